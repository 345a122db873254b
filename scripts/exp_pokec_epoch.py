"""C5 end to end: one evaluation pass over a Pokec-shaped graph in mini-batches, as node classification/main-batch.py:121-137
does it (random node permutation -> batches of 100,000 nodes -> induced subgraph with relabelling -> forward), with every
step on the GPU: dif_subgraph, CSR build, forward (bf16 storage / fp32 accumulate, BASELINE config C5).
    python scripts/exp_pokec_epoch.py
"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difformer_amd import DIFFormer, ops, graph_utils as gu

dev = torch.device("cuda:0")
N, PAIRS, F_IN, BATCH = 1632803, 15311282, 65, 100000        # Pokec: 1.63 M nodes, 30.6 M directed edges
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randint(0, N, (PAIRS,), generator=g, device=dev)
b = torch.randint(0, N, (PAIRS,), generator=g, device=dev)
edge_index = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
edge_index, _ = gu.remove_self_loops(edge_index)
edge_index, _ = gu.add_self_loops(edge_index, num_nodes=N)     # main-batch.py:97-98
x = torch.randn(N, F_IN, device=dev, generator=g)
for store in (torch.float32, torch.bfloat16):
    torch.manual_seed(0)
    model = DIFFormer(F_IN, 64, 2, num_layers=3, kernel="simple").to(dev).to(store).eval()
    xs = x.to(store)
    be = ops.get_backend()

    def epoch(timed):
        perm = torch.randperm(N, device=dev, generator=g)
        t = dict(subgraph=0.0, csr=0.0, forward=0.0)
        outs = []
        for i in range((N + BATCH - 1) // BATCH):
            idx = perm[i * BATCH:(i + 1) * BATCH]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ei, _ = gu.subgraph(idx, edge_index, num_nodes=N, relabel_nodes=True)      # main-batch.py:131
            torch.cuda.synchronize(); t1 = time.perf_counter()
            ops.csr_cache.get(ei, None, idx.numel(), 64 * xs.element_size())
            torch.cuda.synchronize(); t2 = time.perf_counter()
            with torch.no_grad():
                outs.append(model(xs[idx], ei))
            torch.cuda.synchronize(); t3 = time.perf_counter()
            t["subgraph"] += t1 - t0; t["csr"] += t2 - t1; t["forward"] += t3 - t2
        return t, len(outs)

    epoch(False)
    t, nb = epoch(True)
    tot = sum(t.values())
    print(f"{str(store):16s} {nb} batches of {BATCH}: {tot * 1e3:7.1f} ms per pass over {N} nodes "
          f"({N / tot / 1e6:.1f} M nodes/s)  per batch: subgraph {t['subgraph'] / nb * 1e3:.2f} ms, "
          f"CSR build {t['csr'] / nb * 1e3:.2f} ms, forward {t['forward'] / nb * 1e3:.2f} ms", flush=True)

    # round 2: every batch's subgraph from ONE pass over the edge list (dif_subgraph_batches_*)
    def epoch_batched():
        perm = torch.randperm(N, device=dev, generator=g)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        batches = gu.subgraph_batches(perm, BATCH, edge_index, None, num_nodes=N)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        outs = []
        with torch.no_grad():
            for i, (ei, _) in enumerate(batches):
                outs.append(model(xs[perm[i * BATCH:(i + 1) * BATCH]], ei))
        torch.cuda.synchronize(); t2 = time.perf_counter()
        return t1 - t0, t2 - t1, len(outs)

    epoch_batched()
    tg, tf, nb = epoch_batched()
    print(f"{str(store):16s} batched: all {nb} subgraphs {tg * 1e3:.2f} ms + CSR builds and forwards {tf * 1e3:.2f} ms = "
          f"{(tg + tf) * 1e3:.1f} ms per pass ({N / (tg + tf) / 1e6:.1f} M nodes/s)", flush=True)

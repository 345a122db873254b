"""Training epoch of node classification/main-batch.py:121-142 at the scripts' configurations (run.sh:36-44), every step on the GPU:
    random permutation -> batches -> induced subgraph with relabelling (graph_utils.subgraph = dif_subgraph) -> forward -> loss on the
    batch's training nodes -> backward -> Adam
  ogbn-proteins  132,534 nodes, 79.3 M directed entries, 8 features, 112 labels (BCE), hidden 64, 3 layers, batches of 10,000
  pokec          1,632,803 nodes, 30.6 M directed entries + loops, 65 features, 2 classes (NLL), hidden 128, 3 layers, batches of 100,000
Synthetic graphs of those sizes.    python scripts/nc_batch_epoch.py [proteins|pokec]"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from difformer_amd import DIFFormer, graph_utils as gu  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1:] or ["proteins", "pokec"]
CFG = {"proteins": (132534, 39561252, 8, 112, 64, 10000, "bce"), "pokec": (1632803, 15311282, 65, 2, 128, 100000, "nll")}
for name in which:
    N, PAIRS, F_IN, C, HIDDEN, BATCH, LOSS = CFG[name]
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randint(0, N, (PAIRS,), generator=g, device=dev)
    b = torch.randint(0, N, (PAIRS,), generator=g, device=dev)
    edge_index = torch.stack([torch.cat([a, b]), torch.cat([b, a])])
    del a, b
    edge_index, _ = gu.remove_self_loops(edge_index)
    edge_index, _ = gu.add_self_loops(edge_index, num_nodes=N)     # main-batch.py:97-98
    x = torch.randn(N, F_IN, device=dev, generator=g)
    y = (torch.rand(N, C, device=dev, generator=g) > 0.5).float() if LOSS == "bce" else torch.randint(0, C, (N,), device=dev, generator=g)
    train_mask = torch.rand(N, device=dev, generator=g) < 0.5
    torch.manual_seed(0)
    model = DIFFormer(F_IN, HIDDEN, C, num_layers=3, num_heads=1, kernel="simple", use_graph=True, use_bn=True, use_residual=True,
                      use_weight=True, dropout=0.0).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=0.0)

    def epoch():
        model.train()
        perm = torch.randperm(N, device=dev, generator=g)
        t = dict(subgraph=0.0, step=0.0)
        nb = (N + BATCH - 1) // BATCH
        for i in range(nb):
            idx = perm[i * BATCH:(i + 1) * BATCH]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ei, _ = gu.subgraph(idx, edge_index, num_nodes=N, relabel_nodes=True)      # main-batch.py:131
            torch.cuda.synchronize(); t1 = time.perf_counter()
            opt.zero_grad()
            out = model(x[idx], ei)
            m = train_mask[idx]
            if LOSS == "bce":
                loss = F.binary_cross_entropy_with_logits(out[m], y[idx][m])
            else:
                loss = F.nll_loss(F.log_softmax(out, dim=1)[m], y[idx][m])
            loss.backward()
            opt.step()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            t["subgraph"] += t1 - t0; t["step"] += t2 - t1
        return t, nb, int(ei.shape[1])

    epoch()
    t, nb, e_last = epoch()
    tot = t["subgraph"] + t["step"]
    print(f"{name}: {nb} batches of {BATCH} (last subgraph {e_last} entries): epoch {tot * 1e3:.1f} ms = subgraph {t['subgraph'] / nb * 1e3:.2f} ms + "
          f"training step {t['step'] / nb * 1e3:.2f} ms per batch", flush=True)
    model.eval()
    with torch.no_grad():
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            model(x, edge_index)
            torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{name}: full-graph evaluation forward (eval.py:40-43) {1e3 * (t1 - t0):.2f} ms", flush=True)
    del edge_index, x, model
    torch.cuda.empty_cache()

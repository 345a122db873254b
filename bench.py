#!/usr/bin/env python
"""bench.py -- DIFFormer-layer forward throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]           (N=1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N>1)

Workload (BASELINE.json configs[3], the config the north_star target is quoted on; it fits one GPU):
  ogbn-proteins-shaped synthetic graph: N=132,534 nodes, 39,561,252 undirected pairs stored in both
  directions (as OGB stores them) + N self-loops (main.py:73-76) = 79,255,038 CSR entries; F_in=8,
  C=112; DIFFormer-s: hidden 64, H=1, 4 layers, kernel='simple', use_graph, use_weight, use_bn,
  use_residual; fp32; random-init weights (seed 123), x ~ N(0,1); eval mode, no_grad.
A "step" is one DIFFormer.forward over the whole graph with the CSR already cached (warm); the cold
CSR build is timed separately and reported in `cold_csr_build_ms`.  With N>1 GPUs the node rows are
sharded (strong scaling: total work fixed): one all-reduce of the 4,226-float reduce record and one
all-gather of the value rows per layer over RCCL.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel -- the C-ABI entry point with the largest
share of the forward (the gcn_conv product on the headline workload) -- and is FLAT: `frac` is always SURVEY.md section
8(d)'s quantity, algorithmic bytes per launch = 8*nnz + 4*(N+1) + 2*n_rows*H*D*4 over its mean duration from HIP events
recorded on the launching stream in a short pass right after the timed region (the timed region holds nothing but the K
steps), over 8 TB/s (`bound: "hbm"`); the sigmoid kernel is priced against the fp32 MFMA peak (`bound: "mfma"`).  For the
feature-sliced product the keys `limiter: "lds"`, `lds_achieved_tbs`, `lds_peak_tbs`, `lds_frac` say what actually stops it
(its LDS floor is above the HBM floor, profiles/r04_experiments.md).  `ms_per_step_exact_fp32`: a short second pass with
every product on the fp32 MFMA (`config.exact_fp32` says which mode `value` was measured in: false = the default, two
products on split-bfloat16 operands, ~4e-6).  `cpu_baseline` times the whole forward of the oracle on the host cores in both of its
restatements (numpy + OpenMP C; the same lines as CPU torch operations), 1 warm-up + 3 runs, median; `value` is the faster
one, both are printed beside the reference file's own figure from the build container (rank 0, N=1 only).  `whole_forward_frac`:
SURVEY 8(d)'s whole-forward bytes over `ms_per_step` over 8 TB/s; `roofline.share_of_forward`: the dominant entry point bracketed
ALONE over the un-instrumented forward.  `--workload cifar15k-a-h300` / `stl13k-a-h400`: the image-and-text DIFFormer-a lines
(sigmoid attention at 300 / 400 columns, `bound: "mfma"`).  N > 1: when the hidden width splits into 16-byte slices
over the ranks both shard products (rows / feature slices, difformer_amd/dist.py) are timed with K steps each; `value` is the
faster one, named in `config.parallelism`, both in `ms_per_step_by_shard_product`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.0   # fp32-input MFMA, dense (same guide)
MFMA_BF16_PEAK_TFLOPS = 2500.0 # bf16 MFMA, dense (same guide; AMD's 5 PF headline includes 2:1 sparsity)
LDS_PEAK_TBS = 256 * 256 * 2.4e9 / 1e12   # 256 CUs x 256 B/clk (ds_read_b128 / b64) at 2.4 GHz = 157 TB/s (same guide, LDS table)
LDS_MEASURED_TBS = 150.0                   # the same guide's MEASURED aggregate LDS read rate

WORKLOADS = {
    # name: (N, undirected pairs, F_in, classes, hidden, layers, kernel, use_graph)
    "ogbn-proteins-s": (132534, 39561252, 8, 112, 64, 4, "simple", True),
    "cora-s": (2708, 5278, 1433, 7, 64, 2, "simple", True),
    "cora-a": (2708, 5278, 1433, 7, 64, 2, "sigmoid", True),
    "cifar50k-s": (50000, 0, 512, 10, 64, 4, "simple", False),
    # C5: one Pokec-shaped mini-batch (main-batch.py path), bf16 storage / fp32 accumulate; 8 GPUs = 8 replicas
    "pokec-batch-s-bf16": (100000, 115000, 65, 2, 64, 3, "simple", True),
    "pokec-batch-s": (100000, 115000, 65, 2, 64, 3, "simple", True),            # the same batch in fp32 (closed-form layers)
    # not a BASELINE config: the C4 graph in bf16 storage, to show what the gather-bound SpMM does at half the bytes
    "ogbn-proteins-s-bf16": (132534, 39561252, 8, 112, 64, 4, "simple", True),
    # the C4 graph with a skewed (Zipf-like) degree profile, max / mean degree ~13 as in the real
    # ogbn-proteins (7750 / 597), to exercise the SpMM's load balance; not the headline line
    "ogbn-proteins-zipf-s": (132534, 39561252, 8, 112, 64, 4, "simple", True),
    # the C4 graph with COMMUNITY structure (the real ogbn-proteins: 8 species that interact almost only among themselves,
    # node ids grouped by species): 8 contiguous blocks, 95 % of the edges inside a block; the model runs it in a mixed
    # node order (ops.MixedGraph); not the headline line
    "ogbn-proteins-blocks-s": (132534, 39561252, 8, 112, 64, 4, "simple", True),
    # ... and with BOTH: Zipf degrees inside 8 communities (mixing + hub splitting together); not the headline line
    "ogbn-proteins-zipf-blocks-s": (132534, 39561252, 8, 112, 64, 4, "simple", True),
    # the full-graph Pokec forward of the evaluation path (node classification/eval.py:40-43, main-batch.py:144-145: after
    # mini-batch training the model runs ONCE over the whole graph): N = 1,632,803, 15.4 M undirected pairs + loops =
    # 32.4 M entries (mean degree ~19: gather kernel), F_in = 65, C = 2, 3 layers, hidden 64 / 128 (run.sh:42-44)
    "pokec-full-s": (1632803, 15400000, 65, 2, 64, 3, "simple", True),
    "pokec-full-h128": (1632803, 15400000, 65, 2, 128, 3, "simple", True),
    # the widths the reference's scripts train with (node classification/run.sh:42-44 hidden 128 on Pokec batches;
    # image and text/run.sh:27 hidden 300): hidden 128 takes the operator path (projections on the vendor GEMM, stand-alone
    # reduce / wide apply kernels), hidden 300 the closed form on library GEMMs around the hand-written Gram / tail passes
    "pokec-batch-h128": (100000, 115000, 65, 2, 128, 3, "simple", True),
    "cifar50k-h300": (50000, 0, 512, 10, 300, 4, "simple", False),
    # the DIFFormer-a lines of the image-and-text scripts (image and text/run.sh:35: cifar10, 15,000 instances -- dataset.py:168 --,
    # --kernel sigmoid --hidden_channels 300 --num_layers 2 --use_residual --use_bn, no --use_graph, no --use_weight): the
    # O(N^2) attention at 300 columns per head, bound by the matrix pipe (csrc/sigmoid_wide.hip)
    "cifar15k-a-h300": (15000, 0, 512, 10, 300, 2, "sigmoid", False, {"use_weight": False}),
    "stl13k-a-h400": (13000, 0, 512, 10, 400, 2, "sigmoid", False, {"use_weight": False}),           # run.sh:17 (stl10, hidden 400)
}


def make_graph(n, pairs, dev, zipf=False, blocks=0):
    g = torch.Generator(device=dev).manual_seed(0)
    if blocks and zipf:
        # both properties of the real ogbn-proteins at once: `blocks` contiguous groups of nodes (species) with 95 % of the
        # pairs inside the group of their first endpoint, AND skewed degrees (endpoint probability ~ (rank + 1100)^-0.75,
        # ranks scattered over the ids: max / mean degree ~ 13) inside every group
        size = -(-n // blocks)
        rank_of = torch.randperm(n, generator=g, device=dev)
        w = (rank_of.to(torch.float64) + 1100.0) ** -0.75                       # weight of node id v
        cdf = torch.cumsum(w, 0)
        draw = lambda lo, hi: torch.searchsorted(cdf, lo + torch.rand(pairs, generator=g, device=dev, dtype=torch.float64)
                                                 * (hi - lo)).clamp_(max=n - 1)
        zero = torch.zeros(pairs, device=dev, dtype=torch.float64)
        a = draw(zero, cdf[-1].expand(pairs))
        first = (a // size) * size
        last = (first + size).clamp_(max=n) - 1
        lo = torch.where(first > 0, cdf[(first - 1).clamp_(min=0)], zero)
        inside = torch.rand(pairs, generator=g, device=dev) < 0.95
        b = torch.where(inside, draw(lo, cdf[last]), draw(zero, cdf[-1].expand(pairs)))
    elif blocks:
        # `blocks` contiguous groups of nodes; 95 % of the pairs stay inside the group of their first endpoint
        size = -(-n // blocks)
        a = torch.randint(0, n, (pairs,), generator=g, device=dev)
        inside = torch.rand(pairs, generator=g, device=dev) < 0.95
        b_in = ((a // size) * size + torch.randint(0, size, (pairs,), generator=g, device=dev)).clamp_(max=n - 1)
        b = torch.where(inside, b_in, torch.randint(0, n, (pairs,), generator=g, device=dev))
    elif zipf:
        # endpoint probability ~ (rank + 1100)^-0.75 over randomly permuted node ids: max / mean degree ~ 13
        w = (torch.arange(n, device=dev, dtype=torch.float64) + 1100.0) ** -0.75
        cdf = torch.cumsum(w, 0) / w.sum()
        perm = torch.randperm(n, generator=g, device=dev)
        a, b = (perm[torch.searchsorted(cdf, torch.rand(pairs, generator=g, device=dev, dtype=torch.float64)).clamp_(max=n - 1)]
                for _ in range(2))
    else:
        a = torch.randint(0, n, (pairs,), generator=g, device=dev)
        b = torch.randint(0, n, (pairs,), generator=g, device=dev)
    loops = torch.arange(n, device=dev)
    return torch.stack([torch.cat([a, b, loops]), torch.cat([b, a, loops])]).contiguous()


def cpu_baseline(model, x, edge_index, cfg, repeats=3):
    """The oracle timed on the host cores: the WHOLE forward on the full graph, one warm-up + `repeats`
    timed runs, median -- in BOTH restatements the oracle ships: the numpy + OpenMP C port (oracle/difformer_oracle.py) and
    the same lines as CPU torch operations (oracle/difformer_oracle_grad.py under no_grad: what the reference file itself
    executes, minus torch_sparse).  `value` is the FASTER of the two (the numpy einsums are not BLAS calls: on the small
    configs the torch restatement is ~10x faster, on the C4 graph the OpenMP product wins); both are printed."""
    from oracle import difformer_oracle as orc
    from oracle import difformer_oracle_grad as og
    cores = os.cpu_count() or 1
    os.environ["ORACLE_THREADS"] = str(cores)
    torch.set_num_threads(cores)
    n = x.shape[0]

    def timed(fn):
        fn()
        times = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t0)
        return float(np.median(times)), times

    runs = {}
    dense_pairs = cfg["kernel"] == "sigmoid" and n > 4000        # np.einsum("nhm,lhm->nlh") is a scalar loop: minutes at N = 15,000
    if not dense_pairs:
        p = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        xh = x.cpu().numpy()
        ei = None if edge_index is None else edge_index.cpu().numpy()
        runs["numpy+openmp"] = timed(lambda: orc.difformer_forward(p, xh, ei, None, cfg))
    big_graph = edge_index is not None and edge_index.shape[1] > 2_000_000      # index_add_ over 79 M entries: tens of seconds per layer
    threads = {}
    if not big_graph:
        pt = {k: v.detach().cpu().float() for k, v in model.state_dict().items()}
        xt = x.cpu().float()
        eit = None if edge_index is None else edge_index.cpu()
        def torch_port():
            with torch.no_grad():
                og.difformer_forward(pt, xt, eit, None, cfg)
        # (small operands on a 256-thread host: the intra-op pool at full width is slower than 8 threads -- time both)
        for nt in sorted({cores, min(cores, 8)}, reverse=True):
            torch.set_num_threads(nt)
            name = "torch-ops" if nt == cores else f"torch-ops@{nt}threads"
            runs[name] = timed(torch_port)
            threads[name] = nt
        torch.set_num_threads(cores)
    best = min(runs, key=lambda k: runs[k][0])
    med, times = runs[best]
    return {"value": n / med, "unit": "nodes/s", "cores": threads.get(best, cores), "kind": "port", "restatement": best,
            "by_restatement_nodes_per_s": {k: n / v[0] for k, v in runs.items()},
            "sample": f"whole {cfg['num_layers']}-layer forward of the oracle ({best}) on the full graph, 1 warm-up + {repeats} timed "
                      f"runs: median {med:.3f} s (min {min(times):.3f}, max {max(times):.3f})"}


def shard_plan(workload, world, rank):
    """What rank `rank` of `world` processes: C5 mini-batches are independent -> every GPU runs its own batch (replicas,
    weak scaling, no collective); every other workload is ONE graph whose node rows are split into contiguous blocks
    (strong scaling; per layer one all-reduce of the 4,226-float record and one all-gather of the rows)."""
    from difformer_amd.dist import split_rows
    n = WORKLOADS[workload][0]
    replicas = world > 1 and workload.startswith("pokec-batch")
    if world == 1 or replicas:
        begin, count = 0, n
    else:
        counts = split_rows(n, world)
        begin, count = sum(counts[:rank]), counts[rank]
    return {"replicas": replicas, "row_begin": begin, "n_local": count,
            "scaling": "weak" if workload.startswith("pokec-batch") else "strong",       # what adding GPUs does to this workload
            "parallelism": (f"replicas x{world}" if replicas else f"row-shard x{world}") if world > 1 else "single GPU"}


def max_over_ranks(seconds, device):
    """The job is as slow as its slowest rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_value(n, steps, elapsed, world, replicas):
    """Whole-job nodes per second: replicas each push their own n nodes per step, a sharded graph is one set of n."""
    return n * steps / elapsed * (world if replicas else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the clocks need ~30 ms of load after an idle period to settle (the first 20 forwards after a pause run ~5 %
    # slower: scripts/exp_host_time.py); 130 forwards are 0.2 s
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="ogbn-proteins-s", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole forward as one hipGraph (single GPU); per-kernel events then come from a "
                         "short eager pass after the timed region instead of from the timed region itself")
    ap.add_argument("--per-kernel", action="store_true", help="also print mean ms per C-ABI entry point (stderr)")
    ap.add_argument("--shard-product", choices=["auto", "row", "slice"], default=os.environ.get("DIFFORMER_SHARD_PRODUCT", "auto"),
                    help="N > 1: how closed-form layers split the aggregation -- by destination rows (all-gather of the "
                         "layer input) or by feature slices (two all-to-alls of 1/N of it; difformer_amd/dist.py).  auto "
                         "(default): when the hidden width splits into 16-byte slices over the ranks (C %% (4 N) == 0) BOTH are "
                         "timed back to back with K steps each and the faster one is the reported value (DESIGN.md section 4 "
                         "predicts the slice shard: 7.4 MB against 30 MB per rank and layer at C4 on 8 ranks); else rows")
    ap.add_argument("--no-exact-pass", action="store_true",
                    help="skip the short second pass with every product on the fp32 MFMA (ms_per_step_exact_fp32)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # Dry run of the N > 1 path on a one-GPU box (tests/test_gpu_dist.py): every rank on cuda:0, collectives over gloo
    # (RCCL refuses two ranks per device).  Exercises the code the driver launches; its timings mean nothing.
    one_gpu = os.environ.get("DIFFORMER_BENCH_ONE_GPU", "0") == "1"
    dev = torch.device("cuda", 0 if one_gpu else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from difformer_amd import DIFFormer, GraphedForward, RowShard, ops

    n, pairs, f_in, classes, hidden, layers, kernel, use_graph = WORKLOADS[args.workload][:8]
    flags = WORKLOADS[args.workload][8] if len(WORKLOADS[args.workload]) > 8 else {}
    use_weight = flags.get("use_weight", True)
    torch.manual_seed(123)
    model = DIFFormer(f_in, hidden, classes, num_layers=layers, num_heads=1, kernel=kernel, use_graph=use_graph, use_weight=use_weight)
    model.reset_parameters()
    model = model.to(dev).eval()
    store = torch.bfloat16 if args.workload.endswith("-bf16") else torch.float32
    model = model.to(store)
    gx = torch.Generator(device=dev).manual_seed(1)
    x_full = torch.randn(n, f_in, generator=gx, device=dev).to(store)
    edge_index = make_graph(n, pairs, dev, zipf="-zipf" in args.workload,
                            blocks=8 if "-blocks" in args.workload else 0) if use_graph else None
    nnz = 0 if edge_index is None else int(edge_index.shape[1])

    # C5: mini-batches are independent -> every GPU runs its own batch, no collective (replicas)
    plan = shard_plan(args.workload, world, rank)
    replicas = plan["replicas"]
    shard = RowShard.from_process_group(n) if (world > 1 and not replicas) else None
    assert shard is None or (shard.row_begin, shard.n_local) == (plan["row_begin"], plan["n_local"])
    x = x_full if shard is None else shard.local_rows(x_full).contiguous()
    slice_ok = (shard is not None and kernel == "simple" and hidden <= 64 and hidden % (4 * world) == 0 and store == torch.float32)
    products = [None]
    if shard is not None:
        if args.shard_product == "auto":
            products = ["slice", "row"] if slice_ok else ["row"]
        else:
            products = [args.shard_product]
        shard.product = products[0]
        model.set_row_shard(shard)
    n_local = x.shape[0]

    be = ops.get_backend()
    cold_ms = None
    with torch.no_grad():
        if use_graph:  # cold: CSR build (degree, values, stable sort), once per graph
            esz = 2 if store == torch.bfloat16 else 4
            def cold_build():      # CSR + (dense unweighted fp32 graphs, one GPU) the feature-sliced LDS format
                csr = ops.csr_cache.get(edge_index, None, n, hidden * esz, shard, esz)
                if shard is None and esz == 4:
                    csr.sliced(0, n, hidden)
            cold_build()           # first build also pays hipMalloc for the workspaces
            ops.csr_cache.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cold_build()           # timed: what every new graph / mini-batch costs
            torch.cuda.synchronize()
            cold_ms = (time.perf_counter() - t0) * 1e3
        use_graph_replay = (world == 1) and args.graph

        def timed_region():
            """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides -> (seconds, step)."""
            for _ in range(args.warmup):
                model(x, edge_index)
            # --graph: the whole forward is captured once and replayed as one hipGraph launch per step
            step = GraphedForward(model, x, edge_index) if use_graph_replay else (lambda: model(x, edge_index))
            if use_graph_replay:
                for _ in range(2):
                    step()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            return time.perf_counter() - t0, step

        by_product = {}
        for prod in products:                  # N > 1 with a width that splits into slices: both shard products, K steps each
            if prod is not None:
                shard.product = prod
            sec, step = timed_region()
            by_product[prod] = max_over_ranks(sec, dev)
        best_product = min(by_product, key=by_product.get)
        if best_product is not products[-1]:   # leave the model in the faster configuration for the passes below
            shard.product = best_product
            for _ in range(2):
                model(x, edge_index)
        elapsed = by_product[best_product]
        # Separate short pass right after the timed region (nothing but the K steps sits inside it): per-forward HIP
        # events on the launching stream -> median / spread, and per-entry-point events -> the dominant kernel's mean
        # launch duration for the roofline.
        fwd_events = []
        for _ in range(min(args.steps, 10)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(torch.cuda.current_stream(dev))
            step()
            b.record(torch.cuda.current_stream(dev))
            fwd_events.append((a, b))
        be.kernel_events = {}
        for _ in range(min(args.steps, 5)):
            model(x, edge_index)
        torch.cuda.synchronize()
        fwd_ms = sorted(a.elapsed_time(b) for a, b in fwd_events)
    ag = getattr(model, "_ag_state", None)
    if use_graph_replay:
        launch_mode = "hipGraph replay"
    elif ag is not None and ag[2] is not None:
        # plain model(x, edge_index) calls: DIFFormer.forward captured itself as one hipGraph on the third identical call
        # (warm-up) and the timed steps replayed it; DIFFORMER_AUTO_GRAPH=0 gives the kernel-by-kernel launches
        launch_mode = "model(x, edge_index): auto-captured hipGraph replay"
    else:
        launch_mode = "eager"
    ktimes = be.kernel_times_ms()
    be.kernel_events = None
    # N > 1: where a rank's forward goes -- exposed collective waits (events on the compute stream around every exchange
    # step, difformer_amd/dist.py) beside the kernels by group -- so that the first run on real xGMI links is diagnostic
    per_rank = None
    if shard is not None:
        shard.timeline = []
        be.kernel_events = {}
        reps = min(args.steps, 3)
        with torch.no_grad():
            ta, tb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ta.record(torch.cuda.current_stream(dev))
            for _ in range(reps):
                model(x, edge_index)
            tb.record(torch.cuda.current_stream(dev))
        kt = be.kernel_times_ms()
        be.kernel_events = None
        coll = {k: v / reps for k, v in shard.timeline_ms().items()}
        shard.timeline = None
        group_of = lambda k: ("product" if ("spmm" in k or "prescale" in k) else "layer" if "simple_layer" in k or "layer_tail" in k
                              else "record" if ("gram" in k or "coeffs" in k or "reduce" in k) else "other")
        kern = {}
        for k, v in kt.items():
            kern[group_of(k)] = kern.get(group_of(k), 0.0) + float(np.sum(v)) / reps
        mine = {"rank": rank, "rows": int(n_local), "forward_ms": ta.elapsed_time(tb) / reps, "collectives_ms": coll, "kernels_ms": kern,
                "note": "instrumented eager forwards (an event pair around every call): read the SPLIT, not the total"}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
    def dominant_alone(entry):
        # Second event pass with ONLY the dominant entry point bracketed: two events around every C-ABI call cost the
        # host enough that, on a slow host, the GPU queue runs dry in the pass above and the brackets then include the
        # wait for the launch (0.375 ms against rocprofv3's 0.333 ms for the same kernel on one box).  With one bracket
        # per layer the forward stays GPU-bound and the bracket is the kernel.
        be.kernel_events, be.kernel_events_only = {}, {entry}
        with torch.no_grad():
            for i in range(2 + min(args.steps, 5)):
                if i == 2:
                    be.kernel_events = {}          # the first forwards refill the queue; their brackets are dropped
                model(x, edge_index)
        got = be.kernel_times_ms().get(entry)
        be.kernel_events, be.kernel_events_only = None, None
        return got

    ms_per_step = elapsed / args.steps * 1e3
    value = job_value(n, args.steps, elapsed, world, replicas)

    # The default f32 forward runs two products on split-bfloat16 operands (hi + lo parts, three bf16 MFMAs, ~4e-6 of the
    # float64 result: the last layer's output Linear, long-row input Linears, the hidden 65-416 layer kernels).  A short second
    # pass with EVERY product on the fp32 MFMA (DIFFORMER_EXACT_FP32=1 / ops.set_exact_fp32) says what that buys.
    ms_exact = None
    if not args.no_exact_pass and store == torch.float32:
        was = ops.set_exact_fp32(True)
        model.invalidate_caches()
        reps = max(5, min(args.steps, 20))
        with torch.no_grad():
            for _ in range(5):
                model(x, edge_index)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                model(x, edge_index)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            ms_exact = max_over_ranks(time.perf_counter() - t0, dev) / reps * 1e3
        ops.set_exact_fp32(was)
        model.invalidate_caches()

    # roofline of the dominant kernel on this rank: the C-ABI entry point with the largest share of the forward (event
    # brackets of the eager pass above), priced with the algorithmic bytes / FLOP of SURVEY.md section 8d
    esz = 2 if store == torch.bfloat16 else 4
    share = {k: float(np.sum(v)) for k, v in ktimes.items() if v and not k.startswith(("dif_csr", "dif_sliced_measure", "dif_sliced_emit",
                                                                                       "dif_row_order", "dif_linear_pack"))}
    gcn_bytes = 8.0 * nnz * (n_local / n) + 4.0 * (n + 1) + 2.0 * n_local * hidden * esz
    # (entry point, kernel name, key into the tracked rocprofv3 / PMC files, bound, algorithmic bytes, algorithmic FLOP)
    table = {
        "dif_sliced_spmm_f32": ("sliced_spmm_kernel (gcn_conv)", "sliced_spmm_kernel", "lds", gcn_bytes, None),
        "dif_gcn_spmm_f32": ("spmm_* (gcn_conv, gather kernels)", "spmm_", "hbm", gcn_bytes, None),
        "dif_simple_layer_f32": (("simple_layer_kernel<GATHER> (closed-form layer with gcn_conv's aggregation inside)", "simple_layer_kernel",
                                  "hbm", gcn_bytes, None) if (use_graph and not share.get("dif_gcn_spmm_f32") and not share.get("dif_sliced_spmm_f32"))
                                 else ("simple_layer_kernel (closed-form simple layer)", "simple_layer_kernel", "hbm",
                                       (3.0 if use_graph else 2.0) * n_local * hidden * esz, None)),
        "dif_sigmoid_attn_f32": (("sigw_fwd_kernel (sigmoid attention, heads of 65..512 columns: split-bfloat16 plane sweep) + its pack / combine launches",
                                  "sigw_fwd_kernel") if (hidden > 64 and store == torch.float32 and not ops.EXACT_FP32)
                                 else ("sigmoid_attn_kernel", "sigmoid_attn_kernel")) + ("mfma", 4.0 * n_local * hidden * 4,
                                                                                         4.0 * n_local * n * hidden),
        "dif_gram_sym_f32": ("simple_reduce_kernel<sym> (Gram record of the wide closed form)", "simple_reduce_kernel", "mfma",
                             1.0 * n_local * hidden * 4, 1.0 * n_local * hidden * (hidden + 1)),
        "dif_linear_f32": ("linear kernels (input MLP / output Linear)", "linear_", "hbm", 1.0 * n_local * (f_in + hidden) * esz, None),
        "dif_input_gram_f32": ("input_gram_kernel (input layer + Gram record + slice-major copy)", "input_gram_kernel", "hbm",
                               1.0 * n_local * (f_in + 2 * hidden) * 4, None),
        "dif_simple_apply_f32": ("simple_apply_kernel", "simple_apply_kernel", "hbm", 2.0 * n_local * hidden * esz, None),
        "dif_gram_f32": ("gram_kernel (Gram record of the layer input)", "gram_kernel", "hbm", 1.0 * n_local * hidden * esz, None),
    }
    ranked = sorted((k for k in share if k in table), key=lambda k: -share[k])
    dom = ranked[0] if ranked else None
    dom_name, dom_key, bound, alg_bytes, alg_flop = table[dom] if dom else ("none", "", "hbm", 0.0, None)
    alone = dominant_alone(dom) if dom else None   # unconditional: every rank runs the same number of forwards (collectives inside)
    dom_ms = float(np.mean(alone)) if alone else (float(np.mean(ktimes[dom])) if dom and ktimes.get(dom) else None)
    hbm_gbs = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms else None
    # HBM-side bytes per launch of the dominant kernel come from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
    # correction + WRITE_SIZE, calibrated; scripts/pmc_traffic.sh) stored under profiles/ -- a counter run cannot
    # share a process with this timed run.  Only valid for the single-GPU workload it was collected on.
    traffic = tsrc = None
    for tfile in ("r06_pmc_traffic_c4.json", "r06_pmc_traffic_pokec_full.json", "r05_pmc_traffic_c4.json", "r04_pmc_traffic_c4.json",
                  "r03_pmc_traffic_c4.json", "r02_pmc_traffic_c4.json", "r01_pmc_traffic_c4.json"):
        tpath = os.path.join(ROOT, "profiles", tfile)
        if world == 1 and use_graph and os.path.exists(tpath):
            tj = json.load(open(tpath))
            # the key is a kernel name or the common prefix of a kernel family (`spmm_`: the heaviest member counts)
            hits = [v for k, v in tj.get("kernels", {}).items() if dom_key and k.startswith(dom_key)]
            if tj.get("workload") == args.workload and hits:
                traffic = max(h.get("hbm_bytes_per_launch", 0.0) for h in hits)
                tsrc = f"profiles/{tfile} (rocprofv3 PMC, separate passes)"
                break
    src = "HIP events on the launching stream, only this entry point bracketed (bench.py::dominant_alone)"
    # FLAT record (the driver's `parsed` keeps no nested dicts).  `frac` is ALWAYS the SURVEY 8(d) quantity of the dominant
    # kernel: algorithmic bytes / launch time / 8 TB/s for the HBM-bound rows (a1, a3, a4/a5), algorithmic FLOP / launch time
    # / 157 TFLOP/s for a2.  What actually limits the sliced product (LDS array + vector-ALU issue, profiles/r04_experiments.md
    # section 1) rides beside it in the lds_* siblings and in `limiter`.
    # share of the forward: the dominant entry point's launches of ONE forward (bracketed alone, so the queue stays full) over the
    # un-instrumented forward (median of the per-forward event pairs) -- the all-entry-point pass above inflates the small kernels
    n_alone_fwd = min(args.steps, 5)
    share_alone = (float(np.sum(alone)) / n_alone_fwd / fwd_ms[len(fwd_ms) // 2]) if (alone and fwd_ms) else None
    roofline = {"bound": "mfma" if bound == "mfma" else "hbm", "kernel": dom_name, "entry_point": dom,
                "share_of_forward": share_alone if share_alone is not None else ((share[dom] / sum(share.values())) if dom else None),
                "share_of_forward_source": "dominant entry point bracketed alone / median un-instrumented forward",
                "traffic": traffic, "traffic_source": tsrc, "avg_launch_ms": dom_ms, "avg_launch_ms_source": src}
    if bound == "mfma" and dom_ms:
        tf = alg_flop / (dom_ms * 1e-3) / 1e12
        roofline.update({"achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                         "algorithmic_flop_per_launch": alg_flop, "algorithmic_bytes_per_launch": alg_bytes,
                         "hbm_achieved_gbs": hbm_gbs, "hbm_frac": (hbm_gbs / HBM_PEAK_GBS) if hbm_gbs else None})
        if dom == "dif_sigmoid_attn_f32" and hidden > 64 and store == torch.float32 and not ops.EXACT_FP32:
            roofline.update({"mfma_bf16_peak_tflops": MFMA_BF16_PEAK_TFLOPS, "products_per_fp32_product": 3,
                             "frac_of_split_bf16_peak": tf / (MFMA_BF16_PEAK_TFLOPS / 3.0),
                             "limiter_note": "every operand as two bfloat16 planes (hi + lo), three v_mfma_f32_16x16x32_bf16 per 32-deep "
                                             "step: `frac` is algorithmic fp32 FLOP (4 N L D) over the fp32 MFMA peak and may exceed 1; "
                                             "frac_of_split_bf16_peak prices the same FLOP against the dense bf16 peak / 3.  avg_launch_ms "
                                             "brackets the ENTRY POINT: three pack launches + the sweep + the split combine "
                                             "(avg_launch_ms_rocprofv3 = the sweep kernel alone)"})
        elif dom == "dif_sigmoid_attn_f32" and hidden <= 64 and store == torch.float32 and not ops.EXACT_FP32:
            roofline["limiter_note"] = ("heads of <= 64 channels contract on split-bfloat16 operands: 3 v_mfma_f32_16x16x32_bf16 per "
                                        "32-deep step; frac stays algorithmic fp32 FLOP over the fp32 MFMA peak (the contract's figure)")
    else:
        roofline.update({"achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (hbm_gbs / HBM_PEAK_GBS) if hbm_gbs else None, "algorithmic_bytes_per_launch": alg_bytes})
    if traffic and dom_ms and bound != "mfma":
        # what the memory system actually moved per launch over the same time: a gather of neighbour rows that no cache holds
        # (full Pokec: 417 MB of rows against 32 MB of L2) sits at the HBM roofline on ITS traffic while `frac` -- algorithmic bytes,
        # every row once -- stays small; a low traffic_frac would mean latency- or issue-bound instead
        roofline["traffic_gbs"] = traffic / (dom_ms * 1e-3) / 1e9
        roofline["traffic_frac"] = roofline["traffic_gbs"] / HBM_PEAK_GBS
        roofline["traffic_over_algorithmic"] = traffic / alg_bytes if alg_bytes else None
    if bound == "lds" and dom_ms:
        # every entry is one 16-byte LDS read per 16-byte feature slice -- nnz x F x 4 bytes per launch whatever the padding
        lds_bytes = 1.0 * nnz * (n_local / n) * hidden * 4
        lds_tbs = lds_bytes / (dom_ms * 1e-3) / 1e12
        roofline.update({"limiter": "lds", "lds_achieved_tbs": lds_tbs, "lds_peak_tbs": LDS_PEAK_TBS, "lds_frac": lds_tbs / LDS_PEAK_TBS,
                         "lds_frac_of_measured_peak": lds_tbs / LDS_MEASURED_TBS, "lds_bytes_per_launch": lds_bytes,
                         "limiter_note": "one ds_read_b128 per (entry, 16-byte slice); lds_peak_tbs = 256 CUs x 256 B/clk x 2.4 GHz "
                                         "(spec), lds_frac_of_measured_peak against the guide's measured ~150 TB/s; the LDS floor of any "
                                         "fp32 gather-from-LDS design (129-147 us) is above the HBM floor (88 us): frac is the contract's "
                                         "HBM figure, the lds_* keys say why it stops there (profiles/r04_experiments.md section 1)"})
    # the rocprofv3 figure the event bracket is checked against (same workload, tracked summary of this round if present)
    import glob
    cands = [c for r in ("r06", "r05", "r04", "r03") for c in
             sorted(glob.glob(os.path.join(ROOT, "profiles", f"{r}_*{args.workload}*kernel_stats.csv")), reverse=True)]
    for cand in cands:
        try:
            import csv
            rows = [r for r in csv.DictReader(open(cand)) if dom_key and r.get("Name", "").find(dom_key) >= 0]
            if rows:
                best = max(rows, key=lambda r: float(r["TotalDurationNs"]))
                roofline["avg_launch_ms_rocprofv3"] = float(best["AverageNs"]) / 1e6
                roofline["rocprofv3_source"] = "profiles/" + os.path.basename(cand)
                break
        except Exception:
            continue

    # SURVEY 8(d) "whole forward": N F_in s + N C s + L (B_a1 + use_graph B_a3), B_a1 = 4 N H D s, B_a3 = 8 nnz + 4 (N + 1) + 2 N H D s
    whole_bytes = (n_local * f_in * esz + n_local * classes * esz +
                   layers * (4.0 * n_local * hidden * esz + (gcn_bytes if use_graph else 0.0)))
    whole_flop = layers * 4.0 * n_local * n * hidden if kernel == "sigmoid" else None
    whole = {"whole_forward_bytes": whole_bytes, "whole_forward_gbs": whole_bytes / (ms_per_step * 1e-3) / 1e9,
             "whole_forward_frac": whole_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if whole_flop:
        whole.update({"whole_forward_attention_flop": whole_flop, "whole_forward_tflops": whole_flop / (ms_per_step * 1e-3) / 1e12,
                      "whole_forward_mfma_frac": whole_flop / (ms_per_step * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS})

    if args.per_kernel and rank == 0:
        for k, v in sorted(ktimes.items()):
            print(f"[per-kernel] {k}: calls={len(v)} mean={np.mean(v) * 1e3:.1f} us min={np.min(v) * 1e3:.1f} us",
                  file=sys.stderr)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and store == torch.float32:
        cfg = dict(hidden_channels=hidden, num_layers=layers, num_heads=1, kernel=kernel, alpha=0.5, use_bn=True,
                   use_residual=True, use_weight=use_weight, use_graph=use_graph, graph_weight=-1, use_source=False)
        cpu = cpu_baseline(model, x_full, edge_index, cfg)

    if cpu is not None:
        # beside the port timed here: the reference file imported VERBATIM, timed in the build container (the GPU box has no
        # /root/reference); scripts/cpu_reference_verbatim.py, BASELINE.md section 2
        vpath = os.path.join(ROOT, "profiles", "cpu_reference_verbatim.json")
        if os.path.exists(vpath):
            ver = json.load(open(vpath)).get(args.workload)
            if ver:
                cpu["reference_verbatim"] = {k: ver[k] for k in ("value", "unit", "cores", "kind", "where", "sample")}
    if rank == 0:
        print(json.dumps({
            "metric": "DIFFormer-layer forward nodes/sec", "value": value, "unit": "nodes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "ms_per_step_exact_fp32": ms_exact,
            "ms_per_step_events": {"median": fwd_ms[len(fwd_ms) // 2], "min": fwd_ms[0], "max": fwd_ms[-1], "n": len(fwd_ms)},
            "higher_is_better": True,
            "scaling": plan["scaling"], "vs_baseline": None, "dtype": "bf16" if store == torch.bfloat16 else "f32", "data": "synthetic",
            "config": {"workload": args.workload, "nodes": n, "csr_entries": nnz, "in_channels": f_in,
                       "hidden": hidden, "heads": 1, "layers": layers, "kernel": kernel, "use_graph": use_graph,
                       "parallelism": plan["parallelism"] + (f", closed-form aggregation split by {'feature slices (two all-to-alls per layer)' if best_product == 'slice' else 'destination rows (one all-gather per layer)'}" if best_product is not None else ""),
                       "exact_fp32": bool(ops.EXACT_FP32),
                       "csr": "warm (cached); cold build reported in cold_csr_build_ms",
                       "launch": launch_mode},
            "cold_csr_build_ms": cold_ms, **whole, "roofline": roofline, "cpu_baseline": cpu,
            **({"ms_per_step_by_shard_product": {k: v / args.steps * 1e3 for k, v in by_product.items()}} if shard is not None else {}),
            **({"per_rank_phases": per_rank} if per_rank is not None else {}),
        }))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

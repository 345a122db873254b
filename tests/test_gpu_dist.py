"""Two ranks on ONE MI355X: the row-sharded forward and training step on the HIP kernels, with the collectives over
gloo (RCCL refuses two ranks on one device; gloo moves device tensors through the host).  What the rank-by-rank tests
emulate runs here for real: both ranks in flight at once, every collective of dist.py and autograd_ops.py executed, each
rank's rows compared with the single-process run of the same kernels."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, kernel, n, deg, heads, out_q):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from difformer_amd import DIFFormer, RowShard
        dev = torch.device("cuda:0")
        torch.manual_seed(3)
        model = DIFFormer(24, 64, 7, num_layers=2, num_heads=heads, kernel=kernel, dropout=0.0, use_source=True).to(dev)
        g = torch.Generator().manual_seed(9)
        x = torch.randn(n, 24, generator=g).to(dev)
        ei = torch.cat([torch.randint(0, n, (2, deg * n), generator=g), torch.arange(n).repeat(2, 1)], dim=1).to(dev)
        target = torch.randn(n, 7, generator=g).to(dev)
        shard = RowShard.from_process_group(n)
        # ---- inference
        model.eval()
        with torch.no_grad():
            full = model(x, ei)
            model.set_row_shard(shard)
            local = model(shard.local_rows(x).contiguous(), ei)
            model.set_row_shard(None)
        err_f = float((local - shard.local_rows(full)).abs().max() / full.abs().max())
        # ---- one training step: loss summed over nodes; parameter gradients summed over ranks
        model.train()
        xf = x.clone().requires_grad_(True)
        ((model(xf, ei) - target) ** 2).sum().backward()
        ref = [p.grad.clone() for p in model.parameters()]
        ref_x = xf.grad.clone()
        model.zero_grad()
        model.set_row_shard(shard)
        xl = shard.local_rows(x).contiguous().requires_grad_(True)
        ((model(xl, ei) - shard.local_rows(target)) ** 2).sum().backward()
        shard.all_reduce_gradients(model.parameters())
        scale = max(float(r.abs().max()) for r in ref)
        err_p = max(float((p.grad - r).abs().max()) for p, r in zip(model.parameters(), ref)) / scale
        err_x = float((xl.grad - shard.local_rows(ref_x)).abs().max() / ref_x.abs().max())
        torch.cuda.synchronize()
        from difformer_amd import ops
        sliced = [sl for _, _, csr in ops.csr_cache.entries.values() for sl in csr._sliced.values() if sl is not None]
        splits = max([int(ops.get_backend().lib.dif_sliced_spmm_workspace_bytes(n, shard.n_local, 64 * heads) > 0)] if sliced else [0])
        out_q.put((rank, err_f, err_p, err_x, len(sliced), splits))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kernel,n,deg,heads", [("simple", 20000, 60, 1), ("simple", 6000, 8, 2), ("sigmoid", 3000, 6, 1),
                                                ("simple", 18000, 64, 2)])
def test_two_ranks_on_one_gpu_forward_and_training_step(kernel, n, deg, heads):
    """simple / one head / dense graph: closed-form layers with the sliced product of a shard (source splits) in
    inference, the operator kernels and their backward kernels (two all-reduces inside the attention backward, the
    adjoint product over all-gathered gradient rows) in training; several heads and sigmoid: the operator path -- on a
    dense graph with the shard's sliced product there too (forward and adjoint)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kernel, n, deg, heads, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=420) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, err_f, err_p, err_x, n_sliced, splits in sorted(results):
        assert err_f < 1e-5, (rank, err_f)
        assert err_p < 1e-4 and err_x < 1e-4, (rank, err_p, err_x)
        if deg >= 48:                        # the shard ran the feature-sliced product, with its source tiles split
            assert n_sliced >= 1 and splits == 1


def _slice_worker(rank, world, port, n, deg, out_q):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from difformer_amd import DIFFormer, RowShard, ops
        dev = torch.device("cuda:0")
        torch.manual_seed(3)
        model = DIFFormer(24, 64, 7, num_layers=3, num_heads=1, kernel="simple", use_source=True).to(dev).eval()
        g = torch.Generator().manual_seed(9)
        x = torch.randn(n, 24, generator=g).to(dev)
        ei = torch.cat([torch.randint(0, n, (2, deg * n), generator=g), torch.arange(n).repeat(2, 1)], dim=1).to(dev)
        shard = RowShard.from_process_group(n)
        shard.product = "slice"
        be = ops.get_backend()
        with torch.no_grad():
            full = model(x, ei)
            model.set_row_shard(shard)
            be.kernel_events = {}
            local = model(shard.local_rows(x).contiguous(), ei)
            launched = set(be.kernel_events)
            be.kernel_events = None
        err = float((local - shard.local_rows(full)).abs().max() / full.abs().max())
        torch.cuda.synchronize()
        # the product ran over ALL rows at the slice width (64 / world columns)
        widths = sorted({k[2] for _, _, csr in ops.csr_cache.entries.values() for k, sl in csr._sliced.items()
                         if sl is not None and k[1] == n})
        out_q.put((rank, err, widths, "dif_sliced_spmm_f32" in launched))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,deg", [(20000, 60), (9001, 50)])
def test_two_ranks_on_one_gpu_slice_sharded_product(n, deg):
    """RowShard.product = "slice" on the HIP kernels: both ranks sweep the whole graph at 32 of the 64 columns between
    two all-to-alls (over gloo here); every rank's rows equal the single-process run."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slice_worker, args=(r, world, port, n, deg, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=420) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, err, widths, sliced in sorted(results):
        assert err < 1e-5, (rank, err)
        assert sliced and 32 in widths, (rank, widths)


@pytest.mark.parametrize("workload,product", [("ogbn-proteins-s", "row"), ("pokec-batch-s-bf16", "row"), ("ogbn-proteins-s", "slice"),
                                              ("ogbn-proteins-s", "auto")])
def test_bench_with_two_ranks_runs_end_to_end(workload, product):
    """The command the driver launches for N = 2 (python -m torch.distributed.run ... bench.py --gpus 2), with both ranks
    on this box's one GPU and the collectives over gloo: the row-sharded headline workload (closed-form layers, the
    shard's sliced product) and the replica workload print one well-formed JSON line from rank 0."""
    import json
    import subprocess
    env = dict(os.environ, DIFFORMER_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", workload, "--shard-product", product]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["unit"] == "nodes/s"
    if workload == "ogbn-proteins-s":
        assert d["scaling"] == "strong" and d["config"]["parallelism"].startswith("row-shard x2, closed-form aggregation split by ")
        by = d["ms_per_step_by_shard_product"]
        assert set(by) == ({"slice", "row"} if product == "auto" else {product}) and all(v > 0 for v in by.values())
        chosen = "feature slices" if min(by, key=by.get) == "slice" else "destination rows"
        assert chosen in d["config"]["parallelism"] and abs(d["ms_per_step"] - min(by.values())) < 1e-9
        assert d["ms_per_step_exact_fp32"] > 0 and d["config"]["exact_fp32"] is (os.environ.get("DIFFORMER_EXACT_FP32") == "1") and d["roofline"]["bound"] == "hbm"
        assert 0 < d["roofline"]["frac"] < 1 and "lds_frac" in d["roofline"] and not any(isinstance(v, dict) for v in d["roofline"].values())
        assert d["roofline"]["kernel"].startswith("sliced_spmm_kernel") and d["roofline"]["avg_launch_ms"] > 0
        # per-rank diagnostics of a sharded run: exposed collective waits beside the kernel groups (round 4)
        phases = d["per_rank_phases"]
        assert [p["rank"] for p in phases] == [0, 1] and all(p["kernels_ms"].get("product", 0) > 0 for p in phases)
        assert all(any(k.startswith("all_") for k in p["collectives_ms"]) for p in phases), phases
    else:
        assert d["scaling"] == "weak" and d["config"]["parallelism"] == "replicas x2"

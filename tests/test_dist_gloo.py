"""world_size-2 (and 3, uneven blocks) runs of the row-sharded forward on CPU over gloo.

The collectives, row offsets, global-N handling and all-gather compaction are the product's
(difformer_amd/dist.py, ops.py); the per-rank arithmetic is the test-only OracleBackend.  Each rank
checks its slice of the output against the single-process result."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, kernel, n, out_q, heads=2):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from difformer_amd import DIFFormer, RowShard, ops
        from fake_backend import OracleBackend
        ops._BACKEND = OracleBackend()
        if n >= 64 * world * world:
            # force the source-blocked layout at test size, so that the split product of the sharded SpMM (own blocks
            # under the all-gather, the rest after it) runs: 2 blocks per rank
            ops.choose_source_blocks = lambda num_nodes, row_bytes, nnz: 4
            ops.L2_SLICE_BYTES = (-(-n // world) + 7) // 8 * 8 // 2 * 128 / 1.1
        torch.manual_seed(7)
        model = DIFFormer(12, 16, 5, num_layers=2, num_heads=heads, kernel=kernel, use_source=True).eval()
        g = torch.Generator().manual_seed(3)
        x = torch.randn(n, 12, generator=g)
        ei = torch.cat([torch.randint(0, n, (2, 6 * n), generator=g), torch.arange(n).repeat(2, 1)], dim=1)
        with torch.no_grad():
            full = model(x, ei)                                  # single-process result, no shard
            shard = RowShard.from_process_group(n)
            assert shard.world == world and shard.rank == rank
            model.set_row_shard(shard)
            shard.timeline = []                                  # bench.py --gpus N: where the exchange steps' time goes
            local = model(shard.local_rows(x).contiguous(), ei)  # LOCAL rows of x, GLOBAL edge_index
            spans = shard.timeline_ms()
            shard.timeline = None
            spans_ok = "all_gather(rows): exposed wait" in spans and all(v >= 0.0 for v in spans.values())
            if heads == 1 and kernel == "simple":
                spans_ok = spans_ok and "all_reduce(record)" in spans          # closed form: the Gram record's all-reduce
        want = shard.local_rows(full)
        err = float((local - want).abs().max() / full.abs().max())
        # the gathered value rows must come back in global row order even when blocks are uneven
        gathered = shard.all_gather_rows(shard.local_rows(x).contiguous())
        ok = bool(torch.equal(gathered, x)) and spans_ok
        # ... and also for caller-chosen blocks that do not start at multiples of the largest block
        if world == 3:
            from difformer_amd.dist import RowShard as RS
            odd = RS(n, rank, world, None, counts=[n - 30, 7, 23])
            ok = ok and bool(torch.equal(odd.all_gather_rows(odd.local_rows(x).contiguous()), x))
        if n >= 64 * world * world and heads > 1:
            csr = list(ops.csr_cache.entries.values())[-1][2]
            ok = ok and csr.n_blocks == 2 * world and ops._BACKEND.part_calls == {0: 2, 1: 2}     # 2 layers x 2 parts
        if heads == 1 and kernel == "simple":
            # one head: the layers ran in closed form (Gram record all-reduced, source rows all-gathered)
            ok = ok and getattr(ops._BACKEND, "closed_form_calls", 0) == 4                          # 2 layers x 2 forwards
        out_q.put((rank, err, tuple(local.shape), ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kernel,world,n,heads", [("simple", 2, 64, 2), ("simple", 3, 50, 2), ("sigmoid", 2, 41, 2),
                                                  ("simple", 2, 300, 2), ("simple", 3, 620, 2), ("simple", 2, 300, 1),
                                                  ("simple", 3, 50, 1)])
def test_row_sharded_forward_matches_single_process(kernel, world, n, heads):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kernel, n, q, heads)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from difformer_amd.dist import split_rows
    counts = split_rows(n, world)
    for rank, err, shape, gathered_ok in sorted(results):
        assert shape == (counts[rank], 5)
        assert gathered_ok
        assert err < 1e-5, (rank, err)


def _slice_worker(rank, world, port, n, hidden, out_q):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from difformer_amd import DIFFormer, RowShard, ops
        from fake_backend import OracleBackend
        ops._BACKEND = OracleBackend()
        torch.manual_seed(7)
        model = DIFFormer(12, hidden, 5, num_layers=3, num_heads=1, kernel="simple", use_source=True).eval()
        g = torch.Generator().manual_seed(3)
        x = torch.randn(n, 12, generator=g)
        ei = torch.cat([torch.randint(0, n, (2, 6 * n), generator=g), torch.arange(n).repeat(2, 1)], dim=1)
        with torch.no_grad():
            full = model(x, ei)
            shard = RowShard.from_process_group(n)
            shard.product = "slice"
            model.set_row_shard(shard)
            # the two exchanges are inverse to each other and move the right columns
            xl = shard.local_rows(full).contiguous()
            w = shard.slice_width(full.shape[1]) if full.shape[1] % (4 * world) == 0 else 0
            t = torch.arange(n * hidden, dtype=torch.float32).reshape(n, hidden)
            cols = shard.all_to_all_columns(shard.local_rows(t).contiguous())
            wh = hidden // world
            ok = bool(torch.equal(cols, t[:, rank * wh: (rank + 1) * wh]))
            ok = ok and bool(torch.equal(shard.all_to_all_rows(cols), shard.local_rows(t)))
            ops._BACKEND.closed_form_calls = 0
            local = model(shard.local_rows(x).contiguous(), ei)
            ok = ok and ops._BACKEND.closed_form_calls == 3 and ops.slice_sharded(shard, hidden)
            # row-sharded run of the same model for comparison
            shard.product = "row"
            local_row = model(shard.local_rows(x).contiguous(), ei)
        want = shard.local_rows(full)
        err = float((local - want).abs().max() / full.abs().max())
        err_row = float((local_row - want).abs().max() / full.abs().max())
        out_q.put((rank, err, err_row, tuple(local.shape), ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,hidden", [(2, 64, 16), (3, 50, 24), (2, 301, 64), (8, 203, 32)])
def test_slice_sharded_product_matches_single_process(world, n, hidden):
    """RowShard.product = "slice": closed-form layers split the aggregation by feature columns (all-to-all in, product of
    the whole graph at C / world columns, all-to-all out); rows stay where they are for everything else."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slice_worker, args=(r, world, port, n, hidden, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from difformer_amd.dist import split_rows
    counts = split_rows(n, world)
    for rank, err, err_row, shape, ok in sorted(results):
        assert shape == (counts[rank], 5) and ok
        assert err < 1e-5 and err_row < 1e-5, (rank, err, err_row)


def test_slice_width_rules():
    from difformer_amd.dist import RowShard
    assert RowShard(100, 0, 8).slice_width(64) == 8 and RowShard(100, 0, 2).slice_width(64) == 32
    assert RowShard(100, 0, 3).slice_width(64) == 0          # 64 columns do not split into three 16-byte-aligned blocks
    assert RowShard(100, 0, 1).slice_width(64) == 0


def _bench_worker(rank, world, port, out_q):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from difformer_amd import RowShard
        res = {}
        for wl in ("ogbn-proteins-s", "pokec-batch-s-bf16", "cifar50k-s"):
            plan = bench.shard_plan(wl, world, rank)
            n = bench.WORKLOADS[wl][0]
            if not plan["replicas"]:                     # the plan is what RowShard hands the model
                sh = RowShard.from_process_group(n)
                assert (sh.row_begin, sh.n_local) == (plan["row_begin"], plan["n_local"])
            elapsed = bench.max_over_ranks(0.1 * (rank + 1), torch.device("cpu"))       # rank r "took" 0.1 (r + 1) s
            res[wl] = (plan, elapsed, bench.job_value(n, 10, elapsed, world, plan["replicas"]))
        out_q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bench_sharding_arithmetic_dry_run(world):
    """bench.py --gpus N without GPUs: which rows each rank takes, max-over-ranks timing and the whole-job value."""
    import bench
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for wl in ("ogbn-proteins-s", "cifar50k-s"):                     # one graph, rows split: strong scaling
        n = bench.WORKLOADS[wl][0]
        spans = sorted((got[r][wl][0]["row_begin"], got[r][wl][0]["n_local"]) for r in range(world))
        assert spans[0][0] == 0 and all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert spans[-1][0] + spans[-1][1] == n and min(c for _, c in spans) > 0
        for r in range(world):
            plan, elapsed, value = got[r][wl]
            assert plan["scaling"] == "strong" and plan["parallelism"] == f"row-shard x{world}"
            assert abs(elapsed - 0.1 * world) < 1e-9 and abs(value - n * 10 / (0.1 * world)) < 1e-3
    for r in range(world):                                            # independent batches: replicas, weak scaling
        plan, elapsed, value = got[r]["pokec-batch-s-bf16"]
        n = bench.WORKLOADS["pokec-batch-s-bf16"][0]
        assert plan["replicas"] and plan["scaling"] == "weak" and (plan["row_begin"], plan["n_local"]) == (0, n)
        assert abs(value - world * n * 10 / (0.1 * world)) < 1e-3
    one = bench.shard_plan("ogbn-proteins-s", 1, 0)
    assert one["parallelism"] == "single GPU" and one["n_local"] == 132534


def _train_worker(rank, world, port, kernel, n, heads, use_graph, out_q, product="row", hidden=16):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from difformer_amd import DIFFormer, RowShard, ops
        from fake_backend import OracleBackend
        ops._BACKEND = OracleBackend()
        torch.manual_seed(11)
        model = DIFFormer(12, hidden, 5, num_layers=2, num_heads=heads, kernel=kernel, dropout=0.0, use_source=True,
                          use_graph=use_graph).train()
        g = torch.Generator().manual_seed(5)
        x = torch.randn(n, 12, generator=g)
        ei = torch.cat([torch.randint(0, n, (2, 6 * n), generator=g), torch.arange(n).repeat(2, 1)], dim=1)
        target = torch.randn(n, 5, generator=g)
        # one process, all rows: loss = sum over nodes
        xf = x.clone().requires_grad_(True)
        ((model(xf, ei) - target) ** 2).sum().backward()
        ref = [p.grad.clone() for p in model.parameters()]
        ref_x = xf.grad.clone()
        model.zero_grad()
        # row-sharded: this rank's rows of the loss; parameter gradients summed over the ranks afterwards
        shard = RowShard.from_process_group(n)
        shard.product = product
        model.set_row_shard(shard)
        xl = shard.local_rows(x).contiguous().requires_grad_(True)
        out = model(xl, ei)
        ((out - shard.local_rows(target)) ** 2).sum().backward()
        shard.all_reduce_gradients(model.parameters())
        scale = max(float(r.abs().max()) for r in ref)
        err_p = max(float((p.grad - r).abs().max()) for p, r in zip(model.parameters(), ref)) / scale
        err_x = float((xl.grad - shard.local_rows(ref_x)).abs().max() / ref_x.abs().max())
        out_q.put((rank, err_p, err_x))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kernel,world,n,heads,use_graph,product,hidden",
                         [("simple", 2, 64, 2, True, "row", 16), ("simple", 3, 50, 1, True, "row", 16),
                          ("sigmoid", 2, 41, 2, True, "row", 16), ("simple", 2, 300, 2, True, "row", 16),
                          ("simple", 2, 40, 1, False, "row", 16),
                          # the slice shard (RowShard.product = "slice") is an INFERENCE split of closed-form layers: a training
                          # step under it must take the row-sharded operator path and give the same gradients -- at the
                          # driver's largest world size, uneven last block
                          ("simple", 8, 203, 1, True, "slice", 32), ("simple", 2, 64, 1, True, "slice", 16)])
def test_row_sharded_training_step_matches_single_process(kernel, world, n, heads, use_graph, product, hidden):
    """loss.backward() on row shards (main.py:130): the attention's sums over nodes and the aggregation's gathered rows
    carry their gradients back through the same collectives; parameter gradients summed over ranks and the gradient
    of the local input rows equal the single-process ones."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, kernel, n, heads, use_graph, q, product, hidden))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err_p, err_x in sorted(results):
        assert err_p < 2e-4 and err_x < 2e-4, (rank, err_p, err_x)

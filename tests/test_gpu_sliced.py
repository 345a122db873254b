"""Feature-sliced gcn_conv product with LDS-staged sources (csrc/gcn_sliced.hip), through the C ABI.

Integer work is checked exactly: the format must hold, for every destination row and source tile, exactly the CSR
entries of that group (as tile-local row numbers), every other slot a zero-row read, and the 16 lanes that share an
LDS cycle must hit 16 different bank quads in every step.  The product is held to the float64 oracle
(oracle.gcn_conv <- node classification/difformer.py:63-79) at 1e-5 (north_star: 1e-4).
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import difformer_oracle as orc

pytestmark = pytest.mark.gpu

# lane sets that share one LDS cycle of ds_read_b128 (MI355X_MICROARCH.md, LDS table)
HW_GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return torch.device("cuda:0")


def _dense_graph(n, deg, seed, self_loops=True, duplicates=True):
    g = torch.Generator().manual_seed(seed)
    e = n * deg
    ei = torch.randint(0, n, (2, e), generator=g)
    if duplicates:
        ei[:, : e // 50] = ei[:, e // 50: 2 * (e // 50)]         # repeated edges must be summed twice
    if self_loops:
        ei = torch.cat([ei, torch.arange(n).repeat(2, 1)], dim=1)
    return ei


def _sliced(csr, n, F):
    sl = csr.sliced(0, n, F)
    assert sl is not None, "expected the feature-sliced format for this graph"
    return sl


def _skewed_graph(n, deg, seed, hubs=40):
    """Half of the entries land on `hubs` destination rows (max degree ~ n * deg / (2 * hubs) >> mean)."""
    ei = _dense_graph(n, deg, seed)
    g = torch.Generator().manual_seed(seed + 1)
    half = ei.shape[1] // 2
    ei[1, :half] = torch.randint(0, hubs, (half,), generator=g) * (n // hubs)
    return ei


def _check_format(sl, ei, n, csr):
    slices, panels, G, PW, W, R, T, NT = (int(v) for v in sl.plan)
    n_pos = n if sl.n_pos is None else int(sl.n_pos)
    assert NT == csr.n_blocks and T % 16 == 0 and PW == panels * W and G == -(-n_pos // 64) and (R - 1) * PW < G <= R * PW
    order = np.arange(n) if sl.order is None else sl.order.cpu().numpy().astype(np.int64)
    assert order.size == n_pos
    if sl.parts is None:
        part, nparts = np.zeros(n_pos, np.int64), np.ones(n_pos, np.int64)
        assert np.array_equal(np.sort(order), np.arange(n))
    else:
        pp = sl.parts.cpu().numpy().view(np.uint16).astype(np.int64)
        part, nparts = pp & 0xFF, pp >> 8
        live = order >= 0
        assert np.array_equal(np.sort(order[live & (part == 0)]), np.arange(n)), "every row has exactly one part 0"
        assert np.all((nparts >= 1) & (nparts <= 64) & (part < nparts))
        # the parts of a row: consecutive positions of one slot, in part order
        heads = np.nonzero(live & (part == 0))[0]
        for h in heads[nparts[heads] > 1]:
            P = nparts[h]
            assert h // 64 == (h + P - 1) // 64 and np.all(order[h:h + P] == order[h]) and np.array_equal(part[h:h + P], np.arange(P))
            assert np.all(nparts[h:h + P] == P)
        assert int(nparts[heads].sum()) == int(live.sum())
    table = sl.table.cpu().numpy()
    n_ptw = panels * NT * W
    n_blocks = int(table[-1])
    rows = table[:-1].reshape(n_ptw, R + 1)
    ent = sl.entries.cpu().numpy().view(np.uint16).reshape(-1, 64, 8)[:n_blocks]
    # every step of every block: the 16 lanes of a hardware group read 16 different bank quads
    quads = (ent & 15).transpose(0, 2, 1)                       # [block, step, lane]
    for grp in HW_GROUPS:
        q = np.sort(quads[:, :, grp], axis=2)
        assert np.array_equal(q, np.broadcast_to(np.arange(16), q.shape)), "bank conflict in the schedule"
    assert ent.max() < T + 16
    # the real entries of (row position, tile) == its share of the CSR group
    src, dst = ei[0].numpy(), ei[1].numpy()
    perm = np.lexsort((src, dst))
    src_s, dst_s = src[perm], dst[perm]
    rowptr = np.searchsorted(dst_s, np.arange(n + 1))
    assert np.array_equal(rowptr, csr.rowptr.cpu().numpy())
    if sl.order is not None and sl.parts is None:
        deg = np.diff(rowptr)[order]
        assert np.all(deg[:-1] >= deg[1:]), "slots are formed in descending-degree order"
    # a part takes its share of the group in CSR order (entries of a (row, tile) group are filed by edge id)
    csr_src = csr.src.cpu().numpy().astype(np.int64)
    blkptr = csr.blkptr.cpu().numpy().reshape(NT + 1, n) if NT > 1 else None
    total_real, expect_start, seen_slots = 0, 0, 0
    for p in range(panels):
        for t in range(NT):
            for w in range(W):
                start, nb = int(rows[(p * NT + t) * W + w, 0]), rows[(p * NT + t) * W + w, 1:].astype(np.int64)
                assert start == expect_start and np.all(nb[:-1] >= nb[1:]), "round lengths must not increase"
                expect_start += int(nb.sum())
                pw = w * panels + p
                for j in range(R):
                    g = j * PW + (PW - 1 - pw if j & 1 else pw)
                    if g >= G:
                        assert nb[j] == 0
                        continue
                    seen_slots += (t == 0)
                    blocks = [start + int(np.minimum(nb, k).sum()) + j for k in range(int(nb[j]))]
                    lists = ent[blocks].transpose(1, 0, 2).reshape(64, -1) if blocks else np.zeros((64, 0), np.uint16)
                    for lane in range(64):
                        pos = g * 64 + lane
                        got = np.sort(lists[lane][lists[lane] < T].astype(np.int64))
                        if pos >= n_pos or order[pos] < 0:
                            assert got.size == 0
                            continue
                        row = order[pos]
                        seg = src_s[rowptr[row]: rowptr[row + 1]]
                        e0, e1 = (rowptr[row], rowptr[row + 1]) if NT == 1 else (blkptr[t, row], blkptr[t + 1, row])
                        grp_e = csr_src[e0:e1] - t * T
                        assert np.array_equal(np.sort(grp_e), seg[(seg >= t * T) & (seg < (t + 1) * T)] - t * T)
                        c = grp_e.size
                        want = np.sort(grp_e[c * part[pos] // nparts[pos]: c * (part[pos] + 1) // nparts[pos]])
                        assert np.array_equal(got, want), (p, t, w, j, lane)
                        total_real += got.size
    assert expect_start == n_blocks and seen_slots == G and total_real == ei.shape[1]


@pytest.mark.parametrize("n,deg,F", [(20000, 60, 64), (9000, 70, 64), (33000, 50, 128), (12000, 64, 32)])
def test_sliced_format_holds_the_csr_and_is_bank_conflict_free(n, deg, F, dev):
    from difformer_amd import ops
    ei = _dense_graph(n, deg, seed=n + F)
    csr = ops.csr_cache.get(ei.to(dev), None, n, F * 4)
    sl = _sliced(csr, n, F)
    assert sl.order is None                     # degrees about equal: natural row order
    _check_format(sl, ei, n, csr)


@pytest.mark.parametrize("n,deg,F", [(12000, 64, 64), (25000, 50, 32)])
def test_sliced_format_on_skewed_degrees_sorts_rows_into_slots(n, deg, F, dev):
    from difformer_amd import ops
    ei = _skewed_graph(n, deg, seed=n)
    csr = ops.csr_cache.get(ei.to(dev), None, n, F * 4)
    sl = _sliced(csr, n, F)
    assert sl.order is not None and sl.parts is not None       # 40 hub rows far beyond twice the mean degree: split
    _check_format(sl, ei, n, csr)


@pytest.mark.parametrize("frac,parts", [(4, 1), (16, 3)])
def test_sliced_format_on_mild_skew(frac, parts, dev):
    """1/frac of the rows carry half of the entries (mean 60).  frac = 4: ~150 entries, beyond twice the mean (rows are
    sorted into slots) but below the split threshold of 4x mean; frac = 16: ~510 entries -> three lanes each."""
    from difformer_amd import ops
    n, F = 12000, 64
    g = torch.Generator().manual_seed(2)
    light = torch.stack([torch.randint(0, n, (n * 30,), generator=g), torch.randint(0, n, (n * 30,), generator=g)])
    heavy = torch.stack([torch.randint(0, n, (n * 30,), generator=g), torch.randint(0, n // frac, (n * 30,), generator=g)])
    ei = torch.cat([light, heavy], dim=1)
    csr = ops.csr_cache.get(ei.to(dev), None, n, F * 4)
    sl = _sliced(csr, n, F)
    assert sl.order is not None
    got = 1 if sl.parts is None else int((sl.parts.cpu().numpy().view(np.uint16) >> 8).max())
    assert got == parts
    _check_format(sl, ei, n, csr)


@pytest.mark.parametrize("n,deg,h,d", [(20000, 60, 1, 64), (9000, 70, 1, 64), (33000, 50, 2, 64), (12000, 64, 1, 32),
                                       (50000, 100, 1, 64), (150000, 50, 1, 64)])
def test_sliced_product_vs_oracle_and_gather_kernel(n, deg, h, d, dev):
    from difformer_amd import gcn_conv, ops
    ei = _dense_graph(n, deg, seed=3 * n + d)
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, h, d, generator=g)
    eid, xd = ei.to(dev), x.to(dev)
    csr = ops.csr_cache.get(eid, None, n, h * d * 4)
    _sliced(csr, n, h * d)
    out = gcn_conv(xd, eid, None)
    ref = orc.gcn_conv(x.double().numpy(), ei.numpy(), None)
    assert rel_err(out.cpu().numpy(), ref) < 1e-5
    # same numbers as the round-1 gather kernel up to fp32 rounding; bitwise reproducible
    old = ops.get_backend().spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, csr.nnz, xd.reshape(n, h * d), 0,
                                 n)
    assert rel_err(out.reshape(n, h * d).cpu().numpy(), old.cpu().numpy()) < 1e-5
    assert torch.equal(gcn_conv(xd, eid, None), out)


def test_sliced_product_with_attention_combine_and_tail(dev):
    """The combine of difformer.py:130-134 rides in the epilogue; the LayerNorm tail follows as its own pass."""
    from difformer_amd import ops
    n, d = 15000, 64
    ei = _dense_graph(n, 64, seed=5).to(dev)
    g = torch.Generator().manual_seed(9)
    x, attn, prev = (torch.randn(n, 1, d, generator=g).to(dev) for _ in range(3))
    lw, lb = torch.rand(d, generator=g).to(dev) + 0.5, torch.randn(d, generator=g).to(dev)
    csr = ops.csr_cache.get(ei, None, n, d * 4)
    _sliced(csr, n, d)
    tail = dict(x0=None, prev=prev[:, 0, :], alpha=0.3, ln_weight=lw, ln_bias=lb, eps=1e-5)
    got = ops.gcn_aggregate(csr, x, attn, 0.25, 0.75, None, tail)[:, 0, :]
    old = ops.get_backend().spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, csr.nnz, x.reshape(n, d), 0, n,
                                 attn.reshape(n, d), 0.25, 0.75, tail)
    assert rel_err(got.cpu().numpy(), old.cpu().numpy()) < 1e-5


def test_sliced_is_declined_for_varying_weights_and_sparse_graphs(dev):
    from difformer_amd import ops
    n = 12000
    ei = _dense_graph(n, 64, seed=1).to(dev)
    w = torch.rand(ei.shape[1]).to(dev)
    assert ops.csr_cache.get(ei, w, n, 256).sliced(0, n, 64) is None             # edge weights that vary
    sparse = torch.randint(0, n, (2, 5 * n)).to(dev)
    assert ops.csr_cache.get(sparse, None, n, 256).sliced(0, n, 64) is None      # ~5 entries per row


@pytest.mark.parametrize("const", [1.0, 0.37, -2.5, 0.0, float("nan"), float("inf")])
def test_constant_weights_take_the_unweighted_product(const, dev):
    """`edge_attr = torch.ones(E)` (spatial-temporal/main.py:99,103) or any other constant: value_e = w d_in d_out is w times
    the unweighted value (difformer.py:70-74; a non-finite w: nan_to_num -> 0), so the graph takes the feature-sliced product
    and w rides in gcn_scale -- against the float64 oracle WITH the weights, forward and the gradient of x."""
    from difformer_amd import gcn_conv, ops
    n = 12000
    ei = _dense_graph(n, 64, seed=2)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, 1, 64, generator=g)
    w = torch.full((ei.shape[1],), const)
    eid, wd = ei.to(dev), w.to(dev)
    csr = ops.csr_cache.get(eid, wd, n, 256)
    assert not csr.weighted and csr.sliced(0, n, 64) is not None
    assert csr.weight_scale == (float(np.float32(const)) if np.isfinite(const) else 0.0)
    ref = orc.gcn_conv(x.double().numpy(), ei.numpy(), w.double().numpy())
    xd = x.to(dev).requires_grad_(True)
    out = gcn_conv(xd, eid, wd)
    assert rel_err(out.detach().cpu().numpy(), ref) < 1e-5 or (not np.abs(ref).max() and not out.abs().max())
    go = torch.randn(n, 1, 64, generator=g)
    out.backward(go.to(dev))
    # the adjoint through the identity <A x, g> = <x, A^T g>
    lhs = float((out.detach().double().cpu() * go.double()).sum())
    rhs = float((x.double() * xd.grad.double().cpu()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1e-3)
    # weights that are constant but WANT a gradient keep the weighted operator (difformer.py:73 is differentiable in them)
    wg = wd.clone().requires_grad_(True)
    assert ops.csr_cache.get(eid, wg, n, 256).weighted


@pytest.mark.parametrize("n,deg,hubs", [(12000, 64, 40), (30000, 80, 7), (64000, 50, 2000), (20000, 100, 2)])
def test_sliced_product_on_skewed_degrees(n, deg, hubs, dev):
    """Hub rows (up to 500,000 entries: split into up to 64 lock-step parts) and a long tail of short rows in the same
    launch: results against the float64 oracle."""
    from difformer_amd import gcn_conv, ops
    ei = _skewed_graph(n, deg, seed=7 * n, hubs=hubs)
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, 1, 64, generator=g)
    eid, xd = ei.to(dev), x.to(dev)
    csr = ops.csr_cache.get(eid, None, n, 256)
    sl = csr.sliced(0, n, 64)
    ref = orc.gcn_conv(x.double().numpy(), ei.numpy(), None)
    out = gcn_conv(xd, eid, None)
    assert rel_err(out.cpu().numpy(), ref) < 1e-5
    assert sl is not None and sl.order is not None and sl.parts is not None
    # hubs == 2: 500,000 entries on one row -- 64 lock-step parts of ~7,800 (a single lane could not hold a
    # (row, tile) group beyond the 16-bit counters, and would run alone for a millisecond)
    assert torch.equal(gcn_conv(xd, eid, None), out)


@pytest.mark.parametrize("n,deg,world,splits,skewed", [(90000, 60, 8, 8, False), (90000, 60, 4, 4, False),
                                                         (50000, 60, 2, 2, False), (90000, 60, 8, 8, True)])
def test_sliced_product_of_a_row_shard_splits_the_source_tiles(n, deg, world, splits, skewed, dev):
    """One rank's destination rows over ALL the source rows: fewer, fuller panels and `splits` workgroups per
    (panel, slice), each sweeping NT / splits source tiles; the partial sums meet in sliced_combine_kernel.  Every rank
    reproduces its rows of the float64 oracle, bit for bit from run to run."""
    from difformer_amd import ops
    from difformer_amd.dist import RowShard
    ei = _skewed_graph(n, deg, seed=11 * n, hubs=30) if skewed else _dense_graph(n, deg, seed=13 * n + world)
    g = torch.Generator().manual_seed(world)
    x = torch.randn(n, 64, generator=g)
    eid, xd = ei.to(dev), x.to(dev)
    be = ops.get_backend()
    ref = orc.gcn_conv(x.double().numpy()[:, None, :], ei.numpy(), None)[:, 0, :]
    for rank in sorted({0, world // 2, world - 1}):
        sh = RowShard(n, rank=rank, world=world)
        lo, cnt = sh.row_begin, sh.n_local
        csr = ops.csr_cache.get(eid, None, n, 256, sh)
        sl = csr.sliced(lo, cnt, 64)
        assert sl is not None
        n_pos = cnt if sl.n_pos is None else sl.n_pos
        assert be.lib.dif_sliced_spmm_workspace_bytes(n, n_pos, 64) == splits * 16 * int(sl.plan[2]) * 64 * 16
        assert int(sl.plan[7]) % splits == 0 and (sl.parts is not None) == skewed
        ys = be.sliced_prescale(xd, csr.rowptr, n, sl.plan)
        out = be.sliced_spmm(sl, ys, csr.rowptr, n, lo, cnt, 64)
        assert rel_err(out.cpu().numpy(), ref[lo: lo + cnt]) < 1e-5, (world, rank)
        assert torch.equal(be.sliced_spmm(sl, ys, csr.rowptr, n, lo, cnt, 64), out)
        attn = torch.randn(cnt, 64, generator=g).to(dev)
        mix = be.sliced_spmm(sl, ys, csr.rowptr, n, lo, cnt, 64, attn, 0.25, 0.75)
        assert rel_err(mix.cpu().numpy(), 0.75 * ref[lo: lo + cnt] + 0.25 * attn.cpu().numpy()) < 1e-5


def test_sliced_nodes_without_incoming_entries(dev):
    """deg = 0 -> infinite normaliser -> nan_to_num drops the entry (difformer.py:66-74)."""
    from difformer_amd import gcn_conv
    n = 10000
    g = torch.Generator().manual_seed(4)
    ei = torch.stack([torch.randint(0, n, (n * 60,), generator=g), torch.randint(n // 10, n, (n * 60,), generator=g)])
    x = torch.randn(n, 1, 64, generator=g)              # nodes < n/10 have no incoming entry but do send
    out = gcn_conv(x.to(dev), ei.to(dev), None).cpu().numpy()
    ref = orc.gcn_conv(x.double().numpy(), ei.numpy(), None)
    assert np.isfinite(out).all() and rel_err(out, ref) < 1e-5


@pytest.mark.parametrize("skewed", [False, True])
def test_sliced_adjoint_product_is_the_gradient(skewed, dev):
    """backward of difformer.py:75-78: grad_x = A_hat^T g over the CSR of the transposed graph, whose row lengths are
    OUT-degrees -- the normaliser comes from the forward graph (dinv vector through the C ABI)."""
    from difformer_amd import gcn_conv, ops
    n, d = 14000, 64
    ei = _skewed_graph(n, 60, seed=21) if skewed else _dense_graph(n, 60, seed=20)
    ei[0, : n] = torch.randint(0, 50, (n,), generator=torch.Generator().manual_seed(3))       # out-degrees != in-degrees
    g = torch.Generator().manual_seed(8)
    x = torch.randn(n, 1, d, generator=g)
    gout = torch.randn(n, 1, d, generator=g)
    eid = ei.to(dev)
    xd = x.to(dev).requires_grad_(True)
    out = gcn_conv(xd, eid, None)
    out.backward(gout.to(dev))
    csr = ops.csr_cache.get(eid, None, n, d * 4)
    adj = csr.adjoint()
    assert adj.sliced(0, n, d) is not None and adj.dinv is not None
    row, col = ei[0].numpy(), ei[1].numpy()                       # difformer.py:63-75 in float64
    deg = np.bincount(col, minlength=n).astype(np.float64)
    dinv = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1)), 0.0)
    val = dinv[col] * dinv[row]
    want = np.zeros((n, d))
    np.add.at(want, row, val[:, None] * gout[:, 0, :].double().numpy()[col])
    assert rel_err(xd.grad[:, 0, :].cpu().numpy(), want) < 1e-5


def _block_graph(n, deg, nb, intra, seed):
    """Nodes in nb contiguous blocks; a fraction `intra` of the edges stays inside the destination's block."""
    g = torch.Generator().manual_seed(seed)
    e = n * deg
    dst = torch.randint(0, n, (e,), generator=g)
    size = -(-n // nb)
    inside = torch.rand(e, generator=g) < intra
    src_in = ((dst // size) * size + torch.randint(0, size, (e,), generator=g)).clamp_(max=n - 1)
    src = torch.where(inside, src_in, torch.randint(0, n, (e,), generator=g))
    return torch.cat([torch.stack([src, dst]), torch.arange(n).repeat(2, 1)], dim=1)


def test_model_runs_community_structured_graphs_in_a_mixed_node_order(dev):
    """Every row's entries in one or two source tiles (8 blocks, 95 % of the edges inside): the model permutes x once,
    runs on the relabelled graph (ops.MixedGraph) and permutes the logits back -- same result as the float64 oracle on
    the original graph, and a format without the empty-round padding."""
    from difformer_amd import DIFFormer, ops
    n, deg, hidden = 24000, 60, 64
    ei = _block_graph(n, deg, 8, 0.95, seed=3)
    torch.manual_seed(1)
    cfg = dict(hidden_channels=hidden, num_layers=2, num_heads=1, kernel="simple", alpha=0.5, use_bn=True, use_residual=True,
               use_weight=True, use_graph=True, graph_weight=-1, use_source=False)
    model = DIFFormer(12, hidden, 5, num_layers=2, num_heads=1, kernel="simple").to(dev).eval()
    x = torch.randn(n, 12, generator=torch.Generator().manual_seed(2))
    eid = ei.to(dev)
    with torch.no_grad():
        out = model(x.to(dev), eid)
    mix = ops.mix_cache.get(eid, n, hidden)
    assert mix is not None and torch.equal(mix.inv[mix.perm], torch.arange(n, device=dev))
    p = {k: v.detach().cpu().double().numpy() for k, v in model.state_dict().items()}
    ref = orc.difformer_forward(p, x.double().numpy(), ei.numpy(), None, cfg)
    assert rel_err(out.cpu().numpy(), ref) < 1e-4
    # the natural order pads to the empty rounds, the mixed one does not
    nat = ops.csr_cache.get(eid, None, n, hidden * 4).sliced(0, n, hidden)
    mixed = ops.csr_cache.get(mix.edge_index, None, n, hidden * 4).sliced(0, n, hidden)
    assert nat is not None and mixed is not None
    assert int(mixed.table[-1]) < int(nat.table[-1])          # 12 % fewer blocks at this size (3 tiles, 2 rounds); 2.2x at the C4 size
    # a uniform graph keeps its order
    uni = _dense_graph(n, deg, seed=9).to(dev)
    assert ops.mix_cache.get(uni, n, hidden) is None
    # training goes through the two permutations too (main.py:117-131): same parameter gradients as in the natural order
    # (whose kernels are held to float64 autograd in test_gpu_parity.py)
    model.train()
    model.dropout = 0.0
    xg = x.to(dev)

    def grads():
        model.zero_grad()
        model(xg, eid).square().mean().backward()
        return {k: v.grad.detach().clone() for k, v in model.named_parameters()}
    g_mixed = grads()
    saved, ops.MIX_THRESHOLD = ops.MIX_THRESHOLD, 1e9
    ops.mix_cache.clear()
    try:
        g_nat = grads()
        assert ops.mix_cache.get(eid, n, hidden) is None
    finally:
        ops.MIX_THRESHOLD = saved
        ops.mix_cache.clear()
    for k in g_nat:
        assert rel_err(g_mixed[k].cpu().numpy(), g_nat[k].cpu().numpy()) < 1e-4, k

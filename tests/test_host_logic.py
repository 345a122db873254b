"""Host-side logic of the package on CPU: module surface, state_dict/initialisation parity with the
reference, row-shard arithmetic, CSR caching, blocking heuristics and loud failure on CPU operands.
Arithmetic is delegated to tests/fake_backend.OracleBackend (test-only)."""
import inspect

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err, split_model_case
from fake_backend import OracleBackend

MODEL = load_golden("model")


@pytest.fixture()
def fake_backend(monkeypatch):
    from difformer_amd import ops
    be = OracleBackend()
    monkeypatch.setattr(ops, "_BACKEND", be)
    ops.csr_cache.clear()
    yield be
    ops.csr_cache.clear()


def _build(c):
    from difformer_amd import DIFFormer
    cfg, sd = split_model_case(c)
    kw = {k: cfg[k] for k in ("num_layers", "num_heads", "kernel", "alpha", "use_bn", "use_residual", "use_weight",
                              "use_graph", "graph_weight", "use_source")}
    kw["kernel"] = str(kw["kernel"])
    model = DIFFormer(int(cfg["in_channels"]), int(cfg["hidden_channels"]), int(cfg["out_channels"]), **kw)
    return model, cfg, sd, kw


def test_module_surface_matches_reference_signatures():
    """Names, argument order and defaults of difformer.py:10,63,85-92,113,154-155,184."""
    import difformer_amd.difformer as m
    assert set(m.__all__) == {"full_attention_conv", "gcn_conv", "DIFFormerConv", "DIFFormer"}
    sig = inspect.signature(m.DIFFormer.__init__)
    assert list(sig.parameters)[1:] == ["in_channels", "hidden_channels", "out_channels", "num_layers", "num_heads",
                                        "kernel", "alpha", "dropout", "use_bn", "use_residual", "use_weight",
                                        "use_graph", "graph_weight", "use_source"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["num_layers"], d["num_heads"], d["kernel"], d["alpha"], d["dropout"]) == (2, 1, "simple", 0.5, 0.5)
    assert (d["use_bn"], d["use_residual"], d["use_weight"], d["use_graph"], d["graph_weight"], d["use_source"]) == \
        (True, True, True, True, -1, False)
    assert list(inspect.signature(m.DIFFormer.forward).parameters) == ["self", "x", "edge_index", "edge_weight"]
    assert list(inspect.signature(m.DIFFormerConv.forward).parameters) == \
        ["self", "query_input", "source_input", "edge_index", "edge_weight", "x_0", "output_attn"]
    assert list(inspect.signature(m.full_attention_conv).parameters) == ["qs", "ks", "vs", "kernel", "output_attn"]
    assert list(inspect.signature(m.gcn_conv).parameters) == ["x", "edge_index", "edge_weight"]


@pytest.mark.parametrize("name", sorted(MODEL))
def test_state_dict_keys_and_seeded_init_match_reference(name):
    """Strict load of the reference's state_dict (test_large_dataset.py:88) and identical seeded
    initialisation (parameter creation / reset order, main.py:110)."""
    model, cfg, sd, _ = _build(MODEL[name])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    torch.manual_seed(123)
    fresh, *_ = _build(MODEL[name])
    fresh.reset_parameters()
    for k, v in fresh.state_dict().items():
        if k.startswith("bns."):
            continue  # the fixture perturbs LayerNorm affine parameters after init
        assert np.array_equal(v.numpy(), sd[k]), k


@pytest.mark.parametrize("name", sorted(MODEL))
def test_module_plumbing_against_golden(name, fake_backend):
    """DIFFormer / DIFFormerConv sequencing (projection slicing, combine scales, tail arguments, CSR
    cache use) is right: with the oracle doing the arithmetic the reference outputs come back."""
    c = MODEL[name]
    model, cfg, sd, _ = _build(c)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    ei = torch.from_numpy(c["edge_index"]) if cfg["use_graph"] else None
    w = torch.from_numpy(c["edge_weight"]) if "edge_weight" in c else None
    with torch.no_grad():
        out = model(torch.from_numpy(c["x"]), ei, w)
        h0 = model._input_layer(torch.from_numpy(c["x"]), False)
        conv0 = model.convs[0](h0, h0, ei, w, h0)
    assert rel_err(out.numpy(), c["out_f64"]) < 1e-5
    assert rel_err(conv0.numpy(), c["conv0_f64"]) < 1e-5


def test_get_attentions_shape_and_rows_sum(fake_backend):
    from difformer_amd import DIFFormer
    torch.manual_seed(0)
    for kernel in ("simple", "sigmoid"):
        model = DIFFormer(6, 8, 3, num_layers=2, kernel=kernel, use_graph=False).eval()
        with torch.no_grad():
            att = model.get_attentions(torch.randn(11, 6))
        assert att.shape == (2, 11, 11, 1)
        if kernel == "sigmoid":
            assert torch.allclose(att.sum(dim=2), torch.ones(2, 11, 1), atol=1e-5)


def test_split_rows_and_row_shard():
    from difformer_amd.dist import RowShard, split_rows
    assert split_rows(132534, 8) == [16568] * 7 + [16558]       # ceil(N / world) rounded up to a multiple of 8
    assert split_rows(1000, 8) == [125] * 8                     # small graphs: plain ceil(N / world)
    assert split_rows(5, 8) == [1, 1, 1, 1, 1, 0, 0, 0]
    with pytest.raises(ValueError, match="at least one row"):       # a rank without rows would hang the collectives
        RowShard(5, rank=0, world=8)
    with pytest.raises(ValueError, match="at least one row"):
        RowShard(9, rank=3, world=4)                                 # [3, 3, 3, 0]
    assert split_rows(64, 2) == [32, 32]
    s = RowShard(10, rank=2, world=3)
    assert s.counts == [4, 4, 2] and s.offsets == [0, 4, 8, 10] and (s.row_begin, s.n_local) == (8, 2)
    assert torch.equal(s.local_rows(torch.arange(10)), torch.tensor([8, 9]))
    with pytest.raises(ValueError):
        RowShard(10, 0, 2, counts=[4, 4])
    one = RowShard(7)
    t = torch.randn(7, 3)
    assert one.all_gather_rows(t) is t and one.all_reduce_sum(t) is t     # world 1: no collective


def test_choose_source_blocks():
    from difformer_amd.ops import choose_source_blocks as c
    assert c(2708, 256, 13264) == 1                    # Cora: x fits L2
    assert c(100000, 256, 330000) == 1                 # Pokec batch: rows too sparse to block
    assert c(132534, 256, 79255038) == 13              # ogbn-proteins: ~2.5 MiB of x per block
    assert 2 <= c(10 ** 7, 256, 10 ** 9) <= 64


def test_csr_cache_identity_version_and_eviction(fake_backend):
    from difformer_amd import ops
    ei = torch.randint(0, 50, (2, 300))
    a = ops.csr_cache.get(ei, None, 50)
    assert ops.csr_cache.get(ei, None, 50) is a
    ei[0, 0] = (ei[0, 0] + 1) % 50                       # in-place edit -> _version bump -> rebuild
    b = ops.csr_cache.get(ei, None, 50)
    assert b is not a
    w = torch.rand(300)
    assert ops.csr_cache.get(ei, w, 50) is not b          # edge_weight is part of the key
    for _ in range(20):
        ops.csr_cache.get(torch.randint(0, 50, (2, 10)), None, 50)
    assert len(ops.csr_cache.entries) <= ops.csr_cache.capacity
    # entries whose edge tensors are gone are dropped at the next lookup (a mini-batch loop makes one graph per batch)
    ops.csr_cache.clear()
    keep = torch.randint(0, 50, (2, 40))
    ops.csr_cache.get(keep, None, 50)
    for _ in range(3):
        ops.csr_cache.get(torch.randint(0, 50, (2, 10)), None, 50)     # temporaries: freed right after the call
    ops.csr_cache.get(keep, None, 50)
    assert len(ops.csr_cache.entries) == 1
    wg = torch.rand(40, requires_grad=True)
    # the reference differentiates through the values (difformer.py:73): one GPU returns that gradient
    # (autograd_ops._GcnAggregate), a row-sharded run refuses instead of dropping it silently
    assert ops.csr_cache.get(keep, wg, 50).weight_leaf() is wg
    with torch.no_grad():
        assert ops.csr_cache.get(keep, wg, 50).weight_leaf() is None
    from difformer_amd.dist import RowShard
    with pytest.raises(NotImplementedError, match="edge_weight"):
        ops.csr_cache.get(keep, wg, 50, shard=RowShard(50, 0, 2))


def test_simple_attention_under_grad_checks_the_lengths(fake_backend):
    from difformer_amd import full_attention_conv
    q = torch.randn(6, 1, 4, requires_grad=True)
    with pytest.raises(RuntimeError, match="as many queries as sources"):
        full_attention_conv(q, torch.randn(5, 1, 4), torch.randn(5, 1, 4), "simple")


def test_row_major_layout_helper():
    from difformer_amd.backend_hip import _row_major
    qkv = torch.randn(10, 3 * 8)
    v = qkv[:, 16:].reshape(10, 2, 4)
    t, ld = _row_major(v, 8)
    assert t.data_ptr() == v.data_ptr() and ld == 24       # strided view passes through, no copy
    t, ld = _row_major(torch.randn(4, 10).t(), 4)
    assert t.is_contiguous() and ld == 4
    t, ld = _row_major(torch.randn(1, 2, 4), 8)
    assert ld == 8


def test_cpu_operands_raise_without_fallback():
    from difformer_amd.backend_hip import HipBackend
    be = HipBackend()                                        # loads the .so: fine without a GPU
    q = torch.randn(4, 1, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        be.simple_reduce(q, q, q)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        be.csr_build(torch.zeros(2, 3, dtype=torch.long), None, 4)


def test_training_backward_flows_through_autograd_wrappers(fake_backend):
    """loss.backward() works through every operator wrapper (forward via the backend, backward by
    recomputation) and matches autograd of the plain closed forms."""
    from difformer_amd import autograd_ops as ag
    torch.manual_seed(0)
    q, k, v = (torch.randn(9, 2, 4, requires_grad=True) for _ in range(3))
    out = ag.simple_attention(q, k, v)
    out.square().sum().backward()
    g = [t.grad.clone() for t in (q, k, v)]
    for t in (q, k, v):
        t.grad = None
    ag._simple_expr(q, k, v).square().sum().backward()
    for a, t in zip(g, (q, k, v)):
        assert torch.allclose(a, t.grad, rtol=1e-3, atol=1e-5)
    # gcn aggregate: transposed product
    from difformer_amd import ops
    ei = torch.randint(0, 9, (2, 40))
    csr = ops.GraphCSR.build(ei, None, 9)
    x = torch.randn(9, 1, 4, requires_grad=True)
    ag.gcn_aggregate(csr, x).square().sum().backward()
    gx = x.grad.clone()
    rp, src, val = csr.rowptr.long(), csr.src.long()[:40], csr.val[:40]
    dst = torch.repeat_interleave(torch.arange(9), rp[1:] - rp[:-1])
    A = torch.zeros(9, 9).index_put_((dst, src), val, accumulate=True)
    x2 = x.detach().clone().requires_grad_(True)
    torch.einsum("ij,jhd->ihd", A, x2).square().sum().backward()
    assert torch.allclose(gx, x2.grad, rtol=1e-3, atol=1e-5)


# ---- f4: DIFFormer_v2 (physical particle/difformer-v2.py) ---------------------------------------------------------
V2 = load_golden("v2")
V2_MODELS = sorted(n for n in V2 if n.startswith("model/"))


def _build_v2(c):
    from difformer_amd import DIFFormer_v2
    cfg, sd = split_model_case(c)
    kw = {k: cfg[k] for k in ("num_layers", "kernel", "alpha", "use_bn", "use_residual", "use_weight", "use_graph",
                              "graph_weight")}
    kw["kernel"] = str(kw["kernel"])
    h = int(cfg["hidden_channels"])
    return DIFFormer_v2(int(cfg["in_channels"]), h, h, **kw), cfg, sd


def test_v2_module_surface_matches_reference_signatures():
    """difformer-v2.py:48,71,137,162-163,193 and the public helper names :8-27."""
    import difformer_amd.difformer_v2 as m
    assert set(m.__all__) == {"make_batch_mask", "make_batch", "to_pad", "gcn_conv", "TransConv", "DIFFormer_v2"}
    sig = inspect.signature(m.DIFFormer_v2.__init__)
    assert list(sig.parameters)[1:] == ["in_channels", "hidden_channels", "out_channels", "num_layers", "kernel", "alpha",
                                        "dropout", "use_bn", "use_residual", "use_weight", "use_graph", "graph_weight"]
    assert list(inspect.signature(m.DIFFormer_v2.forward).parameters) == ["self", "x", "edge_index", "n_nodes"]
    assert list(inspect.signature(m.TransConv.__init__).parameters)[1:] == \
        ["in_channels", "out_channels", "num_heads", "kernel", "use_graph", "use_weight", "graph_weight"]
    assert list(inspect.signature(m.TransConv.forward).parameters) == \
        ["self", "query_input", "source_input", "n_nodes", "edge_index", "edge_weight"]
    assert list(inspect.signature(m.TransConv.full_attention).parameters) == ["self", "qs", "ks", "vs", "kernel", "n_nodes"]
    n = torch.tensor([2, 0, 3])
    mask, mx = m.make_batch_mask(n)
    assert mx == 3 and mask.tolist() == [[True, True, False], [False, False, False], [True, True, True]]
    assert m.make_batch(n).tolist() == [0, 0, 2, 2, 2]
    pad = m.to_pad(torch.arange(5.0).reshape(5, 1, 1), mask, mx, 3)
    assert pad[:, :, 0, 0].tolist() == [[0, 1, 0], [0, 0, 0], [2, 3, 4]]


@pytest.mark.parametrize("name", V2_MODELS)
def test_v2_state_dict_and_plumbing_against_golden(name, fake_backend):
    c = V2[name]
    model, cfg, sd = _build_v2(c)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    torch.manual_seed(321)
    fresh, *_ = _build_v2(c)
    fresh.reset_parameters()
    for k, v in fresh.state_dict().items():
        if not k.startswith("bns."):
            assert np.array_equal(v.numpy(), sd[k]), k
    model.eval()
    ei = torch.from_numpy(c["edge_index"]) if cfg["use_graph"] else None
    with torch.no_grad():
        out = model(torch.from_numpy(c["x"]), ei, torch.from_numpy(c["n_nodes"]))
    assert rel_err(out.numpy(), c["out_f64"]) < 1e-5


def test_batch_layout_tables():
    from difformer_amd.ops import BatchLayout, layout_cache
    n = torch.tensor([3, 1, 0, 4, 3])
    lay = BatchLayout(n, "cpu")
    assert (lay.n_graphs, lay.n_rows, lay.max_nodes) == (5, 11, 4)
    assert lay.graph_ptr.tolist() == [0, 3, 4, 4, 8, 11]
    assert lay.ranked_first.tolist() == [4, 0, 8, 3, 4]            # sizes 4, 3, 3, 1, 0 (stable among equals)
    assert lay.pos_count.tolist() == [4, 3, 3, 1]
    a = layout_cache.get(n, "cpu")
    assert layout_cache.get(n, "cpu") is a
    n[0] = 2                                                       # in-place edit -> new version -> rebuilt
    assert layout_cache.get(n, "cpu") is not a
    with pytest.raises(ValueError):
        BatchLayout(torch.tensor([2, -1]), "cpu")


def test_v2_use_weight_false_fails_like_the_reference(fake_backend):
    from difformer_amd import DIFFormer_v2
    model = DIFFormer_v2(4, 8, 8, use_weight=False).eval()
    with pytest.raises(UnboundLocalError):
        model(torch.randn(5, 4), torch.zeros(2, 0, dtype=torch.long), torch.tensor([5]))


def test_v2_training_backward_matches_reference_expression(fake_backend):
    """Gradients through the batched attention wrappers equal autograd through the padded formulation."""
    from difformer_amd import autograd_ops as ag, ops
    torch.manual_seed(3)
    n = torch.tensor([4, 1, 6])
    lay = ops.BatchLayout(n, "cpu")
    for kernel, expr in (("simple", ag._batched_simple_expr), ("sigmoid", ag._batched_sigmoid_expr)):
        q, k, v = (torch.randn(11, 2, 8, requires_grad=True) for _ in range(3))
        out = ag.batched_attention(q, k, v, lay, kernel)
        g = torch.randn_like(out)
        out.backward(g)
        q2, k2, v2 = (t.detach().double().requires_grad_(True) for t in (q, k, v))
        expr(lay)(q2, k2, v2).backward(g.double())
        for a, b in ((q, q2), (k, k2), (v, v2)):
            assert rel_err(a.grad.numpy(), b.grad.numpy()) < 1e-5


def test_split_positions_layout():
    """Row positions of the feature-sliced format when hub rows are split (ops.split_positions; "row positions" in
    include/difformer_hip.h): P = min(ceil(d / cap), 64) consecutive positions of one slot per row, part 0 first."""
    import numpy as np
    from difformer_amd.ops import split_positions
    g = torch.Generator().manual_seed(0)
    deg = torch.cat([torch.tensor([100000, 5000, 4999, 1300, 1201, 1200, 601, 601, 601]),
                     torch.randint(0, 600, (3000,), generator=g)])
    deg, idx = torch.sort(deg, descending=True, stable=True)
    order = torch.randperm(deg.numel(), generator=g).to(torch.int32)     # any row ids
    cap = 600
    order2, parts, n_pos = split_positions(deg, order, cap)
    order2, parts = order2.numpy().astype(np.int64), parts.numpy().view(np.uint16).astype(np.int64)
    assert order2.size == parts.size == n_pos
    p, P = parts & 0xFF, parts >> 8
    live = order2 >= 0
    want_P = np.minimum(-(-deg.numpy() // cap), 64).clip(min=1)
    heads = np.nonzero(live & (p == 0))[0]
    assert np.array_equal(np.sort(order2[heads]), np.sort(order.numpy()))            # every row once
    by_row = {int(r): int(k) for r, k in zip(order.numpy(), want_P)}
    for h in heads:
        k = by_row[int(order2[h])]
        assert P[h] == k and h // 64 == (h + k - 1) // 64
        assert np.all(order2[h:h + k] == order2[h]) and np.array_equal(p[h:h + k], np.arange(k)) and np.all(P[h:h + k] == k)
    assert int(live.sum()) == int(want_P.sum())
    assert np.all(P[~live] == 1) and np.all(p[~live] == 0)
    # the unsplit rows keep their (descending-degree) order after the hub slots
    n_hub = int((want_P > 1).sum())
    tail = order2[n_pos - (deg.numel() - n_hub):]
    assert np.array_equal(tail, order.numpy()[n_hub:])


def _attention_from_coefficients(x, Mn, u, cn, cd):
    """out_i = (x_i Mn + cn) / (x_i . u + cd): the closed form of difformer.py:18-39 for query == source == x."""
    return (x @ Mn + cn) / (x @ u + cd)[:, None]


@pytest.mark.parametrize("use_weight", [True, False])
def test_wide_coefficients_algebra_matches_the_oracle(use_weight):
    """ops.WideCoefficients (weight-only factors of the closed form at the scripts' widths) + the two per-layer products
    of ops.simple_layer_closed_form_wide, replayed on the CPU in float64, reproduce full_attention_conv(q, k, v, 'simple')
    of the oracle -- every bias term of difformer.py:20-38 rides inside the augmented matrices."""
    import numpy as np
    from difformer_amd.ops import WideCoefficients
    from oracle import difformer_oracle as orc
    g = torch.Generator().manual_seed(3)
    n, C = 500, 20
    D = C
    x = torch.randn(n, C, generator=g, dtype=torch.float64)
    mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64) * 0.3
    Wq, bq, Wk, bk = mk(D, C), mk(D), mk(D, C), mk(D)
    Wv, bv = (mk(D, C), mk(D)) if use_weight else (None, None)
    co = WideCoefficients(Wq, bq, Wk, bk, Wv, bv)
    Gt = torch.zeros(C + 1, C + 1, dtype=torch.float64)
    Gt[:C, :C], Gt[:C, C], Gt[C, :C], Gt[C, C] = x.t() @ x, x.sum(0), x.sum(0), float(n)
    norms = co.S @ Gt.reshape(-1)
    s = torch.rsqrt(norms[0] * norms[1])
    T = Gt @ co.V
    R = co.P @ T
    B, bias = s * R[:C], s * R[C] + T[C]
    out = _attention_from_coefficients(x, B[:, :D], B[:, D], bias[:D], bias[D]).numpy()
    q, k = (x @ Wq.t() + bq).numpy(), (x @ Wk.t() + bk).numpy()
    v = (x @ Wv.t() + bv).numpy() if use_weight else x.numpy()
    assert abs(float(norms[0]) - (q * q).sum()) < 1e-9 * (q * q).sum() and abs(float(norms[1]) - (k * k).sum()) < 1e-9 * (k * k).sum()
    ref = orc.simple_attention(q[:, None, :], k[:, None, :], v[:, None, :])[:, 0, :]
    assert np.abs(out - ref).max() < 1e-10 * np.abs(ref).max()


@pytest.mark.parametrize("C,D,use_weight", [(64, 64, True), (32, 48, True), (24, 24, False)])
def test_narrow_factors_algebra_matches_the_oracle(C, D, use_weight):
    """ops.NarrowFactors (80 x 80 zero-padded, augmented index 64) + the products of csrc/side_chain.hip replayed with
    numpy: T^T = (G~ V~)^T, R = P~ T, coef = [MnT | cn | u | cd] -- against the oracle's simple attention."""
    import numpy as np
    from difformer_amd.ops import NarrowFactors
    from oracle import difformer_oracle as orc
    g = torch.Generator().manual_seed(C + D)
    n, B_, A_ = 300, 80, 64
    x = torch.randn(n, C, generator=g)
    mk = lambda *s: torch.randn(*s, generator=g) * 0.3
    Wq, bq, Wk, bk = mk(D, C), mk(D), mk(D, C), mk(D)
    Wv, bv = (mk(D, C), mk(D)) if use_weight else (None, None)
    f = NarrowFactors(Wq, bq, Wk, bk, Wv, bv)
    pt, vtt = f.pt.double().numpy().reshape(B_, B_), f.vtt.double().numpy().reshape(B_, B_)
    st = f.st.double().numpy().reshape(2, B_ * B_)
    x64 = x.double().numpy()
    gt = np.zeros((B_, B_))
    gt[:C, :C], gt[A_, :C], gt[:C, A_], gt[A_, A_] = x64.T @ x64, x64.sum(0), x64.sum(0), n
    q2, k2 = st @ gt.reshape(-1)
    s = 1.0 / np.sqrt(q2 * k2)
    T = gt @ vtt.T                                   # tile_kk(gt, vtt): D[i][j] = sum_k gt[i][k] vtt[j][k]
    R = pt @ T
    Mn, u = s * R[:C, :D], s * R[:C, A_]
    cn, cd = s * R[A_, :D] + T[A_, :D], s * R[A_, A_] + T[A_, A_]
    out = (x64 @ Mn + cn) / (x64 @ u + cd)[:, None]
    q, k = x64 @ Wq.double().numpy().T + bq.double().numpy(), x64 @ Wk.double().numpy().T + bk.double().numpy()
    v = x64 @ Wv.double().numpy().T + bv.double().numpy() if use_weight else x64
    ref = orc.simple_attention(q[:, None, :], k[:, None, :], v[:, None, :])[:, 0, :]
    assert abs(q2 - (q * q).sum()) < 1e-5 * (q * q).sum()           # the factors are stored in float32
    assert np.abs(out - ref).max() < 1e-5 * np.abs(ref).max()


# ---- training step through the autograd glue against GRADIENTS of the reference (tests/golden/golden_grad.npz) -----------
GRAD = load_golden("grad")


@pytest.mark.parametrize("name", sorted(n for n in GRAD if n.startswith("model/")))
def test_training_step_plumbing_against_reference_gradients(name, fake_backend):
    """main.py:117-131 on the package's modules with the arithmetic on the fake backend: what is checked here is the
    autograd glue (which operator sees which tensor, the scales of the combine, the edge_weight gradient), against the
    reference's own loss / parameter gradients / dx / d edge_weight.  float32 operators: 1e-4 norm-wise."""
    import torch.nn.functional as F
    from conftest import grad_err, grad_scale
    from difformer_amd import DIFFormer
    c = GRAD[name]
    cfg, sd = split_model_case(c)
    kw = {k: cfg[k] for k in ("num_layers", "num_heads", "kernel", "alpha", "use_bn", "use_residual", "use_weight",
                              "use_graph", "graph_weight", "use_source")}
    kw["kernel"] = str(kw["kernel"])
    model = DIFFormer(int(cfg["in_channels"]), int(cfg["hidden_channels"]), int(cfg["out_channels"]), dropout=0.0, **kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.train()
    x = torch.from_numpy(c["x"]).requires_grad_(True)
    ei = torch.from_numpy(c["edge_index"]) if cfg["use_graph"] else None
    w = torch.from_numpy(c["edge_weight"]).requires_grad_(True) if "edge_weight" in c else None
    out = model(x, ei, w)
    idx, y = torch.from_numpy(c["train_idx"]), torch.from_numpy(c["y"])
    if str(c["loss_kind"]) == "bce":
        loss = F.binary_cross_entropy_with_logits(out[idx], y[idx])
    else:
        loss = F.nll_loss(F.log_softmax(out, dim=1)[idx], y[idx])
    loss.backward()
    assert rel_err(out.detach().numpy(), c["out_f64"]) < 1e-4
    assert abs(float(loss.detach()) - float(c["loss_f64"])) < 1e-4 * abs(float(c["loss_f64"]))
    gmax = grad_scale(c)
    assert grad_err(x.grad.numpy(), c["dx_f64"], gmax) < 1e-4
    if w is not None:
        got, ref = w.grad.numpy(), c["dw_f64"]
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        assert grad_err(np.nan_to_num(got), np.nan_to_num(ref), gmax) < 1e-4
    for k, p in model.named_parameters():
        ref = c["grad_f64/" + k]
        got = np.zeros_like(ref) if p.grad is None else p.grad.numpy()
        assert grad_err(got, ref, gmax) < 1e-4, k


def test_parameter_caches_invalidate(fake_backend):
    """the inference caches are keyed on (data_ptr, _version); `.data` writes bypass the version counter, so
    there is an explicit hook, and load_state_dict / _apply / reset_parameters call it.  Parameters made under
    inference_mode (no version counter) are never cached."""
    from difformer_amd import DIFFormer, ops
    torch.manual_seed(0)
    model = DIFFormer(8, 16, 3, num_layers=2, num_heads=2, kernel="sigmoid", use_graph=False).eval()
    x = torch.randn(30, 8)
    with torch.no_grad():
        a = model(x, None)
        assert model.convs[0]._fused_wb is not None
        model.convs[0].Wq.weight.data.mul_(2.0)                 # bypasses _version: stale cache until invalidated
        model.invalidate_caches()
        assert model.convs[0]._fused_wb is None
        b = model(x, None)
        assert not torch.allclose(a, b)
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        model(x, None)
        model.load_state_dict(sd)
        assert model.convs[0]._fused_wb is None
        model(x, None)
        model.double().float()
        assert model.convs[1]._fused_wb is None
    with torch.inference_mode():
        w = torch.randn(4, 4)
        ei = torch.randint(0, 30, (2, 50))
    assert ops.tensor_version(w) == -1 and ops.param_key([w]) is None
    assert ops.csr_cache.get(ei, None, 30) is ops.csr_cache.get(ei, None, 30)      # keyed without a version, no raise


@pytest.mark.parametrize("use_graph,use_weight,graph_weight,use_source,ln,residual,c",
                         [(True, True, -1, False, True, True, 16), (True, False, -1, False, True, True, 16),
                          (True, True, 0.3, True, True, True, 8), (False, True, -1, True, False, True, 12),
                          (True, True, -1, True, True, False, 16), (True, False, 0.6, False, False, False, 8),
                          (False, False, -1, False, True, True, 16)])
def test_closed_form_layer_backward_matches_the_operator_path(use_graph, use_weight, graph_weight, use_source, ln, residual, c,
                                                              fake_backend, monkeypatch):
    """ag._ClosedFormLayer (training through the Gram record: difformer.py:18-39, :107-140 without q, k, v) against the
    q / k / v operator path, whose gradients the golden fixtures of the reference pin: every input and parameter gradient
    of one layer, every flag of DIFFormerConv."""
    from conftest import grad_err
    from difformer_amd import DIFFormerConv
    from difformer_amd import difformer as dmod
    torch.manual_seed(c + int(use_graph) + 2 * int(use_weight))
    n = 70
    conv = DIFFormerConv(c, c, 1, kernel="simple", use_graph=use_graph, use_weight=use_weight, graph_weight=graph_weight,
                         use_source=use_source).train()
    g = torch.Generator().manual_seed(5)
    ei = torch.randint(0, n, (2, 300), generator=g) if use_graph else None
    R = torch.randn(n, c, generator=g)
    base = dict(x=torch.randn(n, c, generator=g), x0=torch.randn(n, c, generator=g), lw=torch.rand(c, generator=g) + 0.5,
                lb=torch.randn(c, generator=g))

    def run(closed):
        monkeypatch.setattr(dmod, "_CLOSED_FORM_TRAINING", closed)
        leaves = {k: v.clone().requires_grad_(True) for k, v in base.items()}
        conv.zero_grad()
        fake_backend.closed_form_calls = 0
        x = leaves["x"]
        out, _, _ = conv._layer(x, x, ei, None, leaves["x0"] if use_source else None, x if residual else None, 0.4,
                                leaves["lw"] if ln else None, leaves["lb"] if ln else None, 1e-5)
        (out * R).sum().backward()
        assert (fake_backend.closed_form_calls > 0) == closed
        grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
        grads.update({k: p.grad.clone() for k, p in conv.named_parameters() if p.grad is not None})
        return out.detach(), grads

    out_c, g_c = run(True)
    out_o, g_o = run(False)
    assert rel_err(out_c.numpy(), out_o.numpy()) < 1e-5
    assert set(g_c) == set(g_o)
    gmax = max(float(v.abs().max()) for v in g_o.values())
    for k in g_o:
        assert grad_err(g_c[k].numpy(), g_o[k].numpy(), gmax) < 1e-4, k


def test_small_graphs_take_the_one_call_gram_and_coefficients(fake_backend, monkeypatch):
    """ops.simple_layer_closed_form: up to GRAM_COEFFS_MAX_ROWS float32 rows, no row shard, no slice-major copy wanted -> the
    backend's gram_coeffs (record and coefficients from one call); beyond the limit the two-call path.  Same result."""
    from difformer_amd import DIFFormerConv, ops
    torch.manual_seed(3)
    conv = DIFFormerConv(16, 16, 1, kernel="simple", use_graph=True).eval()
    g = torch.Generator().manual_seed(4)
    x, ei = torch.randn(90, 16, generator=g), torch.randint(0, 90, (2, 400), generator=g)
    with torch.no_grad():
        fake_backend.gram_coeffs_calls = 0
        out_one = conv._layer(x, x, ei, None, None, x, 0.5, None, None, 1e-5)[0]
        assert fake_backend.gram_coeffs_calls == 1
        monkeypatch.setattr(ops, "GRAM_COEFFS_MAX_ROWS", 10)
        out_two = conv._layer(x, x, ei, None, None, x, 0.5, None, None, 1e-5)[0]
        assert fake_backend.gram_coeffs_calls == 1
    assert torch.allclose(out_one, out_two, rtol=1e-6, atol=1e-6)


def test_wide_linear_dispatch_rule():
    """ops.linear_xwide_covers: float32, 65..416 output features, more than 128 input channels in one product (<= 416) or two
    equal halves (<= 832), multiples of 4, at least LINEAR_XWIDE_MIN_ROWS rows; never under DIFFORMER_EXACT_FP32."""
    from difformer_amd import ops
    n = ops.LINEAR_XWIDE_MIN_ROWS
    ok = lambda rows, ci, co, dt=torch.float32: ops.linear_xwide_covers(torch.empty(rows, ci, dtype=dt), torch.empty(co, ci, dtype=dt))
    assert ok(n, 512, 300) and ok(n, 832, 416) and ok(n, 300, 68) and ok(n, 132, 400)
    assert not ok(n - 1, 512, 300) and not ok(n, 128, 300) and not ok(n, 512, 64) and not ok(n, 512, 420)
    assert not ok(n, 836, 300) and not ok(n, 510, 300) and not ok(n, 420, 300) and not ok(n, 512, 302)
    assert not ok(n, 512, 300, torch.bfloat16)


def test_closed_form_training_keeps_a_temporary_edge_index_alive(fake_backend):
    """`model(x, ei.to(dev))`: the edge tensor is a temporary that dies with the call, while the adjoint CSR of the backward
    pass is built from it on first use -- the autograd node of the record path holds it (as the aggregation's node does)."""
    import gc
    from difformer_amd import DIFFormer
    torch.manual_seed(0)
    model = DIFFormer(6, 8, 3, num_layers=2, kernel="simple", dropout=0.0).train()
    x = torch.randn(40, 6)
    out = model(x, torch.randint(0, 40, (2, 150)).clone())
    gc.collect()
    out.sum().backward()
    assert fake_backend.closed_form_calls > 0 and all(p.grad is not None for p in model.parameters())


def test_constant_edge_weights_build_the_unweighted_csr(fake_backend):
    """`edge_attr = torch.ones(E)` of `--special_treat dense` / knn (spatial-temporal/main.py:99,103), or any constant: the CSR
    is the unweighted one and the constant rides in gcn_scale (ops._CSRCache.get); model output = the oracle's WITH weights."""
    import numpy as np
    from difformer_amd import DIFFormer, gcn_conv, ops
    from oracle import difformer_oracle as orc
    n = 80
    row = torch.arange(n).unsqueeze(1).repeat(1, n).reshape(-1)
    col = torch.arange(n).unsqueeze(0).repeat(n, 1).reshape(-1)
    ei = torch.stack([row, col])                                          # 6,400 entries: above UNIFORM_MIN_EDGES
    x = torch.randn(n, 1, 8)
    ops.csr_cache.UNIFORM_ALWAYS = True                                    # (the check is reserved for sliced-eligible sizes)
    request_finalizer = lambda: setattr(ops.csr_cache, "UNIFORM_ALWAYS", False)
    for const in (1.0, 0.25, float("nan")):
        w = torch.full((ei.shape[1],), const)
        csr = ops.csr_cache.get(ei, w, n, 32)
        assert not csr.weighted and csr.weight_scale == (const if np.isfinite(const) else 0.0)
        ref = orc.gcn_conv(x.double().numpy(), ei.numpy(), w.double().numpy())
        assert np.abs(gcn_conv(x, ei, w).numpy() - ref).max() < 1e-5
    wv = torch.rand(ei.shape[1]) + 0.5
    assert ops.csr_cache.get(ei, wv, n, 32).weighted                      # weights that vary: the weighted CSR
    torch.manual_seed(0)
    for kernel, use_weight in (("simple", False), ("sigmoid", True)):
        model = DIFFormer(6, 8, 3, num_layers=2, kernel=kernel, use_weight=use_weight).eval()
        xin = torch.randn(n, 6)
        w = torch.full((ei.shape[1],), 0.5)
        with torch.no_grad():
            out = model(xin, ei, w)
        cfg = dict(hidden_channels=8, num_layers=2, num_heads=1, kernel=kernel, alpha=0.5, use_bn=True, use_residual=True,
                   use_weight=use_weight, use_graph=True, graph_weight=-1, use_source=False)
        p = {k: v.detach().numpy().astype(np.float64) for k, v in model.state_dict().items()}
        ref = orc.difformer_forward(p, xin.double().numpy(), ei.numpy(), w.double().numpy(), cfg)
        assert np.abs(out.numpy() - ref).max() / np.abs(ref).max() < 1e-4
    # learnable weights initialised to a constant -- an eval pass under no_grad caches the UNWEIGHTED CSR under the
    # weight tensor's key; the grad-enabled call that follows must not take that entry (edge_weight would get no gradient)
    wl = torch.ones(ei.shape[1], requires_grad=True)
    with torch.no_grad():
        assert not ops.csr_cache.get(ei, wl, n, 32).weighted
    assert ops.csr_cache.get(ei, wl, n, 32).weighted
    gcn_conv(x, ei, wl).sum().backward()
    assert wl.grad is not None and float(wl.grad.abs().sum()) > 0
    request_finalizer()
    assert ops.csr_cache.get(ei, torch.full((ei.shape[1],), 2.0), n, 32).weighted    # 80 nodes: not worth the check by default


def test_exact_fp32_switch_reaches_the_backend(fake_backend):
    """ops.set_exact_fp32: returns the previous setting, moves the Python-side width threshold of the closed form and tells the
    backend (whose launchers make the per-kernel choice: dif_set_exact_fp32) -- a backend without the hook is left alone."""
    from difformer_amd import ops
    seen = []
    was = ops.set_exact_fp32(True)                       # the host test backend has no hook: nothing to call
    try:
        assert ops.EXACT_FP32 is True and ops.CLOSED_FORM_WIDE_MIN == 128
        fake_backend.set_exact_fp32 = seen.append
        assert ops.set_exact_fp32(False) is True and ops.EXACT_FP32 is False and ops.CLOSED_FORM_WIDE_MIN == 64
        assert ops.set_exact_fp32(True) is False
        assert seen == [False, True]
    finally:
        if hasattr(fake_backend, "set_exact_fp32"):
            del fake_backend.set_exact_fp32
        ops.set_exact_fp32(was)

"""Kernel-INSTANTIATION coverage: libdifformer_hip.so holds ~400 template instantiations behind ~100 entry
points, and every boolean of the reference's CLI (node classification/parse.py:48-52), every width and both storage types pick
among them.  `profiles/r06_kernel_coverage_before.txt` (rocprofv3 --kernel-trace around `pytest -m gpu`) listed 128 that no test
launched.  This file sweeps the SELECTORS of each family -- feature width / vector alignment / rows per lane group / storage type
/ walk order / 32-bit vs 64-bit row offsets / fused tails -- through the C ABI (the backend's entry points) against the oracle,
so that `scripts/kernel_coverage.py` ends with an empty UNLAUNCHED list (tests/test_kernel_coverage.py holds the tracked list to
the library's symbols).  Tolerances: 1e-4 (float32), 1e-2 (bfloat16 storage) norm-wise, SURVEY.md 8d."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import difformer_oracle as orc

pytestmark = pytest.mark.gpu

TOL, BF16_TOL = 1e-4, 1e-2


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return torch.device("cuda:0")


def _store(t, dt):
    """-> (tensor in storage type dt, its exact float64 numpy value)."""
    s = t.to(dt)
    return s, s.to(torch.float64).numpy()


def _graph(n, deg, seed, zipf=False):
    g = torch.Generator().manual_seed(seed)
    e = n * deg
    if zipf:
        w = (torch.arange(n, dtype=torch.float64) + max(n // 120, 1)) ** -0.75
        cdf = torch.cumsum(w, 0) / w.sum()
        perm = torch.randperm(n, generator=g)
        a, b = (perm[torch.searchsorted(cdf, torch.rand(e, generator=g, dtype=torch.float64)).clamp_(max=n - 1)] for _ in range(2))
    else:
        a, b = torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)
    return torch.stack([torch.cat([a, torch.arange(n)]), torch.cat([b, torch.arange(n)])])


# ================================================================== a3: the gather SpMM kernels (csrc/gcn_spmm.hip)
# spmm_wave_row / spmm_group_row <G, W, T>: G lanes x W elements cover a feature row -- vector rows: (1..64, 4); rows that are not
# 4-element aligned: (4..64, 1); a wave per row from 16 entries per row on, else a lane group per row.
_ROW_WIDTHS = [4, 8, 16, 32, 64, 128, 260, 3, 7, 13, 30, 50]


@pytest.mark.parametrize("deg", [24, 4])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("F", _ROW_WIDTHS)
def test_spmm_row_kernels_every_width(F, dt, deg, dev):
    from difformer_amd import ops
    be = ops.get_backend()
    n = 3000
    ei = _graph(n, deg, F + deg)
    csr = ops.GraphCSR.build(ei.to(dev), None, n, 1)
    g = torch.Generator().manual_seed(F)
    x, x64 = _store(torch.randn(n, F, generator=g), dt)
    a, a64 = _store(torch.randn(n, F, generator=g), dt)
    out = be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, csr.nnz, x.to(dev), 0, n, a.to(dev), 0.5, 2.0, None, None)
    ref = 2.0 * orc.gcn_conv(x64[:, None, :], ei.numpy(), None)[:, 0, :] + 0.5 * a64
    assert out.dtype == dt and rel_err(out.float().cpu().numpy(), ref) < (TOL if dt == torch.float32 else BF16_TOL)


# spmm_blocked_kernel<G, W, ..., WIDE, ORD, T>: source-blocked CSR; (16 | 32 | 64, 4) by width, bfloat16 rows of 8-element
# multiples (8 | 16 | 32, 8); ORD = degree-ordered walk (row_order); WIDE = 64-bit row offsets once n_nodes * ld reaches 2^31
# elements (here: rows that are column slices of a [32768, 65536] buffer -- 8.6 GB in float32, allocated once).
@pytest.fixture(scope="module")
def wide_buffers(dev):
    bufs = {}

    def get(dt):
        if dt not in bufs:
            bufs[dt] = torch.empty((32768, 65536), dtype=dt, device=dev)
        return bufs[dt]
    yield get
    bufs.clear()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("ordered", [False, True])
@pytest.mark.parametrize("F,dt", [(64, torch.float32), (128, torch.float32), (256, torch.float32),
                                  (64, torch.bfloat16), (128, torch.bfloat16), (256, torch.bfloat16),
                                  (60, torch.bfloat16), (100, torch.bfloat16), (204, torch.bfloat16)])
def test_spmm_blocked_kernel_every_instantiation(F, dt, ordered, wide, dev, wide_buffers):
    from difformer_amd import ops
    be = ops.get_backend()
    n = 32768
    ei = _graph(n, 12, F, zipf=True)
    csr = ops.GraphCSR.build(ei.to(dev), None, n, 3)
    g = torch.Generator().manual_seed(F + 1)
    x, x64 = _store(torch.randn(n, F, generator=g), dt)
    a, a64 = _store(torch.randn(n, F, generator=g), dt)
    if wide:
        xd = wide_buffers(dt)[:, 128: 128 + F]              # ld = 65536: n * ld = 2^31 elements
        xd.copy_(x.to(dev))
        assert xd.stride(0) * n >= 2 ** 31
    else:
        xd = x.to(dev)
    order = csr.row_order(0, n) if ordered else None
    assert order is None or order[0] is not None
    out = be.spmm(csr.rowptr, csr.blkptr, csr.n_blocks, csr.src, csr.val, n, csr.nnz, xd, 0, n, a.to(dev), 0.5, 2.0, None, order)
    ref = 2.0 * orc.gcn_conv(x64[:, None, :], ei.numpy(), None)[:, 0, :] + 0.5 * a64
    assert rel_err(out.float().cpu().numpy(), ref) < (TOL if dt == torch.float32 else BF16_TOL)


# ================================================================== a4 / a5 tail (csrc/layer_tail.hip)
def _tail_ref(conv64, x0, prev, alpha, lw, lb, relu):
    z = conv64.mean(axis=1)                                  # difformer.py:137
    if x0 is not None:
        z = z + x0                                           # :139-140
    if prev is not None:
        z = alpha * z + (1.0 - alpha) * prev                 # :200-201
    if lw is not None:
        z = orc.layer_norm(z, lw, lb)                        # :202-203
    return np.maximum(z, 0.0) if relu else z


# layer_tail_vec_kernel<G, V, T>: D / 4 lanes per row (G = 1 .. 64; 65+ quads: two per lane); layer_tail_generic_kernel: any other D
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D,H", [(4, 1), (8, 2), (16, 1), (32, 3), (64, 1), (128, 1), (256, 2), (300, 1), (6, 2), (70, 1)])
def test_layer_tail_every_width(D, H, dt, dev):
    from difformer_amd import ops
    be = ops.get_backend()
    n = 1500
    g = torch.Generator().manual_seed(D * 10 + H)
    conv, c64 = _store(torch.randn(n, H, D, generator=g), dt)
    x0, x064 = _store(torch.randn(n, D, generator=g), dt)
    prev, p64 = _store(torch.randn(n, D, generator=g), dt)
    lw, lw64 = _store(torch.rand(D, generator=g) + 0.5, dt)
    lb, lb64 = _store(torch.randn(D, generator=g), dt)
    tol = TOL if dt == torch.float32 else BF16_TOL
    for use_x0, use_prev, ln, relu in ((True, True, True, False), (False, True, False, True), (False, False, True, True)):
        out = be.layer_tail(conv.to(dev), x0.to(dev) if use_x0 else None, prev.to(dev) if use_prev else None, 0.4,
                            lw.to(dev) if ln else None, lb.to(dev) if ln else None, 1e-5, relu)
        ref = _tail_ref(c64, x064 if use_x0 else None, p64 if use_prev else None, 0.4, lw64 if ln else None, lb64 if ln else None, relu)
        assert rel_err(out.float().cpu().numpy(), ref) < tol, (use_x0, use_prev, ln, relu)


# layer_tail_vec_kernel<G, V, float, MIX>: the tail of the wide closed form -- numerator / denominator columns of one row GEMM,
# + the aggregated values (A x) Wv^T + (A 1) bv^T, + x0, residual, LayerNorm (dif_layer_tail_mix_f32)
@pytest.mark.parametrize("D", [64, 128, 256, 300])
def test_layer_tail_mix_every_width(D, dev):
    from difformer_amd import ops
    be = ops.get_backend()
    n = 2000
    g = torch.Generator().manual_seed(D)
    ldz = ((D + 1 + 3) // 4) * 4
    Z = torch.randn(n, ldz, generator=g)
    Z[:, D] = torch.rand(n, generator=g) + 1.0               # denominators
    add, rs, bv = torch.randn(n, D, generator=g), torch.rand(n, generator=g), torch.randn(D, generator=g)
    x0, prev = torch.randn(n, D, generator=g), torch.randn(n, D, generator=g)
    lw, lb = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g)
    d = lambda t: t.to(dev)
    out = be.layer_tail_mix(d(Z), D, D, 0.7, d(add), 1.3, d(rs), d(bv), d(x0), d(prev), 0.4, d(lw), d(lb), 1e-5)
    z = 0.7 * Z[:, :D].double() / Z[:, D:D + 1].double() + 1.3 * (add.double() + rs.double()[:, None] * bv.double()[None, :])
    z = 0.4 * (z + x0.double()) + 0.6 * prev.double()
    ref = orc.layer_norm(z.numpy(), lw.double().numpy(), lb.double().numpy())
    assert rel_err(out.cpu().numpy(), ref) < TOL
    out2 = be.layer_tail_mix(d(Z), D, None, 1.0, None, 1.0, None, None, None, None, 0.5, None, None, 1e-5, relu=True)
    assert rel_err(out2.cpu().numpy(), np.maximum(Z[:, :D].double().numpy(), 0.0)) < TOL


# ================================================================== a1 at every head shape (csrc/simple_attn.hip)
# simple_reduce_kernel<VEC, T> / simple_apply_kernel<VEC, SINGLE, T> / simple_apply_wide_kernel<VEC, T>: 4-element aligned rows or
# not; one 64 x 64 tile per head (SINGLE), 65 .. 512 columns (wide: s KtV^T resident in LDS), beyond 512 (tiled apply)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,h,m,d", [(700, 2, 10, 10), (900, 1, 64, 64), (800, 1, 70, 70), (600, 1, 128, 72), (500, 1, 516, 516),
                                     (400, 1, 517, 130), (5000, 1, 128, 128), (300, 3, 33, 65)])
def test_simple_attention_every_head_shape(n, h, m, d, dt, dev):
    from difformer_amd import full_attention_conv
    g = torch.Generator().manual_seed(n + m)
    q, q64 = _store(torch.randn(n, h, m, generator=g), dt)
    k, k64 = _store(torch.randn(n, h, m, generator=g), dt)
    v, v64 = _store(torch.randn(n, h, d, generator=g) + 0.2, dt)
    out = full_attention_conv(q.to(dev), k.to(dev), v.to(dev), "simple")
    ref = orc.simple_attention(q64, k64, v64)
    assert out.dtype == dt and rel_err(out.float().cpu().numpy(), ref) < (TOL if dt == torch.float32 else BF16_TOL)


def test_gram_record_of_rows_that_are_not_vector_aligned(dev):
    """simple_reduce_kernel<false, float, SYM>: dif_gram_sym_f32 at a width that is not a multiple of 4."""
    from difformer_amd import ops
    be = ops.get_backend()
    n, c = 3000, 70
    x = (torch.randn(n, c, generator=torch.Generator().manual_seed(2)) + 0.3).to(dev)
    rec = be.gram_sym(x)
    x64 = x.double().cpu().numpy()
    got = rec[: c * c].cpu().numpy().reshape(c, c).astype(np.float64)
    blk = np.arange(c) // 64
    got = np.where(blk[:, None] <= blk[None, :], got, got.T)
    assert rel_err(got, x64.T @ x64) < 1e-5 and rel_err(rec[c * c: c * c + c].cpu().numpy(), x64.sum(0)) < 1e-5


# ================================================================== Linear layers (csrc/skinny_linear.hip)
def _linear_ref(x64, w64, b64, lw, lb, relu):
    y = x64 @ w64.T + b64
    if lw is not None:
        y = orc.layer_norm(y, lw, lb)
    return np.maximum(y, 0.0) if relu else y


# skinny_linear_kernel<KQ, T>: KQ = ceil(C_in / 16) = 1 .. 8 input steps; bfloat16 rows into MORE than 64 outputs take it too
# (up to 64 outputs they run on the bf16 matrix core: skinny_linear_bf16_kernel)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c_in", [16, 30, 48, 64, 65, 96, 112, 128])
def test_skinny_linear_every_input_width(c_in, dt, dev):
    from difformer_amd import ops
    be = ops.get_backend()
    n, c_out = 2500, 112
    g = torch.Generator().manual_seed(c_in)
    x, x64 = _store(torch.randn(n, c_in, generator=g), dt)
    w, w64 = _store(torch.randn(c_out, c_in, generator=g) / c_in ** 0.5, dt)
    b, b64 = _store(torch.randn(c_out, generator=g), dt)
    lw, lw64 = _store(torch.rand(c_out, generator=g) + 0.5, dt)
    lb, lb64 = _store(torch.randn(c_out, generator=g), dt)
    tol = TOL if dt == torch.float32 else BF16_TOL
    out = be.linear(x.to(dev), w.to(dev), b.to(dev))
    assert rel_err(out.float().cpu().numpy(), _linear_ref(x64, w64, b64, None, None, False)) < tol
    out = be.linear(x.to(dev), w.to(dev), b.to(dev), lw.to(dev), lb.to(dev), 1e-5, True)
    assert rel_err(out.float().cpu().numpy(), _linear_ref(x64, w64, b64, lw64, lb64, True)) < tol


def test_long_rows_into_a_narrow_layer_on_the_fp32_chain(dev):
    """long_linear_kernel<float>: 512 -> 64 on >= 16,384 aligned rows under ops.set_exact_fp32(True) (the default build takes the
    split-bfloat16 kernels, fewer / unaligned rows the K-split kernel)."""
    from difformer_amd import ops
    be = ops.get_backend()
    n, c_in, c_out = 20000, 512, 64
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, c_in, generator=g)
    w, b = torch.randn(c_out, c_in, generator=g) / c_in ** 0.5, torch.randn(c_out, generator=g)
    lw, lb = torch.rand(c_out, generator=g) + 0.5, torch.randn(c_out, generator=g)
    was = ops.set_exact_fp32(True)
    try:
        out = be.linear(x.to(dev), w.to(dev), b.to(dev), lw.to(dev), lb.to(dev), 1e-5, True)
    finally:
        ops.set_exact_fp32(was)
    ref = _linear_ref(x.double().numpy(), w.double().numpy(), b.double().numpy(), lw.double().numpy(), lb.double().numpy(), True)
    assert rel_err(out.cpu().numpy(), ref) < 1e-5


# ================================================================== the closed-form layer kernel (csrc/simple_layer.hip)
# simple_layer_kernel<EXACT, GRAPH_W, NEXT, T, HEAD, GATHER, SPLIT>: 64 x 64 layers or narrower; aggregated values with / without
# Wv; float32 / bfloat16 activations; the last layer with the output Linear inside (HEAD); the aggregation inside the kernel for
# sparse graphs (GATHER) or ahead of it (dense graphs: the sliced product / an SpMM); split-bfloat16 or fp32-MFMA products.
# Model-level sweep of exactly those switches (two layers: one hidden, one HEAD) against the float64 oracle.
def _model_forward(dev, n, f_in, hidden, classes, layers, kernel, graph, use_weight, dt, exact=False, heads=1, **flags):
    from difformer_amd import DIFFormer, ops
    torch.manual_seed(n + hidden + layers)
    model = DIFFormer(f_in, hidden, classes, num_layers=layers, num_heads=heads, kernel=kernel, use_graph=graph is not None,
                      use_weight=use_weight, **flags).eval().to(dt)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(n, f_in, generator=g).to(dt)
    cfg = dict(hidden_channels=hidden, num_layers=layers, num_heads=heads, kernel=kernel, alpha=0.5, use_bn=True, use_residual=True,
               use_weight=use_weight, use_graph=graph is not None, graph_weight=flags.get("graph_weight", -1),
               use_source=flags.get("use_source", False))
    p = {k: v.double().numpy() for k, v in model.state_dict().items()}
    ref = orc.difformer_forward(p, x.double().numpy(), None if graph is None else graph.numpy(), None, cfg)
    model = model.to(dev)
    was = ops.set_exact_fp32(True) if exact else None
    try:
        if exact:
            model.invalidate_caches()
        with torch.no_grad():
            out = model(x.to(dev), None if graph is None else graph.to(dev))
    finally:
        if exact:
            ops.set_exact_fp32(was)
            model.invalidate_caches()
    assert out.dtype == dt
    return rel_err(out.float().cpu().numpy(), ref)


@pytest.mark.parametrize("use_weight", [True, False])
@pytest.mark.parametrize("graph_kind", ["none", "sparse", "dense"])
@pytest.mark.parametrize("hidden", [32, 64])
@pytest.mark.parametrize("mode", ["f32", "bf16", "f32-exact"])
def test_closed_form_layer_kernel_every_switch(mode, hidden, graph_kind, use_weight, dev):
    n = 9000
    graph = None if graph_kind == "none" else _graph(n, 5 if graph_kind == "sparse" else 60, hidden)
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    err = _model_forward(dev, n, 24, hidden, 7, 2, "simple", graph, use_weight, dt, exact=(mode == "f32-exact"))
    assert err < (2 * BF16_TOL if dt == torch.bfloat16 else TOL)


def test_input_linear_in_two_halves_at_hidden_400(dev):
    """simple_layer_xwide_kernel<8, 13>: 512 input channels into 400 features (image and text/run.sh:10: stl10, hidden 400) run
    as two accumulating products over 256 channels each (from 16,384 rows: ops.LINEAR_XWIDE_MIN_ROWS)."""
    assert _model_forward(dev, 17000, 512, 400, 10, 1, "simple", None, True, torch.float32) < TOL


@pytest.mark.parametrize("dense_graph", [False, True])
def test_last_layer_with_the_output_linear_inside_on_the_fp32_chain(dense_graph, dev):
    """simple_layer_kernel<EXACT, GRAPH_W, ..., HEAD, SPLIT = false>: the model keeps the output Linear out of the last layer
    kernel under ops.set_exact_fp32(True) (difformer.py:572 of the package); a C-ABI caller of dif_simple_layer_head_f32 does not
    have to -- the layer with `head` handed over directly, against the oracle's layer + Linear (difformer.py:113-145, :208)."""
    from difformer_amd import DIFFormerConv, ops
    n, c, co = 9000, 64, 10
    torch.manual_seed(3)
    g = torch.Generator().manual_seed(4)
    conv = DIFFormerConv(c, c, 1, kernel="simple", use_graph=dense_graph, use_weight=True).to(dev).eval()
    x = torch.randn(n, c, generator=g)
    lw, lb = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    hw, hb = torch.randn(co, c, generator=g) * 0.2, torch.randn(co, generator=g)
    ei = _graph(n, 60, 5) if dense_graph else None
    xd = x.to(dev)
    was = ops.set_exact_fp32(True)
    try:
        carry = {"head": (hw.to(dev), hb.to(dev))}
        with torch.no_grad():
            out, names = _launched(lambda: conv._layer(xd, xd, None if ei is None else ei.to(dev), None, None, xd, 0.4, lw.to(dev),
                                                       lb.to(dev), 1e-5, carry=carry)[0])
    finally:
        ops.set_exact_fp32(was)
    assert carry.get("head_done") and out.shape == (n, co)
    p = {"c." + k: v.detach().cpu().double().numpy() for k, v in conv.state_dict().items()}
    cfg = dict(num_heads=1, kernel="simple", use_graph=dense_graph, use_weight=True, graph_weight=-1, use_source=False, hidden_channels=c)
    x64 = x.double().numpy()
    z = orc.difformer_conv(p, "c.", x64, x64, None if ei is None else ei.numpy(), None, None, cfg)
    z = orc.layer_norm(0.4 * z + 0.6 * x64, lw.double().numpy(), lb.double().numpy())
    ref = z @ hw.double().numpy().T + hb.double().numpy()
    assert rel_err(out.cpu().numpy(), ref) < 1e-5


@pytest.mark.parametrize("n,c,d", [(4099, 64, 64), (3000, 32, 48)])
def test_closed_form_attention_backward_lean_variant(n, c, d, dev):
    """closed_form_attn_bwd_kernel<EXACT, SUMS = false>: dif_closed_form_attn_bwd_f32 with sums = NULL, against the variant that
    leaves the partial sums (itself held to float64 autograd in tests/test_gpu_closed_form.py)."""
    from difformer_amd import ops
    be = ops.get_backend()
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, c, generator=g).to(dev)
    coef = (torch.randn(d * c + d + c + 4, generator=g) * 0.2).to(dev)
    coef[d * c + d + c] = 25.0
    dd = torch.randn(n, d, generator=g).to(dev)
    d_num, d_den, dx, _, _, _ = be.closed_form_attn_backward(x, coef, d, dd)
    o_num, o_den, o_dx = torch.empty_like(d_num), torch.empty_like(d_den), torch.empty_like(dx)
    rc = be.lib.dif_closed_form_attn_bwd_f32(x.data_ptr(), c, n, c, d, coef.data_ptr(), dd.data_ptr(), d, None, 0, o_num.data_ptr(),
                                             o_den.data_ptr(), o_dx.data_ptr(), c, None, None, torch.cuda.current_stream(dev).cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(o_num, d_num) and torch.equal(o_den, d_den) and torch.equal(o_dx, dx)
    x64, cf = x.double().cpu(), coef.double().cpu()
    MnT, cn, u, cd = cf[: d * c].view(d, c), cf[d * c: d * c + d], cf[d * c + d: d * c + d + c], cf[d * c + d + c]
    den = x64 @ u + cd
    rn = dd.double().cpu() / den[:, None]
    assert rel_err(o_num.cpu().numpy(), rn.numpy()) < 1e-5


@pytest.mark.parametrize("c,h,d", [(20, 3, 10), (64, 1, 64), (30, 2, 33)])
def test_project_reduce_bf16_rows(c, h, d, dev):
    """project_reduce_kernel<EXACT, bf16>: projections fused with stage 1 of the simple kernel on bfloat16 rows, widths that are /
    are not 4-element aligned."""
    from difformer_amd import ops
    n = 1500
    g = torch.Generator().manual_seed(c + d)
    x, x64 = _store(torch.randn(n, c, generator=g), torch.bfloat16)
    W = [_store(torch.randn(h * d, c, generator=g) / c ** 0.5, torch.bfloat16) for _ in range(3)]
    b = [_store(torch.randn(h * d, generator=g) * 0.3, torch.bfloat16) for _ in range(3)]
    q, v, rec = ops.get_backend().project_reduce(x.to(dev), W[0][0].to(dev), b[0][0].to(dev), W[1][0].to(dev), b[1][0].to(dev),
                                                 W[2][0].to(dev), b[2][0].to(dev), h, d)
    q64, k64, v64 = ((x64 @ W[i][1].T + b[i][1]).reshape(n, h, d) for i in range(3))
    assert q.dtype == torch.bfloat16 and rel_err(q.float().cpu().numpy(), q64) < BF16_TOL and rel_err(v.float().cpu().numpy(), v64) < BF16_TOL
    ktv = np.einsum("lhm,lhd->hmd", k64, v64)
    assert rel_err(rec[: ktv.size].cpu().numpy().reshape(ktv.shape), ktv) < BF16_TOL


# ================================================================== feature-sliced product at every round count (csrc/gcn_sliced.hip)
# sliced_spmm_kernel<R>: R = rounds of 64-row slots per wave, 1 .. 9 by the number of destination rows (C4 runs 9)
@pytest.mark.parametrize("rounds", [1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_sliced_product_every_round_count(rounds, dev):
    from difformer_amd import gcn_conv, ops
    be = ops.get_backend()
    n = None
    for cand in range(8192, 140000, 512):                    # smallest node count whose plan has the wanted rounds
        plan = be.sliced_plan(cand, cand, 64)
        if plan is not None and plan[5] == rounds:
            n = cand
            break
    assert n is not None, f"no node count below 140,000 gives {rounds} rounds"
    ei = _graph(n, 50, rounds)
    x = torch.randn(n, 1, 64, generator=torch.Generator().manual_seed(rounds))
    out, names = _launched(lambda: gcn_conv(x.to(dev), ei.to(dev), None))
    assert "dif_sliced_spmm_f32" in names
    ref = orc.gcn_conv(x.double().numpy(), ei.numpy(), None)
    assert rel_err(out.cpu().numpy(), ref) < TOL


def _launched(fn):
    from difformer_amd import ops
    be = ops.get_backend()
    be.kernel_events = {}
    try:
        r = fn()
    finally:
        names, be.kernel_events = set(be.kernel_events), None
    return r, names


# ================================================================== a2 on the fp32 chain at widths beyond its fragment registers
# sigmoid_attn_kernel<VEC, QREG = false, SEG, T>: more than 64 columns per head with the query fragments re-read per key tile --
# float32 beyond the plane kernels' 512 columns (or under ops.set_exact_fp32), bfloat16 storage, the batched (v2) form
@pytest.mark.parametrize("m,dt", [(517, torch.float32), (520, torch.float32), (70, torch.bfloat16), (72, torch.bfloat16)])
def test_sigmoid_attention_generic_kernel_wide_heads(m, dt, dev):
    from difformer_amd import full_attention_conv
    n, l = 150, 210
    g = torch.Generator().manual_seed(m)
    q, q64 = _store(torch.randn(n, 1, m, generator=g) * (3.0 / m ** 0.5), dt)
    k, k64 = _store(torch.randn(l, 1, m, generator=g) * 0.5, dt)
    v, v64 = _store(torch.randn(l, 1, m, generator=g), dt)
    out = full_attention_conv(q.to(dev), k.to(dev), v.to(dev), "sigmoid")
    assert rel_err(out.float().cpu().numpy(), orc.sigmoid_attention(q64, k64, v64)) < (TOL if dt == torch.float32 else BF16_TOL)


@pytest.mark.parametrize("m", [70, 72, 30])
def test_batched_sigmoid_attention_odd_and_wide_heads(m, dev):
    """sigmoid_attn_kernel<VEC, QREG, SEG = true> and (M <= 64) sigmoid_bwd_kernel<MODE, VEC = false, SEG = true>: the v2 model's
    sigmoid attention (physical particle/difformer-v2.py:113-135) at widths that are not 4-element aligned / beyond 64."""
    from difformer_amd import autograd_ops as ag, ops
    from oracle import difformer_oracle_grad as og
    rng = np.random.default_rng(m)
    n_nodes = rng.integers(1, 9, size=40)
    n = int(n_nodes.sum())
    mk = lambda w, s: torch.from_numpy((rng.standard_normal((n, 1, w)) * s).astype(np.float32))
    q, k, v, go = mk(m, 0.4), mk(m, 0.4), mk(m, 1.0), mk(m, 1.0)
    layout = ops.BatchLayout(torch.from_numpy(n_nodes), dev)
    leaves = [t.to(dev).requires_grad_(True) for t in (q, k, v)]
    out = ag.batched_attention(*leaves, layout, "sigmoid")
    out.backward(go.to(dev))
    l64 = [t.double().requires_grad_(True) for t in (q, k, v)]
    ref = og.v2_sigmoid_attention(*l64, torch.from_numpy(n_nodes))
    ref.backward(go.double())
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < TOL
    for got, want in zip(leaves, l64):
        assert rel_err(got.grad.cpu().numpy(), want.grad.numpy()) < TOL


def test_sigmoid_backward_split_operands_on_unaligned_heads():
    """sigmoid_bwd_kernel<MODE, VEC = false, SEG = false, SPLIT = true>: the opt-in split-bfloat16 backward
    (DIFFORMER_SIGMOID_BWD_SPLIT=1, read at library load: child process) at a head width that is not 4-element aligned."""
    import json, os, subprocess, sys
    code = '''
import json, numpy as np, torch
from difformer_amd import autograd_ops as ag
from oracle import difformer_oracle as orc
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(4)
q, k, v, go = (torch.randn(n, 1, 30, generator=g) * s for n, s in ((300, 0.4), (411, 0.4), (411, 1.0), (300, 1.0)))
leaves = [t.to(dev).requires_grad_(True) for t in (q, k, v)]
ag.sigmoid_attention(*leaves).backward(go.to(dev))
ref = orc.sigmoid_attention_grad_blocked(q.double().numpy(), k.double().numpy(), v.double().numpy(), go.double().numpy())
gm = max(np.abs(r).max() for r in ref)
print(json.dumps([float(np.abs(t.grad.cpu().double().numpy() - r).max() / gm) for t, r in zip(leaves, ref)]))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, DIFFORMER_SIGMOID_BWD_SPLIT="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    errs = json.loads(r.stdout.strip().splitlines()[-1])
    assert max(errs) < TOL, errs


# ================================================================== f4: batched simple attention (csrc/batched_attn.hip)
# batched_simple_kernel<MT, VEC, WAVES, RAW>: M <= 64 / <= 128 / wider; vector rows or not; one wave or four per (graph, head, tile)
# (few graphs of many rows); RAW = the three launches of the backward
@pytest.mark.parametrize("B,mx,M,D", [(6, 500, 50, 30), (64, 20, 97, 35), (5, 400, 97, 35), (5, 400, 128, 128), (64, 20, 128, 128),
                                      (64, 20, 197, 37), (4, 300, 197, 37), (64, 20, 200, 200), (4, 300, 200, 200), (6, 500, 64, 64),
                                      (64, 20, 50, 30)])
def test_batched_simple_attention_every_shape(B, mx, M, D, dev):
    from difformer_amd import autograd_ops as ag, ops
    from oracle import difformer_oracle_grad as og
    rng = np.random.default_rng(B + M)
    n_nodes = rng.integers(max(mx // 2, 1), mx + 1, size=B)
    n = int(n_nodes.sum())
    mk = lambda w: torch.from_numpy(rng.standard_normal((n, 1, w)).astype(np.float32))
    q, k, v, go = mk(M), mk(M), mk(D), mk(D)
    layout = ops.BatchLayout(torch.from_numpy(n_nodes), dev)
    leaves = [t.to(dev).requires_grad_(True) for t in (q, k, v)]
    out = ag.batched_attention(*leaves, layout, "simple")
    out.backward(go.to(dev))
    l64 = [t.double().requires_grad_(True) for t in (q, k, v)]
    ref = og.v2_simple_attention(*l64, torch.from_numpy(n_nodes))
    ref.backward(go.double())
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < TOL
    gmax = max(float(t.grad.abs().max()) for t in l64)
    for got, want, nm in zip(leaves, l64, "qkv"):
        assert float((got.grad.cpu().double() - want.grad).abs().max()) / gmax < TOL, nm


# ================================================================== f3: layer-tail backward at every lane-group width
@pytest.mark.parametrize("hidden", [4, 8, 16, 32, 64, 128, 256, 260, 300, 400, 512])
def test_layer_tail_backward_every_width(hidden, dev):
    """layer_tail_bwd_kernel<G, V>: D / 4 = 1 .. 64 lanes per row (hidden 8: G = 2), two vectors per lane from 257 columns (hidden 300 /
    400: image and text/run.sh) -- against float64 autograd of
    the tail expression (difformer.py:137-140, 200-203)."""
    from difformer_amd import ops
    be = ops.get_backend()
    n, H = 5000, 2
    g = torch.Generator().manual_seed(hidden)
    conv, x0, prev = torch.randn(n, H, hidden, generator=g), torch.randn(n, hidden, generator=g), torch.randn(n, hidden, generator=g)
    lw, lb, go = torch.rand(hidden, generator=g) + 0.5, torch.randn(hidden, generator=g), torch.randn(n, hidden, generator=g)
    d = lambda t: t.to(dev)
    got = be.layer_tail_bwd(d(conv), d(x0), d(prev), 0.4, d(lw), d(lb), 1e-5, False, d(go), (True, True, True, True))
    leaves = [t.double().requires_grad_(True) for t in (conv, x0, prev, lw, lb)]
    z = 0.4 * (leaves[0].mean(dim=1) + leaves[1]) + 0.6 * leaves[2]
    torch.nn.functional.layer_norm(z, (hidden,), leaves[3], leaves[4], 1e-5).backward(go.double())
    assert got is not None
    refs = [leaves[0].grad, leaves[1].grad, leaves[2].grad, leaves[3].grad, leaves[4].grad]
    for a, b, nm in zip(got, refs, ("d_conv", "d_x0", "d_prev", "d_ln_weight", "d_ln_bias")):
        assert rel_err(a.cpu().numpy(), b.numpy()) < TOL, nm


# ================================================================== both sides of the host's dispatch thresholds
# The host picks kernel families by thresholds fitted on synthetic graphs (ops.SLICED_MIN_DEGREE = 48 entries per row,
# ops.SLICED_MIN_ROWS = 8,192 nodes, ops.MIX_THRESHOLD = 2.5, 16 entries per row for a wave per row): a graph just below and
# one just above each must give the oracle's numbers, and must take the families the threshold is meant to separate.
def _regular_graph(n, per_row, seed):
    """exactly `per_row` entries in every row: per_row - 1 random sources + the self loop (main.py:75-76)."""
    g = torch.Generator().manual_seed(seed)
    dst = torch.arange(n).repeat_interleave(per_row - 1)
    src = torch.randint(0, n, (n * (per_row - 1),), generator=g)
    return torch.stack([torch.cat([src, torch.arange(n)]), torch.cat([dst, torch.arange(n)])])


def _conv_both_ways(dev, n, ei, width=64):
    from difformer_amd import gcn_conv
    x = torch.randn(n, 1, width, generator=torch.Generator().manual_seed(n))
    out, names = _launched(lambda: gcn_conv(x.to(dev), ei.to(dev), None))
    ref = orc.gcn_conv(x.double().numpy(), ei.numpy(), None)
    return rel_err(out.cpu().numpy(), ref), names


@pytest.mark.parametrize("per_row,sliced", [(47, False), (48, True), (49, True)])
def test_threshold_entries_per_row_of_the_sliced_product(per_row, sliced, dev):
    err, names = _conv_both_ways(dev, 9000, _regular_graph(9000, per_row, per_row))
    assert err < TOL and ("dif_sliced_spmm_f32" in names) == sliced and ("dif_gcn_spmm_f32" in names) != sliced, names


@pytest.mark.parametrize("n,sliced", [(8191, False), (8192, True), (8193, True)])
def test_threshold_node_count_of_the_sliced_product(n, sliced, dev):
    err, names = _conv_both_ways(dev, n, _regular_graph(n, 60, n))
    assert err < TOL and ("dif_sliced_spmm_f32" in names) == sliced, names


@pytest.mark.parametrize("per_row", [15, 16, 17])
def test_threshold_wave_per_row_or_lane_group_per_row(per_row, dev):
    """dif_gcn_spmm: from 16 entries per row a whole wave walks a row, below a lane group does (csrc/gcn_spmm.hip)."""
    err, names = _conv_both_ways(dev, 5000, _regular_graph(5000, per_row, per_row))
    assert err < TOL and "dif_gcn_spmm_f32" in names


@pytest.mark.parametrize("inside,mixed", [(0.70, False), (0.80, True)])
def test_threshold_mixed_node_order_for_community_graphs(inside, mixed, dev):
    """ops.MIX_THRESHOLD: the mean over rows of (largest per-tile share of the row's entries) x tiles.  Three source tiles, a
    fraction `inside` of every row's entries from the row's own tile, the rest uniform: score = 1 + 2 inside = 2.4 / 2.6 -- the
    model runs the second graph in a mixed node order (ops.MixedGraph) and both give the oracle's logits."""
    from difformer_amd import DIFFormer, ops
    n, per_row = 30000, 56
    be = ops.get_backend()
    plan = be.sliced_plan(n, n, 64)
    tiles, tile_rows = int(plan[7]), int(plan[6])
    assert tiles == 3
    g = torch.Generator().manual_seed(int(inside * 100))
    dst = torch.arange(n).repeat_interleave(per_row - 1)
    own = (dst // tile_rows) * tile_rows
    width = torch.minimum(torch.full_like(own, tile_rows), n - own)
    local = own + (torch.rand(dst.shape, generator=g) * width).long()
    anywhere = torch.randint(0, n, dst.shape, generator=g)
    src = torch.where(torch.rand(dst.shape, generator=g) < inside, local, anywhere)
    ei = torch.stack([torch.cat([src, torch.arange(n)]), torch.cat([dst, torch.arange(n)])])
    eid = ei.to(dev)
    mix = ops.mix_cache.get(eid, n, 64)
    assert (mix is not None) == mixed
    torch.manual_seed(1)
    model = DIFFormer(16, 64, 5, num_layers=2, kernel="simple", use_graph=True).to(dev).eval()
    x = torch.randn(n, 16, generator=g)
    with torch.no_grad():
        out = model(x.to(dev), eid)
    cfg = dict(hidden_channels=64, num_layers=2, num_heads=1, kernel="simple", alpha=0.5, use_bn=True, use_residual=True,
               use_weight=True, use_graph=True, graph_weight=-1, use_source=False)
    p = {k: v.detach().cpu().double().numpy() for k, v in model.state_dict().items()}
    ref = orc.difformer_forward(p, x.double().numpy(), ei.numpy(), None, cfg)
    assert rel_err(out.cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("hidden", [64, 68, 128, 132])
def test_threshold_widths_of_the_closed_form_families(hidden, dev):
    """ops.CLOSED_FORM_WIDE_MIN (64): up to 64 columns the record kernels of csrc/simple_layer.hip, 65..128 the one-pass kernel of
    csrc/simple_layer_wide.hip, beyond it csrc/simple_layer_xwide.hip -- each side of both borders against the oracle."""
    assert _model_forward(dev, 9000, 24, hidden, 7, 2, "simple", _graph(9000, 6, hidden), True, torch.float32) < TOL


@pytest.mark.parametrize("width", [64, 65, 512, 513])
def test_threshold_widths_of_the_sigmoid_families(width, dev):
    """a2: <= 64 columns per head the register-fragment kernels (csrc/sigmoid_attn.hip), 65..512 the plane kernels
    (csrc/sigmoid_wide.hip), beyond 512 the generic fp32 kernel -- forward and gradient on each side of both borders."""
    from difformer_amd import autograd_ops as ag
    g = torch.Generator().manual_seed(width)
    q, k, v, go = (torch.randn(n, 1, width, generator=g) * s for n, s in ((300, 3.0 / width ** 0.5), (260, 0.5), (260, 1.0), (300, 1.0)))
    leaves = [t.to(dev).requires_grad_(True) for t in (q, k, v)]
    out = ag.sigmoid_attention(*leaves)
    out.backward(go.to(dev))
    q64, k64, v64, g64 = (t.double().numpy() for t in (q, k, v, go))
    assert rel_err(out.detach().cpu().numpy(), orc.sigmoid_attention(q64, k64, v64)) < TOL
    refs = orc.sigmoid_attention_grad_blocked(q64, k64, v64, g64)
    gmax = max(np.abs(r).max() for r in refs)
    for got, want in zip(leaves, refs):
        assert float(np.abs(got.grad.cpu().double().numpy() - want).max()) / gmax < TOL

"""Backward parity of the HIP path (training step, loss / gradients).

Two yardsticks, both derived from the reference:
  * tests/golden/golden_grad.npz -- gradients of the reference itself (float64 run), small cases;
  * oracle/difformer_oracle_grad.py -- float64 CPU restatement pinned to those fixtures (tests/test_oracle_grad_golden.py),
    for sizes the fixtures cannot hold: a graph that takes the feature-sliced product and its adjoint inside a training
    step, and Cora size.
Tolerance: 1e-4 norm-wise per tensor (conftest.grad_err: relative to the tensor's own largest entry, floored at 1e-6 of the
step's largest gradient entry).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import grad_err, grad_scale, load_golden, rel_err, split_model_case
from oracle import difformer_oracle_grad as og

pytestmark = pytest.mark.gpu

TOL = 1e-4
GRAD = load_golden("grad")


def cases(prefix):
    return sorted(n for n in GRAD if n.startswith(prefix + "/"))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return torch.device("cuda:0")


def t(a, dev, grad=False):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return x.requires_grad_(True) if grad else x


def nan_err(got, ref, gmax=None):
    got, ref = np.asarray(got), np.asarray(ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), "NaN positions differ from the reference's"
    got, ref = np.nan_to_num(got), np.nan_to_num(ref)
    return rel_err(got, ref) if gmax is None else grad_err(got, ref, gmax)


# ------------------------------------------------------------------ a1 / a2
@pytest.mark.parametrize("name", cases("attn"))
def test_full_attention_conv_gradients_golden(name, dev):
    from difformer_amd import full_attention_conv
    c = GRAD[name]
    q, k, v = (t(c[a], dev, True) for a in "qkv")
    out = full_attention_conv(q, k, v, str(c["kernel"]))
    out.backward(t(c["g"], dev))
    assert rel_err(out.detach().cpu().numpy(), c["out_f64"]) < TOL
    gmax = max(np.abs(c[f"d{a}_f64"]).max() for a in "qkv")
    for x, a in ((q, "dq"), (k, "dk"), (v, "dv")):
        assert grad_err(x.grad.cpu().numpy(), c[a + "_f64"], gmax) < TOL, a


# ------------------------------------------------------------------ a3
@pytest.mark.parametrize("name", cases("gcn"))
def test_gcn_conv_gradients_golden(name, dev):
    """dx through the adjoint product, d edge_weight through csrc/gcn_edge_grad.hip (NaN exactly where the reference's is)."""
    from difformer_amd import gcn_conv
    c = GRAD[name]
    x = t(c["x"], dev, True)
    ei = t(c["edge_index"], dev)
    w = t(c["edge_weight"], dev, True) if "edge_weight" in c else None
    out = gcn_conv(x, ei, w)
    out.backward(t(c["g"], dev))
    assert rel_err(out.detach().cpu().numpy(), c["out_f64"]) < TOL
    assert rel_err(x.grad.cpu().numpy(), c["dx_f64"]) < TOL
    if w is not None:
        assert nan_err(w.grad.cpu().numpy(), c["dw_f64"]) < TOL


@pytest.mark.parametrize("n,e,f,iso", [(5000, 200000, 64, 7), (3000, 40000, 10, 0), (20000, 900000, 128, 50)])
def test_edge_weight_gradient_vs_oracle(n, e, f, iso, dev):
    """Bigger graphs (vector and scalar loads, several heads' worth of columns) against float64 autograd of the oracle."""
    from difformer_amd import gcn_conv
    g = torch.Generator().manual_seed(n)
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n - iso, (e,), generator=g)])
    h = 2 if f % 2 == 0 else 1
    x = torch.randn(n, h, f // h, generator=g)
    w = torch.rand(e, generator=g) + 0.1
    go = torch.randn(n, h, f // h, generator=g)
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    og.gcn_conv(x64, ei, w64).backward(go.double())
    xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    gcn_conv(xd, ei.to(dev), wd).backward(go.to(dev))
    assert rel_err(xd.grad.cpu().numpy(), x64.grad.numpy()) < TOL
    assert nan_err(wd.grad.cpu().numpy(), w64.grad.numpy()) < TOL


# ------------------------------------------------------------------ a4 / a5: training step of main.py:117-131
def _build(c, dev):
    from difformer_amd import DIFFormer
    cfg, sd = split_model_case(c)
    kw = {k: cfg[k] for k in ("num_layers", "num_heads", "kernel", "alpha", "use_bn", "use_residual", "use_weight",
                              "use_graph", "graph_weight", "use_source")}
    kw["kernel"] = str(kw["kernel"])
    model = DIFFormer(int(cfg["in_channels"]), int(cfg["hidden_channels"]), int(cfg["out_channels"]), dropout=0.0, **kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return model.to(dev).train(), cfg


def _loss(out, y, idx, kind):
    if kind == "bce":
        return F.binary_cross_entropy_with_logits(out[idx], y[idx].to(out.dtype))
    return F.nll_loss(F.log_softmax(out, dim=1)[idx], y[idx])


@pytest.mark.parametrize("name", cases("model"))
def test_training_step_gradients_golden(name, dev):
    """loss, every parameter gradient, dx (and d edge_weight) of one training step against the reference's own."""
    c = GRAD[name]
    model, cfg = _build(c, dev)
    x = t(c["x"], dev, True)
    ei = t(c["edge_index"], dev) if cfg["use_graph"] else None
    w = t(c["edge_weight"], dev, True) if "edge_weight" in c else None
    out = model(x, ei, w)
    loss = _loss(out, t(c["y"], dev), t(c["train_idx"], dev), str(c["loss_kind"]))
    loss.backward()
    assert rel_err(out.detach().cpu().numpy(), c["out_f64"]) < TOL
    assert abs(float(loss.detach()) - float(c["loss_f64"])) < TOL * abs(float(c["loss_f64"]))
    gmax = grad_scale(c)
    assert grad_err(x.grad.cpu().numpy(), c["dx_f64"], gmax) < TOL
    if w is not None:
        assert nan_err(w.grad.cpu().numpy(), c["dw_f64"], gmax) < TOL
    for k, p in model.named_parameters():
        ref = c["grad_f64/" + k]
        got = np.zeros_like(ref) if p.grad is None else p.grad.cpu().numpy()
        assert np.isfinite(got).all(), k
        assert grad_err(got, ref, gmax) < TOL, k


def _oracle_step(model, x, ei, cfg, y, idx, kind="nll", w=None):
    """float64 CPU oracle of the same step -> (out, loss, {param: grad}, dx)."""
    p = og.leaves({k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
    x64 = x.detach().cpu().double().requires_grad_(True)
    out = og.difformer_forward(p, x64, None if ei is None else ei.cpu(), w, cfg)
    loss = og.training_loss(out, y.cpu(), idx.cpu(), kind)
    loss.backward()
    grads = {k: (None if v.grad is None else v.grad.numpy()) for k, v in p.items()}      # None: unused (LayerNorms of use_bn=False)
    return out.detach().numpy(), float(loss.detach()), grads, x64.grad.numpy()


def _check_step(model, x, ei, cfg, y, idx, launched=None, kind="nll"):
    out = model(x, ei)
    loss = _loss(out, y, idx, kind)
    loss.backward()
    r_out, r_loss, r_grads, r_dx = _oracle_step(model, x, ei, cfg, y, idx, kind)
    assert rel_err(out.detach().cpu().numpy(), r_out) < TOL
    assert abs(float(loss.detach()) - r_loss) < TOL * abs(r_loss)
    gmax = max(float(np.abs(v).max()) for v in r_grads.values() if v is not None)
    if x.grad is not None:
        assert grad_err(x.grad.cpu().numpy(), r_dx, gmax) < TOL
    for k, prm in model.named_parameters():
        if r_grads[k] is None:
            assert prm.grad is None or not prm.grad.any(), k
            continue
        assert prm.grad is not None and torch.isfinite(prm.grad).all(), k
        assert grad_err(prm.grad.cpu().numpy(), r_grads[k], gmax) < TOL, k


@pytest.mark.parametrize("layers,heads,skew", [(2, 1, False), (3, 1, True), (2, 2, False)])
def test_training_step_on_a_graph_that_takes_the_sliced_product(layers, heads, skew, dev):
    """>= 8,192 nodes and >= 48 entries per row: the forward aggregation is the feature-sliced product and its gradient the
    same kernel over the transposed CSR (the headline kernels inside loss.backward()).  skew: hub rows (split positions)."""
    from difformer_amd import DIFFormer, ops
    n, per = 9000, 56
    g = torch.Generator().manual_seed(11 + layers)
    if skew:
        wgt = 1.0 / torch.arange(1, n + 1, dtype=torch.float64) ** 0.8
        dst = torch.multinomial(wgt, n * per // 2, replacement=True, generator=g)
    else:
        dst = torch.randint(0, n, (n * per // 2,), generator=g)
    src = torch.randint(0, n, (n * per // 2,), generator=g)
    ei = torch.cat([torch.stack([src, dst]), torch.stack([dst, src]), torch.arange(n).repeat(2, 1)], dim=1)
    torch.manual_seed(5)
    model = DIFFormer(12, 64 // heads, 9, num_layers=layers, num_heads=heads, kernel="simple", dropout=0.0).to(dev).train()
    cfg = dict(hidden_channels=64 // heads, num_layers=layers, num_heads=heads, kernel="simple", alpha=0.5, use_bn=True,
               use_residual=True, use_weight=True, use_graph=True, graph_weight=-1, use_source=False)
    x = torch.randn(n, 12, generator=g).to(dev).requires_grad_(True)
    y = torch.randint(0, 9, (n,), generator=g).to(dev)
    idx = torch.randperm(n, generator=g)[: n // 2].to(dev)
    be = ops.get_backend()
    be.kernel_events = {}
    try:
        _check_step(model, x, ei.to(dev), cfg, y, idx)
        launched = set(be.kernel_events)
    finally:
        be.kernel_events = None
    assert "dif_sliced_spmm_f32" in launched, launched


@pytest.mark.parametrize("hidden,use_graph,use_weight,graph_weight,use_source,use_bn,use_residual,deg",
                         [(64, True, True, 0.3, True, True, True, 4), (64, True, False, -1, False, True, True, 60),
                          (32, True, True, -1, True, False, False, 6), (48, False, True, -1, True, True, True, 0),
                          (64, True, False, 0.7, True, False, True, 3), (16, False, False, -1, False, True, False, 0)])
def test_training_step_through_the_record_every_flag(hidden, use_graph, use_weight, graph_weight, use_source, use_bn,
                                                     use_residual, deg, dev):
    """Layers that train through the Gram record (autograd_ops._ClosedFormLayer: one head, `simple`, up to 64 columns) with
    every flag of DIFFormerConv / DIFFormer (difformer.py:107-140, :195-207), on sparse graphs (gather product) and on one
    that takes the sliced product, against float64 autograd of the oracle; the record path is the one that ran."""
    from difformer_amd import DIFFormer, ops
    n = 9000 if deg >= 48 else 3001
    g = torch.Generator().manual_seed(hidden + deg)
    ei = None
    if use_graph:
        pairs = torch.randint(0, n, (2, n * deg // 2), generator=g)
        ei = torch.cat([pairs, pairs.flip(0), torch.arange(n).repeat(2, 1)], dim=1).to(dev)
    torch.manual_seed(7)
    kw = dict(num_layers=2, num_heads=1, kernel="simple", alpha=0.4, use_bn=use_bn, use_residual=use_residual,
              use_weight=use_weight, use_graph=use_graph, graph_weight=graph_weight, use_source=use_source)
    model = DIFFormer(20, hidden, 6, dropout=0.0, **kw).to(dev).train()
    cfg = dict(hidden_channels=hidden, **kw)
    x = torch.randn(n, 20, generator=g).to(dev).requires_grad_(True)
    y = torch.randint(0, 6, (n,), generator=g).to(dev)
    idx = torch.randperm(n, generator=g)[: n // 3].to(dev)
    be = ops.get_backend()
    be.kernel_events = {}
    try:
        _check_step(model, x, ei, cfg, y, idx)
        launched = set(be.kernel_events)
    finally:
        be.kernel_events = None
    assert (({"dif_gram_f32", "dif_simple_coeffs_f32"} <= launched or "dif_gram_coeffs_f32" in launched) and
            "dif_simple_apply_f32" not in launched), launched
    if deg >= 48:
        assert "dif_sliced_spmm_f32" in launched


@pytest.mark.parametrize("hidden,heads,n,deg", [(128, 1, 6000, 6), (128, 1, 9000, 60), (300, 1, 2000, 5), (64, 2, 5000, 8)])
def test_training_step_with_wide_heads(hidden, heads, n, deg, dev):
    """--hidden_channels 128 / 300 (run.sh's large-graph and image lines): a training step whose attention gradient runs on
    the HIP backward for heads wider than 64 (128-column prep, wide row-GEMM; no tensor-op re-derivation), against float64
    autograd of the oracle."""
    from difformer_amd import DIFFormer, ops
    g = torch.Generator().manual_seed(hidden + n)
    pairs = torch.randint(0, n, (2, n * deg // 2), generator=g)
    ei = torch.cat([pairs, pairs.flip(0), torch.arange(n).repeat(2, 1)], dim=1).to(dev)
    torch.manual_seed(9)
    kw = dict(num_layers=2, num_heads=heads, kernel="simple", alpha=0.5, use_bn=True, use_residual=True, use_weight=True,
              use_graph=True, graph_weight=-1, use_source=False)
    model = DIFFormer(30, hidden, 5, dropout=0.0, **kw).to(dev).train()
    cfg = dict(hidden_channels=hidden, **kw)
    x = torch.randn(n, 30, generator=g).to(dev).requires_grad_(True)
    y = torch.randint(0, 5, (n,), generator=g).to(dev)
    idx = torch.randperm(n, generator=g)[: n // 2].to(dev)
    be = ops.get_backend()
    be.kernel_events = {}
    try:
        _check_step(model, x, ei, cfg, y, idx)
        launched = set(be.kernel_events)
    finally:
        be.kernel_events = None
    assert {"dif_simple_bwd_prep_f32", "dif_rowgemm_f32"} <= launched, launched


def test_training_step_at_cora_size(dev):
    """BASELINE config C1 as a training step (main.py:117-131 on Cora: 2,708 nodes, 1,433 features, 7 classes)."""
    from difformer_amd import DIFFormer
    n, f_in = 2708, 1433
    g = torch.Generator().manual_seed(0)
    x = torch.rand(n, f_in, generator=g)
    x = (x / x.sum(dim=1, keepdim=True))
    pairs = torch.randint(0, n, (2, 5278), generator=g)
    ei = torch.cat([pairs, pairs.flip(0), torch.arange(n).repeat(2, 1)], dim=1)
    torch.manual_seed(123)
    model = DIFFormer(f_in, 64, 7, num_layers=2, kernel="simple", dropout=0.0).to(dev).train()
    cfg = dict(hidden_channels=64, num_layers=2, num_heads=1, kernel="simple", alpha=0.5, use_bn=True, use_residual=True,
               use_weight=True, use_graph=True, graph_weight=-1, use_source=False)
    y = torch.randint(0, 7, (n,), generator=g).to(dev)
    idx = torch.randperm(n, generator=g)[:140].to(dev)
    _check_step(model, x.to(dev), ei.to(dev), cfg, y, idx)


def test_training_step_sigmoid_at_cora_size(dev):
    """BASELINE config C2 (DIFFormer-a) as a training step: the sigmoid backward kernels at N = 2,708."""
    from difformer_amd import DIFFormer
    n, f_in = 2708, 200
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, f_in, generator=g)
    pairs = torch.randint(0, n, (2, 5278), generator=g)
    ei = torch.cat([pairs, pairs.flip(0), torch.arange(n).repeat(2, 1)], dim=1)
    torch.manual_seed(124)
    model = DIFFormer(f_in, 64, 7, num_layers=2, kernel="sigmoid", dropout=0.0).to(dev).train()
    cfg = dict(hidden_channels=64, num_layers=2, num_heads=1, kernel="sigmoid", alpha=0.5, use_bn=True, use_residual=True,
               use_weight=True, use_graph=True, graph_weight=-1, use_source=False)
    y = torch.randint(0, 7, (n,), generator=g).to(dev)
    idx = torch.randperm(n, generator=g)[:140].to(dev)
    _check_step(model, x.to(dev), ei.to(dev), cfg, y, idx)


# ------------------------------------------------------------------ f4: physical particle/difformer-v2.py
@pytest.mark.parametrize("name", cases("v2attn"))
def test_v2_attention_gradients_golden(name, dev):
    from difformer_amd.difformer_v2 import TransConv
    c = GRAD[name]
    h, d = c["q"].shape[1:]
    conv = TransConv(d, d, num_heads=h, kernel=str(c["kernel"])).to(dev)
    q, k, v = (t(c[a], dev, True) for a in "qkv")
    out = conv.full_attention(q, k, v, str(c["kernel"]), torch.from_numpy(c["n_nodes"]))
    out.backward(t(c["g"], dev))
    assert rel_err(out.detach().cpu().numpy(), c["out_f64"]) < TOL
    gmax = max(np.abs(c[f"d{a}_f64"]).max() for a in "qkv")
    for x, a in ((q, "dq"), (k, "dk"), (v, "dv")):
        assert grad_err(x.grad.cpu().numpy(), c[a + "_f64"], gmax) < TOL, a


@pytest.mark.parametrize("name", cases("v2model"))
def test_v2_training_step_gradients_golden(name, dev):
    from difformer_amd.difformer_v2 import DIFFormer_v2
    c = GRAD[name]
    cfg, sd = split_model_case(c)
    kw = {k: cfg[k] for k in ("num_layers", "kernel", "alpha", "use_bn", "use_residual", "use_weight", "use_graph",
                              "graph_weight")}
    kw["kernel"] = str(kw["kernel"])
    hid = int(cfg["hidden_channels"])
    model = DIFFormer_v2(int(cfg["in_channels"]), hid, hid, dropout=0.0, **kw)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model = model.to(dev).train()
    x = t(c["x"], dev, True)
    out = model(x, t(c["edge_index"], dev) if cfg["use_graph"] else None, torch.from_numpy(c["n_nodes"]))
    loss = F.mse_loss(out, t(c["target"], dev))
    loss.backward()
    assert rel_err(out.detach().cpu().numpy(), c["out_f64"]) < TOL
    assert abs(float(loss.detach()) - float(c["loss_f64"])) < TOL * abs(float(c["loss_f64"]))
    gmax = grad_scale(c)
    assert grad_err(x.grad.cpu().numpy(), c["dx_f64"], gmax) < TOL
    for k, p in model.named_parameters():
        ref = c["grad_f64/" + k]
        got = np.zeros_like(ref) if p.grad is None else p.grad.cpu().numpy()
        assert grad_err(got, ref, gmax) < TOL, k


@pytest.mark.parametrize("n_graphs,max_n,h,d", [(300, 40, 1, 64), (70, 25, 2, 16), (1, 50, 1, 32), (40, 1, 1, 64), (513, 9, 1, 48)])
def test_v2_sigmoid_attention_backward_kernel_vs_oracle(n_graphs, max_n, h, d, dev):
    """The batched (per position, across graphs) sigmoid attention of physical particle/difformer-v2.py:113-135: forward with
    the full denominators kept + the sweep kernels of csrc/sigmoid_attn_bwd.hip with the position groups' row mapping,
    against float64 autograd of the oracle -- ragged batches, one graph, one node per graph, more graphs than a 32-row group."""
    from difformer_amd import ops
    from difformer_amd.difformer_v2 import TransConv
    g = torch.Generator().manual_seed(n_graphs * 100 + max_n)
    n_nodes = torch.randint(1, max_n + 1, (n_graphs,), generator=g)
    n = int(n_nodes.sum())
    q, k, v, go = (torch.randn(n, h, d, generator=g) for _ in range(4))
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    ref = og.v2_sigmoid_attention(q64, k64, v64, n_nodes.tolist())
    ref.backward(go.double())
    conv = TransConv(d, d, num_heads=h, kernel="sigmoid").to(dev)
    qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    be = ops.get_backend()
    be.kernel_events = {}
    try:
        out = conv.full_attention(qd, kd, vd, "sigmoid", n_nodes)
        out.backward(go.to(dev))
        launched = set(be.kernel_events)
    finally:
        be.kernel_events = None
    assert "dif_batched_sigmoid_attn_bwd_f32" in launched
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < TOL
    gmax = max(float(t.grad.abs().max()) for t in (q64, k64, v64))
    # floor 1e-2: with a single graph every position group has ONE node, the weight is 1 - 1e-9 and dq, dk = O(1e-3) come
    # out of (g.v - delta), which cancels to 1e-9 of its operands: float32 leaves ~2e-6 there -- the float32 run of the
    # reference's own expression is off by exactly as much (measured 1.97e-6 against this kernel's 1.59e-6)
    for got, want, name in ((qd, q64, "dq"), (kd, k64, "dk"), (vd, v64, "dv")):
        assert grad_err(got.grad.cpu().numpy(), want.grad.numpy(), gmax, floor=1e-2) < TOL, name


@pytest.mark.parametrize("kernel,hidden", [("simple", 64), ("sigmoid", 32)])
def test_graphed_training_replays_forward_and_backward(kernel, hidden, dev):
    """difformer_amd.graphed_training: the training forward and its backward of main.py:117-131 as two hipGraphs -- three
    optimiser steps give the losses, outputs and parameters of the kernel-by-kernel run; eval calls keep the plain forward."""
    import copy
    import difformer_amd
    from difformer_amd import DIFFormer
    n, f_in, classes = 2708, 96, 7
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, f_in, generator=g).to(dev)
    pairs = torch.randint(0, n, (2, 5278), generator=g)
    ei = torch.cat([pairs, pairs.flip(0), torch.arange(n).repeat(2, 1)], dim=1).to(dev)
    y = torch.randint(0, classes, (n,), generator=g).to(dev)
    idx = torch.randperm(n, generator=g)[:300].to(dev)
    torch.manual_seed(11)
    eager = DIFFormer(f_in, hidden, classes, num_layers=2, kernel=kernel, dropout=0.0).to(dev).train()
    twin = copy.deepcopy(eager)
    graphed = difformer_amd.graphed_training(twin, x, ei)
    assert graphed.training
    opts = [torch.optim.SGD(m.parameters(), lr=0.05) for m in (eager, graphed)]
    for step in range(3):
        losses = []
        for m, opt in zip((eager, graphed), opts):
            opt.zero_grad(set_to_none=True)
            out = m(x, ei)
            loss = F.nll_loss(F.log_softmax(out, dim=1)[idx], y[idx])
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        assert abs(losses[0] - losses[1]) < 1e-5 * abs(losses[0]), (step, losses)
    for (k, a), (_, b) in zip(eager.named_parameters(), graphed.named_parameters()):
        assert rel_err(b.detach().cpu().numpy(), a.detach().cpu().numpy()) < 1e-5, k
    eager.eval()
    graphed.eval()
    with torch.no_grad():
        assert rel_err(graphed(x, ei).cpu().numpy(), eager(x, ei).cpu().numpy()) < 1e-5


def test_whole_training_step_as_one_graph(dev):
    """difformer_amd.GraphedTrainStep: forward, loss, backward and Adam of main.py:117-131 in ONE hipGraph; the warm-up
    steps and three replays end on the parameters of the same number of kernel-by-kernel steps."""
    import copy
    import difformer_amd
    from difformer_amd import DIFFormer
    n, f_in, classes = 2708, 64, 7
    g = torch.Generator().manual_seed(4)
    x = torch.randn(n, f_in, generator=g).to(dev)
    pairs = torch.randint(0, n, (2, 5278), generator=g)
    ei = torch.cat([pairs, pairs.flip(0), torch.arange(n).repeat(2, 1)], dim=1).to(dev)
    y = torch.randint(0, classes, (n,), generator=g).to(dev)
    idx = torch.randperm(n, generator=g)[:300].to(dev)
    loss_fn = lambda out: F.nll_loss(F.log_softmax(out, dim=1)[idx], y[idx])
    torch.manual_seed(12)
    eager = DIFFormer(f_in, 64, classes, num_layers=2, kernel="simple", dropout=0.0).to(dev).train()
    twin = copy.deepcopy(eager)
    opt_e = torch.optim.Adam(eager.parameters(), lr=1e-3)
    opt_g = torch.optim.Adam(twin.parameters(), lr=1e-3, capturable=True)
    step = difformer_amd.GraphedTrainStep(twin, opt_g, loss_fn, x, ei, warmup=3)      # 3 warm-up steps run; the capture does not
    losses_g = [float(step()) for _ in range(3)]
    losses_e = []
    for _ in range(3 + 3):
        opt_e.zero_grad(set_to_none=True)
        loss = loss_fn(eager(x, ei))
        loss.backward()
        opt_e.step()
        losses_e.append(float(loss.detach()))
    assert max(abs(a - b) / abs(b) for a, b in zip(losses_g, losses_e[-3:])) < 1e-4, (losses_g, losses_e)
    for (k, a), (_, b) in zip(eager.named_parameters(), twin.named_parameters()):
        assert rel_err(b.detach().cpu().numpy(), a.detach().cpu().numpy()) < 1e-3, k


@pytest.mark.parametrize("exact", [False, True])
def test_deepest_sigmoid_script_eight_layers(exact, dev):
    """`node classification/run.sh:10`: Cora, DIFFormer-a, --num_layers 8 --hidden_channels 64 --kernel sigmoid --use_graph --use_bn
    --use_residual (the deepest script had no parity case).  Inference forward (default build: split-bfloat16 operands in
    the attention; exact: every product on the fp32 matrix core) and one training step, against the float64 oracle."""
    from difformer_amd import DIFFormer, ops
    torch.manual_seed(8)
    n, f_in, c = 2708, 1433, 7
    g = torch.Generator().manual_seed(9)
    x = torch.rand(n, f_in, generator=g)
    x = x / x.sum(dim=1, keepdim=True)
    pairs = torch.randint(0, n, (2, 5278), generator=g)
    ei = torch.cat([pairs, pairs.flip(0), torch.arange(n).repeat(2, 1)], dim=1)
    cfg = dict(hidden_channels=64, num_layers=8, num_heads=1, kernel="sigmoid", alpha=0.5, use_bn=True, use_residual=True,
               use_weight=False, use_graph=True, graph_weight=-1, use_source=False)
    model = DIFFormer(f_in, 64, c, num_layers=8, alpha=0.5, dropout=0.0, num_heads=1, kernel="sigmoid", use_bn=True, use_residual=True,
                      use_graph=True, use_weight=False).to(dev)
    y, idx = torch.randint(0, c, (n,), generator=g), torch.randperm(n, generator=g)[:140]
    was = ops.set_exact_fp32(exact)
    try:
        model.invalidate_caches()
        model.eval()
        with torch.no_grad():
            out_eval = model(x.to(dev), ei.to(dev))
        model.train()
        out = model(x.to(dev), ei.to(dev))
        loss = F.nll_loss(torch.log_softmax(out, dim=1)[idx.to(dev)], y.to(dev)[idx.to(dev)])
        loss.backward()
    finally:
        ops.set_exact_fp32(was)
        model.invalidate_caches()
    pl = og.leaves({k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
    oref = og.difformer_forward(pl, x.double(), ei, None, cfg)
    lref = og.training_loss(oref, y, idx)
    lref.backward()
    assert rel_err(out_eval.cpu().numpy(), oref.detach().numpy()) < TOL and rel_err(out.detach().cpu().numpy(), oref.detach().numpy()) < TOL
    assert abs(float(loss.detach()) - float(lref.detach())) < TOL * abs(float(lref.detach()))
    gmax = max(float(v.grad.abs().max()) for v in pl.values())
    errs = {k: grad_err(p.grad.cpu().numpy(), pl[k].grad.numpy(), gmax) for k, p in model.named_parameters()}
    assert max(errs.values()) < TOL, [(f"{e:.2e}", k) for e, k in sorted(((e, k) for k, e in errs.items()), reverse=True)[:5]]


@pytest.mark.parametrize("hidden,use_graph,use_weight,use_source,n", [(128, True, True, False, 6000), (128, False, True, False, 4000),
                                                                     (128, True, False, True, 3000), (128, True, True, True, 2500)])
def test_training_through_the_record_at_128_columns(hidden, use_graph, use_weight, use_source, n, dev):
    """node classification/run.sh:42-44 trains Pokec batches at hidden 128 (main-batch.py:135-142): one training step through the
    Gram record at that width (ag._ClosedFormLayerWide, OPT-IN: neither pass forms q, k, v) -- loss, dx and every parameter gradient
    against float64 autograd of the oracle, and the same step on the operator path (DIFFORMER_CLOSED_FORM_TRAINING_WIDE off)."""
    from difformer_amd import DIFFormer, autograd_ops as ag, difformer as dmod, ops
    torch.manual_seed(hidden + n)
    f_in, c, layers = 20, 5, 2
    g = torch.Generator().manual_seed(hidden)
    cfg = dict(hidden_channels=hidden, num_layers=layers, num_heads=1, kernel="simple", alpha=0.5, use_bn=True, use_residual=True,
               use_weight=use_weight, use_graph=use_graph, graph_weight=0.3 if use_source else -1, use_source=use_source)
    model = DIFFormer(f_in, hidden, c, num_layers=layers, kernel="simple", dropout=0.0, use_graph=use_graph, use_weight=use_weight,
                      graph_weight=cfg["graph_weight"], use_source=use_source).to(dev).train()
    x = torch.randn(n, f_in, generator=g)
    ei = torch.cat([torch.randint(0, n, (2, 6 * n), generator=g), torch.arange(n).repeat(2, 1)], dim=1) if use_graph else None
    y, idx = torch.randint(0, c, (n,), generator=g), torch.randperm(n, generator=g)[: n // 2]

    def step():
        model.zero_grad()
        xd = x.to(dev).requires_grad_(True)
        out = model(xd, None if ei is None else ei.to(dev))
        loss = F.nll_loss(torch.log_softmax(out, dim=1)[idx.to(dev)], y.to(dev)[idx.to(dev)])
        loss.backward()
        return loss, xd.grad, {k: p.grad.clone() for k, p in model.named_parameters()}
    be = ops.get_backend()
    was, dmod._CLOSED_FORM_TRAINING_WIDE = dmod._CLOSED_FORM_TRAINING_WIDE, True                 # opt-in path (slower than the operator path)
    be.kernel_events = {}
    try:
        loss, dx, grads = step()
    finally:
        names, be.kernel_events = set(be.kernel_events), None
        dmod._CLOSED_FORM_TRAINING_WIDE = was
    assert "dif_simple_layer_f32" in names and "dif_simple_apply_f32" not in names             # the record path, not q / k / v
    pl = og.leaves({k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
    x64 = x.double().requires_grad_(True)
    lref = og.training_loss(og.difformer_forward(pl, x64, ei, None, cfg), y, idx)
    lref.backward()
    assert abs(float(loss.detach()) - float(lref.detach())) < TOL * abs(float(lref.detach()))
    assert rel_err(dx.cpu().numpy(), x64.grad.numpy()) < TOL
    gmax = max(float(v.grad.abs().max()) for v in pl.values())
    errs = {k: grad_err(v.cpu().numpy(), pl[k].grad.numpy(), gmax) for k, v in grads.items()}
    assert max(errs.values()) < TOL, [(f"{e:.2e}", k) for e, k in sorted(((e, k) for k, e in errs.items()), reverse=True)[:5]]
    # the operator path (the default at these widths) gives the same step
    assert not dmod._CLOSED_FORM_TRAINING_WIDE
    loss2, dx2, grads2 = step()
    assert abs(float(loss2.detach()) - float(loss.detach())) < 1e-5 * abs(float(loss.detach()))
    assert max(grad_err(grads[k].cpu().numpy(), grads2[k].cpu().numpy(), gmax) for k in grads) < TOL

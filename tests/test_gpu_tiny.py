"""The whole-model kernels for tiny graphs (csrc/tiny_model.hip, difformer_amd/tiny.py) through the C ABI: the graph
preparation against dif_csr_build (integers exact, values bit-exact), forward and backward against float64 autograd of the
oracle for every constructor flag, widths 1..8, 1..3,000 nodes, both kernels, dropout with the kernel's own uniforms replayed
in the oracle, and the fixtures of the reference's own gradients that fall in its range."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import grad_err, load_golden, rel_err, split_model_case
from oracle import difformer_oracle_grad as og

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return torch.device("cuda:0")


def _graph(n, e, seed, isolated=0):
    g = torch.Generator().manual_seed(seed)
    row = torch.randint(0, n, (e,), generator=g)
    col = torch.randint(0, max(n - isolated, 1), (e,), generator=g)
    if e >= 20:
        row[: e // 10], col[: e // 10] = row[e // 10: 2 * (e // 10)], col[e // 10: 2 * (e // 10)]      # duplicates
    return torch.stack([row, col]).long()


@pytest.mark.parametrize("n,e,weighted,iso", [(20, 102, True, 0), (129, 2100, True, 3), (1068, 27000, True, 0), (1068, 65535, False, 10),
                                               (1, 3, False, 0), (4096, 5000, True, 900), (50, 0, False, 0)])
def test_graph_build_matches_the_general_csr(n, e, weighted, iso, dev):
    from difformer_amd import ops, tiny
    ei = _graph(n, e, seed=n + e, isolated=iso).to(dev)
    w = (torch.rand(e, generator=torch.Generator().manual_seed(e)) * 3 + 0.05).to(dev) if weighted else None
    g = tiny.build_graph(ei, w, n)
    csr = ops.GraphCSR.build(ei, w, n, 1)
    adj = ops.GraphCSR.build(ei, w, n, 1, transpose=True)
    tiny._poll_status(wait=True)                                                                 # no bad index
    rowptr, src, val, rowptr_t, dst_t, val_t = g.tensors()
    assert torch.equal(rowptr, csr.rowptr) and torch.equal(src, csr.src[:e])
    assert torch.equal(val.view(torch.int32), csr.val[:e].view(torch.int32))                    # bit-exact values
    assert torch.equal(rowptr_t, adj.rowptr) and torch.equal(dst_t, adj.src[:e])
    assert torch.equal(val_t.view(torch.int32), adj.val[:e].view(torch.int32))


def test_graph_build_flags_bad_indices(dev):
    from difformer_amd import tiny
    ei = _graph(30, 100, seed=1)
    ei[1, 17] = 30
    tiny._poll_status(wait=True)
    g = tiny.build_graph(ei.to(dev), None, 30)
    with pytest.raises(ValueError):
        tiny._poll_status(wait=True)
    tiny._poll_status(wait=True)                         # reported once
    for _ in range(300):                                 # the ring of status slots wraps without a host read per build
        tiny.build_graph(ei[:, :16].to(dev), None, 30)
    tiny._poll_status(wait=True)


def test_bad_indices_surface_at_the_next_mode_switch_and_inference_tensors_work(dev):
    """(a) a run of fewer than 128 snapshots used to return silently wrong numbers for a node id outside [0, n) --
    model.eval() / model.train() now reads the pending checks; DIFFORMER_DEBUG=1 reads them at the call (child process);
    (b) graphs made under torch.inference_mode() track no version counter: the graph cache rebuilds instead of raising;
    (c) dropout = 1.0 in eval mode is the identity (the whole-model path used to reject p >= 1 in both modes)."""
    import os, subprocess, sys
    from difformer_amd import DIFFormer, tiny
    torch.manual_seed(0)
    model = DIFFormer(4, 4, 1, num_layers=2, kernel="simple", use_graph=True, use_weight=False).to(dev).train()
    x = torch.randn(30, 4, device=dev)
    ei = _graph(30, 100, seed=2)
    ei[0, 5] = 31
    tiny._poll_status(wait=True)
    before = tiny.stats["forward"]
    model(x, ei.to(dev))
    assert tiny.stats["forward"] == before + 1                       # took the whole-model path, no error yet
    with pytest.raises(ValueError):
        model.eval()
    model.eval()                                                      # reported once
    good = _graph(30, 100, seed=3).to(dev)
    with torch.inference_mode():
        gi = good.clone()
        y1 = model(x, gi)
        y2 = model(x, gi)
    assert torch.equal(y1, y2)
    with torch.no_grad():
        assert torch.allclose(model(x, good), y1, rtol=0, atol=0)
        model.dropout = 1.0
        before = tiny.stats["forward"]
        assert torch.equal(model(x, good), y1) and tiny.stats["forward"] == before + 1      # eval: identity, still one launch
    model.dropout = 0.0
    code = """
import torch
from difformer_amd import DIFFormer
dev = torch.device('cuda:0')
model = DIFFormer(4, 4, 1, num_layers=2, kernel='simple', use_graph=True, use_weight=False).to(dev)
ei = torch.randint(0, 30, (2, 100)); ei[1, 3] = 99
try:
    model(torch.randn(30, 4, device=dev), ei.to(dev))
    print('NO ERROR')
except ValueError as e:
    print('RAISED AT THE CALL')
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, DIFFORMER_DEBUG="1"), capture_output=True, text=True,
                       timeout=600)
    assert "RAISED AT THE CALL" in r.stdout, r.stdout + r.stderr[-1500:]


def _oracle(p, x, ei, w, cfg, masks=None):
    """og.difformer_forward with the dropout masks of :192 / :204 applied where the reference applies them."""
    if masks is None:
        return og.difformer_forward(p, x, ei, w, cfg)
    alpha = cfg["alpha"]
    ln = lambda t, k: F.layer_norm(t, (t.shape[-1],), p[k + ".weight"], p[k + ".bias"], 1e-5)
    h = F.linear(x, p["fcs.0.weight"], p["fcs.0.bias"])
    if cfg["use_bn"]:
        h = ln(h, "bns.0")
    h = torch.relu(h) * masks[0]
    layers = [h]
    for i in range(cfg["num_layers"]):
        h = og.difformer_conv(p, f"convs.{i}.", h, h, ei, w, layers[0], cfg)
        if cfg["use_residual"]:
            h = alpha * h + (1 - alpha) * layers[i]
        if cfg["use_bn"]:
            h = ln(h, f"bns.{i + 1}")
        h = h * masks[i + 1]
        layers.append(h)
    return F.linear(h, p["fcs.1.weight"], p["fcs.1.bias"])


def _cases():
    rng = np.random.RandomState(5)
    out = []
    sizes = [1, 2, 7, 20, 64, 65, 129, 513, 700, 1068, 3000]
    for k in range(36):
        n = int(sizes[k % len(sizes)])
        kernel = "sigmoid" if k % 2 else "simple"
        out.append(dict(n=n, f_in=int(rng.choice([1, 4, 8, 14, 33, 64])), hidden=int(rng.choice([1, 2, 3, 4, 4, 4, 5, 8])),
                        c=int(rng.choice([1, 1, 2, 8])), layers=int(rng.choice([1, 2, 2, 3, 4])), kernel=kernel,
                        use_bn=bool(rng.rand() < 0.7), use_residual=bool(rng.rand() < 0.7), use_weight=bool(rng.rand() < 0.5),
                        use_graph=bool(rng.rand() < 0.75), use_source=bool(rng.rand() < 0.3),
                        graph_weight=float(rng.choice([-1, -1, 0.3, 0.8])), alpha=float(rng.choice([0.5, 0.5, 0.2])),
                        weighted=bool(rng.rand() < 0.6), deg=int(rng.choice([1, 4, 12])), iso=int(rng.choice([0, 0, 2])),
                        dropout=float(rng.choice([0.0, 0.0, 0.2, 0.5])), seed=k))
        if out[-1]["hidden"] <= 2:           # LayerNorm over one or two features is +-1 whatever the input: nobody trains that
            out[-1]["use_bn"] = False
    return out


_ID = lambda c: f"{c['kernel']}-n{c['n']}-d{c['hidden']}-L{c['layers']}-s{c['seed']}"


@pytest.mark.parametrize("c", _cases(), ids=_ID)
def test_forward_and_backward_every_flag(c, dev):
    """Every case on the plan the library picks by size (`sigmoid` above 64 nodes: one launch per layer over the chip,
    csrc/tiny_sigmoid_grid.hip; everything else: one workgroup, csrc/tiny_model.hip)."""
    _run_case(c, dev)


@pytest.mark.parametrize("plan", [1, 2], ids=["one-workgroup", "grid"])
@pytest.mark.parametrize("c", _cases(), ids=_ID)
def test_both_launch_plans(c, plan, dev, monkeypatch):
    """Every case with the plan forced: both sides of the size thresholds run both sets of kernels (`sigmoid` on one workgroup
    capped at 700 nodes here: 9 M pairs per layer on one compute unit is what the grid plan exists to avoid)."""
    from difformer_amd import tiny
    if plan == 1 and c["kernel"] == "sigmoid" and c["n"] > 700:
        c = dict(c, n=700)
    monkeypatch.setattr(tiny, "PLAN", plan)
    _run_case(c, dev)


def _run_case(c, dev):
    from difformer_amd import DIFFormer, tiny
    n, d, L = c["n"], c["hidden"], c["layers"]
    torch.manual_seed(100 + c["seed"])
    model = DIFFormer(c["f_in"], d, c["c"], num_layers=L, num_heads=1, kernel=c["kernel"], alpha=c["alpha"], dropout=c["dropout"],
                      use_bn=c["use_bn"], use_residual=c["use_residual"], use_weight=c["use_weight"], use_graph=c["use_graph"],
                      graph_weight=c["graph_weight"], use_source=c["use_source"])
    with torch.no_grad():
        for bn in model.bns:
            bn.weight.add_(0.2 * torch.randn(bn.weight.shape))
            bn.bias.add_(0.2 * torch.randn(bn.bias.shape))
    g = torch.Generator().manual_seed(c["seed"])
    x = torch.randn(n, c["f_in"], generator=g)
    iso = min(c["iso"], n - 1)
    ei = _graph(n, n * c["deg"], seed=c["seed"], isolated=iso)
    ei = torch.cat([ei, torch.arange(n - iso).repeat(2, 1)], dim=1)
    w = (torch.rand(ei.shape[1], generator=g) * 2 + 0.05) if c["weighted"] else None
    go = torch.randn(n, c["c"], generator=g)
    cfg = dict(in_channels=c["f_in"], hidden_channels=d, out_channels=c["c"], num_layers=L, num_heads=1, kernel=c["kernel"],
               alpha=c["alpha"], use_bn=c["use_bn"], use_residual=c["use_residual"], use_weight=c["use_weight"],
               use_graph=c["use_graph"], graph_weight=c["graph_weight"], use_source=c["use_source"])
    model = model.to(dev).train()
    xd = x.to(dev).requires_grad_(True)
    eid, wd = ei.to(dev), (None if w is None else w.to(dev))
    before = dict(tiny.stats)
    masks = None
    if c["dropout"] > 0:
        # the kernel draws ONE torch.rand((L + 1, n, d)) per forward from the device generator: replay it for the oracle
        torch.manual_seed(4242)
        state = torch.cuda.get_rng_state(dev)
        rnd = torch.rand((L + 1, n, d), device=dev)
        torch.cuda.set_rng_state(state, dev)
        p = c["dropout"]
        masks = ((rnd >= p).double() / (1.0 - p)).cpu()
    out = model(xd, eid if c["use_graph"] else None, wd if c["use_graph"] else None)
    out.backward(go.to(dev))
    assert tiny.stats["forward"] == before["forward"] + 1 and tiny.stats["backward"] == before["backward"] + 1
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    p64 = og.leaves(sd)
    x64 = x.double().requires_grad_(True)
    ref = _oracle(p64, x64, ei if c["use_graph"] else None, None if w is None else w.double(), cfg, masks)
    ref.backward(go.double())
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < TOL
    grads = {k: v.grad for k, v in p64.items()}
    gmax = max([float(v.abs().max()) for v in grads.values() if v is not None] + [1e-30])
    # Second yardstick, for the tensors a float32 backward cannot resolve to 1e-4 of themselves: the SAME oracle run in
    # float32 on the CPU.  Widths 1-3 under LayerNorm (one feature: the output is the bias whatever the input), a handful of
    # nodes (one node: the attention is the node's own value whatever q and k are) make whole gradients cancel in float64; any
    # float32 run leaves ~1e-7 of the step's largest entry there.  A tensor passes at 1e-4 (conftest.grad_err) or within 4x
    # the error of the reference's own arithmetic in float32.
    p32 = og.leaves(sd, torch.float32)
    x32 = x.clone().requires_grad_(True)
    ref32 = _oracle(p32, x32, ei if c["use_graph"] else None, w, cfg, None if masks is None else masks.float())
    ref32.backward(go)

    floor = 2e-6 if n >= 16 else 2e-3          # (one node: sigma / sigma is exactly 1 on the CPU, p v / p in the kernel is not)

    def ok(got, r64, r32, what):
        e = grad_err(got, r64, gmax, floor)
        e32 = grad_err(r32, r64, gmax, floor)
        assert e < TOL or e < 4.0 * e32, (what, e, e32)

    ok(xd.grad.cpu().numpy(), x64.grad.numpy(), x32.grad.numpy(), "dx")
    for k, prm in model.named_parameters():
        if grads[k] is None:
            assert prm.grad is None or not prm.grad.any(), k
            continue
        assert prm.grad is not None and torch.isfinite(prm.grad).all(), k
        ok(prm.grad.cpu().numpy(), grads[k].numpy(), p32[k].grad.numpy(), k)
    # bitwise reproducible, and backward(retain_graph=True) twice gives the same gradients
    if c["dropout"] == 0:
        out2 = model(xd, eid if c["use_graph"] else None, wd if c["use_graph"] else None)
        assert torch.equal(out2, out)
        g1 = [p_.grad.clone() for p_ in model.parameters() if p_.grad is not None]
        model.zero_grad()
        out2.backward(go.to(dev), retain_graph=True)
        g2 = [p_.grad.clone() for p_ in model.parameters() if p_.grad is not None]
        model.zero_grad()
        out2.backward(go.to(dev))
        g3 = [p_.grad.clone() for p_ in model.parameters() if p_.grad is not None]
        assert all(torch.equal(a, b) and torch.equal(b, c_) for a, b, c_ in zip(g1, g2, g3))


def test_eval_calls_and_frozen_parameters(dev):
    from difformer_amd import DIFFormer, tiny
    torch.manual_seed(0)
    model = DIFFormer(8, 4, 1, num_layers=2, kernel="simple", use_weight=False).to(dev).eval()
    x, ei = torch.randn(129, 8).to(dev), _graph(129, 1500, seed=3).to(dev)
    before = tiny.stats["forward"]
    with torch.no_grad():
        a = model(x, ei)
        b = model(x, ei)
    assert tiny.stats["forward"] == before + 2 and torch.equal(a, b)
    model.train()
    model.fcs[0].weight.requires_grad_(False)
    model(x, ei).sum().backward()
    assert model.fcs[0].weight.grad is None and model.convs[0].Wq.weight.grad is not None


def test_larger_models_keep_the_layer_path(dev):
    from difformer_amd import DIFFormer, tiny
    before = tiny.stats["forward"]
    x, ei = torch.randn(300, 8).to(dev), _graph(300, 1500, seed=3).to(dev)
    for kw in (dict(hidden=16), dict(hidden=4, heads=2)):
        model = DIFFormer(8, kw["hidden"], 2, num_heads=kw.get("heads", 1)).to(dev).eval()
        with torch.no_grad():
            assert torch.isfinite(model(x, ei)).all()
    w = torch.rand(ei.shape[1], device=dev, requires_grad=True)
    model = DIFFormer(8, 4, 2).to(dev).train()
    model(x, ei, w).sum().backward()                        # edge_weight wants a gradient: the operator path returns it
    assert w.grad is not None and tiny.stats["forward"] == before


@pytest.mark.parametrize("n,hidden,layers,edges", [(4096, 8, 8, 65535), (4095, 7, 3, 30000), (1068, 4, 2, 27000), (577, 5, 5, 0)])
@pytest.mark.parametrize("kernel", ["sigmoid", "simple"])
def test_grid_plan_at_its_limits(n, hidden, layers, edges, kernel, dev, monkeypatch):
    """The grid plan at the limits of the C entry points (4,096 nodes, hidden 8, 8 layers, 65,535 entries), at a node count that
    leaves a ragged last block and ragged key splits, at wikimath's shape and without a graph: output, dx and every parameter
    gradient against the float64 oracle."""
    from difformer_amd import tiny
    monkeypatch.setattr(tiny, "PLAN", 2)
    c = dict(n=n, f_in=14, hidden=hidden, c=3, layers=layers, kernel=kernel, use_bn=True, use_residual=True, use_weight=hidden != 4,
             use_graph=edges > 0, use_source=hidden == 5, graph_weight=0.3 if hidden == 7 else -1.0, alpha=0.5, weighted=hidden != 4,
             deg=1, iso=0, dropout=0.0, seed=n + hidden)
    if edges:
        c["deg"] = max(1, (edges - n) // n)
    _run_case(c, dev)


def test_summed_costs_with_packed_parameters_match_the_separate_path(dev, monkeypatch):
    """spatial-temporal/main.py:105-121 sums the snapshots' costs before ONE backward(retain_graph=True): from the second forward on
    unchanged parameters they enter as one flat tensor (tiny._Pack).  Same gradients as with separate inputs, a second backward
    over the retained graph doubles them, a frozen parameter gets none, an optimiser step retires the flat copy, and the model
    can still be deep-copied."""
    import copy
    from difformer_amd import DIFFormer, tiny
    n, T = 20, 6
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(n, 4, generator=g).to(dev) for _ in range(T)]
    ys = [torch.randn(n, generator=g).to(dev) for _ in range(T)]
    ei = torch.cat([torch.randint(0, n, (2, 70), generator=g), torch.arange(n).repeat(2, 1)], 1).to(dev)

    def run(pack):
        monkeypatch.setattr(tiny, "PACK", pack)
        torch.manual_seed(11)
        model = DIFFormer(4, 4, 1, num_layers=2, num_heads=1, kernel="simple", use_bn=True, use_residual=True, use_graph=True,
                          use_weight=False, dropout=0.0).to(dev).train()
        model.fcs[0].bias.requires_grad_(False)
        cost = 0
        for x, y in zip(xs, ys):
            cost = cost + torch.mean((model(x, ei)[:, 0] - y) ** 2)
        cost = cost / T
        cost.backward(retain_graph=True)
        once = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        cost.backward()
        twice = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        return model, float(cost.detach()), once, twice
    m0, c0, once0, twice0 = run(False)
    before = tiny.stats["forward"]
    m1, c1, once1, twice1 = run(True)
    assert tiny.stats["forward"] == before + T
    assert tiny._packs.get(m1) is not None and m1.fcs[0].bias.grad is None and "fcs.0.bias" not in once1
    assert abs(c0 - c1) <= 1e-6 * abs(c0) and once0.keys() == once1.keys()
    gmax = max(float(v.abs().max()) for v in once0.values())
    for k in once0:
        assert float((once0[k] - once1[k]).abs().max()) <= 2e-6 * gmax, k
        assert float((twice1[k] - 2 * once1[k]).abs().max()) <= 2e-6 * gmax, k
    # the flat copy belongs to one set of parameter versions
    opt = torch.optim.SGD([p for p in m1.parameters() if p.requires_grad], lr=0.1)
    opt.step()
    out = m1(xs[0], ei)
    assert tiny._packs[m1][1] is None                       # first forward after the step: separate inputs again
    out.sum().backward()
    assert torch.isfinite(copy.deepcopy(m1)(xs[0], ei)).all()

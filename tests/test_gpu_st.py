"""The `spatial-temporal` task folder's REAL configuration on the HIP kernels: every `--method difformer`
line of `spatial-temporal/run.sh:5-40` / `run_hyper_search.sh:12-15` trains hidden 4, two layers, one head, `--use_bn
--use_residual`, WITHOUT `--use_weight` (value = the layer input, difformer.py:116), `snapshot.edge_attr` positionally as
`edge_weight` (`main.py:105`), `simple` and `sigmoid`, with and without the graph term, on 20 / 129 / 1,068-node graphs, the
complete graph with unit weights under `--special_treat dense` (`main.py:98-103`), and -- for every dataset but wikimath --
hundreds of forwards whose costs are summed before ONE `cost_tr.backward(retain_graph=True)` (`main.py:94-120`).

Yardsticks: tests/golden/golden_st.npz (outputs and gradients of `spatial-temporal/difformer.py` itself, float64 run) and,
for the long epochs, float64 autograd of the oracle pinned to those fixtures (tests/test_oracle_st_golden.py).  The caller
lines are restated in tests/st_common.py with citations (no /root/reference on the GPU box).  Tolerance 1e-4 per tensor
(conftest.grad_err)."""
import gc

import numpy as np
import pytest
import torch

from conftest import grad_err, grad_scale, rel_err, split_model_case
from oracle import difformer_oracle_grad as og
from st_common import (ST, build_model, cases, cost_fn, cumulative_epoch, dense_graph, evaluate, incremental_epoch, snapshots)

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return torch.device("cuda:0")


def _check_grads(model, c):
    gmax = grad_scale(c)
    for k, p in model.named_parameters():
        ref = c["grad_f64/" + k]
        got = np.zeros_like(ref) if p.grad is None else p.grad.cpu().numpy()
        assert np.isfinite(got).all(), k
        assert grad_err(got, ref, gmax) < TOL, k


def _model(c, dev):
    from difformer_amd import DIFFormer
    return build_model(DIFFormer, c, dev)


@pytest.fixture(autouse=True)
def _which_path(request):
    """Every model of this file is the scripts' configuration (hidden 4, one head, <= 1,068 nodes): its forwards run as the
    whole-model kernels of csrc/tiny_model.hip / csrc/tiny_sigmoid_grid.hip (difformer_amd/tiny.py; wikimath with `sigmoid`:
    the grid plan) -- except the one shape that path hands back to the layer-by-layer kernels: graphs beyond 65,535 entries
    (`dense` at n = 1,068)."""
    from difformer_amd import tiny
    before = dict(tiny.stats)
    yield
    name = request.node.name
    layer_path = ("wikimath" in name and "dense" in name) or ("test_wikimath_branch" in name and name.endswith("True]"))
    took = tiny.stats["forward"] > before["forward"]
    assert took != layer_path, (name, took)
    tiny._poll_status(wait=True)


@pytest.mark.parametrize("name", cases("step"))
def test_one_snapshot_step_golden(name, dev):
    """y_hat, cost, every parameter gradient and dx of ONE snapshot against the folder's own model (main.py:105-114)."""
    c = ST[name]
    model, cfg = _model(c, dev)
    x = torch.from_numpy(c["x"]).to(dev).requires_grad_(True)
    y_hat = model(x, torch.from_numpy(c["edge_index"]).to(dev), torch.from_numpy(c["edge_weight"]).to(dev))
    cost = cost_fn(y_hat, torch.from_numpy(c["y"]).to(dev))
    cost.backward()
    assert rel_err(y_hat.detach().cpu().numpy(), c["out_f64"]) < TOL
    assert abs(float(cost.detach()) - float(c["loss_f64"])) < TOL * abs(float(c["loss_f64"]))
    assert grad_err(x.grad.cpu().numpy(), c["dx_f64"], grad_scale(c)) < TOL
    _check_grads(model, c)
    # eval.py:5-23: the same snapshot in eval mode under no_grad (dropout 0: the same numbers), three times = replayed
    model.eval()
    with torch.no_grad():
        ei, ea = torch.from_numpy(c["edge_index"]).to(dev), torch.from_numpy(c["edge_weight"]).to(dev)
        for _ in range(4):
            out = model(x.detach(), ei, ea)
            assert rel_err(out.cpu().numpy(), c["out_f64"]) < TOL


@pytest.mark.parametrize("name", cases("dense"))
def test_special_treat_dense_golden(name, dev):
    """`--special_treat dense`: the complete graph, all weights 1 (main.py:98-103) -- 1.14 M weighted entries at n = 1,068."""
    c = ST[name]
    model, cfg = _model(c, dev)
    ei = dense_graph(int(c["n"])).to(dev)
    ea = torch.ones(ei.shape[1]).to(dev)
    y_hat = model(torch.from_numpy(c["x"]).to(dev), ei, ea)
    cost_fn(y_hat, torch.from_numpy(c["y"]).to(dev)).backward()
    assert rel_err(y_hat.detach().cpu().numpy(), c["out_f64"]) < TOL
    _check_grads(model, c)


@pytest.mark.parametrize("name", cases("cumul"))
def test_summed_cost_one_backward_golden(name, dev):
    """Six snapshots, costs summed, ONE backward(retain_graph=True): against the folder's own model run the same way."""
    c = ST[name]
    model, cfg = _model(c, dev)
    cost_tr, outs = cumulative_epoch(model, snapshots(c, dev))
    for t, o in enumerate(outs):
        assert rel_err(o.cpu().numpy(), c["out_f64"][t]) < TOL
    assert abs(float(cost_tr.detach()) - float(c["loss_f64"])) < TOL * abs(float(c["loss_f64"]))
    _check_grads(model, c)


def _long_epoch_data(n, d, deg, T, dynamic, seed, dev, dense=False):
    g = torch.Generator().manual_seed(seed)
    xs, ys = torch.randn(T, n, d, generator=g), torch.randn(T, n, generator=g)

    def graph():
        if dense:
            return dense_graph(n)
        row = torch.arange(n).repeat_interleave(deg)
        return torch.cat([torch.stack([row, torch.randint(0, n, (n * deg,), generator=g)]), torch.arange(n).repeat(2, 1)], 1)

    static = graph()
    host = []
    for t in range(T):
        ei = graph() if dynamic else static
        ea = torch.ones(ei.shape[1]) if dense else torch.rand(ei.shape[1], generator=g) * 3.0 + 0.05
        host.append((xs[t], ei, ea, ys[t]))
    fresh = lambda: [tuple(a.clone().to(dev) for a in s) for s in host]      # `snapshot.to(device)`: new tensors every epoch
    return host, fresh


def _oracle_epoch(model, host, cfg):
    p = og.leaves({k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
    cost_tr = 0
    for x, ei, ea, y in host:
        out = og.difformer_forward(p, x.double(), ei if cfg["use_graph"] else None, ea.double(), cfg)
        cost_tr = cost_tr + cost_fn(out, y.double())
    cost_tr = cost_tr / len(host)
    cost_tr.backward()
    return float(cost_tr.detach()), {k: (None if v.grad is None else v.grad.numpy()) for k, v in p.items()}


def _state(dev):
    from difformer_amd import ops
    gc.collect()
    torch.cuda.synchronize()
    return len(ops.csr_cache.entries), torch.cuda.memory_allocated(dev)


@pytest.mark.parametrize("kernel,use_graph,n,d,deg,T,dynamic", [
    ("simple", True, 20, 4, 4, 104, False),      # chickenpox: 20 counties, ~100 training weeks, a static graph
    ("sigmoid", True, 20, 4, 4, 60, False),
    ("simple", False, 20, 4, 4, 60, False),
    ("simple", True, 129, 8, 12, 50, True),      # covid: the edge list changes from day to day
    ("sigmoid", True, 129, 8, 12, 50, True),
    ("sigmoid", False, 129, 8, 12, 50, True),
])
def test_cumulative_epochs(kernel, use_graph, n, d, deg, T, dynamic, dev):
    """main.py:86-121 as the scripts run it: >= 50 snapshots with FRESH device tensors per snapshot, costs summed, one
    backward(retain_graph=True), Adam step, then further epochs -- every parameter gradient of the first epoch against float64
    autograd of the oracle over the same snapshots, the second epoch's too (after the step), and neither the CSR cache nor
    the allocated device memory grows from epoch to epoch."""
    from difformer_amd import DIFFormer
    torch.manual_seed(123)
    model = DIFFormer(d, 4, 1, num_layers=2, alpha=0.5, dropout=0.0, num_heads=1, kernel=kernel, use_bn=True,
                      use_residual=True, use_graph=use_graph, use_weight=False).to(dev)
    model.reset_parameters()                                             # main.py:79
    cfg = dict(in_channels=d, hidden_channels=4, out_channels=1, num_layers=2, num_heads=1, kernel=kernel, alpha=0.5,
               use_bn=True, use_residual=True, use_weight=False, use_graph=use_graph, graph_weight=-1, use_source=False)
    host, fresh = _long_epoch_data(n, d, deg, T, dynamic, 17, dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=0.0)   # main.py:80
    states = []
    for epoch in range(4):
        r_cost, r_grads = _oracle_epoch(model, host, cfg) if epoch < 2 else (None, None)
        cost_tr, outs = cumulative_epoch(model, fresh(), None)
        if r_grads is not None:
            assert abs(float(cost_tr.detach()) - r_cost) < TOL * abs(r_cost)
            gmax = max(float(np.abs(v).max()) for v in r_grads.values() if v is not None)
            for k, prm in model.named_parameters():
                assert prm.grad is not None and torch.isfinite(prm.grad).all(), k
                assert grad_err(prm.grad.cpu().numpy(), r_grads[k], gmax) < TOL, (epoch, k)
        opt.step()                                                       # main.py:120-121
        opt.zero_grad()
        del cost_tr, outs
        states.append(_state(dev))
        assert np.isfinite(evaluate(model, fresh()))                     # main.py:123 -> eval.py:5-23
    assert states[-1][0] <= states[1][0] and states[-1][1] <= states[1][1] + (1 << 20), states


@pytest.mark.parametrize("kernel,use_graph,dense", [("simple", True, False), ("sigmoid", True, False), ("simple", False, False),
                                                    ("simple", True, True)])
def test_wikimath_branch(kernel, use_graph, dense, dev):
    """main.py:110-114: backward and optimiser step per snapshot on the 1,068-node graph (optionally `--special_treat dense`:
    1.14 M unit-weight entries); the first snapshot's gradients against the oracle, the cost falls over three epochs, no growth."""
    from difformer_amd import DIFFormer
    n, d, T = 1068, 14, 12
    torch.manual_seed(123)
    model = DIFFormer(d, 4, 1, num_layers=2, alpha=0.5, dropout=0.0, num_heads=1, kernel=kernel, use_bn=True,
                      use_residual=True, use_graph=use_graph, use_weight=False).to(dev)
    model.reset_parameters()
    cfg = dict(in_channels=d, hidden_channels=4, out_channels=1, num_layers=2, num_heads=1, kernel=kernel, alpha=0.5,
               use_bn=True, use_residual=True, use_weight=False, use_graph=use_graph, graph_weight=-1, use_source=False)
    host, fresh = _long_epoch_data(n, d, 10, T, False, 23, dev, dense=dense)
    r_cost, r_grads = _oracle_epoch(model, host[:1], cfg)
    x, ei, ea, y = fresh()[0]
    model.train()
    cost = cost_fn(model(x, ei, ea), y)
    cost.backward()
    assert abs(float(cost.detach()) - r_cost) < TOL * abs(r_cost)
    gmax = max(float(np.abs(v).max()) for v in r_grads.values() if v is not None)
    for k, prm in model.named_parameters():
        assert grad_err(prm.grad.cpu().numpy(), r_grads[k], gmax) < TOL, k
    opt = torch.optim.Adam(model.parameters(), lr=0.005)
    opt.zero_grad()
    costs, states = [], []
    for epoch in range(3):
        costs.append(incremental_epoch(model, fresh(), opt))
        states.append(_state(dev))
    assert np.isfinite(costs).all() and costs[-1] < costs[0]
    assert states[-1][0] <= states[0][0] and states[-1][1] <= states[0][1] + (1 << 20), states

"""The oracle pinned against the `spatial-temporal` folder's OWN copy of the model (`spatial-temporal/difformer.py`, the
subset without graph_weight / use_source) in the configuration its scripts use: hidden 4, no Wv (`use_weight=False`),
`edge_attr` as `edge_weight`, both kernels, with and without the graph term, the complete graph with unit weights, and
the summed cost over several snapshots with one backward (`spatial-temporal/main.py:94-120`).  Fixtures:
tests/golden/golden_st.npz <- tests/golden/make_golden_st.py.  Tolerances as tests/test_oracle_grad_golden.py."""
import numpy as np
import pytest
import torch

from conftest import grad_err, grad_scale, rel_err, split_model_case
from oracle import difformer_oracle as orc
from oracle import difformer_oracle_grad as og
from st_common import ST, cases, cost_fn, dense_graph

F64, F32 = ("f64", torch.float64, 1e-10), ("f32", torch.float32, 1e-4)


def test_fixture_inventory():
    """What VERDICT r4 item 1(a) asks the fixtures to hold."""
    step, dense, cumul = cases("step"), cases("dense"), cases("cumul")
    assert len(step) == 12 and len(dense) == 6 and len(cumul) == 8
    for n in step + dense + cumul:
        c = ST[n]
        assert int(c["cfg/hidden_channels"]) == 4 and not bool(c["cfg/use_weight"])
        assert bool(c["cfg/use_bn"]) and bool(c["cfg/use_residual"]) and int(c["cfg/out_channels"]) == 1
    assert {ST[n]["x"].shape[0] for n in step} == {20, 129, 1068}
    assert {str(ST[n]["cfg/kernel"]) for n in step} == {"simple", "sigmoid"}
    assert {bool(ST[n]["cfg/use_graph"]) for n in cumul} == {True, False}
    assert {bool(ST[n]["dynamic"]) for n in cumul} == {True, False}


def _step(c, ei, w):
    cfg, sd = split_model_case(c)
    for sfx, dt, tol in (F64, F32):
        p = og.leaves(sd, dt)
        x = torch.from_numpy(c["x"]).to(dt).requires_grad_(True)
        out = og.difformer_forward(p, x, ei if cfg["use_graph"] else None, w.to(dt), cfg)
        loss = cost_fn(out, torch.from_numpy(c["y"]).to(dt))
        loss.backward()
        assert rel_err(out.detach().numpy(), c["out_" + sfx]) < tol
        assert abs(float(loss.detach()) - float(c["loss_" + sfx])) <= tol * abs(float(c["loss_" + sfx]))
        assert rel_err(x.grad.numpy(), c["dx_" + sfx]) < tol
        for k, t in p.items():
            ref = c[f"grad_{sfx}/{k}"]
            got = np.zeros_like(ref) if t.grad is None else t.grad.numpy()
            assert grad_err(got, ref, grad_scale(c, sfx)) < tol, (k, sfx)
    ref64 = orc.difformer_forward(orc.cast_params(sd, np.float64), c["x"].astype(np.float64),
                                  ei.numpy() if cfg["use_graph"] else None, w.double().numpy(), cfg)
    assert rel_err(ref64, c["out_f64"]) < 1e-11


@pytest.mark.parametrize("name", cases("step"))
def test_one_snapshot_step(name):
    c = ST[name]
    _step(c, torch.from_numpy(c["edge_index"]), torch.from_numpy(c["edge_weight"]))


@pytest.mark.parametrize("name", cases("dense"))
def test_special_treat_dense(name):
    c = ST[name]
    ei = dense_graph(int(c["n"]))
    _step(c, ei, torch.ones(ei.shape[1]))


@pytest.mark.parametrize("name", cases("cumul"))
def test_summed_cost_one_backward(name):
    c = ST[name]
    cfg, sd = split_model_case(c)
    T = c["x"].shape[0]
    for sfx, dt, tol in (F64, F32):
        p = og.leaves(sd, dt)
        cost_tr = 0
        for t in range(T):
            ei = torch.from_numpy(c[f"edge_index/{t}"] if f"edge_index/{t}" in c else c["edge_index/0"])
            out = og.difformer_forward(p, torch.from_numpy(c["x"][t]).to(dt), ei if cfg["use_graph"] else None,
                                       torch.from_numpy(c[f"edge_weight/{t}"]).to(dt), cfg)
            assert rel_err(out.detach().numpy(), c["out_" + sfx][t]) < tol
            cost_tr = cost_tr + cost_fn(out, torch.from_numpy(c["y"][t]).to(dt))
        cost_tr = cost_tr / T
        cost_tr.backward(retain_graph=True)
        assert abs(float(cost_tr.detach()) - float(c["loss_" + sfx])) <= tol * abs(float(c["loss_" + sfx]))
        for k, t_ in p.items():
            ref = c[f"grad_{sfx}/{k}"]
            got = np.zeros_like(ref) if t_.grad is None else t_.grad.numpy()
            assert grad_err(got, ref, grad_scale(c, sfx)) < tol, (k, sfx)

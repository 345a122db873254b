"""`spatial-temporal/main.py:86-120` on the drop-in's HOST logic (autograd glue, CSR cache, workspaces) with the arithmetic on
tests/fake_backend.OracleBackend: many forwards whose costs are summed before ONE `backward(retain_graph=True)`, fresh
`edge_index` / `edge_attr` tensor objects per snapshot, a second epoch after the optimiser step.  Gradients against the
fixtures written from `spatial-temporal/difformer.py` itself (tests/golden/golden_st.npz).  The same sequences run on the
HIP kernels in tests/test_gpu_st.py."""
import numpy as np
import pytest
import torch

from conftest import grad_err, grad_scale, rel_err
from fake_backend import OracleBackend
from st_common import ST, cases, cost_fn, cumulative_epoch, dense_graph, build_model, evaluate, incremental_epoch, snapshots

TOL = 1e-4


@pytest.fixture()
def fake_backend(monkeypatch):
    from difformer_amd import ops
    be = OracleBackend()
    monkeypatch.setattr(ops, "_BACKEND", be)
    ops.csr_cache.clear()
    yield be
    ops.csr_cache.clear()


def _check_grads(model, c):
    gmax = grad_scale(c)
    for k, p in model.named_parameters():
        ref = c["grad_f64/" + k]
        got = np.zeros_like(ref) if p.grad is None else p.grad.cpu().numpy()
        assert np.isfinite(got).all(), k
        assert grad_err(got, ref, gmax) < TOL, k


@pytest.mark.parametrize("name", cases("step"))
def test_one_snapshot_step(name, fake_backend):
    from difformer_amd import DIFFormer
    c = ST[name]
    model, cfg = build_model(DIFFormer, c)
    x = torch.from_numpy(c["x"]).requires_grad_(True)
    y_hat = model(x, torch.from_numpy(c["edge_index"]), torch.from_numpy(c["edge_weight"]))
    cost = cost_fn(y_hat, torch.from_numpy(c["y"]))
    cost.backward()
    assert rel_err(y_hat.detach().numpy(), c["out_f64"]) < TOL
    assert grad_err(x.grad.numpy(), c["dx_f64"], grad_scale(c)) < TOL
    _check_grads(model, c)


@pytest.mark.parametrize("name", [n for n in cases("dense") if "wikimath" not in n])
def test_special_treat_dense(name, fake_backend):
    from difformer_amd import DIFFormer
    c = ST[name]
    model, cfg = build_model(DIFFormer, c)
    ei = dense_graph(int(c["n"]))
    y_hat = model(torch.from_numpy(c["x"]), ei, torch.ones(ei.shape[1]))
    cost_fn(y_hat, torch.from_numpy(c["y"])).backward()
    assert rel_err(y_hat.detach().numpy(), c["out_f64"]) < TOL
    _check_grads(model, c)


@pytest.mark.parametrize("name", cases("cumul"))
def test_summed_cost_one_backward_then_a_second_epoch(name, fake_backend):
    from difformer_amd import DIFFormer, ops
    c = ST[name]
    model, cfg = build_model(DIFFormer, c)
    cost_tr, outs = cumulative_epoch(model, snapshots(c))
    for t, o in enumerate(outs):
        assert rel_err(o.numpy(), c["out_f64"][t]) < TOL
    assert abs(float(cost_tr.detach()) - float(c["loss_f64"])) < TOL * abs(float(c["loss_f64"]))
    _check_grads(model, c)
    # second epoch after an optimiser step (main.py:120-121): NEW tensor objects again; the first epoch's graphs are gone
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    opt.step()
    opt.zero_grad()
    before = len(ops.csr_cache.entries)
    cost2, _ = cumulative_epoch(model, snapshots(c), opt)
    assert torch.isfinite(cost2) and float(cost2) != float(cost_tr)
    import gc
    gc.collect()
    cost3, _ = cumulative_epoch(model, snapshots(c), opt)
    gc.collect()
    assert len(ops.csr_cache.entries) <= max(before, 2 * len(outs))      # entries of freed snapshots do not pile up
    assert np.isfinite(evaluate(model, snapshots(c)))


def test_wikimath_branch_backward_per_snapshot(fake_backend):
    from difformer_amd import DIFFormer
    c = ST["cumul/covid_simple_graph"]
    model, cfg = build_model(DIFFormer, c)
    opt = torch.optim.Adam(model.parameters(), lr=0.005)
    first = incremental_epoch(model, snapshots(c), opt)
    for _ in range(3):
        last = incremental_epoch(model, snapshots(c), opt)
    assert np.isfinite(first) and last < first

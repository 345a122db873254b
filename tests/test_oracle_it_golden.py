"""The oracle pinned against the `image and text` folder's OWN copy of the model (`image and text/difformer.py`) in the
configuration of its DIFFormer-a script lines (`run.sh:17,35,54`): `--kernel sigmoid`, hidden 300 / 400, two layers, LayerNorm +
residual, no graph term, no Wv, and the training step of `main.py:97-113`.  Fixtures: tests/golden/golden_it.npz <-
tests/golden/make_golden_it.py.  Also holds the BLOCKED restatements (oracle.sigmoid_attention_blocked /
sigmoid_attention_grad_blocked: what the GPU tests at N = 15,000 are checked against) to the plain ones and to the fixtures."""
import numpy as np
import pytest
import torch

from conftest import grad_err, load_golden, rel_err, split_model_case
from oracle import difformer_oracle as orc
from oracle import difformer_oracle_grad as og

IT = load_golden("it")
ATTN = sorted(k for k in IT if k.startswith("attn/"))
STEP = sorted(k for k in IT if k.startswith("step/"))


def test_fixture_inventory():
    """What VERDICT r5 item 1(c) asks the fixtures to hold: the script flags at hidden 300 and 400, with gradients."""
    assert len(ATTN) == 3 and len(STEP) == 2
    assert {int(IT[n]["cfg/hidden_channels"]) for n in STEP} == {300, 400}
    for n in STEP:
        c = IT[n]
        assert str(c["cfg/kernel"]) == "sigmoid" and not bool(c["cfg/use_weight"]) and not bool(c["cfg/use_graph"])
        assert bool(c["cfg/use_bn"]) and bool(c["cfg/use_residual"]) and float(c["cfg/alpha"]) == 0.5
        assert any(k.startswith("grad_f64/convs.0.Wq") for k in c)
    assert {IT[n]["q"].shape[2] for n in ATTN} == {129, 300, 400}


@pytest.mark.parametrize("name", ATTN)
def test_sigmoid_attention_and_gradients(name):
    c = IT[name]
    q, k, v, g = (c[x].astype(np.float64) for x in ("q", "k", "v", "g"))
    assert rel_err(orc.sigmoid_attention(q, k, v), c["out_f64"]) < 1e-6           # fixtures hold float64 results rounded to float32
    out32 = orc.sigmoid_attention(c["q"], c["k"], c["v"])
    assert rel_err(out32, c["out_f32"]) < 1e-5
    for blk in (16, 2048):
        ob, den = orc.sigmoid_attention_blocked(q, k, v, block=blk, return_den=True)
        assert rel_err(ob, c["out_f64"]) < 1e-6
        s = 1.0 / (1.0 + np.exp(-np.einsum("nhm,lhm->nlh", q, k)))
        assert rel_err(den, s.sum(axis=1)) < 1e-12
        dq, dk, dv = orc.sigmoid_attention_grad_blocked(q, k, v, g, block=blk)
        for got, key in ((dq, "dq"), (dk, "dk"), (dv, "dv")):
            assert rel_err(got, c[key + "_f64"]) < 1e-6, (key, blk)
    qq, kk, vv = (torch.from_numpy(x).requires_grad_(True) for x in (q, k, v))
    og.sigmoid_attention(qq, kk, vv).backward(torch.from_numpy(g))
    for t, key in ((qq, "dq"), (kk, "dk"), (vv, "dv")):
        assert rel_err(t.grad.numpy(), c[key + "_f64"]) < 1e-6


def test_blocked_gradient_matches_autograd_many_heads():
    g_ = torch.Generator().manual_seed(11)
    q, k, v, go = (torch.randn(n, 3, 70, generator=g_, dtype=torch.float64) * 0.3 for n in (37, 53, 53, 37))
    dq, dk, dv = orc.sigmoid_attention_grad_blocked(q.numpy(), k.numpy(), v.numpy(), go.numpy(), block=10)
    qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
    og.sigmoid_attention(qq, kk, vv).backward(go)
    for got, t in ((dq, qq), (dk, kk), (dv, vv)):
        assert rel_err(got, t.grad.numpy()) < 1e-12


@pytest.mark.parametrize("name", STEP)
def test_training_step(name):
    c = IT[name]
    cfg, sd = split_model_case(c)
    gmax = max(float(np.max(np.abs(v))) for k, v in c.items() if k.startswith("grad_f64/") and v.size)
    for sfx, dt, tol in (("f64", torch.float64, 2e-6), ("f32", torch.float32, 1e-4)):
        p = og.leaves(sd, dt)
        x = torch.from_numpy(c["x"]).to(dt).requires_grad_(True)
        out = og.difformer_forward(p, x, None, None, cfg)
        loss = og.training_loss(out, torch.from_numpy(c["y"]), torch.from_numpy(c["train_idx"]))
        loss.backward()
        assert rel_err(out.detach().numpy(), c["out_" + sfx]) < tol
        assert abs(float(loss.detach()) - float(c["loss_" + sfx])) <= tol * abs(float(c["loss_" + sfx]))
        assert rel_err(x.grad.numpy(), c["dx_" + sfx]) < tol
        for k, t in p.items():
            ref = c["grad_f64/" + k]
            got = np.zeros_like(ref) if t.grad is None else t.grad.numpy()
            assert grad_err(got, ref, gmax) < max(tol, 1e-5), (k, sfx)
    ref64 = orc.difformer_forward(orc.cast_params(sd, np.float64), c["x"].astype(np.float64), None, None, cfg)
    assert rel_err(ref64, c["out_f64"]) < 1e-6

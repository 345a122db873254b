"""Worst per-tensor gradient error (tests/conftest.py::grad_err) of the golden `sigmoid` training steps, for the kernel choices
of the sigmoid attention's backward: split-bfloat16 operands (DIFFORMER_SIGMOID_BWD_SPLIT=1) / the fp32 chain (default).  The
training forward (the one that leaves the row sums) is always the fp32 chain; with the split forward feeding an fp32 backward the
worst tensor was 2.0e-5 (model/a_nobn_src convs.1.Wk.bias; measured before the training forward was pinned to fp32).
    python scripts/exp_sigmoid_grad_parity.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import grad_err, grad_scale
import test_gpu_grad as tg
from difformer_amd import ops

dev = torch.device("cuda:0")
for exact in (True, False):
    ops.set_exact_fp32(exact)
    for name in tg.cases("model"):
        if not name.startswith("model/a"):
            continue
        c = tg.GRAD[name]
        model, cfg = tg._build(c, dev)
        x = tg.t(c["x"], dev, True)
        ei = tg.t(c["edge_index"], dev) if cfg["use_graph"] else None
        w = tg.t(c["edge_weight"], dev, True) if "edge_weight" in c else None
        out = model(x, ei, w)
        loss = tg._loss(out, tg.t(c["y"], dev), tg.t(c["train_idx"], dev), str(c["loss_kind"]))
        loss.backward()
        gmax = grad_scale(c)
        worst = max((grad_err(np.zeros_like(c["grad_f64/" + k]) if p.grad is None else p.grad.cpu().numpy(), c["grad_f64/" + k], gmax), k)
                    for k, p in model.named_parameters())
        ref32 = max((grad_err(c["grad_f32/" + k], c["grad_f64/" + k], gmax), k) for k, p in model.named_parameters())
        print(f"{'exact switch on' if exact else 'default        '}: training forward fp32, backward {'split' if os.environ.get('DIFFORMER_SIGMOID_BWD_SPLIT') == '1' and not exact else 'fp32 '} "
              f"{name}: worst {worst[0]:.2e} ({worst[1]}), the reference's own float32 run {ref32[0]:.2e} ({ref32[1]})", flush=True)
ops.set_exact_fp32(False)

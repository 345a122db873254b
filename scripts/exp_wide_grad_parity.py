"""Worst per-tensor gradient error (tests/conftest.py::grad_err, against float64 autograd of the oracle) of a training step at
hidden 128 on >= 4,096 nodes -- the shapes whose records / row-GEMMs take the split-bfloat16 slab kernels (reduce_slab_kernel,
rowgemm_split_kernel) -- in the default build and with every product on the fp32 core.   python scripts/exp_wide_grad_parity.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import grad_err, rel_err
import test_gpu_grad as tg
from difformer_amd import DIFFormer, ops

dev = torch.device("cuda:0")
for hidden, heads, n, deg in [(128, 1, 6000, 6), (128, 1, 9000, 60), (96, 1, 20000, 10)]:
    g = torch.Generator().manual_seed(hidden + n)
    pairs = torch.randint(0, n, (2, n * deg // 2), generator=g)
    ei = torch.cat([pairs, pairs.flip(0), torch.arange(n).repeat(2, 1)], dim=1).to(dev)
    kw = dict(num_layers=2, num_heads=heads, kernel="simple", alpha=0.5, use_bn=True, use_residual=True, use_weight=True,
              use_graph=True, graph_weight=-1, use_source=False)
    cfg = dict(hidden_channels=hidden, **kw)
    x0 = torch.randn(n, 30, generator=g)
    y = torch.randint(0, 5, (n,), generator=g).to(dev)
    idx = torch.randperm(n, generator=g)[: n // 2].to(dev)
    ref = None
    for exact in (True, False):
        ops.set_exact_fp32(exact)
        torch.manual_seed(9)
        model = DIFFormer(30, hidden, 5, dropout=0.0, **kw).to(dev).train()
        x = x0.to(dev).requires_grad_(True)
        out = model(x, ei)
        loss = tg._loss(out, y, idx, "nll")
        loss.backward()
        if ref is None:
            ref = tg._oracle_step(model, x, ei, cfg, y, idx, "nll")
        r_out, r_loss, r_grads, r_dx = ref
        gmax = max(float(np.abs(v).max()) for v in r_grads.values() if v is not None)
        errs = sorted(((grad_err(p.grad.cpu().numpy(), r_grads[k], gmax), k, float(np.abs(r_grads[k]).max()) / gmax)
                       for k, p in model.named_parameters() if r_grads[k] is not None), reverse=True)
        print(f"hidden {hidden} n {n} deg {deg} [{'fp32 MFMA' if exact else 'default  '}]: out {rel_err(out.detach().cpu().numpy(), r_out):.1e}, "
              f"dx {grad_err(x.grad.cpu().numpy(), r_dx, gmax):.1e}, worst tensors: " +
              "; ".join(f"{k} {e:.1e} (|g| {m:.0e} of the largest)" for e, k, m in errs[:3]), flush=True)
    ops.set_exact_fp32(False)

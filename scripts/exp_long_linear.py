"""Experiment: the long-row input Linear (512 -> 64 + LayerNorm + ReLU at CIFAR scale; difformer.py:188-191): split-bfloat16
operands on the bf16 matrix core against the exact fp32-MFMA kernel (DIFFORMER_LINEAR_FP32_MFMA=1), time and error.
    python scripts/exp_long_linear.py;  DIFFORMER_LINEAR_FP32_MFMA=1 python scripts/exp_long_linear.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from difformer_amd import ops

dev = torch.device("cuda:0")
be = ops.get_backend()
for n, ci, co in ((50000, 512, 64), (30000, 512, 64), (15000, 512, 64), (8000, 512, 64), (100000, 1432, 64), (50000, 300, 64)):
    g = torch.Generator().manual_seed(ci)
    x = torch.randn(n, ci, generator=g)
    W, b = torch.randn(co, ci, generator=g) / np.sqrt(ci), torch.randn(co, generator=g)
    lw, lb = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g)
    xd, Wd, bd, lwd, lbd = (t.to(dev) for t in (x, W, b, lw, lb))
    f = lambda: be.linear(xd, Wd, bd, lwd, lbd, 1e-5, True)
    out = f().cpu().double().numpy()
    raw = be.linear(xd, Wd, bd).cpu().double().numpy()
    ref_raw = x.double().numpy() @ W.double().numpy().T + b.double().numpy()
    mu = ref_raw.mean(1, keepdims=True); var = ((ref_raw - mu) ** 2).mean(1, keepdims=True)
    ref = np.maximum((ref_raw - mu) / np.sqrt(var + 1e-5) * lw.double().numpy() + lb.double().numpy(), 0)
    for _ in range(10): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) / 20 * 1e3)
    byt = (n * ci + n * co) * 4
    print(f"{'fp32 MFMA' if os.environ.get('DIFFORMER_LINEAR_FP32_MFMA') == '1' else 'split bf16'}: {n} x {ci} -> {co}: {min(ts):.1f} us "
          f"({byt / min(ts) / 1e3:.0f} GB/s, {2 * n * ci * co / min(ts) / 1e6:.1f} TFLOP/s), max err / max|ref|: Linear {np.abs(raw - ref_raw).max() / np.abs(ref_raw).max():.2e}, "
          f"+ LayerNorm + ReLU {np.abs(out - ref).max() / np.abs(ref).max():.2e}", flush=True)

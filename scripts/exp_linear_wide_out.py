"""Narrow Linear kernel against the vendor GEMM for a fused q|k|v projection at hidden 128 (C_out = 384) and 64."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops
dev = torch.device("cuda:0")
be = ops.get_backend()


def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for rows, cin, cout in ((100000, 128, 384), (100000, 128, 256), (100000, 128, 128), (132534, 64, 192), (50000, 128, 384)):
    x = torch.randn(rows, cin, device=dev); W = torch.randn(cout, cin, device=dev) / 8; b = torch.randn(cout, device=dev)
    ref = torch.nn.functional.linear(x, W, b)
    got = be.linear(x, W, b)
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f"{rows} x {cin} -> {cout}: kernel {t(lambda: be.linear(x, W, b)):.1f} us, vendor {t(lambda: torch.nn.functional.linear(x, W, b)):.1f} us, diff {err:.1e}")

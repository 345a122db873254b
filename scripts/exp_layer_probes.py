"""The closed-form layer kernel as the C4 forward launches it (132,534 rows x 64, graph term with the value projection,
LayerNorm, residual, row-major rows + slice-major scaled copy for the next product), for probe builds of its own source
(csrc/simple_layer.hip, -DDIF_LAYER_PROBE=n; scripts/build_sliced_variants.sh with OBJ=simple_layer):
    DIFFORMER_HIP_LIB=scripts/bin/libdifformer_hip_<name>.so python scripts/exp_layer_probes.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops, _lib

dev = torch.device("cuda:0")
be = ops.get_backend()
torch.manual_seed(0)
n, C = 132534, 64
plan = be.sliced_plan(n, n, C)
rowptr = torch.arange(n + 1, dtype=torch.int32, device=dev) * 600


def timed(f, reps=200):
    for _ in range(20): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): f()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps // 20): g.replay()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e3)
    return min(ts)


x, ax = torch.randn(n, C, device=dev), torch.randn(n, C, device=dev)
W = [torch.randn(C, C, device=dev) / 8 for _ in range(3)]
b = [torch.randn(C, device=dev) * 0.1 for _ in range(3)]
lw, lb = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
rs = torch.rand(n, device=dev)
Wo, bo = torch.randn(112, C, device=dev) / 8, torch.randn(112, device=dev)
rec, _ = be.gram(x)
coef = be.simple_coeffs(rec, n, C, C, W[0], b[0], W[1], b[1], W[2], b[2], 1.0)
t_copy = timed(lambda: be.simple_layer(x, coef, C, ax, W[2], b[2], rs, 1.0, None, True, 0.5, lw, lb, 1e-5, False, rowptr, plan))
t_plain = timed(lambda: be.simple_layer(x, coef, C, ax, W[2], b[2], rs, 1.0, None, True, 0.5, lw, lb, 1e-5))
t_head = timed(lambda: be.simple_layer(x, coef, C, ax, W[2], b[2], rs, 1.0, None, True, 0.5, lw, lb, 1e-5, head=(Wo, bo)))
print(f"{os.path.basename(_lib.LIB_PATH)}: layer + slice-major copy {t_copy:.1f} us, rows only {t_plain:.1f}, head(112) {t_head:.1f}", flush=True)

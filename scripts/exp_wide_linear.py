"""Input layer of the hidden-300 configs (image and text/run.sh:27: 512 -> 300, LayerNorm, ReLU): dif_linear_xwide_f32 against
the library GEMM + tail pass it replaces."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops

dev = torch.device("cuda:0")
be = ops.get_backend()


def timed(f, reps=100):
    for _ in range(10): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps // 10): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for n, ci, co in ((50000, 512, 300), (50000, 512, 400), (16384, 512, 300), (100000, 256, 128), (50000, 832, 416), (50000, 384, 300)):
    x = torch.randn(n, ci, device=dev)
    W, b = torch.randn(co, ci, device=dev) / ci ** 0.5, torch.randn(co, device=dev)
    lw, lb = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev)
    t_new = timed(lambda: be.linear(x, W, b, lw, lb, 1e-5, True))
    t_gemm = timed(lambda: torch.nn.functional.linear(x, W, b))
    y = torch.nn.functional.linear(x, W, b)
    t_tail = timed(lambda: ops.layer_tail(y.unsqueeze(1), None, None, 0.5, lw, lb, 1e-5, relu=True))
    ref = torch.relu(torch.nn.functional.layer_norm(x.double() @ W.double().T + b.double(), (co,), lw.double(), lb.double(), 1e-5))
    err = float((be.linear(x, W, b, lw, lb, 1e-5, True).double() - ref).abs().max() / ref.abs().max())
    print(f"{n} x {ci} -> {co}: one pass {t_new:.1f} us (err {err:.1e}); library GEMM {t_gemm:.1f} + tail {t_tail:.1f} = {t_gemm + t_tail:.1f} us", flush=True)

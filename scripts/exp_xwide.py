"""Closed-form layer at the image / text scripts' widths: the one-pass kernel with streamed weights (csrc/simple_layer_xwide.hip)
against the library-GEMM path of round 3, layer by layer (no graph: image and text/run.sh uses none at these widths)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import DIFFormerConv, ops
dev = torch.device("cuda:0")


def timed(f, reps=50):
    for _ in range(5): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(5): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps // 5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for n, c in ((50000, 192), (50000, 256), (50000, 300), (50000, 400), (15000, 300)):
    torch.manual_seed(0)
    conv = DIFFormerConv(c, c, 1, kernel="simple", use_graph=False).to(dev).eval()
    x = torch.randn(n, c, device=dev)
    lw, lb = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    res = {}
    with torch.no_grad():
        for name, mx in (("one pass", 416), ("library GEMM + tail", 0)):
            ops.XWIDE_MAX = mx
            res[name] = (timed(lambda: conv._layer(x, x, None, None, None, x, 0.5, lw, lb, 1e-5)), conv._layer(x, x, None, None, None, x, 0.5, lw, lb, 1e-5)[0])
    err = float((res["one pass"][1] - res["library GEMM + tail"][1]).abs().max() / res["library GEMM + tail"][1].abs().max())
    print(f"{n} x {c}: whole layer {res['one pass'][0]:.1f} us with the one-pass kernel, {res['library GEMM + tail'][0]:.1f} us with library GEMM + tail (max diff {err:.1e})", flush=True)

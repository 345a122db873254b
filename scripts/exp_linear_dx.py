"""dx = g W of the Linear layers in training (g [n, C_out], W [C_out, C_in]): the row-GEMM kernel against the library GEMM, and
the weight gradient g^T x through the streaming reduce against the library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops

dev = torch.device("cuda:0")
be = ops.get_backend()


def timed(f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for n, co, ci in ((132534, 112, 64), (132534, 192, 64), (100000, 384, 128), (100000, 128, 65), (100000, 256, 128), (100000, 128, 128),
                  (50000, 300, 512), (100000, 2, 128), (16384, 384, 128)):
    g, W, x = torch.randn(n, co, device=dev), torch.randn(co, ci, device=dev), torch.randn(n, ci, device=dev)
    t_lib = timed(lambda: g @ W)
    t_row = timed(lambda: be.row_gemm(g, W))
    g3 = g.reshape(n, 1, co)
    t_wlib = timed(lambda: (g.t() @ x, g.sum(0)))
    t_wred = timed(lambda: be.simple_reduce(g3, g3, x.reshape(n, 1, ci)))
    print(f"{n} x {co} -> {ci}: dx library {t_lib:.1f} us, row-GEMM {t_row:.1f} us; dW + db library {t_wlib:.1f} us, streaming reduce {t_wred:.1f} us", flush=True)

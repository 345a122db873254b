"""Experiment: the closed-form layer kernel alone (csrc/simple_layer.hip) at C4 / CIFAR-50k row counts: plain, with the
slice-major copy for the next layer, and with the output Linear (head) -- for A/B runs of two builds:
    [DIFFORMER_HIP_LIB=difformer_amd/lib/libdifformer_hip_base.so] python scripts/exp_layer_kernel.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops, _lib

dev = torch.device("cuda:0")
be = ops.get_backend()
torch.manual_seed(0)


def timed(f, reps=30, rounds=5):
    for _ in range(10): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): f()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2]


print(os.path.basename(_lib.LIB_PATH))
for n, classes in ((132534, 112), (50000, 10), (100000, 2)):
    C = D = 64
    x = torch.randn(n, C, device=dev)
    ax = torch.randn(n, C, device=dev)
    W = [torch.randn(D, C, device=dev) / 8 for _ in range(3)]
    b = [torch.randn(D, device=dev) * 0.1 for _ in range(3)]
    lw, lb = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
    rs = torch.rand(n, device=dev)
    Wo, bo = torch.randn(classes, D, device=dev) / 8, torch.randn(classes, device=dev)
    rec, _ = be.gram(x)
    coef = be.simple_coeffs(rec, n, C, D, W[0], b[0], W[1], b[1], W[2], b[2], 1.0)
    t_gram = timed(lambda: be.gram(x))
    t_coef = timed(lambda: be.simple_coeffs(rec, n, C, D, W[0], b[0], W[1], b[1], W[2], b[2], 1.0))
    t_fused = (0, 0)      # the fused Gram + coefficients kernel of profiles/r03_experiments.md section 3 (removed: slower)
    t_chain = timed(lambda: be.simple_coeffs(be.gram(x)[0], n, C, D, W[0], b[0], W[1], b[1], W[2], b[2], 1.0))
    t_plain = timed(lambda: be.simple_layer(x, coef, D, ax, W[2], b[2], rs, 1.0, None, True, 0.5, lw, lb, 1e-5))
    t_nog = timed(lambda: be.simple_layer(x, coef, D, None, W[2], b[2], None, 1.0, None, True, 0.5, lw, lb, 1e-5))
    t_head = timed(lambda: be.simple_layer(x, coef, D, ax, W[2], b[2], rs, 1.0, None, True, 0.5, lw, lb, 1e-5, head=(Wo, bo)))
    t_head_nog = timed(lambda: be.simple_layer(x, coef, D, None, W[2], b[2], None, 1.0, None, True, 0.5, lw, lb, 1e-5, head=(Wo, bo)))
    gb = lambda byt, us: byt / us / 1e3
    print(f"n={n}: gram {t_gram[0]:.1f}/{t_gram[1]:.1f} us, coeffs {t_coef[0]:.1f}/{t_coef[1]:.1f}, gram -> finalize -> coeffs {t_chain[0]:.1f}/{t_chain[1]:.1f}, "
          f"fused gram + coeffs {t_fused[0]:.1f}/{t_fused[1]:.1f}, layer with graph {t_plain[0]:.1f}/{t_plain[1]:.1f} "
          f"({gb(3 * n * C * 4, t_plain[0]):.0f} GB/s), no graph {t_nog[0]:.1f}/{t_nog[1]:.1f} ({gb(2 * n * C * 4, t_nog[0]):.0f} GB/s), "
          f"head({classes}) with graph {t_head[0]:.1f}/{t_head[1]:.1f} ({gb(2 * n * C * 4 + n * classes * 4, t_head[0]):.0f} GB/s), "
          f"head no graph {t_head_nog[0]:.1f}/{t_head_nog[1]:.1f}   [min/median us]", flush=True)

"""Experiment: sigmoid attention backward (csrc/sigmoid_attn_bwd.hip) against re-deriving the gradient with tensor ops
(which materialises the [N,L,H] score tensor several times).   python scripts/exp_sigmoid_bwd.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import ops, autograd_ops as ag

dev = torch.device("cuda:0")
be = ops.get_backend()


def timeit(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


bwd_split = os.environ.get("DIFFORMER_SIGMOID_BWD_SPLIT") == "1"      # the backward's split operands are opt-in (gradient parity)
for n in (2708, 8192, 20000):
    q, k, v, g = (torch.randn(n, 1, 64, device=dev) * 0.5 for _ in range(4))
    out, den = be.sigmoid_attention(q, k, v, want_den=True)
    fl = 14.0 * n * n * 64
    ref = None
    for exact in (True, False):              # every product on the fp32 core | split-bfloat16 operands on the bf16 core
        ops.set_exact_fp32(exact)
        t_f = timeit(lambda: be.sigmoid_attention(q, k, v))                       # inference: split unless exact
        t_t = timeit(lambda: be.sigmoid_attention(q, k, v, want_den=True))        # training forward: always the fp32 chain
        t_b = timeit(lambda: be.sigmoid_backward(q, k, v, out, den, g))
        grads = be.sigmoid_backward(q, k, v, out, den, g)
        if exact:
            ref = grads
            dev_s = ""
        else:
            dev_s = ", gradients vs the fp32 chain: " + " ".join(
                f"{float((a - b).abs().max() / b.abs().max()):.1e}" for a, b in zip(grads, ref))
        print(f"N = L = {n} [{'fp32 MFMA' if exact else 'default'}]: inference forward {t_f:.0f} us ({4.0 * n * n * 64 / t_f / 1e6:.1f} TFLOP/s), "
              f"training forward {t_t:.0f} us, backward kernels ({'split' if bwd_split and not exact else 'fp32'}) {t_b:.0f} us "
              f"({fl / t_b / 1e6:.1f} TFLOP/s algorithmic){dev_s}", flush=True)
    ops.set_exact_fp32(False)
    if n <= 8192 and not bwd_split:
        t_r = timeit(lambda: ag._grad_by_recompute(ag._sigmoid_expr, (q, k, v), g), it=5)
        print(f"N = L = {n}: tensor-op recompute {t_r:.0f} us", flush=True)

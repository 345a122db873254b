"""Do the C5 kernels get faster at half the bytes?  Gram pass (x [n, 64]) and input Linear (x [n, 65] -> 64, LayerNorm, ReLU) in
float32 and bfloat16 storage at n = 100,000 (the C5 batch) ... 1,600,000: kernel time by HIP events (mean of 30 after warm-up),
bytes read + written, GB/s.  A pass that is bandwidth-bound halves with the bytes; one at its fixed cost does not."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from difformer_amd import ops
be = ops.get_backend()
dev = torch.device("cuda:0")

def timed(fn, reps=30):
    for _ in range(5):
        fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev])) * 1e3

print(f"{'rows':>9s} {'kernel':>12s} {'f32 us':>8s} {'bf16 us':>8s} {'f32 GB/s':>9s} {'bf16 GB/s':>9s}")
for n in (100000, 200000, 400000, 800000, 1600000):
    for name in ("gram", "linear65"):
        res = {}
        for dt in (torch.float32, torch.bfloat16):
            esz = 2 if dt == torch.bfloat16 else 4
            if name == "gram":
                x = torch.randn(n, 64, device=dev).to(dt)
                us = timed(lambda: be.gram(x))
                bytes_ = n * 64 * esz
            else:
                x = torch.randn(n, 65, device=dev).to(dt)
                W, b = torch.randn(64, 65, device=dev).to(dt) * 0.1, torch.randn(64, device=dev).to(dt)
                lw, lb = torch.ones(64, device=dev).to(dt), torch.zeros(64, device=dev).to(dt)
                us = timed(lambda: be.linear(x, W, b, lw, lb, 1e-5, True))
                bytes_ = n * (65 + 64) * esz
            res[dt] = (us, bytes_ / us / 1e3)
        print(f"{n:9d} {name:>12s} {res[torch.float32][0]:8.1f} {res[torch.bfloat16][0]:8.1f} {res[torch.float32][1]:9.0f} {res[torch.bfloat16][1]:9.0f}", flush=True)

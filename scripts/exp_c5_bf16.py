"""C5 (one Pokec mini-batch, 100,000 x 65 -> 64, 3 layers) in float32 and bfloat16 storage: ms per forward (hipGraph replay and
kernel by kernel) and mean microseconds per C-ABI entry point (HIP events).  python scripts/exp_c5_bf16.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from difformer_amd import DIFFormer, ops

dev = torch.device("cuda:0")
n, pairs, f_in, classes, hidden, layers, kernel, use_graph = bench.WORKLOADS["pokec-batch-s"]
ei = bench.make_graph(n, pairs, dev)
res = {}
for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("f32", torch.float32), ("bf16", torch.bfloat16)):
    torch.manual_seed(123)
    model = DIFFormer(f_in, hidden, classes, num_layers=layers, num_heads=1, kernel=kernel, use_graph=use_graph)
    model.reset_parameters()
    model = model.to(dev).eval().to(dt)
    x = torch.randn(n, f_in, generator=torch.Generator(device=dev).manual_seed(1), device=dev).to(dt)
    be = ops.get_backend()
    with torch.no_grad():
        for _ in range(20):
            model(x, ei)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            model(x, ei)
        torch.cuda.synchronize()
        replay = (time.perf_counter() - t0) / 200 * 1e3
        model.auto_graph = False
        for _ in range(10):
            model(x, ei)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            model(x, ei)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 200 * 1e3
        be.kernel_events = {}
        for _ in range(30):
            model(x, ei)
        kt = be.kernel_times_ms()
        be.kernel_events = None
    print(f"{name:5s} replay {replay:.4f} ms  eager {eager:.4f} ms   " +
          "  ".join(f"{k.replace('dif_', '')}: {np.mean(v) * 1e3:.1f}us x{len(v) // 30}" for k, v in sorted(kt.items())), flush=True)

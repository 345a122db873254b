"""Whole-model kernels (python scripts/exp_tiny_grid_time.py [sigmoid|simple]): one workgroup (plan 1) against one launch per layer over the chip (plan 2), forward + backward
of the spatial-temporal configuration (hidden 4, 2 layers, LayerNorm, residual, graph term) per node count; device time by
events over 50 training snapshots after 10 warm-up ones.  Sets kGridFromNodes in csrc/tiny_model.hip."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from difformer_amd import DIFFormer, tiny  # noqa: E402

dev = torch.device("cuda:0")
KERNEL = sys.argv[1] if len(sys.argv) > 1 else "sigmoid"
print(f"# kernel {KERNEL}")
print("# nodes hidden layers  one-workgroup us   grid us   (forward + backward, device time per snapshot)")
for n, d, L in ((20, 4, 2), (64, 4, 2), (129, 4, 2), (192, 4, 2), (256, 4, 2), (384, 4, 2), (512, 4, 2), (1068, 4, 2), (2048, 4, 2),
                (4096, 4, 2), (129, 8, 2), (256, 8, 2), (1068, 8, 2), (1068, 8, 4), (4096, 8, 2)):
    torch.manual_seed(0)
    model = DIFFormer(14, d, 1, num_layers=L, num_heads=1, kernel=KERNEL, use_bn=True, use_residual=True, use_weight=False,
                      use_graph=True, dropout=0.0).to(dev).train()
    x = torch.randn(n, 14, device=dev)
    row = torch.arange(n).repeat_interleave(8)
    ei = torch.cat([torch.stack([row, torch.randint(0, n, (n * 8,))]), torch.arange(n).repeat(2, 1)], 1).to(dev)
    res = []
    for plan in (1, 2):
        if plan == 1 and n > 2048 and KERNEL == "sigmoid":
            res.append(float("nan"))
            continue
        tiny.PLAN = plan
        for it in range(60):
            if it == 10:
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            model.zero_grad(set_to_none=True)
            model(x, ei).square().mean().backward()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 50 * 1e3)
    print(f"{n:7d} {d:6d} {L:6d} {res[0]:14.1f} {res[1]:12.1f}")

"""Training step of the physical-particle folder's model (physical particle/run.sh: DIFFormer_v2, hidden 64, 2 layers, --use_bn
--use_residual --use_graph --use_weight, batches of 1,024 ActsTrack-sized / 8,192 Tau3Mu-sized graphs): forward over the batch of
graphs, mean pooling per graph + a linear read-out + BCE (main.py's loop), backward, Adam.  Synthetic batches.   python scripts/pp_step.py"""
import os, sys, time
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from difformer_amd import DIFFormer_v2
from exp_v2_batch import batch
dev = torch.device("cuda:0")
print(f"{'batch':30s} {'kernel':8s} {'nodes':>8s} {'train step ms':>14s} {'eval forward ms':>16s}")
for name, B, lo, hi in (("actstrack-like 1024 x ~110", 1024, 40, 180), ("tau3mu-like 8192 x ~20", 8192, 8, 32)):
    for kernel in ("simple", "sigmoid"):
        n_nodes, ei, n = batch(B, lo, hi)
        torch.manual_seed(0)
        model = DIFFormer_v2(7, 64, 64, num_layers=2, kernel=kernel, use_bn=True, use_residual=True, use_graph=True, use_weight=True,
                             dropout=0.4).to(dev)
        head = torch.nn.Linear(64, 1).to(dev)
        opt = torch.optim.Adam(list(model.parameters()) + list(head.parameters()), lr=1.5e-3, weight_decay=1e-3)
        x = torch.randn(n, 7, device=dev)
        y = (torch.rand(B, device=dev) > 0.5).float()
        gid = torch.repeat_interleave(torch.arange(B, device=dev), n_nodes)
        inv = (1.0 / n_nodes.float())[:, None]

        def step():
            model.train(); opt.zero_grad()
            h = model(x, ei, n_nodes)
            pooled = torch.zeros(B, h.shape[1], device=dev).index_add_(0, gid, h) * inv          # global_mean_pool
            loss = F.binary_cross_entropy_with_logits(head(pooled)[:, 0], y)
            loss.backward(); opt.step()

        def ev():
            model.eval()
            with torch.no_grad():
                model(x, ei, n_nodes)

        def timed(fn, reps=20):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3
        print(f"{name:30s} {kernel:8s} {n:8d} {timed(step):14.3f} {timed(ev):16.3f}", flush=True)

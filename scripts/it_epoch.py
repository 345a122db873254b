"""Epoch of the image-and-text training loop (image and text/main.py:94-117) at the scripts' configurations (run.sh: 2 layers, hidden
300 / 400, --use_residual --use_bn --alpha 0.5, no graph, no Wv... `use_weight` as parse.py leaves it; 10 classes): one training step
(forward, log_softmax + nll on the labelled rows, backward, Adam with weight decay) + one evaluation forward, on synthetic features.
    python scripts/it_epoch.py [--epochs 20]"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from difformer_amd import DIFFormer, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--epochs", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--kernel", default="")
ap.add_argument("--train-only", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
print(f"{'dataset':10s} {'kernel':8s} {'N':>6s} {'hidden':>6s} {'train step ms':>14s} {'eval forward ms':>16s} {'epoch ms':>9s}")
for name, n, f_in, hidden, classes in (("stl10", 13000, 512, 400, 10), ("cifar10", 15000, 512, 300, 10), ("20news", 18846, 236, 300, 20)):
    if args.only not in name:
        continue
    for kernel in ("simple", "sigmoid"):
        if args.kernel and kernel != args.kernel:
            continue
        torch.manual_seed(0)
        model = DIFFormer(f_in, hidden, classes, num_layers=2, num_heads=1, kernel=kernel, alpha=0.5, dropout=0.0, use_bn=True,
                          use_residual=True, use_graph=False, use_weight=False).to(dev)
        opt = torch.optim.Adam(model.parameters(), weight_decay=0.1, lr=5e-4)
        x = torch.randn(n, f_in, device=dev)
        y = torch.randint(0, classes, (n,), device=dev)
        train_idx = torch.randperm(n, device=dev)[: classes * 100]

        def train_step():
            model.train()
            opt.zero_grad()
            out = F.log_softmax(model(x, None), dim=1)
            loss = F.nll_loss(out[train_idx], y[train_idx])
            loss.backward()
            opt.step()
            return loss

        def evaluate():
            model.eval()
            with torch.no_grad():
                return model(x, None).argmax(dim=1)

        def timed(fn, reps):
            fn(); fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3
        ts = timed(train_step, args.epochs)
        if args.train_only:
            print(f"{name} {kernel} train step {ts:.3f} ms")
            continue
        te = timed(evaluate, args.epochs)
        tb = timed(lambda: (train_step(), evaluate()), args.epochs)
        print(f"{name:10s} {kernel:8s} {n:6d} {hidden:6d} {ts:14.3f} {te:16.3f} {tb:9.3f}", flush=True)

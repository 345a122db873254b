"""Small-graph training (node classification/main.py:117-131 on a Cora-sized graph) with forward AND backward replayed as
hipGraphs (torch.cuda.make_graphed_callables over the module), against kernel-by-kernel launches.
    python scripts/exp_graphed_training.py [nodes] [features]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difformer_amd import DIFFormer  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2708
f_in = int(sys.argv[2]) if len(sys.argv) > 2 else 1433
classes = 7
g = torch.Generator().manual_seed(0)
x = torch.rand(n, f_in, generator=g).to(dev)
pairs = torch.randint(0, n, (2, 2 * n), generator=g)
ei = torch.cat([pairs, pairs.flip(0), torch.arange(n).repeat(2, 1)], dim=1).to(dev)
y = torch.randint(0, classes, (n,), generator=g).to(dev)
idx = torch.randperm(n, generator=g)[:140].to(dev)


def make():
    torch.manual_seed(1)
    m = DIFFormer(f_in, 64, classes, num_layers=2, kernel="simple", dropout=0.5).to(dev).train()
    return m, torch.optim.Adam(m.parameters(), lr=1e-2, weight_decay=5e-4)


def run(fwd, opt, steps):
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        out = fwd(x, ei)
        loss = torch.nn.functional.nll_loss(torch.log_softmax(out, dim=1)[idx], y[idx])
        loss.backward()
        opt.step()
    return loss


def timed(fwd, opt, steps=50):
    run(fwd, opt, 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = run(fwd, opt, steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, float(loss)


model, opt = make()
ms, loss = timed(model, opt)
print(f"n={n}: kernel-by-kernel training step {ms:.3f} ms (loss {loss:.4f})")
model, opt = make()
model(x, ei)                                              # builds the CSR caches outside the capture
graphed = torch.cuda.make_graphed_callables(model, (x, ei), num_warmup_iters=3)
ms, loss = timed(graphed, opt)
print(f"n={n}: forward and backward as hipGraphs          {ms:.3f} ms (loss {loss:.4f})")

# the whole step (forward, loss, backward, Adam) as ONE hipGraph: needs a capturable optimiser and static operands
torch.manual_seed(1)
model = DIFFormer(f_in, 64, classes, num_layers=2, kernel="simple", dropout=0.5).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=5e-4, capturable=True)


def step_fn():
    out = model(x, ei)
    loss = torch.nn.functional.nll_loss(torch.log_softmax(out, dim=1)[idx], y[idx])
    loss.backward()
    opt.step()
    return loss


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        step_fn()
torch.cuda.current_stream().wait_stream(side)
whole = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(whole):
    static_loss = step_fn()
for _ in range(5):
    whole.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    whole.replay()
torch.cuda.synchronize()
print(f"n={n}: the whole step as ONE hipGraph                 {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms (loss {float(static_loss):.4f})")

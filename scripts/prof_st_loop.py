"""cProfile of the spatial-temporal training loop's forward side on a chickenpox-sized model (20 nodes, hidden 4): host time per call."""
import cProfile, pstats, io, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from difformer_amd import DIFFormer
dev = torch.device("cuda:0")
n, d, T = 20, 4, 104
model = DIFFormer(d, 4, 1, num_layers=2, alpha=0.5, dropout=0.2, num_heads=1, kernel="simple", use_bn=True, use_residual=True,
                  use_graph=True, use_weight=False).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=0.01)
g = torch.Generator().manual_seed(0)
row = torch.arange(n).repeat_interleave(4)
ei = torch.cat([torch.stack([row, torch.randint(0, n, (n * 4,), generator=g)]), torch.arange(n).repeat(2, 1)], 1)
def snaps():
    return [(torch.randn(n, d).to(dev), ei.clone().to(dev), (torch.rand(ei.shape[1]) * 3 + 0.05).to(dev), torch.randn(n).to(dev)) for _ in range(T)]
def epoch(data):
    cost = 0
    for x, e, w, y in data:
        cost = cost + torch.mean((model(x, e, w) - y) ** 2)
    cost = cost / T
    cost.backward(retain_graph=True)
    opt.step(); opt.zero_grad()
    return float(cost)
for _ in range(3):
    epoch(snaps())
data = snaps(); torch.cuda.synchronize()
t0 = time.perf_counter(); epoch(data); torch.cuda.synchronize(); print("us per snapshot", (time.perf_counter() - t0) / T * 1e6)
data = snaps(); torch.cuda.synchronize()
def fwd_only(data):
    c = 0
    for x, e, w, y in data:
        c = c + torch.mean((model(x, e, w) - y) ** 2)
    return c
t0 = time.perf_counter(); c = fwd_only(data); torch.cuda.synchronize(); print("forward + loss us per snapshot", (time.perf_counter() - t0) / T * 1e6)
t0 = time.perf_counter(); (c / T).backward(); torch.cuda.synchronize(); print("backward us per snapshot", (time.perf_counter() - t0) / T * 1e6)
opt.zero_grad()
data = snaps(); torch.cuda.synchronize()
with torch.no_grad():
    t0 = time.perf_counter()
    for x, e, w, y in data:
        model(x, e, w)
    torch.cuda.synchronize(); print("model() alone (no grad) us per snapshot", (time.perf_counter() - t0) / T * 1e6)
data = snaps(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); c = fwd_only(data); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])

"""Entry-point calls and the kernel trace of ONE ogbn-proteins mini-batch training step (10,000 nodes, ~460 k entries)."""
import os, sys, time
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from difformer_amd import DIFFormer, graph_utils as gu, ops
dev = torch.device("cuda:0")
N, PAIRS, F_IN, C, HIDDEN, BATCH = 132534, 39561252, 8, 112, 64, 10000
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randint(0, N, (PAIRS,), generator=g, device=dev); b = torch.randint(0, N, (PAIRS,), generator=g, device=dev)
edge_index = torch.stack([torch.cat([a, b]), torch.cat([b, a])]); del a, b
x = torch.randn(N, F_IN, device=dev, generator=g)
y = (torch.rand(N, C, device=dev, generator=g) > 0.5).float()
model = DIFFormer(F_IN, HIDDEN, C, num_layers=3, num_heads=1, kernel="simple", use_graph=True, use_bn=True, use_residual=True, use_weight=True, dropout=0.0).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-2)
perm = torch.randperm(N, device=dev)
subs = [gu.subgraph(perm[i * BATCH:(i + 1) * BATCH], edge_index, num_nodes=N, relabel_nodes=True)[0] for i in range(6)]
be = ops.get_backend()
def step(i):
    idx = perm[i * BATCH:(i + 1) * BATCH]
    opt.zero_grad()
    out = model(x[idx], subs[i])
    loss = F.binary_cross_entropy_with_logits(out, y[idx])
    loss.backward()
    opt.step()
model.train()
for i in range(5): step(i)
torch.cuda.synchronize()
be.kernel_events = {}
t0 = time.perf_counter(); step(5); torch.cuda.synchronize(); t1 = time.perf_counter()
ev = be.kernel_times_ms(); be.kernel_events = None
print(f"step {1e3 * (t1 - t0):.2f} ms (instrumented)")
for k, v in sorted(ev.items()):
    print(f"   {k}: {len(v)} calls, {sum(v) * 1e3:.0f} us")
print("entry-point calls:", sum(len(v) for v in ev.values()))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    step(4)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30)[:7000])

"""Training step on the C4 shape as node classification/main.py:117-131 runs it: forward (HIP kernels), BCE-with-logits
loss on a training split, backward (HIP adjoint SpMM + simple-attention backward kernels, torch autograd elsewhere), Adam.
    python scripts/exp_train_step.py [cora|c4|h128]
"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from difformer_amd import DIFFormer, ops
from bench import make_graph

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
hidden = 64
if which == "c4":
    n, pairs, f_in, classes, layers = 132534, 39561252, 8, 112, 3
elif which == "h128":          # node classification/run.sh:42-44: a Pokec mini-batch at hidden 128
    n, pairs, f_in, classes, layers, hidden = 100000, 115000, 65, 2, 3, 128
else:
    n, pairs, f_in, classes, layers = 2708, 5278, 1433, 7, 2
torch.manual_seed(0)
model = DIFFormer(f_in, hidden, classes, num_layers=layers, kernel="simple", dropout=0.0 if which != "cora" else 0.2).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=0.0)
x = torch.randn(n, f_in, device=dev)
ei = make_graph(n, pairs, dev)
y = (torch.rand(n, classes, device=dev) > 0.5).float()
train_idx = torch.randperm(n, device=dev)[: n // 2]
crit = torch.nn.BCEWithLogitsLoss()
be = ops.get_backend()


def step():
    model.train()
    opt.zero_grad()
    out = model(x, ei)
    loss = crit(out[train_idx], y[train_idx])
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    l0 = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    l1 = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"{which}: training step {dt * 1e3:.2f} ms ({n / dt / 1e6:.2f} M nodes/s), loss {l0.item():.4f} -> {l1.item():.4f}")
be.kernel_events = {}
step(); torch.cuda.synchronize()
for k, v in sorted(be.kernel_times_ms().items()):
    print(f"   {k}: {len(v)} calls, {sum(v):.3f} ms")
be.kernel_events = None
with torch.no_grad():
    model.eval()
    for _ in range(3): model(x, ei)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): model(x, ei)
    torch.cuda.synchronize()
    print(f"   eval forward {(time.perf_counter() - t0) / K * 1e3:.2f} ms")

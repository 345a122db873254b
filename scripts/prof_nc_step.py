"""cProfile of the ogbn-proteins mini-batch training step (host side): python scripts/prof_nc_step.py"""
import cProfile, pstats, os, sys, io
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from difformer_amd import DIFFormer, graph_utils as gu
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
subs = [gu.subgraph(perm[i * BATCH:(i + 1) * BATCH], edge_index, num_nodes=N, relabel_nodes=True)[0] for i in range(13)]
def steps():
    model.train()
    for i, ei in enumerate(subs):
        idx = perm[i * BATCH:(i + 1) * BATCH]
        opt.zero_grad()
        out = model(x[idx], ei)
        loss = F.binary_cross_entropy_with_logits(out, y[idx])
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
steps(); steps()
import time
t0 = time.perf_counter(); steps(); print("per step ms", (time.perf_counter() - t0) / 13 * 1e3)
pr = cProfile.Profile(); pr.enable(); steps(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:3000])
# the forward alone (the autograd engine runs the backward on its own thread: not seen by cProfile)
def fwd():
    model.train()
    for i, ei in enumerate(subs):
        out = model(x[perm[i * BATCH:(i + 1) * BATCH]], ei)
    torch.cuda.synchronize()
fwd()
t0 = time.perf_counter(); fwd(); print("forward per batch ms", (time.perf_counter() - t0) / 13 * 1e3)
pr = cProfile.Profile(); pr.enable(); fwd(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats("difformer_amd|built-in", 45); print(s.getvalue()[:9000])

"""What could a locality-preserving node order buy the gather SpMM on a mid-degree graph with regional structure (VERDICT r4
item 6)?  The full-Pokec-sized graph (1,632,803 nodes, 15.4 M undirected pairs + loops, hidden 64) with 95 % of a node's
neighbours within +-W ids, run (a) in its IDEAL order -- what a perfect reordering would recover -- and (b) after a hidden random
relabelling (what the kernels see today).  Forward time and the aggregation kernel's time (HIP events), W = 50,000 / 10,000 /
2,000 and the structure-free graph of `pokec-full-s`.   python scripts/exp_regional_order.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from difformer_amd import DIFFormer, ops

dev = torch.device("cuda:0")
n, pairs, f_in, classes, hidden, layers = 1632803, 15400000, 65, 2, 64, 3

def graph(window, relabel, seed=0, hubs=()):
    g = torch.Generator(device=dev).manual_seed(seed)
    a = torch.randint(0, n, (pairs,), generator=g, device=dev)
    if window:
        off = torch.randint(-window, window + 1, (pairs,), generator=g, device=dev)
        b = (a + off) % n                 # (wrap around: clamping would pile ~120,000 entries on the first and the last node)
        far = torch.rand(pairs, generator=g, device=dev) >= 0.95
        b = torch.where(far, torch.randint(0, n, (pairs,), generator=g, device=dev), b)
    else:
        b = torch.randint(0, n, (pairs,), generator=g, device=dev)
    at = 0
    for k, h in enumerate(hubs):               # hub k: h pairs end at node 1000 k + 17
        b[at: at + h] = 1000 * k + 17
        at += h
    loops = torch.arange(n, device=dev)
    ei = torch.stack([torch.cat([a, b, loops]), torch.cat([b, a, loops])])
    if relabel:
        perm = torch.randperm(n, generator=g, device=dev)
        ei = perm[ei]
    return ei.contiguous()

torch.manual_seed(123)
model = DIFFormer(f_in, hidden, classes, num_layers=layers, num_heads=1, kernel="simple", use_graph=True)
model.reset_parameters()
model = model.to(dev).eval()
x = torch.randn(n, f_in, generator=torch.Generator(device=dev).manual_seed(1), device=dev)
be = ops.get_backend()
print(f"{'graph':34s} {'forward ms':>10s} {'spmm us/launch':>15s}")
for name, window, relabel, hubs in (("structure-free (pokec-full-s)", 0, False, ()), ("+-50,000 hidden by relabelling", 50000, True, ()),
                                    ("+-50,000 ideal order", 50000, False, ()), ("+-10,000 ideal order", 10000, False, ()),
                                    ("+-2,000 ideal order", 2000, False, ()),
                                    ("structure-free + 30 hubs of 15,000", 0, False, (15000,) * 30),
                                    ("structure-free + 2 hubs of 120,000", 0, False, (120000, 120000))):
    ei = graph(window, relabel, hubs=hubs)
    ops.csr_cache.clear()
    with torch.no_grad():
        for _ in range(3):
            model(x, ei)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            model(x, ei)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        be.kernel_events = {}
        for _ in range(3):
            model(x, ei)
        kt = be.kernel_times_ms()
        be.kernel_events = None
    sp = [np.mean(v) * 1e3 for k, v in kt.items() if "spmm" in k]
    print(f"{name:34s} {ms:10.3f} {sp[0] if sp else float('nan'):15.1f}", flush=True)
    del ei

import sys, os, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from difformer_amd import DIFFormer
dev = torch.device("cuda:0")
n = 2000
model = DIFFormer(65, 64, 2, num_layers=3).to(dev).eval()
x = torch.randn(n, 65, device=dev)
ei = torch.randint(0, n, (2, 8000), device=dev)
with torch.no_grad():
    for _ in range(20): model(x, ei)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): model(x, ei)
    torch.cuda.synchronize()
    print("forward (host-bound) us:", (time.perf_counter() - t0) / 300 * 1e6)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300): model(x, ei)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

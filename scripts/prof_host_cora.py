"""Host time of the eager cora-a / cora-s forward (2 layers, hidden 64): cProfile by own time."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from difformer_amd import DIFFormer
dev = torch.device("cuda:0")
n, kernel = 2708, (sys.argv[1] if len(sys.argv) > 1 else "sigmoid")
model = DIFFormer(1433, 64, 7, num_layers=2, kernel=kernel).to(dev).eval()
x = torch.randn(n, 1433, device=dev)
ei = torch.randint(0, n, (2, 10556), device=dev)
with torch.no_grad():
    for _ in range(30): model(x, ei)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): model(x, ei)
    torch.cuda.synchronize()
    print(kernel, "forward us:", (time.perf_counter() - t0) / 500 * 1e6)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(500): model(x, ei)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)

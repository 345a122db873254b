import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.getcwd())
os.environ["DIFFORMER_AUTO_GRAPH"] = "0"
from difformer_amd import DIFFormer, ops
from bench import make_graph, WORKLOADS
dev = torch.device("cuda:0")
n, pairs, f_in, classes, hidden, layers, kernel, use_graph = WORKLOADS["pokec-batch-s-bf16"]
torch.manual_seed(123)
model = DIFFormer(f_in, hidden, classes, num_layers=layers, kernel=kernel).to(dev).eval().to(torch.bfloat16)
x = torch.randn(n, f_in, device=dev).to(torch.bfloat16)
ei = make_graph(n, pairs, dev)
with torch.no_grad():
    for _ in range(5): model(x, ei)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(50): model(x, ei)
    torch.cuda.synchronize()
    print("ms per forward", (time.perf_counter() - t0) / 50 * 1e3)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50): model(x, ei)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

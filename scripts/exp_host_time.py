"""Experiment: host time to ENQUEUE one C4 forward (no synchronisation) against its GPU time; cProfile of the enqueue.
    python scripts/exp_host_time.py"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from difformer_amd import DIFFormer
from bench import make_graph, WORKLOADS

dev = torch.device("cuda:0")
n, pairs, f_in, classes, hidden, layers, kernel, use_graph = WORKLOADS["ogbn-proteins-s"]
torch.manual_seed(123)
model = DIFFormer(f_in, hidden, classes, num_layers=layers, kernel=kernel).to(dev).eval()
x = torch.randn(n, f_in, device=dev)
ei = make_graph(n, pairs, dev)
with torch.no_grad():
    for _ in range(3): model(x, ei)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20): model(x, ei)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"20 forwards: host enqueue {(t1 - t0) / 20 * 1e3:.3f} ms each, until the GPU is done {(t2 - t0) / 20 * 1e3:.3f} ms each", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20): model(x, ei)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

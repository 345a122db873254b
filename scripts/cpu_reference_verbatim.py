"""CPU baseline as BASELINE.md section 2 defines it: the reference file `/root/reference/node classification/difformer.py`
imported VERBATIM (three un-vendored symbols shimmed: torch_sparse.SparseTensor / matmul, torch_geometric.utils.degree),
model.eval(), no_grad, fp32, the same synthetic inputs and state_dict as bench.py -- timed on THIS container's cores
(build container only: the GPU box has no /root/reference; bench.py's `cpu_baseline` there times the oracle port).

    python scripts/cpu_reference_verbatim.py [workload ...]  ->  profiles/cpu_reference_verbatim.json

SpMM stand-in: torch_sparse 0.6.10 sorts the COO entries in SparseTensor.__init__ and multiplies through CSR; the shim does
the same with torch ops (stable sort by (row, col), bincount -> rowptr, torch.sparse_csr_tensor @ x), once per gcn_conv
call, i.e. once per layer per forward exactly as the reference does (difformer.py:66-78).
"""
import importlib.util, json, os, sys, time, types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/node classification/difformer.py"


def load_reference():
    ts = types.ModuleType("torch_sparse")

    class SparseTensor:
        def __init__(self, row, col, value, sparse_sizes):
            n = sparse_sizes[1]
            perm = torch.argsort(row * n + col, stable=True)                  # torch_sparse sorts in the constructor
            self.rowptr = torch.zeros(sparse_sizes[0] + 1, dtype=torch.int64)
            self.rowptr[1:] = torch.cumsum(torch.bincount(row, minlength=sparse_sizes[0]), 0)
            self.col, self.value, self.sizes = col[perm], value[perm], sparse_sizes

    def matmul(adj, x):
        a = torch.sparse_csr_tensor(adj.rowptr, adj.col, adj.value.to(x.dtype), size=tuple(adj.sizes))
        return a @ x

    ts.SparseTensor, ts.matmul = SparseTensor, matmul
    tg, tgu = types.ModuleType("torch_geometric"), types.ModuleType("torch_geometric.utils")
    tgu.degree = lambda index, num_nodes: torch.zeros(num_nodes).scatter_add_(0, index, torch.ones(index.shape[0]))
    tg.utils = tgu
    sys.modules.update({"torch_sparse": ts, "torch_geometric": tg, "torch_geometric.utils": tgu})
    spec = importlib.util.spec_from_file_location("ref_difformer", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    import bench
    ref = load_reference()
    cores = os.cpu_count()
    torch.set_num_threads(cores)
    names = sys.argv[1:] or ["cora-s", "cora-a", "cifar50k-s", "pokec-batch-s", "ogbn-proteins-s"]
    path = os.path.join(ROOT, "profiles", "cpu_reference_verbatim.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for wl in names:
        n, pairs, f_in, classes, hidden, layers, kernel, use_graph = bench.WORKLOADS[wl]
        torch.manual_seed(123)
        model = ref.DIFFormer(f_in, hidden, classes, num_layers=layers, num_heads=1, kernel=kernel, use_graph=use_graph)
        model.reset_parameters()
        model.eval()
        x = torch.randn(n, f_in, generator=torch.Generator().manual_seed(1))
        ei = bench.make_graph(n, pairs, torch.device("cpu"), zipf="-zipf" in wl, blocks=8 if "-blocks" in wl else 0) if use_graph else None
        reps = 3 if n > 60000 else 10
        times = []
        with torch.no_grad():
            for i in range(reps + 1):
                t0 = time.perf_counter()
                model(x, ei)
                dt = time.perf_counter() - t0
                if i:
                    times.append(dt)
                print(f"{wl}: forward {i} {dt:.3f} s", flush=True)
        med = float(np.median(times))
        out[wl] = {"value": n / med, "unit": "nodes/s", "cores": cores, "kind": "reference",
                   "seconds_per_forward": {"median": med, "min": min(times), "max": max(times), "timed_runs": reps},
                   "sample": f"whole {layers}-layer forward of the reference file imported verbatim (model.eval(), no_grad, fp32) on "
                             f"the full graph ({n} nodes, {0 if ei is None else ei.shape[1]} entries); 1 warm-up + {reps} timed",
                   "where": "build container (the GPU box has no /root/reference)",
                   "spmm_stand_in": "torch ops: stable sort of the COO entries + torch.sparse_csr_tensor @ x per gcn_conv call"}
        json.dump(out, open(path, "w"), indent=1)
        print(wl, out[wl]["value"], "nodes/s", flush=True)


if __name__ == "__main__":
    main()

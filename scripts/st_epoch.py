"""Epoch time of the `spatial-temporal` folder's training loop (main.py:86-121) on the drop-in, at the shapes of its three
datasets with the scripts' configuration (hidden 4, two layers, no Wv, edge_attr as edge_weight):

    python scripts/st_epoch.py [--epochs 3] [--cpu-port] [--reference]

  chickenpox  n = 20,   d = 4,  104 training snapshots (train_ratio 0.2 of 517), static graph, cumulative cost
  covid       n = 129,  d = 8,   50 snapshots, a new edge list per snapshot, cumulative cost
  wikimath    n = 1068, d = 14, 146 snapshots (0.2 of 731), static graph, backward + step per snapshot
  + `--special_treat dense` at chickenpox and wikimath size (complete graph, unit weights)

Every snapshot hands the model FRESH device tensors, as `snapshot.to(device)` does (main.py:96).  --cpu-port times the
oracle (float32 CPU torch restatement) on this machine's cores; --reference times `spatial-temporal/difformer.py` itself
(build container only).  Synthetic data."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def dense_graph(n):
    row = torch.arange(0, n).unsqueeze(1).repeat(1, n)
    col = torch.arange(0, n).unsqueeze(0).repeat(n, 1)
    return torch.stack([row.reshape(-1), col.reshape(-1)], dim=0)


def data(n, d, deg, T, dynamic, dense, seed=0):
    g = torch.Generator().manual_seed(seed)
    xs, ys = torch.randn(T, n, d, generator=g), torch.randn(T, n, generator=g)

    def graph():
        if dense:
            return dense_graph(n)
        row = torch.arange(n).repeat_interleave(deg)
        return torch.cat([torch.stack([row, torch.randint(0, n, (n * deg,), generator=g)]), torch.arange(n).repeat(2, 1)], 1)

    static = graph()
    out = []
    for t in range(T):
        ei = graph() if dynamic else static
        ea = torch.ones(ei.shape[1]) if dense else torch.rand(ei.shape[1], generator=g) * 3.0 + 0.05
        out.append((xs[t], ei, ea, ys[t]))
    return out


def epoch(model, snaps, opt, cumulative):
    model.train()
    cost_tr = 0
    for time_, (x, ei, ea, y) in enumerate(snaps):
        y_hat = model(x, ei, ea)
        cost = torch.mean((y_hat - y) ** 2)
        if cumulative:
            cost_tr += cost
        else:
            cost_tr += cost.detach().item()
            cost.backward()
            opt.step()
            opt.zero_grad()
    cost_tr = cost_tr / (time_ + 1)
    if cumulative:
        cost_tr.backward(retain_graph=True)
        opt.step()
        opt.zero_grad()
    return float(cost_tr)


def run(cls, host, d, kernel, use_graph, cumulative, device, epochs):
    torch.manual_seed(123)
    model = cls(d, 4, 1, num_layers=2, alpha=0.5, dropout=0.2, num_heads=1, kernel=kernel, use_bn=True, use_residual=True,
                use_graph=use_graph, use_weight=False).to(device)
    model.reset_parameters()
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    times = []
    for e in range(epochs + 1):
        snaps = [tuple(a.clone().to(device) for a in s) for s in host]     # snapshot.to(device)
        if device.type == "cuda":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        cost = epoch(model, snaps, opt, cumulative)
        if device.type == "cuda":
            torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    return min(times[1:]) if epochs else times[0], cost


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--cpu-port", action="store_true")
    ap.add_argument("--reference", action="store_true")
    ap.add_argument("--no-gpu", action="store_true")
    ap.add_argument("--only", default="", help="run the datasets whose name contains this")
    args = ap.parse_args()
    shapes = [("chickenpox", 20, 4, 4, 104, False, False, True), ("covid", 129, 8, 12, 50, True, False, True),
              ("wikimath", 1068, 14, 10, 146, False, False, False), ("chickenpox-dense", 20, 4, 4, 104, False, True, True),
              ("wikimath-dense", 1068, 14, 10, 30, False, True, False)]
    impls = []
    if not args.no_gpu:
        from difformer_amd import DIFFormer
        impls.append(("hip", DIFFormer, torch.device("cuda:0")))
    if args.cpu_port:
        import torch.nn as nn
        from oracle import difformer_oracle_grad as og

        class Port(nn.Module):              # the oracle's differentiable restatement behind the module interface
            def __init__(self, d, hidden, c, **kw):
                super().__init__()
                from difformer_amd import DIFFormer as D
                self.m = D(d, hidden, c, **kw)
                self.cfg = dict(in_channels=d, hidden_channels=hidden, out_channels=c, num_layers=kw["num_layers"], num_heads=1,
                                kernel=kw["kernel"], alpha=kw["alpha"], use_bn=True, use_residual=True, use_weight=False,
                                use_graph=kw["use_graph"], graph_weight=-1, use_source=False)

            def reset_parameters(self):
                self.m.reset_parameters()

            def forward(self, x, ei, ea):
                return og.difformer_forward(dict(self.m.named_parameters()), x, ei if self.cfg["use_graph"] else None, ea, self.cfg)

        impls.append((f"cpu-port({torch.get_num_threads()} threads)", Port, torch.device("cpu")))
    if args.reference:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        from make_golden_st import load_st
        impls.append((f"reference-file({torch.get_num_threads()} threads)", load_st().DIFFormer, torch.device("cpu")))
    print(f"{'dataset':18s} {'kernel':8s} {'graph':6s} {'T':>4s}  " + "  ".join(f"{n:>28s}" for n, _, _ in impls))
    for name, n, d, deg, T, dynamic, dense, cumulative in shapes:
        if args.only not in name:
            continue
        host = data(n, d, deg, T, dynamic, dense)
        for kernel in ("simple", "sigmoid"):
            for use_graph in ((True,) if dense else (True, False)):
                cells = []
                for label, cls, device in impls:
                    ep = 1 if (device.type == "cpu" and n > 500) else args.epochs
                    sec, cost = run(cls, host, d, kernel, use_graph, cumulative, device, ep)
                    cells.append(f"{sec * 1e3:9.1f} ms ({sec / T * 1e6:7.0f} us/snap)")
                print(f"{name:18s} {kernel:8s} {str(use_graph):6s} {T:4d}  " + "  ".join(f"{c:>28s}" for c in cells), flush=True)


if __name__ == "__main__":
    main()

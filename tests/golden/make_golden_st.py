"""Golden vectors for the `spatial-temporal` task folder, by RUNNING THAT FOLDER'S OWN COPY of the model:
`spatial-temporal/difformer.py` (the subset copy without graph_weight / use_source) imported verbatim with the three
shims of make_golden.py, in the configuration its scripts actually use (`spatial-temporal/run.sh:5-40`,
`run_hyper_search.sh:12-15`): `--hidden_channels 4 --num_layers 2 --num_heads 1 --use_bn --use_residual --alpha 0.5`,
WITHOUT `--use_weight` (value = the layer input itself, difformer.py:116), `--kernel simple` and `sigmoid`, with and without
`--use_graph`, `snapshot.edge_attr` passed positionally as `edge_weight` (`main.py:105`), out_channels c = 1, node counts
and lags of the three datasets (`main.py:40-63`: chickenpox n = 20, d = 4; covid n = 129, d = 8; wikimath n = 1068, d = 14),
the cost of `main.py:107` (mean squared error against `snapshot.y`).

Build container only (needs /root/reference):

    python tests/golden/make_golden_st.py      ->  tests/golden/golden_st.npz

Families
  step/   ONE snapshot: y_hat, cost, every parameter gradient and dx      (the wikimath branch, main.py:110-114)
  dense/  `--special_treat dense` (main.py:98-103): the complete graph with unit weights; the edge list is regenerated
          by the test from n (row-major arange pairs as main.py:100-102 builds them), not stored
  cumul/  T snapshots forwarded one after another, the costs SUMMED, divided by T, then ONE
          `cost_tr.backward(retain_graph=True)` (main.py:94-120, every dataset but wikimath): per-snapshot y_hat, the
          mean cost and every parameter gradient; static graph with per-snapshot weights (chickenpox-like) and a
          different edge list per snapshot (covid-like)
Each case under float32 and float64 default dtype, dropout = 0 (the wikimath lines' value: deterministic).
"""
import importlib.util
import os
import sys

import numpy as np
import torch

from make_golden import both, load_reference

REF_ST = "/root/reference/spatial-temporal/difformer.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_st():
    load_reference()                     # installs the shims into sys.modules
    spec = importlib.util.spec_from_file_location("ref_difformer_st", REF_ST)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def st_graph(g, n, deg, loops=True):
    """Directed edge list like the datasets': `deg` random out-neighbours per node (duplicates possible) + self loops
    (the chickenpox / covid graphs carry them), every node reached at least once."""
    row = torch.arange(n).repeat_interleave(deg)
    col = torch.randint(0, n, (n * deg,), generator=g)
    ei = torch.stack([row, col])
    if loops:
        ei = torch.cat([ei, torch.arange(n).repeat(2, 1)], dim=1)
    return ei.long()


def dense_graph(n):
    row = torch.arange(0, n).unsqueeze(1).repeat(1, n)                   # main.py:100-102
    col = torch.arange(0, n).unsqueeze(0).repeat(n, 1)
    return torch.stack([row.reshape(-1), col.reshape(-1)], dim=0)


def leaf(t, dt):
    return t.to(dt).clone().requires_grad_(True)


def build(ref, d, mc, dt):
    torch.set_default_dtype(torch.float32)
    torch.manual_seed(123)                                               # run.sh: --seed 123
    model = ref.DIFFormer(d, 4, 1, num_layers=2, alpha=0.5, dropout=0.0, num_heads=1, kernel=mc["kernel"], use_bn=True,
                          use_residual=True, use_graph=mc["use_graph"], use_weight=False)      # parse.py:54-55
    model.reset_parameters()                                             # main.py:79
    with torch.no_grad():
        for bn in model.bns:
            bn.weight.add_(0.1 * torch.randn(bn.weight.shape, generator=torch.Generator().manual_seed(7)))
            bn.bias.add_(0.1 * torch.randn(bn.bias.shape, generator=torch.Generator().manual_seed(8)))
    torch.set_default_dtype(dt)
    return model.to(dt).train()                                          # main.py:91


def grads_of(model, dt):
    out = {}
    for k, p in model.named_parameters():
        gr = torch.zeros_like(p) if p.grad is None else p.grad
        out[k] = gr.float().numpy() if dt == torch.float32 else gr.numpy()
    return out


def store_cfg(case, d, mc):
    cfg = dict(in_channels=d, hidden_channels=4, out_channels=1, num_layers=2, num_heads=1, kernel=mc["kernel"], alpha=0.5,
               use_bn=True, use_residual=True, use_weight=False, use_graph=mc["use_graph"], graph_weight=-1, use_source=False)
    for k, v in cfg.items():
        case["cfg/" + k] = np.array(v)


def main():
    torch.set_num_threads(1)
    torch.use_deterministic_algorithms(True)
    ref = load_st()
    g = torch.Generator().manual_seed(20260927)
    flat = {}

    def put(case, d):
        for k, v in d.items():
            flat[f"{case}::{k}"] = v

    datasets = [("chickenpox", 20, 4, 4), ("covid", 129, 8, 12), ("wikimath", 1068, 14, 10)]      # name, n, d (lags), degree

    # ---- step/: one snapshot (main.py:105-114) ------------------------------------------------------------
    for name, n, d, deg in datasets:
        for kernel in ("simple", "sigmoid"):
            for use_graph in (True, False):
                mc = dict(kernel=kernel, use_graph=use_graph)
                x = torch.randn(n, d, generator=g)
                y = torch.randn(n, generator=g)
                ei = st_graph(g, n, deg)
                w = torch.rand(ei.shape[1], generator=g) * 3.0 + 0.05

                def run(dt):
                    model = build(ref, d, mc, dt)
                    xx = leaf(x, dt)
                    y_hat = model(xx, ei, w.to(dt))                       # main.py:105: edge_attr positional
                    cost = torch.mean((y_hat - y.to(dt)) ** 2)           # main.py:107 ([n,1] - [n] broadcasts to [n,n]: as written)
                    cost.backward()
                    sd = {k: v.detach().float().numpy() for k, v in model.state_dict().items()}
                    return dict(out=y_hat.detach().numpy(), loss=cost.detach().numpy(), dx=xx.grad.numpy(),
                                grads=grads_of(model, dt), sd=sd)

                r = both(run)
                case = dict(x=x.numpy(), y=y.numpy(), edge_index=ei.numpy(), edge_weight=w.numpy())
                for p in ("f32", "f64"):
                    case[f"out_{p}"], case[f"loss_{p}"], case[f"dx_{p}"] = r[p]["out"], r[p]["loss"], r[p]["dx"]
                    for k, v in r[p]["grads"].items():
                        case[f"grad_{p}/" + k] = v
                for k, v in r["f32"]["sd"].items():
                    case["sd/" + k] = v
                store_cfg(case, d, mc)
                put(f"step/{name}_{kernel}_{'graph' if use_graph else 'nograph'}", case)

    # ---- dense/: --special_treat dense (main.py:98-103) ---------------------------------------------------
    for name, n, d, deg in datasets:
        for kernel in ("simple", "sigmoid"):
            mc = dict(kernel=kernel, use_graph=True)
            x = torch.randn(n, d, generator=g)
            y = torch.randn(n, generator=g)
            ei = dense_graph(n)
            w = torch.ones(ei.shape[1])                                  # main.py:103

            def run(dt):
                model = build(ref, d, mc, dt)
                xx = leaf(x, dt)
                y_hat = model(xx, ei, w.to(dt))
                cost = torch.mean((y_hat - y.to(dt)) ** 2)
                cost.backward()
                sd = {k: v.detach().float().numpy() for k, v in model.state_dict().items()}
                return dict(out=y_hat.detach().numpy(), loss=cost.detach().numpy(), dx=xx.grad.numpy(),
                            grads=grads_of(model, dt), sd=sd)

            r = both(run)
            case = dict(x=x.numpy(), y=y.numpy(), n=np.array(n))
            for p in ("f32", "f64"):
                case[f"out_{p}"], case[f"loss_{p}"], case[f"dx_{p}"] = r[p]["out"], r[p]["loss"], r[p]["dx"]
                for k, v in r[p]["grads"].items():
                    case[f"grad_{p}/" + k] = v
            for k, v in r["f32"]["sd"].items():
                case["sd/" + k] = v
            store_cfg(case, d, mc)
            put(f"dense/{name}_{kernel}", case)

    # ---- cumul/: T forwards, summed cost, one backward(retain_graph=True) (main.py:94-120) -----------------
    T = 6
    for name, n, d, deg, dynamic in (("chickenpox", 20, 4, 4, False), ("covid", 129, 8, 12, True)):
        for kernel in ("simple", "sigmoid"):
            for use_graph in (True, False):
                mc = dict(kernel=kernel, use_graph=use_graph)
                xs = torch.randn(T, n, d, generator=g)
                ys = torch.randn(T, n, generator=g)
                static = st_graph(g, n, deg)
                eis = [st_graph(g, n, deg) if dynamic else static for _ in range(T)]
                ws = [torch.rand(e.shape[1], generator=g) * 3.0 + 0.05 for e in eis]

                def run(dt):
                    model = build(ref, d, mc, dt)
                    for param in model.parameters():                     # main.py:86-89
                        if param.requires_grad:
                            param.retain_grad()
                    cost_tr = 0
                    outs = []
                    for time in range(T):                                # main.py:94-109
                        y_hat = model(xs[time].to(dt), eis[time], ws[time].to(dt))
                        cost = torch.mean((y_hat - ys[time].to(dt)) ** 2)
                        cost_tr += cost
                        outs.append(y_hat.detach().numpy())
                    cost_tr = cost_tr / (time + 1)                       # main.py:116
                    cost_tr.backward(retain_graph=True)                  # main.py:119
                    sd = {k: v.detach().float().numpy() for k, v in model.state_dict().items()}
                    return dict(out=np.stack(outs), loss=cost_tr.detach().numpy(), grads=grads_of(model, dt), sd=sd)

                r = both(run)
                case = dict(x=xs.numpy(), y=ys.numpy(), dynamic=np.array(dynamic))
                for time in range(T):
                    if dynamic or time == 0:
                        case[f"edge_index/{time}"] = eis[time].numpy()
                    case[f"edge_weight/{time}"] = ws[time].numpy()
                for p in ("f32", "f64"):
                    case[f"out_{p}"], case[f"loss_{p}"] = r[p]["out"], r[p]["loss"]
                    for k, v in r[p]["grads"].items():
                        case[f"grad_{p}/" + k] = v
                for k, v in r["f32"]["sd"].items():
                    case["sd/" + k] = v
                store_cfg(case, d, mc)
                put(f"cumul/{name}_{kernel}_{'graph' if use_graph else 'nograph'}", case)

    np.savez_compressed(os.path.join(OUT, "golden_st.npz"), **flat)
    print("wrote golden_st.npz:", len(flat), "arrays")


if __name__ == "__main__":
    main()

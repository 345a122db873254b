"""Golden vectors for the batched-graph model (f4) by RUNNING THE REFERENCE ITSELF:
`physical particle/difformer-v2.py` imported verbatim with the same three shims as make_golden.py
(torch_sparse.SparseTensor / matmul, torch_geometric.utils.degree).  Build container only.

    python tests/golden/make_golden_v2.py        ->  tests/golden/golden_v2.npz
"""
import importlib.util
import os

import numpy as np
import torch

from make_golden import both, load_reference, rand_graph

REF_V2 = "/root/reference/physical particle/difformer-v2.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_v2():
    load_reference()                                   # installs the shims in sys.modules
    spec = importlib.util.spec_from_file_location("ref_difformer_v2", REF_V2)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def batch_edges(g, n_nodes, per_node):
    """Random edges inside each graph of the batch (global node ids) + self loops."""
    parts, off = [], 0
    for nb in n_nodes.tolist():
        if nb > 1:
            parts.append(rand_graph(g, nb, per_node * nb) + off)
        parts.append(torch.arange(off, off + nb).repeat(2, 1))
        off += nb
    return torch.cat(parts, dim=1).long()


def main():
    ref = load_v2()
    g = torch.Generator().manual_seed(20240926)
    flat = {}

    # ---- TransConv.full_attention --------------------------------------------------------
    shapes = [  # (tag, kernel, n_nodes, H, D)
        ("simple_b5_d64", "simple", [7, 1, 19, 4, 33], 1, 64),
        ("simple_b3_h2_d16", "simple", [12, 12, 5], 2, 16),
        ("simple_b1_d32", "simple", [40], 1, 32),
        ("simple_b6_d10", "simple", [3, 9, 2, 70, 1, 6], 1, 10),
        ("sigmoid_b5_d64", "sigmoid", [7, 1, 19, 4, 33], 1, 64),
        ("sigmoid_b4_h2_d16", "sigmoid", [6, 6, 2, 9], 2, 16),
        ("sigmoid_b40_d64", "sigmoid", [int(v) for v in torch.randint(1, 12, (40,), generator=g)], 1, 64),
    ]
    for tag, kern, nn_, h, d in shapes:
        n_nodes = torch.tensor(nn_)
        n = int(n_nodes.sum())
        q, k, v = (torch.randn(n, h, d, generator=g) for _ in range(3))
        conv = ref.TransConv(d, d, num_heads=h, kernel=kern)
        r = both(lambda dt: conv.full_attention(q.to(dt), k.to(dt), v.to(dt), kern, n_nodes).numpy())
        for key, val in dict(q=q.numpy(), k=k.numpy(), v=v.numpy(), n_nodes=n_nodes.numpy(), out_f32=r["f32"],
                             out_f64=r["f64"], kernel=np.array(kern)).items():
            flat[f"attn/{tag}::{key}"] = val

    # ---- DIFFormer_v2.forward ---------------------------------------------------------------
    cfgs = [
        dict(tag="s_default", n_nodes=[9, 17, 3, 26, 11], f_in=7, hidden=64, num_layers=2, kernel="simple"),
        dict(tag="s_alpha_gw", n_nodes=[5, 5, 14, 1, 8, 20], f_in=12, hidden=32, num_layers=3, kernel="simple",
             alpha=0.3, graph_weight=0.4),
        dict(tag="s_nograph_nobn", n_nodes=[10, 4, 6], f_in=5, hidden=16, num_layers=2, kernel="simple",
             use_graph=False, use_bn=False, use_residual=False),
        dict(tag="a_default", n_nodes=[9, 17, 3, 26, 11], f_in=7, hidden=64, num_layers=2, kernel="sigmoid"),
    ]
    for mc in cfgs:
        mc = dict(mc)
        tag, nn_, f_in, hidden = (mc.pop(k) for k in ("tag", "n_nodes", "f_in", "hidden"))
        n_nodes = torch.tensor(nn_)
        n = int(n_nodes.sum())
        x = torch.randn(n, f_in, generator=g)
        ei = batch_edges(g, n_nodes, 3)
        use_graph = mc.get("use_graph", True)

        def run(dt):
            torch.set_default_dtype(torch.float32)
            torch.manual_seed(321)
            model = ref.DIFFormer_v2(f_in, hidden, hidden, **mc)
            model.reset_parameters()
            with torch.no_grad():
                for bn in model.bns:
                    bn.weight.add_(0.1 * torch.randn(bn.weight.shape, generator=torch.Generator().manual_seed(7)))
                    bn.bias.add_(0.1 * torch.randn(bn.bias.shape, generator=torch.Generator().manual_seed(8)))
            torch.set_default_dtype(dt)
            model = model.to(dt).eval()
            with torch.no_grad():
                out = model(x.to(dt), ei if use_graph else None, n_nodes)
            return out.numpy(), {k: v.float().numpy() for k, v in model.state_dict().items()}

        r = both(run)
        case = dict(x=x.numpy(), edge_index=ei.numpy(), n_nodes=n_nodes.numpy(), out_f32=r["f32"][0], out_f64=r["f64"][0])
        for k, v in r["f32"][1].items():
            case["sd/" + k] = v
        cfg = dict(hidden_channels=hidden, in_channels=f_in, num_layers=2, kernel="simple", alpha=0.5, use_bn=True,
                   use_residual=True, use_weight=True, use_graph=True, graph_weight=-1)
        cfg.update(mc)
        for k, v in cfg.items():
            case["cfg/" + k] = np.array(v)
        for k, v in case.items():
            flat[f"model/{tag}::{k}"] = v
    np.savez_compressed(os.path.join(OUT, "golden_v2.npz"), **flat)
    print("wrote golden_v2.npz:", len(flat), "arrays")


if __name__ == "__main__":
    main()

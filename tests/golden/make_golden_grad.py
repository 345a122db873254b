"""Golden GRADIENTS by RUNNING THE REFERENCE ITSELF under autograd (the training step of
`node classification/main.py:117-131` and `main-batch.py:135-142` differentiates through
`node classification/difformer.py:10-79,113-145,184-209`).

Build container only (needs /root/reference):

    python tests/golden/make_golden_grad.py      ->  tests/golden/golden_grad.npz

The reference files are imported verbatim with the same three shims as make_golden.py; the shimmed
`torch_sparse.matmul` is an `index_add_`, which autograd differentiates with respect to the rows AND the values, so the
gradient of `edge_weight` (difformer.py:73) is the reference's too.  Every case runs under float32 and float64 default
dtype; both results are stored.

Families
  attn/   dq, dk, dv of full_attention_conv for a fixed upstream gradient g  (both kernels, N != L, H > 1)
  gcn/    dx (and d edge_weight) of gcn_conv                                 (weighted, duplicates, zero in-degree)
  model/  loss, d loss / d every parameter, d loss / dx of DIFFormer in train() mode with dropout = 0
          (the two criteria of main.py:121-129: NLL of log_softmax over the training rows; BCE-with-logits)
  v2attn/ dq, dk, dv of TransConv.full_attention (physical particle/difformer-v2.py:71-137)
  v2model/ loss and all gradients of DIFFormer_v2 (MSE against random targets)
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from make_golden import both, load_reference, rand_graph
from make_golden_v2 import batch_edges, load_v2

OUT = os.path.dirname(os.path.abspath(__file__))


def leaf(t, dt):
    return t.to(dt).clone().requires_grad_(True)


def main():
    # the float32 backward passes (index_add / matmul reductions) sum in a thread-dependent order: one thread and the
    # deterministic algorithms make every run of this script produce the same bits
    torch.set_num_threads(1)
    torch.use_deterministic_algorithms(True)
    ref = load_reference()
    ref2 = load_v2()
    g = torch.Generator().manual_seed(20240927)
    flat = {}

    def put(case, d):
        for k, v in d.items():
            flat[f"{case}::{k}"] = v

    # ---- a1 / a2: full_attention_conv ---------------------------------------------------------
    attn_shapes = [  # (tag, kernel, N, L, H, M)
        ("simple_n64_h1_d64", "simple", 64, 64, 1, 64),
        ("simple_n37_h2_d16", "simple", 37, 37, 2, 16),
        ("simple_n50_h3_d10", "simple", 50, 50, 3, 10),
        ("simple_n130_h1_d32", "simple", 130, 130, 1, 32),
        ("sigmoid_n48_h1_d64", "sigmoid", 48, 48, 1, 64),
        ("sigmoid_n30_l45_h2_d16", "sigmoid", 30, 45, 2, 16),
        ("sigmoid_n33_h1_d10", "sigmoid", 33, 33, 1, 10),
        ("sigmoid_n70_l20_h1_d64", "sigmoid", 70, 20, 1, 64),
    ]
    def attn_cases(shapes, g):
        for tag, kern, n, l, h, m in shapes:
            q = torch.randn(n, h, m, generator=g)
            k = torch.randn(l, h, m, generator=g)
            v = torch.randn(l, h, m, generator=g)
            go = torch.randn(n, h, m, generator=g)

            def run(dt):
                qq, kk, vv = leaf(q, dt), leaf(k, dt), leaf(v, dt)
                out = ref.full_attention_conv(qq, kk, vv, kern)
                out.backward(go.to(dt))
                return [t.detach().numpy() for t in (out, qq.grad, kk.grad, vv.grad)]

            r = both(run)
            c = dict(q=q.numpy(), k=k.numpy(), v=v.numpy(), g=go.numpy(), kernel=np.array(kern))
            for p in ("f32", "f64"):
                for name, a in zip(("out", "dq", "dk", "dv"), r[p]):
                    c[f"{name}_{p}"] = a
            put("attn/" + tag, c)

    attn_cases(attn_shapes, g)

    # ---- a3: gcn_conv ----------------------------------------------------------------------------
    gcn_shapes = [  # (tag, N, E, H, D, weighted, isolated)
        ("n60_e300_h1_d64", 60, 300, 1, 64, False, 0),
        ("n60_e300_h1_d64_w", 60, 300, 1, 64, True, 0),
        ("n45_e200_h2_d16_iso", 45, 200, 2, 16, False, 5),
        ("n45_e200_h2_d16_iso_w", 45, 200, 2, 16, True, 5),
        ("n200_e5000_h1_d10_w", 200, 5000, 1, 10, True, 3),
    ]
    for tag, n, e, h, d, weighted, iso in gcn_shapes:
        x = torch.randn(n, h, d, generator=g)
        ei = rand_graph(g, n, e, iso)
        go = torch.randn(n, h, d, generator=g)
        w = (torch.rand(e, generator=g) + 0.1) if weighted else None      # strictly positive: 0 * inf has no gradient

        def run(dt):
            xx = leaf(x, dt)
            ww = None if w is None else leaf(w, dt)
            out = ref.gcn_conv(xx, ei, ww)
            out.backward(go.to(dt))
            res = [out.detach().numpy(), xx.grad.numpy()]
            if ww is not None:
                res.append(ww.grad.numpy())
            return res

        r = both(run)
        c = dict(x=x.numpy(), edge_index=ei.numpy(), g=go.numpy())
        if w is not None:
            c["edge_weight"] = w.numpy()
        for p in ("f32", "f64"):
            for name, a in zip(("out", "dx", "dw"), r[p]):
                c[f"{name}_{p}"] = a
        put("gcn/" + tag, c)

    # ---- a4 / a5: DIFFormer training step (main.py:117-131) ------------------------------------
    model_cfgs = [
        dict(tag="s_default", n=96, f_in=24, hidden=64, c=7, num_layers=2, num_heads=1, kernel="simple"),
        dict(tag="s_cli_flags", n=80, f_in=64, hidden=64, c=5, num_layers=3, num_heads=1, kernel="simple",
             use_bn=False, use_residual=False, use_weight=False, use_graph=True),
        dict(tag="s_h2_src_gw", n=70, f_in=12, hidden=16, c=4, num_layers=2, num_heads=2, kernel="simple",
             graph_weight=0.3, use_source=True, alpha=0.3),
        dict(tag="s_nograph_l4", n=128, f_in=20, hidden=64, c=10, num_layers=4, num_heads=1, kernel="simple",
             use_graph=False),
        dict(tag="s_bce_l3", n=90, f_in=8, hidden=64, c=12, num_layers=3, num_heads=1, kernel="simple", loss="bce"),
        dict(tag="s_d10_weighted", n=40, f_in=6, hidden=10, c=2, num_layers=2, num_heads=1, kernel="simple",
             weighted=True, use_source=True),
        dict(tag="a_default", n=72, f_in=24, hidden=64, c=7, num_layers=2, num_heads=1, kernel="sigmoid"),
        dict(tag="a_h2_weighted", n=50, f_in=10, hidden=16, c=3, num_layers=2, num_heads=2, kernel="sigmoid",
             weighted=True),
        dict(tag="a_nobn_src", n=44, f_in=9, hidden=32, c=3, num_layers=2, num_heads=1, kernel="sigmoid",
             use_bn=False, use_source=True, graph_weight=0.6),
    ]
    def model_cases(cfgs, g):
        for mc in cfgs:
            mc = dict(mc)
            tag, n, f_in, hidden, c = (mc.pop(k) for k in ("tag", "n", "f_in", "hidden", "c"))
            weighted = mc.pop("weighted", False)
            loss_kind = mc.pop("loss", "nll")
            x = torch.randn(n, f_in, generator=g)
            ei = rand_graph(g, n, 6 * n, isolated=2)
            ei = torch.cat([ei, torch.arange(n - 2).repeat(2, 1)], dim=1)            # self loops as main.py:76 adds
            w = (torch.rand(ei.shape[1], generator=g) + 0.05) if weighted else None
            use_graph = mc.get("use_graph", True)
            train_idx = torch.randperm(n, generator=g)[: n // 2]
            if loss_kind == "bce":
                y = (torch.rand(n, c, generator=g) < 0.3).float()                    # multi-label (ogbn-proteins, main.py:121-127)
            else:
                y = torch.randint(0, c, (n,), generator=g)

            def run(dt):
                torch.set_default_dtype(torch.float32)
                torch.manual_seed(123)
                model = ref.DIFFormer(f_in, hidden, c, dropout=0.0, **mc)
                model.reset_parameters()
                with torch.no_grad():
                    for bn in model.bns:
                        bn.weight.add_(0.1 * torch.randn(bn.weight.shape, generator=torch.Generator().manual_seed(7)))
                        bn.bias.add_(0.1 * torch.randn(bn.bias.shape, generator=torch.Generator().manual_seed(8)))
                torch.set_default_dtype(dt)
                model = model.to(dt).train()                                         # main.py:115
                xx = leaf(x, dt)
                ww = None if w is None else leaf(w, dt)
                out = model(xx, ei if use_graph else None, ww)                       # main.py:118
                if loss_kind == "bce":
                    loss = F.binary_cross_entropy_with_logits(out[train_idx], y[train_idx].to(dt))      # main.py:124-125
                else:
                    loss = F.nll_loss(F.log_softmax(out, dim=1)[train_idx], y[train_idx])                # main.py:127-129
                loss.backward()                                                      # main.py:130
                grads = {k: (torch.zeros_like(p) if p.grad is None else p.grad).float().numpy() if dt == torch.float32
                         else (torch.zeros_like(p) if p.grad is None else p.grad).numpy() for k, p in model.named_parameters()}
                sd = {k: v.detach().float().numpy() for k, v in model.state_dict().items()}
                return dict(out=out.detach().numpy(), loss=loss.detach().numpy(), dx=xx.grad.numpy(),
                            dw=None if ww is None else ww.grad.numpy(), grads=grads, sd=sd)

            r = both(run)
            case = dict(x=x.numpy(), edge_index=ei.numpy(), train_idx=train_idx.numpy(), y=y.numpy(),
                        loss_kind=np.array(loss_kind))
            if w is not None:
                case["edge_weight"] = w.numpy()
            for p in ("f32", "f64"):
                case[f"out_{p}"], case[f"loss_{p}"], case[f"dx_{p}"] = r[p]["out"], r[p]["loss"], r[p]["dx"]
                if r[p]["dw"] is not None:
                    case[f"dw_{p}"] = r[p]["dw"]
                for k, v in r[p]["grads"].items():
                    case[f"grad_{p}/" + k] = v
            for k, v in r["f32"]["sd"].items():
                case["sd/" + k] = v
            cfg = dict(hidden_channels=hidden, out_channels=c, in_channels=f_in, num_layers=2, num_heads=1,
                       kernel="simple", alpha=0.5, use_bn=True, use_residual=True, use_weight=True, use_graph=True,
                       graph_weight=-1, use_source=False)
            cfg.update(mc)
            for k, v in cfg.items():
                case["cfg/" + k] = np.array(v)
            put("model/" + tag, case)

    model_cases(model_cfgs, g)

    # ---- f4: TransConv.full_attention of the batched model -------------------------------------------
    v2_shapes = [  # (tag, kernel, n_nodes, H, D)
        ("simple_b5_d64", "simple", [7, 1, 19, 4, 33], 1, 64),
        ("simple_b3_h2_d16", "simple", [12, 12, 5], 2, 16),
        ("simple_b6_d10", "simple", [3, 9, 2, 70, 1, 6], 1, 10),
        ("sigmoid_b5_d64", "sigmoid", [7, 1, 19, 4, 33], 1, 64),
        ("sigmoid_b4_h2_d16", "sigmoid", [6, 6, 2, 9], 2, 16),
    ]
    for tag, kern, nn_, h, d in v2_shapes:
        n_nodes = torch.tensor(nn_)
        n = int(n_nodes.sum())
        q, k, v, go = (torch.randn(n, h, d, generator=g) for _ in range(4))
        conv = ref2.TransConv(d, d, num_heads=h, kernel=kern)

        def run(dt):
            qq, kk, vv = leaf(q, dt), leaf(k, dt), leaf(v, dt)
            out = conv.full_attention(qq, kk, vv, kern, n_nodes)
            out.backward(go.to(dt))
            return [t.detach().numpy() for t in (out, qq.grad, kk.grad, vv.grad)]

        r = both(run)
        c = dict(q=q.numpy(), k=k.numpy(), v=v.numpy(), g=go.numpy(), n_nodes=n_nodes.numpy(), kernel=np.array(kern))
        for p in ("f32", "f64"):
            for name, a in zip(("out", "dq", "dk", "dv"), r[p]):
                c[f"{name}_{p}"] = a
        put("v2attn/" + tag, c)

    v2_cfgs = [
        dict(tag="s_default", n_nodes=[9, 17, 3, 26, 11], f_in=7, hidden=64, num_layers=2, kernel="simple"),
        dict(tag="s_alpha_gw", n_nodes=[5, 5, 14, 1, 8, 20], f_in=12, hidden=32, num_layers=3, kernel="simple",
             alpha=0.3, graph_weight=0.4),
        dict(tag="a_default", n_nodes=[9, 17, 3, 26, 11], f_in=7, hidden=64, num_layers=2, kernel="sigmoid"),
    ]
    for mc in v2_cfgs:
        mc = dict(mc)
        tag, nn_, f_in, hidden = (mc.pop(k) for k in ("tag", "n_nodes", "f_in", "hidden"))
        n_nodes = torch.tensor(nn_)
        n = int(n_nodes.sum())
        x = torch.randn(n, f_in, generator=g)
        target = torch.randn(n, hidden, generator=g)
        ei = batch_edges(g, n_nodes, 3)

        def run(dt):
            torch.set_default_dtype(torch.float32)
            torch.manual_seed(321)
            model = ref2.DIFFormer_v2(f_in, hidden, hidden, dropout=0.0, **mc)
            model.reset_parameters()
            with torch.no_grad():
                for bn in model.bns:
                    bn.weight.add_(0.1 * torch.randn(bn.weight.shape, generator=torch.Generator().manual_seed(7)))
                    bn.bias.add_(0.1 * torch.randn(bn.bias.shape, generator=torch.Generator().manual_seed(8)))
            torch.set_default_dtype(dt)
            model = model.to(dt).train()
            xx = leaf(x, dt)
            out = model(xx, ei, n_nodes)
            loss = F.mse_loss(out, target.to(dt))
            loss.backward()
            grads = {k: (torch.zeros_like(p) if p.grad is None else p.grad).numpy() for k, p in model.named_parameters()}
            sd = {k: v.detach().float().numpy() for k, v in model.state_dict().items()}
            return dict(out=out.detach().numpy(), loss=loss.detach().numpy(), dx=xx.grad.numpy(), grads=grads, sd=sd)

        r = both(run)
        case = dict(x=x.numpy(), edge_index=ei.numpy(), n_nodes=n_nodes.numpy(), target=target.numpy())
        for p in ("f32", "f64"):
            case[f"out_{p}"], case[f"loss_{p}"], case[f"dx_{p}"] = r[p]["out"], r[p]["loss"], r[p]["dx"]
            for k, v in r[p]["grads"].items():
                case[f"grad_{p}/" + k] = v
        for k, v in r["f32"]["sd"].items():
            case["sd/" + k] = v
        cfg = dict(hidden_channels=hidden, in_channels=f_in, num_layers=2, kernel="simple", alpha=0.5, use_bn=True,
                   use_residual=True, use_weight=True, use_graph=True, graph_weight=-1)
        cfg.update(mc)
        for k, v in cfg.items():
            case["cfg/" + k] = np.array(v)
        put("v2model/" + tag, case)

    # ---- round 4: heads wider than 64 columns (run.sh trains at hidden 128 / 300) -- drawn from a generator of their own,
    # AFTER everything above, so that the arrays of the earlier cases keep their bits
    g_wide = torch.Generator().manual_seed(20260926)
    attn_cases([("simple_n70_h1_d128", "simple", 70, 70, 1, 128),
                ("simple_n40_h1_d300", "simple", 40, 40, 1, 300),
                ("simple_n33_h2_d100", "simple", 33, 33, 2, 100)], g_wide)
    model_cases([dict(tag="s_h128", n=100, f_in=20, hidden=128, c=5, num_layers=2, num_heads=1, kernel="simple")], g_wide)

    np.savez_compressed(os.path.join(OUT, "golden_grad.npz"), **flat)
    print("wrote golden_grad.npz:", len(flat), "arrays")


if __name__ == "__main__":
    main()

"""Golden vectors for the `image and text` task folder, by RUNNING THAT FOLDER'S OWN COPY of the model:
`image and text/difformer.py` (the subset copy without graph_weight / use_source) imported verbatim with the three shims of
make_golden.py, in the configuration its DIFFormer-a lines use (`image and text/run.sh:17,35,54`):
`--kernel sigmoid --num_layers 2 --hidden_channels 300 | 400 --use_residual --use_bn --alpha 0.5 --dropout 0.0`, WITHOUT
`--use_graph` and WITHOUT `--use_weight` (parse.py:64-65, 111-112: value = the layer input itself, difformer.py:116; the k-NN
edge list of main.py:52-54 is handed over and ignored), one head; the training step of `main.py:97-110`
(log_softmax + NLLLoss on the training nodes, backward).

Build container only (needs /root/reference):

    python tests/golden/make_golden_it.py      ->  tests/golden/golden_it.npz

Families
  attn/   full_attention_conv(qs, ks, vs, 'sigmoid') at 300 / 400 columns (N != L too), with dq, dk, dv under a random cotangent
  step/   one training step of the model: logits, loss, every parameter gradient and dx
          (hidden 300 with two layers as the scripts run it; hidden 400 with ONE layer to keep the file small)
The model's parameters are inputs and stored in float32; outputs and gradients come from the float32 run (`*_f32`) and from the
float64 run (`*_f64`, stored rounded to float32: 6e-8, far inside the 1e-4 bar they are used at).
"""
import importlib.util
import os

import numpy as np
import torch

from make_golden import both, load_reference

REF_IT = "/root/reference/image and text/difformer.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_it():
    load_reference()                     # installs the shims into sys.modules
    spec = importlib.util.spec_from_file_location("ref_difformer_it", REF_IT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def leaf(t, dt):
    return t.to(dt).clone().requires_grad_(True)


def main():
    torch.set_num_threads(1)
    torch.use_deterministic_algorithms(True)
    ref = load_it()
    g = torch.Generator().manual_seed(20261001)
    flat = {}

    def put(case, d):
        for k, v in d.items():
            flat[f"{case}::{k}"] = v

    # ---- attn/: difformer.py:45-56 at the scripts' widths ----------------------------------------------------------
    for tag, n, l, m in (("n40_l56_d300", 40, 56, 300), ("n33_d400", 33, 33, 400), ("n70_l20_d129", 70, 20, 129)):
        q = torch.randn(n, 1, m, generator=g) * 0.2
        k = torch.randn(l, 1, m, generator=g) * 0.2
        v = torch.randn(l, 1, m, generator=g)
        go = torch.randn(n, 1, m, generator=g)

        def run(dt):
            qq, kk, vv = leaf(q, dt), leaf(k, dt), leaf(v, dt)
            out = ref.full_attention_conv(qq, kk, vv, "sigmoid")
            out.backward(go.to(dt))
            f = (lambda t: t.detach().float().numpy())
            return dict(out=f(out), dq=f(qq.grad), dk=f(kk.grad), dv=f(vv.grad))

        r = both(run)
        case = dict(q=q.numpy(), k=k.numpy(), v=v.numpy(), g=go.numpy())
        for p in ("f32", "f64"):
            for key, val in r[p].items():
                case[f"{key}_{p}"] = val
        put("attn/" + tag, case)

    # ---- step/: main.py:97-110 -------------------------------------------------------------------------------------
    for tag, n, d, hidden, layers, c in (("h300_l2", 96, 24, 300, 2, 10), ("h400_l1", 80, 16, 400, 1, 10)):
        x = torch.randn(n, d, generator=g)
        y = torch.randint(0, c, (n,), generator=g)
        train_idx = torch.randperm(n, generator=g)[: n // 2]
        ei = torch.randint(0, n, (2, 5 * n), generator=g)                 # main.py:52-54: there, and unused without --use_graph

        def run(dt):
            torch.set_default_dtype(torch.float32)
            torch.manual_seed(123)                                       # run.sh: --seed 123
            model = ref.DIFFormer(d, hidden, c, num_layers=layers, alpha=0.5, dropout=0.0, num_heads=1, kernel="sigmoid",
                                  use_bn=True, use_residual=True, use_graph=False, use_weight=False)      # parse.py:64-65
            model.reset_parameters()                                     # main.py:94
            with torch.no_grad():
                for bn in model.bns:
                    bn.weight.add_(0.1 * torch.randn(bn.weight.shape, generator=torch.Generator().manual_seed(7)))
                    bn.bias.add_(0.1 * torch.randn(bn.bias.shape, generator=torch.Generator().manual_seed(8)))
            torch.set_default_dtype(dt)
            model = model.to(dt).train()                                 # main.py:97
            xx = leaf(x, dt)
            out = model(xx, ei)                                          # main.py:101
            loss = torch.nn.NLLLoss()(torch.nn.functional.log_softmax(out, dim=1)[train_idx], y[train_idx])    # main.py:107-109
            loss.backward()                                              # main.py:113
            f = (lambda t: t.detach().float().numpy())
            grads = {k_: f(torch.zeros_like(p_) if p_.grad is None else p_.grad) for k_, p_ in model.named_parameters()}
            sd = {k_: f(v_) for k_, v_ in model.state_dict().items()}
            return dict(out=f(out), loss=f(loss), dx=f(xx.grad), grads=grads, sd=sd)

        r = both(run)
        case = dict(x=x.numpy(), y=y.numpy(), train_idx=train_idx.numpy(), edge_index=ei.numpy())
        for p in ("f32", "f64"):
            case[f"out_{p}"], case[f"loss_{p}"], case[f"dx_{p}"] = r[p]["out"], r[p]["loss"], r[p]["dx"]
        for k_, v_ in r["f64"]["grads"].items():                          # (the float32 run's parameter gradients are not stored: size)
            case["grad_f64/" + k_] = v_
        for k_, v_ in r["f32"]["sd"].items():
            case["sd/" + k_] = v_
        cfg = dict(in_channels=d, hidden_channels=hidden, out_channels=c, num_layers=layers, num_heads=1, kernel="sigmoid", alpha=0.5,
                   use_bn=True, use_residual=True, use_weight=False, use_graph=False, graph_weight=-1, use_source=False)
        for k_, v_ in cfg.items():
            case["cfg/" + k_] = np.array(v_)
        put("step/" + tag, case)

    path = os.path.join(OUT, "golden_it.npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path}: {len(flat)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()

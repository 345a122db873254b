"""The table VERDICT r3 asked for (soft spot i): error of every Wq / Wk gradient of the reference's own training-step fixtures
(tests/golden/golden_grad.npz, float64 run of the reference) on the CLOSED-FORM training path and on the OPERATOR path
(DIFFORMER_CLOSED_FORM_TRAINING=0), side by side with the error of the reference's own float32 run, at the two floors
conftest.grad_err has had (1e-6 and 2e-6 of the step's largest gradient entry).
    python scripts/exp_grad_floor.py            (spawns itself once per path: the switch is read at import)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    import torch.nn.functional as F
    from conftest import grad_err, grad_scale, load_golden, split_model_case
    from difformer_amd import DIFFormer
    dev = torch.device("cuda:0")
    GRAD = load_golden("grad")
    out = {}
    for name in sorted(n for n in GRAD if n.startswith("model/")):
        c = GRAD[name]
        cfg, sd = split_model_case(c)
        if str(cfg["kernel"]) != "simple":
            continue
        kw = {k: cfg[k] for k in ("num_layers", "num_heads", "kernel", "alpha", "use_bn", "use_residual", "use_weight", "use_graph",
                                  "graph_weight", "use_source")}
        kw["kernel"] = str(kw["kernel"])
        model = DIFFormer(int(cfg["in_channels"]), int(cfg["hidden_channels"]), int(cfg["out_channels"]), dropout=0.0, **kw)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        model = model.to(dev).train()
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        ei = t(c["edge_index"]) if cfg["use_graph"] else None
        w = t(c["edge_weight"]) if "edge_weight" in c else None
        o = model(t(c["x"]), ei, w)
        idx = t(c["train_idx"])
        if str(c["loss_kind"]) == "bce":
            loss = F.binary_cross_entropy_with_logits(o[idx], t(c["y"])[idx].to(o.dtype))
        else:
            loss = F.nll_loss(F.log_softmax(o, dim=1)[idx], t(c["y"])[idx])
        loss.backward()
        gmax = grad_scale(c)
        for k, p in model.named_parameters():
            if ".Wq." in k or ".Wk." in k:
                ref = c["grad_f64/" + k]
                out[f"{name}:{k}"] = {"size": float(np.abs(ref).max() / gmax),
                                      "err@1e-6": grad_err(p.grad.cpu().numpy(), ref, gmax, floor=1e-6),
                                      "err@2e-6": grad_err(p.grad.cpu().numpy(), ref, gmax, floor=2e-6),
                                      "ref32@1e-6": grad_err(c["grad_f32/" + k], ref, gmax, floor=1e-6)}
    print("RESULT" + json.dumps(out))
    sys.exit(0)

res = {}
for flag in ("1", "0"):
    env = dict(os.environ, DIFFORMER_CLOSED_FORM_TRAINING=flag)
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        print(r.stdout[-2000:], r.stderr[-3000:]); sys.exit(1)
    res[flag] = json.loads(line[0][6:])
print("| case : parameter | max entry / largest gradient of the step | closed form, floor 1e-6 | closed form, floor 2e-6 | operator path, floor 1e-6 | operator path, floor 2e-6 | reference's own fp32 run, floor 1e-6 |")
print("|---|---|---|---|---|---|---|")
worst = {"cf1": 0, "cf2": 0, "op1": 0, "op2": 0, "ref": 0}
for k in sorted(res["1"]):
    a, b = res["1"][k], res["0"][k]
    worst = {"cf1": max(worst["cf1"], a["err@1e-6"]), "cf2": max(worst["cf2"], a["err@2e-6"]), "op1": max(worst["op1"], b["err@1e-6"]),
             "op2": max(worst["op2"], b["err@2e-6"]), "ref": max(worst["ref"], a["ref32@1e-6"])}
    if max(a["err@1e-6"], b["err@1e-6"], a["ref32@1e-6"]) > 2e-5:
        print(f"| {k.replace('model/', '')} | {a['size']:.1e} | {a['err@1e-6']:.2e} | {a['err@2e-6']:.2e} | {b['err@1e-6']:.2e} | {b['err@2e-6']:.2e} | {a['ref32@1e-6']:.2e} |")
print(f"| WORST over {len(res['1'])} Wq / Wk tensors (rows above: those beyond 2e-5 on any path) | | {worst['cf1']:.2e} | {worst['cf2']:.2e} | {worst['op1']:.2e} | {worst['op2']:.2e} | {worst['ref']:.2e} |")

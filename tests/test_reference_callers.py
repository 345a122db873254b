"""The reference's own CALLERS executed against the drop-in: `parse.py` of every task folder is
loaded verbatim with `difformer` resolving to `dropin/difformer.py` (or `dropin/physical_particle/difformer.py`), every
`--method difformer` command line of the folder's `run.sh` goes through the reference's own `parser_add_main_args`, and
`parse_method(...)` builds the model exactly as `main.py` does (node classification/main.py:30-31,82; image and
text/main.py:31-32,77; spatial-temporal/main.py:24-25,70; physical particle/main.py:27-28,53).  Then what the scripts do with
it: `reset_parameters()` (main.py:110), one forward with the script's call shape (main.py:118; spatial-temporal/main.py:105
passes `edge_attr` positionally; physical particle/main.py:85 passes `n_nodes`), and a strict `state_dict` round trip
(test_large_dataset.py:88).

CPU test: the arithmetic runs on tests/fake_backend.OracleBackend; skipped where /root/reference is absent (GPU box).
The baseline-GNN zoo the reference's `gnns.py` / `models.py` import (torch_geometric.nn, torch_sparse) is out of scope
and absent from this image: those imports resolve to inert stubs -- only the `difformer` branch of parse_method is run.
"""
import argparse
import importlib.util
import os
import re
import shlex
import sys
import types

import pytest
import torch
import torch.nn as nn

from fake_backend import OracleBackend

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")


@pytest.fixture()
def fake_backend(monkeypatch):
    from difformer_amd import ops
    be = OracleBackend()
    monkeypatch.setattr(ops, "_BACKEND", be)
    ops.csr_cache.clear()
    yield be
    ops.csr_cache.clear()


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


class _Inert(nn.Module):
    """Stands in for the torch_geometric layers of the baseline zoo (never instantiated here)."""

    def __init__(self, *a, **k):
        super().__init__()


def _load(folder, dropin_file, extra=()):
    """-> the folder's `parse` module, loaded from the reference source with `difformer` = the drop-in."""
    saved = {k: sys.modules.get(k) for k in ("difformer", "gnns", "models", "parse", "data_utils", "torch_sparse",
                                              "torch_geometric", "torch_geometric.nn", "torch_geometric.nn.conv",
                                              "torch_geometric.nn.conv.gcn_conv", "torch_geometric.utils")}
    layers = {n: _Inert for n in ("GCNConv", "SGConv", "GATConv", "JumpingKnowledge", "APPNP", "MessagePassing", "GINConv",
                                  "global_mean_pool", "global_add_pool", "global_max_pool", "InstanceNorm", "SAGEConv")}
    tg_nn = _stub("torch_geometric.nn", **layers)
    tg_nn.__getattr__ = lambda name: _Inert                           # whatever else the zoo imports
    tg_conv = _stub("torch_geometric.nn.conv", MessagePassing=_Inert)
    tg_gcn = _stub("torch_geometric.nn.conv.gcn_conv", gcn_norm=lambda *a, **k: None)
    tg_utils = _stub("torch_geometric.utils", to_dense_adj=None, dense_to_sparse=None, degree=None)
    tg = _stub("torch_geometric", nn=tg_nn, utils=tg_utils)
    ts = _stub("torch_sparse", SparseTensor=object, matmul=None)
    sys.modules.update({"torch_sparse": ts, "torch_geometric": tg, "torch_geometric.nn": tg_nn,
                        "torch_geometric.nn.conv": tg_conv, "torch_geometric.nn.conv.gcn_conv": tg_gcn,
                        "torch_geometric.utils": tg_utils})
    spec = importlib.util.spec_from_file_location("difformer", os.path.join(ROOT, "dropin", dropin_file))
    dropin = importlib.util.module_from_spec(spec)
    sys.modules["difformer"] = dropin                                   # what `from difformer import *` finds
    spec.loader.exec_module(dropin)
    path = os.path.join(REF, folder)
    sys.path.insert(0, path)
    try:
        for dep in extra + ("parse",):
            sys.modules.pop(dep, None)
            sp = importlib.util.spec_from_file_location(dep, os.path.join(path, dep + ".py"))
            mod = importlib.util.module_from_spec(sp)
            sys.modules[dep] = mod
            sp.loader.exec_module(mod)
        return sys.modules["parse"], dropin
    finally:
        sys.path.remove(path)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _difformer_commands(folder, script="run.sh"):
    """The `--method difformer` command lines of a run script -> list of argv lists (shell variables -> '1')."""
    text = open(os.path.join(REF, folder, script)).read().replace("\\\n", " ")
    out = []
    for line in text.splitlines():
        line = line.strip()
        m = re.match(r"python\s+(main(?:-batch)?\.py|test_large_dataset\.py)\s+(.*)", line)
        if not m or "--method difformer" not in line:
            continue
        args = re.sub(r"\$\{?\w+\}?", "1", m.group(2))
        out.append(shlex.split(args))
    return out


def _args(parse, argv):
    parser = argparse.ArgumentParser()
    parse.parser_add_main_args(parser)
    return parser.parse_args(argv)


def _graph(n, e, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.cat([torch.randint(0, n, (2, e), generator=g), torch.arange(n).repeat(2, 1)], dim=1)


def _roundtrip(model, rebuild):
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    other = rebuild()
    missing = other.load_state_dict(sd, strict=True)                    # test_large_dataset.py:88
    assert not missing.missing_keys and not missing.unexpected_keys
    return other


NODE_CMDS = _difformer_commands("node classification") if os.path.isdir(REF) else []


def test_run_scripts_were_found():
    assert len(NODE_CMDS) >= 14                                         # run.sh: cora ... chameleon
    assert len(_difformer_commands("image and text")) >= 7
    assert len(_difformer_commands("spatial-temporal")) >= 12
    assert len(_difformer_commands("physical particle")) >= 6
    assert len(_difformer_commands("node classification", "run_test_large.sh")) == 2


@pytest.mark.parametrize("script", ["run.sh", "run_test_large.sh"])
def test_node_classification_callers(script, fake_backend):
    parse, dropin = _load("node classification", "difformer.py", extra=("gnns",))
    assert parse.DIFFormer is dropin.DIFFormer                          # `from difformer import *` (parse.py:2)
    n, c, d = 60, 5, 12
    x, ei = torch.randn(n, d), _graph(n, 200)
    for argv in _difformer_commands("node classification", script):
        args = _args(parse, argv)
        model = parse.parse_method(args, n, c, d, torch.device("cpu"))   # parse.py:4-8
        assert isinstance(model, dropin.DIFFormer) and len(model.convs) == args.num_layers
        assert model.convs[0].use_graph == args.use_graph and model.convs[0].use_weight == args.use_weight
        assert model.convs[0].kernel == args.kernel and model.convs[0].num_heads == args.num_heads
        model.reset_parameters()                                         # main.py:110
        model.eval()
        xin = x
        with torch.no_grad():
            out = model(xin, ei)                                         # main.py:118 / eval.py:10
        assert out.shape == (n, c) and torch.isfinite(out).all()
        other = _roundtrip(model, lambda: parse.parse_method(args, n, c, d, torch.device("cpu")))
        with torch.no_grad():
            assert torch.equal(other.eval()(xin, ei), out)


def test_image_and_text_callers(fake_backend):
    parse, dropin = _load("image and text", "difformer.py", extra=("gnns", "data_utils"))
    n, c, d = 48, 10, 20
    x = torch.randn(n, d)
    for argv in _difformer_commands("image and text"):
        args = _args(parse, argv)
        model = parse.parse_method(args, None, n, c, d, torch.device("cpu"))     # parse.py:5, :64-65
        assert isinstance(model, dropin.DIFFormer)
        model.reset_parameters()
        model.eval()
        ei = _graph(n, 150) if args.use_graph else None                  # Readme.md:54-55: no graph -> edge_index None
        with torch.no_grad():
            out = model(x, ei)                                           # main.py:101
        assert out.shape == (n, c) and torch.isfinite(out).all()
        _roundtrip(model, lambda: parse.parse_method(args, None, n, c, d, torch.device("cpu")))


def test_spatial_temporal_callers(fake_backend):
    parse, dropin = _load("spatial-temporal", "difformer.py", extra=("gnns",))
    assert parse.DIFFormer is dropin.DIFFormer                          # `from difformer import DIFFormer` (parse.py:2)
    n, c, d = 20, 1, 4
    x, ei = torch.randn(n, d), _graph(n, 60)
    edge_attr = torch.rand(ei.shape[1]) + 0.1
    for script in ("run.sh", "run_hyper_search.sh"):
        for argv in _difformer_commands("spatial-temporal", script):
            args = _args(parse, argv)
            model, suffix = parse.parse_method(args, n, c, d, torch.device("cpu"))   # parse.py:54-57
            assert isinstance(model, dropin.DIFFormer) and "kernel" + args.kernel in suffix
            model.reset_parameters()
            model.eval()
            with torch.no_grad():
                out = model(x, ei, edge_attr)                            # main.py:105: edge_attr positional
            assert out.shape == (n, c) and torch.isfinite(out).all()
            _roundtrip(model, lambda: parse.parse_method(args, n, c, d, torch.device("cpu"))[0])


def test_physical_particle_callers(fake_backend):
    parse, dropin = _load("physical particle", os.path.join("physical_particle", "difformer.py"), extra=("models",))
    assert parse.DIFFormer_v2 is dropin.DIFFormer_v2                    # parse.py:3
    n_nodes = torch.tensor([7, 3, 12, 5])
    n, d = int(n_nodes.sum()), 6
    x = torch.randn(n, d)
    parts, off = [], 0
    for nb in n_nodes.tolist():
        parts.append(_graph(nb, 3 * nb, seed=nb) + off)
        off += nb
    ei = torch.cat(parts, dim=1)
    for argv in _difformer_commands("physical particle"):
        args = _args(parse, argv)
        model = parse.parse_method(args, 1, d, torch.device("cpu"))      # parse.py:5, :20-32
        assert isinstance(model, dropin.DIFFormer_v2)
        model.reset_parameters()
        model.eval()
        with torch.no_grad():
            out = model(x, ei, n_nodes)                                  # main.py:85 (through models.GraphModel)
        assert out.shape == (n, args.hidden_channels) and torch.isfinite(out).all()
        _roundtrip(model, lambda: parse.parse_method(args, 1, d, torch.device("cpu")))


def test_reference_evaluate_and_training_loop_run_on_the_dropin(fake_backend):
    """`node classification/eval.py::evaluate` loaded verbatim and called as main.py:132 calls it, between optimisation steps
    written as main.py:114-131 writes them (model.train(), zero_grad, forward, log_softmax + NLL on the training split,
    backward, Adam step), on the model parse_method builds for the `cora` line of run.sh, with that line's learning rate: the
    loss goes down over twelve steps and the evaluation returns accuracies and a finite validation loss -- the drop-in under the reference's own evaluation code and call sequence."""
    import torch.nn.functional as F
    parse, dropin = _load("node classification", "difformer.py", extra=("gnns",))
    spec = importlib.util.spec_from_file_location("ref_eval", os.path.join(REF, "node classification", "eval.py"))
    ref_eval = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_eval)
    n, c, d = 80, 4, 10
    g = torch.Generator().manual_seed(5)
    y = torch.randint(0, c, (n, 1), generator=g)
    x = torch.randn(n, d, generator=g) + 2.0 * F.one_hot(y.squeeze(1), d).float()          # learnable labels
    dataset = types.SimpleNamespace(graph={"node_feat": x, "edge_index": _graph(n, 300, seed=6), "num_nodes": n}, label=y)
    perm = torch.randperm(n, generator=g)
    split_idx = {"train": perm[:40], "valid": perm[40:60], "test": perm[60:]}
    argv = next(a for a in NODE_CMDS if "cora" in a)
    args = _args(parse, argv)
    model = parse.parse_method(args, n, c, d, torch.device("cpu"))
    assert isinstance(model, dropin.DIFFormer)
    criterion = nn.NLLLoss()                                             # main.py:95
    eval_func = lambda y_true, out: float((out.argmax(dim=-1, keepdim=True) == y_true).float().mean())   # data_utils.eval_acc
    torch.manual_seed(0)
    model.reset_parameters()
    optimizer = torch.optim.Adam(model.parameters(), weight_decay=args.weight_decay, lr=args.lr)
    train_idx = split_idx["train"]
    losses = []
    for epoch in range(12):
        model.train()                                                    # main.py:115-131
        optimizer.zero_grad()
        out = model(dataset.graph["node_feat"], dataset.graph["edge_index"])
        out = F.log_softmax(out, dim=1)
        loss = criterion(out[train_idx], dataset.label.squeeze(1)[train_idx])
        loss.backward()
        optimizer.step()
        result = ref_eval.evaluate(model, dataset, split_idx, eval_func, criterion, args)    # main.py:132
        losses.append(float(loss))
    train_acc, valid_acc, test_acc, valid_loss, out = result
    assert losses[-1] < losses[0] and all(0.0 <= a <= 1.0 for a in (train_acc, valid_acc, test_acc))
    assert torch.isfinite(valid_loss) and out.shape == (n, c)
    assert fake_backend.closed_form_calls > 0                            # the `simple` layers of the cora line: closed form


# ------------------------------------------------------------------------------------------------------------------
# CPU-RESIDENT callers (eval.py:43 through evaluate_cpu, test_large_dataset.py:93): model and graph in host
# memory.  The product stages such a call onto the GPU (difformer_amd/staging.py); here the staging logic itself -- the
# device twin and its refresh after optimiser steps / load_state_dict / .to(), the operand cache, the gradient routing --
# runs with the "device" forced to the host so that the test backend can do the arithmetic.  The same call sequences
# run on the real backend in tests/test_gpu_staging.py.
# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture()
def staged_on_host(fake_backend, monkeypatch):
    from difformer_amd import staging
    monkeypatch.setattr(staging, "FORCE_DEVICE", torch.device("cpu"))
    staging.operands.clear()
    yield staging
    staging.operands.clear()


def _ref_eval_module():
    spec = importlib.util.spec_from_file_location("ref_eval", os.path.join(REF, "node classification", "eval.py"))
    ref_eval = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_eval)
    return ref_eval


def test_reference_evaluate_cpu_runs_staged(staged_on_host):
    """`eval.py::evaluate_cpu` verbatim, called as main-batch.py:144-145 calls it after an epoch of optimiser steps: the
    model goes `.to(cpu)`, the full graph is forwarded from host memory, the result is a host tensor -- and equals the plain
    (unstaged) forward of the same parameters; a second evaluation after another optimiser step sees the new parameters."""
    import torch.nn.functional as F
    from difformer_amd import staging
    parse, dropin = _load("node classification", "difformer.py", extra=("gnns",))
    ref_eval = _ref_eval_module()
    n, c, d = 70, 3, 9
    g = torch.Generator().manual_seed(11)
    y = torch.randint(0, c, (n, 1), generator=g)
    x = torch.randn(n, d, generator=g)
    dataset = types.SimpleNamespace(graph={"node_feat": x, "edge_index": _graph(n, 250, seed=12), "num_nodes": n}, label=y)
    perm = torch.randperm(n, generator=g)
    split_idx = {"train": perm[:30], "valid": perm[30:50], "test": perm[50:]}
    argv = next(a for a in _difformer_commands("node classification") if "pokec" in a)         # a main-batch.py line
    args = _args(parse, argv)
    model = parse.parse_method(args, n, c, d, torch.device("cpu"))
    criterion = nn.NLLLoss()
    eval_func = lambda y_true, out: float((out.argmax(dim=-1, keepdim=True) == y_true).float().mean())
    model.reset_parameters()
    optimizer = torch.optim.Adam(model.parameters(), lr=0.01)
    outs = []
    for epoch in range(2):
        model.to(torch.device("cpu"))                                   # main-batch.py:121 (`device` is the host here)
        model.train()
        optimizer.zero_grad()
        out = model(x, dataset.graph["edge_index"])                     # :135 -- a staged call with gradients
        loss = criterion(F.log_softmax(out, dim=1)[split_idx["train"]], y.squeeze(1)[split_idx["train"]])
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
        optimizer.step()
        result = ref_eval.evaluate_cpu(model, dataset, split_idx, eval_func, criterion, args, torch.device("cpu"))   # :144-145
        out = result[-1]
        assert out.device.type == "cpu" and out.shape == (n, c) and torch.isfinite(result[3])
        monkey_off = staging.FORCE_DEVICE
        staging.FORCE_DEVICE = None                                     # the same parameters, unstaged (test backend on the host)
        try:
            with torch.no_grad():
                plain = F.log_softmax(model(x, dataset.graph["edge_index"]), dim=1)
        finally:
            staging.FORCE_DEVICE = monkey_off
        assert torch.allclose(out, plain, atol=1e-6)
        outs.append(out)
    assert not torch.allclose(outs[0], outs[1])                          # the twin followed the optimiser step
    assert "_staged" in model.__dict__ and not any(k.startswith("_staged") for k in model.state_dict())


def test_reference_test_large_dataset_body_runs_staged(staged_on_host, tmp_path, monkeypatch):
    """The model-handling lines of `test_large_dataset.py` (:68-69 parse_method(...).to(cpu); :85-88 torch.load +
    load_state_dict; :90-93 eval + no_grad forward of host tensors) EXECUTED from the reference source, for both command
    lines of run_test_large.sh, around a stub dataset and a checkpoint saved from another instance."""
    from difformer_amd import staging
    parse, dropin = _load("node classification", "difformer.py", extra=("gnns",))
    src = open(os.path.join(REF, "node classification", "test_large_dataset.py")).read().splitlines()
    pick = lambda lo, hi: "\n".join(src[lo - 1: hi])
    body = "\n".join([pick(69, 69), pick(87, 88), pick(91, 93)])
    assert "parse_method(args, n, c, d, device).to(torch.device(\"cpu\"))" in body and "model.load_state_dict(checkpoint)" in body
    assert "out = model(dataset.graph['node_feat'], dataset.graph['edge_index'])" in body
    n, c, d = 64, 4, 11
    x, ei = torch.randn(n, d), _graph(n, 220, seed=3)
    dataset = types.SimpleNamespace(graph={"node_feat": x, "edge_index": ei, "num_nodes": n})
    for argv in _difformer_commands("node classification", "run_test_large.sh"):
        args = _args(parse, argv)
        trained = parse.parse_method(args, n, c, d, torch.device("cpu"))
        trained.reset_parameters()
        ckpt = tmp_path / "ckpt.pkl"
        torch.save(trained.state_dict(), ckpt)
        ns = dict(parse_method=parse.parse_method, args=args, n=n, c=c, d=d, device=torch.device("cpu"), torch=torch,
                  dataset=dataset, checkpoint_dir=str(ckpt))
        exec(compile(body, "test_large_dataset.py", "exec"), ns)
        out = ns["out"]
        assert out.device.type == "cpu" and out.shape == (n, c)
        assert "_staged" in ns["model"].__dict__                        # it did go through the staging path
        staging_dev = staging.FORCE_DEVICE
        staging.FORCE_DEVICE = None
        try:
            with torch.no_grad():
                plain = trained.eval()(x, ei)
        finally:
            staging.FORCE_DEVICE = staging_dev
        assert torch.allclose(out, plain, atol=1e-6)


def test_spatial_temporal_training_loop_executed_from_the_reference_source(fake_backend):
    """`spatial-temporal/main.py:81-123` -- reset_parameters, Adam, retain_grad on the parameters, the snapshot loop with
    `snapshot.to(device)`, `model(snapshot.x, snapshot.edge_index, snapshot.edge_attr)`, the summed cost with ONE
    `cost_tr.backward(retain_graph=True)` (chickenpox, covid) or a backward + step per snapshot (wikimath), then
    `evaluate(model, val_dataset, device, args)` from the folder's own eval.py -- EXECUTED from the reference files on the drop-in,
    for every `--method difformer` line of run.sh of the chickenpox and wikimath blocks (the two branches of the loop)."""
    import numpy as np
    parse, dropin = _load("spatial-temporal", "difformer.py", extra=("gnns",))
    src = open(os.path.join(REF, "spatial-temporal", "main.py")).read().splitlines()
    body = "\n".join(src[80:123])                                      # lines 81-123
    assert body.startswith("for run in range(args.runs):") and "cost_tr.backward(retain_graph=True)" in body
    assert "y_hat = model(snapshot.x, snapshot.edge_index, snapshot.edge_attr)" in body and body.rstrip().endswith(
        "cost_val = evaluate(model, val_dataset, device, args)")
    saved = {k: sys.modules.get(k) for k in ("torch_geometric", "torch_geometric.nn")}
    sys.modules["torch_geometric"] = _stub("torch_geometric", nn=_stub("torch_geometric.nn", knn_graph=None))
    sys.modules["torch_geometric.nn"] = sys.modules["torch_geometric"].nn
    try:
        spec = importlib.util.spec_from_file_location("ref_st_eval", os.path.join(REF, "spatial-temporal", "eval.py"))
        ref_eval = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref_eval)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v

    class Snapshot:                                                     # torch_geometric.data.Data as the loop uses it
        def __init__(self, x, edge_index, edge_attr, y):
            self.x, self.edge_index, self.edge_attr, self.y = x, edge_index, edge_attr, y

        def to(self, device):                                           # main.py:95: NEW tensors every time
            return Snapshot(*(t.clone().to(device) for t in (self.x, self.edge_index, self.edge_attr, self.y)))

    g = torch.Generator().manual_seed(3)
    n, c = 20, 1
    cmds = [a for a in _difformer_commands("spatial-temporal") if a[a.index("--dataset") + 1] in ("chickenpox", "wikimath")]
    assert len(cmds) == 8
    for argv in cmds:
        args = _args(parse, argv)
        d = 4 if args.dataset == "chickenpox" else 14
        ei = _graph(n, 60, seed=4)
        data = [Snapshot(torch.randn(n, d, generator=g), ei, torch.rand(ei.shape[1], generator=g) + 0.1, torch.randn(n, generator=g))
                for _ in range(7)]
        args.runs, args.epochs = 1, 2
        device = torch.device("cpu")
        torch.manual_seed(args.seed)
        model, _ = parse.parse_method(args, n, c, d, device)
        ns = dict(args=args, model=model, torch=torch, np=np, train_dataset=data[:5], val_dataset=data[5:], device=device,
                  evaluate=ref_eval.evaluate, knn_graph=None)
        before = [p.detach().clone() for p in model.parameters()]
        exec(compile(body, "spatial-temporal/main.py", "exec"), ns)
        assert np.isfinite(ns["cost_val"]) and np.isfinite(float(ns["cost_tr"]))
        assert any(not torch.equal(a, b) for a, b in zip(before, model.parameters()))      # the optimiser moved them
        assert all(p.grad is None or not p.grad.any() for p in model.parameters())         # zero_grad() after the last step

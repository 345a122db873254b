"""The reference's CPU-RESIDENT callers on the real backend.

`node classification/eval.py:34-43` (`evaluate_cpu`, called from `main-batch.py:144-145`) and
`node classification/test_large_dataset.py:69,85-93` keep the model and the graph in host memory.  The package stages such a
call onto the GPU (difformer_amd/staging.py): the HIP kernels run, the logits come back as a host tensor.

/root/reference does not exist on the GPU box, so the two call sequences are restated here line by line (each line cites
the reference line it stands for); `tests/test_reference_callers.py` executes the SAME lines from the reference's own source
in the build container, on the staging logic with the test backend.  Results are held to the float64 oracle at 1e-4.
"""
import copy
import io
import os
import pickle
import types

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import grad_err, rel_err
from oracle import difformer_oracle as orc
from oracle import difformer_oracle_grad as og

pytestmark = pytest.mark.gpu
TOL = 1e-4
CPU = torch.device("cpu")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return torch.device("cuda:0")


def _graph(n, e, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.cat([torch.randint(0, n, (2, e), generator=g), torch.arange(n).repeat(2, 1)], dim=1)


def _cfg(model, hidden):
    c0 = model.convs[0]
    return dict(hidden_channels=hidden, num_layers=len(model.convs), num_heads=c0.num_heads, kernel=c0.kernel, alpha=model.alpha,
                use_bn=model.use_bn, use_residual=model.residual, use_weight=c0.use_weight, use_graph=c0.use_graph,
                graph_weight=c0.graph_weight, use_source=c0.use_source)


def _oracle(model, x, ei, hidden, log_softmax=False):
    p = {k: v.detach().cpu().double().numpy() for k, v in model.state_dict().items()}
    ref = orc.difformer_forward(p, x.double().numpy(), None if ei is None else ei.numpy(), None, _cfg(model, hidden))
    if log_softmax:
        ref = ref - ref.max(axis=1, keepdims=True)
        ref = ref - np.log(np.exp(ref).sum(axis=1, keepdims=True))
    return ref


def evaluate_cpu(model, dataset, split_idx, eval_func, criterion, args, device, result=None):
    """eval.py:33-60, line for line (the branch taken by every dataset but the multi-label ones)."""
    with torch.no_grad():                                                        # :33 @torch.no_grad()
        model.eval()                                                             # :38
        model.to(torch.device("cpu"))                                            # :40
        dataset.label = dataset.label.to(torch.device("cpu"))                    # :41
        edge_index, x = dataset.graph['edge_index'], dataset.graph['node_feat']  # :42
        out = model(x, edge_index)                                               # :43
        train_acc = eval_func(dataset.label[split_idx['train']], out[split_idx['train']])      # :45-46
        valid_acc = eval_func(dataset.label[split_idx['valid']], out[split_idx['valid']])      # :47-48
        test_acc = eval_func(dataset.label[split_idx['test']], out[split_idx['test']])         # :49-50
        out = F.log_softmax(out, dim=1)                                          # :59
        valid_loss = criterion(out[split_idx['valid']], dataset.label.squeeze(1)[split_idx['valid']])   # :60-61
    return train_acc, valid_acc, test_acc, valid_loss, out


def eval_acc(y_true, y_pred):
    return float((y_pred.argmax(dim=-1, keepdim=True) == y_true).float().mean())


@pytest.mark.parametrize("n,e,hidden,layers", [(3000, 9000, 64, 3),          # Pokec-like: sparse graph, gather kernels
                                               (9000, 460000, 64, 2)])        # dense graph: closed form + sliced product
def test_evaluate_cpu_after_gpu_training_epochs(n, e, hidden, layers, dev):
    """main-batch.py:119-145: an epoch of optimiser steps on the GPU (`model.to(device)`, `model.train()`, forward, backward,
    step), then `evaluate_cpu(model, dataset, ...)` with the model moved to the host and the full graph in host memory --
    twice, with `model.to(device)` in between, as the epoch loop does."""
    from difformer_amd import DIFFormer, ops
    d, c = 24, 5
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, d, generator=g)
    y = torch.randint(0, c, (n, 1), generator=g)
    ei = _graph(n, e, seed=4)
    dataset = types.SimpleNamespace(graph={"node_feat": x, "edge_index": ei, "num_nodes": n}, label=y)
    perm = torch.randperm(n, generator=g)
    split_idx = {"train": perm[: n // 2], "valid": perm[n // 2: 3 * n // 4], "test": perm[3 * n // 4:]}
    torch.manual_seed(0)
    model = DIFFormer(d, hidden, c, num_layers=layers, num_heads=1, kernel="simple", use_bn=True, use_residual=True,
                      use_weight=True, use_graph=True, dropout=0.0)
    criterion = nn.NLLLoss()
    optimizer = torch.optim.Adam(model.parameters(), lr=0.01)
    be = ops.get_backend()
    eig = ei.to(dev)
    outs = []
    for epoch in range(2):
        model.to(dev)                                                            # main-batch.py:121
        model.train()                                                            # :122
        optimizer.zero_grad()                                                    # :133
        out_i = F.log_softmax(model(x.to(dev), eig), dim=1)                      # :135, :139
        loss = criterion(out_i[split_idx['train'].to(dev)], y.squeeze(1).to(dev)[split_idx['train'].to(dev)])
        loss.backward()                                                          # :141
        optimizer.step()                                                         # :142
        be.kernel_events = {}
        result = evaluate_cpu(model, dataset, split_idx, eval_acc, criterion, None, dev)     # :144-145
        launched, be.kernel_events = set(be.kernel_events), None
        out = result[-1]
        assert out.device.type == "cpu" and out.shape == (n, c)
        assert all(p.device.type == "cpu" for p in model.parameters())          # the caller's model stays where it put it
        assert "dif_simple_layer_f32" in launched, launched
        if epoch == 1:
            assert "dif_csr_build" not in launched, "the second evaluation must find the cached CSR"
        assert rel_err(out.numpy(), _oracle(model, x, ei, hidden, log_softmax=True)) < TOL
        assert 0.0 <= result[0] <= 1.0 and torch.isfinite(result[3])
        outs.append(out)
    assert not torch.equal(outs[0], outs[1])                                     # the second evaluation saw the optimiser step


@pytest.mark.parametrize("kernel,use_graph", [("simple", True), ("sigmoid", True), ("simple", False)])
def test_test_large_dataset_body(kernel, use_graph, dev, tmp_path):
    """test_large_dataset.py:69 (`parse_method(...).to(torch.device("cpu"))`), :85-88 (torch.load + strict load_state_dict),
    :90-93 (`model.eval()`, `with torch.no_grad(): out = model(node_feat, edge_index)`), :94 (the metric on host tensors)."""
    from difformer_amd import DIFFormer
    n, d, c, hidden = 2500, 40, 6, 64
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(n, d, generator=g), torch.randint(0, c, (n, 1), generator=g)
    ei = _graph(n, 12000, seed=8) if use_graph else None
    dataset = types.SimpleNamespace(graph={"node_feat": x, "edge_index": ei, "num_nodes": n}, label=y)
    kw = dict(num_layers=2, num_heads=1, kernel=kernel, use_bn=True, use_residual=True, use_weight=True, use_graph=use_graph)
    torch.manual_seed(1)
    trained = DIFFormer(d, hidden, c, **kw).to(dev)                              # the run that saved the checkpoint
    checkpoint_dir = str(tmp_path / "pokec-difformer.pkl")
    torch.save(trained.state_dict(), checkpoint_dir)                             # main-batch.py:154
    model = DIFFormer(d, hidden, c, **kw).to(dev).to(torch.device("cpu"))        # parse.py:6-7 `.to(device)`; test_large_dataset.py:69
    checkpoint = torch.load(checkpoint_dir)                                      # :87
    model.load_state_dict(checkpoint)                                            # :88 (strict; CUDA tensors into host parameters)
    model.eval()                                                                 # :91
    with torch.no_grad():                                                        # :92
        out = model(dataset.graph['node_feat'], dataset.graph['edge_index'])     # :93
        test_acc = eval_acc(dataset.label, out)                                  # :94
    assert out.device.type == "cpu" and 0.0 <= test_acc <= 1.0
    assert rel_err(out.numpy(), _oracle(trained, x, ei, hidden)) < TOL


def test_staged_training_step_matches_the_oracle_gradients(dev):
    """`main.py --cpu` keeps the whole training on the host device: forward, `loss.backward()` (main.py:130) and the
    optimiser see host tensors; the arithmetic of both passes runs on the GPU and the gradients come back."""
    from difformer_amd import DIFFormer
    n, d, c, hidden = 1200, 16, 4, 32
    g = torch.Generator().manual_seed(21)
    x, y = torch.randn(n, d, generator=g), torch.randint(0, c, (n,), generator=g)
    ei = _graph(n, 5000, seed=22)
    idx = torch.randperm(n, generator=g)[: n // 2]
    torch.manual_seed(2)
    model = DIFFormer(d, hidden, c, num_layers=2, kernel="simple", dropout=0.0)
    model.train()
    out = model(x, ei)
    assert out.device.type == "cpu" and out.requires_grad
    loss = F.nll_loss(F.log_softmax(out, dim=1)[idx], y[idx])
    loss.backward()
    pl = og.leaves({k: v.detach().numpy() for k, v in model.state_dict().items()})
    lref = og.training_loss(og.difformer_forward(pl, x.double(), ei, None, _cfg(model, hidden)), y, idx)
    lref.backward()
    gmax = max(float(v.grad.abs().max()) for v in pl.values())
    assert abs(float(loss) - float(lref)) < 1e-4 * abs(float(lref))
    for k, prm in model.named_parameters():
        assert prm.grad is not None and prm.grad.device.type == "cpu"
        assert grad_err(prm.grad.numpy(), pl[k].grad.numpy(), gmax) < TOL, k


def test_host_operands_of_the_module_functions(dev):
    """`full_attention_conv` / `gcn_conv` with host tensors (the reference's functions accept whatever device they get)."""
    from difformer_amd import full_attention_conv, gcn_conv
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(700, 2, 16, generator=g) for _ in range(3))
    for kernel in ("simple", "sigmoid"):
        out = full_attention_conv(q, k, v, kernel)
        assert out.device.type == "cpu"
        assert rel_err(out.numpy(), orc.full_attention_conv(q.double().numpy(), k.double().numpy(), v.double().numpy(), kernel)) < TOL
    ei = _graph(700, 3000, seed=6)
    out = gcn_conv(v, ei, None)
    assert out.device.type == "cpu" and rel_err(out.numpy(), orc.gcn_conv(v.double().numpy(), ei.numpy(), None)) < TOL
    qg = q.clone().requires_grad_(True)
    full_attention_conv(qg, k, v, "simple").sum().backward()
    assert qg.grad is not None and qg.grad.device.type == "cpu" and torch.isfinite(qg.grad).all()


def test_mixed_placement_still_raises(dev):
    from difformer_amd import DIFFormer
    model = DIFFormer(8, 16, 3).to(dev).eval()
    with pytest.raises(RuntimeError), torch.no_grad():
        model(torch.randn(50, 8), _graph(50, 100).to(dev))                       # host x, device model: as in the reference


def test_deepcopy_and_pickle_after_eval_calls(dev):
    """`copy.deepcopy(model)` (best-checkpoint / EMA copies) and `torch.save(model)` after the auto-captured
    hipGraph exists, and after a staged call left a device twin behind."""
    from difformer_amd import DIFFormer
    n = 3000
    x, ei = torch.randn(n, 12).to(dev), _graph(n, 9000).to(dev)
    model = DIFFormer(12, 64, 4).to(dev).eval()
    with torch.no_grad():
        outs = [model(x, ei) for _ in range(4)]                                  # the third call captures, the fourth replays
    assert model._ag_state is not None and model._ag_state[2] is not None
    clone = copy.deepcopy(model)
    buf = io.BytesIO()
    torch.save(model, buf)
    buf.seek(0)
    loaded = torch.load(buf, weights_only=False)
    with torch.no_grad():
        assert torch.equal(clone(x, ei), outs[-1]) and torch.equal(loaded(x, ei), outs[-1])
    host = copy.deepcopy(model).to(CPU)
    with torch.no_grad():
        staged = host(x.cpu(), ei.cpu())
    assert "_staged" in host.__dict__ and rel_err(staged.numpy(), outs[-1].cpu().numpy()) < 1e-5
    again = pickle.loads(pickle.dumps(host))
    assert "_staged" not in again.__dict__ and set(again.state_dict()) == set(model.state_dict())

"""CPU-resident callers of the reference, run on the MI355X without editing them.

Two of the reference's eight call sites keep the model and the graph in host memory: `test_large_dataset.py:69,91-93`
(`parse_method(...).to(torch.device("cpu"))`, then `model(dataset.graph['node_feat'], dataset.graph['edge_index'])`; both
lines of `run_test_large.sh` also pass `--cpu`) and `eval.py:34-43` (`evaluate_cpu`, called from `main-batch.py:144-145`:
`model.to(torch.device("cpu"))`, `out = model(x, edge_index)` on the full graph after every epoch of mini-batch training on
the GPU).  This package has no host arithmetic, so such a call is STAGED: the operands go to the GPU, the HIP path runs
there, the logits come back on the caller's device.

    parameters  ->  a device twin of the module (same class, same flags), its parameters refreshed by `copy_` whenever the
                    host parameters' (data_ptr, _version) key changes -- so the twin's own caches (concatenated projections,
                    weight-only factors, hipGraph capture), keyed on ITS parameters' versions, stay valid;
    x, edge_index, edge_weight  ->  device copies kept per host tensor (identity + version, weak references), so the CSR /
                    sliced-format cache -- keyed on the device `edge_index` -- hits on the next evaluation;
    gradients   ->  under autograd (`main.py --cpu`: the whole training on the host device) the twin's forward is recorded on
                    the device and `_StagedCall.backward` hands the parameter / input gradients back as host tensors.

Staging only applies when EVERYTHING is on the host (parameters and every tensor argument), a GPU is present and the
backend is the HIP one; mixed placements raise as they do in the reference (`F.linear` device mismatch), and without a GPU
the usual "no CPU fallback" error stands.  Writes through `.data` do not bump `_version`: call `model.invalidate_caches()`
after them, as for the other caches.
"""
from __future__ import annotations

import copy
import weakref
from collections import OrderedDict

import torch

from . import ops

# Tests may point this at torch.device("cpu") to run the staging logic (twin refresh, operand cache, gradient routing)
# on the host-side test backend; the product leaves it None (= the current HIP device).
FORCE_DEVICE = None


def staging_device(module, tensors):
    """Device to stage this call on, or None when the call is not a CPU-resident one (or cannot be staged)."""
    ts = [t for t in tensors if torch.is_tensor(t)]
    if not ts or any(t.device.type != "cpu" for t in ts):
        return None
    if module is not None and module.__dict__.get("_is_twin"):
        return None                                   # (only reachable with FORCE_DEVICE = the host)
    if module is not None and any(p.device.type != "cpu" for p in module.parameters()):
        return None
    if FORCE_DEVICE is not None:
        return FORCE_DEVICE
    if not torch.cuda.is_available():
        return None                                   # the HIP path then raises its "no CPU fallback" error
    be = ops._BACKEND
    if be is not None and not hasattr(be, "lib"):     # a host-side test backend: nothing to stage for
        return None
    return torch.device("cuda", torch.cuda.current_device())


class _OperandCache:
    """Device copies of host operands (x, edge_index, edge_weight), one per host tensor: keyed on identity + data_ptr +
    shape + version and checked against a weak reference, like the CSR cache.  The device edge_index has to be the SAME
    tensor from call to call or every evaluation would rebuild the CSR and the sliced format."""

    def __init__(self, capacity=6):
        self.capacity, self.entries = capacity, OrderedDict()

    def get(self, t, dev):
        if t is None or not torch.is_tensor(t):
            return t
        ver = ops.tensor_version(t)
        key = (id(t), t.data_ptr(), tuple(t.shape), t.dtype, ver, str(dev))
        # on EVERY call: the device copy of a freed host tensor (a full-graph x can be hundreds of MB) goes at once, not at
        # the next miss
        for k in [k for k, v in self.entries.items() if v[0]() is None]:
            del self.entries[k]
        hit = self.entries.get(key)
        if ver >= 0 and hit is not None and hit[0]() is t:
            self.entries.move_to_end(key)
            return hit[1]
        d = t.detach().to(dev)
        if ver >= 0:
            self.entries[key] = (weakref.ref(t), d)
            while len(self.entries) > self.capacity:
                self.entries.popitem(last=False)
        return d

    def clear(self):
        self.entries.clear()


operands = _OperandCache()

_FLAGS = ("training", "dropout", "use_bn", "residual", "alpha", "auto_graph")
_CONV_FLAGS = ("training", "kernel", "use_graph", "use_weight", "graph_weight", "use_source", "num_heads", "out_channels")


class _Twin:
    """The device copy of a host-resident module."""

    def __init__(self, host, dev):
        self.module = copy.deepcopy(host).to(dev)      # __getstate__ of the module classes drops every cache first
        self.module.__dict__["_is_twin"] = True
        self.dev = dev
        self.key = None

    def refresh(self, host):
        hp, tp = list(host.parameters()), list(self.module.parameters())
        key = ops.param_key(hp)
        if key is None or key != self.key:
            with torch.no_grad():
                for t, h in zip(tp, hp):
                    t.copy_(h)                          # in place: bumps the twin's versions, so its caches refresh
            self.key = key
        for t, h in zip(tp, hp):                        # a host parameter frozen / unfrozen after the twin was made
            if t.requires_grad != h.requires_grad:
                t.requires_grad_(h.requires_grad)
        for (_, tb), (_, hb) in zip(self.module.named_buffers(), host.named_buffers()):
            if tb is not None and hb is not None:
                with torch.no_grad():
                    tb.copy_(hb)
        for name in _FLAGS:
            if hasattr(host, name):
                setattr(self.module, name, getattr(host, name))
        self.module.train(host.training)
        for hc, tc in zip(getattr(host, "convs", ()), getattr(self.module, "convs", ())):
            for name in _CONV_FLAGS:
                if hasattr(hc, name) and getattr(tc, name, None) != getattr(hc, name):
                    setattr(tc, name, getattr(hc, name))
        return tp


def twin_of(host, dev):
    st = host.__dict__.get("_staged")
    if st is None or st[0].dev != dev or len(list(st[0].module.parameters())) != len(list(host.parameters())):
        st = (_Twin(host, dev),)                       # a tuple: nn.Module.__setattr__ must not register the twin
        host.__dict__["_staged"] = st
    return st[0]


class _StagedCall(torch.autograd.Function):
    """forward: the twin's forward, recorded on the device; backward: its gradients, returned as host tensors."""

    @staticmethod
    def forward(ctx, twin, fwd, x, n_extra, *rest):
        extra, host_params = rest[:n_extra], rest[n_extra:]
        dev = twin.dev
        tparams = twin.module.parameters()
        tparams = list(tparams)
        with torch.enable_grad():
            xg = x.detach().to(dev).requires_grad_(x.requires_grad)
            out = fwd(twin.module, xg, *extra)
        ctx.twin_params, ctx.xg, ctx.out = tparams, xg, out
        ctx.x_device = x.device
        ctx.need = [p.requires_grad for p in host_params]
        return out.detach().to(x.device)

    @staticmethod
    def backward(ctx, g):
        wanted = [p for p, need in zip(ctx.twin_params, ctx.need) if need]
        inputs = ([ctx.xg] if ctx.xg.requires_grad else []) + wanted
        grads = list(torch.autograd.grad(ctx.out, inputs, g.to(ctx.out.device), allow_unused=True)) if inputs else []
        gx = None
        if ctx.xg.requires_grad:
            gx = grads.pop(0)
            gx = None if gx is None else gx.to(ctx.x_device)
        it = iter(grads)
        gp = []
        for need in ctx.need:
            gr = next(it) if need else None
            gp.append(None if gr is None else gr.to(ctx.x_device))
        ctx.out = ctx.xg = None
        return (None, None, gx, None) + (None,) * (len(ctx.needs_input_grad) - 4 - len(gp)) + tuple(gp)


def staged_forward(host, dev, fwd, x, *extra):
    """Run `fwd(twin, x_dev, *extra_dev)` for the host-resident module `host`; -> the result on x's device.
    `extra`: further operands (edge_index, edge_weight, n_nodes ...); host tensors among them are staged (and cached)."""
    twin = twin_of(host, dev)
    twin.refresh(host)
    staged = tuple(operands.get(t, dev) if (torch.is_tensor(t) and t.device.type == "cpu" and t.dim() > 0) else t
                   for t in extra)
    host_params = list(host.parameters())
    if torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in extra):
        raise NotImplementedError("difformer_amd: gradients with respect to edge_weight are not routed through a staged "
                                  "(host-resident) call; move the model and its operands to the GPU")
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in host_params)):
        return _StagedCall.apply(twin, fwd, x, len(staged), *staged, *host_params)
    with torch.no_grad():
        out = fwd(twin.module, operands.get(x, dev), *staged)
    return out.to(x.device)


def drop(host):
    """Forget the twin of `host` (invalidate_caches: parameter writes through .data, flag surgery on sub-modules ...)."""
    host.__dict__.pop("_staged", None)

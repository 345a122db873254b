"""Whole-model path for TINY graphs (csrc/tiny_model.hip; include/difformer_hip.h `dif_tiny_*`).

`spatial-temporal/` trains `DIFFormer(d, 4, 1, num_layers=2, num_heads=1, use_weight=False)` on 20 / 129 / 1,068-node
snapshots -- hundreds of forwards per epoch, each on tensors fresh from `snapshot.to(device)`, their costs summed before ONE
`cost_tr.backward(retain_graph=True)` (spatial-temporal/run.sh:5-40, main.py:94-121).  Layer by layer that is ~110 kernel
launches and ~1.1 ms of host time per training snapshot for 0.33 ms of kernels (profiles/r05_experiments.md); here a forward
is ONE launch (+ one for a graph not seen before, + one `torch.rand` when dropout is on), its backward one more, under one
autograd node that hands every parameter its gradient.

Taken for: one head, float32, n <= 4,096 (`sigmoid` above 64 nodes: L + 1 launches forward and L + 2 backward, a layer's
O(n^2) pair loop spread over the chip -- csrc/tiny_sigmoid_grid.hip --, `simple` above 256 nodes likewise --
csrc/tiny_simple_grid.hip --, behind the same two C calls), hidden <= 8,
<= 64 input features, <= 8 outputs, <= 8 layers, every flag of the constructor, graphs of <= 65,535 entries (prepared by
`dif_tiny_graph_build`, one workgroup per direction).  Everything else -- `--special_treat dense` at n = 1,068 (1.14 M
entries), `edge_weight` tensors that want a gradient -- takes the layer-by-layer path.  DIFFORMER_TINY=0 switches it off.
"""
from __future__ import annotations

import ctypes
import math
import os
import weakref
from collections import OrderedDict

import torch

from . import _lib, ops
from .backend_hip import _stream

ENABLED = os.environ.get("DIFFORMER_TINY", "1") != "0"
MAX_NODES, MAX_HIDDEN, MAX_IN, MAX_OUT, MAX_LAYERS, MAX_EDGES = 4096, 8, 64, 8, 8, 65535
MAX_NODES_SIGMOID = int(os.environ.get("DIFFORMER_TINY_SIGMOID_NODES", "4096"))
PACK = os.environ.get("DIFFORMER_TINY_PACK", "1") != "0"     # the parameters as one flat autograd input when forwards repeat (_Pack)
PLAN = int(os.environ.get("DIFFORMER_TINY_PLAN", "0"))     # dif_tiny_cfg.launch_plan: 0 by size, 1 one workgroup, 2 one launch per layer stage over the chip


class TinyGraph:
    """Destination-major CSR and its transpose for the tiny kernels: raw device addresses into ONE buffer (int32 pointers /
    neighbours, float32 values) -- the kernels take addresses, and seven tensor views per snapshot cost more than the launch."""

    __slots__ = ("rowptr", "src", "val", "rowptr_t", "dst_t", "val_t", "nnz", "n", "scale", "keep", "__weakref__")

    def tensors(self):
        """(rowptr, src, val, rowptr_t, dst_t, val_t) as tensor views (tests)."""
        buf, n, E = self.keep[0], self.n, self.nnz
        o = [0, n + 1, 2 * (n + 1), 2 * (n + 1) + E, 2 * (n + 1) + 2 * E, 2 * (n + 1) + 3 * E, 2 * (n + 1) + 4 * E]
        rp, rpt, src, dst, val, valt = (buf[o[k]: o[k + 1]] for k in range(6))
        return rp, src, val.view(torch.float32), rpt, dst, valt.view(torch.float32)


class _GraphCache:
    """TinyGraph per (edge_index, edge_weight) tensor pair, keyed on identity + version like ops.csr_cache.  A training loop
    over snapshots makes new tensors per snapshot (main.py:96): every miss is one launch."""

    def __init__(self, capacity=8):
        self.capacity, self.entries = capacity, OrderedDict()

    def get(self, edge_index, edge_weight, n):
        # (tensors made under torch.inference_mode() track no version: ops.tensor_version gives -1 and the graph is rebuilt
        # on every call, as ops.csr_cache does)
        vi = ops.tensor_version(edge_index)
        vw = 0 if edge_weight is None else ops.tensor_version(edge_weight)
        if vi < 0 or vw < 0:
            return build_graph(edge_index, edge_weight, n)
        key = (id(edge_index), edge_index.data_ptr(), edge_index.shape[1], vi, n,
               None if edge_weight is None else (id(edge_weight), edge_weight.data_ptr(), vw))
        hit = self.entries.get(key)
        if hit is not None and hit[0]() is edge_index and (edge_weight is None or hit[1]() is edge_weight):
            self.entries.move_to_end(key)
            return hit[2]
        g = build_graph(edge_index, edge_weight, n)
        self.entries[key] = (weakref.ref(edge_index), None if edge_weight is None else weakref.ref(edge_weight), g)
        while len(self.entries) > self.capacity:
            self.entries.popitem(last=False)
        return g

    def clear(self):
        self.entries.clear()


graphs = _GraphCache()
stats = {"forward": 0, "backward": 0, "graph_builds": 0}          # calls that took this path (tests assert on them)

# The index check of dif_tiny_graph_build (status[0] != 0: a node id outside [0, n); the entry is filed under node 0) is read
# WITHOUT a host synchronisation per snapshot: every build writes its two status words into the next slot of a ring on its
# device, and ONE read checks all slots
#   * when the ring is full (128 builds),
#   * at every model.train() / model.eval() switch (DIFFormer.train: once per epoch in the reference's loops,
#     spatial-temporal/main.py:91, eval.py:9) and at interpreter exit,
#   * right after the build when DIFFORMER_DEBUG=1 (one synchronisation per new graph: the error then comes from the offending
#     call, as the reference's own index ops raise, difformer.py:66).
# So a bad edge_index raises ValueError at the latest at the next mode switch -- DEFERRED, unlike the layer-by-layer path.
DEBUG = os.environ.get("DIFFORMER_DEBUG", "0") == "1"
_RING = 128
_rings = {}            # device -> [int32 tensor [2 * _RING], next slot]


def _status_slot(dev):
    ring = _rings.get(dev)
    if ring is None:
        ring = _rings[dev] = [torch.zeros(2 * _RING, dtype=torch.int32, device=dev), 0]
    if ring[1] >= _RING:
        _poll_status(wait=True)
    slot = ring[1]
    ring[1] += 1
    return ring[0].data_ptr() + 8 * slot


def _poll_status(wait=False):
    """Raise for a bad edge_index seen by any build since the last check (wait=False: nothing -- the ring reads itself when
    full).  Synchronises with the device when any build is pending."""
    if not wait:
        return
    for ring in _rings.values():
        used, ring[1] = ring[1], 0
        if used and any(ring[0][: 2 * used: 2].tolist()):
            ring[0].zero_()
            raise ValueError("difformer_amd: edge_index holds node ids outside [0, num_nodes)")


def build_graph(edge_index, edge_weight, n):
    """-> TinyGraph by dif_tiny_graph_build (<= 65,535 entries; one launch, no host synchronisation; the index check is read
    later, `_poll_status`)."""
    lib = _lib.load()
    dev = edge_index.device
    E = int(edge_index.shape[1])
    if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise TypeError("difformer_amd: edge_index must be an int64 tensor of shape [2, E]")
    ei = edge_index if edge_index.is_contiguous() else edge_index.contiguous()
    w = None
    if edge_weight is not None:
        w = edge_weight.detach()
        if w.dtype != torch.float32 or not w.is_contiguous():
            w = w.to(torch.float32).contiguous()
        if w.numel() != E:
            raise ValueError("difformer_amd: edge_weight must have one entry per edge")
    # one allocation: [rowptr N+1][rowptr_t N+1][src E][dst_t E][val E][val_t E][pad][workspace 4 E + 4]
    ints = 2 * (n + 1) + 4 * E
    ints += (-ints) % 4
    ws_bytes = 16 * E + 16
    buf = torch.empty(ints + 4 * E + 4, dtype=torch.int32, device=dev)
    base = buf.data_ptr()
    g = TinyGraph()
    g.nnz, g.n, g.scale = E, n, 1.0
    g.rowptr = base
    g.rowptr_t = base + 4 * (n + 1)
    g.src = base + 8 * (n + 1)
    g.dst_t = g.src + 4 * E
    g.val = g.dst_t + 4 * E
    g.val_t = g.val + 4 * E
    rc = lib.dif_tiny_graph_build(ei.data_ptr(), None if w is None else w.data_ptr(), E, n, g.rowptr, g.src, g.val, g.rowptr_t, g.dst_t,
                                  g.val_t, _status_slot(dev), base + 4 * ints, ws_bytes, _stream(dev))
    _lib.check(rc, "dif_tiny_graph_build")
    stats["graph_builds"] += 1
    g.keep = (buf, ei, w)
    if DEBUG:
        _poll_status(wait=True)
    return g


def _poll_at_exit():
    try:
        _poll_status(wait=True)
    except ValueError as e:           # nothing to unwind to any more: say it
        import sys
        print(f"difformer_amd: {e} (seen by an earlier forward of this process)", file=sys.stderr)
    except Exception:
        pass


import atexit  # noqa: E402
atexit.register(_poll_at_exit)


def _plan(model, x, edge_index, edge_weight):
    """The static side of a call -> (cfg fields, parameter list in the C ABI's order) or None when the call is not covered.
    Runs on every forward of a launch-bound loop: sub-modules and parameters are read through the registration dicts (plain
    dict reads; nn.Module.__getattr__ costs ~1 us each and a plan needs ~40 of them)."""
    if not ENABLED or not x.is_cuda or x.dim() != 2 or x.dtype != torch.float32:
        return None
    n, f_in = x.shape
    mods = model._modules
    convs = list(mods["convs"]._modules.values())
    L = len(convs)
    if not (1 <= n <= MAX_NODES and 1 <= f_in <= MAX_IN and 1 <= L <= MAX_LAYERS):
        return None
    fcs = list(mods["fcs"]._modules.values())
    bns = list(mods["bns"]._modules.values())
    w0, b0 = fcs[0]._parameters["weight"], fcs[0]._parameters["bias"]
    w1, b1 = fcs[-1]._parameters["weight"], fcs[-1]._parameters["bias"]
    d, c = w0.shape[0], w1.shape[0]
    if d > MAX_HIDDEN or c > MAX_OUT or w0.shape[1] != f_in or w0.dtype != torch.float32 or b0 is None or b1 is None:
        return None
    c0 = convs[0].__dict__
    kernel, use_graph, use_weight, graph_weight, use_source = (c0["kernel"], c0["use_graph"], c0["use_weight"], c0["graph_weight"],
                                                               c0["use_source"])
    if kernel not in ("simple", "sigmoid") or c0["num_heads"] != 1 or c0["out_channels"] != d:
        return None
    if kernel == "sigmoid" and n > MAX_NODES_SIGMOID:
        return None
    md = model.__dict__
    if md["training"] and not (0.0 <= float(md["dropout"]) < 1.0):
        return None                          # dropout = 1 (all zeros in training): the layer path's F.dropout semantics
    use_bn = bool(md["use_bn"])
    params = [w0, b0]
    params += [bns[0]._parameters["weight"], bns[0]._parameters["bias"]] if use_bn else [None, None]
    params += [w1, b1]
    for l, cv in enumerate(convs):
        cd = cv.__dict__
        if (cd["row_shard"] is not None or cd["kernel"] != kernel or cd["num_heads"] != 1 or cd["use_graph"] != use_graph or
                cd["use_weight"] != use_weight or cd["graph_weight"] != graph_weight or cd["use_source"] != use_source or
                cd["out_channels"] != d):
            return None
        sub = cv._modules
        params += [sub["Wk"]._parameters["weight"], sub["Wk"]._parameters["bias"], sub["Wq"]._parameters["weight"],
                   sub["Wq"]._parameters["bias"]]
        params += [sub["Wv"]._parameters["weight"], sub["Wv"]._parameters["bias"]] if use_weight else [None, None]
        params += [bns[l + 1]._parameters["weight"], bns[l + 1]._parameters["bias"]] if use_bn else [None, None]
    if use_graph:
        if edge_index is None:
            raise ValueError("use_graph=True needs an edge_index")
        if edge_index.shape[1] > MAX_EDGES:
            return None                      # (`--special_treat dense` at n = 1,068: 1.14 M entries -- the parallel kernels)
        if edge_weight is not None and edge_weight.requires_grad and torch.is_grad_enabled():
            return None                      # difformer.py:73 is differentiable in edge_weight: layer-by-layer path
    a_s, g_s = (1.0 - graph_weight, float(graph_weight)) if (use_graph and graph_weight > 0) else (1.0, 1.0)
    fields = dict(n=n, in_channels=f_in, hidden=d, out_channels=c, num_layers=L, kernel=1 if kernel == "sigmoid" else 0,
                  use_bn=int(use_bn), use_residual=int(bool(md["residual"])), use_weight=int(bool(use_weight)),
                  use_graph=int(bool(use_graph)), use_source=int(bool(use_source)), alpha=float(md["alpha"]),
                  attn_scale=a_s, gcn_scale=g_s, eps=float(bns[0].eps))
    return fields, params


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def _layout(live):
    """Offsets (floats, 16-byte aligned) of the parameters in ONE flat buffer -> (offsets, total)."""
    offs, o = [], 0
    for p in live:
        offs.append(o)
        o += p.numel() + (-p.numel()) % 4
    return offs, o


class _Pack(torch.autograd.Function):
    """The model's parameters as ONE flat tensor.  The spatial-temporal loop sums the costs of a hundred snapshots before ONE
    backward (main.py:105-121); with the parameters as separate inputs of every snapshot's node the autograd engine adds a
    hundred gradients per PARAMETER -- 16 small launches per snapshot, 80 of the 173 us a snapshot's backward took.  Through
    this node every snapshot hands back one flat gradient and the split into parameters runs once per backward."""

    @staticmethod
    def forward(ctx, slot, *live):
        ctx.slot = slot
        offs, total = _layout(live)
        flat = torch.zeros(total, dtype=torch.float32, device=live[0].device)
        torch._foreach_copy_([flat[o: o + p.numel()].view(p.shape) for o, p in zip(offs, live)], list(live))
        ctx.meta = (offs, [p.shape for p in live])
        return flat

    @staticmethod
    def backward(ctx, g):
        offs, shapes = ctx.meta
        ctx.slot[1] = None                                # a graph that has been walked may have been freed: the next forward packs anew
        return (None,) + tuple(g[o: o + math.prod(sh)].view(sh) if need else None
                               for o, sh, need in zip(offs, shapes, ctx.needs_input_grad[1:]))


_packs = weakref.WeakKeyDictionary()        # model -> [key of the parameter versions, flat copy or None]  (not ON the model: deepcopy)


def _packed(model, live):
    """The flat copy of the parameters for this forward, or None.  Built on the SECOND training forward that sees the same
    parameter versions (a loop that steps the optimiser after every snapshot, wikimath, never pays for it), reused until a
    parameter changes or a backward has run through it."""
    key = tuple((id(p), p._version, p.requires_grad) for p in live)
    slot = _packs.get(model)
    if slot is not None and slot[0] == key:
        if slot[1] is None:
            slot[1] = _Pack.apply(slot, *live)
        return slot[1]
    _packs[model] = [key, None]
    return None


class _TinyModel(torch.autograd.Function):
    """DIFFormer.forward as one launch; backward = one launch handing x and every parameter its gradient.  The parameters come
    as separate tensors (`flat_layout` None) or as ONE flat tensor (_Pack) with its layout."""

    @staticmethod
    def forward(ctx, state, x, *live_params):
        fields, params, graph, p_drop, training, flat_layout = state
        lib = _lib.load()
        dev = x.device
        n, d, L, c = fields["n"], fields["hidden"], fields["num_layers"], fields["out_channels"]
        xc = x if (x.stride(1) == 1 and (n == 1 or x.stride(0) >= x.shape[1])) else x.contiguous()
        ldx = xc.stride(0) if n > 1 else xc.shape[1]
        rnd = torch.rand((L + 1, n, d), device=dev) if (training and p_drop > 0.0) else None
        cfg = _lib.TinyCfg(training=int(training), dropout=float(p_drop), nnz=0 if graph is None else graph.nnz,
                           launch_plan=PLAN, **fields)
        tape = torch.empty(int(lib.dif_tiny_tape_floats(n, d, L)), dtype=torch.float32, device=dev)
        y = torch.empty((n, c), dtype=torch.float32, device=dev)
        if flat_layout is None:
            pa = _ptr_array(params)
        else:                                             # addresses inside the flat copy, in the C ABI's slot order
            base, it = live_params[0].data_ptr(), iter(flat_layout)
            pa = (ctypes.c_void_p * len(params))(*[None if p is None else base + 4 * next(it) for p in params])
        rc = lib.dif_tiny_forward_f32(ctypes.byref(cfg), xc.data_ptr(), ldx, pa,
                                      None if graph is None else graph.rowptr, None if graph is None else graph.src,
                                      None if graph is None else graph.val,
                                      None if rnd is None else rnd.data_ptr(), tape.data_ptr(), y.data_ptr(), _stream(dev))
        _lib.check(rc, "dif_tiny_forward_f32")
        stats["forward"] += 1
        ctx.cfg, ctx.graph, ctx.rnd, ctx.tape, ctx.ldx = cfg, graph, rnd, tape, ldx
        ctx.slots = [p is not None for p in params]
        ctx.flat_layout, ctx.pa = flat_layout, pa
        ctx.sizes = [p.numel() for p in params if p is not None]
        ctx.shapes = [p.shape for p in params if p is not None]
        ctx.save_for_backward(xc, *live_params)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        xc, *live = ctx.saved_tensors
        dev = xc.device
        cfg, graph = ctx.cfg, ctx.graph
        n, d = cfg.n, cfg.hidden
        sizes, shapes = ctx.sizes, ctx.shapes
        want_dx = ctx.needs_input_grad[1]
        offs, o = [], 0
        for sz in sizes:
            offs.append(o)
            o += sz + (-sz) % 4
        # (packed: the padding between parameters is part of the gradient handed back -- zeros, not stale memory)
        flat = (torch.zeros if ctx.flat_layout is not None else torch.empty)(o + (n * cfg.in_channels if want_dx else 0),
                                                                             dtype=torch.float32, device=dev)
        base, it = flat.data_ptr(), iter(offs)
        grads = (ctypes.c_void_p * len(ctx.slots))(*[base + 4 * next(it) if has else None for has in ctx.slots])
        dx = flat[o: o + n * cfg.in_channels].view(n, cfg.in_channels) if want_dx else None
        scratch = torch.empty(int(lib.dif_tiny_scratch_floats(n, d, cfg.num_layers)), dtype=torch.float32, device=dev)
        gyc = gy if gy.is_contiguous() else gy.contiguous()
        if ctx.flat_layout is not None:
            pa = ctx.pa
        else:
            li = iter(live)
            pa = _ptr_array([next(li) if has else None for has in ctx.slots])
        rc = lib.dif_tiny_backward_f32(ctypes.byref(cfg), xc.data_ptr(), ctx.ldx, pa,
                                       None if graph is None else graph.rowptr_t, None if graph is None else graph.dst_t,
                                       None if graph is None else graph.val_t,
                                       None if ctx.rnd is None else ctx.rnd.data_ptr(), ctx.tape.data_ptr(), gyc.data_ptr(),
                                       grads, None if dx is None else dx.data_ptr(), scratch.data_ptr(), _stream(dev))
        _lib.check(rc, "dif_tiny_backward_f32")
        stats["backward"] += 1
        if ctx.flat_layout is not None:
            return (None, dx, flat[:o])
        return (None, dx) + tuple(flat[a: a + sz].view(sh) for a, sz, sh in zip(offs, sizes, shapes))


def forward(model, x, edge_index, edge_weight):
    """-> y [n, out_channels], or None when this call takes the layer-by-layer path."""
    be = ops._BACKEND if ops._BACKEND is not None else ops.get_backend()
    if not getattr(be, "has_tiny", False):
        return None
    plan = _plan(model, x, edge_index, edge_weight)
    if plan is None:
        return None
    fields, params = plan
    graph = None
    if fields["use_graph"]:
        graph = graphs.get(edge_index, edge_weight, fields["n"])
        fields["gcn_scale"] *= graph.scale
    live = [p for p in params if p is not None]
    training = bool(model.training)
    p_drop = float(model.__dict__["dropout"]) if training else 0.0           # eval: dropout is the identity
    flat = None
    if PACK and torch.is_grad_enabled() and any(p.requires_grad for p in live):
        flat = _packed(model, live)
    if flat is not None:
        return _TinyModel.apply((fields, params, graph, p_drop, training, _layout(live)[0]), x, flat)
    return _TinyModel.apply((fields, params, graph, p_drop, training, None), x, *live)

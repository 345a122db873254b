"""Tensor-level wrapper of the C ABI: allocation (torch caching allocator), stream plumbing
(torch's current HIP stream) and layout checks.  No arithmetic happens here.

Every method requires float32 tensors resident on a ROCm device and raises otherwise --
the package has no CPU or eager path.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib


def _ptr(t):
    """Device address as a plain int (None = NULL): ctypes converts it for `c_void_p` argtypes itself, and a Python int is
    several times cheaper to make than a c_void_p object -- the host path has ~20 of these per call."""
    return t.data_ptr() if t is not None else None


# Part 0 of the row-sharded SpMM runs while the all-gather of the value rows is in flight.  A workgroup of the blocked
# kernel owns a whole CU, so it is launched on fewer workgroups than the 256 CUs and the collective's kernel keeps CUs
# of its own (RCCL uses up to ~32 channels = workgroups).
PART0_WORKGROUPS = int(os.environ.get("DIFFORMER_SPMM_PART0_WGS", "224"))

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(dev):
    """Raw handle of torch's current HIP stream on `dev` (the fast accessor when this torch has it)."""
    if _raw_stream is not None and dev.index is not None:
        return _raw_stream(dev.index)
    return torch.cuda.current_stream(dev).cuda_stream


def _require_device(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "difformer_amd: operands must live on the MI355X (got a CPU tensor); this package has no "
                "CPU fallback -- move the model and inputs to the GPU with .to('cuda')")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"difformer_amd: operands on different devices ({dev} vs {t.device})")
    return dev


def _f32(t, name):
    if t.dtype != torch.float32:
        raise TypeError(f"difformer_amd: {name} must be float32 (got {t.dtype}); the reference path is fp32 "
                        "(difformer.py:27)")
    return t


_SUFFIX = {torch.float32: "f32", torch.bfloat16: "bf16"}


def _storage(*tensors):
    """Common storage dtype of the operands -> C-ABI suffix ('f32' | 'bf16')."""
    dt = None
    for t in tensors:
        if t is None:
            continue
        if t.dtype not in _SUFFIX:
            raise TypeError(f"difformer_amd: unsupported dtype {t.dtype}: float32 (reference dtype) or bfloat16 "
                            "(storage-only variant) expected")
        if dt is None:
            dt = t.dtype
        elif t.dtype != dt:
            raise TypeError(f"difformer_amd: operands mix {dt} and {t.dtype}")
    return dt, _SUFFIX[dt]


def _row_major(t, width):
    """Return (tensor, leading dimension in elements) with the trailing dims dense so that
    row r starts at data_ptr + r*ld*4 and holds `width` contiguous floats.  Strided row views
    (e.g. column slices of a fused projection) are passed through without a copy."""
    ok = t.stride(-1) == 1 or t.shape[-1] == 1
    if ok and t.dim() == 3:
        ok = t.stride(1) == t.shape[2] or t.shape[1] == 1
    if ok and t.shape[0] > 1:
        ok = t.stride(0) >= width
    if not ok:
        t = t.contiguous()
    ld = t.stride(0) if t.shape[0] > 1 else width
    return t, int(ld)


class _Timed:
    """Optional HIP-event bracket around one C-ABI call (events on the launching stream)."""

    def __init__(self, backend, name, dev):
        only = backend.kernel_events_only          # bracket just these entry points (keeps the host ahead of the GPU)
        self.rec = backend.kernel_events if (only is None or name in only) else None
        self.name, self.dev = name, dev

    def __enter__(self):
        # device guard only when the operands live on another device than the current one
        self.ctx = None
        if self.dev.index is None or torch.cuda.current_device() != self.dev.index:
            self.ctx = torch.cuda.device(self.dev)
            self.ctx.__enter__()
        if self.rec is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record(torch.cuda.current_stream(self.dev))
        return self

    def __exit__(self, *exc):
        if self.rec is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record(torch.cuda.current_stream(self.dev))
            self.rec.setdefault(self.name, []).append((self.start, end))
        return self.ctx.__exit__(*exc) if self.ctx is not None else False


class _NoBracket:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_BRACKET = _NoBracket()


def _timed(backend, name, dev):
    """The common case -- no event collection, operands on the current device -- costs one shared no-op context."""
    if backend.kernel_events is None and dev.index is not None and torch.cuda.current_device() == dev.index:
        return _NO_BRACKET
    return _Timed(backend, name, dev)


class HipBackend:
    has_tiny = True          # the whole-model kernels for tiny graphs (tiny.py; csrc/tiny_model.hip) are in this library
    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        # bench.py sets this to a dict to collect (start, end) HIP events per entry point
        self.kernel_events = None
        self.kernel_events_only = None
        self._ones = {}                             # device -> float32 [1] = 1.0 (device-side beta of dif_rowgemm_f32)
        from . import ops
        self.lib.dif_set_exact_fp32(1 if ops.EXACT_FP32 else 0)      # a set_exact_fp32 made before the library was loaded

    def set_exact_fp32(self, flag):
        """The launchers' side of ops.set_exact_fp32: which matrix core their products take (dif_set_exact_fp32)."""
        return bool(self.lib.dif_set_exact_fp32(1 if flag else 0))

    def kernel_times_ms(self):
        """{entry point: [ms per call]} from the collected events (synchronises)."""
        torch.cuda.synchronize()
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in (self.kernel_events or {}).items()}

    # ---- a1 --------------------------------------------------------------------------------
    def simple_reduce(self, q, k, v):
        dev = _require_device(q, k, v)
        n, H, M = q.shape
        D = v.shape[2]
        _, sfx = _storage(q, k, v)
        q, ldq = _row_major(q, H * M)
        k, ldk = _row_major(k, H * M)
        v, ldv = _row_major(v, H * D)
        reduced = torch.empty(self.lib.dif_simple_reduced_len(H, M, D), dtype=torch.float32, device=dev)
        ws_bytes = self.lib.dif_simple_workspace_bytes(n, H, M, D)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        fn = getattr(self.lib, "dif_simple_reduce_" + sfx)
        with _timed(self, "dif_simple_reduce_f32", dev):
            rc = fn(_ptr(q), ldq, _ptr(k), ldk, _ptr(v), ldv, n, H, M, D, _ptr(reduced), _ptr(ws), ws_bytes, _stream(dev))
        _lib.check(rc, "dif_simple_reduce_" + sfx)
        return reduced

    def project_reduce(self, x, Wq, bq, Wk, bk, Wv, bv, H, D):
        """x [n,C] -> (q [n,H,D], v [n,H,D], reduced): projections fused with stage 1 of the simple kernel."""
        dev = _require_device(x, Wq, bq, Wk, bk, Wv, bv)
        n, C = x.shape
        dt, sfx = _storage(x, Wq, bq, Wk, bk, Wv, bv)
        x, ldx = _row_major(x, C)
        ws_ = [t.contiguous() for t in (Wq, bq, Wk, bk, Wv, bv)]
        q = torch.empty((n, H, D), dtype=dt, device=dev)
        v = torch.empty((n, H, D), dtype=dt, device=dev)
        reduced = torch.empty(self.lib.dif_simple_reduced_len(H, D, D), dtype=torch.float32, device=dev)
        ws_bytes = self.lib.dif_project_reduce_workspace_bytes(n, H, D)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        fn = getattr(self.lib, "dif_project_reduce_" + sfx)
        with _timed(self, "dif_project_reduce_f32", dev):
            rc = fn(_ptr(x), ldx, n, C, *[_ptr(t) for t in ws_], H, D, _ptr(q), H * D, _ptr(v), H * D, _ptr(reduced),
                    _ptr(ws), ws_bytes, _stream(dev))
        _lib.check(rc, "dif_project_reduce_" + sfx)
        return q, v, reduced

    def simple_apply(self, q, reduced, n_global, D):
        dev = _require_device(q, reduced)
        n, H, M = q.shape
        dt, sfx = _storage(q)
        _f32(reduced, "reduced")
        q, ldq = _row_major(q, H * M)
        out = torch.empty((n, H, D), dtype=dt, device=dev)
        fn = getattr(self.lib, "dif_simple_apply_" + sfx)
        with _timed(self, "dif_simple_apply_f32", dev):
            rc = fn(_ptr(q), ldq, _ptr(reduced), n, int(n_global), H, M, D, _ptr(out), H * D, _stream(dev))
        _lib.check(rc, "dif_simple_apply_" + sfx)
        return out

    # ---- a1 backward ------------------------------------------------------------------------
    def simple_backward(self, q, k, v, reduced, out, g, shard=None):
        """(dq, dk, dv) of the simple kernel for fp32 q,k [n,H,M], v/out/g [n,H,D] with M, D <= 512.  Row-sharded
        (`shard`, `reduced` already summed over the ranks): the sums over nodes of the backward -- q^T gn, sum gn, the
        ks-gradient -- are all-reduced in one buffer, then the scalar T: two small exchange steps."""
        dev = _require_device(q, k, v, reduced, out, g)
        n, H, M = q.shape
        sharded = shard is not None and shard.world > 1
        n_global = shard.n_global if sharded else n
        D = v.shape[2]
        for t_, nm in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (g, "grad")):
            _f32(t_, nm)
        # q, k, v are usually column slices of one fused projection [n, 3 H D]: every kernel below takes a leading
        # dimension, so they go in as they are (three 34-MB copies per layer at C4 otherwise)
        (q, ldq), (k, ldk), (v, ldv) = _row_major(q, H * M), _row_major(k, H * M), _row_major(v, H * D)
        out, g = out.contiguous(), g.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        gn, gd = torch.empty((n, H, D), **f32), torch.empty((n, H), **f32)
        sums = torch.empty(H * M + 1, **f32)
        ws_bytes = self.lib.dif_simple_bwd_workspace_bytes(n, H, M, D)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _timed(self, "dif_simple_bwd_prep_f32", dev):
            rc = self.lib.dif_simple_bwd_prep_f32(_ptr(q), ldq, _ptr(g), H * D, _ptr(out), H * D, _ptr(reduced), n,
                                                  int(n_global), H, M, D, _ptr(gn), _ptr(gd), _ptr(sums), _ptr(ws),
                                                  ws_bytes, _stream(dev))
        _lib.check(rc, "dif_simple_bwd_prep_f32")
        rec2 = self.simple_reduce(q, q, gn)                      # q^T gn, (sum q), sum gn
        if sharded:
            both = torch.cat([rec2, sums])
            shard.all_reduce_sum(both)
            rec2, sums = both[: rec2.numel()], both[rec2.numel():]
        # small per-head coefficient tensors, all on the device (no host synchronisation)
        hmd = H * M * D
        s = torch.rsqrt(reduced[-2]) * torch.rsqrt(reduced[-1])
        ktv_s = (reduced[:hmd] * s).contiguous()                 # s KtV          [H,M,D]
        ks_s = (reduced[hmd:hmd + H * M] * s).contiguous()       # s ks           [H,M]
        dktv = (rec2[:hmd] * s).contiguous()                     # s q^T gn       [H,M,D]
        dvs = rec2[hmd + H * M: hmd + H * M + H * D].contiguous()
        dks = (sums[: H * M] * s).contiguous()
        if M == D and (H * M) % 4 == 0:
            # dq | dk | dv as the column blocks of ONE [n, 3 H D] buffer: when q, k, v were the column slices of a fused
            # projection, the gradient of that projection is this buffer as it stands (autograd_ops._SplitColumns finds the
            # views adjacent and skips its concatenation: 308 MB of copy per layer at 100,000 x 128)
            fused = torch.empty((n, 3 * H * M), **f32)
            fused._difformer_fused_grad = True      # only a buffer made HERE may be completed in place by _SplitColumns
            dq, dk, dv = (fused[:, i * H * M: (i + 1) * H * M].view(n, H, M) for i in range(3))
            ldg = 3 * H * M
        else:
            dq, dk, dv = torch.empty((n, H, M), **f32), torch.empty((n, H, M), **f32), torch.empty((n, H, D), **f32)
            ldg = None

        def rowgemm(A, K, mat, mat_t, bias, r, u, cin, beta, C, dst, lda=None, ldc=None):
            with _timed(self, "dif_rowgemm_f32", dev):
                rc_ = self.lib.dif_rowgemm_f32(_ptr(A), lda or H * K, _ptr(mat), D, M * D, mat_t, 1.0, _ptr(bias), _ptr(r), _ptr(u),
                                               1.0, _ptr(cin), ldc or H * C, _ptr(beta), n, H, K, C, _ptr(dst), ldg or H * C,
                                               _stream(dev))
            _lib.check(rc_, "dif_rowgemm_f32")

        rowgemm(gn, D, ktv_s, 1, None, gd, ks_s, None, None, M, dq)   # dq_main = gn (s KtV)^T + gd (s ks)
        # T = s * dL/ds = sum q . dq_main.  (Algebraically also -(vs . dvs) - N sum gd, but those two terms cancel to
        # ~1e-4 of their size at N ~ 1e5; this form has no cancellation.)
        T = (q * dq).sum()
        if sharded:
            T = shard.all_reduce_sum(T.reshape(1))[0]
        dq.addcmul_(q, (-T / reduced[-2]).expand_as(q))               # - (T/|Q|^2) q
        beta_k = (-T / reduced[-1]).reshape(1).contiguous()
        rowgemm(v, D, dktv, 1, dks, None, None, k, beta_k, M, dk, lda=ldv, ldc=ldk)     # v dKtV^T + dks - (T/|K|^2) k
        rowgemm(k, M, dktv, 0, dvs, None, None, None, None, D, dv, lda=ldk)             # k dKtV + dvs
        return dq, dk, dv

    # ---- a2 --------------------------------------------------------------------------------
    def sigmoid_attention(self, q, k, v, want_den=False):
        """-> out [N,H,D]; want_den (fp32, training): (out, den float32 [N,H]) for sigmoid_backward."""
        dev = _require_device(q, k, v)
        N, H, M = q.shape
        L, D = k.shape[0], v.shape[2]
        dt, sfx = _storage(q, k, v)          # bfloat16: storage only, scores / sigma / sums in fp32
        q, ldq = _row_major(q, H * M)
        k, ldk = _row_major(k, H * M)
        v, ldv = _row_major(v, H * D)
        out = torch.empty((N, H, D), dtype=dt, device=dev)
        ws_bytes = self.lib.dif_sigmoid_workspace_bytes(N, L, H, M, D)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
        if want_den:
            _f32(q, "q")
            den = torch.empty((N, H), dtype=torch.float32, device=dev)
            with _timed(self, "dif_sigmoid_attn_fwd_f32", dev):
                rc = self.lib.dif_sigmoid_attn_fwd_f32(_ptr(q), ldq, _ptr(k), ldk, _ptr(v), ldv, N, L, H, M, D, _ptr(out),
                                                       H * D, _ptr(den), _ptr(ws), ws_bytes, _stream(dev))
            _lib.check(rc, "dif_sigmoid_attn_fwd_f32")
            return out, den
        name = "dif_sigmoid_attn_" + sfx
        with _timed(self, name, dev):
            rc = getattr(self.lib, name)(_ptr(q), ldq, _ptr(k), ldk, _ptr(v), ldv, N, L, H, M, D, _ptr(out), H * D, _ptr(ws),
                                         ws_bytes, _stream(dev))
        _lib.check(rc, name)
        return out

    def sigmoid_backward(self, q, k, v, out, den, g):
        """(dq, dk, dv) of the sigmoid kernel for fp32 q [N,H,M], k [L,H,M], v [L,H,D], out / g [N,H,D], den [N,H];
        M, D <= 512 (csrc/sigmoid_attn_bwd.hip up to 64 columns, csrc/sigmoid_wide.hip beyond: sigma recomputed tile by tile,
        nothing of size N x L stored)."""
        dev = _require_device(q, k, v, out, den, g)
        N, H, M = q.shape
        L, D = k.shape[0], v.shape[2]
        for t_, nm in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (den, "den"), (g, "grad")):
            _f32(t_, nm)
        q, ldq = _row_major(q, H * M)
        k, ldk = _row_major(k, H * M)
        v, ldv = _row_major(v, H * D)
        g, ldg = _row_major(g, H * D)
        out, den = out.contiguous(), den.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        dq, dk, dv = torch.empty((N, H, M), **f32), torch.empty((L, H, M), **f32), torch.empty((L, H, D), **f32)
        ws_bytes = self.lib.dif_sigmoid_bwd_workspace_bytes(N, L, H, M, D)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _timed(self, "dif_sigmoid_attn_bwd_f32", dev):
            rc = self.lib.dif_sigmoid_attn_bwd_f32(_ptr(q), ldq, _ptr(k), ldk, _ptr(v), ldv, _ptr(out), H * D, _ptr(den),
                                                   _ptr(g), ldg, N, L, H, M, D, _ptr(dq), H * M, _ptr(dk), H * M, _ptr(dv),
                                                   H * D, _ptr(ws), ws_bytes, _stream(dev))
        _lib.check(rc, "dif_sigmoid_attn_bwd_f32")
        return dq, dk, dv

    # ---- f4: batch of graphs (physical particle/difformer-v2.py:71-137) ---------------------
    def batched_simple_attention(self, q, k, v, graph_ptr, want_den=False):
        """graph_ptr: int32 [B+1] device tensor, graph b = rows [graph_ptr[b], graph_ptr[b+1]).
        want_den (training): -> (out, den float32 [N,H], sumsq float32 [2] = |Q|^2, |K|^2) for batched_simple_backward."""
        dev = _require_device(q, k, v, graph_ptr)
        N, H, M = q.shape
        D = v.shape[2]
        for t_, nm in ((q, "q"), (k, "k"), (v, "v")):
            _f32(t_, nm)
        if graph_ptr.dtype != torch.int32 or not graph_ptr.is_contiguous():
            raise TypeError("difformer_amd: graph_ptr must be a contiguous int32 tensor [B+1]")
        q, ldq = _row_major(q, H * M)
        k, ldk = _row_major(k, H * M)
        v, ldv = _row_major(v, H * D)
        out = torch.empty((N, H, D), dtype=torch.float32, device=dev)
        ws_bytes = self.lib.dif_batched_simple_workspace_bytes()
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        if want_den:
            den = torch.empty((N, H), dtype=torch.float32, device=dev)
            with _timed(self, "dif_batched_simple_attn_fwd_f32", dev):
                rc = self.lib.dif_batched_simple_attn_fwd_f32(_ptr(q), ldq, _ptr(k), ldk, _ptr(v), ldv, _ptr(graph_ptr),
                                                              graph_ptr.numel() - 1, N, H, M, D, _ptr(out), H * D, _ptr(den),
                                                              _ptr(ws), ws_bytes, _stream(dev))
            _lib.check(rc, "dif_batched_simple_attn_fwd_f32")
            return out, den, ws.view(torch.float32)[-2:].clone()
        with _timed(self, "dif_batched_simple_attn_f32", dev):
            rc = self.lib.dif_batched_simple_attn_f32(_ptr(q), ldq, _ptr(k), ldk, _ptr(v), ldv, _ptr(graph_ptr),
                                                      graph_ptr.numel() - 1, N, H, M, D, _ptr(out), H * D, _ptr(ws),
                                                      ws_bytes, _stream(dev))
        _lib.check(rc, "dif_batched_simple_attn_f32")
        return out

    def batched_simple_backward(self, q, k, v, out, den, sumsq, g, graph_ptr):
        """(dq, dk, dv) of batched_simple_attention: three launches of the forward kernel's raw mode
        (dif_batched_simple_raw_f32: per-graph sum of outer products applied to the graph's own rows) plus row-wise
        tensor arithmetic; formulas in include/difformer_hip.h."""
        dev = _require_device(q, k, v, out, den, sumsq, g, graph_ptr)
        N, H, M = q.shape
        D = v.shape[2]
        for t_, nm in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (den, "den"), (g, "grad")):
            _f32(t_, nm)
        q, k, v, g = (t_.contiguous() for t_ in (q, k, v, g))
        gn = (g / den.unsqueeze(-1)).contiguous()
        gd = (-(g * out).sum(dim=-1) / den).contiguous()
        B = graph_ptr.numel() - 1

        def raw(a, b, c, rs, vw, vs_is_s):
            Ma, Dc = a.shape[2], c.shape[2]
            dst = torch.empty((N, H, Dc), dtype=torch.float32, device=dev)
            with _timed(self, "dif_batched_simple_raw_f32", dev):
                rc = self.lib.dif_batched_simple_raw_f32(_ptr(a), H * Ma, _ptr(b), H * Ma, _ptr(c), H * Dc, _ptr(graph_ptr), B, N,
                                                         H, Ma, Dc, _ptr(sumsq), _ptr(rs), _ptr(vw), int(vs_is_s), _ptr(dst),
                                                         H * Dc, _stream(dev))
            _lib.check(rc, "dif_batched_simple_raw_f32")
            return dst

        dq = raw(gn, v, k, gd, None, 1)                 # s gn (sum v (x) k) + s gd ksum_b
        T = (q * dq).sum()                              # s dL/ds (no cancellation: as in simple_backward)
        dk = raw(v, gn, q, None, gd, 1)                 # s v (sum gn (x) q) + s sum gd q
        dv = raw(k, q, gn, None, None, 0)               # s k (sum q (x) gn) + sum gn
        dq.addcmul_(q, (-T / sumsq[0]).expand_as(q))
        dk.addcmul_(k, (-T / sumsq[1]).expand_as(k))
        return dq, dk, dv

    def batched_sigmoid_attention(self, q, k, v, ranked_first, pos_count, want_den=False):
        """ranked_first: int32 [B] first row of the r-th largest graph; pos_count: int32 [max_nodes].
        want_den (training, D <= 64): -> (out, den float32 [N, H]) for batched_sigmoid_backward."""
        dev = _require_device(q, k, v, ranked_first, pos_count)
        N, H, M = q.shape
        D = v.shape[2]
        for t_, nm in ((q, "q"), (k, "k"), (v, "v")):
            _f32(t_, nm)
        for t_ in (ranked_first, pos_count):
            if t_.dtype != torch.int32 or not t_.is_contiguous():
                raise TypeError("difformer_amd: ranked_first / pos_count must be contiguous int32 tensors")
        q, ldq = _row_major(q, H * M)
        k, ldk = _row_major(k, H * M)
        v, ldv = _row_major(v, H * D)
        out = torch.empty((N, H, D), dtype=torch.float32, device=dev)
        if want_den:
            den = torch.empty((N, H), dtype=torch.float32, device=dev)
            with _timed(self, "dif_batched_sigmoid_attn_fwd_f32", dev):
                rc = self.lib.dif_batched_sigmoid_attn_fwd_f32(_ptr(q), ldq, _ptr(k), ldk, _ptr(v), ldv, _ptr(ranked_first),
                                                               _ptr(pos_count), ranked_first.numel(), pos_count.numel(), H, M,
                                                               D, _ptr(out), H * D, _ptr(den), _stream(dev))
            _lib.check(rc, "dif_batched_sigmoid_attn_fwd_f32")
            return out, den
        with _timed(self, "dif_batched_sigmoid_attn_f32", dev):
            rc = self.lib.dif_batched_sigmoid_attn_f32(_ptr(q), ldq, _ptr(k), ldk, _ptr(v), ldv, _ptr(ranked_first),
                                                       _ptr(pos_count), ranked_first.numel(), pos_count.numel(), H, M, D,
                                                       _ptr(out), H * D, _stream(dev))
        _lib.check(rc, "dif_batched_sigmoid_attn_f32")
        return out

    def batched_sigmoid_backward(self, q, k, v, out, den, g, ranked_first, pos_count):
        """(dq, dk, dv) of the batched sigmoid attention (difformer-v2.py:113-135) for fp32 operands with M, D <= 64:
        csrc/sigmoid_attn_bwd.hip with the position groups' row mapping; nothing of size B x B x max_nodes is stored."""
        dev = _require_device(q, k, v, out, den, g, ranked_first, pos_count)
        N, H, M = q.shape
        D = v.shape[2]
        for t_, nm in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (den, "den"), (g, "grad")):
            _f32(t_, nm)
        q, ldq = _row_major(q, H * M)
        k, ldk = _row_major(k, H * M)
        v, ldv = _row_major(v, H * D)
        out, den, g = out.contiguous(), den.contiguous(), g.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        dq, dk, dv = torch.empty((N, H, M), **f32), torch.empty((N, H, M), **f32), torch.empty((N, H, D), **f32)
        ws_bytes = self.lib.dif_batched_sigmoid_bwd_workspace_bytes(N, H)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        with _timed(self, "dif_batched_sigmoid_attn_bwd_f32", dev):
            rc = self.lib.dif_batched_sigmoid_attn_bwd_f32(_ptr(q), ldq, _ptr(k), ldk, _ptr(v), ldv, _ptr(out), H * D, _ptr(den),
                                                           _ptr(g), H * D, _ptr(ranked_first), _ptr(pos_count),
                                                           ranked_first.numel(), pos_count.numel(), N, H, M, D, _ptr(dq), H * M,
                                                           _ptr(dk), H * M, _ptr(dv), H * D, _ptr(ws), ws_bytes, _stream(dev))
        _lib.check(rc, "dif_batched_sigmoid_attn_bwd_f32")
        return dq, dk, dv

    # ---- a3 --------------------------------------------------------------------------------
    def csr_build(self, edge_index, edge_weight, num_nodes, n_blocks=1, transpose=False, block_rows=0):
        dev = _require_device(edge_index, edge_weight)
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise TypeError("difformer_amd: edge_index must be an int64 tensor of shape [2, E]")
        ei = edge_index.contiguous()
        E = int(ei.shape[1])
        ew = None
        if edge_weight is not None:
            ew = _f32(edge_weight, "edge_weight").contiguous()
            if ew.numel() != E:
                raise ValueError("difformer_amd: edge_weight must have one entry per edge")
        rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
        src = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        val = torch.empty(max(E, 1), dtype=torch.float32, device=dev)
        status = torch.empty(2, dtype=torch.int32, device=dev)          # {index out of range, longest row}
        blkptr = None
        if n_blocks > 1:
            blkptr = torch.empty((n_blocks + 1) * num_nodes, dtype=torch.int32, device=dev)
        ws_bytes = self.lib.dif_csr_workspace_bytes(E, num_nodes, n_blocks)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _timed(self, "dif_csr_build", dev):
            rc = self.lib.dif_csr_build(_ptr(ei), E, num_nodes, _ptr(ew), n_blocks, int(block_rows), int(bool(transpose)), _ptr(rowptr),
                                        _ptr(blkptr),
                                        _ptr(src), _ptr(val), _ptr(status), _ptr(ws), ws_bytes, _stream(dev))
        _lib.check(rc, "dif_csr_build")
        bad, longest = status.tolist()  # one sync per (cold) build; the longest row rides along (kernel selection)
        if bad != 0:
            raise IndexError(f"difformer_amd: edge_index holds node ids outside [0, {num_nodes})")
        return rowptr, blkptr, src, val, int(longest)

    def subgraph(self, subset, edge_index, edge_weight, num_nodes):
        """Induced subgraph with relabelling (main-batch.py:131) -> (edge_index [2,E'], edge_weight [E'] | None)."""
        dev = _require_device(subset, edge_index, edge_weight)
        if edge_index.dtype != torch.int64 or subset.dtype != torch.int64:
            raise TypeError("difformer_amd: subset and edge_index must be int64")
        ei = edge_index.contiguous()
        sub = subset.contiguous()
        E, B = int(ei.shape[1]), int(sub.numel())
        ew = None if edge_weight is None else _f32(edge_weight, "edge_weight").contiguous()
        if E == 0:                                             # nothing to filter (an empty tensor has no address to hand over)
            if B and bool(((sub < 0) | (sub >= num_nodes)).any()):
                raise IndexError(f"difformer_amd: subset / edge_index hold node ids outside [0, {num_nodes})")
            return (torch.empty((2, 0), dtype=torch.int64, device=dev),
                    None if ew is None else torch.empty(0, dtype=torch.float32, device=dev))
        out_ei = torch.empty((2, max(E, 1)), dtype=torch.int64, device=dev)
        out_w = None if ew is None else torch.empty(max(E, 1), dtype=torch.float32, device=dev)
        res = torch.zeros(2, dtype=torch.int64, device=dev)    # [kept count | status word]: ONE read-back
        count, status = res[:1], res[1:].view(torch.int32)
        ws_bytes = self.lib.dif_subgraph_workspace_bytes(E, num_nodes)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _timed(self, "dif_subgraph", dev):
            rc = self.lib.dif_subgraph(_ptr(ei), E, num_nodes, _ptr(sub), B, _ptr(ew), _ptr(out_ei), _ptr(out_w),
                                       _ptr(count), _ptr(status), _ptr(ws), ws_bytes, _stream(dev))
        _lib.check(rc, "dif_subgraph")
        kept, bad = res.tolist()                                # one sync: the result size is data dependent
        bad &= 0xFFFFFFFF
        if bad:
            raise IndexError(f"difformer_amd: subset / edge_index hold node ids outside [0, {num_nodes})")
        out = torch.stack([out_ei[0, :kept], out_ei[1, :kept]])
        return out, (None if out_w is None else out_w[:kept].clone())

    def graph_prepare(self, edge_index, num_nodes, undirected=False, remove_loops=False, add_loops=False):
        """to_undirected -> remove_self_loops -> add_self_loops (any subset, in that order) on the device -> [2, E']."""
        dev = _require_device(edge_index)
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise TypeError("difformer_amd: edge_index must be an int64 tensor of shape [2, E]")
        ei = edge_index.contiguous()
        E = int(ei.shape[1])
        cap = (2 * E if undirected else E) + (num_nodes if add_loops else 0)
        out = torch.empty((2, max(cap, 1)), dtype=torch.int64, device=dev)
        res = torch.zeros(2, dtype=torch.int64, device=dev)    # [kept count | status word]: ONE read-back
        count, status = res[:1], res[1:].view(torch.int32)
        ws_bytes = self.lib.dif_graph_prepare_workspace_bytes(E, num_nodes, int(bool(undirected)))
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)
        with _timed(self, "dif_graph_prepare", dev):
            rc = self.lib.dif_graph_prepare(_ptr(ei), E, num_nodes, int(bool(undirected)), int(bool(remove_loops)),
                                            int(bool(add_loops)), max(cap, 1), _ptr(out), _ptr(count), _ptr(status), _ptr(ws),
                                            ws_bytes, _stream(dev))
        _lib.check(rc, "dif_graph_prepare")
        kept, bad = res.tolist()                                # one sync: the result size is data dependent
        bad &= 0xFFFFFFFF
        if bad:
            raise IndexError(f"difformer_amd: edge_index holds node ids outside [0, {num_nodes})")
        return torch.stack([out[0, :kept], out[1, :kept]])

    def subgraph_batches(self, perm, batch_size, edge_index, edge_weight, num_nodes, build_csr=False):
        """All induced subgraphs of an epoch (main-batch.py:121-131) from one pass over the edge list ->
        (edge_index [2, kept] grouped by batch, edge_weight [kept] | None, batch_ptr: python list of n_batches + 1 ints,
        csr | None) with csr = (rowptr int32 [M + 1], src int32 [kept], val float32 [kept]) over all batches when
        build_csr (one sort instead of one dif_csr_build per batch)."""
        dev = _require_device(perm, edge_index, edge_weight)
        if edge_index.dtype != torch.int64 or perm.dtype != torch.int64:
            raise TypeError("difformer_amd: perm and edge_index must be int64")
        ei, pm = edge_index.contiguous(), perm.contiguous()
        E, M = int(ei.shape[1]), int(pm.numel())
        nb = -(-M // int(batch_size))
        ew = None if edge_weight is None else _f32(edge_weight, "edge_weight").contiguous()
        bptr = torch.empty(nb + 1, dtype=torch.int64, device=dev)
        status = torch.empty(1, dtype=torch.int32, device=dev)
        ws_bytes = self.lib.dif_subgraph_batches_workspace_bytes(E, num_nodes, nb)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _timed(self, "dif_subgraph_batches", dev):
            rc = self.lib.dif_subgraph_batches_group(_ptr(ei), E, num_nodes, _ptr(pm), M, int(batch_size), _ptr(bptr),
                                                     _ptr(status), _ptr(ws), ws_bytes, _stream(dev))
        _lib.check(rc, "dif_subgraph_batches_group")
        ptr = bptr.tolist()                                   # one sync per epoch: the result size is data dependent
        bad = int(status.item())
        if bad & 1:
            raise IndexError(f"difformer_amd: perm / edge_index hold node ids outside [0, {num_nodes})")
        if bad & 2:
            raise ValueError("difformer_amd: perm repeats a node id")
        kept = ptr[-1]
        out_ei = torch.empty((2, max(kept, 1)), dtype=torch.int64, device=dev)
        out_w = None if ew is None else torch.empty(max(kept, 1), dtype=torch.float32, device=dev)
        with _timed(self, "dif_subgraph_batches", dev):
            rc = self.lib.dif_subgraph_batches_emit(_ptr(ei), E, num_nodes, M, int(batch_size), _ptr(ew), _ptr(bptr), kept,
                                                    _ptr(out_ei), _ptr(out_w), _ptr(ws), ws_bytes, _stream(dev))
        _lib.check(rc, "dif_subgraph_batches_emit")
        csr = None
        if build_csr:
            rowptr = torch.empty(M + 1, dtype=torch.int32, device=dev)
            src = torch.empty(max(kept, 1), dtype=torch.int32, device=dev)
            val = torch.empty(max(kept, 1), dtype=torch.float32, device=dev)
            ws2_bytes = self.lib.dif_subgraph_batches_csr_workspace_bytes(kept, M)
            ws2 = torch.empty(ws2_bytes, dtype=torch.uint8, device=dev)
            with _timed(self, "dif_subgraph_batches", dev):
                rc = self.lib.dif_subgraph_batches_csr(_ptr(ei), E, num_nodes, M, int(batch_size), _ptr(ew), kept, _ptr(ws),
                                                       ws_bytes, _ptr(rowptr), _ptr(src), _ptr(val), _ptr(ws2), ws2_bytes,
                                                       _stream(dev))
            _lib.check(rc, "dif_subgraph_batches_csr")
            csr = (rowptr, src, val)
        return out_ei[:, :kept], (None if out_w is None else out_w[:kept]), ptr, csr

    def edge_weight_grad(self, edge_index, edge_weight, rowptr, n_nodes, g, x, scale=1.0):
        """d loss / d edge_weight of the aggregation (difformer.py:73 under autograd): g, x [n_nodes, F] -> [E]."""
        dev = _require_device(edge_index, edge_weight, rowptr, g, x)
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise TypeError("difformer_amd: edge_index must be int64 [2, E]")
        E, F = edge_index.shape[1], x.shape[1]
        if g.shape != x.shape or x.shape[0] != n_nodes:
            raise ValueError(f"difformer_amd: edge_weight_grad needs g and x as [{n_nodes}, F] (got {tuple(g.shape)}, {tuple(x.shape)})")
        _f32(g, "g"), _f32(x, "x")
        w = _f32(edge_weight, "edge_weight").detach().contiguous()
        ei = edge_index.contiguous()
        g, ldg = _row_major(g, F)
        x, ldx = _row_major(x, F)
        dw = torch.empty(E, dtype=torch.float32, device=dev)
        with _timed(self, "dif_gcn_edge_weight_grad_f32", dev):
            rc = self.lib.dif_gcn_edge_weight_grad_f32(_ptr(ei), E, n_nodes, _ptr(w), _ptr(rowptr), _ptr(g), ldg, _ptr(x), ldx,
                                                       F, float(scale), _ptr(dw), _stream(dev))
        _lib.check(rc, "dif_gcn_edge_weight_grad_f32")
        return dw

    def spmm(self, rowptr, blkptr, n_blocks, src, val, n_nodes, nnz, x, row_begin, n_rows, attn=None, attn_scale=1.0,
             gcn_scale=1.0, tail=None, order=None, part=None):
        """tail = None | dict(x0, prev, alpha, ln_weight, ln_bias, eps[, relu]): fuse the layer tail (H == 1).
        part = None | (0 | 1, own_blk_begin, own_blk_end, scratch | None, x_row0): one half of a split product
        (row-sharded runs).  Part 0 takes `x` = this rank's OWN value rows, whose first row is source row x_row0 (the
        sources of blocks own_blk_begin .. own_blk_end-1), parks the accumulators and returns the scratch tensor; part 1
        takes all n_nodes gathered rows plus that scratch and returns the finished rows.
        order = None | (int32 [n_rows], n_split) from row_order(): degree-sorted rows (the first n_split of them split
        over a whole quad) for the blocked kernel's load balance."""
        order, n_split = order if order is not None else (None, 0)
        dev = _require_device(rowptr, blkptr, src, val, x, attn, order)
        if order is not None and (order.dtype != torch.int32 or order.numel() != n_rows or not order.is_contiguous()):
            raise TypeError("difformer_amd: order must be a contiguous int32 tensor with one entry per local row")
        F = x.shape[1]
        t = tail or {}
        x0, prev, lw, lb = t.get("x0"), t.get("prev"), t.get("ln_weight"), t.get("ln_bias")
        _require_device(x0, prev, lw, lb)
        dt, sfx = _storage(x, attn, x0, prev, lw, lb)
        _f32(val, "CSR values")
        x, ldx = _row_major(x, F)
        phase, own_lo, own_hi, scratch, x_row0 = part if part is not None else (None, 0, 0, None, 0)
        if phase != 0 and x.shape[0] != n_nodes:
            raise ValueError(f"difformer_amd: spmm needs all {n_nodes} source rows, got {x.shape[0]}")
        lda = ldx0 = ldp = 0
        if attn is not None:
            attn, lda = _row_major(attn, F)
        if x0 is not None:
            x0, ldx0 = _row_major(x0, F)
        if prev is not None:
            prev, ldp = _row_major(prev, F)
        if lw is not None:
            lw, lb = lw.contiguous(), lb.contiguous()
        out = torch.empty((n_rows, F), dtype=dt, device=dev)
        head = (_ptr(rowptr), _ptr(blkptr), n_blocks, _ptr(src), _ptr(val), n_nodes, nnz, _ptr(x), ldx, row_begin, n_rows, F,
                _ptr(attn), lda, float(attn_scale), float(gcn_scale), _ptr(order), int(n_split))
        tail_args = (_ptr(x0), ldx0, _ptr(prev), ldp, float(t.get("alpha", 0.5)), _ptr(lw), _ptr(lb),
                     float(t.get("eps", 1e-5)), int(bool(t.get("relu", False))))
        if part is not None:
            sbytes = self.lib.dif_gcn_spmm_part_scratch_bytes(n_rows, int(n_split), F)
            if scratch is None:
                scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev)
            # part 0 indexes x by GLOBAL source row but only ever touches this rank's own rows: hand it the pointer
            # the first own row would have inside a full [n_nodes, F] array
            xp = ctypes.c_void_p(x.data_ptr() - int(x_row0) * ldx * x.element_size())
            head = head[:7] + (xp,) + head[8:]
            fn = self.lib.dif_gcn_spmm_part_bf16 if sfx == "bf16" else self.lib.dif_gcn_spmm_part_f32
            with _timed(self, "dif_gcn_spmm_f32", dev):
                rc = fn(*head, int(tail is not None), *tail_args, int(phase), int(own_lo), int(own_hi),
                        PART0_WORKGROUPS if phase == 0 else 0, _ptr(scratch), sbytes, _ptr(out), F, _stream(dev))
            _lib.check(rc, "dif_gcn_spmm_part")
            return scratch if phase == 0 else out
        with _timed(self, "dif_gcn_spmm_f32", dev):
            if sfx == "bf16":
                name = "dif_gcn_spmm_tail_bf16"
                rc = self.lib.dif_gcn_spmm_tail_bf16(*head, int(tail is not None), *tail_args, _ptr(out), F, _stream(dev))
            elif tail is None:
                name = "dif_gcn_spmm_f32"
                rc = self.lib.dif_gcn_spmm_f32(*head, _ptr(out), F, _stream(dev))
            else:
                name = "dif_gcn_spmm_tail_f32"
                rc = self.lib.dif_gcn_spmm_tail_f32(*head, *tail_args, _ptr(out), F, _stream(dev))
        _lib.check(rc, name)
        return out

    # ---- a1 + a4 + tail in closed form (csrc/simple_layer.hip) ---------------------------------------------------------
    def gram(self, x, rowptr=None, plan=None):
        """x [n, C] fp32 -> (record [C*C + C + 2] = {X^T X, column sums}, ys | None).  With rowptr + plan the pass also
        writes the slice-major copy of x scaled by deg^-1/2 that sliced_spmm reads."""
        dev = _require_device(x, rowptr)
        dt, sfx = _storage(x)
        n, C = x.shape
        x, ldx = _row_major(x, C)
        if ldx % 4 or x.data_ptr() % (4 * x.element_size()):
            x, ldx = x.contiguous(), C
        record = torch.empty(C * C + C + 2, dtype=torch.float32, device=dev)
        ws_bytes = self.lib.dif_gram_workspace_bytes(n, C)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        if sfx == "bf16":        # bfloat16 rows, float32 record; no slice-major copy (the sliced product is float32-only)
            if plan is not None:
                raise TypeError("difformer_amd: the slice-major copy of the sliced product is float32-only")
            with _timed(self, "dif_gram_f32", dev):
                rc = self.lib.dif_gram_bf16(_ptr(x), ldx, n, C, _ptr(record), _ptr(ws), ws_bytes, _stream(dev))
            _lib.check(rc, "dif_gram_bf16")
            return record, None
        ys = None
        if plan is not None:
            ys = torch.empty((C // 4, int(plan[6]) * int(plan[7]), 4), dtype=torch.float32, device=dev)
        with _timed(self, "dif_gram_f32", dev):
            rc = self.lib.dif_gram_f32(_ptr(x), ldx, n, C, _ptr(rowptr) if plan is not None else None, plan, _ptr(ys),
                                       _ptr(record), _ptr(ws), ws_bytes, _stream(dev))
        _lib.check(rc, "dif_gram_f32")
        return record, ys

    def input_gram(self, x, weight, bias, ln_weight, ln_bias, eps, relu, rowptr=None, plan=None):
        """Input layer + the first closed-form layer's products in one pass (csrc/simple_layer.hip, input_gram_kernel):
        x [n, C_in <= 64] fp32 -> (h = ReLU(LayerNorm(x W^T + b)) [n, D], record of h as `gram` leaves it, ys | None)."""
        dev = _require_device(x, weight, bias, ln_weight, ln_bias, rowptr)
        for t, name in ((x, "x"), (weight, "weight"), (bias, "bias")):
            _f32(t, name)
        n, C = x.shape
        D = weight.shape[0]
        x, ldx = _row_major(x, C)
        weight, bias = weight.contiguous(), bias.contiguous()
        if ln_weight is not None:
            ln_weight, ln_bias = ln_weight.contiguous(), ln_bias.contiguous()
        out = torch.empty((n, D), dtype=torch.float32, device=dev)
        record = torch.empty(D * D + D + 2, dtype=torch.float32, device=dev)
        ws_bytes = self.lib.dif_gram_workspace_bytes(n, D)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        ys = None
        if plan is not None:
            ys = torch.empty((D // 4, int(plan[6]) * int(plan[7]), 4), dtype=torch.float32, device=dev)
        with _timed(self, "dif_input_gram_f32", dev):
            rc = self.lib.dif_input_gram_f32(_ptr(x), ldx, n, C, _ptr(weight), _ptr(bias), D, _ptr(ln_weight), _ptr(ln_bias),
                                             float(eps), int(bool(relu)), _ptr(out), D, _ptr(rowptr) if plan is not None else None,
                                             plan, _ptr(ys), _ptr(record), _ptr(ws), ws_bytes, _stream(dev))
        _lib.check(rc, "dif_input_gram_f32")
        return out, record, ys

    def simple_coeffs(self, record, n_global, C, D, Wq, bq, Wk, bk, Wv, bv, attn_scale):
        dev = _require_device(record, Wq, bq, Wk, bk, Wv, bv)
        ws_ = [None if t is None else _f32(t, "weight").contiguous() for t in (Wq, bq, Wk, bk, Wv, bv)]
        coef = torch.empty(self.lib.dif_simple_coeffs_len(C, D), dtype=torch.float32, device=dev)
        with _timed(self, "dif_simple_coeffs_f32", dev):
            rc = self.lib.dif_simple_coeffs_f32(_ptr(record), int(n_global), C, D, *[_ptr(t) for t in ws_], float(attn_scale),
                                                _ptr(coef), _stream(dev))
        _lib.check(rc, "dif_simple_coeffs_f32")
        return coef

    def gram_coeffs(self, x, n_global, C, D, Wq, bq, Wk, bk, Wv, bv, attn_scale):
        """gram(x) + simple_coeffs in one call (dif_gram_coeffs_f32): x [n, C] float32 -> (record, coef).  Up to 48 partial
        records of the Gram pass (<= 24,576 rows) are summed inside the coefficient kernel: two launches instead of three."""
        dev = _require_device(x, Wq, bq, Wk, bk, Wv, bv)
        _f32(x, "x")
        n = x.shape[0]
        x, ldx = _row_major(x, C)
        if ldx % 4 or x.data_ptr() % 16:
            x, ldx = x.contiguous(), C
        ws_ = [None if t is None else _f32(t, "weight").contiguous() for t in (Wq, bq, Wk, bk, Wv, bv)]
        record = torch.empty(C * C + C + 2, dtype=torch.float32, device=dev)
        coef = torch.empty(self.lib.dif_simple_coeffs_len(C, D), dtype=torch.float32, device=dev)
        ws_bytes = self.lib.dif_gram_workspace_bytes(n, C)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        with _timed(self, "dif_gram_coeffs_f32", dev):
            rc = self.lib.dif_gram_coeffs_f32(_ptr(x), ldx, n, C, D, *[_ptr(t) for t in ws_], int(n_global), float(attn_scale),
                                              _ptr(coef), _ptr(record), _ptr(ws), ws_bytes, _stream(dev))
        _lib.check(rc, "dif_gram_coeffs_f32")
        return record, coef

    def row_gemm(self, A, mat, bias=None, accumulate=None):
        """A [n, K] @ mat [K, C] (+ bias [C]) (+ accumulate [n, C]) in one pass over the rows (dif_rowgemm_f32; K <= 512,
        float32) -> [n, C], or None when the shape is not covered (the caller then uses the vendor GEMM)."""
        dev = _require_device(A, mat, bias, accumulate)
        n, K = A.shape
        C = mat.shape[1]
        if K > 512 or any(t_ is not None and t_.dtype != torch.float32 for t_ in (A, mat, bias, accumulate)):
            return None
        A, lda = _row_major(A, K)
        mat = mat.contiguous()
        ldc = 0
        one = None
        if accumulate is not None:
            accumulate, ldc = _row_major(accumulate, C)
            one = self._ones.get(dev)
            if one is None:
                one = self._ones[dev] = torch.ones(1, dtype=torch.float32, device=dev)
        out = torch.empty((n, C), dtype=torch.float32, device=dev)
        with _timed(self, "dif_rowgemm_f32", dev):
            rc = self.lib.dif_rowgemm_f32(_ptr(A), lda, _ptr(mat), C, 0, 0, 1.0, _ptr(None if bias is None else bias.contiguous()),
                                          None, None, 1.0, _ptr(accumulate), ldc, _ptr(one), n, 1, K, C, _ptr(out), C,
                                          _stream(dev))
        _lib.check(rc, "dif_rowgemm_f32")
        return out

    def closed_form_attn_backward(self, x, coef, D, d, dx_in=None, row_sums=None):
        """Backward of att = (x Mn + cn) / (x u + cd) in one pass (csrc/simple_layer.hip, closed_form_attn_bwd_kernel):
        -> (d_num [n, D], d_den [n], dx [n, C] = dx_in + d_num Mn^T + d_den u^T, d_u [C] = x^T d_den, d_cd [] = sum d_den,
        rs_d [D] = row_sums^T d or None), or None when the shape is not covered."""
        dev = _require_device(x, coef, d, dx_in, row_sums)
        n, C = x.shape
        if C % 4 or D % 4 or C > 64 or D > 64 or any(t_ is not None and t_.dtype != torch.float32 for t_ in (x, coef, d, dx_in)):
            return None
        x, ldx = _row_major(x, C)
        d, ldd = _row_major(d, D)
        ldi = 0
        if dx_in is not None:
            dx_in, ldi = _row_major(dx_in, C)
        if any(ld % 4 for ld in (ldx, ldd, ldi)) or any(t_ is not None and t_.data_ptr() % 16 for t_ in (x, d, dx_in)):
            return None
        d_num = torch.empty((n, D), dtype=torch.float32, device=dev)
        d_den = torch.empty((n,), dtype=torch.float32, device=dev)
        dx = torch.empty((n, C), dtype=torch.float32, device=dev)
        if row_sums is not None:
            row_sums = _f32(row_sums, "row_sums").contiguous()
        parts = torch.empty((self.lib.dif_closed_form_attn_bwd_groups(n), 132), dtype=torch.float32, device=dev)
        with _timed(self, "dif_simple_layer_f32", dev):
            rc = self.lib.dif_closed_form_attn_bwd_f32(_ptr(x), ldx, n, C, D, _ptr(coef), _ptr(d), ldd, _ptr(dx_in), ldi,
                                                       _ptr(d_num), _ptr(d_den), _ptr(dx), C, _ptr(row_sums), _ptr(parts),
                                                       _stream(dev))
        _lib.check(rc, "dif_closed_form_attn_bwd_f32")
        sums = parts.sum(dim=0)                      # one partial record per workgroup, added in a fixed order
        return d_num, d_den, dx, sums[:C], sums[128], (sums[64: 64 + D] if row_sums is not None else None)

    def simple_coeffs_backward(self, record, n_global, C, D, Wq, bq, Wk, bk, Wv, bv, attn_scale, coef, dcoef):
        """Backward of simple_coeffs in one launch (csrc/simple_coeffs_bwd.hip): dcoef in coef's layout ->
        (S [C, C], t [C], dWq, dbq, dWk, dbk, dWv | None, dbv | None): dx = x S + 1 t^T through the record."""
        dev = _require_device(record, Wq, bq, Wk, bk, Wv, bv, coef, dcoef)
        ws_ = [None if t_ is None else _f32(t_, "weight").contiguous() for t_ in (Wq, bq, Wk, bk, Wv, bv)]
        if dcoef.numel() < D * C + D + C + 1 or not dcoef.is_contiguous():
            raise TypeError("difformer_amd: dcoef must be contiguous with coef's layout [D*C | D | C | 1]")
        out = torch.empty(self.lib.dif_simple_coeffs_bwd_len(C, D), dtype=torch.float32, device=dev)
        with _timed(self, "dif_simple_coeffs_f32", dev):
            rc = self.lib.dif_simple_coeffs_bwd_f32(_ptr(record), int(n_global), C, D, *[_ptr(t_) for t_ in ws_],
                                                    float(attn_scale), _ptr(_f32(coef, "coef")), _ptr(_f32(dcoef, "dcoef")),
                                                    _ptr(out), _stream(dev))
        _lib.check(rc, "dif_simple_coeffs_bwd_f32")
        o = 0
        parts = []
        for shape in ((C, C), (C,), (D, C), (D,), (D, C), (D,), (D, C), (D,)):
            k = shape[0] * (shape[1] if len(shape) > 1 else 1)
            parts.append(out[o: o + k].view(shape))
            o += k
        if Wv is None:
            parts[6] = parts[7] = None
        return parts

    def simple_layer(self, x, coef, D, ax=None, Wv=None, bv=None, row_sums=None, gcn_scale=1.0, x0=None, residual=False,
                     alpha=0.5, ln_weight=None, ln_bias=None, eps=1e-5, relu=False, next_rowptr=None, next_plan=None,
                     next_record=False, head=None, gather=None):
        """-> out [n, D]; with next_plan -> (out, ys, record | None): also the slice-major scaled copy of `out` for the
        next layer's SpMM (see gram()), and with next_record its Gram record from the same pass.
        head = (Wo [Co, D], bo [Co]) float32, Co <= 128 (the model's output Linear, difformer.py:208): -> logits [n, Co]
        from the same pass; the layer's rows themselves are not stored.
        gather = (rowptr, src, val) of a one-block CSR over the same n nodes: the aggregation runs inside the layer kernel
        (no `ax`, no `row_sums`, no next-layer products)."""
        dev = _require_device(x, coef, ax, Wv, bv, row_sums, x0, ln_weight, ln_bias)
        if gather is not None:
            return self._simple_layer_gather(x, coef, D, gather, Wv, bv, gcn_scale, x0, residual, alpha, ln_weight, ln_bias,
                                             eps, relu, head, ax is not None or next_plan is not None or next_record)
        dt, sfx = _storage(x, ax, x0)              # activations: float32 or bfloat16; parameters always float32 here
        for t_, nm in ((coef, "coef"), (Wv, "Wv"), (bv, "bv"), (ln_weight, "ln_weight"), (ln_bias, "ln_bias"), (row_sums, "row_sums")):
            if t_ is not None:
                _f32(t_, nm)
        n, C = x.shape
        x, ldx = _row_major(x, C)
        if ldx % 4 or x.data_ptr() % (4 * x.element_size()):
            x, ldx = x.contiguous(), C
        ldax = ldx0 = 0
        if ax is not None:
            ax, ldax = _row_major(ax, C)
        if x0 is not None:
            x0, ldx0 = _row_major(x0, D)
        if Wv is not None:
            Wv, bv = Wv.contiguous(), bv.contiguous()
        if ln_weight is not None:
            ln_weight, ln_bias = ln_weight.contiguous(), ln_bias.contiguous()
        if head is not None:
            Wo, bo = (_f32(t_, "head").contiguous() for t_ in head)          # float32 (exact copies of bf16 parameters)
            Co = Wo.shape[0]
            if next_plan is not None or next_record or Co > 128 or Wo.shape[1] != D:
                raise TypeError("difformer_amd: the fused output Linear needs Co <= 128 and no next-layer products")
            logits = torch.empty((n, Co), dtype=dt, device=dev)
            fn = self.lib.dif_simple_layer_head_bf16 if sfx == "bf16" else self.lib.dif_simple_layer_head_f32
            with _timed(self, "dif_simple_layer_f32", dev):
                rc = fn(_ptr(x), ldx, n, C, D, _ptr(coef), _ptr(ax), ldax, _ptr(Wv), _ptr(bv),
                        _ptr(row_sums), float(gcn_scale), _ptr(x0), ldx0, int(bool(residual)),
                        float(alpha), _ptr(ln_weight), _ptr(ln_bias), float(eps),
                        int(bool(relu)), None, 0, _ptr(Wo), _ptr(bo), Co, _ptr(logits), Co,
                        _stream(dev))
            _lib.check(rc, "dif_simple_layer_head")
            return logits
        out = torch.empty((n, D), dtype=dt, device=dev)
        record = ys = ws = None
        ws_bytes = 0
        if sfx == "bf16":
            if next_plan is not None or next_record:
                raise TypeError("difformer_amd: products for the next layer are float32-only")
            with _timed(self, "dif_simple_layer_f32", dev):
                rc = self.lib.dif_simple_layer_bf16(_ptr(x), ldx, n, C, D, _ptr(coef), _ptr(ax), ldax, _ptr(Wv), _ptr(bv),
                                                    _ptr(row_sums), float(gcn_scale), _ptr(x0), ldx0, int(bool(residual)),
                                                    float(alpha), _ptr(ln_weight), _ptr(ln_bias), float(eps),
                                                    int(bool(relu)), _ptr(out), D, _stream(dev))
            _lib.check(rc, "dif_simple_layer_bf16")
            return out
        if next_plan is not None:
            ys = torch.empty((D // 4, int(next_plan[6]) * int(next_plan[7]), 4), dtype=torch.float32, device=dev)
        if next_record:
            record = torch.empty(D * D + D + 2, dtype=torch.float32, device=dev)
            ws_bytes = self.lib.dif_gram_workspace_bytes(n, D)
            ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        with _timed(self, "dif_simple_layer_f32", dev):
            rc = self.lib.dif_simple_layer_f32(_ptr(x), ldx, n, C, D, _ptr(coef), _ptr(ax), ldax, _ptr(Wv), _ptr(bv),
                                               _ptr(row_sums), float(gcn_scale), _ptr(x0), ldx0, int(bool(residual)),
                                               float(alpha), _ptr(ln_weight), _ptr(ln_bias), float(eps), int(bool(relu)),
                                               _ptr(out), D, _ptr(record), _ptr(next_rowptr) if ys is not None else None,
                                               next_plan if ys is not None else None, _ptr(ys), _ptr(ws), ws_bytes,
                                               _stream(dev))
        _lib.check(rc, "dif_simple_layer_f32")
        if next_plan is None and not next_record:
            return out
        return out, ys, record

    def _simple_layer_gather(self, x, coef, D, gather, Wv, bv, gcn_scale, x0, residual, alpha, ln_weight, ln_bias, eps, relu,
                             head, conflicting):
        dev = x.device
        dt, sfx = _storage(x, None, x0)
        rowptr, src, val = gather
        n, C = x.shape
        if conflicting or rowptr.numel() != n + 1 or rowptr.dtype != torch.int32 or src.dtype != torch.int32:
            raise TypeError("difformer_amd: the in-kernel aggregation takes an int32 one-block CSR over the rows of x and "
                            "neither ax nor next-layer products")
        for t_, nm in ((coef, "coef"), (Wv, "Wv"), (bv, "bv"), (ln_weight, "ln_weight"), (ln_bias, "ln_bias"), (val, "val")):
            if t_ is not None:
                _f32(t_, nm)
        x, ldx = _row_major(x, C)
        if ldx % 4 or x.data_ptr() % (4 * x.element_size()):
            x, ldx = x.contiguous(), C
        ldx0 = 0
        if x0 is not None:
            x0, ldx0 = _row_major(x0, D)
        if Wv is not None:
            Wv, bv = Wv.contiguous(), bv.contiguous()
        if ln_weight is not None:
            ln_weight, ln_bias = ln_weight.contiguous(), ln_bias.contiguous()
        Wo = bo = out = logits = None
        Co = 0
        if head is not None:
            Wo, bo = (_f32(t_, "head").contiguous() for t_ in head)
            Co = Wo.shape[0]
            if Co > 128 or Wo.shape[1] != D:
                raise TypeError("difformer_amd: the fused output Linear needs Co <= 128")
            logits = torch.empty((n, Co), dtype=dt, device=dev)
        else:
            out = torch.empty((n, D), dtype=dt, device=dev)
        fn = self.lib.dif_simple_layer_gather_bf16 if sfx == "bf16" else self.lib.dif_simple_layer_gather_f32
        with _timed(self, "dif_simple_layer_f32", dev):
            rc = fn(_ptr(x), ldx, n, C, D, _ptr(coef), _ptr(rowptr), _ptr(src), _ptr(val), _ptr(Wv), _ptr(bv),
                    float(gcn_scale), _ptr(x0), ldx0, int(bool(residual)), float(alpha), _ptr(ln_weight), _ptr(ln_bias),
                    float(eps), int(bool(relu)), _ptr(out), D, _ptr(Wo), _ptr(bo), Co, _ptr(logits), Co, _stream(dev))
        _lib.check(rc, "dif_simple_layer_gather")
        return logits if head is not None else out

    # ---- a3, dense unweighted graphs: feature-sliced product with LDS-staged sources (csrc/gcn_sliced.hip) ----------
    def sliced_plan(self, n_src, n_rows, F):
        """int32[8] geometry (ctypes array) or None when the shape is not covered."""
        plan = (ctypes.c_int32 * 8)()
        rc = self.lib.dif_sliced_plan(int(n_src), int(n_rows), int(F), plan)
        return plan if rc == 0 else None

    def sliced_build(self, rowptr, blkptr, src, n_src, nnz, row_begin, n_rows, F, plan, order=None, parts=None, n_pos=None):
        """-> (entries uint16 [512 * n_blocks], table int32) or None when a (row position, tile) group exceeds the 16-bit
        counters.  order: the shard's rows by descending degree (row_order) for skewed graphs, None = natural order;
        parts (uint16 [n_pos], with order int32 [n_pos]): hub rows split into lock-step parts (include/difformer_hip.h,
        "row positions"); plan = sliced_plan(n_src, n_pos, F).  Two host syncs (status, block count): cold path, once
        per (graph, shard, F)."""
        dev = _require_device(rowptr, blkptr, src, order, parts)
        n_pos = int(n_rows) if n_pos is None else int(n_pos)
        slices, panels, G, PW, W, R, T, NT = (int(v) for v in plan)
        i32 = dict(dtype=torch.int32, device=dev)
        srt = torch.empty(max(int(nnz), 1), dtype=torch.int16, device=dev)
        counts = torch.empty(n_pos * NT * 32, dtype=torch.uint8, device=dev)
        lengths = torch.empty(G * NT * 4, **i32)
        table = torch.empty((R + 1) * panels * NT * W + 1, **i32)
        status = torch.empty(1, **i32)
        with _timed(self, "dif_sliced_measure", dev):
            rc = self.lib.dif_sliced_measure(_ptr(rowptr), _ptr(blkptr), _ptr(src), int(n_src), int(nnz), int(row_begin),
                                             int(n_rows), int(F), plan, _ptr(order), _ptr(parts), n_pos, _ptr(srt),
                                             _ptr(counts), _ptr(lengths), _ptr(table), _ptr(status), _stream(dev))
        _lib.check(rc, "dif_sliced_measure")
        bad, n_blocks = (int(v) for v in torch.stack([status[0], table[-1]]).tolist())
        if bad:
            return None
        entries = torch.empty(512 * max(n_blocks, 1), dtype=torch.int16, device=dev)
        with _timed(self, "dif_sliced_emit", dev):
            rc = self.lib.dif_sliced_emit(_ptr(rowptr), _ptr(blkptr), int(n_src), int(row_begin), int(n_rows), int(F), plan,
                                          _ptr(order), _ptr(parts), n_pos, _ptr(srt), _ptr(counts), _ptr(table),
                                          max(n_blocks, 1), _ptr(entries), _stream(dev))
        _lib.check(rc, "dif_sliced_emit")
        return entries, table

    def sliced_prescale(self, x, rowptr, n_src, plan, dinv=None):
        """x [n_src, F] fp32 -> ys [F/4, T*NT, 4]: rows scaled by deg^-1/2 (dinv, or from the row lengths), slice-major."""
        dev = _require_device(x, rowptr, dinv)
        _f32(x, "x")
        F = x.shape[1]
        x, ldx = _row_major(x, F)
        if ldx % 4 or x.data_ptr() % 16:
            x, ldx = x.contiguous(), F
        ys = torch.empty((F // 4, int(plan[6]) * int(plan[7]), 4), dtype=torch.float32, device=dev)
        with _timed(self, "dif_sliced_prescale_f32", dev):
            rc = self.lib.dif_sliced_prescale_f32(_ptr(x), ldx, _ptr(rowptr), _ptr(dinv), int(n_src), F, plan, _ptr(ys),
                                                  _stream(dev))
        _lib.check(rc, "dif_sliced_prescale_f32")
        return ys

    def sliced_spmm(self, sl, ys, rowptr, n_src, row_begin, n_rows, F, attn=None, attn_scale=1.0, gcn_scale=1.0, dinv=None):
        """sl: the format (ops.SlicedAdjacency: entries, table, plan, order, parts, n_pos) built for these rows."""
        dev = _require_device(sl.entries, sl.table, ys, rowptr, attn, sl.order, sl.parts, dinv)
        lda = 0
        if attn is not None:
            _f32(attn, "attn")
            attn, lda = _row_major(attn, F)
            if lda % 4 or attn.data_ptr() % 16:
                attn, lda = attn.contiguous(), F
        out = torch.empty((n_rows, F), dtype=torch.float32, device=dev)
        n_pos = int(n_rows) if sl.n_pos is None else int(sl.n_pos)
        ws_bytes = self.lib.dif_sliced_spmm_workspace_bytes(int(n_src), n_pos, int(F))   # > 0: a row shard (source splits)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
        with _timed(self, "dif_sliced_spmm_f32", dev):
            rc = self.lib.dif_sliced_spmm_f32(_ptr(sl.entries), _ptr(sl.table), sl.plan, _ptr(ys), _ptr(rowptr), _ptr(dinv),
                                              _ptr(sl.order), _ptr(sl.parts), n_pos, int(n_src), int(row_begin), int(n_rows),
                                              int(F), _ptr(attn), lda, float(attn_scale), float(gcn_scale), _ptr(out), F,
                                              _ptr(ws), ws_bytes, _stream(dev))
        _lib.check(rc, "dif_sliced_spmm_f32")
        return out

    def row_order(self, rowptr, row_begin, n_rows):
        """Rows [row_begin, row_begin + n_rows) by descending degree (indices inside the shard) ->
        (order int32 [n_rows], stats int32 [2] = {rows with degree > 4x mean, max degree}), both on the device."""
        dev = _require_device(rowptr)
        order = torch.empty(n_rows, dtype=torch.int32, device=dev)
        stats = torch.empty(2, dtype=torch.int32, device=dev)
        ws_bytes = self.lib.dif_row_order_workspace_bytes(n_rows)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _timed(self, "dif_row_order", dev):
            rc = self.lib.dif_row_order(_ptr(rowptr), row_begin, n_rows, _ptr(order), _ptr(stats), _ptr(ws), ws_bytes,
                                        _stream(dev))
        _lib.check(rc, "dif_row_order")
        return order, stats

    # ---- a5 ends: narrow Linear (+ LayerNorm + ReLU) -------------------------------------------
    def linear(self, x, weight, bias, ln_weight=None, ln_bias=None, eps=1e-5, relu=False):
        dev = _require_device(x, weight, bias, ln_weight, ln_bias)
        n, C = x.shape
        Co = weight.shape[0]
        dt, sfx = _storage(x, weight, bias, ln_weight, ln_bias)
        x, ldx = _row_major(x, C)
        if C > 128 and (ldx % 4 or x.data_ptr() % (4 * x.element_size())):      # long rows: 4-element aligned rows
            x, ldx = x.contiguous(), C
        weight, bias = weight.contiguous(), bias.contiguous()
        if ln_weight is not None:
            ln_weight, ln_bias = ln_weight.contiguous(), ln_bias.contiguous()
        out = torch.empty((n, Co), dtype=dt, device=dev)
        from . import ops
        if sfx == "f32" and Co > 64 and ops.linear_xwide_covers(x, weight):
            # wide rows into a wide layer: one product (C <= 416) or the two halves of the input channels as the two accumulating
            # products of the hidden-300 layer kernel; weights packed once per parameter version
            if ldx % 4 or x.data_ptr() % 16:
                x, ldx = x.contiguous(), C
            two = C > ops.XWIDE_MAX
            Ch = C // 2 if two else C
            pa = self.xwide_pack(weight, False, Ch, Co, cache=True)
            pb = self.xwide_pack(weight, False, Ch, Co, cache=True, col0=Ch) if two else None
            with _timed(self, "dif_linear_f32", dev):
                rc = self.lib.dif_linear_xwide_f32(_ptr(x), ldx, n, C, _ptr(pa), _ptr(pb), _ptr(bias), Co, _ptr(ln_weight), _ptr(ln_bias),
                                                   float(eps), int(bool(relu)), _ptr(out), Co, _stream(dev))
            _lib.check(rc, "dif_linear_xwide_f32")
            return out
        if (sfx == "f32" and C > 128 and Co <= 64 and C <= 8192 and not ops.EXACT_FP32 and
                (n < 16384 or C % 4 or ldx % 4 or x.data_ptr() % 16 or weight.data_ptr() % 16)):
            # few rows, or rows that are only 4-byte aligned (Cora: 2,708 x 1,433): K split over the waves of a workgroup,
            # the weights packed once per parameter version (bfloat16 hi / lo parts in MFMA fragment order)
            packed = self._packed_weight(weight, C, Co, dev)
            with _timed(self, "dif_linear_f32", dev):
                rc = self.lib.dif_linear_packed_f32(_ptr(x), ldx, n, C, _ptr(packed), _ptr(bias), Co, _ptr(ln_weight), _ptr(ln_bias),
                                                    float(eps), int(bool(relu)), _ptr(out), Co, _stream(dev))
            _lib.check(rc, "dif_linear_packed_f32")
            return out
        fn = getattr(self.lib, "dif_linear_" + sfx)
        with _timed(self, "dif_linear_f32", dev):
            rc = fn(_ptr(x), ldx, n, C, _ptr(weight), _ptr(bias), Co, _ptr(ln_weight), _ptr(ln_bias), float(eps),
                    int(bool(relu)), _ptr(out), Co, _stream(dev))
        _lib.check(rc, "dif_linear_" + sfx)
        return out

    def _packed_weight(self, weight, C, Co, dev):
        """dif_linear_pack_f32 of a [Co, C] float32 weight, cached per tensor (identity through a weak reference + data_ptr +
        version, like the CSR cache: a new tensor at a recycled address must not find the old packing): rebuilt after
        optimiser steps / load_state_dict (they bump the version); `.data` writes need model.invalidate_caches()."""
        import weakref
        from . import ops
        ver = ops.tensor_version(weight)
        key = (id(weight), weight.data_ptr(), ver, C, Co, str(dev))
        cache = self.__dict__.setdefault("_packed", {})
        hit = cache.get(key)
        if hit is not None and ver >= 0 and hit[0]() is weight:
            return self._pin(hit[1])
        packed = torch.empty(self.lib.dif_linear_packed_bytes(C), dtype=torch.uint8, device=dev)
        with _timed(self, "dif_linear_pack_f32", dev):
            rc = self.lib.dif_linear_pack_f32(_ptr(weight), C, Co, _ptr(packed), _stream(dev))
        _lib.check(rc, "dif_linear_pack_f32")
        if ver >= 0:
            self._packed_insert(cache, key, weight, packed)
        return self._pin(packed)

    @staticmethod
    def _packed_insert(cache, key, tensor, packed):
        """Insert into a packed-weight cache.  Evicted: entries of freed tensors and OLDER VERSIONS of this tensor (an optimiser
        step per epoch would otherwise add an entry per epoch); beyond 64 live entries the oldest.  Never everything at once:
        a captured hipGraph bakes raw pointers to these buffers in (it pins the ones it used itself, `capture_pins`)."""
        import weakref
        for k in [k for k, v in cache.items() if v[0]() is None or (v[0]() is tensor and k[:2] == key[:2] and k[3:] == key[3:])]:
            del cache[k]
        while len(cache) >= 64:
            del cache[next(iter(cache))]
        cache[key] = (weakref.ref(tensor), packed)

    def _pin(self, packed):
        """While a forward is being captured, the capture keeps every packed buffer it was handed alive (DIFFormer.
        _forward_graphed / graphs.GraphedForward set `capture_pins` to a list around the capture)."""
        pins = getattr(self, "capture_pins", None)
        if pins is not None:
            pins.append(packed)
        return packed

    # ---- a4 / a5 tail ----------------------------------------------------------------------
    def layer_tail_bwd(self, conv, x0, prev, alpha, ln_weight, ln_bias, eps, relu, grad_out, want):
        """Backward of layer_tail in one pass (csrc/layer_tail_bwd.hip).  want = (conv, x0, prev) booleans.
        -> (d_conv [n,H,D] | None, d_x0 | None, d_prev | None, d_ln_weight | None, d_ln_bias | None); None when the
        shape is not covered (D % 4 != 0, D > 512, bf16 storage): the caller re-derives the gradient with tensor ops."""
        dev = _require_device(conv, x0, prev, ln_weight, ln_bias, grad_out)
        n, H, D = conv.shape
        if D % 4 or D > 512 or any(t is not None and t.dtype != torch.float32 for t in (conv, x0, prev, ln_weight, grad_out)):
            return None
        conv, ldc = _row_major(conv, H * D)
        g, ldg = _row_major(grad_out, D)
        ldx0 = ldp = 0
        if x0 is not None:
            x0, ldx0 = _row_major(x0, D)
        if prev is not None:
            prev, ldp = _row_major(prev, D)
        if any(ld % 4 for ld in (ldc, ldg, ldx0, ldp)) or any(t is not None and t.data_ptr() % 16 for t in (conv, g, x0, prev)):
            return None
        f32 = dict(dtype=torch.float32, device=dev)
        d_conv = torch.empty((n, H, D), **f32) if want[0] else None
        d_x0 = torch.empty((n, D), **f32) if (want[1] and x0 is not None) else None
        d_prev = torch.empty((n, D), **f32) if (want[2] and prev is not None) else None
        d_ln = ws = None
        ws_bytes = 0
        if ln_weight is not None:
            ln_weight, ln_bias = ln_weight.contiguous(), ln_bias.contiguous()
            d_ln = torch.empty(2 * D, **f32)
            ws_bytes = self.lib.dif_layer_tail_bwd_workspace_bytes(n, D)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _timed(self, "dif_layer_tail_bwd_f32", dev):
            rc = self.lib.dif_layer_tail_bwd_f32(_ptr(conv), ldc, n, H, D, _ptr(x0), ldx0, _ptr(prev), ldp, float(alpha),
                                                 _ptr(ln_weight), _ptr(ln_bias), float(eps), int(bool(relu)), _ptr(g), ldg,
                                                 _ptr(d_conv), H * D, _ptr(d_x0), D, _ptr(d_prev), D, _ptr(d_ln), _ptr(ws),
                                                 ws_bytes, _stream(dev))
        _lib.check(rc, "dif_layer_tail_bwd_f32")
        return (d_conv, d_x0, d_prev, None if d_ln is None else d_ln[:D], None if d_ln is None else d_ln[D:2 * D])

    def coeffs_bg(self, x, record, n_global, factors, C, D, attn_scale):
        """Coefficients of the closed-form layer through the background kernels (csrc/side_chain.hip), enqueued on the
        CURRENT stream (the caller puts a side stream there): from x [n, C] (one pass: Gram partials per wave), or from a
        finished `record` [X^T X | sum x] when the layer input's Gram pass has already run.  factors:
        ops.NarrowFactors (pt, vtt, st float32 [80 * 80]).  -> coef (layout of simple_coeffs)."""
        dev = _require_device(x, record, factors.pt)
        f32 = dict(dtype=torch.float32, device=dev)
        gt = torch.empty(80 * 80 + 400, **f32)           # G~ + 100 float64 pairs of partial norm products
        if record is not None:
            ws, ws_bytes, xp, ldx, n = record, record.numel() * 4, None, 0, 1
        else:
            _f32(x, "x")
            n = x.shape[0]
            x, ldx = _row_major(x, C)
            ws_bytes = self.lib.dif_gram_bg_workspace_bytes(n, C)
            ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
            xp = x
        with _timed(self, "dif_gram_bg_f32", dev):
            rc = self.lib.dif_gram_bg_f32(_ptr(xp), ldx, n, C, int(n_global), _ptr(factors.st), _ptr(gt), _ptr(ws), ws_bytes,
                                          _stream(dev))
        _lib.check(rc, "dif_gram_bg_f32")
        scratch = torch.empty(80 * 80 + 4, **f32)
        coef = torch.empty(self.lib.dif_simple_coeffs_len(C, D), **f32)
        with _timed(self, "dif_simple_coeffs_bg_f32", dev):
            rc = self.lib.dif_simple_coeffs_bg_f32(_ptr(gt), _ptr(factors.pt), _ptr(factors.vtt), _ptr(factors.st), C, D,
                                                   float(attn_scale), _ptr(scratch), _ptr(coef), _stream(dev))
        _lib.check(rc, "dif_simple_coeffs_bg_f32")
        return coef

    def gram_sym(self, x):
        """x [n, C] fp32 -> record [C*C + 2*C + 2]: X^T X (blocks of 64 on and above the diagonal valid) | sum x | unused
        (csrc/simple_attn.hip, dif_gram_sym_f32) -- the Gram record of the closed form at the scripts' widths."""
        dev = _require_device(x)
        _f32(x, "x")
        n, C = x.shape
        x, ldx = _row_major(x, C)
        rec = torch.empty(self.lib.dif_simple_reduced_len(1, C, C), dtype=torch.float32, device=dev)
        from . import ops
        if 64 < C <= 128 and C % 4 == 0 and ldx % 4 == 0 and x.data_ptr() % 16 == 0 and (ops.EXACT_FP32 or n < 4096):
            # hidden 128: one pass over x with the whole upper half of X^T X in a wave's registers (csrc/simple_layer_wide.hip)
            ws_bytes = self.lib.dif_gram128_workspace_bytes(n, C)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            with _timed(self, "dif_gram_sym_f32", dev):
                rc = self.lib.dif_gram128_f32(_ptr(x), ldx, n, C, _ptr(rec), _ptr(ws), ws_bytes, _stream(dev))
            _lib.check(rc, "dif_gram128_f32")
            return rec
        ws_bytes = self.lib.dif_gram_sym_workspace_bytes(n, C)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _timed(self, "dif_gram_sym_f32", dev):
            rc = self.lib.dif_gram_sym_f32(_ptr(x), ldx, n, C, _ptr(rec), _ptr(ws), ws_bytes, _stream(dev))
        _lib.check(rc, "dif_gram_sym_f32")
        return rec

    def wide_gram(self, rec, C, n_global, S):
        """Record of gram_sym -> (G~ float64 [(C+1), (C+1)], partial sums of the two norm products) -- dif_wide_gram_f64."""
        dev = _require_device(rec, S)
        Gt = torch.empty((C + 1, C + 1), dtype=torch.float64, device=dev)
        partial = torch.empty(2 * self.lib.dif_wide_partials(C), dtype=torch.float64, device=dev)
        with _timed(self, "dif_wide_gram_f64", dev):
            rc = self.lib.dif_wide_gram_f64(_ptr(rec), C, int(n_global), _ptr(S), _ptr(Gt), _ptr(partial), _stream(dev))
        _lib.check(rc, "dif_wide_gram_f64")
        return Gt, partial

    def wide_coeffs(self, rec, C, n_global, S, V, P):
        """Record of gram_sym -> (B float32 [C, DV], bias float32 [DV]), the row GEMM's operands [Mn | u], [cn | cd]: both
        float64 products and their bookkeeping in two launches (dif_wide_coeffs_f64; no library GEMM)."""
        dev = _require_device(rec, S, V, P)
        DV = V.shape[1]
        T = torch.empty((C + 1, DV), dtype=torch.float64, device=dev)
        partial = torch.empty(2 * ((C + 16) // 16), dtype=torch.float64, device=dev)
        B = torch.empty((C, DV), dtype=torch.float32, device=dev)
        bias = torch.empty(DV, dtype=torch.float32, device=dev)
        with _timed(self, "dif_wide_coeffs_f64", dev):
            rc = self.lib.dif_wide_coeffs_f64(_ptr(rec), C, int(n_global), _ptr(S), _ptr(V), _ptr(P), DV, _ptr(T), _ptr(partial),
                                              _ptr(B), _ptr(bias), _stream(dev))
        _lib.check(rc, "dif_wide_coeffs_f64")
        return B, bias

    def wide_scale(self, R, T, partial, C):
        """R, T float64 [(C+1), DV] -> (B float32 [C, DV], bias float32 [DV]) = (s R[:C], s R[C] + T[C]) -- dif_wide_scale_f64."""
        dev = _require_device(R, T, partial)
        DV = R.shape[1]
        R, T = R.contiguous(), T.contiguous()
        B = torch.empty((C, DV), dtype=torch.float32, device=dev)
        bias = torch.empty(DV, dtype=torch.float32, device=dev)
        with _timed(self, "dif_wide_scale_f64", dev):
            rc = self.lib.dif_wide_scale_f64(_ptr(R), _ptr(T), _ptr(partial), C, DV, _ptr(B), _ptr(bias), _stream(dev))
        _lib.check(rc, "dif_wide_scale_f64")
        return B, bias

    def layer_tail_mix(self, Z, D, den_col, conv_scale, add, add_scale, rs, bv, x0, prev, alpha, ln_weight, ln_bias, eps,
                       relu=False):
        """Tail of the closed form at the scripts' widths (dif_layer_tail_mix_f32): Z [n, >= D + 1] fp32 holds the
        numerator in columns [0, D) and the denominator in column den_col; add [n, D] / rs [n] / bv [D] optional."""
        dev = _require_device(Z, add, rs, bv, x0, prev, ln_weight, ln_bias)
        n, ldz = Z.shape
        for t_, nm in ((Z, "Z"), (add, "add"), (rs, "rs"), (bv, "bv"), (x0, "x0"), (prev, "prev")):
            if t_ is not None:
                _f32(t_, nm)
        if not Z.is_contiguous() or ldz % 4:
            raise ValueError("difformer_amd: layer_tail_mix needs a contiguous Z with a row length that is a multiple of 4")
        lda = ldx0 = ldp = 0
        if add is not None:
            add, lda = _row_major(add, D)
        if x0 is not None:
            x0, ldx0 = _row_major(x0, D)
        if prev is not None:
            prev, ldp = _row_major(prev, D)
        if ln_weight is not None:
            ln_weight, ln_bias = ln_weight.contiguous(), ln_bias.contiguous()
        if rs is not None:
            rs, bv = rs.contiguous(), bv.contiguous()
        out = torch.empty((n, D), dtype=torch.float32, device=dev)
        den_ptr = None if den_col is None else Z.data_ptr() + 4 * int(den_col)
        with _timed(self, "dif_layer_tail_mix_f32", dev):
            rc = self.lib.dif_layer_tail_mix_f32(_ptr(Z), ldz, den_ptr, ldz, float(conv_scale), _ptr(add), lda,
                                                 float(add_scale), _ptr(rs), _ptr(bv), n, D, _ptr(x0), ldx0, _ptr(prev), ldp,
                                                 float(alpha), _ptr(ln_weight), _ptr(ln_bias), float(eps), int(bool(relu)),
                                                 _ptr(out), D, _stream(dev))
        _lib.check(rc, "dif_layer_tail_mix_f32")
        return out

    def simple_layer_wide(self, x, B, bias, D, attn_scale, ax, Wv, bv, rs, gcn_scale, x0, residual, alpha, ln_weight, ln_bias,
                          eps, relu=False):
        """Closed-form `simple` layer for 64 < max(C, D) <= 128 in one pass (csrc/simple_layer_wide.hip): x [n, C], B [C, dv]
        = [Mn | u | ...], bias [dv] = [cn | cd | ...] (wide_scale), ax = A_hat x [n, C] or None, Wv [D, C] / bv [D] / rs [n]."""
        dev = _require_device(x, B, bias, ax, Wv, bv, rs, x0, ln_weight, ln_bias)
        n, C = x.shape
        for t_, nm in ((x, "x"), (B, "B"), (bias, "bias"), (ax, "ax"), (Wv, "Wv"), (x0, "x0")):
            if t_ is not None:
                _f32(t_, nm)
        x, ldx = _row_major(x, C)
        if ldx % 4 or x.data_ptr() % 16:
            x, ldx = x.contiguous(), C
        B, bias = B.contiguous(), bias.contiguous()
        ldax = ldx0 = 0
        if ax is not None:
            ax, ldax = _row_major(ax, C)
            if ldax % 4 or ax.data_ptr() % 16:
                ax, ldax = ax.contiguous(), C
        if x0 is not None:
            x0, ldx0 = _row_major(x0, D)
            if ldx0 % 4 or x0.data_ptr() % 16:
                x0, ldx0 = x0.contiguous(), D
        if Wv is not None:
            Wv, bv = Wv.contiguous(), bv.contiguous()
        if rs is not None:
            rs = rs.contiguous()
        if ln_weight is not None:
            ln_weight, ln_bias = ln_weight.contiguous(), ln_bias.contiguous()
        out = torch.empty((n, D), dtype=torch.float32, device=dev)
        with _timed(self, "dif_simple_layer_f32", dev):
            rc = self.lib.dif_simple_layer_wide_f32(_ptr(x), ldx, n, C, D, _ptr(B), int(B.shape[1]), _ptr(bias), float(attn_scale),
                                                    _ptr(ax), ldax, _ptr(Wv), _ptr(bv), _ptr(rs), float(gcn_scale), _ptr(x0), ldx0,
                                                    int(bool(residual)), float(alpha), _ptr(ln_weight), _ptr(ln_bias), float(eps),
                                                    int(bool(relu)), _ptr(out), D, _stream(dev))
        _lib.check(rc, "dif_simple_layer_wide_f32")
        return out

    def xwide_pack(self, src, transposed, C, D, cache=False, col0=0):
        """dif_xwide_pack_f32: src [C, ld] (transposed: the [Mn | u] operand) or [D, ld] (an nn.Linear weight; col0: the C
        input channels start at that column) -> packed MFMA fragments.  cache=True keeps the packing per weight tensor
        (identity + version, as _packed_weight)."""
        import weakref
        from . import ops
        dev = src.device
        key = None
        if cache:
            ver = ops.tensor_version(src)
            key = (id(src), src.data_ptr(), ver, C, D, bool(transposed), str(dev), col0)
            store = self.__dict__.setdefault("_packed", {})
            hit = store.get(key)
            if hit is not None and ver >= 0 and hit[0]() is src:
                return self._pin(hit[1])
        src_c = src.contiguous()
        packed = torch.empty(self.lib.dif_xwide_packed_bytes(C, D), dtype=torch.uint8, device=dev)
        with _timed(self, "dif_xwide_pack_f32", dev):
            rc = self.lib.dif_xwide_pack_f32(src_c.data_ptr() + 4 * col0, int(src_c.shape[1]), int(bool(transposed)), C, D, _ptr(packed),
                                             _stream(dev))
        _lib.check(rc, "dif_xwide_pack_f32")
        if cache and key[2] >= 0:
            self._packed_insert(store, key, src, packed)
        return self._pin(packed)

    def simple_layer_xwide(self, x, B, bias, D, attn_scale, ax, Wv, bv, rs, gcn_scale, x0, residual, alpha, ln_weight, ln_bias,
                           eps, relu=False):
        """Closed-form `simple` layer for 128 < max(C, D) <= 416 in one pass (csrc/simple_layer_xwide.hip); arguments as
        simple_layer_wide.  Mn is packed per call (it changes with the Gram record), Wv once per parameter version."""
        dev = _require_device(x, B, bias, ax, Wv, bv, rs, x0, ln_weight, ln_bias)
        n, C = x.shape
        for t_, nm in ((x, "x"), (B, "B"), (bias, "bias"), (ax, "ax"), (Wv, "Wv"), (x0, "x0")):
            if t_ is not None:
                _f32(t_, nm)
        x, ldx = _row_major(x, C)
        if ldx % 4 or x.data_ptr() % 16:
            x, ldx = x.contiguous(), C
        B, bias = B.contiguous(), bias.contiguous()
        ldax = ldx0 = 0
        if ax is not None:
            ax, ldax = _row_major(ax, C)
            if ldax % 4 or ax.data_ptr() % 16:
                ax, ldax = ax.contiguous(), C
        if x0 is not None:
            x0, ldx0 = _row_major(x0, D)
            if ldx0 % 4 or x0.data_ptr() % 16:
                x0, ldx0 = x0.contiguous(), D
        pm = self.xwide_pack(B, True, C, D)
        pv = None
        if Wv is not None:
            pv, bv = self.xwide_pack(Wv, False, C, D, cache=True), bv.contiguous()
        if rs is not None:
            rs = rs.contiguous()
        if ln_weight is not None:
            ln_weight, ln_bias = ln_weight.contiguous(), ln_bias.contiguous()
        out = torch.empty((n, D), dtype=torch.float32, device=dev)
        with _timed(self, "dif_simple_layer_f32", dev):
            rc = self.lib.dif_simple_layer_xwide_f32(_ptr(x), ldx, n, C, D, _ptr(pm), _ptr(pv), _ptr(B), int(B.shape[1]), _ptr(bias),
                                                     float(attn_scale), _ptr(ax), ldax, _ptr(bv), _ptr(rs), float(gcn_scale), _ptr(x0),
                                                     ldx0, int(bool(residual)), float(alpha), _ptr(ln_weight), _ptr(ln_bias), float(eps),
                                                     int(bool(relu)), _ptr(out), D, _stream(dev))
        _lib.check(rc, "dif_simple_layer_xwide_f32")
        return out

    def layer_tail(self, conv, x0, prev, alpha, ln_weight, ln_bias, eps, relu=False):
        dev = _require_device(conv, x0, prev, ln_weight, ln_bias)
        n, H, D = conv.shape
        dt, sfx = _storage(conv, x0, prev, ln_weight, ln_bias)
        conv, ldc = _row_major(conv, H * D)
        ldx0 = ldp = 0
        if x0 is not None:
            x0, ldx0 = _row_major(x0, D)
        if prev is not None:
            prev, ldp = _row_major(prev, D)
        if ln_weight is not None:
            ln_weight, ln_bias = ln_weight.contiguous(), ln_bias.contiguous()
        out = torch.empty((n, D), dtype=dt, device=dev)
        fn = getattr(self.lib, "dif_layer_tail_" + sfx)
        with _timed(self, "dif_layer_tail_f32", dev):
            rc = fn(_ptr(conv), ldc, n, H, D, _ptr(x0), ldx0, _ptr(prev), ldp, float(alpha), _ptr(ln_weight),
                    _ptr(ln_bias), float(eps), int(bool(relu)), _ptr(out), D, _stream(dev))
        _lib.check(rc, "dif_layer_tail_" + sfx)
        return out

"""CPU oracle for the DIFFormer propagation layer -- TEST INFRASTRUCTURE ONLY.

This module is a numpy restatement of the reference algorithm
(`/root/reference/node classification/difformer.py`).  It exists to *check*
the HIP path; it is never the thing shipped or measured.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it.  Nothing under `difformer_amd/` imports it.

Parity pinning: the reference ships no golden vectors / KATs for this path
(SURVEY.md section 8c).  The oracle is therefore pinned against outputs of the
reference *itself*, generated in the build container by importing the
reference source verbatim (`tests/golden/make_golden.py` for
`node classification/difformer.py`, `tests/golden/make_golden_v2.py` for
`physical particle/difformer-v2.py`, `tests/golden/make_golden_st.py` / `make_golden_it.py` for the `spatial-temporal` and
`image and text` copies; fixtures under `tests/golden/*.npz`).  `tests/test_oracle_golden.py` checks every function
here against those fixtures.

Every function cites the reference lines it restates.  All arithmetic runs in
the dtype of the inputs (float32 mirrors the reference CPU path, float64 is
the high-precision ground truth the parity metric of SURVEY.md section 8d is
taken against).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CLIB = None


def _load_clib():
    """Optional C restatement of gcn_conv (oracle/gcn_conv_ref.c) for large graphs."""
    global _CLIB
    if _CLIB is None:
        path = os.path.join(_HERE, "_build", "liboracle_gcn.so")
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            for name in ("oracle_gcn_conv_f32", "oracle_gcn_conv_f64"):
                fn = getattr(lib, name)
                fn.restype = ctypes.c_int
                fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                               ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                               ctypes.c_void_p, ctypes.c_int]
            _CLIB = lib
        else:
            _CLIB = False
    return _CLIB


# --------------------------------------------------------------------------
# a1: full_attention_conv, kernel == 'simple'   (difformer.py:18-39)
# --------------------------------------------------------------------------
def simple_attention(qs, ks, vs):
    """qs [N,H,M], ks [L,H,M], vs [L,H,D] -> [N,H,D].

    difformer.py:20-21  global Frobenius normalisation of the WHOLE q / k tensors
    difformer.py:25-26  numerator  q . (K^T V)
    difformer.py:27-29  + sum_l v_l, broadcast over queries (needs N == L)
    difformer.py:32-34  normaliser q . (sum_l k_l)
    difformer.py:37-39  + N, divide
    """
    dt = qs.dtype
    if qs.shape[0] != vs.shape[0]:
        # difformer.py:29 adds a [L,H,D] tensor to a [N,H,D] one
        raise ValueError("simple kernel requires N == L (difformer.py:29)")
    qn = qs / np.sqrt(np.sum(qs * qs, dtype=dt)).astype(dt)
    kn = ks / np.sqrt(np.sum(ks * ks, dtype=dt)).astype(dt)
    n_query = dt.type(qs.shape[0])
    ktv = np.einsum("lhm,lhd->hmd", kn, vs)              # :25
    num = np.einsum("nhm,hmd->nhd", qn, ktv)             # :26
    num = num + vs.sum(axis=0, dtype=dt)[None]           # :27-29
    ksum = kn.sum(axis=0, dtype=dt)                      # :32-33
    den = np.einsum("nhm,hm->nh", qn, ksum)[..., None]   # :34,:37
    den = den + n_query                                  # :38
    return (num / den).astype(dt)                        # :39


def simple_attention_weights(qs, ks):
    """output_attn branch, difformer.py:42-43 (shape-valid only for H == 1)."""
    dt = qs.dtype
    qn = qs / np.sqrt(np.sum(qs * qs, dtype=dt)).astype(dt)
    kn = ks / np.sqrt(np.sum(ks * ks, dtype=dt)).astype(dt)
    den = np.einsum("nhm,hm->nh", qn, kn.sum(axis=0, dtype=dt))[..., None] + dt.type(qs.shape[0])
    return (np.einsum("nhm,lhm->nlh", qn, kn) / den).astype(dt)


# --------------------------------------------------------------------------
# a2: full_attention_conv, kernel == 'sigmoid'  (difformer.py:45-56)
# --------------------------------------------------------------------------
def sigmoid_attention(qs, ks, vs, return_weights=False):
    """qs [N,H,M], ks [L,H,M], vs [L,H,D] -> [N,H,D]; N may differ from L.

    difformer.py:47     S = sigmoid(q . k)           [N,L,H]
    difformer.py:50-52  row sums over l
    difformer.py:55-56  (S / rowsum) . V
    """
    dt = qs.dtype
    s = np.einsum("nhm,lhm->nlh", qs, ks)
    s = (1.0 / (1.0 + np.exp(-s))).astype(dt)
    den = s.sum(axis=1, dtype=dt)[:, None, :]
    att = (s / den).astype(dt)
    out = np.einsum("nlh,lhd->nhd", att, vs).astype(dt)
    return (out, att) if return_weights else out


def sigmoid_attention_blocked(qs, ks, vs, block=2048, return_den=False):
    """difformer.py:45-56 for sizes whose [N,L,H] score tensor does not fit (N = L = 15,000 at 300 columns: the
    image-and-text scripts): the same lines over blocks of `block` queries, one BLAS product per head and block.
    Identical to sigmoid_attention up to summation order (tests/test_oracle_it_golden.py holds the two together)."""
    dt = qs.dtype
    n, h, _ = qs.shape
    out = np.empty((n, h, vs.shape[2]), dtype=dt)
    dens = np.empty((n, h), dtype=dt)
    for hh in range(h):
        kt = np.ascontiguousarray(ks[:, hh, :].T)
        vh = np.ascontiguousarray(vs[:, hh, :])
        for r0 in range(0, n, block):
            s = qs[r0:r0 + block, hh, :] @ kt                             # :47 q . k
            s = (1.0 / (1.0 + np.exp(-s))).astype(dt)                     # :47 sigmoid
            den = s.sum(axis=1, dtype=dt)                                 # :50-52
            dens[r0:r0 + block, hh] = den
            out[r0:r0 + block, hh, :] = ((s / den[:, None]).astype(dt) @ vh).astype(dt)       # :55-56
    return (out, dens) if return_den else out


def sigmoid_attention_grad_blocked(qs, ks, vs, g, block=2048):
    """(dq, dk, dv) of difformer.py:45-56 under the cotangent g = dL/dout, i.e. what `loss.backward()` (main.py:113 of the
    image-and-text folder) derives from those lines, written out and evaluated over blocks of queries:
        P = sigmoid(q . k) (:47), den = sum_l P (:50-52), A = P / den (:55), out = A v (:56)
        dv = A^T g,  dA = g v^T,  dP = (dA - sum_l dA A) / den,  dS = dP P (1 - P),  dq = dS k,  dk = dS^T q
    Pinned to autograd of the reference itself through tests/golden/golden_it.npz and golden_grad.npz."""
    dt = qs.dtype
    n, h, _ = qs.shape
    dq, dk, dv = np.zeros_like(qs), np.zeros_like(ks), np.zeros_like(vs)
    for hh in range(h):
        kh, vh = np.ascontiguousarray(ks[:, hh, :]), np.ascontiguousarray(vs[:, hh, :])
        for r0 in range(0, n, block):
            qb, gb = qs[r0:r0 + block, hh, :], g[r0:r0 + block, hh, :]
            p = (1.0 / (1.0 + np.exp(-(qb @ kh.T)))).astype(dt)
            den = p.sum(axis=1, dtype=dt)[:, None]
            a = p / den
            dv[:, hh, :] += a.T @ gb
            da = gb @ vh.T
            dp = (da - (da * a).sum(axis=1, dtype=dt)[:, None]) / den
            ds = dp * p * (1.0 - p)
            dq[r0:r0 + block, hh, :] = ds @ kh
            dk[:, hh, :] += ds.T @ qb
    return dq, dk, dv


def full_attention_conv(qs, ks, vs, kernel, output_attn=False):
    """Dispatch mirroring difformer.py:10-61."""
    if kernel == "simple":
        out = simple_attention(qs, ks, vs)
        return (out, simple_attention_weights(qs, ks)) if output_attn else out
    if kernel == "sigmoid":
        return sigmoid_attention(qs, ks, vs, return_weights=output_attn)
    raise ValueError(f"unknown kernel {kernel!r}")


# --------------------------------------------------------------------------
# a3: gcn_conv  (difformer.py:63-79; torch_sparse 0.6.10 SparseTensor+matmul and
#     torch_geometric.utils.degree are un-vendored dependencies: their published
#     semantics are restated here -- degree = bincount, matmul(adj, x) = sum-SpMM)
# --------------------------------------------------------------------------
def gcn_edge_values(edge_index, num_nodes, edge_weight=None, dtype=np.float32):
    """Per-edge value of the normalised adjacency, difformer.py:65-74.

    d = in-degree counted over `col` only (:66); value_e = w_e * d[col]^-1/2 *
    d[row]^-1/2 (:67-73); non-finite -> 0 (:74; a zero-degree source gives inf).
    The degree vector and both d^-1/2 factors are float32 in the reference
    (`.float()`, :66) whatever the feature dtype, and so they are here.
    """
    row = np.asarray(edge_index[0], dtype=np.int64)
    col = np.asarray(edge_index[1], dtype=np.int64)
    f32 = np.float32
    deg = np.bincount(col, minlength=num_nodes).astype(f32)       # `.float()`, :66
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        dn_in = np.sqrt(f32(1.0) / deg[col])                       # float32 whatever x is
        dn_out = np.sqrt(f32(1.0) / deg[row])
        if edge_weight is None:
            val = (np.ones(row.shape[0], dtype=f32) * dn_in * dn_out).astype(dtype)   # :71
        else:
            val = np.asarray(edge_weight).astype(dtype) * dn_in.astype(dtype) * dn_out.astype(dtype)  # :73
    val = np.where(np.isfinite(val), val, dtype(0.0)).astype(dtype)
    return row, col, val


def gcn_conv(x, edge_index, edge_weight=None):
    """x [N,H,D] -> [N,H,D]:  out[col_e] += value_e * x[row_e]  for every edge.

    difformer.py:75 builds SparseTensor(row=col, col=row, value) i.e. the edge
    (row -> col) lands on output row `col`; :76-78 apply the same adjacency to
    every head.  Duplicate edges accumulate.
    """
    dt = x.dtype
    n, h, d = x.shape
    e = int(np.asarray(edge_index).shape[1])
    lib = _load_clib()
    if lib and e >= 100000 and dt in (np.float32, np.float64):
        ei = np.ascontiguousarray(edge_index, dtype=np.int64)
        xc = np.ascontiguousarray(x).reshape(n, h * d)
        out = np.zeros_like(xc)
        w = None if edge_weight is None else np.ascontiguousarray(edge_weight, dtype=dt)
        fn = lib.oracle_gcn_conv_f32 if dt == np.float32 else lib.oracle_gcn_conv_f64
        rc = fn(xc.ctypes.data, ei.ctypes.data, None if w is None else w.ctypes.data,
                n, e, h * d, out.ctypes.data, int(os.environ.get("ORACLE_THREADS", os.cpu_count() or 1)))
        if rc != 0:
            raise RuntimeError(f"oracle_gcn_conv failed rc={rc}")
        return out.reshape(n, h, d)
    row, col, val = gcn_edge_values(edge_index, n, edge_weight, dtype=dt.type)
    out = np.zeros((n, h * d), dtype=dt)
    np.add.at(out, col, val[:, None] * x.reshape(n, h * d)[row])
    return out.reshape(n, h, d)


# --------------------------------------------------------------------------
# a4: DIFFormerConv.forward  (difformer.py:113-145)
# --------------------------------------------------------------------------
def linear(x, weight, bias):
    """nn.Linear: x @ W^T + b."""
    return (x @ weight.T + bias).astype(x.dtype)


def layer_norm(x, weight, bias, eps=1e-5):
    """nn.LayerNorm over the last dim (biased variance, eps inside the sqrt)."""
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return ((x - mu) / np.sqrt(var + x.dtype.type(eps)) * weight + bias).astype(x.dtype)


def difformer_conv(p, prefix, query_input, source_input, edge_index, edge_weight, x_0, cfg):
    """One propagation layer.  `p` maps state_dict keys -> numpy arrays.

    difformer.py:115-120 projections (use_weight=False: value = source reshaped [N,1,D])
    difformer.py:126     attention
    difformer.py:129-136 + gcn_conv, or convex mix when graph_weight > 0
    difformer.py:137     mean over heads
    difformer.py:139-140 += x_0 when use_source
    """
    h, d = cfg["num_heads"], cfg["hidden_channels"]
    q = linear(query_input, p[prefix + "Wq.weight"], p[prefix + "Wq.bias"]).reshape(-1, h, d)
    k = linear(source_input, p[prefix + "Wk.weight"], p[prefix + "Wk.bias"]).reshape(-1, h, d)
    if cfg.get("use_weight", True):
        v = linear(source_input, p[prefix + "Wv.weight"], p[prefix + "Wv.bias"]).reshape(-1, h, d)
    else:
        v = source_input.reshape(-1, 1, d)
    att = full_attention_conv(q, k, v, cfg.get("kernel", "simple"))
    if cfg.get("use_graph", True):
        g = gcn_conv(v, edge_index, edge_weight)
        gw = cfg.get("graph_weight", -1)
        dt = att.dtype.type
        out = (dt(1 - gw) * att + dt(gw) * g) if gw > 0 else (att + g)
    else:
        out = att
    out = out.mean(axis=1).astype(query_input.dtype)
    if cfg.get("use_source", False):
        out = out + x_0
    return out


# --------------------------------------------------------------------------
# a5: DIFFormer.forward (eval mode: dropout is identity)  (difformer.py:184-209)
# --------------------------------------------------------------------------
def difformer_forward(p, x, edge_index, edge_weight, cfg, return_layers=False):
    """cfg keys: hidden_channels, num_layers, num_heads, kernel, alpha, use_bn,
    use_residual, use_weight, use_graph, graph_weight, use_source."""
    dt = x.dtype.type
    alpha = dt(cfg.get("alpha", 0.5))
    h = linear(x, p["fcs.0.weight"], p["fcs.0.bias"])                  # :188
    if cfg.get("use_bn", True):
        h = layer_norm(h, p["bns.0.weight"], p["bns.0.bias"])          # :189-190
    h = np.maximum(h, dt(0))                                           # :191
    layers = [h]                                                       # :195
    for i in range(cfg["num_layers"]):
        h = difformer_conv(p, f"convs.{i}.", h, h, edge_index, edge_weight, layers[0], cfg)  # :199
        if cfg.get("use_residual", True):
            h = alpha * h + (dt(1) - alpha) * layers[i]                # :200-201
        if cfg.get("use_bn", True):
            h = layer_norm(h, p[f"bns.{i + 1}.weight"], p[f"bns.{i + 1}.bias"])  # :202-203
        layers.append(h)                                               # :205
    out = linear(h, p["fcs.1.weight"], p["fcs.1.bias"])                # :208
    return (out, layers) if return_layers else out


def cast_params(p, dtype):
    return {k: np.asarray(v).astype(dtype) for k, v in p.items()}


# --------------------------------------------------------------------------
# f2: induced subgraph with relabelling (caller side of the mini-batch path)
#     node classification/main-batch.py:131  subgraph(idx_i, edge_index, num_nodes=n, relabel_nodes=True)
#     torch_geometric 1.7.2 (un-vendored): node_mask[subset] = True; edge_mask = mask[row] & mask[col];
#     kept edges stay in their original order; node_idx[subset] = arange(len(subset)) relabels them.
# --------------------------------------------------------------------------
def subgraph(subset, edge_index, edge_attr=None, relabel_nodes=False, num_nodes=None):
    subset = np.asarray(subset, dtype=np.int64)
    edge_index = np.asarray(edge_index, dtype=np.int64)
    n = int(num_nodes) if num_nodes is not None else int(edge_index.max()) + 1
    newid = np.full(n, -1, dtype=np.int64)
    newid[subset] = np.arange(subset.shape[0])
    keep = (newid[edge_index[0]] >= 0) & (newid[edge_index[1]] >= 0)
    out = edge_index[:, keep]
    if relabel_nodes:
        out = newid[out]
    return out, (None if edge_attr is None else np.asarray(edge_attr)[keep])


# --------------------------------------------------------------------------
# f4: DIFFormer_v2 (physical particle/difformer-v2.py) - a batch of B independent graphs whose nodes are
#     stored back to back; n_nodes[b] = node count of graph b.  The reference pads to [B, max_node, H, D]
#     (:8-27); here the same sums are restated per graph / per position without padding.
# --------------------------------------------------------------------------
def v2_simple_attention(qs, ks, vs, n_nodes):
    """TransConv.full_attention, kernel 'simple' (difformer-v2.py:80-111).

    :82-83  q, k divided by the Frobenius norm over the WHOLE batch (all graphs together)
    :93     per-graph K^T V                      :95-98   per-graph sum of v (UNscaled values)
    :100-101 numerator = q.KtV_b + vsum_b        :103-109 denominator = q.ksum_b + n_b
    """
    qs, ks, vs = (np.asarray(a) for a in (qs, ks, vs))
    n_nodes = np.asarray(n_nodes, dtype=np.int64)
    dt = qs.dtype.type
    qn = qs / np.sqrt((qs * qs).sum(dtype=qs.dtype))
    kn = ks / np.sqrt((ks * ks).sum(dtype=ks.dtype))
    out = np.empty(vs.shape, dtype=vs.dtype)
    off = 0
    for nb in n_nodes.tolist():
        q, k, v = qn[off:off + nb], kn[off:off + nb], vs[off:off + nb]
        ktv = np.einsum("lhm,lhd->hmd", k, v)
        num = np.einsum("nhm,hmd->nhd", q, ktv) + v.sum(axis=0)[None]
        den = np.einsum("nhm,hm->nh", q, k.sum(axis=0)) + dt(nb)
        out[off:off + nb] = num / den[..., None]
        off += nb
    return out


def v2_sigmoid_attention(qs, ks, vs, n_nodes):
    """TransConv.full_attention, kernel 'sigmoid' (difformer-v2.py:113-135).

    :124 einsum("abcd,ebcd->aebc"): the node at POSITION b of graph a scores against the node at the SAME
    position b of every graph e (not against the other nodes of its own graph).  Graphs shorter than b+1
    contribute a padded zero key: sigma(0) = 0.5 to the denominator (:127-129, + 1e-9) and a zero value (:134).
    """
    qs, ks, vs = (np.asarray(a) for a in (qs, ks, vs))
    n_nodes = np.asarray(n_nodes, dtype=np.int64)
    dt = qs.dtype.type
    B = n_nodes.shape[0]
    offs = np.concatenate([[0], np.cumsum(n_nodes)])
    out = np.empty(vs.shape, dtype=vs.dtype)
    for p in range(int(n_nodes.max()) if B else 0):
        idx = offs[:-1][n_nodes > p] + p                       # the nodes at position p
        q, k, v = qs[idx], ks[idx], vs[idx]
        s = 1.0 / (1.0 + np.exp(-np.einsum("nhm,lhm->nlh", q, k)))
        den = s.sum(axis=1) + dt(0.5) * dt(B - idx.shape[0]) + dt(1e-9)
        out[idx] = np.einsum("nlh,lhd->nhd", s, v) / den[..., None]
    return out


def difformer_v2_forward(p, x, edge_index, n_nodes, cfg):
    """DIFFormer_v2.forward in eval mode (difformer-v2.py:193-223): as DIFFormer.forward but one head, no
    use_source, attention per graph (above), and ReLU AFTER each layer's LayerNorm (:216-217)."""
    dt = x.dtype.type
    d = cfg["hidden_channels"]
    alpha = dt(cfg.get("alpha", 0.5))
    h = linear(x, p["fcs.0.weight"], p["fcs.0.bias"])                  # :197
    if cfg.get("use_bn", True):
        h = layer_norm(h, p["bns.0.weight"], p["bns.0.bias"])          # :198-199
    h = np.maximum(h, dt(0))                                           # :200
    layers = [h]
    attend = v2_simple_attention if cfg.get("kernel", "simple") == "simple" else v2_sigmoid_attention
    for i in range(cfg["num_layers"]):
        pre = f"convs.{i}."
        q = linear(h, p[pre + "Wq.weight"], p[pre + "Wq.bias"]).reshape(-1, 1, d)      # :143
        k = linear(h, p[pre + "Wk.weight"], p[pre + "Wk.bias"]).reshape(-1, 1, d)      # :144
        v = linear(h, p[pre + "Wv.weight"], p[pre + "Wv.bias"]).reshape(-1, 1, d)      # :145-146 (use_weight)
        att = attend(q, k, v, n_nodes)                                                 # :148
        if cfg.get("use_graph", True):
            g = gcn_conv(v, edge_index, None)
            gw = cfg.get("graph_weight", -1)
            att = (dt(1 - gw) * att + dt(gw) * g) if gw > 0 else (att + g)             # :150-154
        hh = att.mean(axis=1).astype(x.dtype)                                          # :157
        if cfg.get("use_residual", True):
            hh = alpha * hh + (dt(1) - alpha) * layers[i]                              # :212-213
        if cfg.get("use_bn", True):
            hh = layer_norm(hh, p[f"bns.{i + 1}.weight"], p[f"bns.{i + 1}.bias"])      # :214-215
        h = np.maximum(hh, dt(0))                                                      # :217
        layers.append(h)
    return linear(h, p["fcs.1.weight"], p["fcs.1.bias"])                               # :221

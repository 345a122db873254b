"""Differentiable CPU oracle for the TRAINING step of the DIFFormer propagation layer -- TEST INFRASTRUCTURE ONLY.

`difformer_oracle.py` (numpy) restates the forward; this file restates the same lines of
`/root/reference/node classification/difformer.py` and `physical particle/difformer-v2.py` as CPU torch expressions,
because the reference obtains every gradient from autograd (`node classification/main.py:130`, `main-batch.py:141`) and a
gradient oracle has to be the derivative of exactly that forward.  Only `tests/` may import it; nothing under
`difformer_amd/` does.  It never runs on the GPU and is never timed.

Parity pinning: `tests/golden/make_golden_grad.py` runs the reference itself under autograd in the build container and
writes `tests/golden/golden_grad.npz` (loss, dq / dk / dv, d x, d edge_weight, every parameter gradient, float32 and
float64).  `tests/test_oracle_golden.py` holds the functions below to those fixtures (forward and backward) and to the numpy
oracle's forward; the GPU tests then use them at sizes the fixtures cannot cover (a graph that takes the sliced product,
Cora size).

All arithmetic runs in the dtype of the inputs; float64 is the yardstick of SURVEY.md section 8d.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


# ---- a1: full_attention_conv, 'simple' (difformer.py:18-39) ------------------------------------------------------
def simple_attention(qs, ks, vs):
    """qs [N,H,M], ks [L,H,M], vs [L,H,D] -> [N,H,D]."""
    if qs.shape[0] != vs.shape[0]:
        raise ValueError("simple kernel requires N == L (difformer.py:29)")
    qn = qs / torch.norm(qs, p=2)                                        # :20
    kn = ks / torch.norm(ks, p=2)                                        # :21
    num = torch.einsum("nhm,hmd->nhd", qn, torch.einsum("lhm,lhd->hmd", kn, vs))   # :25-26
    num = num + vs.sum(dim=0).unsqueeze(0)                               # :27-29
    den = torch.einsum("nhm,hm->nh", qn, kn.sum(dim=0))                  # :32-34
    den = den.unsqueeze(-1) + qs.shape[0]                                # :37-38
    return num / den                                                     # :39


# ---- a2: full_attention_conv, 'sigmoid' (difformer.py:45-56) -----------------------------------------------------
def sigmoid_attention(qs, ks, vs):
    s = torch.sigmoid(torch.einsum("nhm,lhm->nlh", qs, ks))              # :47
    att = s / s.sum(dim=1, keepdim=True)                                 # :50-55
    return torch.einsum("nlh,lhd->nhd", att, vs)                         # :56


def full_attention_conv(qs, ks, vs, kernel):
    if kernel == "simple":
        return simple_attention(qs, ks, vs)
    if kernel == "sigmoid":
        return sigmoid_attention(qs, ks, vs)
    raise ValueError(f"unknown kernel {kernel!r}")


# ---- a3: gcn_conv (difformer.py:63-79) ---------------------------------------------------------------------------
def gcn_conv(x, edge_index, edge_weight=None):
    """x [N,H,D]; the degree and both d^-1/2 factors are float32 whatever x is (`.float()`, :66); the edge (row -> col)
    lands on `col` (:75); duplicates sum; non-finite values become 0 (:74)."""
    n = x.shape[0]
    row, col = edge_index[0], edge_index[1]
    deg = torch.zeros(n, dtype=torch.float32).index_add_(0, col, torch.ones(col.shape[0], dtype=torch.float32))   # :66
    d_in = (1.0 / deg[col]).sqrt()                                       # :67
    d_out = (1.0 / deg[row]).sqrt()                                      # :68
    if edge_weight is None:
        value = torch.ones_like(row) * d_in * d_out                      # :71
    else:
        value = edge_weight * d_in * d_out                               # :73
    value = torch.nan_to_num(value, nan=0.0, posinf=0.0, neginf=0.0)     # :74
    out = torch.zeros_like(x)
    return out.index_add(0, col, x[row] * value.to(x.dtype).reshape(-1, 1, 1))   # :75-78


# ---- a4: DIFFormerConv.forward (difformer.py:113-145) ------------------------------------------------------------
def difformer_conv(p, prefix, query_input, source_input, edge_index, edge_weight, x_0, cfg):
    h, d = cfg["num_heads"], cfg["hidden_channels"]
    q = F.linear(query_input, p[prefix + "Wq.weight"], p[prefix + "Wq.bias"]).reshape(-1, h, d)    # :115
    k = F.linear(source_input, p[prefix + "Wk.weight"], p[prefix + "Wk.bias"]).reshape(-1, h, d)   # :116
    if cfg.get("use_weight", True):
        v = F.linear(source_input, p[prefix + "Wv.weight"], p[prefix + "Wv.bias"]).reshape(-1, h, d)   # :118
    else:
        v = source_input.reshape(-1, 1, d)                               # :120
    out = full_attention_conv(q, k, v, cfg.get("kernel", "simple"))      # :126
    if cfg.get("use_graph", True):
        g = gcn_conv(v, edge_index, edge_weight)
        gw = cfg.get("graph_weight", -1)
        out = (1 - gw) * out + gw * g if gw > 0 else out + g             # :129-134
    out = out.mean(dim=1)                                                # :137
    if cfg.get("use_source", False):
        out = out + x_0                                                  # :139-140
    return out


# ---- a5: DIFFormer.forward, train() mode with dropout 0 == eval (difformer.py:184-209) ---------------------------
def difformer_forward(p, x, edge_index, edge_weight, cfg):
    """`p`: state_dict keys -> tensors (leaves that require grad for a gradient check)."""
    alpha = cfg.get("alpha", 0.5)
    ln = lambda t, k: F.layer_norm(t, (t.shape[-1],), p[k + ".weight"], p[k + ".bias"], 1e-5)
    h = F.linear(x, p["fcs.0.weight"], p["fcs.0.bias"])                  # :188
    if cfg.get("use_bn", True):
        h = ln(h, "bns.0")                                               # :189-190
    h = torch.relu(h)                                                    # :191
    layers = [h]                                                         # :195
    for i in range(cfg["num_layers"]):
        h = difformer_conv(p, f"convs.{i}.", h, h, edge_index, edge_weight, layers[0], cfg)   # :199
        if cfg.get("use_residual", True):
            h = alpha * h + (1 - alpha) * layers[i]                      # :200-201
        if cfg.get("use_bn", True):
            h = ln(h, f"bns.{i + 1}")                                    # :202-203
        layers.append(h)                                                 # :205
    return F.linear(h, p["fcs.1.weight"], p["fcs.1.bias"])               # :208


def training_loss(out, y, train_idx, kind="nll"):
    """The two criteria of node classification/main.py:121-129."""
    if kind == "bce":
        return F.binary_cross_entropy_with_logits(out[train_idx], y[train_idx].to(out.dtype))   # :124-125
    return F.nll_loss(F.log_softmax(out, dim=1)[train_idx], y[train_idx])                        # :127-129


# ---- f4: physical particle/difformer-v2.py ------------------------------------------------------------------------
def _segments(n_nodes):
    offs = [0]
    for nb in [int(v) for v in n_nodes]:
        offs.append(offs[-1] + nb)
    return offs


def v2_simple_attention(qs, ks, vs, n_nodes):
    """difformer-v2.py:80-111 without the padding: norms over the WHOLE batch (:82-83), per-graph K^T V, sum v, sum k, n_b."""
    qn = qs / torch.norm(qs, p=2)
    kn = ks / torch.norm(ks, p=2)
    offs, outs = _segments(n_nodes), []
    for b in range(len(offs) - 1):
        q, k, v = (t[offs[b]:offs[b + 1]] for t in (qn, kn, vs))
        num = torch.einsum("nhm,hmd->nhd", q, torch.einsum("lhm,lhd->hmd", k, v)) + v.sum(dim=0).unsqueeze(0)   # :93-101
        den = torch.einsum("nhm,hm->nh", q, k.sum(dim=0)) + float(offs[b + 1] - offs[b])                        # :103-109
        outs.append(num / den.unsqueeze(-1))
    return torch.cat(outs, dim=0)


def v2_sigmoid_attention(qs, ks, vs, n_nodes):
    """difformer-v2.py:113-135: position b of graph a scores against position b of EVERY graph e (:124); graphs shorter
    than b + 1 contribute sigma(0) = 0.5 to the denominator and a zero value; + 1e-9 (:127-129)."""
    nn_ = [int(v) for v in n_nodes]
    offs = _segments(nn_)
    out = torch.zeros_like(vs)
    B = len(nn_)
    for pos in range(max(nn_) if B else 0):
        idx = torch.tensor([offs[b] + pos for b in range(B) if nn_[b] > pos])
        q, k, v = qs[idx], ks[idx], vs[idx]
        s = torch.sigmoid(torch.einsum("nhm,lhm->nlh", q, k))
        den = s.sum(dim=1) + 0.5 * (B - idx.shape[0]) + 1e-9
        out = out.index_add(0, idx, torch.einsum("nlh,lhd->nhd", s, v) / den.unsqueeze(-1))
    return out


def difformer_v2_forward(p, x, edge_index, n_nodes, cfg):
    """DIFFormer_v2.forward (difformer-v2.py:193-223), dropout 0: one head, ReLU after every layer's LayerNorm (:216-217)."""
    d = cfg["hidden_channels"]
    alpha = cfg.get("alpha", 0.5)
    ln = lambda t, k: F.layer_norm(t, (t.shape[-1],), p[k + ".weight"], p[k + ".bias"], 1e-5)
    h = F.linear(x, p["fcs.0.weight"], p["fcs.0.bias"])                  # :197
    if cfg.get("use_bn", True):
        h = ln(h, "bns.0")
    h = torch.relu(h)                                                    # :200
    layers = [h]
    attend = v2_simple_attention if cfg.get("kernel", "simple") == "simple" else v2_sigmoid_attention
    for i in range(cfg["num_layers"]):
        pre = f"convs.{i}."
        q = F.linear(h, p[pre + "Wq.weight"], p[pre + "Wq.bias"]).reshape(-1, 1, d)     # :143
        k = F.linear(h, p[pre + "Wk.weight"], p[pre + "Wk.bias"]).reshape(-1, 1, d)     # :144
        v = F.linear(h, p[pre + "Wv.weight"], p[pre + "Wv.bias"]).reshape(-1, 1, d)     # :146
        att = attend(q, k, v, n_nodes)                                   # :148
        if cfg.get("use_graph", True):
            g = gcn_conv(v, edge_index, None)
            gw = cfg.get("graph_weight", -1)
            att = (1 - gw) * att + gw * g if gw > 0 else att + g         # :150-154
        hh = att.mean(dim=1)                                             # :157
        if cfg.get("use_residual", True):
            hh = alpha * hh + (1 - alpha) * layers[i]                    # :212-213
        if cfg.get("use_bn", True):
            hh = ln(hh, f"bns.{i + 1}")                                  # :214-215
        h = torch.relu(hh)                                               # :217
        layers.append(h)
    return F.linear(h, p["fcs.1.weight"], p["fcs.1.bias"])               # :221


def leaves(arrays, dtype=torch.float64):
    """{name: ndarray} -> {name: leaf tensor requiring grad} in `dtype`."""
    return {k: torch.as_tensor(v).to(dtype).clone().requires_grad_(True) for k, v in arrays.items()}

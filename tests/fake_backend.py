"""TEST-ONLY stand-in for difformer_amd.backend_hip.HipBackend, built on the oracle.

It lets the host logic (row sharding, collectives, CSR caching, module plumbing) run on CPU tensors
in the `-m "not gpu"` suite.  It mirrors the HipBackend method contract (same arguments, same
`reduced` record layout, same CSR layout); the product never imports it.
"""
import numpy as np
import torch

from oracle import difformer_oracle as orc


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


class OracleBackend:
    name = "oracle-fake"
    kernel_events = None

    def simple_reduce(self, q, k, v):
        q, k, v = (_np(t).astype(np.float64) for t in (q, k, v))
        rec = np.concatenate([np.einsum("lhm,lhd->hmd", k, v).ravel(), k.sum(0).ravel(), v.sum(0).ravel(),
                              [(q * q).sum(), (k * k).sum()]])
        return torch.from_numpy(rec.astype(np.float32))

    def project_reduce(self, x, Wq, bq, Wk, bk, Wv, bv, H, D):
        lin = torch.nn.functional.linear
        q, k, v = (lin(x, w, b).reshape(-1, H, D) for w, b in ((Wq, bq), (Wk, bk), (Wv, bv)))
        return q, v, self.simple_reduce(q, k, v)

    def simple_apply(self, q, reduced, n_global, D):
        q = _np(q).astype(np.float64)
        n, H, M = q.shape
        r = _np(reduced).astype(np.float64)
        ktv = r[: H * M * D].reshape(H, M, D)
        ksum = r[H * M * D: H * M * D + H * M].reshape(H, M)
        vsum = r[H * M * D + H * M: H * M * D + H * M + H * D].reshape(H, D)
        s = 1.0 / (np.sqrt(r[-2]) * np.sqrt(r[-1]))
        num = s * np.einsum("nhm,hmd->nhd", q, ktv) + vsum[None]
        den = s * np.einsum("nhm,hm->nh", q, ksum)[..., None] + n_global
        return torch.from_numpy((num / den).astype(np.float32))

    def sigmoid_attention(self, q, k, v):
        return torch.from_numpy(orc.sigmoid_attention(*(_np(t).astype(np.float64) for t in (q, k, v))).astype(np.float32))

    def batched_simple_attention(self, q, k, v, graph_ptr):
        n_nodes = np.diff(_np(graph_ptr).astype(np.int64))
        return torch.from_numpy(orc.v2_simple_attention(_np(q), _np(k), _np(v), n_nodes))

    def batched_sigmoid_attention(self, q, k, v, ranked_first, pos_count):
        # rebuild n_nodes from the layout tables: graph r (by rank) has #{p : pos_count[p] > r} nodes
        first, cnt = _np(ranked_first).astype(np.int64), _np(pos_count).astype(np.int64)
        sizes = (cnt[None, :] > np.arange(first.shape[0])[:, None]).sum(axis=1)
        order = np.argsort(first, kind="stable")                  # memory order of the graphs
        assert np.array_equal(np.cumsum(sizes[order]) - sizes[order], first[order])
        return torch.from_numpy(orc.v2_sigmoid_attention(_np(q), _np(k), _np(v), sizes[order]))

    def csr_build(self, edge_index, edge_weight, num_nodes, n_blocks=1, transpose=False, block_rows=0):
        ei = _np(edge_index)
        row, col, val = orc.gcn_edge_values(ei, num_nodes, _np(edge_weight), dtype=np.float32)
        if transpose:
            row, col = col, row
        block_rows = block_rows or -(-num_nodes // n_blocks)
        assert block_rows * n_blocks >= num_nodes
        key = col * n_blocks + row // block_rows
        order = np.argsort(key, kind="stable")
        kptr = np.concatenate([[0], np.cumsum(np.bincount(key, minlength=num_nodes * n_blocks))]).astype(np.int32)
        rowptr = kptr[::n_blocks].copy()
        blk = None
        if n_blocks > 1:
            blk = np.concatenate([kptr[:-1].reshape(num_nodes, n_blocks).T, kptr[n_blocks::n_blocks][None]], axis=0)
            blk = torch.from_numpy(blk.astype(np.int32).ravel())
        return (torch.from_numpy(rowptr), blk, torch.from_numpy(row[order].astype(np.int32)),
                torch.from_numpy(val[order]), int(np.diff(rowptr.astype(np.int64)).max()) if num_nodes > 0 else 0)

    def edge_weight_grad(self, edge_index, edge_weight, rowptr, n_nodes, g, x, scale=1.0):
        """difformer.py:73-74 under autograd, restated term by term in float32 (NaN where the source has no incoming entry)."""
        ei, w = _np(edge_index), _np(edge_weight).astype(np.float32)
        deg = np.diff(_np(rowptr).astype(np.int64)).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            d_in, d_out = np.sqrt(np.float32(1) / deg[ei[1]]), np.sqrt(np.float32(1) / deg[ei[0]])
            dot = np.einsum("ef,ef->e", _np(g).astype(np.float64)[ei[1]], _np(x).astype(np.float64)[ei[0]])
            gv = np.where(np.isfinite(w * d_in * d_out), scale * dot, 0.0).astype(np.float32)
            return torch.from_numpy(((gv * d_out) * d_in).astype(np.float32))

    def spmm(self, rowptr, blkptr, n_blocks, src, val, n_nodes, nnz, x, row_begin, n_rows, attn=None, attn_scale=1.0,
             gcn_scale=1.0, tail=None, order=None, part=None):
        rp, s, w, xx = _np(rowptr), _np(src)[:nnz], _np(val)[:nnz].astype(np.float64), _np(x).astype(np.float64)
        dst = np.repeat(np.arange(n_nodes), np.diff(rp))
        full = np.zeros((n_nodes, xx.shape[1]))
        if part is None:
            assert xx.shape[0] == n_nodes
            np.add.at(full, dst, w[:, None] * xx[s])
        else:
            # split product (row-sharded runs): part 0 sees ONLY this rank's own value rows and must not need any other
            phase, own_lo, own_hi, scratch, x_row0 = part
            self.part_calls = getattr(self, "part_calls", {})
            self.part_calls[phase] = self.part_calls.get(phase, 0) + 1
            own = (s >= x_row0) & (s < x_row0 + xx.shape[0]) if phase == 0 else None
            if phase == 0:
                blk = _np(blkptr).reshape(n_blocks + 1, n_nodes)
                in_own_blocks = np.zeros(nnz, dtype=bool)
                for r in range(n_nodes):
                    in_own_blocks[blk[own_lo, r]: blk[own_hi, r]] = True
                assert np.array_equal(own, in_own_blocks), "own blocks and own value rows must coincide"
                np.add.at(full, dst[own], w[own, None] * xx[s[own] - x_row0])
                return torch.from_numpy(full[row_begin:row_begin + n_rows].copy())          # the parked accumulators
            assert xx.shape[0] == n_nodes
            blk = _np(blkptr).reshape(n_blocks + 1, n_nodes)
            rest = np.ones(nnz, dtype=bool)
            for r in range(n_nodes):
                rest[blk[own_lo, r]: blk[own_hi, r]] = False
            np.add.at(full, dst[rest], w[rest, None] * xx[s[rest]])
            full[row_begin:row_begin + n_rows] += _np(scratch)
        out = gcn_scale * full[row_begin:row_begin + n_rows]
        if attn is not None:
            out = out + attn_scale * _np(attn).astype(np.float64)
        out = torch.from_numpy(out.astype(np.float32))
        if tail is not None:
            out = self.layer_tail(out[:, None, :], tail.get("x0"), tail.get("prev"), tail.get("alpha", 0.5),
                                  tail.get("ln_weight"), tail.get("ln_bias"), tail.get("eps", 1e-5),
                                  tail.get("relu", False))
        return out

    # ---- closed-form `simple` layer (csrc/simple_layer.hip), restated with numpy in float64 --------------------------
    def gram(self, x, rowptr=None, plan=None):
        xx = _np(x).astype(np.float64)
        rec = np.concatenate([(xx.T @ xx).ravel(), xx.sum(0), [0.0, 0.0]])
        return torch.from_numpy(rec.astype(np.float32)), None

    def simple_coeffs(self, record, n_global, C, D, Wq, bq, Wk, bk, Wv, bv, attn_scale):
        r = _np(record).astype(np.float64)
        G, sx, N = r[: C * C].reshape(C, C), r[C * C: C * C + C], float(n_global)
        Wq, bq, Wk, bk = (_np(t).astype(np.float64) for t in (Wq, bq, Wk, bk))
        Wv = np.eye(D, C) if Wv is None else _np(Wv).astype(np.float64)
        bv = np.zeros(D) if bv is None else _np(bv).astype(np.float64)
        ktv = Wk @ G @ Wv.T + np.outer(Wk @ sx, bv) + np.outer(bk, Wv @ sx) + N * np.outer(bk, bv)
        ksum, vsum = Wk @ sx + N * bk, Wv @ sx + N * bv
        q2 = np.trace(Wq @ G @ Wq.T) + 2 * bq @ (Wq @ sx) + N * bq @ bq
        k2 = np.trace(Wk @ G @ Wk.T) + 2 * bk @ (Wk @ sx) + N * bk @ bk
        s = 1.0 / (np.sqrt(q2) * np.sqrt(k2))
        coef = np.concatenate([(attn_scale * s * (Wq.T @ ktv)).T.ravel(), attn_scale * (s * bq @ ktv + vsum),
                               s * (Wq.T @ ksum), [s * bq @ ksum + N, s, q2, k2]])
        return torch.from_numpy(coef.astype(np.float32))

    def gram_coeffs(self, x, n_global, C, D, Wq, bq, Wk, bk, Wv, bv, attn_scale):
        self.gram_coeffs_calls = getattr(self, "gram_coeffs_calls", 0) + 1
        record, _ = self.gram(x)
        return record, self.simple_coeffs(record, n_global, C, D, Wq, bq, Wk, bk, Wv, bv, attn_scale)

    def simple_layer(self, x, coef, D, ax=None, Wv=None, bv=None, row_sums=None, gcn_scale=1.0, x0=None, residual=False,
                     alpha=0.5, ln_weight=None, ln_bias=None, eps=1e-5, relu=False, next_rowptr=None, next_plan=None,
                     next_record=False, head=None, gather=None):
        assert gather is None            # the in-kernel aggregation is the HIP backend's (ops checks for _simple_layer_gather)
        self.closed_form_calls = getattr(self, "closed_form_calls", 0) + 1
        xx, cf = _np(x).astype(np.float64), _np(coef).astype(np.float64)
        C = xx.shape[1]
        MnT, cn, u, cd = cf[: D * C].reshape(D, C), cf[D * C: D * C + D], cf[D * C + D: D * C + D + C], cf[D * C + D + C]
        z = (xx @ MnT.T + cn) / (xx @ u + cd)[:, None]
        if ax is not None:
            a = _np(ax).astype(np.float64)
            if Wv is not None:
                z = z + a @ _np(Wv).astype(np.float64).T
                if row_sums is not None:
                    z = z + gcn_scale * np.outer(_np(row_sums).astype(np.float64), _np(bv).astype(np.float64))
            else:
                z = z + a
        if x0 is not None:
            z = z + _np(x0)
        if residual:
            z = alpha * z + (1.0 - alpha) * xx
        if ln_weight is not None:
            z = orc.layer_norm(z, _np(ln_weight).astype(np.float64), _np(ln_bias).astype(np.float64), eps)
        if relu:
            z = np.maximum(z, 0.0)
        if head is not None:                    # the model's output Linear in the same pass (difformer.py:208)
            z = z @ _np(head[0]).astype(np.float64).T + _np(head[1]).astype(np.float64)
        out = torch.from_numpy(z.astype(np.float32))
        return out if (next_plan is None and not next_record) else (out, None, None)

    def row_order(self, rowptr, row_begin, n_rows):
        deg = np.diff(_np(rowptr).astype(np.int64))[row_begin: row_begin + n_rows]
        stats = np.array([(deg * n_rows > 4 * deg.sum()).sum(), deg.max()], dtype=np.int32)
        return torch.from_numpy(np.argsort(-deg, kind="stable").astype(np.int32)), torch.from_numpy(stats)

    def linear(self, x, weight, bias, ln_weight=None, ln_bias=None, eps=1e-5, relu=False):
        y = torch.nn.functional.linear(x, weight, bias)
        return self.layer_tail(y.unsqueeze(1), None, None, 0.5, ln_weight, ln_bias, eps, relu)

    def layer_tail_bwd(self, *args):
        return None          # -> autograd_ops re-derives the gradient with tensor ops

    def layer_tail(self, conv, x0, prev, alpha, ln_weight, ln_bias, eps, relu=False):
        z = _np(conv).astype(np.float64).mean(axis=1)
        if x0 is not None:
            z = z + _np(x0)
        if prev is not None:
            z = alpha * z + (1.0 - alpha) * _np(prev)
        if ln_weight is not None:
            z = orc.layer_norm(z, _np(ln_weight).astype(np.float64), _np(ln_bias).astype(np.float64), eps)
        if relu:
            z = np.maximum(z, 0.0)
        return torch.from_numpy(z.astype(np.float32))

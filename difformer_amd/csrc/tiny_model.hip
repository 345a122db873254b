// Whole-model kernels for TINY graphs: `spatial-temporal/` trains DIFFormer at hidden 4 on 20 / 129 / 1,068-node snapshots
// (spatial-temporal/run.sh:5-40, main.py:94-120: hundreds of forwards per epoch, each on tensors fresh from
// `snapshot.to(device)`).  At that size the layer-by-layer path is ~110 launches per training snapshot (0.33 ms of kernels,
// 1.1 ms of host time); here the whole of
//     DIFFormer.forward            node classification/difformer.py:184-209  (= spatial-temporal/difformer.py:173-198)
//       DIFFormerConv.forward      :113-145      full_attention_conv :10-61      gcn_conv :63-79
// is ONE launch of one workgroup, its backward (what autograd derives for main.py:119 `cost_tr.backward`) a second one, and
// the graph preparation (degree, normalised values, destination-major CSR and its transpose, both in stable edge order) a
// third.  One head, hidden <= 8, <= 4,096 nodes, <= 64 input features, <= 8 outputs, float32.
//
// Layout of the work: a thread owns node i (i += blockDim for more nodes than threads); everything a node needs from
// OTHER nodes goes through per-node arrays in a caller-owned scratch / tape buffer (L1 / L2 resident: 17 KB per array at
// 1,068 x 4) between __syncthreads(); sums over nodes (K^T V, weight gradients, LayerNorm gradients) are computed by
// `outer_sum`: outputs x node-chunks spread over the threads, chunk partials in LDS, added in chunk order in float64 --
// bitwise reproducible.  The sigmoid kernel's O(N^2) pair loop keeps the stationary node in registers and streams the
// others through LDS tiles (uniform addresses: broadcast reads).
#include "tiny_common.h"

using namespace tiny;

namespace {

// ======================================================================================================================
// forward
// ======================================================================================================================
template <int DP>
__global__ __launch_bounds__(512) void tiny_forward_kernel(const TinyArgs a) {
    constexpr int TK = 4096 / DP;                 // keys per LDS tile of the sigmoid sweep (K and V rows: 32 KiB)
    __shared__ float sW0[DP * kMaxIn];
    __shared__ float sB0[DP], sLn0w[DP], sLn0b[DP];
    __shared__ LayerW<DP> sL;
    __shared__ float sWo[kMaxOut * DP], sBo[kMaxOut];
    __shared__ float sSum[96];
    __shared__ double sPart[1024];
    __shared__ float sKV[TK * 2 * DP];
    const int n = a.n, d = a.d, L = a.layers, T = blockDim.x, t = threadIdx.x;
    Tape<DP> tp(a.tape, n, L);
    const bool drop = a.training && a.rnd != nullptr && a.p_drop > 0.f;

    for (int k = t; k < DP * kMaxIn; k += T) {
        const int m = k / kMaxIn, f = k % kMaxIn;
        sW0[k] = (m < d && f < a.f_in) ? a.w0[m * a.f_in + f] : 0.f;
    }
    for (int k = t; k < DP; k += T) {
        sB0[k] = k < d ? a.b0[k] : 0.f;
        sLn0w[k] = (k < d && a.use_bn) ? a.ln0w[k] : 0.f;
        sLn0b[k] = (k < d && a.use_bn) ? a.ln0b[k] : 0.f;
    }
    for (int k = t; k < kMaxOut * DP; k += T) {
        const int c = k / DP, m = k % DP;
        sWo[k] = (c < a.c && m < d) ? a.wo[c * d + m] : 0.f;
    }
    for (int k = t; k < kMaxOut; k += T) sBo[k] = k < a.c ? a.bo[k] : 0.f;
    __syncthreads();

    // ---- input layer: Linear -> LayerNorm -> ReLU -> dropout (:188-192) ----
    for (int i = t; i < n; i += T) {
        float h[DP];
#pragma unroll
        for (int m = 0; m < DP; ++m) h[m] = sB0[m];
        const float* xr = a.x + i * a.ldx;
        for (int f = 0; f < a.f_in; ++f) {
            const float xv = xr[f];
#pragma unroll
            for (int m = 0; m < DP; ++m) h[m] += sW0[m * kMaxIn + f] * xv;
        }
        store_row<DP>(tp.Z + static_cast<size_t>(i) * DP, h);
        if (a.use_bn) {
            float mean, rstd;
            ln_stats<DP>(h, d, a.eps, mean, rstd);
#pragma unroll
            for (int m = 0; m < DP; ++m) h[m] = (m < d) ? (h[m] - mean) * rstd * sLn0w[m] + sLn0b[m] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < DP; ++m) h[m] = fmaxf(h[m], 0.f);
        if (drop) dropout_row<DP>(h, a.rnd, static_cast<int64_t>(i) * d, d, a.p_drop);
        store_row<DP>(tp.H + static_cast<size_t>(i) * DP, h);
    }

    for (int l = 0; l < L; ++l) {
        __syncthreads();
        load_layer<DP>(sL, a.lp[l], d, a.use_weight, a.use_bn);
        __syncthreads();
        const float* Hl = tp.H + static_cast<size_t>(l) * n * DP;
        // ---- projections (:115-120) ----
        for (int i = t; i < n; i += T) {
            float h[DP], q[DP], k[DP], v[DP];
            load_row<DP>(Hl + static_cast<size_t>(i) * DP, h);
            matvec<DP>(sL.wq, sL.bq, h, q);
            matvec<DP>(sL.wk, sL.bk, h, k);
            if (a.use_weight) matvec<DP>(sL.wv, sL.bv, h, v);
            else {
#pragma unroll
                for (int m = 0; m < DP; ++m) v[m] = h[m];
            }
            store_row<DP>(tp.Q + static_cast<size_t>(i) * DP, q);
            store_row<DP>(tp.K + static_cast<size_t>(i) * DP, k);
            store_row<DP>(tp.V + static_cast<size_t>(i) * DP, v);
        }
        __syncthreads();
        if (!a.sigmoid) {
            // K^T V, ksum, vsum, |Q|^2, |K|^2 (:20-34): sums over nodes
            outer_sum(tp.K, DP, DP, tp.V, DP, DP, n, 1.f, sSum, sPart);
            outer_sum(tp.K, DP, DP, nullptr, 0, 1, n, 1.f, sSum + DP * DP, sPart);
            outer_sum(tp.V, DP, DP, nullptr, 0, 1, n, 1.f, sSum + DP * DP + DP, sPart);
            // sum of squares: diagonal of Q^T Q / K^T K summed -- one output per column, then over the columns
            __shared__ float sSq[2 * DP];
            for (int which = 0; which < 2; ++which) {
                const float* P = which ? tp.K : tp.Q;
                // outer_sum of P with itself restricted to the diagonal: A = B = P, but only m == c wanted -> use M = DP, C = 1
                // on the element-wise squares: done by a dedicated chunked loop below
                const int Ob = DP;
                int chunks = T / Ob;
                if (chunks > 64) chunks = 64;
                const int len = (n + chunks - 1) / chunks;
                const int o = t % Ob, ch = t / Ob;
                if (ch < chunks) {
                    double acc = 0.0;
                    const int i1 = min(n, (ch + 1) * len);
                    for (int i = ch * len; i < i1; ++i) {
                        const double vq = P[static_cast<size_t>(i) * DP + o];
                        acc += vq * vq;
                    }
                    sPart[ch * Ob + o] = acc;
                }
                __syncthreads();
                if (t < Ob) {
                    double s = 0.0;
                    for (int k = 0; k < chunks; ++k) s += sPart[k * Ob + t];
                    sSq[which * DP + t] = static_cast<float>(s);
                }
                __syncthreads();
            }
            if (t == 0) {
                double q2 = 0.0, k2 = 0.0;
                for (int m = 0; m < DP; ++m) { q2 += sSq[m]; k2 += sSq[DP + m]; }
                sSum[DP * DP + 2 * DP] = static_cast<float>(q2);
                sSum[DP * DP + 2 * DP + 1] = static_cast<float>(k2);
            }
            __syncthreads();
            for (int k = t; k < DP * DP + 2 * DP + 2; k += T) tp.SUM[l * 96 + k] = sSum[k];
        }
        // ---- attention + aggregation + tail per node ----
        const int slots = (n + T - 1) / T;
        for (int slot = 0; slot < slots; ++slot) {
            const int i = slot * T + t;
            const bool live = i < n;
            float q[DP], att[DP];
#pragma unroll
            for (int m = 0; m < DP; ++m) { q[m] = 0.f; att[m] = 0.f; }
            if (live) load_row<DP>(tp.Q + static_cast<size_t>(i) * DP, q);
            if (a.sigmoid) {
                // :47-56  sigma(q k^T) / row sum, applied to v; the keys stream through LDS tiles; a tile's sums in float32,
                // the tiles added in float64
                double den = 0.0, accd[DP];
#pragma unroll
                for (int m = 0; m < DP; ++m) accd[m] = 0.0;
                for (int j0 = 0; j0 < n; j0 += TK) {
                    const int cnt = min(TK, n - j0);
                    __syncthreads();
                    for (int k = t; k < cnt * DP; k += T) {
                        sKV[k] = tp.K[static_cast<size_t>(j0) * DP + k];
                        sKV[TK * DP + k] = tp.V[static_cast<size_t>(j0) * DP + k];
                    }
                    __syncthreads();
                    if (live) {
                        for (int j0b = 0; j0b < cnt; j0b += 64) {
                            float denf = 0.f, accf[DP];
#pragma unroll
                            for (int m = 0; m < DP; ++m) accf[m] = 0.f;
                            const int j1 = min(cnt, j0b + 64);
                            for (int j = j0b; j < j1; ++j) {
                                float dot = 0.f;
#pragma unroll
                                for (int m = 0; m < DP; ++m) dot += q[m] * sKV[j * DP + m];
                                const float p = sigmoidf(dot);
                                denf += p;
#pragma unroll
                                for (int m = 0; m < DP; ++m) accf[m] += p * sKV[TK * DP + j * DP + m];
                            }
                            den += static_cast<double>(denf);
#pragma unroll
                            for (int m = 0; m < DP; ++m) accd[m] += static_cast<double>(accf[m]);
                        }
                    }
                }
                if (live) {
#pragma unroll
                    for (int m = 0; m < DP; ++m) att[m] = static_cast<float>(accd[m] / den);
                    tp.DEN[static_cast<size_t>(l) * n + i] = static_cast<float>(den);
                }
            } else if (live) {
                const float q2 = sSum[DP * DP + 2 * DP], k2 = sSum[DP * DP + 2 * DP + 1];
                const float s = 1.0f / (sqrtf(q2) * sqrtf(k2));                      // :20-21 global Frobenius norms
                float den = 0.f;
#pragma unroll
                for (int m = 0; m < DP; ++m) den += q[m] * sSum[DP * DP + m];
                den = s * den + static_cast<float>(n);                               // :31-38  (+N from qs.shape[0])
#pragma unroll
                for (int dd = 0; dd < DP; ++dd) {
                    float num = 0.f;
#pragma unroll
                    for (int m = 0; m < DP; ++m) num += q[m] * sSum[m * DP + dd];
                    att[dd] = (s * num + sSum[DP * DP + DP + dd]) / den;             // :25-29
                }
            }
            if (live) {
                store_row<DP>(tp.ATT + (static_cast<size_t>(l) * n + i) * DP, att);
                float out[DP];
                if (a.use_graph) {
                    float g[DP];
#pragma unroll
                    for (int m = 0; m < DP; ++m) g[m] = 0.f;
                    const int e1 = a.rowptr[i + 1];
                    for (int e = a.rowptr[i]; e < e1; ++e) {                          // :75-78, entries in edge order
                        const float w = a.val[e];
                        float vr[DP];
                        load_row<DP>(tp.V + static_cast<size_t>(a.nbr[e]) * DP, vr);
#pragma unroll
                        for (int m = 0; m < DP; ++m) g[m] += w * vr[m];
                    }
#pragma unroll
                    for (int m = 0; m < DP; ++m) out[m] = a.a_s * att[m] + a.g_s * g[m]; // :130-134
                } else {
#pragma unroll
                    for (int m = 0; m < DP; ++m) out[m] = att[m];
                }
                float h[DP];
                load_row<DP>(Hl + static_cast<size_t>(i) * DP, h);
                if (a.use_source) {                                                  // :139-140
                    float x0[DP];
                    load_row<DP>(tp.H + static_cast<size_t>(i) * DP, x0);
#pragma unroll
                    for (int m = 0; m < DP; ++m) out[m] += x0[m];
                }
                if (a.residual) {                                                    // :200-201
#pragma unroll
                    for (int m = 0; m < DP; ++m) out[m] = a.alpha * out[m] + (1.0f - a.alpha) * h[m];
                }
                store_row<DP>(tp.Z + (static_cast<size_t>(l + 1) * n + i) * DP, out);
                if (a.use_bn) {                                                      // :202-203
                    float mean, rstd;
                    ln_stats<DP>(out, d, a.eps, mean, rstd);
#pragma unroll
                    for (int m = 0; m < DP; ++m) out[m] = (m < d) ? (out[m] - mean) * rstd * sL.lnw[m] + sL.lnb[m] : 0.f;
                }
                if (drop) dropout_row<DP>(out, a.rnd, (static_cast<int64_t>(l + 1) * n + i) * d, d, a.p_drop);   // :204
                store_row<DP>(tp.H + (static_cast<size_t>(l + 1) * n + i) * DP, out);
            }
        }
    }
    // ---- output Linear (:208): a node's own row, no exchange needed ----
    for (int i = t; i < n; i += T) {
        float h[DP];
        load_row<DP>(tp.H + (static_cast<size_t>(L) * n + i) * DP, h);
        for (int c = 0; c < a.c; ++c) {
            float acc = sBo[c];
#pragma unroll
            for (int m = 0; m < DP; ++m) acc += sWo[c * DP + m] * h[m];
            a.y[static_cast<size_t>(i) * a.c + c] = acc;
        }
    }
}

// ======================================================================================================================
// backward
// ======================================================================================================================
// scratch slots ([n][DP] each): 0 DH (gradient of the current layer input)  1 DX0  2 SG (g_s d_out)  3 DNUM / DA  4 DY
// 5 DYX  6 DIR  7 DQ  8 DK  9 DV  10 Q  11 K  12 V  13 DPRE;  then [n] arrays: DDEN / DL, TS, spare
template <int DP>
__global__ __launch_bounds__(512) void tiny_backward_kernel(const TinyArgs a) {
    constexpr int TK = 2048 / DP;                 // nodes per LDS tile of the sigmoid backward sweep (4 rows + 1 scalar each)
    __shared__ float sW0[DP * kMaxIn];
    __shared__ float sLn0w[DP], sLn0b[DP];
    __shared__ LayerW<DP> sL;
    __shared__ float sWo[kMaxOut * DP];
    __shared__ float sSum[96];                    // forward sums of the layer (simple)
    __shared__ float sG[96];                      // their gradients: dKtV [DP*DP], dksum [DP], dvsum [DP], ds
    __shared__ float sVec[4 * DP];
    __shared__ double sPart[1024];
    __shared__ float sTile[TK * (4 * DP + 1)];
    const int n = a.n, d = a.d, L = a.layers, T = blockDim.x, t = threadIdx.x;
    Tape<DP> tp(a.tape, n, L);
    const size_t nd = static_cast<size_t>(n) * DP;
    float* S = a.scratch;
    float *DH = S, *DX0 = S + nd, *SG = S + 2 * nd, *DNUM = S + 3 * nd, *DY = S + 4 * nd, *DYX = S + 5 * nd, *DIR = S + 6 * nd,
          *DQ = S + 7 * nd, *DK = S + 8 * nd, *DV = S + 9 * nd, *Q = S + 10 * nd, *K = S + 11 * nd, *V = S + 12 * nd,
          *DPRE = S + 13 * nd;
    float* DDEN = S + kBwdSlots * nd;
    float* TS = DDEN + n;
    const bool drop = a.training && a.rnd != nullptr && a.p_drop > 0.f;
    const float keep = drop ? 1.0f / (1.0f - a.p_drop) : 1.0f;

    for (int k = t; k < DP * kMaxIn; k += T) {
        const int m = k / kMaxIn, f = k % kMaxIn;
        sW0[k] = (m < d && f < a.f_in) ? a.w0[m * a.f_in + f] : 0.f;
    }
    for (int k = t; k < DP; k += T) {
        sLn0w[k] = (k < d && a.use_bn) ? a.ln0w[k] : 0.f;
        sLn0b[k] = (k < d && a.use_bn) ? a.ln0b[k] : 0.f;
    }
    for (int k = t; k < kMaxOut * DP; k += T) {
        const int c = k / DP, m = k % DP;
        sWo[k] = (c < a.c && m < d) ? a.wo[c * d + m] : 0.f;
    }
    __syncthreads();

    // ---- output Linear: d H[L] = gy Wo; d Wo = gy^T H[L]; d bo = sum gy ----
    for (int i = t; i < n; i += T) {
        float g[DP];
#pragma unroll
        for (int m = 0; m < DP; ++m) g[m] = 0.f;
        for (int c = 0; c < a.c; ++c) {
            const float gv = a.gy[static_cast<size_t>(i) * a.c + c];
#pragma unroll
            for (int m = 0; m < DP; ++m) g[m] += gv * sWo[c * DP + m];
        }
        store_row<DP>(DH + static_cast<size_t>(i) * DP, g);
        if (a.use_source) {
#pragma unroll
            for (int m = 0; m < DP; ++m) g[m] = 0.f;
            store_row<DP>(DX0 + static_cast<size_t>(i) * DP, g);
        }
    }
    {
        const float* HL = tp.H + static_cast<size_t>(L) * nd;
        // d Wo [c][m] = sum_i gy[i][c] H[L][i][m]  (true width d in the output)
        for (int c = 0; c < a.c; ++c) {
            outer_sum(a.gy + c, a.c, 1, HL, DP, DP, n, 1.f, sVec, sPart);
            if (t < d) a.gwo[c * d + t] = sVec[t];
            __syncthreads();
        }
        outer_sum(a.gy, a.c, a.c, nullptr, 0, 1, n, 1.f, sVec, sPart);
        if (t < a.c) a.gbo[t] = sVec[t];
        __syncthreads();
    }

    for (int l = L - 1; l >= 0; --l) {
        load_layer<DP>(sL, a.lp[l], d, a.use_weight, a.use_bn);
        if (!a.sigmoid)
            for (int k = t; k < DP * DP + 2 * DP + 2; k += T) sSum[k] = tp.SUM[l * 96 + k];
        __syncthreads();
        const float* Hl = tp.H + static_cast<size_t>(l) * nd;
        const float q2 = sSum[DP * DP + 2 * DP], k2 = sSum[DP * DP + 2 * DP + 1];
        const float s = a.sigmoid ? 0.f : 1.0f / (sqrtf(q2) * sqrtf(k2));
        // ---- phase 1: tail backward, attention set-up, per node ----
        for (int i = t; i < n; i += T) {
            float dy[DP], z[DP], dz[DP];
            load_row<DP>(DH + static_cast<size_t>(i) * DP, dy);
            if (drop) {
                const int64_t at = (static_cast<int64_t>(l + 1) * n + i) * d;
#pragma unroll
                for (int m = 0; m < DP; ++m)
                    if (m < d) dy[m] = (a.rnd[at + m] >= a.p_drop) ? dy[m] * keep : 0.f;
            }
            if (a.use_bn) {
                load_row<DP>(tp.Z + (static_cast<size_t>(l + 1) * n + i) * DP, z);
                float mean, rstd;
                ln_stats<DP>(z, d, a.eps, mean, rstd);
                float xh[DP], m1 = 0.f, m2 = 0.f;
#pragma unroll
                for (int m = 0; m < DP; ++m) {
                    xh[m] = (m < d) ? (z[m] - mean) * rstd : 0.f;
                    const float gw = sL.lnw[m] * dy[m];
                    m1 += gw;
                    m2 += gw * xh[m];
                }
                m1 /= static_cast<float>(d);
                m2 /= static_cast<float>(d);
                float dyx[DP];
#pragma unroll
                for (int m = 0; m < DP; ++m) {
                    dyx[m] = dy[m] * xh[m];
                    dz[m] = (m < d) ? rstd * (sL.lnw[m] * dy[m] - m1 - xh[m] * m2) : 0.f;
                }
                store_row<DP>(DY + static_cast<size_t>(i) * DP, dy);
                store_row<DP>(DYX + static_cast<size_t>(i) * DP, dyx);
            } else {
#pragma unroll
                for (int m = 0; m < DP; ++m) dz[m] = dy[m];
            }
            float dout[DP], dir[DP];
#pragma unroll
            for (int m = 0; m < DP; ++m) {
                dout[m] = a.residual ? a.alpha * dz[m] : dz[m];
                dir[m] = a.residual ? (1.0f - a.alpha) * dz[m] : 0.f;
            }
            store_row<DP>(DIR + static_cast<size_t>(i) * DP, dir);
            if (a.use_source) {
                float acc[DP];
                load_row<DP>(DX0 + static_cast<size_t>(i) * DP, acc);
#pragma unroll
                for (int m = 0; m < DP; ++m) acc[m] += dout[m];
                store_row<DP>(DX0 + static_cast<size_t>(i) * DP, acc);
            }
            float datt[DP];
            if (a.use_graph) {
                float sg[DP];
#pragma unroll
                for (int m = 0; m < DP; ++m) { sg[m] = a.g_s * dout[m]; datt[m] = a.a_s * dout[m]; }
                store_row<DP>(SG + static_cast<size_t>(i) * DP, sg);
            } else {
#pragma unroll
                for (int m = 0; m < DP; ++m) datt[m] = dout[m];
            }
            float h[DP], q[DP], k[DP], v[DP];
            load_row<DP>(Hl + static_cast<size_t>(i) * DP, h);
            matvec<DP>(sL.wq, sL.bq, h, q);
            matvec<DP>(sL.wk, sL.bk, h, k);
            if (a.use_weight) matvec<DP>(sL.wv, sL.bv, h, v);
            else {
#pragma unroll
                for (int m = 0; m < DP; ++m) v[m] = h[m];
            }
            store_row<DP>(Q + static_cast<size_t>(i) * DP, q);
            store_row<DP>(K + static_cast<size_t>(i) * DP, k);
            store_row<DP>(V + static_cast<size_t>(i) * DP, v);
            if (a.sigmoid) {
                float att[DP], da[DP];
                load_row<DP>(tp.ATT + (static_cast<size_t>(l) * n + i) * DP, att);
                const float rden = 1.0f / tp.DEN[static_cast<size_t>(l) * n + i];
                float dl = 0.f;
#pragma unroll
                for (int m = 0; m < DP; ++m) { da[m] = datt[m] * rden; dl += datt[m] * att[m]; }
                store_row<DP>(DNUM + static_cast<size_t>(i) * DP, da);          // DA_i = d att_i / den_i
                DDEN[i] = dl * rden;                                             // DL_i = (d att_i . att_i) / den_i
            } else {
                float A[DP], b = 0.f;
#pragma unroll
                for (int dd = 0; dd < DP; ++dd) {
                    float acc = 0.f;
#pragma unroll
                    for (int m = 0; m < DP; ++m) acc += q[m] * sSum[m * DP + dd];
                    A[dd] = acc;
                }
#pragma unroll
                for (int m = 0; m < DP; ++m) b += q[m] * sSum[DP * DP + m];
                const float den = s * b + static_cast<float>(n);
                float dnum[DP], dden = 0.f, ts = 0.f;
#pragma unroll
                for (int dd = 0; dd < DP; ++dd) {
                    const float att = (s * A[dd] + sSum[DP * DP + DP + dd]) / den;
                    dnum[dd] = datt[dd] / den;
                    dden -= dnum[dd] * att;
                    ts += dnum[dd] * A[dd];
                }
                ts += dden * b;
                float dq[DP];
#pragma unroll
                for (int m = 0; m < DP; ++m) {
                    float acc = 0.f;
#pragma unroll
                    for (int dd = 0; dd < DP; ++dd) acc += sSum[m * DP + dd] * dnum[dd];
                    dq[m] = s * (acc + dden * sSum[DP * DP + m]);
                }
                store_row<DP>(DNUM + static_cast<size_t>(i) * DP, dnum);
                store_row<DP>(DQ + static_cast<size_t>(i) * DP, dq);
                DDEN[i] = dden;
                TS[i] = ts;
            }
        }
        __syncthreads();
        // ---- phase 2: sums over nodes ----
        if (a.use_bn) {
            outer_sum(DYX, DP, DP, nullptr, 0, 1, n, 1.f, sVec, sPart);
            if (t < d) a.lg[l].lnw[t] = sVec[t];
            __syncthreads();
            outer_sum(DY, DP, DP, nullptr, 0, 1, n, 1.f, sVec, sPart);
            if (t < d) a.lg[l].lnb[t] = sVec[t];
            __syncthreads();
        }
        float gq2 = 0.f, gk2 = 0.f;
        if (!a.sigmoid) {
            outer_sum(Q, DP, DP, DNUM, DP, DP, n, s, sG, sPart);                       // d KtV [m][dd] = s sum q[m] dnum[dd]
            outer_sum(Q, DP, DP, DDEN, 1, 1, n, s, sG + DP * DP, sPart);               // d ksum [m]    = s sum dden q[m]
            outer_sum(DNUM, DP, DP, nullptr, 0, 1, n, 1.f, sG + DP * DP + DP, sPart);  // d vsum [dd]   = sum dnum[dd]
            outer_sum(TS, 1, 1, nullptr, 0, 1, n, 1.f, sG + DP * DP + 2 * DP, sPart);  // d s
            const float ds = sG[DP * DP + 2 * DP];
            gq2 = -0.5f * s * ds / q2;                                                // s = q2^-1/2 k2^-1/2
            gk2 = -0.5f * s * ds / k2;
        }
        // ---- phase 3: attention + aggregation + projection backward per node ----
        const int slots = (n + T - 1) / T;
        for (int slot = 0; slot < slots; ++slot) {
            const int i = slot * T + t;
            const bool live = i < n;
            float q[DP], k[DP], v[DP], dq[DP], dk[DP], dv[DP];
#pragma unroll
            for (int m = 0; m < DP; ++m) { q[m] = k[m] = v[m] = dq[m] = dk[m] = dv[m] = 0.f; }
            if (live) {
                load_row<DP>(Q + static_cast<size_t>(i) * DP, q);
                load_row<DP>(K + static_cast<size_t>(i) * DP, k);
                load_row<DP>(V + static_cast<size_t>(i) * DP, v);
            }
            if (a.sigmoid) {
                float da[DP], dl = 0.f;
                double dqd[DP], dkd[DP], dvd[DP];
#pragma unroll
                for (int m = 0; m < DP; ++m) { da[m] = 0.f; dqd[m] = 0.0; dkd[m] = 0.0; dvd[m] = 0.0; }
                if (live) {
                    load_row<DP>(DNUM + static_cast<size_t>(i) * DP, da);
                    dl = DDEN[i];
                }
                for (int o0 = 0; o0 < n; o0 += TK) {
                    const int cnt = min(TK, n - o0);
                    __syncthreads();
                    for (int kk = t; kk < cnt * DP; kk += T) {
                        sTile[kk] = Q[static_cast<size_t>(o0) * DP + kk];
                        sTile[TK * DP + kk] = K[static_cast<size_t>(o0) * DP + kk];
                        sTile[2 * TK * DP + kk] = V[static_cast<size_t>(o0) * DP + kk];
                        sTile[3 * TK * DP + kk] = DNUM[static_cast<size_t>(o0) * DP + kk];
                    }
                    for (int kk = t; kk < cnt; kk += T) sTile[4 * TK * DP + kk] = DDEN[o0 + kk];
                    __syncthreads();
                    if (live) {
                      for (int ob = 0; ob < cnt; ob += 64) {
                        float dqf[DP], dkf[DP], dvf[DP];
#pragma unroll
                        for (int m = 0; m < DP; ++m) { dqf[m] = 0.f; dkf[m] = 0.f; dvf[m] = 0.f; }
                        const int o1 = min(cnt, ob + 64);
                        for (int o = ob; o < o1; ++o) {
                            const float* qo = sTile + o * DP;
                            const float* ko = sTile + TK * DP + o * DP;
                            const float* vo = sTile + 2 * TK * DP + o * DP;
                            const float* dao = sTile + 3 * TK * DP + o * DP;
                            const float dlo = sTile[4 * TK * DP + o];
                            // this node as the QUERY against key o:   dq_i += dS k_o
                            float dot = 0.f, dp = -dl;
#pragma unroll
                            for (int m = 0; m < DP; ++m) { dot += q[m] * ko[m]; dp += da[m] * vo[m]; }
                            float p = sigmoidf(dot);
                            float dsv = dp * p * (1.0f - p);
#pragma unroll
                            for (int m = 0; m < DP; ++m) dqf[m] += dsv * ko[m];
                            // this node as the KEY against query o:   dv_j += P DA_o;  dk_j += dS' q_o
                            dot = 0.f;
                            dp = -dlo;
#pragma unroll
                            for (int m = 0; m < DP; ++m) { dot += qo[m] * k[m]; dp += dao[m] * v[m]; }
                            p = sigmoidf(dot);
                            dsv = dp * p * (1.0f - p);
#pragma unroll
                            for (int m = 0; m < DP; ++m) { dvf[m] += p * dao[m]; dkf[m] += dsv * qo[m]; }
                        }
                        // 64 pairs in float32, the blocks of 64 added in float64 (the sums cancel: their terms are larger than they)
#pragma unroll
                        for (int m = 0; m < DP; ++m) {
                            dqd[m] += static_cast<double>(dqf[m]);
                            dkd[m] += static_cast<double>(dkf[m]);
                            dvd[m] += static_cast<double>(dvf[m]);
                        }
                      }
                    }
                }
#pragma unroll
                for (int m = 0; m < DP; ++m) {
                    dq[m] = static_cast<float>(dqd[m]);
                    dk[m] = static_cast<float>(dkd[m]);
                    dv[m] = static_cast<float>(dvd[m]);
                }
            } else if (live) {
                load_row<DP>(DQ + static_cast<size_t>(i) * DP, dq);
#pragma unroll
                for (int m = 0; m < DP; ++m) {
                    dq[m] += 2.0f * gq2 * q[m];
                    float acc = 0.f;
#pragma unroll
                    for (int dd = 0; dd < DP; ++dd) acc += sG[m * DP + dd] * v[dd];
                    dk[m] = acc + sG[DP * DP + m] + 2.0f * gk2 * k[m];
                }
#pragma unroll
                for (int dd = 0; dd < DP; ++dd) {
                    float acc = 0.f;
#pragma unroll
                    for (int m = 0; m < DP; ++m) acc += sG[m * DP + dd] * k[m];
                    dv[dd] = acc + sG[DP * DP + DP + dd];
                }
            }
            if (live) {
                if (a.use_graph) {                      // adjoint of the aggregation: entries of the TRANSPOSED CSR, edge order
                    const int e1 = a.rowptr[i + 1];
                    for (int e = a.rowptr[i]; e < e1; ++e) {
                        const float w = a.val[e];
                        float gr[DP];
                        load_row<DP>(SG + static_cast<size_t>(a.nbr[e]) * DP, gr);
#pragma unroll
                        for (int m = 0; m < DP; ++m) dv[m] += w * gr[m];
                    }
                }
                // padded columns carry nothing
#pragma unroll
                for (int m = 0; m < DP; ++m)
                    if (m >= d) { dq[m] = 0.f; dk[m] = 0.f; dv[m] = 0.f; }
                store_row<DP>(DQ + static_cast<size_t>(i) * DP, dq);
                store_row<DP>(DK + static_cast<size_t>(i) * DP, dk);
                store_row<DP>(DV + static_cast<size_t>(i) * DP, dv);
                float dh[DP];
                load_row<DP>(DIR + static_cast<size_t>(i) * DP, dh);
                matvec_t_add<DP>(sL.wq, dq, dh);
                matvec_t_add<DP>(sL.wk, dk, dh);
                if (a.use_weight) matvec_t_add<DP>(sL.wv, dv, dh);
                else {
#pragma unroll
                    for (int m = 0; m < DP; ++m) dh[m] += dv[m];
                }
                store_row<DP>(DH + static_cast<size_t>(i) * DP, dh);
            }
        }
        __syncthreads();
        // ---- phase 4: weight gradients ----
        for (int which = 0; which < (a.use_weight ? 3 : 2); ++which) {
            const float* G = which == 0 ? DQ : (which == 1 ? DK : DV);
            float* gw = which == 0 ? a.lg[l].wq : (which == 1 ? a.lg[l].wk : a.lg[l].wv);
            float* gb = which == 0 ? a.lg[l].bq : (which == 1 ? a.lg[l].bk : a.lg[l].bv);
            outer_sum(G, DP, DP, Hl, DP, DP, n, 1.f, sG, sPart);
            for (int kk = t; kk < d * d; kk += T) gw[kk] = sG[(kk / d) * DP + kk % d];
            __syncthreads();
            outer_sum(G, DP, DP, nullptr, 0, 1, n, 1.f, sVec, sPart);
            if (t < d) gb[t] = sVec[t];
            __syncthreads();
        }
    }

    // ---- input layer backward (:188-192) ----
    for (int i = t; i < n; i += T) {
        float dy[DP], z[DP];
        load_row<DP>(DH + static_cast<size_t>(i) * DP, dy);
        if (a.use_source) {
            float acc[DP];
            load_row<DP>(DX0 + static_cast<size_t>(i) * DP, acc);
#pragma unroll
            for (int m = 0; m < DP; ++m) dy[m] += acc[m];
        }
        if (drop) {
            const int64_t at = static_cast<int64_t>(i) * d;
#pragma unroll
            for (int m = 0; m < DP; ++m)
                if (m < d) dy[m] = (a.rnd[at + m] >= a.p_drop) ? dy[m] * keep : 0.f;
        }
        load_row<DP>(tp.Z + static_cast<size_t>(i) * DP, z);
        float dpre[DP];
        if (a.use_bn) {
            float mean, rstd;
            ln_stats<DP>(z, d, a.eps, mean, rstd);
            float xh[DP], m1 = 0.f, m2 = 0.f, dyx[DP];
#pragma unroll
            for (int m = 0; m < DP; ++m) {
                xh[m] = (m < d) ? (z[m] - mean) * rstd : 0.f;
                const float yv = xh[m] * sLn0w[m] + sLn0b[m];
                if (!(yv > 0.f)) dy[m] = 0.f;                                        // ReLU
                const float gw = sLn0w[m] * dy[m];
                m1 += gw;
                m2 += gw * xh[m];
                dyx[m] = dy[m] * xh[m];
            }
            m1 /= static_cast<float>(d);
            m2 /= static_cast<float>(d);
#pragma unroll
            for (int m = 0; m < DP; ++m) dpre[m] = (m < d) ? rstd * (sLn0w[m] * dy[m] - m1 - xh[m] * m2) : 0.f;
            store_row<DP>(DY + static_cast<size_t>(i) * DP, dy);
            store_row<DP>(DYX + static_cast<size_t>(i) * DP, dyx);
        } else {
#pragma unroll
            for (int m = 0; m < DP; ++m) dpre[m] = (m < d && z[m] > 0.f) ? dy[m] : 0.f;
        }
        store_row<DP>(DPRE + static_cast<size_t>(i) * DP, dpre);
        if (a.dx) {
            for (int f = 0; f < a.f_in; ++f) {
                float acc = 0.f;
#pragma unroll
                for (int m = 0; m < DP; ++m) acc += dpre[m] * sW0[m * kMaxIn + f];
                a.dx[static_cast<size_t>(i) * a.f_in + f] = acc;
            }
        }
    }
    __syncthreads();
    if (a.use_bn) {
        outer_sum(DYX, DP, DP, nullptr, 0, 1, n, 1.f, sVec, sPart);
        if (t < d) a.gln0w[t] = sVec[t];
        __syncthreads();
        outer_sum(DY, DP, DP, nullptr, 0, 1, n, 1.f, sVec, sPart);
        if (t < d) a.gln0b[t] = sVec[t];
        __syncthreads();
    }
    for (int m = 0; m < d; ++m) {                  // d W0 [m][f] = sum_i dpre[i][m] x[i][f]
        outer_sum(DPRE + m, DP, 1, a.x, a.ldx, a.f_in, n, 1.f, sW0, sPart);
        for (int f = t; f < a.f_in; f += T) a.gw0[m * a.f_in + f] = sW0[f];
        __syncthreads();
    }
    outer_sum(DPRE, DP, DP, nullptr, 0, 1, n, 1.f, sVec, sPart);
    if (t < d) a.gb0[t] = sVec[t];
}

// ======================================================================================================================
// graph preparation in one launch: gcn_conv's degree / values (:66-74) and the CSR of the adjacency and of its transpose,
// entries of a row in edge order (stable), by LSD radix sort on 4-bit digits of the row key, one workgroup per direction:
// a thread owns a contiguous chunk of the list; per-thread digit histograms (LDS, [16][512] uint16) scanned digit-major
// give every thread its write cursor per digit; walking its chunk in order keeps the sort stable.  ceil(log2 N / 4) passes.
// ======================================================================================================================
constexpr int kGraphThreads = 512;

// exclusive prefix sum over the block's threads (wave scans by shuffles + the wave totals in LDS); returns the exclusive
// prefix of `v`; every thread must call
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* sWave /*[16]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(inc, off, 64);
        if (lane >= off) inc += up;
    }
    __syncthreads();                       // sWave may still be read by the previous scan
    if (lane == 63) sWave[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += sWave[w];
    return base + inc - v;
}

__device__ void radix_pass(const uint32_t* in, uint32_t* out, const int64_t* keys64, int E, int shift, uint16_t* hist /*[16][T]*/,
                           uint32_t* sWave, int N) {
    const int T = kGraphThreads, t = threadIdx.x;
    const int c = (E + T - 1) / T;
    const int e0 = min(E, t * c), e1 = min(E, e0 + c);
    for (int k = 0; k < 16; ++k) hist[k * T + t] = 0;
    auto key_of = [&](int e) -> uint32_t {
        if (keys64) {
            int64_t kv = keys64[e];
            if (kv < 0 || kv >= N) kv = 0;                  // flagged by the counting loop
            return (static_cast<uint32_t>(kv) << 16) | static_cast<uint32_t>(e);
        }
        return in[e];
    };
    for (int e = e0; e < e1; ++e) hist[(((key_of(e) >> 16) >> shift) & 15) * T + t] += 1;
    __syncthreads();
    // exclusive scan over the 16 x T counters in digit-major order: thread t owns flat entries [16 t, 16 t + 16)
    uint32_t local = 0;
    for (int k = 0; k < 16; ++k) local += hist[16 * t + k];
    uint32_t run = block_exclusive_scan(local, sWave);
    for (int k = 0; k < 16; ++k) {
        const uint32_t cnt = hist[16 * t + k];
        hist[16 * t + k] = static_cast<uint16_t>(run);
        run += cnt;
    }
    __syncthreads();
    for (int e = e0; e < e1; ++e) {
        const uint32_t kv = key_of(e);
        const uint32_t dg = ((kv >> 16) >> shift) & 15;
        const uint32_t pos = hist[dg * T + t];
        hist[dg * T + t] = static_cast<uint16_t>(pos + 1);
        out[pos] = kv;
    }
    __syncthreads();
}

// workgroup 0: the CSR over destinations; workgroup 1: its transpose.  Both count the in-degrees themselves (the values
// need deg^-1/2 of both ends), sort their own copy of the list, and share nothing.
__global__ __launch_bounds__(kGraphThreads) void tiny_graph_kernel(const int64_t* __restrict__ ei, const float* __restrict__ w, int E,
                                                                     int N, int* rowptr, int* src, float* val, int* rowptr_t,
                                                                     int* dst_t, float* val_t, int* status, uint32_t* ws /*[4 E]*/) {
    __shared__ uint16_t hist[16 * kGraphThreads];
    __shared__ uint32_t sWave[16];
    __shared__ int sCnt[kMaxNodes];             // entries per row of THIS workgroup's CSR
    __shared__ float sDinv[kMaxNodes];          // first the in-degree counts (as int bits), then deg^-1/2
    __shared__ int sBad;
    const int T = kGraphThreads, t = threadIdx.x;
    const int direction = blockIdx.x;
    uint32_t* bufA = ws + static_cast<size_t>(direction) * 2 * E;
    uint32_t* bufB = bufA + E;
    const int64_t* keys = direction == 0 ? ei + E : ei;        // forward: filed under the destination (`col`, :65)
    const int64_t* other = direction == 0 ? ei : ei + E;
    int* rp = direction == 0 ? rowptr : rowptr_t;
    int* nb = direction == 0 ? src : dst_t;
    float* vl = direction == 0 ? val : val_t;
    int* sIn = reinterpret_cast<int*>(sDinv);
    if (t == 0) sBad = 0;
    for (int k = t; k < N; k += T) { sCnt[k] = 0; sIn[k] = 0; }
    __syncthreads();
    for (int e = t; e < E; e += T) {
        const int64_t r = ei[e], c = ei[E + e];
        const bool okr = r >= 0 && r < N, okc = c >= 0 && c < N;
        if (!okr || !okc) sBad = 1;
        const int ri = okr ? static_cast<int>(r) : 0, ci = okc ? static_cast<int>(c) : 0;
        atomicAdd(&sIn[ci], 1);
        if (direction == 1) atomicAdd(&sCnt[ri], 1);
    }
    __syncthreads();
    {
        // dinv = sqrt(1 / in-degree), float32, correctly rounded (:66-68); 0 incoming entries -> inf
        int mx = 0;
        for (int k = t; k < N; k += T) {
            const int deg = sIn[k];
            if (direction == 0) sCnt[k] = deg;
            mx = max(mx, deg);
            sDinv[k] = sqrtf(1.0f / static_cast<float>(deg));
        }
        if (direction == 0) atomicMax(&status[1], mx);
    }
    __syncthreads();
    {
        // exclusive scan of the counts -> row pointers (a thread owns up to 8 consecutive rows)
        const int per = (N + T - 1) / T;
        const int r0 = min(N, t * per), r1 = min(N, r0 + per);
        uint32_t local = 0;
        for (int r = r0; r < r1; ++r) local += sCnt[r];
        uint32_t run = block_exclusive_scan(local, sWave);
        for (int r = r0; r < r1; ++r) {
            rp[r] = static_cast<int>(run);
            run += sCnt[r];
        }
        if (t == T - 1) rp[N] = E;
    }
    if (E > 0) {
        int bits = 1;
        while ((1 << bits) < N) ++bits;
        const int passes = (bits + 3) / 4;
        const uint32_t* cur = nullptr;
        uint32_t* dst = bufA;
        for (int p = 0; p < passes; ++p) {
            radix_pass(cur, dst, p == 0 ? keys : nullptr, E, 4 * p, hist, sWave, N);
            cur = dst;
            dst = (dst == bufA) ? bufB : bufA;
        }
        for (int k = t; k < E; k += T) {
            const uint32_t kv = cur[k];
            const int e = static_cast<int>(kv & 0xffffu), grp = static_cast<int>(kv >> 16);
            int64_t o = other[e];
            if (o < 0 || o >= N) o = 0;
            const int r = direction == 0 ? static_cast<int>(o) : grp;       // source      (`row`)
            const int c = direction == 0 ? grp : static_cast<int>(o);       // destination (`col`)
            float v = w ? __fmul_rn(__fmul_rn(w[e], sDinv[c]), sDinv[r]) : __fmul_rn(sDinv[c], sDinv[r]);   // :71 / :73
            if (!isfinite(v)) v = 0.f;                                       // :74
            nb[k] = static_cast<int>(o);
            vl[k] = v;
        }
    }
    __syncthreads();
    if (t == 0 && sBad) status[0] = 1;
}

int check_cfg(const dif_tiny_cfg* cfg) {
    DIF_REQUIRE(cfg != nullptr, DIF_E_BADARG, "dif_tiny: null configuration");
    DIF_REQUIRE(cfg->n >= 1 && cfg->n <= kMaxNodes, DIF_E_SHAPE, "dif_tiny: 1 <= n <= %d nodes (got %d)", kMaxNodes, cfg->n);
    DIF_REQUIRE(cfg->hidden >= 1 && cfg->hidden <= 8, DIF_E_SHAPE, "dif_tiny: hidden width 1..8 (got %d)", cfg->hidden);
    DIF_REQUIRE(cfg->in_channels >= 1 && cfg->in_channels <= kMaxIn, DIF_E_SHAPE, "dif_tiny: 1..%d input features (got %d)", kMaxIn,
                cfg->in_channels);
    DIF_REQUIRE(cfg->out_channels >= 1 && cfg->out_channels <= kMaxOut, DIF_E_SHAPE, "dif_tiny: 1..%d outputs (got %d)", kMaxOut,
                cfg->out_channels);
    DIF_REQUIRE(cfg->num_layers >= 1 && cfg->num_layers <= kMaxLayers, DIF_E_SHAPE, "dif_tiny: 1..%d layers (got %d)", kMaxLayers,
                cfg->num_layers);
    DIF_REQUIRE(cfg->kernel == 0 || cfg->kernel == 1, DIF_E_BADARG, "dif_tiny: kernel 0 (simple) or 1 (sigmoid)");
    DIF_REQUIRE(cfg->dropout >= 0.f && cfg->dropout < 1.f, DIF_E_BADARG, "dif_tiny: dropout in [0, 1)");
    DIF_REQUIRE(cfg->launch_plan >= 0 && cfg->launch_plan <= 2, DIF_E_BADARG, "dif_tiny: launch_plan 0 (by size), 1 (one workgroup) or 2 (grid)");
    return 0;
}

// One workgroup for the whole model, or one launch per layer stage over the chip (tiny_sigmoid_grid.hip / tiny_simple_grid.hip).
// `sigmoid` (the n^2 pairs of a layer), measured on MI355X (profiles/r06_tiny_sigmoid_grid.txt): device time crosses at ~64 nodes
// (49 us against 48 per training snapshot; 72 against 225 at 129 nodes, 90 against 3,350 at 1,068); in the launch-bound training
// loop of spatial-temporal/main.py the 20-node dataset is 7 % faster on one workgroup (2 launches, not 7), the 129-node one 13 %
// faster on the grid.  `simple` has no pair loop: per training snapshot one workgroup takes 100 us at 129 nodes, 116 at 256, 300 at 1,068
// against 72 / 75 / 88 on the grid (7 launches instead of 2); the launch-bound loop at 129 nodes is 8 % FASTER on one workgroup,
// wikimath's (1,068 nodes, graph term) 24 % faster on the grid (profiles/r06_tiny_sigmoid_grid.txt section 4).
constexpr int kGridFromNodes = 64;
constexpr int kGridFromNodesSimple = 256;
bool grid_plan(const dif_tiny_cfg* cfg) {
    if (cfg->launch_plan == 1) return false;
    if (cfg->launch_plan == 2) return true;
    return cfg->n > (cfg->kernel == 1 ? kGridFromNodes : kGridFromNodesSimple);
}

// params / grads: 6 + 8 * layers pointers in the order
//   fcs.0.weight, fcs.0.bias, bns.0.weight, bns.0.bias, fcs.1.weight, fcs.1.bias,
//   then per layer: Wk.weight, Wk.bias, Wq.weight, Wq.bias, Wv.weight, Wv.bias, bns.{l+1}.weight, bns.{l+1}.bias
int fill(TinyArgs& a, const dif_tiny_cfg* cfg, const float* x, int64_t ldx, const void* const* params) {
    a.n = cfg->n; a.f_in = cfg->in_channels; a.d = cfg->hidden; a.c = cfg->out_channels; a.layers = cfg->num_layers;
    a.sigmoid = cfg->kernel; a.use_bn = cfg->use_bn; a.residual = cfg->use_residual; a.use_weight = cfg->use_weight;
    a.use_graph = cfg->use_graph; a.use_source = cfg->use_source; a.training = cfg->training;
    a.alpha = cfg->alpha; a.a_s = cfg->attn_scale; a.g_s = cfg->gcn_scale; a.p_drop = cfg->dropout; a.eps = cfg->eps;
    a.x = x; a.ldx = ldx;
    DIF_REQUIRE(x != nullptr && params != nullptr && ldx >= cfg->in_channels, DIF_E_BADARG, "dif_tiny: null x / params or ldx < in_channels");
    auto P = [&](int k) { return static_cast<const float*>(params[k]); };
    a.w0 = P(0); a.b0 = P(1); a.ln0w = P(2); a.ln0b = P(3); a.wo = P(4); a.bo = P(5);
    DIF_REQUIRE(a.w0 && a.b0 && a.wo && a.bo, DIF_E_BADARG, "dif_tiny: null Linear parameter");
    DIF_REQUIRE(!cfg->use_bn || (a.ln0w && a.ln0b), DIF_E_BADARG, "dif_tiny: use_bn without LayerNorm parameters");
    for (int l = 0; l < cfg->num_layers; ++l) {
        LayerPtrs& p = a.lp[l];
        p.wk = P(6 + 8 * l); p.bk = P(7 + 8 * l); p.wq = P(8 + 8 * l); p.bq = P(9 + 8 * l);
        p.wv = P(10 + 8 * l); p.bv = P(11 + 8 * l); p.lnw = P(12 + 8 * l); p.lnb = P(13 + 8 * l);
        DIF_REQUIRE(p.wk && p.bk && p.wq && p.bq, DIF_E_BADARG, "dif_tiny: null Wq / Wk of layer %d", l);
        DIF_REQUIRE(!cfg->use_weight || (p.wv && p.bv), DIF_E_BADARG, "dif_tiny: use_weight without Wv of layer %d", l);
        DIF_REQUIRE(!cfg->use_bn || (p.lnw && p.lnb), DIF_E_BADARG, "dif_tiny: use_bn without LayerNorm of layer %d", l);
    }
    return 0;
}

int block_threads(int n) {
    int t = (n + 63) / 64 * 64;
    return t < 64 ? 64 : (t > 512 ? 512 : t);     // 512: two waves per SIMD, 256 VGPRs each (no spills at hidden 8)
}

}  // namespace

extern "C" size_t dif_tiny_tape_floats(int n, int hidden, int num_layers) {
    return tape_floats(n, hidden <= 4 ? 4 : 8, num_layers);
}

extern "C" size_t dif_tiny_scratch_floats(int n, int hidden, int num_layers) { return scratch_floats(n, hidden <= 4 ? 4 : 8, num_layers); }

extern "C" size_t dif_tiny_graph_workspace_bytes(int64_t E, int64_t N) {
    (void)N;
    return 4 * static_cast<size_t>(E) * 4 + 16;
}

extern "C" int dif_tiny_graph_build(const int64_t* edge_index, const float* edge_weight, int64_t E, int64_t N, int32_t* rowptr,
                                    int32_t* src, float* val, int32_t* rowptr_t, int32_t* dst_t, float* val_t, int32_t* status,
                                    void* workspace, size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(N >= 1 && N <= kMaxNodes && E >= 0 && E <= kMaxEdges, DIF_E_SHAPE,
                "dif_tiny_graph_build: N <= %d nodes and E <= %d edges (got %lld, %lld)", kMaxNodes, kMaxEdges,
                static_cast<long long>(N), static_cast<long long>(E));
    DIF_REQUIRE((E == 0 || edge_index) && rowptr && rowptr_t && status && workspace, DIF_E_BADARG, "dif_tiny_graph_build: null pointer");
    DIF_REQUIRE(E == 0 || (src && val && dst_t && val_t), DIF_E_BADARG, "dif_tiny_graph_build: null output");
    DIF_REQUIRE(workspace_bytes >= dif_tiny_graph_workspace_bytes(E, N), DIF_E_WORKSPACE, "dif_tiny_graph_build: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(status, 0, 2 * sizeof(int32_t), st);
    if (e != hipSuccess) return dif::fail(static_cast<int>(e), "dif_tiny_graph_build: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(tiny_graph_kernel, dim3(2), dim3(kGraphThreads), 0, st, edge_index, edge_weight, static_cast<int>(E), static_cast<int>(N),
                       rowptr, src, val, rowptr_t, dst_t, val_t, status, static_cast<uint32_t*>(workspace));
    return dif::launch_status("dif_tiny_graph_build");
}

extern "C" int dif_tiny_forward_f32(const dif_tiny_cfg* cfg, const float* x, int64_t ldx, const void* const* params,
                                    const int32_t* rowptr, const int32_t* src, const float* val, const float* rnd, float* tape,
                                    float* y, dif_stream_t stream) {
    if (int rc = check_cfg(cfg)) return rc;
    TinyArgs a{};
    if (int rc = fill(a, cfg, x, ldx, params)) return rc;
    DIF_REQUIRE(tape && y, DIF_E_BADARG, "dif_tiny_forward_f32: null tape / y");
    DIF_REQUIRE(dif::aligned16(tape), DIF_E_BADARG, "dif_tiny_forward_f32: tape must be 16-byte aligned");
    DIF_REQUIRE(!cfg->use_graph || (rowptr && (src || cfg->nnz == 0) && (val || cfg->nnz == 0)), DIF_E_BADARG,
                "dif_tiny_forward_f32: use_graph without a CSR");
    a.rowptr = rowptr; a.nbr = src; a.val = val; a.rnd = rnd; a.tape = tape; a.y = y;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (grid_plan(cfg)) return cfg->kernel == 1 ? grid_sigmoid_forward(a, st) : grid_simple_forward(a, st);
    const int T = block_threads(cfg->n);
    if (cfg->hidden <= 4) hipLaunchKernelGGL(tiny_forward_kernel<4>, dim3(1), dim3(T), 0, st, a);
    else hipLaunchKernelGGL(tiny_forward_kernel<8>, dim3(1), dim3(T), 0, st, a);
    return dif::launch_status("dif_tiny_forward_f32");
}

extern "C" int dif_tiny_backward_f32(const dif_tiny_cfg* cfg, const float* x, int64_t ldx, const void* const* params,
                                     const int32_t* rowptr_t, const int32_t* dst_t, const float* val_t, const float* rnd,
                                     float* tape, const float* grad_y, void* const* grads, float* dx, float* scratch,
                                     dif_stream_t stream) {
    if (int rc = check_cfg(cfg)) return rc;
    TinyArgs a{};
    if (int rc = fill(a, cfg, x, ldx, params)) return rc;
    DIF_REQUIRE(tape && grad_y && grads && scratch, DIF_E_BADARG, "dif_tiny_backward_f32: null tape / grad_y / grads / scratch");
    DIF_REQUIRE(dif::aligned16(tape) && dif::aligned16(scratch), DIF_E_BADARG, "dif_tiny_backward_f32: tape / scratch must be 16-byte aligned");
    DIF_REQUIRE(!cfg->use_graph || (rowptr_t && (dst_t || cfg->nnz == 0) && (val_t || cfg->nnz == 0)), DIF_E_BADARG,
                "dif_tiny_backward_f32: use_graph without the transposed CSR");
    a.rowptr = rowptr_t; a.nbr = dst_t; a.val = val_t; a.rnd = rnd; a.tape = tape; a.gy = grad_y; a.dx = dx; a.scratch = scratch;
    auto G = [&](int k) { return static_cast<float*>(grads[k]); };
    a.gw0 = G(0); a.gb0 = G(1); a.gln0w = G(2); a.gln0b = G(3); a.gwo = G(4); a.gbo = G(5);
    DIF_REQUIRE(a.gw0 && a.gb0 && a.gwo && a.gbo && (!cfg->use_bn || (a.gln0w && a.gln0b)), DIF_E_BADARG, "dif_tiny_backward_f32: null gradient buffer");
    for (int l = 0; l < cfg->num_layers; ++l) {
        LayerGrads& g = a.lg[l];
        g.wk = G(6 + 8 * l); g.bk = G(7 + 8 * l); g.wq = G(8 + 8 * l); g.bq = G(9 + 8 * l);
        g.wv = G(10 + 8 * l); g.bv = G(11 + 8 * l); g.lnw = G(12 + 8 * l); g.lnb = G(13 + 8 * l);
        DIF_REQUIRE(g.wk && g.bk && g.wq && g.bq && (!cfg->use_weight || (g.wv && g.bv)) && (!cfg->use_bn || (g.lnw && g.lnb)),
                    DIF_E_BADARG, "dif_tiny_backward_f32: null gradient buffer of layer %d", l);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (grid_plan(cfg)) return cfg->kernel == 1 ? grid_sigmoid_backward(a, st) : grid_simple_backward(a, st);
    const int T = block_threads(cfg->n);
    if (cfg->hidden <= 4) hipLaunchKernelGGL(tiny_backward_kernel<4>, dim3(1), dim3(T), 0, st, a);
    else hipLaunchKernelGGL(tiny_backward_kernel<8>, dim3(1), dim3(T), 0, st, a);
    return dif::launch_status("dif_tiny_backward_f32");
}

// The whole model on a tiny graph with the `simple` kernel, spread over the chip (the counterpart of tiny_sigmoid_grid.hip):
// spatial-temporal/run.sh:33-37 trains `simple` on wikimath's 1,068 nodes, where the one-workgroup kernel of tiny_model.hip walks
// three slots of nodes per phase and ~13 dependent edge loads per node on ONE compute unit (97 + 188 us per training snapshot).
//
// With `simple` (node classification/difformer.py:18-39) the nodes meet in five sums per layer -- K^T V, sum k, sum v, |Q|^2,
// |K|^2 -- and the backward in four -- d K^T V, d sum k, d sum v, d s.  A launch is one layer's stage for all nodes, workgroup =
// 64 nodes (a thread per node) + the threads that add the block's share of the NEXT sums (<= 82 values, float64, rows in node
// order) into memory; the next launch's workgroups each add the blocks' shares in block order (every workgroup the same
// float64 sum: bitwise reproducible, no atomics) and go on:
//     forward   L + 1 launches:  [input layer, projections 0, sums 0] , [attention l, aggregation, tail, projections l + 1, sums l + 1 | output] x L
//     backward  L + 2 launches:  [output Linear, tail L-1, attention set-up, sums] , [attention l, projections l, tail l - 1 ... | input layer] x L ,
//                                [every parameter gradient: tiny::grid_sums]
// Same arithmetic as the one-workgroup kernel (closed form of the attention and of its backward), same tape, scratch and C calls.
#include "tiny_grid.h"

using namespace tiny;

namespace {

constexpr int kThreads = 128;                   // 64 node threads + the sum threads (<= 82 sums)

template <int DP>
__host__ __device__ constexpr int fwd_sums() { return DP * DP + 2 * DP + 2; }       // K^T V | sum k | sum v | |Q|^2 | |K|^2
template <int DP>
__host__ __device__ constexpr int bwd_sums() { return DP * DP + 2 * DP + 1; }       // d K^T V | d sum k | d sum v | d s

// the blocks' shares [G][96] of one set -> sOut[0 .. NS) (float), every workgroup alike; ends with __syncthreads()
__device__ __forceinline__ void add_block_sums(const double* shares, int G, int NS, float* sOut) {
    const int t = threadIdx.x;
    if (t < NS) {
        double s = 0.0;
        int g = 0;
        for (; g + 8 <= G; g += 8) {                   // eight loads in flight, added in block order
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = shares[static_cast<size_t>(g + u) * 96 + t];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; g < G; ++g) s += shares[static_cast<size_t>(g) * 96 + t];
        sOut[t] = static_cast<float>(s);
    }
    __syncthreads();
}

// ======================================================================================================================
// forward
// ======================================================================================================================
template <int DP>
__global__ __launch_bounds__(kThreads) void grid_simple_forward_kernel(const TinyArgs a, const int stage) {
    __shared__ float sW0[DP * kMaxIn];
    __shared__ float sB0[DP], sLn0w[DP], sLn0b[DP];
    __shared__ LayerW<DP> sL, sNext;
    __shared__ float sWo[kMaxOut * DP], sBo[kMaxOut];
    __shared__ float sSum[96];
    __shared__ float sRows[kNodes][3 * DP];         // q | k | v of the block's nodes (zero rows beyond n)
    const int n = a.n, d = a.d, L = a.layers, T = blockDim.x, t = threadIdx.x;
    const int i = blockIdx.x * kNodes + t;
    const bool node = t < kNodes, live = node && i < n;
    const int G = gridDim.x;
    Tape<DP> tp(a.tape, n, L);
    const size_t nd = static_cast<size_t>(n) * DP;
    const bool drop = a.training && a.rnd != nullptr && a.p_drop > 0.f;
    double* shares = tp.partials();                                       // [2 sets][G][96]
    float h[DP], q[DP], k[DP], v[DP];
#pragma unroll
    for (int m = 0; m < DP; ++m) { h[m] = 0.f; q[m] = 0.f; k[m] = 0.f; v[m] = 0.f; }
    int next_set = 0;

    if (stage == 0) {
        for (int kk = t; kk < DP * kMaxIn; kk += T) {
            const int m = kk / kMaxIn, f = kk % kMaxIn;
            sW0[kk] = (m < d && f < a.f_in) ? a.w0[m * a.f_in + f] : 0.f;
        }
        for (int kk = t; kk < DP; kk += T) {
            sB0[kk] = kk < d ? a.b0[kk] : 0.f;
            sLn0w[kk] = (kk < d && a.use_bn) ? a.ln0w[kk] : 0.f;
            sLn0b[kk] = (kk < d && a.use_bn) ? a.ln0b[kk] : 0.f;
        }
        load_layer<DP>(sNext, a.lp[0], d, a.use_weight, a.use_bn);
        __syncthreads();
        if (live) {                                                        // input layer (:188-192)
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
    } else {
        const int l = stage - 1;
        const bool last = l + 1 == L;
        load_layer<DP>(sL, a.lp[l], d, a.use_weight, a.use_bn);
        if (!last) load_layer<DP>(sNext, a.lp[l + 1], d, a.use_weight, a.use_bn);
        else {
            for (int kk = t; kk < kMaxOut * DP; kk += T) {
                const int c = kk / DP, m = kk % DP;
                sWo[kk] = (c < a.c && m < d) ? a.wo[c * d + m] : 0.f;
            }
            for (int kk = t; kk < kMaxOut; kk += T) sBo[kk] = kk < a.c ? a.bo[kk] : 0.f;
        }
        add_block_sums(shares + static_cast<size_t>(l & 1) * G * 96, G, fwd_sums<DP>(), sSum);      // (syncs)
        if (blockIdx.x == 0 && t < fwd_sums<DP>()) tp.SUM[l * 96 + t] = sSum[t];                      // the backward reads them
        const float* set = tp.qkv(l & 1);
        const float* Vl = set + 2 * nd;
        next_set = (l + 1) & 1;
        if (live) {
            load_row<DP>(set + static_cast<size_t>(i) * DP, q);
            // :20-38  closed form of the attention from the layer's sums
            const float q2 = sSum[DP * DP + 2 * DP], k2 = sSum[DP * DP + 2 * DP + 1];
            const float s = 1.0f / (sqrtf(q2) * sqrtf(k2));
            float den = 0.f, att[DP];
#pragma unroll
            for (int m = 0; m < DP; ++m) den += q[m] * sSum[DP * DP + m];
            den = s * den + static_cast<float>(n);
#pragma unroll
            for (int dd = 0; dd < DP; ++dd) {
                float num = 0.f;
#pragma unroll
                for (int m = 0; m < DP; ++m) num += q[m] * sSum[m * DP + dd];
                att[dd] = (s * num + sSum[DP * DP + DP + dd]) / den;
            }
            store_row<DP>(tp.ATT + (static_cast<size_t>(l) * n + i) * DP, att);
            float out[DP];
            if (a.use_graph) {
                float g[DP];
#pragma unroll
                for (int m = 0; m < DP; ++m) g[m] = 0.f;
                const int e1 = a.rowptr[i + 1];
                for (int e = a.rowptr[i]; e < e1; ++e) {                              // :75-78, entries in edge order
                    const float wgt = a.val[e];
                    float vr[DP];
                    load_row<DP>(Vl + static_cast<size_t>(a.nbr[e]) * DP, vr);
#pragma unroll
                    for (int m = 0; m < DP; ++m) g[m] += wgt * vr[m];
                }
#pragma unroll
                for (int m = 0; m < DP; ++m) out[m] = a.a_s * att[m] + a.g_s * g[m];  // :130-134
            } else {
#pragma unroll
                for (int m = 0; m < DP; ++m) out[m] = att[m];
            }
            float hl[DP];
            load_row<DP>(tp.H + (static_cast<size_t>(l) * n + i) * DP, hl);
            if (a.use_source) {                                                       // :139-140
                float x0[DP];
                load_row<DP>(tp.H + static_cast<size_t>(i) * DP, x0);
#pragma unroll
                for (int m = 0; m < DP; ++m) out[m] += x0[m];
            }
            if (a.residual) {                                                         // :200-201
#pragma unroll
                for (int m = 0; m < DP; ++m) out[m] = a.alpha * out[m] + (1.0f - a.alpha) * hl[m];
            }
            store_row<DP>(tp.Z + (static_cast<size_t>(l + 1) * n + i) * DP, out);
            if (a.use_bn) {                                                           // :202-203
                float mean, rstd;
                ln_stats<DP>(out, d, a.eps, mean, rstd);
#pragma unroll
                for (int m = 0; m < DP; ++m) out[m] = (m < d) ? (out[m] - mean) * rstd * sL.lnw[m] + sL.lnb[m] : 0.f;
            }
            if (drop) dropout_row<DP>(out, a.rnd, (static_cast<int64_t>(l + 1) * n + i) * d, d, a.p_drop);   // :204
            store_row<DP>(tp.H + (static_cast<size_t>(l + 1) * n + i) * DP, out);
#pragma unroll
            for (int m = 0; m < DP; ++m) h[m] = out[m];
            if (last) {
                for (int c = 0; c < a.c; ++c) {                                       // :208
                    float acc = sBo[c];
#pragma unroll
                    for (int m = 0; m < DP; ++m) acc += sWo[c * DP + m] * out[m];
                    a.y[static_cast<size_t>(i) * a.c + c] = acc;
                }
            }
        }
        if (last) return;
    }
    // ---- projections of the next layer (:115-120) and this block's share of its sums ----
    if (live) {
        project<DP>(sNext, a.use_weight, h, q, k, v);
        float* nxt = tp.qkv(next_set);
        store_row<DP>(nxt + static_cast<size_t>(i) * DP, q);
        store_row<DP>(nxt + nd + static_cast<size_t>(i) * DP, k);
        store_row<DP>(nxt + 2 * nd + static_cast<size_t>(i) * DP, v);
    }
    if (node) {
#pragma unroll
        for (int m = 0; m < DP; ++m) {
            sRows[t][m] = live ? q[m] : 0.f;
            sRows[t][DP + m] = live ? k[m] : 0.f;
            sRows[t][2 * DP + m] = live ? v[m] : 0.f;
        }
    }
    __syncthreads();
    if (t < fwd_sums<DP>()) {
        double s = 0.0;
        if (t < DP * DP) {
            const int m = t / DP, c = t % DP;
            for (int r = 0; r < kNodes; ++r) s += static_cast<double>(sRows[r][DP + m]) * static_cast<double>(sRows[r][2 * DP + c]);
        } else if (t < DP * DP + 2 * DP) {
            const int c = DP + (t - DP * DP);                                          // k columns, then v columns
            for (int r = 0; r < kNodes; ++r) s += static_cast<double>(sRows[r][c]);
        } else {
            const int c0 = (t == DP * DP + 2 * DP) ? 0 : DP;                           // |Q|^2, |K|^2
            for (int r = 0; r < kNodes; ++r) {
#pragma unroll
                for (int m = 0; m < DP; ++m) s += static_cast<double>(sRows[r][c0 + m]) * static_cast<double>(sRows[r][c0 + m]);
            }
        }
        shares[(static_cast<size_t>(next_set) * G + blockIdx.x) * 96 + t] = s;
    }
}

// ======================================================================================================================
// backward
// ======================================================================================================================
// tail_backward_common, then the attention's set-up for node i of layer l (the one-workgroup kernel's phase 1, `simple`):
// d num, d den, the part of d q that does not wait for the sums, t_s; -> the row the block's sums are made of
template <int DP>
__device__ __forceinline__ void tail_backward(const TinyArgs& a, const Tape<DP>& tp, const GridScratch<DP>& gs, const LayerW<DP>& w,
                                              const float* sSum, int l, int i, float (&dy)[DP], bool drop, float keep,
                                              float (&row)[2 * DP + 2]) {
    const int n = a.n;
    const size_t at_i = static_cast<size_t>(i) * DP;
    float datt[DP], q[DP], k[DP], v[DP];
    tail_backward_common<DP>(a, tp, gs, w, l, i, dy, drop, keep, datt, q, k, v);
    const float q2 = sSum[DP * DP + 2 * DP], k2 = sSum[DP * DP + 2 * DP + 1];
    const float s = 1.0f / (sqrtf(q2) * sqrtf(k2));
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
    store_row<DP>(gs.per_layer(l, kDQ) + at_i, dq);            // completed by the next launch (+ 2 g_q2 q)
#pragma unroll
    for (int m = 0; m < DP; ++m) { row[m] = q[m]; row[DP + m] = dnum[m]; }
    row[2 * DP] = dden;
    row[2 * DP + 1] = ts;
}

template <int DP>
__global__ __launch_bounds__(kThreads) void grid_simple_backward_kernel(const TinyArgs a, const int stage) {
    __shared__ float sW0[DP * kMaxIn];
    __shared__ float sLn0w[DP], sLn0b[DP];
    __shared__ LayerW<DP> sL, sPrev;
    __shared__ float sWo[kMaxOut * DP];
    __shared__ float sSum[96], sSumPrev[96], sG[96];
    __shared__ float sRows[kNodes][2 * DP + 2];     // q | d num | d den | t_s of the block's nodes
    const int n = a.n, d = a.d, L = a.layers, T = blockDim.x, t = threadIdx.x;
    const int i = blockIdx.x * kNodes + t;
    const bool node = t < kNodes, live = node && i < n;
    const int G = gridDim.x;
    Tape<DP> tp(a.tape, n, L);
    GridScratch<DP> gs(a.scratch, n, L);
    const size_t at_i = static_cast<size_t>(i) * DP;
    const bool drop = a.training && a.rnd != nullptr && a.p_drop > 0.f;
    const float keep = drop ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    double* shares = gs.partials();                                       // [2 sets][G][96]
    float row[2 * DP + 2];
#pragma unroll
    for (int m = 0; m < 2 * DP + 2; ++m) row[m] = 0.f;
    int lp = 0;                                                            // the layer whose set-up this launch ends with

    if (stage == 0) {
        lp = L - 1;
        for (int kk = t; kk < kMaxOut * DP; kk += T) {
            const int c = kk / DP, m = kk % DP;
            sWo[kk] = (c < a.c && m < d) ? a.wo[c * d + m] : 0.f;
        }
        load_layer<DP>(sPrev, a.lp[lp], d, a.use_weight, a.use_bn);
        for (int kk = t; kk < fwd_sums<DP>(); kk += T) sSumPrev[kk] = tp.SUM[lp * 96 + kk];
        __syncthreads();
        if (live) {
            float g[DP];
#pragma unroll
            for (int m = 0; m < DP; ++m) g[m] = 0.f;
            if (a.use_source) store_row<DP>(gs.DX0 + at_i, g);
            for (int c = 0; c < a.c; ++c) {
                const float gv = a.gy[static_cast<size_t>(i) * a.c + c];
#pragma unroll
                for (int m = 0; m < DP; ++m) g[m] += gv * sWo[c * DP + m];
            }
            tail_backward<DP>(a, tp, gs, sPrev, sSumPrev, lp, i, g, drop, keep, row);
        }
    } else {
        const int l = L - stage;
        lp = l - 1;
        load_layer<DP>(sL, a.lp[l], d, a.use_weight, a.use_bn);
        for (int kk = t; kk < fwd_sums<DP>(); kk += T) sSum[kk] = tp.SUM[l * 96 + kk];
        if (l > 0) {
            load_layer<DP>(sPrev, a.lp[lp], d, a.use_weight, a.use_bn);
            for (int kk = t; kk < fwd_sums<DP>(); kk += T) sSumPrev[kk] = tp.SUM[lp * 96 + kk];
        } else {
            for (int kk = t; kk < DP * kMaxIn; kk += T) {
                const int m = kk / kMaxIn, f = kk % kMaxIn;
                sW0[kk] = (m < d && f < a.f_in) ? a.w0[m * a.f_in + f] : 0.f;
            }
            for (int kk = t; kk < DP; kk += T) {
                sLn0w[kk] = (kk < d && a.use_bn) ? a.ln0w[kk] : 0.f;
                sLn0b[kk] = (kk < d && a.use_bn) ? a.ln0b[kk] : 0.f;
            }
        }
        add_block_sums(shares + static_cast<size_t>(l & 1) * G * 96, G, bwd_sums<DP>(), sG);       // (syncs)
        const int set = l & 1;
        if (live) {
            // ---- the attention's backward in closed form (the one-workgroup kernel's phases 2-3, `simple`) ----
            const float q2 = sSum[DP * DP + 2 * DP], k2 = sSum[DP * DP + 2 * DP + 1];
            const float s = 1.0f / (sqrtf(q2) * sqrtf(k2));
            const float ds = sG[DP * DP + 2 * DP];
            const float gq2 = -0.5f * s * ds / q2, gk2 = -0.5f * s * ds / k2;                // s = q2^-1/2 k2^-1/2
            float q[DP], k[DP], v[DP], dq[DP], dk[DP], dv[DP];
            load_row<DP>(gs.in_set(set, kQ) + at_i, q);
            load_row<DP>(gs.in_set(set, kK) + at_i, k);
            load_row<DP>(gs.in_set(set, kV) + at_i, v);
            load_row<DP>(gs.per_layer(l, kDQ) + at_i, dq);
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
            if (a.use_graph) {                      // adjoint of the aggregation: entries of the TRANSPOSED CSR, edge order
                const float* SG = gs.in_set(set, kSG);
                const int e1 = a.rowptr[i + 1];
                for (int e = a.rowptr[i]; e < e1; ++e) {
                    const float wgt = a.val[e];
                    float gr[DP];
                    load_row<DP>(SG + static_cast<size_t>(a.nbr[e]) * DP, gr);
#pragma unroll
                    for (int m = 0; m < DP; ++m) dv[m] += wgt * gr[m];
                }
            }
#pragma unroll
            for (int m = 0; m < DP; ++m)
                if (m >= d) { dq[m] = 0.f; dk[m] = 0.f; dv[m] = 0.f; }
            store_row<DP>(gs.per_layer(l, kDQ) + at_i, dq);
            store_row<DP>(gs.per_layer(l, kDK) + at_i, dk);
            store_row<DP>(gs.per_layer(l, kDV) + at_i, dv);
            float dh[DP];
            load_row<DP>(gs.DIR + at_i, dh);
            matvec_t_add<DP>(sL.wq, dq, dh);
            matvec_t_add<DP>(sL.wk, dk, dh);
            if (a.use_weight) matvec_t_add<DP>(sL.wv, dv, dh);
            else {
#pragma unroll
                for (int m = 0; m < DP; ++m) dh[m] += dv[m];
            }
            if (l > 0) tail_backward<DP>(a, tp, gs, sPrev, sSumPrev, lp, i, dh, drop, keep, row);
            else input_backward<DP>(a, tp, gs, sW0, sLn0w, sLn0b, i, dh, drop, keep);
        }
        if (l == 0) return;
    }
    // ---- this block's share of layer lp's backward sums: d K^T V = s sum q (x) d num, d sum k = s sum d den q, d sum v, d s ----
    if (node) {
#pragma unroll
        for (int m = 0; m < 2 * DP + 2; ++m) sRows[t][m] = row[m];
    }
    __syncthreads();
    if (t < bwd_sums<DP>()) {
        const float q2 = sSumPrev[DP * DP + 2 * DP], k2 = sSumPrev[DP * DP + 2 * DP + 1];
        const double s = 1.0 / (static_cast<double>(sqrtf(q2)) * static_cast<double>(sqrtf(k2)));
        double acc = 0.0;
        if (t < DP * DP) {
            const int m = t / DP, dd = t % DP;
            for (int r = 0; r < kNodes; ++r) acc += static_cast<double>(sRows[r][m]) * static_cast<double>(sRows[r][DP + dd]);
            acc *= s;
        } else if (t < DP * DP + DP) {
            const int m = t - DP * DP;
            for (int r = 0; r < kNodes; ++r) acc += static_cast<double>(sRows[r][2 * DP]) * static_cast<double>(sRows[r][m]);
            acc *= s;
        } else if (t < DP * DP + 2 * DP) {
            const int dd = t - DP * DP - DP;
            for (int r = 0; r < kNodes; ++r) acc += static_cast<double>(sRows[r][DP + dd]);
        } else {
            for (int r = 0; r < kNodes; ++r) acc += static_cast<double>(sRows[r][2 * DP + 1]);
        }
        shares[(static_cast<size_t>(lp & 1) * G + blockIdx.x) * 96 + t] = acc;
    }
}

template <int DP>
int forward_launches(const TinyArgs& a, hipStream_t st) {
    const int G = (a.n + kNodes - 1) / kNodes;
    for (int s = 0; s <= a.layers; ++s) hipLaunchKernelGGL(grid_simple_forward_kernel<DP>, dim3(G), dim3(kThreads), 0, st, a, s);
    return dif::launch_status("dif_tiny_forward_f32");
}

template <int DP>
int backward_launches(const TinyArgs& a, hipStream_t st) {
    const int G = (a.n + kNodes - 1) / kNodes;
    for (int s = 0; s <= a.layers; ++s) hipLaunchKernelGGL(grid_simple_backward_kernel<DP>, dim3(G), dim3(kThreads), 0, st, a, s);
    if (int rc = dif::launch_status("dif_tiny_backward_f32")) return rc;
    return grid_sums(a, st);
}

}  // namespace

int tiny::grid_simple_forward(const TinyArgs& a, hipStream_t st) {
    return a.d <= 4 ? forward_launches<4>(a, st) : forward_launches<8>(a, st);
}

int tiny::grid_simple_backward(const TinyArgs& a, hipStream_t st) {
    return a.d <= 4 ? backward_launches<4>(a, st) : backward_launches<8>(a, st);
}

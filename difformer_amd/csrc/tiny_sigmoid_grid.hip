// The whole model on a tiny graph with the `sigmoid` kernel, for MORE nodes than one workgroup sweeps in time
// (spatial-temporal/run.sh:39-43: wikimath, 1,068 nodes, `--kernel sigmoid`, with and without `--use_graph`).
//
// With `sigmoid` (node classification/difformer.py:45-56 = spatial-temporal/difformer.py:45-56) nothing in DIFFormer.forward
// (:184-209) sums over nodes except the attention itself -- LayerNorm is per row -- so a node's chain
//     Linear -> LayerNorm -> ReLU -> [ Wq / Wk / Wv | sigma(q K^T) V / row sum | gcn_conv row | residual | LayerNorm ] x L -> Linear
// only waits for the OTHER nodes' k / v rows once per layer.  The one-workgroup kernel of tiny_model.hip walks all n^2 pairs
// on one compute unit (3.6 ms per training snapshot at 1,068 nodes); here a launch is one layer's pair loop over the chip,
//     workgroup = 64 nodes (lane = node) x 8 waves (wave = a contiguous chunk of the other nodes, rows fetched by scalar loads:
//                 the addresses are wave-uniform),
// with the per-node work before and after it done by the workgroup's first wave, and the launch boundary as the barrier:
//     forward   L + 1 launches:  [input layer, projections 0] , [pairs l, tail l, projections l + 1 | output Linear] x L
//     backward  L + 2 launches:  [output Linear, tail l = L-1] , [pairs l, projections l, tail l - 1 | input layer] x L ,
//                                [every sum over nodes: one workgroup per parameter gradient]
// Beyond 512 nodes the keys of a node block are split over K workgroups (~2,000 waves a launch: two per SIMD) whose partial
// sums meet in a second, 64-thread launch per layer ([pairs l] , [tail l ...]: 2 L + 1 / 2 L + 2 launches).
// The q / k / v rows (and the backward's per-node operands) alternate between two sets from layer to layer, because a workgroup
// that is done with layer l writes layer l +- 1's rows while others still read layer l's.  Partial sums: every product rounded once
// in float32, added in float64; the waves' chunks in a fixed order -- bitwise reproducible, as the one-workgroup kernels are.
// Same tape, same scratch buffer, same C entry points (dif_tiny_forward_f32 / dif_tiny_backward_f32 pick the plan).
#include "tiny_grid.h"

using namespace tiny;

namespace {

constexpr int kWaves = 8;                       // waves per workgroup in the pair launches

// 1 / (1 + e^-v) from v_exp_f32 / v_rcp_f32 (~2 ulp; the forward and the backward sweep evaluate the SAME expression, which is
// what their sums' consistency needs; the oracle bar is 1e-4)
__device__ __forceinline__ float fast_sigmoid(float v) { return __frcp_rn(1.0f + __expf(-v)); }

// A layer's stage as ONE launch (kFused: a node block's 8 waves cover all keys), or as TWO when the keys are split over K
// workgroups per node block: kPairs leaves each split's NS sums per node in `part` [K][n][NS], kTail (64 threads per node block)
// adds them in split order and goes on.  (The splits do not meet inside one launch: a device-scope release / acquire per
// workgroup writes back and invalidates the XCD's L2 -- 272 workgroups doing so cost 55 us a launch, measured.)
enum { kFused = 0, kPairs = 1, kTail = 2 };

template <int NS>
__device__ __forceinline__ void store_split(const double (&s)[NS], double* part, int n, int i) {
    double* mine = part + (static_cast<size_t>(blockIdx.y) * n + i) * NS;
#pragma unroll
    for (int m = 0; m < NS; ++m) mine[m] = s[m];
}
template <int NS>
__device__ __forceinline__ void add_splits(double (&s)[NS], const double* part, int K, int n, int i) {
#pragma unroll
    for (int m = 0; m < NS; ++m) s[m] = 0.0;
#pragma unroll 4
    for (int kb = 0; kb < K; ++kb) {
        const double* r = part + (static_cast<size_t>(kb) * n + i) * NS;
#pragma unroll
        for (int m = 0; m < NS; ++m) s[m] += r[m];
    }
}

// ======================================================================================================================
// forward: stage 0 = input layer + projections of layer 0 (64 threads per workgroup); stage l + 1 = layer l
// ======================================================================================================================
template <int DP>
__global__ __launch_bounds__(kWaves * 64) void grid_sigmoid_forward_kernel(const TinyArgs a, const int stage, const int part, const int K) {
    __shared__ float sW0[DP * kMaxIn];
    __shared__ float sB0[DP], sLn0w[DP], sLn0b[DP];
    __shared__ LayerW<DP> sL, sNext;
    __shared__ float sWo[kMaxOut * DP], sBo[kMaxOut];
    __shared__ double sPart[kWaves][kNodes][DP + 1];
    const int n = a.n, d = a.d, L = a.layers, T = blockDim.x, t = threadIdx.x;
    const int lane = t & 63;
    const int i = blockIdx.x * kNodes + lane;
    const bool live = i < n;
    Tape<DP> tp(a.tape, n, L);
    const size_t nd = static_cast<size_t>(n) * DP;
    const bool drop = a.training && a.rnd != nullptr && a.p_drop > 0.f;

    if (stage == 0) {
        for (int k = t; k < DP * kMaxIn; k += T) {
            const int m = k / kMaxIn, f = k % kMaxIn;
            sW0[k] = (m < d && f < a.f_in) ? a.w0[m * a.f_in + f] : 0.f;
        }
        for (int k = t; k < DP; k += T) {
            sB0[k] = k < d ? a.b0[k] : 0.f;
            sLn0w[k] = (k < d && a.use_bn) ? a.ln0w[k] : 0.f;
            sLn0b[k] = (k < d && a.use_bn) ? a.ln0b[k] : 0.f;
        }
        load_layer<DP>(sNext, a.lp[0], d, a.use_weight, a.use_bn);
        __syncthreads();
        if (!live || t >= 64) return;
        // input layer: Linear -> LayerNorm -> ReLU -> dropout (:188-192)
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
        float q[DP], k[DP], v[DP];
        project<DP>(sNext, a.use_weight, h, q, k, v);
        float* set = tp.qkv(0);
        store_row<DP>(set + static_cast<size_t>(i) * DP, q);
        store_row<DP>(set + nd + static_cast<size_t>(i) * DP, k);
        store_row<DP>(set + 2 * nd + static_cast<size_t>(i) * DP, v);
        return;
    }

    const int l = stage - 1;
    const bool last = l + 1 == L;
    if (part != kPairs) {                       // (the tail's operands)
        load_layer<DP>(sL, a.lp[l], d, a.use_weight, a.use_bn);
        if (!last) load_layer<DP>(sNext, a.lp[l + 1], d, a.use_weight, a.use_bn);
        else {
            for (int k = t; k < kMaxOut * DP; k += T) {
                const int c = k / DP, m = k % DP;
                sWo[k] = (c < a.c && m < d) ? a.wo[c * d + m] : 0.f;
            }
            for (int k = t; k < kMaxOut; k += T) sBo[k] = k < a.c ? a.bo[k] : 0.f;
        }
    }
    const float* set = tp.qkv(l & 1);
    const float* __restrict__ Kl = set + nd;
    const float* __restrict__ Vl = set + 2 * nd;
    // ---- :47-56  sigma(q k^T) v and the row sum over this wave's chunk of the keys ----
    if (part != kTail) {
        float q[DP];
#pragma unroll
        for (int m = 0; m < DP; ++m) q[m] = 0.f;
        if (live) load_row<DP>(set + static_cast<size_t>(i) * DP, q);
        const int w = __builtin_amdgcn_readfirstlane(t >> 6), W = T >> 6;
        const int span = (n + gridDim.y - 1) / gridDim.y;                     // this workgroup's keys, then this wave's
        const int b0 = blockIdx.y * span, b1 = min(n, b0 + span);
        const int len = (span + W - 1) / W;
        const int j0 = min(b1, b0 + w * len), j1 = min(b1, j0 + len);
        double den = 0.0, accd[DP];
#pragma unroll
        for (int m = 0; m < DP; ++m) accd[m] = 0.0;
        // every product rounded once in float32 and added in float64 (see the backward sweep: the row's att must be the
        // quotient of exactly these sums for the backward's  sum_j P_ij (DA_i . v_j - DL_i) = 0  to hold)
#pragma unroll 4
        for (int j = j0; j < j1; ++j) {
            const float* kr = Kl + static_cast<size_t>(j) * DP;
            const float* vr = Vl + static_cast<size_t>(j) * DP;
            float dot = 0.f;
#pragma unroll
            for (int m = 0; m < DP; ++m) dot += q[m] * kr[m];
            const float p = fast_sigmoid(dot);
            den += static_cast<double>(p);
#pragma unroll
            for (int m = 0; m < DP; ++m) accd[m] += static_cast<double>(p * vr[m]);
        }
#pragma unroll
        for (int m = 0; m < DP; ++m) sPart[w][lane][m] = accd[m];
        sPart[w][lane][DP] = den;
    }
    __syncthreads();
    // ---- the waves' chunks in wave order (the key splits in split order); then the node's tail on its wave 0 thread:
    //      aggregation, residual, LayerNorm, next projections ----
    if (!live || t >= 64) return;
    double sums[DP + 1];
    if (part == kTail) add_splits<DP + 1>(sums, tp.partials(), K, n, i);
    else {
#pragma unroll
        for (int m = 0; m <= DP; ++m) sums[m] = 0.0;
        const int W = T >> 6;
        for (int w = 0; w < W; ++w) {
#pragma unroll
            for (int m = 0; m <= DP; ++m) sums[m] += sPart[w][lane][m];
        }
        if (part == kPairs) {
            store_split<DP + 1>(sums, tp.partials(), n, i);
            return;
        }
    }
    float att[DP];
    {
        const double den = sums[DP];
        float lo[DP];
#pragma unroll
        for (int m = 0; m < DP; ++m) {
            const double q64 = sums[m] / den;
            att[m] = static_cast<float>(q64);
            lo[m] = static_cast<float>(q64 - static_cast<double>(att[m]));
        }
        tp.DEN[static_cast<size_t>(l) * n + i] = static_cast<float>(den);
        store_row<DP>(tp.att_lo() + (static_cast<size_t>(l) * n + i) * DP, lo);
    }
    store_row<DP>(tp.ATT + (static_cast<size_t>(l) * n + i) * DP, att);
    float out[DP];
    if (a.use_graph) {
        float g[DP];
#pragma unroll
        for (int m = 0; m < DP; ++m) g[m] = 0.f;
        const int e1 = a.rowptr[i + 1];
        for (int e = a.rowptr[i]; e < e1; ++e) {                                      // :75-78, entries in edge order
            const float wgt = a.val[e];
            float vr[DP];
            load_row<DP>(Vl + static_cast<size_t>(a.nbr[e]) * DP, vr);
#pragma unroll
            for (int m = 0; m < DP; ++m) g[m] += wgt * vr[m];
        }
#pragma unroll
        for (int m = 0; m < DP; ++m) out[m] = a.a_s * att[m] + a.g_s * g[m];          // :130-134
    } else {
#pragma unroll
        for (int m = 0; m < DP; ++m) out[m] = att[m];
    }
    float h[DP];
    load_row<DP>(tp.H + (static_cast<size_t>(l) * n + i) * DP, h);
    if (a.use_source) {                                                               // :139-140
        float x0[DP];
        load_row<DP>(tp.H + static_cast<size_t>(i) * DP, x0);
#pragma unroll
        for (int m = 0; m < DP; ++m) out[m] += x0[m];
    }
    if (a.residual) {                                                                 // :200-201
#pragma unroll
        for (int m = 0; m < DP; ++m) out[m] = a.alpha * out[m] + (1.0f - a.alpha) * h[m];
    }
    store_row<DP>(tp.Z + (static_cast<size_t>(l + 1) * n + i) * DP, out);
    if (a.use_bn) {                                                                   // :202-203
        float mean, rstd;
        ln_stats<DP>(out, d, a.eps, mean, rstd);
#pragma unroll
        for (int m = 0; m < DP; ++m) out[m] = (m < d) ? (out[m] - mean) * rstd * sL.lnw[m] + sL.lnb[m] : 0.f;
    }
    if (drop) dropout_row<DP>(out, a.rnd, (static_cast<int64_t>(l + 1) * n + i) * d, d, a.p_drop);   // :204
    store_row<DP>(tp.H + (static_cast<size_t>(l + 1) * n + i) * DP, out);
    if (!last) {
        float q[DP], k[DP], v[DP];
        project<DP>(sNext, a.use_weight, out, q, k, v);
        float* nxt = tp.qkv((l + 1) & 1);
        store_row<DP>(nxt + static_cast<size_t>(i) * DP, q);
        store_row<DP>(nxt + nd + static_cast<size_t>(i) * DP, k);
        store_row<DP>(nxt + 2 * nd + static_cast<size_t>(i) * DP, v);
    } else {
        for (int c = 0; c < a.c; ++c) {                                               // :208
            float acc = sBo[c];
#pragma unroll
            for (int m = 0; m < DP; ++m) acc += sWo[c * DP + m] * out[m];
            a.y[static_cast<size_t>(i) * a.c + c] = acc;
        }
    }
}

// ======================================================================================================================
// backward
// ======================================================================================================================
// backward
// ======================================================================================================================
// tail_backward_common, then the operands of the pair sweep
template <int DP>
__device__ __forceinline__ void tail_backward(const TinyArgs& a, const Tape<DP>& tp, const GridScratch<DP>& gs, const LayerW<DP>& w,
                                              int l, int i, float (&dy)[DP], bool drop, float keep) {
    const int n = a.n;
    const size_t at_i = static_cast<size_t>(i) * DP;
    const int set = l & 1;
    float datt[DP], q[DP], k[DP], v[DP];
    tail_backward_common<DP>(a, tp, gs, w, l, i, dy, drop, keep, datt, q, k, v);
    float att[DP], att_lo[DP], da[DP];
    load_row<DP>(tp.ATT + (static_cast<size_t>(l) * n + i) * DP, att);
    load_row<DP>(tp.att_lo() + (static_cast<size_t>(l) * n + i) * DP, att_lo);
    const float rden = 1.0f / tp.DEN[static_cast<size_t>(l) * n + i];
    // DA_i = d att_i / den_i;  DL_i = DA_i . att_i from the ROUNDED DA_i and the forward's att_i before ITS rounding (hi + lo
    // in the tape), in float64, kept as float32 hi + lo: the pair sweep
    // forms DA_i . v_j - DL_i, whose sum over j weighted by P_ij is zero by construction -- an inconsistency of one float32
    // rounding between the two terms is the same for all n keys and does not average out as the per-pair roundings do
    double dl = 0.0;
#pragma unroll
    for (int m = 0; m < DP; ++m) {
        da[m] = datt[m] * rden;
        dl += static_cast<double>(da[m]) * (static_cast<double>(att[m]) + static_cast<double>(att_lo[m]));
    }
    store_row<DP>(gs.in_set(set, kDA) + at_i, da);
    const float dl_hi = static_cast<float>(dl);
    gs.dl(set, 0)[i] = dl_hi;
    gs.dl(set, 1)[i] = static_cast<float>(dl - static_cast<double>(dl_hi));
}

// stage 0 = output Linear + tail of layer L - 1 (64 threads per workgroup); stage k = layer L - k
template <int DP>
__global__ __launch_bounds__(kWaves * 64) void grid_sigmoid_backward_kernel(const TinyArgs a, const int stage, const int part, const int K) {
    __shared__ float sW0[DP * kMaxIn];
    __shared__ float sLn0w[DP], sLn0b[DP];
    __shared__ LayerW<DP> sL, sPrev;
    __shared__ float sWo[kMaxOut * DP];
    __shared__ double sPart[kWaves][kNodes][3 * DP];
    const int n = a.n, d = a.d, L = a.layers, T = blockDim.x, t = threadIdx.x;
    const int lane = t & 63;
    const int i = blockIdx.x * kNodes + lane;
    const bool live = i < n;
    Tape<DP> tp(a.tape, n, L);
    GridScratch<DP> gs(a.scratch, n, L);
    const size_t at_i = static_cast<size_t>(i) * DP;
    const bool drop = a.training && a.rnd != nullptr && a.p_drop > 0.f;
    const float keep = drop ? 1.0f / (1.0f - a.p_drop) : 1.0f;

    if (stage == 0) {
        for (int k = t; k < kMaxOut * DP; k += T) {
            const int c = k / DP, m = k % DP;
            sWo[k] = (c < a.c && m < d) ? a.wo[c * d + m] : 0.f;
        }
        load_layer<DP>(sPrev, a.lp[L - 1], d, a.use_weight, a.use_bn);
        __syncthreads();
        if (!live || t >= 64) return;
        float g[DP];
#pragma unroll
        for (int m = 0; m < DP; ++m) g[m] = 0.f;
        if (a.use_source) store_row<DP>(gs.DX0 + at_i, g);
        for (int c = 0; c < a.c; ++c) {
            const float gv = a.gy[static_cast<size_t>(i) * a.c + c];
#pragma unroll
            for (int m = 0; m < DP; ++m) g[m] += gv * sWo[c * DP + m];
        }
        tail_backward<DP>(a, tp, gs, sPrev, L - 1, i, g, drop, keep);
        return;
    }

    const int l = L - stage;
    if (part != kPairs) {
        load_layer<DP>(sL, a.lp[l], d, a.use_weight, a.use_bn);
        if (l > 0) load_layer<DP>(sPrev, a.lp[l - 1], d, a.use_weight, a.use_bn);
        else {
            for (int k = t; k < DP * kMaxIn; k += T) {
                const int m = k / kMaxIn, f = k % kMaxIn;
                sW0[k] = (m < d && f < a.f_in) ? a.w0[m * a.f_in + f] : 0.f;
            }
            for (int k = t; k < DP; k += T) {
                sLn0w[k] = (k < d && a.use_bn) ? a.ln0w[k] : 0.f;
                sLn0b[k] = (k < d && a.use_bn) ? a.ln0b[k] : 0.f;
            }
        }
    }
    const int set = l & 1;
    const float* __restrict__ Qs = gs.in_set(set, kQ);
    const float* __restrict__ Ks = gs.in_set(set, kK);
    const float* __restrict__ Vs = gs.in_set(set, kV);
    const float* __restrict__ DAs = gs.in_set(set, kDA);
    const float* __restrict__ DLs = gs.dl(set, 0);
    const float* __restrict__ DLlo = gs.dl(set, 1);
    // ---- pairs: this node as the QUERY against key o (dq) and as the KEY against query o (dk, dv), o over the wave's chunk ----
    if (part != kTail) {
        float q[DP], k[DP], v[DP], da[DP], dl = 0.f, dl_lo = 0.f;
#pragma unroll
        for (int m = 0; m < DP; ++m) { q[m] = k[m] = v[m] = da[m] = 0.f; }
        if (live) {
            load_row<DP>(Qs + at_i, q);
            load_row<DP>(Ks + at_i, k);
            load_row<DP>(Vs + at_i, v);
            load_row<DP>(DAs + at_i, da);
            dl = DLs[i];
            dl_lo = DLlo[i];
        }
        const int w = __builtin_amdgcn_readfirstlane(t >> 6), W = T >> 6;
        const int span = (n + gridDim.y - 1) / gridDim.y;
        const int b0 = blockIdx.y * span, b1 = min(n, b0 + span);
        const int len = (span + W - 1) / W;
        const int o0 = min(b1, b0 + w * len), o1 = min(b1, o0 + len);
        double dqd[DP], dkd[DP], dvd[DP];
#pragma unroll
        for (int m = 0; m < DP; ++m) { dqd[m] = 0.0; dkd[m] = 0.0; dvd[m] = 0.0; }
        // The sums cancel (sum_j P_ij (DA_i . v_j - DL_i) = 0 by construction, and P (1 - P) is nearly constant for small
        // scores): every product is rounded once in float32 and ADDED in float64 -- v_add_f64 issues at v_fma_f32's rate here --
        // so the rounding of partial sums, which dominates a float32 accumulation, never enters.
#pragma unroll 2
        for (int o = o0; o < o1; ++o) {
            const float* qo = Qs + static_cast<size_t>(o) * DP;
            const float* ko = Ks + static_cast<size_t>(o) * DP;
            const float* vo = Vs + static_cast<size_t>(o) * DP;
            const float* dao = DAs + static_cast<size_t>(o) * DP;
            const float dlo = DLs[o], dlo_lo = DLlo[o];
            float dot = 0.f, dp = -dl;
#pragma unroll
            for (int m = 0; m < DP; ++m) { dot += q[m] * ko[m]; dp += da[m] * vo[m]; }
            dp -= dl_lo;
            float p = fast_sigmoid(dot);
            float dsv = dp * p * (1.0f - p);
#pragma unroll
            for (int m = 0; m < DP; ++m) dqd[m] += static_cast<double>(dsv * ko[m]);
            dot = 0.f;
            dp = -dlo;
#pragma unroll
            for (int m = 0; m < DP; ++m) { dot += qo[m] * k[m]; dp += dao[m] * v[m]; }
            dp -= dlo_lo;
            p = fast_sigmoid(dot);
            dsv = dp * p * (1.0f - p);
#pragma unroll
            for (int m = 0; m < DP; ++m) {
                dvd[m] += static_cast<double>(p * dao[m]);
                dkd[m] += static_cast<double>(dsv * qo[m]);
            }
        }
#pragma unroll
        for (int m = 0; m < DP; ++m) {
            sPart[w][lane][m] = dqd[m];
            sPart[w][lane][DP + m] = dkd[m];
            sPart[w][lane][2 * DP + m] = dvd[m];
        }
    }
    __syncthreads();
    if (!live || t >= 64) return;
    float dq[DP], dk[DP], dv[DP];
    {
        double s[3 * DP];
        if (part == kTail) add_splits<3 * DP>(s, gs.partials(), K, n, i);
        else {
#pragma unroll
            for (int m = 0; m < 3 * DP; ++m) s[m] = 0.0;
            const int W = T >> 6;
            for (int w = 0; w < W; ++w) {
#pragma unroll
                for (int m = 0; m < 3 * DP; ++m) s[m] += sPart[w][lane][m];
            }
            if (part == kPairs) {
                store_split<3 * DP>(s, gs.partials(), n, i);
                return;
            }
        }
#pragma unroll
        for (int m = 0; m < DP; ++m) {
            dq[m] = static_cast<float>(s[m]);
            dk[m] = static_cast<float>(s[DP + m]);
            dv[m] = static_cast<float>(s[2 * DP + m]);
        }
    }
    if (a.use_graph) {                              // adjoint of the aggregation: entries of the TRANSPOSED CSR, edge order
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
    if (l > 0) {
        tail_backward<DP>(a, tp, gs, sPrev, l - 1, i, dh, drop, keep);
        return;
    }
    input_backward<DP>(a, tp, gs, sW0, sLn0w, sLn0b, i, dh, drop, keep);
}

// Every sum over nodes of the backward, one workgroup per parameter gradient (outer_sum: chunk partials in float64, fixed order):
//   block 0 .. c-1        d fcs.1.weight row c     c                d fcs.1.bias
//   c+1 + 8 l + {0..7}    layer l: d bns.weight, d bns.bias, d Wq.weight, d Wq.bias, d Wk.weight, d Wk.bias, d Wv.weight, d Wv.bias
//   then                  d bns.0.weight, d bns.0.bias, d fcs.0.bias, d fcs.0.weight row m (m < d)
template <int DP>
__global__ __launch_bounds__(512) void grid_sums_kernel(const TinyArgs a) {
    __shared__ double sPart[1024];
    __shared__ float sOut[DP * kMaxIn > 64 ? DP * kMaxIn : 64];
    const int n = a.n, d = a.d, L = a.layers, T = blockDim.x, t = threadIdx.x;
    Tape<DP> tp(a.tape, n, L);
    GridScratch<DP> gs(a.scratch, n, L);
    int job = blockIdx.x;
    if (job < a.c) {
        outer_sum(a.gy + job, a.c, 1, tp.H + static_cast<size_t>(L) * gs.nd, DP, DP, n, 1.f, sOut, sPart);
        if (t < d) a.gwo[job * d + t] = sOut[t];
        return;
    }
    job -= a.c;
    if (job == 0) {
        outer_sum(a.gy, a.c, a.c, nullptr, 0, 1, n, 1.f, sOut, sPart);
        if (t < a.c) a.gbo[t] = sOut[t];
        return;
    }
    job -= 1;
    if (job < 8 * L) {
        const int l = job / 8, what = job % 8;
        const LayerGrads& g = a.lg[l];
        if (what < 2) {
            if (!a.use_bn) return;
            outer_sum(gs.per_layer(l, what == 0 ? kDYX : kDY), DP, DP, nullptr, 0, 1, n, 1.f, sOut, sPart);
            float* dst = what == 0 ? g.lnw : g.lnb;
            if (t < d) dst[t] = sOut[t];
            return;
        }
        const int which = (what - 2) / 2;            // 0 q, 1 k, 2 v
        if (which == 2 && !a.use_weight) return;
        const float* G = gs.per_layer(l, which == 0 ? kDQ : (which == 1 ? kDK : kDV));
        if ((what & 1) == 0) {
            outer_sum(G, DP, DP, tp.H + static_cast<size_t>(l) * gs.nd, DP, DP, n, 1.f, sOut, sPart);
            float* gw = which == 0 ? g.wq : (which == 1 ? g.wk : g.wv);
            for (int kk = t; kk < d * d; kk += T) gw[kk] = sOut[(kk / d) * DP + kk % d];
        } else {
            outer_sum(G, DP, DP, nullptr, 0, 1, n, 1.f, sOut, sPart);
            float* gb = which == 0 ? g.bq : (which == 1 ? g.bk : g.bv);
            if (t < d) gb[t] = sOut[t];
        }
        return;
    }
    job -= 8 * L;
    if (job < 2) {
        if (!a.use_bn) return;
        outer_sum(job == 0 ? gs.DYX0 : gs.DY0, DP, DP, nullptr, 0, 1, n, 1.f, sOut, sPart);
        float* dst = job == 0 ? a.gln0w : a.gln0b;
        if (t < d) dst[t] = sOut[t];
        return;
    }
    if (job == 2) {
        outer_sum(gs.DPRE, DP, DP, nullptr, 0, 1, n, 1.f, sOut, sPart);
        if (t < d) a.gb0[t] = sOut[t];
        return;
    }
    const int m = job - 3;                          // d W0 [m][f] = sum_i dpre[i][m] x[i][f]
    if (m < d) {
        outer_sum(gs.DPRE + m, DP, 1, a.x, a.ldx, a.f_in, n, 1.f, sOut, sPart);
        for (int f = t; f < a.f_in; f += T) a.gw0[m * a.f_in + f] = sOut[f];
    }
}

template <int DP>
int forward_launches(const TinyArgs& a, hipStream_t st) {
    const int G = (a.n + kNodes - 1) / kNodes, K = key_splits(a.n);
    hipLaunchKernelGGL(grid_sigmoid_forward_kernel<DP>, dim3(G), dim3(64), 0, st, a, 0, kFused, 1);
    for (int l = 0; l < a.layers; ++l) {
        if (K == 1) hipLaunchKernelGGL(grid_sigmoid_forward_kernel<DP>, dim3(G), dim3(kWaves * 64), 0, st, a, l + 1, kFused, 1);
        else {
            hipLaunchKernelGGL(grid_sigmoid_forward_kernel<DP>, dim3(G, K), dim3(kWaves * 64), 0, st, a, l + 1, kPairs, K);
            hipLaunchKernelGGL(grid_sigmoid_forward_kernel<DP>, dim3(G), dim3(64), 0, st, a, l + 1, kTail, K);
        }
    }
    return dif::launch_status("dif_tiny_forward_f32");
}

template <int DP>
int backward_launches(const TinyArgs& a, hipStream_t st) {
    const int G = (a.n + kNodes - 1) / kNodes, K = key_splits(a.n);
    hipLaunchKernelGGL(grid_sigmoid_backward_kernel<DP>, dim3(G), dim3(64), 0, st, a, 0, kFused, 1);
    for (int k = 1; k <= a.layers; ++k) {
        if (K == 1) hipLaunchKernelGGL(grid_sigmoid_backward_kernel<DP>, dim3(G), dim3(kWaves * 64), 0, st, a, k, kFused, 1);
        else {
            hipLaunchKernelGGL(grid_sigmoid_backward_kernel<DP>, dim3(G, K), dim3(kWaves * 64), 0, st, a, k, kPairs, K);
            hipLaunchKernelGGL(grid_sigmoid_backward_kernel<DP>, dim3(G), dim3(64), 0, st, a, k, kTail, K);
        }
    }
    return grid_sums(a, st);
}

}  // namespace

int tiny::grid_sums(const TinyArgs& a, hipStream_t st) {
    if (a.d <= 4) hipLaunchKernelGGL(grid_sums_kernel<4>, dim3(a.c + 1 + 8 * a.layers + 3 + a.d), dim3(512), 0, st, a);
    else hipLaunchKernelGGL(grid_sums_kernel<8>, dim3(a.c + 1 + 8 * a.layers + 3 + a.d), dim3(512), 0, st, a);
    return dif::launch_status("dif_tiny_backward_f32");
}

int tiny::grid_sigmoid_forward(const TinyArgs& a, hipStream_t st) {
    return a.d <= 4 ? forward_launches<4>(a, st) : forward_launches<8>(a, st);
}

int tiny::grid_sigmoid_backward(const TinyArgs& a, hipStream_t st) {
    return a.d <= 4 ? backward_launches<4>(a, st) : backward_launches<8>(a, st);
}

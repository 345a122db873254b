// Shared by the grid plans of the whole-model kernels (tiny_sigmoid_grid.hip, tiny_simple_grid.hip): one launch per layer stage with
// 64 nodes per workgroup; the backward's scratch layout and the per-node pieces that do not depend on the attention kernel.
#pragma once
#include "tiny_common.h"

namespace tiny {

template <int DP>
struct GridScratch {                            // backward scratch (floats), nd = n * DP
    float *DIR, *DX0, *DPRE, *DY0, *DYX0;
    float* layer0;                              // [L][5][nd]: DQ DK DV DY DYX
    float* set0;                                // [2][5][nd]: SG Q K V DA
    float* dl0;                                 // [2][2][n]: DL as float32 hi + lo
    float* tail;                                // the key splits' partial sums [K][n][3 DP] (double)
    size_t nd;
    __device__ GridScratch(float* S, int n, int L) {
        nd = static_cast<size_t>(n) * DP;
        DIR = S; DX0 = S + nd; DPRE = S + 2 * nd; DY0 = S + 3 * nd; DYX0 = S + 4 * nd;
        layer0 = S + 5 * nd;
        set0 = layer0 + 5 * nd * L;
        dl0 = set0 + 10 * nd;
        tail = dl0 + 4 * static_cast<size_t>(n);
        tail += (-(tail - S)) & 3;              // 16-byte aligned (S is)
    }
    __device__ double* partials() const { return reinterpret_cast<double*>(tail); }
    __device__ float* per_layer(int l, int which) const { return layer0 + (static_cast<size_t>(l) * 5 + which) * nd; }
    __device__ float* in_set(int set, int which) const { return set0 + (static_cast<size_t>(set) * 5 + which) * nd; }
    __device__ float* dl(int set, int part) const { return dl0 + static_cast<size_t>(2 * set + part) * (nd / DP); }
};
enum { kDQ = 0, kDK = 1, kDV = 2, kDY = 3, kDYX = 4 };
enum { kSG = 0, kQ = 1, kK = 2, kV = 3, kDA = 4 };

// Wq / Wk / Wv of node i's layer input h -> the set's rows
template <int DP>
__device__ __forceinline__ void project(const LayerW<DP>& w, bool use_weight, const float (&h)[DP], float (&q)[DP], float (&k)[DP],
                                        float (&v)[DP]) {
    matvec<DP>(w.wq, w.bq, h, q);
    matvec<DP>(w.wk, w.bk, h, k);
    if (use_weight) matvec<DP>(w.wv, w.bv, h, v);
    else {
#pragma unroll
        for (int m = 0; m < DP; ++m) v[m] = h[m];
    }
}

// ======================================================================================================================
// What the one-workgroup kernel's "phase 1" does for node i of layer l, from the gradient dy of the layer's OUTPUT (H[l + 1]),
// up to the attention: dropout / LayerNorm / residual backward (d LayerNorm operands kept per layer for the sums at the end),
// the direct term DIR, + x0, the aggregation's operand SG, and q, k, v recomputed into the layer's set.  -> datt, q, k, v
template <int DP>
__device__ __forceinline__ void tail_backward_common(const TinyArgs& a, const Tape<DP>& tp, const GridScratch<DP>& gs, const LayerW<DP>& w,
                                                     int l, int i, float (&dy)[DP], bool drop, float keep, float (&datt)[DP],
                                                     float (&q)[DP], float (&k)[DP], float (&v)[DP]) {
    const int n = a.n, d = a.d;
    const size_t at_i = static_cast<size_t>(i) * DP;
    float z[DP], dz[DP];
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
            const float gw = w.lnw[m] * dy[m];
            m1 += gw;
            m2 += gw * xh[m];
        }
        m1 /= static_cast<float>(d);
        m2 /= static_cast<float>(d);
        float dyx[DP];
#pragma unroll
        for (int m = 0; m < DP; ++m) {
            dyx[m] = dy[m] * xh[m];
            dz[m] = (m < d) ? rstd * (w.lnw[m] * dy[m] - m1 - xh[m] * m2) : 0.f;
        }
        store_row<DP>(gs.per_layer(l, kDY) + at_i, dy);
        store_row<DP>(gs.per_layer(l, kDYX) + at_i, dyx);
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
    store_row<DP>(gs.DIR + at_i, dir);
    if (a.use_source) {
        float acc[DP];
        load_row<DP>(gs.DX0 + at_i, acc);
#pragma unroll
        for (int m = 0; m < DP; ++m) acc[m] += dout[m];
        store_row<DP>(gs.DX0 + at_i, acc);
    }
    const int set = l & 1;
    if (a.use_graph) {
        float sg[DP];
#pragma unroll
        for (int m = 0; m < DP; ++m) { sg[m] = a.g_s * dout[m]; datt[m] = a.a_s * dout[m]; }
        store_row<DP>(gs.in_set(set, kSG) + at_i, sg);
    } else {
#pragma unroll
        for (int m = 0; m < DP; ++m) datt[m] = dout[m];
    }
    float h[DP];
    load_row<DP>(tp.H + (static_cast<size_t>(l) * n + i) * DP, h);
    project<DP>(w, a.use_weight, h, q, k, v);
    store_row<DP>(gs.in_set(set, kQ) + at_i, q);
    store_row<DP>(gs.in_set(set, kK) + at_i, k);
    store_row<DP>(gs.in_set(set, kV) + at_i, v);
}

// input layer backward (:188-192) for node i from dh = the gradient of H[0]: d pre-activation -> DPRE (and the LayerNorm operands)
// for the sums at the end, dx
template <int DP>
__device__ __forceinline__ void input_backward(const TinyArgs& a, const Tape<DP>& tp, const GridScratch<DP>& gs, const float* sW0,
                                               const float* sLn0w, const float* sLn0b, int i, float (&dh)[DP], bool drop, float keep) {
    const int n = a.n, d = a.d;
    (void)n;
    const size_t at_i = static_cast<size_t>(i) * DP;
    float z[DP];
    if (a.use_source) {
        float acc[DP];
        load_row<DP>(gs.DX0 + at_i, acc);
#pragma unroll
        for (int m = 0; m < DP; ++m) dh[m] += acc[m];
    }
    if (drop) {
        const int64_t at = static_cast<int64_t>(i) * d;
#pragma unroll
        for (int m = 0; m < DP; ++m)
            if (m < d) dh[m] = (a.rnd[at + m] >= a.p_drop) ? dh[m] * keep : 0.f;
    }
    load_row<DP>(tp.Z + at_i, z);
    float dpre[DP];
    if (a.use_bn) {
        float mean, rstd;
        ln_stats<DP>(z, d, a.eps, mean, rstd);
        float xh[DP], m1 = 0.f, m2 = 0.f, dyx[DP];
#pragma unroll
        for (int m = 0; m < DP; ++m) {
            xh[m] = (m < d) ? (z[m] - mean) * rstd : 0.f;
            const float yv = xh[m] * sLn0w[m] + sLn0b[m];
            if (!(yv > 0.f)) dh[m] = 0.f;                                            // ReLU
            const float gw = sLn0w[m] * dh[m];
            m1 += gw;
            m2 += gw * xh[m];
            dyx[m] = dh[m] * xh[m];
        }
        m1 /= static_cast<float>(d);
        m2 /= static_cast<float>(d);
#pragma unroll
        for (int m = 0; m < DP; ++m) dpre[m] = (m < d) ? rstd * (sLn0w[m] * dh[m] - m1 - xh[m] * m2) : 0.f;
        store_row<DP>(gs.DY0 + at_i, dh);
        store_row<DP>(gs.DYX0 + at_i, dyx);
    } else {
#pragma unroll
        for (int m = 0; m < DP; ++m) dpre[m] = (m < d && z[m] > 0.f) ? dh[m] : 0.f;
    }
    store_row<DP>(gs.DPRE + at_i, dpre);
    if (a.dx) {
        for (int f = 0; f < a.f_in; ++f) {
            float acc = 0.f;
#pragma unroll
            for (int m = 0; m < DP; ++m) acc += dpre[m] * sW0[m * kMaxIn + f];
            a.dx[static_cast<size_t>(i) * a.f_in + f] = acc;
        }
    }
}

// every sum over nodes of the backward (parameter gradients) from the per-layer operands: one workgroup per gradient
int grid_sums(const TinyArgs& a, hipStream_t st);

}  // namespace tiny

// Backward of the coefficient stage of the closed-form `simple` layer (csrc/simple_layer.hip, coeffs_kernel): training
// through the Gram record (difformer_amd/autograd_ops.py, _ClosedFormLayer) needs the gradients of
//     coef = [MnT: D x C][cn: D][u: C][cd] = f(G, sx, N; Wq, bq, Wk, bk, Wv, bv)          (difformer.py:18-38 in closed form)
// with respect to the record and the six parameters, given d coef.  Everything is 64 x 64: ONE workgroup of 16 waves, one
// 16 x 16 tile of every product per wave on the fp32 matrix core, operands in LDS (eight 64 x 68 blocks, reused in phases).
//
// With wq = Wq sx, ks = Wk sx + N bk (sum k), vs, qs alike, Tq = Wq G, Tk = Wk G, Tv = Wv G:
//   KtV = Tk Wv^T + bk wv^T + ks bv^T      |Q|^2 = <Tq, Wq> + 2 bq.wq + N bq.bq     s = (|Q|^2 |K|^2)^-1/2
//   Mn = a s Wq^T KtV    cn = a (s bq^T KtV + vs)    u = s Wq^T ks    cd = s bq.ks + N
// Backward (dMn = dMnT^T, a = attn_scale):
//   core = Wq dMn                     dKtV = a s (core + bq dcn^T)              d vs = a dcn
//   ds   = a (<core, KtV> + dcn.(bq^T KtV)) + du.(Wq^T ks) + dcd bq.ks         d|Q|^2 = -ds s / (2 |Q|^2), d|K|^2 alike
//   d ks = s (Wq du + bq dcd)
//   dWq = a s KtV dMn^T + s ks du^T + 2 d|Q|^2 (Tq + bq sx^T)      dbq = a s KtV dcn + s dcd ks + 2 d|Q|^2 qs
//   dWk = 2 d|K|^2 (Tk + bk sx^T) + dKtV Tv + (dKtV bv + d ks) sx^T    dbk = 2 d|K|^2 ks + dKtV vs + N d ks
//   dWv = dKtV^T Tk + (dKtV^T bk + a dcn) sx^T                         dbv = dKtV^T ks + N a dcn
//   dG  = d|Q|^2 Wq^T Wq + d|K|^2 Wk^T Wk + Wk^T (dKtV Wv)            -> S = dG + dG^T
//   t   = 2 d|Q|^2 Wq^T bq + 2 d|K|^2 Wk^T bk + Wk^T (dKtV bv + d ks) + Wv^T (a dcn) + (dKtV Wv)^T bk
// so that the gradient of the rows is  dx = x S + 1 t^T  (G = x^T x, sx = x^T 1).
// Checked against torch autograd of the float64 forward (tests/test_gpu_closed_form.py) and, through the training step,
// against the reference's own gradients (tests/test_gpu_grad.py <- tests/golden/golden_grad.npz).
#include "dif_common.h"

namespace {

using dif::f32x4;

constexpr int kLd = 68;
constexpr int kBlock = 64 * kLd;

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// one 16 x 16 tile of a 64^3 product: lane holds D[16 ti + 4 lg + reg][16 tj + l15]
template <typename FA, typename FB>
__device__ __forceinline__ f32x4 tile_product(FA A, FB B, int ti, int tj, int l15, int lg) {
    f32x4 d = zero4();
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(A(16 * ti + l15, 4 * ks + lg), B(4 * ks + lg, 16 * tj + l15), d, 0, 0, 0);
    return d;
}

// 256 consecutive threads: out[o] = sum_i blk[o][i] v[i]  (COL = false)  or  sum_i blk[i][o] v[i]  (COL = true); four
// lanes per output fold 16 terms each.  `t` = thread index inside the group of 256.
template <bool COL>
__device__ __forceinline__ void matvec(const float* __restrict__ blk, const float* __restrict__ v, float* __restrict__ out,
                                       int t) {
    const int o = t >> 2, part = t & 3;
    float a = 0.f;
#pragma unroll
    for (int i = 16 * part; i < 16 * part + 16; ++i) a += (COL ? blk[i * kLd + o] : blk[o * kLd + i]) * v[i];
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    if (part == 0) out[o] = a;
}

enum Vec { V_SX, V_BQ, V_BK, V_BV, V_DCN, V_DU, V_WQ, V_WK, V_WV, V_KS, V_VS, V_QS, V_BQK, V_WQKS, V_DKS, V_KDCN, V_DKBV,
           V_DKVS, V_DKTBK, V_DKTKS, V_ADCN, V_T0, V_T1, V_T2, V_T3, V_T4, V_T5, V_COUNT };
constexpr int kSmemFloats = 8 * kBlock + V_COUNT * 64 + 16 + 4;

__global__ __launch_bounds__(1024) void coeffs_bwd_kernel(const float* __restrict__ rec, float N, int C, int D,
                                                          const float* __restrict__ Wq, const float* __restrict__ bq,
                                                          const float* __restrict__ Wk, const float* __restrict__ bk,
                                                          const float* __restrict__ Wv, const float* __restrict__ bv,
                                                          float a_s, const float* __restrict__ coef,
                                                          const float* __restrict__ dcoef, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float smem[kSmemFloats];            // 146 KB of the CU's 160
    float* sG = smem;                    // G, then dKtV
    float* sWq = sG + kBlock;
    float* sWk = sWq + kBlock;
    float* sWv = sWk + kBlock;
    float* sDM = sWv + kBlock;           // dMnT [d][c]
    float* sTk = sDM + kBlock;           // Wk G, then Y = dKtV Wv
    float* sTv = sTk + kBlock;           // Wv G, then dG
    float* sKtV = sTv + kBlock;
    float* vec = sKtV + kBlock;          // V_COUNT x 64
    float* s_red = vec + V_COUNT * 64;   // [16]
    float* s_scal = s_red + 16;          // [4]: d|Q|^2, d|K|^2
    auto V = [&](int which) { return vec + 64 * which; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int ti = wave >> 2, tj = wave & 3;
    const bool has_wv = Wv != nullptr;
    const int base = D * C + D;
    const float s = coef[base + C + 1], q2 = coef[base + C + 2], k2 = coef[base + C + 3];
    const float dcd = dcoef[base + C];

    // ---- stage --------------------------------------------------------------------------------------------------
    for (int e = tid; e < 64 * 64; e += 1024) {
        const int r = e >> 6, c = e & 63;
        sG[r * kLd + c] = (r < C && c < C) ? rec[r * C + c] : 0.f;
        const bool in = r < D && c < C;
        sWq[r * kLd + c] = in ? Wq[r * C + c] : 0.f;
        sWk[r * kLd + c] = in ? Wk[r * C + c] : 0.f;
        sWv[r * kLd + c] = has_wv ? (in ? Wv[r * C + c] : 0.f) : ((r == c && r < D) ? 1.f : 0.f);
        sDM[r * kLd + c] = in ? dcoef[r * C + c] : 0.f;
    }
    if (tid < 64) {
        V(V_SX)[tid] = tid < C ? rec[C * C + tid] : 0.f;
        V(V_BQ)[tid] = tid < D ? bq[tid] : 0.f;
        V(V_BK)[tid] = tid < D ? bk[tid] : 0.f;
        V(V_BV)[tid] = (has_wv && tid < D) ? bv[tid] : 0.f;
        V(V_DCN)[tid] = tid < D ? dcoef[D * C + tid] : 0.f;
        V(V_ADCN)[tid] = tid < D ? a_s * dcoef[D * C + tid] : 0.f;
        V(V_DU)[tid] = tid < C ? dcoef[base + tid] : 0.f;
    }
    __syncthreads();

    // ---- 1: W sx, Tk, Tv (LDS), Tq (registers) ------------------------------------------------------------------
    if (tid < 768) {
        const int which = tid >> 8;
        matvec<false>(which == 0 ? sWq : which == 1 ? sWk : sWv, V(V_SX), V(which == 0 ? V_WQ : which == 1 ? V_WK : V_WV),
                      tid & 255);
    }
    const auto G_ = [&](int k, int j) { return sG[k * kLd + j]; };
    const f32x4 tq = tile_product([&](int i, int k) { return sWq[i * kLd + k]; }, G_, ti, tj, l15, lg);
    {
        const f32x4 tk = tile_product([&](int i, int k) { return sWk[i * kLd + k]; }, G_, ti, tj, l15, lg);
        const f32x4 tv = tile_product([&](int i, int k) { return sWv[i * kLd + k]; }, G_, ti, tj, l15, lg);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = 16 * ti + 4 * lg + reg, c = 16 * tj + l15;
            sTk[m * kLd + c] = tk[reg];
            sTv[m * kLd + c] = tv[reg];
        }
    }
    __syncthreads();                                                   // G is free from here: its block takes dKtV
    if (tid < 64) {
        V(V_KS)[tid] = V(V_WK)[tid] + N * V(V_BK)[tid];
        V(V_VS)[tid] = V(V_WV)[tid] + N * V(V_BV)[tid];
        V(V_QS)[tid] = V(V_WQ)[tid] + N * V(V_BQ)[tid];
    }
    __syncthreads();

    // ---- 2: KtV, dKtV, <core, KtV> ------------------------------------------------------------------------------
    float* sDK = sG;
    {
        const f32x4 kv = tile_product([&](int i, int k) { return sTk[i * kLd + k]; }, [&](int k, int j) { return sWv[j * kLd + k]; },
                                      ti, tj, l15, lg);
        const f32x4 core = tile_product([&](int i, int k) { return sWq[i * kLd + k]; },
                                        [&](int k, int j) { return sDM[j * kLd + k]; }, ti, tj, l15, lg);
        float p = 0.f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = 16 * ti + 4 * lg + reg, d = 16 * tj + l15;
            const float ktv = kv[reg] + V(V_BK)[m] * V(V_WV)[d] + V(V_KS)[m] * V(V_BV)[d];
            sKtV[m * kLd + d] = ktv;
            sDK[m * kLd + d] = a_s * s * (core[reg] + V(V_BQ)[m] * V(V_DCN)[d]);
            p += core[reg] * ktv;
        }
        p = dif::wave_sum(p);
        if (lane == 0) s_red[wave] = p;
    }
    __syncthreads();

    // ---- 3: the vectors that need KtV / dKtV whole, then the scalars --------------------------------------------
    {
        const int gi = tid >> 8, t = tid & 255;
        if (gi == 0) { matvec<true>(sKtV, V(V_BQ), V(V_BQK), t); matvec<false>(sDK, V(V_BV), V(V_DKBV), t); }
        else if (gi == 1) { matvec<true>(sWq, V(V_KS), V(V_WQKS), t); matvec<false>(sDK, V(V_VS), V(V_DKVS), t); }
        else if (gi == 2) { matvec<false>(sWq, V(V_DU), V(V_T0), t); matvec<true>(sDK, V(V_BK), V(V_DKTBK), t); }
        else { matvec<false>(sKtV, V(V_DCN), V(V_KDCN), t); matvec<true>(sDK, V(V_KS), V(V_DKTKS), t); }
    }
    __syncthreads();
    if (tid < 64) V(V_DKS)[tid] = s * (V(V_T0)[tid] + V(V_BQ)[tid] * dcd);
    if (wave == 1) {
        float ds = (lane < 16 ? s_red[lane] : 0.f) + V(V_DCN)[lane] * V(V_BQK)[lane];
        ds = a_s * ds + V(V_DU)[lane] * V(V_WQKS)[lane] + dcd * V(V_BQ)[lane] * V(V_KS)[lane];
        ds = dif::wave_sum(ds);
        if (lane == 0) {
            s_scal[0] = -0.5f * ds * s / q2;
            s_scal[1] = -0.5f * ds * s / k2;
        }
    }
    __syncthreads();
    const float dq2 = s_scal[0], dk2 = s_scal[1];

    // ---- 4: parameter gradients ----------------------------------------------------------------------------------
    float* oS = out;
    float* ot = oS + C * C;
    float* oWq = ot + C;
    float* obq = oWq + D * C;
    float* oWk = obq + D;
    float* obk = oWk + D * C;
    float* oWv = obk + D;
    float* obv = oWv + D * C;
    {
        const f32x4 e4 = tile_product([&](int i, int k) { return sKtV[i * kLd + k]; }, [&](int k, int j) { return sDM[k * kLd + j]; },
                                      ti, tj, l15, lg);
        const f32x4 f4 = tile_product([&](int i, int k) { return sDK[i * kLd + k]; }, [&](int k, int j) { return sTv[k * kLd + j]; },
                                      ti, tj, l15, lg);
        const f32x4 h4 = tile_product([&](int i, int k) { return sDK[k * kLd + i]; }, [&](int k, int j) { return sTk[k * kLd + j]; },
                                      ti, tj, l15, lg);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = 16 * ti + 4 * lg + reg, c = 16 * tj + l15;
            if (m < D && c < C) {
                const float sxc = V(V_SX)[c];
                oWq[m * C + c] = a_s * s * e4[reg] + s * V(V_KS)[m] * V(V_DU)[c] + 2.f * dq2 * (tq[reg] + V(V_BQ)[m] * sxc);
                oWk[m * C + c] = 2.f * dk2 * (sTk[m * kLd + c] + V(V_BK)[m] * sxc) + f4[reg] + (V(V_DKBV)[m] + V(V_DKS)[m]) * sxc;
                if (has_wv) oWv[m * C + c] = h4[reg] + (V(V_DKTBK)[m] + V(V_ADCN)[m]) * sxc;
            }
        }
    }
    if (tid < D) {
        obq[tid] = a_s * s * V(V_KDCN)[tid] + s * dcd * V(V_KS)[tid] + 2.f * dq2 * V(V_QS)[tid];
        obk[tid] = 2.f * dk2 * V(V_KS)[tid] + V(V_DKVS)[tid] + N * V(V_DKS)[tid];
        if (has_wv) obv[tid] = V(V_DKTKS)[tid] + N * V(V_ADCN)[tid];
    }
    if (tid < 64) V(V_T5)[tid] = V(V_DKBV)[tid] + V(V_DKS)[tid];          // dKtV bv + d ks
    __syncthreads();                                                      // Tk has been read: its block takes Y

    // ---- 5: Y = dKtV Wv, dG ----------------------------------------------------------------------------------------
    float* sY = sTk;
    {
        const f32x4 y4 = tile_product([&](int i, int k) { return sDK[i * kLd + k]; }, [&](int k, int j) { return sWv[k * kLd + j]; },
                                      ti, tj, l15, lg);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) sY[(16 * ti + 4 * lg + reg) * kLd + 16 * tj + l15] = y4[reg];
    }
    __syncthreads();
    float* sdG = sTv;
    {
        const f32x4 qq = tile_product([&](int i, int k) { return sWq[k * kLd + i]; }, [&](int k, int j) { return sWq[k * kLd + j]; },
                                      ti, tj, l15, lg);
        const f32x4 kk = tile_product([&](int i, int k) { return sWk[k * kLd + i]; }, [&](int k, int j) { return sWk[k * kLd + j]; },
                                      ti, tj, l15, lg);
        const f32x4 ky = tile_product([&](int i, int k) { return sWk[k * kLd + i]; }, [&](int k, int j) { return sY[k * kLd + j]; },
                                      ti, tj, l15, lg);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            sdG[(16 * ti + 4 * lg + reg) * kLd + 16 * tj + l15] = dq2 * qq[reg] + dk2 * kk[reg] + ky[reg];
    }
    {
        const int gi = tid >> 8, t = tid & 255;
        if (gi == 0) { matvec<true>(sWq, V(V_BQ), V(V_T0), t); matvec<true>(sWk, V(V_BK), V(V_T1), t); }
        else if (gi == 1) matvec<true>(sWk, V(V_T5), V(V_T2), t);
        else if (gi == 2) matvec<true>(sWv, V(V_ADCN), V(V_T3), t);
        else matvec<true>(sY, V(V_BK), V(V_T4), t);
    }
    __syncthreads();

    // ---- 6: S = dG + dG^T, t ---------------------------------------------------------------------------------------
    for (int e = tid; e < C * C; e += 1024) {
        const int i = e / C, j = e - i * C;
        oS[e] = sdG[i * kLd + j] + sdG[j * kLd + i];
    }
    if (tid < C) ot[tid] = 2.f * dq2 * V(V_T0)[tid] + 2.f * dk2 * V(V_T1)[tid] + V(V_T2)[tid] + V(V_T3)[tid] + V(V_T4)[tid];
}

}  // namespace

extern "C" size_t dif_simple_coeffs_bwd_len(int C, int D) {
    if (C <= 0 || D <= 0) return 0;
    return static_cast<size_t>(C) * C + C + 3 * (static_cast<size_t>(D) * C + D);
}

// out = [S: C x C][t: C][dWq: D x C][dbq: D][dWk: D x C][dbk: D][dWv: D x C][dbv: D]   (dWv, dbv untouched without Wv)
// dcoef = the gradient with respect to coef, in coef's layout [dMnT: D x C][dcn: D][du: C][dcd]; coef = the forward's output
// (its scale and norms are read back).
extern "C" int dif_simple_coeffs_bwd_f32(const float* record, int64_t n_global, int C, int D, const float* Wq, const float* bq,
                                         const float* Wk, const float* bk, const float* Wv, const float* bv, float attn_scale,
                                         const float* coef, const float* dcoef, float* out, dif_stream_t stream) {
    DIF_REQUIRE(record && Wq && bq && Wk && bk && coef && dcoef && out && n_global > 0, DIF_E_BADARG,
                "dif_simple_coeffs_bwd: null pointer");
    DIF_REQUIRE(C > 0 && C <= 64 && D > 0 && D <= 64, DIF_E_SHAPE, "dif_simple_coeffs_bwd: covers C, D <= 64 (got %d, %d)", C, D);
    DIF_REQUIRE((Wv == nullptr) == (bv == nullptr), DIF_E_BADARG, "dif_simple_coeffs_bwd: Wv and bv go together");
    DIF_REQUIRE(Wv != nullptr || C == D, DIF_E_SHAPE, "dif_simple_coeffs_bwd: without a value projection C must equal D");
    hipLaunchKernelGGL(coeffs_bwd_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), record,
                       static_cast<float>(n_global), C, D, Wq, bq, Wk, bk, Wv, bv, attn_scale, coef, dcoef, out);
    return dif::launch_status("coeffs_bwd_kernel");
}

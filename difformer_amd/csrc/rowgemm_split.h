// Shared by csrc/simple_attn_bwd.hip (dif_rowgemm_f32) and csrc/simple_attn.hip (dif_simple_apply_f32): a kernel per translation
// unit (anonymous namespace), one source.
#pragma once
#include "dif_common.h"

namespace {

using dif::f32x4;

// ---- row-GEMM at one head of 65..128 x 65..128 on split-bfloat16 operands (round 5: training at hidden 128) -------------
// rowgemm_wide_kernel multiplies on the fp32 matrix core and gives each 64 output columns their own workgroup: A is read once
// per column block and 128 v_mfma_f32_16x16x4_f32 run per 16 rows and block (nine launches of ~70 us per training step at
// 100,000 x 128).  Here a workgroup computes ALL (<= 128) output columns of its rows -- A read once -- with Mat resident in LDS
// as split-bf16 A fragments ([hi | lo][ft][kb][lane]: feature 16 ft + l15, k = 32 kb + 4 lg .. + 3 and the same + 16; 64 KiB at
// 128 x 128) and the rows split into hi + lo as they arrive: three v_mfma_f32_16x16x32_bf16 per (feature tile, 32 channels),
// 96 instructions of 16 cycles per 16 rows instead of 256 of 32.  The dropped lo.lo term is 2^-16 of a product (~4e-6 of the
// result; the gradients are held to 1e-4).  DIFFORMER_EXACT_FP32=1 keeps the fp32 kernel.
typedef __bf16 rg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 rg_bf16x4 __attribute__((ext_vector_type(4)));
constexpr int kRsWaves = 8;

__global__ __launch_bounds__(64 * kRsWaves) void rowgemm_split_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ Mat,
                                                                      int ldm, int mat_t, float mat_scale, const float* __restrict__ bias,
                                                                      const float* __restrict__ r, const float* __restrict__ u,
                                                                      float u_scale, const float* __restrict__ Cin, int64_t ldc,
                                                                      const float* __restrict__ beta_dev, int64_t n_rows, int K, int C,
                                                                      float* __restrict__ out, int64_t ldo,
                                                                      const float* __restrict__ norm2, const float* __restrict__ den_vec,
                                                                      float den_add) {
    extern __shared__ __attribute__((aligned(16))) rg_bf16x8 sm_frag[];          // [hi | lo][ft < 8][kb < 4][lane]
    __shared__ __attribute__((aligned(16))) float sm_bias[128], sm_u[128], sm_den[128];   // per-column / per-channel epilogue operands (zero when absent)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    // APPLY mode (den_vec != NULL; dif_simple_apply_f32 at one head of 65..128 columns, difformer.py:29-39): Mat = K^T V and
    // den_vec = sum k of the record, both scaled by 1 / (|Q| |K|) from the record's two squared norms (norm2), bias = sum v,
    // out = (A Mat + bias) / (A . den_vec + den_add), den_add = N.
    if (norm2) mat_scale *= 1.0f / (sqrtf(norm2[0]) * sqrtf(norm2[1]));
    for (int i = threadIdx.x; i < 128; i += 64 * kRsWaves) {
        sm_bias[i] = (bias && i < C) ? bias[i] : 0.f;
        sm_u[i] = (r && i < C) ? u[i] : 0.f;
        sm_den[i] = (den_vec && i < K) ? mat_scale * den_vec[i] : 0.f;
    }
    for (int e = threadIdx.x; e < 8 * 4 * 64; e += 64 * kRsWaves) {
        const int ln = e & 63, kb = (e >> 6) & 3, ft = e >> 8;
        const int c = 16 * ft + (ln & 15), k0 = 32 * kb + 4 * (ln >> 4);
        // (eight raw loads from clamped indices in flight, masked afterwards: guarded, each was a serialised round trip)
        f32x4 w0, w1;
        const int cc = c < C ? c : C - 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ka = (k0 + t < K) ? k0 + t : K - 1, kc = (k0 + 16 + t < K) ? k0 + 16 + t : K - 1;
            w0[t] = mat_t ? Mat[static_cast<int64_t>(cc) * ldm + ka] : Mat[static_cast<int64_t>(ka) * ldm + cc];
            w1[t] = mat_t ? Mat[static_cast<int64_t>(cc) * ldm + kc] : Mat[static_cast<int64_t>(kc) * ldm + cc];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            w0[t] = (c < C && k0 + t < K) ? mat_scale * w0[t] : 0.f;
            w1[t] = (c < C && k0 + 16 + t < K) ? mat_scale * w1[t] : 0.f;
        }
        const rg_bf16x4 h0 = __builtin_convertvector(w0, rg_bf16x4), h1 = __builtin_convertvector(w1, rg_bf16x4);
        const rg_bf16x4 l0 = __builtin_convertvector(w0 - __builtin_convertvector(h0, f32x4), rg_bf16x4);
        const rg_bf16x4 l1 = __builtin_convertvector(w1 - __builtin_convertvector(h1, f32x4), rg_bf16x4);
        sm_frag[e] = rg_bf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        sm_frag[8 * 4 * 64 + e] = rg_bf16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    }
    __syncthreads();
    const float beta = (Cin && beta_dev) ? *beta_dev : 1.0f;
    const int64_t n_steps = (n_rows + 15) / 16;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kRsWaves + wave;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kRsWaves;
    // Every global operand of a tile is requested a tile ahead (A) or before the tile's products (the accumulated-into rows,
    // the row scalar), as RAW loads from clamped (valid) addresses that are masked only when used: a guarded load is its own
    // exec-masked block and a masked one waits for its data where it is issued -- the first version of this kernel had 8 + 24
    // such blocks per tile, each epilogue operand a serialised round trip.
    auto load_a = [&](f32x4 (&av)[8], int64_t st) {
        const int64_t row = st * 16 + l15;
        const float* base = A + (row < n_rows ? row : n_rows - 1) * lda;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int k0 = 16 * c + 4 * lg;
            av[c] = *reinterpret_cast<const f32x4*>(base + (k0 < K ? k0 : 0));
        }
    };
    f32x4 an[8];
    load_a(an, first < n_steps ? first : n_steps - 1);
    for (int64_t st = first; st < n_steps; st += stride) {
        const int64_t row = st * 16 + l15;
        const bool row_ok = row < n_rows;
        const int64_t rowc = row_ok ? row : n_rows - 1;
        rg_bf16x8 xh[4], xl[4];
        float dpart = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 a0 = an[2 * kb], a1 = an[2 * kb + 1];
            if (!(row_ok && 32 * kb + 4 * lg < K)) a0 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!(row_ok && 32 * kb + 16 + 4 * lg < K)) a1 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (den_vec) {
                const f32x4 d0 = *reinterpret_cast<const f32x4*>(&sm_den[32 * kb + 4 * lg]), d1 = *reinterpret_cast<const f32x4*>(&sm_den[32 * kb + 16 + 4 * lg]);
#pragma unroll
                for (int t = 0; t < 4; ++t) dpart += a0[t] * d0[t] + a1[t] * d1[t];
            }
            const rg_bf16x4 h0 = __builtin_convertvector(a0, rg_bf16x4), h1 = __builtin_convertvector(a1, rg_bf16x4);
            const rg_bf16x4 l0 = __builtin_convertvector(a0 - __builtin_convertvector(h0, f32x4), rg_bf16x4);
            const rg_bf16x4 l1 = __builtin_convertvector(a1 - __builtin_convertvector(h1, f32x4), rg_bf16x4);
            xh[kb] = rg_bf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            xl[kb] = rg_bf16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        }
        load_a(an, st + stride < n_steps ? st + stride : st);               // the next rows arrive under this tile's products
        f32x4 cin[8];
        float rv = 0.f;
        if (Cin) {
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) {
                const int c0 = 16 * ft + 4 * lg;
                cin[ft] = *reinterpret_cast<const f32x4*>(Cin + rowc * ldc + (c0 < C ? c0 : 0));
            }
        }
        if (r) rv = r[rowc];
        __builtin_amdgcn_sched_barrier(0);   // requested HERE, ahead of the products
        asm volatile("" ::: "memory");       // the 64 weight fragments are re-read from LDS per tile (hoisted they would need 256 VGPRs)
        f32x4 acc[8];
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) acc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) {
                const rg_bf16x8 wh = sm_frag[(ft * 4 + kb) * 64 + lane], wl = sm_frag[8 * 4 * 64 + (ft * 4 + kb) * 64 + lane];
                acc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[kb], acc[ft], 0, 0, 0);      // small terms first
                acc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[kb], acc[ft], 0, 0, 0);
                acc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[kb], acc[ft], 0, 0, 0);
            }
        rv *= u_scale;
        float rden = 1.0f;
        if (den_vec) {                                                       // the four lanes of a row hold a quarter of its channels each
            dpart += __shfl_xor(dpart, 16, 64);
            dpart += __shfl_xor(dpart, 32, 64);
            rden = 1.0f / (dpart + den_add);
        }
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
            const int c0 = 16 * ft + 4 * lg;                                 // the lane holds out^T[c0 .. c0 + 3][row]
            f32x4 o = acc[ft] + *reinterpret_cast<const f32x4*>(&sm_bias[c0]) + rv * *reinterpret_cast<const f32x4*>(&sm_u[c0]);
            if (den_vec) o *= rden;
            if (Cin) o += beta * cin[ft];
            if (row_ok && c0 < C) *reinterpret_cast<f32x4*>(out + row * ldo + c0) = o;
        }
    }
}

// n_rows >= 1, K, C <= 128 and multiples of 4, 16-byte aligned rows: checked by the callers.
inline int rowgemm_split_launch(const char* who, hipStream_t st, const float* A, int64_t lda, const float* Mat, int ldm, int mat_t,
                                float mat_scale, const float* bias, const float* r, const float* u, float u_scale, const float* Cin,
                                int64_t ldc, const float* beta_dev, int64_t n_rows, int K, int C, float* out, int64_t ldo,
                                const float* norm2, const float* den_vec, float den_add) {
    constexpr int kFragBytes = 2 * 8 * 4 * 64 * 16;                       // 64 KiB of fragments
    static const hipError_t ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_split_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kFragBytes);
    if (ok != hipSuccess) return dif::fail(static_cast<int>(ok), "%s: LDS attribute: %s", who, hipGetErrorString(ok));
    const int64_t n_steps = (n_rows + 15) / 16;
    int64_t gw = (n_steps + 2 * kRsWaves - 1) / (2 * kRsWaves);
    if (gw > 2 * dif::kCUs) gw = 2 * dif::kCUs;
    hipLaunchKernelGGL(rowgemm_split_kernel, dim3(static_cast<unsigned>(gw)), dim3(64 * kRsWaves), kFragBytes, st, A, lda, Mat, ldm,
                       mat_t, mat_scale, bias, r, u, u_scale, Cin, ldc, beta_dev, n_rows, K, C, out, ldo, norm2, den_vec, den_add);
    return dif::launch_status("rowgemm_split_kernel");
}

}  // namespace

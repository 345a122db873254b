// MEASUREMENT FORK of difformer_amd/csrc/simple_layer_xwide.hip as of round 5 (probes, traces, alternative layouts behind -D flags; results of most are
// WRONG by design).  Not part of the product: built only by scripts/build_sliced_variants.sh (OBJ=simple_layer_xwide) into scripts/bin/.  The product
// file carries none of these branches and compiles to the same device code as this fork without flags (round 6, checked).
// a1 + a4 + a5 tail at the widths the reference's image / text scripts train with (image and text/run.sh:27: hidden 300;
// two lines at 400): the closed-form `simple` layer for 128 < max(C, D) <= 416 in ONE pass over the rows.
//
//   out = LN( alpha * ( a_s (x Mn + cn) / (x.u + cd)  +  g_s ((A_hat x) Wv^T + (A_hat 1) bv^T)  [+ x0] ) + (1 - alpha) x )
//
// Round 3 ran these widths as library GEMMs around a tail pass (x [Mn | u] by rocBLAS: 103 us for 50,000 x 300, then
// layer_tail_mix: 40 us; with a graph a second GEMM for (A_hat x) Wv^T).  At 300 columns the weights no longer fit the LDS
// (csrc/simple_layer_wide.hip keeps two 128 x 128 matrices there), so the roles swap:
//   * the ROWS stay in registers -- a wave keeps its 16-row tile as ready-split bfloat16 B fragments (hi, lo: 8 VGPRs per
//     32 channels) and ALL its output accumulators (4 VGPRs per 16 features) until the LayerNorm;
//   * the WEIGHTS stream through LDS in chunks of 32 output features: packed once per call into split-bfloat16 A fragments
//     ([chunk][hi | lo][k-block][feature tile][lane] x 16 bytes, dif_xwide_pack_f32; Wv: once per parameter version), a chunk
//     is one contiguous LDS-DMA copy of KB x 4 KiB, shared by the eight waves (128 rows) of a workgroup.
// Products on v_mfma_f32_16x16x32_bf16 with three terms per step (hi hi, hi lo, lo hi; fp32 accumulation, ~4e-6 of the
// float64 result).  The residual operand is rebuilt from the fragments (hi + lo = x to 2^-17).  DIFFORMER_EXACT_FP32=1 keeps
// the layer on the library-GEMM path.
#include <stdlib.h>
#include "dif_common.h"

namespace {

using dif::f32x4;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void split_bf16(const f32x4& v, bf16x4& hi, bf16x4& lo) {
    hi = __builtin_convertvector(v, bf16x4);
    const f32x4 back = __builtin_convertvector(hi, f32x4);
    lo = __builtin_convertvector(v - back, bf16x4);
}
__device__ __forceinline__ bf16x8 cat8(const bf16x4& a, const bf16x4& b) {
    return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

constexpr int kXWaves = 8;             // 128 rows per workgroup and weight chunk
// timing probes of measurement builds (results WRONG): 1 = no weight staging, 2 = no products at all, 3 = no LDS fragment reads
#ifndef DIF_XWIDE_PROBE
#define DIF_XWIDE_PROBE 0
#endif

struct XArgs {
    const float* x; int64_t ldx;
    const bf16x8* pm; const bf16x8* pv;     // packed Mn^T, packed Wv (nullable)
    const float* bmat; int dv;              // [C][dv]: column D = u
    const float* bias;                      // [dv]: cn | cd
    float attn_scale;
    const float* ax; int64_t ldax;
    const float* bv; const float* rs;
    float gcn_scale;
    const float* x0; int64_t ldx0;
    int residual; float alpha;
    const float* ln_w; const float* ln_b; float eps; int relu;
    float* out; int64_t ldo;
    int64_t n_rows; int C, D, KB, FC;
};

// one thread per (chunk, k-block, feature tile, lane): the lane's eight k-slots of feature 32 fc + 16 ft + l15 in k-block kb
// (channels 32 kb + 4 lg .. + 3 and the same + 16), hi and lo parts.  transposed: src is [C][ld] (Mn inside [Mn | u]),
// else [D][ld] (nn.Linear weight).
__global__ __launch_bounds__(256) void xwide_pack_kernel(const float* __restrict__ src, int64_t ld, int transposed, int C, int D,
                                                         int KB, int FC, bf16x8* __restrict__ out) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= FC * KB * 128) return;
    const int lane = id & 63, ft = (id >> 6) & 1, kb = (id >> 7) % KB, fc = (id >> 7) / KB;
    const int f = 32 * fc + 16 * ft + (lane & 15), c0 = 32 * kb + 4 * (lane >> 4);
    f32x4 w0 = zero4(), w1 = zero4();
    if (f < D) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ca = c0 + t, cb = c0 + 16 + t;
            if (ca < C) w0[t] = transposed ? src[static_cast<int64_t>(ca) * ld + f] : src[static_cast<int64_t>(f) * ld + ca];
            if (cb < C) w1[t] = transposed ? src[static_cast<int64_t>(cb) * ld + f] : src[static_cast<int64_t>(f) * ld + cb];
        }
    }
    bf16x4 h0, l0, h1, l1;
    split_bf16(w0, h0, l0);
    split_bf16(w1, h1, l1);
    const int64_t base = static_cast<int64_t>(fc) * 2 * KB * 128 + kb * 128 + ft * 64 + lane;
    out[base] = cat8(h0, h1);
    out[base + static_cast<int64_t>(KB) * 128] = cat8(l0, l1);
}

template <int KBMAX, int FCMAX>
__global__ __launch_bounds__(64 * kXWaves, 2) void simple_layer_xwide_kernel(XArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16x8 sm_w[];          // two chunk buffers, each [hi | lo][kb][ft][lane]
    __shared__ __attribute__((aligned(16))) float sm_u[KBMAX * 32], sm_cn[FCMAX * 32], sm_bv[FCMAX * 32], sm_lw[FCMAX * 32],
        sm_lb[FCMAX * 32];
    __shared__ float sm_cd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int C = a.C, D = a.D;
    constexpr int KB = KBMAX;                      // the geometry is the template's: packed weights are zero-padded up to it
    // bmat == NULL (dif_linear_xwide_f32): no denominator -- u = 0, cd = 1
    for (int i = threadIdx.x; i < KBMAX * 32; i += 64 * kXWaves)
        sm_u[i] = (a.bmat && i < C) ? a.bmat[static_cast<int64_t>(i) * a.dv + D] : 0.f;
    for (int i = threadIdx.x; i < FCMAX * 32; i += 64 * kXWaves) {
        sm_cn[i] = i < D ? a.bias[i] : 0.f;
        sm_bv[i] = (a.pv && a.rs && i < D) ? a.bv[i] * a.gcn_scale : 0.f;
        sm_lw[i] = (a.ln_w && i < D) ? a.ln_w[i] : 1.f;
        sm_lb[i] = (a.ln_b && i < D) ? a.ln_b[i] : 0.f;
    }
    if (threadIdx.x == 0) sm_cd = a.bmat ? a.bias[D] : 1.0f;
    __syncthreads();
    const float cd = sm_cd;
    const float inv_d = 1.0f / static_cast<float>(D);
    const int chunk = 2 * KB * 128;                                         // bf16x8 elements of one weight chunk (32 output features)

    // y[2 fc + ft] += W_chunk x^T for every chunk of 32 output features; all eight waves walk the chunks together.  A chunk
    // (4 KB KiB) goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: one KiB per wave instruction, no registers), into
    // one of TWO buffers: chunk fc + 1 is in flight while chunk fc is multiplied, one barrier per chunk.  (Staged through
    // registers two loads at a time the copy alone took ~8 us per chunk; a single DMA buffer left the copy's latency exposed.)
    auto stage = [&](const bf16x8* __restrict__ packed, int fc) {
        const bf16x8* src = packed + static_cast<int64_t>(fc) * chunk;
        bf16x8* dst = sm_w + (fc & 1) * chunk;
        for (int piece = wave; piece < 4 * KB; piece += kXWaves)
            __builtin_amdgcn_global_load_lds(src + piece * 64 + lane, dst + piece * 64, 16, 0, 0);
    };
    auto product = [&](f32x4 (&y)[FCMAX * 2], const bf16x8 (&xh)[KBMAX], const bf16x8 (&xl)[KBMAX], const bf16x8* __restrict__ packed) {
#if DIF_XWIDE_PROBE == 2
        return;
#endif
        __syncthreads();                                                    // everyone is done with both buffers
#if DIF_XWIDE_PROBE != 1
        stage(packed, 0);
#endif
#pragma unroll
        for (int fc = 0; fc < FCMAX; ++fc) {
            __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): this wave's pieces of chunk fc have landed
            __syncthreads();                                                // ... everybody's have, and chunk fc - 1 is done with
#if DIF_XWIDE_PROBE != 1
            if (fc + 1 < FCMAX) stage(packed, fc + 1);                      // flies under this chunk's products
#endif
            const bf16x8* w = sm_w + (fc & 1) * chunk;
#pragma unroll
            for (int kb = 0; kb < KBMAX; ++kb) {
                bf16x8 wh[2], wl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#if DIF_XWIDE_PROBE == 3
                    wh[u] = xh[(kb + u) % KBMAX];
                    wl[u] = xl[(kb + u) % KBMAX];
#else
                    wh[u] = w[kb * 128 + u * 64 + lane];
                    wl[u] = w[KB * 128 + kb * 128 + u * 64 + lane];
#endif
                }
                // small terms first; the two feature tiles alternate so that no MFMA waits on the one just issued
                y[2 * fc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[0], xl[kb], y[2 * fc], 0, 0, 0);
                y[2 * fc + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[1], xl[kb], y[2 * fc + 1], 0, 0, 0);
                y[2 * fc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[0], xh[kb], y[2 * fc], 0, 0, 0);
                y[2 * fc + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[1], xh[kb], y[2 * fc + 1], 0, 0, 0);
                y[2 * fc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[0], xh[kb], y[2 * fc], 0, 0, 0);
                y[2 * fc + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[1], xh[kb], y[2 * fc + 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);                          // keep the fragment reads next to their products
            }
        }
    };
    // a row's pieces -> split fragments (and, for x, the denominator's dot product with u).  throttle: four k-blocks of loads in
    // flight at a time -- the aggregated rows are loaded while the accumulators are live, and 2 KB loads + 2 KB fragments + the
    // accumulators do not fit 256 registers
    auto load_split = [&](bf16x8 (&xh)[KBMAX], bf16x8 (&xl)[KBMAX], const float* __restrict__ base, int64_t ld, int64_t row, bool ok,
                          float scale, float* den, bool throttle) {
#pragma unroll
        for (int kb = 0; kb < KBMAX; ++kb) {
            const int c0 = 32 * kb + 4 * lg;
            f32x4 v0 = (ok && c0 < C) ? *reinterpret_cast<const f32x4*>(base + row * ld + c0) : zero4();
            f32x4 v1 = (ok && c0 + 16 < C) ? *reinterpret_cast<const f32x4*>(base + row * ld + c0 + 16) : zero4();
            if (den) {
                const f32x4 u0 = *reinterpret_cast<const f32x4*>(&sm_u[c0]), u1 = *reinterpret_cast<const f32x4*>(&sm_u[c0 + 16]);
#pragma unroll
                for (int t = 0; t < 4; ++t) *den += v0[t] * u0[t] + v1[t] * u1[t];
            } else {
                v0 *= scale;
                v1 *= scale;
            }
            bf16x4 h0, l0, h1, l1;
            split_bf16(v0, h0, l0);
            split_bf16(v1, h1, l1);
            xh[kb] = cat8(h0, h1);
            xl[kb] = cat8(l0, l1);
            if (throttle && (kb & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int64_t n_blocks = (a.n_rows + 16 * kXWaves - 1) / (16 * kXWaves);
    for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int64_t row = (blk * kXWaves + wave) * 16 + l15;
        const bool row_ok = row < a.n_rows;
        bf16x8 xh[KBMAX], xl[KBMAX];
        float den = 0.f;
        load_split(xh, xl, a.x, a.ldx, row, row_ok, 1.0f, &den, false);
        den += __shfl_xor(den, 16, 64);
        den += __shfl_xor(den, 32, 64);
        const float rden = 1.0f / (den + cd);
        f32x4 y[FCMAX * 2];
#pragma unroll
        for (int i = 0; i < FCMAX * 2; ++i) y[i] = zero4();
        product(y, xh, xl, a.pm);
        // alpha (y_att + y_graph + x0) + (1 - alpha) x, the residual first: the x fragments are then free for A_hat x
        const float mixw = a.residual ? a.alpha : 1.0f;
        const float as_rden = a.attn_scale * rden * mixw, cn_w = a.attn_scale * rden * mixw, keep = a.residual ? 1.0f - a.alpha : 0.f;
#pragma unroll
        for (int ft = 0; ft < FCMAX * 2; ++ft) {
            {
                f32x4 xr = zero4();
                if (ft < 2 * KBMAX && a.residual) {         // feature 16 ft + 4 lg + r = channel: k-block ft / 2, half ft % 2
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        xr[r] = static_cast<float>(xh[ft / 2][4 * (ft & 1) + r]) + static_cast<float>(xl[ft / 2][4 * (ft & 1) + r]);
                }
                y[ft] = y[ft] * as_rden + *reinterpret_cast<const f32x4*>(&sm_cn[16 * ft + 4 * lg]) * cn_w + keep * xr;
            }
        }
        if (a.ax) {
            const float gw = a.gcn_scale * mixw;
            load_split(xh, xl, a.ax, a.ldax, row, row_ok, gw, nullptr, true);
            if (a.pv) {
                const float rsv = (a.rs && row_ok) ? a.rs[row] * mixw : 0.f;
#pragma unroll
                for (int ft = 0; ft < FCMAX * 2; ++ft)
                    y[ft] += *reinterpret_cast<const f32x4*>(&sm_bv[16 * ft + 4 * lg]) * rsv;
                product(y, xh, xl, a.pv);
            } else {          // use_weight = False: the aggregated rows are the graph term (C == D)
#pragma unroll
                for (int ft = 0; ft < FCMAX * 2; ++ft) {
                    if (ft < 2 * KBMAX) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            y[ft][r] += static_cast<float>(xh[ft / 2][4 * (ft & 1) + r]) + static_cast<float>(xl[ft / 2][4 * (ft & 1) + r]);
                    }
                }
            }
        }
        if (a.x0) {
#pragma unroll
            for (int ft = 0; ft < FCMAX * 2; ++ft) {
                const int f = 16 * ft + 4 * lg;
                if (row_ok && f < D) y[ft] += mixw * *reinterpret_cast<const f32x4*>(a.x0 + row * a.ldx0 + f);
            }
        }
        if (a.ln_w) {
            float mu = 0.f;
#pragma unroll
            for (int ft = 0; ft < FCMAX * 2; ++ft)
                if (16 * ft + 4 * lg < D) mu += (y[ft][0] + y[ft][1]) + (y[ft][2] + y[ft][3]);
            mu += __shfl_xor(mu, 16, 64);
            mu += __shfl_xor(mu, 32, 64);
            mu *= inv_d;
            float var = 0.f;
#pragma unroll
            for (int ft = 0; ft < FCMAX * 2; ++ft) {
                {
                    const bool ok = 16 * ft + 4 * lg < D;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dz = ok ? y[ft][r] - mu : 0.f;
                        y[ft][r] = dz;
                        var += dz * dz;
                    }
                }
            }
            var += __shfl_xor(var, 16, 64);
            var += __shfl_xor(var, 32, 64);
            const float rstd = 1.0f / sqrtf(var * inv_d + a.eps);
#pragma unroll
            for (int ft = 0; ft < FCMAX * 2; ++ft)
                
                    y[ft] = y[ft] * rstd * *reinterpret_cast<const f32x4*>(&sm_lw[16 * ft + 4 * lg]) +
                            *reinterpret_cast<const f32x4*>(&sm_lb[16 * ft + 4 * lg]);
        }
#pragma unroll
        for (int ft = 0; ft < FCMAX * 2; ++ft) {
            const int f = 16 * ft + 4 * lg;
            if (row_ok && f < D) {
                f32x4 v = y[ft];
                if (a.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                *reinterpret_cast<f32x4*>(a.out + row * a.ldo + f) = v;
            }
        }
    }
}

bool xwide_covers(int C, int D) { return C > 0 && D > 0 && C <= 416 && D <= 416 && C % 4 == 0 && D % 4 == 0; }

// k-blocks (of 32 channels) x feature chunks (of 32) the kernel is instantiated for: squares that hold max(C, D) (the layer:
// C == D), and two oblong ones for the input Linear's halves (256 channels -> 300 / 400 features: a square would multiply
// 25 % / 62 % zero padding).  The packed weights are laid out (and zero-padded) for the geometry.
struct XGeo { int kb, fc; };
XGeo xwide_geometry(int C, int D) {
    auto bucket = [](int v) { const int need = (v + 31) / 32; return need <= 6 ? 6 : (need <= 8 ? 8 : (need <= 10 ? 10 : 13)); };
    const int kb = bucket(C), fc = bucket(D);
    if (kb == 8 && (fc == 10 || fc == 13)) return {kb, fc};
    const int g = kb > fc ? kb : fc;
    return {g, g};
}

template <int KB, int FC>
int xwide_launch(const XArgs& a, unsigned P, hipStream_t st) {
    constexpr int lds = 2 * KB * 2 * 128 * 16;             // two chunk buffers
    static const hipError_t he = hipFuncSetAttribute(reinterpret_cast<const void*>(&simple_layer_xwide_kernel<KB, FC>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (he != hipSuccess) return dif::fail(static_cast<int>(he), "dif_simple_layer_xwide: LDS attribute: %s", hipGetErrorString(he));
    hipLaunchKernelGGL((simple_layer_xwide_kernel<KB, FC>), dim3(P), dim3(64 * kXWaves), lds, st, a);
    return dif::launch_status("simple_layer_xwide_kernel");
}
int xwide_dispatch(const XArgs& a, hipStream_t st) {
    const int64_t blocks = (a.n_rows + 16 * kXWaves - 1) / (16 * kXWaves);
    const unsigned P = static_cast<unsigned>(blocks < dif::kCUs ? blocks : dif::kCUs);
    switch (a.KB * 100 + a.FC) {
        case 606: return xwide_launch<6, 6>(a, P, st);
        case 808: return xwide_launch<8, 8>(a, P, st);
        case 810: return xwide_launch<8, 10>(a, P, st);
        case 813: return xwide_launch<8, 13>(a, P, st);
        case 1010: return xwide_launch<10, 10>(a, P, st);
        default: return xwide_launch<13, 13>(a, P, st);
    }
}

}  // namespace

extern "C" int64_t dif_xwide_packed_bytes(int C, int D) {
    if (!xwide_covers(C, D)) return 0;
    const XGeo g = xwide_geometry(C, D);
    return static_cast<int64_t>(g.fc) * 2 * g.kb * 128 * 16;
}

// src: [C][ld] with the matrix in columns [0, D) when transposed (the [Mn | u] operand of dif_wide_scale_f64), else [D][ld]
// (an nn.Linear weight [D, C]).  packed: dif_xwide_packed_bytes(C, D), 16-byte aligned.
extern "C" int dif_xwide_pack_f32(const float* src, int64_t ld, int transposed, int C, int D, void* packed, dif_stream_t stream) {
    DIF_REQUIRE(src && packed && xwide_covers(C, D) && dif::aligned16(packed), DIF_E_BADARG,
                "dif_xwide_pack: needs src, a 16-byte aligned buffer and C, D <= 416, multiples of 4");
    DIF_REQUIRE(ld >= (transposed ? D : C), DIF_E_BADARG, "dif_xwide_pack: leading dimension smaller than a row");
    const XGeo g = xwide_geometry(C, D);
    hipLaunchKernelGGL(xwide_pack_kernel, dim3(static_cast<unsigned>((g.kb * g.fc + 1) / 2)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       src, ld, transposed, C, D, g.kb, g.fc, static_cast<bf16x8*>(packed));
    return dif::launch_status("xwide_pack_kernel");
}

// Closed-form `simple` layer for 128 < max(C, D) <= 416 in one pass (see the head of this file).  packed_m = dif_xwide_pack_f32 of
// bmat (transposed), packed_v = of Wv (NULL: no graph, or use_weight = False with C == D); bmat / dv / bias as for
// dif_simple_layer_wide_f32 (u = column D of bmat, cn | cd = bias).
extern "C" int dif_simple_layer_xwide_f32(const float* x, int64_t ldx, int64_t n_rows, int C, int D, const void* packed_m,
                                          const void* packed_v, const float* bmat, int dv, const float* bias, float attn_scale,
                                          const float* ax, int64_t ldax, const float* bv, const float* row_sums, float gcn_scale,
                                          const float* x0, int64_t ldx0, int residual, float alpha, const float* ln_weight,
                                          const float* ln_bias, float ln_eps, int relu, float* out, int64_t ldo,
                                          dif_stream_t stream) {
    DIF_REQUIRE(x && packed_m && bmat && bias && out && n_rows > 0, DIF_E_BADARG, "dif_simple_layer_xwide: null pointer or no rows");
    DIF_REQUIRE(xwide_covers(C, D) && dv >= D + 1, DIF_E_SHAPE,
                "dif_simple_layer_xwide: covers C, D <= 416, multiples of 4, dv >= D + 1 (got C = %d, D = %d, dv = %d)", C, D, dv);
    DIF_REQUIRE((ln_weight == nullptr) == (ln_bias == nullptr), DIF_E_BADARG, "dif_simple_layer_xwide: ln_weight and ln_bias must be given together");
    DIF_REQUIRE((!packed_v || (ax && bv)) && (!row_sums || packed_v), DIF_E_BADARG,
                "dif_simple_layer_xwide: packed_v needs ax and bv; row_sums only with packed_v");
    DIF_REQUIRE(!(ax && !packed_v) || C == D, DIF_E_SHAPE, "dif_simple_layer_xwide: use_weight = False needs C == D");
    DIF_REQUIRE(!residual || C == D, DIF_E_SHAPE, "dif_simple_layer_xwide: the residual mixes with x itself (C == D)");
    DIF_REQUIRE(ldx >= C && ldx % 4 == 0 && dif::aligned16(x) && ldo >= D && ldo % 4 == 0 && dif::aligned16(out) &&
                (!ax || (ldax >= C && ldax % 4 == 0 && dif::aligned16(ax))) && (!x0 || (ldx0 >= D && ldx0 % 4 == 0 && dif::aligned16(x0))) &&
                dif::aligned16(packed_m) && dif::aligned16(packed_v), DIF_E_BADARG,
                "dif_simple_layer_xwide: rows and packed weights must be 16-byte aligned with ld >= the row length");
    const XGeo g = xwide_geometry(C, D);
    const XArgs a = {x, ldx, static_cast<const bf16x8*>(packed_m), static_cast<const bf16x8*>(packed_v), bmat, dv, bias, attn_scale,
                     ax, ldax, bv, row_sums, gcn_scale, x0, ldx0, residual, alpha, ln_weight, ln_bias, ln_eps, relu, out, ldo,
                     n_rows, C, D, g.kb, g.fc};
    return xwide_dispatch(a, static_cast<hipStream_t>(stream));
}

// nn.Linear (-> LayerNorm) (-> ReLU) with a WIDE result (64 < D <= 416; image and text/run.sh:27: the 512 -> 300 input layer of
// difformer.py:188-191) on the same kernel: the layer's two accumulating products are the two halves of the input channels,
//   out = LN( x[:, :Ch] Wa^T + x[:, Ch:] Wb^T + bias ),   Ch = C_in / 2 <= 416,   or one product for C_in <= 416 (packed_b NULL)
// packed_a / packed_b = dif_xwide_pack_f32(W + 0 / + Ch, ld = C_in, transposed = 0, Ch, D).  Replaces the library GEMM + tail
// pass (143 + 26 us at 50,000 x 512 -> 300).
extern "C" int dif_linear_xwide_f32(const float* x, int64_t ldx, int64_t n_rows, int C_in, const void* packed_a, const void* packed_b,
                                    const float* bias, int D, const float* ln_weight, const float* ln_bias, float ln_eps, int relu,
                                    float* out, int64_t ldo, dif_stream_t stream) {
    DIF_REQUIRE(x && packed_a && bias && out && n_rows > 0, DIF_E_BADARG, "dif_linear_xwide: null pointer or no rows");
    const int Ch = packed_b ? C_in / 2 : C_in;
    DIF_REQUIRE(C_in > 0 && (!packed_b || C_in % 2 == 0) && xwide_covers(Ch, D), DIF_E_SHAPE,
                "dif_linear_xwide: covers D <= 416 and C_in <= 416 (one product) or C_in / 2 <= 416 (two), multiples of 4 (got %d -> %d)",
                C_in, D);
    DIF_REQUIRE((ln_weight == nullptr) == (ln_bias == nullptr), DIF_E_BADARG, "dif_linear_xwide: ln_weight and ln_bias must be given together");
    DIF_REQUIRE(ldx >= C_in && ldx % 4 == 0 && dif::aligned16(x) && ldo >= D && ldo % 4 == 0 && dif::aligned16(out) &&
                dif::aligned16(packed_a) && dif::aligned16(packed_b), DIF_E_BADARG,
                "dif_linear_xwide: rows and packed weights must be 16-byte aligned with ld >= the row length");
    const XGeo g = xwide_geometry(Ch, D);
    const XArgs a = {x, ldx, static_cast<const bf16x8*>(packed_a), static_cast<const bf16x8*>(packed_b), nullptr, 0, bias, 1.0f,
                     packed_b ? x + Ch : nullptr, ldx, bias, nullptr, 1.0f, nullptr, 0, 0, 1.0f, ln_weight, ln_bias, ln_eps, relu,
                     out, ldo, n_rows, Ch, D, g.kb, g.fc};
    return xwide_dispatch(a, static_cast<hipStream_t>(stream));
}

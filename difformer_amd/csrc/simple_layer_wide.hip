// a1 + a4 + a5 tail at the widths the reference's scripts train with on Pokec (node classification/run.sh:42-44: hidden 128):
// the closed-form `simple` layer for 64 < C, D <= 128 in ONE pass over the rows.
//
//   out = LN( alpha * ( a_s (x Mn + cn) / (x.u + cd)  +  g_s ((A_hat x) Wv^T + (A_hat 1) bv^T)  [+ x0] ) + (1 - alpha) x )
//   (difformer.py:18-39 through the Gram record -- csrc/simple_layer.hip has the algebra --, :75-78, :130-140, :200-203)
//
// Round 3 ran this width on the operator path (q | k | v projected by the vendor GEMM, stand-alone reduce / apply / tail
// kernels: ~290 us per layer at 100,000 rows, 36 % of it in the library) because the two row products of the closed form,
// as library GEMMs around a tail pass, cost the same.  Here both 128 x 128 products, the division, the combine, the residual
// and the LayerNorm run on the rows while they are in registers: x and A_hat x are read once, out is written once.
//
// Matrix core: both products run TRANSPOSED on split-bfloat16 operands (v = hi + lo, three v_mfma_f32_16x16x32_bf16 per
// product step with fp32 accumulation, ~4e-6 of the float64 result; csrc/skinny_linear.hip has the error argument) -- on the
// fp32 MFMA the 512 products of a 16-row tile would take 16 k cycles per SIMD and bound the kernel at ~50 us for 100,000
// rows; split they take 3.5 k and the rows' bytes bound it.  DIFFORMER_EXACT_FP32=1 keeps the layer off this kernel.
//   A (weights, from LDS)  [i = feature 16 ft + l15][k]      B (rows, from registers)  [k][j = row l15]
//   k-slots of lane group lg in the 32-channel block h: channels 32 h + 4 lg .. + 3 and 32 h + 16 + 4 lg .. + 3 -- what a lane
//   holds after two 16-byte loads of its row -- so a lane ends with features 16 ft + 4 lg .. + 3 of ONE row: the residual
//   operand is the x fragment already in registers, LayerNorm folds over the four lane groups, rows leave as 16-byte stores.
// LDS: Mn^T and Wv as ready-made A fragments, [part hi | lo][h][ft][lane] x 16 bytes = 64 KiB each (one 16-wave workgroup per CU).
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

constexpr int kWideWaves = 16;
constexpr int kFrags = 4 * 8 * 64;          // (h, ft, lane) fragments of one 128 x 128 matrix part

struct WideArgs {
    const float* x; int64_t ldx;
    const float* bmat; int dv;              // [C][dv]: columns [0, D) = Mn, column D = u
    const float* bias;                      // [dv]: cn | cd
    float attn_scale;
    const float* ax; int64_t ldax;          // A_hat x (unscaled), nullable
    const float* Wv; const float* bv;       // [D][C], [D]; nullable (use_weight = False: the aggregated rows are the graph term)
    const float* rs;                        // A_hat 1 per row (with Wv)
    float gcn_scale;
    const float* x0; int64_t ldx0;
    int residual; float alpha;
    const float* ln_w; const float* ln_b; float eps; int relu;
    float* out; int64_t ldo;
    int64_t n_rows; int C, D;
};

// y[ft] += W_tile x^T over the lane's eight row pieces xa (split once per 32-channel block)
__device__ __forceinline__ void project_wide(f32x4 (&y)[8], const f32x4 (&xa)[8], const bf16x8* __restrict__ w, int lane) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        bf16x4 h0, l0, h1, l1;
        split_bf16(xa[2 * h], h0, l0);
        split_bf16(xa[2 * h + 1], h1, l1);
        const bf16x8 xh = cat8(h0, h1), xl = cat8(l0, l1);
        // two feature tiles at a time (four fragment reads in flight, six MFMAs): left to itself the scheduler hoists all 64
        // fragment reads of the product -- 256 registers -- above the first MFMA and spills
#pragma unroll
        for (int fp = 0; fp < 4; ++fp) {
            bf16x8 wh[2], wl[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                wh[u] = w[(h * 8 + 2 * fp + u) * 64 + lane];
                wl[u] = w[kFrags + (h * 8 + 2 * fp + u) * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ft = 2 * fp + u;
                y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[u], xl, y[ft], 0, 0, 0);      // small terms first
                y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[u], xh, y[ft], 0, 0, 0);
                y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[u], xh, y[ft], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <bool GRAPH_W>
__global__ __launch_bounds__(64 * kWideWaves, 4) void simple_layer_wide_kernel(WideArgs a) {
    __shared__ __attribute__((aligned(16))) bf16x8 sm_m[2 * kFrags];                    // Mn^T: hi | lo
    __shared__ __attribute__((aligned(16))) bf16x8 sm_v[GRAPH_W ? 2 * kFrags : 1];      // Wv:   hi | lo
    __shared__ __attribute__((aligned(16))) float sm_cn[128], sm_u[128], sm_bv[128], sm_lw[128], sm_lb[128];
    __shared__ float sm_cd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int C = a.C, D = a.D;
    // A fragments: thread -> (h, ft, lane'): feature 16 ft + l15', channels 32 h + 4 lg' + t and + 16
    for (int e = threadIdx.x; e < kFrags; e += 64 * kWideWaves) {
        const int ln = e & 63, ft = (e >> 6) & 7, h = e >> 9;
        const int f = 16 * ft + (ln & 15), c0 = 32 * h + 4 * (ln >> 4);
        f32x4 m0 = zero4(), m1 = zero4(), v0 = zero4(), v1 = zero4();
        if (f < D) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (c0 + t < C) m0[t] = a.bmat[static_cast<int64_t>(c0 + t) * a.dv + f];
                if (c0 + 16 + t < C) m1[t] = a.bmat[static_cast<int64_t>(c0 + 16 + t) * a.dv + f];
                if (GRAPH_W) {
                    if (c0 + t < C) v0[t] = a.Wv[static_cast<int64_t>(f) * C + c0 + t];
                    if (c0 + 16 + t < C) v1[t] = a.Wv[static_cast<int64_t>(f) * C + c0 + 16 + t];
                }
            }
        }
        bf16x4 h0, l0, h1, l1;
        split_bf16(m0, h0, l0);
        split_bf16(m1, h1, l1);
        sm_m[e] = cat8(h0, h1);
        sm_m[kFrags + e] = cat8(l0, l1);
        if (GRAPH_W) {
            split_bf16(v0, h0, l0);
            split_bf16(v1, h1, l1);
            sm_v[e] = cat8(h0, h1);
            sm_v[kFrags + e] = cat8(l0, l1);
        }
    }
    if (threadIdx.x < 128) {
        const int i = threadIdx.x;
        sm_cn[i] = i < D ? a.bias[i] * a.attn_scale : 0.f;           // a_s folded into the numerator's constant ...
        sm_u[i] = i < C ? a.bmat[static_cast<int64_t>(i) * a.dv + D] : 0.f;
        sm_bv[i] = (GRAPH_W && a.rs && i < D) ? a.bv[i] * a.gcn_scale : 0.f;
        sm_lw[i] = (a.ln_w && i < D) ? a.ln_w[i] : 1.f;
        sm_lb[i] = (a.ln_b && i < D) ? a.ln_b[i] : 0.f;
    }
    if (threadIdx.x == 0) sm_cd = a.bias[D];
    __syncthreads();
    const float cd = sm_cd;
    const float inv_d = 1.0f / static_cast<float>(D);
    const int64_t n_tiles = (a.n_rows + 15) / 16;
    for (int64_t tile = static_cast<int64_t>(blockIdx.x) * kWideWaves + wave; tile < n_tiles;
         tile += static_cast<int64_t>(gridDim.x) * kWideWaves) {
        const int64_t row = tile * 16 + l15;
        const bool row_ok = row < a.n_rows;
        f32x4 xa[8];
#pragma unroll
        for (int cq = 0; cq < 8; ++cq) {
            const int c = 16 * cq + 4 * lg;
            xa[cq] = (row_ok && c < C) ? *reinterpret_cast<const f32x4*>(a.x + row * a.ldx + c) : zero4();
        }
        float den = 0.f;
#pragma unroll
        for (int cq = 0; cq < 8; ++cq) {
            const f32x4 uu = *reinterpret_cast<const f32x4*>(&sm_u[16 * cq + 4 * lg]);
#pragma unroll
            for (int t = 0; t < 4; ++t) den += xa[cq][t] * uu[t];
        }
        den += __shfl_xor(den, 16, 64);
        den += __shfl_xor(den, 32, 64);
        const float rden = 1.0f / (den + cd);
        f32x4 y[8];
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) y[ft] = zero4();
        project_wide(y, xa, sm_m, lane);
        // alpha (y_att + y_graph + x0) + (1 - alpha) x, with the residual taken FIRST: the x fragments die before the
        // aggregated rows are loaded (three live 32-register sets would spill at 128 VGPRs)
        const float mixw = a.residual ? a.alpha : 1.0f;
        const float as_rden = a.attn_scale * rden * mixw;              // ... and into the product's scale
        const float cn_w = rden * mixw, keep = a.residual ? 1.0f - a.alpha : 0.f;
#pragma unroll
        for (int ft = 0; ft < 8; ++ft)
            y[ft] = y[ft] * as_rden + *reinterpret_cast<const f32x4*>(&sm_cn[16 * ft + 4 * lg]) * cn_w + keep * xa[ft];
        if (a.ax) {
            f32x4 ga[8];
            const float gw = a.gcn_scale * mixw;
#pragma unroll
            for (int cq = 0; cq < 8; ++cq) {
                const int c = 16 * cq + 4 * lg;
                ga[cq] = (row_ok && c < C) ? *reinterpret_cast<const f32x4*>(a.ax + row * a.ldax + c) * gw : zero4();
            }
            if (GRAPH_W) {
                const float rsv = (a.rs && row_ok) ? a.rs[row] * mixw : 0.f;
#pragma unroll
                for (int ft = 0; ft < 8; ++ft) y[ft] += *reinterpret_cast<const f32x4*>(&sm_bv[16 * ft + 4 * lg]) * rsv;
                project_wide(y, ga, sm_v, lane);
            } else {          // use_weight = False: the aggregated rows are the graph term (C == D), same layout
#pragma unroll
                for (int ft = 0; ft < 8; ++ft) y[ft] += ga[ft];
            }
        }
        if (a.x0) {
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) {
                const int f = 16 * ft + 4 * lg;
                if (row_ok && f < D) y[ft] += mixw * *reinterpret_cast<const f32x4*>(a.x0 + row * a.ldx0 + f);
            }
        }
        if (a.ln_w) {
            float mu = 0.f;
#pragma unroll
            for (int ft = 0; ft < 8; ++ft)
#pragma unroll
                for (int r = 0; r < 4; ++r) mu += (16 * ft + 4 * lg < D) ? y[ft][r] : 0.f;
            mu += __shfl_xor(mu, 16, 64);
            mu += __shfl_xor(mu, 32, 64);
            mu *= inv_d;
            float var = 0.f;
#pragma unroll
            for (int ft = 0; ft < 8; ++ft)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dz = (16 * ft + 4 * lg < D) ? y[ft][r] - mu : 0.f;
                    y[ft][r] = dz;
                    var += dz * dz;
                }
            var += __shfl_xor(var, 16, 64);
            var += __shfl_xor(var, 32, 64);
            const float rstd = 1.0f / sqrtf(var * inv_d + a.eps);
#pragma unroll
            for (int ft = 0; ft < 8; ++ft)
                y[ft] = y[ft] * rstd * *reinterpret_cast<const f32x4*>(&sm_lw[16 * ft + 4 * lg]) +
                        *reinterpret_cast<const f32x4*>(&sm_lb[16 * ft + 4 * lg]);
        }
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
            const int f = 16 * ft + 4 * lg;
            f32x4 v = y[ft];
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (row_ok && f < D) *reinterpret_cast<f32x4*>(a.out + row * a.ldo + f) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Gram record X^T X | sum x for 64 < C <= 128 (the record of the closed form at hidden 128).  csrc/simple_layer.hip's
// gram_kernel widened: a wave reads four whole rows per step (lane (lg, l15): row base + lg, columns 4 l15 .. + 3 and
// 64 + 4 l15 .. + 3), which is the coalesced stream AND the A / B operand of v_mfma_f32_16x16x4_f32 contracting over rows
// -- the lane's eight components p select the 16-column group {4 i + p % 4 + 64 (p / 4)}; the 36 products with pa <= pb
// cover the upper half of X^T X, the record write mirrors them.  One pass over x on the exact fp32 MFMA (the coefficient
// algebra downstream cancels large terms: no split operands here).  Round 3's form for this width (stage 1 of the
// attention reduce over three 64 x 64 blocks, x read three times) took 49 us at 100,000 rows.
// ------------------------------------------------------------------------------------------------------------
constexpr int kG128Waves = 8;

__global__ __launch_bounds__(64 * kG128Waves, 2) void gram128_kernel(const float* __restrict__ x, int64_t ldx, int64_t n_rows, int C,
                                                                     float* __restrict__ ws, int64_t ws_stride) {
    __shared__ float sm_f[4 * 36 * 64];               // fold buffer: 9 accumulators (36 registers) x 64 lanes for up to 4 waves
    __shared__ float sm_s[kG128Waves][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const bool ok0 = 4 * l15 < C, ok1 = 64 + 4 * l15 < C;
    f32x4 acc[36];
#pragma unroll
    for (int a = 0; a < 36; ++a) acc[a] = zero4();
    f32x4 sx0 = zero4(), sx1 = zero4();
    const int64_t n16 = (n_rows + 15) / 16;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kG128Waves + wave;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kG128Waves;
    auto load16 = [&](f32x4 (&xv)[4][2], int64_t tile) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t row = tile * 16 + 4 * u + lg;
            const float* p = x + row * ldx + 4 * l15;
            xv[u][0] = (row < n_rows && ok0) ? *reinterpret_cast<const f32x4*>(p) : zero4();
            xv[u][1] = (row < n_rows && ok1) ? *reinterpret_cast<const f32x4*>(p + 64) : zero4();
        }
    };
    f32x4 nxt[4][2];
    if (first < n16) load16(nxt, first);
    for (int64_t tile = first; tile < n16; tile += stride) {
        f32x4 xv[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) { xv[u][0] = nxt[u][0]; xv[u][1] = nxt[u][1]; }
        if (tile + stride < n16) load16(nxt, tile + stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sx0 += xv[u][0];
            sx1 += xv[u][1];
            const float p[8] = {xv[u][0][0], xv[u][0][1], xv[u][0][2], xv[u][0][3], xv[u][1][0], xv[u][1][1], xv[u][1][2], xv[u][1][3]};
            int a = 0;
#pragma unroll
            for (int pa = 0; pa < 8; ++pa)
#pragma unroll
                for (int pb = pa; pb < 8; ++pb, ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[pa], p[pb], acc[a], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float a0 = sx0[t], a1 = sx1[t];
        a0 += __shfl_xor(a0, 16, 64); a0 += __shfl_xor(a0, 32, 64);
        a1 += __shfl_xor(a1, 16, 64); a1 += __shfl_xor(a1, 32, 64);
        if (lg == 0) { sm_s[wave][4 * l15 + t] = a0; sm_s[wave][64 + 4 * l15 + t] = a1; }
    }
    // fold the eight waves in registers, nine accumulators at a time: ((w0 + w4) + (w2 + w6)) + ((w1 + w5) + (w3 + w7))
#pragma unroll
    for (int half = kG128Waves / 2; half >= 1; half >>= 1) {
#pragma unroll
        for (int c0 = 0; c0 < 36; c0 += 9) {
            if (wave >= half && wave < 2 * half) {
#pragma unroll
                for (int i = 0; i < 9; ++i)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) sm_f[((wave - half) * 36 + i * 4 + reg) * 64 + lane] = acc[c0 + i][reg];
            }
            __syncthreads();
            if (wave < half) {
#pragma unroll
                for (int i = 0; i < 9; ++i)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) acc[c0 + i][reg] += sm_f[(wave * 36 + i * 4 + reg) * 64 + lane];
            }
            __syncthreads();
        }
    }
    float* rec = ws + static_cast<int64_t>(blockIdx.x) * ws_stride;
    if (wave == 0) {
        int a = 0;
#pragma unroll
        for (int pa = 0; pa < 8; ++pa)
#pragma unroll
            for (int pb = pa; pb < 8; ++pb, ++a)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int gi = 4 * (4 * lg + reg) + (pa & 3) + 64 * (pa >> 2), gj = 4 * l15 + (pb & 3) + 64 * (pb >> 2);
                    if (gi < C && gj < C) {
                        rec[gi * C + gj] = acc[a][reg];
                        if (pa != pb) rec[gj * C + gi] = acc[a][reg];
                    }
                }
    } else if (wave == 1) {
        for (int c = lane; c < C; c += 64) {
            float a = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < kG128Waves; ++w2) a += sm_s[w2][c];
            rec[C * C + c] = a;
        }
    }
}

int gram128_chunks(int64_t n_rows) {
    const int64_t tiles = (n_rows + 15) / 16;
    int64_t p = (tiles + 2 * kG128Waves - 1) / (2 * kG128Waves);       // at least two tiles per wave, every CU busy from ~65,000 rows
    if (p > dif::kCUs) p = dif::kCUs;
    return static_cast<int>(p < 1 ? 1 : p);
}

}  // namespace

extern "C" size_t dif_gram128_workspace_bytes(int64_t n_rows, int C) {
    if (n_rows <= 0 || C <= 64 || C > 128) return 0;
    const size_t rec = (static_cast<size_t>(C) * C + C + 3) & ~size_t(3);
    return rec * sizeof(float) * static_cast<size_t>(gram128_chunks(n_rows));
}

// record = [X^T X: C x C row-major, all of it][sum x: C] (+ whatever the caller's record holds beyond): the layout
// dif_gram_sym_f32 leaves, for 64 < C <= 128, C % 4 == 0, rows 16-byte aligned.
extern "C" int dif_gram128_f32(const float* x, int64_t ldx, int64_t n_rows, int C, float* record, void* workspace,
                               size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(x && record && workspace && n_rows > 0, DIF_E_BADARG, "dif_gram128: null pointer or no rows");
    DIF_REQUIRE(C > 64 && C <= 128 && C % 4 == 0, DIF_E_SHAPE, "dif_gram128: covers 64 < C <= 128, C %% 4 == 0 (got %d)", C);
    DIF_REQUIRE(ldx >= C && ldx % 4 == 0 && dif::aligned16(x) && dif::aligned16(workspace), DIF_E_BADARG,
                "dif_gram128: rows of x and the workspace must be 16-byte aligned");
    DIF_REQUIRE(workspace_bytes >= dif_gram128_workspace_bytes(n_rows, C), DIF_E_WORKSPACE, "dif_gram128: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int P = gram128_chunks(n_rows);
    const int64_t rec = (static_cast<int64_t>(C) * C + C + 3) & ~int64_t(3);
    float* ws = static_cast<float*>(workspace);
    hipLaunchKernelGGL(gram128_kernel, dim3(P), dim3(64 * kG128Waves), 0, st, x, ldx, n_rows, C, ws, rec);
    if (int rc = dif::launch_status("gram128_kernel")) return rc;
    return dif::launch_record_finalize(ws, P, rec, C * C + C, 0, record, st);
}

// Closed-form `simple` layer for 64 < max(C, D) <= 128 (C % 4 == 0, D % 4 == 0) in one pass; see the head of this file.
//   bmat [C][dv] (dv >= D + 1; the host's dif_wide_scale_f64 output: columns [0, D) = s Mn, column D = s u), bias [dv] = cn | cd;
//   ax (nullable) = A_hat x unscaled [n, C]; Wv [D][C], bv [D], rs [n] (nullable together: use_weight = False needs C == D);
//   x0 (nullable) [n, D]; residual mixes with x itself (needs C == D).
extern "C" int dif_simple_layer_wide_f32(const float* x, int64_t ldx, int64_t n_rows, int C, int D, const float* bmat, int dv,
                                         const float* bias, float attn_scale, const float* ax, int64_t ldax, const float* Wv,
                                         const float* bv, const float* row_sums, float gcn_scale, const float* x0, int64_t ldx0,
                                         int residual, float alpha, const float* ln_weight, const float* ln_bias, float ln_eps,
                                         int relu, float* out, int64_t ldo, dif_stream_t stream) {
    DIF_REQUIRE(x && bmat && bias && out && n_rows > 0, DIF_E_BADARG, "dif_simple_layer_wide: null pointer or no rows");
    DIF_REQUIRE(C > 0 && D > 0 && C <= 128 && D <= 128 && C % 4 == 0 && D % 4 == 0 && dv >= D + 1, DIF_E_SHAPE,
                "dif_simple_layer_wide: covers C, D <= 128, multiples of 4, dv >= D + 1 (got C = %d, D = %d, dv = %d)", C, D, dv);
    DIF_REQUIRE((ln_weight == nullptr) == (ln_bias == nullptr), DIF_E_BADARG, "dif_simple_layer_wide: ln_weight and ln_bias must be given together");
    DIF_REQUIRE((Wv == nullptr) == (bv == nullptr) && (!row_sums || Wv) && (!Wv || ax), DIF_E_BADARG,
                "dif_simple_layer_wide: Wv and bv come together, with ax; row_sums only with Wv");
    DIF_REQUIRE(!(ax && !Wv) || C == D, DIF_E_SHAPE, "dif_simple_layer_wide: use_weight = False needs C == D");
    DIF_REQUIRE(!residual || C == D, DIF_E_SHAPE, "dif_simple_layer_wide: the residual mixes with x itself (C == D)");
    DIF_REQUIRE(ldx >= C && ldx % 4 == 0 && dif::aligned16(x) && ldo >= D && ldo % 4 == 0 && dif::aligned16(out) &&
                (!ax || (ldax >= C && ldax % 4 == 0 && dif::aligned16(ax))) && (!x0 || (ldx0 >= D && ldx0 % 4 == 0 && dif::aligned16(x0))),
                DIF_E_BADARG, "dif_simple_layer_wide: rows must be 16-byte aligned with ld >= the row length");
    const int64_t tiles = (n_rows + 15) / 16;
    int64_t P = (tiles + kWideWaves - 1) / kWideWaves;
    if (P > dif::kCUs) P = dif::kCUs;                      // 128 KiB of weights in LDS: one workgroup per CU
    const WideArgs a = {x, ldx, bmat, dv, bias, attn_scale, ax, ldax, Wv, bv, row_sums, gcn_scale, x0, ldx0, residual, alpha,
                        ln_weight, ln_bias, ln_eps, relu, out, ldo, n_rows, C, D};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (Wv) hipLaunchKernelGGL((simple_layer_wide_kernel<true>), dim3(static_cast<unsigned>(P)), dim3(64 * kWideWaves), 0, st, a);
    else hipLaunchKernelGGL((simple_layer_wide_kernel<false>), dim3(static_cast<unsigned>(P)), dim3(64 * kWideWaves), 0, st, a);
    return dif::launch_status("simple_layer_wide_kernel");
}

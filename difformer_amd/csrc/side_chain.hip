// The coefficient chain of the closed-form `simple` layer (Gram record -> coefficients; simple_layer.hip) as BACKGROUND
// kernels: single-wave workgroups, no LDS, <= 128 VGPRs -- the footprint that fits BESIDE a workgroup of the
// feature-sliced product (gcn_sliced.hip: 15 waves x 128 VGPRs and all but 256 bytes of a CU's LDS leave one wave slot
// per CU).  The chain does not depend on the product (both read the layer input), so on a second stream it runs under it
// instead of in front of it (node classification/difformer.py:18-39, :115-118 for query == source, one head, C, D <= 64).
//
// With the augmented matrices  X~ = [X | 1],  W~ = [W | b]  (q = X~ W~q^T ...) and  G~ = X~^T X~ = [[G, sx], [sx^T, N]]:
//     |Q|^2 = <W~q^T W~q, G~>,  |K|^2 = <W~k^T W~k, G~>                     (:20-21)
//     T = G~ V~,   V~ = [W~v^T | e]        -> T[last] = [sum v | N]
//     R = P~ T,    P~ = W~q^T W~k          -> R[c] = [Mn[c] | u[c]] / s,  R[last] = [bq KtV | bq . sum k] / s      (:25-38)
// so the layer needs two small GEMMs and two dot products; the weight-only factors P~, V~^T, S~ are cached by the host.
// All matrices are zero-padded to 80 x 80 floats (kB), the augmented index is 64, every operand is k-contiguous so a
// lane's MFMA inputs are 16-byte loads:  D = M1 M2  with  A = M1[i][k],  B = M2^T[j][k].
//   gram_bg_kernel      per wave: rows -> partial [G | sx] record (as gram_kernel, no fold across waves)
//   finalize_bg_kernel  sums the partials into the padded G~ (fixed order: deterministic)
//   coeffs_bg_kernel<0> 25 tile workgroups: T^T = (G~ V~)^T;  one more: |Q|^2, |K|^2, s
//   coeffs_bg_kernel<1> 25 tile workgroups: R = P~ T -> coef = [MnT | cn | u | cd | s | |Q|^2 | |K|^2] (dif_simple_coeffs_f32's layout)
#include "dif_common.h"

namespace {

using dif::f32x4;

constexpr int kB = 80;                 // padded matrix extent: 64 + the augmented index + padding to 5 MFMA tiles
constexpr int kAug = 64;               // index of the augmented row / column
constexpr int kBgChunksMax = 512;

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// ---- Gram partials: one wave = one workgroup = one partial record [G: C x C][sx: C] ------------------------
__global__ __launch_bounds__(64) void gram_bg_kernel(const float* __restrict__ x, int64_t ldx, int64_t n_rows, int C,
                                                     float* __restrict__ ws, int64_t ws_stride) {
    const int lane = threadIdx.x;
    const int l15 = lane & 15, lg = lane >> 4;
    const bool col_ok = 4 * l15 < C;
    f32x4 acc[10];
#pragma unroll
    for (int a = 0; a < 10; ++a) acc[a] = zero4();
    f32x4 sx = zero4();
    const int64_t n16 = (n_rows + 15) / 16;
    auto load16 = [&](f32x4 (&xv)[4], int64_t tile) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t row = tile * 16 + 4 * u + lg;
            xv[u] = (row < n_rows && col_ok) ? *reinterpret_cast<const f32x4*>(x + row * ldx + 4 * l15) : zero4();
        }
    };
    f32x4 nxt[4];
    const int64_t first = blockIdx.x, stride = gridDim.x;
    if (first < n16) load16(nxt, first);
    for (int64_t tile = first; tile < n16; tile += stride) {
        f32x4 xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = nxt[u];
        if (tile + stride < n16) load16(nxt, tile + stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sx += xv[u];
            int a = 0;
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = ta; tb < 4; ++tb, ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][ta], xv[u][tb], acc[a], 0, 0, 0);
        }
    }
    float* rec = ws + static_cast<int64_t>(blockIdx.x) * ws_stride;
    int i = 0;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = ta; tb < 4; ++tb, ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int gi = 4 * (4 * lg + reg) + ta, gj = 4 * l15 + tb;
                if (gi < C && gj < C) {
                    rec[gi * C + gj] = acc[i][reg];
                    if (ta != tb) rec[gj * C + gi] = acc[i][reg];
                }
            }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float a = sx[t];
        a = dif::rows4_sum(a);
        if (lg == 0 && 4 * l15 + t < C) rec[C * C + 4 * l15 + t] = a;
    }
}

// ---- G~ (kB x kB, zero padded) from P partial records, one entry per lane ------------------------------------
// Each workgroup also leaves its share of the two norm products <S~q, G~>, <S~k, G~> (float64) in npart[2 * block].
__global__ __launch_bounds__(64) void finalize_bg_kernel(const float* __restrict__ ws, int P, int64_t ws_stride, int C,
                                                         float n_global, const float* __restrict__ st,
                                                         float* __restrict__ gt, double* __restrict__ npart) {
    const int e = blockIdx.x * 64 + threadIdx.x;          // kB * kB is a multiple of 64: every lane owns an entry
    const int i = e / kB, k = e % kB;
    int col = -1;                                         // entry of the partial records this element sums
    if (i < C && k < C) col = i * C + k;
    else if (i == kAug && k < C) col = C * C + k;
    else if (k == kAug && i < C) col = C * C + i;
    float a = 0.f;
    if (col >= 0) {
        int p = 0;
        for (; p + 15 < P; p += 16) {                     // 16 loads in flight, summed in the order of a plain loop
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = ws[static_cast<int64_t>(p + u) * ws_stride + col];
#pragma unroll
            for (int u = 0; u < 16; ++u) a += v[u];
        }
        for (; p < P; ++p) a += ws[static_cast<int64_t>(p) * ws_stride + col];
    } else if (i == kAug && k == kAug) {
        a = n_global;
    }
    gt[e] = a;
    double q2 = static_cast<double>(st[e]) * a, k2 = static_cast<double>(st[kB * kB + e]) * a;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        q2 += __shfl_xor(q2, off, 64);
        k2 += __shfl_xor(k2, off, 64);
    }
    if (threadIdx.x == 0) { npart[2 * blockIdx.x] = q2; npart[2 * blockIdx.x + 1] = k2; }
}

// one 16 x 16 tile of M1 M2 (both kB x kB): A = M1[i][k], B = M2^T[j][k]; lane holds D[16ti + 4lg + reg][16tj + l15]
__device__ __forceinline__ f32x4 tile_kk(const float* __restrict__ m1, const float* __restrict__ m2t, int ti, int tj, int l15,
                                         int lg) {
    f32x4 a4[kB / 16], b4[kB / 16];
#pragma unroll
    for (int q = 0; q < kB / 16; ++q) {
        a4[q] = *reinterpret_cast<const f32x4*>(m1 + (16 * ti + l15) * kB + 16 * q + 4 * lg);
        b4[q] = *reinterpret_cast<const f32x4*>(m2t + (16 * tj + l15) * kB + 16 * q + 4 * lg);
    }
    f32x4 d = zero4();
#pragma unroll
    for (int q = 0; q < kB / 16; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q][t], b4[q][t], d, 0, 0, 0);
    return d;
}

// STAGE 0: tt = (G~ V~)^T, scal = {s, |Q|^2, |K|^2}.   STAGE 1: coef from R = P~ T.
template <int STAGE>
__global__ __launch_bounds__(64) void coeffs_bg_kernel(const float* __restrict__ gt, const float* __restrict__ pt,
                                                       const float* __restrict__ vtt, const float* __restrict__ st,
                                                       int C, int D, float attn_scale, float* __restrict__ tt,
                                                       float* __restrict__ scal, const double* __restrict__ npart,
                                                       float* __restrict__ coef) {
    const int lane = threadIdx.x, l15 = lane & 15, lg = lane >> 4;
    constexpr int kT = kB / 16;
    const int b = blockIdx.x;
    if (STAGE == 0) {
        if (b < kT * kT) {
            const int ti = b / kT, tj = b % kT;
            const f32x4 d = tile_kk(gt, vtt, ti, tj, l15, lg);                   // T[i][j]
            *reinterpret_cast<f32x4*>(tt + (16 * tj + l15) * kB + 16 * ti + 4 * lg) = d;      // T^T[j][i .. i + 3]
        } else {
            double q2 = 0.0, k2 = 0.0;                                            // <S~q, G~>, <S~k, G~>: the 100 partials of finalize_bg
            for (int e = lane; e < kB * kB / 64; e += 64) {
                q2 += npart[2 * e];
                k2 += npart[2 * e + 1];
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                q2 += __shfl_xor(q2, off, 64);
                k2 += __shfl_xor(k2, off, 64);
            }
            if (lane == 0) {
                const float fq = static_cast<float>(q2), fk = static_cast<float>(k2);
                scal[0] = 1.0f / (sqrtf(fq) * sqrtf(fk));          // difformer.py:20-21: zero norms give inf / nan as there
                scal[1] = fq;
                scal[2] = fk;
            }
        }
    } else {
        const int ti = b / kT, tj = b % kT;
        const f32x4 r = tile_kk(pt, tt, ti, tj, l15, lg);                        // R[c = 16ti + 4lg + reg][d = 16tj + l15]
        const float s = scal[0];
        float* MnT = coef;
        float* cn = coef + D * C;
        float* u = cn + D;
        const int c0 = 16 * ti + 4 * lg, d = 16 * tj + l15;
        if (c0 < C && d < D) *reinterpret_cast<f32x4*>(MnT + d * C + c0) = (attn_scale * s) * r;       // C % 4 == 0
        if (d == kAug && c0 < C) *reinterpret_cast<f32x4*>(u + c0) = s * r;                             // u[c] = s R[c][aug]
        if (c0 == kAug) {                                                         // the augmented row: reg 0
            if (d < D) cn[d] = attn_scale * (s * r[0] + tt[d * kB + kAug]);       // + sum v = T[aug][d]
            if (d == kAug) {
                u[C] = s * r[0] + tt[kAug * kB + kAug];                           // cd: + N = T[aug][aug]
                u[C + 1] = s;
                u[C + 2] = scal[1];
                u[C + 3] = scal[2];
            }
        }
    }
}

int bg_chunks(int64_t n_rows) {
    const int64_t tiles = (n_rows + 15) / 16;
    int64_t p = (tiles + 3) / 4;                          // at least four 16-row tiles per wave
    if (p > kBgChunksMax) p = kBgChunksMax;
    if (p < 1) p = 1;
    return static_cast<int>(p);
}

int64_t rec_stride(int C) { return (static_cast<int64_t>(C) * C + C + 3) & ~int64_t(3); }

}  // namespace

extern "C" size_t dif_gram_bg_workspace_bytes(int64_t n_rows, int C) {
    if (n_rows <= 0 || C <= 0 || C > 64) return 0;
    return static_cast<size_t>(rec_stride(C)) * sizeof(float) * static_cast<size_t>(bg_chunks(n_rows));
}

// gt float[80 * 80 + 400]: the zero-padded G~ = [[X^T X, sum x], [sum x^T, n_global]] (augmented index 64), followed by the 100
// float64 pairs of partial norm products <S~q, G~>, <S~k, G~> (sfac = st of dif_simple_coeffs_bg_f32).  x == NULL: `workspace`
// already holds ONE record [G | sx] (e.g. dif_gram_f32's, whose pass also wrote the slice-major copy) and is only re-laid.
extern "C" int dif_gram_bg_f32(const float* x, int64_t ldx, int64_t n_rows, int C, int64_t n_global, const float* sfac, float* gt,
                               void* workspace, size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(gt && sfac && workspace && n_rows > 0 && n_global > 0, DIF_E_BADARG, "dif_gram_bg: null pointer or no rows");
    DIF_REQUIRE(dif::aligned16(gt), DIF_E_BADARG, "dif_gram_bg: gt must be 16-byte aligned");
    DIF_REQUIRE(C > 0 && C <= 64 && C % 4 == 0, DIF_E_SHAPE, "dif_gram_bg: covers C <= 64, C %% 4 == 0 (got %d)", C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* ws = static_cast<float*>(workspace);
    int P = 1;
    if (x) {
        DIF_REQUIRE(ldx >= C && ldx % 4 == 0 && dif::aligned16(x), DIF_E_BADARG, "dif_gram_bg: rows of x must be 16-byte aligned");
        DIF_REQUIRE(workspace_bytes >= dif_gram_bg_workspace_bytes(n_rows, C), DIF_E_WORKSPACE, "dif_gram_bg: workspace too small");
        P = bg_chunks(n_rows);
        hipLaunchKernelGGL(gram_bg_kernel, dim3(P), dim3(64), 0, st, x, ldx, n_rows, C, ws, rec_stride(C));
        if (int rc = dif::launch_status("gram_bg_kernel")) return rc;
    } else {
        DIF_REQUIRE(workspace_bytes >= static_cast<size_t>(C * C + C) * sizeof(float), DIF_E_WORKSPACE, "dif_gram_bg: record too small");
    }
    hipLaunchKernelGGL(finalize_bg_kernel, dim3(kB * kB / 64), dim3(64), 0, st, ws, P, rec_stride(C), C,
                       static_cast<float>(n_global), sfac, gt, reinterpret_cast<double*>(gt + kB * kB));
    return dif::launch_status("finalize_bg_kernel");
}

// pt = P~ = W~q^T W~k, vtt = V~^T, st = [W~q^T W~q ; W~k^T W~k]: float[80 * 80] (x 2 for st), zero padded, augmented index
// 64 (weight-only: the host caches them).  scratch: float[80 * 80 + 4].  coef: dif_simple_coeffs_len(C, D) floats, same
// layout and meaning as dif_simple_coeffs_f32's.
extern "C" int dif_simple_coeffs_bg_f32(const float* gt, const float* pt, const float* vtt, const float* st, int C, int D,
                                        float attn_scale, float* scratch, float* coef, dif_stream_t stream) {
    DIF_REQUIRE(gt && pt && vtt && st && scratch && coef, DIF_E_BADARG, "dif_simple_coeffs_bg: null pointer");
    DIF_REQUIRE(C > 0 && C <= 64 && C % 4 == 0 && D > 0 && D <= 64, DIF_E_SHAPE,
                "dif_simple_coeffs_bg: covers C <= 64 (C %% 4 == 0), D <= 64 (got %d, %d)", C, D);
    DIF_REQUIRE(dif::aligned16(gt) && dif::aligned16(pt) && dif::aligned16(vtt) && dif::aligned16(scratch) && dif::aligned16(coef) &&
                    (D * C + D) % 4 == 0, DIF_E_BADARG, "dif_simple_coeffs_bg: buffers must be 16-byte aligned (and D %% 4 == 0)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* tt = scratch;
    float* scal = scratch + kB * kB;
    const double* npart = reinterpret_cast<const double*>(gt + kB * kB);
    constexpr int kT = kB / 16;
    hipLaunchKernelGGL((coeffs_bg_kernel<0>), dim3(kT * kT + 1), dim3(64), 0, s, gt, pt, vtt, st, C, D, attn_scale, tt, scal, npart, coef);
    if (int rc = dif::launch_status("coeffs_bg_kernel<0>")) return rc;
    hipLaunchKernelGGL((coeffs_bg_kernel<1>), dim3(kT * kT), dim3(64), 0, s, gt, pt, vtt, st, C, D, attn_scale, tt, scal, npart, coef);
    return dif::launch_status("coeffs_bg_kernel<1>");
}

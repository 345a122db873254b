// Small float64 steps of the closed form at the scripts' widths (hidden 128 / 300 / 400; ops.simple_layer_closed_form_wide):
// between the Gram pass and the row GEMM sit two float64 library GEMMs on (C + 1)-square matrices -- and, as tensor ops,
// a dozen tiny launches around them (mirror the Gram blocks, append sum x and N, two norm products, rsqrt, scale, convert).
// Two kernels do that bookkeeping:
//   wide_gram_kernel   G~ = [[X^T X, sx], [sx^T, N]] in float64 from the Gram record (of X^T X only the 64 x 64 blocks on and
//                      above the diagonal are valid: the rest is mirrored), plus per-workgroup partial sums of the two norm
//                      products <W~q^T W~q, G~> = |Q|^2 and <W~k^T W~k, G~> = |K|^2 (difformer.py:20-21)
//   wide_scale_kernel  s = 1 / (|Q| |K|) from the partial sums (added in index order: deterministic), then the operands of the
//                      row GEMM in float32: B = s R[0..C), bias = s R[C] + T[C]      ([Mn | u] and [cn | cd], :25-38)
#include "dif_common.h"

namespace {

constexpr int kWideThreads = 256;

__device__ __forceinline__ double block_sum(double v, double* sm) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[wave] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ __launch_bounds__(kWideThreads) void wide_gram_kernel(const float* __restrict__ rec, int C, double n_global,
                                                                 const double* __restrict__ S, double* __restrict__ Gt,
                                                                 double* __restrict__ partial) {
    __shared__ double sm[4];
    const int C1 = C + 1;
    const int64_t total = static_cast<int64_t>(C1) * C1;
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * kWideThreads + threadIdx.x;
    double g = 0.0, a = 0.0, b = 0.0;
    if (idx < total) {
        const int i = static_cast<int>(idx / C1), j = static_cast<int>(idx % C1);
        if (i < C && j < C) g = ((i >> 6) <= (j >> 6)) ? rec[static_cast<int64_t>(i) * C + j] : rec[static_cast<int64_t>(j) * C + i];
        else if (i < C) g = rec[static_cast<int64_t>(C) * C + i];
        else if (j < C) g = rec[static_cast<int64_t>(C) * C + j];
        else g = n_global;
        Gt[idx] = g;
        a = S[idx] * g;
        b = S[total + idx] * g;
    }
    const double sa = block_sum(a, sm);
    const double sb = block_sum(b, sm);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = sa; partial[2 * blockIdx.x + 1] = sb; }
}

__global__ __launch_bounds__(kWideThreads) void wide_scale_kernel(const double* __restrict__ R, const double* __restrict__ T,
                                                                  const double* __restrict__ partial, int n_partial, int C,
                                                                  int DV, float* __restrict__ B, float* __restrict__ bias) {
    __shared__ double sm[4];
    // every workgroup adds the partial sums itself, in the same order: a few hundred values
    double a = 0.0, b = 0.0;
    for (int p = threadIdx.x; p < n_partial; p += kWideThreads) { a += partial[2 * p]; b += partial[2 * p + 1]; }
    const double q2 = block_sum(a, sm);
    const double k2 = block_sum(b, sm);
    const double s = 1.0 / (sqrt(q2) * sqrt(k2));
    const int64_t total = static_cast<int64_t>(C + 1) * DV;
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * kWideThreads + threadIdx.x;
    if (idx >= total) return;
    const int r = static_cast<int>(idx / DV), c = static_cast<int>(idx % DV);
    if (r < C) B[idx] = static_cast<float>(s * R[idx]);
    else bias[c] = static_cast<float>(s * R[idx] + T[idx]);
}

// ---- round 5: the two float64 products themselves, no library GEMM left in the forward ---------------------------------
// T = G~ V~ and R = P~ T are (C + 1)-square times (C + 1) x (D + 4): 55 MFLOP at hidden 300 -- nothing for the chip, but as
// two library GEMMs between two bookkeeping kernels they were four dependent launches of 5-10 us each (33 us per layer, 17 %
// of the cifar50k-h300 forward).  Two launches do all of it:
//   wide_gemm_kernel<0>   T = G~ V~ with G~ read straight from the Gram record (float32 -> float64, lower blocks mirrored,
//                         sum x and N appended) and, in the workgroups of the first column of tiles, the partial sums of
//                         <W~q^T W~q, G~> and <W~k^T W~k, G~> for their 32 rows
//   wide_gemm_kernel<1>   R = P~ T, then the row GEMM's operands in float32: B = s R[0..C), bias = s R[C] + T[C], with
//                         s = 1 / (|Q| |K|) from the partial sums (added in index order by every workgroup: deterministic)
// The work is tiny and the chain is dependent, so the kernels are built for LATENCY: one 1,024-thread workgroup per 16 x 16
// output tile, its 16 waves split K (a wave's <= 5 steps of v_mfma_f64_16x16x4_f64 need <= 10 loads, all issued before the
// first product -- one memory round trip per kernel), the 16 partial tiles meet in LDS and are added in wave order.
// (A first version that walked K in 32-wide LDS tiles paid one round trip per tile: 40 us per product.)
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int kWT = 16;
constexpr int kWWaves = 16;

__device__ __forceinline__ double gt_entry(const float* __restrict__ rec, int C, double n_global, int i, int j) {
    if (i < C && j < C) return ((i >> 6) <= (j >> 6)) ? rec[static_cast<int64_t>(i) * C + j] : rec[static_cast<int64_t>(j) * C + i];
    if (i < C) return rec[static_cast<int64_t>(C) * C + i];
    if (j < C) return rec[static_cast<int64_t>(C) * C + j];
    return n_global;
}

template <int MODE>
__global__ __launch_bounds__(64 * kWWaves) void wide_gemm_kernel(const float* __restrict__ rec, int C, double n_global,
                                                                 const double* __restrict__ S, const double* __restrict__ A,
                                                                 const double* __restrict__ Bm, int DV, double* __restrict__ T,
                                                                 double* __restrict__ partial, int n_partial,
                                                                 float* __restrict__ Bout, float* __restrict__ bias) {
    __shared__ double sRed[kWWaves][256];
    __shared__ double smw[kWWaves], smb[kWWaves];
    const int C1 = C + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    const int i0 = blockIdx.y * kWT, j0 = blockIdx.x * kWT;
    const int steps = (C1 + 3) / 4;
    constexpr int kMaxSteps = 9;                       // per wave: 16 waves x 9 steps x 4 columns >= C + 1 = 513 (hidden 512, ops.CLOSED_FORM_WIDE_MAX)
    const int64_t tot = static_cast<int64_t>(C1) * C1;
    double av[kMaxSteps], bv[kMaxSteps];
    double na = 0.0, nb = 0.0;
#pragma unroll
    for (int q = 0; q < kMaxSteps; ++q) {
        const int st = wave + kWWaves * q;
        const int k = 4 * st + kg, i = i0 + l15, j = j0 + l15;
        av[q] = 0.0;
        bv[q] = 0.0;
        if (st < steps && k < C1) {
            if (i < C1) {
                if (MODE == 0) {
                    av[q] = gt_entry(rec, C, n_global, i, k);
                    if (blockIdx.x == 0) {
                        na += S[static_cast<int64_t>(i) * C1 + k] * av[q];
                        nb += S[tot + static_cast<int64_t>(i) * C1 + k] * av[q];
                    }
                } else {
                    av[q] = A[static_cast<int64_t>(i) * C1 + k];
                }
            }
            if (j < DV) bv[q] = Bm[static_cast<int64_t>(k) * DV + j];
        }
    }
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < kMaxSteps; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) sRed[wave][64 * r + lane] = acc[r];
    double q2 = 0.0, k2 = 0.0;
    if (MODE == 0 && blockIdx.x == 0) {                // norm partials of this row block: wave sums, then the waves in order
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { na += __shfl_xor(na, m, 64); nb += __shfl_xor(nb, m, 64); }
        if (lane == 0) { smw[wave] = na; smb[wave] = nb; }
    }
    __syncthreads();
    if (MODE == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        double sa = 0.0, sb = 0.0;
        for (int w = 0; w < kWWaves; ++w) { sa += smw[w]; sb += smb[w]; }
        partial[2 * blockIdx.y] = sa;
        partial[2 * blockIdx.y + 1] = sb;
    }
    if (MODE == 1) {
        for (int p = 0; p < n_partial; ++p) { q2 += partial[2 * p]; k2 += partial[2 * p + 1]; }      // same order in every thread
    }
    if (threadIdx.x < 256) {
        const int e = threadIdx.x;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < kWWaves; ++w) v += sRed[w][e];
        const int r = e >> 6, ln = e & 63;
        const int i = i0 + 4 * r + (ln >> 4), j = j0 + (ln & 15);          // v_mfma_f64_16x16x4_f64: register r of lane l holds
                                                                           // D[4 r + l / 16][l % 16] (NOT the float32 layout)
        if (MODE == 0) {
            if (i < C1 && j < DV) T[static_cast<int64_t>(i) * DV + j] = v;
        } else {
            const double s = 1.0 / (sqrt(q2) * sqrt(k2));
            if (i < C && j < DV) Bout[static_cast<int64_t>(i) * DV + j] = static_cast<float>(s * v);
            else if (i == C && j < DV) bias[j] = static_cast<float>(s * v + T[static_cast<int64_t>(C) * DV + j]);
        }
    }
}

}  // namespace

// Gram record -> the row GEMM's operands in two launches (see above).  record: dif_gram_sym_f32's; S double [2][(C+1)^2],
// V double [(C+1)][DV], P double [(C+1)][(C+1)]: the weight-only factors (ops.WideCoefficients); T double [(C+1)][DV] and
// partial double [2 * ceil((C+1) / 16)]: scratch; B float [C][DV], bias float [DV]: out.
extern "C" int dif_wide_coeffs_f64(const float* record, int C, int64_t n_global, const double* S, const double* V, const double* P,
                                   int DV, double* T, double* partial, float* B, float* bias, dif_stream_t stream) {
    DIF_REQUIRE(record && S && V && P && T && partial && B && bias && C > 0 && C <= 8192 && DV > 0 && n_global > 0, DIF_E_BADARG,
                "dif_wide_coeffs: bad argument");
    const int C1 = C + 1;
    DIF_REQUIRE(C1 <= 513, DIF_E_SHAPE, "dif_wide_coeffs: C <= 512 (got %d)", C);
    const int gy = (C1 + kWT - 1) / kWT, gx = (DV + kWT - 1) / kWT;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(wide_gemm_kernel<0>, dim3(gx, gy), dim3(64 * kWWaves), 0, st, record, C, static_cast<double>(n_global), S, nullptr, V, DV, T,
                       partial, gy, nullptr, nullptr);
    if (int rc = dif::launch_status("wide_gemm_kernel<0>")) return rc;
    hipLaunchKernelGGL(wide_gemm_kernel<1>, dim3(gx, gy), dim3(64 * kWWaves), 0, st, nullptr, C, 0.0, nullptr, P, T, DV, T, partial, gy, B, bias);
    return dif::launch_status("wide_gemm_kernel<1>");
}

extern "C" int64_t dif_wide_partials(int C) {
    const int64_t total = static_cast<int64_t>(C + 1) * (C + 1);
    return (total + kWideThreads - 1) / kWideThreads;
}

extern "C" int dif_wide_gram_f64(const float* record, int C, int64_t n_global, const double* S, double* Gt, double* partial,
                                 dif_stream_t stream) {
    DIF_REQUIRE(record && S && Gt && partial && C > 0 && C <= 8192 && n_global > 0, DIF_E_BADARG, "dif_wide_gram: bad argument");
    hipLaunchKernelGGL(wide_gram_kernel, dim3(static_cast<unsigned>(dif_wide_partials(C))), dim3(kWideThreads), 0,
                       static_cast<hipStream_t>(stream), record, C, static_cast<double>(n_global), S, Gt, partial);
    return dif::launch_status("wide_gram_kernel");
}

extern "C" int dif_wide_scale_f64(const double* R, const double* T, const double* partial, int C, int DV, float* B,
                                  float* bias, dif_stream_t stream) {
    DIF_REQUIRE(R && T && partial && B && bias && C > 0 && C <= 8192 && DV > 0, DIF_E_BADARG, "dif_wide_scale: bad argument");
    const int64_t total = static_cast<int64_t>(C + 1) * DV;
    hipLaunchKernelGGL(wide_scale_kernel, dim3(static_cast<unsigned>((total + kWideThreads - 1) / kWideThreads)),
                       dim3(kWideThreads), 0, static_cast<hipStream_t>(stream), R, T, partial,
                       static_cast<int>(dif_wide_partials(C)), C, DV, B, bias);
    return dif::launch_status("wide_scale_kernel");
}

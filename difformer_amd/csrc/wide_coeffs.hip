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

}  // namespace

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

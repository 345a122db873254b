// Column-sum of the P partial records of the simple kernel's stage 1 -> the `reduced` record (shared by
// simple_attn.hip and project_reduce.hip).  Deterministic: fixed summation order, no atomics.
// Record layout: [t_main floats: KtV | ksum | vsum][2 floats per tile: sum q*q, sum k*k]; output: t_main + 2.
#include "dif_common.h"

namespace {

constexpr int kFinSlices = 16;

// Blocks 0..nb-2: 64 columns x 16 record slices each; last block: the two Frobenius scalars (P x tiles entries each).
__global__ __launch_bounds__(1024) void record_finalize_kernel(const float* __restrict__ ws, int P, int64_t ws_stride,
                                                               int t_main, int tiles, float* __restrict__ reduced) {
    __shared__ float sm[kFinSlices][64];
    const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
    if (static_cast<int>(blockIdx.x) < (t_main + 63) / 64) {
        const int col = blockIdx.x * 64 + c;
        float a = 0.f;
        if (col < t_main) {
            // the partials come from other XCDs' workgroups (MALL / HBM latency): 8 loads in flight, summed in the
            // same fixed order as a plain loop would
            int p = sl;
            for (; p + 7 * kFinSlices < P; p += 8 * kFinSlices) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = ws[static_cast<int64_t>(p + u * kFinSlices) * ws_stride + col];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += v[u];
            }
            for (; p < P; p += kFinSlices) a += ws[static_cast<int64_t>(p) * ws_stride + col];
        }
        sm[sl][c] = a;
        __syncthreads();
        if (sl == 0 && col < t_main) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < kFinSlices; ++i) t += sm[i][c];
            reduced[col] = t;
        }
    } else {
        float a0 = 0.f, a1 = 0.f;
        const int total = P * tiles;
        for (int i = threadIdx.x; i < total; i += 1024) {
            const int p = i / tiles, yy = i % tiles;
            a0 += ws[p * ws_stride + t_main + 2 * yy];
            a1 += ws[p * ws_stride + t_main + 2 * yy + 1];
        }
        a0 = dif::wave_sum(a0);
        a1 = dif::wave_sum(a1);
        if (c == 0) { sm[sl][0] = a0; sm[sl][1] = a1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float t0 = 0.f, t1 = 0.f;
#pragma unroll
            for (int i = 0; i < kFinSlices; ++i) { t0 += sm[i][0]; t1 += sm[i][1]; }
            reduced[t_main] = t0;
            reduced[t_main + 1] = t1;
        }
    }
}

}  // namespace

namespace dif {

int launch_record_finalize(const float* ws, int P, int64_t ws_stride, int t_main, int tiles, float* reduced,
                           hipStream_t st) {
    const int nb = (t_main + 63) / 64 + (tiles >= 0 ? 1 : 0);       // tiles < 0: column sums only, no scalar pair
    hipLaunchKernelGGL(record_finalize_kernel, dim3(nb), dim3(1024), 0, st, ws, P, ws_stride, t_main, tiles, reduced);
    return launch_status("record_finalize_kernel");
}

}  // namespace dif

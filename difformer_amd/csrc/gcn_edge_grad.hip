// Gradient of gcn_conv with respect to edge_weight (node classification/difformer.py:63-79 under loss.backward(),
// main.py:130): the reference builds value = edge_weight * d_in * d_out (:73), nan_to_num (:74), and torch_sparse.matmul
// is differentiable in the values, so
//     d loss / d w_e = [value_e finite] * <g[col_e, :], x[row_e, :]> * d_out[row_e] * d_in[col_e]
// in the ORIGINAL edge order.  An edge that leaves a node without incoming entries has d_out = inf: the reference's
// gradient there is 0 * inf = NaN, and it is NaN here (the arithmetic below is the reference's, term by term).
//
// HBM/L2-bound gather: two F-float rows per edge.  16 lanes own an edge (a 256-byte row at F = 64 is one 16-byte load per
// lane), four edges per wave instruction, partial dot products folded with four xor-shuffles inside the lane group.
#include "dif_common.h"

namespace {

template <int VEC>
__global__ __launch_bounds__(256) void edge_weight_grad_kernel(const int64_t* __restrict__ edge_index, int64_t E,
                                                               const float* __restrict__ w,
                                                               const int32_t* __restrict__ rowptr,
                                                               const float* __restrict__ g, int64_t ldg,
                                                               const float* __restrict__ x, int64_t ldx, int F, float scale,
                                                               float* __restrict__ dw) {
    const int lane16 = threadIdx.x & 15;
    const int64_t group = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 4;
    const int64_t n_groups = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 4;
    for (int64_t e = group; e < E; e += n_groups) {
        const int64_t r = edge_index[e];          // source   (row of :64)
        const int64_t c = edge_index[E + e];      // destination (col)
        const float* gr = g + c * ldg;
        const float* xr = x + r * ldx;
        float acc = 0.f;
        if (VEC == 4) {
            for (int j = 4 * lane16; j < F; j += 64) {
                const dif::f32x4 a = *reinterpret_cast<const dif::f32x4*>(gr + j);
                const dif::f32x4 b = *reinterpret_cast<const dif::f32x4*>(xr + j);
                acc += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
            }
        } else {
            for (int j = lane16; j < F; j += 16) acc += gr[j] * xr[j];
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane16 == 0) {
            const float deg_c = static_cast<float>(rowptr[c + 1] - rowptr[c]);     // :66, in-degree as float32
            const float deg_r = static_cast<float>(rowptr[r + 1] - rowptr[r]);
            const float d_in = sqrtf(1.0f / deg_c);                                // :67
            const float d_out = sqrtf(1.0f / deg_r);                               // :68 (inf for deg 0)
            const float value = w[e] * d_in * d_out;                               // :73
            const float gv = isfinite(value) ? scale * acc : 0.f;                  // backward of nan_to_num (:74)
            dw[e] = (gv * d_out) * d_in;                                           // backward of (w * d_in) * d_out
        }
    }
}

}  // namespace

extern "C" int dif_gcn_edge_weight_grad_f32(const int64_t* edge_index, int64_t E, int64_t N, const float* edge_weight,
                                            const int32_t* rowptr, const float* g, int64_t ldg, const float* x,
                                            int64_t ldx, int F, float scale, float* dw, dif_stream_t stream) {
    DIF_REQUIRE(E >= 0 && N >= 0 && F > 0, DIF_E_SHAPE, "dif_gcn_edge_weight_grad: bad extents E=%lld N=%lld F=%d",
                static_cast<long long>(E), static_cast<long long>(N), F);
    if (E == 0) return 0;
    DIF_REQUIRE(edge_index && edge_weight && rowptr && g && x && dw, DIF_E_BADARG, "dif_gcn_edge_weight_grad: null pointer");
    DIF_REQUIRE(ldg >= F && ldx >= F, DIF_E_SHAPE, "dif_gcn_edge_weight_grad: leading dimension below F");
    const bool vec = F % 4 == 0 && ldg % 4 == 0 && ldx % 4 == 0 && dif::aligned16(g) && dif::aligned16(x);
    const int64_t want = (E + 15) / 16;                           // 16 edges per 256-thread workgroup and sweep
    const int grid = static_cast<int>(want < 8LL * dif::kCUs ? want : 8LL * dif::kCUs);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (vec)
        hipLaunchKernelGGL(edge_weight_grad_kernel<4>, dim3(grid), dim3(256), 0, st, edge_index, E, edge_weight, rowptr, g, ldg,
                           x, ldx, F, scale, dw);
    else
        hipLaunchKernelGGL(edge_weight_grad_kernel<1>, dim3(grid), dim3(256), 0, st, edge_index, E, edge_weight, rowptr, g, ldg,
                           x, ldx, F, scale, dw);
    return dif::launch_status("dif_gcn_edge_weight_grad_f32");
}

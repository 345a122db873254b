// f3: backward of the layer tail (dif_layer_tail_f32) -- what loss.backward() (node classification/main.py:130) asks
// of   node classification/difformer.py:137 (head mean), :139-140 (+= x_0), :200-201 (alpha residual), :202-203
// (LayerNorm) and of the input layer :188-191 (Linear -> LayerNorm -> ReLU, relu = 1).  The reference leaves these to
// autograd (one pass per line, LayerNorm backward as three kernels); here the row is re-derived from the saved inputs
// and every gradient leaves in ONE pass:
//   z  = mean_h conv (+ x0);  z2 = prev ? alpha z + (1 - alpha) prev : z;  xh = (z2 - mu) rstd;  y = xh w + b;  out = relu?
//   gy = g (* [y > 0]);  d_w += gy xh;  d_b += gy;  gw = gy w;
//   dz2 = rstd (gw - mean(gw) - xh mean(gw xh));   d_prev = (1 - alpha) dz2;  dz = alpha dz2 (or dz2);  d_x0 = dz;
//   d_conv[h] = dz / H
// d_w / d_b: per-thread partial sums over its rows, folded per workgroup in LDS (fixed order), one partial record per
// workgroup, column sums by record_finalize (no atomics: bitwise reproducible).
#include "dif_common.h"

namespace {

using dif::f32x4;

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// G lanes x V vectors of 4 elements hold one row (D <= 4 G V, D % 4 == 0; V = 2 with G = 64 for 257 .. 512 columns: hidden 300 / 400,
// image and text/run.sh); 256 / G rows per workgroup pass.
template <int G, int V = 1>
__global__ __launch_bounds__(256) void layer_tail_bwd_kernel(const float* __restrict__ conv, int64_t ldc, int64_t n_rows, int H,
                                                             int D, const float* __restrict__ x0, int64_t ldx0,
                                                             const float* __restrict__ prev, int64_t ldp, float alpha,
                                                             const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                             float eps, int relu, const float* __restrict__ g, int64_t ldg,
                                                             float* __restrict__ dconv, int64_t lddc,
                                                             float* __restrict__ dx0, int64_t lddx0,
                                                             float* __restrict__ dprev, int64_t lddp,
                                                             float* __restrict__ ws, int64_t ws_stride) {
    constexpr int RPB = 256 / G;
    __shared__ f32x4 sm[2][V][256];
    const int li = threadIdx.x % G, rl = threadIdx.x / G;
    int col[V];
    bool active[V];
    const float inv_h = 1.0f / static_cast<float>(H);
    const float inv_d = 1.0f / static_cast<float>(D);
    f32x4 w4[V], b4[V], dw[V], db[V];
#pragma unroll
    for (int u = 0; u < V; ++u) {
        col[u] = 4 * (li + u * G);
        active[u] = col[u] < D;
        w4[u] = f32x4{1.f, 1.f, 1.f, 1.f};
        b4[u] = zero4();
        if (ln_w && active[u]) {
            w4[u] = *reinterpret_cast<const f32x4*>(ln_w + col[u]);
            b4[u] = *reinterpret_cast<const f32x4*>(ln_b + col[u]);
        }
        dw[u] = zero4();
        db[u] = zero4();
    }
    const int64_t nrb = (n_rows + RPB - 1) / RPB;
    for (int64_t rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        const int64_t row = rb * RPB + rl;
        bool ok[V];
        f32x4 z[V], gy[V], dz2[V];
#pragma unroll
        for (int u = 0; u < V; ++u) {
            ok[u] = active[u] && row < n_rows;
            z[u] = zero4();
            gy[u] = zero4();
            if (ok[u]) {
                const float* c = conv + row * ldc + col[u];
                for (int h = 0; h < H; ++h) z[u] += *reinterpret_cast<const f32x4*>(c + static_cast<int64_t>(h) * D);
                if (H > 1) z[u] *= inv_h;
                if (x0) z[u] += *reinterpret_cast<const f32x4*>(x0 + row * ldx0 + col[u]);
                if (prev) z[u] = alpha * z[u] + (1.0f - alpha) * *reinterpret_cast<const f32x4*>(prev + row * ldp + col[u]);
                gy[u] = *reinterpret_cast<const f32x4*>(g + row * ldg + col[u]);
            }
            dz2[u] = gy[u];
        }
        if (ln_w) {
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < V; ++u) s += z[u][0] + z[u][1] + z[u][2] + z[u][3];
#pragma unroll
            for (int m = 1; m < G; m <<= 1) s += __shfl_xor(s, m, 64);
            const float mu = s * inv_d;
            f32x4 dz[V], xh[V], gw[V];
            float v = 0.f;
#pragma unroll
            for (int u = 0; u < V; ++u) {
                dz[u] = ok[u] ? (z[u] - mu) : zero4();
                v += dz[u][0] * dz[u][0] + dz[u][1] * dz[u][1] + dz[u][2] * dz[u][2] + dz[u][3] * dz[u][3];
            }
#pragma unroll
            for (int m = 1; m < G; m <<= 1) v += __shfl_xor(v, m, 64);
            const float rstd = 1.0f / sqrtf(v * inv_d + eps);
            float m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int u = 0; u < V; ++u) {
                xh[u] = dz[u] * rstd;
                if (relu) {
                    const f32x4 y = xh[u] * w4[u] + b4[u];
#pragma unroll
                    for (int i = 0; i < 4; ++i) gy[u][i] = y[i] > 0.f ? gy[u][i] : 0.f;
                }
                dw[u] += gy[u] * xh[u];
                db[u] += gy[u];
                gw[u] = gy[u] * w4[u];
                m1 += gw[u][0] + gw[u][1] + gw[u][2] + gw[u][3];
                m2 += gw[u][0] * xh[u][0] + gw[u][1] * xh[u][1] + gw[u][2] * xh[u][2] + gw[u][3] * xh[u][3];
            }
#pragma unroll
            for (int m = 1; m < G; m <<= 1) {
                m1 += __shfl_xor(m1, m, 64);
                m2 += __shfl_xor(m2, m, 64);
            }
#pragma unroll
            for (int u = 0; u < V; ++u) dz2[u] = rstd * (gw[u] - m1 * inv_d - xh[u] * (m2 * inv_d));
        } else if (relu) {
#pragma unroll
            for (int u = 0; u < V; ++u) {
#pragma unroll
                for (int i = 0; i < 4; ++i) dz2[u][i] = z[u][i] > 0.f ? gy[u][i] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < V; ++u) {
            if (!ok[u]) continue;
            f32x4 dzz = dz2[u];
            if (prev) {
                if (dprev) *reinterpret_cast<f32x4*>(dprev + row * lddp + col[u]) = (1.0f - alpha) * dz2[u];
                dzz = alpha * dz2[u];
            }
            if (dx0) *reinterpret_cast<f32x4*>(dx0 + row * lddx0 + col[u]) = dzz;
            if (dconv) {
                const f32x4 dc = H > 1 ? dzz * inv_h : dzz;
                for (int h = 0; h < H; ++h) *reinterpret_cast<f32x4*>(dconv + row * lddc + static_cast<int64_t>(h) * D + col[u]) = dc;
            }
        }
    }
    if (!ws) return;
    // fold the RPB row slots of the workgroup (same column group li) in a fixed order
#pragma unroll
    for (int u = 0; u < V; ++u) {
        sm[0][u][threadIdx.x] = dw[u];
        sm[1][u][threadIdx.x] = db[u];
    }
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int u = 0; u < V; ++u) {
            if (!active[u]) continue;
            f32x4 a = zero4(), b = zero4();
            for (int r = 0; r < RPB; ++r) {
                a += sm[0][u][r * G + li];
                b += sm[1][u][r * G + li];
            }
            float* rec = ws + static_cast<int64_t>(blockIdx.x) * ws_stride;
            *reinterpret_cast<f32x4*>(rec + col[u]) = a;
            *reinterpret_cast<f32x4*>(rec + D + col[u]) = b;
        }
    }
}

int tail_bwd_blocks(int64_t n_rows, int G) {
    const int64_t rpb = 256 / G;
    int64_t gx = (n_rows + rpb - 1) / rpb;
    const int64_t cap = 4 * dif::kCUs;
    if (gx > cap) gx = cap;
    return static_cast<int>(gx < 1 ? 1 : gx);
}

int tail_group(int D) {
    const int q = D / 4;
    return q <= 1 ? 1 : q <= 2 ? 2 : q <= 4 ? 4 : q <= 8 ? 8 : q <= 16 ? 16 : q <= 32 ? 32 : 64;
}

}  // namespace

extern "C" size_t dif_layer_tail_bwd_workspace_bytes(int64_t n_rows, int D) {
    if (n_rows <= 0 || D <= 0 || D % 4 != 0 || D > 512) return 0;
    return static_cast<size_t>(tail_bwd_blocks(n_rows, tail_group(D))) * 2 * D * sizeof(float);
}

extern "C" int dif_layer_tail_bwd_f32(const float* conv, int64_t ldc, int64_t n_rows, int H, int D, const float* x0,
                                      int64_t ldx0, const float* prev, int64_t ldp, float alpha, const float* ln_weight,
                                      const float* ln_bias, float ln_eps, int relu, const float* grad_out, int64_t ldg,
                                      float* d_conv, int64_t lddc, float* d_x0, int64_t lddx0, float* d_prev, int64_t lddp,
                                      float* d_ln, void* workspace, size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(n_rows > 0 && H > 0 && D > 0, DIF_E_BADARG, "dif_layer_tail_bwd: n_rows, H, D must be positive");
    DIF_REQUIRE(D % 4 == 0 && D <= 512, DIF_E_SHAPE, "dif_layer_tail_bwd: needs D %% 4 == 0 and D <= 512 (got %d)", D);
    DIF_REQUIRE(conv && grad_out, DIF_E_BADARG, "dif_layer_tail_bwd: null pointer");
    DIF_REQUIRE((ln_weight == nullptr) == (ln_bias == nullptr) && (ln_weight == nullptr) == (d_ln == nullptr), DIF_E_BADARG,
                "dif_layer_tail_bwd: ln_weight, ln_bias and d_ln must be given together");
    DIF_REQUIRE(ldc >= static_cast<int64_t>(H) * D && ldg >= D && (!x0 || ldx0 >= D) && (!prev || ldp >= D) &&
                    (!d_conv || lddc >= static_cast<int64_t>(H) * D) && (!d_x0 || lddx0 >= D) && (!d_prev || lddp >= D),
                DIF_E_BADARG, "dif_layer_tail_bwd: leading dimension smaller than a row");
    auto ok4 = [](const void* p, int64_t ld) { return !p || (ld % 4 == 0 && dif::aligned16(p)); };
    DIF_REQUIRE(ok4(conv, ldc) && ok4(grad_out, ldg) && ok4(x0, ldx0) && ok4(prev, ldp) && ok4(d_conv, lddc) &&
                    ok4(d_x0, lddx0) && ok4(d_prev, lddp) && ok4(ln_weight, 4) && ok4(ln_bias, 4),
                DIF_E_BADARG, "dif_layer_tail_bwd: rows must be 16-byte aligned");
    const int G = tail_group(D);
    const int blocks = tail_bwd_blocks(n_rows, G);
    float* ws = nullptr;
    if (ln_weight) {
        DIF_REQUIRE(workspace && workspace_bytes >= dif_layer_tail_bwd_workspace_bytes(n_rows, D) && dif::aligned16(workspace),
                    DIF_E_WORKSPACE, "dif_layer_tail_bwd: workspace too small (see dif_layer_tail_bwd_workspace_bytes)");
        ws = static_cast<float*>(workspace);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
#define DIF_TAILB(...)                                                                                                    \
    hipLaunchKernelGGL((layer_tail_bwd_kernel<__VA_ARGS__>), dim3(blocks), dim3(256), 0, st, conv, ldc, n_rows, H, D, x0, ldx0, prev, \
                       ldp, alpha, ln_weight, ln_bias, ln_eps, relu, grad_out, ldg, d_conv, lddc, d_x0, lddx0, d_prev, lddp, \
                       ws, static_cast<int64_t>(2 * D))
    switch (G) {
        case 1: DIF_TAILB(1); break;
        case 2: DIF_TAILB(2); break;
        case 4: DIF_TAILB(4); break;
        case 8: DIF_TAILB(8); break;
        case 16: DIF_TAILB(16); break;
        case 32: DIF_TAILB(32); break;
        default:
            if (D <= 256) DIF_TAILB(64);
            else DIF_TAILB(64, 2);
            break;
    }
#undef DIF_TAILB
    if (int rc = dif::launch_status("layer_tail_bwd_kernel")) return rc;
    if (!ln_weight) return 0;
    // d_ln float[2 D] = {d ln_weight, d ln_bias}: column sums of the workgroup records, fixed order
    return dif::launch_record_finalize(ws, blocks, 2 * D, 2 * D, -1, d_ln, st);
}

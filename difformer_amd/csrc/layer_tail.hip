// a4/a5 tail: everything between the propagation operators and the next layer, in one pass.
//   node classification/difformer.py:137      final_output.mean(dim=1)         (heads)
//   node classification/difformer.py:139-140  final_output += x_0               (use_source)
//   node classification/difformer.py:200-201  x = alpha*x + (1-alpha)*layer_[i] (use_residual)
//   node classification/difformer.py:202-203  x = LayerNorm(x)                  (use_bn; eps 1e-5, affine)
// (dropout, :204, is the identity in eval mode.)  HBM-bound: reads conv [n,H,D] (+x0, prev),
// writes [n,D]; the reference does each line as its own full pass.
#include "dif_common.h"

namespace {

using dif::f32x4;
using dif::Elem;

// G lanes x V x 4 elements hold one row (D <= 4 G V, D % 4 == 0; V = 2 only with G = 64: rows of 260..512 elements, the
// widths of image and text/run.sh); 256/G rows per block.  T = float | dif::bf16 (storage).
// MIX (closed form at the scripts' widths, H == 1, fp32): the propagation result is assembled here as
//   conv_scale * conv[row] / den[row] + add_scale * (add[row] + rs[row] * bv)        (difformer.py:36-39, :75-78, :130-134)
// from the row GEMM outputs (numerator | denominator, aggregated values) -- no intermediate pass.
struct Mix {
    const float* den;       // per-row divisor, element row * ldden (NULL: none)
    int64_t ldden;
    float conv_scale;
    const float* add;       // [n, D] second operand (NULL: none)
    int64_t lda;
    float add_scale;
    const float* rs;        // [n] with bv [D]: rank-one term rs[row] * bv[col] inside the second operand (NULL: none)
    const float* bv;
};

template <int G, int V, typename T, bool MIX = false>
__global__ __launch_bounds__(256) void layer_tail_vec_kernel(const T* __restrict__ conv, int64_t ldc,
                                                             int64_t n_rows, int H, int D,
                                                             const T* __restrict__ x0, int64_t ldx0,
                                                             const T* __restrict__ prev, int64_t ldp, float alpha,
                                                             const T* __restrict__ ln_w,
                                                             const T* __restrict__ ln_b, float eps, int relu,
                                                             T* __restrict__ out, int64_t ldo, Mix mix = Mix{}) {
    constexpr int RPB = 256 / G;
    const int li = threadIdx.x % G;
    const float inv_h = 1.0f / static_cast<float>(H);
    const float inv_d = 1.0f / static_cast<float>(D);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    int col[V];
    bool active[V];
    f32x4 w4[V], b4[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        col[v] = 4 * (li + v * G);
        active[v] = col[v] < D;
        w4[v] = f32x4{1.f, 1.f, 1.f, 1.f};
        b4[v] = zero;
        if (ln_w && active[v]) { w4[v] = Elem<T>::ld4(ln_w + col[v]); b4[v] = Elem<T>::ld4(ln_b + col[v]); }
    }
    const int64_t nrb = (n_rows + RPB - 1) / RPB;
    for (int64_t rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        const int64_t row = rb * RPB + threadIdx.x / G;
        f32x4 z[V];
        bool ok[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            ok[v] = active[v] && row < n_rows;
            z[v] = zero;
            if (ok[v]) {
                const T* c = conv + row * ldc + col[v];
                for (int h = 0; h < H; ++h) z[v] += Elem<T>::ld4(c + static_cast<int64_t>(h) * D);
                if (H > 1) z[v] *= inv_h;                                                              // :137
                if (MIX) {
                    float sc = mix.conv_scale;
                    if (mix.den) sc /= mix.den[row * mix.ldden];
                    z[v] *= sc;
                    if (mix.add) {
                        f32x4 a = *reinterpret_cast<const f32x4*>(mix.add + row * mix.lda + col[v]);
                        if (mix.rs) a += mix.rs[row] * *reinterpret_cast<const f32x4*>(mix.bv + col[v]);
                        z[v] += mix.add_scale * a;
                    }
                }
                if (x0) z[v] += Elem<T>::ld4(x0 + row * ldx0 + col[v]);                                // :139-140
                if (prev) z[v] = alpha * z[v] + (1.0f - alpha) * Elem<T>::ld4(prev + row * ldp + col[v]);  // :201
            }
        }
        if (ln_w) {                                                                 // :202-203
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < V; ++v) s += z[v][0] + z[v][1] + z[v][2] + z[v][3];
#pragma unroll
            for (int m = 1; m < G; m <<= 1) s += __shfl_xor(s, m, 64);
            const float mu = s * inv_d;
            float var = 0.f;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                z[v] = ok[v] ? (z[v] - mu) : zero;
                var += z[v][0] * z[v][0] + z[v][1] * z[v][1] + z[v][2] * z[v][2] + z[v][3] * z[v][3];
            }
#pragma unroll
            for (int m = 1; m < G; m <<= 1) var += __shfl_xor(var, m, 64);
            const float rstd = 1.0f / sqrtf(var * inv_d + eps);
#pragma unroll
            for (int v = 0; v < V; ++v) z[v] = z[v] * rstd * w4[v] + b4[v];
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (relu) {
#pragma unroll
                for (int i = 0; i < 4; ++i) z[v][i] = fmaxf(z[v][i], 0.f);
            }
            if (ok[v]) Elem<T>::st4(out + row * ldo + col[v], z[v]);
        }
    }
}

// generic shapes: one wave per row; the row is re-derived in each of the three passes (inputs come from L1/L2),
// so nothing is parked in `out` at reduced precision.
template <typename T>
__global__ __launch_bounds__(256) void layer_tail_generic_kernel(const T* __restrict__ conv, int64_t ldc,
                                                                 int64_t n_rows, int H, int D,
                                                                 const T* __restrict__ x0, int64_t ldx0,
                                                                 const T* __restrict__ prev, int64_t ldp,
                                                                 float alpha, const T* __restrict__ ln_w,
                                                                 const T* __restrict__ ln_b, float eps, int relu,
                                                                 T* __restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const int64_t nw = static_cast<int64_t>(gridDim.x) * 4;
    const float inv_h = 1.0f / static_cast<float>(H);
    const float inv_d = 1.0f / static_cast<float>(D);
    for (int64_t row = gw; row < n_rows; row += nw) {
        auto zval = [&](int d) {
            float z = 0.f;
            for (int h = 0; h < H; ++h) z += Elem<T>::ld(conv + row * ldc + static_cast<int64_t>(h) * D + d);
            if (H > 1) z *= inv_h;
            if (x0) z += Elem<T>::ld(x0 + row * ldx0 + d);
            if (prev) z = alpha * z + (1.0f - alpha) * Elem<T>::ld(prev + row * ldp + d);
            return z;
        };
        if (!ln_w) {
            for (int d = lane; d < D; d += 64) {
                const float z = zval(d);
                Elem<T>::st(out + row * ldo + d, relu ? fmaxf(z, 0.f) : z);
            }
            continue;
        }
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += zval(d);
        const float mu = dif::wave_sum(s) * inv_d;
        float v = 0.f;
        for (int d = lane; d < D; d += 64) { const float dz = zval(d) - mu; v += dz * dz; }
        const float rstd = 1.0f / sqrtf(dif::wave_sum(v) * inv_d + eps);
        for (int d = lane; d < D; d += 64) {
            const float y = (zval(d) - mu) * rstd * Elem<T>::ld(ln_w + d) + Elem<T>::ld(ln_b + d);
            Elem<T>::st(out + row * ldo + d, relu ? fmaxf(y, 0.f) : y);
        }
    }
}

template <typename T>
int layer_tail_entry(const T* conv, int64_t ldc, int64_t n_rows, int H, int D, const T* x0, int64_t ldx0, const T* prev,
                     int64_t ldp, float alpha, const T* ln_weight, const T* ln_bias, float ln_eps, int relu, T* out,
                     int64_t ldo, dif_stream_t stream) {
    DIF_REQUIRE(n_rows > 0 && H > 0 && D > 0, DIF_E_BADARG, "dif_layer_tail: n_rows, H, D must be positive");
    DIF_REQUIRE(conv && out, DIF_E_BADARG, "dif_layer_tail: null pointer");
    DIF_REQUIRE((ln_weight == nullptr) == (ln_bias == nullptr), DIF_E_BADARG,
                "dif_layer_tail: ln_weight and ln_bias must be given together");
    DIF_REQUIRE(ldc >= static_cast<int64_t>(H) * D && ldo >= D && (!x0 || ldx0 >= D) && (!prev || ldp >= D), DIF_E_BADARG,
                "dif_layer_tail: leading dimension smaller than a row");
    DIF_REQUIRE(out != conv || H == 1, DIF_E_BADARG, "dif_layer_tail: in-place only for H == 1");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = (D % 4 == 0) && D <= 512 && (ldc % 4 == 0) && (ldo % 4 == 0) && (!x0 || ldx0 % 4 == 0) &&
                     (!prev || ldp % 4 == 0) && dif::aligned_v4<T>(conv) && dif::aligned_v4<T>(out) &&
                     (!x0 || dif::aligned_v4<T>(x0)) && (!prev || dif::aligned_v4<T>(prev)) &&
                     (!ln_weight || (dif::aligned_v4<T>(ln_weight) && dif::aligned_v4<T>(ln_bias)));
    // the generic kernel re-reads its inputs after writing `out`: no aliasing there
    DIF_REQUIRE(vec || (out != conv && out != x0 && out != prev), DIF_E_BADARG,
                "dif_layer_tail: in-place needs D % 4 == 0, D <= 512 and aligned rows");
    const int64_t cap = 8 * dif::kCUs;
    if (vec) {
        const int q = D / 4;
#define DIF_TAIL2(G, V)                                                                                        \
    do {                                                                                                       \
        int64_t gx = (n_rows + (256 / G) - 1) / (256 / G);                                                     \
        if (gx > cap) gx = cap;                                                                                \
        hipLaunchKernelGGL((layer_tail_vec_kernel<G, V, T>), dim3(static_cast<unsigned>(gx)), dim3(256), 0, st, conv, \
                           ldc, n_rows, H, D, x0, ldx0, prev, ldp, alpha, ln_weight, ln_bias, ln_eps, relu, out, ldo); \
    } while (0)
#define DIF_TAIL(G) DIF_TAIL2(G, 1)
        if (q <= 1) DIF_TAIL(1);
        else if (q <= 2) DIF_TAIL(2);
        else if (q <= 4) DIF_TAIL(4);
        else if (q <= 8) DIF_TAIL(8);
        else if (q <= 16) DIF_TAIL(16);
        else if (q <= 32) DIF_TAIL(32);
        else if (q <= 64) DIF_TAIL(64);
        else DIF_TAIL2(64, 2);
#undef DIF_TAIL
#undef DIF_TAIL2
    } else {
        int64_t gx = (n_rows + 3) / 4;
        if (gx > cap) gx = cap;
        hipLaunchKernelGGL((layer_tail_generic_kernel<T>), dim3(static_cast<unsigned>(gx)), dim3(256), 0, st, conv, ldc,
                           n_rows, H, D, x0, ldx0, prev, ldp, alpha, ln_weight, ln_bias, ln_eps, relu, out, ldo);
    }
    return dif::launch_status("layer_tail kernel");
}

}  // namespace

extern "C" int dif_layer_tail_f32(const float* conv, int64_t ldc, int64_t n_rows, int H, int D, const float* x0,
                                  int64_t ldx0, const float* prev, int64_t ldp, float alpha, const float* ln_weight,
                                  const float* ln_bias, float ln_eps, int relu, float* out, int64_t ldo,
                                  dif_stream_t stream) {
    return layer_tail_entry<float>(conv, ldc, n_rows, H, D, x0, ldx0, prev, ldp, alpha, ln_weight, ln_bias, ln_eps, relu,
                                   out, ldo, stream);
}

// closed form at the scripts' widths: see struct Mix.  fp32, H == 1, D % 4 == 0, D <= 512, 16-byte aligned rows.
extern "C" int dif_layer_tail_mix_f32(const float* conv, int64_t ldc, const float* den, int64_t ldden, float conv_scale,
                                      const float* add, int64_t lda, float add_scale, const float* rs, const float* bv,
                                      int64_t n_rows, int D, const float* x0, int64_t ldx0, const float* prev, int64_t ldp,
                                      float alpha, const float* ln_weight, const float* ln_bias, float ln_eps, int relu,
                                      float* out, int64_t ldo, dif_stream_t stream) {
    DIF_REQUIRE(n_rows > 0 && D > 0 && D % 4 == 0 && D <= 512, DIF_E_SHAPE,
                "dif_layer_tail_mix_f32: covers D %% 4 == 0, D <= 512 (got %d)", D);
    DIF_REQUIRE(conv && out && ((rs == nullptr) == (bv == nullptr)) && (!rs || add), DIF_E_BADARG,
                "dif_layer_tail_mix_f32: null pointer (rs and bv come together, with add)");
    DIF_REQUIRE((ln_weight == nullptr) == (ln_bias == nullptr), DIF_E_BADARG,
                "dif_layer_tail_mix_f32: ln_weight and ln_bias must be given together");
    auto rows_ok = [&](const float* p, int64_t ld) { return !p || (ld >= D && ld % 4 == 0 && dif::aligned16(p)); };
    DIF_REQUIRE(rows_ok(conv, ldc) && rows_ok(add, lda) && rows_ok(x0, ldx0) && rows_ok(prev, ldp) && rows_ok(out, ldo) &&
                    rows_ok(ln_weight, D) && rows_ok(ln_bias, D) && rows_ok(bv, D) && (!den || ldden >= 1), DIF_E_BADARG,
                "dif_layer_tail_mix_f32: rows must be 16-byte aligned with ld >= D, ld %% 4 == 0");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Mix mix = {den, ldden, conv_scale, add, lda, add_scale, rs, bv};
    const int64_t cap = 8 * dif::kCUs;
    const int q = D / 4;
#define DIF_MIX(G, V)                                                                                              \
    do {                                                                                                           \
        int64_t gx = (n_rows + (256 / G) - 1) / (256 / G);                                                         \
        if (gx > cap) gx = cap;                                                                                    \
        hipLaunchKernelGGL((layer_tail_vec_kernel<G, V, float, true>), dim3(static_cast<unsigned>(gx)), dim3(256), 0, st, conv, \
                           ldc, n_rows, 1, D, x0, ldx0, prev, ldp, alpha, ln_weight, ln_bias, ln_eps, relu, out, ldo, mix); \
    } while (0)
    if (q <= 16) DIF_MIX(16, 1);
    else if (q <= 32) DIF_MIX(32, 1);
    else if (q <= 64) DIF_MIX(64, 1);
    else DIF_MIX(64, 2);
#undef DIF_MIX
    return dif::launch_status("layer_tail_vec_kernel<mix>");
}

extern "C" int dif_layer_tail_bf16(const void* conv, int64_t ldc, int64_t n_rows, int H, int D, const void* x0,
                                   int64_t ldx0, const void* prev, int64_t ldp, float alpha, const void* ln_weight,
                                   const void* ln_bias, float ln_eps, int relu, void* out, int64_t ldo,
                                   dif_stream_t stream) {
    using B = dif::bf16;
    return layer_tail_entry<B>(static_cast<const B*>(conv), ldc, n_rows, H, D, static_cast<const B*>(x0), ldx0,
                               static_cast<const B*>(prev), ldp, alpha, static_cast<const B*>(ln_weight),
                               static_cast<const B*>(ln_bias), ln_eps, relu, static_cast<B*>(out), ldo, stream);
}

// f4: TransConv.full_attention(kernel='simple') over a batch of graphs -- physical particle/difformer-v2.py:80-111
//
//   out_i = (s q_i KtV_b + vsum_b) / (s q_i . ksum_b + n_b),   b = graph of node i,
//   s = 1 / (|Q|_F |K|_F) over the WHOLE batch (:82-83), KtV_b / ksum_b / vsum_b over graph b only (:93-106).
// The reference pads every graph to the longest one ([B, max_node, H, D], :87-91).  Here nothing is padded and
// nothing per graph ever reaches memory: one wave owns one (graph, head, 64-column tile), builds KtV_b in its
// accumulator registers with v_mfma_f32_16x16x4_f32 and applies it to the graph's own query rows straight from
// those registers.  The k-steps of the second product are ordered so that the accumulator a lane holds after the
// first product IS the B operand it must supply (m = 64 mt + 16 (lane/16) + 4 reg + t), so there is no LDS, no
// shuffle and no barrier.  ksum_b rides along as a 17th accumulator column (B = 1.0) and lands in the layout the
// denominator product needs.
//   phase 1, per 4 rows : kx[mt] = K[row0 + lane/16][64 mt + 4 (lane%16) ..+3],  vx = V[same row][64 dt + 4 (lane%16) ..+3]
//                         acc[mt][t][u] += kx[mt][t] (x) vx[u]     -> KtV[64 mt + 4 (4 (lane/16) + reg) + t][64 dt + 4 (lane%16) + u]
//   phase 2, per 16 rows: qf[mt][c]  = Q[row0 + lane%16][64 mt + 16 (lane/16) + 4 c ..+3]
//                         o[u] += qf[mt][reg][t] (x) acc[mt][t][u][reg]    -> out[row0 + 4 (lane/16) + r][64 dt + 4 (lane%16) + u]
// HBM-bound: 4 N H D 4 bytes per layer as a1 (+ one more pass over Q and K for the two global norms).
#include "dif_common.h"

namespace {

using dif::f32x4;

constexpr int kNormWG = 512;      // workgroups (= partial sums) of the norm pass

template <bool VEC>
__device__ __forceinline__ f32x4 ld_row4(const float* __restrict__ base, int64_t ld, int64_t r, bool rok, int col0,
                                         int c, int width) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (!rok) return z;
    const float* p = base + r * ld + col0 + c;
    if (VEC) {
        if (c < width) z = *reinterpret_cast<const f32x4*>(p);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (c + i < width) z[i] = p[i];
    }
    return z;
}

// ---- sum q^2, sum k^2 over all rows: partials in a fixed order, then one workgroup folds them (deterministic) ----
__global__ __launch_bounds__(256) void sumsq_pair_kernel(const float* __restrict__ q, int64_t ldq,
                                                         const float* __restrict__ k, int64_t ldk, int64_t n_rows,
                                                         int width, float* __restrict__ part) {
    __shared__ float sm[4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w4 = (width + 3) / 4;                       // float4 slots per row
    const int64_t total = n_rows * w4;
    const bool vec = (width % 4 == 0) && (ldq % 4 == 0) && (ldk % 4 == 0) && ((reinterpret_cast<uintptr_t>(q) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(k) & 15) == 0);
    float a = 0.f, b = 0.f;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * 256) {
        const int64_t r = e / w4;
        const int c = static_cast<int>(e % w4) * 4;
        if (vec) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(q + r * ldq + c);
            const f32x4 y = *reinterpret_cast<const f32x4*>(k + r * ldk + c);
            a += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
            b += y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3];
        } else {
            for (int i = 0; i < 4 && c + i < width; ++i) {
                const float x = q[r * ldq + c + i], y = k[r * ldk + c + i];
                a += x * x;
                b += y * y;
            }
        }
    }
    a = dif::wave_sum(a);
    b = dif::wave_sum(b);
    if (lane == 0) { sm[wave][0] = a; sm[wave][1] = b; }
    __syncthreads();
    if (threadIdx.x < 2)
        part[2 * blockIdx.x + threadIdx.x] = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
}

__global__ __launch_bounds__(64) void sumsq_fold_kernel(const float* __restrict__ part, int n_part, float* __restrict__ out2) {
    const int lane = threadIdx.x;
    float a = 0.f, b = 0.f;
    for (int i = lane; i < n_part; i += 64) { a += part[2 * i]; b += part[2 * i + 1]; }
    a = dif::wave_sum(a);
    b = dif::wave_sum(b);
    if (lane == 0) { out2[0] = a; out2[1] = b; }
}

// ---- one wave (WAVES == 1) or one 4-wave workgroup (WAVES == 4) per (graph, head, 64-column tile) ------------------
// WAVES == 4 serves batches of few, larger graphs (not enough items to fill the chip with one wave each): the four
// waves split the graph's rows in both phases and fold their K^T V accumulators through LDS in a fixed order.
// RAW (the three products of the backward, dif_batched_simple_raw_f32): no "+ n_b", no division --
//   out_i = s q_i KtV_b + coef_i vsum_b,  coef_i = (rs ? rs[i, h] : 1) (vs_is_s ? s : 1),  vsum_b = sum_l (vw ? vw[l, h] : 1) v_l
struct BatchedExtra {
    const float* rs;       // RAW: per (row, head) factor of the vsum term
    const float* vw;       // RAW: per (row, head) weight inside vsum
    int vs_is_s;           // RAW: the vsum term carries the scale s
    float* den_out;        // !RAW: s q.ksum_b + n_b per (row, head), for the backward
};

template <int MT, bool VEC, int WAVES, bool RAW>
__global__ __launch_bounds__(256) void batched_simple_kernel(const float* __restrict__ q, int64_t ldq,
                                                             const float* __restrict__ k, int64_t ldk,
                                                             const float* __restrict__ v, int64_t ldv,
                                                             const int32_t* __restrict__ graph_ptr, int n_graphs, int H,
                                                             int M, int D, const float* __restrict__ sumsq,
                                                             float* __restrict__ out, int64_t ldo, BatchedExtra ex) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int DT = (D + 63) / 64;
    const int wv = (WAVES == 1) ? 0 : (threadIdx.x >> 6);                    // this wave's share of the item
    const int64_t item = (WAVES == 1) ? static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6) : blockIdx.x;
    if (item >= static_cast<int64_t>(n_graphs) * H * DT) return;             // uniform over everything that syncs
    const int g = static_cast<int>(item % n_graphs);
    const int hd = static_cast<int>(item / n_graphs);
    const int h = hd / DT, dt = hd % DT;
    const int64_t r0 = graph_ptr[g], r1 = graph_ptr[g + 1];
    if (r1 <= r0) return;
    const float s = 1.0f / (sqrtf(sumsq[0]) * sqrtf(sumsq[1]));             // difformer-v2.py:82-83
    const float n_b = static_cast<float>(r1 - r0);                           // :107-109

    f32x4 acc[MT][4][4];   // [mt][t][u]
    f32x4 acck[MT][4];     // [mt][t]: ksum as an extra output column
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acck[mt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[mt][t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    f32x4 vs = {0.f, 0.f, 0.f, 0.f};

    // ---- phase 1: KtV_b, ksum_b, vsum_b over the graph's rows (:93-106) ----------------------------------------
    constexpr int kU = (MT == 1) ? 4 : (MT == 2) ? 2 : 1;                                    // 4-row steps in flight
    for (int64_t rb = r0 + 4 * kU * wv; rb < r1; rb += 4 * kU * WAVES) {
        f32x4 kx[kU][MT], vx[kU];
#pragma unroll
        for (int st = 0; st < kU; ++st) {
            const int64_t r = rb + 4 * st + lg;
            const bool ok = r < r1;
            vx[st] = ld_row4<VEC>(v, ldv, r, ok, h * D, dt * 64 + 4 * l15, D);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) kx[st][mt] = ld_row4<VEC>(k, ldk, r, ok, h * M, mt * 64 + 4 * l15, M);
        }
#pragma unroll
        for (int st = 0; st < kU; ++st) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        acc[mt][t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kx[st][mt][t], vx[st][u], acc[mt][t][u], 0, 0, 0);
                    acck[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kx[st][mt][t], 1.0f, acck[mt][t], 0, 0, 0);
                }
            if (RAW && ex.vw) {
                const int64_t r = rb + 4 * st + lg;
                vs += (r < r1 ? ex.vw[r * H + h] : 0.f) * vx[st];
            } else {
                vs += vx[st];
            }
        }
    }
    // vsum: fold the four row groups; every lane then holds vsum[64 dt + 4 l15 + u], the columns it will write
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float a = vs[u];
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        vs[u] = a;
    }
    if (WAVES > 1) {
        // slot-major [slot][lane] float4: every lane reads back exactly the registers it wrote
        __shared__ __attribute__((aligned(16))) float sm[(WAVES > 1) ? (20 * MT + 1) * 256 : 4];
        f32x4* base = reinterpret_cast<f32x4*>(sm) + lane;
        for (int w = 0; w < WAVES; ++w) {
            if (wv == w) {
                int slot = 0;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int u = 0; u < 4; ++u, ++slot) {
                            if (w == 0) base[slot * 64] = acc[mt][t][u]; else base[slot * 64] += acc[mt][t][u];
                        }
                        if (w == 0) base[slot * 64] = acck[mt][t]; else base[slot * 64] += acck[mt][t];
                        ++slot;
                    }
                if (w == 0) base[slot * 64] = vs; else base[slot * 64] += vs;
            }
            __syncthreads();
        }
        int slot = 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int u = 0; u < 4; ++u, ++slot) acc[mt][t][u] = base[slot * 64];
                acck[mt][t] = base[slot * 64];
                ++slot;
            }
        vs = base[slot * 64];
    }

    // ---- phase 2: apply to the graph's own queries (:100-111) --------------------------------------------------
    for (int64_t rb = r0 + 16 * wv; rb < r1; rb += 16 * WAVES) {
        f32x4 qf[MT][4];
        const int64_t rq = rb + l15;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int c = 0; c < 4; ++c) qf[mt][c] = ld_row4<VEC>(q, ldq, rq, rq < r1, h * M, mt * 64 + 16 * lg + 4 * c, M);
        f32x4 o[4], den = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        o[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[mt][reg][t], acc[mt][t][u][reg], o[u], 0, 0, 0);
                    den = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[mt][reg][t], acck[mt][t][reg], den, 0, 0, 0);
                }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = rb + 4 * lg + r;
            if (row >= r1) continue;
            f32x4 y;
            if (RAW) {
                const float coef = (ex.rs ? ex.rs[row * H + h] : 1.0f) * (ex.vs_is_s ? s : 1.0f);
#pragma unroll
                for (int u = 0; u < 4; ++u) y[u] = s * o[u][r] + coef * vs[u];
            } else {
                const float dn = s * den[r] + n_b;
                const float inv = 1.0f / dn;
                if (ex.den_out && dt == 0 && l15 == 0) ex.den_out[row * H + h] = dn;
#pragma unroll
                for (int u = 0; u < 4; ++u) y[u] = (s * o[u][r] + vs[u]) * inv;
            }
            float* dst = out + row * ldo + h * D + dt * 64 + 4 * l15;
            const int c0 = dt * 64 + 4 * l15;
            if (VEC) {
                if (c0 < D) *reinterpret_cast<f32x4*>(dst) = y;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (c0 + u < D) dst[u] = y[u];
            }
        }
    }
}

}  // namespace

extern "C" size_t dif_batched_simple_workspace_bytes(void) { return static_cast<size_t>(2 * kNormWG + 2) * sizeof(float); }

namespace {

int launch_batched(hipStream_t st, const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                   const int32_t* graph_ptr, int n_graphs, int64_t n_rows, int H, int M, int D, const float* sumsq,
                   float* out, int64_t ldo, bool raw, const BatchedExtra& ex) {
    const int DT = (D + 63) / 64;
    const int64_t items = static_cast<int64_t>(n_graphs) * H * DT;
    // few items of many rows: a 4-wave workgroup per item (>= ~16 rows per wave and phase to be worth the fold)
    const bool wide = items < 8 * dif::kCUs && n_rows >= 64 * static_cast<int64_t>(n_graphs);
    const int64_t grid = wide ? items : (items + 3) / 4;
    DIF_REQUIRE(grid < (1ll << 31), DIF_E_RANGE, "dif_batched_simple: grid too large");
    const bool vec = (M % 4 == 0) && (D % 4 == 0) && (ldq % 4 == 0) && (ldk % 4 == 0) && (ldv % 4 == 0) && (ldo % 4 == 0) &&
                     dif::aligned16(q) && dif::aligned16(k) && dif::aligned16(v) && dif::aligned16(out);
    const int MT = (M + 63) / 64;
#define DIF_LAUNCH_BS2(MTV, V, W, R) \
    hipLaunchKernelGGL((batched_simple_kernel<MTV, V, W, R>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, st, q, ldq, k, ldk, \
                       v, ldv, graph_ptr, n_graphs, H, M, D, sumsq, out, ldo, ex)
#define DIF_LAUNCH_BS(MTV, V) \
    do { \
        if (wide) { if (raw) DIF_LAUNCH_BS2(MTV, V, 4, true); else DIF_LAUNCH_BS2(MTV, V, 4, false); } \
        else { if (raw) DIF_LAUNCH_BS2(MTV, V, 1, true); else DIF_LAUNCH_BS2(MTV, V, 1, false); } \
    } while (0)
    if (MT == 1) { if (vec) DIF_LAUNCH_BS(1, true); else DIF_LAUNCH_BS(1, false); }
    else if (MT == 2) { if (vec) DIF_LAUNCH_BS(2, true); else DIF_LAUNCH_BS(2, false); }
    else { if (vec) DIF_LAUNCH_BS(4, true); else DIF_LAUNCH_BS(4, false); }
#undef DIF_LAUNCH_BS
#undef DIF_LAUNCH_BS2
    return dif::launch_status("batched_simple_kernel");
}

int batched_forward(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                    const int32_t* graph_ptr, int n_graphs, int64_t n_rows, int H, int M, int D, float* out, int64_t ldo,
                    float* den, void* workspace, size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(n_graphs > 0 && n_rows > 0 && H > 0 && M > 0 && D > 0, DIF_E_BADARG,
                "dif_batched_simple_attn_f32: n_graphs, n_rows, H, M, D must be positive");
    DIF_REQUIRE(q && k && v && out && graph_ptr, DIF_E_BADARG, "dif_batched_simple_attn_f32: null pointer");
    DIF_REQUIRE(ldq >= H * M && ldk >= H * M && ldv >= H * D && ldo >= H * D, DIF_E_BADARG,
                "dif_batched_simple_attn_f32: leading dimension smaller than a row");
    DIF_REQUIRE(M <= 256, DIF_E_SHAPE,
                "dif_batched_simple_attn_f32: per-head width M = %d; the register-resident K^T V covers M <= 256", M);
    DIF_REQUIRE(n_rows < (1ll << 31), DIF_E_RANGE, "dif_batched_simple_attn_f32: graph_ptr is int32 (n_rows = %lld)",
                static_cast<long long>(n_rows));
    const size_t need = dif_batched_simple_workspace_bytes();
    DIF_REQUIRE(workspace && workspace_bytes >= need, DIF_E_WORKSPACE,
                "dif_batched_simple_attn_f32: workspace too small (%zu < %zu)", workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    float* sumsq = part + 2 * kNormWG;

    int64_t ng = (n_rows * ((static_cast<int64_t>(H) * M + 3) / 4) + 1023) / 1024;     // >= 4 float4 per thread
    if (ng > kNormWG) ng = kNormWG;
    if (ng < 1) ng = 1;
    hipLaunchKernelGGL(sumsq_pair_kernel, dim3(static_cast<unsigned>(ng)), dim3(256), 0, st, q, ldq, k, ldk, n_rows, H * M, part);
    if (int rc = dif::launch_status("sumsq_pair_kernel")) return rc;
    hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(64), 0, st, part, static_cast<int>(ng), sumsq);
    if (int rc = dif::launch_status("sumsq_fold_kernel")) return rc;
    const BatchedExtra ex = {nullptr, nullptr, 0, den};
    return launch_batched(st, q, ldq, k, ldk, v, ldv, graph_ptr, n_graphs, n_rows, H, M, D, sumsq, out, ldo, false, ex);
}

}  // namespace

extern "C" int dif_batched_simple_attn_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                                           int64_t ldv, const int32_t* graph_ptr, int n_graphs, int64_t n_rows, int H,
                                           int M, int D, float* out, int64_t ldo, void* workspace,
                                           size_t workspace_bytes, dif_stream_t stream) {
    return batched_forward(q, ldq, k, ldk, v, ldv, graph_ptr, n_graphs, n_rows, H, M, D, out, ldo, nullptr, workspace,
                           workspace_bytes, stream);
}

// training: also den float[n_rows * H] (= s q.ksum_b + n_b); the two squared norms stay in the LAST two floats of the
// workspace for dif_batched_simple_raw_f32
extern "C" int dif_batched_simple_attn_fwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                                               int64_t ldv, const int32_t* graph_ptr, int n_graphs, int64_t n_rows, int H,
                                               int M, int D, float* out, int64_t ldo, float* den, void* workspace,
                                               size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(den != nullptr, DIF_E_BADARG, "dif_batched_simple_attn_fwd_f32: den is null");
    return batched_forward(q, ldq, k, ldk, v, ldv, graph_ptr, n_graphs, n_rows, H, M, D, out, ldo, den, workspace,
                           workspace_bytes, stream);
}

extern "C" int dif_batched_simple_raw_f32(const float* a, int64_t lda, const float* b, int64_t ldb, const float* c,
                                          int64_t ldc, const int32_t* graph_ptr, int n_graphs, int64_t n_rows, int H, int M,
                                          int D, const float* sumsq, const float* rs, const float* vw, int vs_is_s,
                                          float* out, int64_t ldo, dif_stream_t stream) {
    DIF_REQUIRE(n_graphs > 0 && n_rows > 0 && H > 0 && M > 0 && D > 0, DIF_E_BADARG,
                "dif_batched_simple_raw_f32: n_graphs, n_rows, H, M, D must be positive");
    DIF_REQUIRE(a && b && c && out && graph_ptr && sumsq, DIF_E_BADARG, "dif_batched_simple_raw_f32: null pointer");
    DIF_REQUIRE(lda >= H * M && ldb >= H * M && ldc >= H * D && ldo >= H * D, DIF_E_BADARG,
                "dif_batched_simple_raw_f32: leading dimension smaller than a row");
    DIF_REQUIRE(M <= 256, DIF_E_SHAPE, "dif_batched_simple_raw_f32: width of a / b = %d; covered up to 256", M);
    DIF_REQUIRE(n_rows < (1ll << 31), DIF_E_RANGE, "dif_batched_simple_raw_f32: graph_ptr is int32");
    const BatchedExtra ex = {rs, vw, vs_is_s, nullptr};
    return launch_batched(static_cast<hipStream_t>(stream), a, lda, b, ldb, c, ldc, graph_ptr, n_graphs, n_rows, H, M, D, sumsq,
                          out, ldo, true, ex);
}

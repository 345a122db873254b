// a1 backward: gradient of full_attention_conv(..., 'simple')
// (node classification/difformer.py:18-39) with respect to q, k, v -- what loss.backward() in
// main.py:130 / main-batch.py:141 needs.  The reference leaves this to autograd over ~17 ATen ops.
//
// With s = 1/(|Q||K|), num = s q KtV + vs, den = s q.ks + N, out = num/den and g = dL/dout:
//   gn = g / den                      gd = -(g . out) / den                         (per row, head)
//   dKtV = s sum_n q_n^T gn_n         dks = s sum_n q_n gd_n        dvs = sum_n gn_n
//   T    = -(sum_h vs_h . dvs_h) - N sum gd          ( = s * dL/ds )
//   dq = gn (s KtV)^T + gd (s ks) - (T / sum q^2) q
//   dk = v dKtV^T + dks - (T / sum k^2) k           dv = k dKtV + dvs
// Pass structure:  bwd_prep (rows -> gn, gd; partial sums of q*gd and gd)  ->  the forward's reduce kernel on
// (q, q, gn) gives q^T gn and sum gn  ->  three row-GEMMs  out = A Mat + bias + r (x) u + beta C  (this file).
#include "dif_common.h"
#include "rowgemm_split.h"

namespace {

using dif::f32x4;

// ---- prep: one 16-lane group per (row, head); D, M <= 64*4... handled by striding ------------------------------
// Writes gn [n,H,D], gd [n,H]; per-block partial of  sum_n q[n,h,m]*gd[n,h]  ([H*M]) and  sum gd  (1) into `part`.
__global__ __launch_bounds__(256) void simple_bwd_prep_kernel(const float* __restrict__ q, int64_t ldq,
                                                              const float* __restrict__ g, int64_t ldg,
                                                              const float* __restrict__ out, int64_t ldo,
                                                              const float* __restrict__ reduced, int64_t n_rows,
                                                              float n_global, int H, int M, int D,
                                                              float* __restrict__ gn, float* __restrict__ gd,
                                                              float* __restrict__ part, int part_stride) {
    extern __shared__ float sm[];   // [H*M + 1] block accumulators
    const int t_ks = H * M * D, t_main = H * M * D + H * M + H * D;
    const float s = 1.0f / (sqrtf(reduced[t_main]) * sqrtf(reduced[t_main + 1]));
    for (int i = threadIdx.x; i <= H * M; i += 256) sm[i] = 0.f;
    __syncthreads();
    const int lane16 = threadIdx.x & 15;
    const int64_t groups = n_rows * H;
    const int64_t gstride = static_cast<int64_t>(gridDim.x) * 16;
    float gd_acc = 0.f;
    for (int64_t gi = static_cast<int64_t>(blockIdx.x) * 16 + (threadIdx.x >> 4); gi < groups; gi += gstride) {
        const int64_t n = gi / H;
        const int h = static_cast<int>(gi % H);
        const float* qr = q + n * ldq + h * M;
        const float* gr = g + n * ldg + h * D;
        const float* orow = out + n * ldo + h * D;
        const float* ks = reduced + t_ks + h * M;
        float dot = 0.f, go = 0.f;
        for (int m = lane16; m < M; m += 16) dot += qr[m] * ks[m];
        for (int d = lane16; d < D; d += 16) go += gr[d] * orow[d];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { dot += __shfl_xor(dot, o, 64); go += __shfl_xor(go, o, 64); }
        const float den = s * dot + n_global;
        const float gdv = -go / den;
        for (int d = lane16; d < D; d += 16) gn[(n * H + h) * D + d] = gr[d] / den;
        if (lane16 == 0) { gd[n * H + h] = gdv; gd_acc += gdv; }
        for (int m = lane16; m < M; m += 16) atomicAdd(&sm[h * M + m], qr[m] * gdv);   // LDS atomics
    }
    if (lane16 == 0) atomicAdd(&sm[H * M], gd_acc);
    __syncthreads();
    for (int i = threadIdx.x; i <= H * M; i += 256) part[static_cast<int64_t>(blockIdx.x) * part_stride + i] = sm[i];
}

// The same pass for M, D <= 64 W (W = 1, 2: every width the reference's scripts reach with `simple` in one head up to 128)
// with 16-byte aligned rows: a 16-lane group owns ONE head and walks rows, a lane holds four fixed columns per 64-column
// chunk of q / g / out (one 16-byte load each), keeps its share of sum_n q * gd in registers, and the 16 groups of a
// workgroup fold through LDS in a fixed order -- deterministic, and no LDS atomics (the generic kernel above issues M of
// them per row: 98 us at C4 against 34 us for this one).  grid.x * 16 must be a multiple of H, so that a group's head never
// changes.
template <int W>
__global__ __launch_bounds__(256) void simple_bwd_prep_vec_kernel(const float* __restrict__ q, int64_t ldq,
                                                                  const float* __restrict__ g, int64_t ldg,
                                                                  const float* __restrict__ out, int64_t ldo,
                                                                  const float* __restrict__ reduced, int64_t n_rows,
                                                                  float n_global, int H, int M, int D,
                                                                  float* __restrict__ gn, float* __restrict__ gd,
                                                                  float* __restrict__ part, int part_stride) {
    __shared__ float sm[16][64 * W + 4];
    const int t_ks = H * M * D, t_main = H * M * D + H * M + H * D;
    const float s = 1.0f / (sqrtf(reduced[t_main]) * sqrtf(reduced[t_main + 1]));
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int64_t groups = n_rows * H;
    const int64_t gstride = static_cast<int64_t>(gridDim.x) * 16;
    const int64_t g0 = static_cast<int64_t>(blockIdx.x) * 16 + grp;
    const int h = static_cast<int>(g0 % H);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    bool mq[W], md[W];
    f32x4 ks[W], acc[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
        mq[w] = 64 * w + 4 * l16 < M;
        md[w] = 64 * w + 4 * l16 < D;
        ks[w] = mq[w] ? *reinterpret_cast<const f32x4*>(reduced + t_ks + h * M + 64 * w + 4 * l16) : z4;
        acc[w] = z4;
    }
    float gd_acc = 0.f;
    // The rows of the NEXT step are requested before this step's arithmetic: raw loads from clamped (valid) columns, masked
    // when used.  (Guarded -- `mq ? load : 0` -- every one of the 3 W loads was its own exec-masked block and a serialised round
    // trip to HBM: 66 us for 100,000 x 128 where the bytes take 26.)
    f32x4 qn[W], gnx[W], on[W];
    auto fetch = [&](int64_t gi) {
        const int64_t n = gi / H;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int c = 64 * w + 4 * l16;
            qn[w] = *reinterpret_cast<const f32x4*>(q + n * ldq + h * M + (mq[w] ? c : 0));
            gnx[w] = *reinterpret_cast<const f32x4*>(g + n * ldg + h * D + (md[w] ? c : 0));
            on[w] = *reinterpret_cast<const f32x4*>(out + n * ldo + h * D + (md[w] ? c : 0));
        }
    };
    if (g0 < groups) fetch(g0);
    for (int64_t gi = g0; gi < groups; gi += gstride) {
        const int64_t n = gi / H;
        f32x4 q4[W], g4[W], o4[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            q4[w] = mq[w] ? qn[w] : z4;
            g4[w] = md[w] ? gnx[w] : z4;
            o4[w] = md[w] ? on[w] : z4;
        }
        fetch(gi + gstride < groups ? gi + gstride : gi);
        __builtin_amdgcn_sched_barrier(0);
        float dot = 0.f, go = 0.f;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            dot += (q4[w][0] * ks[w][0] + q4[w][1] * ks[w][1]) + (q4[w][2] * ks[w][2] + q4[w][3] * ks[w][3]);
            go += (g4[w][0] * o4[w][0] + g4[w][1] * o4[w][1]) + (g4[w][2] * o4[w][2] + g4[w][3] * o4[w][3]);
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { dot += __shfl_xor(dot, o, 64); go += __shfl_xor(go, o, 64); }
        const float rden = 1.0f / (s * dot + n_global);
        const float gdv = -go * rden;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            if (md[w]) *reinterpret_cast<f32x4*>(gn + (n * H + h) * D + 64 * w + 4 * l16) = g4[w] * rden;
            acc[w] += q4[w] * gdv;
        }
        if (l16 == 0) { gd[n * H + h] = gdv; gd_acc += gdv; }
    }
#pragma unroll
    for (int w = 0; w < W; ++w) *reinterpret_cast<f32x4*>(&sm[grp][64 * w + 4 * l16]) = acc[w];
    if (l16 == 0) sm[grp][64 * W] = gd_acc;
    __syncthreads();
    // groups with the same head: grp = h', h' + H, ... when 16 % H == 0; in general (block * 16 + grp) % H
    for (int i = threadIdx.x; i <= H * M; i += 256) {
        const bool scalar = i == H * M;
        const int hh = scalar ? -1 : i / M, m = scalar ? 64 * W : i % M;
        float a = 0.f;
        for (int g2 = 0; g2 < 16; ++g2) {
            const int h2 = static_cast<int>((static_cast<int64_t>(blockIdx.x) * 16 + g2) % H);
            if (scalar || h2 == hh) a += sm[g2][m];
        }
        part[static_cast<int64_t>(blockIdx.x) * part_stride + i] = a;
    }
}

// ---- row-GEMM: out[n, :C] = A[n, :K] Mat[K x C] + bias[C] + r[n] u[C] + beta Cin[n, :C]   (K, C <= 64, per head) ----
// Same transposed MFMA formulation as simple_apply_kernel: D[i <-> c][j <-> row] = sum_k Mat[k][c] A[row][k].
// mat_t != 0: Mat is given transposed in memory (Mat[k][c] = mem[c * ldm + k]).
__global__ __launch_bounds__(256) void rowgemm_kernel(const float* __restrict__ A, int64_t lda, int a_head_stride,
                                                      const float* __restrict__ Mat, int ldm, int mat_head_stride,
                                                      int mat_t, float mat_scale, const float* __restrict__ bias,
                                                      int bias_head_stride, const float* __restrict__ r, int H,
                                                      const float* __restrict__ u, int u_head_stride, float u_scale,
                                                      const float* __restrict__ Cin, int64_t ldc,
                                                      const float* __restrict__ beta_dev,
                                                      int64_t n_rows, int K, int C, float* __restrict__ out,
                                                      int64_t ldo, int vec) {
    __shared__ float sm_m[64 * 68];
    const int h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const float* mat = Mat + static_cast<int64_t>(h) * mat_head_stride;
    {   // all 16 loads of a thread in flight before the first LDS store (a load -> store loop is 16 round trips)
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = threadIdx.x + 256 * u;
            const int k = mat_t ? (e & 63) : (e >> 6), c = mat_t ? (e >> 6) : (e & 63);     // coalesced either way
            v[u] = (k < K && c < C) ? mat_scale * (mat_t ? mat[c * ldm + k] : mat[k * ldm + c]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = threadIdx.x + 256 * u;
            const int k = mat_t ? (e & 63) : (e >> 6), c = mat_t ? (e >> 6) : (e & 63);
            sm_m[k * 68 + c] = v[u];
        }
    }
    __syncthreads();
    float afrag[4][4][4];   // [ctile][kq][t] = Mat[16kq + 4lg + t][16 ctile + l15]
#pragma unroll
    for (int kq = 0; kq < 4; ++kq)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) afrag[ct][kq][t] = sm_m[(16 * kq + 4 * lg + t) * 68 + 16 * ct + l15];

    const float beta = (Cin && beta_dev) ? *beta_dev : 1.0f;
    const int64_t n_steps = (n_rows + 15) / 16;
    for (int64_t st = static_cast<int64_t>(blockIdx.x) * 4 + wave; st < n_steps; st += static_cast<int64_t>(gridDim.x) * 4) {
        const int64_t row = st * 16 + l15;
        const bool rok = row < n_rows;
        f32x4 av[4];
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (rok) {
                const int k0 = 16 * kq + 4 * lg;
                if (vec) {
                    if (k0 < K) z = *reinterpret_cast<const f32x4*>(A + row * lda + h * a_head_stride + k0);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (k0 + i < K) z[i] = A[row * lda + h * a_head_stride + k0 + i];
                }
            }
            av[kq] = z;
        }
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kq = 0; kq < 4; ++kq)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[ct][kq][t], av[kq][t], acc[ct], 0, 0, 0);
        if (rok) {
            const float rv = r ? r[row * H + h] : 0.f;
            if (vec) {           // 4 consecutive output columns per lane and tile: 16-byte operands and stores
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const int c = 16 * ct + 4 * lg;
                    if (c >= C) continue;
                    f32x4 o = acc[ct];
                    if (bias) o += *reinterpret_cast<const f32x4*>(bias + h * bias_head_stride + c);
                    if (r) o += (rv * u_scale) * *reinterpret_cast<const f32x4*>(u + h * u_head_stride + c);
                    if (Cin) o += beta * *reinterpret_cast<const f32x4*>(Cin + row * ldc + h * C + c);
                    *reinterpret_cast<f32x4*>(out + row * ldo + h * C + c) = o;
                }
                continue;
            }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = 16 * ct + 4 * lg + i;      // lane holds out^T[c = 16ct + 4lg + i][row]
                    if (c < C) {
                        float o = acc[ct][i];
                        if (bias) o += bias[h * bias_head_stride + c];
                        if (r) o += rv * u_scale * u[h * u_head_stride + c];
                        if (Cin) o += beta * Cin[row * ldc + h * C + c];
                        out[row * ldo + h * C + c] = o;
                    }
                }
        }
    }
}

// ---- wide row-GEMM: the same contraction for heads wider than 64 (hidden 128 / 300 of the reference's scripts) ----
// A workgroup owns 64 output columns of one head (blockIdx.y) and keeps the whole [K x 64] slab of Mat in LDS, transposed
// (smT[c][k], so that a lane's four k-consecutive multipliers are one ds_read_b128); its eight waves walk 16-row steps,
// the next 64 channels of A arriving under the MFMAs of the current ones.  K <= 512 (132 KiB of LDS at 512).
constexpr int kRgWaves = 8;

template <bool VEC>
__global__ __launch_bounds__(64 * kRgWaves) void rowgemm_wide_kernel(const float* __restrict__ A, int64_t lda, int a_head_stride,
                                                                     const float* __restrict__ Mat, int ldm, int mat_head_stride,
                                                                     int mat_t, float mat_scale, const float* __restrict__ bias,
                                                                     const float* __restrict__ r, int H,
                                                                     const float* __restrict__ u, float u_scale,
                                                                     const float* __restrict__ Cin, int64_t ldc,
                                                                     const float* __restrict__ beta_dev, int64_t n_rows, int K,
                                                                     int C, float* __restrict__ out, int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) float sm_dyn[];
    const int h = blockIdx.z, ct = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int KT = (K + 63) >> 6, Kp = KT * 64, ldt = Kp + 4;
    const float* mat = Mat + static_cast<int64_t>(h) * mat_head_stride;
    const int total = Kp * 64;
    for (int base = threadIdx.x; base < total; base += 64 * kRgWaves * 8) {           // eight loads in flight per thread
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = base + 64 * kRgWaves * i;
            const int k = mat_t ? e % Kp : e >> 6, c = ct * 64 + (mat_t ? e / Kp : e & 63);        // coalesced either way
            v[i] = (e < total && k < K && c < C) ? mat_scale * (mat_t ? mat[static_cast<int64_t>(c) * ldm + k]
                                                                      : mat[static_cast<int64_t>(k) * ldm + c]) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = base + 64 * kRgWaves * i;
            if (e < total) sm_dyn[(mat_t ? e / Kp : e & 63) * ldt + (mat_t ? e % Kp : e >> 6)] = v[i];
        }
    }
    __syncthreads();

    const float beta = (Cin && beta_dev) ? *beta_dev : 1.0f;
    const int64_t n_steps = (n_rows + 15) / 16;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kRgWaves + wave;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kRgWaves;
    auto load_a = [&](f32x4 (&av)[4], int64_t st, int kt) {
        const int64_t row = st * 16 + l15;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const int k0 = kt * 64 + 16 * c + 4 * lg;
            if (row < n_rows) {
                const float* p = A + row * lda + h * a_head_stride + k0;
                if (VEC) {
                    if (k0 < K) z = *reinterpret_cast<const f32x4*>(p);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (k0 + i < K) z[i] = p[i];
                }
            }
            av[c] = z;
        }
    };
    f32x4 an[4];
    if (first < n_steps) load_a(an, first, 0);
    for (int64_t st = first; st < n_steps; st += stride) {
        const int64_t row = st * 16 + l15;
        f32x4 acc[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) acc[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < KT; ++kt) {
            f32x4 av[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) av[c] = an[c];
            if (kt + 1 < KT) load_a(an, st, kt + 1);
            else if (st + stride < n_steps) load_a(an, st + stride, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k0 = kt * 64 + 16 * c + 4 * lg;
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(&sm_dyn[(16 * t4 + l15) * ldt + k0]);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[t4] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[t], av[c][t], acc[t4], 0, 0, 0);
                }
            }
        }
        if (row >= n_rows) continue;
        const float rv = r ? r[row * H + h] * u_scale : 0.f;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const int c0 = ct * 64 + 16 * t4 + 4 * lg;              // the lane holds out^T[c0 .. c0 + 3][row]
            if (VEC) {
                if (c0 >= C) continue;
                f32x4 o = acc[t4];
                if (bias) o += *reinterpret_cast<const f32x4*>(bias + h * C + c0);
                if (r) o += rv * *reinterpret_cast<const f32x4*>(u + h * C + c0);
                if (Cin) o += beta * *reinterpret_cast<const f32x4*>(Cin + row * ldc + h * C + c0);
                *reinterpret_cast<f32x4*>(out + row * ldo + h * C + c0) = o;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = c0 + i;
                    if (c >= C) continue;
                    float o = acc[t4][i];
                    if (bias) o += bias[h * C + c];
                    if (r) o += rv * u[h * C + c];
                    if (Cin) o += beta * Cin[row * ldc + h * C + c];
                    out[row * ldo + h * C + c] = o;
                }
            }
        }
    }
}

// ---- the same contraction on split-bfloat16 operands (round 6: `simple` training at hidden 300 / 400, image and text/run.sh) ----
// rowgemm_wide_kernel is bound by the fp32 matrix core (64 v_mfma_f32_16x16x4_f32 of 32 cycles per 16 rows and 64 channels:
// 87 us for 15,000 x 300 -> 300, 20 % of that pipe's peak with the staging and the column blocks' re-reads of A).  Same shape of
// work -- a workgroup owns 64 output columns of one head and keeps their [K x 64] slab of Mat in LDS, its eight waves walk
// 16-row steps with the next 64 channels of A arriving under the products -- but the slab sits there as ready-split A fragments
// ([hi | lo][feature tile < 4][k-block][lane] x 16 bytes, as in rowgemm_split.h) and the rows are split as they arrive: three
// v_mfma_f32_16x16x32_bf16 per (feature tile, 32 channels) = 24 instructions of 16 cycles per 64 channels.  The dropped lo.lo
// term is 2^-16 of a product (~4e-6 of the result; the gradients are held to 1e-4); DIFFORMER_EXACT_FP32=1 keeps the fp32 kernel.
// K <= 512 and K, C multiples of 4 with 16-byte aligned rows (the launcher checks); LDS: 16 KiB per 64 channels.
typedef __bf16 rw_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 rw_bf16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64 * kRgWaves) void rowgemm_wide_split_kernel(const float* __restrict__ A, int64_t lda, int a_head_stride,
                                                                           const float* __restrict__ Mat, int ldm, int mat_head_stride,
                                                                           int mat_t, float mat_scale, const float* __restrict__ bias,
                                                                           const float* __restrict__ r, int H,
                                                                           const float* __restrict__ u, float u_scale,
                                                                           const float* __restrict__ Cin, int64_t ldc,
                                                                           const float* __restrict__ beta_dev, int64_t n_rows, int K,
                                                                           int C, float* __restrict__ out, int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) rw_bf16x8 sm_wfrag[];      // [hi | lo][ft < 4][kb < KB][lane]
    const int h = blockIdx.z, ct = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int KT = (K + 63) >> 6, KB = 2 * KT;
    const float* mat = Mat + static_cast<int64_t>(h) * mat_head_stride;
    const int n_frag = 4 * KB * 64;
    for (int e = threadIdx.x; e < n_frag; e += 64 * kRgWaves) {
        const int ln = e & 63, kb = (e >> 6) % KB, ft = (e >> 6) / KB;
        const int c = ct * 64 + 16 * ft + (ln & 15), k0 = 32 * kb + 4 * (ln >> 4);
        const int cc = c < C ? c : C - 1;
        f32x4 w0, w1;                       // raw loads from clamped indices, masked afterwards (a guarded load waits where it is issued)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ka = (k0 + t < K) ? k0 + t : K - 1, kc = (k0 + 16 + t < K) ? k0 + 16 + t : K - 1;
            w0[t] = mat_t ? mat[static_cast<int64_t>(cc) * ldm + ka] : mat[static_cast<int64_t>(ka) * ldm + cc];
            w1[t] = mat_t ? mat[static_cast<int64_t>(cc) * ldm + kc] : mat[static_cast<int64_t>(kc) * ldm + cc];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            w0[t] = (c < C && k0 + t < K) ? mat_scale * w0[t] : 0.f;
            w1[t] = (c < C && k0 + 16 + t < K) ? mat_scale * w1[t] : 0.f;
        }
        const rw_bf16x4 h0 = __builtin_convertvector(w0, rw_bf16x4), h1 = __builtin_convertvector(w1, rw_bf16x4);
        const rw_bf16x4 l0 = __builtin_convertvector(w0 - __builtin_convertvector(h0, f32x4), rw_bf16x4);
        const rw_bf16x4 l1 = __builtin_convertvector(w1 - __builtin_convertvector(h1, f32x4), rw_bf16x4);
        sm_wfrag[e] = rw_bf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        sm_wfrag[n_frag + e] = rw_bf16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    }
    __syncthreads();

    const float beta = (Cin && beta_dev) ? *beta_dev : 1.0f;
    const int64_t n_steps = (n_rows + 15) / 16;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kRgWaves + wave;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kRgWaves;
    // a tile of A: channels 64 kt + 16 c + 4 lg .. + 3 of row 16 st + l15 (raw loads from clamped addresses, masked when used)
    auto load_a = [&](f32x4 (&av)[4], int64_t st, int kt) {
        const int64_t row = st * 16 + l15;
        const float* base = A + (row < n_rows ? row : n_rows - 1) * lda + h * a_head_stride;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k0 = kt * 64 + 16 * c + 4 * lg;
            av[c] = *reinterpret_cast<const f32x4*>(base + (k0 < K ? k0 : 0));
        }
    };
    f32x4 an[4];
    load_a(an, first < n_steps ? first : n_steps - 1, 0);
    for (int64_t st = first; st < n_steps; st += stride) {
        const int64_t row = st * 16 + l15;
        const bool row_ok = row < n_rows;
        f32x4 acc[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) acc[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < KT; ++kt) {
            rw_bf16x8 xh[2], xl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 a0 = an[2 * j], a1 = an[2 * j + 1];
                if (!(row_ok && kt * 64 + 32 * j + 4 * lg < K)) a0 = f32x4{0.f, 0.f, 0.f, 0.f};
                if (!(row_ok && kt * 64 + 32 * j + 16 + 4 * lg < K)) a1 = f32x4{0.f, 0.f, 0.f, 0.f};
                const rw_bf16x4 h0 = __builtin_convertvector(a0, rw_bf16x4), h1 = __builtin_convertvector(a1, rw_bf16x4);
                const rw_bf16x4 l0 = __builtin_convertvector(a0 - __builtin_convertvector(h0, f32x4), rw_bf16x4);
                const rw_bf16x4 l1 = __builtin_convertvector(a1 - __builtin_convertvector(h1, f32x4), rw_bf16x4);
                xh[j] = rw_bf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                xl[j] = rw_bf16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            }
            if (kt + 1 < KT) load_a(an, st, kt + 1);                           // the next channels / rows arrive under these products
            else load_a(an, st + stride < n_steps ? st + stride : st, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const int at = (t4 * KB + 2 * kt + j) * 64 + lane;
                    const rw_bf16x8 wh = sm_wfrag[at], wl = sm_wfrag[n_frag + at];
                    acc[t4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[j], acc[t4], 0, 0, 0);      // small terms first
                    acc[t4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[j], acc[t4], 0, 0, 0);
                    acc[t4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[j], acc[t4], 0, 0, 0);
                }
        }
        if (!row_ok) continue;
        const float rv = r ? r[row * H + h] * u_scale : 0.f;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const int c0 = ct * 64 + 16 * t4 + 4 * lg;              // the lane holds out^T[c0 .. c0 + 3][row]
            if (c0 >= C) continue;
            f32x4 o = acc[t4];
            if (bias) o += *reinterpret_cast<const f32x4*>(bias + h * C + c0);
            if (r) o += rv * *reinterpret_cast<const f32x4*>(u + h * C + c0);
            if (Cin) o += beta * *reinterpret_cast<const f32x4*>(Cin + row * ldc + h * C + c0);
            *reinterpret_cast<f32x4*>(out + row * ldo + h * C + c0) = o;
        }
    }
}

}  // namespace

extern "C" size_t dif_simple_bwd_workspace_bytes(int64_t n_rows, int H, int M, int D) {
    if (n_rows <= 0 || H <= 0 || M <= 0 || D <= 0) return 0;
    const size_t P = 512;
    const size_t len = static_cast<size_t>(H) * M + 1;
    return (P * ((len + 3) & ~size_t(3)) + ((len + 3) & ~size_t(3))) * sizeof(float);
}

// gn [n,H,D], gd [n,H] and sums [H*M + 1] = { sum_n q*gd per (h,m), sum gd }.
extern "C" int dif_simple_bwd_prep_f32(const float* q, int64_t ldq, const float* g, int64_t ldg, const float* out,
                                       int64_t ldo, const float* reduced, int64_t n_rows, int64_t n_global, int H,
                                       int M, int D, float* gn, float* gd, float* sums, void* workspace,
                                       size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(n_rows > 0 && H > 0 && M > 0 && D > 0, DIF_E_BADARG, "dif_simple_bwd_prep_f32: sizes must be positive");
    DIF_REQUIRE(q && g && out && reduced && gn && gd && sums && workspace, DIF_E_BADARG,
                "dif_simple_bwd_prep_f32: null pointer");
    DIF_REQUIRE(workspace_bytes >= dif_simple_bwd_workspace_bytes(n_rows, H, M, D), DIF_E_WORKSPACE,
                "dif_simple_bwd_prep_f32: workspace too small");
    DIF_REQUIRE(static_cast<size_t>(H) * M + 1 <= 12000, DIF_E_SHAPE, "dif_simple_bwd_prep_f32: H*M too large for LDS");
    const int len = H * M + 1;
    const int stride = (len + 3) & ~3;
    int64_t P = (n_rows * H + 15) / 16;
    if (P > 512) P = 512;
    float* part = static_cast<float*>(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto row4 = [](const float* p, int64_t ld) { return ld % 4 == 0 && dif::aligned16(p); };
    const bool vec = M <= 128 && D <= 128 && M % 4 == 0 && D % 4 == 0 && row4(q, ldq) && row4(g, ldg) && row4(out, ldo) &&
                     dif::aligned16(gn) && dif::aligned16(reduced) && (static_cast<int64_t>(H) * M * D + H * M) % 4 == 0 && H <= 512;
    if (vec) {
        // a group keeps its head: the stride over (row, head) pairs, 16 * P, must be a multiple of H
        int64_t Pv = P;
        while ((Pv * 16) % H != 0) --Pv;
        if (Pv < 1) Pv = H;                                  // (H * 16) % H == 0; at most 512 partial records
        if (M <= 64 && D <= 64)
            hipLaunchKernelGGL(simple_bwd_prep_vec_kernel<1>, dim3(static_cast<unsigned>(Pv)), dim3(256), 0, st, q, ldq, g, ldg, out,
                               ldo, reduced, n_rows, static_cast<float>(n_global), H, M, D, gn, gd, part, stride);
        else
            hipLaunchKernelGGL(simple_bwd_prep_vec_kernel<2>, dim3(static_cast<unsigned>(Pv)), dim3(256), 0, st, q, ldq, g, ldg, out,
                               ldo, reduced, n_rows, static_cast<float>(n_global), H, M, D, gn, gd, part, stride);
        if (int rc = dif::launch_status("simple_bwd_prep_vec_kernel")) return rc;
        return dif::launch_record_finalize(part, static_cast<int>(Pv), stride, len, -1, sums, st);
    }
    hipLaunchKernelGGL(simple_bwd_prep_kernel, dim3(static_cast<unsigned>(P)), dim3(256), sizeof(float) * (len + 1), st, q,
                       ldq, g, ldg, out, ldo, reduced, n_rows, static_cast<float>(n_global), H, M, D, gn, gd, part, stride);
    if (int rc = dif::launch_status("simple_bwd_prep_kernel")) return rc;
    // column sums of the prep partials -> sums [H*M + 1] (16 slices per column, fixed order)
    return dif::launch_record_finalize(part, static_cast<int>(P), stride, len, -1, sums, st);
}

// out[n,h,:C] = A[n,h,:K] Mat_h + bias_h + r[n,h] * u_scale * u_h + (*beta_dev) * Cin[n,h,:C]    (K <= 512)
// beta_dev: DEVICE scalar (the backward's coefficients are computed on the device; no host round trip); NULL = 1.
extern "C" int dif_rowgemm_f32(const float* A, int64_t lda, const float* Mat, int ldm, int mat_head_stride, int mat_t,
                               float mat_scale, const float* bias, const float* r, const float* u, float u_scale,
                               const float* Cin, int64_t ldc, const float* beta_dev, int64_t n_rows, int H, int K,
                               int C, float* out, int64_t ldo, dif_stream_t stream) {
    DIF_REQUIRE(n_rows > 0 && H > 0 && K > 0 && C > 0, DIF_E_BADARG, "dif_rowgemm_f32: sizes must be positive");
    DIF_REQUIRE(K <= 512 && C <= 65535 * 64, DIF_E_SHAPE, "dif_rowgemm_f32: covers K <= 512 (got %d, %d)", K, C);
    DIF_REQUIRE(A && Mat && out && ((r == nullptr) == (u == nullptr)), DIF_E_BADARG, "dif_rowgemm_f32: null pointer");
    DIF_REQUIRE(H <= 65535, DIF_E_RANGE, "dif_rowgemm_f32: too many heads");
    const int64_t n_steps = (n_rows + 15) / 16;
    int64_t gx = (n_steps + 3) / 4;
    if (gx > 3 * dif::kCUs) gx = 3 * dif::kCUs;
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto ok4 = [](const void* p, int64_t ld) { return !p || (ld % 4 == 0 && dif::aligned16(p)); };
    const int vec = (K % 4 == 0) && (C % 4 == 0) && ok4(A, lda) && ok4(out, ldo) && ok4(Cin, ldc) && ok4(bias, 4) && ok4(u, 4);
    if ((K > 64 || C > 64) && K <= 128 && C <= 128 && H == 1 && vec && n_rows >= 4096 && !dif::exact_fp32()) {
        return rowgemm_split_launch("dif_rowgemm_f32", st, A, lda, Mat, ldm, mat_t, mat_scale, bias, r, u, u_scale, Cin, ldc, beta_dev, n_rows, K, C,
                                    out, ldo, nullptr, nullptr, 0.f);
    }
    if ((K > 64 || C > 64) && vec && n_rows >= 1024 && !dif::exact_fp32()) {
        const int KT = (K + 63) / 64, CT = (C + 63) / 64;
        const size_t lds = static_cast<size_t>(2) * 4 * (2 * KT) * 64 * 16;                  // 16 KiB per 64 channels
        constexpr int kLdsMaxSplit = 2 * 4 * 16 * 64 * 16;
        static const hipError_t allowed_split = hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_wide_split_kernel),
                                                                    hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMaxSplit);
        if (allowed_split != hipSuccess)
            return dif::fail(static_cast<int>(allowed_split), "dif_rowgemm_f32: LDS attribute: %s", hipGetErrorString(allowed_split));
        int64_t gw = (n_steps + 2 * kRgWaves - 1) / (2 * kRgWaves);
        int64_t cap = (lds <= 80 * 1024 ? 2 : 1) * dif::kCUs / (static_cast<int64_t>(CT) * H);      // one round of workgroups
        if (cap < 1) cap = 1;
        if (gw > cap) gw = cap;
        hipLaunchKernelGGL(rowgemm_wide_split_kernel, dim3(static_cast<unsigned>(gw), CT, H), dim3(64 * kRgWaves), lds, st, A, lda, K, Mat,
                           ldm, mat_head_stride, mat_t, mat_scale, bias, r, H, u, u_scale, Cin, ldc, beta_dev, n_rows, K, C, out, ldo);
        return dif::launch_status("rowgemm_wide_split_kernel");
    }
    if (K > 64 || C > 64) {
        const int KT = (K + 63) / 64, CT = (C + 63) / 64;
        const size_t lds = static_cast<size_t>(64) * (KT * 64 + 4) * sizeof(float);
        constexpr int kLdsMax = 64 * (8 * 64 + 4) * static_cast<int>(sizeof(float));
        static const hipError_t allowed[2] = {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_wide_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMax),
            hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_wide_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMax)};
        if (allowed[vec] != hipSuccess)
            return dif::fail(static_cast<int>(allowed[vec]), "dif_rowgemm_f32: LDS attribute: %s", hipGetErrorString(allowed[vec]));
        int64_t gw = (n_steps + 2 * kRgWaves - 1) / (2 * kRgWaves);          // >= 2 steps per wave: the staging is amortised
        // ONE round of workgroups over the chip, column blocks and heads included: 15,000 x 300 -> 300 (hidden 300 training,
        // image and text/run.sh) was 59 x 5 = 295 workgroups of 82 KB LDS on 256 compute units -- a second round for 39 of them
        // doubled the 40 us; a workgroup walks more row steps instead (its slab is staged once either way)
        int64_t cap = (lds <= 80 * 1024 ? 2 : 1) * dif::kCUs / (static_cast<int64_t>(CT) * H);
        if (cap < 1) cap = 1;
        if (gw > cap) gw = cap;
        const dim3 grid(static_cast<unsigned>(gw), CT, H), block(64 * kRgWaves);
        if (vec)
            hipLaunchKernelGGL(rowgemm_wide_kernel<true>, grid, block, lds, st, A, lda, K, Mat, ldm, mat_head_stride, mat_t,
                               mat_scale, bias, r, H, u, u_scale, Cin, ldc, beta_dev, n_rows, K, C, out, ldo);
        else
            hipLaunchKernelGGL(rowgemm_wide_kernel<false>, grid, block, lds, st, A, lda, K, Mat, ldm, mat_head_stride, mat_t,
                               mat_scale, bias, r, H, u, u_scale, Cin, ldc, beta_dev, n_rows, K, C, out, ldo);
        return dif::launch_status("rowgemm_wide_kernel");
    }
    hipLaunchKernelGGL(rowgemm_kernel, dim3(static_cast<unsigned>(gx), H), dim3(256), 0, st, A, lda, K, Mat, ldm,
                       mat_head_stride, mat_t, mat_scale, bias, C, r, H, u, C, u_scale, Cin, ldc, beta_dev, n_rows, K, C, out,
                       ldo, vec);
    return dif::launch_status("rowgemm_kernel");
}

// a1 + a4 + a5 tail in closed form for the `simple` kernel with query_input == source_input == x
// (node classification/difformer.py:18-39, :115-140, :200-203), H == 1, C_in <= 64, D <= 64, eval mode.
//
// With q = x Wq^T + bq, k = x Wk^T + bk, v = x Wv^T + bv every quantity stage 1 of the simple kernel needs is a
// function of the GRAM matrix G = X^T X, the column sums sx = sum_rows x and the weights:
//     K^T V  = Wk G Wv^T + (Wk sx) bv^T + bk (Wv sx)^T + N bk bv^T       sum k = Wk sx + N bk      sum v = Wv sx + N bv
//     |Q|^2  = tr(Wq G Wq^T) + 2 bq.(Wq sx) + N |bq|^2                    (|K|^2 alike)
// and stage 2 is linear in x:   num_i = x_i Mn + cn,  den_i = x_i.u + cd   with
//     Mn = s Wq^T KtV,  cn = s bq KtV + sum v,  u = s Wq^T (sum k),  cd = s bq.(sum k) + N,  s = 1/(|Q| |K|).
// gcn_conv is linear as well:  A_hat (x Wv^T + 1 bv^T) = (A_hat x) Wv^T + (A_hat 1) bv^T, so the SpMM runs on x itself.
// A layer is then
//     dif_gram_f32           one pass over x: G, sx (+ the pre-scaled slice-major copy of x the sliced SpMM reads)
//     dif_simple_coeffs_f32  one workgroup: Mn, cn, u, cd from the 4,160-float record (the only thing that would cross GPUs)
//     SpMM on x              ax = g_s A_hat x                                       (gcn_sliced.hip / gcn_spmm.hip)
//     dif_simple_layer_f32   out = LN(alpha (a_s num/den + ax Wv^T + g_s rs bv [+ x0]) + (1-alpha) x)
// q, k, v and the attention output never reach HBM: per layer the non-SpMM traffic is x read twice, ax read once,
// out (and the slice-major copy) written once, against q, v, attn, conv written and re-read before.
#include <type_traits>

#include "dif_common.h"

namespace {

using dif::f32x4;
using dif::Elem;

constexpr int kWaves = 4;
#ifndef DIF_HEAD_WAVES
#define DIF_HEAD_WAVES 8      // waves per workgroup of the HEAD variant (four 64 x 64 weight blocks in LDS: two workgroups per CU)
#endif
constexpr int kHeadWaves = DIF_HEAD_WAVES;
constexpr int kWStride = 68;     // padded LDS row (floats): 16 lanes x b128 land on 64 distinct banks
constexpr int kRecordChunksPerCU = 3;   // most workgroups per CU of any kernel that writes partial Gram records

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

__device__ __forceinline__ float dinv_of(const int32_t* __restrict__ rowptr, int64_t row) {
    const int32_t d = rowptr[row + 1] - rowptr[row];
    return d > 0 ? sqrtf(1.0f / static_cast<float>(d)) : 0.f;
}

// ------------------------------------------------------------------------------------------------------------
// Gram record.  A wave reads four whole rows per instruction (lane = 16*lg + l15: row base + lg, columns 4*l15..+3), which
// is at once the coalesced stream, the A / B operand of v_mfma_f32_16x16x4_f32 contracting over ROWS
//   D[i][j] += sum_k A[i][k] B[k][j],  A[i][k] = X[row k][4i + ta],  B[k][j] = X[row k][4j + tb]
//   -> lane holds G[4*(4lg + reg) + ta][4*l15 + tb]  in acc[ta][tb][reg]
// and one 16-byte slice of the row for the slice-major copy (ys[l15][row] = dinv[row] * x[row][4*l15..]).
// ------------------------------------------------------------------------------------------------------------
constexpr int kGramWaves = 8;

template <bool WRITE_YS, typename T = float>
__global__ __launch_bounds__(64 * kGramWaves, 2) void gram_kernel(const T* __restrict__ x, int64_t ldx, int64_t n_rows, int C,
                                                                  const int32_t* __restrict__ rowptr, f32x4* __restrict__ ys,
                                                                  int64_t npad, float* __restrict__ ws, int64_t ws_stride) {
    __shared__ float sm_f[4 * 40 * 64];          // fold buffer: 40 accumulator registers x 64 lanes for up to 4 waves
    __shared__ float sm_s[kGramWaves][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const bool col_ok = 4 * l15 < C;
    // G is symmetric: only the products with ta <= tb are formed (10 of 16), the record write mirrors them
    f32x4 acc[10];
#pragma unroll
    for (int a = 0; a < 10; ++a) acc[a] = zero4();
    f32x4 sx = zero4();
    const int64_t n16 = (n_rows + 15) / 16;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kGramWaves + wave;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kGramWaves;
    auto load16 = [&](f32x4 (&xv)[4], int64_t tile) {      // 16 consecutive rows: four loads in flight
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t row = tile * 16 + 4 * u + lg;
            xv[u] = (row < n_rows && col_ok) ? Elem<T>::ld4(x + row * ldx + 4 * l15) : zero4();
        }
    };
    if constexpr (sizeof(T) == 2 && !WRITE_YS) {
        // bfloat16 rows: the products of two bfloat16 numbers are exact in float32, so the record can come straight off the
        // bf16 matrix core -- v_mfma_f32_16x16x32_bf16 contracts 32 ROWS per instruction (two 16-row tiles) where the float32
        // path needs eight v_mfma_f32_16x16x4_f32, and no element is converted.  A lane's eight raw loads hold rows 4 u + lg
        // (u = 0..7) of the 32, columns 4 l15 .. + 3; its operand for column offset t is the t-th element of each of them
        // (k = 8 lg + u <-> row 4 u + lg: the same map on both sides of the product), packed by v_perm_b32.  The column sums
        // are one more product, against ones.  Float32-converted rows made this pass issue-bound at every size
        // (1,600,000 x 64: 131 us against 91 us for float32 rows, profiles/r05_experiments.md).
        typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        auto load32 = [&](uint2 (&raw)[8], int64_t tile) {        // tile: index of a 32-row block
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t row = tile * 32 + 4 * u + lg;
                raw[u] = (row < n_rows && col_ok) ? *reinterpret_cast<const uint2*>(x + row * ldx + 4 * l15) : make_uint2(0u, 0u);
            }
        };
        const int64_t n32 = (n_rows + 31) / 32;
        const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
        f32x4 accs[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) accs[t] = zero4();
        uint2 nraw[8];
        if (first < n32) load32(nraw, first);
        for (int64_t tile = first; tile < n32; tile += stride) {
            uint2 raw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) raw[u] = nraw[u];
            if (tile + stride < n32) load32(nraw, tile + stride);
            bf16x8_t op[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                u32x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t a0 = (t < 2) ? raw[2 * j].x : raw[2 * j].y, a1 = (t < 2) ? raw[2 * j + 1].x : raw[2 * j + 1].y;
                    // low halves (t even) or high halves (t odd) of the two dwords, row 2 j below row 2 j + 1
                    w[j] = __builtin_amdgcn_perm(a1, a0, (t & 1) ? 0x07060302u : 0x05040100u);
                }
                op[t] = __builtin_bit_cast(bf16x8_t, w);
            }
            int a = 0;
#pragma unroll
            for (int ta = 0; ta < 4; ++ta) {
                accs[ta] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op[ta], ones, accs[ta], 0, 0, 0);
#pragma unroll
                for (int tb = ta; tb < 4; ++tb, ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op[ta], op[tb], acc[a], 0, 0, 0);
            }
        }
        // accs[ta][reg] = column sum of column 4 (4 lg + reg) + ta, the same in every l15: the lanes with l15 == 0 file them
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                if (l15 == 0) sm_s[wave][4 * (4 * lg + reg) + ta] = accs[ta][reg];
    } else {
    f32x4 nxt[4];
    if (first < n16) load16(nxt, first);
    for (int64_t tile = first; tile < n16; tile += stride) {
        f32x4 xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = nxt[u];
        if (tile + stride < n16) load16(nxt, tile + stride);          // next tile's rows arrive under this tile's MFMAs
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (WRITE_YS) {
                const int64_t row = tile * 16 + 4 * u + lg;
                if (row < n_rows && col_ok) ys[static_cast<int64_t>(l15) * npad + row] = xv[u] * dinv_of(rowptr, row);
            }
            sx += xv[u];
            int a = 0;
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = ta; tb < 4; ++tb, ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][ta], xv[u][tb], acc[a], 0, 0, 0);
        }
    }
    }
    if (WRITE_YS) {          // rows n_rows .. npad-1 of the copy are read by the last tile of the sweep: zero them
        const int64_t pad = npad - n_rows;
        for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < pad * 16;
             i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
            const int64_t sl = i / pad, r = n_rows + i % pad;
            if (4 * sl < C) ys[sl * npad + r] = zero4();
        }
    }
    if constexpr (!(sizeof(T) == 2 && !WRITE_YS)) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float a = sx[t];
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        if (lg == 0) sm_s[wave][4 * l15 + t] = a;
    }
    }
    // fold the eight waves in registers, halving the live set three times through LDS (fixed order):
    // ((w0 + w4) + (w2 + w6)) + ((w1 + w5) + (w3 + w7))
#pragma unroll
    for (int half = kGramWaves / 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
#pragma unroll
            for (int i = 0; i < 10; ++i)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) sm_f[((wave - half) * 40 + i * 4 + reg) * 64 + lane] = acc[i][reg];
        }
        __syncthreads();
        if (wave < half) {
#pragma unroll
            for (int i = 0; i < 10; ++i)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) acc[i][reg] += sm_f[(wave * 40 + i * 4 + reg) * 64 + lane];
        }
        __syncthreads();
    }
    float* rec = ws + static_cast<int64_t>(blockIdx.x) * ws_stride;
    if (wave == 0) {
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
    } else if (wave == 1 && lane < C) {
        float a = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < kGramWaves; ++w2) a += sm_s[w2][lane];
        rec[C * C + lane] = a;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Input layer + Gram record + slice-major copy in ONE pass (difformer.py:188-191 feeding the first closed-form layer on a
// dense graph): h = ReLU(LayerNorm(x W^T + b)) for narrow inputs (C_in <= 64: ogbn-proteins has 8 features), written row-major
// (the layer kernel's operand), written again slice-major and scaled by deg^-1/2 (the sliced product's operand), and
// accumulated into h^T h | sum h (the coefficients' operand) -- three kernels of round 3 (skinny_linear 22 us, gram + copy
// 24 us at C4: h written once and read back once) in one, with h never read back.
// The product runs UNtransposed (rows of x as the A operand): a lane then holds y[ft][reg] = h[r0 + 4 lg + reg][4 l15 + ft],
// (rows permuted inside the tile, see load_x), i.e. for each of its four rows the 16-byte slice l15 -- one row-major store,
// one slice-major store -- and, read as
// A[i = l15][k = lg] / B[k = lg][j = l15], the operands of the row-contracting Gram MFMAs without a trip through LDS:
//   acc[(ta, tb)] += sum over reg of mfma(y[ta][reg], y[tb][reg])  ->  G[4 (4 lg + reg) + ta][4 l15 + tb], as gram_kernel.
// ------------------------------------------------------------------------------------------------------------
template <int KQ>
__global__ __launch_bounds__(64 * kGramWaves, 2) void input_gram_kernel(const float* __restrict__ x, int64_t ldx, int64_t n_rows, int C_in,
                                                                        const float* __restrict__ W, const float* __restrict__ bias,
                                                                        int D, const float* __restrict__ ln_w,
                                                                        const float* __restrict__ ln_b, float eps, int relu,
                                                                        float* __restrict__ out, int64_t ldo,
                                                                        const int32_t* __restrict__ rowptr, f32x4* __restrict__ ys,
                                                                        int64_t npad, float* __restrict__ ws, int64_t ws_stride,
                                                                        int xvec) {
    constexpr int kStride = 16 * KQ + 4;             // floats per LDS weight row: b128 reads of 16 rows spread over the banks
    __shared__ __attribute__((aligned(16))) float sm_w[64 * kStride];     // row 16 (f % 4) + f / 4 = feature f
    __shared__ float sm_f[4 * 40 * 64];
    __shared__ float sm_s[kGramWaves][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    for (int e = threadIdx.x; e < 64 * 16 * KQ; e += 64 * kGramWaves) {
        const int f = e / (16 * KQ), c = e % (16 * KQ);
        sm_w[(16 * (f & 3) + (f >> 2)) * kStride + c] = (f < D && c < C_in) ? W[static_cast<int64_t>(f) * C_in + c] : 0.f;
    }
    f32x4 bv = zero4(), lw = zero4(), lb = zero4();
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
        const int f = 4 * l15 + ft;
        if (f < D) {
            bv[ft] = bias[f];
            if (ln_w) { lw[ft] = ln_w[f]; lb[ft] = ln_b[f]; }
        }
    }
    const bool col_ok = 4 * l15 < D;                 // D % 4 == 0: the lane's four features are all real or all padding
    const float inv_d = 1.0f / static_cast<float>(D);
    __syncthreads();
    f32x4 acc[10];
#pragma unroll
    for (int a = 0; a < 10; ++a) acc[a] = zero4();
    f32x4 sx = zero4();
    const int64_t n16 = (n_rows + 15) / 16;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kGramWaves + wave;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kGramWaves;
    // Row i of the MFMA tile is row 4 (i % 4) + i / 4 of the 16-row tile: the lane's four result rows (i = 4 lg + reg) are
    // then rows 4 reg + lg, so that for one reg the four lanes of a slice hold four CONSECUTIVE rows -- 64 contiguous bytes
    // of the slice-major copy per store instruction instead of four 16-byte pieces 64 bytes apart.
    const int prow = 4 * (l15 & 3) + (l15 >> 2);
    auto load_x = [&](f32x4 (&xa)[KQ], int64_t tile) {
        const int64_t r = tile * 16 + prow;
#pragma unroll
        for (int cq = 0; cq < KQ; ++cq) {
            const int c = 16 * cq + 4 * lg;
            f32x4 z = zero4();
            if (r < n_rows) {
                const float* p = x + r * ldx + c;
                if (xvec && c + 3 < C_in) z = *reinterpret_cast<const f32x4*>(p);
                else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (c + i < C_in) z[i] = p[i];
                }
            }
            xa[cq] = z;
        }
    };
    f32x4 xa[KQ], xn[KQ];
    if (first < n16) load_x(xa, first);
    for (int64_t tile = first; tile < n16; tile += stride) {
        if (tile + stride < n16) load_x(xn, tile + stride);
        f32x4 y[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) y[ft] = f32x4{bv[ft], bv[ft], bv[ft], bv[ft]};
        const float* wrow = sm_w + l15 * kStride + 4 * lg;
#pragma unroll
        for (int cq = 0; cq < KQ; ++cq)
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                const f32x4 wf = *reinterpret_cast<const f32x4*>(wrow + 16 * ft * kStride + 16 * cq);
#pragma unroll
                for (int t = 0; t < 4; ++t) y[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[cq][t], wf[t], y[ft], 0, 0, 0);
            }
        // LayerNorm over the D features of each of the lane's four rows (a row's features: 16 lanes x 4 tiles), ReLU
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            if (ln_w) {
                float sm = 0.f;
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) sm += col_ok ? y[ft][reg] : 0.f;
                sm = dif::row16_sum(sm);
                const float mu = sm * inv_d;
                float v = 0.f;
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) { const float dz = col_ok ? y[ft][reg] - mu : 0.f; v += dz * dz; }
                v = dif::row16_sum(v);
                const float rstd = 1.0f / sqrtf(v * inv_d + eps);
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) y[ft][reg] = (y[ft][reg] - mu) * rstd * lw[ft] + lb[ft];
            }
            const int64_t row = tile * 16 + 4 * reg + lg;
            const bool ok = row < n_rows && col_ok;
            f32x4 h = {y[0][reg], y[1][reg], y[2][reg], y[3][reg]};
            if (relu) {
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = fmaxf(h[i], 0.f);
            }
            if (!ok) h = zero4();                       // rows past the matrix / padding columns stay out of the record
            if (ok) {
                *reinterpret_cast<f32x4*>(out + row * ldo + 4 * l15) = h;
                if (ys) ys[static_cast<int64_t>(l15) * npad + row] = h * dinv_of(rowptr, row);
            }
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) y[ft][reg] = h[ft];
            sx += h;
        }
        // Gram: contraction over the 16 rows of the tile, four rows (k = lg) per MFMA
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            int a = 0;
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = ta; tb < 4; ++tb, ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[ta][reg], y[tb][reg], acc[a], 0, 0, 0);
        }
#pragma unroll
        for (int cq = 0; cq < KQ; ++cq) xa[cq] = xn[cq];
    }
    if (ys) {                 // rows n_rows .. npad-1 of the copy are read by the last tile of the sweep: zero them
        const int64_t pad = npad - n_rows;
        for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < pad * 16;
             i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
            const int64_t sl = i / pad, r = n_rows + i % pad;
            if (4 * sl < D) ys[sl * npad + r] = zero4();
        }
    }
    // column sums: this lane's sx covers rows 4 lg + reg of every tile, features 4 l15 + t
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float a = sx[t];
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        if (lg == 0) sm_s[wave][4 * l15 + t] = a;
    }
    // fold the eight waves as gram_kernel does: ((w0 + w4) + (w2 + w6)) + ((w1 + w5) + (w3 + w7))
#pragma unroll
    for (int half = kGramWaves / 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
#pragma unroll
            for (int i = 0; i < 10; ++i)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) sm_f[((wave - half) * 40 + i * 4 + reg) * 64 + lane] = acc[i][reg];
        }
        __syncthreads();
        if (wave < half) {
#pragma unroll
            for (int i = 0; i < 10; ++i)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) acc[i][reg] += sm_f[(wave * 40 + i * 4 + reg) * 64 + lane];
        }
        __syncthreads();
    }
    float* rec = ws + static_cast<int64_t>(blockIdx.x) * ws_stride;
    if (wave == 0) {
        int i = 0;
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int tb = ta; tb < 4; ++tb, ++i)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int gi = 4 * (4 * lg + reg) + ta, gj = 4 * l15 + tb;
                    if (gi < D && gj < D) {
                        rec[gi * D + gj] = acc[i][reg];
                        if (ta != tb) rec[gj * D + gi] = acc[i][reg];
                    }
                }
    } else if (wave == 1 && lane < D) {
        float a = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < kGramWaves; ++w2) a += sm_s[w2][lane];
        rec[D * D + lane] = a;
    }
}

// workgroups of kWaves waves; `per_cu` = how many of them fit a CU (registers / LDS of the kernel in question)
int row_chunks(int64_t n_rows, int per_cu, int waves = kWaves) {
    const int64_t tiles = (n_rows + 15) / 16;
    int64_t p = (tiles + waves - 1) / waves;
    if (p > static_cast<int64_t>(per_cu) * dif::kCUs) p = static_cast<int64_t>(per_cu) * dif::kCUs;
    if (p < 1) p = 1;
    return static_cast<int>(p);
}
// the Gram kernel: eight waves per workgroup, at least four tiles per wave, one workgroup per CU at most (every
// workgroup pays a fold and a 16.6-KB partial record)
int gram_chunks(int64_t n_rows) {
    const int64_t tiles = (n_rows + 15) / 16;
    int64_t p = (tiles + 4 * kGramWaves - 1) / (4 * kGramWaves);
    if (p > dif::kCUs) p = dif::kCUs;
    if (p < 1) p = 1;
    return static_cast<int>(p);
}
// the input-layer kernel: ONE workgroup per CU, as the Gram kernel below.  Measured at the ogbn-proteins rows (8,283 tiles = 4.04 per
// wave, profiles/r06_experiments.md section 8): 36.3 us with one workgroup per CU, 40.6 with two, 50.4 with three -- every workgroup
// pays its weight staging, the fold of eight waves' Gram accumulators and a 16.6-KB partial record, which costs more than the
// second round of waves hides.  DIF_INPUT_GRAM_PER_CU=2 / 3 reproduces the comparison.
int input_gram_chunks(int64_t n_rows) {
    static const int per_cu = [] { const char* e = getenv("DIF_INPUT_GRAM_PER_CU"); return e ? atoi(e) : 1; }();
    if (per_cu <= 1) return gram_chunks(n_rows);
    const int64_t tiles = (n_rows + 15) / 16;
    int64_t p = (tiles + 2 * kGramWaves - 1) / (2 * kGramWaves);
    const int64_t cap = static_cast<int64_t>(per_cu > kRecordChunksPerCU ? kRecordChunksPerCU : per_cu) * dif::kCUs;
    if (p > cap) p = cap;
    if (p < 1) p = 1;
    return static_cast<int>(p);
}

// ------------------------------------------------------------------------------------------------------------
// Coefficients: one workgroup of 1024 threads, four 64^3 products through LDS.
//   coef = [MnT: D x C (feature-major, a_s folded in)][cn: D][u: C][cd][s][|Q|^2][|K|^2]
// ------------------------------------------------------------------------------------------------------------
constexpr int kLd = 68;

// one 16 x 16 tile of a 64 x 64 x 64 product on the fp32 MFMA: A(i, k), B(k, j) are LDS accessors;
// lane holds D[16ti + 4lg + reg][16tj + l15]
template <typename FA, typename FB>
__device__ __forceinline__ f32x4 tile_product(FA A, FB B, int ti, int tj, int l15, int lg) {
    f32x4 d = zero4();
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(A(16 * ti + l15, 4 * ks + lg), B(4 * ks + lg, 16 * tj + l15), d, 0, 0, 0);
    return d;
}

// LDS of the coefficient stage (1024 threads): six padded 64 x 64 blocks and a few vectors
struct CoefSmem {
    float *sG, *sWq, *sWk, *sWv, *sT, *sKtV;                                   // 64 x kLd each
    float *s_sx, *s_bq, *s_bk, *s_bv, *s_wk, *s_wq, *s_wv, *s_ks, *s_vs;       // 64 each
    float* s_red;                                                              // [2][16]
    float* s_scal;                                                             // [4]
};
constexpr int kCoefSmemFloats = 6 * 64 * kLd + 9 * 64 + 32 + 4;

__device__ __forceinline__ CoefSmem carve_coef_smem(float* base) {
    CoefSmem m;
    m.sG = base; m.sWq = m.sG + 64 * kLd; m.sWk = m.sWq + 64 * kLd; m.sWv = m.sWk + 64 * kLd; m.sT = m.sWv + 64 * kLd;
    m.sKtV = m.sT + 64 * kLd;
    m.s_sx = m.sKtV + 64 * kLd; m.s_bq = m.s_sx + 64; m.s_bk = m.s_bq + 64; m.s_bv = m.s_bk + 64; m.s_wk = m.s_bv + 64;
    m.s_wq = m.s_wk + 64; m.s_wv = m.s_wq + 64; m.s_ks = m.s_wv + 64; m.s_vs = m.s_ks + 64;
    m.s_red = m.s_vs + 64; m.s_scal = m.s_red + 32;
    return m;
}

// weights (and, with rec != nullptr, the Gram record) -> LDS.  1024 threads; no barrier at the end.
__device__ __forceinline__ void coeffs_stage(const CoefSmem& m, const float* __restrict__ rec, int C, int D,
                                             const float* __restrict__ Wq, const float* __restrict__ bq,
                                             const float* __restrict__ Wk, const float* __restrict__ bk,
                                             const float* __restrict__ Wv, const float* __restrict__ bv) {
    const int tid = threadIdx.x;
    const bool has_wv = Wv != nullptr;            // use_weight = False: v = x (needs C == D), Wv = I, bv = 0
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    if (C == 64 && D == 64 && (!rec || al16(rec)) && al16(Wq) && al16(Wk) && (!has_wv || al16(Wv))) {
        // dense 64 x 64 blocks: one 16-byte load per matrix and thread, all in flight together
        const int e = 4 * tid, a = e >> 6, b = e & 63;
        f32x4 g4 = zero4();
        if (rec) g4 = *reinterpret_cast<const f32x4*>(rec + e);
        const f32x4 q4 = *reinterpret_cast<const f32x4*>(Wq + e), k4 = *reinterpret_cast<const f32x4*>(Wk + e);
        f32x4 v4 = zero4();
        if (has_wv) v4 = *reinterpret_cast<const f32x4*>(Wv + e);
        else
            for (int r = 0; r < 4; ++r) v4[r] = (a == b + r) ? 1.f : 0.f;
        if (rec) *reinterpret_cast<f32x4*>(&m.sG[a * kLd + b]) = g4;
        *reinterpret_cast<f32x4*>(&m.sWq[a * kLd + b]) = q4;
        *reinterpret_cast<f32x4*>(&m.sWk[a * kLd + b]) = k4;
        *reinterpret_cast<f32x4*>(&m.sWv[a * kLd + b]) = v4;
    } else {
        for (int e = tid; e < 64 * 64; e += 1024) {
            const int a = e >> 6, b = e & 63;
            if (rec) m.sG[a * kLd + b] = (a < C && b < C) ? rec[a * C + b] : 0.f;
            m.sWq[a * kLd + b] = (a < D && b < C) ? Wq[a * C + b] : 0.f;
            m.sWk[a * kLd + b] = (a < D && b < C) ? Wk[a * C + b] : 0.f;
            m.sWv[a * kLd + b] = has_wv ? ((a < D && b < C) ? Wv[a * C + b] : 0.f) : (a == b && a < D ? 1.f : 0.f);
        }
    }
    if (tid < 64) {
        if (rec) m.s_sx[tid] = tid < C ? rec[C * C + tid] : 0.f;
        m.s_bq[tid] = tid < D ? bq[tid] : 0.f;
        m.s_bk[tid] = tid < D ? bk[tid] : 0.f;
        m.s_bv[tid] = (has_wv && tid < D) ? bv[tid] : 0.f;
    }
}

// Mn, cn, u, cd from G, sx and the weights, all staged in LDS (a barrier must separate the staging from this call).
// 1024 threads = 16 waves = the 16 tiles of each 64^3 product.
__device__ __forceinline__ void coeffs_compute(const CoefSmem& m, float n_global, int C, int D, float attn_scale,
                                               float* __restrict__ coef) {
    float *sG = m.sG, *sWq = m.sWq, *sWk = m.sWk, *sWv = m.sWv, *sT = m.sT, *sKtV = m.sKtV;
    float *s_sx = m.s_sx, *s_bq = m.s_bq, *s_bk = m.s_bk, *s_bv = m.s_bv, *s_wk = m.s_wk, *s_wq = m.s_wq, *s_wv = m.s_wv,
          *s_ks = m.s_ks, *s_vs = m.s_vs, *s_scal = m.s_scal;
    float (*s_red)[16] = reinterpret_cast<float (*)[16]>(m.s_red);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int ti = wave >> 2, tj = wave & 3;       // this wave's 16 x 16 tile of every 64 x 64 product
    // W sx for the three projections: 192 dot products, four lanes per row (16 columns each), folded by two shuffles
    if (tid < 768) {
        const int r = tid >> 2, part = tid & 3;
        const int mm = r & 63, which = r >> 6;
        const float* W = (which == 0 ? sWk : which == 1 ? sWq : sWv) + mm * kLd + 16 * part;
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) a += W[c] * s_sx[16 * part + c];
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        if (part == 0) (which == 0 ? s_wk : which == 1 ? s_wq : s_wv)[mm] = a;
    }
    // T = Wk G (kept), Wq G (only its trace against Wq): |K|^2, |Q|^2 main terms
    {
        const f32x4 tk = tile_product([&](int i, int k) { return sWk[i * kLd + k]; }, [&](int k, int j) { return sG[k * kLd + j]; },
                                      ti, tj, l15, lg);
        const f32x4 tq = tile_product([&](int i, int k) { return sWq[i * kLd + k]; }, [&](int k, int j) { return sG[k * kLd + j]; },
                                      ti, tj, l15, lg);
        float pk = 0.f, pq = 0.f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int mm = 16 * ti + 4 * lg + reg, c = 16 * tj + l15;
            sT[mm * kLd + c] = tk[reg];
            pk += tk[reg] * sWk[mm * kLd + c];
            pq += tq[reg] * sWq[mm * kLd + c];
        }
        pq = dif::wave_sum(pq);
        pk = dif::wave_sum(pk);
        if (lane == 0) { s_red[0][wave] = pq; s_red[1][wave] = pk; }
    }
    __syncthreads();
    // KtV = T Wv^T + rank-one terms; sum k, sum v; the scale
    {
        const f32x4 kv = tile_product([&](int i, int k) { return sT[i * kLd + k]; }, [&](int k, int j) { return sWv[j * kLd + k]; },
                                      ti, tj, l15, lg);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int mm = 16 * ti + 4 * lg + reg, d = 16 * tj + l15;
            sKtV[mm * kLd + d] = kv[reg] + s_wk[mm] * s_bv[d] + s_bk[mm] * s_wv[d] + n_global * s_bk[mm] * s_bv[d];
        }
    }
    if (tid < 64) {
        s_ks[tid] = s_wk[tid] + n_global * s_bk[tid];
        s_vs[tid] = s_wv[tid] + n_global * s_bv[tid];
    }
    if (wave == 15) {
        float q2 = lane < 16 ? s_red[0][lane] : 0.f, k2 = lane < 16 ? s_red[1][lane] : 0.f;
        q2 += 2.f * s_bq[lane] * s_wq[lane] + n_global * s_bq[lane] * s_bq[lane];
        k2 += 2.f * s_bk[lane] * s_wk[lane] + n_global * s_bk[lane] * s_bk[lane];
        q2 = dif::wave_sum(q2);
        k2 = dif::wave_sum(k2);
        if (lane == 0) {
            s_scal[0] = 1.0f / (sqrtf(q2) * sqrtf(k2));          // difformer.py:20-21: zero norms give inf / nan as there
            s_scal[1] = q2;
            s_scal[2] = k2;
        }
    }
    __syncthreads();
    const float s = s_scal[0];
    float* MnT = coef;
    float* cn = coef + D * C;
    float* u = cn + D;
    {   // MnT[d][c] = a_s s sum_m KtV[m][d] Wq[m][c]
        const f32x4 mn = tile_product([&](int i, int k) { return sKtV[k * kLd + i]; }, [&](int k, int j) { return sWq[k * kLd + j]; },
                                      ti, tj, l15, lg);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int d = 16 * ti + 4 * lg + reg, c = 16 * tj + l15;
            if (d < D && c < C) MnT[d * C + c] = attn_scale * s * mn[reg];
        }
    }
    if (tid < 256) {                     // cn[d] = a_s (s bq KtV + sum v): four lanes per column
        const int d = tid >> 2, part = tid & 3;
        float a = 0.f;
#pragma unroll
        for (int mm = 16 * part; mm < 16 * part + 16; ++mm) a += s_bq[mm] * sKtV[mm * kLd + d];
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        if (part == 0 && d < D) cn[d] = attn_scale * (s * a + s_vs[d]);
    } else if (tid < 512) {              // u[c] = s Wq^T (sum k)
        const int c = (tid - 256) >> 2, part = tid & 3;
        float a = 0.f;
#pragma unroll
        for (int mm = 16 * part; mm < 16 * part + 16; ++mm) a += sWq[mm * kLd + c] * s_ks[mm];
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        if (part == 0 && c < C) u[c] = s * a;
    } else if (wave == 8) {
        const float a = dif::wave_sum(s_bq[lane] * s_ks[lane]);
        if (lane == 0) {
            u[C] = s * a + n_global;
            u[C + 1] = s;
            u[C + 2] = s_scal[1];
            u[C + 3] = s_scal[2];
        }
    }
}

__global__ __launch_bounds__(1024) void coeffs_kernel(const float* __restrict__ rec, float n_global, int C, int D,
                                                      const float* __restrict__ Wq, const float* __restrict__ bq,
                                                      const float* __restrict__ Wk, const float* __restrict__ bk,
                                                      const float* __restrict__ Wv, const float* __restrict__ bv,
                                                      float attn_scale, float* __restrict__ coef) {
    __shared__ __attribute__((aligned(16))) float smem[kCoefSmemFloats];
    const CoefSmem m = carve_coef_smem(smem);
    coeffs_stage(m, rec, C, D, Wq, bq, Wk, bk, Wv, bv);
    __syncthreads();
    coeffs_compute(m, n_global, C, D, attn_scale, coef);
}

// The same with the Gram pass's PARTIAL records folded here (small graphs: Cora 6, PubMed 39 partial records): the chain
// Gram -> finalize -> coefficients -> layer of a small configuration is four dependent launches of 5-12 us each, and the
// finalize launch (4.6 us + its gap) only adds up a few 16.6-KB records.  Partials are summed in ascending chunk order
// (fixed: deterministic), P independent loads per thread in flight; `record` (nullable) receives the summed record.
constexpr int kFoldedPartsMax = 48;
__global__ __launch_bounds__(1024) void coeffs_parts_kernel(const float* __restrict__ parts, int P, int64_t stride, float n_global,
                                                            int C, int D, const float* __restrict__ Wq, const float* __restrict__ bq,
                                                            const float* __restrict__ Wk, const float* __restrict__ bk,
                                                            const float* __restrict__ Wv, const float* __restrict__ bv,
                                                            float attn_scale, float* __restrict__ coef, float* __restrict__ record) {
    __shared__ __attribute__((aligned(16))) float smem[kCoefSmemFloats];
    const CoefSmem m = carve_coef_smem(smem);
    coeffs_stage(m, nullptr, C, D, Wq, bq, Wk, bk, Wv, bv);
    const int tid = threadIdx.x;
    if (C == 64 && (stride & 3) == 0 && (reinterpret_cast<uintptr_t>(parts) & 15u) == 0) {
        const int e = 4 * tid;
        f32x4 a = zero4();
        for (int p0 = 0; p0 < P; p0 += 8) {          // eight partial records in flight per thread, summed in ascending order
            f32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = p0 + i < P ? *reinterpret_cast<const f32x4*>(parts + (p0 + i) * stride + e) : zero4();
#pragma unroll
            for (int i = 0; i < 8; ++i) a += v[i];
        }
        *reinterpret_cast<f32x4*>(&m.sG[(e >> 6) * kLd + (e & 63)]) = a;
        if (record) *reinterpret_cast<f32x4*>(record + e) = a;
    } else {
        for (int e = tid; e < 64 * 64; e += 1024) {
            const int a = e >> 6, b = e & 63;
            float v = 0.f;
            if (a < C && b < C) {
                for (int p = 0; p < P; ++p) v += parts[p * stride + a * C + b];
                if (record) record[a * C + b] = v;
            }
            m.sG[a * kLd + b] = v;
        }
    }
    if (tid < 64) {
        float v = 0.f;
        if (tid < C) {
            for (int p0 = 0; p0 < P; p0 += 8) {
                float w[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) w[i] = p0 + i < P ? parts[(p0 + i) * stride + C * C + tid] : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) v += w[i];
            }
            if (record) record[C * C + tid] = v;
        }
        m.s_sx[tid] = v;
    }
    __syncthreads();
    coeffs_compute(m, n_global, C, D, attn_scale, coef);
}

// ------------------------------------------------------------------------------------------------------------
// The layer: per 16-row tile and wave, two 64 x 64 MFMA products (x Mn, ax Wv^T), computed TRANSPOSED
//   D[i = feature][j = row] = sum_c W[feature][c] X[row][c]
//   A[i = lane%16][k = lane/16] = W[16ft + lane%16][16cq + 4*(lane/16) + t]    (ds_read_b128, LDS row stride 68)
//   B[k = lane/16][j = lane%16] = X[r0 + lane%16][16cq + 4*(lane/16) + t]      (one dwordx4 per cq: the coalesced stream)
//   -> lane (lg, l15) holds Y[row r0 + l15][features 16ft + 4lg .. +3]: four consecutive features of ONE row,
// i.e. the very layout of the loaded x fragment.  So the residual operand is already in registers, the denominator of
// the row is in the lane, LayerNorm folds over the four lane groups with two shuffles and rows leave as 16-byte stores.
// ------------------------------------------------------------------------------------------------------------
// T: storage type of the ACTIVATIONS (x, ax, x0, out: float or bfloat16); coefficients and parameters are float32
template <typename T>
struct LayerArgsT {
    const T* x; int64_t ldx;              // layer input [n, C]
    const T* ax; int64_t ldax;            // g_s * A_hat x [n, C] or null (use_graph = False)
    const float* coef;                    // dif_simple_coeffs_f32 output
    const float* Wv; const float* bv;     // [D, C], [D] or null (use_weight = False: the graph term is ax itself)
    const float* rs; float gcn_scale;     // row sums of A_hat (for the bias of the value projection) or null
    const T* x0; int64_t ldx0;            // use_source
    int residual; float alpha;            // alpha * z + (1 - alpha) * x
    const float* ln_w; const float* ln_b; float eps; int relu;
    T* out; int64_t ldo;
    int64_t n_rows; int C, D;
    // NEXT: the Gram record of `out` (the next layer's input) and its slice-major pre-scaled copy, from the same pass
    const int32_t* rowptr; f32x4* ys_next; int64_t npad; float* ws; int64_t ws_stride;
    // HEAD: the model's output Linear (difformer.py:208) applied to the finished rows in the same pass
    const float* Wo; const float* bo; int Co; T* logits; int64_t ldl;
    // GATHER: the aggregation itself in this pass (sparse graphs: a few entries per row) -- CSR over destination rows
    const int32_t* g_rowptr; const int32_t* g_src; const float* g_val;
};
using LayerArgs = LayerArgsT<float>;

template <bool GUARD, typename T>
__device__ __forceinline__ void load_rows(f32x4 (&xa)[4], const T* __restrict__ x, int64_t ldx, int64_t r, int64_t n, int lg,
                                          int C) {
    if (!GUARD) {
        const T* p = x + r * ldx + 4 * lg;
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) xa[cq] = Elem<T>::ld4g(p + 16 * cq);        // (ld4g: flat loads in the GATHER variants otherwise)
    } else {
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
            const int c = 16 * cq + 4 * lg;
            xa[cq] = (r < n && c < C) ? Elem<T>::ld4g(x + r * ldx + c) : zero4();
        }
    }
}

// GATHER: this lane's fragment of (A_hat x)[row] and the row sum of A_hat, straight from the CSR -- for graphs with a few
// entries per row (a Pokec mini-batch: 3.3, Cora: 4.9) the separate SpMM launch costs more than the rows it moves (22 us
// per layer at 100,000 rows).  The four lanes (lg) of a row walk its entries together; the 16 rows of a tile in lock step
// up to the longest of them, two entries per step.  The chain rowptr -> (src, val) -> x rows is three dependent loads:
// gather_begin issues the first two links BEFORE the tile's own projection (their latency hides under it), and inside the
// loop the indices of the next step are requested before the rows of this step are consumed.
// Measured at 100,000 rows x 64 columns (scripts/exp_layer_gather.py): +3.7 us per entry per row, i.e. 6.9 TB/s of gathered
// rows (L2 / MALL): throughput-, not latency-bound.  Two variants came out no faster (profiles/r03_experiments.md, 5): a
// walk over the tile's flattened entry range (4 rows per 16-lane group, eight rows in flight, transposed into this layout
// by 16 selector MFMAs), and the graph term finished before the attention term (no spills, but +1.5 .. 5 us).
struct GatherCursor {
    int e, e1;
    int s0, s1;
    float w0, w1;
};

__device__ __forceinline__ void gather_fetch(GatherCursor& g, const int32_t* __restrict__ src, const float* __restrict__ val) {
    // four INDEPENDENT loads: `two ? src[e + 1] : s0` made the second index wait for the first -- and, the memory counter being in
    // order, for the eight row loads issued before it: the "prefetch" of the next step's indices ran after this step's rows arrived
    const bool one = g.e < g.e1, two = g.e + 1 < g.e1;
    g.s0 = one ? src[g.e] : 0;
    g.w0 = one ? val[g.e] : 0.f;
    g.s1 = one ? src[two ? g.e + 1 : g.e] : 0;          // an odd tail re-reads its own entry (weight 0)
    g.w1 = two ? val[g.e + 1] : 0.f;
}

__device__ __forceinline__ GatherCursor gather_begin(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
                                                     const float* __restrict__ val, int64_t row, bool row_ok) {
    GatherCursor g = {0, 0, 0, 0, 0.f, 0.f};
    if (row_ok) { g.e = rowptr[row]; g.e1 = rowptr[row + 1]; }
    gather_fetch(g, src, val);
    return g;
}

template <bool EXACT, typename T>
__device__ __forceinline__ void gather_rows(f32x4 (&ga)[4], float& wsum, GatherCursor g, const T* __restrict__ x, int64_t ldx,
                                            const int32_t* __restrict__ src, const float* __restrict__ val, int lg, int C) {
#pragma unroll
    for (int cq = 0; cq < 4; ++cq) ga[cq] = zero4();
    wsum = 0.f;
    while (g.e < g.e1) {
        const T* p0 = x + static_cast<int64_t>(g.s0) * ldx;
        const T* p1 = x + static_cast<int64_t>(g.s1) * ldx;
        const float w0 = g.w0, w1 = g.w1;
        f32x4 r0[4], r1[4];
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
            const int c = 16 * cq + 4 * lg;
            const bool ok = EXACT || c < C;
            r0[cq] = ok ? Elem<T>::ld4g(p0 + c) : zero4();        // (rows addressed through loaded indices: flat loads without the hint)
            r1[cq] = ok ? Elem<T>::ld4g(p1 + c) : zero4();
        }
        g.e += 2;
        gather_fetch(g, src, val);                 // next step's indices: in flight while this step's rows arrive
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) ga[cq] += r0[cq] * w0 + r1[cq] * w1;
        wsum += w0 + w1;
    }
}

// Split-bfloat16 operands for the output Linear of the HEAD variant (as the long-row input Linear, csrc/skinny_linear.hip):
// v = hi + lo with hi = bf16(v), lo = bf16(v - hi); x.w ~ xl.wh + xh.wl + xh.wh on v_mfma_f32_16x16x32_bf16, which runs 16x
// the fp32 matrix rate (the dropped xl.wl term is 2^-16 of the product: ~4e-6 on the logits).  The 64 -> 128-class product
// was half of the kernel's 256 fp32 MFMAs per tile and the kernel was bound by them.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_bf16(const f32x4& v, bf16x4& hi, bf16x4& lo) {
    hi = __builtin_convertvector(v, bf16x4);
    const f32x4 back = __builtin_convertvector(hi, f32x4);
    lo = __builtin_convertvector(v - back, bf16x4);
}
__device__ __forceinline__ bf16x8 cat8(const bf16x4& a, const bf16x4& b) {
    return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// LDS layout of a 64 x 64 weight block for project_t: element (f, c) at [f/16][c/16][(c/4)%4][f%16][c%4].  The A fragment of
// a lane (lg, l15) is then the 16 bytes at lg * 256 + l15 * 16 inside the (ft, cq) sub-block: the bank quad depends on l15
// only, and every hardware lane group of a ds_read_b128 ({0-3,12-15,20-27}, ... MI355X_MICROARCH.md, LDS table) holds each
// l15 exactly once -> no bank conflicts.  (The row-major layout with a 68-float stride, conflict-free for groups of 16
// CONSECUTIVE lanes, put lanes (0, 12) and (1, 11) of the first hardware group on one quad: SQ_LDS_BANK_CONFLICT was 34 %
// of the kernel's LDS cycles, profiles/r02_pmc_traffic_c4.json.)
constexpr int kWBlock = 64 * 64;           // (the NEXT variant folds 2 x 40 x 64 floats through the two blocks: 8,192 there)
__device__ __forceinline__ int widx(int f, int c) {
    return ((((f >> 4) * 4 + (c >> 4)) * 4 + ((c >> 2) & 3)) << 6) + ((f & 15) << 2) + (c & 3);
}

// y[ft] (+)= W_tile x^T for the four feature tiles; W in LDS in the widx layout
__device__ __forceinline__ void project_t(f32x4 (&y)[4], const f32x4 (&xa)[4], const float* __restrict__ w, int l15, int lg) {
#pragma unroll
    for (int cq = 0; cq < 4; ++cq) {
        f32x4 wf[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) wf[ft] = *reinterpret_cast<const f32x4*>(&w[widx(16 * ft + l15, 16 * cq + 4 * lg)]);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)      // four independent accumulator chains back to back
                y[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ft][t], xa[cq][t], y[ft], 0, 0, 0);
    }
}

// Split-bfloat16 form of project_t for the 64 x 64 float32 layers (round 5): x = xh + xl, W = wh + wl, three terms
// xl.wh + xh.wl + xh.wh on v_mfma_f32_16x16x32_bf16 -- 24 instructions of 16 cycles per product where project_t issues 64 of 32
// (the 128 fp32 MFMAs per tile were 9 of the kernel's 34 us at the ogbn-proteins size, profiles/r04_experiments.md section 7);
// the dropped xl.wl term is 2^-16 of the product (~4e-6 on the layer's rows, as the output Linear and the hidden-128 kernel).
// W fragments [hi | lo][ft * 2 + kb][lane] in LDS: lane (lg, l15) holds feature 16 ft + l15, columns 16 (2 kb) + 4 lg .. + 3 and
// 16 (2 kb + 1) + 4 lg .. + 3 -- the k order of a row piece.  DIFFORMER_EXACT_FP32=1 keeps project_t.
__device__ __forceinline__ void project_split(f32x4 (&y)[4], const f32x4 (&xa)[4], const bf16x8* __restrict__ wh,
                                              const bf16x8* __restrict__ wl, int lane) {
    bf16x8 xh[2], xl[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        bf16x4 h0, l0, h1, l1;
        split_bf16(xa[2 * kb], h0, l0);
        split_bf16(xa[2 * kb + 1], h1, l1);
        xh[kb] = cat8(h0, h1);
        xl[kb] = cat8(l0, l1);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const bf16x8 ah = wh[(ft * 2 + kb) * 64 + lane], al = wl[(ft * 2 + kb) * 64 + lane];
            y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl[kb], y[ft], 0, 0, 0);      // small terms first
            y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh[kb], y[ft], 0, 0, 0);
            y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh[kb], y[ft], 0, 0, 0);
        }
}

#ifndef DIF_GATHER_WG
#define DIF_GATHER_WG 4           // workgroups per CU the GATHER variants are compiled for (probes / old layouts: scripts/variants/simple_layer.hip)
#endif
template <bool EXACT, bool GRAPH_W, bool NEXT, typename T = float, bool HEAD = false, bool GATHER = false, bool SPLIT = false>
__global__ __launch_bounds__(64 * (HEAD ? kHeadWaves : kWaves),
                             HEAD ? (2 * kHeadWaves + 3) / 4 : (NEXT ? 2 : (GATHER ? DIF_GATHER_WG : 4)))
void simple_layer_kernel(LayerArgsT<T> a) {
    static_assert(!SPLIT || (EXACT && !NEXT), "split-bf16 products: the dense 64 x 64 float32 layers");
    constexpr int NW = HEAD ? kHeadWaves : kWaves;          // waves per workgroup
    __shared__ __attribute__((aligned(16))) float sm_w[2][kWBlock];   // MnT, Wv (zero padded; widx layout)
    // HEAD: up to 128 output classes as split-bf16 A fragments, [hi | lo][(blk * 4 + ft) * 2 + kb][lane]: lane (lg, l15) holds
    // class 64 blk + 16 ft + l15, columns 16 (2 kb) + 4 lg .. + 3 and 16 (2 kb + 1) + 4 lg .. + 3 (the k order of a row piece)
    __shared__ __attribute__((aligned(16))) bf16x8 sm_wo[HEAD ? 2 : 1][HEAD ? 16 * 64 : 1];
    __shared__ __attribute__((aligned(16))) float sm_bo[HEAD ? 128 : 4];
    __shared__ __attribute__((aligned(16))) float sm_cn[64], sm_u[64], sm_bv[64], sm_lw[64], sm_lb[64];
    __shared__ __attribute__((aligned(16))) float sm_t[NEXT ? NW : 1][16 * kWStride];   // NEXT: a wave's finished tile
    __shared__ float sm_s[NW][64];
    __shared__ float sm_cd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int C = a.C, D = a.D;
    const int64_t n_tiles = (a.n_rows + 15) / 16;
    const int64_t n_fast = EXACT ? a.n_rows / 16 : 0;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * NW + wave, stride = static_cast<int64_t>(gridDim.x) * NW;
    if (EXACT && (reinterpret_cast<uintptr_t>(a.coef) & 15u) == 0 && (!GRAPH_W || (reinterpret_cast<uintptr_t>(a.Wv) & 15u) == 0)) {
        // dense 64 x 64 blocks: all eight 16-byte loads of a thread are in flight before the first LDS store (an
        // element-wise loop runs 32 load -> store round trips back to back: ~30 us of prologue per workgroup)
        // Thread -> fragment map of the staging: a wave instruction reads eight ROWS x 128 contiguous bytes (whole cache
        // lines), and the eight consecutive lanes that share a cycle of ds_write_b128 hold eight different rows, i.e.
        // eight different bank quads of the widx layout (conflict-free stores).
        constexpr int NL = 16 / NW;                 // 16-byte loads per thread and matrix
        f32x4 wreg[2][NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = wave + NW * i;                          // 0..15: (row block of 8, half row)
            const int f = 8 * (idx & 7) + (lane & 7), c = 32 * (idx >> 3) + 4 * (lane >> 3);
            wreg[0][i] = *reinterpret_cast<const f32x4*>(a.coef + f * 64 + c);
            if (GRAPH_W) wreg[1][i] = *reinterpret_cast<const f32x4*>(a.Wv + f * 64 + c);
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = wave + NW * i;
            const int f = 8 * (idx & 7) + (lane & 7), c = 32 * (idx >> 3) + 4 * (lane >> 3);
            if constexpr (SPLIT) {
                // the same 16-byte pieces, split into hi | lo halves of the MFMA fragment they belong to: piece (f, c) is half
                // (c / 16) % 2 of fragment (f / 16) * 2 + c / 32, lane 16 ((c / 4) % 4) + f % 16; the 64 lanes of a store
                // instruction fill 512 contiguous bytes (no bank conflicts)
                bf16x4* frag = reinterpret_cast<bf16x4*>(&sm_w[0][0]);      // [matrix][hi | lo][fragment][lane][half] x 8 bytes
                const int slot = ((((f >> 4) * 2 + (c >> 5)) * 64 + 16 * ((c >> 2) & 3) + (f & 15)) << 1) + ((c >> 4) & 1);
#pragma unroll
                for (int m = 0; m < (GRAPH_W ? 2 : 1); ++m) {
                    bf16x4 h, l;
                    split_bf16(wreg[m][i], h, l);
                    frag[(m * 2 + 0) * 1024 + slot] = h;
                    frag[(m * 2 + 1) * 1024 + slot] = l;
                }
            } else {
                *reinterpret_cast<f32x4*>(&sm_w[0][widx(f, c)]) = wreg[0][i];
                if (GRAPH_W) *reinterpret_cast<f32x4*>(&sm_w[1][widx(f, c)]) = wreg[1][i];
            }
        }
    } else {
        for (int e = threadIdx.x; e < 64 * 64; e += 64 * NW) {         // e = LDS dword: [ft][cq][lg][l15][t]
            const int f = 16 * (e >> 10) + ((e >> 2) & 15), c = 16 * ((e >> 8) & 3) + 4 * ((e >> 6) & 3) + (e & 3);
            sm_w[0][widx(f, c)] = (f < D && c < C) ? a.coef[f * C + c] : 0.f;
            if (GRAPH_W) sm_w[1][widx(f, c)] = (f < D && c < C) ? a.Wv[f * C + c] : 0.f;
        }
    }
    if (threadIdx.x < 64) {
        const int i = threadIdx.x;
        sm_cn[i] = i < D ? a.coef[D * C + i] : 0.f;
        sm_u[i] = i < C ? a.coef[D * C + D + i] : 0.f;
        sm_bv[i] = (GRAPH_W && (GATHER || a.rs) && i < D) ? a.bv[i] * a.gcn_scale : 0.f;
        sm_lw[i] = (a.ln_w && i < D) ? a.ln_w[i] : 1.f;
        sm_lb[i] = (a.ln_b && i < D) ? a.ln_b[i] : 0.f;
    }
    if (threadIdx.x == 0) sm_cd = a.coef[D * C + D + C];
    if constexpr (HEAD) {
        for (int e = threadIdx.x; e < 16 * 64; e += 64 * NW) {         // one A fragment (two 4-column pieces of a class row) each
            const int frag = e >> 6, ln = e & 63;
            const int cls = 64 * (frag >> 3) + 16 * ((frag >> 1) & 3) + (ln & 15);
            const int c0 = 32 * (frag & 1) + 4 * (ln >> 4);
            f32x4 w0 = zero4(), w1 = zero4();
            if (cls < a.Co) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (c0 + t < D) w0[t] = a.Wo[cls * D + c0 + t];
                    if (c0 + 16 + t < D) w1[t] = a.Wo[cls * D + c0 + 16 + t];
                }
            }
            bf16x4 h0, l0, h1, l1;
            split_bf16(w0, h0, l0);
            split_bf16(w1, h1, l1);
            sm_wo[0][e] = cat8(h0, h1);
            sm_wo[1][e] = cat8(l0, l1);
        }
        if (threadIdx.x < 128) sm_bo[threadIdx.x] = threadIdx.x < a.Co ? a.bo[threadIdx.x] : 0.f;
    }
    __syncthreads();
    const float cd = sm_cd;
    const float inv_d = 1.0f / static_cast<float>(D);

    f32x4 gacc[NEXT ? 10 : 1];          // NEXT: upper half of out^T out (see gram_kernel)
    f32x4 gsx = zero4();
#pragma unroll
    for (int i = 0; i < (NEXT ? 10 : 1); ++i) gacc[i] = zero4();

    auto body = [&](int64_t tile, auto guard_tag) {
        constexpr bool G = decltype(guard_tag)::value;
        const int64_t row = tile * 16 + l15;
        const bool row_ok = !G || row < a.n_rows;
        GatherCursor gc;
        if (GATHER) gc = gather_begin(a.g_rowptr, a.g_src, a.g_val, row, row_ok);
        f32x4 xa[4];
        load_rows<G>(xa, a.x, a.ldx, row, a.n_rows, lg, C);
        // denominator of this lane's row: x.u + cd, folded over the four lane groups
        float den = 0.f;
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
            const f32x4 uu = *reinterpret_cast<const f32x4*>(&sm_u[16 * cq + 4 * lg]);
#pragma unroll
            for (int t = 0; t < 4; ++t) den += xa[cq][t] * uu[t];
        }
        den += __shfl_xor(den, 16, 64);
        den += __shfl_xor(den, 32, 64);
        const float rden = 1.0f / (den + cd);
        f32x4 y[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) y[ft] = *reinterpret_cast<const f32x4*>(&sm_cn[16 * ft + 4 * lg]);
        if constexpr (SPLIT) project_split(y, xa, reinterpret_cast<const bf16x8*>(&sm_w[0][0]), reinterpret_cast<const bf16x8*>(&sm_w[0][0]) + 512, lane);
        else project_t(y, xa, sm_w[0], l15, lg);
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) y[ft] *= rden;
        if (GATHER || a.ax) {
            f32x4 ga[4];
            float rsv = 0.f;
            if (GATHER) {
                gather_rows<EXACT, T>(ga, rsv, gc, a.x, a.ldx, a.g_src, a.g_val, lg, C);
#pragma unroll
                for (int cq = 0; cq < 4; ++cq) ga[cq] *= a.gcn_scale;          // ax = g_s A_hat x; the bias term scales below
            } else {
                load_rows<G>(ga, a.ax, a.ldax, row, a.n_rows, lg, C);
                rsv = (a.rs && row_ok) ? a.rs[row] : 0.f;
            }
            if (GRAPH_W) {        // the second product accumulates on top of the attention term
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) y[ft] += *reinterpret_cast<const f32x4*>(&sm_bv[16 * ft + 4 * lg]) * rsv;
                if constexpr (SPLIT) project_split(y, ga, reinterpret_cast<const bf16x8*>(&sm_w[0][0]) + 1024, reinterpret_cast<const bf16x8*>(&sm_w[0][0]) + 1536, lane);
                else project_t(y, ga, sm_w[1], l15, lg);
            } else {          // use_weight = False: the aggregated rows are the graph term (C == D), same layout
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) y[ft] += ga[ft];
            }
        }
        if (a.x0) {
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                const int f = 16 * ft + 4 * lg;
                if (row_ok && (EXACT || f < D)) {
                    if (EXACT || ((a.ldx0 & 3) == 0 && f + 3 < D)) y[ft] += Elem<T>::ld4g(a.x0 + row * a.ldx0 + f);
                    else
                        for (int r = 0; r < 4; ++r) if (f + r < D) y[ft][r] += Elem<T>::ld(a.x0 + row * a.ldx0 + f + r);
                }
            }
        }
        if (a.residual) {
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) y[ft] = a.alpha * y[ft] + (1.0f - a.alpha) * xa[ft];
        }
        if (a.ln_w) {
            float mu = 0.f;
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
#pragma unroll
                for (int r = 0; r < 4; ++r) mu += (EXACT || 16 * ft + 4 * lg + r < D) ? y[ft][r] : 0.f;
            mu += __shfl_xor(mu, 16, 64);
            mu += __shfl_xor(mu, 32, 64);
            mu *= inv_d;
            float var = 0.f;
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dz = (EXACT || 16 * ft + 4 * lg + r < D) ? y[ft][r] - mu : 0.f;
                    y[ft][r] = dz;
                    var += dz * dz;
                }
            var += __shfl_xor(var, 16, 64);
            var += __shfl_xor(var, 32, 64);
            const float rstd = 1.0f / sqrtf(var * inv_d + a.eps);
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
                y[ft] = y[ft] * rstd * *reinterpret_cast<const f32x4*>(&sm_lw[16 * ft + 4 * lg]) +
                        *reinterpret_cast<const f32x4*>(&sm_lb[16 * ft + 4 * lg]);
        }
        float dscale = 0.f;
        if (!NEXT && a.ys_next && row_ok) dscale = dinv_of(a.rowptr, row);
        if constexpr (HEAD) {
            // logits^T = Wo out^T: the finished row piece has the layout of a loaded x fragment (features 16ft + 4lg .. + 3
            // of row r0 + l15), so it is the B operand of the same transposed product; two blocks of 64 classes
            f32x4 yo[4];
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                yo[ft] = y[ft];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (a.relu) yo[ft][r] = fmaxf(yo[ft][r], 0.f);
                    if (!EXACT && 16 * ft + 4 * lg + r >= D) yo[ft][r] = 0.f;
                }
            }
            const int nblk = (a.Co + 63) >> 6;
            bf16x8 oh[2], ol[2];                                       // the row piece, split once for both class blocks
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                bf16x4 h0, l0, h1, l1;
                split_bf16(yo[2 * kb], h0, l0);
                split_bf16(yo[2 * kb + 1], h1, l1);
                oh[kb] = cat8(h0, h1);
                ol[kb] = cat8(l0, l1);
            }
            for (int blk = 0; blk < nblk; ++blk) {
                f32x4 z[4];
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) z[ft] = *reinterpret_cast<const f32x4*>(&sm_bo[64 * blk + 16 * ft + 4 * lg]);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int ft = 0; ft < 4; ++ft) {
                        const int fr = (((blk * 4 + ft) * 2 + kb) << 6) + lane;
                        const bf16x8 wh = sm_wo[0][fr], wl = sm_wo[1][fr];
                        z[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ol[kb], z[ft], 0, 0, 0);      // small terms first
                        z[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, oh[kb], z[ft], 0, 0, 0);
                        z[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, oh[kb], z[ft], 0, 0, 0);
                    }
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) {
                    const int cls = 64 * blk + 16 * ft + 4 * lg;
                    if (row_ok && cls < a.Co) {
                        if ((a.ldl & 3) == 0 && cls + 3 < a.Co) Elem<T>::st4(a.logits + row * a.ldl + cls, z[ft]);
                        else
                            for (int r = 0; r < 4; ++r) if (cls + r < a.Co) Elem<T>::st(a.logits + row * a.ldl + cls + r, z[ft][r]);
                    }
                }
            }
        }
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const int f = 16 * ft + 4 * lg;
            f32x4 v = y[ft];
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (a.out && row_ok && (EXACT || f < D)) {
                if (EXACT || ((a.ldo & 3) == 0 && f + 3 < D)) Elem<T>::st4(a.out + row * a.ldo + f, v);
                else
                    for (int r = 0; r < 4; ++r) if (f + r < D) Elem<T>::st(a.out + row * a.ldo + f + r, v[r]);
            }
            if (!NEXT && a.ys_next) {
                // slice-major pre-scaled copy for the next layer's SpMM straight from the registers: this lane holds
                // slice 4ft + lg of its row, the 16 lanes of a group 16 consecutive rows -> 256 contiguous bytes
                if (row_ok && (EXACT || f < D)) a.ys_next[static_cast<int64_t>(4 * ft + lg) * a.npad + row] = v * dscale;
            }
            if (NEXT) {         // park the finished row piece (zero outside the matrix) for the row-contracting re-read
                if (!(row_ok && (EXACT || f < D))) v = zero4();
                else if (!EXACT)
                    for (int r = 0; r < 4; ++r) if (f + r >= D) v[r] = 0.f;
                *reinterpret_cast<f32x4*>(&sm_t[wave][l15 * kWStride + f]) = v;
            }
        }
        if (NEXT) {
            // the tile again, four whole rows per read (row 4u + lg, columns 4*l15..+3): operand of the Gram product
            // contracting over rows, and one 16-byte slice of the row for the slice-major copy
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(&sm_t[wave][(4 * u + lg) * kWStride + 4 * l15]);
                const int64_t r2 = tile * 16 + 4 * u + lg;
                if (a.ys_next && (!G || r2 < a.n_rows) && 4 * l15 < D) a.ys_next[static_cast<int64_t>(l15) * a.npad + r2] = xv * dinv_of(a.rowptr, r2);
                gsx += xv;
                int i = 0;
#pragma unroll
                for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                    for (int tb = ta; tb < 4; ++tb, ++i)
                        gacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[ta], xv[tb], gacc[i], 0, 0, 0);
            }
        }
    };
    int64_t tile = first;
    if (EXACT) {
        for (; tile < n_fast; tile += stride) {
            asm volatile("" ::: "memory");
            body(tile, std::false_type{});
        }
    }
    for (; tile < n_tiles; tile += stride) {
        asm volatile("" ::: "memory");
        body(tile, std::true_type{});
    }
    if (!NEXT && a.ys_next) {
        const int64_t pad = a.npad - a.n_rows;          // rows of the copy past the matrix are read by the last source tile
        for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < pad * 16;
             i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
            const int64_t sl = i / pad, r = a.n_rows + i % pad;
            if (4 * sl < D) a.ys_next[sl * a.npad + r] = zero4();
        }
    }
    if (NEXT) {
        const int64_t pad = a.ys_next ? a.npad - a.n_rows : 0;   // rows of the copy past the matrix are read by the last source tile
        for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < pad * 16;
             i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
            const int64_t sl = i / pad, r = a.n_rows + i % pad;
            if (4 * sl < D) a.ys_next[sl * a.npad + r] = zero4();
        }
        // fold the four waves' Gram partials through the weight region (no longer needed): (w0 + w2) + (w1 + w3)
        __syncthreads();
        float* buf = &sm_w[0][0];                       // 2 x 40 x 64 floats needed, 2 x 64 x 68 there (NEXT keeps the padded size)
        if (wave >= 2) {
#pragma unroll
            for (int i = 0; i < 10; ++i)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) buf[((wave - 2) * 40 + i * 4 + reg) * 64 + lane] = gacc[i][reg];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v = gsx[t];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lg == 0) sm_s[wave][4 * l15 + t] = v;
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
            for (int i = 0; i < 10; ++i)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) gacc[i][reg] += buf[(wave * 40 + i * 4 + reg) * 64 + lane];
        }
        __syncthreads();
        if (wave == 1) {
#pragma unroll
            for (int i = 0; i < 10; ++i)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) buf[(i * 4 + reg) * 64 + lane] = gacc[i][reg];
        }
        __syncthreads();
        if (wave == 0) {
            float* rec = a.ws + static_cast<int64_t>(blockIdx.x) * a.ws_stride;
            int i = 0;
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = ta; tb < 4; ++tb, ++i)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const float v = gacc[i][reg] + buf[(i * 4 + reg) * 64 + lane];
                        const int gi = 4 * (4 * lg + reg) + ta, gj = 4 * l15 + tb;
                        if (gi < D && gj < D) {
                            rec[gi * D + gj] = v;
                            if (ta != tb) rec[gj * D + gi] = v;
                        }
                    }
            if (lane < D) rec[D * D + lane] = ((sm_s[0][lane] + sm_s[1][lane]) + sm_s[2][lane]) + sm_s[3][lane];
        }
    }
}

}  // namespace

extern "C" size_t dif_gram_workspace_bytes(int64_t n_rows, int C) {
    if (n_rows <= 0 || C <= 0 || C > 64) return 0;
    const size_t rec = (static_cast<size_t>(C) * C + C + 3) & ~size_t(3);
    return rec * sizeof(float) * static_cast<size_t>(row_chunks(n_rows, kRecordChunksPerCU));
}

// record = [G: C x C][sx: C][2 unused floats]; ys / rowptr / plan may be null (no slice-major copy).
namespace {
template <typename T>
int gram_entry(const T* x, int64_t ldx, int64_t n_rows, int C, const int32_t* rowptr, const int32_t* plan, float* ys,
               float* record, void* workspace, size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(x && record && workspace && n_rows > 0, DIF_E_BADARG, "dif_gram: null pointer or no rows");
    DIF_REQUIRE(C > 0 && C <= 64 && C % 4 == 0, DIF_E_SHAPE, "dif_gram: covers C <= 64, C %% 4 == 0 (got %d)", C);
    DIF_REQUIRE(ldx >= C && ldx % 4 == 0 && dif::aligned_v4<T>(x), DIF_E_BADARG, "dif_gram: rows of x must be aligned to 4 elements");
    DIF_REQUIRE(workspace_bytes >= dif_gram_workspace_bytes(n_rows, C), DIF_E_WORKSPACE, "dif_gram: workspace too small");
    DIF_REQUIRE((ys == nullptr) || (rowptr && plan && dif::aligned16(ys) && std::is_same<T, float>::value), DIF_E_BADARG,
                "dif_gram: the slice-major copy needs float32 rows, rowptr, the plan and a 16-byte aligned buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int P = gram_chunks(n_rows);
    const int64_t rec = (static_cast<int64_t>(C) * C + C + 3) & ~int64_t(3);
    float* ws = static_cast<float*>(workspace);
    if (ys) {
        if constexpr (std::is_same<T, float>::value) {        // (the slice-major copy is float32-only: no bfloat16 instantiation)
            const int64_t npad = static_cast<int64_t>(plan[6]) * plan[7];
            DIF_REQUIRE(plan[0] == C / 4 && npad >= n_rows, DIF_E_BADARG, "dif_gram: plan does not match C / n_rows");
            hipLaunchKernelGGL((gram_kernel<true, T>), dim3(P), dim3(64 * kGramWaves), 0, st, x, ldx, n_rows, C, rowptr,
                               reinterpret_cast<f32x4*>(ys), npad, ws, rec);
        }
    } else {
        hipLaunchKernelGGL((gram_kernel<false, T>), dim3(P), dim3(64 * kGramWaves), 0, st, x, ldx, n_rows, C, nullptr, nullptr,
                           int64_t(0), ws, rec);
    }
    if (int rc = dif::launch_status("gram_kernel")) return rc;
    return dif::launch_record_finalize(ws, P, rec, C * C + C, 0, record, st);
}
}  // namespace

extern "C" int dif_gram_f32(const float* x, int64_t ldx, int64_t n_rows, int C, const int32_t* rowptr, const int32_t* plan,
                            float* ys, float* record, void* workspace, size_t workspace_bytes, dif_stream_t stream) {
    return gram_entry<float>(x, ldx, n_rows, C, rowptr, plan, ys, record, workspace, workspace_bytes, stream);
}

// Input layer (Linear -> LayerNorm -> ReLU, C_in <= 64 -> D <= 64, D % 4 == 0) + Gram record of its OUTPUT + slice-major copy
// (ys / rowptr / plan nullable together) in one pass; out [n_rows, D] row-major.  workspace: dif_gram_workspace_bytes(n_rows, D).
extern "C" int dif_input_gram_f32(const float* x, int64_t ldx, int64_t n_rows, int C_in, const float* W, const float* bias, int D,
                                  const float* ln_weight, const float* ln_bias, float ln_eps, int relu, float* out, int64_t ldo,
                                  const int32_t* rowptr, const int32_t* plan, float* ys, float* record, void* workspace,
                                  size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(x && W && bias && out && record && workspace && n_rows > 0, DIF_E_BADARG, "dif_input_gram: null pointer or no rows");
    DIF_REQUIRE(C_in > 0 && C_in <= 64 && D > 0 && D <= 64 && D % 4 == 0, DIF_E_SHAPE,
                "dif_input_gram: covers C_in <= 64 -> D <= 64, D %% 4 == 0 (got %d -> %d)", C_in, D);
    DIF_REQUIRE((ln_weight == nullptr) == (ln_bias == nullptr), DIF_E_BADARG, "dif_input_gram: ln_weight and ln_bias must be given together");
    DIF_REQUIRE(ldx >= C_in && ldo >= D && ldo % 4 == 0 && dif::aligned16(out), DIF_E_BADARG,
                "dif_input_gram: rows of out must be 16-byte aligned, leading dimensions >= the row lengths");
    DIF_REQUIRE(workspace_bytes >= dif_gram_workspace_bytes(n_rows, D), DIF_E_WORKSPACE, "dif_input_gram: workspace too small");
    DIF_REQUIRE((ys == nullptr) || (rowptr && plan && dif::aligned16(ys)), DIF_E_BADARG,
                "dif_input_gram: the slice-major copy needs rowptr, the plan and a 16-byte aligned buffer");
    int64_t npad = 0;
    if (ys) {
        npad = static_cast<int64_t>(plan[6]) * plan[7];
        DIF_REQUIRE(plan[0] == D / 4 && npad >= n_rows, DIF_E_BADARG, "dif_input_gram: plan does not match D / n_rows");
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int P = input_gram_chunks(n_rows);
    const int64_t rec = (static_cast<int64_t>(D) * D + D + 3) & ~int64_t(3);
    float* ws = static_cast<float*>(workspace);
    const int xvec = (C_in % 4 == 0) && (ldx % 4 == 0) && dif::aligned16(x);
    const int kq = (C_in + 15) / 16;
#define DIF_IG(KQ) hipLaunchKernelGGL((input_gram_kernel<KQ>), dim3(P), dim3(64 * kGramWaves), 0, st, x, ldx, n_rows, C_in, W, bias, D, \
                                      ln_weight, ln_bias, ln_eps, relu, out, ldo, rowptr, reinterpret_cast<f32x4*>(ys), npad, ws, rec, xvec)
    if (kq == 1) DIF_IG(1); else if (kq == 2) DIF_IG(2); else if (kq == 3) DIF_IG(3); else DIF_IG(4);
#undef DIF_IG
    if (int rc = dif::launch_status("input_gram_kernel")) return rc;
    return dif::launch_record_finalize(ws, P, rec, D * D + D, 0, record, st);
}

// bfloat16 rows (BASELINE config C5), float32 record; no slice-major copy (the sliced product is float32-only)
extern "C" int dif_gram_bf16(const void* x, int64_t ldx, int64_t n_rows, int C, float* record, void* workspace,
                             size_t workspace_bytes, dif_stream_t stream) {
    return gram_entry<dif::bf16>(static_cast<const dif::bf16*>(x), ldx, n_rows, C, nullptr, nullptr, nullptr, record, workspace,
                                 workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------------------
// Training through the record: backward of the attention term  att = (x Mn + cn) / (x u + cd)  of the layer kernel, given
// d = dL/d(att) [n, D] (difformer_amd/autograd_ops.py, _ClosedFormLayer).  Per 16-row tile and wave, in one pass:
//   att again (the first product of the layer kernel),  d_num = d / den,  d_den = -<d_num, att>,
//   dx = dx_in + d_num Mn^T + d_den u^T                 (second product: the same weights, read transposed)
// d_num and d_den go on to one streaming reduce (d Mn = x^T d_num, d cn, d u, d cd).  Replaces a recompute launch, four
// element-wise passes, a vendor GEMV and a row GEMM of the tensor-op formulation (0.21 ms per layer at 132,534 rows).
// ------------------------------------------------------------------------------------------------------------
namespace {
struct AttnBwdArgs {
    const float* x; int64_t ldx;
    const float* coef;
    const float* d; int64_t ldd;
    const float* dx_in; int64_t ldi;
    float* d_num; float* d_den; float* dx; int64_t ldo;
    int64_t n_rows; int C, D;
    const float* rs;            // SUMS: row sums of the adjacency (nullable)
    float* sums;                // SUMS: [workgroups][kAttnBwdSums] partial column sums
};
constexpr int kAttnBwdSums = 132;       // [sum_r d_den[r] x[r, :]: 64][sum_r rs[r] d[r, :]: 64][sum_r d_den[r]][pad: 3]

// SUMS: the three row-contracted sums the backward needs besides the streaming reduce (d u = x^T d_den, d cd = sum d_den,
// and the bias gradient of the graph branch, rs^T d) are folded over the rows this workgroup sees -- per-lane accumulators,
// one butterfly over the 16 rows of the lane groups at the end, the waves' sums through LDS -- and left as one partial
// record per workgroup (summed by the caller): two streaming reduces + two finalize launches fewer per layer.
template <bool EXACT, bool SUMS>
__global__ __launch_bounds__(64 * kWaves, SUMS ? 3 : 4) void closed_form_attn_bwd_kernel(AttnBwdArgs a) {
    __shared__ float sm_part[SUMS ? kWaves : 1][SUMS ? kAttnBwdSums : 1];
    __shared__ __attribute__((aligned(16))) float sm_w[2][kWBlock];   // MnT as W[f][c], and its transpose W'[c][f]
    __shared__ __attribute__((aligned(16))) float sm_cn[64], sm_u[64];
    __shared__ float sm_cd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
    const int C = a.C, D = a.D;
    for (int e = threadIdx.x; e < 64 * 64; e += 64 * kWaves) {          // e = LDS dword of block 0: [ft][cq][lg][l15][t]
        const int f = 16 * (e >> 10) + ((e >> 2) & 15), c = 16 * ((e >> 8) & 3) + 4 * ((e >> 6) & 3) + (e & 3);
        const float v = (f < D && c < C) ? a.coef[f * C + c] : 0.f;
        sm_w[0][widx(f, c)] = v;
        sm_w[1][widx(c, f)] = v;
    }
    if (threadIdx.x < 64) {
        const int i = threadIdx.x;
        sm_cn[i] = i < D ? a.coef[D * C + i] : 0.f;
        sm_u[i] = i < C ? a.coef[D * C + D + i] : 0.f;
    }
    if (threadIdx.x == 0) sm_cd = a.coef[D * C + D + C];
    __syncthreads();
    const float cd = sm_cd;
    const int64_t n_tiles = (a.n_rows + 15) / 16;
    const int64_t n_fast = EXACT ? a.n_rows / 16 : 0;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kWaves + wave, stride = static_cast<int64_t>(gridDim.x) * kWaves;

    f32x4 au[4], ab[4];
    float acd = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) au[i] = ab[i] = zero4();
    auto body = [&](int64_t tile, auto guard_tag) {
        constexpr bool G = decltype(guard_tag)::value;
        const int64_t row = tile * 16 + l15;
        const bool row_ok = !G || row < a.n_rows;
        f32x4 xa[4], dn[4];
        load_rows<G>(xa, a.x, a.ldx, row, a.n_rows, lg, C);
        load_rows<G>(dn, a.d, a.ldd, row, a.n_rows, lg, D);
        if (SUMS && a.rs) {
            const float rsv = row_ok ? a.rs[row] : 0.f;
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) ab[ft] += dn[ft] * rsv;
        }
        float den = 0.f;
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
            const f32x4 uu = *reinterpret_cast<const f32x4*>(&sm_u[16 * cq + 4 * lg]);
#pragma unroll
            for (int t = 0; t < 4; ++t) den += xa[cq][t] * uu[t];
        }
        den += __shfl_xor(den, 16, 64);
        den += __shfl_xor(den, 32, 64);
        const float rden = 1.0f / (den + cd);
        f32x4 y[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) y[ft] = *reinterpret_cast<const f32x4*>(&sm_cn[16 * ft + 4 * lg]);
        project_t(y, xa, sm_w[0], l15, lg);                                        // numerator of att
        float dd = 0.f;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            dn[ft] *= rden;
#pragma unroll
            for (int r = 0; r < 4; ++r) dd += dn[ft][r] * y[ft][r];
        }
        dd += __shfl_xor(dd, 16, 64);
        dd += __shfl_xor(dd, 32, 64);
        dd *= -rden;                                                               // -<d_num, att>
        if (row_ok && lg == 0) a.d_den[row] = dd;
        if (SUMS) {                                   // rows past the end contribute zeros (their d and x were loaded as 0)
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) au[cq] += xa[cq] * dd;
            acd += (lg == 0) ? dd : 0.f;
        }
        f32x4 z[4];
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
            const int c = 16 * cq + 4 * lg;
            z[cq] = *reinterpret_cast<const f32x4*>(&sm_u[c]) * dd;
            if (a.dx_in && row_ok && (EXACT || c < C)) z[cq] += *reinterpret_cast<const f32x4*>(a.dx_in + row * a.ldi + c);
        }
        project_t(z, dn, sm_w[1], l15, lg);                                        // + d_num Mn^T
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = 16 * q + 4 * lg;
            if (row_ok && (EXACT || f < D)) *reinterpret_cast<f32x4*>(a.d_num + row * D + f) = dn[q];
            if (row_ok && (EXACT || f < C)) *reinterpret_cast<f32x4*>(a.dx + row * a.ldo + f) = z[q];
        }
    };
    int64_t tile = first;
    if (EXACT) {
        for (; tile < n_fast; tile += stride) {
            asm volatile("" ::: "memory");
            body(tile, std::false_type{});
        }
    }
    for (; tile < n_tiles; tile += stride) {
        asm volatile("" ::: "memory");
        body(tile, std::true_type{});
    }
    if constexpr (SUMS) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {            // over the 16 rows of a lane group (lanes with the same lg)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    au[i][t] += __shfl_xor(au[i][t], m, 64);
                    ab[i][t] += __shfl_xor(ab[i][t], m, 64);
                }
            acd += __shfl_xor(acd, m, 64);
        }
        if (l15 == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    sm_part[wave][16 * i + 4 * lg + t] = au[i][t];
                    sm_part[wave][64 + 16 * i + 4 * lg + t] = ab[i][t];
                }
            if (lg == 0) sm_part[wave][128] = acd;
        }
        __syncthreads();
        if (threadIdx.x < kAttnBwdSums) {
            float v = 0.f;
            if (threadIdx.x <= 128) {
#pragma unroll
                for (int w = 0; w < kWaves; ++w) v += sm_part[w][threadIdx.x];
            }
            a.sums[static_cast<int64_t>(blockIdx.x) * kAttnBwdSums + threadIdx.x] = v;
        }
    }
}
}  // namespace

extern "C" int dif_closed_form_attn_bwd_groups(int64_t n_rows) {       // partial records dif_closed_form_attn_bwd_f32 leaves in `sums`
    return n_rows > 0 ? row_chunks(n_rows, 3) : 0;
}

// d [n, D] = gradient with respect to the attention term; dx_in (nullable) [n, C] = what dx already holds.
// -> d_num [n, D] (dense), d_den [n], dx [n, C] = dx_in + d_num Mn^T + d_den u^T.  C, D multiples of 4, <= 64; 16-byte rows.
// sums (nullable) [dif_closed_form_attn_bwd_groups(n)][132]: per-workgroup partial sums [x^T d_den: 64][row_sums^T d: 64][sum d_den]
// (row_sums nullable: that block stays zero); the caller adds the records up.
extern "C" int dif_closed_form_attn_bwd_f32(const float* x, int64_t ldx, int64_t n_rows, int C, int D, const float* coef,
                                            const float* d, int64_t ldd, const float* dx_in, int64_t ldi, float* d_num,
                                            float* d_den, float* dx, int64_t ldo, const float* row_sums, float* sums,
                                            dif_stream_t stream) {
    DIF_REQUIRE(x && coef && d && d_num && d_den && dx && n_rows > 0, DIF_E_BADARG, "dif_closed_form_attn_bwd: null pointer or no rows");
    DIF_REQUIRE(C > 0 && C <= 64 && D > 0 && D <= 64 && C % 4 == 0 && D % 4 == 0, DIF_E_SHAPE,
                "dif_closed_form_attn_bwd: covers C, D <= 64, multiples of 4 (got %d, %d)", C, D);
    DIF_REQUIRE(ldx >= C && ldd >= D && ldo >= C && (!dx_in || ldi >= C) && ldx % 4 == 0 && ldd % 4 == 0 && ldo % 4 == 0 &&
                (!dx_in || ldi % 4 == 0), DIF_E_BADARG, "dif_closed_form_attn_bwd: leading dimensions must cover a row and be multiples of 4");
    DIF_REQUIRE(dif::aligned16(x) && dif::aligned16(d) && dif::aligned16(d_num) && dif::aligned16(dx) && (!dx_in || dif::aligned16(dx_in)),
                DIF_E_BADARG, "dif_closed_form_attn_bwd: rows must be 16-byte aligned");
    DIF_REQUIRE(sums || !row_sums, DIF_E_BADARG, "dif_closed_form_attn_bwd: row_sums without sums");
    AttnBwdArgs a = {x, ldx, coef, d, ldd, dx_in, ldi, d_num, d_den, dx, ldo, n_rows, C, D, row_sums, sums};
    const int P = row_chunks(n_rows, sums ? 3 : 4);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool exact = C == 64 && D == 64;
    if (sums) {
        if (exact) hipLaunchKernelGGL((closed_form_attn_bwd_kernel<true, true>), dim3(P), dim3(64 * kWaves), 0, st, a);
        else hipLaunchKernelGGL((closed_form_attn_bwd_kernel<false, true>), dim3(P), dim3(64 * kWaves), 0, st, a);
    } else {
        if (exact) hipLaunchKernelGGL((closed_form_attn_bwd_kernel<true, false>), dim3(P), dim3(64 * kWaves), 0, st, a);
        else hipLaunchKernelGGL((closed_form_attn_bwd_kernel<false, false>), dim3(P), dim3(64 * kWaves), 0, st, a);
    }
    return dif::launch_status("closed_form_attn_bwd_kernel");
}

extern "C" size_t dif_simple_coeffs_len(int C, int D) {
    if (C <= 0 || D <= 0) return 0;
    return static_cast<size_t>(D) * C + D + C + 4;
}

extern "C" int dif_simple_coeffs_f32(const float* record, int64_t n_global, int C, int D, const float* Wq, const float* bq,
                                     const float* Wk, const float* bk, const float* Wv, const float* bv, float attn_scale,
                                     float* coef, dif_stream_t stream) {
    DIF_REQUIRE(record && Wq && bq && Wk && bk && coef && n_global > 0, DIF_E_BADARG, "dif_simple_coeffs: null pointer");
    DIF_REQUIRE(C > 0 && C <= 64 && D > 0 && D <= 64, DIF_E_SHAPE, "dif_simple_coeffs: covers C, D <= 64 (got %d, %d)", C, D);
    DIF_REQUIRE((Wv == nullptr) == (bv == nullptr), DIF_E_BADARG, "dif_simple_coeffs: Wv and bv go together");
    DIF_REQUIRE(Wv != nullptr || C == D, DIF_E_SHAPE, "dif_simple_coeffs: without a value projection C must equal D (difformer.py:120)");
    hipLaunchKernelGGL(coeffs_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), record,
                       static_cast<float>(n_global), C, D, Wq, bq, Wk, bk, Wv, bv, attn_scale, coef);
    return dif::launch_status("coeffs_kernel");
}

// dif_gram_f32 (no slice-major copy) + dif_simple_coeffs_f32 for one layer input: coef from x in two launches when the Gram
// pass leaves at most 48 partial records (<= 24,576 rows: the folded kernel above), three otherwise.  record (nullable)
// receives the Gram record.  workspace: dif_gram_workspace_bytes(n_rows, C).
extern "C" int dif_gram_coeffs_f32(const float* x, int64_t ldx, int64_t n_rows, int C, int D, const float* Wq, const float* bq,
                                   const float* Wk, const float* bk, const float* Wv, const float* bv, int64_t n_global,
                                   float attn_scale, float* coef, float* record, void* workspace, size_t workspace_bytes,
                                   dif_stream_t stream) {
    DIF_REQUIRE(x && Wq && bq && Wk && bk && coef && workspace && n_rows > 0 && n_global >= n_rows, DIF_E_BADARG,
                "dif_gram_coeffs: null pointer or no rows");
    DIF_REQUIRE(C > 0 && C <= 64 && C % 4 == 0 && D > 0 && D <= 64, DIF_E_SHAPE,
                "dif_gram_coeffs: covers C <= 64 (C %% 4 == 0), D <= 64 (got %d, %d)", C, D);
    DIF_REQUIRE((Wv == nullptr) == (bv == nullptr), DIF_E_BADARG, "dif_gram_coeffs: Wv and bv go together");
    DIF_REQUIRE(Wv != nullptr || C == D, DIF_E_SHAPE, "dif_gram_coeffs: without a value projection C must equal D (difformer.py:120)");
    DIF_REQUIRE(ldx >= C && ldx % 4 == 0 && dif::aligned16(x), DIF_E_BADARG, "dif_gram_coeffs: rows of x must be 16-byte aligned");
    DIF_REQUIRE(workspace_bytes >= dif_gram_workspace_bytes(n_rows, C), DIF_E_WORKSPACE, "dif_gram_coeffs: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int P = gram_chunks(n_rows);
    const int64_t rec = (static_cast<int64_t>(C) * C + C + 3) & ~int64_t(3);
    float* ws = static_cast<float*>(workspace);
    DIF_REQUIRE(P <= kFoldedPartsMax || record, DIF_E_BADARG, "dif_gram_coeffs: more than 48 partial records need `record` as the finalize target");
    hipLaunchKernelGGL((gram_kernel<false, float>), dim3(P), dim3(64 * kGramWaves), 0, st, x, ldx, n_rows, C, nullptr, nullptr,
                       int64_t(0), ws, rec);
    if (int rc = dif::launch_status("gram_kernel")) return rc;
    if (P <= kFoldedPartsMax) {
        hipLaunchKernelGGL(coeffs_parts_kernel, dim3(1), dim3(1024), 0, st, ws, P, rec, static_cast<float>(n_global), C, D, Wq, bq, Wk,
                           bk, Wv, bv, attn_scale, coef, record);
        return dif::launch_status("coeffs_parts_kernel");
    }
    if (int rc = dif::launch_record_finalize(ws, P, rec, C * C + C, 0, record, st)) return rc;
    hipLaunchKernelGGL(coeffs_kernel, dim3(1), dim3(1024), 0, st, record, static_cast<float>(n_global), C, D, Wq, bq, Wk, bk, Wv, bv,
                       attn_scale, coef);
    return dif::launch_status("coeffs_kernel");
}

namespace {
template <typename T>
int layer_entry(const T* x, int64_t ldx, int64_t n_rows, int C, int D, const float* coef, const T* ax, int64_t ldax,
                const float* Wv, const float* bv, const float* row_sums, float gcn_scale, const T* x0, int64_t ldx0,
                int residual, float alpha, const float* ln_weight, const float* ln_bias, float ln_eps, int relu, T* out,
                int64_t ldo, float* next_record, const int32_t* rowptr, const int32_t* plan, float* next_ys, void* workspace,
                size_t workspace_bytes, dif_stream_t stream, const float* Wo = nullptr, const float* bo = nullptr, int Co = 0,
                T* logits = nullptr, int64_t ldl = 0, const int32_t* g_rowptr = nullptr, const int32_t* g_src = nullptr,
                const float* g_val = nullptr) {
    const bool head = Wo != nullptr;
    DIF_REQUIRE(x && coef && (out || head) && n_rows > 0, DIF_E_BADARG, "dif_simple_layer: null pointer or no rows");
    DIF_REQUIRE(!head || (bo && logits && Co > 0 && Co <= 128 && ldl >= Co && !next_record && !next_ys), DIF_E_BADARG,
                "dif_simple_layer: the fused output Linear needs bo, logits, 1 <= Co <= 128, ldl >= Co, no next-layer products");
    DIF_REQUIRE(C > 0 && C <= 64 && C % 4 == 0 && D > 0 && D <= 64, DIF_E_SHAPE,
                "dif_simple_layer: covers C <= 64 (C %% 4 == 0) and D <= 64 (got %d, %d)", C, D);
    DIF_REQUIRE(ldx >= C && ldx % 4 == 0 && dif::aligned_v4<T>(x) && (!out || ldo >= D), DIF_E_BADARG,
                "dif_simple_layer: rows of x must be aligned to 4 elements, ldo >= D");
    DIF_REQUIRE((!out || dif::aligned_v4<T>(out)) && (!x0 || dif::aligned_v4<T>(x0)) && (!logits || dif::aligned_v4<T>(logits)),
                DIF_E_BADARG, "dif_simple_layer: out / x0 / logits must be aligned to 4 elements");
    DIF_REQUIRE(!ax || (ldax >= C && ldax % 4 == 0 && dif::aligned_v4<T>(ax)), DIF_E_BADARG,
                "dif_simple_layer: rows of ax must be aligned to 4 elements");
    DIF_REQUIRE((ln_weight == nullptr) == (ln_bias == nullptr), DIF_E_BADARG, "dif_simple_layer: ln_weight and ln_bias go together");
    DIF_REQUIRE((Wv == nullptr) == (bv == nullptr), DIF_E_BADARG, "dif_simple_layer: Wv and bv go together");
    DIF_REQUIRE(Wv != nullptr || !ax || C == D, DIF_E_SHAPE, "dif_simple_layer: without a value projection C must equal D");
    DIF_REQUIRE(!residual || C == D, DIF_E_SHAPE, "dif_simple_layer: the residual needs C == D");
    DIF_REQUIRE(!x0 || ldx0 >= D, DIF_E_BADARG, "dif_simple_layer: ldx0 smaller than a row");
    const bool f32 = std::is_same<T, float>::value;
    DIF_REQUIRE(f32 || (!next_record && !next_ys), DIF_E_BADARG, "dif_simple_layer: products for the next layer are float32-only");
    const bool next = next_record != nullptr;          // Gram record of the output from the same pass (slower)
    const bool gather = g_rowptr != nullptr;
    const int P = head ? row_chunks(n_rows, 2, kHeadWaves) : row_chunks(n_rows, next ? kRecordChunksPerCU : (gather ? DIF_GATHER_WG : 4));
    const int64_t rec = (static_cast<int64_t>(D) * D + D + 3) & ~int64_t(3);
    int64_t npad = 0;
    if (next) {
        DIF_REQUIRE(workspace && workspace_bytes >= dif_gram_workspace_bytes(n_rows, D), DIF_E_WORKSPACE,
                    "dif_simple_layer: workspace too small for the next record (dif_gram_workspace_bytes(n_rows, D))");
    }
    if (next || next_ys) {
        DIF_REQUIRE(D % 4 == 0, DIF_E_SHAPE, "dif_simple_layer: products for the next layer need D %% 4 == 0");
        DIF_REQUIRE((next_ys == nullptr) || (rowptr && plan && dif::aligned16(next_ys)), DIF_E_BADARG,
                    "dif_simple_layer: the slice-major copy needs rowptr, the plan and a 16-byte aligned buffer");
        if (next_ys) {
            npad = static_cast<int64_t>(plan[6]) * plan[7];
            DIF_REQUIRE(plan[0] == D / 4 && npad >= n_rows, DIF_E_BADARG, "dif_simple_layer: plan does not match D / n_rows");
        }
    }
    DIF_REQUIRE(!gather || (g_src && g_val && !ax && !next && !next_ys), DIF_E_BADARG,
                "dif_simple_layer: the in-kernel aggregation takes rowptr, src AND val, no ax and no next-layer products");
    LayerArgsT<T> a = {x, ldx, ax, ldax, coef, Wv, bv, row_sums, gcn_scale, x0, ldx0, residual, alpha, ln_weight, ln_bias, ln_eps,
                       relu, out, ldo, n_rows, C, D, rowptr, reinterpret_cast<f32x4*>(next_ys), npad, static_cast<float*>(workspace), rec,
                       Wo, bo, Co, logits, ldl, g_rowptr, g_src, g_val};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool exact = C == 64 && D == 64 && ldo % 4 == 0 && (!x0 || ldx0 % 4 == 0);
    const bool gw = (ax != nullptr || gather) && Wv != nullptr;
    // dense 64 x 64 float32 layers (the headline shape): both products on split-bfloat16 operands unless DIFFORMER_EXACT_FP32=1
    const bool split = exact && f32 && !next && !gather && !dif::exact_fp32() &&
                       (reinterpret_cast<uintptr_t>(coef) & 15u) == 0 && (!gw || (reinterpret_cast<uintptr_t>(Wv) & 15u) == 0);
    if (split) {
        if constexpr (std::is_same<T, float>::value) {
            if (head) {
                if (gw) hipLaunchKernelGGL((simple_layer_kernel<true, true, false, T, true, false, true>), dim3(P), dim3(64 * kHeadWaves), 0, st, a);
                else hipLaunchKernelGGL((simple_layer_kernel<true, false, false, T, true, false, true>), dim3(P), dim3(64 * kHeadWaves), 0, st, a);
            } else {
                if (gw) hipLaunchKernelGGL((simple_layer_kernel<true, true, false, T, false, false, true>), dim3(P), dim3(64 * kWaves), 0, st, a);
                else hipLaunchKernelGGL((simple_layer_kernel<true, false, false, T, false, false, true>), dim3(P), dim3(64 * kWaves), 0, st, a);
            }
            return dif::launch_status("simple_layer_kernel");
        }
    }
#define DIF_LAYER(E, G, N) hipLaunchKernelGGL((simple_layer_kernel<E, G, N, T>), dim3(P), dim3(64 * kWaves), 0, st, a)
#define DIF_LAYER_HEAD(E, G) hipLaunchKernelGGL((simple_layer_kernel<E, G, false, T, true>), dim3(P), dim3(64 * kHeadWaves), 0, st, a)
#define DIF_LAYER_GATHER(E, G, H) hipLaunchKernelGGL((simple_layer_kernel<E, G, false, T, H, true>), dim3(P), dim3(64 * (H ? kHeadWaves : kWaves)), 0, st, a)
#define DIF_LAYER2(E, G) do { if (gather) { if (head) DIF_LAYER_GATHER(E, G, true); else DIF_LAYER_GATHER(E, G, false); } \
                              else if (head) DIF_LAYER_HEAD(E, G); else if (f32 && next) DIF_LAYER(E, G, (std::is_same<T, float>::value)); else DIF_LAYER(E, G, false); } while (0)
    if (exact) { if (gw) DIF_LAYER2(true, true); else DIF_LAYER2(true, false); }
    else { if (gw) DIF_LAYER2(false, true); else DIF_LAYER2(false, false); }
#undef DIF_LAYER2
#undef DIF_LAYER_GATHER
#undef DIF_LAYER_HEAD
#undef DIF_LAYER
    if (int rc = dif::launch_status("simple_layer_kernel")) return rc;
    if (next) return dif::launch_record_finalize(static_cast<float*>(workspace), P, rec, D * D + D, 0, next_record, st);
    return 0;
}
}  // namespace

extern "C" int dif_simple_layer_f32(const float* x, int64_t ldx, int64_t n_rows, int C, int D, const float* coef,
                                    const float* ax, int64_t ldax, const float* Wv, const float* bv, const float* row_sums,
                                    float gcn_scale, const float* x0, int64_t ldx0, int residual, float alpha,
                                    const float* ln_weight, const float* ln_bias, float ln_eps, int relu, float* out,
                                    int64_t ldo, float* next_record, const int32_t* rowptr, const int32_t* plan,
                                    float* next_ys, void* workspace, size_t workspace_bytes, dif_stream_t stream) {
    return layer_entry<float>(x, ldx, n_rows, C, D, coef, ax, ldax, Wv, bv, row_sums, gcn_scale, x0, ldx0, residual, alpha,
                              ln_weight, ln_bias, ln_eps, relu, out, ldo, next_record, rowptr, plan, next_ys, workspace,
                              workspace_bytes, stream);
}

// The LAST layer of a model with the output Linear of difformer.py:208 in the same pass: logits [n, Co] = out Wo^T + bo
// (Co <= 128; Wo [Co, D], bo [Co] float32).  out may be NULL (the finished rows are then not stored at all).
extern "C" int dif_simple_layer_head_f32(const float* x, int64_t ldx, int64_t n_rows, int C, int D, const float* coef,
                                         const float* ax, int64_t ldax, const float* Wv, const float* bv,
                                         const float* row_sums, float gcn_scale, const float* x0, int64_t ldx0, int residual,
                                         float alpha, const float* ln_weight, const float* ln_bias, float ln_eps, int relu,
                                         float* out, int64_t ldo, const float* Wo, const float* bo, int Co, float* logits,
                                         int64_t ldl, dif_stream_t stream) {
    DIF_REQUIRE(Wo != nullptr, DIF_E_BADARG, "dif_simple_layer_head_f32: Wo is null");
    return layer_entry<float>(x, ldx, n_rows, C, D, coef, ax, ldax, Wv, bv, row_sums, gcn_scale, x0, ldx0, residual, alpha,
                              ln_weight, ln_bias, ln_eps, relu, out, ldo, nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream,
                              Wo, bo, Co, logits, ldl);
}

// The layer with the AGGREGATION in the same pass (graphs with a few entries per row, single GPU): instead of ax the kernel
// takes the CSR (rowptr int32 [n_rows + 1], src int32, val float32: dif_csr_build's, n_blocks = 1) and gathers the rows of
// x itself -- x holds all n_rows nodes.  Wo != NULL: also the output Linear (logits [n, Co]); then `out` may be NULL.
extern "C" int dif_simple_layer_gather_f32(const float* x, int64_t ldx, int64_t n_rows, int C, int D, const float* coef,
                                           const int32_t* rowptr, const int32_t* src, const float* val, const float* Wv,
                                           const float* bv, float gcn_scale, const float* x0, int64_t ldx0, int residual,
                                           float alpha, const float* ln_weight, const float* ln_bias, float ln_eps, int relu,
                                           float* out, int64_t ldo, const float* Wo, const float* bo, int Co, float* logits,
                                           int64_t ldl, dif_stream_t stream) {
    DIF_REQUIRE(rowptr && src && val, DIF_E_BADARG, "dif_simple_layer_gather_f32: rowptr / src / val are null");
    return layer_entry<float>(x, ldx, n_rows, C, D, coef, nullptr, 0, Wv, bv, nullptr, gcn_scale, x0, ldx0, residual, alpha,
                              ln_weight, ln_bias, ln_eps, relu, out, ldo, nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream,
                              Wo, bo, Co, logits, ldl, rowptr, src, val);
}

extern "C" int dif_simple_layer_gather_bf16(const void* x, int64_t ldx, int64_t n_rows, int C, int D, const float* coef,
                                            const int32_t* rowptr, const int32_t* src, const float* val, const float* Wv,
                                            const float* bv, float gcn_scale, const void* x0, int64_t ldx0, int residual,
                                            float alpha, const float* ln_weight, const float* ln_bias, float ln_eps, int relu,
                                            void* out, int64_t ldo, const float* Wo, const float* bo, int Co, void* logits,
                                            int64_t ldl, dif_stream_t stream) {
    using B = dif::bf16;
    DIF_REQUIRE(rowptr && src && val, DIF_E_BADARG, "dif_simple_layer_gather_bf16: rowptr / src / val are null");
    return layer_entry<B>(static_cast<const B*>(x), ldx, n_rows, C, D, coef, nullptr, 0, Wv, bv, nullptr, gcn_scale,
                          static_cast<const B*>(x0), ldx0, residual, alpha, ln_weight, ln_bias, ln_eps, relu, static_cast<B*>(out),
                          ldo, nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream, Wo, bo, Co, static_cast<B*>(logits), ldl,
                          rowptr, src, val);
}

// bfloat16 activations, output Linear in the same pass: logits [n, Co] bfloat16 (Wo, bo float32 copies of the parameters)
extern "C" int dif_simple_layer_head_bf16(const void* x, int64_t ldx, int64_t n_rows, int C, int D, const float* coef,
                                          const void* ax, int64_t ldax, const float* Wv, const float* bv,
                                          const float* row_sums, float gcn_scale, const void* x0, int64_t ldx0, int residual,
                                          float alpha, const float* ln_weight, const float* ln_bias, float ln_eps, int relu,
                                          void* out, int64_t ldo, const float* Wo, const float* bo, int Co, void* logits,
                                          int64_t ldl, dif_stream_t stream) {
    using B = dif::bf16;
    DIF_REQUIRE(Wo != nullptr, DIF_E_BADARG, "dif_simple_layer_head_bf16: Wo is null");
    return layer_entry<B>(static_cast<const B*>(x), ldx, n_rows, C, D, coef, static_cast<const B*>(ax), ldax, Wv, bv, row_sums,
                          gcn_scale, static_cast<const B*>(x0), ldx0, residual, alpha, ln_weight, ln_bias, ln_eps, relu,
                          static_cast<B*>(out), ldo, nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream, Wo, bo, Co,
                          static_cast<B*>(logits), ldl);
}

// bfloat16 ACTIVATIONS (x, ax, x0, out); coefficients and parameters (Wv, bv, LayerNorm) float32 -- the host keeps exact
// float32 copies of the bfloat16 parameters.  All arithmetic float32 (BASELINE config C5).
extern "C" int dif_simple_layer_bf16(const void* x, int64_t ldx, int64_t n_rows, int C, int D, const float* coef, const void* ax,
                                     int64_t ldax, const float* Wv, const float* bv, const float* row_sums, float gcn_scale,
                                     const void* x0, int64_t ldx0, int residual, float alpha, const float* ln_weight,
                                     const float* ln_bias, float ln_eps, int relu, void* out, int64_t ldo, dif_stream_t stream) {
    using B = dif::bf16;
    return layer_entry<B>(static_cast<const B*>(x), ldx, n_rows, C, D, coef, static_cast<const B*>(ax), ldax, Wv, bv, row_sums,
                          gcn_scale, static_cast<const B*>(x0), ldx0, residual, alpha, ln_weight, ln_bias, ln_eps, relu,
                          static_cast<B*>(out), ldo, nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

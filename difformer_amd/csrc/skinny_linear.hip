// a5 ends: the input MLP (node classification/difformer.py:188-191  Linear -> LayerNorm -> ReLU) and the
// output Linear (:208) for the narrow shapes DIFFormer uses (C_in <= 128: ogbn-proteins 8->64, hidden->classes
// 64->112; long_linear_kernel below: long rows into a narrow layer, e.g. 512 -> 64).  One pass: x read once, y written once, LayerNorm/ReLU applied in registers.  Vendor GEMMs are
// tuned for large K and spend ~100 us on these (K = 8: 4 MB in, 34 MB out).
//
// MFMA plan (v_mfma_f32_16x16x4_f32, exact fp32), per wave and 16-row tile.  A workgroup keeps up to 256 output
// features of W in LDS (grid.y covers wider layers) and sweeps them in blocks of 64 per row tile, so x is read once.
// Inside a block the four 16x16 output tiles INTERLEAVE the features (tile ft owns features 4j + ft): a lane holds 4
// consecutive features of a row across its four accumulators and 16 lanes store one whole 256-B row.
//   D[i <-> row][j <-> feature fb + 4j + ft] = sum_c X[row][c] W[feature][c]
//   A[i=lane%16][k=lane/16] = X[r0 + lane%16][16cq + 4*(lane/16) + t]                 (dwordx4 loads, next tile prefetched)
//   B[k=lane/16][j=lane%16] = W[fb + 4*(lane%16) + ft][16cq + 4*(lane/16) + t]        (LDS row 16 ft + lane%16 of the block)
//   lane then holds out[r0 + 4*(lane/16) + reg][fb + 4*(lane%16) + ft]: one 16-byte store per reg.
// HBM-bound: (C_in + C_out) * 4 bytes per row.
#include <stdlib.h>
#include <type_traits>

#include "dif_common.h"

namespace {

using dif::f32x4;
using dif::Elem;

// LDS weight rows hold 64 input channels (+4 floats: b128 reads of 16 rows hit all banks) for C_in <= 64, 128 (+4) for
// C_in <= 128 (Pokec's 65 features); a workgroup keeps 4 resp. 2 blocks of 64 output features.
constexpr int lin_stride(int kq) { return kq <= 4 ? 68 : 132; }
constexpr int lin_channels(int kq) { return kq <= 4 ? 64 : 128; }
constexpr int lin_max_blocks(int kq) { return kq <= 4 ? 4 : 2; }

// Epilogue of one 16-row x 64-feature tile: lane holds y[ft][reg] = out[r0 + 4 lg + reg][fb + 4 l15 + ft].  LayerNorm over the
// features of each row (C_out <= 64: one block), ReLU, 16-byte stores.
template <typename T>
__device__ __forceinline__ void finish_tile(f32x4 (&y)[4], int64_t r0, int fb, int l15, int lg, int C_out, bool has_ln,
                                            const f32x4& lw, const f32x4& lb, float inv_c, float eps, int relu,
                                            T* __restrict__ out, int64_t ldo, int64_t n_rows, int ovec) {
    if (has_ln) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            float s = 0.f;
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
                if (4 * l15 + ft < C_out) s += y[ft][reg];
            s = dif::row16_sum(s);
            const float mu = s * inv_c;
            float v = 0.f;
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
                if (4 * l15 + ft < C_out) { const float dz = y[ft][reg] - mu; v += dz * dz; }
            v = dif::row16_sum(v);
            const float rstd = 1.0f / sqrtf(v * inv_c + eps);
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) y[ft][reg] = (y[ft][reg] - mu) * rstd * lw[ft] + lb[ft];
        }
    }
    const bool vst = ovec && (fb + 4 * l15 + 3 < C_out);   // this lane's 4 features are all real
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int64_t rr = r0 + 4 * lg + reg;
        if (rr >= n_rows) continue;
        f32x4 o = {y[0][reg], y[1][reg], y[2][reg], y[3][reg]};
        if (relu) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = fmaxf(o[i], 0.f);
        }
        T* dst = out + rr * ldo + fb + 4 * l15;
        if (vst) {
            Elem<T>::st4(dst, o);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (fb + 4 * l15 + i < C_out) Elem<T>::st(dst + i, o[i]);
        }
    }
}

// Long rows into a narrow layer (C_in up to a few thousand -> C_out <= 64: the input MLP on image / text embeddings,
// image and text/run.sh; 512 -> 64 at BASELINE config C3).  W does not fit the registers-per-tile scheme above, so K runs
// in chunks of 64 channels: the four waves of a workgroup (one 16-row tile each) share the chunk's 64 x 64 weights in LDS
// (double buffered: the next chunk's global loads fly under this chunk's 64 MFMAs per wave), x arrives chunk by chunk as
// the same coalesced 16-byte loads.  One pass: x read once, LayerNorm / ReLU in registers (the vendor GEMM + a separate
// tail pass took 51 + 6 us for 50,000 x 512 -> 64).  Needs C_in % 4 == 0 and 16-byte aligned rows of x and W.
constexpr int kLongStride = 68;

template <typename T>
__global__ __launch_bounds__(256) void long_linear_kernel(const T* __restrict__ x, int64_t ldx, int64_t n_rows, int C_in,
                                                          const T* __restrict__ W, const T* __restrict__ bias, int C_out,
                                                          const T* __restrict__ ln_w, const T* __restrict__ ln_b, float eps,
                                                          int relu, T* __restrict__ out, int64_t ldo, int ovec) {
    __shared__ __attribute__((aligned(16))) float sm_w[2][64 * kLongStride];
    __shared__ float sm_b[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    if (threadIdx.x < 64) sm_b[threadIdx.x] = threadIdx.x < C_out ? Elem<T>::ld(bias + threadIdx.x) : 0.f;
    f32x4 lw = {0.f, 0.f, 0.f, 0.f}, lb = {0.f, 0.f, 0.f, 0.f};
    if (ln_w) {
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const int f = 4 * l15 + ft;
            if (f < C_out) { lw[ft] = Elem<T>::ld(ln_w + f); lb[ft] = Elem<T>::ld(ln_b + f); }
        }
    }
    const float inv_c = 1.0f / static_cast<float>(C_out);
    const int n_chunks = (C_in + 63) / 64;
    const int64_t n_tiles = (n_rows + 15) / 16;
    const int64_t n_groups = (n_tiles + 3) / 4;
    // staging of a weight chunk: thread t owns four 16-byte units (feature e / 16, channels 4 (e % 16) .. + 3)
    auto load_w = [&](f32x4 (&wv)[4], int ch) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = threadIdx.x + 256 * u;
            const int f = e >> 4, c = 64 * ch + 4 * (e & 15);
            wv[u] = (f < C_out && c < C_in) ? Elem<T>::ld4(W + static_cast<int64_t>(f) * C_in + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_w = [&](const f32x4 (&wv)[4], int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = threadIdx.x + 256 * u;
            const int f = e >> 4, c = 4 * (e & 15);
            *reinterpret_cast<f32x4*>(&sm_w[buf][(16 * (f & 3) + (f >> 2)) * kLongStride + c]) = wv[u];
        }
    };
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t tile = grp * 4 + wave;
        const int64_t r0 = tile * 16, r = r0 + l15;
        const bool rok = r < n_rows;
        auto load_x = [&](f32x4 (&xa)[4], int ch) {
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) {
                const int c = 64 * ch + 16 * cq + 4 * lg;
                xa[cq] = (rok && c < C_in) ? Elem<T>::ld4(x + r * ldx + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        f32x4 wv[4], xa[4], xn[4];
        load_w(wv, 0);
        load_x(xa, 0);
        __syncthreads();                       // the previous group is done with both buffers (and sm_b is there)
        store_w(wv, 0);
        __syncthreads();
        const f32x4 bv = *reinterpret_cast<const f32x4*>(sm_b + 4 * l15);
        f32x4 y[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) y[ft] = f32x4{bv[ft], bv[ft], bv[ft], bv[ft]};
        for (int ch = 0; ch < n_chunks; ++ch) {
            const bool more = ch + 1 < n_chunks;
            if (more) { load_w(wv, ch + 1); load_x(xn, ch + 1); }              // in flight under this chunk's MFMAs
            const float* wrow = sm_w[ch & 1] + l15 * kLongStride + 4 * lg;
            f32x4 wf[4][4];                    // all 16 weight fragments of the chunk: the LDS latency is paid once
#pragma unroll
            for (int cq = 0; cq < 4; ++cq)
#pragma unroll
                for (int ft = 0; ft < 4; ++ft)
                    wf[cq][ft] = *reinterpret_cast<const f32x4*>(wrow + 16 * ft * kLongStride + 16 * cq);
#pragma unroll
            for (int cq = 0; cq < 4; ++cq)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int ft = 0; ft < 4; ++ft)      // four independent accumulator chains back to back
                        y[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[cq][t], wf[cq][ft][t], y[ft], 0, 0, 0);
            if (more) {
                store_w(wv, (ch + 1) & 1);     // last read in iteration ch - 1, which every wave left at the barrier below
#pragma unroll
                for (int cq = 0; cq < 4; ++cq) xa[cq] = xn[cq];
            }
            __syncthreads();
        }
        if (tile < n_tiles)
            finish_tile<T>(y, r0, 0, l15, lg, C_out, ln_w != nullptr, lw, lb, inv_c, eps, relu, out, ldo, n_rows, ovec);
    }
}

// ---- long rows into a narrow layer, float32 storage, on the bfloat16 matrix core with split operands --------------------
// 512 -> 64 at CIFAR scale is 3.3 GFLOP per 115 MB: on the fp32 MFMA (157 TFLOP/s) the product alone takes 21 us, more than
// the 14 us the bytes take at 8 TB/s (measured 55 us).  Every float32 operand is split into two bfloat16 numbers,
//     x = xh + xl,   xh = bf16(x),  xl = bf16(x - xh)            (|x - xh - xl| <= 2^-17 |x|)
// and the product is formed as  xh wh + xh wl + xl wh  with float32 accumulation on v_mfma_f32_16x16x32_bf16 (16x the fp32
// rate): three MFMAs of K = 32 instead of eight of K = 4 per 32 input channels.  The dropped terms (xl wl and the second
// truncation) are <= 2^-16 relative per product: the result agrees with the fp32 kernel to ~1e-5 norm-wise, inside the 1e-4
// budget of the path (tests/test_gpu_parity.py holds it to the float64 oracle).
// The contraction index may be permuted freely as long as both operands agree: k-slot s of lane group lg in half h of a
// 64-channel chunk is channel 32 h + 16 (s / 4) + 4 lg + s % 4 -- exactly the columns a lane already holds after its four
// coalesced 16-byte loads of x.  W is split while it is staged: fragment (part, h, ft) of a chunk is 64 lanes x 16 bytes
// (8 bfloat16 = the lane's eight k-slots of feature 4 l15 + ft), read back with one conflict-free ds_read_b128.
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

__global__ __launch_bounds__(256) void long_linear_split_kernel(const float* __restrict__ x, int64_t ldx, int64_t n_rows, int C_in,
                                                                const float* __restrict__ W, const float* __restrict__ bias,
                                                                int C_out, const float* __restrict__ ln_w,
                                                                const float* __restrict__ ln_b, float eps, int relu,
                                                                float* __restrict__ out, int64_t ldo, int ovec) {
    __shared__ __attribute__((aligned(16))) bf16x8 sm_w[2][16 * 64];        // [buffer][(part * 2 + h) * 4 + ft][lane]
    __shared__ float sm_b[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    if (threadIdx.x < 64) sm_b[threadIdx.x] = threadIdx.x < C_out ? bias[threadIdx.x] : 0.f;
    f32x4 lw = {0.f, 0.f, 0.f, 0.f}, lb = {0.f, 0.f, 0.f, 0.f};
    if (ln_w) {
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const int f = 4 * l15 + ft;
            if (f < C_out) { lw[ft] = ln_w[f]; lb[ft] = ln_b[f]; }
        }
    }
    const float inv_c = 1.0f / static_cast<float>(C_out);
    const int n_chunks = (C_in + 63) / 64;
    const int64_t n_tiles = (n_rows + 15) / 16;
    const int64_t n_groups = (n_tiles + 3) / 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // staging of a weight chunk: thread t owns fragment lanes id = t, t + 256 (frag = id / 64 = 4 h + ft): two 16-byte loads
    auto load_w = [&](f32x4 (&wv)[4], int ch) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int id = threadIdx.x + 256 * u, ln_ = id & 63, frag = id >> 6;
            const int f = 4 * (ln_ & 15) + (frag & 3), c = 64 * ch + 32 * (frag >> 2) + 4 * (ln_ >> 4);
            const float* wp = W + static_cast<int64_t>(f) * C_in + c;
            wv[2 * u] = (f < C_out && c < C_in) ? *reinterpret_cast<const f32x4*>(wp) : z4;
            wv[2 * u + 1] = (f < C_out && c + 16 < C_in) ? *reinterpret_cast<const f32x4*>(wp + 16) : z4;
        }
    };
    auto store_w = [&](const f32x4 (&wv)[4], int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int id = threadIdx.x + 256 * u, ln_ = id & 63, frag = id >> 6;
            bf16x4 h0, l0, h1, l1;
            split_bf16(wv[2 * u], h0, l0);
            split_bf16(wv[2 * u + 1], h1, l1);
            sm_w[buf][frag * 64 + ln_] = cat8(h0, h1);
            sm_w[buf][(8 + frag) * 64 + ln_] = cat8(l0, l1);
        }
    };
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t tile = grp * 4 + wave;
        const int64_t r0 = tile * 16, r = r0 + l15;
        const bool rok = r < n_rows;
        auto load_x = [&](f32x4 (&xa)[4], int ch) {
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) {
                const int c = 64 * ch + 16 * cq + 4 * lg;
                xa[cq] = (rok && c < C_in) ? *reinterpret_cast<const f32x4*>(x + r * ldx + c) : z4;
            }
        };
        f32x4 wv[4], xa[4], xn[4];
        load_w(wv, 0);
        load_x(xa, 0);
        __syncthreads();                       // the previous group is done with both buffers (and sm_b is there)
        store_w(wv, 0);
        __syncthreads();
        const f32x4 bv = *reinterpret_cast<const f32x4*>(sm_b + 4 * l15);
        f32x4 y[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) y[ft] = f32x4{bv[ft], bv[ft], bv[ft], bv[ft]};
        for (int ch = 0; ch < n_chunks; ++ch) {
            const bool more = ch + 1 < n_chunks;
            if (more) { load_w(wv, ch + 1); load_x(xn, ch + 1); }              // in flight under this chunk's MFMAs
            const bf16x8* wb = sm_w[ch & 1] + lane;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf16x4 h0, l0, h1, l1;
                split_bf16(xa[2 * h], h0, l0);
                split_bf16(xa[2 * h + 1], h1, l1);
                const bf16x8 xh = cat8(h0, h1), xl = cat8(l0, l1);
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) {
                    const bf16x8 wh = wb[(4 * h + ft) * 64], wl = wb[(8 + 4 * h + ft) * 64];
                    y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, wh, y[ft], 0, 0, 0);      // small terms first
                    y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wl, y[ft], 0, 0, 0);
                    y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wh, y[ft], 0, 0, 0);
                }
            }
            if (more) {
                store_w(wv, (ch + 1) & 1);     // last read in iteration ch - 1, which every wave left at the barrier below
#pragma unroll
                for (int cq = 0; cq < 4; ++cq) xa[cq] = xn[cq];
            }
            __syncthreads();
        }
        if (tile < n_tiles)
            finish_tile<float>(y, r0, 0, l15, lg, C_out, ln_w != nullptr, lw, lb, inv_c, eps, relu, out, ldo, n_rows, ovec);
    }
}

// C_in <= 512: the whole split weight matrix (C_in x 64 x 2 parts x 2 bytes <= 128 KiB) stays in LDS.  One workgroup of 16
// waves per CU stages it once; after that a wave streams its 16-row tiles with no barrier at all: per 64-channel chunk four
// 16-byte loads of x (the next chunk's already in flight), 16 ds_read_b128, 24 MFMAs.  (The chunked kernel above re-stages
// every 16-KiB weight chunk for every group of 64 rows -- as many bytes out of L2 as x brings in from HBM -- and meets at
// two barriers per chunk.)
constexpr int kResidentChunks = 8;

__global__ __launch_bounds__(1024) void long_linear_resident_kernel(const float* __restrict__ x, int64_t ldx, int64_t n_rows,
                                                                    int C_in, const float* __restrict__ W,
                                                                    const float* __restrict__ bias, int C_out,
                                                                    const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                                    float eps, int relu, float* __restrict__ out, int64_t ldo,
                                                                    int ovec) {
    __shared__ __attribute__((aligned(16))) bf16x8 sm_w[kResidentChunks][16 * 64];   // [chunk][(part * 2 + h) * 4 + ft][lane]
    __shared__ float sm_b[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int n_chunks = (C_in + 63) / 64;
    const int64_t n_tiles = (n_rows + 15) / 16;
    const int64_t t_stride = static_cast<int64_t>(gridDim.x) * 16;
    int64_t tile = static_cast<int64_t>(blockIdx.x) * 16 + wave;
    auto load_x = [&](f32x4 (&xa)[4], int64_t t, int ch) {
        const int64_t r = t * 16 + l15;
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
            const int c = 64 * ch + 16 * cq + 4 * lg;
            xa[cq] = (r < n_rows && c < C_in) ? *reinterpret_cast<const f32x4*>(x + r * ldx + c) : z4;
        }
    };
    // A wave usually owns ONE tile (up to 65,536 rows): its first chunk of x is requested before the weights are staged, and
    // the loop keeps one chunk in flight ahead of the one it multiplies (two ahead needs a third buffer: 100 VGPRs spilled,
    // 56 us instead of 30 at 50,000 x 512).
    f32x4 xa[4], xn[4];
    if (tile < n_tiles) load_x(xa, tile, 0);
    if (threadIdx.x < 64) sm_b[threadIdx.x] = threadIdx.x < C_out ? bias[threadIdx.x] : 0.f;
    {   // all of a thread's weight loads (up to eight 16-byte loads) are in flight before the first conversion
        f32x4 w0[kResidentChunks / 2], w1[kResidentChunks / 2];
#pragma unroll
        for (int it = 0; it < kResidentChunks / 2; ++it) {
            const int idx = threadIdx.x + 1024 * it;
            const int ch = idx >> 9, id = idx & 511, ln_ = id & 63, frag = id >> 6;
            const int f = 4 * (ln_ & 15) + (frag & 3), c = 64 * ch + 32 * (frag >> 2) + 4 * (ln_ >> 4);
            const float* wp = W + static_cast<int64_t>(f) * C_in + c;
            w0[it] = (ch < n_chunks && f < C_out && c < C_in) ? *reinterpret_cast<const f32x4*>(wp) : z4;
            w1[it] = (ch < n_chunks && f < C_out && c + 16 < C_in) ? *reinterpret_cast<const f32x4*>(wp + 16) : z4;
        }
#pragma unroll
        for (int it = 0; it < kResidentChunks / 2; ++it) {
            const int idx = threadIdx.x + 1024 * it;
            const int ch = idx >> 9, id = idx & 511, ln_ = id & 63, frag = id >> 6;
            if (ch < n_chunks) {
                bf16x4 h0, l0, h1, l1;
                split_bf16(w0[it], h0, l0);
                split_bf16(w1[it], h1, l1);
                sm_w[ch][frag * 64 + ln_] = cat8(h0, h1);
                sm_w[ch][(8 + frag) * 64 + ln_] = cat8(l0, l1);
            }
        }
    }
    f32x4 lw = z4, lb = z4;
    if (ln_w) {
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const int f = 4 * l15 + ft;
            if (f < C_out) { lw[ft] = ln_w[f]; lb[ft] = ln_b[f]; }
        }
    }
    const float inv_c = 1.0f / static_cast<float>(C_out);
    __syncthreads();
    const f32x4 bv = *reinterpret_cast<const f32x4*>(sm_b + 4 * l15);
    for (; tile < n_tiles; tile += t_stride) {
        f32x4 y[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) y[ft] = f32x4{bv[ft], bv[ft], bv[ft], bv[ft]};
        for (int ch = 0; ch < n_chunks; ++ch) {
            if (ch + 1 < n_chunks) load_x(xn, tile, ch + 1);
            else if (tile + t_stride < n_tiles) load_x(xn, tile + t_stride, 0);
            const bf16x8* wb = sm_w[ch] + lane;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf16x4 h0, l0, h1, l1;
                split_bf16(xa[2 * h], h0, l0);
                split_bf16(xa[2 * h + 1], h1, l1);
                const bf16x8 xh = cat8(h0, h1), xl = cat8(l0, l1);
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) {
                    const bf16x8 wh = wb[(4 * h + ft) * 64], wl = wb[(8 + 4 * h + ft) * 64];
                    y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, wh, y[ft], 0, 0, 0);      // small terms first
                    y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wl, y[ft], 0, 0, 0);
                    y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wh, y[ft], 0, 0, 0);
                }
            }
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) xa[cq] = xn[cq];
        }
        finish_tile<float>(y, tile * 16, 0, l15, lg, C_out, ln_w != nullptr, lw, lb, inv_c, eps, relu, out, ldo, n_rows, ovec);
    }
}

// ---- long rows, FEW rows or rows that are not 16-byte aligned (Cora: 2,708 x 1,433 -> 64, difformer.py:188-191) ------------
// The chunked kernels above give every workgroup four whole row tiles and walk K chunk by chunk: at 170 row tiles that is 43
// workgroups of 23 dependent chunk steps each.  Here a workgroup owns ONE 16-row tile and its eight waves split K: wave w takes
// the 64-channel chunks w, w + 8, ... with two chunks of loads in flight, the partial tiles meet in LDS in wave order
// (deterministic), then bias, LayerNorm, ReLU and the 16-byte stores.  Rows of x only need 4-byte alignment:
// `global_load_dwordx4` takes dword-aligned addresses, a vector that straddles the end of a row is read element by element.
// W comes PACKED (dif_linear_pack_f32, once per parameter version): already split into bfloat16 hi / lo parts and stored
// fragment by fragment in the order a lane feeds the MFMA -- [chunk][part * 8 + h * 4 + ft][lane] x 16 bytes -- so a wave
// reads each fragment as ONE contiguous KiB (a lane picking its own elements out of W touches 16 weight rows per load
// instruction, two cache lines each: the vector-memory tag rate then bounds the kernel at ~20 us for Cora) and there is no
// per-tile conversion work.  Exact mode (DIFFORMER_EXACT_FP32=1): long_linear_ksplit_exact_kernel on the fp32 MFMA.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ f32x4 ld4_tail(const float* __restrict__ row, int c, int C_in) {
    if (c + 3 < C_in) return *reinterpret_cast<const f32x4u*>(row + c);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (c + i < C_in) v[i] = row[c + i];
    return v;
}

// one thread per (chunk, h * 4 + ft, lane): the lane's eight k-slots of feature 4 l15 + ft in half h of the chunk (channels
// 64 ch + 32 h + 4 lg .. + 3 and the same + 16), hi and lo parts
__global__ __launch_bounds__(256) void linear_pack_kernel(const float* __restrict__ W, int C_in, int C_out, int n_chunks,
                                                          bf16x8* __restrict__ packed) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= n_chunks * 512) return;
    const int ch = id >> 9, fr = (id >> 6) & 7, lane = id & 63;
    const int l15 = lane & 15, lg = lane >> 4, h = fr >> 2, ft = fr & 3;
    const int f = 4 * l15 + ft, c = 64 * ch + 32 * h + 4 * lg;
    f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
    if (f < C_out) {
        const float* wr = W + static_cast<int64_t>(f) * C_in;
        w0 = ld4_tail(wr, c, C_in);
        w1 = ld4_tail(wr, c + 16, C_in);
    }
    bf16x4 h0, l0, h1, l1;
    split_bf16(w0, h0, l0);
    split_bf16(w1, h1, l1);
    packed[(static_cast<int64_t>(ch) * 16 + fr) * 64 + lane] = cat8(h0, h1);
    packed[(static_cast<int64_t>(ch) * 16 + 8 + fr) * 64 + lane] = cat8(l0, l1);
}

constexpr int kPackedWaves = 8;

__global__ __launch_bounds__(64 * kPackedWaves) void long_linear_packed_kernel(const float* __restrict__ x, int64_t ldx, int64_t n_rows,
                                                                               int C_in, const bf16x8* __restrict__ packed,
                                                                               const float* __restrict__ bias, int C_out,
                                                                               const float* __restrict__ ln_w,
                                                                               const float* __restrict__ ln_b, float eps, int relu,
                                                                               float* __restrict__ out, int64_t ldo, int ovec) {
    __shared__ __attribute__((aligned(16))) f32x4 part[kPackedWaves - 1][4][64];      // partial tiles of waves 1 ..
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;                         // min(kPackedWaves, chunks): no wave without a chunk
    const int l15 = lane & 15, lg = lane >> 4;
    const int n_chunks = (C_in + 63) / 64;
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * 16, r = r0 + l15;
    const bool rok = r < n_rows;
    const float* xrow = x + (rok ? r : 0) * ldx;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    auto load = [&](f32x4 (&xa)[4], bf16x8 (&wf)[16], int ch) {
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) xa[cq] = rok ? ld4_tail(xrow, 64 * ch + 16 * cq + 4 * lg, C_in) : z4;
        const bf16x8* wp = packed + static_cast<int64_t>(ch) * 16 * 64 + lane;
#pragma unroll
        for (int fr = 0; fr < 16; ++fr) wf[fr] = wp[fr * 64];
    };
    f32x4 y[4] = {z4, z4, z4, z4};
    auto multiply = [&](const f32x4 (&xa)[4], const bf16x8 (&wf)[16]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x4 h0, l0, h1, l1;
            split_bf16(xa[2 * h], h0, l0);
            split_bf16(xa[2 * h + 1], h1, l1);
            const bf16x8 xh = cat8(h0, h1), xl = cat8(l0, l1);
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, wf[4 * h + ft], y[ft], 0, 0, 0);      // small terms first
                y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wf[8 + 4 * h + ft], y[ft], 0, 0, 0);
                y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wf[4 * h + ft], y[ft], 0, 0, 0);
            }
        }
    };
    f32x4 xa[4], xb[4];
    bf16x8 wa[16], wb[16];
    int ca = wave, cb = wave + nw;
    if (ca < n_chunks) load(xa, wa, ca);
    if (cb < n_chunks) load(xb, wb, cb);
    while (ca < n_chunks) {
        multiply(xa, wa);
        ca += 2 * nw;
        if (ca < n_chunks) load(xa, wa, ca);
        if (cb < n_chunks) {
            multiply(xb, wb);
            cb += 2 * nw;
            if (cb < n_chunks) load(xb, wb, cb);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) part[wave - 1][ft][lane] = y[ft];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < nw - 1; ++w)
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) y[ft] += part[w][ft][lane];
    f32x4 lw = z4, lb = z4;
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
        const int f = 4 * l15 + ft;
        const float b = f < C_out ? bias[f] : 0.f;
        y[ft] += f32x4{b, b, b, b};
        if (ln_w && f < C_out) { lw[ft] = ln_w[f]; lb[ft] = ln_b[f]; }
    }
    finish_tile<float>(y, r0, 0, l15, lg, C_out, ln_w != nullptr, lw, lb, 1.0f / static_cast<float>(C_out), eps, relu, out, ldo,
                       n_rows, ovec);
}

// the same split of K over four waves on the exact fp32 MFMA, operands straight from global memory (DIFFORMER_EXACT_FP32=1)
__global__ __launch_bounds__(256) void long_linear_ksplit_exact_kernel(const float* __restrict__ x, int64_t ldx, int64_t n_rows, int C_in,
                                                                       const float* __restrict__ W, const float* __restrict__ bias,
                                                                       int C_out, const float* __restrict__ ln_w,
                                                                       const float* __restrict__ ln_b, float eps, int relu,
                                                                       float* __restrict__ out, int64_t ldo, int ovec) {
    __shared__ __attribute__((aligned(16))) f32x4 part[3][4][64];           // partial tiles of waves 1..3
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int n_chunks = (C_in + 63) / 64;
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * 16, r = r0 + l15;
    const bool rok = r < n_rows;
    const float* xrow = x + (rok ? r : 0) * ldx;
    const float* wrow[4];
    bool fok[4];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
        fok[ft] = 4 * l15 + ft < C_out;
        wrow[ft] = W + static_cast<int64_t>(fok[ft] ? 4 * l15 + ft : 0) * C_in;
    }
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    auto load = [&](f32x4 (&xa)[4], f32x4 (&wa)[4][4], int ch) {
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
            const int c = 64 * ch + 16 * cq + 4 * lg;
            xa[cq] = rok ? ld4_tail(xrow, c, C_in) : z4;
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) wa[ft][cq] = fok[ft] ? ld4_tail(wrow[ft], c, C_in) : z4;
        }
    };
    f32x4 y[4] = {z4, z4, z4, z4};
    f32x4 xa[4], wa[4][4], xn[4], wn[4][4];
    if (wave < n_chunks) load(xa, wa, wave);
    for (int ch = wave; ch < n_chunks; ch += 4) {
        const bool more = ch + 4 < n_chunks;
        if (more) load(xn, wn, ch + 4);
#pragma unroll
        for (int cq = 0; cq < 4; ++cq)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ft = 0; ft < 4; ++ft)
                    y[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[cq][t], wa[ft][cq][t], y[ft], 0, 0, 0);
        if (more) {
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) {
                xa[cq] = xn[cq];
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) wa[ft][cq] = wn[ft][cq];
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) part[wave - 1][ft][lane] = y[ft];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) y[ft] += part[w][ft][lane];
    f32x4 lw = z4, lb = z4;
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
        const int f = 4 * l15 + ft;
        const float b = f < C_out ? bias[f] : 0.f;
        y[ft] += f32x4{b, b, b, b};
        if (ln_w && f < C_out) { lw[ft] = ln_w[f]; lb[ft] = ln_b[f]; }
    }
    finish_tile<float>(y, r0, 0, l15, lg, C_out, ln_w != nullptr, lw, lb, 1.0f / static_cast<float>(C_out), eps, relu, out, ldo,
                       n_rows, ovec);
}

// grid (row chunks, ceil(C_out/256)); 256 threads; dynamic LDS = blocks * 64 * (kLinStride + 1) floats.
// KQ = number of 16-channel groups of C_in actually used (1..8).
template <int KQ, typename T>
__global__ __launch_bounds__(256) void skinny_linear_kernel(const T* __restrict__ x, int64_t ldx, int64_t n_rows,
                                                            int C_in, const T* __restrict__ W,
                                                            const T* __restrict__ bias, int C_out,
                                                            const T* __restrict__ ln_w,
                                                            const T* __restrict__ ln_b, float eps, int relu,
                                                            T* __restrict__ out, int64_t ldo, int vec, int ovec, int wvec) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    constexpr int kLinStride = lin_stride(KQ), kCW = lin_channels(KQ), kLinMaxBlocks = lin_max_blocks(KQ);
    const int f0 = blockIdx.y * 64 * kLinMaxBlocks;
    int nblk = (C_out - f0 + 63) / 64;
    if (nblk > kLinMaxBlocks) nblk = kLinMaxBlocks;
    float* sm_w = sm;                                       // [nblk*64][kLinStride], block-local row 16 (f%4) + (f%64)/4
    float* sm_b = sm + nblk * 64 * kLinStride;              // [nblk*64] bias in feature order

    if (wvec) {
        // 16-byte loads, eight in flight per thread before their LDS stores: the staging is the fixed cost of a launch
        // (a 64 -> 192 projection of a Cora-sized graph spent 20 of its 27 us here with 4-byte loads)
        constexpr int kC4 = kCW / 4;
        const int units = nblk * 64 * kC4;
        for (int base = threadIdx.x; base < units; base += 256 * 8) {
            f32x4 wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + 256 * u;
                const int f = e / kC4, c = 4 * (e % kC4);
                wv[u] = (e < units && f0 + f < C_out && c < C_in) ? Elem<T>::ld4(W + static_cast<int64_t>(f0 + f) * C_in + c)
                                                                  : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = base + 256 * u;
                const int f = e / kC4, c = 4 * (e % kC4);
                if (e < units)
                    *reinterpret_cast<f32x4*>(&sm_w[((f & ~63) + 16 * (f & 3) + ((f & 63) >> 2)) * kLinStride + c]) = wv[u];
            }
        }
    } else {
    // 8 loads in flight per thread before their LDS stores (a plain load -> store loop is 16 round trips per block)
    for (int base = threadIdx.x; base < nblk * 64 * kCW; base += 256 * 8) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = base + 256 * u;                      // nblk * 64 * kCW is a multiple of 2048: e stays in range
            const int f = e / kCW, c = e % kCW;
            wv[u] = (f0 + f < C_out && c < C_in) ? Elem<T>::ld(W + static_cast<int64_t>(f0 + f) * C_in + c) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = base + 256 * u;
            const int f = e / kCW, c = e % kCW;
            sm_w[((f & ~63) + 16 * (f & 3) + ((f & 63) >> 2)) * kLinStride + c] = wv[u];
        }
    }
    }
    for (int f = threadIdx.x; f < nblk * 64; f += 256) sm_b[f] = (f0 + f < C_out) ? Elem<T>::ld(bias + f0 + f) : 0.f;
    __syncthreads();

    f32x4 lw = {0.f, 0.f, 0.f, 0.f}, lb = {0.f, 0.f, 0.f, 0.f};   // LayerNorm: C_out <= 64, one block
    if (ln_w) {
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const int f = 4 * l15 + ft;
            if (f < C_out) { lw[ft] = Elem<T>::ld(ln_w + f); lb[ft] = Elem<T>::ld(ln_b + f); }
        }
    }
    const float inv_c = 1.0f / static_cast<float>(C_out);

    auto load_x = [&](int64_t tile, f32x4 (&xa)[KQ]) {
        const int64_t r = tile * 16 + l15;
#pragma unroll
        for (int cq = 0; cq < KQ; ++cq) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const int c = 16 * cq + 4 * lg;
            if (r < n_rows) {
                const T* p = x + r * ldx + c;
                if (c + 3 < C_in) {            // rows that are only element-aligned (65 columns: Pokec): still one load
                    z = vec ? Elem<T>::ld4(p) : Elem<T>::ld4u(p);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (c + i < C_in) z[i] = Elem<T>::ld(p + i);
                }
            }
            xa[cq] = z;
        }
    };

    const int64_t n_tiles = (n_rows + 15) / 16;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 4;
    // bfloat16 rows: half the bytes per load instruction -> two tiles ahead (see gram_kernel)
    constexpr bool kTwoAhead = sizeof(T) == 2;
    f32x4 xa[KQ], xn[KQ], xn2[KQ];
    if (first < n_tiles) load_x(first, xa);
    if (kTwoAhead && first + stride < n_tiles) load_x(first + stride, xn);
    // LayerNorm over 65..128 output features (the input layer at the scripts' hidden 128, run.sh:42-44): both 64-feature blocks
    // of a row tile are formed first, normalised together, then stored (the host guarantees one workgroup column: f0 == 0)
    const bool wide_ln = ln_w != nullptr && C_out > 64;
    f32x4 lw2 = {0.f, 0.f, 0.f, 0.f}, lb2 = {0.f, 0.f, 0.f, 0.f};
    if (wide_ln) {
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const int f = 64 + 4 * l15 + ft;
            if (f < C_out) { lw2[ft] = Elem<T>::ld(ln_w + f); lb2[ft] = Elem<T>::ld(ln_b + f); }
        }
    }
    for (int64_t tile = first; tile < n_tiles; tile += stride) {
        const int64_t r0 = tile * 16;
        if (kTwoAhead) {
            if (tile + 2 * stride < n_tiles) load_x(tile + 2 * stride, xn2);
        } else if (tile + stride < n_tiles) load_x(tile + stride, xn);        // in flight under this tile's work
        if (wide_ln) {
            f32x4 y2[2][4];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const float* wrow = sm_w + (64 * blk + l15) * kLinStride + 4 * lg;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(sm_b + 64 * blk + 4 * l15);
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) y2[blk][ft] = f32x4{bv[ft], bv[ft], bv[ft], bv[ft]};
#pragma unroll
                for (int cq = 0; cq < KQ; ++cq)
#pragma unroll
                    for (int ft = 0; ft < 4; ++ft) {
                        const f32x4 wf = *reinterpret_cast<const f32x4*>(wrow + 16 * ft * kLinStride + 16 * cq);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            y2[blk][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[cq][t], wf[t], y2[blk][ft], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float sm = 0.f;
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) sm += y2[0][ft][reg] + ((64 + 4 * l15 + ft < C_out) ? y2[1][ft][reg] : 0.f);
                sm = dif::row16_sum(sm);
                const float mu = sm * inv_c;
                float v = 0.f;
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) {
                    const float d0 = y2[0][ft][reg] - mu, d1 = (64 + 4 * l15 + ft < C_out) ? y2[1][ft][reg] - mu : 0.f;
                    v += d0 * d0 + d1 * d1;
                }
                v = dif::row16_sum(v);
                const float rstd = 1.0f / sqrtf(v * inv_c + eps);
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) {
                    y2[0][ft][reg] = (y2[0][ft][reg] - mu) * rstd * lw[ft] + lb[ft];
                    y2[1][ft][reg] = (y2[1][ft][reg] - mu) * rstd * lw2[ft] + lb2[ft];
                }
            }
            finish_tile<T>(y2[0], r0, 0, l15, lg, C_out, false, lw, lb, inv_c, eps, relu, out, ldo, n_rows, ovec);
            finish_tile<T>(y2[1], r0, 64, l15, lg, C_out, false, lw2, lb2, inv_c, eps, relu, out, ldo, n_rows, ovec);
        } else
        for (int blk = 0; blk < nblk; ++blk) {
            const int fb = f0 + 64 * blk;
            const float* wrow = sm_w + (64 * blk + l15) * kLinStride + 4 * lg;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(sm_b + 64 * blk + 4 * l15);
            f32x4 y[4];
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) y[ft] = f32x4{bv[ft], bv[ft], bv[ft], bv[ft]};
#pragma unroll
            for (int cq = 0; cq < KQ; ++cq)
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) {
                    const f32x4 wf = *reinterpret_cast<const f32x4*>(wrow + 16 * ft * kLinStride + 16 * cq);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        y[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[cq][t], wf[t], y[ft], 0, 0, 0);
                }
            finish_tile<T>(y, r0, fb, l15, lg, C_out, ln_w != nullptr, lw, lb, inv_c, eps, relu, out, ldo, n_rows, ovec);
        }
#pragma unroll
        for (int cq = 0; cq < KQ; ++cq) {
            xa[cq] = xn[cq];
            if (kTwoAhead) xn[cq] = xn2[cq];
        }
    }
}

// bfloat16 storage, C_in <= 128 -> C_out <= 64 (the input layer of a Pokec mini-batch in bf16: 100,000 x 65 -> 64, BASELINE
// config C5): x and W are bfloat16, so their products are exact in float32 and the product can run on
// v_mfma_f32_16x16x32_bf16 -- 32 input channels per instruction where the converted-to-float32 path above needs eight
// v_mfma_f32_16x16x4_f32 and four conversions per load (that path is issue-bound at every size: 1.3 TB/s at 1.6 M rows
// against 3.2 TB/s for float32 storage, profiles/r05_experiments.md).  A lane (row l15, k-group lg) reads the 16 contiguous
// bytes of channels 32 kb + 8 lg .. + 7 of its row: rows of 65 elements start on odd 2-byte boundaries, so the load is five
// ALIGNED dwords shifted into place (v_alignbit).  W sits in LDS as ready-made B fragments [kb][ft][lane] (feature
// 4 l15 + ft: the epilogue's layout), zero beyond C_in.  The LAST 16-row tile reads element by element (no read past the
// end of the tensor).
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

template <int KB>
__global__ __launch_bounds__(256) void skinny_linear_bf16_kernel(const dif::bf16* __restrict__ x, int64_t ldx, int64_t n_rows, int C_in,
                                                                 const dif::bf16* __restrict__ W, const dif::bf16* __restrict__ bias,
                                                                 int C_out, const dif::bf16* __restrict__ ln_w,
                                                                 const dif::bf16* __restrict__ ln_b, float eps, int relu,
                                                                 dif::bf16* __restrict__ out, int64_t ldo, int ovec) {
    using B = dif::bf16;
    __shared__ __attribute__((aligned(16))) uint16_t sm_w[KB * 4 * 64 * 8];
    __shared__ __attribute__((aligned(16))) float sm_b[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    for (int e = threadIdx.x; e < KB * 4 * 64 * 8; e += 256) {
        const int s_ = e & 7, ln_ = (e >> 3) & 63, ft = (e >> 9) & 3, kb = e >> 11;
        const int f = 4 * (ln_ & 15) + ft, c = 32 * kb + 8 * (ln_ >> 4) + s_;
        sm_w[e] = (f < C_out && c < C_in) ? W[static_cast<int64_t>(f) * C_in + c].bits : uint16_t(0);
    }
    if (threadIdx.x < 64) sm_b[threadIdx.x] = threadIdx.x < C_out ? Elem<B>::ld(bias + threadIdx.x) : 0.f;
    __syncthreads();
    f32x4 lw = {0.f, 0.f, 0.f, 0.f}, lb = {0.f, 0.f, 0.f, 0.f};
    if (ln_w) {
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const int f = 4 * l15 + ft;
            if (f < C_out) { lw[ft] = Elem<B>::ld(ln_w + f); lb[ft] = Elem<B>::ld(ln_b + f); }
        }
    }
    const float inv_c = 1.0f / static_cast<float>(C_out);
    const int64_t n_tiles = (n_rows + 15) / 16;

    // Rows are 2-byte aligned at best (F_in = 65: 130-byte rows): a lane's eight channels come as five aligned dwords, shifted
    // into place.  Full tiles (every tile but the last: its 16 rows and the row after them exist, so 20 bytes from any of their
    // channels stay inside the tensor) are fetched RAW a tile ahead and assembled -- shifted, masked beyond C_in -- only when
    // used; the last tile takes the element-wise path.  (Assembled at issue time, and the ragged channel block element-wise on
    // every tile, each load waited for itself: 53 of the kernel's 58 loads were serialised round trips.)
    typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
    struct Raw { u32x4a lo; uint32_t hi; };
    const bool fast_ok = ldx >= 9 && n_tiles > 1;
    u32x4v cmask[KB];                           // channels of this lane that exist, per k-block
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int nv = C_in - (32 * kb + 8 * lg);           // valid channels among the lane's eight
#pragma unroll
        for (int i = 0; i < 4; ++i) cmask[kb][i] = nv >= 2 * i + 2 ? 0xffffffffu : (nv == 2 * i + 1 ? 0x0000ffffu : 0u);
    }
    // dword view of x by pointer arithmetic only (an integer round trip of the address loses the address space: flat loads)
    const int64_t xoff = static_cast<int64_t>((reinterpret_cast<uintptr_t>(x) >> 1) & 1u);          // x starts xoff elements into a dword
    const uint32_t* xw = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(x) - 2 * xoff);
    auto fetch_raw = [&](int64_t tile, Raw (&rw)[KB]) {     // tile < n_tiles - 1
        const int64_t r = tile * 16 + l15;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int c = 32 * kb + 8 * lg;
            const uint32_t* q = xw + ((r * ldx + (c < C_in ? c : 0) + xoff) >> 1);
            rw[kb].lo = *reinterpret_cast<const u32x4a*>(q);
            rw[kb].hi = q[4];
        }
    };
    auto assemble = [&](int64_t tile, const Raw (&rw)[KB], u32x4v (&xa)[KB]) {
        const int64_t r = tile * 16 + l15;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int c = 32 * kb + 8 * lg;
            const uint32_t sh = static_cast<uint32_t>((r * ldx + (c < C_in ? c : 0) + xoff) & 1) * 16u;
            u32x4v z;
            z[0] = __builtin_amdgcn_alignbit(rw[kb].lo[1], rw[kb].lo[0], sh);
            z[1] = __builtin_amdgcn_alignbit(rw[kb].lo[2], rw[kb].lo[1], sh);
            z[2] = __builtin_amdgcn_alignbit(rw[kb].lo[3], rw[kb].lo[2], sh);
            z[3] = __builtin_amdgcn_alignbit(rw[kb].hi, rw[kb].lo[3], sh);
            xa[kb] = z & cmask[kb];
        }
    };
    auto load_careful = [&](int64_t tile, u32x4v (&xa)[KB]) {                // the last tile: element by element
        const int64_t r = tile * 16 + l15;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            u32x4v z = {0u, 0u, 0u, 0u};
            const int c = 32 * kb + 8 * lg;
            if (r < n_rows && c < C_in) {
                const B* p = x + r * ldx + c;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (c + i < C_in) z[i >> 1] |= static_cast<uint32_t>(p[i].bits) << (16 * (i & 1));
            }
            xa[kb] = z;
        }
    };

    const int64_t first = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 4;
    const int64_t last_fast = n_tiles - 2;                                  // only meaningful with fast_ok
    Raw rw[KB];
    if (fast_ok) fetch_raw(first <= last_fast ? first : last_fast, rw);
    const bf16x8v* wfrag = reinterpret_cast<const bf16x8v*>(sm_w);
    for (int64_t tile = first; tile < n_tiles; tile += stride) {
        u32x4v xa[KB];
        if (fast_ok && tile <= last_fast) assemble(tile, rw, xa);
        else load_careful(tile, xa);
        if (fast_ok) {
            const int64_t nxt = tile + stride;
            fetch_raw(nxt <= last_fast ? nxt : (tile <= last_fast ? tile : last_fast), rw);     // unconditional: no wait where it is issued
            __builtin_amdgcn_sched_barrier(0);
        }
        const f32x4 bv = *reinterpret_cast<const f32x4*>(sm_b + 4 * l15);
        f32x4 y[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) y[ft] = f32x4{bv[ft], bv[ft], bv[ft], bv[ft]};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const bf16x8v av = __builtin_bit_cast(bf16x8v, xa[kb]);
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
                y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, wfrag[(kb * 4 + ft) * 64 + lane], y[ft], 0, 0, 0);
        }
        finish_tile<B>(y, tile * 16, 0, l15, lg, C_out, ln_w != nullptr, lw, lb, inv_c, eps, relu, out, ldo, n_rows, ovec);
    }
}

template <typename T>
int linear_entry(const T* x, int64_t ldx, int64_t n_rows, int C_in, const T* W, const T* bias, int C_out,
                 const T* ln_weight, const T* ln_bias, float ln_eps, int relu, T* out, int64_t ldo, dif_stream_t stream) {
    DIF_REQUIRE(n_rows > 0 && C_in > 0 && C_out > 0, DIF_E_BADARG, "dif_linear: n_rows, C_in, C_out must be positive");
    DIF_REQUIRE(x && W && bias && out, DIF_E_BADARG, "dif_linear: null pointer");
    if (C_in > 128) {                                     // long rows into a narrow layer
        DIF_REQUIRE((ln_weight == nullptr) == (ln_bias == nullptr), DIF_E_BADARG,
                    "dif_linear: ln_weight and ln_bias must be given together");
        DIF_REQUIRE(ldx >= C_in && ldo >= C_out, DIF_E_BADARG, "dif_linear: leading dimension smaller than a row");
        const bool aligned = C_in % 4 == 0 && ldx % 4 == 0 && dif::aligned_v4<T>(x) && dif::aligned_v4<T>(W);
        if constexpr (std::is_same<T, float>::value) {
            // few rows, or rows that are only 4-byte aligned (C_in % 4 != 0: Cora's 1,433 features): one workgroup per 16-row
            // tile, K split over its waves (long_linear_ksplit_kernel)
            if (C_out <= 64 && C_in <= 8192 && (dif::exact_fp32() ? (!aligned || n_rows < 16384) : !aligned)) {
                // exact mode, or rows the chunked kernels cannot take: K split over four waves on the fp32 MFMA.  (The fast form
                // of this shape is dif_linear_packed_f32, which the host calls with the packed weights it caches.)
                const int ov = (ldo % 4 == 0) && dif::aligned_v4<T>(out);
                const unsigned tiles = static_cast<unsigned>((n_rows + 15) / 16);
                hipLaunchKernelGGL(long_linear_ksplit_exact_kernel, dim3(tiles), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx,
                                   n_rows, C_in, W, bias, C_out, ln_weight, ln_bias, ln_eps, relu, out, ldo, ov);
                return dif::launch_status("long_linear_ksplit_exact_kernel");
            }
        }
        DIF_REQUIRE(C_out <= 64 && C_in <= 8192 && aligned,
                    DIF_E_SHAPE, "dif_linear: C_in > 128 needs C_out <= 64, C_in <= 8192 and (bfloat16 storage) C_in %% 4 == 0 with "
                    "rows of x / W aligned to 4 elements (got %d -> %d); use the vendor GEMM", C_in, C_out);
        const int64_t groups = ((n_rows + 15) / 16 + 3) / 4;
        int64_t g = groups < 4 * dif::kCUs ? groups : 4 * dif::kCUs;      // 35 KB of LDS: four workgroups per CU
        const int ov = (ldo % 4 == 0) && dif::aligned_v4<T>(out);
        if constexpr (std::is_same<T, float>::value) {
            // float32 storage: split-bfloat16 operands on the bf16 matrix core (DIFFORMER_LINEAR_FP32_MFMA=1: the exact
            // fp32-MFMA kernel, for A/B measurements)
            static const bool fp32_mfma = [] { const char* e = getenv("DIFFORMER_LINEAR_FP32_MFMA"); return e && e[0] == '1'; }();
            const bool exact = dif::exact_fp32() || fp32_mfma;
            if (!exact && C_in <= 64 * kResidentChunks && n_rows >= 32768) {     // below: the chunked kernel's many small workgroups win
                const int64_t tiles = (n_rows + 15) / 16;
                const int64_t wg = (tiles + 15) / 16 < dif::kCUs ? (tiles + 15) / 16 : dif::kCUs;      // one 16-wave workgroup per CU
                hipLaunchKernelGGL(long_linear_resident_kernel, dim3(static_cast<unsigned>(wg)), dim3(1024), 0,
                                   static_cast<hipStream_t>(stream), x, ldx, n_rows, C_in, W, bias, C_out, ln_weight, ln_bias, ln_eps,
                                   relu, out, ldo, ov);
                return dif::launch_status("long_linear_resident_kernel");
            }
            if (!exact) {
                hipLaunchKernelGGL(long_linear_split_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0,
                                   static_cast<hipStream_t>(stream), x, ldx, n_rows, C_in, W, bias, C_out, ln_weight, ln_bias, ln_eps,
                                   relu, out, ldo, ov);
                return dif::launch_status("long_linear_split_kernel");
            }
        }
        hipLaunchKernelGGL((long_linear_kernel<T>), dim3(static_cast<unsigned>(g)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                           ldx, n_rows, C_in, W, bias, C_out, ln_weight, ln_bias, ln_eps, relu, out, ldo, ov);
        return dif::launch_status("long_linear_kernel");
    }
    DIF_REQUIRE((ln_weight == nullptr) == (ln_bias == nullptr), DIF_E_BADARG,
                "dif_linear: ln_weight and ln_bias must be given together");
    DIF_REQUIRE(!ln_weight || C_out <= 128, DIF_E_SHAPE, "dif_linear: fused LayerNorm needs C_out <= 128");
    DIF_REQUIRE(ldx >= C_in && ldo >= C_out, DIF_E_BADARG, "dif_linear: leading dimension smaller than a row");
    if constexpr (std::is_same<T, dif::bf16>::value) {
        if (C_out <= 64 && C_in <= 128 && !dif::exact_fp32()) {
            const int64_t n_tiles = (n_rows + 15) / 16;
            int64_t gx = (n_tiles + 15) / 16;
            const int64_t one_each = (n_tiles + 3) / 4;
            if (gx < dif::kCUs) gx = one_each < dif::kCUs ? one_each : dif::kCUs;
            if (gx > 4 * dif::kCUs) gx = 4 * dif::kCUs;
            const int ovec = (ldo % 4 == 0) && dif::aligned_v4<T>(out);
            const int kb = (C_in + 31) / 32;
            hipStream_t st = static_cast<hipStream_t>(stream);
#define DIF_LINB(KB) \
            hipLaunchKernelGGL((skinny_linear_bf16_kernel<KB>), dim3(static_cast<unsigned>(gx)), dim3(256), 0, st, x, ldx, n_rows, C_in, W, \
                               bias, C_out, ln_weight, ln_bias, ln_eps, relu, out, ldo, ovec)
            if (kb == 1) DIF_LINB(1);
            else if (kb == 2) DIF_LINB(2);
            else if (kb == 3) DIF_LINB(3);
            else DIF_LINB(4);
#undef DIF_LINB
            return dif::launch_status("skinny_linear_bf16_kernel");
        }
    }
    const int kq = (C_in + 15) / 16;
    const int kLinMaxBlocks = lin_max_blocks(kq), kLinStride = lin_stride(kq);
    const int gy = (C_out + 64 * kLinMaxBlocks - 1) / (64 * kLinMaxBlocks);
    DIF_REQUIRE(gy <= 65535, DIF_E_RANGE, "dif_linear: C_out too large");
    const int nblk = (C_out >= 64 * kLinMaxBlocks) ? kLinMaxBlocks : (C_out + 63) / 64;
    const size_t lds = static_cast<size_t>(nblk) * 64 * (kLinStride + 1) * sizeof(float);
    const int vec = (C_in % 4 == 0) && (ldx % 4 == 0) && dif::aligned_v4<T>(x);
    const int ovec = (ldo % 4 == 0) && dif::aligned_v4<T>(out);
    const int wvec = (C_in % 4 == 0) && dif::aligned_v4<T>(W);
    const int64_t n_tiles = (n_rows + 15) / 16;
    int64_t gx = (n_tiles + 15) / 16;                       // >= 4 row tiles per wave: the weight staging is amortised ...
    const int64_t one_each = (n_tiles + 3) / 4;             // ... unless the chip would sit idle: then one tile per wave
    if (gx < dif::kCUs) gx = one_each < dif::kCUs ? one_each : dif::kCUs;
    const int64_t cap = (nblk <= 2 ? 4 : 2) * dif::kCUs;    // what the LDS footprint lets a CU hold
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(static_cast<unsigned>(gx), gy), block(256);
#define DIF_LIN(KQ) \
    hipLaunchKernelGGL((skinny_linear_kernel<KQ, T>), grid, block, lds, st, x, ldx, n_rows, C_in, W, bias, C_out, ln_weight, \
                       ln_bias, ln_eps, relu, out, ldo, vec, ovec, wvec)
    if (kq == 1) DIF_LIN(1);
    else if (kq == 2) DIF_LIN(2);
    else if (kq == 3) DIF_LIN(3);
    else if (kq == 4) DIF_LIN(4);
    else if (kq == 5) DIF_LIN(5);
    else if (kq == 6) DIF_LIN(6);
    else if (kq == 7) DIF_LIN(7);
    else DIF_LIN(8);
#undef DIF_LIN
    return dif::launch_status("skinny_linear_kernel");
}

}  // namespace

extern "C" int dif_linear_f32(const float* x, int64_t ldx, int64_t n_rows, int C_in, const float* W, const float* bias,
                              int C_out, const float* ln_weight, const float* ln_bias, float ln_eps, int relu,
                              float* out, int64_t ldo, dif_stream_t stream) {
    return linear_entry<float>(x, ldx, n_rows, C_in, W, bias, C_out, ln_weight, ln_bias, ln_eps, relu, out, ldo, stream);
}

extern "C" int64_t dif_linear_packed_bytes(int C_in) {
    return C_in > 0 ? static_cast<int64_t>((C_in + 63) / 64) * 16 * 64 * 16 : 0;
}

extern "C" int dif_linear_pack_f32(const float* W, int C_in, int C_out, void* packed, dif_stream_t stream) {
    DIF_REQUIRE(W && packed && C_in > 0 && C_out > 0 && C_out <= 64, DIF_E_BADARG, "dif_linear_pack_f32: needs W, packed, 0 < C_out <= 64");
    DIF_REQUIRE(dif::aligned16(packed), DIF_E_BADARG, "dif_linear_pack_f32: packed must be 16-byte aligned");
    const int n_chunks = (C_in + 63) / 64;
    hipLaunchKernelGGL(linear_pack_kernel, dim3(static_cast<unsigned>(n_chunks * 2)), dim3(256), 0, static_cast<hipStream_t>(stream), W,
                       C_in, C_out, n_chunks, static_cast<bf16x8*>(packed));
    return dif::launch_status("linear_pack_kernel");
}

extern "C" int dif_linear_packed_f32(const float* x, int64_t ldx, int64_t n_rows, int C_in, const void* packed, const float* bias,
                                     int C_out, const float* ln_weight, const float* ln_bias, float ln_eps, int relu, float* out,
                                     int64_t ldo, dif_stream_t stream) {
    DIF_REQUIRE(n_rows > 0 && C_in > 0 && C_out > 0 && C_out <= 64, DIF_E_BADARG, "dif_linear_packed_f32: needs n_rows, C_in > 0 and 0 < C_out <= 64");
    DIF_REQUIRE(x && packed && bias && out, DIF_E_BADARG, "dif_linear_packed_f32: null pointer");
    DIF_REQUIRE((ln_weight == nullptr) == (ln_bias == nullptr), DIF_E_BADARG,
                "dif_linear_packed_f32: ln_weight and ln_bias must be given together");
    DIF_REQUIRE(ldx >= C_in && ldo >= C_out && dif::aligned16(packed), DIF_E_BADARG,
                "dif_linear_packed_f32: leading dimension smaller than a row, or packed not 16-byte aligned");
    DIF_REQUIRE((n_rows + 15) / 16 < (int64_t(1) << 31), DIF_E_RANGE, "dif_linear_packed_f32: too many rows");
    const int ov = (ldo % 4 == 0) && dif::aligned16(out);
    const int n_chunks = (C_in + 63) / 64;
    hipLaunchKernelGGL(long_linear_packed_kernel, dim3(static_cast<unsigned>((n_rows + 15) / 16)),
                       dim3(64 * (n_chunks < kPackedWaves ? n_chunks : kPackedWaves)), 0,
                       static_cast<hipStream_t>(stream), x, ldx, n_rows, C_in, static_cast<const bf16x8*>(packed), bias, C_out, ln_weight,
                       ln_bias, ln_eps, relu, out, ldo, ov);
    return dif::launch_status("long_linear_packed_kernel");
}

extern "C" int dif_linear_bf16(const void* x, int64_t ldx, int64_t n_rows, int C_in, const void* W, const void* bias,
                               int C_out, const void* ln_weight, const void* ln_bias, float ln_eps, int relu, void* out,
                               int64_t ldo, dif_stream_t stream) {
    using B = dif::bf16;
    return linear_entry<B>(static_cast<const B*>(x), ldx, n_rows, C_in, static_cast<const B*>(W),
                           static_cast<const B*>(bias), C_out, static_cast<const B*>(ln_weight),
                           static_cast<const B*>(ln_bias), ln_eps, relu, static_cast<B*>(out), ldo, stream);
}

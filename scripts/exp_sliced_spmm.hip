// Experiment (round 2): feature-sliced SpMM with the source tile in LDS.
//
// Every workgroup (one per CU) owns a PANEL of destination rows and a 16-byte feature SLICE (4 floats): a lane owns
// whole destination rows (static float4 accumulators in registers), the source rows of the current TILE (their
// 16-byte slices, prescaled) sit in LDS, and an entry is a 16-bit tile-local source index -> one random ds_read_b128.
// The entry lists are stored padded per (panel, tile, wave, round) as 1-KiB blocks of 8 entries x 64 lanes.
// Host builds the lists for a uniform random graph of the C4 shape directly in that form, in two orders:
//   order 0: as generated (random): ds_read_b128 bank conflicts ~3-way;
//   order 1: greedy edge colouring per hardware lane group: the 16 lanes that share an LDS cycle read 16 distinct bank quads.
// Build: hipcc --offload-arch=gfx950 -O3 -fopenmp -fno-slp-vectorize scripts/exp_sliced_spmm.hip -o scripts/bin/exp_sliced_spmm
#include <hip/hip_runtime.h>
#include <omp.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// lane sets that share one LDS cycle of a ds_read_b128 (MI355X_MICROARCH.md, LDS table)
static const int kGroupLanes[4][16] = {
    {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};

template <int R, int UNPACK>
__global__ __launch_bounds__(1024) void sliced_kernel(const uint4* __restrict__ ell, const int32_t* __restrict__ off,
                                                      const f32x4* __restrict__ ys, f32x4* __restrict__ out, int N,
                                                      int Npad, int T, int NT, int P, int W, int slices) {
    __shared__ f32x4 tile[10224];   // T rows + 16 zero rows (static: no base-address add per read)
    const uint32_t four = 4;
    const int b = blockIdx.x;
    const int xcd = b & 7, k = b >> 3;
    const int per = gridDim.x >> 3;
    const int panel = xcd * (per / slices) + k / slices;
    const int slice = k % slices;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc[R];
#pragma unroll
    for (int j = 0; j < R; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (threadIdx.x < 16) tile[T + threadIdx.x] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* ysl = ys + static_cast<size_t>(slice) * Npad;
    const int32_t* offw = off + (static_cast<size_t>(panel) * NT * W + w) * (R + 1);
    for (int t = 0; t < NT; ++t) {
        const int32_t* o = offw + static_cast<size_t>(t) * W * (R + 1);
        const int bend = o[R];
        int bcur = o[0];
        uint4 cur = uint4{0, 0, 0, 0}, nxt = uint4{0, 0, 0, 0};
        if (bcur < bend) cur = ell[static_cast<size_t>(bcur) * 64 + lane];
        if (bcur + 1 < bend) nxt = ell[static_cast<size_t>(bcur + 1) * 64 + lane];
        __syncthreads();
        for (int i = threadIdx.x; i < T; i += blockDim.x) tile[i] = ysl[static_cast<size_t>(t) * T + i];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int b1 = o[j + 1];
            f32x4 a = acc[j];
            for (; bcur < b1; ++bcur) {
                const uint4 e = cur;
                cur = nxt;
                if (bcur + 2 < bend) nxt = ell[static_cast<size_t>(bcur + 2) * 64 + lane];
                const uint32_t wds[4] = {e.x, e.y, e.z, e.w};
                f32x4 v[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t lo, hi;
                    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
                        : "=v"(lo) : "v"(four), "v"(wds[q]));
                    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
                        : "=v"(hi) : "v"(four), "v"(wds[q]));
                    v[2 * q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(tile) + lo);
                    v[2 * q + 1] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(tile) + hi);
                }
                a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
            acc[j] = a;
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int lrow = (j * W + w) * 64 + lane;
        const int64_t row = static_cast<int64_t>(panel) * P + lrow;
        if (lrow < P && row < N) out[row * slices + slice] = acc[j];
    }
}


// ---- v2: rounds interleaved block by block ([k][j] order, every round of a wave padded to the same block count per
// tile): the body of a super-step is NR blocks with static accumulator / entry registers, each entry register is
// reloaded right after its use (a whole super-step ahead of its next use, no register rotation -> vmcnt(NR-1) waits).
template <int NR, int MODE>
__device__ __forceinline__ void sweep2(const f32x4* tile_, const uint4* __restrict__ ell, const int2* __restrict__ tabw,
                                       const f32x4* __restrict__ ysl, f32x4* __restrict__ out, int N, int T, int NT, int P,
                                       int W, int slices, int panel, int slice, int w, int lane) {
    f32x4* tile = const_cast<f32x4*>(tile_);
    const uint32_t four = 4;
    f32x4 acc[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // issue the four LDS reads of one half block (two packed dwords = four 16-bit tile-local source rows)
    auto issue = [&](uint32_t w0, uint32_t w1, f32x4 (&v)[4]) {
        const uint32_t wds[2] = {w0, w1};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            uint32_t lo, hi;
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
                : "=v"(lo) : "v"(four), "v"(wds[q]));
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
                : "=v"(hi) : "v"(four), "v"(wds[q]));
            v[2 * q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(tile) + lo);
            v[2 * q + 1] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(tile) + hi);
        }
    };
    auto consume = [&](const f32x4 (&v)[4], f32x4& a) { a += (v[0] + v[1]) + (v[2] + v[3]); };
    for (int t = 0; t < NT; ++t) {
        const int2 tb = tabw[static_cast<size_t>(t) * W];
        const int start = __builtin_amdgcn_readfirstlane(tb.x), nb = __builtin_amdgcn_readfirstlane(tb.y);
        const uint4* base = ell + static_cast<size_t>(start) * 64 + lane;
        uint4 e[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) e[j] = base[j * 64];      // nb >= 1 always (builder pads with a bubble block)
        __syncthreads();
        if (MODE != 1 || t == 0) {
            const f32x4* src = ysl + static_cast<size_t>(t) * T;
            const int nth = blockDim.x;
            for (int b0 = threadIdx.x; b0 < T; b0 += 5 * nth) {     // batches of five 16-byte loads per thread
                f32x4 r[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int i = b0 + u * nth;
                    r[u] = i < T ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int i = b0 + u * nth;
                    if (i < T) tile[i] = r[u];
                }
            }
        }
        __syncthreads();
        if (MODE == 2) continue;
        // software pipeline over half blocks: the next half's four reads are issued before the current half's values
        // are added, so this wave always has requests in the LDS queue (closed-loop: 16 waves x 4 outstanding reads with
        // ~180 cycles of issue + add time between batches left the LDS array 59 % busy)
        f32x4 va[4], vb[4];
        issue(e[0].x, e[0].y, va);
#pragma unroll 1
        for (int k = 0; k + 1 < nb; ++k) {
            const uint4* nx = base + static_cast<size_t>(k + 1) * NR * 64;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                issue(e[j].z, e[j].w, vb);
                __builtin_amdgcn_sched_barrier(0);
                consume(va, acc[j]);
                e[j] = nx[j * 64];      // reloaded right after its last use; next use is a whole super-step away
                __builtin_amdgcn_sched_barrier(0);
                issue(e[(j + 1) % NR].x, e[(j + 1) % NR].y, va);
                __builtin_amdgcn_sched_barrier(0);
                consume(vb, acc[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            issue(e[j].z, e[j].w, vb);
            __builtin_amdgcn_sched_barrier(0);
            consume(va, acc[j]);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < NR) issue(e[j + 1].x, e[j + 1].y, va);
            __builtin_amdgcn_sched_barrier(0);
            consume(vb, acc[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int lrow = (j * W + w) * 64 + lane;
        const int64_t row = static_cast<int64_t>(panel) * P + lrow;
        if (lrow < P && row < N) out[row * slices + slice] = acc[j];
    }
}

template <int R, int MODE>
__global__ __launch_bounds__(1024) void sliced_kernel2(const uint4* __restrict__ ell, const int2* __restrict__ tab,
                                                       const f32x4* __restrict__ ys, f32x4* __restrict__ out, int N,
                                                       int Npad, int T, int NT, int P, int S, int W, int slices) {
    __shared__ f32x4 tile[10224];
    const int b = blockIdx.x;
    const int xcd = b & 7, k = b >> 3;
    const int per = gridDim.x >> 3;
    const int panel = xcd * (per / slices) + k / slices;
    const int slice = k % slices;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 16) tile[T + threadIdx.x] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* ysl = ys + static_cast<size_t>(slice) * Npad;
    const int2* tabw = tab + static_cast<size_t>(panel) * NT * W + w;
    const int nr = (S - w + W - 1) / W;     // slots s = j*W + w < S
    if (nr == R) sweep2<R, MODE>(tile, ell, tabw, ysl, out, N, T, NT, P, W, slices, panel, slice, w, lane);
    else sweep2<R - 1, MODE>(tile, ell, tabw, ysl, out, N, T, NT, P, W, slices, panel, slice, w, lane);
}

// ---- v4: v1 layout (every round keeps its own block count), rounds interleaved block by block with scalar
// predicates: no padding of the rounds to a common length.
template <int NR>
__device__ __forceinline__ void sweep4(const f32x4* tile_, const uint4* __restrict__ ell, const int32_t* __restrict__ offw,
                                       const f32x4* __restrict__ ysl, f32x4* __restrict__ out, int N, int T, int NT, int P,
                                       int W, int R, int slices, int panel, int slice, int w, int lane) {
    f32x4* tile = const_cast<f32x4*>(tile_);
    const uint32_t four = 4;
    f32x4 acc[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto half = [&](uint32_t w0, uint32_t w1, f32x4& a) {
        const uint32_t wds[2] = {w0, w1};
        f32x4 v[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            uint32_t lo, hi;
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
                : "=v"(lo) : "v"(four), "v"(wds[q]));
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
                : "=v"(hi) : "v"(four), "v"(wds[q]));
            v[2 * q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(tile) + lo);
            v[2 * q + 1] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(tile) + hi);
        }
        a += (v[0] + v[1]) + (v[2] + v[3]);
    };
    for (int t = 0; t < NT; ++t) {
        const int32_t* o = offw + static_cast<size_t>(t) * W * (R + 1);
        int ob[NR + 1];
#pragma unroll
        for (int j = 0; j <= NR; ++j) ob[j] = __builtin_amdgcn_readfirstlane(o[j]);
        int nbmax = 0;
        uint4 e[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int nbj = ob[j + 1] - ob[j];
            nbmax = nbj > nbmax ? nbj : nbmax;
            e[j] = uint4{0, 0, 0, 0};
            if (nbj > 0) e[j] = ell[static_cast<size_t>(ob[j]) * 64 + lane];
        }
        __syncthreads();
        {
            const f32x4* src = ysl + static_cast<size_t>(t) * T;
            const int nth = blockDim.x;
            for (int b0 = threadIdx.x; b0 < T; b0 += 5 * nth) {
                f32x4 r[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int i = b0 + u * nth;
                    r[u] = i < T ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int i = b0 + u * nth;
                    if (i < T) tile[i] = r[u];
                }
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < nbmax; ++k) {
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int nbj = ob[j + 1] - ob[j];
                if (k < nbj) {
                    half(e[j].x, e[j].y, acc[j]);
                    half(e[j].z, e[j].w, acc[j]);
                    if (k + 1 < nbj) e[j] = ell[static_cast<size_t>(ob[j] + k + 1) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int lrow = (j * W + w) * 64 + lane;
        const int64_t row = static_cast<int64_t>(panel) * P + lrow;
        if (lrow < P && row < N) out[row * slices + slice] = acc[j];
    }
}

template <int R>
__global__ __launch_bounds__(1024) void sliced_kernel4(const uint4* __restrict__ ell, const int32_t* __restrict__ off,
                                                       const f32x4* __restrict__ ys, f32x4* __restrict__ out, int N,
                                                       int Npad, int T, int NT, int P, int S, int W, int slices) {
    __shared__ f32x4 tile[10224];
    const int b = blockIdx.x;
    const int xcd = b & 7, k = b >> 3;
    const int per = gridDim.x >> 3;
    const int panel = xcd * (per / slices) + k / slices;
    const int slice = k % slices;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 16) tile[T + threadIdx.x] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* ysl = ys + static_cast<size_t>(slice) * Npad;
    const int32_t* offw = off + (static_cast<size_t>(panel) * NT * W + w) * (R + 1);
    const int nr = (S - w + W - 1) / W;
    if (nr == R) sweep4<R>(tile, ell, offw, ysl, out, N, T, NT, P, W, R, slices, panel, slice, w, lane);
    else sweep4<R - 1>(tile, ell, offw, ysl, out, N, T, NT, P, W, R, slices, panel, slice, w, lane);
}

// ---- v5: v2 layout with options: BSTEP = steps per block (8: uint4 per lane, 4: uint2 per lane), DMA = tile loaded by
// global_load_lds (no VGPR round trip), EXTRA = dummy packed FMAs per 4 steps (VALU headroom probe).
template <int NR, int BSTEP, bool DMA, int EXTRA>
__device__ __forceinline__ void sweep5(f32x4* tile, const uint32_t* __restrict__ ell, const int2* __restrict__ tabw,
                                       const f32x4* __restrict__ ysl, f32x4* __restrict__ out, int N, int T, int NT, int P,
                                       int W, int slices, int panel, int slice, int w, int lane) {
    constexpr int DW = BSTEP / 2;            // dwords per lane per block
    const uint32_t four = 4;
    f32x4 acc[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 dummy = f32x4{0.f, 0.f, 0.f, 0.f};
    auto half = [&](uint32_t w0, uint32_t w1, f32x4& a) {
        const uint32_t wds[2] = {w0, w1};
        f32x4 v[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            uint32_t lo, hi;
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
                : "=v"(lo) : "v"(four), "v"(wds[q]));
            asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
                : "=v"(hi) : "v"(four), "v"(wds[q]));
            v[2 * q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(tile) + lo);
            v[2 * q + 1] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(tile) + hi);
        }
        const f32x4 s = (v[0] + v[1]) + (v[2] + v[3]);
        a += s;
#pragma unroll
        for (int x = 0; x < EXTRA; ++x) dummy = dummy * s + s;       // 2 v_pk_fma_f32 each
    };
    for (int t = 0; t < NT; ++t) {
        const int2 tb = tabw[static_cast<size_t>(t) * W];
        const int start = __builtin_amdgcn_readfirstlane(tb.x), nb = __builtin_amdgcn_readfirstlane(tb.y);
        const uint32_t* base = ell + (static_cast<size_t>(start) * 64 + lane) * DW;
        uint32_t e[NR][DW];
        auto load = [&](uint32_t (&d)[DW], const uint32_t* p) {
            if constexpr (DW == 4) { const uint4 r = *reinterpret_cast<const uint4*>(p); d[0] = r.x; d[1] = r.y; d[2] = r.z; d[3] = r.w; }
            else { const uint2 r = *reinterpret_cast<const uint2*>(p); d[0] = r.x; d[1] = r.y; }
        };
#pragma unroll
        for (int j = 0; j < NR; ++j) load(e[j], base + static_cast<size_t>(j) * 64 * DW);
        __syncthreads();
        {
            const f32x4* src = ysl + static_cast<size_t>(t) * T;
            if (DMA) {
                for (int b0 = w * 64; b0 < T; b0 += W * 64) {
                    const int i = b0 + lane;
                    __builtin_amdgcn_global_load_lds(src + (i < T ? i : T - 1), tile + b0, 16, 0, 0);
                }
            } else {
                const int nth = blockDim.x;
                for (int b0 = threadIdx.x; b0 < T; b0 += 5 * nth) {
                    f32x4 r[5];
#pragma unroll
                    for (int u = 0; u < 5; ++u) { const int i = b0 + u * nth; r[u] = i < T ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                    for (int u = 0; u < 5; ++u) { const int i = b0 + u * nth; if (i < T) tile[i] = r[u]; }
                }
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k + 1 < nb; ++k) {
            const uint32_t* nx = base + static_cast<size_t>(k + 1) * NR * 64 * DW;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                half(e[j][0], e[j][1], acc[j]);
                if constexpr (DW == 4) half(e[j][2], e[j][3], acc[j]);
                load(e[j], nx + static_cast<size_t>(j) * 64 * DW);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            half(e[j][0], e[j][1], acc[j]);
            if constexpr (DW == 4) half(e[j][2], e[j][3], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int lrow = (j * W + w) * 64 + lane;
        const int64_t row = static_cast<int64_t>(panel) * P + lrow;
        if (EXTRA > 0) acc[j] += dummy * 1e-30f;
        if (lrow < P && row < N) out[row * slices + slice] = acc[j];
    }
}

template <int R, int BSTEP, bool DMA, int EXTRA>
__global__ __launch_bounds__(1024) void sliced_kernel5(const uint32_t* __restrict__ ell, const int2* __restrict__ tab,
                                                       const f32x4* __restrict__ ys, f32x4* __restrict__ out, int N,
                                                       int Npad, int T, int NT, int P, int S, int W, int slices) {
    __shared__ f32x4 tile[10224];
    const int b = blockIdx.x;
    const int xcd = b & 7, k = b >> 3;
    const int per = gridDim.x >> 3;
    const int panel = xcd * (per / slices) + k / slices;
    const int slice = k % slices;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 16) tile[T + threadIdx.x] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* ysl = ys + static_cast<size_t>(slice) * Npad;
    const int2* tabw = tab + static_cast<size_t>(panel) * NT * W + w;
    const int nr = (S - w + W - 1) / W;
    if (nr == R) sweep5<R, BSTEP, DMA, EXTRA>(tile, ell, tabw, ysl, out, N, T, NT, P, W, slices, panel, slice, w, lane);
    else sweep5<R - 1, BSTEP, DMA, EXTRA>(tile, ell, tabw, ysl, out, N, T, NT, P, W, slices, panel, slice, w, lane);
}

struct Fmt {
    int N, F, slices, panels, P, S, W, R, T, NT, Npad;
    std::vector<uint16_t> ell;      // blocks of 64 lanes x 8 entries
    std::vector<int32_t> off;       // v1: [panels][NT][W][R+1], block units; v2: [panels][NT][W] {start, nb}
    int v2 = 0;
    int bstep = 8;      // steps per block
    int64_t nnz = 0, steps = 0;
};

// order one hardware lane group's lists (16 lanes) for `len` steps; sched[l][k] = index or bubble (T + free quad)
static void schedule_group(std::vector<uint16_t> lists[16], int T, int order, std::vector<uint16_t> sched[16]) {
    int maxc = 0;
    for (int l = 0; l < 16; ++l) maxc = std::max<int>(maxc, lists[l].size());
    if (order == 0) {
        for (int l = 0; l < 16; ++l) {
            sched[l] = lists[l];
            sched[l].resize(maxc, static_cast<uint16_t>(T + l));
        }
        return;
    }
    // greedy colouring: per step, lanes by remaining count (desc) pick a free bank quad, preferring the quad with the
    // largest remaining demand over all lanes
    std::vector<uint16_t> bucket[16][16];
    int rem[16] = {0}, col[16] = {0};
    for (int l = 0; l < 16; ++l)
        for (uint16_t v : lists[l]) {
            bucket[l][v & 15].push_back(v);
            ++rem[l];
            ++col[v & 15];
        }
    int left = 0;
    for (int l = 0; l < 16; ++l) left += rem[l];
    for (int l = 0; l < 16; ++l) sched[l].clear();
    while (left > 0) {
        int ord[16];
        for (int l = 0; l < 16; ++l) ord[l] = l;
        std::stable_sort(ord, ord + 16, [&](int a, int b2) { return rem[a] > rem[b2]; });
        bool used[16] = {false};
        int choice[16];
        for (int i = 0; i < 16; ++i) {
            const int l = ord[i];
            choice[l] = -1;
            if (rem[l] == 0) continue;
            int best = -1;
            for (int q = 0; q < 16; ++q)
                if (!used[q] && !bucket[l][q].empty() && (best < 0 || col[q] > col[best])) best = q;
            if (best >= 0) { used[best] = true; choice[l] = best; }
        }
        int fq = 0;
        for (int l = 0; l < 16; ++l) {
            if (choice[l] >= 0) {
                const int q = choice[l];
                sched[l].push_back(bucket[l][q].back());
                bucket[l][q].pop_back();
                --rem[l]; --col[q]; --left;
            } else {
                while (used[fq]) ++fq;
                used[fq] = true;
                sched[l].push_back(static_cast<uint16_t>(T + fq));
            }
        }
    }
}

static void build(Fmt& f, int N, int F, double deg, int W, int R, int order, uint64_t seed) {
    f.N = N; f.F = F; f.slices = F / 4; f.panels = 256 / f.slices;
    f.P = (N + f.panels - 1) / f.panels;
    f.S = (f.P + 63) / 64;
    f.W = W; f.R = R;
    if (W * R < f.S) { fprintf(stderr, "W*R < S (%d)\n", f.S); exit(1); }
    const int maxT = (160 * 1024) / 16 - 16;
    f.NT = (N + maxT - 1) / maxT;
    f.T = (((N + f.NT - 1) / f.NT) + 15) / 16 * 16;
    f.Npad = f.T * f.NT;
    const int nchunks = f.panels * f.NT * W * R;
    std::vector<std::vector<uint16_t>> chunks(nchunks);
    std::vector<int32_t> nblk(nchunks, 0);
    int64_t nnz = 0, steps = 0;
    const double lam = deg * f.T / N;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : nnz, steps)
    for (int c = 0; c < nchunks; ++c) {
        const int j = c % R, w = (c / R) % W, t = (c / R / W) % f.NT, p = c / R / W / f.NT;
        const int s = j * W + w;
        std::mt19937_64 rng(seed * 1000003ull + c);
        std::poisson_distribution<int> pois(lam);
        std::vector<uint16_t> lists[64];
        int any = 0;
        for (int l = 0; l < 64; ++l) {
            const int lrow = s * 64 + l;
            const int64_t row = static_cast<int64_t>(p) * f.P + lrow;
            if (s >= f.S || lrow >= f.P || row >= N) continue;
            const int cnt = pois(rng);
            const int tlen = std::min(f.T, N - t * f.T);
            for (int i = 0; i < cnt; ++i) lists[l].push_back(static_cast<uint16_t>(rng() % tlen));
            any += cnt;
        }
        nnz += any;
        if (!any) continue;
        std::vector<uint16_t> sched[64];
        int len = 0;
        for (int g = 0; g < 4; ++g) {
            std::vector<uint16_t> gl[16], gs[16];
            for (int i = 0; i < 16; ++i) gl[i] = lists[kGroupLanes[g][i]];
            schedule_group(gl, f.T, order, gs);
            for (int i = 0; i < 16; ++i) sched[kGroupLanes[g][i]] = gs[i];
            len = std::max<int>(len, gs[0].size());
        }
        const int BS = f.bstep;
        const int nb = (len + BS - 1) / BS;
        nblk[c] = nb;
        steps += nb * BS;
        std::vector<uint16_t>& out = chunks[c];
        out.resize(static_cast<size_t>(nb) * 64 * BS);
        for (int g = 0; g < 4; ++g)
            for (int i = 0; i < 16; ++i) {
                const int l = kGroupLanes[g][i];
                for (int k2 = 0; k2 < nb * BS; ++k2) {
                    const uint16_t v = k2 < static_cast<int>(sched[l].size()) ? sched[l][k2] : static_cast<uint16_t>(f.T + i);
                    out[(static_cast<size_t>(k2 / BS) * 64 + l) * BS + (k2 % BS)] = v;
                }
            }
    }
    f.nnz = nnz; f.steps = steps;
    // bubble block: lane l reads zero row T + (its position inside its hardware lane group)
    uint16_t bub[512];
    const int BSZ = 64 * f.bstep;        // entries per block
    for (int g = 0; g < 4; ++g)
        for (int i = 0; i < 16; ++i)
            for (int k2 = 0; k2 < f.bstep; ++k2) bub[kGroupLanes[g][i] * f.bstep + k2] = static_cast<uint16_t>(f.T + i);
    if (f.v2) {
        // [p][t][w] -> {start, nb}; blocks in [k][j] order, every round of the wave padded to nb blocks
        f.off.assign(static_cast<size_t>(f.panels) * f.NT * W * 2, 0);
        int64_t tot = 0;
        steps = 0;
        for (int p = 0; p < f.panels; ++p)
            for (int t = 0; t < f.NT; ++t)
                for (int w = 0; w < W; ++w) {
                    const int nr = (f.S - w + W - 1) / W;
                    int nb = 0;
                    for (int j = 0; j < nr; ++j) nb = std::max(nb, nblk[((p * f.NT + t) * W + w) * R + j]);
                    int32_t* o = &f.off[((static_cast<size_t>(p) * f.NT + t) * W + w) * 2];
                    o[0] = static_cast<int32_t>(tot);
                    if (nb < 1) nb = 1;
                    o[1] = nb;
                    tot += static_cast<int64_t>(nb) * nr;
                    steps += static_cast<int64_t>(nb) * nr * f.bstep;
                }
        f.steps = steps;
        f.ell.resize(static_cast<size_t>(tot) * BSZ);
#pragma omp parallel for schedule(dynamic, 16)
        for (int c = 0; c < f.panels * f.NT * W; ++c) {
            const int w = c % W;
            const int nr = (f.S - w + W - 1) / W;
            const int32_t start = f.off[static_cast<size_t>(c) * 2], nb = f.off[static_cast<size_t>(c) * 2 + 1];
            for (int j = 0; j < nr; ++j) {
                const std::vector<uint16_t>& ch = chunks[c * R + j];
                const int have = nblk[c * R + j];
                for (int k2 = 0; k2 < nb; ++k2) {
                    uint16_t* dst = &f.ell[(static_cast<size_t>(start) + static_cast<size_t>(k2) * nr + j) * BSZ];
                    if (k2 < have) memcpy(dst, &ch[static_cast<size_t>(k2) * BSZ], BSZ * 2);
                    else memcpy(dst, bub, BSZ * 2);
                }
            }
        }
        return;
    }
    // v1 offsets: [p][t][w][R+1]
    f.off.assign(static_cast<size_t>(f.panels) * f.NT * W * (R + 1), 0);
    int64_t tot = 0;
    for (int p = 0; p < f.panels; ++p)
        for (int t = 0; t < f.NT; ++t)
            for (int w = 0; w < W; ++w) {
                int32_t* o = &f.off[((static_cast<size_t>(p) * f.NT + t) * W + w) * (R + 1)];
                for (int j = 0; j < R; ++j) {
                    o[j] = static_cast<int32_t>(tot);
                    tot += nblk[((p * f.NT + t) * W + w) * R + j];
                }
                o[R] = static_cast<int32_t>(tot);
            }
    f.ell.resize(static_cast<size_t>(tot) * 512);
#pragma omp parallel for schedule(dynamic, 64)
    for (int c = 0; c < nchunks; ++c) {
        if (!nblk[c]) continue;
        const int j = c % R, w = (c / R) % W, t = (c / R / W) % f.NT, p = c / R / W / f.NT;
        const int32_t o = f.off[((static_cast<size_t>(p) * f.NT + t) * W + w) * (R + 1) + j];
        memcpy(&f.ell[static_cast<size_t>(o) * 512], chunks[c].data(), chunks[c].size() * 2);
    }
}

template <int R, int MODE>
static float run2(const Fmt& f, const uint16_t* d_ell, const int32_t* d_off, const float* d_ys, float* d_out, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto go = [&]() {
        hipLaunchKernelGGL((sliced_kernel2<R, MODE>), dim3(256), dim3(f.W * 64), 0, 0, reinterpret_cast<const uint4*>(d_ell),
                           reinterpret_cast<const int2*>(d_off), reinterpret_cast<const f32x4*>(d_ys),
                           reinterpret_cast<f32x4*>(d_out), f.N, f.Npad, f.T, f.NT, f.P, f.S, f.W, f.slices);
    };
    for (int i = 0; i < 3; ++i) go();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) go();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int R>
static float run(const Fmt& f, const uint16_t* d_ell, const int32_t* d_off, const float* d_ys, float* d_out, int iters) {
    const size_t lds = 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((sliced_kernel<R, 0>), dim3(256), dim3(f.W * 64), lds, 0, reinterpret_cast<const uint4*>(d_ell), d_off,
                           reinterpret_cast<const f32x4*>(d_ys), reinterpret_cast<f32x4*>(d_out), f.N, f.Npad, f.T, f.NT,
                           f.P, f.W, f.slices);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL((sliced_kernel<R, 0>), dim3(256), dim3(f.W * 64), lds, 0, reinterpret_cast<const uint4*>(d_ell), d_off,
                           reinterpret_cast<const f32x4*>(d_ys), reinterpret_cast<f32x4*>(d_out), f.N, f.Npad, f.T, f.NT,
                           f.P, f.W, f.slices);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int R>
static float run4(const Fmt& f, const uint16_t* d_ell, const int32_t* d_off, const float* d_ys, float* d_out, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto go = [&]() {
        hipLaunchKernelGGL((sliced_kernel4<R>), dim3(256), dim3(f.W * 64), 0, 0, reinterpret_cast<const uint4*>(d_ell),
                           d_off, reinterpret_cast<const f32x4*>(d_ys), reinterpret_cast<f32x4*>(d_out), f.N, f.Npad, f.T,
                           f.NT, f.P, f.S, f.W, f.slices);
    };
    for (int i = 0; i < 3; ++i) go();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) go();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int R, int BSTEP, bool DMA, int EXTRA>
static float run5(const Fmt& f, const uint16_t* d_ell, const int32_t* d_off, const float* d_ys, float* d_out, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto go = [&]() {
        hipLaunchKernelGGL((sliced_kernel5<R, BSTEP, DMA, EXTRA>), dim3(256), dim3(f.W * 64), 0, 0,
                           reinterpret_cast<const uint32_t*>(d_ell), reinterpret_cast<const int2*>(d_off),
                           reinterpret_cast<const f32x4*>(d_ys), reinterpret_cast<f32x4*>(d_out), f.N, f.Npad, f.T, f.NT, f.P,
                           f.S, f.W, f.slices);
    };
    for (int i = 0; i < 3; ++i) go();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) go();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 132534;
    const double deg = argc > 2 ? atof(argv[2]) : 598.0;
    const int only = argc > 3 ? atoi(argv[3]) : -1;     // run a single configuration (for profiling)
    const int F = 64;
    // v5 variants on the equalised [k][j] layout, W = 15 waves x 9 rounds (the product's geometry at C4)
    struct Cfg { int bstep, dma, extra; const char* what; } cfgs[] = {
        {8, 0, 0, "8-step blocks, register-staged tile (the product kernel)"},
        {8, 1, 0, "8-step blocks, tile by global_load_lds"},
        {4, 0, 0, "4-step blocks, register-staged tile"},
        {4, 1, 0, "4-step blocks, tile by global_load_lds"},
        {8, 0, 1, "8-step, +2 dummy v_pk_fma per 4 steps"},
        {8, 0, 2, "8-step, +4 dummy v_pk_fma per 4 steps"},
        {8, 0, 4, "8-step, +8 dummy v_pk_fma per 4 steps"},
    };
    std::vector<float> ys;
    int ci = -1;
    Fmt f8, f4;
    for (const Cfg& c : cfgs) {
        ++ci;
        if (only >= 0 && ci != only) continue;
        Fmt& f = c.bstep == 8 ? f8 : f4;
        if (f.ell.empty()) {
            f.v2 = 1;
            f.bstep = c.bstep;
            double t0 = omp_get_wtime();
            build(f, N, F, deg, 15, 9, 1, 7);
            printf("format bstep=%d: T=%d NT=%d P=%d S=%d nnz=%lld padded steps x64=%lld (x%.3f) %.1f MB, host build %.1fs\n", c.bstep,
                   f.T, f.NT, f.P, f.S, (long long)f.nnz, (long long)f.steps * 64, f.steps * 64.0 / f.nnz, f.ell.size() * 2 / 1e6,
                   omp_get_wtime() - t0);
        }
        ys.resize(static_cast<size_t>(f.slices) * f.Npad * 4);
        std::mt19937 rng(3);
        std::uniform_real_distribution<float> u(-1.f, 1.f);
        for (size_t i2 = 0; i2 < ys.size(); ++i2) ys[i2] = u(rng);
        for (int s2 = 0; s2 < f.slices; ++s2)
            for (int r = N; r < f.Npad; ++r)
                for (int q = 0; q < 4; ++q) ys[(static_cast<size_t>(s2) * f.Npad + r) * 4 + q] = 0.f;
        uint16_t* d_ell; int32_t* d_off; float *d_ys, *d_out;
        CK(hipMalloc(&d_ell, f.ell.size() * 2));
        CK(hipMalloc(&d_off, f.off.size() * 4));
        CK(hipMalloc(&d_ys, ys.size() * 4));
        CK(hipMalloc(&d_out, static_cast<size_t>(N) * F * 4));
        CK(hipMemcpy(d_ell, f.ell.data(), f.ell.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_off, f.off.data(), f.off.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_ys, ys.data(), ys.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(d_out, 0, static_cast<size_t>(N) * F * 4));
        float ms = 0;
        const int it = 30;
        if (c.bstep == 8 && !c.dma && c.extra == 0) ms = run5<9, 8, false, 0>(f, d_ell, d_off, d_ys, d_out, it);
        else if (c.bstep == 8 && c.dma) ms = run5<9, 8, true, 0>(f, d_ell, d_off, d_ys, d_out, it);
        else if (c.bstep == 4 && !c.dma) ms = run5<9, 4, false, 0>(f, d_ell, d_off, d_ys, d_out, it);
        else if (c.bstep == 4 && c.dma) ms = run5<9, 4, true, 0>(f, d_ell, d_off, d_ys, d_out, it);
        else if (c.extra == 1) ms = run5<9, 8, false, 1>(f, d_ell, d_off, d_ys, d_out, it);
        else if (c.extra == 2) ms = run5<9, 8, false, 2>(f, d_ell, d_off, d_ys, d_out, it);
        else ms = run5<9, 8, false, 4>(f, d_ell, d_off, d_ys, d_out, it);
        printf("[%d] %-62s %.4f ms  (702 MB -> %.0f GB/s)\n", ci, c.what, ms, 702.4e6 / ms / 1e6);
        std::vector<float> out(static_cast<size_t>(N) * F);
        CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0;
        const int BS = f.bstep;
        for (int trial = 0; trial < 40; ++trial) {
            const int p = trial % f.panels, s2 = (trial * 7) % f.S, l = (trial * 13) % 64, sl = (trial * 5) % f.slices;
            const int j = s2 / f.W, w = s2 % f.W;
            const int nr = (f.S - w + f.W - 1) / f.W;
            const int64_t row = static_cast<int64_t>(p) * f.P + s2 * 64 + l;
            if (s2 * 64 + l >= f.P || row >= N) continue;
            double ref[4] = {0, 0, 0, 0};
            for (int t = 0; t < f.NT; ++t) {
                const int32_t* o = &f.off[((static_cast<size_t>(p) * f.NT + t) * f.W + w) * 2];
                for (int k2 = 0; k2 < o[1]; ++k2) {
                    const int64_t b = static_cast<int64_t>(o[0]) + static_cast<int64_t>(k2) * nr + j;
                    for (int k3 = 0; k3 < BS; ++k3) {
                        const int v = f.ell[(static_cast<size_t>(b) * 64 + l) * BS + k3];
                        if (v >= f.T) continue;
                        for (int q = 0; q < 4; ++q) ref[q] += ys[(static_cast<size_t>(sl) * f.Npad + t * f.T + v) * 4 + q];
                    }
                }
            }
            for (int q = 0; q < 4; ++q) maxerr = std::max(maxerr, std::abs(ref[q] - out[row * F + sl * 4 + q]));
        }
        printf("      max abs err on sampled rows %.3e\n", maxerr);
        CK(hipFree(d_ell)); CK(hipFree(d_off)); CK(hipFree(d_ys)); CK(hipFree(d_out));
    }
    return 0;
}

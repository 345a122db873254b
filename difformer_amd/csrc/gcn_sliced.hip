// a3 (hot half, dense graphs): feature-sliced normalised-adjacency product with the SOURCE rows staged in LDS
//   out[r,:] = gcn_scale * dinv[r] * sum_{e in CSR row r} (dinv[src_e] * x[src_e,:])   (+ attn_scale * attn[r,:])
// node classification/difformer.py:63-79 for edge_weight = None: value_e = dinv[col] * dinv[row] factors into a
// pre-scaled source row and a per-destination scale, so an entry is nothing but a source index.
//
// Why: the round-1 kernel gathers a 256-byte row per entry through L1/L2 and is bound by the L2->L1 request rate
// (profiles/r02_pmc_spmm_blocked.json: 174 M requests per launch at C4, 154 G/s, 1.13 ms).  LDS serves random 16-byte
// reads an order of magnitude faster, so:
//   * a workgroup (one per CU) owns a PANEL of destination rows and one 16-byte SLICE (4 floats) of the feature row;
//     16 slices x 16 panels fill the chip at F = 64.  A lane owns whole destination rows: round j of wave w holds rows
//     (j*W + w)*64 + lane of the panel in a float4 register accumulator -- static registers, no LDS read-modify-write;
//   * the sources are swept tile by tile: the slice of T <= 10,208 pre-scaled source rows sits in LDS (all 160 KiB);
//     an entry is a 16-bit tile-local row number -> ONE ds_read_b128 + one float4 add per entry and lane;
//   * the lists are stored per (panel, tile, wave) as 1-KiB blocks of 8 entries x 64 lanes, rounds interleaved block
//     by block and padded to a common length (lock-step lanes), so the body is branch-free and every entry register
//     is reloaded right after its last use, a whole super-step ahead of its next one;
//   * inside a (tile, round) the entries of the 16 lanes that share an LDS cycle of ds_read_b128 are scheduled at build
//     time (greedy edge colouring) so that they hit 16 different bank quads: no bank conflicts (SQ_LDS_BANK_CONFLICT
//     = 0); idle slots read one of 16 zero rows, also on a free quad.  Random order costs 1.6x.
// The same adjacency slice is swept by the 16 slice-workgroups of a panel, which sit on one XCD (block b -> XCD b % 8),
// so the entry stream comes from HBM once.
// Measured at C4 (79.3 M entries): 0.36 ms per launch against 1.12 ms (profiles/r02_experiments.md).
#include "dif_common.h"

namespace {

using dif::f32x4;

constexpr int kTileRowsMax = 10208;              // + 16 zero rows = 10,224 rows x 16 B = 163,584 B of LDS
constexpr int kLdsRows = kTileRowsMax + 16;
constexpr int kMaxRounds = 10;                   // destination rows per lane (float4 accumulator + entry registers each)
constexpr int kMaxWaves = 16;
constexpr int kGroupCap = 255;                   // entries of one (row, tile) group (byte counters)

// lane sets that share one LDS cycle of a ds_read_b128 (MI355X_MICROARCH.md, LDS table): position -> lane
__device__ const uint8_t kLaneOf[64] = {
    0,  1,  2,  3,  12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27,
    4,  5,  6,  7,  8,  9,  10, 11, 16, 17, 18, 19, 28, 29, 30, 31,
    32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59,
    36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63};

struct Plan {
    int slices, panels, P, S, W, R, T, NT;
};

// Geometry for n_rows destination rows (a shard) over n_src source rows at F feature columns.
int make_plan(int64_t n_src, int64_t n_rows, int F, Plan& p) {
    if (n_src <= 0 || n_rows <= 0 || F <= 0 || F % 4 != 0 || F / 4 > 256) return DIF_E_SHAPE;
    p.slices = F / 4;
    const int64_t cap = static_cast<int64_t>(kMaxWaves) * kMaxRounds * 64;      // rows one workgroup can own
    int64_t panels = dif::kCUs / p.slices;
    if (panels < 1) panels = 1;
    if (panels * cap < n_rows) panels = (n_rows + cap - 1) / cap;
    if (panels > n_rows) panels = n_rows;
    if (panels * p.slices > (int64_t(1) << 20)) return DIF_E_RANGE;
    p.panels = static_cast<int>(panels);
    p.P = static_cast<int>((n_rows + panels - 1) / panels);
    p.S = (p.P + 63) / 64;
    // fewest rounds first (they run back to back), then the fewest idle (wave, round) slots
    int bestW = 0, bestR = 1 << 30, bestWaste = 1 << 30;
    for (int W = kMaxWaves; W >= 1; --W) {
        if (W > p.S) continue;
        const int R = (p.S + W - 1) / W;
        const int waste = R * W - p.S;
        if (R < bestR || (R == bestR && waste < bestWaste)) { bestW = W; bestR = R; bestWaste = waste; }
    }
    if (bestR > kMaxRounds) return DIF_E_SHAPE;
    p.W = bestW;
    p.R = bestR;
    const int64_t nt = (n_src + kTileRowsMax - 1) / kTileRowsMax;
    if (nt > 32767) return DIF_E_RANGE;
    p.NT = static_cast<int>(nt);
    p.T = static_cast<int>((((n_src + nt - 1) / nt) + 15) / 16 * 16);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// build, step 1: one thread per (row, tile) group: entries re-ordered by bank quad (source row mod 16) as 16-bit
// tile-local row numbers, plus the 16 byte counters.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void group_bounds(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ blkptr,
                                             int64_t n_src, int NT, int64_t row, int t, int32_t& e0, int32_t& e1) {
    if (NT == 1) { e0 = rowptr[row]; e1 = rowptr[row + 1]; }
    else { e0 = blkptr[static_cast<int64_t>(t) * n_src + row]; e1 = blkptr[static_cast<int64_t>(t + 1) * n_src + row]; }
}

__global__ __launch_bounds__(256) void sliced_sort_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ blkptr,
                                                          const int32_t* __restrict__ src, int64_t n_src, int NT, int T,
                                                          int64_t row_begin, int64_t n_rows, uint16_t* __restrict__ srt,
                                                          uint2* __restrict__ cnt, int32_t* __restrict__ status) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= n_rows * NT) return;
    const int64_t lrow = idx / NT;
    const int t = static_cast<int>(idx % NT);
    int32_t e0, e1;
    group_bounds(rowptr, blkptr, n_src, NT, row_begin + lrow, t, e0, e1);
    const int32_t base = t * T;
    if (e1 - e0 > kGroupCap) {       // byte counters: such a graph takes the gather kernel
        atomicOr(status, 1);
        cnt[idx * 2] = uint2{0, 0};
        cnt[idx * 2 + 1] = uint2{0, 0};
        return;
    }
    uint64_t c0 = 0, c1 = 0;          // 16 byte counters
    for (int32_t e = e0; e < e1; ++e) {
        const uint32_t q = static_cast<uint32_t>(src[e] - base) & 15u;
        const uint64_t one = 1ull << ((q & 7u) * 8u);
        c0 += (q < 8u) ? one : 0ull;
        c1 += (q < 8u) ? 0ull : one;
    }
    cnt[idx * 2] = uint2{static_cast<uint32_t>(c0), static_cast<uint32_t>(c0 >> 32)};
    cnt[idx * 2 + 1] = uint2{static_cast<uint32_t>(c1), static_cast<uint32_t>(c1 >> 32)};
    // exclusive starts: byte k of c * 0x0101.. = sum of bytes 0..k (no carries: the total is <= 255)
    const uint64_t ones = 0x0101010101010101ull;
    const uint64_t i0 = c0 * ones;
    uint64_t n0 = i0 - c0;
    uint64_t n1 = c1 * ones - c1 + (i0 >> 56) * ones;
    for (int32_t e = e0; e < e1; ++e) {
        const uint32_t loc = static_cast<uint32_t>(src[e] - base);
        const uint32_t q = loc & 15u, sh = (q & 7u) * 8u;
        const uint32_t pos = static_cast<uint32_t>(((q < 8u) ? n0 : n1) >> sh) & 0xffu;
        srt[e0 + pos] = static_cast<uint16_t>(loc);
        const uint64_t one = 1ull << sh;
        n0 += (q < 8u) ? one : 0ull;
        n1 += (q < 8u) ? 0ull : one;
    }
}

// ------------------------------------------------------------------------------------------------------------
// build, step 2: greedy edge colouring, one THREAD per (panel, tile, slot, hardware lane group).  A step serves the
// 16 lanes in rotating order; a lane takes the bank quad it still has the most entries on among the quads no earlier
// lane of the step took (within ~0.2 % of the lower bound max(longest lane, fullest quad)).  EMIT = false only counts
// the steps; EMIT = true writes the schedule (and zero-row reads on the free quads for idle lanes) into the blocks.
// ------------------------------------------------------------------------------------------------------------
constexpr int kColorThreads = 128;
constexpr int kColorWords = 64 + 16;         // per thread in LDS: rem[16 lanes][16 quads] bytes + first entry of each lane

template <bool EMIT>
__global__ __launch_bounds__(kColorThreads) void sliced_color_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ blkptr, int64_t n_src, int64_t row_begin,
    int64_t n_rows, Plan pl, const uint16_t* __restrict__ srt, const uint2* __restrict__ cnt, int32_t* __restrict__ len,
    const int32_t* __restrict__ tab, uint16_t* __restrict__ ell) {
    __shared__ uint32_t sm[kColorWords * kColorThreads];
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * kColorThreads + threadIdx.x;
    const int64_t n_groups = static_cast<int64_t>(pl.panels) * pl.NT * pl.S * 4;
    if (gid >= n_groups) return;
    const int g = static_cast<int>(gid & 3);
    const int64_t c = gid >> 2;
    const int s = static_cast<int>(c % pl.S);
    const int t = static_cast<int>((c / pl.S) % pl.NT);
    const int p = static_cast<int>(c / pl.S / pl.NT);
    uint32_t* my = sm + threadIdx.x;             // word k of this thread: my[k * kColorThreads]
    int remaining = 0;
    for (int i = 0; i < 16; ++i) {
        const int lane = kLaneOf[g * 16 + i];
        const int64_t prow = static_cast<int64_t>(s) * 64 + lane;         // row inside the panel
        const int64_t lrow = static_cast<int64_t>(p) * pl.P + prow;       // row inside the shard
        uint2 a = uint2{0, 0}, b = uint2{0, 0};
        int32_t e0 = 0, e1 = 0;
        if (prow < pl.P && lrow < n_rows) {
            a = cnt[(lrow * pl.NT + t) * 2];
            b = cnt[(lrow * pl.NT + t) * 2 + 1];
            group_bounds(rowptr, blkptr, n_src, pl.NT, row_begin + lrow, t, e0, e1);
            if (e1 - e0 > kGroupCap) e1 = e0;    // flagged by the sort kernel; keep the walk bounded
        }
        my[(i * 4 + 0) * kColorThreads] = a.x;
        my[(i * 4 + 1) * kColorThreads] = a.y;
        my[(i * 4 + 2) * kColorThreads] = b.x;
        my[(i * 4 + 3) * kColorThreads] = b.y;
        my[(64 + i) * kColorThreads] = static_cast<uint32_t>(e0);
        remaining += e1 - e0;
    }
    // where this slot's blocks go
    const int w = s % pl.W, j = s / pl.W;
    const int nr = (pl.S - w + pl.W - 1) / pl.W;
    int64_t blk0 = 0;
    int nb = 0;
    if (EMIT) {
        const int32_t* tb = tab + (static_cast<int64_t>(p) * pl.NT + t) * pl.W * 2 + w * 2;
        blk0 = tb[0];
        nb = tb[1];
    }
    auto slot_addr = [&](int step, int lane) -> int64_t {
        return ((blk0 + static_cast<int64_t>(step >> 3) * nr + j) * 64 + lane) * 8 + (step & 7);
    };
    int step = 0;
    while (remaining > 0) {
        uint32_t used = 0, picked = 0;
        for (int ii = 0; ii < 16; ++ii) {
            const int i = (ii + step) & 15;
            int best = -1;
            uint32_t bestv = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t wd = my[(i * 4 + k) * kColorThreads];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t v = (wd >> (8 * b)) & 0xffu;
                    const int q = k * 4 + b;
                    if (v > bestv && !((used >> q) & 1u)) { bestv = v; best = q; }
                }
            }
            if (best < 0) continue;
            used |= 1u << best;
            picked |= 1u << i;
            my[(i * 4 + (best >> 2)) * kColorThreads] -= 1u << (8 * (best & 3));
            --remaining;
            if (EMIT) {
                // pop from the end of the lane's quad bucket: entry e0 + (entries of lower quads) + (left on this quad)
                const int lane = kLaneOf[g * 16 + i];
                const int64_t lrow = static_cast<int64_t>(p) * pl.P + static_cast<int64_t>(s) * 64 + lane;
                const uint2 a = cnt[(lrow * pl.NT + t) * 2], b2 = cnt[(lrow * pl.NT + t) * 2 + 1];
                const uint64_t c0 = static_cast<uint64_t>(a.x) | (static_cast<uint64_t>(a.y) << 32);
                const uint64_t c1 = static_cast<uint64_t>(b2.x) | (static_cast<uint64_t>(b2.y) << 32);
                const uint64_t ones = 0x0101010101010101ull;
                const uint64_t i0 = c0 * ones;
                const uint64_t x0 = i0 - c0, x1 = c1 * ones - c1 + (i0 >> 56) * ones;
                const uint32_t start = static_cast<uint32_t>(((best < 8) ? x0 : x1) >> ((best & 7) * 8)) & 0xffu;
                const uint32_t e0 = my[(64 + i) * kColorThreads];
                ell[slot_addr(step, lane)] = srt[e0 + start + (bestv - 1)];
            }
        }
        if (EMIT) {             // idle lanes read a zero row on a quad nobody uses in this step
            for (int i = 0; i < 16; ++i) {
                if ((picked >> i) & 1u) continue;
                const int fq = __builtin_ctz(~used & 0xffffu);
                used |= 1u << fq;
                ell[slot_addr(step, kLaneOf[g * 16 + i])] = static_cast<uint16_t>(pl.T + fq);
            }
        }
        ++step;
    }
    if (!EMIT) {
        len[gid] = step;
    } else {
        for (; step < nb * 8; ++step)
            for (int i = 0; i < 16; ++i) ell[slot_addr(step, kLaneOf[g * 16 + i])] = static_cast<uint16_t>(pl.T + i);
    }
}

// build, step 3: blocks per (panel, tile, wave) = rounds of the wave x the longest of its rounds (in 8-step blocks,
// at least one), then an exclusive scan.  tab[(p*NT + t)*W + w] = {first block, blocks per round}; tab[2*n] = total.
__global__ __launch_bounds__(1024) void sliced_table_kernel(const int32_t* __restrict__ len, Plan pl, int32_t* __restrict__ tab) {
    __shared__ int32_t sm[1024];
    __shared__ int32_t carry;
    const int64_t n = static_cast<int64_t>(pl.panels) * pl.NT * pl.W;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t idx = base + threadIdx.x;
        int32_t blocks = 0, nb = 1;
        if (idx < n) {
            const int w = static_cast<int>(idx % pl.W);
            const int64_t pt = idx / pl.W;
            const int nr = (pl.S - w + pl.W - 1) / pl.W;
            for (int j = 0; j < nr; ++j) {
                const int64_t c = pt * pl.S + (j * pl.W + w);
                for (int g = 0; g < 4; ++g) {
                    const int v = (len[c * 4 + g] + 7) >> 3;
                    nb = v > nb ? v : nb;
                }
            }
            blocks = nb * nr;
        }
        sm[threadIdx.x] = blocks;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int32_t tv = (static_cast<int>(threadIdx.x) >= off) ? sm[threadIdx.x - off] : 0;
            __syncthreads();
            sm[threadIdx.x] += tv;
            __syncthreads();
        }
        const int32_t cin = carry;
        if (idx < n) {
            tab[idx * 2] = cin + sm[threadIdx.x] - blocks;
            tab[idx * 2 + 1] = nb;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = cin + sm[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) tab[n * 2] = carry;
}

// ------------------------------------------------------------------------------------------------------------
// run, step 1: ys[slice][row] = dinv[row] * x[row][4*slice .. 4*slice+3]   (slice-major: a tile of one slice is
// one contiguous stream), rows >= n_src zero.  dinv = sqrt(1 / in-degree) as difformer.py:66-68; a node without
// incoming entries contributes nothing (nan_to_num of the infinite value, :74).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dinv_of(const int32_t* __restrict__ rowptr, int64_t row) {
    const int32_t d = rowptr[row + 1] - rowptr[row];
    return d > 0 ? sqrtf(1.0f / static_cast<float>(d)) : 0.f;
}

constexpr int kPreSlices = 32;      // slices per block of the prescale pass (LDS staging: 32 x 65 float4)

__global__ __launch_bounds__(256) void sliced_prescale_kernel(const float* __restrict__ x, int64_t ldx,
                                                              const int32_t* __restrict__ rowptr, int64_t n_src,
                                                              int64_t npad, int slices, f32x4* __restrict__ ys) {
    __shared__ f32x4 stage[kPreSlices * 65];
    const int64_t row0 = static_cast<int64_t>(blockIdx.x) * 64;
    const int s0 = blockIdx.y * kPreSlices;
    const int ns = (slices - s0 < kPreSlices) ? slices - s0 : kPreSlices;
    const int total = 64 * ns;
    for (int e = threadIdx.x; e < total; e += 256) {       // coalesced over the slices of a row
        const int r = e / ns, sl = e % ns;
        const int64_t row = row0 + r;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < n_src) v = *reinterpret_cast<const f32x4*>(x + row * ldx + (s0 + sl) * 4) * dinv_of(rowptr, row);
        stage[sl * 65 + r] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += 256) {       // 64 consecutive rows of one slice = 1 KiB contiguous
        const int sl = e / 64, r = e % 64;
        const int64_t row = row0 + r;
        if (row < npad) ys[static_cast<int64_t>(s0 + sl) * npad + row] = stage[sl * 65 + r];
    }
}

// ------------------------------------------------------------------------------------------------------------
// run, step 2: the sweep
// ------------------------------------------------------------------------------------------------------------
struct Epilogue {
    const int32_t* rowptr;
    int64_t row_begin, n_rows;
    const float* attn;
    int64_t lda;
    float attn_scale, gcn_scale;
    float* out;
    int64_t ldo;
};

template <int NR>
__device__ __forceinline__ void sweep(f32x4* tile, const uint4* __restrict__ ell, const int2* __restrict__ tabw,
                                      const f32x4* __restrict__ ysl, const Plan pl, const Epilogue ep, int panel,
                                      int slice, int w, int lane) {
    const uint32_t four = 4;
    const int T = pl.T;
    f32x4 acc[NR > 0 ? NR : 1];
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // four entries (two packed dwords): 16-bit row number -> LDS byte address with one SDWA shift each
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
    for (int t = 0; t < pl.NT; ++t) {
        uint4 e[NR > 0 ? NR : 1];
        const uint4* base = ell;
        int nb = 0;
        if (NR > 0) {
            const int2 tb = tabw[static_cast<int64_t>(t) * pl.W];
            nb = __builtin_amdgcn_readfirstlane(tb.y);                       // >= 1
            base = ell + static_cast<int64_t>(__builtin_amdgcn_readfirstlane(tb.x)) * 64 + lane;
#pragma unroll
            for (int j = 0; j < NR; ++j) e[j] = base[j * 64];                // in flight across the tile load
        }
        __syncthreads();                                                      // everyone is done with the previous tile
        {
            const f32x4* src = ysl + static_cast<int64_t>(t) * T;
            const int nth = blockDim.x;
            for (int b0 = threadIdx.x; b0 < T; b0 += 5 * nth) {              // batches of five 16-byte loads per thread
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
        if (NR > 0) {
#pragma unroll 1
            for (int k = 0; k + 1 < nb; ++k) {
                const uint4* nx = base + static_cast<int64_t>(k + 1) * NR * 64;
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    half(e[j].x, e[j].y, acc[j]);
                    half(e[j].z, e[j].w, acc[j]);
                    e[j] = nx[j * 64];          // reloaded right after its last use; next use is a whole super-step away
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                half(e[j].x, e[j].y, acc[j]);
                half(e[j].z, e[j].w, acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int64_t prow = static_cast<int64_t>(j * pl.W + w) * 64 + lane;
        const int64_t lrow = static_cast<int64_t>(panel) * pl.P + prow;
        if (prow < pl.P && lrow < ep.n_rows) {
            f32x4 o = acc[j] * (ep.gcn_scale * dinv_of(ep.rowptr, ep.row_begin + lrow));
            if (ep.attn) o += ep.attn_scale * *reinterpret_cast<const f32x4*>(ep.attn + lrow * ep.lda + slice * 4);
            *reinterpret_cast<f32x4*>(ep.out + lrow * ep.ldo + slice * 4) = o;
        }
    }
}

template <int R>
__global__ __launch_bounds__(64 * kMaxWaves) void sliced_spmm_kernel(const uint4* __restrict__ ell, const int2* __restrict__ tab,
                                                                     const f32x4* __restrict__ ys, int64_t npad, Plan pl,
                                                                     Epilogue ep) {
    __shared__ f32x4 tile[kLdsRows];
    const int b = blockIdx.x;
    int panel, slice;
    const int per = gridDim.x >> 3;
    if ((gridDim.x & 7) == 0 && per % pl.slices == 0) {      // the slices of a panel share an XCD (block b -> XCD b % 8)
        const int xcd = b & 7, k = b >> 3;
        panel = xcd * (per / pl.slices) + k / pl.slices;
        slice = k % pl.slices;
    } else {
        panel = b / pl.slices;
        slice = b % pl.slices;
    }
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 16) tile[pl.T + threadIdx.x] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* ysl = ys + static_cast<int64_t>(slice) * npad;
    const int2* tabw = tab + static_cast<int64_t>(panel) * pl.NT * pl.W + w;
    const int nr = (pl.S - w + pl.W - 1) / pl.W;             // rounds of this wave: R or R - 1
    if (nr == R) sweep<R>(tile, ell, tabw, ysl, pl, ep, panel, slice, w, lane);
    else sweep<R - 1>(tile, ell, tabw, ysl, pl, ep, panel, slice, w, lane);
}

template <int R>
int launch_sweep(hipStream_t st, const uint4* ell, const int2* tab, const f32x4* ys, int64_t npad, const Plan& pl,
                 const Epilogue& ep) {
    hipLaunchKernelGGL((sliced_spmm_kernel<R>), dim3(static_cast<unsigned>(pl.panels * pl.slices)), dim3(64 * pl.W), 0, st,
                       ell, tab, ys, npad, pl, ep);
    return dif::launch_status("sliced_spmm_kernel");
}

int check_plan(const int32_t* plan, int64_t n_src, int64_t n_rows, int F, Plan& pl) {
    if (!plan) return dif::fail(DIF_E_BADARG, "dif_sliced: plan is null");
    Plan want;
    const int rc = make_plan(n_src, n_rows, F, want);
    if (rc) return dif::fail(rc, "dif_sliced: shape not covered (n_src=%lld, n_rows=%lld, F=%d)",
                             static_cast<long long>(n_src), static_cast<long long>(n_rows), F);
    pl = Plan{plan[0], plan[1], plan[2], plan[3], plan[4], plan[5], plan[6], plan[7]};
    if (pl.slices != want.slices || pl.panels != want.panels || pl.P != want.P || pl.S != want.S || pl.W != want.W ||
        pl.R != want.R || pl.T != want.T || pl.NT != want.NT)
        return dif::fail(DIF_E_BADARG, "dif_sliced: plan does not match dif_sliced_plan(n_src, n_rows, F)");
    return 0;
}

}  // namespace

extern "C" int dif_sliced_plan(int64_t n_src, int64_t n_rows, int F, int32_t* plan) {
    DIF_REQUIRE(plan != nullptr, DIF_E_BADARG, "dif_sliced_plan: plan is null");
    Plan p;
    const int rc = make_plan(n_src, n_rows, F, p);
    if (rc) return dif::fail(rc, "dif_sliced_plan: shape not covered (n_src=%lld, n_rows=%lld, F=%d): needs F %% 4 == 0, "
                             "F <= 1024", static_cast<long long>(n_src), static_cast<long long>(n_rows), F);
    const int32_t v[8] = {p.slices, p.panels, p.P, p.S, p.W, p.R, p.T, p.NT};
    for (int i = 0; i < 8; ++i) plan[i] = v[i];
    return 0;
}

extern "C" int dif_sliced_measure(const int32_t* rowptr, const int32_t* blkptr, const int32_t* src, int64_t n_src,
                                  int64_t nnz, int64_t row_begin, int64_t n_rows, int F, const int32_t* plan,
                                  uint16_t* sorted, void* counts, int32_t* lengths, int32_t* table, int32_t* status,
                                  dif_stream_t stream) {
    Plan pl;
    if (int rc = check_plan(plan, n_src, n_rows, F, pl)) return rc;
    DIF_REQUIRE(row_begin >= 0 && row_begin + n_rows <= n_src && nnz >= 0, DIF_E_BADARG, "dif_sliced_measure: bad row range");
    DIF_REQUIRE(rowptr && (nnz == 0 || src) && sorted && counts && lengths && table && status, DIF_E_BADARG,
                "dif_sliced_measure: null pointer");
    DIF_REQUIRE(pl.NT == 1 || blkptr, DIF_E_BADARG,
                "dif_sliced_measure: more than one tile needs the CSR built with n_blocks = plan[7], block_rows = plan[6]");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t he = hipMemsetAsync(status, 0, 4, st);
    if (he != hipSuccess) return dif::fail(static_cast<int>(he), "dif_sliced_measure: memset: %s", hipGetErrorString(he));
    const int64_t n_groups = n_rows * pl.NT;
    hipLaunchKernelGGL(sliced_sort_kernel, dim3(static_cast<unsigned>((n_groups + 255) / 256)), dim3(256), 0, st, rowptr,
                       blkptr, src, n_src, pl.NT, pl.T, row_begin, n_rows, sorted, static_cast<uint2*>(counts), status);
    if (int rc = dif::launch_status("sliced_sort_kernel")) return rc;
    const int64_t n_hw = static_cast<int64_t>(pl.panels) * pl.NT * pl.S * 4;
    hipLaunchKernelGGL((sliced_color_kernel<false>), dim3(static_cast<unsigned>((n_hw + kColorThreads - 1) / kColorThreads)),
                       dim3(kColorThreads), 0, st, rowptr, blkptr, n_src, row_begin, n_rows, pl, sorted,
                       static_cast<const uint2*>(counts), lengths, nullptr, nullptr);
    if (int rc = dif::launch_status("sliced_color_kernel")) return rc;
    hipLaunchKernelGGL(sliced_table_kernel, dim3(1), dim3(1024), 0, st, lengths, pl, table);
    return dif::launch_status("sliced_table_kernel");
}

extern "C" int dif_sliced_emit(const int32_t* rowptr, const int32_t* blkptr, int64_t n_src, int64_t row_begin,
                               int64_t n_rows, int F, const int32_t* plan, const uint16_t* sorted, const void* counts,
                               const int32_t* table, int64_t n_blocks, uint16_t* entries, dif_stream_t stream) {
    Plan pl;
    if (int rc = check_plan(plan, n_src, n_rows, F, pl)) return rc;
    DIF_REQUIRE(rowptr && sorted && counts && table && entries && n_blocks >= 1, DIF_E_BADARG, "dif_sliced_emit: null pointer");
    DIF_REQUIRE(pl.NT == 1 || blkptr, DIF_E_BADARG, "dif_sliced_emit: more than one tile needs blkptr");
    DIF_REQUIRE((reinterpret_cast<uintptr_t>(entries) & 15u) == 0, DIF_E_BADARG, "dif_sliced_emit: entries must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n_hw = static_cast<int64_t>(pl.panels) * pl.NT * pl.S * 4;
    hipLaunchKernelGGL((sliced_color_kernel<true>), dim3(static_cast<unsigned>((n_hw + kColorThreads - 1) / kColorThreads)),
                       dim3(kColorThreads), 0, st, rowptr, blkptr, n_src, row_begin, n_rows, pl, sorted,
                       static_cast<const uint2*>(counts), nullptr, table, entries);
    return dif::launch_status("sliced_color_kernel");
}

extern "C" int dif_sliced_prescale_f32(const float* x, int64_t ldx, const int32_t* rowptr, int64_t n_src, int F,
                                       const int32_t* plan, float* ys, dif_stream_t stream) {
    DIF_REQUIRE(x && rowptr && plan && ys && n_src > 0 && F > 0 && F % 4 == 0 && ldx >= F, DIF_E_BADARG,
                "dif_sliced_prescale: bad argument");
    DIF_REQUIRE(ldx % 4 == 0 && dif::aligned16(x) && dif::aligned16(ys), DIF_E_BADARG,
                "dif_sliced_prescale: x rows and ys must be 16-byte aligned");
    const int slices = plan[0];
    const int64_t npad = static_cast<int64_t>(plan[6]) * plan[7];
    DIF_REQUIRE(slices == F / 4 && npad >= n_src, DIF_E_BADARG, "dif_sliced_prescale: plan does not match F / n_src");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(sliced_prescale_kernel, dim3(static_cast<unsigned>((npad + 63) / 64), (slices + kPreSlices - 1) / kPreSlices),
                       dim3(256), 0, st, x, ldx, rowptr, n_src, npad, slices, reinterpret_cast<f32x4*>(ys));
    return dif::launch_status("sliced_prescale_kernel");
}

extern "C" int dif_sliced_spmm_f32(const uint16_t* entries, const int32_t* table, const int32_t* plan, const float* ys,
                                   const int32_t* rowptr, int64_t n_src, int64_t row_begin, int64_t n_rows, int F,
                                   const float* attn, int64_t lda, float attn_scale, float gcn_scale, float* out,
                                   int64_t ldo, dif_stream_t stream) {
    Plan pl;
    if (int rc = check_plan(plan, n_src, n_rows, F, pl)) return rc;
    DIF_REQUIRE(entries && table && ys && rowptr && out, DIF_E_BADARG, "dif_sliced_spmm: null pointer");
    DIF_REQUIRE(row_begin >= 0 && row_begin + n_rows <= n_src, DIF_E_BADARG, "dif_sliced_spmm: row range exceeds n_src");
    DIF_REQUIRE(ldo >= F && ldo % 4 == 0 && dif::aligned16(out) && (!attn || (lda >= F && lda % 4 == 0 && dif::aligned16(attn))),
                DIF_E_BADARG, "dif_sliced_spmm: out / attn rows must be 16-byte aligned with ld >= F");
    DIF_REQUIRE(dif::aligned16(entries) && dif::aligned16(ys), DIF_E_BADARG, "dif_sliced_spmm: entries / ys must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t npad = static_cast<int64_t>(pl.T) * pl.NT;
    const Epilogue ep = {rowptr, row_begin, n_rows, attn, lda, attn_scale, gcn_scale, out, ldo};
    const uint4* e4 = reinterpret_cast<const uint4*>(entries);
    const int2* tb = reinterpret_cast<const int2*>(table);
    const f32x4* y4 = reinterpret_cast<const f32x4*>(ys);
    switch (pl.R) {
        case 1: return launch_sweep<1>(st, e4, tb, y4, npad, pl, ep);
        case 2: return launch_sweep<2>(st, e4, tb, y4, npad, pl, ep);
        case 3: return launch_sweep<3>(st, e4, tb, y4, npad, pl, ep);
        case 4: return launch_sweep<4>(st, e4, tb, y4, npad, pl, ep);
        case 5: return launch_sweep<5>(st, e4, tb, y4, npad, pl, ep);
        case 6: return launch_sweep<6>(st, e4, tb, y4, npad, pl, ep);
        case 7: return launch_sweep<7>(st, e4, tb, y4, npad, pl, ep);
        case 8: return launch_sweep<8>(st, e4, tb, y4, npad, pl, ep);
        case 9: return launch_sweep<9>(st, e4, tb, y4, npad, pl, ep);
        case 10: return launch_sweep<10>(st, e4, tb, y4, npad, pl, ep);
    }
    return dif::fail(DIF_E_SHAPE, "dif_sliced_spmm: %d rounds per wave not covered", pl.R);
}

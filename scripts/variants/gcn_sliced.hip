// MEASUREMENT FORK of difformer_amd/csrc/gcn_sliced.hip as of round 5 (probes, traces, alternative layouts behind -D flags; results of most are
// WRONG by design).  Not part of the product: built only by scripts/build_sliced_variants.sh (OBJ=gcn_sliced) into scripts/bin/.  The product
// file carries none of these branches and compiles to the same device code as this fork without flags (round 6, checked).
// a3 (hot half, dense graphs): feature-sliced normalised-adjacency product with the SOURCE rows staged in LDS
//   out[r,:] = gcn_scale * dinv[r] * sum_{e in CSR row r} (dinv[src_e] * x[src_e,:])   (+ attn_scale * attn[r,:])
// node classification/difformer.py:63-79 for edge_weight = None: value_e = dinv[col] * dinv[row] factors into a
// pre-scaled source row and a per-destination scale, so an entry is nothing but a source index.
//
// Why: the round-1 kernel gathers a 256-byte row per entry through L1/L2 and is bound by the L2->L1 request rate
// (profiles/r02_pmc_spmm_blocked.json: 174 M requests per launch at C4, 154 G/s, 1.13 ms).  LDS serves random 16-byte
// reads an order of magnitude faster, so:
//   * a workgroup (one per CU) owns a PANEL of destination rows and one 16-byte SLICE (4 floats) of the feature row;
//     16 slices x 16 panels fill the chip at F = 64.  A lane owns whole destination rows: round j of wave w holds one
//     row per lane in a float4 register accumulator -- static registers, no LDS read-modify-write;
//   * the sources are swept tile by tile: the slice of T <= 10,208 pre-scaled source rows sits in LDS (all 160 KiB);
//     an entry is a 16-bit tile-local row number -> ONE ds_read_b128 + one float4 add per entry and lane;
//   * rows are grouped 64 at a time into SLOTS (in descending-degree order when the degrees are skewed, so that the
//     lock-step lanes of a slot carry lists of similar length).  Slot g = stratum j x pair index, dealt over the
//     (panel, wave) pairs in snake order: every wave gets one slot of every degree stratum and the totals balance;
//   * the lists are stored per (panel, tile, wave) as 1-KiB blocks of 8 entries x 64 lanes, rounds interleaved block
//     by block.  Round lengths are padded to a NON-INCREASING sequence (round 0 the longest): the rounds still active
//     at block row k are then a prefix {0 .. m(k)-1}, and the sweep runs one branch-free, statically unrolled phase per
//     prefix length.  Every entry register is reloaded right after its last use, a whole block row ahead of its next
//     one (no register rotation -> counted vmcnt waits);
//   * inside a (tile, round) the entries of the 16 lanes that share an LDS cycle of ds_read_b128 are scheduled at build
//     time (greedy edge colouring) so that they hit 16 different bank quads: no bank conflicts (SQ_LDS_BANK_CONFLICT
//     = 0); idle slots read one of 16 zero rows, also on a free quad.  Random order costs 1.6x.
// The same adjacency slice is swept by the 16 slice-workgroups of a panel, which sit on one XCD (block b -> XCD b % 8),
// so the entry stream comes from HBM once.
// Measured at C4 (79.3 M entries): 0.33 ms per launch against 1.12 ms (profiles/r02_experiments.md).
#include <stdlib.h>
#include "dif_common.h"

namespace {

using dif::f32x4;

// Geometry knobs of measurement builds (profiles/r03_experiments.md): the default is one 16-wave workgroup per CU with the
// whole LDS; -DDIF_SLICED_TILE_ROWS=5104 -DDIF_SLICED_MAX_WAVES=8 -DDIF_SLICED_WG_PER_CU=2 builds two 8-wave workgroups
// per CU with 80 KiB each (one drains / loads its tile while the other computes; half-length lists).
#ifndef DIF_SLICED_TILE_ROWS
#define DIF_SLICED_TILE_ROWS 10208
#endif
#ifndef DIF_SLICED_MAX_WAVES
#define DIF_SLICED_MAX_WAVES 16
#endif
#ifndef DIF_SLICED_WG_PER_CU
#define DIF_SLICED_WG_PER_CU 1
#endif
// Timing probes (results are WRONG; profiles/r04_experiments.md): 1 = entry registers never reloaded inside a tile, 2 = LDS
// reads without the adds, 3 = adds without the LDS reads, 4 = tiles loaded once (barriers kept), 5 = no tile loads, no barriers
#ifndef DIF_SLICED_PROBE
#define DIF_SLICED_PROBE 0
#endif
// Entry format of measurement builds (profiles/r04_experiments.md): -DDIF_SLICED_ENTRY32 stores every entry as the 32-bit
// LDS byte address of its row (no address shift in the sweep) in blocks of FOUR steps x 64 lanes -- the same 1-KiB block,
// twice the entry stream.  The default is the 16-bit tile-local row number in blocks of eight steps.
#ifdef DIF_SLICED_ENTRY32
constexpr int kSteps = 4;
#else
constexpr int kSteps = 8;
#endif
constexpr int kTileRowsMax = DIF_SLICED_TILE_ROWS;   // + 16 zero rows = 10,224 rows x 16 B = 163,584 B of LDS
constexpr int kLdsRows = kTileRowsMax + 16;
constexpr int kWgPerCU = DIF_SLICED_WG_PER_CU;
constexpr int kMaxRounds = 10;                   // destination rows per lane (float4 accumulator + entry registers each)
constexpr int kMaxWaves = DIF_SLICED_MAX_WAVES;
constexpr int kGroupCap = 65535;                 // entries of one (row, tile) group (16-bit counters)

// lane sets that share one LDS cycle of a ds_read_b128 (MI355X_MICROARCH.md, LDS table): position -> lane
__device__ const uint8_t kLaneOf[64] = {
    0,  1,  2,  3,  12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27,
    4,  5,  6,  7,  8,  9,  10, 11, 16, 17, 18, 19, 28, 29, 30, 31,
    32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59,
    36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63};

struct Plan {
    int slices, panels, G, PW, W, R, T, NT;      // G = 64-row slots, PW = panels * W (panel, wave) pairs, R = rounds
    int S = 1;                                   // source splits (derived, not part of the int32[8] record): see make_plan
};
constexpr int kMaxSplits = 8;

// slot of stratum j for pair index pw = wave * panels + panel (snake: odd strata run backwards)
__host__ __device__ __forceinline__ int64_t slot_of(int j, int pw, int PW) {
    return static_cast<int64_t>(j) * PW + ((j & 1) ? PW - 1 - pw : pw);
}

// Geometry for n_rows destination rows (a shard) over n_src source rows at F feature columns.
int make_plan(int64_t n_src, int64_t n_rows, int F, Plan& p) {
    if (n_src <= 0 || n_rows <= 0 || F <= 0 || F % 4 != 0 || F / 4 > 256) return DIF_E_SHAPE;
    p.slices = F / 4;
    const int64_t G = (n_rows + 63) / 64;
    int64_t panels = dif::kCUs * kWgPerCU / p.slices;
    if (panels < 1) panels = 1;
    const int64_t cap = static_cast<int64_t>(kMaxWaves) * kMaxRounds;            // slots one workgroup can own
    if (panels * cap < G) panels = (G + cap - 1) / cap;
    // A row shard (few destination rows, all the source rows): every workgroup would sweep ALL the source tiles for a
    // few rounds of work, and the sweep does not shrink with the rows.  Fewer, fuller panels instead, and S workgroups
    // per (panel, slice) that take 1/S of the source tiles each; their partial sums meet in a second pass
    // (sliced_combine_kernel), in a fixed order.
    const int64_t nt0 = (n_src + kTileRowsMax - 1) / kTileRowsMax;
    const int64_t need = (G + (cap - kMaxWaves) - 1) / (cap - kMaxWaves);        // panels at kMaxRounds - 1 rounds
    int S = 1;
    while (S * 2 <= kMaxSplits && n_src + (n_src >> 4) >= int64_t(2) * S * n_rows && need * S * 2 <= panels && nt0 >= S * 2) S *= 2;   // (1/16 slack: ranks' row counts are rounded up)
    p.S = S;
    panels /= S;
    if (panels > G) panels = G;
    if (panels * p.slices > (int64_t(1) << 20) || G >= (int64_t(1) << 30)) return DIF_E_RANGE;
    p.panels = static_cast<int>(panels);
    p.G = static_cast<int>(G);
    // fewest rounds first (they run back to back), then the fewest idle (wave, round) slots
    int bestW = 0, bestR = 1 << 30;
    int64_t bestWaste = int64_t(1) << 60;
    for (int W = kMaxWaves; W >= 1; --W) {
        const int64_t pw = panels * W;
        if (pw > G && W > 1) continue;
        const int R = static_cast<int>((G + pw - 1) / pw);
        const int64_t waste = static_cast<int64_t>(R) * pw - G;
        if (R < bestR || (R == bestR && waste < bestWaste)) { bestW = W; bestR = R; bestWaste = waste; }
    }
    if (bestR > kMaxRounds) return DIF_E_SHAPE;
    p.W = bestW;
    p.R = bestR;
    p.PW = p.panels * p.W;
    const int64_t nt = (nt0 + S - 1) / S * S;                                    // every split takes NT / S tiles
    if (nt > 32767) return DIF_E_RANGE;
    p.NT = static_cast<int>(nt);
    p.T = static_cast<int>((((n_src + nt - 1) / nt) + 15) / 16 * 16);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// build, step 1: one thread per (row position, tile) group: entries re-ordered by bank quad (source row mod 16) as
// 16-bit tile-local row numbers, plus the 16 16-bit counters (four per 64-bit word).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void group_bounds(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ blkptr,
                                             int64_t n_src, int NT, int64_t row, int t, int32_t& e0, int32_t& e1) {
    if (NT == 1) { e0 = rowptr[row]; e1 = rowptr[row + 1]; }
    else { e0 = blkptr[static_cast<int64_t>(t) * n_src + row]; e1 = blkptr[static_cast<int64_t>(t + 1) * n_src + row]; }
}

// A row POSITION is a whole destination row, or -- when `parts` is given -- part p of P of one (a hub row split over P
// consecutive lanes of one slot: in every tile part p takes the p-th of P equal shares of the row's entries), or
// empty (order[pos] < 0: padding so that the parts of a row never straddle a slot).  parts[pos] = p | P << 8.
__device__ __forceinline__ void position_bounds(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ blkptr,
                                                int64_t n_src, int NT, int64_t row_begin, const int32_t* __restrict__ order,
                                                const uint16_t* __restrict__ parts, int64_t pos, int t, int32_t& e0,
                                                int32_t& e1) {
    const int64_t lrow = order ? order[pos] : pos;
    if (lrow < 0) { e0 = e1 = 0; return; }
    group_bounds(rowptr, blkptr, n_src, NT, row_begin + lrow, t, e0, e1);
    if (parts) {
        const uint32_t pp = parts[pos];
        const int64_t p = pp & 0xffu, P = pp >> 8, c = e1 - e0;
        e1 = e0 + static_cast<int32_t>(c * (p + 1) / P);
        e0 = e0 + static_cast<int32_t>(c * p / P);
    }
}

struct Counts { uint64_t w[4]; };

// exclusive starts of the 16 buckets: lane k of c * 0x0001000100010001 = sum of lanes 0..k (no carries: total <= 65535)
__device__ __forceinline__ Counts bucket_starts(const Counts& c) {
    const uint64_t ones = 0x0001000100010001ull;
    Counts s;
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint64_t inc = c.w[k] * ones;
        s.w[k] = inc - c.w[k] + carry * ones;
        carry += inc >> 48;
    }
    return s;
}

__device__ __forceinline__ uint32_t lane16(const Counts& c, uint32_t q) {      // 16-bit counter q (dynamic)
    const uint64_t w = (q < 8u) ? ((q < 4u) ? c.w[0] : c.w[1]) : ((q < 12u) ? c.w[2] : c.w[3]);
    return static_cast<uint32_t>(w >> ((q & 3u) * 16u)) & 0xffffu;
}

__device__ __forceinline__ void bump16(Counts& c, uint32_t q) {
    const uint64_t one = 1ull << ((q & 3u) * 16u);
    const uint32_t k = q >> 2;
    c.w[0] += (k == 0u) ? one : 0ull;
    c.w[1] += (k == 1u) ? one : 0ull;
    c.w[2] += (k == 2u) ? one : 0ull;
    c.w[3] += (k == 3u) ? one : 0ull;
}

__device__ __forceinline__ Counts load_counts(const uint4* __restrict__ cnt, int64_t idx) {
    const uint4 a = cnt[idx * 2], b = cnt[idx * 2 + 1];
    Counts c;
    c.w[0] = static_cast<uint64_t>(a.x) | (static_cast<uint64_t>(a.y) << 32);
    c.w[1] = static_cast<uint64_t>(a.z) | (static_cast<uint64_t>(a.w) << 32);
    c.w[2] = static_cast<uint64_t>(b.x) | (static_cast<uint64_t>(b.y) << 32);
    c.w[3] = static_cast<uint64_t>(b.z) | (static_cast<uint64_t>(b.w) << 32);
    return c;
}

// One WAVE per row position: the entries of a (position, tile) group are contiguous in the CSR, so the wave reads them
// as one coalesced chunk of up to 64 (the usual group has ~46), ranks every entry inside its bank quad with 16 ballots
// (stable: original order inside a quad), and writes the 16-bit local ids + the 16 counters.  (The first version walked
// one group per THREAD: 184-byte runs per lane, 19.8 GB through HBM for 0.5 GB of useful bytes, 3.2 ms at C4.)
__global__ __launch_bounds__(256) void sliced_sort_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ blkptr,
                                                          const int32_t* __restrict__ src, int64_t n_src, int NT, int T,
                                                          int64_t row_begin, int64_t n_pos, const int32_t* __restrict__ order,
                                                          const uint16_t* __restrict__ parts,
                                                          uint16_t* __restrict__ srt, uint4* __restrict__ cnt,
                                                          int32_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int64_t pos = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (pos >= n_pos) return;
    const uint64_t below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t* cnt32 = reinterpret_cast<uint32_t*>(cnt);
    for (int t = 0; t < NT; ++t) {
        int32_t e0, e1;
        position_bounds(rowptr, blkptr, n_src, NT, row_begin, order, parts, pos, t, e0, e1);
        e0 = __builtin_amdgcn_readfirstlane(e0);
        e1 = __builtin_amdgcn_readfirstlane(e1);
        const int32_t base = t * T;
        if (e1 - e0 > kGroupCap) {       // 16-bit counters: such a graph takes the gather kernel
            if (lane == 0) atomicOr(status, 1);
            e1 = e0;
        }
        uint32_t count[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) count[q] = 0;
        for (int32_t c0 = e0; c0 < e1; c0 += 64) {
            const int32_t e = c0 + lane;
            const bool ok = e < e1;
            const uint32_t q = ok ? (static_cast<uint32_t>(src[e] - base) & 15u) : 16u;
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) count[qq] += static_cast<uint32_t>(__popcll(__ballot(q == static_cast<uint32_t>(qq))));
        }
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (lane == j) word = count[2 * j] | (count[2 * j + 1] << 16);
        if (lane < 8) cnt32[(pos * NT + t) * 8 + lane] = word;
        uint32_t run[16];
        uint32_t acc = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) { run[q] = acc; acc += count[q]; }
        for (int32_t c0 = e0; c0 < e1; c0 += 64) {
            const int32_t e = c0 + lane;
            const bool ok = e < e1;
            const uint32_t loc = ok ? static_cast<uint32_t>(src[e] - base) : 0u;
            const uint32_t q = ok ? (loc & 15u) : 16u;
            uint32_t dst = 0;
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) {
                const uint64_t m = __ballot(q == static_cast<uint32_t>(qq));
                if (q == static_cast<uint32_t>(qq)) dst = run[qq] + static_cast<uint32_t>(__popcll(m & below));
                run[qq] += static_cast<uint32_t>(__popcll(m));
            }
            if (ok) srt[e0 + dst] = static_cast<uint16_t>(loc);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// build, step 2: greedy edge colouring, one THREAD per (slot, tile, hardware lane group).  A step serves the 16 lanes
// in rotating order; a lane takes the bank quad it still has the most entries on among the quads no earlier lane of the
// step took (within ~0.2 % of the lower bound max(longest lane, fullest quad)).  EMIT = false only counts the steps;
// EMIT = true writes the schedule of every step as a 16-byte record into the block it belongs to.
// The 16 x 16 remaining counts (16-bit, two per dword) live in 128 REGISTERS: the lane loop is unrolled twice (lanes
// >= start, then lanes < start; start = step & 15 is wave-uniform, the skipped bodies cost a scalar branch) so that every
// register index is static.  There are only ~1,700 waves of this work at C4, so occupancy does not matter and the
// latency of the former LDS copy of the counts (74 KB per 128 threads, one wave per SIMD) was the whole run time:
// 1.16 -> 0.3 ms per pass.
// Table row of (panel, tile, wave): {first block, nb_0 >= nb_1 >= ... >= nb_{R-1}} (blocks per round, padded to a
// non-increasing sequence); block row k holds the rounds j with nb_j > k: block (k, j) = first + sum_j' min(nb_j', k) + j.
// ------------------------------------------------------------------------------------------------------------
constexpr int kColorThreads = 64;

template <bool EMIT>
__global__ __launch_bounds__(kColorThreads) void sliced_color_kernel(int64_t n_pos, Plan pl, const uint4* __restrict__ cnt,
                                                                     int32_t* __restrict__ len,
                                                                     const int32_t* __restrict__ tab,
                                                                     uint16_t* __restrict__ ell) {
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * kColorThreads + threadIdx.x;
    const int64_t n_groups = static_cast<int64_t>(pl.G) * pl.NT * 4;
    if (gid >= n_groups) return;
    const int grp = static_cast<int>(gid & 3);
    const int t = static_cast<int>((gid >> 2) % pl.NT);
    const int64_t g = (gid >> 2) / pl.NT;                     // slot
    uint32_t c[16][8];                           // c[lane][k]: entries left on quads 2k (low half) and 2k + 1 (high half)
    int remaining = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t pos = g * 64 + kLaneOf[grp * 16 + i];
        Counts cc = {{0, 0, 0, 0}};
        if (pos < n_pos) cc = load_counts(cnt, pos * pl.NT + t);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c[i][2 * k] = static_cast<uint32_t>(cc.w[k]);
            c[i][2 * k + 1] = static_cast<uint32_t>(cc.w[k] >> 32);
            remaining += static_cast<int>((cc.w[k] & 0xffffu) + ((cc.w[k] >> 16) & 0xffffu) + ((cc.w[k] >> 32) & 0xffffu) +
                                          (cc.w[k] >> 48));
        }
    }
    // where this slot's blocks go
    const int j = static_cast<int>(g / pl.PW);
    const int idxp = static_cast<int>(g % pl.PW);
    const int pw = (j & 1) ? pl.PW - 1 - idxp : idxp;
    const int p = pw % pl.panels, w = pw / pl.panels;
    const int32_t* tb = nullptr;
    int64_t blk0 = 0;
    int nbj = 0;
    if (EMIT) {
        tb = tab + ((static_cast<int64_t>(p) * pl.NT + t) * pl.W + w) * (pl.R + 1);
        blk0 = tb[0];
        nbj = tb[1 + j];
    }
    int64_t rowbase = 0;            // blocks before block row k = step >> 3 of this (panel, tile, wave)
    // EMIT: the schedule of a step goes into the block it belongs to as one 16-byte record {bank quads of lanes 0-7,
    // of lanes 8-15 (4 bits each), mask of lanes holding a real entry, 0} at uint4 grp * 8 + (step & 7);
    // sliced_fill_kernel then turns the records of a block into its entries, in place.
    auto record_at = [&](int step) -> uint4* {
        return reinterpret_cast<uint4*>(ell) + (blk0 + rowbase + j) * 64 + grp * kSteps + (step % kSteps);
    };
    auto new_row = [&](int step) {
        if (EMIT && (step % kSteps) == 0) {
            const int k = step / kSteps;
            rowbase = 0;
            for (int j2 = 0; j2 < pl.R; ++j2) { const int v = tb[1 + j2]; rowbase += v < k ? v : k; }
        }
    };
    int step = 0;
    while (remaining > 0) {
        new_row(step);
        uint32_t used = 0, picked = 0, qlo = 0, qhi = 0;
        const int start = step & 15;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if ((i >= start) != (pass == 0)) continue;
                // most entries left among the free quads; ties go to the lower quad (key = count << 4 | 15 - quad)
                uint32_t key = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t lo = ((c[i][k] & 0xffffu) << 4) | static_cast<uint32_t>(15 - 2 * k);
                    const uint32_t hi = ((c[i][k] >> 16) << 4) | static_cast<uint32_t>(14 - 2 * k);
                    const uint32_t flo = ((used >> (2 * k)) & 1u) ? 0u : lo;
                    const uint32_t fhi = ((used >> (2 * k + 1)) & 1u) ? 0u : hi;
                    key = max(key, max(flo, fhi));
                }
                if (key < 16u) continue;                       // nothing left on a free quad
                const uint32_t best = 15u - (key & 15u);
                used |= 1u << best;
                picked |= 1u << i;
                const uint32_t dec = 1u << (16u * (best & 1u)), wsel = best >> 1;
#pragma unroll
                for (int k = 0; k < 8; ++k) c[i][k] -= (wsel == static_cast<uint32_t>(k)) ? dec : 0u;
                --remaining;
                if (EMIT) {
                    if (i < 8) qlo |= best << (4 * i);
                    else qhi |= best << (4 * (i - 8));
                }
            }
        }
        if (EMIT) {             // idle lanes read a zero row on a quad nobody uses in this step
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if ((picked >> i) & 1u) continue;
                const uint32_t fq = __builtin_ctz(~used & 0xffffu);
                used |= 1u << fq;
                if (i < 8) qlo |= fq << (4 * i);
                else qhi |= fq << (4 * (i - 8));
            }
            *record_at(step) = uint4{qlo, qhi, picked, 0u};
        }
        ++step;
    }
    if (!EMIT) {
        len[gid] = step;
    } else {
        for (; step < nbj * kSteps; ++step) {       // padding steps: lane i reads zero row i
            new_row(step);
            *record_at(step) = uint4{0x76543210u, 0xfedcba98u, 0u, 0u};
        }
    }
}

// build, step 4: one wave per (slot, tile) walks the slot's blocks in step order and replaces the schedule records by
// the entries -- lane L takes, from the bank-quad bucket the record names, the next entry of its (row position, tile)
// group (16-bit tile-local source row), or the zero row T + quad where it idles; eight steps = one 16-byte store.
// The 64 groups of the slot are first copied into LDS with coalesced loads (64 lanes picking single entries out of 64
// different groups would cost one cache line per lane and load: 1.08 ms at C4 against 0.2 ms); slots whose groups do not
// fit (hub rows) read global memory directly.
constexpr int kFillStage = 4096;                 // entries of one slot in LDS (8 KiB per wave)

template <bool STAGED>
__device__ __forceinline__ void fill_blocks(const uint16_t* __restrict__ mine, const uint16_t* stage, Counts next,
                                            const int32_t* __restrict__ tb, int R, int j, int nbj, int64_t blk0, int grp,
                                            int i, int lane, uint32_t T, uint4* __restrict__ ell) {
    uint64_t n0 = next.w[0], n1 = next.w[1], n2 = next.w[2], n3 = next.w[3];
    const uint32_t himask = (i >= 8) ? ~0u : 0u, sh = 4u * (i & 7);
    for (int k = 0; k < nbj; ++k) {
        int64_t rowbase = 0;
        for (int j2 = 0; j2 < R; ++j2) { const int v = tb[1 + j2]; rowbase += v < k ? v : k; }
        uint4* blk = ell + (blk0 + rowbase + j) * 64;
        uint4 rec[kSteps];
#pragma unroll
        for (int s8 = 0; s8 < kSteps; ++s8) rec[s8] = blk[grp * kSteps + s8];
        uint32_t ent[kSteps];
#pragma unroll
        for (int s8 = 0; s8 < kSteps; ++s8) {
            // masks, not selects between array elements: those turn `rec` / `next` into indexed scratch arrays
            const uint32_t q = ((((rec[s8].x & ~himask) | (rec[s8].y & himask)) >> sh)) & 15u;
            const bool real = (rec[s8].z >> i) & 1u;
            const uint32_t k = q >> 2;
            const uint64_t m0 = k == 0u ? ~0ull : 0ull, m1 = k == 1u ? ~0ull : 0ull, m2 = k == 2u ? ~0ull : 0ull,
                           m3 = k == 3u ? ~0ull : 0ull;
            const uint64_t wsel = (n0 & m0) | (n1 & m1) | (n2 & m2) | (n3 & m3);
            const uint32_t qs = (q & 3u) * 16u;
            const uint32_t at = real ? (static_cast<uint32_t>(wsel >> qs) & 0xffffu) : 0u;   // idle lanes read entry 0 and drop it
            const uint32_t v = STAGED ? stage[at] : mine[at];
            const uint64_t one = real ? (1ull << qs) : 0ull;
            n0 += one & m0; n1 += one & m1; n2 += one & m2; n3 += one & m3;
            ent[s8] = real ? v : T + q;
        }
        // every lane has read its records (the loads above belong to instructions that completed for the whole wave
        // before the first dependent use); now the block takes its final content
#ifdef DIF_SLICED_ENTRY32
        blk[lane] = uint4{ent[0] << 4, ent[1] << 4, ent[2] << 4, ent[3] << 4};
#else
        blk[lane] = uint4{ent[0] | (ent[1] << 16), ent[2] | (ent[3] << 16), ent[4] | (ent[5] << 16), ent[6] | (ent[7] << 16)};
#endif
    }
}

__global__ __launch_bounds__(256) void sliced_fill_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ blkptr, int64_t n_src, int64_t row_begin,
    int64_t n_pos, const int32_t* __restrict__ order, const uint16_t* __restrict__ parts, Plan pl,
    const uint16_t* __restrict__ srt, const uint4* __restrict__ cnt, const int32_t* __restrict__ tab,
    uint4* __restrict__ ell) {
    __shared__ uint16_t stage_s[4][kFillStage + 8];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t gw = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    const int64_t g = gw / pl.NT;
    const int t = static_cast<int>(gw % pl.NT);
    if (g >= pl.G) return;
    const int j = static_cast<int>(g / pl.PW);
    const int idxp = static_cast<int>(g % pl.PW);
    const int pw = (j & 1) ? pl.PW - 1 - idxp : idxp;
    const int p = pw % pl.panels, w = pw / pl.panels;
    const int32_t* tb = tab + ((static_cast<int64_t>(p) * pl.NT + t) * pl.W + w) * (pl.R + 1);
    const int64_t blk0 = tb[0];
    const int nbj = tb[1 + j];
    if (nbj == 0) return;
    int inv = 0;                                   // lane = kLaneOf[inv]: group inv >> 4, index inv & 15
#pragma unroll
    for (int k = 0; k < 64; ++k) inv = (kLaneOf[k] == lane) ? k : inv;
    const int grp = inv >> 4, i = inv & 15;
    const int64_t pos = g * 64 + lane;
    Counts next = {{0, 0, 0, 0}};
    int32_t e0 = 0, e1 = 0;
    if (pos < n_pos) {
        position_bounds(rowptr, blkptr, n_src, pl.NT, row_begin, order, parts, pos, t, e0, e1);
        next = bucket_starts(load_counts(cnt, pos * pl.NT + t));     // next unread entry of each quad bucket
    }
    // where the groups of the 64 lanes go in LDS (exclusive scan of their sizes); entry 0 of `stage` / of the lane's
    // group is what idle steps read, so an empty group must still point at readable memory
    const int mycnt = e1 - e0;
    int inc = mycnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    const int total = __shfl(inc, 63, 64);
    uint16_t* stage = stage_s[wave];
    if (total <= kFillStage) {
        const int off = inc - mycnt;
        for (int r0 = 0; r0 < 64; r0 += 16) {          // 16 groups per batch: all their loads in flight, then the LDS stores
            uint16_t v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int n_r = __builtin_amdgcn_readlane(mycnt, r0 + u);
                const int32_t e_r = __builtin_amdgcn_readlane(e0, r0 + u);
                v[u] = (lane < n_r) ? srt[e_r + lane] : uint16_t(0);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int n_r = __builtin_amdgcn_readlane(mycnt, r0 + u), off_r = __builtin_amdgcn_readlane(off, r0 + u);
                if (lane < n_r) stage[off_r + lane] = v[u];
            }
            for (int u = 0; u < 16; ++u) {              // groups beyond 64 entries: the rest
                const int n_r = __builtin_amdgcn_readlane(mycnt, r0 + u), off_r = __builtin_amdgcn_readlane(off, r0 + u);
                const int32_t e_r = __builtin_amdgcn_readlane(e0, r0 + u);
                for (int x = lane + 64; x < n_r; x += 64) stage[off_r + x] = srt[e_r + x];
            }
        }
        __builtin_amdgcn_wave_barrier();
        fill_blocks<true>(nullptr, stage + off, next, tb, pl.R, j, nbj, blk0, grp, i, lane, static_cast<uint32_t>(pl.T),
                          ell);
    } else {
        fill_blocks<false>(mycnt > 0 ? srt + e0 : srt, nullptr, next, tb, pl.R, j, nbj, blk0, grp, i, lane,
                           static_cast<uint32_t>(pl.T), ell);
    }
}

// build, step 3: per (panel, tile, wave) the blocks of every round (8-step blocks, longest of the slot's four lane
// groups), padded from the right to a non-increasing sequence; exclusive scan of the totals.  tab[n * (R + 1)] = total.
__global__ __launch_bounds__(1024) void sliced_table_kernel(const int32_t* __restrict__ len, Plan pl, int32_t* __restrict__ tab) {
    __shared__ int32_t sm[1024];
    __shared__ int32_t carry;
    const int64_t n = static_cast<int64_t>(pl.panels) * pl.NT * pl.W;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t idx = base + threadIdx.x;
        int32_t blocks = 0;
        if (idx < n) {
            const int w = static_cast<int>(idx % pl.W);
            const int t = static_cast<int>((idx / pl.W) % pl.NT);
            const int p = static_cast<int>(idx / pl.W / pl.NT);
            const int pw = w * pl.panels + p;
            int32_t* row = tab + idx * (pl.R + 1);
            int32_t env = 0;
            for (int j = pl.R - 1; j >= 0; --j) {
                const int64_t g = slot_of(j, pw, pl.PW);
                int32_t nb = 0;
                if (g < pl.G) {
                    for (int q = 0; q < 4; ++q) {
                        const int32_t v = (len[(g * pl.NT + t) * 4 + q] + kSteps - 1) / kSteps;
                        nb = v > nb ? v : nb;
                    }
                }
                env = nb > env ? nb : env;
                row[1 + j] = env;
                blocks += env;
            }
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
        if (idx < n) tab[idx * (pl.R + 1)] = cin + sm[threadIdx.x] - blocks;
        __syncthreads();
        if (threadIdx.x == 1023) carry = cin + sm[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) tab[n * (pl.R + 1)] = carry;
}

// ------------------------------------------------------------------------------------------------------------
// run, step 1: ys[slice][row] = dinv[row] * x[row][4*slice .. 4*slice+3]   (slice-major: a tile of one slice is
// one contiguous stream), rows >= n_src zero.  dinv = sqrt(1 / in-degree) as difformer.py:66-68; a node without
// incoming entries contributes nothing (nan_to_num of the infinite value, :74).  The in-degree is the row length of
// the CSR; for the ADJOINT product (CSR of the transposed graph, rows = sources) the caller passes the vector.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dinv_of(const int32_t* __restrict__ rowptr, int64_t row) {
    const int32_t d = rowptr[row + 1] - rowptr[row];
    return d > 0 ? sqrtf(1.0f / static_cast<float>(d)) : 0.f;
}

constexpr int kPreSlices = 32;      // slices per block of the prescale pass (LDS staging: 32 x 65 float4)

__global__ __launch_bounds__(256) void sliced_prescale_kernel(const float* __restrict__ x, int64_t ldx,
                                                              const int32_t* __restrict__ rowptr,
                                                              const float* __restrict__ dinv, int64_t n_src,
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
        if (row < n_src)
            v = *reinterpret_cast<const f32x4*>(x + row * ldx + (s0 + sl) * 4) * (dinv ? dinv[row] : dinv_of(rowptr, row));
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
    const float* dinv;
    const int32_t* order;
    const uint16_t* parts;
    int64_t row_begin, n_pos;
    const float* attn;
    int64_t lda;
    float attn_scale, gcn_scale;
    float* out;
    int64_t ldo;
    f32x4* partial;        // S > 1: [S][slices][G * 64] raw sums of the splits, finished by sliced_combine_kernel
#ifdef DIF_SLICED_TRACE
    long long* trace;      // measurement build only (scripts/exp_sliced_trace.py): per-tile wall-clock stamps of the first and last wave
#endif
};

// one block of a round: eight entries (four packed dwords), 16-bit row number -> LDS byte address with one SDWA shift
// each; the entry register is reloaded (its next block) as soon as the addresses are out, BEFORE the LDS reads, so the
// load has the whole block in flight even in the phases where few rounds are active
// float4 sum; measurement builds: -DDIF_SLICED_SCALAR_ADD forces four v_add_f32 instead of two v_pk_add_f32
__device__ __forceinline__ f32x4 add4(f32x4 x, f32x4 y) {
#ifdef DIF_SLICED_SCALAR_ADD
    f32x4 r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r.x) : "v"(x.x), "v"(y.x));
    asm("v_add_f32 %0, %1, %2" : "=v"(r.y) : "v"(x.y), "v"(y.y));
    asm("v_add_f32 %0, %1, %2" : "=v"(r.z) : "v"(x.z), "v"(y.z));
    asm("v_add_f32 %0, %1, %2" : "=v"(r.w) : "v"(x.w), "v"(y.w));
    return r;
#else
    return x + y;
#endif
}

__device__ __forceinline__ void block8(const f32x4* tile, uint4& e, const uint4* reload, f32x4& a) {
#ifdef DIF_SLICED_ENTRY32
    // four steps: the entries ARE the LDS byte addresses
    const uint32_t ad[4] = {e.x, e.y, e.z, e.w};
    f32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(tile) + ad[q]);
    e = *reload;
    a = add4(a, add4(add4(v[0], v[1]), add4(v[2], v[3])));
    asm volatile("" : "+v"(a));          // the sum is due HERE (the machine sinker otherwise parks every round's adds behind the last round's reads)
#else
    const uint32_t four = 4;
    const uint32_t wds[4] = {e.x, e.y, e.z, e.w};
    uint32_t ad[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
            : "=v"(ad[2 * q]) : "v"(four), "v"(wds[q]));
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
            : "=v"(ad[2 * q + 1]) : "v"(four), "v"(wds[q]));
    }
#if DIF_SLICED_PROBE != 1
    e = *reload;
#endif
#if DIF_SLICED_PROBE == 2          // reads only: results kept alive, no adds
    {
        f32x4 v[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(tile) + ad[4 * h + q]);
            asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
        }
        return;
    }
#elif DIF_SLICED_PROBE == 3        // adds only: the addresses stand in for the rows
    {
        f32x4 v[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float f = __uint_as_float(ad[4 * h + q]); v[q] = f32x4{f, f, f, f}; }
            a = add4(a, add4(add4(v[0], v[1]), add4(v[2], v[3])));
            asm volatile("" : "+v"(a));
        }
        return;
    }
#endif
#ifdef DIF_SLICED_READS8
    // all eight reads in flight before the first add (32 result registers instead of 16)
    f32x4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(tile) + ad[q]);
    a = add4(a, add4(add4(add4(v[0], v[1]), add4(v[2], v[3])), add4(add4(v[4], v[5]), add4(v[6], v[7]))));
    asm volatile("" : "+v"(a));
#else
    f32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(tile) + ad[q]);
    a = add4(a, add4(add4(v[0], v[1]), add4(v[2], v[3])));
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(tile) + ad[4 + q]);
    a = add4(a, add4(add4(v[0], v[1]), add4(v[2], v[3])));
#endif
#endif
}

// Phase M: the block rows in which exactly the rounds 0 .. M-1 are active (nb[M] <= k < nb[M-1]).
template <int M, int NR>
struct Phases {
    static __device__ __forceinline__ void run(const f32x4* tile, int& k, const uint4*& cur, uint4 (&e)[NR], f32x4 (&acc)[NR],
                                               const int (&nb)[NR]) {
        const int kend = nb[M - 1];
#pragma unroll 1
        for (; k + 1 < kend; ++k) {
            const uint4* nx = cur + M * 64;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                block8(tile, e[j], nx + j * 64, acc[j]);      // reloaded inside; next use is a whole block row away
                __builtin_amdgcn_sched_barrier(0);
            }
            cur = nx;
        }
        if (k < kend) {                     // last row of the phase: the next row keeps only the rounds with nb[j] > k + 1
            const uint4* nx = cur + M * 64;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                block8(tile, e[j], (nb[j] > k + 1) ? nx + j * 64 : cur, acc[j]);      // a finished round re-reads a block that exists
                __builtin_amdgcn_sched_barrier(0);
            }
            cur = nx;
            ++k;
            // the registers of the rounds that just ended are re-used by the shorter phases while their last (unused)
            // load is still in flight; draining here keeps the per-round counted waits inside the next phase's loop
            __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
        }
        Phases<M - 1, NR>::run(tile, k, cur, e, acc, nb);
    }
};
template <int NR>
struct Phases<0, NR> {
    static __device__ __forceinline__ void run(const f32x4*, int&, const uint4*&, uint4 (&)[NR], f32x4 (&)[NR], const int (&)[NR]) {}
};

// The sum of one row position (all 64 lanes of a slot call this together): parts of a split row are added up by
// part 0, then the row is scaled, combined with the attention branch and stored.
__device__ __forceinline__ void finish_position(const Epilogue& ep, int64_t pos, int slice, f32x4 acc) {
    int64_t lrow = -1;
    if (pos < ep.n_pos) lrow = ep.order ? ep.order[pos] : pos;
    f32x4 sum = acc;
    if (ep.parts) {
        // the parts of a split row sit in consecutive lanes of this slot: part 0 adds them up in part order
        const uint32_t pp = (pos < ep.n_pos && lrow >= 0) ? ep.parts[pos] : 0x0100u;
        const int p = pp & 0xffu, P = pp >> 8;
        for (int k = 1; __ballot(k < P) != 0ull; ++k) {
            f32x4 o;
            o.x = __shfl_down(acc.x, k);
            o.y = __shfl_down(acc.y, k);
            o.z = __shfl_down(acc.z, k);
            o.w = __shfl_down(acc.w, k);
            if (p == 0 && k < P) sum += o;
        }
        if (p != 0) lrow = -1;
    }
    if (lrow >= 0) {
        const int64_t row = ep.row_begin + lrow;
        f32x4 o = sum * (ep.gcn_scale * (ep.dinv ? ep.dinv[row] : dinv_of(ep.rowptr, row)));
        if (ep.attn) o += ep.attn_scale * *reinterpret_cast<const f32x4*>(ep.attn + lrow * ep.lda + slice * 4);
        *reinterpret_cast<f32x4*>(ep.out + lrow * ep.ldo + slice * 4) = o;
    }
}

template <int NR>
__device__ __forceinline__ void sweep(f32x4* tile, const uint4* __restrict__ ell, const int32_t* __restrict__ tabw,
                                      int64_t tab_stride, const f32x4* __restrict__ ysl, const Plan pl, const Epilogue ep,
                                      int pw, int slice, int lane, int t0, int z) {
    constexpr int NA = NR > 0 ? NR : 1;
    const int T = pl.T;
    f32x4 acc[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntl = pl.NT / pl.S;                                             // this split's tiles: [z * ntl, (z + 1) * ntl)
    for (int tt = 0; tt < ntl; ++tt) {
        int t = tt + t0;                                                      // XCDs start on different tiles (t0)
        if (t >= ntl) t -= ntl;
        t += z * ntl;
#ifdef DIF_SLICED_TRACE
        const int tw = (threadIdx.x >> 6) == 0 ? 0 : ((static_cast<int>(threadIdx.x >> 6) == pl.W - 1) ? 1 : -1);
        long long* tr4 = (tw >= 0 && lane == 0) ? ep.trace + ((static_cast<int64_t>(blockIdx.x) * 2 + tw) * pl.NT + tt) * 4 : nullptr;
#define DIF_STAMP(i) do { if (tr4) tr4[i] = wall_clock64(); } while (0)
#else
#define DIF_STAMP(i) do { } while (0)
#endif
        uint4 e[NA];
        int nb[NA];
        const uint4* cur = ell;
        if (NR > 0) {
            const int32_t* tr = tabw + static_cast<int64_t>(t) * tab_stride;
            cur = ell + static_cast<int64_t>(__builtin_amdgcn_readfirstlane(tr[0])) * 64 + lane;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                nb[j] = __builtin_amdgcn_readfirstlane(tr[1 + j]);
                e[j] = *((nb[j] > 0) ? cur + j * 64 : ell);                  // in flight across the tile load
            }
        }
#if DIF_SLICED_PROBE == 5
        if (tt == 0)
#endif
        {
            // The tile's first loads are issued BEFORE the barrier (they land in registers, not in LDS), so their latency
            // runs under the wait for the slowest wave of the previous tile; the registers written first take the tail
            // rows.  Wave-uniform bases + one shared lane offset keep the addressing in scalar registers.
            constexpr int U = NR >= 10 ? 4 : (NR == 9 ? 7 : (NR == 8 ? 9 : 11));
            const f32x4* src = ysl + static_cast<int64_t>(t) * T;
            const int nth = blockDim.x;
            const uint32_t tid = threadIdx.x;
            f32x4 r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f32x4* su = src + u * nth;
                r[u] = (u * nth + static_cast<int>(tid) < T) ? su[tid] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            DIF_STAMP(0);
            __syncthreads();                                                  // everyone is done with the previous tile
            DIF_STAMP(1);
#if DIF_SLICED_PROBE == 4
            if (tt == 0)
#endif
#pragma unroll
            for (int b = 0; b < 11; b += U) {                                 // batches of U rows per thread, 11 in all
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (b + u < 11) {
                        f32x4* du = tile + (b + u) * nth;
                        if ((b + u) * nth + static_cast<int>(tid) < T) du[tid] = r[u];
                    }
                    if (b + U + u < 11) {                                     // the register just written takes its next row
                        const f32x4* su = src + (b + U + u) * nth;
                        r[u] = ((b + U + u) * nth + static_cast<int>(tid) < T) ? su[tid] : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            for (int b0 = static_cast<int>(tid) + 11 * nth; b0 < T; b0 += nth) tile[b0] = src[b0];      // fewer than 15 waves
        }
        // The tile loads sit under exec masks, so the compiler's wait-count model would carry them as "possibly
        // pending" into every phase loop and wait for ALL loads at the top of each block row.  They are complete here
        // (the LDS stores consumed them, and the entry loads were issued before them): say so.
        __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
#if DIF_SLICED_PROBE == 5
        if (tt == 0)
#endif
        __syncthreads();
        DIF_STAMP(2);
        if (NR > 0) {
            int k = 0;
            Phases<NR, NA>::run(tile, k, cur, e, acc, nb);
        }
        DIF_STAMP(3);
    }
#undef DIF_STAMP
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int64_t pos = slot_of(j, pw, pl.PW) * 64 + lane;
        if (pl.S > 1) ep.partial[(static_cast<int64_t>(z) * pl.slices + slice) * pl.G * 64 + pos] = acc[j];
        else finish_position(ep, pos, slice, acc[j]);
    }
}

template <int R>
__global__ __launch_bounds__(64 * kMaxWaves, (kWgPerCU * kMaxWaves + 3) / 4) void sliced_spmm_kernel(const uint4* __restrict__ ell, const int32_t* __restrict__ tab,
                                                                     const f32x4* __restrict__ ys, int64_t npad, Plan pl,
                                                                     Epilogue ep) {
    __shared__ f32x4 tile[kLdsRows];
    const int b = blockIdx.x;
    int vp, slice, t0 = 0;                                   // vp = panel * S + split
    const int per = gridDim.x >> 3;
    if ((gridDim.x & 7) == 0 && per % pl.slices == 0) {      // the slices of a panel share an XCD (block b -> XCD b % 8)
        const int xcd = b & 7, k = b >> 3;
        vp = xcd * (per / pl.slices) + k / pl.slices;
        slice = k % pl.slices;
        t0 = (xcd * (pl.NT / pl.S)) >> 3;                    // XCDs start on different tiles: their tile loads do not coincide
    } else {
        vp = b / pl.slices;
        slice = b % pl.slices;
    }
    const int panel = vp / pl.S, z = vp % pl.S;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 16) tile[pl.T + threadIdx.x] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* ysl = ys + static_cast<int64_t>(slice) * npad;
    const int32_t* tabw = tab + (static_cast<int64_t>(panel) * pl.NT * pl.W + w) * (pl.R + 1);
    const int64_t tab_stride = static_cast<int64_t>(pl.W) * (pl.R + 1);
    const int pw = w * pl.panels + panel;
    const bool full = slot_of(R - 1, pw, pl.PW) < pl.G;      // rounds of this wave: R or R - 1
    if (full) sweep<R>(tile, ell, tabw, tab_stride, ysl, pl, ep, pw, slice, lane, t0, z);
    else sweep<R - 1>(tile, ell, tabw, tab_stride, ysl, pl, ep, pw, slice, lane, t0, z);
}

// second pass of a source-split product: the S partial sums of every (row position, slice), added in split order
__global__ __launch_bounds__(256) void sliced_combine_kernel(Plan pl, Epilogue ep) {
    const int64_t pair = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (pair >= static_cast<int64_t>(pl.G) * pl.slices) return;
    const int slice = static_cast<int>(pair % pl.slices);
    const int64_t pos = (pair / pl.slices) * 64 + (threadIdx.x & 63);
    const int64_t plane = static_cast<int64_t>(pl.G) * 64;
    const f32x4* src = ep.partial + slice * plane + pos;
    f32x4 v[kMaxSplits];
#pragma unroll
    for (int z = 0; z < kMaxSplits; ++z)
        if (z < pl.S) v[z] = src[static_cast<int64_t>(z) * pl.slices * plane];
    f32x4 sum = v[0];
#pragma unroll
    for (int z = 1; z < kMaxSplits; ++z)
        if (z < pl.S) sum += v[z];
    finish_position(ep, pos, slice, sum);
}

template <int R>
int launch_sweep(hipStream_t st, const uint4* ell, const int32_t* tab, const f32x4* ys, int64_t npad, const Plan& pl,
                 const Epilogue& ep) {
    hipLaunchKernelGGL((sliced_spmm_kernel<R>), dim3(static_cast<unsigned>(pl.panels * pl.S * pl.slices)), dim3(64 * pl.W),
                       0, st, ell, tab, ys, npad, pl, ep);
    if (int rc = dif::launch_status("sliced_spmm_kernel")) return rc;
    if (pl.S == 1) return 0;
    const int64_t pairs = static_cast<int64_t>(pl.G) * pl.slices;            // one wave per (slot, slice)
    hipLaunchKernelGGL(sliced_combine_kernel, dim3(static_cast<unsigned>((pairs + 3) / 4)), dim3(256), 0, st, pl, ep);
    return dif::launch_status("sliced_combine_kernel");
}

// n_pos row positions: the n_rows rows themselves, or (parts != NULL) their parts plus padding
int check_positions(const int32_t* row_order, const uint16_t* parts, int64_t n_rows, int64_t n_pos, const char* who) {
    if (parts ? (row_order == nullptr || n_pos < n_rows) : (n_pos != n_rows))
        return dif::fail(DIF_E_BADARG, "%s: n_pos must equal n_rows without `parts`; with `parts`, row_order and n_pos >= "
                         "n_rows are required", who);
    return 0;
}

int check_plan(const int32_t* plan, int64_t n_src, int64_t n_rows, int F, Plan& pl) {
    if (!plan) return dif::fail(DIF_E_BADARG, "dif_sliced: plan is null");
    Plan want;
    const int rc = make_plan(n_src, n_rows, F, want);
    if (rc) return dif::fail(rc, "dif_sliced: shape not covered (n_src=%lld, n_rows=%lld, F=%d)",
                             static_cast<long long>(n_src), static_cast<long long>(n_rows), F);
    pl = Plan{plan[0], plan[1], plan[2], plan[3], plan[4], plan[5], plan[6], plan[7], want.S};
    if (pl.slices != want.slices || pl.panels != want.panels || pl.G != want.G || pl.PW != want.PW || pl.W != want.W ||
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
    const int32_t v[8] = {p.slices, p.panels, p.G, p.PW, p.W, p.R, p.T, p.NT};
    for (int i = 0; i < 8; ++i) plan[i] = v[i];
    return 0;
}

extern "C" int64_t dif_sliced_spmm_workspace_bytes(int64_t n_src, int64_t n_rows, int F) {
    Plan p;
    if (make_plan(n_src, n_rows, F, p) != 0 || p.S == 1) return 0;
    return static_cast<int64_t>(p.S) * p.slices * p.G * 64 * 16;
}

extern "C" int dif_sliced_measure(const int32_t* rowptr, const int32_t* blkptr, const int32_t* src, int64_t n_src,
                                  int64_t nnz, int64_t row_begin, int64_t n_rows, int F, const int32_t* plan,
                                  const int32_t* row_order, const uint16_t* parts, int64_t n_pos, uint16_t* sorted,
                                  void* counts, int32_t* lengths, int32_t* table, int32_t* status, dif_stream_t stream) {
    Plan pl;
    if (int rc = check_positions(row_order, parts, n_rows, n_pos, "dif_sliced_measure")) return rc;
    if (int rc = check_plan(plan, n_src, n_pos, F, pl)) return rc;
    DIF_REQUIRE(row_begin >= 0 && row_begin + n_rows <= n_src && nnz >= 0, DIF_E_BADARG, "dif_sliced_measure: bad row range");
    DIF_REQUIRE(rowptr && (nnz == 0 || src) && sorted && counts && lengths && table && status, DIF_E_BADARG,
                "dif_sliced_measure: null pointer");
    DIF_REQUIRE(pl.NT == 1 || blkptr, DIF_E_BADARG,
                "dif_sliced_measure: more than one tile needs the CSR built with n_blocks = plan[7], block_rows = plan[6]");
    DIF_REQUIRE(dif::aligned16(counts), DIF_E_BADARG, "dif_sliced_measure: counts must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t he = hipMemsetAsync(status, 0, 4, st);
    if (he != hipSuccess) return dif::fail(static_cast<int>(he), "dif_sliced_measure: memset: %s", hipGetErrorString(he));
    hipLaunchKernelGGL(sliced_sort_kernel, dim3(static_cast<unsigned>((n_pos + 3) / 4)), dim3(256), 0, st, rowptr,
                       blkptr, src, n_src, pl.NT, pl.T, row_begin, n_pos, row_order, parts, sorted,
                       static_cast<uint4*>(counts), status);
    if (int rc = dif::launch_status("sliced_sort_kernel")) return rc;
    const int64_t n_hw = static_cast<int64_t>(pl.G) * pl.NT * 4;
    hipLaunchKernelGGL((sliced_color_kernel<false>), dim3(static_cast<unsigned>((n_hw + kColorThreads - 1) / kColorThreads)),
                       dim3(kColorThreads), 0, st, n_pos, pl, static_cast<const uint4*>(counts), lengths, nullptr, nullptr);
    if (int rc = dif::launch_status("sliced_color_kernel")) return rc;
    hipLaunchKernelGGL(sliced_table_kernel, dim3(1), dim3(1024), 0, st, lengths, pl, table);
    return dif::launch_status("sliced_table_kernel");
}

extern "C" int dif_sliced_emit(const int32_t* rowptr, const int32_t* blkptr, int64_t n_src, int64_t row_begin,
                               int64_t n_rows, int F, const int32_t* plan, const int32_t* row_order, const uint16_t* parts,
                               int64_t n_pos, const uint16_t* sorted, const void* counts, const int32_t* table,
                               int64_t n_blocks, uint16_t* entries, dif_stream_t stream) {
    Plan pl;
    if (int rc = check_positions(row_order, parts, n_rows, n_pos, "dif_sliced_emit")) return rc;
    if (int rc = check_plan(plan, n_src, n_pos, F, pl)) return rc;
    DIF_REQUIRE(rowptr && sorted && counts && table && entries && n_blocks >= 1, DIF_E_BADARG, "dif_sliced_emit: null pointer");
    DIF_REQUIRE(pl.NT == 1 || blkptr, DIF_E_BADARG, "dif_sliced_emit: more than one tile needs blkptr");
    DIF_REQUIRE(dif::aligned16(entries) && dif::aligned16(counts), DIF_E_BADARG,
                "dif_sliced_emit: entries / counts must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n_hw = static_cast<int64_t>(pl.G) * pl.NT * 4;
    hipLaunchKernelGGL((sliced_color_kernel<true>), dim3(static_cast<unsigned>((n_hw + kColorThreads - 1) / kColorThreads)),
                       dim3(kColorThreads), 0, st, n_pos, pl, static_cast<const uint4*>(counts), nullptr, table, entries);
    if (int rc = dif::launch_status("sliced_color_kernel")) return rc;
    const int64_t n_st = static_cast<int64_t>(pl.G) * pl.NT;                 // one wave per (slot, tile)
    hipLaunchKernelGGL(sliced_fill_kernel, dim3(static_cast<unsigned>((n_st + 3) / 4)), dim3(256), 0, st, rowptr, blkptr,
                       n_src, row_begin, n_pos, row_order, parts, pl, sorted, static_cast<const uint4*>(counts), table,
                       reinterpret_cast<uint4*>(entries));
    return dif::launch_status("sliced_fill_kernel");
}

extern "C" int dif_sliced_prescale_f32(const float* x, int64_t ldx, const int32_t* rowptr, const float* dinv, int64_t n_src,
                                       int F, const int32_t* plan, float* ys, dif_stream_t stream) {
    DIF_REQUIRE(x && rowptr && plan && ys && n_src > 0 && F > 0 && F % 4 == 0 && ldx >= F, DIF_E_BADARG,
                "dif_sliced_prescale: bad argument");
    DIF_REQUIRE(ldx % 4 == 0 && dif::aligned16(x) && dif::aligned16(ys), DIF_E_BADARG,
                "dif_sliced_prescale: x rows and ys must be 16-byte aligned");
    const int slices = plan[0];
    const int64_t npad = static_cast<int64_t>(plan[6]) * plan[7];
    DIF_REQUIRE(slices == F / 4 && npad >= n_src, DIF_E_BADARG, "dif_sliced_prescale: plan does not match F / n_src");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(sliced_prescale_kernel, dim3(static_cast<unsigned>((npad + 63) / 64), (slices + kPreSlices - 1) / kPreSlices),
                       dim3(256), 0, st, x, ldx, rowptr, dinv, n_src, npad, slices, reinterpret_cast<f32x4*>(ys));
    return dif::launch_status("sliced_prescale_kernel");
}

extern "C" int dif_sliced_spmm_f32(const uint16_t* entries, const int32_t* table, const int32_t* plan, const float* ys,
                                   const int32_t* rowptr, const float* dinv, const int32_t* row_order, const uint16_t* parts,
                                   int64_t n_pos, int64_t n_src, int64_t row_begin, int64_t n_rows, int F, const float* attn,
                                   int64_t lda, float attn_scale, float gcn_scale, float* out, int64_t ldo, void* ws,
                                   int64_t ws_bytes, dif_stream_t stream) {
    Plan pl;
    if (int rc = check_positions(row_order, parts, n_rows, n_pos, "dif_sliced_spmm")) return rc;
    if (int rc = check_plan(plan, n_src, n_pos, F, pl)) return rc;
    const int64_t ws_need = pl.S > 1 ? static_cast<int64_t>(pl.S) * pl.slices * pl.G * 64 * 16 : 0;
    DIF_REQUIRE(ws_need == 0 || (ws && ws_bytes >= ws_need && dif::aligned16(ws)), DIF_E_BADARG,
                "dif_sliced_spmm: a row shard's product needs dif_sliced_spmm_workspace_bytes() of 16-byte aligned workspace");
    DIF_REQUIRE(entries && table && ys && rowptr && out, DIF_E_BADARG, "dif_sliced_spmm: null pointer");
    DIF_REQUIRE(row_begin >= 0 && row_begin + n_rows <= n_src, DIF_E_BADARG, "dif_sliced_spmm: row range exceeds n_src");
    DIF_REQUIRE(ldo >= F && ldo % 4 == 0 && dif::aligned16(out) && (!attn || (lda >= F && lda % 4 == 0 && dif::aligned16(attn))),
                DIF_E_BADARG, "dif_sliced_spmm: out / attn rows must be 16-byte aligned with ld >= F");
    DIF_REQUIRE(dif::aligned16(entries) && dif::aligned16(ys), DIF_E_BADARG, "dif_sliced_spmm: entries / ys must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t npad = static_cast<int64_t>(pl.T) * pl.NT;
#ifdef DIF_SLICED_TRACE
    const char* tp = getenv("DIF_SLICED_TRACE");
    DIF_REQUIRE(tp != nullptr, DIF_E_BADARG, "trace build: DIF_SLICED_TRACE = device address of int64[blocks * 2 * tiles * 4]");
    const Epilogue ep = {rowptr, dinv, row_order, parts, row_begin, n_pos, attn, lda, attn_scale, gcn_scale, out, ldo,
                         static_cast<f32x4*>(ws), reinterpret_cast<long long*>(strtoull(tp, nullptr, 0))};
#else
    const Epilogue ep = {rowptr, dinv, row_order, parts, row_begin, n_pos, attn, lda, attn_scale, gcn_scale, out, ldo,
                         static_cast<f32x4*>(ws)};
#endif
    const uint4* e4 = reinterpret_cast<const uint4*>(entries);
    const f32x4* y4 = reinterpret_cast<const f32x4*>(ys);
    switch (pl.R) {
        case 1: return launch_sweep<1>(st, e4, table, y4, npad, pl, ep);
        case 2: return launch_sweep<2>(st, e4, table, y4, npad, pl, ep);
        case 3: return launch_sweep<3>(st, e4, table, y4, npad, pl, ep);
        case 4: return launch_sweep<4>(st, e4, table, y4, npad, pl, ep);
        case 5: return launch_sweep<5>(st, e4, table, y4, npad, pl, ep);
        case 6: return launch_sweep<6>(st, e4, table, y4, npad, pl, ep);
        case 7: return launch_sweep<7>(st, e4, table, y4, npad, pl, ep);
        case 8: return launch_sweep<8>(st, e4, table, y4, npad, pl, ep);
        case 9: return launch_sweep<9>(st, e4, table, y4, npad, pl, ep);
        case 10: return launch_sweep<10>(st, e4, table, y4, npad, pl, ep);
    }
    return dif::fail(DIF_E_SHAPE, "dif_sliced_spmm: %d rounds per wave not covered", pl.R);
}

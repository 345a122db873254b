// a3 (hot half): normalised-adjacency SpMM  -- node classification/difformer.py:75-78
//   out[r,:] = gcn_scale * sum_{e in CSR row r} val_e * x[src_e,:]   (+ attn_scale * attn[r,:])
// One launch covers all H*D feature columns (the reference loops H torch_sparse.matmul calls
// and stacks), and the `attention + gcn` / convex mix of difformer.py:130-134 rides in the
// epilogue so the layer does not re-read both operands.
//
// Gather-bound: per entry one 4*F-byte source row is fetched through L1/L2 (x stays resident in
// L2 / Infinity Cache), against 8 bytes of streamed CSR.  Layout choices:
//   - G = F/4 lanes (rounded to a power of two) each own one float4 of the feature row, so a
//     wave issues 64/G independent 16-byte gathers per instruction, every one a full 16*G-byte
//     contiguous segment of a source row;
//   - CSR src/val are read 64 entries at a time with one coalesced dword load each and handed
//     to the gathering lanes by ds_bpermute (no per-entry index loads);
//   - accumulation stays in registers; the 64/G partial rows are folded with xor-shuffles in a
//     fixed order -> deterministic, no atomics.
// High-degree rows (ogbn-proteins, ~600 entries) use a whole wave per row; low-degree graphs
// (Cora ~5, Pokec batches ~3) use one G-lane group per row.
#include "dif_common.h"

namespace {

using dif::f32x4;
using dif::Elem;

typedef float f32x8 __attribute__((ext_vector_type(8)));

// W floats of a feature row per lane.  W = 8 serves bfloat16 storage in the blocked kernel: 8 elements are one
// 16-byte load, so a 64-wide row is 8 lanes x 16 B = ONE 128-byte line (W = 4 would fetch it with 8-byte loads,
// which the texture path runs at ~0.6x the rate).
template <int W> struct Vec;
template <> struct Vec<8> { using T = f32x8; };
template <> struct Vec<4> { using T = f32x4; };
template <> struct Vec<1> { using T = float; };

template <int W>
__device__ __forceinline__ typename Vec<W>::T vzero() {
    if constexpr (W == 8) return f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    else if constexpr (W == 4) return f32x4{0.f, 0.f, 0.f, 0.f};
    else return 0.f;
}

template <int W>
__device__ __forceinline__ typename Vec<W>::T vload(const float* p) {
    if constexpr (W == 8) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
        return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    } else if constexpr (W == 4) return *reinterpret_cast<const f32x4*>(p);
    else return *p;
}

template <int W>
__device__ __forceinline__ void vstore(float* p, typename Vec<W>::T v) {
    if constexpr (W == 8) {
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else if constexpr (W == 4) *reinterpret_cast<f32x4*>(p) = v;
    else *p = v;
}

// global-memory row pieces in the storage type E (float | dif::bf16); registers and LDS stay fp32
template <int W, typename E>
__device__ __forceinline__ typename Vec<W>::T gload(const E* p) {
    if constexpr (W == 8) {
        if constexpr (sizeof(E) == 2) {                       // 8 bfloat16 = one 16-byte load
            const uint4 r = *reinterpret_cast<const uint4*>(p);
            const uint32_t w[4] = {r.x, r.y, r.z, r.w};
            f32x8 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[2 * i] = __uint_as_float(w[i] << 16);
                v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
            }
            return v;
        } else {
            const f32x4 a = Elem<E>::ld4(p), b = Elem<E>::ld4(p + 4);
            return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        }
    } else if constexpr (W == 4) return Elem<E>::ld4(p);
    else return Elem<E>::ld(p);
}

template <int W, typename E>
__device__ __forceinline__ void gstore(E* p, typename Vec<W>::T v) {
    if constexpr (W == 8) {
        Elem<E>::st4(p, f32x4{v[0], v[1], v[2], v[3]});
        Elem<E>::st4(p + 4, f32x4{v[4], v[5], v[6], v[7]});
    } else if constexpr (W == 4) Elem<E>::st4(p, v);
    else Elem<E>::st(p, v);
}

template <int W>
__device__ __forceinline__ typename Vec<W>::T vshfl_xor(typename Vec<W>::T v, int m) {
    if constexpr (W > 1) {
        typename Vec<W>::T r;
#pragma unroll
        for (int i = 0; i < W; ++i) r[i] = __shfl_xor(v[i], m, 64);
        return r;
    } else {
        return __shfl_xor(v, m, 64);
    }
}

// sum of the W components (left to right) / of their squares
template <int W>
__device__ __forceinline__ float vsum(typename Vec<W>::T v) {
    if constexpr (W > 1) {
        float s = v[0];
#pragma unroll
        for (int i = 1; i < W; ++i) s += v[i];
        return s;
    } else {
        return v;
    }
}

template <int W>
__device__ __forceinline__ float vsumsq(typename Vec<W>::T v) {
    if constexpr (W > 1) {
        float s = v[0] * v[0];
#pragma unroll
        for (int i = 1; i < W; ++i) s += v[i] * v[i];
        return s;
    } else {
        return v * v;
    }
}

// A gathered row piece as it sits in registers between the load and the FMA.  bfloat16 x 8 stays PACKED (4 registers
// instead of 8): eight gathers in flight then cost the same 32 registers as in fp32 -- the blocked kernel is bound by
// the gathers a wave keeps in flight, so unpacking at load time (and halving the unroll to fit the register budget)
// gave no gain over fp32 rows (1.16 ms at C4).
template <int W, typename E>
struct Packed {
    using T = typename Vec<W>::T;
    static __device__ __forceinline__ T load(const E* p) { return gload<W, E>(p); }
    static __device__ __forceinline__ T zero() { return vzero<W>(); }
    static __device__ __forceinline__ typename Vec<W>::T unpack(T r) { return r; }
};
template <>
struct Packed<8, dif::bf16> {
    using T = uint4;
    static __device__ __forceinline__ T load(const dif::bf16* p) { return *reinterpret_cast<const uint4*>(p); }
    static __device__ __forceinline__ T zero() { return uint4{0u, 0u, 0u, 0u}; }
    static __device__ __forceinline__ f32x8 unpack(T r) {
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
        f32x8 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
        return v;
    }
};

// Optional fused layer tail (H == 1 only; difformer.py:139-140, :200-203): applied to the finished
// row `o` held by a G-lane group (W floats per lane, lanes with col >= F inactive).
template <typename E>
struct Tail {
    const E* x0;   int64_t ldx0;   // += x_0            (use_source)
    const E* prev; int64_t ldp;    // alpha-residual    (use_residual)
    float alpha;
    const E* ln_w; const E* ln_b; float eps;   // LayerNorm (use_bn)
    int enabled;
    int relu;                                   // ReLU after the norm (difformer-v2.py:217)
};

template <int G, int W, typename E>
__device__ __forceinline__ typename Vec<W>::T apply_tail(typename Vec<W>::T o, const Tail<E>& t, int64_t row, int col,
                                                         bool ok, int F) {
    using V = typename Vec<W>::T;
    if (ok) {
        if (t.x0) o += gload<W, E>(t.x0 + row * t.ldx0 + col);
        if (t.prev) o = t.alpha * o + (1.0f - t.alpha) * gload<W, E>(t.prev + row * t.ldp + col);
    }
    if (t.ln_w) {   // wave-uniform
        const float inv_d = 1.0f / static_cast<float>(F);
        float s = 0.f;
        if (ok) s = vsum<W>(o);
#pragma unroll
        for (int m = 1; m < G; m <<= 1) s += __shfl_xor(s, m, 64);
        const float mu = s * inv_d;
        V dz = vzero<W>();
        float v = 0.f;
        if (ok) {
            dz = o - mu;
            v = vsumsq<W>(dz);
        }
#pragma unroll
        for (int m = 1; m < G; m <<= 1) v += __shfl_xor(v, m, 64);
        const float rstd = 1.0f / sqrtf(v * inv_d + t.eps);
        if (ok) o = dz * rstd * gload<W, E>(t.ln_w + col) + gload<W, E>(t.ln_b + col);
    }
    if (t.relu) {
        if constexpr (W > 1) { for (int i = 0; i < W; ++i) o[i] = fmaxf(o[i], 0.f); } else o = fmaxf(o, 0.f);
    }
    return o;
}

constexpr int kGatherUnroll = 8;
constexpr int kLongRow = 64;          // entries; longer rows of a low-degree graph are taken by a whole block (spmm_group_row_kernel)

// Element offset of source row `s`: 32-bit arithmetic (one v_mul_lo_u32) when the whole matrix spans < 2^31 elements
// -- the 64-bit form costs five extra VALU instructions per gather.
template <bool WIDE>
__device__ __forceinline__ int64_t row_off(int32_t s, int64_t ldx) {
    if (WIDE) return static_cast<int64_t>(s) * ldx;
    return static_cast<int64_t>(static_cast<uint32_t>(s) * static_cast<uint32_t>(ldx));
}

// ---- whole wave per destination row ---------------------------------------------------------
// grid (row blocks, column chunks of G*W floats); 256 threads = 4 rows in flight per block.
template <int G, int W, typename E>
__global__ __launch_bounds__(256) void spmm_wave_row_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const float* __restrict__ val,
    const E* __restrict__ x, int64_t ldx, int64_t row_begin, int64_t n_rows, int F,
    const E* __restrict__ attn, int64_t lda, float attn_scale, float gcn_scale, Tail<E> tail,
    E* __restrict__ out, int64_t ldo) {
    using V = typename Vec<W>::T;
    constexpr int EPW = 64 / G;  // entries gathered per wave instruction
    const int lane = threadIdx.x & 63;
    const int sub = lane / G;    // which of the EPW concurrent entries
    const int li = lane % G;
    const int col = blockIdx.y * (G * W) + li * W;
    const bool active = col < F;
    const int wave = threadIdx.x >> 6;
    const int64_t gw = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    const int64_t nw = static_cast<int64_t>(gridDim.x) * 4;
    const E* xcol = x + col;
    // HUB ROWS.  A wave walks its row 64 entries per index load, 8 gathers in flight: ~1 us per 32 entries, so the 14,854-entry
    // hub of the real Pokec graph would hold one wave for ~0.5 ms of a 1.3-ms launch and a 120,000-entry row for 4 ms
    // (scripts/exp_regional_order.py, round 5).  Rows beyond kWaveLong entries are left out of the walk and noted in LDS; when the
    // block has finished its ordinary rows its four waves take a noted row together, a quarter of the entries each (cut at
    // multiples of 64), and wave 0 adds the four partial rows in wave order (deterministic) and finishes the row.
    constexpr int kWaveLong = 1024, kNoted = 16;
    __shared__ int64_t s_row[kNoted];
    __shared__ int s_count;
    __shared__ V s_part[3][64];
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();

    auto walk = [&](int32_t a0, int32_t a1) {
        V acc = vzero<W>();
        for (int32_t base = a0; base < a1; base += 64) {
            const int32_t idx = base + lane;
            const int32_t my_src = (idx < a1) ? src[idx] : 0;
            const float my_val = (idx < a1) ? val[idx] : 0.f;
            const int cnt = (a1 - base < 64) ? (a1 - base) : 64;
            const int steps = (cnt + EPW - 1) / EPW;
            for (int j0 = 0; j0 < steps; j0 += kGatherUnroll) {
                V xv[kGatherUnroll];
                float w[kGatherUnroll];
#pragma unroll
                for (int u = 0; u < kGatherUnroll; ++u) {
                    const int ent = ((j0 + u) * EPW + sub) & 63;
                    const int32_t s = __shfl(my_src, ent, 64);
                    w[u] = (j0 + u < steps) ? __shfl(my_val, ent, 64) : 0.f;
                    xv[u] = (active && j0 + u < steps) ? gload<W, E>(xcol + static_cast<int64_t>(s) * ldx) : vzero<W>();
                }
#pragma unroll
                for (int u = 0; u < kGatherUnroll; ++u) acc += w[u] * xv[u];
            }
        }
        // fold the EPW partial rows (fixed tree)
#pragma unroll
        for (int m = G; m < 64; m <<= 1) acc += vshfl_xor<W>(acc, m);
        return acc;
    };
    auto finish = [&](V acc, int64_t row) {
        const bool ok = (sub == 0 && active);
        V o = gcn_scale * acc;
        if (ok && attn) o += attn_scale * gload<W, E>(attn + row * lda + col);
        if (tail.enabled) o = apply_tail<G, W, E>(o, tail, row, col, ok, F);
        if (ok) gstore<W, E>(out + row * ldo + col, o);
    };

    for (int64_t row = gw; row < n_rows; row += nw) {
        const int64_t r = row_begin + row;
        const int32_t e0 = rowptr[r], e1 = rowptr[r + 1];
        if (e1 - e0 > kWaveLong) {                       // wave-uniform
            int slot = 0;
            if (lane == 0) slot = atomicAdd(&s_count, 1);
            slot = __shfl(slot, 0, 64);
            if (slot < kNoted) {
                if (lane == 0) s_row[slot] = row;
                continue;
            }
        }
        finish(walk(e0, e1), row);
    }
    __syncthreads();
    const int noted = s_count < kNoted ? s_count : kNoted;
    for (int k = 0; k < noted; ++k) {
        const int64_t row = s_row[k];
        const int64_t r = row_begin + row;
        const int32_t e0 = rowptr[r], e1 = rowptr[r + 1];
        const int32_t chunk = (((e1 - e0 + 3) / 4) + 63) & ~63;
        const int32_t a0 = e0 + wave * chunk < e1 ? e0 + wave * chunk : e1;
        const int32_t a1 = a0 + chunk < e1 ? a0 + chunk : e1;
        V acc = walk(a0, a1);
        if (wave > 0) s_part[wave - 1][lane] = acc;
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w2 = 0; w2 < 3; ++w2) acc += s_part[w2][lane];
            finish(acc, row);
        }
        __syncthreads();
    }
}

// ---- one G-lane group per destination row (low-degree graphs) ---------------------------------
template <int G, int W, typename E>
__global__ __launch_bounds__(256) void spmm_group_row_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const float* __restrict__ val,
    const E* __restrict__ x, int64_t ldx, int64_t row_begin, int64_t n_rows, int F,
    const E* __restrict__ attn, int64_t lda, float attn_scale, float gcn_scale, Tail<E> tail,
    E* __restrict__ out, int64_t ldo) {
    using V = typename Vec<W>::T;
    constexpr int RPB = 256 / G;  // rows per block
    const int li = threadIdx.x % G;
    const int col = blockIdx.y * (G * W) + li * W;
    const bool active = col < F;
    const E* xcol = x + col;
    // LONG ROWS.  A lane group walks its row four entries at a time; a hub row (citation and social graphs have rows of
    // hundreds to thousands of entries among rows of three) would hold its group -- and with it the kernel -- for ~0.3 us per
    // entry (a 4,100-entry row: 1.2 ms at 100,000 rows, scripts/exp_layer_gather.py hubs).  Rows longer than kLongRow entries
    // are therefore left out of the walk and noted in LDS; when the block has finished its ordinary rows, ALL its lane groups
    // take the noted rows together, striding through a row four entries each, and group 0 adds their partial rows in group
    // order (deterministic) and finishes the row.  The ordinary rows pay one LDS counter and one barrier per block.
    constexpr int NG = RPB;              // lane groups per block
    constexpr int kNoted = 64;           // long rows a block can defer (further ones are walked in place)
    __shared__ int64_t s_row[kNoted];
    __shared__ int s_count;
    __shared__ V s_part[256];
    const int grp = threadIdx.x / G;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    auto walk = [&](int32_t a0, int32_t a1, int32_t step) {
        V sum = vzero<W>();
        for (int32_t e = a0; e < a1; e += step) {
            V xv[4];
            float w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = e + u < a1;
                const int32_t s = ok ? src[e + u] : 0;
                w[u] = ok ? val[e + u] : 0.f;
                xv[u] = (ok && active) ? gload<W, E>(xcol + static_cast<int64_t>(s) * ldx) : vzero<W>();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) sum += w[u] * xv[u];
        }
        return sum;
    };
    auto finish = [&](V acc, int64_t row, bool ok) {
        V o = gcn_scale * acc;
        if (ok && attn) o += attn_scale * gload<W, E>(attn + row * lda + col);
        if (tail.enabled) o = apply_tail<G, W, E>(o, tail, row, col, ok, F);
        if (ok) gstore<W, E>(out + row * ldo + col, o);
    };
    const int64_t nrb = (n_rows + RPB - 1) / RPB;
    for (int64_t rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        const int64_t row = rb * RPB + grp;
        const bool rok = row < n_rows;
        const int64_t r = row_begin + (rok ? row : 0);
        const int32_t e0 = rok ? rowptr[r] : 0, e1 = rok ? rowptr[r + 1] : 0;
        if (e1 - e0 > kLongRow) {
            int slot = 0;
            if (li == 0) slot = atomicAdd(&s_count, 1);
            slot = __shfl(slot, (threadIdx.x & 63) / G * G, 64);          // the group's first lane
            if (slot < kNoted) {
                if (li == 0) s_row[slot] = row;
                continue;
            }
        }
        finish(walk(e0, e1, 4), row, active && rok);
    }
    __syncthreads();
    const int noted = s_count < kNoted ? s_count : kNoted;             // the same in every thread
    for (int j = 0; j < noted; ++j) {
        const int64_t row = s_row[j];
        const int64_t r = row_begin + row;
        s_part[threadIdx.x] = walk(rowptr[r] + 4 * grp, rowptr[r + 1], 4 * NG);
        __syncthreads();
        V sum = vzero<W>();
        if (grp == 0) {
            sum = s_part[li];
            for (int q = 1; q < NG; ++q) sum += s_part[q * G + li];
        }
        if (threadIdx.x < 64) finish(sum, row, grp == 0 && active);      // the first wave: group 0 holds the row
        __syncthreads();
    }
}

// ---- source-blocked sweep (large dense graphs: x does not fit the 4 MiB L2 of an XCD) -------------
// CSR rows are grouped by source block (gcn_csr.hip).  One persistent 16-wave workgroup per CU;
// every wave owns a panel of `rpw` consecutive destination rows whose accumulators live in its
// slice of LDS, and sweeps the source blocks 0..NB-1 in order.  All CUs therefore gather from the
// same ~2.5 MiB slice of x at about the same time and the slice is served by L2 instead of the
// Infinity Cache (measured on MI355X: 17.7 TB/s of gathers when the touched span is <= 4 MiB
// against 8.0 TB/s over the whole 32 MiB -- scripts/exp_spmm_locality.py).
// Inside a block visit the wave's S = 64/G lane groups each walk one row's group of entries:
// G entries are fetched per lane group with one coalesced (non-temporal) load of src/val, then
// handed out by ds_bpermute, so there is no per-entry index load and no cross-lane reduction.
// Which source blocks a launch sweeps, and where its accumulators start / end up.  A row-sharded run (dist.py) splits
// the product in two launches so that the all-gather of the value rows hides behind the first one:
//   part 0: only the blocks [first, end) whose sources are this rank's OWN rows (available before the collective),
//           sums written to `acc_out` (one fp32 row of G*W floats per shard row) instead of the epilogue;
//   part 1: all other blocks (skip_lo..skip_hi-1 left out), accumulators preloaded from `acc_in`, normal epilogue.
// The two parts may use different launch geometries (rows are parked by row index).  A persistent workgroup of this
// kernel takes a whole CU (16 waves x 128 VGPRs, 144 KiB LDS): nothing else can share it, so part 0 runs on fewer
// workgroups than there are CUs and the all-gather's kernel keeps (or finds) CUs of its own.
// A whole product is {0, -1, -1, n_blocks, nullptr, nullptr, 0}.
struct Sweep {
    int first, skip_lo, skip_hi, end;
    const float* acc_in;
    float* acc_out;
    int max_wgs;      // 0 = one workgroup per CU; part 0 leaves CUs free for the collective's own kernel (see below)
};

constexpr int kBlkWaves = 16;          // waves per workgroup
constexpr int kBlkLdsBytesPerCU = 147456;  // 144 KiB of the CU's 160 KiB for accumulators
constexpr int kBlkPre = 4;             // G-entry chunks of a (row, block) group fetched in one batch

// WPC = workgroups per CU (1: 16 waves/CU, 128-VGPR budget; 2: 32 waves/CU, 64-VGPR budget)
// ORD: walk the rows through `order` (degree-sorted, hub rows split) -- a separate instantiation, so graphs that do
// not need it run the plain contiguous-panel code.
template <int G, int W, int WPC, int UNROLL, bool PACE, bool WIDE, bool ORD, typename E>
__global__ __launch_bounds__(64 * kBlkWaves, WPC * kBlkWaves / 4) void spmm_blocked_kernel(
    const int32_t* __restrict__ blkptr, int64_t n_nodes, int n_blocks, const int32_t* __restrict__ src,
    const float* __restrict__ val, const E* __restrict__ x, int64_t ldx, int64_t row_begin, int64_t n_rows,
    int F, const E* __restrict__ attn, int64_t lda, float attn_scale, float gcn_scale, Tail<E> tail,
    E* __restrict__ out, int64_t ldo, int rpw, const int32_t* __restrict__ order, int64_t n_split, Sweep sw) {
    using V = typename Vec<W>::T;
    constexpr int S = 64 / G;       // rows walked concurrently by one wave
    constexpr int kPre = (G >= 16) ? kBlkPre : 64 / G;   // chunks prefetched per group: >= 64 entries (6 chunks: +3 %)
    constexpr int RW = G * W;       // floats of LDS per accumulator row
    constexpr int kLdsFloats = kBlkLdsBytesPerCU / 4 / WPC;
    __shared__ __attribute__((aligned(16))) float acc_lds[kLdsFloats];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int slot = lane / G;
    const int li = lane % G;
    const int col = li * W;
    const bool active = col < F;
    const E* xcol = x + (active ? col : 0);     // inactive column lanes read (and discard) column 0
    float* my = acc_lds + wave * rpw * RW;

    // Panels.  A "quad" is the S rows a wave walks side by side; a panel is rpw / S quads.
    //  * order == nullptr (degrees about equal): panel p = rows [p*rpw, (p+1)*rpw) -- contiguous CSR, fewest TLB entries.
    //  * order != nullptr (skewed degrees; real graphs: ogbn-proteins has max / mean degree ~13): a quad costs the
    //    LONGEST of its groups and the block barrier costs the slowest wave, so `order` lists the shard's rows by
    //    descending degree (dif_row_order), a quad is S consecutive entries of it (similar lengths) and the quads are
    //    dealt round-robin over the panels (every wave gets one quad of each degree stratum).  The first n_split rows of
    //    the order (degree > 4x mean) each take a WHOLE quad: every lane group walks a quarter of each of the row's
    //    groups and the epilogue adds the S partial rows.  Zipf-profile C4: 3.05 ms natural order, see
    //    profiles/r01_experiments.md for the ordered / split figures.
    const int qpp = rpw / S;                                    // ORD: quads per panel (rpw is a multiple of S then)
    const int64_t n_quads = n_split + (n_rows - n_split + S - 1) / S;
    const int64_t n_panels = ORD ? (n_quads + qpp - 1) / qpp : (n_rows + rpw - 1) / rpw;
    // panels are dealt round-robin over the workgroups (panel p -> workgroup p % grid, wave (p / grid) % 16), so every
    // CU carries the same number of panels to within one whatever the panel count is (vs 16 consecutive panels per
    // workgroup: 1.165 -> 1.125 ms at C4, 0.271 -> 0.181 ms on a 16.5k-row shard; PMC fetch 1.4 -> 2.1 GB per launch)
    const int64_t first = static_cast<int64_t>(wave) * gridDim.x + blockIdx.x;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlkWaves;
    // every wave runs the same number of panel rounds (a wave without a panel only paces the barriers)
    const int64_t rounds = (n_panels + stride - 1) / stride;
    for (int64_t rnd = 0; rnd < rounds; ++rnd) {
        const int64_t panel = first + rnd * stride;
        const bool has = panel < n_panels;
        // lane i <-> panel slot i: member i % S of the panel's quad i / S
        const int64_t myquad = static_cast<int64_t>(lane / S) * n_panels + panel;      // ORD only
        const bool split = ORD && myquad < n_split;             // this slot holds a quarter of a split row
        const int64_t mypos = !ORD ? panel * rpw + lane : split ? myquad : n_split + (myquad - n_split) * S + (lane % S);
        const bool mine = has && lane < rpw && (!ORD || myquad < n_quads) && mypos < n_rows;
        const int32_t lrow = mine ? (ORD ? order[mypos] : static_cast<int32_t>(mypos)) : 0;   // row inside the shard
        // the part of a (row, block) group [e0, e1) this slot walks
        auto part_of = [&](int32_t& e0, int32_t& e1) {
            if (ORD && split) {
                const int32_t len = e1 - e0;
                const int32_t chunk = ((len + S * G - 1) / (S * G)) * G;
                e0 += (lane % S) * chunk;
                if (e0 > e1) e0 = e1;
                if (e1 - e0 > chunk) e1 = e0 + chunk;
            }
        };
        int nq = 0;
        if (has) {
            if (ORD) {
                const int64_t left = (n_quads - panel + n_panels - 1) / n_panels;   // q with q*n_panels + panel < n_quads
                nq = left < qpp ? static_cast<int>(left) : qpp;
            } else {
                const int64_t left = n_rows - panel * rpw;
                nq = static_cast<int>(((left < rpw ? left : rpw) + S - 1) / S);
            }
        }
        for (int i = lane; i < rpw * RW; i += 64) my[i] = 0.f;
        if (sw.acc_in) {        // part 1 of a split product: start from the sums part 0 parked, one fp32 row per shard row
            for (int q = 0; q < nq; ++q) {
                const int rl = q * S + slot;
                const int flags = __shfl(static_cast<int>(mine) | (static_cast<int>(split) << 1), rl, 64);
                const int64_t row = __shfl(lrow, rl, 64);
                // a split row keeps S partial sums: the parked total goes to lane group 0, the others start at zero
                if ((flags & 1) && active && (!(flags & 2) || slot == 0))
                    vstore<W>(my + rl * RW + col, vload<W>(sw.acc_in + row * RW + col));
            }
        }

        // (row, block) group pointers of the current and of the next block, one lane per row of the panel
        const int32_t* pb = blkptr + row_begin + lrow;
        int32_t e0v = 0, e1v = 0;
        if (sw.first < sw.end) {
            const int32_t* p0 = pb + static_cast<int64_t>(sw.first) * n_nodes;
            e0v = mine ? p0[0] : 0;
            e1v = mine ? p0[n_nodes] : 0;
        }
        part_of(e0v, e1v);
        // entries of the NEXT row-quad (of this block, or the first quad of the next block) are always in flight
        // while the current quad gathers
        int32_t e0n = __shfl(e0v, slot, 64), e1n = __shfl(e1v, slot, 64);
        int32_t sn[kPre];
        float wn[kPre];
#pragma unroll
        for (int c = 0; c < kPre; ++c) {
            const int32_t idx = e0n + c * G + li;
            const bool ok = idx < e1n;
            sn[c] = ok ? __builtin_nontemporal_load(src + idx) : 0;
            wn[c] = ok ? __builtin_nontemporal_load(val + idx) : 0.f;
        }
        for (int b = sw.first, nb; b < sw.end; b = nb) {
            nb = (b + 1 == sw.skip_lo) ? sw.skip_hi : b + 1;       // next block of this launch's sweep
            // pace the workgroup's 16 waves block by block: without it they drift apart over the sweep, the
            // XCD's L2 has to hold two source blocks and ~24 % of the gathers miss (PMC FETCH_SIZE per launch
            // at C4: 5.8 GB unpaced -> 1.34 GB paced, 1.0 GB compulsory; 1.29 -> 1.19 ms)
            if (PACE) __syncthreads();
            int32_t e0x = 0, e1x = 0;     // pointers of block b+1, requested a whole block ahead
            if (nb < sw.end) {
                const int32_t* pn = pb + static_cast<int64_t>(nb) * n_nodes;
                e0x = mine ? pn[0] : 0;
                e1x = mine ? pn[n_nodes] : 0;
                part_of(e0x, e1x);
            }
            for (int q = 0; q < nq; ++q) {
                const int rl = q * S + slot;
                const int32_t e0 = e0n, e1 = e1n;
                int32_t sv[kPre];
                float wv[kPre];
#pragma unroll
                for (int c = 0; c < kPre; ++c) { sv[c] = sn[c]; wv[c] = wn[c]; }
                if (q + 1 < nq || nb < sw.end) {
                    if (q + 1 < nq) {
                        e0n = __shfl(e0v, rl + S, 64);
                        e1n = __shfl(e1v, rl + S, 64);
                    } else {
                        e0n = __shfl(e0x, slot, 64);
                        e1n = __shfl(e1x, slot, 64);
                    }
#pragma unroll
                    for (int c = 0; c < kPre; ++c) {
                        const int32_t idx = e0n + c * G + li;
                        const bool ok = idx < e1n;
                        sn[c] = ok ? __builtin_nontemporal_load(src + idx) : 0;
                        wn[c] = ok ? __builtin_nontemporal_load(val + idx) : 0.f;
                    }
                }
                const int len = e1 - e0;
                int maxlen = len;
#pragma unroll
                for (int m = G; m < 64; m <<= 1) {
                    const int o = __shfl_xor(maxlen, m, 64);
                    maxlen = o > maxlen ? o : maxlen;
                }
                if (maxlen <= 0) continue;
                V acc = vzero<W>();
                for (int c0 = 0; c0 < maxlen; c0 += kPre * G) {
                    if (c0 > 0) {  // rare: a group longer than kPre*G entries
#pragma unroll
                        for (int c = 0; c < kPre; ++c) {
                            const int32_t idx = e0 + c0 + c * G + li;
                            const bool ok = idx < e1;
                            sv[c] = ok ? __builtin_nontemporal_load(src + idx) : 0;
                            wv[c] = ok ? __builtin_nontemporal_load(val + idx) : 0.f;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < kPre; ++c) {
#pragma unroll
                        for (int j0 = 0; j0 < G; j0 += UNROLL) {
                            if (c0 + c * G + j0 < maxlen) {        // wave-uniform
                                typename Packed<W, E>::T xv[UNROLL];
                                float w[UNROLL];
#pragma unroll
                                for (int u = 0; u < UNROLL; ++u) {
                                    const int from = slot * G + j0 + u;
                                    // lanes past the end of their group carry weight 0 (masked at the entry load) and
                                    // skip the gather (measured: fetching a fallback row instead costs +6 % and
                                    // +0.7 GB of L2 misses per launch)
                                    const bool take = active && (c0 + c * G + j0 + u < len);
                                    const int32_t sidx = __shfl(sv[c], from, 64);
                                    w[u] = __shfl(wv[c], from, 64);
                                    xv[u] = take ? Packed<W, E>::load(xcol + row_off<WIDE>(sidx, ldx)) : Packed<W, E>::zero();
                                }
#pragma unroll
                                for (int u = 0; u < UNROLL; ++u) acc += w[u] * Packed<W, E>::unpack(xv[u]);
                            }
                        }
                    }
                }
                if (len > 0 && active) {
                    float* a = my + rl * RW + col;
                    vstore<W>(a, vload<W>(a) + acc);
                }
            }
            e0v = e0x;
            e1v = e1x;
        }
        if (sw.acc_out) {       // part 0 of a split product: park the sums row by row (fp32), no epilogue
            for (int q = 0; q < nq; ++q) {
                const int rl = q * S + slot;
                const int flags = __shfl(static_cast<int>(mine) | (static_cast<int>(split) << 1), rl, 64);
                const int64_t row = __shfl(lrow, rl, 64);
                if (!((flags & 1) && active)) continue;
                V t = vload<W>(my + rl * RW + col);
                if (ORD && (flags & 2)) {
                    t = vload<W>(my + (q * S) * RW + col);
#pragma unroll
                    for (int p2 = 1; p2 < S; ++p2) t += vload<W>(my + (q * S + p2) * RW + col);
                }
                if (!(flags & 2) || slot == 0) vstore<W>(sw.acc_out + row * RW + col, t);
            }
            continue;
        }
        // panel epilogue: S rows per step, each a full contiguous row segment
        for (int q = 0; q < nq; ++q) {
            const int rl = q * S + slot;
            const int flags = __shfl(static_cast<int>(mine) | (static_cast<int>(split) << 1), rl, 64);
            const bool ok = (flags & 1) && active;
            const int64_t row = __shfl(lrow, rl, 64);
            V o = vzero<W>();
            if (ok) {
                if (ORD && (flags & 2)) {       // split row: the S lane groups each hold a partial sum; all add them, group 0 stores
                    V t = vload<W>(my + (q * S) * RW + col);
#pragma unroll
                    for (int p2 = 1; p2 < S; ++p2) t += vload<W>(my + (q * S + p2) * RW + col);
                    o = gcn_scale * t;
                } else {
                    o = gcn_scale * vload<W>(my + rl * RW + col);
                }
                if (attn) o += attn_scale * gload<W, E>(attn + row * lda + col);
            }
            if (tail.enabled) o = apply_tail<G, W, E>(o, tail, row, col, ok, F);
            if (ok && (!ORD || !(flags & 2) || slot == 0)) gstore<W, E>(out + row * ldo + col, o);
        }
    }
}

template <int G, int W, int WPC, int UNROLL, typename E>
int launch_blocked_v(hipStream_t st, const int32_t* blkptr, int64_t n_nodes, int n_blocks, const int32_t* src,
                     const float* val, const E* x, int64_t ldx, int64_t row_begin, int64_t n_rows, int F,
                     const E* attn, int64_t lda, float attn_scale, float gcn_scale, const Tail<E>& tail, E* out,
                     int64_t ldo, const int32_t* order, int64_t n_split, const Sweep& sw) {
    constexpr int S = 64 / G;
    constexpr int RW = G * W;
    constexpr int kLdsFloats = kBlkLdsBytesPerCU / 4 / WPC;
    int64_t n_wg = static_cast<int64_t>(dif::kCUs) * WPC;                     // one persistent workgroup per CU
    if (sw.max_wgs > 0 && sw.max_wgs < n_wg) n_wg = sw.max_wgs;
    const int64_t slots = n_wg * kBlkWaves;                                      // waves resident on the chip
    const int rpw_max = (kLdsFloats / kBlkWaves / RW < 64) ? kLdsFloats / kBlkWaves / RW : 64;
    // rows per wave panel: the busiest CU does ceil(panels / workgroups) panels of ceil(rpw / S) row-quad visits per
    // block; pick the panel height near n_rows / slots that minimises that product (ties -> shorter panels: more waves busy).
    // Measured on a 16,567-row shard of C4: 4 rows/wave (two rounds) 0.249 ms, 6-8 rows 0.215 ms, 16 rows 0.371 ms.
    int64_t lo = (n_rows + 3 * (order ? n_split : 0) + slots - 1) / slots;   // a split row fills S slots
    if (lo < 1) lo = 1;
    int64_t rpw = 0, best = -1;
    if (!order) n_split = 0;
    const int64_t n_quads = n_split + (n_rows - n_split + S - 1) / S;
    const int64_t step = order ? S : 1;                       // ordered walk: whole quads only
    for (int64_t cand = order ? ((lo + S - 1) / S) * S : (lo > S ? lo - S + 1 : 1); cand <= lo + 3 * S; cand += step) {
        if (cand > rpw_max) break;
        const int64_t panels = order ? (n_quads + cand / S - 1) / (cand / S) : (n_rows + cand - 1) / cand;
        const int64_t wgs = panels < n_wg ? panels : n_wg;
        // a second round of panels re-runs the whole block sweep (with its barriers) for a few stragglers: avoid
        const int64_t rounds = (panels + wgs * kBlkWaves - 1) / (wgs * kBlkWaves);
        const int64_t cost = (rounds - 1) * 1000000 + ((panels + wgs - 1) / wgs) * ((cand + S - 1) / S);
        if (best < 0 || cost < best) { best = cost; rpw = cand; }
    }
    if (rpw == 0) rpw = (rpw_max / S) * S;
    const int64_t n_panels = order ? (n_quads + rpw / S - 1) / (rpw / S) : (n_rows + rpw - 1) / rpw;
    int64_t grid = n_panels < n_wg ? n_panels : n_wg;
    const bool wide = n_nodes * ldx >= (int64_t(1) << 31);
#define DIF_BLK_LAUNCH(WIDEV, ORDV) \
    hipLaunchKernelGGL((spmm_blocked_kernel<G, W, WPC, UNROLL, true, WIDEV, ORDV, E>), dim3(static_cast<unsigned>(grid)), \
                       dim3(64 * kBlkWaves), 0, st, blkptr, n_nodes, n_blocks, src, val, x, ldx, row_begin, n_rows, F, attn, \
                       lda, attn_scale, gcn_scale, tail, out, ldo, static_cast<int>(rpw), order, n_split, sw)
    if (order) { if (wide) DIF_BLK_LAUNCH(true, true); else DIF_BLK_LAUNCH(false, true); }
    else { if (wide) DIF_BLK_LAUNCH(true, false); else DIF_BLK_LAUNCH(false, false); }
#undef DIF_BLK_LAUNCH
    return dif::launch_status("spmm_blocked_kernel");
}

template <int G, int W, typename E>
int launch_blocked(hipStream_t st, const int32_t* blkptr, int64_t n_nodes, int n_blocks, const int32_t* src,
                   const float* val, const E* x, int64_t ldx, int64_t row_begin, int64_t n_rows, int F,
                   const E* attn, int64_t lda, float attn_scale, float gcn_scale, const Tail<E>& tail, E* out,
                   int64_t ldo, const int32_t* order, int64_t n_split, const Sweep& sw) {
    // 1 workgroup (16 waves) per CU, 8 gathers in flight per wave: best of the measured variants
    // (2 workgroups per CU need a 64-VGPR budget and spill)
    return launch_blocked_v<G, W, 1, 8, E>(st, blkptr, n_nodes, n_blocks, src, val, x, ldx, row_begin, n_rows, F, attn,
                                        lda, attn_scale, gcn_scale, tail, out, ldo, order, n_split, sw);
}

template <int G, int W, typename E>
int launch(bool wave_mode, hipStream_t st, const int32_t* rowptr, const int32_t* src, const float* val,
           const E* x, int64_t ldx, int64_t row_begin, int64_t n_rows, int F, const E* attn, int64_t lda,
           float attn_scale, float gcn_scale, const Tail<E>& tail, E* out, int64_t ldo) {
    const int gy = (F + G * W - 1) / (G * W);
    const int64_t cap = 8 * dif::kCUs;
    if (wave_mode) {
        int64_t gx = (n_rows + 3) / 4;
        if (gx > cap) gx = cap;
        hipLaunchKernelGGL((spmm_wave_row_kernel<G, W, E>), dim3(static_cast<unsigned>(gx), gy), dim3(256), 0, st, rowptr,
                           src, val, x, ldx, row_begin, n_rows, F, attn, lda, attn_scale, gcn_scale, tail, out, ldo);
    } else {
        constexpr int RPB = 256 / G;
        int64_t gx = (n_rows + RPB - 1) / RPB;
        if (gx > cap) gx = cap;
        hipLaunchKernelGGL((spmm_group_row_kernel<G, W, E>), dim3(static_cast<unsigned>(gx), gy), dim3(256), 0, st,
                           rowptr, src, val, x, ldx, row_begin, n_rows, F, attn, lda, attn_scale, gcn_scale, tail, out, ldo);
    }
    return dif::launch_status("spmm kernel");
}

}  // namespace

template <typename E>
static int spmm_dispatch(bool wave_mode, hipStream_t st, const int32_t* rowptr, const int32_t* src,
                         const float* val, const E* x, int64_t ldx, int64_t row_begin, int64_t n_rows, int F,
                         const E* attn, int64_t lda, float attn_scale, float gcn_scale, const Tail<E>& tail,
                         E* out, int64_t ldo, bool vec) {
#define DIF_SPMM(G, W) \
    return launch<G, W, E>(wave_mode, st, rowptr, src, val, x, ldx, row_begin, n_rows, F, attn, lda, attn_scale, gcn_scale, tail, out, ldo)
    if (vec) {
        const int q = F / 4;
        if (q <= 1) DIF_SPMM(1, 4);
        if (q <= 2) DIF_SPMM(2, 4);
        if (q <= 4) DIF_SPMM(4, 4);
        if (q <= 8) DIF_SPMM(8, 4);
        if (q <= 16) DIF_SPMM(16, 4);
        if (q <= 32) DIF_SPMM(32, 4);
        DIF_SPMM(64, 4);
    } else {
        if (F <= 4) DIF_SPMM(4, 1);
        if (F <= 8) DIF_SPMM(8, 1);
        if (F <= 16) DIF_SPMM(16, 1);
        if (F <= 32) DIF_SPMM(32, 1);
        DIF_SPMM(64, 1);
    }
#undef DIF_SPMM
}

template <typename E>
static int spmm_entry(const int32_t* rowptr, const int32_t* blkptr, int n_blocks, const int32_t* src, const float* val,
                      int64_t n_nodes, int64_t nnz, const E* x, int64_t ldx, int64_t row_begin, int64_t n_rows,
                      int F, const E* attn, int64_t lda, float attn_scale, float gcn_scale, const Tail<E>& tail,
                      E* out, int64_t ldo, const int32_t* order, int64_t n_split, dif_stream_t stream,
                      const Sweep* part = nullptr) {
    DIF_REQUIRE(n_rows > 0 && F > 0 && row_begin >= 0 && n_nodes > 0 && nnz >= 0 && n_blocks >= 1, DIF_E_BADARG,
                "dif_gcn_spmm: need n_rows > 0, F > 0, row_begin >= 0, n_nodes > 0, nnz >= 0, n_blocks >= 1");
    DIF_REQUIRE(row_begin + n_rows <= n_nodes, DIF_E_BADARG, "dif_gcn_spmm: row range exceeds n_nodes");
    DIF_REQUIRE(rowptr && x && out && (nnz == 0 || (src && val)), DIF_E_BADARG, "dif_gcn_spmm: null pointer");
    DIF_REQUIRE(n_blocks == 1 || blkptr, DIF_E_BADARG, "dif_gcn_spmm: n_blocks > 1 needs blkptr");
    DIF_REQUIRE(n_split >= 0 && n_split <= n_rows && (order || n_split == 0), DIF_E_BADARG,
                "dif_gcn_spmm: n_split_rows must lie in [0, n_rows] and needs row_order");
    DIF_REQUIRE(ldx >= F && ldo >= F && (!attn || lda >= F), DIF_E_BADARG,
                "dif_gcn_spmm: leading dimension smaller than a row");
    DIF_REQUIRE((F + 255) / 256 <= 65535, DIF_E_RANGE, "dif_gcn_spmm: F too large");
    bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && (!attn || lda % 4 == 0) &&
               dif::aligned_v4<E>(x) && dif::aligned_v4<E>(out) && (!attn || dif::aligned_v4<E>(attn));
    if (tail.enabled) {
        DIF_REQUIRE((tail.ln_w == nullptr) == (tail.ln_b == nullptr), DIF_E_BADARG,
                    "dif_gcn_spmm_tail: ln_weight and ln_bias must be given together");
        DIF_REQUIRE((!tail.x0 || tail.ldx0 >= F) && (!tail.prev || tail.ldp >= F), DIF_E_BADARG,
                    "dif_gcn_spmm_tail: leading dimension smaller than a row");
        vec = vec && (!tail.x0 || (tail.ldx0 % 4 == 0 && dif::aligned_v4<E>(tail.x0))) &&
              (!tail.prev || (tail.ldp % 4 == 0 && dif::aligned_v4<E>(tail.prev))) &&
              (!tail.ln_w || (dif::aligned_v4<E>(tail.ln_w) && dif::aligned_v4<E>(tail.ln_b)));
        // the LayerNorm statistics are folded inside one lane group: the whole row must fit it
        DIF_REQUIRE(F <= (vec ? 256 : 64), DIF_E_SHAPE, "dif_gcn_spmm_tail: fused tail needs F <= %d here",
                    vec ? 256 : 64);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Sweep sw = part ? *part : Sweep{0, -1, -1, n_blocks, nullptr, nullptr, 0};
    DIF_REQUIRE(!part || (n_blocks > 1 && vec && F <= 256 && nnz > 0), DIF_E_SHAPE,
                "dif_gcn_spmm_part: split products run on the blocked kernel only (n_blocks > 1, F %% 4 == 0, F <= 256, aligned)");
    if (n_blocks > 1 && vec && F <= 256 && nnz > 0) {
#define DIF_BLK(G, W) \
    return launch_blocked<G, W, E>(st, blkptr, n_nodes, n_blocks, src, val, x, ldx, row_begin, n_rows, F, attn, lda, \
                                   attn_scale, gcn_scale, tail, out, ldo, order, n_split, sw)
        if constexpr (sizeof(E) == 2) {
            // bfloat16 rows: 8 elements per lane = 16-byte gathers (a 64-wide row is one 128-byte line)
            auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
            const bool v8 = (F % 8 == 0) && (ldx % 8 == 0) && (ldo % 8 == 0) && (!attn || lda % 8 == 0) && a16(x) &&
                            a16(out) && (!attn || a16(attn)) && (!tail.x0 || (tail.ldx0 % 8 == 0 && a16(tail.x0))) &&
                            (!tail.prev || (tail.ldp % 8 == 0 && a16(tail.prev))) &&
                            (!tail.ln_w || (a16(tail.ln_w) && a16(tail.ln_b)));
            if (v8) {
                if (F <= 64) DIF_BLK(8, 8);
                if (F <= 128) DIF_BLK(16, 8);
                DIF_BLK(32, 8);
            }
        }
        if (F <= 64) DIF_BLK(16, 4);
        if (F <= 128) DIF_BLK(32, 4);
        DIF_BLK(64, 4);
#undef DIF_BLK
    }
    // row mapping: a whole wave per row pays off once a row keeps the wave's gather slots busy
    const bool wave_mode = nnz / n_nodes >= 16;
    return spmm_dispatch<E>(wave_mode, st, rowptr, src, val, x, ldx, row_begin, n_rows, F, attn, lda, attn_scale,
                            gcn_scale, tail, out, ldo, vec);
}

extern "C" int dif_gcn_spmm_f32(const int32_t* rowptr, const int32_t* blkptr, int n_blocks, const int32_t* src,
                                const float* val, int64_t n_nodes, int64_t nnz, const float* x, int64_t ldx,
                                int64_t row_begin, int64_t n_rows, int F, const float* attn, int64_t lda,
                                float attn_scale, float gcn_scale, const int32_t* row_order, int64_t n_split_rows,
                                float* out, int64_t ldo, dif_stream_t stream) {
    Tail<float> tail = {};
    return spmm_entry<float>(rowptr, blkptr, n_blocks, src, val, n_nodes, nnz, x, ldx, row_begin, n_rows, F, attn, lda,
                             attn_scale, gcn_scale, tail, out, ldo, row_order, n_split_rows, stream);
}

extern "C" int dif_gcn_spmm_tail_f32(const int32_t* rowptr, const int32_t* blkptr, int n_blocks, const int32_t* src,
                                     const float* val, int64_t n_nodes, int64_t nnz, const float* x, int64_t ldx,
                                     int64_t row_begin, int64_t n_rows, int F, const float* attn, int64_t lda,
                                     float attn_scale, float gcn_scale, const int32_t* row_order, int64_t n_split_rows,
                                     const float* x0, int64_t ldx0,
                                     const float* prev, int64_t ldp, float alpha, const float* ln_weight,
                                     const float* ln_bias, float ln_eps, int relu, float* out, int64_t ldo,
                                     dif_stream_t stream) {
    Tail<float> tail = {x0, ldx0, prev, ldp, alpha, ln_weight, ln_bias, ln_eps, 1, relu};
    return spmm_entry<float>(rowptr, blkptr, n_blocks, src, val, n_nodes, nnz, x, ldx, row_begin, n_rows, F, attn, lda,
                             attn_scale, gcn_scale, tail, out, ldo, row_order, n_split_rows, stream);
}

// bfloat16 storage: x / attn / x0 / prev / LayerNorm parameters / out are bf16, `val` and every accumulation fp32.
// tail_enabled = 0 ignores the tail arguments (plain SpMM + combine).
extern "C" int dif_gcn_spmm_tail_bf16(const int32_t* rowptr, const int32_t* blkptr, int n_blocks, const int32_t* src,
                                      const float* val, int64_t n_nodes, int64_t nnz, const void* x, int64_t ldx,
                                      int64_t row_begin, int64_t n_rows, int F, const void* attn, int64_t lda,
                                      float attn_scale, float gcn_scale, const int32_t* row_order, int64_t n_split_rows,
                                      int tail_enabled, const void* x0, int64_t ldx0,
                                      const void* prev, int64_t ldp, float alpha, const void* ln_weight,
                                      const void* ln_bias, float ln_eps, int relu, void* out, int64_t ldo,
                                      dif_stream_t stream) {
    using B = dif::bf16;
    auto c = [](const void* p) { return static_cast<const B*>(p); };
    Tail<B> tail = {};
    if (tail_enabled) tail = Tail<B>{c(x0), ldx0, c(prev), ldp, alpha, c(ln_weight), c(ln_bias), ln_eps, 1, relu};
    return spmm_entry<B>(rowptr, blkptr, n_blocks, src, val, n_nodes, nnz, c(x), ldx, row_begin, n_rows, F, c(attn), lda,
                         attn_scale, gcn_scale, tail, static_cast<B*>(out), ldo, row_order, n_split_rows, stream);
}

// ---- split product for row-sharded runs (dist.py): part 0 = the blocks whose sources are this rank's own rows (run
// ---- while the all-gather of the value rows is in flight), part 1 = everything else + the epilogue -----------------
extern "C" size_t dif_gcn_spmm_part_scratch_bytes(int64_t n_rows, int64_t n_split_rows, int F) {
    if (n_rows <= 0 || F <= 0 || n_split_rows < 0) return 0;
    int rw = 64;                                   // floats per accumulator row: G * W of the blocked kernel
    while (rw < F) rw *= 2;
    (void)n_split_rows;                            // one parked fp32 row per shard row, whatever the walk order
    return static_cast<size_t>(n_rows) * rw * sizeof(float);
}

template <typename E>
static int spmm_part_entry(const int32_t* rowptr, const int32_t* blkptr, int n_blocks, const int32_t* src, const float* val,
                           int64_t n_nodes, int64_t nnz, const E* x, int64_t ldx, int64_t row_begin, int64_t n_rows, int F,
                           const E* attn, int64_t lda, float attn_scale, float gcn_scale, const int32_t* row_order,
                           int64_t n_split_rows, const Tail<E>& tail, int part, int own_blk_begin, int own_blk_end,
                           int max_workgroups, float* scratch, size_t scratch_bytes, E* out, int64_t ldo,
                           dif_stream_t stream) {
    DIF_REQUIRE(part == 0 || part == 1, DIF_E_BADARG, "dif_gcn_spmm_part: part must be 0 or 1");
    DIF_REQUIRE(max_workgroups >= 0, DIF_E_BADARG, "dif_gcn_spmm_part: max_workgroups must be >= 0 (0 = one per CU)");
    DIF_REQUIRE(0 <= own_blk_begin && own_blk_begin <= own_blk_end && own_blk_end <= n_blocks, DIF_E_BADARG,
                "dif_gcn_spmm_part: own block range [%d, %d) outside [0, %d]", own_blk_begin, own_blk_end, n_blocks);
    DIF_REQUIRE(scratch && scratch_bytes >= dif_gcn_spmm_part_scratch_bytes(n_rows, n_split_rows, F), DIF_E_WORKSPACE,
                "dif_gcn_spmm_part: scratch too small");
    Sweep sw;
    if (part == 0) {
        sw = Sweep{own_blk_begin, -1, -1, own_blk_end, nullptr, scratch, max_workgroups};
    } else {
        const bool none = own_blk_begin == own_blk_end;
        sw = Sweep{(!none && own_blk_begin == 0) ? own_blk_end : 0, none ? -1 : own_blk_begin, none ? -1 : own_blk_end,
                   n_blocks, scratch, nullptr, max_workgroups};
    }
    Tail<E> t = tail;
    if (part == 0) t = Tail<E>{};
    return spmm_entry<E>(rowptr, blkptr, n_blocks, src, val, n_nodes, nnz, x, ldx, row_begin, n_rows, F,
                         part == 0 ? nullptr : attn, lda, attn_scale, gcn_scale, t, out, ldo, row_order, n_split_rows, stream,
                         &sw);
}

extern "C" int dif_gcn_spmm_part_f32(const int32_t* rowptr, const int32_t* blkptr, int n_blocks, const int32_t* src,
                                     const float* val, int64_t n_nodes, int64_t nnz, const float* x, int64_t ldx,
                                     int64_t row_begin, int64_t n_rows, int F, const float* attn, int64_t lda,
                                     float attn_scale, float gcn_scale, const int32_t* row_order, int64_t n_split_rows,
                                     int tail_enabled, const float* x0, int64_t ldx0, const float* prev, int64_t ldp,
                                     float alpha, const float* ln_weight, const float* ln_bias, float ln_eps, int relu,
                                     int part, int own_blk_begin, int own_blk_end, int max_workgroups, float* scratch,
                                     size_t scratch_bytes, float* out, int64_t ldo, dif_stream_t stream) {
    Tail<float> tail = {};
    if (tail_enabled) tail = Tail<float>{x0, ldx0, prev, ldp, alpha, ln_weight, ln_bias, ln_eps, 1, relu};
    return spmm_part_entry<float>(rowptr, blkptr, n_blocks, src, val, n_nodes, nnz, x, ldx, row_begin, n_rows, F, attn, lda,
                                  attn_scale, gcn_scale, row_order, n_split_rows, tail, part, own_blk_begin, own_blk_end,
                                  max_workgroups, scratch, scratch_bytes, out, ldo, stream);
}

extern "C" int dif_gcn_spmm_part_bf16(const int32_t* rowptr, const int32_t* blkptr, int n_blocks, const int32_t* src,
                                      const float* val, int64_t n_nodes, int64_t nnz, const void* x, int64_t ldx,
                                      int64_t row_begin, int64_t n_rows, int F, const void* attn, int64_t lda,
                                      float attn_scale, float gcn_scale, const int32_t* row_order, int64_t n_split_rows,
                                      int tail_enabled, const void* x0, int64_t ldx0, const void* prev, int64_t ldp,
                                      float alpha, const void* ln_weight, const void* ln_bias, float ln_eps, int relu,
                                      int part, int own_blk_begin, int own_blk_end, int max_workgroups, float* scratch,
                                      size_t scratch_bytes, void* out, int64_t ldo, dif_stream_t stream) {
    using B = dif::bf16;
    auto c = [](const void* p) { return static_cast<const B*>(p); };
    Tail<B> tail = {};
    if (tail_enabled) tail = Tail<B>{c(x0), ldx0, c(prev), ldp, alpha, c(ln_weight), c(ln_bias), ln_eps, 1, relu};
    return spmm_part_entry<B>(rowptr, blkptr, n_blocks, src, val, n_nodes, nnz, c(x), ldx, row_begin, n_rows, F, c(attn), lda,
                              attn_scale, gcn_scale, row_order, n_split_rows, tail, part, own_blk_begin, own_blk_end,
                              max_workgroups, scratch, scratch_bytes, static_cast<B*>(out), ldo, stream);
}

// a3 (cold half): COO edge_index -> normalised CSR over destination rows.
// Replaces, for node classification/difformer.py:63-75,
//   torch_geometric.utils.degree(col, N)           (:66)   -> csr_count_kernel (int atomics)
//   d_norm_in/out, value, nan_to_num               (:67-74) -> csr_fill_kernel
//   torch_sparse.SparseTensor(row=col, col=row,..) (:75)   -> stable LSD radix sort by destination
// The reference redoes all of this in every layer of every forward; here it runs once per
// (edge_index, edge_weight) and the host caches the result.
//
// Everything is integer / index work except the per-entry value, which is computed with the
// same float32 operations as the reference (correctly rounded 1/d, sqrt, two multiplies), so
// rowptr/src are exact and val is bit-identical to the CPU path.
// The sort is stable (entries of a row keep edge order) => the SpMM sums in a fixed order.
//
// Source blocking (n_blocks > 1): the sources are cut into n_blocks contiguous ranges of
// block_rows = ceil(N / n_blocks) nodes and the sort key becomes  dst * n_blocks + src / block_rows,
// so a destination row's entries are grouped by source block (edge order inside a group).
// blkptr[b][r] (block-major, (n_blocks+1) x N) is where group (r,b) starts; blkptr[n_blocks][r] is
// the end of row r.  The blocked SpMM sweeps the blocks in order on every CU at once, so the
// slice of x being gathered (block_rows * F * 4 bytes, sized by the host to ~2.5 MiB) stays in
// the 4 MiB L2 of each XCD instead of being fetched from Infinity Cache / HBM on every entry.
#include "dif_common.h"

namespace {

constexpr int kScanBlock = 256;
constexpr int kScanItems = 4;
constexpr int kScanTile = kScanBlock * kScanItems;  // 1024 values per block

constexpr int kSortWaves = 4;       // independent waves per block
constexpr int kSortRoundsMax = 64;  // 64 lanes x 64 rounds = 4096 keys per wave chunk on large graphs
constexpr int kSortRoundsMin = 8;   // small graphs (mini-batches): shorter chunks so that >= ~1000 waves share the sort
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

struct Plan {
    int64_t E, N;
    int64_t NB, block_rows, n_keys;  // source blocks, nodes per block, N * NB sort keys
    int rounds;            // 64-key rounds per wave chunk
    int64_t n_chunks;      // sort chunks (waves)
    int64_t table_len;     // kRadix * n_chunks
    int passes;            // radix passes over the destination id
    size_t off_keys_a, off_keys_b, off_vals_a, off_vals_b, off_deg, off_dinv, off_degc, off_table, off_bsum, total;
};

Plan make_plan(int64_t E, int64_t N, int64_t NB, int64_t block_rows = 0) {
    Plan p;
    p.E = E; p.N = N;
    p.NB = NB < 1 ? 1 : NB;
    p.block_rows = block_rows > 0 ? block_rows : (N + p.NB - 1) / p.NB;
    p.n_keys = N * p.NB;
    int64_t rounds = (E + 64 * 4096 - 1) / (64 * 4096);     // aim at ~4096 waves
    if (rounds < kSortRoundsMin) rounds = kSortRoundsMin;
    if (rounds > kSortRoundsMax) rounds = kSortRoundsMax;
    p.rounds = static_cast<int>(rounds);
    const int64_t chunk = 64 * rounds;
    p.n_chunks = (E + chunk - 1) / chunk;
    if (p.n_chunks < 1) p.n_chunks = 1;
    p.table_len = p.n_chunks * kRadix;
    int bits = 1;
    while ((int64_t(1) << bits) < p.n_keys) ++bits;
    p.passes = (bits + kRadixBits - 1) / kRadixBits;
    const size_t e = static_cast<size_t>(E > 0 ? E : 1);
    const int64_t scan_n = (p.table_len > p.n_keys + 1) ? p.table_len : (p.n_keys + 1);
    size_t o = 0;
    p.off_keys_a = o; o += align256(e * 4);
    p.off_keys_b = o; o += align256(e * 4);
    p.off_vals_a = o; o += align256(e * 4);
    p.off_vals_b = o; o += align256(e * 4);
    p.off_deg = o;    o += align256(static_cast<size_t>(p.n_keys + 1) * 4);   // per-key counts, then key pointers
    p.off_dinv = o;   o += align256(static_cast<size_t>(N + 1) * 4);
    p.off_degc = o;   o += align256(static_cast<size_t>(N + 1) * 4);   // in-degree over `col` (transposed build)
    p.off_table = o;  o += align256(static_cast<size_t>(p.table_len) * 4);
    p.off_bsum = o;   o += align256(static_cast<size_t>((scan_n + kScanTile - 1) / kScanTile + 1) * 4);
    p.total = o;
    return p;
}

// ---- sort keys (row it is filed under x source block) and payloads --------------------------------
// Payload: the edge id when the graph is weighted (the weight is fetched after the sort), otherwise directly the node
// id the entry will gather -- the sorted payload then IS the CSR `src` array and the fill pass needs no random access
// into edge_index.  No per-key counting here: the pointer table is read off the sorted keys (csr_bounds_kernel).
__global__ __launch_bounds__(256) void csr_count_kernel(const int64_t* __restrict__ edge_index, int64_t E,
                                                        int64_t N, int64_t NB, int64_t block_rows, int transpose,
                                                        int weighted, uint32_t* __restrict__ keys,
                                                        uint32_t* __restrict__ vals, int32_t* __restrict__ degc,
                                                        int32_t* __restrict__ status) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < E; e += stride) {
        int64_t r = edge_index[e];             // source       (row, difformer.py:65)
        int64_t c = edge_index[E + e];         // destination  (col)
        if (r < 0 || r >= N || c < 0 || c >= N) {
            atomicOr(status, 1);
            r = 0; c = 0;  // keep the build memory-safe; the host rejects the result
        }
        // forward: entries grouped by destination (col), blocked by source; transposed (backward of the aggregation):
        // grouped by source (row), blocked by destination
        // node ids are < 2^31 here (range check above) and the key fits 32 bits (checked by the host): 32-bit arithmetic --
        // a 64-bit integer division per edge made this pass 1.7 ms at C4 (1.1 TB/s for a streaming kernel)
        const uint32_t ru = static_cast<uint32_t>(r), cu = static_cast<uint32_t>(c), br = static_cast<uint32_t>(block_rows),
                       nb = static_cast<uint32_t>(NB);
        const uint32_t key = transpose ? ru * nb + cu / br : cu * nb + ru / br;
        keys[e] = key;
        vals[e] = weighted ? static_cast<uint32_t>(e) : static_cast<uint32_t>(transpose ? c : r);
        if (transpose) atomicAdd(&degc[c], 1); // the normalisation always uses the in-degree over `col` (:66)
    }
}

// kptr[j] = first position of the sorted keys holding a key >= j, for j in [0, n_keys]  (kptr[n_keys] = E): thread k
// owns the gap between key[k-1] and key[k].  Replaces E atomic increments + a scan over the n_keys counters.
__global__ __launch_bounds__(256) void csr_bounds_kernel(const uint32_t* __restrict__ key_sorted, int64_t E,
                                                         int64_t n_keys, int32_t* __restrict__ kptr) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k <= E; k += stride) {
        const int64_t lo = (k == 0) ? 0 : static_cast<int64_t>(key_sorted[k - 1]) + 1;
        const int64_t hi = (k == E) ? n_keys : static_cast<int64_t>(key_sorted[k]);
        for (int64_t j = lo; j <= hi; ++j) kptr[j] = static_cast<int32_t>(k);
    }
}

// ---- device-wide exclusive scan of int32 (3 phases) -----------------------------------------
__global__ __launch_bounds__(kScanBlock) void scan_block_sums_kernel(const int32_t* __restrict__ in, int64_t n,
                                                                     int32_t* __restrict__ bsum) {
    __shared__ int32_t sm[kScanBlock / 64];
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanTile;
    int32_t a = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const int64_t idx = base + i * kScanBlock + threadIdx.x;
        if (idx < n) a += in[idx];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// single block: exclusive scan of the block sums in place; bsum[nb] = grand total
__global__ __launch_bounds__(1024) void scan_of_sums_kernel(int32_t* __restrict__ bsum, int64_t nb) {
    __shared__ int32_t sm[1024];
    __shared__ int32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t idx = base + threadIdx.x;
        const int32_t x = (idx < nb) ? bsum[idx] : 0;
        sm[threadIdx.x] = x;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {          // Hillis-Steele inclusive scan
            const int32_t t = (static_cast<int>(threadIdx.x) >= off) ? sm[threadIdx.x - off] : 0;
            __syncthreads();
            sm[threadIdx.x] += t;
            __syncthreads();
        }
        const int32_t c = carry;
        if (idx < nb) bsum[idx] = c + sm[threadIdx.x] - x;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + sm[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[nb] = carry;
}

// out[i] = exclusive prefix of in; out may alias in.  If out_total != nullptr, *out_total = sum.
__global__ __launch_bounds__(kScanBlock) void scan_apply_kernel(const int32_t* in, int64_t n,
                                                                const int32_t* __restrict__ bsum, int64_t nb,
                                                                int32_t* out, int32_t* out_total) {
    __shared__ int32_t sm_w[kScanBlock / 64];
    const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanTile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // blocked arrangement: thread t owns items [t*4, t*4+4) of the tile
    int32_t x[kScanItems];
    int32_t tsum = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const int64_t idx = base + static_cast<int64_t>(threadIdx.x) * kScanItems + i;
        x[i] = (idx < n) ? in[idx] : 0;
        tsum += x[i];
    }
    int32_t incl = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) sm_w[wave] = incl;
    __syncthreads();
    int32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += sm_w[w];
    int32_t run = bsum[blockIdx.x] + woff + incl - tsum;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const int64_t idx = base + static_cast<int64_t>(threadIdx.x) * kScanItems + i;
        if (idx < n) out[idx] = run;
        run += x[i];
    }
    if (out_total && blockIdx.x == 0 && threadIdx.x == 0) *out_total = bsum[nb];
}

int exclusive_scan(const int32_t* in, int64_t n, int32_t* out, int32_t* out_total, int32_t* bsum,
                   hipStream_t st) {
    const int64_t nb = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(static_cast<unsigned>(nb)), dim3(kScanBlock), 0, st, in, n, bsum);
    hipLaunchKernelGGL(scan_of_sums_kernel, dim3(1), dim3(1024), 0, st, bsum, nb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(static_cast<unsigned>(nb)), dim3(kScanBlock), 0, st, in, n, bsum, nb,
                       out, out_total);
    return dif::launch_status("exclusive_scan");
}

// From the key pointers kptr[dst*NB + blk]: rowptr, block-major blkptr, and
// dinv[n] = sqrt(1/deg[n])  (float32, correctly rounded: difformer.py:67-68); deg 0 -> inf
__global__ __launch_bounds__(256) void csr_ptrs_kernel(const int32_t* __restrict__ kptr, int64_t N, int64_t NB,
                                                       int32_t* __restrict__ rowptr, int32_t* __restrict__ blkptr,
                                                       const int32_t* __restrict__ degc, float* __restrict__ dinv,
                                                       int32_t* __restrict__ longest) {
    const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    int32_t len = 0;
    if (r <= N) {
        const int32_t start = kptr[r * NB];
        rowptr[r] = start;                         // r == N: kptr[N*NB] = E
        if (r < N) {
            const int32_t end = kptr[(r + 1) * NB];
            len = end - start;
            dinv[r] = sqrtf(1.0f / static_cast<float>(degc ? degc[r] : end - start));
            if (blkptr) {
                for (int64_t b = 0; b < NB; ++b) blkptr[b * N + r] = kptr[r * NB + b];
                blkptr[NB * N + r] = end;
            }
        }
    }
    // the longest row of the CSR (status[1]): one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int32_t o = __shfl_xor(len, off, 64); len = o > len ? o : len; }
    if ((threadIdx.x & 63) == 0 && len > 0) atomicMax(longest, len);
}

// ---- stable LSD radix sort, 8 bits per pass; each WAVE owns a 4096-key chunk -------------------
__global__ __launch_bounds__(64 * kSortWaves) void radix_hist_kernel(const uint32_t* __restrict__ keys, int64_t n,
                                                                    int shift, int64_t n_chunks, int rounds,
                                                                    int32_t* __restrict__ table) {
    __shared__ int32_t hist[kSortWaves][kRadix];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t chunk = static_cast<int64_t>(blockIdx.x) * kSortWaves + wave;
#pragma unroll
    for (int i = 0; i < kRadix / 64; ++i) hist[wave][lane + 64 * i] = 0;
    __builtin_amdgcn_wave_barrier();
    if (chunk < n_chunks) {
        const int64_t base = chunk * 64 * rounds;
#pragma unroll 8
        for (int i = 0; i < rounds; ++i) {
            const int64_t idx = base + i * 64 + lane;
            if (idx < n) atomicAdd(&hist[wave][(keys[idx] >> shift) & (kRadix - 1)], 1);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < kRadix / 64; ++i) {
            const int d = lane + 64 * i;
            table[static_cast<int64_t>(d) * n_chunks + chunk] = hist[wave][d];  // digit-major
        }
    }
}

// Scatter of one pass.  Writing every key straight to base[digit] + rank makes 64 unrelated 4-byte stores per wave
// instruction (one 64-B sector each: ~10 GB of write traffic per pass at 79 M pairs, 2.3 ms).  Instead the wave first
// sorts its chunk LOCALLY: `perm` (LDS, 2 bytes per key) lists the chunk's positions digit by digit, in stable order;
// the copy-out then walks perm, so the keys of one digit leave as a run of consecutive addresses (16 keys = 64 B on
// average at 4096 keys per chunk) and re-reads its key / payload from the chunk it has just streamed (L1 / L2 hits).
__global__ __launch_bounds__(64 * kSortWaves) void radix_scatter_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, int64_t n, int shift,
    int64_t n_chunks, int rounds, const int32_t* __restrict__ table, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out) {
    __shared__ uint16_t perm_s[kSortWaves][64 * kSortRoundsMax];
    __shared__ int32_t gbase_s[kSortWaves][kRadix];    // where this chunk's keys of a digit start in the output
    __shared__ int32_t lstart_s[kSortWaves][kRadix];   // ... and in perm
    __shared__ int32_t lcur_s[kSortWaves][kRadix];     // running insert position per digit
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t chunk = static_cast<int64_t>(blockIdx.x) * kSortWaves + wave;
    if (chunk >= n_chunks) return;
    uint16_t* perm = perm_s[wave];
    int32_t* gbase = gbase_s[wave];
    int32_t* lstart = lstart_s[wave];
    volatile int32_t* lcur = lcur_s[wave];
    const int64_t cbase = chunk * 64 * rounds;
    const int cn = static_cast<int>((n - cbase < 64 * rounds) ? (n - cbase) : 64 * rounds);

    // per-digit output base and count of this chunk (the table is an exclusive scan in digit-major order), then the
    // exclusive prefix of the counts over the digits = layout of perm
    const int64_t tlen = static_cast<int64_t>(kRadix) * n_chunks;
    int carry = 0;
#pragma unroll
    for (int i = 0; i < kRadix / 64; ++i) {
        const int d = lane + 64 * i;
        const int64_t ti = static_cast<int64_t>(d) * n_chunks + chunk;
        const int32_t g = table[ti];
        const int32_t cnt = ((ti + 1 < tlen) ? table[ti + 1] : static_cast<int32_t>(n)) - g;
        int inc = cnt;                                   // inclusive scan over the 64 lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        gbase[d] = g;
        lstart[d] = carry + inc - cnt;
        lcur[d] = carry + inc - cnt;
        carry += __shfl(inc, 63, 64);
    }
    __builtin_amdgcn_wave_barrier();

    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int i = 0; i < rounds; ++i) {
        const int local = i * 64 + lane;
        const bool valid = local < cn;
        if (!__any(valid)) break;
        const uint32_t key = valid ? keys_in[cbase + local] : 0u;
        const uint32_t digit = (key >> shift) & (kRadix - 1);
        // lanes holding the same digit (wave64 "match any" from 8 ballots)
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < kRadixBits; ++b) {
            const bool bit = (digit >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        if (valid) perm[lcur[digit] + __popcll(m & lt_mask)] = static_cast<uint16_t>(local);
        __builtin_amdgcn_wave_barrier();
        if (valid && (m & lt_mask) == 0) lcur[digit] += __popcll(m);  // lowest lane of each group
        __builtin_amdgcn_wave_barrier();
    }
    // copy out in digit order: position p of perm goes to gbase[digit] + (p - lstart[digit])
    for (int p = lane; p < cn; p += 64) {
        const int local = perm[p];
        const uint32_t key = keys_in[cbase + local];
        const uint32_t val = vals_in[cbase + local];
        const uint32_t digit = (key >> shift) & (kRadix - 1);
        const int32_t dst = gbase[digit] + (p - lstart[digit]);
        keys_out[dst] = key;
        vals_out[dst] = val;
    }
}

// ---- rows of a shard by descending degree (load balance of the blocked SpMM) ----------------------
// stats[0] = rows with degree > 4 * mean degree of the shard (they come first in the order), stats[1] = max degree
__global__ __launch_bounds__(256) void order_keys_kernel(const int32_t* __restrict__ rowptr, int64_t row_begin,
                                                         int64_t n_rows, uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals, int32_t* __restrict__ stats) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const uint32_t d = static_cast<uint32_t>(rowptr[row_begin + i + 1] - rowptr[row_begin + i]);
    const int64_t total = static_cast<int64_t>(rowptr[row_begin + n_rows]) - rowptr[row_begin];
    if (static_cast<int64_t>(d) * n_rows > 4 * total) atomicAdd(stats, 1);
    atomicMax(stats + 1, static_cast<int32_t>(d));
    keys[i] = 0xFFFFFFu - (d < 0xFFFFFFu ? d : 0xFFFFFFu);      // ascending key = descending degree (24 bits)
    vals[i] = static_cast<uint32_t>(i);                         // row inside the shard
}

// ---- per-entry source id and normalised value ---------------------------------------------------
__global__ __launch_bounds__(256) void csr_fill_kernel(const int64_t* __restrict__ edge_index, int64_t E, int64_t N,
                                                       uint32_t NB, int transpose,
                                                       const float* __restrict__ edge_weight,
                                                       const uint32_t* __restrict__ key_sorted,
                                                       const uint32_t* __restrict__ eid_sorted,
                                                       const float* __restrict__ dinv, int32_t* __restrict__ src,
                                                       float* __restrict__ val) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < E; k += stride) {
        const uint32_t e = eid_sorted[k];                 // weighted: edge id; unweighted: already the node to gather
        const uint32_t grp = key_sorted[k] / NB;          // the row this entry is filed under
        int64_t other = e;
        if (edge_weight) {
            other = edge_index[transpose ? E + e : e];
            if (other < 0 || other >= N) other = 0;       // flagged in status by csr_count_kernel
        }
        const int64_t r = transpose ? grp : other;        // source      (difformer.py:65 `row`)
        const int64_t c = transpose ? other : grp;        // destination (`col`)
        const float dn_in = dinv[c];
        const float dn_out = dinv[r];
        // :71 / :73 -- (w * d_norm_in) * d_norm_out, float32, no contraction
        float v = edge_weight ? __fmul_rn(__fmul_rn(edge_weight[e], dn_in), dn_out) : __fmul_rn(dn_in, dn_out);
        if (!isfinite(v)) v = 0.f;             // :74 nan_to_num(nan=0, posinf=0, neginf=0)
        src[k] = static_cast<int32_t>(other);             // the row to gather when this entry is applied
        val[k] = v;
    }
}

// ---- induced subgraph of a node subset with relabelling (f2: the per-batch graph step of the mini-batch path) ----
// node classification/main-batch.py:131  subgraph(idx_i, edge_index, num_nodes=n, relabel_nodes=True)  (torch_geometric
// 1.7.2, un-vendored: node_mask[subset] = True; keep edges whose two ends are in the subset, in their original order;
// new id of subset[i] is i).  The reference runs it on the CPU over the whole edge list for every batch.
// `member` is a bitmap of the subset (N bits: 200 KB for Pokec, L2-resident), the membership test of the flag pass;
// `newid` (N x int32) is only consulted for the few edges that survive it.
__global__ __launch_bounds__(256) void subgraph_mark_kernel(const int64_t* __restrict__ subset, int64_t B, int64_t N,
                                                            int32_t* __restrict__ newid, uint32_t* __restrict__ member,
                                                            int32_t* __restrict__ status) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const int64_t v = subset[i];
    if (v < 0 || v >= N) { atomicOr(status, 1); return; }
    newid[v] = static_cast<int32_t>(i) + 1;          // 0 = not in the subset
    atomicOr(&member[v >> 5], 1u << (v & 31));
}

// Two passes over the edge list, nothing per edge in between but ONE BIT (round 6; before: an int32 flag per edge, a scan over all
// E + 1 of them and a pass that read the scanned positions back -- 1.10 ms per ogbn-proteins batch for 79 M entries of which
// 0.6 % survive):
//   mask     a wave tests 64 consecutive edges, its ballot IS their mask word; a workgroup owns a chunk of 4,096 edges
//            (64 words) and leaves the chunk's kept count
//   scan     exclusive scan of the E / 4,096 chunk counts
//   compact  a wave per chunk: lane l takes word l, a wave prefix sum of the popcounts places the words, the lane walks its
//            set bits in order -- only the kept edges' ends are read again
constexpr int kSubChunk = 4096;                    // edges per chunk = 64 mask words
__global__ __launch_bounds__(256) void subgraph_mask_kernel(const int64_t* __restrict__ edge_index, int64_t E, int64_t N,
                                                            const uint32_t* __restrict__ member, unsigned long long* __restrict__ mask,
                                                            int32_t* __restrict__ counts, int64_t n_chunks,
                                                            int32_t* __restrict__ status) {
    __shared__ int sCount[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t ch = blockIdx.x; ch <= n_chunks; ch += gridDim.x) {
        if (ch == n_chunks) {                          // sentinel so the scan's last entry is the kept count
            if (threadIdx.x == 0) counts[ch] = 0;
            continue;
        }
        int cnt = 0;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int64_t word = ch * 64 + it * 4 + w;
            const int64_t e = word * 64 + lane;
            int k = 0;
            if (e < E) {
                const int64_t r = edge_index[e], c = edge_index[E + e];
                if (r < 0 || r >= N || c < 0 || c >= N) atomicOr(status, 1);
                else if ((member[r >> 5] >> (r & 31)) & 1u) k = (member[c >> 5] >> (c & 31)) & 1u;   // second test for ~B/N of the edges
            }
            const unsigned long long bits = __ballot(k);
            if (lane == 0 && word * 64 < E) mask[word] = bits;
            cnt += __popcll(bits);
        }
        if (lane == 0) sCount[w] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) counts[ch] = sCount[0] + sCount[1] + sCount[2] + sCount[3];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void subgraph_compact_kernel(const int64_t* __restrict__ edge_index, int64_t E,
                                                               const float* __restrict__ edge_weight,
                                                               const int32_t* __restrict__ newid,
                                                               const unsigned long long* __restrict__ mask,
                                                               const int32_t* __restrict__ base, int64_t n_chunks, int64_t cap,
                                                               int64_t* __restrict__ out_ei, float* __restrict__ out_w,
                                                               int64_t* __restrict__ out_count) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6), stride = static_cast<int64_t>(gridDim.x) * 4;
    for (int64_t ch = wave0; ch < n_chunks; ch += stride) {
        if (base[ch + 1] == base[ch]) continue;        // nothing kept in these 4,096 edges (wave-uniform)
        const int64_t word = ch * 64 + lane;
        unsigned long long bits = (word * 64 < E) ? mask[word] : 0ull;
        const int mine = __popcll(bits);
        int incl = mine;                               // inclusive prefix sum over the lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        int64_t p = static_cast<int64_t>(base[ch]) + incl - mine;
        while (bits) {
            const int b = __builtin_ctzll(bits);
            bits &= bits - 1;
            const int64_t e = word * 64 + b;
            out_ei[p] = newid[edge_index[e]] - 1;
            out_ei[cap + p] = newid[edge_index[E + e]] - 1;
            if (out_w) out_w[p] = edge_weight[e];
            ++p;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = base[n_chunks];
}


// ---- all mini-batches of an epoch in one pass (f2; node classification/main-batch.py:121-131) ------------------------
// The reference draws a permutation of the training nodes once per epoch and then, for every batch of `batch_size`
// consecutive entries, filters the WHOLE edge list (subgraph(..., relabel_nodes=True), on the CPU).  Here the edge list
// is streamed once for all batches: info[v] = 1 + position of v in the permutation; an edge survives iff both ends sit
// in the same batch (position / batch_size), its key is that batch (255 = dropped), and one stable radix pass on the key
// groups the surviving edge ids batch by batch in their original order.
__global__ __launch_bounds__(256) void batches_mark_kernel(const int64_t* __restrict__ perm, int64_t M, int64_t N,
                                                           int32_t* __restrict__ info, int32_t* __restrict__ status) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int64_t v = perm[i];
    if (v < 0 || v >= N) { atomicOr(status, 1); return; }
    if (atomicExch(&info[v], static_cast<int32_t>(i) + 1) != 0) atomicOr(status, 2);      // repeated id
}

__global__ __launch_bounds__(256) void batches_key_kernel(const int64_t* __restrict__ edge_index, int64_t E, int64_t N,
                                                          const int32_t* __restrict__ info, int32_t batch_size, int shift_hi,
                                                          uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                          int32_t* __restrict__ status) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < E; e += stride) {
        const int64_t r = edge_index[e], c = edge_index[E + e];
        uint32_t key = 0xffffu;                                   // dropped: sorts behind every batch
        if (r < 0 || r >= N || c < 0 || c >= N) atomicOr(status, 1);
        else {
            const int32_t pr = info[r];
            if (pr) {
                const int32_t pc = info[c];
                if (pc) {
                    const int32_t br = (pr - 1) / batch_size;
                    if (br == (pc - 1) / batch_size) key = static_cast<uint32_t>(br);
                }
            }
        }
        keys[e] = shift_hi ? key : (key & 0xffu);                 // one 8-bit pass when there are < 255 batches
        vals[e] = static_cast<uint32_t>(e);
    }
}

// batch_ptr[b] = first sorted position holding a key >= b (b = 0 .. n_batches); the kept count is batch_ptr[n_batches]
__global__ __launch_bounds__(256) void batches_ptr_kernel(const uint32_t* __restrict__ keys_sorted, int64_t E, int n_batches,
                                                          int64_t* __restrict__ batch_ptr) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k <= E; k += stride) {
        const int64_t lo = (k == 0) ? 0 : static_cast<int64_t>(keys_sorted[k - 1]) + 1;
        int64_t hi = (k == E) ? n_batches : static_cast<int64_t>(keys_sorted[k]);
        if (hi > n_batches) hi = n_batches;
        for (int64_t j = lo; j <= hi; ++j) batch_ptr[j] = k;
    }
}

__global__ __launch_bounds__(256) void batches_emit_kernel(const int64_t* __restrict__ edge_index, int64_t E,
                                                           const float* __restrict__ edge_weight,
                                                           const int32_t* __restrict__ info, int32_t batch_size,
                                                           const uint32_t* __restrict__ vals_sorted,
                                                           const int64_t* __restrict__ batch_ptr, int n_batches, int64_t cap,
                                                           int64_t* __restrict__ out_ei, float* __restrict__ out_w) {
    const int64_t kept = batch_ptr[n_batches];
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < kept && i < cap; i += stride) {
        const int64_t e = vals_sorted[i];
        out_ei[i] = (info[edge_index[e]] - 1) % batch_size;               // subset[j] becomes node j of its batch
        out_ei[cap + i] = (info[edge_index[E + e]] - 1) % batch_size;
        if (out_w) out_w[i] = edge_weight[e];
    }
}

// Batched CSR of the surviving edges: one sort on key = batch * batch_size + local destination (payload = edge id),
// so every batch's CSR (its own degrees, its own normalisation: difformer.py:63-75 on the subgraph) is a slice of one
// set of arrays -- instead of one dif_csr_build per batch.
__global__ __launch_bounds__(256) void batches_csr_key_kernel(const int64_t* __restrict__ edge_index, int64_t E,
                                                              const int32_t* __restrict__ info,
                                                              const uint32_t* __restrict__ eid_grouped, int64_t kept,
                                                              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < kept; i += stride) {
        const uint32_t e = eid_grouped[i];
        keys[i] = static_cast<uint32_t>(info[edge_index[E + e]] - 1);     // position of the destination in the permutation
        vals[i] = e;                                                      //   = batch * batch_size + local id
    }
}

__global__ __launch_bounds__(256) void batches_csr_dinv_kernel(const int32_t* __restrict__ rowptr, int64_t M,
                                                               float* __restrict__ dinv) {
    const int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g < M) dinv[g] = sqrtf(1.0f / static_cast<float>(rowptr[g + 1] - rowptr[g]));
}

__global__ __launch_bounds__(256) void batches_csr_fill_kernel(const int64_t* __restrict__ edge_index, int64_t E,
                                                               const float* __restrict__ edge_weight,
                                                               const int32_t* __restrict__ info, int32_t batch_size,
                                                               const uint32_t* __restrict__ key_sorted,
                                                               const uint32_t* __restrict__ eid_sorted, int64_t kept,
                                                               const float* __restrict__ dinv, int32_t* __restrict__ src,
                                                               float* __restrict__ val) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < kept; k += stride) {
        const uint32_t e = eid_sorted[k];
        const int64_t gc = key_sorted[k];                                 // destination (`col`), global position
        const int64_t gr = info[edge_index[e]] - 1;                       // source (`row`): same batch by construction
        const float dn_in = dinv[gc], dn_out = dinv[gr];
        float v = edge_weight ? __fmul_rn(__fmul_rn(edge_weight[e], dn_in), dn_out) : __fmul_rn(dn_in, dn_out);
        if (!isfinite(v)) v = 0.f;
        src[k] = static_cast<int32_t>(gr % batch_size);
        val[k] = v;
    }
}

// ---- graph preparation of the drivers (node classification/main.py:72-76, main-batch.py:96-98), on device -----------
//   to_undirected (both directions, duplicates coalesced, sorted by (row, col)) -> remove_self_loops -> add_self_loops
// Pairs are generated (self loops dropped right there when asked), sorted by col then by row with the stable radix
// passes above (= lexicographic (row, col)), runs of equal pairs keep their first element, and the N loops are appended.
__global__ __launch_bounds__(256) void prep_pairs_kernel(const int64_t* __restrict__ edge_index, int64_t E, int64_t N,
                                                         int undirected, uint32_t* __restrict__ key_col,
                                                         uint32_t* __restrict__ val_row, int32_t* __restrict__ status) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < E; e += stride) {
        int64_t r = edge_index[e], c = edge_index[E + e];
        if (r < 0 || r >= N || c < 0 || c >= N) { atomicOr(status, 1); r = c = 0; }
        key_col[e] = static_cast<uint32_t>(c);
        val_row[e] = static_cast<uint32_t>(r);
        if (undirected) {
            key_col[E + e] = static_cast<uint32_t>(r);
            val_row[E + e] = static_cast<uint32_t>(c);
        }
    }
}

// swap roles between the two sorts: the pairs sorted by col become (key = row, payload = col)
__global__ __launch_bounds__(256) void prep_swap_kernel(uint32_t* __restrict__ a, uint32_t* __restrict__ b, int64_t n) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t t = a[i];
        a[i] = b[i];
        b[i] = t;
    }
}

// keep[i] = 1 for the first pair of a run of equal (row, col) (coalesce), 0 for self loops when they are removed
__global__ __launch_bounds__(256) void prep_flag_kernel(const uint32_t* __restrict__ row, const uint32_t* __restrict__ col,
                                                        int64_t n, int coalesce, int drop_loops, int32_t* __restrict__ keep) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i <= n; i += stride) {
        int k = 0;
        if (i < n) {
            k = 1;
            if (coalesce && i > 0 && row[i] == row[i - 1] && col[i] == col[i - 1]) k = 0;
            if (drop_loops && row[i] == col[i]) k = 0;
        }
        keep[i] = k;                                    // keep[n] = 0: the scan's last entry is the kept count
    }
}

__global__ __launch_bounds__(256) void prep_emit_kernel(const uint32_t* __restrict__ row, const uint32_t* __restrict__ col,
                                                        int64_t n, const int32_t* __restrict__ pos, int64_t N, int add_loops,
                                                        int64_t cap, int64_t* __restrict__ out_ei, int64_t* __restrict__ out_count) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t kept = pos[n];
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int32_t p = pos[i];
        if (pos[i + 1] != p) {
            out_ei[p] = row[i];
            out_ei[cap + p] = col[i];
        }
    }
    if (add_loops) {
        for (int64_t v = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; v < N; v += stride) {
            out_ei[kept + v] = v;
            out_ei[cap + kept + v] = v;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = kept + (add_loops ? N : 0);
}

struct PrepPlan { int rounds; int64_t n, n_chunks, table_len; int passes; size_t off_ka, off_kb, off_va, off_vb, off_keep, off_table, off_bsum, total; };
PrepPlan make_prep_plan(int64_t E, int64_t N, int undirected) {
    PrepPlan p;
    p.n = undirected ? 2 * E : E;
    int64_t rounds = (p.n + 64 * 4096 - 1) / (64 * 4096);
    if (rounds < kSortRoundsMin) rounds = kSortRoundsMin;
    if (rounds > kSortRoundsMax) rounds = kSortRoundsMax;
    p.rounds = static_cast<int>(rounds);
    const int64_t chunk = 64 * rounds;
    p.n_chunks = (p.n + chunk - 1) / chunk;
    if (p.n_chunks < 1) p.n_chunks = 1;
    p.table_len = p.n_chunks * kRadix;
    int bits = 1;
    while ((int64_t(1) << bits) < N) ++bits;
    p.passes = (bits + kRadixBits - 1) / kRadixBits;
    const size_t e = static_cast<size_t>(p.n > 0 ? p.n : 1);
    const int64_t scan_n = p.table_len > p.n + 1 ? p.table_len : p.n + 1;
    size_t o = 0;
    p.off_ka = o;    o += align256(e * 4);
    p.off_kb = o;    o += align256(e * 4);
    p.off_va = o;    o += align256(e * 4);
    p.off_vb = o;    o += align256(e * 4);
    p.off_keep = o;  o += align256((e + 1) * 4);
    p.off_table = o; o += align256(static_cast<size_t>(p.table_len) * 4);
    p.off_bsum = o;  o += align256(static_cast<size_t>((scan_n + kScanTile - 1) / kScanTile + 1) * 4);
    p.total = o;
    return p;
}

struct BatchCsrPlan { int rounds; int64_t n_chunks, table_len; int passes; size_t off_ka, off_kb, off_va, off_vb, off_dinv, off_table, off_bsum, total; };
BatchCsrPlan make_batch_csr_plan(int64_t kept, int64_t M) {
    BatchCsrPlan p;
    int64_t rounds = (kept + 64 * 4096 - 1) / (64 * 4096);
    if (rounds < kSortRoundsMin) rounds = kSortRoundsMin;
    if (rounds > kSortRoundsMax) rounds = kSortRoundsMax;
    p.rounds = static_cast<int>(rounds);
    const int64_t chunk = 64 * rounds;
    p.n_chunks = (kept + chunk - 1) / chunk;
    if (p.n_chunks < 1) p.n_chunks = 1;
    p.table_len = p.n_chunks * kRadix;
    int bits = 1;
    while ((int64_t(1) << bits) < M) ++bits;
    p.passes = (bits + kRadixBits - 1) / kRadixBits;
    const size_t e = static_cast<size_t>(kept > 0 ? kept : 1);
    const int64_t scan_n = p.table_len > M + 1 ? p.table_len : M + 1;
    size_t o = 0;
    p.off_ka = o;    o += align256(e * 4);
    p.off_kb = o;    o += align256(e * 4);
    p.off_va = o;    o += align256(e * 4);
    p.off_vb = o;    o += align256(e * 4);
    p.off_dinv = o;  o += align256(static_cast<size_t>(M + 1) * 4);
    p.off_table = o; o += align256(static_cast<size_t>(p.table_len) * 4);
    p.off_bsum = o;  o += align256(static_cast<size_t>((scan_n + kScanTile - 1) / kScanTile + 1) * 4);
    p.total = o;
    return p;
}

struct BatchPlan { int rounds; int64_t n_chunks, table_len; int passes; size_t off_info, off_ka, off_kb, off_va, off_vb, off_table, off_bsum, total; };
BatchPlan make_batch_plan(int64_t E, int64_t N, int n_batches) {
    BatchPlan p;
    int64_t rounds = (E + 64 * 4096 - 1) / (64 * 4096);
    if (rounds < kSortRoundsMin) rounds = kSortRoundsMin;
    if (rounds > kSortRoundsMax) rounds = kSortRoundsMax;
    p.rounds = static_cast<int>(rounds);
    const int64_t chunk = 64 * rounds;
    p.n_chunks = (E + chunk - 1) / chunk;
    if (p.n_chunks < 1) p.n_chunks = 1;
    p.table_len = p.n_chunks * kRadix;
    p.passes = n_batches < 255 ? 1 : 2;
    const size_t e = static_cast<size_t>(E > 0 ? E : 1);
    size_t o = 0;
    p.off_info = o;  o += align256(static_cast<size_t>(N) * 4);
    p.off_ka = o;    o += align256(e * 4);
    p.off_kb = o;    o += align256(e * 4);
    p.off_va = o;    o += align256(e * 4);
    p.off_vb = o;    o += align256(e * 4);
    p.off_table = o; o += align256(static_cast<size_t>(p.table_len) * 4);
    p.off_bsum = o;  o += align256(static_cast<size_t>((p.table_len + kScanTile - 1) / kScanTile + 1) * 4);
    p.total = o;
    return p;
}

}  // namespace

extern "C" size_t dif_csr_workspace_bytes(int64_t E, int64_t N, int n_blocks) {
    if (E < 0 || N <= 0 || n_blocks < 1) return 0;
    return make_plan(E, N, n_blocks).total;
}

extern "C" int dif_csr_build(const int64_t* edge_index, int64_t E, int64_t N, const float* edge_weight,
                             int n_blocks, int64_t block_rows, int transpose, int32_t* rowptr, int32_t* blkptr, int32_t* src, float* val,
                             int32_t* status, void* workspace, size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(N > 0 && E >= 0 && n_blocks >= 1, DIF_E_BADARG, "dif_csr_build: need N > 0, E >= 0, n_blocks >= 1");
    DIF_REQUIRE(E < (int64_t(1) << 31) - 4096 && N < (int64_t(1) << 31) - 1, DIF_E_RANGE,
                "dif_csr_build: E and N must fit int32 (E=%lld, N=%lld)", static_cast<long long>(E),
                static_cast<long long>(N));
    DIF_REQUIRE(N * static_cast<int64_t>(n_blocks) < (int64_t(1) << 31), DIF_E_RANGE,
                "dif_csr_build: N * n_blocks must fit int32");
    DIF_REQUIRE(rowptr && status && workspace && (E == 0 || (edge_index && src && val)), DIF_E_BADARG,
                "dif_csr_build: null pointer");
    DIF_REQUIRE(n_blocks == 1 || blkptr, DIF_E_BADARG, "dif_csr_build: n_blocks > 1 needs blkptr");
    DIF_REQUIRE(block_rows >= 0 && (block_rows == 0 || block_rows * n_blocks >= N), DIF_E_BADARG,
                "dif_csr_build: block_rows * n_blocks must cover N (block_rows = 0: ceil(N / n_blocks))");
    const Plan p = make_plan(E, N, n_blocks, block_rows);
    DIF_REQUIRE(workspace_bytes >= p.total, DIF_E_WORKSPACE, "dif_csr_build: workspace too small (%zu < %zu)",
                workspace_bytes, p.total);
    DIF_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, DIF_E_BADARG,
                "dif_csr_build: workspace must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    uint32_t* keys_a = reinterpret_cast<uint32_t*>(ws + p.off_keys_a);
    uint32_t* keys_b = reinterpret_cast<uint32_t*>(ws + p.off_keys_b);
    uint32_t* vals_a = reinterpret_cast<uint32_t*>(ws + p.off_vals_a);
    uint32_t* vals_b = reinterpret_cast<uint32_t*>(ws + p.off_vals_b);
    int32_t* kcnt = reinterpret_cast<int32_t*>(ws + p.off_deg);   // per-key counts -> key pointers (in place)
    float* dinv = reinterpret_cast<float*>(ws + p.off_dinv);
    int32_t* degc = transpose ? reinterpret_cast<int32_t*>(ws + p.off_degc) : nullptr;
    int32_t* table = reinterpret_cast<int32_t*>(ws + p.off_table);
    int32_t* bsum = reinterpret_cast<int32_t*>(ws + p.off_bsum);

    hipError_t he = hipMemsetAsync(status, 0, 8, st);            // {bad index flag, longest row}
    if (he == hipSuccess && transpose) he = hipMemsetAsync(degc, 0, static_cast<size_t>(N + 1) * 4, st);
    if (he != hipSuccess) return dif::fail(static_cast<int>(he), "dif_csr_build: memset: %s", hipGetErrorString(he));

    const int64_t cap = 8 * dif::kCUs;
    if (E == 0) {           // no entries: every pointer is 0
        he = hipMemsetAsync(kcnt, 0, static_cast<size_t>(p.n_keys + 1) * 4, st);
        if (he != hipSuccess) return dif::fail(static_cast<int>(he), "dif_csr_build: memset: %s", hipGetErrorString(he));
        hipLaunchKernelGGL(csr_ptrs_kernel, dim3(static_cast<unsigned>((N + 256) / 256)), dim3(256), 0, st, kcnt, N, p.NB,
                           rowptr, n_blocks > 1 ? blkptr : nullptr, degc, dinv, status + 1);
        return dif::launch_status("csr_ptrs_kernel");
    }
    {
        int64_t g = (E + 255) / 256;
        if (g > cap) g = cap;
        hipLaunchKernelGGL(csr_count_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0, st, edge_index, E, N, p.NB,
                           p.block_rows, transpose, edge_weight != nullptr, keys_a, vals_a, degc, status);
        if (int rc = dif::launch_status("csr_count_kernel")) return rc;
    }

    uint32_t *kin = keys_a, *kout = keys_b, *vin = vals_a, *vout = vals_b;
    const unsigned sort_grid = static_cast<unsigned>((p.n_chunks + kSortWaves - 1) / kSortWaves);
    for (int pass = 0; pass < p.passes; ++pass) {
        const int shift = pass * kRadixBits;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(sort_grid), dim3(64 * kSortWaves), 0, st, kin, E, shift,
                           p.n_chunks, p.rounds, table);
        if (int rc = dif::launch_status("radix_hist_kernel")) return rc;
        if (int rc = exclusive_scan(table, p.table_len, table, nullptr, bsum, st)) return rc;
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(sort_grid), dim3(64 * kSortWaves), 0, st, kin, vin, E, shift,
                           p.n_chunks, p.rounds, table, kout, vout);
        if (int rc = dif::launch_status("radix_scatter_kernel")) return rc;
        uint32_t* t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    int64_t g = (E + 1 + 255) / 256;
    if (g > cap) g = cap;
    // key pointers kptr[0 .. N*NB] from the sorted keys, then rowptr / blkptr / dinv
    hipLaunchKernelGGL(csr_bounds_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0, st, kin, E, p.n_keys, kcnt);
    if (int rc = dif::launch_status("csr_bounds_kernel")) return rc;
    hipLaunchKernelGGL(csr_ptrs_kernel, dim3(static_cast<unsigned>((N + 256) / 256)), dim3(256), 0, st, kcnt, N, p.NB,
                       rowptr, n_blocks > 1 ? blkptr : nullptr, degc, dinv, status + 1);
    if (int rc = dif::launch_status("csr_ptrs_kernel")) return rc;
    hipLaunchKernelGGL(csr_fill_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0, st, edge_index, E, N,
                       static_cast<uint32_t>(p.NB), transpose, edge_weight, kin, vin, dinv, src, val);
    return dif::launch_status("csr_fill_kernel");
}

namespace {
struct OrderPlan { int64_t n_chunks, table_len; size_t off_keys_a, off_keys_b, off_vals_a, off_table, off_bsum, total; };
OrderPlan make_order_plan(int64_t n) {
    OrderPlan p;
    const int64_t chunk = 64 * kSortRoundsMin;
    p.n_chunks = (n + chunk - 1) / chunk;
    if (p.n_chunks < 1) p.n_chunks = 1;
    p.table_len = p.n_chunks * kRadix;
    const size_t e = static_cast<size_t>(n > 0 ? n : 1);
    size_t o = 0;
    p.off_keys_a = o; o += align256(e * 4);
    p.off_keys_b = o; o += align256(e * 4);
    p.off_vals_a = o; o += align256(e * 4);
    p.off_table = o;  o += align256(static_cast<size_t>(p.table_len) * 4);
    p.off_bsum = o;   o += align256(static_cast<size_t>((p.table_len + kScanTile - 1) / kScanTile + 1) * 4);
    p.total = o;
    return p;
}
}  // namespace

extern "C" size_t dif_row_order_workspace_bytes(int64_t n_rows) {
    if (n_rows <= 0) return 0;
    return make_order_plan(n_rows).total;
}

// order[i] = i-th row of the shard [row_begin, row_begin + n_rows) by descending degree (ties: ascending row), as an
// index INSIDE the shard; stats[0] = number of rows whose degree exceeds 4x the shard's mean (a prefix of the order: the
// rows the blocked SpMM splits over a whole quad), stats[1] = largest degree.  Stable 3-pass LSD radix sort on 24 bits of (2^24 - 1 - degree) with the CSR build's kernels.
extern "C" int dif_row_order(const int32_t* rowptr, int64_t row_begin, int64_t n_rows, int32_t* order, int32_t* stats,
                             void* workspace, size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(n_rows > 0 && row_begin >= 0 && n_rows < (int64_t(1) << 31) - 4096, DIF_E_BADARG,
                "dif_row_order: need 0 < n_rows < 2^31, row_begin >= 0");
    DIF_REQUIRE(rowptr && order && stats && workspace, DIF_E_BADARG, "dif_row_order: null pointer");
    const OrderPlan p = make_order_plan(n_rows);
    DIF_REQUIRE(workspace_bytes >= p.total, DIF_E_WORKSPACE, "dif_row_order: workspace too small (%zu < %zu)",
                workspace_bytes, p.total);
    DIF_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, DIF_E_BADARG,
                "dif_row_order: workspace must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    uint32_t* kin = reinterpret_cast<uint32_t*>(ws + p.off_keys_a);
    uint32_t* kout = reinterpret_cast<uint32_t*>(ws + p.off_keys_b);
    uint32_t* vin = reinterpret_cast<uint32_t*>(ws + p.off_vals_a);
    uint32_t* vout = reinterpret_cast<uint32_t*>(order);            // 3 passes: a -> order -> a -> order
    int32_t* table = reinterpret_cast<int32_t*>(ws + p.off_table);
    int32_t* bsum = reinterpret_cast<int32_t*>(ws + p.off_bsum);
    hipError_t he = hipMemsetAsync(stats, 0, 8, st);
    if (he != hipSuccess) return dif::fail(static_cast<int>(he), "dif_row_order: memset: %s", hipGetErrorString(he));
    hipLaunchKernelGGL(order_keys_kernel, dim3(static_cast<unsigned>((n_rows + 255) / 256)), dim3(256), 0, st, rowptr,
                       row_begin, n_rows, kin, vin, stats);
    if (int rc = dif::launch_status("order_keys_kernel")) return rc;
    const unsigned sort_grid = static_cast<unsigned>((p.n_chunks + kSortWaves - 1) / kSortWaves);
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass * kRadixBits;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(sort_grid), dim3(64 * kSortWaves), 0, st, kin, n_rows, shift, p.n_chunks,
                           kSortRoundsMin, table);
        if (int rc = dif::launch_status("radix_hist_kernel")) return rc;
        if (int rc = exclusive_scan(table, p.table_len, table, nullptr, bsum, st)) return rc;
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(sort_grid), dim3(64 * kSortWaves), 0, st, kin, vin, n_rows, shift,
                           p.n_chunks, kSortRoundsMin, table, kout, vout);
        if (int rc = dif::launch_status("radix_scatter_kernel")) return rc;
        uint32_t* t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    return 0;
}

// workspace: newid [N] | mask words [ceil(E / 64)] | chunk counts [E / 4,096 + 2] | the scan's block sums | member bitmap [N / 32]
inline size_t sub_mask_bytes(int64_t E) { return align256(static_cast<size_t>((E + 63) / 64 + 1) * 8); }
inline size_t sub_count_bytes(int64_t E) { return align256(static_cast<size_t>((E + kSubChunk - 1) / kSubChunk + 2) * 4); }
inline size_t sub_bsum_bytes(int64_t E) {
    const int64_t n = (E + kSubChunk - 1) / kSubChunk + 1;
    return align256(static_cast<size_t>((n + kScanTile - 1) / kScanTile + 1) * 4);
}

extern "C" size_t dif_subgraph_workspace_bytes(int64_t E, int64_t N) {
    if (E < 0 || N <= 0) return 0;
    return align256(static_cast<size_t>(N) * 4) + sub_mask_bytes(E) + sub_count_bytes(E) + sub_bsum_bytes(E) +
           align256(static_cast<size_t>((N + 31) / 32) * 4);
}

extern "C" int dif_subgraph(const int64_t* edge_index, int64_t E, int64_t N, const int64_t* subset, int64_t B,
                            const float* edge_weight, int64_t* out_edge_index, float* out_weight, int64_t* out_count,
                            int32_t* status, void* workspace, size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(N > 0 && E >= 0 && B >= 0, DIF_E_BADARG, "dif_subgraph: need N > 0, E >= 0, B >= 0");
    DIF_REQUIRE(E < (int64_t(1) << 31) - 4096 && N < (int64_t(1) << 31) - 1 && B < (int64_t(1) << 31) - 1, DIF_E_RANGE,
                "dif_subgraph: sizes must fit int32");
    DIF_REQUIRE(out_count && status && workspace && (E == 0 || (edge_index && out_edge_index)) && (B == 0 || subset),
                DIF_E_BADARG, "dif_subgraph: null pointer");
    DIF_REQUIRE((edge_weight == nullptr) == (out_weight == nullptr), DIF_E_BADARG,
                "dif_subgraph: edge_weight and out_weight must be given together");
    DIF_REQUIRE(workspace_bytes >= dif_subgraph_workspace_bytes(E, N), DIF_E_WORKSPACE, "dif_subgraph: workspace too small");
    DIF_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, DIF_E_BADARG,
                "dif_subgraph: workspace must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    int32_t* newid = reinterpret_cast<int32_t*>(ws);
    char* at = ws + align256(static_cast<size_t>(N) * 4);
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(at);
    at += sub_mask_bytes(E);
    int32_t* counts = reinterpret_cast<int32_t*>(at);
    at += sub_count_bytes(E);
    int32_t* bsum = reinterpret_cast<int32_t*>(at);
    at += sub_bsum_bytes(E);
    uint32_t* member = reinterpret_cast<uint32_t*>(at);
    hipError_t he = hipMemsetAsync(newid, 0, static_cast<size_t>(N) * 4, st);
    if (he == hipSuccess) he = hipMemsetAsync(member, 0, static_cast<size_t>((N + 31) / 32) * 4, st);
    if (he == hipSuccess) he = hipMemsetAsync(status, 0, 4, st);
    if (he != hipSuccess) return dif::fail(static_cast<int>(he), "dif_subgraph: memset: %s", hipGetErrorString(he));
    if (B > 0) {
        hipLaunchKernelGGL(subgraph_mark_kernel, dim3(static_cast<unsigned>((B + 255) / 256)), dim3(256), 0, st, subset, B,
                           N, newid, member, status);
        if (int rc = dif::launch_status("subgraph_mark_kernel")) return rc;
    }
    const int64_t n_chunks = (E + kSubChunk - 1) / kSubChunk;
    const int64_t cap = 8 * dif::kCUs;
    int64_t g = n_chunks + 1;
    if (g > cap) g = cap;
    hipLaunchKernelGGL(subgraph_mask_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0, st, edge_index, E, N, member, mask,
                       counts, n_chunks, status);
    if (int rc = dif::launch_status("subgraph_mask_kernel")) return rc;
    if (int rc = exclusive_scan(counts, n_chunks + 1, counts, nullptr, bsum, st)) return rc;
    int64_t g2 = (n_chunks + 3) / 4;
    if (g2 > cap) g2 = cap;
    if (g2 < 1) g2 = 1;
    hipLaunchKernelGGL(subgraph_compact_kernel, dim3(static_cast<unsigned>(g2)), dim3(256), 0, st, edge_index, E, edge_weight,
                       newid, mask, counts, n_chunks, E, out_edge_index, out_weight, out_count);
    return dif::launch_status("subgraph_compact_kernel");
}

extern "C" size_t dif_subgraph_batches_workspace_bytes(int64_t E, int64_t N, int n_batches) {
    if (E < 0 || N <= 0 || n_batches <= 0) return 0;
    return make_batch_plan(E, N, n_batches).total;
}

// Phase 1: batch_ptr int64[n_batches + 1] (device) <- offsets of every batch's edges in the grouped order; the kept
// count is its last element.  The workspace keeps the grouped edge ids for phase 2.
extern "C" int dif_subgraph_batches_group(const int64_t* edge_index, int64_t E, int64_t N, const int64_t* perm, int64_t M,
                                          int64_t batch_size, int64_t* batch_ptr, int32_t* status, void* workspace,
                                          size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(N > 0 && E >= 0 && M > 0 && batch_size > 0, DIF_E_BADARG, "dif_subgraph_batches: need N, M, batch_size > 0, E >= 0");
    DIF_REQUIRE(E < (int64_t(1) << 31) - 4096 && N < (int64_t(1) << 31) - 1 && M <= N && batch_size < (int64_t(1) << 31),
                DIF_E_RANGE, "dif_subgraph_batches: sizes must fit int32 and M <= N");
    const int64_t nb64 = (M + batch_size - 1) / batch_size;
    DIF_REQUIRE(nb64 < 65535, DIF_E_RANGE, "dif_subgraph_batches: at most 65,534 batches");
    const int n_batches = static_cast<int>(nb64);
    DIF_REQUIRE(batch_ptr && status && workspace && perm && (E == 0 || edge_index), DIF_E_BADARG, "dif_subgraph_batches: null pointer");
    const BatchPlan p = make_batch_plan(E, N, n_batches);
    DIF_REQUIRE(workspace_bytes >= p.total, DIF_E_WORKSPACE, "dif_subgraph_batches: workspace too small");
    DIF_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, DIF_E_BADARG, "dif_subgraph_batches: workspace must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    int32_t* info = reinterpret_cast<int32_t*>(ws + p.off_info);
    uint32_t *ka = reinterpret_cast<uint32_t*>(ws + p.off_ka), *kb = reinterpret_cast<uint32_t*>(ws + p.off_kb);
    uint32_t *va = reinterpret_cast<uint32_t*>(ws + p.off_va), *vb = reinterpret_cast<uint32_t*>(ws + p.off_vb);
    int32_t* table = reinterpret_cast<int32_t*>(ws + p.off_table);
    int32_t* bsum = reinterpret_cast<int32_t*>(ws + p.off_bsum);
    hipError_t he = hipMemsetAsync(info, 0, static_cast<size_t>(N) * 4, st);
    if (he == hipSuccess) he = hipMemsetAsync(status, 0, 4, st);
    if (he != hipSuccess) return dif::fail(static_cast<int>(he), "dif_subgraph_batches: memset: %s", hipGetErrorString(he));
    hipLaunchKernelGGL(batches_mark_kernel, dim3(static_cast<unsigned>((M + 255) / 256)), dim3(256), 0, st, perm, M, N, info, status);
    if (int rc = dif::launch_status("batches_mark_kernel")) return rc;
    const int64_t cap = 8 * dif::kCUs;
    int64_t g = (E + 1 + 255) / 256;
    if (g > cap) g = cap;
    if (E > 0) {
        hipLaunchKernelGGL(batches_key_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0, st, edge_index, E, N, info,
                           static_cast<int32_t>(batch_size), p.passes > 1 ? 1 : 0, ka, va, status);
        if (int rc = dif::launch_status("batches_key_kernel")) return rc;
        const unsigned sort_grid = static_cast<unsigned>((p.n_chunks + kSortWaves - 1) / kSortWaves);
        uint32_t *kin = ka, *kout = kb, *vin = va, *vout = vb;
        for (int pass = 0; pass < p.passes; ++pass) {
            const int shift = pass * kRadixBits;
            hipLaunchKernelGGL(radix_hist_kernel, dim3(sort_grid), dim3(64 * kSortWaves), 0, st, kin, E, shift, p.n_chunks, p.rounds, table);
            if (int rc = dif::launch_status("radix_hist_kernel")) return rc;
            if (int rc = exclusive_scan(table, p.table_len, table, nullptr, bsum, st)) return rc;
            hipLaunchKernelGGL(radix_scatter_kernel, dim3(sort_grid), dim3(64 * kSortWaves), 0, st, kin, vin, E, shift, p.n_chunks,
                               p.rounds, table, kout, vout);
            if (int rc = dif::launch_status("radix_scatter_kernel")) return rc;
            uint32_t* t = kin; kin = kout; kout = t;
            t = vin; vin = vout; vout = t;
        }
        // sorted keys / ids now sit in (kin, vin): after one pass that is (kb, vb), after two (ka, va)
        hipLaunchKernelGGL(batches_ptr_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0, st, kin, E, n_batches, batch_ptr);
        return dif::launch_status("batches_ptr_kernel");
    }
    he = hipMemsetAsync(batch_ptr, 0, static_cast<size_t>(n_batches + 1) * 8, st);
    return he == hipSuccess ? 0 : dif::fail(static_cast<int>(he), "dif_subgraph_batches: memset: %s", hipGetErrorString(he));
}

// Phase 2: out_edge_index int64 [2, capacity] (row r at offset r * capacity), capacity >= batch_ptr[n_batches]:
// the surviving edges batch by batch, in their original order, ends renumbered inside their batch.
extern "C" int dif_subgraph_batches_emit(const int64_t* edge_index, int64_t E, int64_t N, int64_t M, int64_t batch_size,
                                         const float* edge_weight, const int64_t* batch_ptr, int64_t capacity,
                                         int64_t* out_edge_index, float* out_weight, const void* workspace,
                                         size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(N > 0 && E >= 0 && M > 0 && batch_size > 0 && capacity >= 0, DIF_E_BADARG, "dif_subgraph_batches_emit: bad sizes");
    const int64_t nb64 = (M + batch_size - 1) / batch_size;
    DIF_REQUIRE(nb64 < 65535, DIF_E_RANGE, "dif_subgraph_batches: at most 65,534 batches");
    const int n_batches = static_cast<int>(nb64);
    DIF_REQUIRE(batch_ptr && workspace && (capacity == 0 || (edge_index && out_edge_index)), DIF_E_BADARG, "dif_subgraph_batches_emit: null pointer");
    DIF_REQUIRE((edge_weight == nullptr) == (out_weight == nullptr), DIF_E_BADARG,
                "dif_subgraph_batches_emit: edge_weight and out_weight must be given together");
    const BatchPlan p = make_batch_plan(E, N, n_batches);
    DIF_REQUIRE(workspace_bytes >= p.total, DIF_E_WORKSPACE, "dif_subgraph_batches_emit: workspace too small");
    if (capacity == 0 || E == 0) return 0;
    const char* ws = static_cast<const char*>(workspace);
    const int32_t* info = reinterpret_cast<const int32_t*>(ws + p.off_info);
    const uint32_t* vals = reinterpret_cast<const uint32_t*>(ws + (p.passes == 1 ? p.off_vb : p.off_va));
    int64_t g = (capacity + 255) / 256;
    if (g > 8 * dif::kCUs) g = 8 * dif::kCUs;
    hipLaunchKernelGGL(batches_emit_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0, static_cast<hipStream_t>(stream), edge_index,
                       E, edge_weight, info, static_cast<int32_t>(batch_size), vals, batch_ptr, n_batches, capacity,
                       out_edge_index, out_weight);
    return dif::launch_status("batches_emit_kernel");
}

extern "C" size_t dif_subgraph_batches_csr_workspace_bytes(int64_t kept, int64_t M) {
    if (kept < 0 || M <= 0) return 0;
    return make_batch_csr_plan(kept, M).total;
}

// Phase 2b (optional): the CSR of EVERY batch from one sort.  rowptr int32[M + 1] over the positions of the permutation
// (row b*batch_size + j = node j of batch b), src int32[kept] (batch-local source ids), val float32[kept]: the slice
// [rowptr[b*batch_size], rowptr[min((b+1)*batch_size, M)]) = [batch_ptr[b], batch_ptr[b+1]) is exactly what dif_csr_build
// returns for that batch's edge list (entries in original edge order, degrees and normalisation of the subgraph), with
// rowptr shifted by batch_ptr[b].  kept = batch_ptr[n_batches] as read back by the caller; group_workspace is the
// workspace of dif_subgraph_batches_group.
extern "C" int dif_subgraph_batches_csr(const int64_t* edge_index, int64_t E, int64_t N, int64_t M, int64_t batch_size,
                                        const float* edge_weight, int64_t kept, const void* group_workspace,
                                        size_t group_workspace_bytes, int32_t* rowptr, int32_t* src, float* val,
                                        void* workspace, size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(N > 0 && E >= 0 && M > 0 && batch_size > 0 && kept >= 0 && kept <= E, DIF_E_BADARG, "dif_subgraph_batches_csr: bad sizes");
    const int64_t nb64 = (M + batch_size - 1) / batch_size;
    DIF_REQUIRE(nb64 < 65535, DIF_E_RANGE, "dif_subgraph_batches: at most 65,534 batches");
    DIF_REQUIRE(rowptr && group_workspace && workspace && (kept == 0 || (edge_index && src && val)), DIF_E_BADARG,
                "dif_subgraph_batches_csr: null pointer");
    const BatchPlan gp = make_batch_plan(E, N, static_cast<int>(nb64));
    DIF_REQUIRE(group_workspace_bytes >= gp.total, DIF_E_WORKSPACE, "dif_subgraph_batches_csr: group workspace too small");
    const BatchCsrPlan p = make_batch_csr_plan(kept, M);
    DIF_REQUIRE(workspace_bytes >= p.total, DIF_E_WORKSPACE, "dif_subgraph_batches_csr: workspace too small");
    DIF_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, DIF_E_BADARG, "dif_subgraph_batches_csr: workspace must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const char* gws = static_cast<const char*>(group_workspace);
    const int32_t* info = reinterpret_cast<const int32_t*>(gws + gp.off_info);
    const uint32_t* grouped = reinterpret_cast<const uint32_t*>(gws + (gp.passes == 1 ? gp.off_vb : gp.off_va));
    char* ws = static_cast<char*>(workspace);
    uint32_t *ka = reinterpret_cast<uint32_t*>(ws + p.off_ka), *kb = reinterpret_cast<uint32_t*>(ws + p.off_kb);
    uint32_t *va = reinterpret_cast<uint32_t*>(ws + p.off_va), *vb = reinterpret_cast<uint32_t*>(ws + p.off_vb);
    float* dinv = reinterpret_cast<float*>(ws + p.off_dinv);
    int32_t* table = reinterpret_cast<int32_t*>(ws + p.off_table);
    int32_t* bsum = reinterpret_cast<int32_t*>(ws + p.off_bsum);
    if (kept == 0) {
        hipError_t he = hipMemsetAsync(rowptr, 0, static_cast<size_t>(M + 1) * 4, st);
        return he == hipSuccess ? 0 : dif::fail(static_cast<int>(he), "dif_subgraph_batches_csr: memset: %s", hipGetErrorString(he));
    }
    const int64_t cap = 8 * dif::kCUs;
    int64_t g = (kept + 1 + 255) / 256;
    if (g > cap) g = cap;
    hipLaunchKernelGGL(batches_csr_key_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0, st, edge_index, E, info, grouped, kept, ka, va);
    if (int rc = dif::launch_status("batches_csr_key_kernel")) return rc;
    const unsigned sort_grid = static_cast<unsigned>((p.n_chunks + kSortWaves - 1) / kSortWaves);
    uint32_t *kin = ka, *kout = kb, *vin = va, *vout = vb;
    for (int pass = 0; pass < p.passes; ++pass) {
        const int shift = pass * kRadixBits;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(sort_grid), dim3(64 * kSortWaves), 0, st, kin, kept, shift, p.n_chunks, p.rounds, table);
        if (int rc = dif::launch_status("radix_hist_kernel")) return rc;
        if (int rc = exclusive_scan(table, p.table_len, table, nullptr, bsum, st)) return rc;
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(sort_grid), dim3(64 * kSortWaves), 0, st, kin, vin, kept, shift, p.n_chunks,
                           p.rounds, table, kout, vout);
        if (int rc = dif::launch_status("radix_scatter_kernel")) return rc;
        uint32_t* t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    hipLaunchKernelGGL(csr_bounds_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0, st, kin, kept, M, rowptr);
    if (int rc = dif::launch_status("csr_bounds_kernel")) return rc;
    hipLaunchKernelGGL(batches_csr_dinv_kernel, dim3(static_cast<unsigned>((M + 255) / 256)), dim3(256), 0, st, rowptr, M, dinv);
    if (int rc = dif::launch_status("batches_csr_dinv_kernel")) return rc;
    hipLaunchKernelGGL(batches_csr_fill_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0, st, edge_index, E, edge_weight, info,
                       static_cast<int32_t>(batch_size), kin, vin, kept, dinv, src, val);
    return dif::launch_status("batches_csr_fill_kernel");
}

extern "C" size_t dif_graph_prepare_workspace_bytes(int64_t E, int64_t N, int undirected) {
    if (E < 0 || N <= 0) return 0;
    return make_prep_plan(E, N, undirected).total;
}

// main.py:72-76 / main-batch.py:96-98 in one call: undirected != 0 -> to_undirected (both directions of every edge, equal
// pairs coalesced, sorted by (row, col)); remove_loops != 0 -> remove_self_loops; add_loops != 0 -> add_self_loops (the N
// loops appended at the end).  Without `undirected` the surviving edges keep their order.  out_edge_index int64
// [2, capacity], capacity >= (undirected ? 2E : E) + (add_loops ? N : 0); out_count (device int64) <- edges written.
extern "C" int dif_graph_prepare(const int64_t* edge_index, int64_t E, int64_t N, int undirected, int remove_loops,
                                 int add_loops, int64_t capacity, int64_t* out_edge_index, int64_t* out_count,
                                 int32_t* status, void* workspace, size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(N > 0 && E >= 0, DIF_E_BADARG, "dif_graph_prepare: need N > 0, E >= 0");
    const PrepPlan p = make_prep_plan(E, N, undirected);
    DIF_REQUIRE(p.n < (int64_t(1) << 31) - 4096 && N < (int64_t(1) << 31) - 1, DIF_E_RANGE, "dif_graph_prepare: sizes must fit int32");
    DIF_REQUIRE(capacity >= p.n + (add_loops ? N : 0), DIF_E_BADARG, "dif_graph_prepare: capacity too small");
    DIF_REQUIRE(out_edge_index && out_count && status && workspace && (E == 0 || edge_index), DIF_E_BADARG, "dif_graph_prepare: null pointer");
    DIF_REQUIRE(workspace_bytes >= p.total, DIF_E_WORKSPACE, "dif_graph_prepare: workspace too small");
    DIF_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, DIF_E_BADARG, "dif_graph_prepare: workspace must be 256-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    uint32_t *ka = reinterpret_cast<uint32_t*>(ws + p.off_ka), *kb = reinterpret_cast<uint32_t*>(ws + p.off_kb);
    uint32_t *va = reinterpret_cast<uint32_t*>(ws + p.off_va), *vb = reinterpret_cast<uint32_t*>(ws + p.off_vb);
    int32_t* keep = reinterpret_cast<int32_t*>(ws + p.off_keep);
    int32_t* table = reinterpret_cast<int32_t*>(ws + p.off_table);
    int32_t* bsum = reinterpret_cast<int32_t*>(ws + p.off_bsum);
    hipError_t he = hipMemsetAsync(status, 0, 4, st);
    if (he != hipSuccess) return dif::fail(static_cast<int>(he), "dif_graph_prepare: memset: %s", hipGetErrorString(he));
    const int64_t cap = 8 * dif::kCUs;
    int64_t g = (p.n + 1 + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    uint32_t *row = va, *col = ka;           // after prep_pairs: key = col, payload = row
    if (E > 0) {
        hipLaunchKernelGGL(prep_pairs_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0, st, edge_index, E, N, undirected, ka, va, status);
        if (int rc = dif::launch_status("prep_pairs_kernel")) return rc;
    }
    if (undirected && p.n > 0) {
        const unsigned sort_grid = static_cast<unsigned>((p.n_chunks + kSortWaves - 1) / kSortWaves);
        uint32_t *kin = ka, *kout = kb, *vin = va, *vout = vb;
        for (int phase = 0; phase < 2; ++phase) {          // phase 0: by col, phase 1: by row (stable) -> (row, col) order
            for (int pass = 0; pass < p.passes; ++pass) {
                const int shift = pass * kRadixBits;
                hipLaunchKernelGGL(radix_hist_kernel, dim3(sort_grid), dim3(64 * kSortWaves), 0, st, kin, p.n, shift, p.n_chunks, p.rounds, table);
                if (int rc = dif::launch_status("radix_hist_kernel")) return rc;
                if (int rc = exclusive_scan(table, p.table_len, table, nullptr, bsum, st)) return rc;
                hipLaunchKernelGGL(radix_scatter_kernel, dim3(sort_grid), dim3(64 * kSortWaves), 0, st, kin, vin, p.n, shift, p.n_chunks,
                                   p.rounds, table, kout, vout);
                if (int rc = dif::launch_status("radix_scatter_kernel")) return rc;
                uint32_t* t = kin; kin = kout; kout = t;
                t = vin; vin = vout; vout = t;
            }
            if (phase == 0) {                               // keys <- rows, payload <- cols
                hipLaunchKernelGGL(prep_swap_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0, st, kin, vin, p.n);
                if (int rc = dif::launch_status("prep_swap_kernel")) return rc;
            }
        }
        row = kin;
        col = vin;
    }
    hipLaunchKernelGGL(prep_flag_kernel, dim3(static_cast<unsigned>(g)), dim3(256), 0, st, row, col, p.n, undirected, remove_loops, keep);
    if (int rc = dif::launch_status("prep_flag_kernel")) return rc;
    if (int rc = exclusive_scan(keep, p.n + 1, keep, nullptr, bsum, st)) return rc;
    int64_t g2 = ((p.n > N ? p.n : N) + 255) / 256;
    if (g2 > cap) g2 = cap;
    if (g2 < 1) g2 = 1;
    hipLaunchKernelGGL(prep_emit_kernel, dim3(static_cast<unsigned>(g2)), dim3(256), 0, st, row, col, p.n, keep, N, add_loops, capacity,
                       out_edge_index, out_count);
    return dif::launch_status("prep_emit_kernel");
}

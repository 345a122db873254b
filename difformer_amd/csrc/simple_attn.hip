// a1: full_attention_conv(..., kernel='simple')  -- node classification/difformer.py:18-39
//
// The reference spends ~17 ATen launches on
//     out = ((q/|Q|) (k/|K|)^T v + sum_l v) / ((q/|Q|) . sum_l (k/|K|) + N)
// Re-associated (difformer.py:25-26 already does the K^T V first) this is two streaming passes:
//   reduce : KtV[h] = K_h^T V_h (MxD), ksum[h], vsum[h], sum q*q, sum k*k      (reads Q,K,V once)
//   apply  : out = (s q KtV + vsum) / (s q.ksum + N),  s = 1/(|Q||K|)          (reads Q, writes out)
// Both contractions run on the exact-f32 matrix core op v_mfma_f32_16x16x4_f32: the 4-row
// contraction step of that shape matches a wave reading 4 whole 256-B rows with one
// global_load_dwordx4 (16 lanes x 16 B per row), so the HBM stream stays fully coalesced and
// no LDS transpose is needed.  HBM-bound: 4*N*H*D*4 bytes per layer.
#include <type_traits>
#include "dif_common.h"
#include "rowgemm_split.h"

namespace {

using dif::f32x4;
using dif::Elem;

constexpr int kTile = 64;       // m / d extent one workgroup accumulates (16 MFMA tiles of 16x16)
constexpr int kRedWaves = 4;    // waves per reduce workgroup
constexpr int kRedUnroll = 4;   // 4-row steps in flight per wave iteration (16 rows, 4 KB per stream)
constexpr int kMaxChunks = 512; // row chunks = partial records = workgroups along x

struct Shape {
    int H, M, D, MT, DT;
    int t_main;  // H*M*D + H*M + H*D
    int tiles;   // H*MT*DT
};

__host__ __device__ inline Shape make_shape(int H, int M, int D) {
    Shape s;
    s.H = H; s.M = M; s.D = D;
    s.MT = (M + kTile - 1) / kTile;
    s.DT = (D + kTile - 1) / kTile;
    s.t_main = H * M * D + H * M + H * D;
    s.tiles = H * s.MT * s.DT;
    return s;
}

// 4 consecutive floats of row r starting at column c (c % 4 == 0); zero outside [0,n) x [0,width).
template <bool VEC, typename T>
__device__ __forceinline__ f32x4 load_row4(const T* __restrict__ base, int64_t ld, int64_t r,
                                           int64_t n, int col0, int c, int width) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (r >= n) return z;
    const T* p = base + r * ld + col0 + c;
    if (VEC) {
        if (c < width) z = Elem<T>::ld4(p);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (c + i < width) z[i] = Elem<T>::ld(p + i);
    }
    return z;
}

// ------------------------------------------------------------------------------------------
// reduce: grid (P row chunks, H*MT*DT tiles), 256 threads.  Per 4-row step a lane holds
//   kx = K[r0 + lane/16][m0 + 4*(lane%16) .. +3],  vx = V[same row][d0 + 4*(lane%16) .. +3]
// and MFMA (t,u) accumulates  D[i][j] += sum_k A[i][k] B[k][j]  with A[i=lane%16][k=lane/16] =
// kx[t], B[k][j=lane%16] = vx[u], i.e. KtV[m0 + 4i + t][d0 + 4j + u].
// ------------------------------------------------------------------------------------------
// SYM (Gram record X^T X of the closed form at the scripts' widths: q = k = v = x, one head, M == D): only the tiles
// with mt <= dt are computed (blockIdx.y enumerates them row by row), a diagonal tile loads its rows once and also
// yields the column sums of its block; the strictly lower tiles of the record stay unwritten (the caller mirrors).
template <bool VEC, typename T, bool SYM = false>
__global__ __launch_bounds__(256) void simple_reduce_kernel(
    const T* __restrict__ q, int64_t ldq, const T* __restrict__ k, int64_t ldk,
    const T* __restrict__ v, int64_t ldv, int64_t n_rows, Shape sh, float* __restrict__ ws,
    int64_t ws_stride) {
    __shared__ __attribute__((aligned(16))) float sm_tile[2][kTile * kTile];      // 32 KiB: three workgroups per CU (was 64 KiB: two)
    __shared__ float sm_k[kRedWaves][kTile];
    __shared__ float sm_v[kRedWaves][kTile];
    __shared__ float sm_s[kRedWaves][2];

    const int y = blockIdx.y;
    int dt = y % sh.DT;
    int mt = (y / sh.DT) % sh.MT;
    const int h = SYM ? 0 : y / (sh.DT * sh.MT);
    if (SYM) {                                   // y -> (mt, dt), mt <= dt, row by row
        int rest = y;
        mt = 0;
        while (rest >= sh.MT - mt) { rest -= sh.MT - mt; ++mt; }
        dt = mt + rest;
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int mc = mt * kTile + 4 * l15;  // this lane's first m column inside head h
    const int dc = dt * kTile + 4 * l15;
    const bool do_k = SYM ? (dt == mt) : (dt == 0);
    const bool do_v = SYM ? false : (mt == 0);
    const bool do_q = SYM ? false : do_k;  // every m-tile of a head squares its own q columns exactly once

    f32x4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ks = {0.f, 0.f, 0.f, 0.f}, vs = {0.f, 0.f, 0.f, 0.f};
    float ksq = 0.f, qsq = 0.f;

    const int64_t n_steps = (n_rows + 3) / 4;
    const int64_t n_iters = (n_steps + kRedUnroll - 1) / kRedUnroll;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kRedWaves + wave;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kRedWaves;

    // the next iteration's rows are requested before this iteration's 64 MFMAs (round 4: without the prefetch a wave sat out
    // the whole load latency between two MFMA bursts -- the Gram pass of the wide closed form ran at 25 % of the fp32 MFMA peak)
    f32x4 nk[kRedUnroll], nv[kRedUnroll], nq[kRedUnroll];
    auto fetch = [&](int64_t it) {
#pragma unroll
        for (int s = 0; s < kRedUnroll; ++s) {
            const int64_t r = (it * kRedUnroll + s) * 4 + lg;
            nk[s] = load_row4<VEC>(k, ldk, r, n_rows, h * sh.M, mc, sh.M);
            if (!(SYM && dt == mt)) nv[s] = load_row4<VEC>(v, ldv, r, n_rows, h * sh.D, dc, sh.D);
            if (do_q) nq[s] = load_row4<VEC>(q, ldq, r, n_rows, h * sh.M, mc, sh.M);
        }
    };
    // (the Gram pass only -- SYM: one or two streams.  With three streams the second register set costs a wave per SIMD and
    // the attention's own reduce ran 20 % slower: it keeps the plain loop.)
    if (SYM && first < n_iters) fetch(first);
    for (int64_t it = first; it < n_iters; it += stride) {
        f32x4 kx[kRedUnroll], vx[kRedUnroll], qx[kRedUnroll];
        if (!SYM) fetch(it);
#pragma unroll
        for (int s = 0; s < kRedUnroll; ++s) {
            kx[s] = nk[s];
            vx[s] = (SYM && dt == mt) ? nk[s] : nv[s];
            if (do_q) qx[s] = nq[s];
        }
        if (SYM && it + stride < n_iters) fetch(it + stride);
#pragma unroll
        for (int s = 0; s < kRedUnroll; ++s) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kx[s][t], vx[s][u], acc[t][u], 0, 0, 0);
            if (do_k) {
                ks += kx[s];
                ksq += kx[s][0] * kx[s][0] + kx[s][1] * kx[s][1] + kx[s][2] * kx[s][2] + kx[s][3] * kx[s][3];
            }
            if (do_v) vs += vx[s];
            if (do_q) qsq += qx[s][0] * qx[s][0] + qx[s][1] * qx[s][1] + qx[s][2] * qx[s][2] + qx[s][3] * qx[s][3];
        }
    }

    // ---- fold the 4 waves (fixed order: deterministic) ---------------------------------
    // lane owns KtV_local[16*lg + 4*reg + t][4*l15 + u]: for fixed (t,reg) the 4 u's are one float4.  The tiles go
    // through two LDS slabs, summed in the order ((w0 + w1) + w2) + w3: slab 0 holds the running sum, slab 1 the next wave.
    auto park = [&](int slab) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const f32x4 val = {acc[t][0][reg], acc[t][1][reg], acc[t][2][reg], acc[t][3][reg]};
                *reinterpret_cast<f32x4*>(&sm_tile[slab][(16 * lg + 4 * reg + t) * kTile + 4 * l15]) = val;
            }
    };
    if (wave < 2) park(wave);
    // column sums: fold the four 16-lane row groups, then the waves
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a = ks[i], b = vs[i];
        a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
        if (lg == 0) { sm_k[wave][4 * l15 + i] = a; sm_v[wave][4 * l15 + i] = b; }
    }
    qsq = dif::wave_sum(qsq);
    ksq = dif::wave_sum(ksq);
    if (lane == 0) { sm_s[wave][0] = qsq; sm_s[wave][1] = ksq; }
    __syncthreads();
#pragma unroll 1
    for (int w = 2; w < kRedWaves; ++w) {
        for (int e = threadIdx.x; e < kTile * kTile; e += 256) sm_tile[0][e] += sm_tile[1][e];
        __syncthreads();
        if (wave == w) park(1);
        __syncthreads();
    }

    float* rec = ws + static_cast<int64_t>(blockIdx.x) * ws_stride;
    float* rec_ktv = rec + static_cast<int64_t>(h) * sh.M * sh.D;
    for (int e = threadIdx.x; e < kTile * kTile; e += 256) {
        const int m = mt * kTile + e / kTile, d = dt * kTile + e % kTile;
        if (m < sh.M && d < sh.D)
            rec_ktv[static_cast<int64_t>(m) * sh.D + d] = sm_tile[0][e] + sm_tile[1][e];
    }
    if (threadIdx.x < kTile) {
        const int c = threadIdx.x;
        if (do_k && mt * kTile + c < sh.M)
            rec[sh.H * sh.M * sh.D + h * sh.M + mt * kTile + c] =
                ((sm_k[0][c] + sm_k[1][c]) + sm_k[2][c]) + sm_k[3][c];
        if (do_v && dt * kTile + c < sh.D)
            rec[sh.H * sh.M * sh.D + sh.H * sh.M + h * sh.D + dt * kTile + c] =
                ((sm_v[0][c] + sm_v[1][c]) + sm_v[2][c]) + sm_v[3][c];
    }
    if (threadIdx.x == 0) {
        rec[sh.t_main + 2 * y + 0] = do_q ? ((sm_s[0][0] + sm_s[1][0]) + sm_s[2][0]) + sm_s[3][0] : 0.f;
        rec[sh.t_main + 2 * y + 1] = do_k ? ((sm_s[0][1] + sm_s[1][1]) + sm_s[2][1]) + sm_s[3][1] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// apply: out^T tile = (s KtV)^T . Q^T on the MFMA:  D[i][j] = sum_k A[i][k] B[k][j] with
//   i <-> d, j <-> row, k <-> m.  Per 16-row step lane holds qv[c] = Q[r0 + lane%16][16c +
//   4*(lane/16) .. +3]; k-step (c,t) contracts m = 16c + 4k + t, so B = qv[c][t] and
//   A = s*KtV[16c + 4*(lane/16) + t][16*dtl + lane%16] (register-resident when M,D <= 64).
//   The lane ends up with out[r0 + lane%16][16*dtl + 4*(lane/16) .. +3]: one float4 store.
// ------------------------------------------------------------------------------------------
template <bool VEC, bool SINGLE, typename T>
__global__ __launch_bounds__(256) void simple_apply_kernel(const T* __restrict__ q, int64_t ldq,
                                                           const float* __restrict__ reduced,
                                                           int64_t n_rows, float n_global, Shape sh,
                                                           T* __restrict__ out, int64_t ldo) {
    const int h = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const float* ktv = reduced + static_cast<int64_t>(h) * sh.M * sh.D;
    const float* ksum = reduced + sh.H * sh.M * sh.D + h * sh.M;
    const float* vsum = reduced + sh.H * sh.M * sh.D + sh.H * sh.M + h * sh.D;
    // difformer.py:20-21: qs / ||qs||, ks / ||ks|| with the norm over the WHOLE tensor
    const float s = 1.0f / (sqrtf(reduced[sh.t_main]) * sqrtf(reduced[sh.t_main + 1]));

    float afrag[4][4][4];  // [dtl][c][t]
    float kfrag[4][4];     // [c][t] = s * ksum[16c + 4lg + t]
    auto load_frags = [&](int mt, int dt) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = mt * kTile + 16 * c + 4 * lg + t;
                kfrag[c][t] = (m < sh.M) ? s * ksum[m] : 0.f;
#pragma unroll
                for (int dtl = 0; dtl < 4; ++dtl) {
                    const int d = dt * kTile + 16 * dtl + l15;
                    afrag[dtl][c][t] = (m < sh.M && d < sh.D) ? s * ktv[static_cast<int64_t>(m) * sh.D + d] : 0.f;
                }
            }
    };
    if (SINGLE) {
        // the 64x64 record tile is fetched once per workgroup (coalesced) and handed to the lanes'
        // fragment registers through LDS; each wave then streams many 16-row steps with it
        __shared__ __attribute__((aligned(16))) float sm_ktv[kTile * (kTile + 4)];
        __shared__ float sm_ks[kTile];
        if (sh.M == kTile && sh.D == kTile) {
            // full tile: four 16-byte loads per thread, all in flight before the first LDS store (the element loop
            // below is 16 load -> store round trips)
            f32x4 r4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) r4[i] = *reinterpret_cast<const f32x4*>(ktv + 4 * (threadIdx.x + 256 * i));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * (threadIdx.x + 256 * i);
                *reinterpret_cast<f32x4*>(&sm_ktv[(e / kTile) * (kTile + 4) + e % kTile]) = s * r4[i];
            }
        } else {
            for (int e = threadIdx.x; e < kTile * kTile; e += 256) {
                const int m = e / kTile, d = e % kTile;
                sm_ktv[m * (kTile + 4) + d] = (m < sh.M && d < sh.D) ? s * ktv[static_cast<int64_t>(m) * sh.D + d] : 0.f;
            }
        }
        if (threadIdx.x < kTile) sm_ks[threadIdx.x] = (threadIdx.x < sh.M) ? s * ksum[threadIdx.x] : 0.f;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = 16 * c + 4 * lg + t;
                kfrag[c][t] = sm_ks[m];
#pragma unroll
                for (int dtl = 0; dtl < 4; ++dtl) afrag[dtl][c][t] = sm_ktv[m * (kTile + 4) + 16 * dtl + l15];
            }
    }

    const int64_t n_steps = (n_rows + 15) / 16;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 4;
    if (SINGLE) {
        // M, D <= 64: one fragment set for the whole sweep; the next step's 16 rows of q are requested before the
        // current step's 64 MFMAs so the HBM latency hides under them
        f32x4 qn[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) qn[c] = load_row4<VEC>(q, ldq, first * 16 + l15, n_rows, h * sh.M, 16 * c + 4 * lg, sh.M);
        for (int64_t st = first; st < n_steps; st += stride) {
            const int64_t r = st * 16 + l15;
            f32x4 qv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) qv[c] = qn[c];
            if (st + stride < n_steps) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    qn[c] = load_row4<VEC>(q, ldq, (st + stride) * 16 + l15, n_rows, h * sh.M, 16 * c + 4 * lg, sh.M);
            }
            f32x4 acc[4];
#pragma unroll
            for (int dtl = 0; dtl < 4; ++dtl) acc[dtl] = f32x4{0.f, 0.f, 0.f, 0.f};
            float dpart = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int dtl = 0; dtl < 4; ++dtl)
                        acc[dtl] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[dtl][c][t], qv[c][t], acc[dtl], 0, 0, 0);
                    dpart += qv[c][t] * kfrag[c][t];
                }
            dpart += __shfl_xor(dpart, 16, 64);
            dpart += __shfl_xor(dpart, 32, 64);
            const float den = dpart + n_global;  // difformer.py:37-38
            if (r < n_rows) {
#pragma unroll
                for (int dtl = 0; dtl < 4; ++dtl) {
                    const int d0 = 16 * dtl + 4 * lg;
                    T* o = out + r * ldo + h * sh.D + d0;
                    if (VEC) {
                        if (d0 < sh.D) Elem<T>::st4(o, (acc[dtl] + *reinterpret_cast<const f32x4*>(vsum + d0)) / den);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (d0 + i < sh.D) Elem<T>::st(o + i, (acc[dtl][i] + vsum[d0 + i]) / den);
                    }
                }
            }
        }
        return;
    }
    for (int64_t st = first; st < n_steps; st += stride) {
        const int64_t r = st * 16 + l15;
        float den = 0.f;
        for (int dt = 0; dt < sh.DT; ++dt) {
            f32x4 acc[4];
#pragma unroll
            for (int dtl = 0; dtl < 4; ++dtl) acc[dtl] = f32x4{0.f, 0.f, 0.f, 0.f};
            float dpart = 0.f;
            for (int mt = 0; mt < sh.MT; ++mt) {
                if (!SINGLE) load_frags(mt, dt);
                f32x4 qv[4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    qv[c] = load_row4<VEC>(q, ldq, r, n_rows, h * sh.M, mt * kTile + 16 * c + 4 * lg, sh.M);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int dtl = 0; dtl < 4; ++dtl)
                            acc[dtl] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[dtl][c][t], qv[c][t], acc[dtl], 0, 0, 0);
                        dpart += qv[c][t] * kfrag[c][t];
                    }
            }
            if (dt == 0) {
                // q . (s ksum): fold the 4 lane groups that share this row
                dpart += __shfl_xor(dpart, 16, 64);
                dpart += __shfl_xor(dpart, 32, 64);
                den = dpart + n_global;  // difformer.py:37-38
            }
            if (r < n_rows) {
#pragma unroll
                for (int dtl = 0; dtl < 4; ++dtl) {
                    const int d0 = dt * kTile + 16 * dtl + 4 * lg;
                    T* o = out + r * ldo + h * sh.D + d0;
                    if (VEC) {
                        if (d0 < sh.D) {
                            const f32x4 vs4 = *reinterpret_cast<const f32x4*>(vsum + d0);
                            Elem<T>::st4(o, (acc[dtl] + vs4) / den);  // :29, :39
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (d0 + i < sh.D) Elem<T>::st(o + i, (acc[dtl][i] + vsum[d0 + i]) / den);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// apply for wide heads (64 < M <= 512, any D): the widths the reference's scripts train with (run.sh: hidden 128, 300,
// 400).  A workgroup owns 64 output columns (grid.y) and keeps s * KtV[:, those columns] TRANSPOSED in LDS for the whole
// sweep (row d holds the M contraction values contiguously, +4 floats of padding: the 16 lanes of a ds_read_b128 phase
// hit 16 different bank quads), so the A operand of every MFMA comes from one 16-byte LDS read per four k-steps and the
// only global traffic in the loop is q (read once per 64 output columns) and the output tile.
// ------------------------------------------------------------------------------------------
constexpr int kWideWaves = 8;

template <bool VEC, typename T>
__global__ __launch_bounds__(64 * kWideWaves) void simple_apply_wide_kernel(const T* __restrict__ q, int64_t ldq,
                                                                            const float* __restrict__ reduced,
                                                                            int64_t n_rows, float n_global, Shape sh,
                                                                            T* __restrict__ out, int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) float sm_dyn[];
    const int h = blockIdx.z, dt = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int Mp = sh.MT * kTile, ldt = Mp + 4;
    float* smT = sm_dyn;                       // [64][ldt]: smT[dl][m] = s * KtV[m][64 dt + dl]
    float* sm_ks = sm_dyn + kTile * ldt;       // [Mp]: s * ksum
    const float* ktv = reduced + static_cast<int64_t>(h) * sh.M * sh.D;
    const float* ksum = reduced + sh.H * sh.M * sh.D + h * sh.M;
    const float* vsum = reduced + sh.H * sh.M * sh.D + sh.H * sh.M + h * sh.D;
    const float s = 1.0f / (sqrtf(reduced[sh.t_main]) * sqrtf(reduced[sh.t_main + 1]));
    const int total = Mp * kTile;
    for (int base = threadIdx.x; base < total; base += 64 * kWideWaves * 8) {      // eight loads in flight per thread
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = base + 64 * kWideWaves * u;
            const int m = e >> 6, d = dt * kTile + (e & 63);
            v[u] = (e < total && m < sh.M && d < sh.D) ? s * ktv[static_cast<int64_t>(m) * sh.D + d] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = base + 64 * kWideWaves * u;
            if (e < total) smT[(e & 63) * ldt + (e >> 6)] = v[u];
        }
    }
    for (int m = threadIdx.x; m < Mp; m += 64 * kWideWaves) sm_ks[m] = (m < sh.M) ? s * ksum[m] : 0.f;
    __syncthreads();

    const int64_t n_steps = (n_rows + 15) / 16;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kWideWaves + wave;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kWideWaves;
    auto load_q = [&](f32x4 (&qv)[4], int64_t st, int mt) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            qv[c] = load_row4<VEC>(q, ldq, st * 16 + l15, n_rows, h * sh.M, mt * kTile + 16 * c + 4 * lg, sh.M);
    };
    f32x4 qn[4];
    if (first < n_steps) load_q(qn, first, 0);
    for (int64_t st = first; st < n_steps; st += stride) {
        const int64_t r = st * 16 + l15;
        f32x4 acc[4];
#pragma unroll
        for (int dtl = 0; dtl < 4; ++dtl) acc[dtl] = f32x4{0.f, 0.f, 0.f, 0.f};
        float dpart = 0.f;
        for (int mt = 0; mt < sh.MT; ++mt) {
            f32x4 qv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) qv[c] = qn[c];
            if (mt + 1 < sh.MT) load_q(qn, st, mt + 1);                 // the next 64 channels arrive under these MFMAs
            else if (st + stride < n_steps) load_q(qn, st + stride, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int m0 = mt * kTile + 16 * c + 4 * lg;
#pragma unroll
                for (int dtl = 0; dtl < 4; ++dtl) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(&smT[(16 * dtl + l15) * ldt + m0]);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[dtl] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[t], qv[c][t], acc[dtl], 0, 0, 0);
                }
                const f32x4 k4 = *reinterpret_cast<const f32x4*>(&sm_ks[m0]);
                dpart += qv[c][0] * k4[0] + qv[c][1] * k4[1] + qv[c][2] * k4[2] + qv[c][3] * k4[3];
            }
        }
        dpart += __shfl_xor(dpart, 16, 64);
        dpart += __shfl_xor(dpart, 32, 64);
        const float den = dpart + n_global;                              // difformer.py:37-38
        if (r < n_rows) {
#pragma unroll
            for (int dtl = 0; dtl < 4; ++dtl) {
                const int d0 = dt * kTile + 16 * dtl + 4 * lg;
                T* o = out + r * ldo + h * sh.D + d0;
                if (VEC) {
                    if (d0 < sh.D) Elem<T>::st4(o, (acc[dtl] + *reinterpret_cast<const f32x4*>(vsum + d0)) / den);   // :29, :39
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (d0 + i < sh.D) Elem<T>::st(o + i, (acc[dtl][i] + vsum[d0 + i]) / den);
                }
            }
        }
    }
}

// Row chunks (= partial records = workgroups along x).  One 64 x 64 tile (narrow heads): up to 512 chunks fill the
// chip.  Many tiles (hidden 300: 25 of them): every chunk writes a whole partial record, so the chunk count is cut to
// ~1024 workgroups in all -- at 512 chunks the partials alone were 185 MB written and read back (profiles/r02_experiments.md).
int reduce_chunks(int64_t n_rows, int tiles) {
    const int64_t n_iters = ((n_rows + 3) / 4 + kRedUnroll - 1) / kRedUnroll;
    int64_t p = (n_iters + kRedWaves - 1) / kRedWaves;
    int64_t cap = kMaxChunks;
    if (tiles > 2) cap = (4 * dif::kCUs + tiles - 1) / tiles;
    if (cap < 8) cap = 8;
    if (p > cap) p = cap;
    if (p < 1) p = 1;
    return static_cast<int>(p);
}

int check_shape(int64_t n_rows, int H, int M, int D) {
    DIF_REQUIRE(n_rows > 0 && H > 0 && M > 0 && D > 0, DIF_E_BADARG,
                "simple attention: n_rows, H, M, D must be positive (got %lld, %d, %d, %d)",
                static_cast<long long>(n_rows), H, M, D);
    DIF_REQUIRE(static_cast<int64_t>(H) * M * D + static_cast<int64_t>(H) * (M + D) + 2 < (1ll << 30),
                DIF_E_RANGE, "simple attention: H*M*D too large");
    DIF_REQUIRE(static_cast<int64_t>(H) * ((M + kTile - 1) / kTile) * ((D + kTile - 1) / kTile) <= 65535,
                DIF_E_RANGE, "simple attention: too many head tiles");
    return 0;
}

constexpr int kSlabRows = 32;          // rows of a slab of the one-read kernels below (gram_slab_kernel, reduce_slab_kernel)
__global__ __launch_bounds__(256, 2) void reduce_slab_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k,
                                                          int64_t ldk, const float* __restrict__ v, int64_t ldv, int64_t n_rows,
                                                          int M, int D, float* __restrict__ ws, int64_t ws_stride, int t_main);

template <typename T>
int simple_reduce_entry(const T* q, int64_t ldq, const T* k, int64_t ldk, const T* v, int64_t ldv, int64_t n_rows, int H,
                        int M, int D, float* reduced, void* workspace, size_t workspace_bytes, dif_stream_t stream) {
    if (int rc = check_shape(n_rows, H, M, D)) return rc;
    DIF_REQUIRE(q && k && v && reduced && workspace, DIF_E_BADARG, "dif_simple_reduce: null pointer");
    DIF_REQUIRE(ldq >= H * M && ldk >= H * M && ldv >= H * D, DIF_E_BADARG,
                "dif_simple_reduce: leading dimension smaller than a row");
    DIF_REQUIRE(workspace_bytes >= dif_simple_workspace_bytes(n_rows, H, M, D), DIF_E_WORKSPACE,
                "dif_simple_reduce: workspace too small (%zu < %zu)", workspace_bytes,
                dif_simple_workspace_bytes(n_rows, H, M, D));
    DIF_REQUIRE(dif::aligned16(workspace), DIF_E_BADARG, "dif_simple_reduce: workspace not 16-byte aligned");
    const Shape sh = make_shape(H, M, D);
    const int P = reduce_chunks(n_rows, sh.tiles);
    const int64_t rec = (static_cast<int64_t>(sh.t_main) + 2 * sh.tiles + 3) & ~int64_t(3);
    const bool vec = (M % 4 == 0) && (D % 4 == 0) && (ldq % 4 == 0) && (ldk % 4 == 0) && (ldv % 4 == 0) &&
                     dif::aligned_v4<T>(q) && dif::aligned_v4<T>(k) && dif::aligned_v4<T>(v);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* ws = static_cast<float*>(workspace);
    if constexpr (std::is_same<T, float>::value) {
        // one head of 65..128 columns (training at hidden 128): every operand read once, products on split-bf16 (reduce_slab_kernel)
        if (H == 1 && vec && sh.tiles >= 2 && sh.MT <= 8 && sh.DT <= 2 && n_rows >= 4096 && !dif::exact_fp32()) {
            const int64_t slabs = (n_rows + kSlabRows - 1) / kSlabRows;
            const int Ps = static_cast<int>(slabs < P ? slabs : P);
            hipLaunchKernelGGL(reduce_slab_kernel, dim3(Ps, (sh.MT + 1) / 2), dim3(256), 0, st, q, ldq, k, ldk, v, ldv, n_rows, M, D, ws, rec,
                               sh.t_main);
            if (int rc = dif::launch_status("reduce_slab_kernel")) return rc;
            return dif::launch_record_finalize(ws, Ps, rec, sh.t_main, sh.tiles, reduced, st);
        }
    }
    dim3 grid(P, sh.tiles), block(256);
    if (vec)
        hipLaunchKernelGGL((simple_reduce_kernel<true, T>), grid, block, 0, st, q, ldq, k, ldk, v, ldv, n_rows, sh, ws, rec);
    else
        hipLaunchKernelGGL((simple_reduce_kernel<false, T>), grid, block, 0, st, q, ldq, k, ldk, v, ldv, n_rows, sh, ws, rec);
    if (int rc = dif::launch_status("simple_reduce_kernel")) return rc;
    return dif::launch_record_finalize(ws, P, rec, sh.t_main, sh.tiles, reduced, st);
}

template <typename T>
int simple_apply_entry(const T* q, int64_t ldq, const float* reduced, int64_t n_rows, int64_t n_global, int H, int M,
                       int D, T* out, int64_t ldo, dif_stream_t stream) {
    if (int rc = check_shape(n_rows, H, M, D)) return rc;
    DIF_REQUIRE(q && reduced && out, DIF_E_BADARG, "dif_simple_apply: null pointer");
    DIF_REQUIRE(ldq >= H * M && ldo >= H * D, DIF_E_BADARG, "dif_simple_apply: leading dimension smaller than a row");
    DIF_REQUIRE(n_global >= n_rows, DIF_E_BADARG, "dif_simple_apply: n_global < n_rows");
    DIF_REQUIRE(H <= 65535, DIF_E_RANGE, "dif_simple_apply: too many heads");
    const Shape sh = make_shape(H, M, D);
    const bool vec = (M % 4 == 0) && (D % 4 == 0) && (ldq % 4 == 0) && (ldo % 4 == 0) && dif::aligned_v4<T>(q) &&
                     dif::aligned_v4<T>(out) && dif::aligned16(reduced) && ((H * M * D + H * M) % 4 == 0);
    const bool single = (sh.MT == 1 && sh.DT == 1);
    const int64_t n_steps = (n_rows + 15) / 16;
    if constexpr (std::is_same<T, float>::value) {
        // one head of 65..128 columns on >= 4,096 rows (the scripts' hidden 128): the row-GEMM kernel on split-bfloat16 operands in
        // its APPLY mode -- q read once for all output columns, 96 bf16 MFMAs per 16 rows instead of 2 x 128 fp32 ones
        // (csrc/rowgemm_split.h; ~4e-6 of the float64 result).  DIFFORMER_EXACT_FP32=1 keeps the fp32 kernel below.
        if (H == 1 && !single && M <= 128 && D <= 128 && vec && n_rows >= 4096 && !dif::exact_fp32()) {
            const float* ksum = reduced + static_cast<int64_t>(M) * D;
            const float* vsum = ksum + M;
            return rowgemm_split_launch("dif_simple_apply_f32", static_cast<hipStream_t>(stream), q, ldq, reduced, D, 0, 1.0f, vsum, nullptr,
                                        nullptr, 0.f, nullptr, 0, nullptr, n_rows, M, D, out, ldo, reduced + sh.t_main, ksum,
                                        static_cast<float>(n_global));
        }
    }
    if (!single && sh.MT <= 8) {              // wide heads: 64 output columns per workgroup, s * KtV^T resident in LDS
        const size_t lds = (static_cast<size_t>(kTile) * (sh.MT * kTile + 4) + sh.MT * kTile) * sizeof(float);
        int64_t gxw = (n_steps + 2 * kWideWaves - 1) / (2 * kWideWaves);       // >= 2 steps per wave: the staging is amortised
        const int64_t capw = (lds <= 80 * 1024 ? 2 : 1) * dif::kCUs;
        if (gxw > capw) gxw = capw;
        if (gxw < 1) gxw = 1;
        hipStream_t stw = static_cast<hipStream_t>(stream);
        dim3 gridw(static_cast<unsigned>(gxw), sh.DT, H), blockw(64 * kWideWaves);
        const float nfw = static_cast<float>(n_global);
        // dynamic LDS beyond 64 KiB must be allowed once per kernel (set to the largest case, M = 512)
        constexpr int kWideLdsMax = (kTile * (8 * kTile + 4) + 8 * kTile) * static_cast<int>(sizeof(float));
        static const hipError_t allowed[2] = {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&simple_apply_wide_kernel<false, T>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kWideLdsMax),
            hipFuncSetAttribute(reinterpret_cast<const void*>(&simple_apply_wide_kernel<true, T>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kWideLdsMax)};
        const hipError_t he = allowed[vec ? 1 : 0];
        if (he != hipSuccess) return dif::fail(static_cast<int>(he), "dif_simple_apply: LDS attribute: %s", hipGetErrorString(he));
        if (vec)
            hipLaunchKernelGGL((simple_apply_wide_kernel<true, T>), gridw, blockw, lds, stw, q, ldq, reduced, n_rows, nfw, sh, out, ldo);
        else
            hipLaunchKernelGGL((simple_apply_wide_kernel<false, T>), gridw, blockw, lds, stw, q, ldq, reduced, n_rows, nfw, sh, out, ldo);
        return dif::launch_status("simple_apply_wide_kernel");
    }
    int64_t gx = (n_steps + 3) / 4;
    const int64_t cap = 3 * dif::kCUs;  // persistent: the per-wave fragment prologue is paid once per ~3+ steps
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(static_cast<unsigned>(gx), H), block(256);
    const float nf = static_cast<float>(n_global);
#define DIF_LAUNCH_APPLY(V, S) \
    hipLaunchKernelGGL((simple_apply_kernel<V, S, T>), grid, block, 0, st, q, ldq, reduced, n_rows, nf, sh, out, ldo)
    if (vec && single) DIF_LAUNCH_APPLY(true, true);
    else if (vec) DIF_LAUNCH_APPLY(true, false);
    else if (single) DIF_LAUNCH_APPLY(false, true);
    else DIF_LAUNCH_APPLY(false, false);
#undef DIF_LAUNCH_APPLY
    return dif::launch_status("simple_apply_kernel");
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// Gram record at hidden 129..320 (image and text/run.sh:27 trains at 300), round 5.  simple_reduce_kernel<sym> gives every
// (row chunk, 64 x 64 tile) pair its own workgroup: each of the 15 upper tiles streams its two column blocks from memory
// (x is read ~5 times: 320 MB for 50,000 x 300) and multiplies on the fp32 matrix core (N C^2 flop at 157 TFLOP/s: 57 us for
// the full matrix) -- 84 us per layer, a third of the cifar50k-h300 forward.  Here a workgroup owns a row chunk and ALL the
// upper tiles: a 32-row slab of x is read ONCE (coalesced 16-byte loads, eight rows of one column quad per thread), split into
// bfloat16 hi + lo and written to LDS as ready-made MFMA operands -- for a column block b and offset t the fragment's lane
// (l15, lg) holds rows 8 lg .. 8 lg + 7 of column 64 b + 4 l15 + t, one ds_write_b128 per operand -- and each of the sixteen waves of
// a row chunk's two workgroups multiplies ITS tile from LDS: per (t, u) three v_mfma_f32_16x16x32_bf16 (xl.yh + xh.yl + xh.yh; the dropped xl.yl term is
// 2^-16 of a product, ~1e-6 of the record) where the fp32 path needs eight v_mfma_f32_16x16x4_f32 at twice the cycles.
// Accumulators (64 VGPRs per wave) live in registers for the whole chunk; the partial record of the chunk has the layout
// simple_reduce_kernel<sym> writes, so record_finalize_kernel sums them unchanged.  DIFFORMER_EXACT_FP32=1 keeps the fp32 pass.
// ------------------------------------------------------------------------------------------------------------------
namespace {
typedef __bf16 gs_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gs_bf16x4 __attribute__((ext_vector_type(4)));
constexpr int kSlabMaxMT = 5;

__global__ __launch_bounds__(512) void gram_slab_kernel(const float* __restrict__ x, int64_t ldx, int64_t n_rows, int C, int MT,
                                                        float* __restrict__ ws, int64_t ws_stride) {
    __shared__ __attribute__((aligned(16))) gs_bf16x8 sm_op[2 * kSlabMaxMT * 4 * 64];      // [hi | lo][block][t][lane]: 40 KiB
    __shared__ float sm_sx[4][kSlabMaxMT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int nq = C / 4;                                  // column quads
    // staging role: thread (slg, cq) reads rows 8 slg .. + 7 of the slab, columns 4 cq .. + 3
    const int slg = threadIdx.x / nq, cq = threadIdx.x % nq;
    const bool stager = threadIdx.x < 4 * nq;
    // compute role: the NT upper tiles are dealt to the gridDim.y workgroups of a row chunk, kTilesPerWave to a wave (both
    // workgroups of a chunk stage the whole slab: x is read twice, from L2 / the Infinity Cache the second time, and the chunk
    // still leaves ONE partial record)
    const int NT = MT * (MT + 1) / 2;
    constexpr int kTilesPerWave = 1;
    int mts[kTilesPerWave], dts[kTilesPerWave], tix[kTilesPerWave];
#pragma unroll
    for (int k = 0; k < kTilesPerWave; ++k) {
        tix[k] = (blockIdx.y * 8 + wave) * kTilesPerWave + k;
        int rest = tix[k], m = 0;
        while (m < MT && rest >= MT - m) { rest -= MT - m; ++m; }
        mts[k] = m;
        dts[k] = m + rest;
    }
    f32x4 acc[kTilesPerWave][4][4];
#pragma unroll
    for (int k = 0; k < kTilesPerWave; ++k)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[k][t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 sx = {0.f, 0.f, 0.f, 0.f};

    const int64_t n_slabs = (n_rows + kSlabRows - 1) / kSlabRows;
    f32x4 nxt[8];
    auto fetch = [&](int64_t slab) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t r = slab * kSlabRows + 8 * slg + j;
            nxt[j] = (stager && r < n_rows) ? *reinterpret_cast<const f32x4*>(x + r * ldx + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    const int64_t first = blockIdx.x, stride = gridDim.x;
    if (first < n_slabs) fetch(first);
    for (int64_t slab = first; slab < n_slabs; slab += stride) {
        // ---- registers -> LDS operands ----
        if (stager) {
            gs_bf16x4 h[8], l[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sx += nxt[j];
                h[j] = __builtin_convertvector(nxt[j], gs_bf16x4);
                const f32x4 back = __builtin_convertvector(h[j], f32x4);
                l[j] = __builtin_convertvector(nxt[j] - back, gs_bf16x4);
            }
            const int b = cq >> 4, ln = 16 * slg + (cq & 15);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                sm_op[((0 * kSlabMaxMT + b) * 4 + t) * 64 + ln] = gs_bf16x8{h[0][t], h[1][t], h[2][t], h[3][t], h[4][t], h[5][t], h[6][t], h[7][t]};
                sm_op[((1 * kSlabMaxMT + b) * 4 + t) * 64 + ln] = gs_bf16x8{l[0][t], l[1][t], l[2][t], l[3][t], l[4][t], l[5][t], l[6][t], l[7][t]};
            }
        }
        __syncthreads();
        if (slab + stride < n_slabs) fetch(slab + stride);          // the next slab's rows arrive under this slab's products
#pragma unroll
        for (int k = 0; k < kTilesPerWave; ++k) {
            if (tix[k] >= NT) continue;
            const int mt = mts[k], dt = dts[k];
            gs_bf16x8 bh[4], bl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bh[u] = sm_op[((0 * kSlabMaxMT + dt) * 4 + u) * 64 + lane];
                bl[u] = sm_op[((1 * kSlabMaxMT + dt) * 4 + u) * 64 + lane];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const gs_bf16x8 ah = sm_op[((0 * kSlabMaxMT + mt) * 4 + t) * 64 + lane];
                const gs_bf16x8 al = sm_op[((1 * kSlabMaxMT + mt) * 4 + t) * 64 + lane];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[k][t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[u], acc[k][t][u], 0, 0, 0);      // small terms first
                    acc[k][t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[u], acc[k][t][u], 0, 0, 0);
                    acc[k][t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[u], acc[k][t][u], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // ---- partial record of this chunk: lane owns G[64 mt + 16 lg + 4 reg + t][64 dt + 4 l15 + u] (as simple_reduce_kernel) ----
    float* rec = ws + static_cast<int64_t>(blockIdx.x) * ws_stride;
#pragma unroll
    for (int k = 0; k < kTilesPerWave; ++k) {
        if (tix[k] >= NT) continue;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int m = 64 * mts[k] + 16 * lg + 4 * reg + t, d0 = 64 * dts[k] + 4 * l15;
                if (m < C && d0 < C)
                    *reinterpret_cast<f32x4*>(&rec[static_cast<int64_t>(m) * C + d0]) =
                        f32x4{acc[k][t][0][reg], acc[k][t][1][reg], acc[k][t][2][reg], acc[k][t][3][reg]};
            }
    }
    // column sums: the four row groups of a column quad, in group order
    if (stager) *reinterpret_cast<f32x4*>(&sm_sx[slg][4 * cq]) = sx;
    __syncthreads();
    if (blockIdx.y != 0) return;
    for (int c = threadIdx.x; c < C; c += 512) rec[static_cast<int64_t>(C) * C + c] = ((sm_sx[0][c] + sm_sx[1][c]) + sm_sx[2][c]) + sm_sx[3][c];
    // (the record's two Frobenius slots per tile are not used by the Gram record; zero them so that the finalize sums zeros)
    const int tiles_all = MT * MT;
    for (int i = threadIdx.x; i < 2 * tiles_all; i += 512) rec[static_cast<int64_t>(C) * C + 2 * C + i] = 0.f;
}
}  // namespace

// The same one-read scheme for the GENERAL record K^T V | sum k | sum v | sum q^2 | sum k^2 at one head of 65..128 columns
// (training at hidden 128, node classification/run.sh:42-44: the attention's forward record, q^T gn in its backward and every
// Linear weight gradient g^T x go through dif_simple_reduce_f32 -- nine launches of 88 us per step on the fp32 matrix core, each
// of the four tiles streaming its two column blocks).  Waves 0..3 multiply the (<= 2 x 2) tiles, 128 + 128 + 128 threads stage
// the k slab, the v slab and square the q rows.
namespace {
__global__ __launch_bounds__(256, 2) void reduce_slab_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k,
                                                          int64_t ldk, const float* __restrict__ v, int64_t ldv, int64_t n_rows,
                                                          int M, int D, float* __restrict__ ws, int64_t ws_stride, int t_main) {
    __shared__ __attribute__((aligned(16))) gs_bf16x8 sm_k[2 * 2 * 4 * 64], sm_v[2 * 2 * 4 * 64];      // [hi | lo][block][t][lane]
    __shared__ float sm_sx[2][4][128];
    __shared__ float sm_sq[2][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    // blockIdx.y: which 128 columns of k (two 64-blocks) this workgroup multiplies against ALL of v (<= 128 columns): the fused
    // q | k | v projection's weight gradient is k = g [n, 384], v = x [n, 128] -- three workgroups per row chunk, one partial record
    const int m0 = blockIdx.y * 128;
    const int Ml = min(128, M - m0);                                  // k columns of this workgroup
    const int MTl = (Ml + 63) / 64, DT = (D + 63) / 64, MT = (M + 63) / 64;
    // 256 threads = four waves, two workgroups per CU (~220 VGPRs: two waves per SIMD): twice the loads in flight of one
    // 512-thread workgroup -- the pass is paced by the latency of a slab's loads, not by its 768 matrix cycles.
    // staging roles: threads [0, 128): k, and the squares of the q columns beside them (q has M columns like k: every one is
    // squared exactly once); [128, 256): v.  Thread (slg, cq) of a role reads rows 8 slg .. + 7 of the slab, columns 4 cq .. + 3.
    const int role = threadIdx.x >> 7, rt = threadIdx.x & 127;
    const int width = role == 1 ? D : Ml;
    const int nq = width / 4;
    const int slg = rt / nq, cq = rt % nq;
    const bool stager = rt < 4 * nq;
    const float* src = role == 0 ? k + m0 : v;
    const int64_t lds_ = role == 0 ? ldk : ldv;
    const bool with_q = role == 0 && q != k;                          // (q == k: the squares are the k squares)
    const bool worker = wave < MTl * DT;
    const int mt = wave / DT, dt = wave % DT;
    f32x4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 sx = {0.f, 0.f, 0.f, 0.f};
    float sq = 0.f;
    const int64_t n_slabs = (n_rows + kSlabRows - 1) / kSlabRows;
    f32x4 nxt[8], nxq[8];
    float sqq = 0.f;
    auto fetch = [&](int64_t slab) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t r = slab * kSlabRows + 8 * slg + j;
            nxt[j] = (stager && r < n_rows) ? *reinterpret_cast<const f32x4*>(src + r * lds_ + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (with_q) nxq[j] = (stager && r < n_rows) ? *reinterpret_cast<const f32x4*>(q + m0 + r * ldq + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    const int64_t first = blockIdx.x, stride = gridDim.x;
    if (first < n_slabs) fetch(first);
    for (int64_t slab = first; slab < n_slabs; slab += stride) {
        if (stager) {
            gs_bf16x4 h[8], l[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sx += nxt[j];
                sq += nxt[j][0] * nxt[j][0] + nxt[j][1] * nxt[j][1] + nxt[j][2] * nxt[j][2] + nxt[j][3] * nxt[j][3];
                if (with_q) sqq += nxq[j][0] * nxq[j][0] + nxq[j][1] * nxq[j][1] + nxq[j][2] * nxq[j][2] + nxq[j][3] * nxq[j][3];
                h[j] = __builtin_convertvector(nxt[j], gs_bf16x4);
                const f32x4 back = __builtin_convertvector(h[j], f32x4);
                l[j] = __builtin_convertvector(nxt[j] - back, gs_bf16x4);
            }
            {
                gs_bf16x8* op = role == 0 ? sm_k : sm_v;
                const int b = cq >> 4, ln = 16 * slg + (cq & 15);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    op[((0 * 2 + b) * 4 + t) * 64 + ln] = gs_bf16x8{h[0][t], h[1][t], h[2][t], h[3][t], h[4][t], h[5][t], h[6][t], h[7][t]};
                    op[((1 * 2 + b) * 4 + t) * 64 + ln] = gs_bf16x8{l[0][t], l[1][t], l[2][t], l[3][t], l[4][t], l[5][t], l[6][t], l[7][t]};
                }
            }
        }
        __syncthreads();
        if (slab + stride < n_slabs) fetch(slab + stride);
        if (worker) {
            gs_bf16x8 bh[4], bl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bh[u] = sm_v[((0 * 2 + dt) * 4 + u) * 64 + lane];
                bl[u] = sm_v[((1 * 2 + dt) * 4 + u) * 64 + lane];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const gs_bf16x8 ah = sm_k[((0 * 2 + mt) * 4 + t) * 64 + lane];
                const gs_bf16x8 al = sm_k[((1 * 2 + mt) * 4 + t) * 64 + lane];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[u], acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[u], acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[u], acc[t][u], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    float* rec = ws + static_cast<int64_t>(blockIdx.x) * ws_stride;
    if (worker) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int m = m0 + 64 * mt + 16 * lg + 4 * reg + t, d0 = 64 * dt + 4 * l15;
                if (m < M && d0 < D)
                    *reinterpret_cast<f32x4*>(&rec[static_cast<int64_t>(m) * D + d0]) =
                        f32x4{acc[t][0][reg], acc[t][1][reg], acc[t][2][reg], acc[t][3][reg]};
            }
    }
    if (stager) *reinterpret_cast<f32x4*>(&sm_sx[role][slg][4 * cq]) = sx;
    if (role == 0) {                                                        // [0]: q squares, [1]: k squares
        sm_sq[1][rt] = stager ? sq : 0.f;
        sm_sq[0][rt] = stager ? (with_q ? sqq : sq) : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Ml; c += 256) rec[static_cast<int64_t>(M) * D + m0 + c] = ((sm_sx[0][0][c] + sm_sx[0][1][c]) + sm_sx[0][2][c]) + sm_sx[0][3][c];
    if (blockIdx.y == 0)
        for (int c = threadIdx.x; c < D; c += 256) rec[static_cast<int64_t>(M) * D + M + c] = ((sm_sx[1][0][c] + sm_sx[1][1][c]) + sm_sx[1][2][c]) + sm_sx[1][3][c];
    // the Frobenius pairs (one per tile of the record): this workgroup owns the pairs of ITS tiles, puts its sums of squares (in
    // thread order) into the first and zeros into the others -- the finalize adds all pairs
    const int pair0 = 2 * blockIdx.y * DT, pairs = MTl * DT;
    if (threadIdx.x < 2) {
        float a = 0.f;
        for (int i = 0; i < 128; ++i) a += sm_sq[threadIdx.x][i];
        rec[t_main + 2 * pair0 + threadIdx.x] = a;
    }
    for (int i = 2 + threadIdx.x; i < 2 * pairs; i += 256) rec[t_main + 2 * pair0 + i] = 0.f;
    (void)MT;
}
}  // namespace

static int gram_slab_chunks(int64_t n_rows, int C) {
    // partial records of C^2 floats each: ~35 MB in all (96 at C = 300; 256 -- every CU -- from C = 185 down)
    const int64_t slabs = (n_rows + kSlabRows - 1) / kSlabRows;
    int64_t p = (int64_t(35) << 20) / (static_cast<int64_t>(C) * C * 4);
    if (p < 64) p = 64;
    if (p > 256) p = 256;
    return static_cast<int>(slabs < p ? slabs : p);
}

// Gram record of the closed form at the scripts' widths: record = [X^T X (C x C) | sum x (C) | unused (C) | 2 unused];
// of X^T X only the 64 x 64 blocks on and above the diagonal are written.  Layout and workspace as
// dif_simple_reduce_f32(x, x, x) with H = 1, M = D = C.
extern "C" int dif_gram_sym_f32(const float* x, int64_t ldx, int64_t n_rows, int C, float* record, void* workspace,
                                size_t workspace_bytes, dif_stream_t stream) {
    if (int rc = check_shape(n_rows, 1, C, C)) return rc;
    DIF_REQUIRE(x && record && workspace, DIF_E_BADARG, "dif_gram_sym_f32: null pointer");
    DIF_REQUIRE(ldx >= C, DIF_E_BADARG, "dif_gram_sym_f32: leading dimension smaller than a row");
    DIF_REQUIRE(workspace_bytes >= dif_simple_workspace_bytes(n_rows, 1, C, C) && dif::aligned16(workspace), DIF_E_WORKSPACE,
                "dif_gram_sym_f32: workspace too small or not 16-byte aligned");
    const Shape sh = make_shape(1, C, C);
    const int tiles_sym = sh.MT * (sh.MT + 1) / 2;
    const int P = reduce_chunks(n_rows, sh.tiles);          // as dif_simple_workspace_bytes sizes it
    const int64_t rec = (static_cast<int64_t>(sh.t_main) + 2 * sh.tiles + 3) & ~int64_t(3);
    const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && dif::aligned16(x);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* ws = static_cast<float*>(workspace);
    // 65..320 columns, enough rows to fill the chip, a workspace sized by dif_gram_sym_workspace_bytes: the one-read slab kernel
    const int Ps = gram_slab_chunks(n_rows, C);
    if (vec && C > 64 && sh.MT <= kSlabMaxMT && n_rows >= 4096 && !dif::exact_fp32() &&
        workspace_bytes >= static_cast<size_t>(rec) * sizeof(float) * static_cast<size_t>(Ps)) {
        hipLaunchKernelGGL(gram_slab_kernel, dim3(Ps, (tiles_sym + 7) / 8), dim3(512), 0, st, x, ldx, n_rows, C, sh.MT, ws, rec);
        if (int rc = dif::launch_status("gram_slab_kernel")) return rc;
        return dif::launch_record_finalize(ws, Ps, rec, sh.t_main, tiles_sym, record, st);
    }
    dim3 grid(P, tiles_sym), block(256);
    if (vec) hipLaunchKernelGGL((simple_reduce_kernel<true, float, true>), grid, block, 0, st, x, ldx, x, ldx, x, ldx, n_rows, sh, ws, rec);
    else hipLaunchKernelGGL((simple_reduce_kernel<false, float, true>), grid, block, 0, st, x, ldx, x, ldx, x, ldx, n_rows, sh, ws, rec);
    if (int rc = dif::launch_status("simple_reduce_kernel<sym>")) return rc;
    return dif::launch_record_finalize(ws, P, rec, sh.t_main, tiles_sym, record, st);
}

extern "C" size_t dif_gram_sym_workspace_bytes(int64_t n_rows, int C) {
    if (n_rows <= 0 || C <= 0) return 0;
    const Shape sh = make_shape(1, C, C);
    const size_t rec = (static_cast<size_t>(sh.t_main) + 2 * static_cast<size_t>(sh.tiles) + 3) & ~size_t(3);
    const size_t a = rec * sizeof(float) * static_cast<size_t>(reduce_chunks(n_rows, sh.tiles));
    const size_t b = rec * sizeof(float) * static_cast<size_t>(gram_slab_chunks(n_rows, C));
    return a > b ? a : b;
}

extern "C" size_t dif_simple_reduced_len(int H, int M, int D) {
    if (H <= 0 || M <= 0 || D <= 0) return 0;
    return static_cast<size_t>(make_shape(H, M, D).t_main) + 2;
}

extern "C" size_t dif_simple_workspace_bytes(int64_t n_rows, int H, int M, int D) {
    if (n_rows <= 0 || H <= 0 || M <= 0 || D <= 0) return 0;
    const Shape sh = make_shape(H, M, D);
    const size_t rec = (static_cast<size_t>(sh.t_main) + 2 * static_cast<size_t>(sh.tiles) + 3) & ~size_t(3);
    return rec * sizeof(float) * static_cast<size_t>(reduce_chunks(n_rows, sh.tiles));
}

extern "C" int dif_simple_reduce_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                                     int64_t ldv, int64_t n_rows, int H, int M, int D, float* reduced, void* workspace,
                                     size_t workspace_bytes, dif_stream_t stream) {
    return simple_reduce_entry<float>(q, ldq, k, ldk, v, ldv, n_rows, H, M, D, reduced, workspace, workspace_bytes, stream);
}

extern "C" int dif_simple_reduce_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                      int64_t n_rows, int H, int M, int D, float* reduced, void* workspace,
                                      size_t workspace_bytes, dif_stream_t stream) {
    using B = dif::bf16;
    return simple_reduce_entry<B>(static_cast<const B*>(q), ldq, static_cast<const B*>(k), ldk, static_cast<const B*>(v),
                                  ldv, n_rows, H, M, D, reduced, workspace, workspace_bytes, stream);
}

extern "C" int dif_simple_apply_f32(const float* q, int64_t ldq, const float* reduced, int64_t n_rows, int64_t n_global,
                                    int H, int M, int D, float* out, int64_t ldo, dif_stream_t stream) {
    return simple_apply_entry<float>(q, ldq, reduced, n_rows, n_global, H, M, D, out, ldo, stream);
}

extern "C" int dif_simple_apply_bf16(const void* q, int64_t ldq, const float* reduced, int64_t n_rows, int64_t n_global,
                                     int H, int M, int D, void* out, int64_t ldo, dif_stream_t stream) {
    using B = dif::bf16;
    return simple_apply_entry<B>(static_cast<const B*>(q), ldq, reduced, n_rows, n_global, H, M, D, static_cast<B*>(out),
                                 ldo, stream);
}

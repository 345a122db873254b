// a2: full_attention_conv(..., kernel='sigmoid')  -- node classification/difformer.py:45-56
//
//   out[n,h,:] = sum_l sigma(q_n.k_l) v_l / sum_l sigma(q_n.k_l)
// The reference materialises the [N,L,H] score tensor three times (:47,:52,:55); here it never
// leaves registers.  sigma is bounded so no running-max rescale is needed: one pass over the
// keys accumulates the row sum and sigma(S).V together.  Both contractions are exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32), computed TRANSPOSED so the score tile feeds the second MFMA with no
// register shuffle:
//   S^T[key][query] = K Q^T      : A[i=lane%16 <-> key][k] = K frag,  B[k][j=lane%16 <-> query] = Q frag
//   lane then holds S^T[key = 4*(lane/16)+reg][query = lane%16]  -> P = sigma(S^T)
//   O^T[d][query]  += V^T P^T    : k-step `reg` contracts key = 4*(lane/16)+reg, which is exactly the
//                                  register the lane already holds (B operand = P[reg]).
// Compute-bound on the f32 matrix pipe: 4*N*L*H*D FLOP.
#include <type_traits>
#include "dif_common.h"
#include "sigmoid_wide.h"

namespace {

using dif::f32x4;

constexpr int kWaves = 8;     // waves per workgroup; each takes every 8th 16-key tile of the workgroup's key range
constexpr int kQT = 2;        // 16-query tiles per wave: every K / V fragment fetched from L2 feeds 2x the MFMAs
constexpr int kQGroup = 16 * kQT;   // queries per workgroup
constexpr int kDTile = 64;    // output columns per workgroup (grid.y covers heads x ceil(D/64))

// Branch-free fragment load: out-of-range rows / columns are read from a clamped (valid) address and
// zeroed afterwards, so the compiler can issue all of a tile's loads back to back (a guarded load is
// its own exec-masked branch region and serialises).
template <bool VEC, typename T>
__device__ __forceinline__ f32x4 ld4(const T* __restrict__ base, int64_t ld, int64_t rc, bool rok,
                                     int col0, int c, int width) {
    f32x4 z;
    if (VEC) {
        const bool cok = c < width;                       // width % 4 == 0 here
        z = dif::Elem<T>::ld4(base + rc * ld + col0 + (cok ? c : 0));
        if (!(rok && cok)) z = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool cok = c + i < width;
            const float t = dif::Elem<T>::ld(base + rc * ld + col0 + (cok ? c + i : 0));
            z[i] = (rok && cok) ? t : 0.f;
        }
    }
    return z;
}

// The two halves of ld4 for software-pipelined loads: the raw (clamped-address) load is issued a step ahead and NOT touched until
// the step that uses it -- masking at load time would make the issuing step wait for the data.
template <bool VEC, typename T>
__device__ __forceinline__ f32x4 ld4_raw(const T* __restrict__ base, int64_t ld, int64_t rc, int col0, int c, int width) {
    f32x4 z;
    if (VEC) {
        z = dif::Elem<T>::ld4(base + rc * ld + col0 + (c < width ? c : 0));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = dif::Elem<T>::ld(base + rc * ld + col0 + (c + i < width ? c + i : 0));
    }
    return z;
}
template <bool VEC>
__device__ __forceinline__ f32x4 mask4(f32x4 z, bool rok, int c, int width) {
    if (VEC) {
        if (!(rok && c < width)) z = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (!(rok && c + i < width)) z[i] = 0.f;
    }
    return z;
}

// sigma(x) with the hardware exp2 / rcp (each ~1 ulp): far inside the 1e-4 parity budget, and ~5x
// fewer VALU instructions than expf + IEEE division next to the MFMAs.
__device__ __forceinline__ float sigmoidf(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// grid: (ceil(N/32), H * ceil(D/64), S key splits); block 512.
// S == 1: writes the normalised output.  S > 1 (small N: not enough query groups to fill 256 CUs): writes
// un-normalised partial sums and row sums to `part` [S][N][H*D] / `pden` [S][N][H*DT]; sigmoid_combine_kernel
// finishes.
//
// SEG (f4, physical particle/difformer-v2.py:113-135): blockIdx.z is a POSITION p inside the graphs of a batch; the
// attention runs among the seg_cnt[p] nodes that sit at position p of their graph.  Graphs are ranked by size
// (descending), so those are exactly ranks 0 .. seg_cnt[p]-1 and local row r is node seg_first[r] + p
// (seg_first[r] = first node of the r-th largest graph).  The n_graphs - seg_cnt[p] shorter graphs are the
// reference's zero padding: sigma(0) = 0.5 each in the denominator (+1e-9), nothing in the numerator.
// SPLIT (round 5; float32 storage, M <= 64): both contractions on split-bfloat16 operands -- every operand v = hi + lo, three
// v_mfma_f32_16x16x32_bf16 per 32-deep step (lo.hi + hi.lo + hi.hi; the dropped lo.lo term is 2^-16 of a product) instead of
// eight v_mfma_f32_16x16x4_f32 at twice the cycles: 5.3x less matrix-pipe time for a kernel that sat at 31-62 % of the fp32 MFMA
// peak.  The wave takes TWO 16-key tiles per step: the second contraction runs 32 keys deep, and its B operand is still the
// lane's own registers -- k-slot 8 lg + s <-> key 4 lg + s of the first tile (s < 4) / of the second (s >= 4), the same map on
// the V side.  Scores ~4e-6 |q||k|, sigma is 1/4-Lipschitz.  DIFFORMER_EXACT_FP32=1 keeps the fp32 chain.
typedef __bf16 sg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sg_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sg_split8(const f32x4& a, const f32x4& b, sg_bf16x8& hi, sg_bf16x8& lo) {
    const sg_bf16x4 h0 = __builtin_convertvector(a, sg_bf16x4), h1 = __builtin_convertvector(b, sg_bf16x4);
    const sg_bf16x4 l0 = __builtin_convertvector(a - __builtin_convertvector(h0, f32x4), sg_bf16x4);
    const sg_bf16x4 l1 = __builtin_convertvector(b - __builtin_convertvector(h1, f32x4), sg_bf16x4);
    hi = sg_bf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    lo = sg_bf16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
}

template <bool VEC, bool QREG, bool SEG, typename T = float, bool SPLIT = false>
__global__ __launch_bounds__(512) void sigmoid_attn_kernel(const T* __restrict__ q, int64_t ldq,
                                                           const T* __restrict__ k, int64_t ldk,
                                                           const T* __restrict__ v, int64_t ldv,
                                                           int64_t N, int64_t L, int H, int M, int D,
                                                           T* __restrict__ out, int64_t ldo,
                                                           float* __restrict__ part, float* __restrict__ pden,
                                                           float* __restrict__ den_out,
                                                           const int32_t* __restrict__ seg_first,
                                                           const int32_t* __restrict__ seg_cnt, int n_graphs) {
    __shared__ __attribute__((aligned(16))) float sm_o[kWaves][kQGroup * kDTile];   // 64 KiB
    __shared__ float sm_den[kWaves][kQGroup];

    const int DT = (D + kDTile - 1) / kDTile;
    const int h = blockIdx.y / DT;
    const int dt = blockIdx.y % DT;
    const int S = SEG ? 1 : gridDim.z;
    const int split = SEG ? 0 : blockIdx.z;
    const int pos = SEG ? blockIdx.z : 0;
    if (SEG) {
        N = L = seg_cnt[pos];
        if (static_cast<int64_t>(blockIdx.x) * kQGroup >= N) return;       // uniform: before any barrier
    }
    // local row -> row in memory (clamped to the last valid row; callers mask with `r < n`)
    auto qrow = [&](int64_t r) -> int64_t {
        const int64_t rc = r < N ? r : N - 1;
        return SEG ? static_cast<int64_t>(seg_first[rc]) + pos : rc;
    };
    auto krow = [&](int64_t r) -> int64_t {
        const int64_t rc = r < L ? r : L - 1;
        return SEG ? static_cast<int64_t>(seg_first[rc]) + pos : rc;
    };
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int64_t q0 = static_cast<int64_t>(blockIdx.x) * kQGroup;
    const int m_chunks = (M + 63) / 64;

    // Q fragments for the (only) m-chunk stay in registers when M <= 64
    f32x4 qv[kQT][4];
    if (QREG) {
#pragma unroll
        for (int t = 0; t < kQT; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                qv[t][c] = ld4<VEC>(q, ldq, qrow(q0 + 16 * t + l15), q0 + 16 * t + l15 < N, h * M, 16 * c + 4 * lg, M);
    }

    f32x4 acc_o[kQT][4];
    float den[kQT];
#pragma unroll
    for (int t = 0; t < kQT; ++t) {
        den[t] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc_o[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // this workgroup's key tiles: [kt0, kt1)
    const int64_t n_ktiles = (L + 15) / 16;
    const int64_t per = (n_ktiles + S - 1) / S;
    const int64_t kt0 = split * per;
    const int64_t kt1 = (kt0 + per < n_ktiles) ? kt0 + per : n_ktiles;
    if constexpr (SPLIT) {
        static_assert(QREG, "split-bf16 sigmoid attention keeps the query fragments in registers (M <= 64)");
        sg_bf16x8 qh[kQT][2], ql[kQT][2];
#pragma unroll
        for (int t = 0; t < kQT; ++t)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) sg_split8(qv[t][2 * kb], qv[t][2 * kb + 1], qh[t][kb], ql[t][kb]);
        const int64_t key_limit = (kt1 * 16 < L) ? kt1 * 16 : L;         // the keys of THIS workgroup's range end here
        // The K and V fragments of the NEXT step are in flight under this step's products (64 VGPRs): a step is a chain
        // loads -> products -> sigma -> products, and one workgroup per CU leaves two waves per SIMD to hide it.  Raw loads
        // from clamped (valid) addresses, issued unconditionally -- the last step re-reads its own rows -- and masked only
        // when used: a mask or a branch at issue time makes the issuing step wait for the data (measured: 665 -> 1,493 us).
        // V: A[i = l15 <-> d][k-slot 8 lg + 4 tile + reg] = V[kbase + 16 tile + 4 lg + reg][16 dtl + l15].
        f32x4 kn[2][4], vn[4][2];
        auto prefetch = [&](int64_t kbase) {
#pragma unroll
            for (int tile = 0; tile < 2; ++tile) {
                const int64_t kr = krow(kbase + 16 * tile + l15);
#pragma unroll
                for (int c = 0; c < 4; ++c) kn[tile][c] = ld4_raw<VEC>(k, ldk, kr, h * M, 16 * c + 4 * lg, M);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const T* vrow = v + krow(kbase + 16 * tile + 4 * lg + reg) * ldv + h * D;
#pragma unroll
                    for (int dtl = 0; dtl < 4; ++dtl) {
                        const int d = dt * kDTile + 16 * dtl + l15;
                        vn[dtl][tile][reg] = dif::Elem<T>::ld(vrow + (d < D ? d : 0));
                    }
                }
            }
        };
        prefetch((kt0 + 2 * wave) * 16);
        for (int64_t kt = kt0 + 2 * wave; kt < kt1; kt += 2 * kWaves) {
            const int64_t kbase = kt * 16;
            f32x4 kc[2][4];
            sg_bf16x8 vh[4], vl[4];
#pragma unroll
            for (int tile = 0; tile < 2; ++tile)
#pragma unroll
                for (int c = 0; c < 4; ++c) kc[tile][c] = mask4<VEC>(kn[tile][c], kbase + 16 * tile + l15 < key_limit, 16 * c + 4 * lg, M);
#pragma unroll
            for (int dtl = 0; dtl < 4; ++dtl) {
                const bool dok = dt * kDTile + 16 * dtl + l15 < D;
                f32x4 va[2];
#pragma unroll
                for (int tile = 0; tile < 2; ++tile)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        va[tile][reg] = (dok && kbase + 16 * tile + 4 * lg + reg < key_limit) ? vn[dtl][tile][reg] : 0.f;
                sg_split8(va[0], va[1], vh[dtl], vl[dtl]);
            }
            {
                const int64_t nxt = kt + 2 * kWaves;
                prefetch((nxt < kt1 ? nxt : kt) * 16);
                __builtin_amdgcn_sched_barrier(0);       // issued HERE, ahead of the products (the scheduler sank the V reads to the end of the step)
            }
            f32x4 s2[2][kQT];
#pragma unroll
            for (int tile = 0; tile < 2; ++tile) {
                const f32x4 (&kx)[4] = kc[tile];
#pragma unroll
                for (int t = 0; t < kQT; ++t) s2[tile][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    sg_bf16x8 kh, kl;
                    sg_split8(kx[2 * kb], kx[2 * kb + 1], kh, kl);
#pragma unroll
                    for (int t = 0; t < kQT; ++t) {
                        s2[tile][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[t][kb], s2[tile][t], 0, 0, 0);      // small terms first
                        s2[tile][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[t][kb], s2[tile][t], 0, 0, 0);
                        s2[tile][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[t][kb], s2[tile][t], 0, 0, 0);
                    }
                }
            }
            // P = sigma(S) masked beyond the range (difformer.py:47), its row sums (:50-51), split for the second contraction
            sg_bf16x8 ph[kQT], pl[kQT];
#pragma unroll
            for (int t = 0; t < kQT; ++t) {
                f32x4 pp[2];
#pragma unroll
                for (int tile = 0; tile < 2; ++tile)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        pp[tile][reg] = (kbase + 16 * tile + 4 * lg + reg < key_limit) ? sigmoidf(s2[tile][t][reg]) : 0.f;
                        den[t] += pp[tile][reg];
                    }
                sg_split8(pp[0], pp[1], ph[t], pl[t]);
            }
#pragma unroll
            for (int dtl = 0; dtl < 4; ++dtl)
#pragma unroll
                for (int t = 0; t < kQT; ++t) {
                    acc_o[t][dtl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl[dtl], ph[t], acc_o[t][dtl], 0, 0, 0);
                    acc_o[t][dtl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh[dtl], pl[t], acc_o[t][dtl], 0, 0, 0);
                    acc_o[t][dtl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh[dtl], ph[t], acc_o[t][dtl], 0, 0, 0);
                }
        }
    } else
    for (int64_t kt = kt0 + wave; kt < kt1; kt += kWaves) {
        const int64_t kbase = kt * 16;
        // ---- S^T tiles (one per query tile) ------------------------------------------------
        f32x4 s[kQT];
#pragma unroll
        for (int t = 0; t < kQT; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int mc = 0; mc < m_chunks; ++mc) {
            f32x4 kx[4];
            const int64_t kr = krow(kbase + l15);
#pragma unroll
            for (int c = 0; c < 4; ++c) kx[c] = ld4<VEC>(k, ldk, kr, kbase + l15 < L, h * M, mc * 64 + 16 * c + 4 * lg, M);
            if (!QREG) {
#pragma unroll
                for (int t = 0; t < kQT; ++t)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        qv[t][c] = ld4<VEC>(q, ldq, qrow(q0 + 16 * t + l15), q0 + 16 * t + l15 < N, h * M,
                                            mc * 64 + 16 * c + 4 * lg, M);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < kQT; ++t)      // independent accumulator chains back to back
                        s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kx[c][u], qv[t][c][u], s[t], 0, 0, 0);
        }
        // ---- V fragments: A[i=l15 <-> d][k=lg] = V[kbase + 4*lg + reg][d0 + 16*dtl + l15] ----
        float vf[4][4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t key = kbase + 4 * lg + reg;
            const bool kok = key < L;
            const T* vrow = v + krow(key) * ldv + h * D;
#pragma unroll
            for (int dtl = 0; dtl < 4; ++dtl) {
                const int d = dt * kDTile + 16 * dtl + l15;
                const bool dok = d < D;
                const float t = dif::Elem<T>::ld(vrow + (dok ? d : 0));
                vf[dtl][reg] = (kok && dok) ? t : 0.f;
            }
        }
        // ---- P = sigma(S), masked beyond L (difformer.py:47) -----------------------------
        f32x4 p[kQT];
#pragma unroll
        for (int t = 0; t < kQT; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                p[t][reg] = (kbase + 4 * lg + reg < L) ? sigmoidf(s[t][reg]) : 0.f;
                den[t] += p[t][reg];                                      // :50-51 row sum
            }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
            for (int dtl = 0; dtl < 4; ++dtl)
#pragma unroll
                for (int t = 0; t < kQT; ++t)
                    acc_o[t][dtl] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[dtl][reg], p[t][reg], acc_o[t][dtl], 0, 0, 0);
    }

    // rows (queries) are shared by the 4 lane groups: fold, then fold the 8 waves through LDS
#pragma unroll
    for (int t = 0; t < kQT; ++t) {
        float dsum = den[t];
        dsum += __shfl_xor(dsum, 16, 64);
        dsum += __shfl_xor(dsum, 32, 64);
        if (lg == 0) sm_den[wave][16 * t + l15] = dsum;
        // lane holds O^T[d = 16*dtl + 4*lg + reg][query = 16t + l15] -> store as [query][d]
#pragma unroll
        for (int dtl = 0; dtl < 4; ++dtl)
            *reinterpret_cast<f32x4*>(&sm_o[wave][(16 * t + l15) * kDTile + 16 * dtl + 4 * lg]) = acc_o[t][dtl];
    }
    __syncthreads();

    for (int e = threadIdx.x; e < kQGroup * kDTile; e += 512) {
        const int qi = e / kDTile, dl = e % kDTile;
        float o = 0.f, dn = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { o += sm_o[w][e]; dn += sm_den[w][qi]; }
        const int64_t row = q0 + qi;
        const int d = dt * kDTile + dl;
        if (row < N && d < D) {
            if (SEG) {
                // difformer-v2.py:127-134: padded graphs add sigma(0) each, then + epsilon
                const float dfull = dn + 0.5f * static_cast<float>(n_graphs - N) + 1e-9f;
                dif::Elem<T>::st(out + qrow(row) * ldo + h * D + d, o / dfull);
                if (den_out && d == 0) den_out[qrow(row) * H + h] = dfull;          // kept for the backward pass
            } else if (S == 1) {
                dif::Elem<T>::st(out + row * ldo + h * D + d, o / dn);     // :55-56
                if (den_out && d == 0) den_out[row * H + h] = dn;          // kept for the backward pass
            } else {
                part[(static_cast<int64_t>(split) * N + row) * (H * D) + h * D + d] = o;
                if (dl == 0) pden[(static_cast<int64_t>(split) * N + row) * (H * DT) + h * DT + dt] = dn;
            }
        }
    }
}

// S > 1: out = (sum_s part[s]) / (sum_s pden[s])
template <typename T>
__global__ __launch_bounds__(256) void sigmoid_combine_kernel(const float* __restrict__ part,
                                                              const float* __restrict__ pden, int64_t N, int H,
                                                              int D, int S, T* __restrict__ out, int64_t ldo,
                                                              float* __restrict__ den_out) {
    const int DT = (D + kDTile - 1) / kDTile;
    const int64_t total = N * H * D;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * 256) {
        const int64_t row = e / (H * D);
        const int c = static_cast<int>(e % (H * D));
        const int h = c / D, d = c % D;
        float o = 0.f, dn = 0.f;
        for (int s = 0; s < S; ++s) {
            o += part[(static_cast<int64_t>(s) * N + row) * (H * D) + c];
            dn += pden[(static_cast<int64_t>(s) * N + row) * (H * DT) + h * DT + d / kDTile];
        }
        dif::Elem<T>::st(out + row * ldo + c, o / dn);
        if (den_out && d == 0) den_out[row * H + h] = dn;
    }
}

// Key splits S: every (query group, head, column tile) is cut into S workgroups over disjoint key ranges (+ a combine
// pass).  Two workgroups fit a CU, so the launch runs in ceil(groups * S / 512) rounds of workgroups that each sweep
// n_ktiles / (8 S) key tiles per wave plus a fixed cost worth ~4 tiles (LDS fold of the 8 waves, partial stores).
// S minimises rounds x per-workgroup time: it fills the chip when there are few query groups (Cora: 85 groups -> S = 6)
// and trims the half-empty last round when there are many (N = 20,000: 625 groups, S = 1 runs 2 rounds for 1.22
// rounds of work; S = 4 runs 5 quarter-rounds).
// The split-bfloat16 kernel (float32 storage, M <= 64) holds hi / lo pairs of every operand: 162 VGPRs, ONE workgroup per CU,
// and a wave step is two key tiles -- its splits are priced on 256 slots and keep >= two key tiles per wave.  (Parking the
// query fragments in LDS brings it to 124 VGPRs and two workgroups per CU: measured slower, 743 -> 800 us at N = 20,000.)
int key_splits(int64_t N, int64_t L, int H, int D, bool split_kernel = false) {
    const int64_t groups = ((N + kQGroup - 1) / kQGroup) * H * ((D + kDTile - 1) / kDTile);
    const int64_t n_ktiles = (L + 15) / 16;
    const int64_t slots = (split_kernel ? 1 : 2) * dif::kCUs;
    int64_t smax = split_kernel ? (n_ktiles + 2 * kWaves - 1) / (2 * kWaves) : (n_ktiles + kWaves - 1) / kWaves;   // keep >= one step per wave
    if (smax > 16) smax = 16;
    int best = 1;
    double best_cost = -1.0;
    for (int64_t s = 1; s <= smax; ++s) {
        const int64_t rounds = (groups * s + slots - 1) / slots;
        const double cost = static_cast<double>(rounds) * (static_cast<double>(n_ktiles) / (kWaves * s) + 4.0);
        if (best_cost < 0 || cost < best_cost * 0.98) { best_cost = cost; best = static_cast<int>(s); }   // ties -> fewer splits
    }
    return best;
}

}  // namespace

extern "C" size_t dif_sigmoid_workspace_bytes(int64_t N, int64_t L, int H, int M, int D) {

    if (N <= 0 || L <= 0 || H <= 0 || D <= 0) return 0;
    if (dif::sigw_covers(M, D)) {                                    // heads of 65 .. 512 columns: csrc/sigmoid_wide.hip (packed planes + partial sums)
        const size_t wide = dif::sigw_fwd_workspace_bytes(N, L, H, M, D);
        const int S0 = key_splits(N, L, H, D);                       // ... or this file's kernel under dif_set_exact_fp32(1) / bfloat16 storage
        const size_t DT0 = (D + kDTile - 1) / kDTile;
        const size_t narrow = S0 == 1 ? 0 : static_cast<size_t>(S0) * N * H * (static_cast<size_t>(D) + DT0) * sizeof(float);
        return wide > narrow ? wide : narrow;
    }
    int S = key_splits(N, L, H, D);
    if (M <= 64) {                                                   // whichever kernel the exact-fp32 switch picks at launch
        const int S2 = key_splits(N, L, H, D, true);
        if (S2 > S) S = S2;
    }
    const size_t DT = (D + kDTile - 1) / kDTile;
    size_t need = S == 1 ? 0 : static_cast<size_t>(S) * N * H * (static_cast<size_t>(D) + DT) * sizeof(float);
    if (dif::sigw_narrow_pays(M, D, N, L, false)) {                  // 33 .. 64 columns with enough pairs: the plane kernels may run
        const size_t planes = dif::sigw_fwd_workspace_bytes(N, L, H, M, D);
        if (planes > need) need = planes;
    }
    return need;
}

namespace {

template <typename T>
int sigmoid_attn(const char* who, const T* q, int64_t ldq, const T* k, int64_t ldk, const T* v, int64_t ldv, int64_t N,
                 int64_t L, int H, int M, int D, T* out, int64_t ldo, float* den, void* workspace, size_t workspace_bytes,
                 dif_stream_t stream) {
    DIF_REQUIRE(N > 0 && L > 0 && H > 0 && M > 0 && D > 0, DIF_E_BADARG, "%s: N, L, H, M, D must be positive", who);
    DIF_REQUIRE(q && k && v && out, DIF_E_BADARG, "%s: null pointer", who);
    DIF_REQUIRE(ldq >= H * M && ldk >= H * M && ldv >= H * D && ldo >= H * D, DIF_E_BADARG,
                "%s: leading dimension smaller than a row", who);
    const int64_t gx = (N + kQGroup - 1) / kQGroup;
    const int64_t gy = static_cast<int64_t>(H) * ((D + kDTile - 1) / kDTile);
    DIF_REQUIRE(gx < (1ll << 31) && gy <= 65535, DIF_E_RANGE, "%s: grid too large", who);
    // Inference only (no row sums asked for): under loss.backward() the few 1e-6 the split operands move out / den come back
    // amplified in gradients that are sums of cancelling rows (Wk.bias of model/a_nobn_src: 2.0e-5 of itself against 4.3e-6
    // with the fp32 chain, the reference's own float32 run 3.7e-6; scripts/exp_sigmoid_grad_parity.py) -- training keeps fp32.
    // (aligned rows only: the scalar-load variant of the split kernel does not fit its hi / lo pairs and the prefetch in 256 VGPRs)
    if constexpr (std::is_same<T, float>::value) {
        // heads of 65 .. 512 columns (image and text/run.sh:17,35,54: hidden 300 / 400): every operand as split-bfloat16 planes, packed
        // in MFMA fragment order, scores formed once per (query, key) pair for all D columns (csrc/sigmoid_wide.hip)
        // ... and heads of 33 .. 64 columns in inference from the sizes where those kernels win (sigw_narrow_pays: never with row
        // sums asked for -- training keeps this file's fp32 chain)
        if ((dif::sigw_covers(M, D) || dif::sigw_narrow_pays(M, D, N, L, den != nullptr)) && !dif::exact_fp32())
            return dif::sigw_fwd(q, ldq, k, ldk, v, ldv, N, L, H, M, D, out, ldo, den, workspace, workspace_bytes,
                                 static_cast<hipStream_t>(stream));
    }
    const bool vec = (M % 4 == 0) && (ldq % 4 == 0) && (ldk % 4 == 0) && dif::aligned_v4<T>(q) && dif::aligned_v4<T>(k);
    const bool split_kernel = std::is_same<T, float>::value && M <= 64 && vec && !dif::exact_fp32() && den == nullptr;
    const int S = key_splits(N, L, H, D, split_kernel);
    const size_t need = dif_sigmoid_workspace_bytes(N, L, H, M, D);
    DIF_REQUIRE(S == 1 || (workspace && workspace_bytes >= need), DIF_E_WORKSPACE, "%s: workspace too small (%zu < %zu)", who,
                workspace_bytes, need);
    const bool qreg = (M <= 64);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(workspace);
    float* pden = (S > 1) ? part + static_cast<size_t>(S) * N * H * D : nullptr;
    dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(gy), S), block(512);
#define DIF_LAUNCH_SIG(V, Q) \
    hipLaunchKernelGGL((sigmoid_attn_kernel<V, Q, false, T>), grid, block, 0, st, q, ldq, k, ldk, v, ldv, N, L, H, M, D, out, \
                       ldo, part, pden, den, nullptr, nullptr, 0)
    if constexpr (std::is_same<T, float>::value) {
        if (split_kernel) {                        // both contractions on split-bfloat16 operands (sigmoid_attn_kernel<..., SPLIT>)
            hipLaunchKernelGGL((sigmoid_attn_kernel<true, true, false, T, true>), grid, block, 0, st, q, ldq, k, ldk, v, ldv, N, L, H, M, D,
                               out, ldo, part, pden, den, nullptr, nullptr, 0);
            if (int rc = dif::launch_status("sigmoid_attn_kernel<split>")) return rc;
            if (S > 1) {
                int64_t g = (N * H * D + 255) / 256;
                if (g > 8 * dif::kCUs) g = 8 * dif::kCUs;
                hipLaunchKernelGGL(sigmoid_combine_kernel<T>, dim3(static_cast<unsigned>(g)), dim3(256), 0, st, part, pden, N, H, D, S,
                                   out, ldo, den);
                return dif::launch_status("sigmoid_combine_kernel");
            }
            return 0;
        }
    }
    if (vec && qreg) DIF_LAUNCH_SIG(true, true);
    else if (vec) DIF_LAUNCH_SIG(true, false);
    else if (qreg) DIF_LAUNCH_SIG(false, true);
    else DIF_LAUNCH_SIG(false, false);
#undef DIF_LAUNCH_SIG
    if (int rc = dif::launch_status("sigmoid_attn_kernel")) return rc;
    if (S > 1) {
        int64_t g = (N * H * D + 255) / 256;
        if (g > 8 * dif::kCUs) g = 8 * dif::kCUs;
        hipLaunchKernelGGL(sigmoid_combine_kernel<T>, dim3(static_cast<unsigned>(g)), dim3(256), 0, st, part, pden, N, H, D, S,
                           out, ldo, den);
        return dif::launch_status("sigmoid_combine_kernel");
    }
    return 0;
}

}  // namespace

extern "C" int dif_sigmoid_attn_f32(const float* q, int64_t ldq, const float* k, int64_t ldk,
                                    const float* v, int64_t ldv, int64_t N, int64_t L, int H, int M, int D,
                                    float* out, int64_t ldo, void* workspace, size_t workspace_bytes,
                                    dif_stream_t stream) {
    return sigmoid_attn<float>("dif_sigmoid_attn_f32", q, ldq, k, ldk, v, ldv, N, L, H, M, D, out, ldo, nullptr, workspace,
                               workspace_bytes, stream);
}

// training forward: also leaves den[n,h] = sum_l sigmoid(q_n . k_l) (float [N,H]) for dif_sigmoid_attn_bwd_f32
extern "C" int dif_sigmoid_attn_fwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                        int64_t N, int64_t L, int H, int M, int D, float* out, int64_t ldo, float* den,
                                        void* workspace, size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(den != nullptr, DIF_E_BADARG, "dif_sigmoid_attn_fwd_f32: den is null");
    return sigmoid_attn<float>("dif_sigmoid_attn_fwd_f32", q, ldq, k, ldk, v, ldv, N, L, H, M, D, out, ldo, den, workspace,
                               workspace_bytes, stream);
}

// bfloat16 storage (q, k, v, out), fp32 scores / sigma / accumulation -- SURVEY.md 8b lists the {f32, bf16} pair
extern "C" int dif_sigmoid_attn_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                     int64_t N, int64_t L, int H, int M, int D, void* out, int64_t ldo, void* workspace,
                                     size_t workspace_bytes, dif_stream_t stream) {
    using B = dif::bf16;
    return sigmoid_attn<B>("dif_sigmoid_attn_bf16", static_cast<const B*>(q), ldq, static_cast<const B*>(k), ldk,
                           static_cast<const B*>(v), ldv, N, L, H, M, D, static_cast<B*>(out), ldo, nullptr, workspace,
                           workspace_bytes, stream);
}

// f4: TransConv.full_attention(kernel='sigmoid') over a batch of graphs -- physical particle/difformer-v2.py:113-135.
// den (optional, float [N, H]): the full denominators (row sum + 0.5 per padded graph + 1e-9) for dif_batched_sigmoid_attn_bwd_f32.
static int batched_sigmoid_entry(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                                 int64_t ldv, const int32_t* ranked_first, const int32_t* pos_count,
                                 int n_graphs, int max_nodes, int H, int M, int D, float* out, int64_t ldo, float* den,
                                 dif_stream_t stream) {
    DIF_REQUIRE(n_graphs > 0 && max_nodes > 0 && H > 0 && M > 0 && D > 0, DIF_E_BADARG,
                "dif_batched_sigmoid_attn_f32: n_graphs, max_nodes, H, M, D must be positive");
    DIF_REQUIRE(q && k && v && out && ranked_first && pos_count, DIF_E_BADARG, "dif_batched_sigmoid_attn_f32: null pointer");
    DIF_REQUIRE(ldq >= H * M && ldk >= H * M && ldv >= H * D && ldo >= H * D, DIF_E_BADARG,
                "dif_batched_sigmoid_attn_f32: leading dimension smaller than a row");
    const int64_t gx = (static_cast<int64_t>(n_graphs) + kQGroup - 1) / kQGroup;
    const int64_t gy = static_cast<int64_t>(H) * ((D + kDTile - 1) / kDTile);
    DIF_REQUIRE(gy <= 65535 && max_nodes <= 65535, DIF_E_RANGE,
                "dif_batched_sigmoid_attn_f32: grid too large (heads x column tiles %lld, max_nodes %d; limit 65535)",
                static_cast<long long>(gy), max_nodes);
    const bool vec = (M % 4 == 0) && (ldq % 4 == 0) && (ldk % 4 == 0) && dif::aligned16(q) && dif::aligned16(k);
    const bool qreg = (M <= 64);
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(gy), static_cast<unsigned>(max_nodes)), block(512);
#define DIF_LAUNCH_SEG(V, Q) \
    hipLaunchKernelGGL((sigmoid_attn_kernel<V, Q, true>), grid, block, 0, st, q, ldq, k, ldk, v, ldv, int64_t{0}, \
                       int64_t{0}, H, M, D, out, ldo, nullptr, nullptr, den, ranked_first, pos_count, n_graphs)
    if (vec && qreg) DIF_LAUNCH_SEG(true, true);
    else if (vec) DIF_LAUNCH_SEG(true, false);
    else if (qreg) DIF_LAUNCH_SEG(false, true);
    else DIF_LAUNCH_SEG(false, false);
#undef DIF_LAUNCH_SEG
    return dif::launch_status("sigmoid_attn_kernel<batched>");
}

extern "C" int dif_batched_sigmoid_attn_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                                            int64_t ldv, const int32_t* ranked_first, const int32_t* pos_count,
                                            int n_graphs, int max_nodes, int H, int M, int D, float* out, int64_t ldo,
                                            dif_stream_t stream) {
    return batched_sigmoid_entry(q, ldq, k, ldk, v, ldv, ranked_first, pos_count, n_graphs, max_nodes, H, M, D, out, ldo, nullptr,
                                 stream);
}

extern "C" int dif_batched_sigmoid_attn_fwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v,
                                                int64_t ldv, const int32_t* ranked_first, const int32_t* pos_count,
                                                int n_graphs, int max_nodes, int H, int M, int D, float* out, int64_t ldo,
                                                float* den, dif_stream_t stream) {
    DIF_REQUIRE(den != nullptr, DIF_E_BADARG, "dif_batched_sigmoid_attn_fwd_f32: den is null");
    DIF_REQUIRE(D <= 64, DIF_E_SHAPE, "dif_batched_sigmoid_attn_fwd_f32: the backward pass covers D <= 64 (got %d)", D);
    return batched_sigmoid_entry(q, ldq, k, ldk, v, ldv, ranked_first, pos_count, n_graphs, max_nodes, H, M, D, out, ldo, den,
                                 stream);
}

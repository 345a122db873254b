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
#include "dif_common.h"

namespace {

using dif::f32x4;

constexpr int kWaves = 8;     // waves per workgroup; each takes every 8th 16-key tile
constexpr int kQTile = 16;    // queries per workgroup
constexpr int kDTile = 64;    // output columns per workgroup (grid.z covers D > 64)

template <bool VEC>
__device__ __forceinline__ f32x4 ld4(const float* __restrict__ base, int64_t ld, int64_t r, int64_t n,
                                     int col0, int c, int width) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (r >= n) return z;
    const float* p = base + r * ld + col0 + c;
    if (VEC) {
        if (c < width) z = *reinterpret_cast<const f32x4*>(p);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (c + i < width) z[i] = p[i];
    }
    return z;
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// grid: (ceil(N/16), H, ceil(D/64)); block 512.
template <bool VEC, bool QREG>
__global__ __launch_bounds__(512) void sigmoid_attn_kernel(const float* __restrict__ q, int64_t ldq,
                                                           const float* __restrict__ k, int64_t ldk,
                                                           const float* __restrict__ v, int64_t ldv,
                                                           int64_t N, int64_t L, int M, int D,
                                                           float* __restrict__ out, int64_t ldo) {
    __shared__ __attribute__((aligned(16))) float sm_o[kWaves][kQTile * kDTile];
    __shared__ float sm_den[kWaves][kQTile];

    const int h = blockIdx.y;
    const int dt = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int64_t qrow = static_cast<int64_t>(blockIdx.x) * kQTile + l15;
    const int m_chunks = (M + 63) / 64;

    // Q fragments for the (only) m-chunk stay in registers when M <= 64
    f32x4 qv[4];
    if (QREG) {
#pragma unroll
        for (int c = 0; c < 4; ++c) qv[c] = ld4<VEC>(q, ldq, qrow, N, h * M, 16 * c + 4 * lg, M);
    }

    f32x4 acc_o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc_o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float den = 0.f;

    const int64_t n_ktiles = (L + 15) / 16;
    for (int64_t kt = wave; kt < n_ktiles; kt += kWaves) {
        const int64_t kbase = kt * 16;
        // ---- S^T tile -------------------------------------------------------------------
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int mc = 0; mc < m_chunks; ++mc) {
            f32x4 kx[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) kx[c] = ld4<VEC>(k, ldk, kbase + l15, L, h * M, mc * 64 + 16 * c + 4 * lg, M);
            if (!QREG) {
#pragma unroll
                for (int c = 0; c < 4; ++c) qv[c] = ld4<VEC>(q, ldq, qrow, N, h * M, mc * 64 + 16 * c + 4 * lg, M);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    s = __builtin_amdgcn_mfma_f32_16x16x4f32(kx[c][t], qv[c][t], s, 0, 0, 0);
        }
        // ---- V fragments: A[i=l15 <-> d][k=lg] = V[kbase + 4*lg + reg][d0 + 16*dtl + l15] ----
        float vf[4][4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t key = kbase + 4 * lg + reg;
#pragma unroll
            for (int dtl = 0; dtl < 4; ++dtl) {
                const int d = dt * kDTile + 16 * dtl + l15;
                vf[dtl][reg] = (key < L && d < D) ? v[key * ldv + h * D + d] : 0.f;
            }
        }
        // ---- P = sigma(S), masked beyond L (difformer.py:47) -----------------------------
        f32x4 p;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            p[reg] = (kbase + 4 * lg + reg < L) ? sigmoidf(s[reg]) : 0.f;
            den += p[reg];                                            // :50-51 row sum
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
            for (int dtl = 0; dtl < 4; ++dtl)
                acc_o[dtl] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[dtl][reg], p[reg], acc_o[dtl], 0, 0, 0);
    }

    // rows (queries) are shared by the 4 lane groups: fold, then fold the 8 waves through LDS
    den += __shfl_xor(den, 16, 64);
    den += __shfl_xor(den, 32, 64);
    if (lg == 0) sm_den[wave][l15] = den;
    // lane holds O^T[d = 16*dtl + 4*lg + reg][query = l15] -> store as [query][d]
#pragma unroll
    for (int dtl = 0; dtl < 4; ++dtl)
        *reinterpret_cast<f32x4*>(&sm_o[wave][l15 * kDTile + 16 * dtl + 4 * lg]) = acc_o[dtl];
    __syncthreads();

    for (int e = threadIdx.x; e < kQTile * kDTile; e += 512) {
        const int qi = e / kDTile, dl = e % kDTile;
        float o = 0.f, dn = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { o += sm_o[w][e]; dn += sm_den[w][qi]; }
        const int64_t row = static_cast<int64_t>(blockIdx.x) * kQTile + qi;
        const int d = dt * kDTile + dl;
        if (row < N && d < D) out[row * ldo + h * D + d] = o / dn;   // :55-56
    }
}

}  // namespace

extern "C" size_t dif_sigmoid_workspace_bytes(int64_t, int64_t, int, int, int) { return 0; }

extern "C" int dif_sigmoid_attn_f32(const float* q, int64_t ldq, const float* k, int64_t ldk,
                                    const float* v, int64_t ldv, int64_t N, int64_t L, int H, int M, int D,
                                    float* out, int64_t ldo, void* /*workspace*/, size_t /*workspace_bytes*/,
                                    dif_stream_t stream) {
    DIF_REQUIRE(N > 0 && L > 0 && H > 0 && M > 0 && D > 0, DIF_E_BADARG,
                "dif_sigmoid_attn_f32: N, L, H, M, D must be positive");
    DIF_REQUIRE(q && k && v && out, DIF_E_BADARG, "dif_sigmoid_attn_f32: null pointer");
    DIF_REQUIRE(ldq >= H * M && ldk >= H * M && ldv >= H * D && ldo >= H * D, DIF_E_BADARG,
                "dif_sigmoid_attn_f32: leading dimension smaller than a row");
    const int64_t gx = (N + kQTile - 1) / kQTile;
    const int gz = (D + kDTile - 1) / kDTile;
    DIF_REQUIRE(gx < (1ll << 31) && H <= 65535 && gz <= 65535, DIF_E_RANGE, "dif_sigmoid_attn_f32: grid too large");
    const bool vec = (M % 4 == 0) && (ldq % 4 == 0) && (ldk % 4 == 0) && dif::aligned16(q) && dif::aligned16(k);
    const bool qreg = (M <= 64);
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(static_cast<unsigned>(gx), H, gz), block(512);
#define DIF_LAUNCH_SIG(V, Q) \
    hipLaunchKernelGGL((sigmoid_attn_kernel<V, Q>), grid, block, 0, st, q, ldq, k, ldk, v, ldv, N, L, M, D, out, ldo)
    if (vec && qreg) DIF_LAUNCH_SIG(true, true);
    else if (vec) DIF_LAUNCH_SIG(true, false);
    else if (qreg) DIF_LAUNCH_SIG(false, true);
    else DIF_LAUNCH_SIG(false, false);
#undef DIF_LAUNCH_SIG
    return dif::launch_status("sigmoid_attn_kernel");
}

// a4 + a1 fused: the Wq/Wk/Wv projections of DIFFormerConv.forward
// (node classification/difformer.py:115-118) together with stage 1 of the simple kernel (:20-34).
//
//   q = x Wq^T + bq,  k = x Wk^T + bk,  v = x Wv^T + bv          (per head h: rows h*D .. h*D+D-1 of W)
//   record += { K_h^T V_h, sum k, sum v, sum q*q, sum k*k }        (same record as dif_simple_reduce_f32)
// One pass over x: k never reaches HBM, q and v are written once (q for stage 2, v for the SpMM
// gather, contiguous 4*H*D-byte rows).  The reference runs three Linear layers and then ~10 ATen
// launches over q,k,v for the same quantities.
//
// MFMA plan (v_mfma_f32_16x16x4_f32, exact fp32), per wave and 16-row tile:
//   projection  D[i <-> row][j <-> feature] = sum_c X[row][c] W[feature][c]
//       A[i=lane%16][k=lane/16] = X[r0 + lane%16][16cq + 4*(lane/16) + t]     (one dwordx4 load per cq)
//       B[k=lane/16][j=lane%16] = W[16ft + lane%16][16cq + 4*(lane/16) + t]   (ds_read_b128 from LDS)
//     -> lane holds tile[ft][reg] = Y[r0 + 4*(lane/16) + reg][16ft + lane%16]
//   K^T V       D[i <-> m][j <-> d] += sum_rows K[row][m] V[row][d], contraction step kk uses
//       row = r0 + 4*(lane/16) + kk, so  A = ktile[ft_k][kk]  and  B = vtile[ft_v][kk]  are the registers
//       the projection just produced: no shuffle, no LDS round trip between the two products.
// Work: 256 MFMAs (8192 cycles) per 16 rows and head -> MFMA-bound (~30 us at C4) rather than HBM-bound
// (x read + q,v written = 3*N*D*4 bytes), still ~4x less time than GEMM + reduce as separate passes.
#include "dif_common.h"

namespace {

using dif::f32x4;
using dif::Elem;

constexpr int kPRWaves = 4;
constexpr int kWStride = 68;     // padded LDS row (floats): 16 lanes x b128 land on 64 distinct banks
constexpr int kPRMaxChunks = 512;

struct PRShape {
    int H, D, C;       // heads, per-head width (M == D), input channels (<= 64)
    int t_main;        // H*D*D + 2*H*D
};

// X fragment loads.  GUARD = false: full 16-row tile, C == 64, 16-byte aligned rows -> 4 plain dwordx4.
template <bool GUARD, typename T>
__device__ __forceinline__ void load_tile(f32x4 (&xa)[4], const T* __restrict__ x, int64_t ldx, int64_t r,
                                          int64_t n, int lg, int C, bool vec) {
    if (!GUARD) {
        const T* p = x + r * ldx + 4 * lg;
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) xa[cq] = Elem<T>::ld4(p + 16 * cq);
    } else {
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const int c = 16 * cq + 4 * lg;
            if (r < n) {
                const T* p = x + r * ldx + c;
                if (vec && c + 3 < C) {
                    z = Elem<T>::ld4(p);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (c + i < C) z[i] = Elem<T>::ld(p + i);
                }
            }
            xa[cq] = z;
        }
    }
}

struct TileState {
    f32x4 acc[4][4];   // KtV[16fk + 4lg + reg][16fv + l15]
    float ks[4], vs[4];
    float ksq, qsq;
};

// One 16-row tile: three projections, the q / v stores, K^T V and the column sums.
template <bool GUARD, typename T>
__device__ __forceinline__ void tile_body(TileState& st, const f32x4 (&xa)[4], const float (*sm_w)[64 * kWStride],
                                          const float (*sm_b)[64], int64_t r0, int64_t n_rows, int h, int D,
                                          int l15, int lg, T* __restrict__ q_out, int64_t ldq,
                                          T* __restrict__ v_out, int64_t ldv) {
    f32x4 kt[4], vt[4];
#pragma unroll
    for (int m = 0; m < 3; ++m) {            // 0: k, 1: v, 2: q
        f32x4 y[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const float bias = sm_b[m][16 * ft + l15];
            y[ft] = f32x4{bias, bias, bias, bias};
        }
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
            f32x4 wf[4];
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
                wf[ft] = *reinterpret_cast<const f32x4*>(&sm_w[m][(16 * ft + l15) * kWStride + 16 * cq + 4 * lg]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ft = 0; ft < 4; ++ft)      // four independent accumulator chains back to back
                    y[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[cq][t], wf[ft][t], y[ft], 0, 0, 0);
        }
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            f32x4 yy = y[ft];
            if (GUARD) {                          // rows past the end would carry the bias
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (r0 + 4 * lg + reg >= n_rows) yy[reg] = 0.f;
            }
            const int f = 16 * ft + l15;
            if (m == 0) {
                kt[ft] = yy;
                st.ks[ft] += (yy[0] + yy[1]) + (yy[2] + yy[3]);
                st.ksq += yy[0] * yy[0] + yy[1] * yy[1] + yy[2] * yy[2] + yy[3] * yy[3];
            } else {
                if (m == 1) {
                    vt[ft] = yy;
                    st.vs[ft] += (yy[0] + yy[1]) + (yy[2] + yy[3]);
                } else {
                    st.qsq += yy[0] * yy[0] + yy[1] * yy[1] + yy[2] * yy[2] + yy[3] * yy[3];
                }
                T* o = (m == 1 ? v_out + (r0 + 4 * lg) * ldv : q_out + (r0 + 4 * lg) * ldq) + h * D + f;
                const int64_t ld = (m == 1) ? ldv : ldq;
                if (!GUARD) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) Elem<T>::st(o + reg * ld, yy[reg]);
                } else if (f < D) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        if (r0 + 4 * lg + reg < n_rows) Elem<T>::st(o + reg * ld, yy[reg]);
                }
            }
        }
    }
    // K^T V on the tiles just produced: step kk contracts rows r0 + 4*lg + kk
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int fk = 0; fk < 4; ++fk)
#pragma unroll
            for (int fv = 0; fv < 4; ++fv)
                st.acc[fk][fv] = __builtin_amdgcn_mfma_f32_16x16x4f32(kt[fk][kk], vt[fv][kk], st.acc[fk][fv], 0, 0, 0);
}

// grid (row chunks P, heads H); 256 threads.  Record layout identical to simple_attn.hip's.
// EXACT: C == 64, D == 64, aligned rows -> full tiles run without any guard.
template <bool EXACT, typename T>
__global__ __launch_bounds__(256, 2) void project_reduce_kernel(
    const T* __restrict__ x, int64_t ldx, int64_t n_rows, PRShape sh, const T* __restrict__ Wq,
    const T* __restrict__ bq, const T* __restrict__ Wk, const T* __restrict__ bk,
    const T* __restrict__ Wv, const T* __restrict__ bv, T* __restrict__ q_out, int64_t ldq,
    T* __restrict__ v_out, int64_t ldv, float* __restrict__ ws, int64_t ws_stride, int vec, int wvec) {
    __shared__ __attribute__((aligned(16))) float sm_w[3][64 * kWStride];   // Wk, Wv, Wq of this head (zero padded)
    __shared__ float sm_b[3][64];
    __shared__ __attribute__((aligned(16))) float sm_tile[64 * 64];
    __shared__ float sm_k[kPRWaves][64];
    __shared__ float sm_v[kPRWaves][64];
    __shared__ float sm_s[kPRWaves][2];

    const int h = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int D = sh.D, C = sh.C;

    // stage this head's three weight blocks [D x C] -> LDS [64 x kWStride], zero padded
    {
        const T* Ws[3] = {Wk + static_cast<int64_t>(h) * D * C, Wv + static_cast<int64_t>(h) * D * C,
                          Wq + static_cast<int64_t>(h) * D * C};
        const T* bs[3] = {bk + h * D, bv + h * D, bq + h * D};
        if (EXACT && wvec) {
            // 64 x 64 blocks, 4-element aligned: all twelve 4-wide loads of a thread are issued before the first LDS
            // store (the element-wise loop below runs 48 load -> store round trips back to back: ~10 us of prologue)
            f32x4 wreg[3][4];
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) wreg[m][i] = Elem<T>::ld4(Ws[m] + 4 * (threadIdx.x + 256 * i));
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 4 * (threadIdx.x + 256 * i);
                    *reinterpret_cast<f32x4*>(&sm_w[m][(e >> 6) * kWStride + (e & 63)]) = wreg[m][i];
                }
        } else {
#pragma unroll
            for (int m = 0; m < 3; ++m)
                for (int e = threadIdx.x; e < 64 * 64; e += 256) {
                    const int f = e >> 6, c = e & 63;
                    sm_w[m][f * kWStride + c] = (f < D && c < C) ? Elem<T>::ld(Ws[m] + f * C + c) : 0.f;
                }
        }
#pragma unroll
        for (int m = 0; m < 3; ++m)
            if (threadIdx.x < 64) sm_b[m][threadIdx.x] = (threadIdx.x < D) ? Elem<T>::ld(bs[m] + threadIdx.x) : 0.f;
    }
    __syncthreads();

    TileState st;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) st.acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) { st.ks[i] = 0.f; st.vs[i] = 0.f; }
    st.ksq = 0.f; st.qsq = 0.f;

    const int64_t n_tiles = (n_rows + 15) / 16;
    const int64_t n_fast = EXACT ? n_rows / 16 : 0;       // tiles that need no guard at all
    const int64_t first = static_cast<int64_t>(blockIdx.x) * kPRWaves + wave;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kPRWaves;
    int64_t tile = first;
    if (EXACT) {
        for (; tile < n_fast; tile += stride) {
            // keep the weight fragments in LDS (re-read per tile): hoisting all 3x64 of them into registers
            // would leave one wave per SIMD and nothing to hide latency behind
            asm volatile("" ::: "memory");
            f32x4 xa[4];
            load_tile<false>(xa, x, ldx, tile * 16 + l15, n_rows, lg, C, true);
            tile_body<false>(st, xa, sm_w, sm_b, tile * 16, n_rows, h, D, l15, lg, q_out, ldq, v_out, ldv);
        }
    }
    for (; tile < n_tiles; tile += stride) {
        asm volatile("" ::: "memory");
        f32x4 xa[4];
        load_tile<true>(xa, x, ldx, tile * 16 + l15, n_rows, lg, C, vec != 0);
        tile_body<true>(st, xa, sm_w, sm_b, tile * 16, n_rows, h, D, l15, lg, q_out, ldq, v_out, ldv);
    }
    f32x4 (&acc)[4][4] = st.acc;
    float (&ks)[4] = st.ks;
    float (&vs)[4] = st.vs;
    float ksq = st.ksq, qsq = st.qsq;

    // ---- fold the 4 waves (fixed order) ------------------------------------------------------
    // every wave parks its 64 x 64 tile in LDS at once -- waves 0..2 in the weight region (3 x 64 x kWStride floats, no
    // longer needed), wave 3 in sm_tile -- and after ONE barrier the record write below adds the four copies in the
    // order ((w0 + w1) + w2) + w3
    __syncthreads();                                  // all waves are done reading the weights
    {
        float* mine = (wave < 3) ? &sm_w[wave][0] : sm_tile;
#pragma unroll
        for (int fk = 0; fk < 4; ++fk)
#pragma unroll
            for (int fv = 0; fv < 4; ++fv)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    mine[(16 * fk + 4 * lg + reg) * 64 + 16 * fv + l15] = acc[fk][fv][reg];
    }
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
        float a = ks[ft], b = vs[ft];
        a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
        if (lg == 0) { sm_k[wave][16 * ft + l15] = a; sm_v[wave][16 * ft + l15] = b; }
    }
    qsq = dif::wave_sum(qsq);
    ksq = dif::wave_sum(ksq);
    if (lane == 0) { sm_s[wave][0] = qsq; sm_s[wave][1] = ksq; }
    __syncthreads();

    float* rec = ws + static_cast<int64_t>(blockIdx.x) * ws_stride;
    float* rec_ktv = rec + static_cast<int64_t>(h) * D * D;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int m = e >> 6, d = e & 63;
        if (m < D && d < D) rec_ktv[m * D + d] = ((sm_w[0][e] + sm_w[1][e]) + sm_w[2][e]) + sm_tile[e];
    }
    if (threadIdx.x < 64 && threadIdx.x < D) {
        const int c = threadIdx.x;
        rec[sh.H * D * D + h * D + c] = ((sm_k[0][c] + sm_k[1][c]) + sm_k[2][c]) + sm_k[3][c];
        rec[sh.H * D * D + sh.H * D + h * D + c] = ((sm_v[0][c] + sm_v[1][c]) + sm_v[2][c]) + sm_v[3][c];
    }
    if (threadIdx.x == 0) {
        rec[sh.t_main + 2 * h + 0] = ((sm_s[0][0] + sm_s[1][0]) + sm_s[2][0]) + sm_s[3][0];
        rec[sh.t_main + 2 * h + 1] = ((sm_s[0][1] + sm_s[1][1]) + sm_s[2][1]) + sm_s[3][1];
    }
}

int pr_chunks(int64_t n_rows) {
    const int64_t tiles = (n_rows + 15) / 16;
    int64_t p = (tiles + kPRWaves - 1) / kPRWaves;     // at least one tile per wave
    const int64_t cap = 2 * dif::kCUs;                  // ~70 KB of LDS per workgroup: two fit a CU
    if (p > cap) p = cap;
    if (p > kPRMaxChunks) p = kPRMaxChunks;
    if (p < 1) p = 1;
    return static_cast<int>(p);
}

template <typename T>
int project_reduce_entry(const T* x, int64_t ldx, int64_t n_rows, int C_in, const T* Wq, const T* bq, const T* Wk,
                         const T* bk, const T* Wv, const T* bv, int H, int D, T* q_out, int64_t ldq, T* v_out,
                         int64_t ldv, float* reduced, void* workspace, size_t workspace_bytes, dif_stream_t stream) {
    DIF_REQUIRE(n_rows > 0 && C_in > 0 && H > 0 && D > 0, DIF_E_BADARG,
                "dif_project_reduce: n_rows, C_in, H, D must be positive");
    DIF_REQUIRE(C_in <= 64 && D <= 64, DIF_E_SHAPE,
                "dif_project_reduce: fused path covers C_in <= 64 and D <= 64 (got %d, %d); use the Linear layers + "
                "dif_simple_reduce", C_in, D);
    DIF_REQUIRE(H <= 65535, DIF_E_RANGE, "dif_project_reduce: too many heads");
    DIF_REQUIRE(x && Wq && bq && Wk && bk && Wv && bv && q_out && v_out && reduced && workspace, DIF_E_BADARG,
                "dif_project_reduce: null pointer");
    DIF_REQUIRE(ldx >= C_in && ldq >= H * D && ldv >= H * D, DIF_E_BADARG,
                "dif_project_reduce: leading dimension smaller than a row");
    DIF_REQUIRE(workspace_bytes >= dif_project_reduce_workspace_bytes(n_rows, H, D), DIF_E_WORKSPACE,
                "dif_project_reduce: workspace too small");
    PRShape sh;
    sh.H = H; sh.D = D; sh.C = C_in;
    sh.t_main = H * D * D + 2 * H * D;
    const int P = pr_chunks(n_rows);
    const int64_t rec = (static_cast<int64_t>(sh.t_main) + 2 * H + 3) & ~int64_t(3);
    const int vec = (C_in % 4 == 0) && (ldx % 4 == 0) && dif::aligned_v4<T>(x);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* ws = static_cast<float*>(workspace);
    const bool exact = vec && C_in == 64 && D == 64;
    const int wvec = dif::aligned_v4<T>(Wq) && dif::aligned_v4<T>(Wk) && dif::aligned_v4<T>(Wv);   // H*D*C is a multiple of 4096 here
    if (exact)
        hipLaunchKernelGGL((project_reduce_kernel<true, T>), dim3(P, H), dim3(256), 0, st, x, ldx, n_rows, sh, Wq, bq, Wk,
                           bk, Wv, bv, q_out, ldq, v_out, ldv, ws, rec, vec, wvec);
    else
        hipLaunchKernelGGL((project_reduce_kernel<false, T>), dim3(P, H), dim3(256), 0, st, x, ldx, n_rows, sh, Wq, bq,
                           Wk, bk, Wv, bv, q_out, ldq, v_out, ldv, ws, rec, vec, wvec);
    if (int rc = dif::launch_status("project_reduce_kernel")) return rc;
    return dif::launch_record_finalize(ws, P, rec, sh.t_main, H, reduced, st);
}

}  // namespace

extern "C" size_t dif_project_reduce_workspace_bytes(int64_t n_rows, int H, int D) {
    if (n_rows <= 0 || H <= 0 || D <= 0) return 0;
    const size_t rec = (static_cast<size_t>(H) * D * D + 2 * static_cast<size_t>(H) * D + 2 * H + 3) & ~size_t(3);
    return rec * sizeof(float) * static_cast<size_t>(pr_chunks(n_rows));
}

extern "C" int dif_project_reduce_f32(const float* x, int64_t ldx, int64_t n_rows, int C_in, const float* Wq,
                                      const float* bq, const float* Wk, const float* bk, const float* Wv,
                                      const float* bv, int H, int D, float* q_out, int64_t ldq, float* v_out,
                                      int64_t ldv, float* reduced, void* workspace, size_t workspace_bytes,
                                      dif_stream_t stream) {
    return project_reduce_entry<float>(x, ldx, n_rows, C_in, Wq, bq, Wk, bk, Wv, bv, H, D, q_out, ldq, v_out, ldv, reduced,
                                       workspace, workspace_bytes, stream);
}

extern "C" int dif_project_reduce_bf16(const void* x, int64_t ldx, int64_t n_rows, int C_in, const void* Wq,
                                       const void* bq, const void* Wk, const void* bk, const void* Wv, const void* bv,
                                       int H, int D, void* q_out, int64_t ldq, void* v_out, int64_t ldv, float* reduced,
                                       void* workspace, size_t workspace_bytes, dif_stream_t stream) {
    using B = dif::bf16;
    auto c = [](const void* p) { return static_cast<const B*>(p); };
    return project_reduce_entry<B>(c(x), ldx, n_rows, C_in, c(Wq), c(bq), c(Wk), c(bk), c(Wv), c(bv), H, D,
                                   static_cast<B*>(q_out), ldq, static_cast<B*>(v_out), ldv, reduced, workspace,
                                   workspace_bytes, stream);
}

// a2 / f3 for WIDE heads: full_attention_conv(..., kernel='sigmoid') with 65 .. 512 columns per head, forward and backward
//   -- node classification/difformer.py:45-56 as the image-and-text scripts run it (image and text/run.sh:17,35,54:
//      --kernel sigmoid --hidden_channels 300 / 400, N = 13,000 .. 18,846, 600 epochs x 5 runs, main.py:94-110).
//
//     S = Q K^T,  P = sigma(S),  den_n = sum_l P_nl,  out = (P / den) V                                     (forward)
//     G~ = G / den,  delta~_n = G~_n . out_n,  dS = (G~ V^T - delta~) P (1 - P),
//     dQ = dS K,  dK = dS^T Q,  dV = P^T G~                                                                 (backward)
//
// At these widths the path is bound by the matrix pipe (4 N L D FLOP forward, 14 N L D backward against 4 N D floats of
// traffic), and the fp32 matrix core runs at 1/16 of the bf16 rate.  So every operand goes in as bfloat16 PLANES,
// x = hi + lo (NP = 2: three bf16 MFMAs per product, lo.hi + hi.lo + hi.hi, ~4e-6 |a||b|; NP = 3 adds a third plane:
// six MFMAs, ~2^-24), accumulation in fp32.  Nothing of size N x L is stored: sigma is recomputed tile by tile.
//
// PACKED OPERANDS.  A pre-pass (sigw_pack_kernel, one read of each tensor) splits the planes and writes every operand in the
// order the matrix core consumes it -- MFMA fragments of 64 lanes x 8 bf16 = 1 KiB, contiguous -- so that the streamed side is
// moved global -> LDS by LDS-DMA in 1-KiB wave instructions and read back conflict-free (`ds_read_b128`, lane l at byte 16 l):
//   row-packed  Xr[h][t][plane][rt][ks][lane][8] = X[32 t + 16 rt + l15][32 ks + 8 lg + j]      (contraction over COLUMNS)
//   col-packed  Xc[h][t][plane][ct]    [lane][8] = X[32 t + slot(lg, j)][16 ct + l15]           (contraction over ROWS)
// with l15 = lane % 16, lg = lane / 16, slot(lg, j) = 16 (j / 4) + 4 lg + j % 4.  A tile t = 32 rows = NP x 2 KS KiB
// (KS = ceil(max(M, D) / 32); columns and rows are zero-padded).  `slot` is the order in which a lane holds the rows of two
// 16 x 16 score tiles in MFMA D-layout (row 4 lg + reg of tile rt): sigma(S^T) goes from the first contraction's result registers
// straight into the second contraction's B operand, no shuffle, no LDS (the trick of csrc/sigmoid_attn.hip).
//
// TWO SWEEP KERNELS, each wave owning 16 rows of the STATIONARY side (its fragments in registers for the whole sweep) with
// the streamed side passing through a ring of two LDS tile buffers, one tile per sub-stage, one barrier per sub-stage:
//   sigw_fwd_kernel:  per step  [Y tile -> S^T = Y X^T]  [Z tile -> O^T += Z^T sigma(S^T)]
//        forward:  X = Q, Y = K, Z = V (normalised by den)          dV:  X = K, Y = Q, Z = G~ (raw)
//   sigw_bwd_kernel:  per step  [Y1 tile -> S^T]  [Y2 tile -> T^T = Y2 X2^T, dS]  [Y1 col tile -> A^T += Y1^T dS]
//        dQ:  X1 = Q, X2 = G~, Y1 = K, Y2 = V, delta~ per stationary row      dK:  X1 = K, X2 = V, Y1 = Q, Y2 = G~, delta~ per streamed row
// The stream is cut into S splits when there are fewer than 256 workgroups' worth of stationary rows (15,000 rows = 118
// workgroups of 8 waves); partial sums are added in split order (bitwise reproducible).
#include <type_traits>
#include "dif_common.h"
#include "sigmoid_wide.h"

namespace {

using dif::f32x4;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxKS = 16;                      // 512 columns

__device__ __forceinline__ float sigmoidf(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// product terms (streamed plane, stationary plane), small terms first
template <int NP> struct Terms;
template <> struct Terms<2> {
    static constexpr int n = 3;
    static __device__ __forceinline__ constexpr int a(int i) { return i == 0 ? 1 : 0; }       // lo.hi  hi.lo  hi.hi
    static __device__ __forceinline__ constexpr int b(int i) { return i == 1 ? 1 : 0; }
};
template <> struct Terms<3> {
    static constexpr int n = 6;                                                                // mid.mid  hi.lo  lo.hi  hi.mid  mid.hi  hi.hi
    static __device__ __forceinline__ constexpr int a(int i) { return i == 0 ? 1 : i == 2 ? 2 : i == 4 ? 1 : 0; }
    static __device__ __forceinline__ constexpr int b(int i) { return i == 0 ? 1 : i == 1 ? 2 : i == 3 ? 1 : 0; }
};

// x -> NP bf16 planes (round to nearest even; plane p + 1 holds what plane p left)
template <int NP>
__device__ __forceinline__ void split_planes(f32x4 x0, f32x4 x1, bf16x8 (&pl)[NP]) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const bf16x4 h0 = __builtin_convertvector(x0, bf16x4), h1 = __builtin_convertvector(x1, bf16x4);
        pl[p] = bf16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        if (p + 1 < NP) {
            x0 -= __builtin_convertvector(h0, f32x4);
            x1 -= __builtin_convertvector(h1, f32x4);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// pack: one workgroup per (32-row tile, head).  x [R][ld] (head h at column h * creal), optional per-row divisor
// rowdiv [R][H] (G~ = G / den).  Writes the row-packed and / or the col-packed planes of the tile.
template <int NP>
__global__ __launch_bounds__(256) void sigw_pack_kernel(const float* __restrict__ x, int64_t ld, int64_t R, int creal, int H,
                                                        int KS, const float* __restrict__ rowdiv, bf16x8* __restrict__ rowp,
                                                        bf16x8* __restrict__ colp, int T) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const int C = 32 * KS, lds = C + 4;
    const int t = blockIdx.x, h = blockIdx.y;
    for (int e = threadIdx.x; e < 32 * C; e += 256) {
        const int r = e / C, c = e - r * C;
        const int64_t row = 32ll * t + r;
        float v = 0.f;
        if (row < R && c < creal) {
            v = x[row * ld + static_cast<int64_t>(h) * creal + c];
            if (rowdiv) v /= rowdiv[row * H + h];
        }
        smf[r * lds + c] = v;
    }
    __syncthreads();
    const int FR = 2 * KS;
    const int64_t tile = (static_cast<int64_t>(h) * T + t) * NP * FR * 64;
    for (int item = threadIdx.x; item < FR * 64; item += 256) {
        const int f = item >> 6, lane = item & 63, l15 = lane & 15, lg = lane >> 4;
        if (rowp) {
            const int rt = f / KS, ks = f - rt * KS;
            const float* s = smf + (16 * rt + l15) * lds + 32 * ks + 8 * lg;
            bf16x8 pl[NP];
            split_planes<NP>(*reinterpret_cast<const f32x4*>(s), *reinterpret_cast<const f32x4*>(s + 4), pl);
#pragma unroll
            for (int p = 0; p < NP; ++p) rowp[tile + (static_cast<int64_t>(p) * FR + f) * 64 + lane] = pl[p];
        }
        if (colp) {
            const int col = 16 * f + l15;
            f32x4 v0, v1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v0[j] = smf[(4 * lg + j) * lds + col];
                v1[j] = smf[(16 + 4 * lg + j) * lds + col];
            }
            bf16x8 pl[NP];
            split_planes<NP>(v0, v1, pl);
#pragma unroll
            for (int p = 0; p < NP; ++p) colp[tile + (static_cast<int64_t>(p) * FR + f) * 64 + lane] = pl[p];
        }
    }
}

// delta~[h][n] = (g_n . out_n) / den_n for n < N, 0 up to NPAD (one wave per (n, h))
__global__ __launch_bounds__(256) void sigw_delta_kernel(const float* __restrict__ g, int64_t ldg, const float* __restrict__ out,
                                                         int64_t ldo, const float* __restrict__ den, int64_t N, int64_t NPAD, int H,
                                                         int D, float* __restrict__ delta) {
    const int lane = threadIdx.x & 63;
    const int64_t item = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (item >= NPAD * H) return;
    const int64_t n = item / H;
    const int h = static_cast<int>(item - n * H);
    float s = 0.f;
    if (n < N) {
        for (int d = lane; d < D; d += 64) s += g[n * ldg + h * D + d] * out[n * ldo + h * D + d];
        s = dif::wave_sum(s) / den[n * H + h];
    }
    if (lane == 0) delta[static_cast<int64_t>(h) * NPAD + n] = s;
}

// ------------------------------------------------------------------------------------------------------------------------
struct SweepOut {
    float* out; int64_t ldo; int cout;          // final rows [NX][ldo], head h at column h * cout (S == 1)
    float* den_out;                             // nullable [NX][H]
    float* part; float* pden;                   // S > 1: [S][H][NXPAD][C], [S][H][NXPAD]
    int64_t NXPAD;
    int normalize;
};

// stores the accumulators of a wave (O^T[col = 16 ct + 4 lg + r][row = l15]) -- final or partial
template <int KS>
__device__ __forceinline__ void store_acc(const f32x4 (&o)[2 * KS], float den_tot, const SweepOut& w, int64_t row, int64_t NX, int h,
                                          int H, int S, int split, int l15, int lg) {
    constexpr int C = 32 * KS;
    if (S == 1) {
        if (row < NX) {
            const float sc = w.normalize ? 1.0f / den_tot : 1.0f;
            float* dst = w.out + row * w.ldo + static_cast<int64_t>(h) * w.cout;
            const bool vec = (w.cout % 4 == 0) && (w.ldo % 4 == 0) && (reinterpret_cast<uintptr_t>(w.out) & 15u) == 0;
#pragma unroll
            for (int ct = 0; ct < 2 * KS; ++ct) {
                const int c = 16 * ct + 4 * lg;
                if (vec) {
                    if (c < w.cout) *reinterpret_cast<f32x4*>(dst + c) = o[ct] * sc;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c + r < w.cout) dst[c + r] = o[ct][r] * sc;
                }
            }
            if (w.den_out && lg == 0) w.den_out[row * H + h] = den_tot;
        }
    } else {
        const int64_t base = (static_cast<int64_t>(split) * H + h) * w.NXPAD + row;       // row < NXPAD by construction
        float* dst = w.part + base * C;
#pragma unroll
        for (int ct = 0; ct < 2 * KS; ++ct) *reinterpret_cast<f32x4*>(dst + 16 * ct + 4 * lg) = o[ct];
        if (w.pden && lg == 0) w.pden[base] = den_tot;
    }
}

struct FwdArgs {
    const bf16x8* xr; const bf16x8* yr; const bf16x8* zc;
    int64_t NX, NY;
    int Tx, Ty, H;
    SweepOut w;
};

// grid (ceil(NX / (16 W)), H, S); block 64 W; dynamic LDS 2 tiles
template <int KS, int NP, int W>
__global__ __launch_bounds__(64 * W) void sigw_fwd_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16x8 sm[];
    constexpr int FR = 2 * KS;
    constexpr int TILE = NP * FR * 64;
    using TT = Terms<NP>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
    const int h = blockIdx.y, S = gridDim.z, split = blockIdx.z;
    const int64_t g = static_cast<int64_t>(blockIdx.x) * W + wave;               // this wave's 16 stationary rows
    const int64_t gc = g < 2ll * a.Tx ? g : 2ll * a.Tx - 1;
    bf16x8 xf[NP][KS];
    {
        const bf16x8* xt = a.xr + ((static_cast<int64_t>(h) * a.Tx + (gc >> 1)) * NP * FR + (gc & 1) * KS) * 64 + lane;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[p][ks] = xt[(p * FR + ks) * 64];
    }
    const int per = (a.Ty + S - 1) / S;
    const int t0 = split * per, t1 = (t0 + per < a.Ty) ? t0 + per : a.Ty;
    const bf16x8* ybase = a.yr + static_cast<int64_t>(h) * a.Ty * TILE;
    const bf16x8* zbase = a.zc + static_cast<int64_t>(h) * a.Ty * TILE;
    auto issue = [&](const bf16x8* tile, int b) {                      // LDS-DMA: one KiB per wave instruction, no registers
#pragma unroll
        for (int i = 0; i < (NP * FR + W - 1) / W; ++i) {
            const int piece = wave + i * W;
            if ((NP * FR) % W == 0 || piece < NP * FR)
                __builtin_amdgcn_global_load_lds(tile + piece * 64 + lane, sm + b * TILE + piece * 64, 16, 0, 0);
        }
    };
    f32x4 o[FR];
#pragma unroll
    for (int ct = 0; ct < FR; ++ct) o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float den = 0.f;
    if (t0 < t1) issue(ybase + static_cast<int64_t>(t0) * TILE, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        issue(zbase + static_cast<int64_t>(t) * TILE, 1);
        // ---- S^T[streamed 4 lg + r of tile rt][stationary l15] ----
        f32x4 sa[TT::n][2];
#pragma unroll
        for (int i = 0; i < TT::n; ++i) sa[i][0] = sa[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bf16x8* b0 = sm + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 y[NP][2];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) y[p][rt] = b0[(p * FR + rt * KS + ks) * 64];
#pragma unroll
            for (int i = 0; i < TT::n; ++i)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    sa[i][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[TT::a(i)][rt], xf[TT::b(i)][ks], sa[i][rt], 0, 0, 0);
        }
        f32x4 p[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x4 s = sa[0][rt];
#pragma unroll
            for (int i = 1; i < TT::n; ++i) s += sa[i][rt];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = 32ll * t + 16 * rt + 4 * lg + r < a.NY;          // padded rows of the stream: sigma(0) = 1/2 stays out of den
                p[rt][r] = ok ? sigmoidf(s[r]) : 0.f;
                den += p[rt][r];
            }
        }
        bf16x8 pb[NP];
        split_planes<NP>(p[0], p[1], pb);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (t + 1 < t1) issue(ybase + static_cast<int64_t>(t + 1) * TILE, 0);
        // ---- O^T[col 16 ct + 4 lg + r][stationary l15] += Z^T P^T ----
        const bf16x8* b1 = sm + TILE + lane;
#pragma unroll
        for (int c0 = 0; c0 < FR; c0 += 4) {
            bf16x8 z[NP][4];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (c0 + i < FR) z[pl][i] = b1[(pl * FR + c0 + i) * 64];
#pragma unroll
            for (int tm = 0; tm < TT::n; ++tm)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (c0 + i < FR)
                        o[c0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(z[TT::a(tm)][i], pb[TT::b(tm)], o[c0 + i], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }
    const float den_tot = dif::rows4_sum(den);
    store_acc<KS>(o, den_tot, a.w, 16 * g + l15, a.NX, h, a.H, S, split, l15, lg);
}

struct BwdArgs {
    const bf16x8* x1r; const bf16x8* x2r;       // stationary, row-packed
    const bf16x8* y1r; const bf16x8* y2r;       // streamed, row-packed
    const bf16x8* y1c;                          // streamed, col-packed
    const float* delta; int64_t NDPAD;          // delta~ [H][NDPAD], zero beyond the rows
    int64_t NX, NY;
    int Tx, Ty, H;
    SweepOut w;
};

// DSTREAM: delta~ belongs to the STREAMED rows (dK: the stream is the queries), else to the stationary rows (dQ)
template <int KS, int NP, int W, bool DSTREAM>
__global__ __launch_bounds__(64 * W) void sigw_bwd_kernel(BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16x8 sm[];
    constexpr int FR = 2 * KS;
    constexpr int TILE = NP * FR * 64;
    using TT = Terms<NP>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
    const int h = blockIdx.y, S = gridDim.z, split = blockIdx.z;
    const int64_t g = static_cast<int64_t>(blockIdx.x) * W + wave;
    const int64_t gc = g < 2ll * a.Tx ? g : 2ll * a.Tx - 1;
    bf16x8 x1[NP][KS], x2[NP][KS];
    {
        const int64_t off = ((static_cast<int64_t>(h) * a.Tx + (gc >> 1)) * NP * FR + (gc & 1) * KS) * 64 + lane;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                x1[p][ks] = a.x1r[off + (p * FR + ks) * 64];
                x2[p][ks] = a.x2r[off + (p * FR + ks) * 64];
            }
    }
    const float* dl = a.delta + static_cast<int64_t>(h) * a.NDPAD;
    const float dstat = DSTREAM ? 0.f : dl[16 * gc + l15];
    const int per = (a.Ty + S - 1) / S;
    const int t0 = split * per, t1 = (t0 + per < a.Ty) ? t0 + per : a.Ty;
    const int64_t hb = static_cast<int64_t>(h) * a.Ty * TILE;
    auto issue = [&](const bf16x8* tile, int b) {                      // LDS-DMA: one KiB per wave instruction, no registers
#pragma unroll
        for (int i = 0; i < (NP * FR + W - 1) / W; ++i) {
            const int piece = wave + i * W;
            if ((NP * FR) % W == 0 || piece < NP * FR)
                __builtin_amdgcn_global_load_lds(tile + piece * 64 + lane, sm + b * TILE + piece * 64, 16, 0, 0);
        }
    };
    // S^T or T^T of the tile in buffer b against the stationary fragments xs
    auto scores = [&](int b, const bf16x8 (&xs)[NP][KS], f32x4 (&res)[2]) {
        f32x4 sa[TT::n][2];
#pragma unroll
        for (int i = 0; i < TT::n; ++i) sa[i][0] = sa[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bf16x8* bb = sm + b * TILE + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 y[NP][2];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) y[p][rt] = bb[(p * FR + rt * KS + ks) * 64];
#pragma unroll
            for (int i = 0; i < TT::n; ++i)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    sa[i][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[TT::a(i)][rt], xs[TT::b(i)][ks], sa[i][rt], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            res[rt] = sa[0][rt];
#pragma unroll
            for (int i = 1; i < TT::n; ++i) res[rt] += sa[i][rt];
        }
    };
    f32x4 o[FR];
#pragma unroll
    for (int ct = 0; ct < FR; ++ct) o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    int q = 0;                                                          // sub-stage counter: tile q sits in buffer q & 1
    if (t0 < t1) issue(a.y1r + hb + static_cast<int64_t>(t0) * TILE, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        f32x4 dv[2];
        if (DSTREAM) {
            dv[0] = *reinterpret_cast<const f32x4*>(dl + 32ll * t + 4 * lg);
            dv[1] = *reinterpret_cast<const f32x4*>(dl + 32ll * t + 16 + 4 * lg);
        } else {
            dv[0] = dv[1] = f32x4{dstat, dstat, dstat, dstat};
        }
        f32x4 s[2], tt[2];
        issue(a.y2r + hb + static_cast<int64_t>(t) * TILE, (q + 1) & 1);
        scores(q & 1, x1, s);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        ++q;
        issue(a.y1c + hb + static_cast<int64_t>(t) * TILE, (q + 1) & 1);
        scores(q & 1, x2, tt);
        // dS = (T - delta~) P (1 - P); padded rows of the stream meet zero rows of Y1 in the last contraction
        f32x4 ds[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pp = sigmoidf(s[rt][r]);
                ds[rt][r] = (tt[rt][r] - dv[rt][r]) * (pp - pp * pp);
            }
        bf16x8 pb[NP];
        split_planes<NP>(ds[0], ds[1], pb);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        ++q;
        if (t + 1 < t1) issue(a.y1r + hb + static_cast<int64_t>(t + 1) * TILE, (q + 1) & 1);
        const bf16x8* b1 = sm + (q & 1) * TILE + lane;
#pragma unroll
        for (int c0 = 0; c0 < FR; c0 += 4) {
            bf16x8 z[NP][4];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (c0 + i < FR) z[pl][i] = b1[(pl * FR + c0 + i) * 64];
#pragma unroll
            for (int tm = 0; tm < TT::n; ++tm)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (c0 + i < FR)
                        o[c0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(z[TT::a(tm)][i], pb[TT::b(tm)], o[c0 + i], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        ++q;
    }
    store_acc<KS>(o, 1.0f, a.w, 16 * g + l15, a.NX, h, a.H, S, split, l15, lg);
}

// S > 1: out = sum_s part[s] (/ sum_s pden[s]); fixed split order
__global__ __launch_bounds__(256) void sigw_combine_kernel(const float* __restrict__ part, const float* __restrict__ pden,
                                                           int64_t NX, int64_t NXPAD, int H, int C, int cout, int S,
                                                           float* __restrict__ out, int64_t ldo, float* __restrict__ den_out,
                                                           int normalize) {
    const int c4 = (cout + 3) / 4;
    const int64_t total = NX * H * c4;
    for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < total; e += static_cast<int64_t>(gridDim.x) * 256) {
        const int cq = static_cast<int>(e % c4);
        const int64_t rh = e / c4;
        const int h = static_cast<int>(rh % H);
        const int64_t row = rh / H;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        float dn = 0.f;
        for (int s = 0; s < S; ++s) {
            const int64_t base = (static_cast<int64_t>(s) * H + h) * NXPAD + row;
            acc += *reinterpret_cast<const f32x4*>(part + base * C + 4 * cq);
            if (pden) dn += pden[base];
        }
        if (normalize) acc *= 1.0f / dn;
        float* dst = out + row * ldo + static_cast<int64_t>(h) * cout + 4 * cq;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * cq + r < cout) dst[r] = acc[r];
        if (den_out && cq == 0) den_out[row * H + h] = dn;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
inline size_t align256(size_t b) { return (b + 255) & ~static_cast<size_t>(255); }
inline int ks_of(int M, int D) { const int c = M > D ? M : D; return (c + 31) / 32; }
inline int fwd_waves(int KS) { return KS <= 10 ? 8 : 4; }
constexpr int kBwdWaves = 4;
constexpr int kNP = 2;

inline size_t packed_bytes(int64_t rows, int H, int KS, int NP) {
    const size_t T = static_cast<size_t>((rows + 31) / 32);
    return align256(static_cast<size_t>(H) * T * NP * 2 * KS * 1024);
}

// Stream splits: one workgroup per CU (its two LDS tiles take 80 .. 128 KiB).  S minimises rounds x (steps per split + a fixed
// cost worth ~3 steps: operand fragments, the epilogue) plus the price of S partial sums (~1 step each).
int sweep_splits(int64_t groups, int64_t steps) {
    int best = 1;
    double best_cost = -1.0;
    const int64_t smax = steps < 8 ? (steps > 0 ? steps : 1) : 8;
    for (int64_t s = 1; s <= smax; ++s) {
        const int64_t rounds = (groups * s + dif::kCUs - 1) / dif::kCUs;
        const double cost = static_cast<double>(rounds) * (static_cast<double>((steps + s - 1) / s) + 3.0) + (s > 1 ? 1.0 * s : 0.0);
        if (best_cost < 0 || cost < best_cost * 0.98) { best_cost = cost; best = static_cast<int>(s); }
    }
    return best;
}

struct SweepPlan { int W; int64_t gx; int S; int64_t NXPAD; size_t part_bytes; };
SweepPlan plan_sweep(int64_t NX, int64_t NY, int H, int KS, int W, bool with_den) {
    SweepPlan p;
    p.W = W;
    p.gx = (NX + 16 * W - 1) / (16 * W);
    p.S = sweep_splits(p.gx * H, (NY + 31) / 32);
    p.NXPAD = p.gx * 16 * W;
    p.part_bytes = p.S > 1 ? align256(static_cast<size_t>(p.S) * H * p.NXPAD * 32 * KS * sizeof(float)) +
                                 (with_den ? align256(static_cast<size_t>(p.S) * H * p.NXPAD * sizeof(float)) : 0)
                           : 0;
    return p;
}

template <typename K>
int set_lds(K kernel, int bytes, const char* who) {
    const hipError_t he = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (he != hipSuccess) return dif::fail(static_cast<int>(he), "%s: LDS attribute: %s", who, hipGetErrorString(he));
    return 0;
}

int launch_pack(const float* x, int64_t ld, int64_t R, int creal, int H, int KS, const float* rowdiv, bf16x8* rowp, bf16x8* colp,
                hipStream_t st) {
    const int T = static_cast<int>((R + 31) / 32);
    const int lds = 32 * (32 * KS + 4) * static_cast<int>(sizeof(float));
    static const int rc = set_lds(&sigw_pack_kernel<kNP>, 32 * (32 * kMaxKS + 4) * static_cast<int>(sizeof(float)), "sigw_pack");
    if (rc) return rc;
    hipLaunchKernelGGL(sigw_pack_kernel<kNP>, dim3(T, H), dim3(256), lds, st, x, ld, R, creal, H, KS, rowdiv, rowp, colp, T);
    return dif::launch_status("sigw_pack_kernel");
}

template <int KS, int W>
int launch_fwd_t(const FwdArgs& a, const SweepPlan& p, hipStream_t st) {
    constexpr int lds = 2 * kNP * 2 * KS * 1024;
    static const int rc = set_lds(&sigw_fwd_kernel<KS, kNP, W>, lds, "sigw_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL((sigw_fwd_kernel<KS, kNP, W>), dim3(static_cast<unsigned>(p.gx), a.H, p.S), dim3(64 * W), lds, st, a);
    return dif::launch_status("sigw_fwd_kernel");
}
template <int KS, bool DSTREAM>
int launch_bwd_t(const BwdArgs& a, const SweepPlan& p, hipStream_t st) {
    constexpr int lds = 2 * kNP * 2 * KS * 1024;
    static const int rc = set_lds(&sigw_bwd_kernel<KS, kNP, kBwdWaves, DSTREAM>, lds, "sigw_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL((sigw_bwd_kernel<KS, kNP, kBwdWaves, DSTREAM>), dim3(static_cast<unsigned>(p.gx), a.H, p.S), dim3(64 * kBwdWaves),
                       lds, st, a);
    return dif::launch_status("sigw_bwd_kernel");
}

int launch_fwd(int KS, const FwdArgs& a, const SweepPlan& p, hipStream_t st) {
    switch (KS) {
#define DIF_CASE(K) case K: return launch_fwd_t<K, (K <= 10 ? 8 : 4)>(a, p, st);
        DIF_CASE(3) DIF_CASE(4) DIF_CASE(5) DIF_CASE(6) DIF_CASE(7) DIF_CASE(8) DIF_CASE(9) DIF_CASE(10)
        DIF_CASE(11) DIF_CASE(12) DIF_CASE(13) DIF_CASE(14) DIF_CASE(15) DIF_CASE(16)
#undef DIF_CASE
    }
    return dif::fail(DIF_E_SHAPE, "sigmoid attention (wide heads): %d columns per head not covered", 32 * KS);
}
template <bool DSTREAM>
int launch_bwd(int KS, const BwdArgs& a, const SweepPlan& p, hipStream_t st) {
    switch (KS) {
#define DIF_CASE(K) case K: return launch_bwd_t<K, DSTREAM>(a, p, st);
        DIF_CASE(3) DIF_CASE(4) DIF_CASE(5) DIF_CASE(6) DIF_CASE(7) DIF_CASE(8) DIF_CASE(9) DIF_CASE(10)
        DIF_CASE(11) DIF_CASE(12) DIF_CASE(13) DIF_CASE(14) DIF_CASE(15) DIF_CASE(16)
#undef DIF_CASE
    }
    return dif::fail(DIF_E_SHAPE, "sigmoid attention backward (wide heads): %d columns per head not covered", 32 * KS);
}

int launch_combine(const SweepPlan& p, const SweepOut& w, int64_t NX, int H, int KS, hipStream_t st) {
    if (p.S == 1) return 0;
    int64_t gr = (NX * H * ((w.cout + 3) / 4) + 255) / 256;
    if (gr > 8 * dif::kCUs) gr = 8 * dif::kCUs;
    hipLaunchKernelGGL(sigw_combine_kernel, dim3(static_cast<unsigned>(gr)), dim3(256), 0, st, w.part, w.pden, NX, p.NXPAD, H, 32 * KS,
                       w.cout, p.S, w.out, w.ldo, w.den_out, w.normalize);
    return dif::launch_status("sigw_combine_kernel");
}

// carve `bytes` out of the workspace
struct Carver {
    char* p; size_t left;
    void* take(size_t bytes) {
        bytes = align256(bytes);
        if (bytes > left) return nullptr;
        void* r = p; p += bytes; left -= bytes;
        return r;
    }
};

}  // namespace

namespace dif {

bool sigw_covers(int M, int D) {
    const int c = M > D ? M : D;
    return c > 64 && c <= 32 * kMaxKS;
}

size_t sigw_fwd_workspace_bytes(int64_t N, int64_t L, int H, int M, int D) {
    const int KS = ks_of(M, D);
    const SweepPlan p = plan_sweep(N, L, H, KS, fwd_waves(KS), true);
    return 256 + packed_bytes(N, H, KS, kNP) + 2 * packed_bytes(L, H, KS, kNP) + p.part_bytes;
}

int sigw_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, int64_t N, int64_t L, int H,
             int M, int D, float* out, int64_t ldo, float* den, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const int KS = ks_of(M, D);
    const SweepPlan p = plan_sweep(N, L, H, KS, fwd_waves(KS), true);
    DIF_REQUIRE(workspace && workspace_bytes >= sigw_fwd_workspace_bytes(N, L, H, M, D), DIF_E_WORKSPACE,
                "sigmoid attention (wide heads): workspace too small (%zu < %zu)", workspace_bytes,
                sigw_fwd_workspace_bytes(N, L, H, M, D));
    DIF_REQUIRE(H <= 65535, DIF_E_RANGE, "sigmoid attention (wide heads): too many heads");
    Carver cv{static_cast<char*>(workspace), workspace_bytes};
    {   // 256-byte alignment of the packed tiles
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(cv.p), a1 = (a0 + 255) & ~static_cast<uintptr_t>(255);
        cv.p += a1 - a0; cv.left -= a1 - a0;
    }
    bf16x8* qr = static_cast<bf16x8*>(cv.take(packed_bytes(N, H, KS, kNP)));
    bf16x8* kr = static_cast<bf16x8*>(cv.take(packed_bytes(L, H, KS, kNP)));
    bf16x8* vc = static_cast<bf16x8*>(cv.take(packed_bytes(L, H, KS, kNP)));
    if (int rc = launch_pack(q, ldq, N, M, H, KS, nullptr, qr, nullptr, st)) return rc;
    if (int rc = launch_pack(k, ldk, L, M, H, KS, nullptr, kr, nullptr, st)) return rc;
    if (int rc = launch_pack(v, ldv, L, D, H, KS, nullptr, nullptr, vc, st)) return rc;
    FwdArgs a;
    a.xr = qr; a.yr = kr; a.zc = vc;
    a.NX = N; a.NY = L; a.Tx = static_cast<int>((N + 31) / 32); a.Ty = static_cast<int>((L + 31) / 32); a.H = H;
    a.w = SweepOut{out, ldo, D, den, nullptr, nullptr, p.NXPAD, 1};
    if (p.S > 1) {
        a.w.part = static_cast<float*>(cv.take(static_cast<size_t>(p.S) * H * p.NXPAD * 32 * KS * sizeof(float)));
        a.w.pden = static_cast<float*>(cv.take(static_cast<size_t>(p.S) * H * p.NXPAD * sizeof(float)));
    }
    if (int rc = launch_fwd(KS, a, p, st)) return rc;
    return launch_combine(p, a.w, N, H, KS, st);
}

size_t sigw_bwd_workspace_bytes(int64_t N, int64_t L, int H, int M, int D) {
    const int KS = ks_of(M, D);
    const SweepPlan pq = plan_sweep(N, L, H, KS, kBwdWaves, false), pk = plan_sweep(L, N, H, KS, kBwdWaves, false),
                    pv = plan_sweep(L, N, H, KS, fwd_waves(KS), false);
    size_t part = pq.part_bytes > pk.part_bytes ? pq.part_bytes : pk.part_bytes;
    if (pv.part_bytes > part) part = pv.part_bytes;
    const size_t ndpad = static_cast<size_t>((N + 63) / 64 * 64 + 64);
    return 256 + 4 * packed_bytes(N, H, KS, kNP) + 3 * packed_bytes(L, H, KS, kNP) + align256(H * ndpad * sizeof(float)) + part;
}

int sigw_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* out, int64_t ldo,
             const float* den, const float* g, int64_t ldg, int64_t N, int64_t L, int H, int M, int D, float* dq, int64_t lddq,
             float* dk, int64_t lddk, float* dv, int64_t lddv, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const int KS = ks_of(M, D);
    DIF_REQUIRE(workspace && workspace_bytes >= sigw_bwd_workspace_bytes(N, L, H, M, D), DIF_E_WORKSPACE,
                "sigmoid attention backward (wide heads): workspace too small (%zu < %zu)", workspace_bytes,
                sigw_bwd_workspace_bytes(N, L, H, M, D));
    DIF_REQUIRE(H <= 65535, DIF_E_RANGE, "sigmoid attention backward (wide heads): too many heads");
    Carver cv{static_cast<char*>(workspace), workspace_bytes};
    {
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(cv.p), a1 = (a0 + 255) & ~static_cast<uintptr_t>(255);
        cv.p += a1 - a0; cv.left -= a1 - a0;
    }
    const size_t pn = packed_bytes(N, H, KS, kNP), pl = packed_bytes(L, H, KS, kNP);
    bf16x8* qr = static_cast<bf16x8*>(cv.take(pn));
    bf16x8* qc = static_cast<bf16x8*>(cv.take(pn));
    bf16x8* gr = static_cast<bf16x8*>(cv.take(pn));
    bf16x8* gc = static_cast<bf16x8*>(cv.take(pn));
    bf16x8* kr = static_cast<bf16x8*>(cv.take(pl));
    bf16x8* kc = static_cast<bf16x8*>(cv.take(pl));
    bf16x8* vr = static_cast<bf16x8*>(cv.take(pl));
    const int64_t ndpad = (N + 63) / 64 * 64 + 64;
    float* delta = static_cast<float*>(cv.take(static_cast<size_t>(H) * ndpad * sizeof(float)));
    float* part = reinterpret_cast<float*>(cv.p);
    hipLaunchKernelGGL(sigw_delta_kernel, dim3(static_cast<unsigned>((ndpad * H + 3) / 4)), dim3(256), 0, st, g, ldg, out, ldo, den, N,
                       ndpad, H, D, delta);
    if (int rc = dif::launch_status("sigw_delta_kernel")) return rc;
    if (int rc = launch_pack(q, ldq, N, M, H, KS, nullptr, qr, qc, st)) return rc;
    if (int rc = launch_pack(k, ldk, L, M, H, KS, nullptr, kr, kc, st)) return rc;
    if (int rc = launch_pack(v, ldv, L, D, H, KS, nullptr, vr, nullptr, st)) return rc;
    if (int rc = launch_pack(g, ldg, N, D, H, KS, den, gr, gc, st)) return rc;
    const int Tn = static_cast<int>((N + 31) / 32), Tl = static_cast<int>((L + 31) / 32);
    {   // dQ: stationary = queries
        const SweepPlan p = plan_sweep(N, L, H, KS, kBwdWaves, false);
        BwdArgs a{qr, gr, kr, vr, kc, delta, ndpad, N, L, Tn, Tl, H, SweepOut{dq, lddq, M, nullptr, p.S > 1 ? part : nullptr, nullptr, p.NXPAD, 0}};
        if (int rc = launch_bwd<false>(KS, a, p, st)) return rc;
        if (int rc = launch_combine(p, a.w, N, H, KS, st)) return rc;
    }
    {   // dK: stationary = keys, delta~ rides on the streamed queries
        const SweepPlan p = plan_sweep(L, N, H, KS, kBwdWaves, false);
        BwdArgs a{kr, vr, qr, gr, qc, delta, ndpad, L, N, Tl, Tn, H, SweepOut{dk, lddk, M, nullptr, p.S > 1 ? part : nullptr, nullptr, p.NXPAD, 0}};
        if (int rc = launch_bwd<true>(KS, a, p, st)) return rc;
        if (int rc = launch_combine(p, a.w, L, H, KS, st)) return rc;
    }
    {   // dV = P^T G~: the forward sweep with the roles of Q and K exchanged, raw sums
        const SweepPlan p = plan_sweep(L, N, H, KS, fwd_waves(KS), false);
        FwdArgs a;
        a.xr = kr; a.yr = qr; a.zc = gc;
        a.NX = L; a.NY = N; a.Tx = Tl; a.Ty = Tn; a.H = H;
        a.w = SweepOut{dv, lddv, D, nullptr, p.S > 1 ? part : nullptr, nullptr, p.NXPAD, 0};
        if (int rc = launch_fwd(KS, a, p, st)) return rc;
        if (int rc = launch_combine(p, a.w, L, H, KS, st)) return rc;
    }
    return 0;
}

}  // namespace dif

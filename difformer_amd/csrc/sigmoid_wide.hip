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

// product terms (streamed plane a, stationary plane b) and the accumulator each adds to (0 = hi.hi, 1 = the small terms):
// consecutive terms alternate accumulators, so that no MFMA waits for the one before it
template <int NP> struct Terms;
template <> struct Terms<2> {
    static constexpr int n = 3;                                                                // lo.hi  hi.hi  hi.lo
    static __device__ __forceinline__ constexpr int a(int i) { return i == 0 ? 1 : 0; }
    static __device__ __forceinline__ constexpr int b(int i) { return i == 2 ? 1 : 0; }
    static __device__ __forceinline__ constexpr int acc(int i) { return i == 1 ? 0 : 1; }
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
// pack: one workgroup per (32-row tile, head, chunk of 64 columns = 2 column steps / 4 column tiles); one fragment lane per
// thread and orientation.  x [R][ld] (head h at column h * creal), optional per-row divisor rowdiv [R][H] (G~ = G / den),
// optional per-column offset colsub [H][creal] that is subtracted (the centred values V - mean V).
template <int NP>
__global__ __launch_bounds__(256) void sigw_pack_kernel(const float* __restrict__ x, int64_t ld, int64_t R, int creal, int H,
                                                        int KS, const float* __restrict__ rowdiv, const float* __restrict__ colsub,
                                                        bf16x8* __restrict__ rowp, bf16x8* __restrict__ colp, int T) {
    __shared__ __attribute__((aligned(16))) float smf[32][68];
    const int t = blockIdx.x, h = blockIdx.y, c0 = 64 * blockIdx.z;
    {   // 32 rows x 64 columns: 8 threads per row, 16 bytes each, twice
        const int r = threadIdx.x >> 3;
        const int64_t row = 32ll * t + r;
        const float* src = x + row * ld + static_cast<int64_t>(h) * creal;
        const bool vec = (creal % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0);
        float inv = 1.0f;
        if (rowdiv && row < R) inv = rowdiv[row * H + h];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cl = 4 * (threadIdx.x & 7) + 32 * half, c = c0 + cl;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < R) {
                if (vec) {
                    if (c < creal) v = *reinterpret_cast<const f32x4*>(src + c);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c + j < creal) v[j] = src[c + j];
                }
                if (rowdiv) v = f32x4{v[0] / inv, v[1] / inv, v[2] / inv, v[3] / inv};
                if (colsub) {                                  // centred values (see sigw_colmean): padded columns stay zero
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c + j < creal) v[j] -= colsub[static_cast<int64_t>(h) * creal + c + j];
                }
            }
            *reinterpret_cast<f32x4*>(&smf[r][cl]) = v;
        }
    }
    __syncthreads();
    const int FR = 2 * KS;
    const int64_t tile = (static_cast<int64_t>(h) * T + t) * NP * FR * 64;
    const int f4 = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4;
    if (rowp) {                                            // fragment (rt, ks): rt = f4 / 2, ks = 2 blockIdx.z + f4 % 2
        const int rt = f4 >> 1, ksl = f4 & 1, ks = 2 * blockIdx.z + ksl;
        if (ks < KS) {
            const float* sp = &smf[16 * rt + l15][32 * ksl + 8 * lg];
            bf16x8 pl[NP];
            split_planes<NP>(*reinterpret_cast<const f32x4*>(sp), *reinterpret_cast<const f32x4*>(sp + 4), pl);
#pragma unroll
            for (int p = 0; p < NP; ++p) rowp[tile + (static_cast<int64_t>(p) * FR + rt * KS + ks) * 64 + lane] = pl[p];
        }
    }
    if (colp) {                                            // column tile ct = 4 blockIdx.z + f4
        const int ct = 4 * blockIdx.z + f4;
        if (ct < FR) {
            const int col = 16 * f4 + l15;
            f32x4 v0, v1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v0[j] = smf[4 * lg + j][col];
                v1[j] = smf[16 + 4 * lg + j][col];
            }
            bf16x8 pl[NP];
            split_planes<NP>(v0, v1, pl);
#pragma unroll
            for (int p = 0; p < NP; ++p) colp[tile + (static_cast<int64_t>(p) * FR + ct) * 64 + lane] = pl[p];
        }
    }
}

// Column means of v [L][H x D] -> cmean [H][D] in two deterministic stages (row blocks of 128, then the blocks in order).
// WHY: out_n is a convex combination of the value rows, so  g.v_l - g.out_n = g.(v_l - c) - g.(out_n - c)  for any c.  The
// backward's dS = (g~.v_l - g~.out_n) P (1 - P) takes the first product off the matrix core at ~4e-6 |g~||v| (split planes) and
// the second from float32 row arithmetic: where the value rows resemble each other (deep layers behind LayerNorm + residual)
// the difference is far smaller than either term and the split error came back amplified (last layer of an 8-layer model:
// 1.5e-4 of a Wk.bias gradient).  With c = the column mean both terms shrink to the rows' spread; the forward sums
// P (v - c) and adds c back for the same reason.
constexpr int kColsumRows = 64;
__host__ __device__ inline int64_t colsum_blocks(int64_t L) { return (L + kColsumRows - 1) / kColsumRows; }
// partial [blocks][HD]: a thread owns a column of the block's 64 rows, eight loads in flight (the column walk is latency-bound:
// one accumulator per thread made this 52 us at 15,000 x 300, two small kernels were 8 % of that forward)
__global__ __launch_bounds__(256) void sigw_colsum_kernel(const float* __restrict__ v, int64_t ldv, int64_t L, int HD,
                                                          float* __restrict__ partial) {
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * kColsumRows;
    const int rows = static_cast<int>(r0 + kColsumRows < L ? kColsumRows : L - r0);
    for (int c = threadIdx.x; c < HD; c += 256) {
        const float* p = v + r0 * ldv + c;
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = 0.f;
        int r = 0;
        for (; r + 8 <= rows; r += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += p[static_cast<int64_t>(r + u) * ldv];
        }
        for (; r < rows; ++r) a[0] += p[static_cast<int64_t>(r) * ldv];
        partial[static_cast<int64_t>(blockIdx.x) * HD + c] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
}
// cmean [HD]: a workgroup owns 64 columns, its four waves every fourth block of partials (fixed order: reproducible)
__global__ __launch_bounds__(256) void sigw_colmean_kernel(const float* __restrict__ partial, int blocks, int64_t L, int HD,
                                                           float* __restrict__ cmean) {
    __shared__ float sS[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < HD) {
        int b = w;
        for (; b + 12 < blocks; b += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += partial[static_cast<int64_t>(b + 4 * u) * HD + c];
        }
        for (; b < blocks; b += 4) a[0] += partial[static_cast<int64_t>(b) * HD + c];
    }
    sS[w][lane] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (w == 0 && c < HD) cmean[c] = ((sS[0][lane] + sS[1][lane]) + (sS[2][lane] + sS[3][lane])) / static_cast<float>(L);
}

// delta~[h][n] = g_n . (out_n - c) / den_n for n < N, 0 up to NPAD (one wave per (n, h)); c [H][D] = the value rows' centre
__global__ __launch_bounds__(256) void sigw_delta_kernel(const float* __restrict__ g, int64_t ldg, const float* __restrict__ out,
                                                         int64_t ldo, const float* __restrict__ den, const float* __restrict__ cmean,
                                                         int64_t N, int64_t NPAD, int H, int D, float* __restrict__ delta) {
    const int lane = threadIdx.x & 63;
    const int64_t item = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (item >= NPAD * H) return;
    const int64_t n = item / H;
    const int h = static_cast<int>(item - n * H);
    float s = 0.f;
    if (n < N) {
        for (int d = lane; d < D; d += 64) s += g[n * ldg + h * D + d] * (out[n * ldo + h * D + d] - cmean[h * D + d]);
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
    const float* coladd;                        // nullable [H][cout]: added to the (normalised) rows -- the centre of the values
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
            const float* add = w.coladd ? w.coladd + static_cast<int64_t>(h) * w.cout : nullptr;
            const bool vec = (w.cout % 4 == 0) && (w.ldo % 4 == 0) && (reinterpret_cast<uintptr_t>(w.out) & 15u) == 0;
#pragma unroll
            for (int ct = 0; ct < 2 * KS; ++ct) {
                const int c = 16 * ct + 4 * lg;
                f32x4 val = o[ct] * sc;
                if (add) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c + r < w.cout) val[r] += add[c + r];
                }
                if (vec) {
                    if (c < w.cout) *reinterpret_cast<f32x4*>(dst + c) = val;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c + r < w.cout) dst[c + r] = val[r];
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

// ---- the two stage bodies shared by both sweep kernels.  `tile` = LDS tile + lane.  Fragments are fetched one step AHEAD of the
// products that use them (two register sets): with one or two waves per SIMD nothing else hides the LDS round trip. ----

// S^T[streamed row 4 lg + r of row tile rt][stationary row l15 of tile qt] = Y X^T over the KS column steps
template <int KS, int NP, int QT>
__device__ __forceinline__ void score_stage(const bf16x8* tile, const bf16x8 (&xs)[QT][NP][KS], f32x4 (&res)[QT][2]) {
    constexpr int FR = 2 * KS;
    using TT = Terms<NP>;
    f32x4 sa[2][QT][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) sa[i][qt][0] = sa[i][qt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 y[2][NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) y[0][p][rt] = tile[(p * FR + rt * KS) * 64];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) y[(ks + 1) & 1][p][rt] = tile[(p * FR + rt * KS + ks + 1) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);               // (the scheduler would sink the reads to just before their use)
#pragma unroll
        for (int i = 0; i < TT::n; ++i)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    sa[TT::acc(i)][qt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y[ks & 1][TT::a(i)][rt], xs[qt][TT::b(i)][ks],
                                                                                   sa[TT::acc(i)][qt][rt], 0, 0, 0);
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) res[qt][rt] = sa[1][qt][rt] + sa[0][qt][rt];
}

// O^T[col 16 ct + 4 lg + r][stationary row l15 of tile qt] += Z^T B over the 32 streamed rows (B = the planes of sigma(S^T) or dS^T)
template <int KS, int NP, int QT>
__device__ __forceinline__ void accumulate_stage(const bf16x8* tile, const bf16x8 (&pb)[QT][NP], f32x4 (&o)[QT][2 * KS]) {
    constexpr int FR = 2 * KS;
    using TT = Terms<NP>;
    bf16x8 z[2][NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < 2; ++i) z[0][p][i] = tile[(p * FR + i) * 64];
#pragma unroll
    for (int c0 = 0; c0 < FR; c0 += 2) {
        if (c0 + 2 < FR) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) z[((c0 >> 1) + 1) & 1][p][i] = tile[(p * FR + c0 + 2 + i) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < TT::n; ++tm)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    o[qt][c0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(z[(c0 >> 1) & 1][TT::a(tm)][i], pb[qt][TT::b(tm)], o[qt][c0 + i],
                                                                           0, 0, 0);
    }
}

// LDS-DMA of one tile (NP x 2 KS KiB): one KiB per wave instruction (lane l's 16 bytes land at M0 + 16 l), no registers.
// Inline assembly on purpose: with `__builtin_amdgcn_global_load_lds` in the kernel the compiler's wait-count pass answers every
// `ds_read` -> MFMA dependence with `s_waitcnt lgkmcnt(0)` -- the fragment reads issued one step ahead are waited for together
// with the ones needed now, and the prefetch buys nothing (the same loop without the builtin gets `lgkmcnt(4)`).  The compiler
// does not count these loads: every sub-stage ends with an explicit vmcnt(0) + barrier.  M0 is saved and restored inside the
// statement (cdna_hip_programming.md 5.7).  Issued at the head of a sub-stage; the sched_barrier keeps it there.
__device__ __forceinline__ void dma_1k(const bf16x8* gsrc, const bf16x8* lds_dst) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        static_cast<unsigned>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) bf16x8*)lds_dst)));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
template <int KS, int NP, int W>
__device__ __forceinline__ void issue_tile(const bf16x8* tile, bf16x8* dst, int wave, int lane) {
    constexpr int PIECES = NP * 2 * KS;
#pragma unroll
    for (int i = 0; i < (PIECES + W - 1) / W; ++i) {
        const int piece = wave + i * W;
        if (PIECES % W == 0 || piece < PIECES) dma_1k(tile + piece * 64 + lane, dst + piece * 64);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// fragments of this wave's QT stationary 16-row tiles (tile index g0 + qt, clamped to the packed rows)
template <int KS, int NP, int QT>
__device__ __forceinline__ void load_stationary(const bf16x8* xr, int h, int Tx, int64_t g0, int lane, bf16x8 (&xf)[QT][NP][KS]) {
    constexpr int FR = 2 * KS;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int64_t g = g0 + qt < 2ll * Tx ? g0 + qt : 2ll * Tx - 1;
        const bf16x8* xt = xr + ((static_cast<int64_t>(h) * Tx + (g >> 1)) * NP * FR + (g & 1) * KS) * 64 + lane;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[qt][p][ks] = xt[(p * FR + ks) * 64];
    }
}

struct FwdArgs {
    const bf16x8* xr; const bf16x8* yr; const bf16x8* zc;
    int64_t NX, NY;
    int Tx, Ty, H;
    SweepOut w;
};

// grid (ceil(NX / (16 W QT)), H, S); block 64 W; dynamic LDS 2 tiles
template <int KS, int NP, int W, int QT>
__global__ __launch_bounds__(64 * W) void sigw_fwd_kernel(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16x8 sm[];
    constexpr int FR = 2 * KS;
    constexpr int TILE = NP * FR * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
    const int h = blockIdx.y, S = gridDim.z, split = blockIdx.z;
    const int64_t g0 = (static_cast<int64_t>(blockIdx.x) * W + wave) * QT;       // this wave's stationary 16-row tiles
    bf16x8 xf[QT][NP][KS];
    load_stationary<KS, NP, QT>(a.xr, h, a.Tx, g0, lane, xf);
    const int per = (a.Ty + S - 1) / S;
    const int t0 = split * per, t1 = (t0 + per < a.Ty) ? t0 + per : a.Ty;
    const bf16x8* ybase = a.yr + static_cast<int64_t>(h) * a.Ty * TILE;
    const bf16x8* zbase = a.zc + static_cast<int64_t>(h) * a.Ty * TILE;
    f32x4 o[QT][FR];
    float den[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        den[qt] = 0.f;
#pragma unroll
        for (int ct = 0; ct < FR; ++ct) o[qt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (t0 < t1) issue_tile<KS, NP, W>(ybase + static_cast<int64_t>(t0) * TILE, sm, wave, lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        issue_tile<KS, NP, W>(zbase + static_cast<int64_t>(t) * TILE, sm + TILE, wave, lane);
        f32x4 s[QT][2];
        score_stage<KS, NP, QT>(sm + lane, xf, s);
        bf16x8 pb[QT][NP];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x4 p[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = 32ll * t + 16 * rt + 4 * lg + r < a.NY;      // padded rows of the stream: sigma(0) = 1/2 stays out of den
                    p[rt][r] = ok ? sigmoidf(s[qt][rt][r]) : 0.f;
                    den[qt] += p[rt][r];
                }
            split_planes<NP>(p[0], p[1], pb[qt]);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (t + 1 < t1) issue_tile<KS, NP, W>(ybase + static_cast<int64_t>(t + 1) * TILE, sm, wave, lane);
        accumulate_stage<KS, NP, QT>(sm + TILE + lane, pb, o);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
        store_acc<KS>(o[qt], dif::rows4_sum(den[qt]), a.w, 16 * (g0 + qt) + l15, a.NX, h, a.H, S, split, l15, lg);
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
    const int h = blockIdx.y, S = gridDim.z, split = blockIdx.z;
    const int64_t g = static_cast<int64_t>(blockIdx.x) * W + wave;
    const int64_t gc = g < 2ll * a.Tx ? g : 2ll * a.Tx - 1;
    bf16x8 x1[1][NP][KS], x2[1][NP][KS];
    load_stationary<KS, NP, 1>(a.x1r, h, a.Tx, g, lane, x1);
    load_stationary<KS, NP, 1>(a.x2r, h, a.Tx, g, lane, x2);
    const float* dl = a.delta + static_cast<int64_t>(h) * a.NDPAD;
    const float dstat = DSTREAM ? 0.f : dl[16 * gc + l15];
    const int per = (a.Ty + S - 1) / S;
    const int t0 = split * per, t1 = (t0 + per < a.Ty) ? t0 + per : a.Ty;
    const int64_t hb = static_cast<int64_t>(h) * a.Ty * TILE;
    f32x4 o[1][FR];
#pragma unroll
    for (int ct = 0; ct < FR; ++ct) o[0][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    int q = 0;                                                          // sub-stage counter: tile q sits in buffer q & 1
    if (t0 < t1) issue_tile<KS, NP, W>(a.y1r + hb + static_cast<int64_t>(t0) * TILE, sm, wave, lane);
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
        f32x4 s[1][2], tt[1][2];
        issue_tile<KS, NP, W>(a.y2r + hb + static_cast<int64_t>(t) * TILE, sm + ((q + 1) & 1) * TILE, wave, lane);
        score_stage<KS, NP, 1>(sm + (q & 1) * TILE + lane, x1, s);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        ++q;
        issue_tile<KS, NP, W>(a.y1c + hb + static_cast<int64_t>(t) * TILE, sm + ((q + 1) & 1) * TILE, wave, lane);
        score_stage<KS, NP, 1>(sm + (q & 1) * TILE + lane, x2, tt);
        // dS = (T - delta~) P (1 - P); padded rows of the stream meet zero rows of Y1 in the last contraction
        f32x4 ds[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pp = sigmoidf(s[0][rt][r]);
                ds[rt][r] = (tt[0][rt][r] - dv[rt][r]) * (pp - pp * pp);
            }
        bf16x8 pb[1][NP];
        split_planes<NP>(ds[0], ds[1], pb[0]);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        ++q;
        if (t + 1 < t1) issue_tile<KS, NP, W>(a.y1r + hb + static_cast<int64_t>(t + 1) * TILE, sm + ((q + 1) & 1) * TILE, wave, lane);
        accumulate_stage<KS, NP, 1>(sm + (q & 1) * TILE + lane, pb, o);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        ++q;
    }
    store_acc<KS>(o[0], 1.0f, a.w, 16 * g + l15, a.NX, h, a.H, S, split, l15, lg);
}

// S > 1: out = sum_s part[s] (/ sum_s pden[s]); fixed split order
__global__ __launch_bounds__(256) void sigw_combine_kernel(const float* __restrict__ part, const float* __restrict__ pden,
                                                           int64_t NX, int64_t NXPAD, int H, int C, int cout, int S,
                                                           float* __restrict__ out, int64_t ldo, float* __restrict__ den_out,
                                                           int normalize, const float* __restrict__ coladd) {
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
            if (4 * cq + r < cout) dst[r] = acc[r] + (coladd ? coladd[static_cast<int64_t>(h) * cout + 4 * cq + r] : 0.f);
        if (den_out && cq == 0) den_out[row * H + h] = dn;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
inline size_t align256(size_t b) { return (b + 255) & ~static_cast<size_t>(255); }
inline int ks_of(int M, int D) { const int c = M > D ? M : D; return (c + 31) / 32; }
// forward sweep: waves per workgroup and stationary 16-row tiles per wave.  Up to 320 columns eight waves fit 256 registers (80
// for the stationary fragments, 80 for the accumulators at KS = 10); beyond, four waves with up to 512.  (Four waves with TWO row
// tiles each -- half the LDS reads per product, one wave per SIMD -- measured the same 0.73 ms at 15,000 x 300; kept as QT.)
constexpr int fwd_waves_c(int KS) { return KS <= 10 ? 8 : 4; }
constexpr int fwd_qt_c(int KS) { return 1; }
inline int fwd_rows(int KS) { return 16 * fwd_waves_c(KS) * fwd_qt_c(KS); }
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

struct SweepPlan { int64_t gx; int S; int64_t NXPAD; size_t part_bytes; };
SweepPlan plan_sweep(int64_t NX, int64_t NY, int H, int KS, int rows_per_wg, bool with_den) {
    SweepPlan p;
    p.gx = (NX + rows_per_wg - 1) / rows_per_wg;
    p.S = sweep_splits(p.gx * H, (NY + 31) / 32);
    p.NXPAD = p.gx * rows_per_wg;
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

int launch_pack(const float* x, int64_t ld, int64_t R, int creal, int H, int KS, const float* rowdiv, const float* colsub, bf16x8* rowp,
                bf16x8* colp, hipStream_t st) {
    const int T = static_cast<int>((R + 31) / 32);
    hipLaunchKernelGGL(sigw_pack_kernel<kNP>, dim3(T, H, (KS + 1) / 2), dim3(256), 0, st, x, ld, R, creal, H, KS, rowdiv, colsub, rowp, colp,
                       T);
    return dif::launch_status("sigw_pack_kernel");
}

template <int KS>
int launch_fwd_t(const FwdArgs& a, const SweepPlan& p, hipStream_t st) {
    constexpr int lds = 2 * kNP * 2 * KS * 1024;
    constexpr int W = fwd_waves_c(KS), QT = fwd_qt_c(KS);
    static const int rc = set_lds(&sigw_fwd_kernel<KS, kNP, W, QT>, lds, "sigw_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL((sigw_fwd_kernel<KS, kNP, W, QT>), dim3(static_cast<unsigned>(p.gx), a.H, p.S), dim3(64 * W), lds, st, a);
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
#define DIF_CASE(K) case K: return launch_fwd_t<K>(a, p, st);
        DIF_CASE(2) DIF_CASE(3) DIF_CASE(4) DIF_CASE(5) DIF_CASE(6) DIF_CASE(7) DIF_CASE(8) DIF_CASE(9) DIF_CASE(10)
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
                       w.cout, p.S, w.out, w.ldo, w.den_out, w.normalize, w.coladd);
    return dif::launch_status("sigw_combine_kernel");
}

// cmean [H][D] = column means of v; partial: ceil(L / 128) x H D floats
inline size_t colmean_bytes(int64_t L, int H, int D) {
    return align256(static_cast<size_t>(colsum_blocks(L)) * H * D * sizeof(float)) + align256(static_cast<size_t>(H) * D * sizeof(float));
}
int launch_colmean(const float* v, int64_t ldv, int64_t L, int H, int D, float* partial, float* cmean, hipStream_t st) {
    const int blocks = static_cast<int>(colsum_blocks(L));
    hipLaunchKernelGGL(sigw_colsum_kernel, dim3(blocks), dim3(256), 0, st, v, ldv, L, H * D, partial);
    if (int rc = dif::launch_status("sigw_colsum_kernel")) return rc;
    hipLaunchKernelGGL(sigw_colmean_kernel, dim3((H * D + 63) / 64), dim3(256), 0, st, partial, blocks, L, H * D, cmean);
    return dif::launch_status("sigw_colmean_kernel");
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

// 33 .. 64 columns per head (KS = 2): the same plane kernels against csrc/sigmoid_attn.hip.  Measured (scripts/exp_sigw_narrow.py,
// M = D = 64): inference forward 0.714 -> 0.406 ms at 20,000 rows, 0.119 -> 0.103 at 8,192, 0.028 -> 0.041 at 2,708 (slower: the
// packs and the partial sums are a fixed cost) -- taken for INFERENCE from 2^25 pairs.  Training forward + backward would go
// 4.86 -> 1.95 ms at 20,000 rows, but the deepest script (node classification/run.sh:10: 8 layers, hidden 64) then has two last-layer
// gradients at 1.9e-4 of themselves (tests/test_gpu_grad.py::test_deepest_sigmoid_script_eight_layers; the fp32 chain: < 1e-4):
// training at <= 64 columns keeps the fp32-chain kernels, as round 5 decided for the split operands of the narrow kernel.
bool sigw_narrow_pays(int M, int D, int64_t N, int64_t L, bool training) {
    const int c = M > D ? M : D;
    return !training && c > 32 && c <= 64 && N * L >= (int64_t(1) << 25);
}

size_t sigw_fwd_workspace_bytes(int64_t N, int64_t L, int H, int M, int D) {
    const int KS = ks_of(M, D);
    const SweepPlan p = plan_sweep(N, L, H, KS, fwd_rows(KS), true);
    return 256 + packed_bytes(N, H, KS, kNP) + 2 * packed_bytes(L, H, KS, kNP) + colmean_bytes(L, H, D) + p.part_bytes;
}

int sigw_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, int64_t N, int64_t L, int H,
             int M, int D, float* out, int64_t ldo, float* den, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const int KS = ks_of(M, D);
    const SweepPlan p = plan_sweep(N, L, H, KS, fwd_rows(KS), true);
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
    float* cpart = static_cast<float*>(cv.take(static_cast<size_t>(colsum_blocks(L)) * H * D * sizeof(float)));
    float* cmean = static_cast<float*>(cv.take(static_cast<size_t>(H) * D * sizeof(float)));
    if (int rc = launch_colmean(v, ldv, L, H, D, cpart, cmean, st)) return rc;
    if (int rc = launch_pack(q, ldq, N, M, H, KS, nullptr, nullptr, qr, nullptr, st)) return rc;
    if (int rc = launch_pack(k, ldk, L, M, H, KS, nullptr, nullptr, kr, nullptr, st)) return rc;
    if (int rc = launch_pack(v, ldv, L, D, H, KS, nullptr, cmean, nullptr, vc, st)) return rc;       // centred values; the centre comes back in the epilogue
    FwdArgs a;
    a.xr = qr; a.yr = kr; a.zc = vc;
    a.NX = N; a.NY = L; a.Tx = static_cast<int>((N + 31) / 32); a.Ty = static_cast<int>((L + 31) / 32); a.H = H;
    a.w = SweepOut{out, ldo, D, den, nullptr, nullptr, p.NXPAD, 1, cmean};
    if (p.S > 1) {
        a.w.part = static_cast<float*>(cv.take(static_cast<size_t>(p.S) * H * p.NXPAD * 32 * KS * sizeof(float)));
        a.w.pden = static_cast<float*>(cv.take(static_cast<size_t>(p.S) * H * p.NXPAD * sizeof(float)));
    }
    if (int rc = launch_fwd(KS, a, p, st)) return rc;
    return launch_combine(p, a.w, N, H, KS, st);
}

size_t sigw_bwd_workspace_bytes(int64_t N, int64_t L, int H, int M, int D) {
    const int KS = ks_of(M, D);
    const SweepPlan pq = plan_sweep(N, L, H, KS, 16 * kBwdWaves, false), pk = plan_sweep(L, N, H, KS, 16 * kBwdWaves, false),
                    pv = plan_sweep(L, N, H, KS, fwd_rows(KS), false);
    size_t part = pq.part_bytes > pk.part_bytes ? pq.part_bytes : pk.part_bytes;
    if (pv.part_bytes > part) part = pv.part_bytes;
    const size_t ndpad = static_cast<size_t>((N + 63) / 64 * 64 + 64);
    return 256 + 4 * packed_bytes(N, H, KS, kNP) + 3 * packed_bytes(L, H, KS, kNP) + align256(H * ndpad * sizeof(float)) +
           colmean_bytes(L, H, D) + part;
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
    float* cpart = static_cast<float*>(cv.take(static_cast<size_t>(colsum_blocks(L)) * H * D * sizeof(float)));
    float* cmean = static_cast<float*>(cv.take(static_cast<size_t>(H) * D * sizeof(float)));
    float* part = reinterpret_cast<float*>(cv.p);
    if (int rc = launch_colmean(v, ldv, L, H, D, cpart, cmean, st)) return rc;
    hipLaunchKernelGGL(sigw_delta_kernel, dim3(static_cast<unsigned>((ndpad * H + 3) / 4)), dim3(256), 0, st, g, ldg, out, ldo, den, cmean,
                       N, ndpad, H, D, delta);
    if (int rc = dif::launch_status("sigw_delta_kernel")) return rc;
    if (int rc = launch_pack(q, ldq, N, M, H, KS, nullptr, nullptr, qr, qc, st)) return rc;
    if (int rc = launch_pack(k, ldk, L, M, H, KS, nullptr, nullptr, kr, kc, st)) return rc;
    if (int rc = launch_pack(v, ldv, L, D, H, KS, nullptr, cmean, vr, nullptr, st)) return rc;       // centred: T - delta~ = g~.(v - c) - g~.(out - c)
    if (int rc = launch_pack(g, ldg, N, D, H, KS, den, nullptr, gr, gc, st)) return rc;
    const int Tn = static_cast<int>((N + 31) / 32), Tl = static_cast<int>((L + 31) / 32);
    {   // dQ: stationary = queries
        const SweepPlan p = plan_sweep(N, L, H, KS, 16 * kBwdWaves, false);
        BwdArgs a{qr, gr, kr, vr, kc, delta, ndpad, N, L, Tn, Tl, H, SweepOut{dq, lddq, M, nullptr, p.S > 1 ? part : nullptr, nullptr, p.NXPAD, 0, nullptr}};
        if (int rc = launch_bwd<false>(KS, a, p, st)) return rc;
        if (int rc = launch_combine(p, a.w, N, H, KS, st)) return rc;
    }
    {   // dK: stationary = keys, delta~ rides on the streamed queries
        const SweepPlan p = plan_sweep(L, N, H, KS, 16 * kBwdWaves, false);
        BwdArgs a{kr, vr, qr, gr, qc, delta, ndpad, L, N, Tl, Tn, H, SweepOut{dk, lddk, M, nullptr, p.S > 1 ? part : nullptr, nullptr, p.NXPAD, 0, nullptr}};
        if (int rc = launch_bwd<true>(KS, a, p, st)) return rc;
        if (int rc = launch_combine(p, a.w, L, H, KS, st)) return rc;
    }
    {   // dV = P^T G~: the forward sweep with the roles of Q and K exchanged, raw sums
        const SweepPlan p = plan_sweep(L, N, H, KS, fwd_rows(KS), false);
        FwdArgs a;
        a.xr = kr; a.yr = qr; a.zc = gc;
        a.NX = L; a.NY = N; a.Tx = Tl; a.Ty = Tn; a.H = H;
        a.w = SweepOut{dv, lddv, D, nullptr, p.S > 1 ? part : nullptr, nullptr, p.NXPAD, 0, nullptr};
        if (int rc = launch_fwd(KS, a, p, st)) return rc;
        if (int rc = launch_combine(p, a.w, L, H, KS, st)) return rc;
    }
    return 0;
}

}  // namespace dif

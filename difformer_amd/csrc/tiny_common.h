// Shared by the whole-model kernels for tiny graphs (tiny_model.hip: one workgroup; tiny_sigmoid_grid.hip: the `sigmoid`
// kernel's O(n^2) pair loops spread over the chip): the argument block, the tape / scratch layouts and the per-row helpers.
#pragma once
#include "dif_common.h"

namespace tiny {

constexpr int kMaxLayers = 8;
constexpr int kMaxIn = 64;
constexpr int kMaxOut = 8;
constexpr int kMaxNodes = 4096;
constexpr int kMaxEdges = 65535;

struct LayerPtrs {
    const float *wk, *bk, *wq, *bq, *wv, *bv, *lnw, *lnb;
};
struct LayerGrads {
    float *wk, *bk, *wq, *bq, *wv, *bv, *lnw, *lnb;
};

struct TinyArgs {
    int n, f_in, d, c, layers, sigmoid, use_bn, residual, use_weight, use_graph, use_source, training;
    float alpha, a_s, g_s, p_drop, eps;
    const float* x;
    int64_t ldx;
    const float *w0, *b0, *ln0w, *ln0b, *wo, *bo;
    LayerPtrs lp[kMaxLayers];
    const int* rowptr;      // forward: destination-major CSR; backward: its transpose (rows = sources, entries = destinations)
    const int* nbr;
    const float* val;
    const float* rnd;       // [(layers + 1), n, d] uniform [0, 1) (training with dropout) or null
    float* tape;
    float* y;               // forward: [n, c]
    // backward only
    const float* gy;        // [n, c]
    float *gw0, *gb0, *gln0w, *gln0b, *gwo, *gbo;
    LayerGrads lg[kMaxLayers];
    float* dx;              // [n, f_in] or null
    float* scratch;
};

// grid kernels: 64 nodes per workgroup; the keys of a layer's pair sweep split over K workgroups per node block so that the
// launch is ~2,000 waves, a wave keeping >= 8 keys
constexpr int kNodes = 64;
constexpr int kMaxBlocks = kMaxNodes / kNodes;
// `simple` on the grid: a node block leaves its share of a layer's sums over nodes (<= 96 values, double) per set
__host__ __device__ inline size_t block_sums_floats(int n) { return 2 * 2 * static_cast<size_t>((n + kNodes - 1) / kNodes) * 96; }
__host__ __device__ inline int key_splits(int n) {
    if (n <= 512) return 1;                    // one launch per layer: 8 waves x <= 64 keys
    const int G = (n + kNodes - 1) / kNodes;
    int K = (256 + G - 1) / G;
    if (K > n / 64) K = n / 64;
    if (K > 16) K = 16;
    return K < 1 ? 1 : K;
}

// ---- tape layout (floats); DP = padded hidden width (4 or 8) ---------------------------------------------------------
//   H   [(L+1)][n][DP]   layer inputs (post LayerNorm / ReLU / dropout)
//   Z   [(L+1)][n][DP]   pre-LayerNorm values (Z[0]: input Linear; Z[l+1]: layer l after the residual)
//   ATT [L][n][DP]       attention output per layer (sigmoid backward needs it)
//   DEN [L][n]           sigmoid row sums
//   SUM [L][96]          simple kernel: K^T V [DP*DP], ksum [DP], vsum [DP], sum q^2, sum k^2
//   QKV [2][3][n][DP]    forward scratch (the grid kernels alternate between the two sets from layer to layer: a workgroup
//                        writes layer l + 1's rows while others still read layer l's)
//   ATTLO [L][n][DP]     grid kernels: att = ATT + ATTLO as the float64 quotient gave it (the backward's DL wants it unrounded)
__host__ __device__ inline size_t tape_floats(int n, int DP, int L) {
    size_t f = static_cast<size_t>(n) * DP * (2 * (L + 1) + 2 * L + 6) + static_cast<size_t>(L) * n + static_cast<size_t>(L) * 96;
    f += (-f) & 3;
    // `sigmoid`: the key splits' partial sums [K][n][DP + 1];  `simple`: the node blocks' partial sums [2 sets][G][96]  (double)
    const size_t pairs = 2 * static_cast<size_t>(key_splits(n)) * n * (DP + 1), blocks = block_sums_floats(n);
    return f + (pairs > blocks ? pairs : blocks);
}
constexpr int kBwdSlots = 14;
// one workgroup: 14 [n][DP] slots + 3 [n] arrays; the grid kernels (tiny_sigmoid_grid.hip): 5 slots per layer kept for the
// sums over nodes at the end, 2 x 5 alternating between layers, 5 single ones, 4 [n] arrays
__host__ __device__ inline size_t scratch_floats(int n, int DP, int L) {
    const size_t nd = static_cast<size_t>(n) * DP;
    const size_t one = nd * kBwdSlots + 3 * static_cast<size_t>(n);
    size_t grid = nd * (5 * static_cast<size_t>(L) + 15) + 4 * static_cast<size_t>(n);
    grid += (-grid) & 3;
    const size_t pairs = 2 * static_cast<size_t>(key_splits(n)) * n * 3 * DP, blocks = block_sums_floats(n);
    grid += pairs > blocks ? pairs : blocks;
    return one > grid ? one : grid;
}

template <int DP>
struct Tape {
    float *H, *Z, *ATT, *DEN, *SUM, *Q, *K, *V;
    int layers;
    __device__ Tape(float* base, int n, int L) : layers(L) {
        const size_t nd = static_cast<size_t>(n) * DP;
        H = base;
        Z = H + nd * (L + 1);
        ATT = Z + nd * (L + 1);
        DEN = ATT + nd * L;
        SUM = DEN + static_cast<size_t>(L) * n;
        Q = SUM + static_cast<size_t>(L) * 96;
        K = Q + nd;
        V = K + nd;
    }
    __device__ float* att_lo() const { return Q + 6 * (K - Q); }
    __device__ float* tail() const {                       // 16-byte aligned (the tape is)
        float* e = Q + (6 + static_cast<size_t>(layers)) * (K - Q);
        return e + ((-(e - H)) & 3);
    }
    __device__ double* partials() const { return reinterpret_cast<double*>(tail()); }
    __device__ float* qkv(int set) const { return Q + static_cast<size_t>(set) * 3 * (K - Q); }    // [3][n][DP] of set 0 / 1
};

// out[m * C + c] = scale * sum_i A[i * lda + m] * (B ? B[i * ldb + c] : 1)    for m < M, c < C
// O = M * C outputs; blockDim / O node-chunks per output (at most 64), partials in sPart (LDS, blockDim floats... doubles),
// added in chunk order.  Ends with a __syncthreads(); `out` may be LDS or global.  Every thread of the block must call.
static __device__ void outer_sum(const float* __restrict__ A, int64_t lda, int M, const float* __restrict__ B, int64_t ldb, int C,
                          int n, float scale, float* out, double* sPart) {
    const int T = blockDim.x, t = threadIdx.x;
    const int O = M * C;
    for (int base = 0; base < O; base += T) {
        const int Ob = min(O - base, T);
        int chunks = T / Ob;
        if (chunks > 64) chunks = 64;
        const int len = (n + chunks - 1) / chunks;
        const int o = t % Ob, ch = t / Ob;
        if (ch < chunks) {
            const int m = (base + o) / C, c = (base + o) % C;
            const int i0 = ch * len, i1 = min(n, i0 + len);
            double acc = 0.0;
            if (B) {
                for (int i = i0; i < i1; ++i) acc += static_cast<double>(A[i * lda + m]) * static_cast<double>(B[i * ldb + c]);
            } else {
                for (int i = i0; i < i1; ++i) acc += static_cast<double>(A[i * lda + m]);
            }
            sPart[ch * Ob + o] = acc;
        }
        __syncthreads();
        if (t < Ob) {
            double s = 0.0;
            for (int k = 0; k < chunks; ++k) s += sPart[k * Ob + t];
            out[base + t] = static_cast<float>(s * static_cast<double>(scale));
        }
        __syncthreads();
    }
}

// LayerNorm statistics of one row (torch.nn.LayerNorm: biased variance, eps inside the root) over its d valid columns
template <int DP>
__device__ __forceinline__ void ln_stats(const float (&z)[DP], int d, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < DP; ++k) s += (k < d) ? z[k] : 0.f;
    mean = s / static_cast<float>(d);
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
        const float c = (k < d) ? z[k] - mean : 0.f;
        v += c * c;
    }
    rstd = 1.0f / sqrtf(v / static_cast<float>(d) + eps);
}

template <int DP>
__device__ __forceinline__ void load_row(const float* p, float (&r)[DP]) {
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int k = 0; k < DP / 4; ++k) {
        const float4 v = q[k];
        r[4 * k] = v.x; r[4 * k + 1] = v.y; r[4 * k + 2] = v.z; r[4 * k + 3] = v.w;
    }
}
template <int DP>
__device__ __forceinline__ void store_row(float* p, const float (&r)[DP]) {
    float4* q = reinterpret_cast<float4*>(p);
#pragma unroll
    for (int k = 0; k < DP / 4; ++k) q[k] = make_float4(r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3]);
}

// weights of one layer in LDS, zero-padded to DP x DP (row m = output feature m)
template <int DP>
struct LayerW {
    float wq[DP * DP], wk[DP * DP], wv[DP * DP], bq[DP], bk[DP], bv[DP], lnw[DP], lnb[DP];
};

template <int DP>
__device__ void load_layer(LayerW<DP>& s, const LayerPtrs& p, int d, bool use_weight, bool use_bn) {
    for (int k = threadIdx.x; k < DP * DP; k += blockDim.x) {
        const int m = k / DP, c = k % DP;
        const bool in = m < d && c < d;
        s.wq[k] = in ? p.wq[m * d + c] : 0.f;
        s.wk[k] = in ? p.wk[m * d + c] : 0.f;
        s.wv[k] = (in && use_weight) ? p.wv[m * d + c] : 0.f;
    }
    for (int k = threadIdx.x; k < DP; k += blockDim.x) {
        const bool in = k < d;
        s.bq[k] = in ? p.bq[k] : 0.f;
        s.bk[k] = in ? p.bk[k] : 0.f;
        s.bv[k] = (in && use_weight) ? p.bv[k] : 0.f;
        s.lnw[k] = (in && use_bn) ? p.lnw[k] : 0.f;
        s.lnb[k] = (in && use_bn) ? p.lnb[k] : 0.f;
    }
}

template <int DP>
__device__ __forceinline__ void matvec(const float* W, const float* b, const float (&h)[DP], float (&o)[DP]) {
#pragma unroll
    for (int m = 0; m < DP; ++m) {
        float acc = b ? b[m] : 0.f;
#pragma unroll
        for (int c = 0; c < DP; ++c) acc += W[m * DP + c] * h[c];
        o[m] = acc;
    }
}
// o[c] += sum_m W[m][c] g[m]
template <int DP>
__device__ __forceinline__ void matvec_t_add(const float* W, const float (&g)[DP], float (&o)[DP]) {
#pragma unroll
    for (int c = 0; c < DP; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int m = 0; m < DP; ++m) acc += W[m * DP + c] * g[m];
        o[c] += acc;
    }
}

__device__ __forceinline__ float sigmoidf(float v) { return 1.0f / (1.0f + expf(-v)); }

// dropout of one row with the caller's uniforms: keep iff r >= p, scaled by 1 / (1 - p) (torch.nn.functional.dropout)
template <int DP>
__device__ __forceinline__ void dropout_row(float (&h)[DP], const float* rnd, int64_t at, int d, float p) {
    const float keep = 1.0f / (1.0f - p);
#pragma unroll
    for (int k = 0; k < DP; ++k)
        if (k < d) h[k] = (rnd[at + k] >= p) ? h[k] * keep : 0.f;
}

// tiny_sigmoid_grid.hip / tiny_simple_grid.hip: `a` filled as for the one-workgroup kernels; forward: L + 1 launches, backward: L + 2
int grid_sigmoid_forward(const TinyArgs& a, hipStream_t st);
int grid_sigmoid_backward(const TinyArgs& a, hipStream_t st);
int grid_simple_forward(const TinyArgs& a, hipStream_t st);
int grid_simple_backward(const TinyArgs& a, hipStream_t st);

}  // namespace tiny

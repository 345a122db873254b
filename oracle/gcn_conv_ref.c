/* CPU oracle for gcn_conv -- TEST INFRASTRUCTURE ONLY (see oracle/difformer_oracle.py).
 *
 * Plain-C restatement of `node classification/difformer.py:63-79` for graphs too
 * large for the numpy scatter-add (ogbn-proteins scale, ~8e7 edges):
 *   :66     d = in-degree counted over `col`
 *   :67-73  value_e = w_e * sqrt(1/d[col_e]) * sqrt(1/d[row_e])
 *   :74     non-finite value -> 0
 *   :75-78  out[col_e] += value_e * x[row_e]   (SparseTensor(row=col, col=row) + sum-SpMM,
 *           torch_sparse 0.6.10 semantics; duplicates accumulate; same adjacency per head)
 * The edges are bucketed by destination with a stable counting sort (what the
 * SparseTensor constructor's sort amounts to) and the rows are then summed in
 * parallel, one destination row per OpenMP task -- so the result does not
 * depend on the thread count.
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC gcn_conv_ref.c -o _build/liboracle_gcn.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DEFINE_GCN(NAME, T)                                                          \
int NAME(const T* x, const int64_t* edge_index, const T* edge_weight, int64_t n,          \
         int64_t e, int64_t f, T* out, int threads) {                                     \
    const int64_t* row = edge_index;      /* source  (difformer.py:65) */                 \
    const int64_t* col = edge_index + e;  /* destination */                               \
    int64_t* ptr = (int64_t*)calloc((size_t)n + 2, sizeof(int64_t));                      \
    int64_t* perm = (int64_t*)malloc((size_t)(e > 0 ? e : 1) * sizeof(int64_t));          \
    if (!ptr || !perm) { free(ptr); free(perm); return -1; }                              \
    for (int64_t i = 0; i < e; ++i) {                                                     \
        if (row[i] < 0 || row[i] >= n || col[i] < 0 || col[i] >= n) {                     \
            free(ptr); free(perm); return -2; }                                           \
        ptr[col[i] + 2]++;                                                                \
    }                                                                                     \
    /* ptr[c+2] = deg(c); shift-scan so that ptr[c+1] becomes the fill cursor of row c */ \
    for (int64_t c = 0; c < n; ++c) ptr[c + 2] += ptr[c + 1];                             \
    for (int64_t i = 0; i < e; ++i) perm[ptr[col[i] + 1]++] = i;                          \
    /* now ptr[c] .. ptr[c+1] delimit destination row c */                                \
    if (threads < 1) threads = 1;                                                         \
    _Pragma("omp parallel for schedule(dynamic, 64) num_threads(threads)")                \
    for (int64_t c = 0; c < n; ++c) {                                                     \
        T* o = out + c * f;                                                               \
        for (int64_t j = 0; j < f; ++j) o[j] = (T)0;                                      \
        /* degree and d^-1/2 are float32 whatever T is (`.float()`, difformer.py:66) */   \
        const float dn_in = sqrtf(1.0f / (float)(ptr[c + 1] - ptr[c]));                   \
        for (int64_t k = ptr[c]; k < ptr[c + 1]; ++k) {                                   \
            const int64_t i = perm[k];                                                    \
            const int64_t r = row[i];                                                     \
            const float dn_out = sqrtf(1.0f / (float)(ptr[r + 1] - ptr[r]));              \
            T v = edge_weight ? edge_weight[i] * (T)dn_in * (T)dn_out                     \
                              : (T)(dn_in * dn_out);                                      \
            if (!isfinite(v)) v = (T)0;                                                   \
            const T* xr = x + r * f;                                                      \
            for (int64_t j = 0; j < f; ++j) o[j] += v * xr[j];                            \
        }                                                                                 \
    }                                                                                     \
    free(ptr); free(perm);                                                                \
    return 0;                                                                             \
}

DEFINE_GCN(oracle_gcn_conv_f32, float)
DEFINE_GCN(oracle_gcn_conv_f64, double)

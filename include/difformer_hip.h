/*
 * difformer_hip.h -- C ABI of libdifformer_hip.so, the MI355X (gfx950) hot path of the
 * DIFFormer propagation layer.
 *
 * The reference (qitianwu/DIFFormer) has no FFI of its own: its boundary for this path is
 * the Python module `difformer` (`node classification/parse.py:2,6-7`).  The Python host in
 * `difformer_amd/difformer.py` keeps that module's names and signatures and binds the entry
 * points below through `ctypes`; each entry point names the reference lines it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (row-major, 16-byte aligned, leading dimensions given in ELEMENTS), including all
 *     workspaces -- the library never allocates, frees or synchronises;
 *   - work is enqueued on the `stream` argument only (a hipStream_t passed as void*; NULL =
 *     the null stream), so every call is stream-ordered and graph-capturable;
 *   - return 0 on success, a negative DIF_E_* code for a rejected argument, or a positive
 *     hipError_t for a runtime failure; `dif_last_error()` returns a thread-local message.
 *     No C++ exception crosses this boundary;
 *   - functions keep no state between calls and are re-entrant, with ONE exception: dif_set_exact_fp32() flips a
 *     process-global switch (seeded from DIFFORMER_EXACT_FP32) that every launcher reads when it picks the matrix core of
 *     its products.  Set it before the first call or between launches; flipping it from one host thread while another is
 *     inside a dif_* call gives that call either setting (workspace sizes are computed for both).  One process per GPU
 *     for multi-GPU.
 */
#ifndef DIFFORMER_HIP_H
#define DIFFORMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIF_ABI_VERSION 2   /* 2: dif_csr_build's status is int32[2] (status[1] = longest row): a caller that allocates one int must be rebuilt */

#define DIF_E_BADARG   (-1)  /* null pointer, non-positive size, misaligned pointer */
#define DIF_E_SHAPE    (-2)  /* shape the kernels do not cover */
#define DIF_E_WORKSPACE (-3) /* workspace smaller than dif_*_workspace_bytes() */
#define DIF_E_RANGE    (-4)  /* size exceeds an internal 32-bit index */

typedef void* dif_stream_t; /* hipStream_t */

int dif_version(void);
const char* dif_last_error(void);
/* Every product on the fp32 matrix core (on != 0) or the default choice per kernel (split-bfloat16 operands on the bf16
 * core where the entry points below say so); seeded from DIFFORMER_EXACT_FP32, returns the previous setting.  The reference
 * has no such switch: its products are whatever torch.matmul does on the device (difformer.py:23-58). */
int dif_set_exact_fp32(int on);

/* ---------------------------------------------------------------------------------------
 * a1  full_attention_conv(qs, ks, vs, 'simple')     node classification/difformer.py:18-39
 *
 * Stage 1 (`reduce`): one pass over K, V (and Q for its norm) producing the per-head global
 * sums the closed form needs.  `reduced` layout (float32, dif_simple_reduced_len() values):
 *     [ KtV  H*M*D ][ ksum  H*M ][ vsum  H*D ][ sum(q*q) ][ sum(k*k) ]
 * all UN-normalised (the Frobenius scale 1/(|Q||K|) of difformer.py:20-21 is applied in
 * stage 2).  In a row-sharded run every rank reduces its own rows and the host all-reduces
 * (sum) this buffer across ranks before stage 2.
 * Stage 2 (`apply`):  out[n,h,:] = (s*q[n,h,:]*KtV[h] + vsum[h]) / (s*q[n,h,:]*ksum[h] + n_global)
 * with s = 1/(sqrt(sum q*q)*sqrt(sum k*k)); n_global is N of difformer.py:22,38.
 * q,k are [n_rows,H,M], v and out [n_rows,H,D]; ld* = elements between consecutive rows.
 * Products: float32, ONE head of 65..128 channels, >= 4,096 rows, 16-byte aligned rows: both stages contract on
 * split-bfloat16 operands (hi + lo, bf16 matrix core; ~4e-6 of the float64 result); every other shape, and
 * dif_set_exact_fp32(1), on the fp32 matrix core.
 * ------------------------------------------------------------------------------------- */
size_t dif_simple_reduced_len(int H, int M, int D);
size_t dif_simple_workspace_bytes(int64_t n_rows, int H, int M, int D);
int dif_simple_reduce_f32(const float* q, int64_t ldq, const float* k, int64_t ldk,
                          const float* v, int64_t ldv, int64_t n_rows, int H, int M, int D,
                          float* reduced, void* workspace, size_t workspace_bytes,
                          dif_stream_t stream);
int dif_simple_apply_f32(const float* q, int64_t ldq, const float* reduced, int64_t n_rows,
                         int64_t n_global, int H, int M, int D, float* out, int64_t ldo,
                         dif_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a4+a1 fused  DIFFormerConv.forward :115-118 (Wq/Wk/Wv Linear) + stage 1 of the simple kernel
 * (:20-34) in one pass over x [n_rows, C_in]:  q = x Wq^T + bq and v = x Wv^T + bv are written
 * ([n_rows, H*D]), k = x Wk^T + bk stays in registers, and `reduced` receives the same record as
 * dif_simple_reduce_f32 (M = D).  W* are the nn.Linear weights [H*D, C_in] row-major, b* [H*D].
 * Covers C_in <= 64 and D <= 64 (DIF_E_SHAPE otherwise: run the Linear layers + dif_simple_reduce_f32).
 * ------------------------------------------------------------------------------------- */
size_t dif_project_reduce_workspace_bytes(int64_t n_rows, int H, int D);
int dif_project_reduce_f32(const float* x, int64_t ldx, int64_t n_rows, int C_in,
                           const float* Wq, const float* bq, const float* Wk, const float* bk,
                           const float* Wv, const float* bv, int H, int D,
                           float* q_out, int64_t ldq, float* v_out, int64_t ldv,
                           float* reduced, void* workspace, size_t workspace_bytes,
                           dif_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a2  full_attention_conv(qs, ks, vs, 'sigmoid')    node classification/difformer.py:45-56
 *     out[n,h,:] = sum_l sigmoid(q[n,h,:].k[l,h,:]) v[l,h,:] / sum_l sigmoid(q[n,h,:].k[l,h,:])
 * q [N,H,M], k [L,H,M], v [L,H,D], out [N,H,D]; N may differ from L.  The [N,L,H] score
 * tensor is never materialised.
 * Products: float32 heads of at most 64 channels with 16-byte aligned rows contract on split-bfloat16 operands
 * (hi + lo, bf16 matrix core; ~7e-6 of the float64 result) in dif_sigmoid_attn_f32; dif_set_exact_fp32(1) keeps the
 * fp32 matrix core.  dif_sigmoid_attn_fwd_f32 / dif_sigmoid_attn_bwd_f32 (training) run heads up to 64 channels on the fp32
 * chain (DIFFORMER_SIGMOID_BWD_SPLIT=1 opts the backward in): gradients made of cancelling rows amplify the operands' error.
 * float32 heads of 65 .. 512 channels (image and text/run.sh:17,35,54: hidden 300 / 400, N ~ 15,000) -- all three entry
 * points: csrc/sigmoid_wide.hip.  Every operand goes in as two bfloat16 PLANES (x = hi + lo) packed in MFMA fragment order by
 * a pre-pass into the workspace, the streamed side moves global -> LDS by LDS-DMA, each wave keeps 16 query (or key) rows'
 * fragments and ALL output columns' accumulators in registers, scores are formed once per (query, key) pair; values enter
 * centred (v - column mean: out_n is a convex combination of the value rows, so the centre passes through exactly and the
 * products carry only the rows' spread).  Forward ~1e-6 .. 1e-5 of the float64 result, gradients ~3e-5 of each tensor's largest
 * entry.  Under dif_set_exact_fp32(1) these widths run the generic fp32 forward kernel and dif_sigmoid_attn_bwd_f32 returns
 * DIF_E_SHAPE (the host then differentiates with tensor operations).  Heads of 33 .. 64 channels take the same plane kernels in
 * dif_sigmoid_attn_f32 (inference) from 2^25 (query, key) pairs; the training entry points never do.  Workspaces: packed planes (8 bytes per element and
 * orientation) + per-split partial sums, dif_sigmoid_workspace_bytes / dif_sigmoid_bwd_workspace_bytes; 256-byte aligned use.
 * ------------------------------------------------------------------------------------- */
size_t dif_sigmoid_workspace_bytes(int64_t N, int64_t L, int H, int M, int D);
int dif_sigmoid_attn_f32(const float* q, int64_t ldq, const float* k, int64_t ldk,
                         const float* v, int64_t ldv, int64_t N, int64_t L, int H, int M, int D,
                         float* out, int64_t ldo, void* workspace, size_t workspace_bytes,
                         dif_stream_t stream);
/* f3 (training, main.py:117-131): dif_sigmoid_attn_fwd_f32 = dif_sigmoid_attn_f32 that also leaves the row sums
 * den float[N,H] = sum_l sigmoid(q_n . k_l); dif_sigmoid_attn_bwd_f32 turns g = dL/dout [N,H,D] into dq [N,H,M],
 * dk [L,H,M], dv [L,H,D] (M, D <= 512; DIF_E_SHAPE otherwise) by recomputing sigma tile by tile -- the [N,L,H] tensors are
 * never materialised:  delta_n = g_n . out_n,  dV_l = sum_n (P_nl / den_n) g_n,
 * dS_nl = (g_n . v_l - delta_n) P_nl (1 - P_nl) / den_n,  dQ = dS K,  dK = dS^T Q.  Deterministic.
 * workspace: dif_sigmoid_bwd_workspace_bytes(N, L, H, M, D), 16-byte aligned. */
int dif_sigmoid_attn_fwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                             int64_t N, int64_t L, int H, int M, int D, float* out, int64_t ldo, float* den,
                             void* workspace, size_t workspace_bytes, dif_stream_t stream);
size_t dif_sigmoid_bwd_workspace_bytes(int64_t N, int64_t L, int H, int M, int D);
int dif_sigmoid_attn_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                             const float* out, int64_t ldo, const float* den, const float* g, int64_t ldg, int64_t N,
                             int64_t L, int H, int M, int D, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv,
                             int64_t lddv, void* workspace, size_t workspace_bytes, dif_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * f4  TransConv.full_attention over a batch of graphs    physical particle/difformer-v2.py:71-137
 *     The batch stores its graphs back to back: graph b owns rows [graph_ptr[b], graph_ptr[b+1]).
 *     Nothing is padded (the reference pads to [B, max_node, H, D], :8-27, :87-91).
 *
 * 'simple' (:80-111):  out_i = (s q_i KtV_b + vsum_b) / (s q_i.ksum_b + n_b),  s = 1/(|Q|_F |K|_F) over the whole
 *     batch, the sums over graph b only.  M <= 256 per head (DIF_E_SHAPE beyond).  workspace: partial norms.
 * 'sigmoid' (:113-135): the node at position p of a graph attends the nodes at the SAME position p of all graphs
 *     (einsum "abcd,ebcd->aebc"); shorter graphs count sigma(0) = 0.5 in the denominator, + 1e-9.
 *     ranked_first[r] = first row of the r-th largest graph (any order among equal sizes), pos_count[p] = number of
 *     graphs with more than p nodes, p < max_nodes = size of the largest graph.
 * ------------------------------------------------------------------------------------- */
size_t dif_batched_simple_workspace_bytes(void);
int dif_batched_simple_attn_f32(const float* q, int64_t ldq, const float* k, int64_t ldk,
                                const float* v, int64_t ldv, const int32_t* graph_ptr, int n_graphs,
                                int64_t n_rows, int H, int M, int D, float* out, int64_t ldo,
                                void* workspace, size_t workspace_bytes, dif_stream_t stream);
/* Training (csrc/batched_attn.hip):  dif_batched_simple_attn_fwd_f32 also writes den float[n_rows * H] = s q_i.ksum_b + n_b
 * and leaves {|Q|^2, |K|^2} in the LAST two floats of `workspace`.  The backward is three launches of
 *   dif_batched_simple_raw_f32:  out_i = s a_i (sum_{l in graph(i)} b_l (x) c_l) + coef_i sum_l w_l c_l
 *       a, b [n_rows, H, M], c / out [n_rows, H, D];  coef_i = (rs ? rs[i, h] : 1) (vs_is_s ? s : 1),  w_l = vw ? vw[l, h] : 1;
 *       s from sumsq = float[2] {|Q|^2, |K|^2} of the forward;  rs, vw float[n_rows * H] or NULL
 * with gn = g / den, gd = -(g . out) / den:   dq = raw(gn, v, k; rs = gd, s)  - (T / |Q|^2) q,  T = sum q . dq_direct
 *                                             dk = raw(v, gn, q; vw = gd, s) - (T / |K|^2) k
 *                                             dv = raw(k, q, gn)  (vsum term unscaled). */
int dif_batched_simple_attn_fwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                    const int32_t* graph_ptr, int n_graphs, int64_t n_rows, int H, int M, int D, float* out,
                                    int64_t ldo, float* den, void* workspace, size_t workspace_bytes, dif_stream_t stream);
int dif_batched_simple_raw_f32(const float* a, int64_t lda, const float* b, int64_t ldb, const float* c, int64_t ldc,
                               const int32_t* graph_ptr, int n_graphs, int64_t n_rows, int H, int M, int D,
                               const float* sumsq, const float* rs, const float* vw, int vs_is_s, float* out, int64_t ldo,
                               dif_stream_t stream);
int dif_batched_sigmoid_attn_f32(const float* q, int64_t ldq, const float* k, int64_t ldk,
                                 const float* v, int64_t ldv, const int32_t* ranked_first,
                                 const int32_t* pos_count, int n_graphs, int max_nodes, int H, int M, int D,
                                 float* out, int64_t ldo, dif_stream_t stream);
/* Training (physical particle/main.py:89-93 differentiates through difformer-v2.py:113-135): the forward that also leaves the
 * full denominators den float[N, H] (row sum + 0.5 per padded graph + 1e-9; D <= 64), and the backward that takes them:
 * dq, dk [N,H,M], dv [N,H,D] with sigma recomputed tile by tile per position group (M, D <= 64, DIF_E_SHAPE beyond; the host
 * then re-derives the gradient with tensor ops on the padded batch).  Padded graphs add constants to the denominator only, so
 * the arithmetic is dif_sigmoid_attn_bwd_f32's.  workspace: dif_batched_sigmoid_bwd_workspace_bytes(N, H), 16-byte aligned. */
int dif_batched_sigmoid_attn_fwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                     const int32_t* ranked_first, const int32_t* pos_count, int n_graphs, int max_nodes, int H,
                                     int M, int D, float* out, int64_t ldo, float* den, dif_stream_t stream);
size_t dif_batched_sigmoid_bwd_workspace_bytes(int64_t N, int H);
int dif_batched_sigmoid_attn_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                     const float* out, int64_t ldo, const float* den, const float* g, int64_t ldg,
                                     const int32_t* ranked_first, const int32_t* pos_count, int n_graphs, int max_nodes,
                                     int64_t N, int H, int M, int D, float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv,
                                     int64_t lddv, void* workspace, size_t workspace_bytes, dif_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a3  gcn_conv(x, edge_index, edge_weight)          node classification/difformer.py:63-79
 *     (replaces torch_geometric.utils.degree :66, torch_sparse.SparseTensor :75 and
 *      torch_sparse.matmul :77)
 *
 * dif_csr_build: COO edge_index [2,E] int64 (row = source, col = destination) -> CSR over
 * DESTINATION rows: rowptr [N+1] int32, src [E] int32 (source node of each entry) and
 * val [E] float32 = w_e * deg[col_e]^-1/2 * deg[row_e]^-1/2 with non-finite -> 0
 * (:66-74; deg = in-degree over `col`).  Entries of a row are ordered by (source block, edge
 * id) -- a stable sort, so the SpMM result is run-to-run deterministic.  n_blocks >= 1 cuts the
 * sources into contiguous blocks of ceil(N/n_blocks) nodes; with n_blocks > 1, blkptr
 * [(n_blocks+1) x N] int32 (block-major) receives the start of every (row, block) group and the
 * SpMM sweeps the blocks in order so the gathered slice of x stays L2-resident (choose
 * n_blocks ~ N*F*4 / 2.5 MiB; 1 = plain CSR).  block_rows = 0 means ceil(N/n_blocks); a row-sharded run passes the
 * rows per rank divided by a small integer so that block boundaries coincide with rank boundaries.  status (device int32 [2]): status[0] is set non-zero if any
 * index is outside [0,N); status[1] receives the length of the LONGEST row (the host reads both with the one
 * synchronisation a build has anyway: kernel selection then never depends on when a statistic arrives).  transpose = 1 files every entry under its SOURCE row instead (same values,
 * `src` then holds the destination): the SpMM over that CSR is the adjoint A_hat^T g, i.e. the gradient
 * of gcn_conv with respect to x (loss.backward() in main.py:130).
 * dif_gcn_spmm_f32: out[r, :] = gcn_scale * sum_{e in row r} val_e * x[src_e, :]
 *                                (+ attn_scale * attn[r, :] when attn != NULL)
 * for r in [row_begin, row_begin + n_rows): the adjacency product of :75-78 over all H*D
 * feature columns at once, with the `attention + gcn` / convex mix of :130-134 folded in.
 * `x` holds ALL n_nodes source rows (the full graph); out/attn hold only the n_rows local
 * rows.  n_nodes / nnz are the CSR extents (rowptr has n_nodes+1 entries, rowptr[n_nodes] = nnz);
 * they pick the row->wave mapping (wave per row for dense rows, lane group per row otherwise).
 * ------------------------------------------------------------------------------------- */
size_t dif_csr_workspace_bytes(int64_t E, int64_t N, int n_blocks);
int dif_csr_build(const int64_t* edge_index, int64_t E, int64_t N, const float* edge_weight,
                  int n_blocks, int64_t block_rows, int transpose, int32_t* rowptr, int32_t* blkptr, int32_t* src, float* val,
                  int32_t* status, void* workspace, size_t workspace_bytes, dif_stream_t stream);
int dif_gcn_spmm_f32(const int32_t* rowptr, const int32_t* blkptr, int n_blocks,
                     const int32_t* src, const float* val, int64_t n_nodes, int64_t nnz,
                     const float* x, int64_t ldx, int64_t row_begin, int64_t n_rows, int F,
                     const float* attn, int64_t lda, float attn_scale, float gcn_scale,
                     const int32_t* row_order, int64_t n_split_rows, float* out, int64_t ldo,
                     dif_stream_t stream);
/* row_order (optional; NULL with n_split_rows = 0 = natural order): the rows of [row_begin, row_begin + n_rows) by
 * descending degree, as produced by dif_row_order (indices inside the shard).  It only changes which rows the blocked
 * kernel walks side by side and in which wave (similar degrees together, every wave a share of each degree stratum):
 * the load balance on graphs with skewed degrees.  The first n_split_rows rows of the order (dif_row_order's stats[0]:
 * degree > 4x the mean) are each split over the wave's four lane groups, whose partial rows are added in a fixed order;
 * every other row is summed exactly as without the order.  Ignored by the unblocked kernels.
 * dif_row_order: order int32 [n_rows]; stats int32 [2] (device) = {rows with degree > 4x mean, max degree}. */
/* dif_gcn_edge_weight_grad_f32: gradient of gcn_conv with respect to edge_weight, in the ORIGINAL edge order
 * (the reference's value = edge_weight * d_in * d_out, :73, feeds torch_sparse.matmul, which autograd differentiates in
 * the values; reached by loss.backward(), main.py:130, whenever edge_weight requires a gradient):
 *     dw[e] = [value_e finite] * scale * <g[col_e, :], x[row_e, :]> * deg[row_e]^-1/2 * deg[col_e]^-1/2
 * with deg from `rowptr` of the destination-row CSR (full graph).  An edge leaving a node without incoming entries has
 * deg^-1/2 = inf and the reference's gradient is 0 * inf = NaN; so it is here.  g: upstream gradient of the aggregation's
 * output rows [N, F]; x: its input rows [N, F]; scale = gcn_scale of the combine (:130-134). */
int dif_gcn_edge_weight_grad_f32(const int64_t* edge_index, int64_t E, int64_t N, const float* edge_weight,
                                 const int32_t* rowptr, const float* g, int64_t ldg, const float* x, int64_t ldx, int F,
                                 float scale, float* dw, dif_stream_t stream);
size_t dif_row_order_workspace_bytes(int64_t n_rows);
int dif_row_order(const int32_t* rowptr, int64_t row_begin, int64_t n_rows, int32_t* order, int32_t* stats,
                  void* workspace, size_t workspace_bytes, dif_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a1 + a4 + a5 tail in closed form (csrc/simple_layer.hip): the `simple` kernel with query_input == source_input == x,
 * one head, C_in <= 64 (C_in % 4 == 0), D <= 64, inference.   node classification/difformer.py:18-39, :115-140, :200-203
 *
 * Stage 1 of the simple kernel only needs G = X^T X and sx = sum_rows x (K^T V = Wk G Wv^T + ..., |Q|^2 = tr(Wq G Wq^T)
 * + ...), stage 2 is linear in x (num = x Mn + cn, den = x.u + cd) and gcn_conv is linear too (A_hat (x Wv^T + 1 bv^T)
 * = (A_hat x) Wv^T + (A_hat 1) bv^T), so q, k, v and the attention output never reach memory:
 *   dif_gram_f32           record float[C*C + C + 2] = {G, sx}; with ys != NULL also the slice-major copy of x scaled by
 *                          deg^-1/2 that dif_sliced_spmm_f32 reads (plan from dif_sliced_plan(n_rows, n_rows, C)).  The
 *                          record is the only thing a row-sharded run has to all-reduce.
 *   dif_simple_coeffs_f32  coef float[dif_simple_coeffs_len(C, D)] = {MnT [D x C], cn [D], u [C], cd, s, |Q|^2, |K|^2} with
 *                          attn_scale (1 - graph_weight, or 1) folded into MnT / cn; n_global = number of rows of the
 *                          whole graph (the +N of :22,:38).  Wv = bv = NULL: use_weight = False (v = x, C == D, :120).
 *   dif_simple_layer_f32   out = LN(alpha * (num/den + ax Wv^T + gcn_scale * row_sums * bv [+ x0]) + (1 - alpha) * x) with
 *                          ax = gcn_scale * A_hat x from the SpMM run on x (NULL: use_graph = False), row_sums = A_hat 1
 *                          (NULL when bv is not needed); residual = 0 skips the alpha mix, ln_weight = NULL the LayerNorm.
 *                          next_ys != NULL: the pass also writes the slice-major copy of `out` scaled by deg^-1/2 (what
 *                          dif_gram_f32 would write for the next layer), 256 contiguous bytes per lane group.
 *                          next_record != NULL: it leaves dif_gram_f32's record of `out` as well (workspace >=
 *                          dif_gram_workspace_bytes(n_rows, D)); measured slower than a separate dif_gram_f32 at C4.
 * ------------------------------------------------------------------------------------- */
size_t dif_gram_workspace_bytes(int64_t n_rows, int C);
int dif_gram_f32(const float* x, int64_t ldx, int64_t n_rows, int C, const int32_t* rowptr, const int32_t* plan,
                 float* ys, float* record, void* workspace, size_t workspace_bytes, dif_stream_t stream);
size_t dif_simple_coeffs_len(int C, int D);
int dif_simple_coeffs_f32(const float* record, int64_t n_global, int C, int D, const float* Wq, const float* bq,
                          const float* Wk, const float* bk, const float* Wv, const float* bv, float attn_scale,
                          float* coef, dif_stream_t stream);
/* dif_gram_f32 (without the slice-major copy) + dif_simple_coeffs_f32 of one layer input in one call: record (nullable up to
 * 24,576 rows) and coef from x.  Up to 48 partial records of the Gram pass are summed inside the coefficient kernel (ascending
 * chunk order): two launches instead of three on the small graphs of node classification/run.sh (Cora: 6 partials).
 * workspace: dif_gram_workspace_bytes(n_rows, C). */
int dif_gram_coeffs_f32(const float* x, int64_t ldx, int64_t n_rows, int C, int D, const float* Wq, const float* bq,
                        const float* Wk, const float* bk, const float* Wv, const float* bv, int64_t n_global,
                        float attn_scale, float* coef, float* record, void* workspace, size_t workspace_bytes,
                        dif_stream_t stream);
int dif_simple_layer_f32(const float* x, int64_t ldx, int64_t n_rows, int C, int D, const float* coef,
                         const float* ax, int64_t ldax, const float* Wv, const float* bv, const float* row_sums,
                         float gcn_scale, const float* x0, int64_t ldx0, int residual, float alpha,
                         const float* ln_weight, const float* ln_bias, float ln_eps, int relu, float* out,
                         int64_t ldo, float* next_record, const int32_t* rowptr, const int32_t* plan, float* next_ys,
                         void* workspace, size_t workspace_bytes, dif_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a3, dense graphs without edge weights: feature-sliced product with the source rows staged in LDS
 *     (same function of node classification/difformer.py:63-79 as dif_gcn_spmm_f32 for edge_weight = None:
 *      value_e = deg[col]^-1/2 * deg[row]^-1/2 factors into a pre-scaled source row and a per-destination scale, so
 *      an entry is only a source index; results agree with dif_gcn_spmm_f32 to fp32 rounding, not bit for bit)
 *
 * A workgroup owns a panel of destination rows and a 16-byte slice of the feature row; source rows are swept in tiles
 * of plan[6] rows held in LDS; an entry is a 16-bit tile-local row number read with one ds_read_b128 (csrc/gcn_sliced.hip).
 *   dif_sliced_plan     plan int32[8] = {slices = F/4, panels, 64-row slots G, (panel, wave) pairs PW = panels * waves,
 *                       waves, rounds R, tile rows T (multiple of 16, <= 10,208), tiles NT}.  Host-only, deterministic
 *                       in its arguments.  DIF_E_SHAPE when F % 4 != 0 or F > 1024.  Slot g holds the rows at positions
 *                       64 g .. 64 g + 63 of row_order (NULL: natural order; pass dif_row_order's descending-degree order
 *                       when the degrees are skewed, so that the 64 lock-step lanes of a slot carry similar lists);
 *                       slot j * PW + (j odd ? PW - 1 - pw : pw) is round j of pair pw = wave * panels + panel.
 *   dif_sliced_measure  needs the CSR built by dif_csr_build(n_blocks = NT, block_rows = T) (blkptr may be NULL when
 *                       NT == 1).  Re-orders every (row, tile) group by LDS bank (-> `sorted` uint16[nnz], `counts`
 *                       32 bytes per (row position, tile)), schedules the entries (-> `lengths` int32[G*NT*4]) and
 *                       writes `table` int32[(R+1)*panels*NT*waves + 1] = {first block, blocks of round 0 >= round 1 >=
 *                       ... (padded to a non-increasing sequence)} per (panel, tile, wave) and the total number of
 *                       1-KiB blocks in its last element.  status[0] != 0: a (row, tile) group holds more than 65,535
 *                       entries -- use dif_gcn_spmm_f32 for this graph.
 *   dif_sliced_emit     writes the blocks: `entries` uint16[512 * n_blocks] with n_blocks = table[(R+1)*panels*NT*waves]
 *                       (read back by the caller: the size is data dependent).  sorted / counts / lengths may be freed
 *                       afterwards; entries + table + plan (+ row_order) are the format.
 *   dif_sliced_prescale_f32  ys float[F/4][T*NT][4] = deg^-1/2 (0 for a node without incoming entries, :74) times x,
 *                       slice-major; x holds all n_src rows.
 *   dif_sliced_spmm_f32 out[r,:] = gcn_scale * deg[r]^-1/2 * sum_e ys[src_e] (+ attn_scale * attn[r,:]) for the n_rows
 *                       rows the format was built for (out / attn hold only those rows; row_order as at build time).
 *                       Deterministic.
 *   source splits       For a row shard (n_src >= 2 n_rows: one rank's destination rows over all the source rows) the
 *                       plan has fewer, fuller panels and S = 2, 4 or 8 workgroups per (panel, slice), each sweeping
 *                       NT / S of the source tiles (NT is a multiple of S); their partial sums meet in a second kernel,
 *                       in split order.  The product then needs `ws` of dif_sliced_spmm_workspace_bytes(n_src, n_pos, F)
 *                       bytes, 16-byte aligned (0 bytes and NULL for S = 1: every whole-graph product).
 *   row positions       Without `parts` (NULL) the n_pos = n_rows positions are the rows themselves (in row_order).  With
 *                       `parts` uint16[n_pos] a position is part p of P of row row_order[pos] (parts[pos] = p | P << 8,
 *                       1 <= P <= 64: in every tile part p takes the p-th of P equal shares of the row's entries) or empty
 *                       (row_order[pos] < 0); the parts of a row occupy P consecutive positions of ONE slot, part 0 first,
 *                       and the product adds them up in part order.  Hub rows split this way run as P lock-step lanes
 *                       instead of one long one, and a (row, tile) group may hold up to P * 65,535 entries.  The plan is
 *                       dif_sliced_plan(n_src, n_pos, F).
 *   dinv (prescale, spmm): NULL = deg^-1/2 from the row lengths of `rowptr` (the forward product).  For the adjoint
 *                       product (the gradient of :75-78: CSR of the transposed graph, rows = sources) pass the forward
 *                       graph's float[n_src] vector: grad_x[s] = dinv[s] * sum_{e: src = s} (dinv * g)[dst_e].
 * ------------------------------------------------------------------------------------- */
int dif_sliced_plan(int64_t n_src, int64_t n_rows, int F, int32_t* plan);
int dif_sliced_measure(const int32_t* rowptr, const int32_t* blkptr, const int32_t* src, int64_t n_src,
                       int64_t nnz, int64_t row_begin, int64_t n_rows, int F, const int32_t* plan,
                       const int32_t* row_order, const uint16_t* parts, int64_t n_pos, uint16_t* sorted, void* counts,
                       int32_t* lengths, int32_t* table, int32_t* status, dif_stream_t stream);
int dif_sliced_emit(const int32_t* rowptr, const int32_t* blkptr, int64_t n_src, int64_t row_begin,
                    int64_t n_rows, int F, const int32_t* plan, const int32_t* row_order, const uint16_t* parts,
                    int64_t n_pos, const uint16_t* sorted, const void* counts, const int32_t* table, int64_t n_blocks,
                    uint16_t* entries, dif_stream_t stream);
int dif_sliced_prescale_f32(const float* x, int64_t ldx, const int32_t* rowptr, const float* dinv, int64_t n_src,
                            int F, const int32_t* plan, float* ys, dif_stream_t stream);
int dif_sliced_spmm_f32(const uint16_t* entries, const int32_t* table, const int32_t* plan, const float* ys,
                        const int32_t* rowptr, const float* dinv, const int32_t* row_order, const uint16_t* parts,
                        int64_t n_pos, int64_t n_src, int64_t row_begin, int64_t n_rows, int F, const float* attn,
                        int64_t lda, float attn_scale, float gcn_scale, float* out, int64_t ldo, void* ws,
                        int64_t ws_bytes, dif_stream_t stream);
int64_t dif_sliced_spmm_workspace_bytes(int64_t n_src, int64_t n_rows, int F);

/* Split product for row-sharded runs (one process per GPU): a rank owns the source rows of the blocks
 * [own_blk_begin, own_blk_end) before the all-gather of the value rows has delivered the others.
 *   part 0: sweeps only those blocks and parks the fp32 accumulators in `scratch` (no epilogue, `out` untouched).  `x` may
 *           be a pointer such that x + s*ldx is valid only for the rank's own source rows s (local rows minus
 *           row offset): no other row is read.
 *   part 1: sweeps all other blocks on top of `scratch` and finishes with the combine / tail epilogue.
 * max_workgroups (0 = one per CU) caps the persistent grid: a workgroup of this kernel fills a CU, so part 0 is launched
 * on fewer workgroups than CUs to leave room for the collective's own kernel.  `scratch` holds one fp32 row per shard
 * row, so the two parts may differ in geometry.  Blocked kernel only: n_blocks > 1,
 * F % 4 == 0, F <= 256, 16-byte aligned rows (DIF_E_SHAPE otherwise).  tail_enabled = 0 ignores the tail arguments. */
size_t dif_gcn_spmm_part_scratch_bytes(int64_t n_rows, int64_t n_split_rows, int F);
int dif_gcn_spmm_part_f32(const int32_t* rowptr, const int32_t* blkptr, int n_blocks,
                          const int32_t* src, const float* val, int64_t n_nodes, int64_t nnz,
                          const float* x, int64_t ldx, int64_t row_begin, int64_t n_rows, int F,
                          const float* attn, int64_t lda, float attn_scale, float gcn_scale,
                          const int32_t* row_order, int64_t n_split_rows, int tail_enabled,
                          const float* x0, int64_t ldx0, const float* prev, int64_t ldp, float alpha,
                          const float* ln_weight, const float* ln_bias, float ln_eps, int relu,
                          int part, int own_blk_begin, int own_blk_end, int max_workgroups,
                          float* scratch, size_t scratch_bytes, float* out, int64_t ldo,
                          dif_stream_t stream);
int dif_gcn_spmm_part_bf16(const int32_t* rowptr, const int32_t* blkptr, int n_blocks,
                           const int32_t* src, const float* val, int64_t n_nodes, int64_t nnz,
                           const void* x, int64_t ldx, int64_t row_begin, int64_t n_rows, int F,
                           const void* attn, int64_t lda, float attn_scale, float gcn_scale,
                           const int32_t* row_order, int64_t n_split_rows, int tail_enabled,
                           const void* x0, int64_t ldx0, const void* prev, int64_t ldp, float alpha,
                           const void* ln_weight, const void* ln_bias, float ln_eps, int relu,
                           int part, int own_blk_begin, int own_blk_end, int max_workgroups,
                           float* scratch, size_t scratch_bytes, void* out, int64_t ldo,
                           dif_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * f2  induced subgraph of a node subset with relabelling -- the per-batch graph step of the mini-batch path,
 *     node classification/main-batch.py:131  subgraph(idx_i, edge_index, num_nodes=n, relabel_nodes=True)
 *     (torch_geometric 1.7.2 semantics: keep the edges whose two ends are in `subset`, in their original order;
 *     subset[i] becomes node i).  out_edge_index has capacity [2, E] (row r at offset r*E); out_count (device int64)
 *     receives the number of kept edges; status[0] != 0 flags ids outside [0, N).  `subset` must not repeat ids.
 * ------------------------------------------------------------------------------------- */
size_t dif_subgraph_workspace_bytes(int64_t E, int64_t N);
int dif_subgraph(const int64_t* edge_index, int64_t E, int64_t N, const int64_t* subset, int64_t B,
                 const float* edge_weight, int64_t* out_edge_index, float* out_weight, int64_t* out_count,
                 int32_t* status, void* workspace, size_t workspace_bytes, dif_stream_t stream);

/* All mini-batches of an epoch in ONE pass over the edge list (main-batch.py:121-131: a permutation of the training
 * nodes is cut into batches of `batch_size` consecutive entries and subgraph(idx_i, edge_index, relabel_nodes=True) is
 * called per batch, each call filtering the whole edge list).  perm int64[M] (no repeats); batch b = perm[b*batch_size ..].
 *   dif_subgraph_batches_group  streams the edge list once: an edge survives iff both ends lie in the same batch; a
 *       stable radix pass groups the surviving edge ids by batch.  batch_ptr int64[n_batches + 1] (device) <- offsets,
 *       its last element = number of surviving edges (read it back to size the output).  status[0]: 1 = id outside
 *       [0, N), 2 = repeated id in perm.  n_batches = ceil(M / batch_size) < 65,535.
 *   dif_subgraph_batches_emit   out_edge_index int64 [2, capacity]: batch b owns columns [batch_ptr[b], batch_ptr[b+1]),
 *       edges in their original order, ends renumbered inside the batch (perm[b*batch_size + j] -> j): exactly what the
 *       per-batch call returns.  Same workspace as the group call (it holds the grouped edge ids). */
size_t dif_subgraph_batches_workspace_bytes(int64_t E, int64_t N, int n_batches);
int dif_subgraph_batches_group(const int64_t* edge_index, int64_t E, int64_t N, const int64_t* perm, int64_t M,
                               int64_t batch_size, int64_t* batch_ptr, int32_t* status, void* workspace,
                               size_t workspace_bytes, dif_stream_t stream);
int dif_subgraph_batches_emit(const int64_t* edge_index, int64_t E, int64_t N, int64_t M, int64_t batch_size,
                              const float* edge_weight, const int64_t* batch_ptr, int64_t capacity,
                              int64_t* out_edge_index, float* out_weight, const void* workspace,
                              size_t workspace_bytes, dif_stream_t stream);
/*   dif_subgraph_batches_csr    (optional) the CSR of EVERY batch from one more sort of the surviving edges (key =
 *       position of the destination in the permutation): rowptr int32[M + 1], src int32[kept] (batch-local ids), val
 *       float32[kept]; the slice of batch b is what dif_csr_build returns for that batch's edge list (same entry order,
 *       degrees and normalisation of the subgraph; rowptr shifted by batch_ptr[b]) -- "direct CSR emission": no per-batch
 *       dif_csr_build.  kept = batch_ptr[n_batches]; group_workspace = the workspace of dif_subgraph_batches_group. */
size_t dif_subgraph_batches_csr_workspace_bytes(int64_t kept, int64_t M);
int dif_subgraph_batches_csr(const int64_t* edge_index, int64_t E, int64_t N, int64_t M, int64_t batch_size,
                             const float* edge_weight, int64_t kept, const void* group_workspace,
                             size_t group_workspace_bytes, int32_t* rowptr, int32_t* src, float* val,
                             void* workspace, size_t workspace_bytes, dif_stream_t stream);

/* Graph preparation of the drivers on device (node classification/main.py:72-76, main-batch.py:96-98; torch_geometric.utils
 * to_undirected / remove_self_loops / add_self_loops, un-vendored), any subset of the three steps in their order:
 *   undirected   != 0: both directions of every edge, equal pairs coalesced, result sorted by (row, col);
 *   remove_loops != 0: edges (v, v) dropped;       add_loops != 0: the N loops (v, v) appended at the end.
 * Without `undirected` the surviving edges keep their order.  out_edge_index int64 [2, capacity] (row r at offset
 * r * capacity), capacity >= (undirected ? 2E : E) + (add_loops ? N : 0); out_count (device int64) <- edges written;
 * status[0] != 0: an id outside [0, N). */
size_t dif_graph_prepare_workspace_bytes(int64_t E, int64_t N, int undirected);
int dif_graph_prepare(const int64_t* edge_index, int64_t E, int64_t N, int undirected, int remove_loops,
                      int add_loops, int64_t capacity, int64_t* out_edge_index, int64_t* out_count,
                      int32_t* status, void* workspace, size_t workspace_bytes, dif_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a4/a5  tail of DIFFormerConv.forward + the per-layer tail of DIFFormer.forward
 *        node classification/difformer.py:137 (mean over heads), :139-140 (+= x_0),
 *        :200-201 (alpha residual), :202-203 (LayerNorm, eps 1e-5, affine)
 *   y = mean_h conv[n,h,:] (+ x0[n,:]) ; z = residual ? alpha*y + (1-alpha)*prev[n,:] : y ;
 *   out = ln_weight ? LayerNorm(z) * ln_weight + ln_bias : z ;  out = relu ? max(out, 0) : out
 *   (relu = 1 serves the input layer, difformer.py:188-191: Linear -> LayerNorm -> ReLU)
 * conv [n_rows,H,D]; x0, prev, out [n_rows,D]; x0 / prev / ln_weight may be NULL.
 * ------------------------------------------------------------------------------------- */
int dif_layer_tail_f32(const float* conv, int64_t ldc, int64_t n_rows, int H, int D,
                       const float* x0, int64_t ldx0, const float* prev, int64_t ldp,
                       float alpha, const float* ln_weight, const float* ln_bias, float ln_eps,
                       int relu, float* out, int64_t ldo, dif_stream_t stream);
/* The same tail with the propagation result assembled in the pass (closed form of the simple kernel at the widths of
 * the reference's scripts, hidden 128 / 300 / 400; one head, fp32):
 *   z = conv_scale * conv[row,:] / den[row * ldden] + add_scale * (add[row,:] + rs[row] * bv[:])      (:36-39, :75-78, :130-134)
 * then + x0, the alpha-residual with prev, LayerNorm, ReLU as dif_layer_tail_f32.  den / add / (rs, bv) may be NULL.
 * conv and den are typically the numerator columns and the denominator column of ONE row-GEMM output. */
/* The LAST layer of a model with its output Linear (difformer.py:208) in the same pass: logits [n, Co] = out Wo^T + bo with
 * Wo float[Co, D], bo float[Co], Co <= 128; the finished row piece is the B operand of one more transposed MFMA product.
 * out may be NULL (the rows themselves are then not stored).  Other arguments as dif_simple_layer_f32. */
int dif_simple_layer_head_f32(const float* x, int64_t ldx, int64_t n_rows, int C, int D, const float* coef,
                              const float* ax, int64_t ldax, const float* Wv, const float* bv, const float* row_sums,
                              float gcn_scale, const float* x0, int64_t ldx0, int residual, float alpha,
                              const float* ln_weight, const float* ln_bias, float ln_eps, int relu, float* out,
                              int64_t ldo, const float* Wo, const float* bo, int Co, float* logits, int64_t ldl,
                              dif_stream_t stream);
/* The coefficient chain of the closed form as BACKGROUND kernels (csrc/side_chain.hip): single-wave workgroups without LDS,
 * the footprint that fits beside a workgroup of the feature-sliced product, so that on a second stream the chain runs
 * under the product instead of in front of it (neither depends on the other).  Augmented formulation: X~ = [X | 1],
 * W~ = [W | b]; all matrices float[80 * 80], zero padded, augmented index 64.
 *   dif_gram_bg_f32          gt float[80 * 80 + 400] = G~ = [[X^T X, sum x], [sum x^T, n_global]] from one pass over x, followed
 *                            by 100 float64 pairs of partial norm products <st, G~> (x == NULL: `workspace` holds one
 *                            finished record [X^T X | sum x], e.g. dif_gram_f32's, and is only re-laid)
 *   dif_simple_coeffs_bg_f32 coef (layout of dif_simple_coeffs_f32) from gt and the weight-only factors
 *                            pt = W~q^T W~k, vtt = [W~v^T | e]^T, st = [W~q^T W~q ; W~k^T W~k]; scratch float[80 * 80 + 4]. */
size_t dif_gram_bg_workspace_bytes(int64_t n_rows, int C);
int dif_gram_bg_f32(const float* x, int64_t ldx, int64_t n_rows, int C, int64_t n_global, const float* st, float* gt,
                    void* workspace, size_t workspace_bytes, dif_stream_t stream);
int dif_simple_coeffs_bg_f32(const float* gt, const float* pt, const float* vtt, const float* st, int C, int D,
                             float attn_scale, float* scratch, float* coef, dif_stream_t stream);
/* Gram record of that closed form: record float[dif_simple_reduced_len(1, C, C)] = [X^T X (C x C, row-major) | sum x (C) |
 * C + 2 unused]; of X^T X only the 64 x 64 blocks on and above the diagonal are written (symmetric: the caller mirrors).
 * One streaming pass on the fp32 MFMA -- or, for 65..320 columns and >= 4,096 rows (round 5), ONE read of x with split-bfloat16
 * operands staged in LDS (csrc/simple_attn.hip, gram_slab_kernel; DIFFORMER_EXACT_FP32=1 keeps the fp32 pass).  workspace:
 * dif_gram_sym_workspace_bytes(n_rows, C) (a smaller one of dif_simple_workspace_bytes(n_rows, 1, C, C) bytes selects the fp32
 * pass), 16-byte aligned. */
size_t dif_gram_sym_workspace_bytes(int64_t n_rows, int C);
int dif_gram_sym_f32(const float* x, int64_t ldx, int64_t n_rows, int C, float* record, void* workspace,
                     size_t workspace_bytes, dif_stream_t stream);
/* Float64 bookkeeping of that closed form around its two library GEMMs (csrc/wide_coeffs.hip):
 *   dif_wide_gram_f64   Gt double[(C+1)^2] = [[X^T X, sx], [sx^T, n_global]] from the record of dif_gram_sym_f32 (lower
 *                       blocks mirrored), and partial double[2 * dif_wide_partials(C)]: per-workgroup sums of
 *                       <S[0], Gt> = |Q|^2 and <S[1], Gt> = |K|^2 with S double[2][(C+1)^2] = {W~q^T W~q, W~k^T W~k}
 *   dif_wide_scale_f64  s = 1 / (|Q| |K|) from the partial sums; B float[C][DV] = s R[0..C), bias float[DV] = s R[C] + T[C]
 *                       for R, T double[(C+1)][DV] (R = P~ T, T = Gt V~): the operands of the row GEMM x B + bias. */
int64_t dif_wide_partials(int C);
int dif_wide_gram_f64(const float* record, int C, int64_t n_global, const double* S, double* Gt, double* partial,
                      dif_stream_t stream);
int dif_wide_scale_f64(const double* R, const double* T, const double* partial, int C, int DV, float* B, float* bias,
                       dif_stream_t stream);
/* Round 5: the same coefficients WITHOUT the two library GEMMs -- record (dif_gram_sym_f32 / dif_gram128_f32) -> B float [C][DV] =
   s R[0..C) and bias float [DV] = s R[C] + T[C] in two launches (T = G~ V~ straight from the record with the partial norm
   products; R = P~ T with the scaling).  S double [2][(C+1)^2], V double [(C+1)][DV], P double [(C+1)][(C+1)]: weight-only factors;
   T double [(C+1)][DV], partial double [2 * ceil((C+1) / 16)]: scratch; C + 1 <= 512.  Replaces ops.simple_layer_closed_form_wide's
   dif_wide_gram_f64 -> GEMM -> GEMM -> dif_wide_scale_f64 chain (difformer.py:20-38 in closed form). */
int dif_wide_coeffs_f64(const float* record, int C, int64_t n_global, const double* S, const double* V, const double* P, int DV,
                        double* T, double* partial, float* B, float* bias, dif_stream_t stream);
/* Closed-form `simple` layer at hidden 129..416 (image and text/run.sh:27 trains at 300, two lines at 400) in ONE pass over the
 * rows (csrc/simple_layer_xwide.hip): same result as dif_simple_layer_wide_f32, with the rows kept in registers as split-bf16
 * fragments and the weights streamed through LDS in chunks of 64 output features.  The weights come PACKED:
 * dif_xwide_pack_f32(src, ld, transposed, C, D, packed) writes dif_xwide_packed_bytes(C, D) bytes of MFMA A fragments
 * ([chunk][hi | lo][k-block][feature tile][lane] x 16 B) from src = bmat ([C][ld], transposed = 1: Mn sits in columns [0, D))
 * or src = Wv ([D][ld], transposed = 0).  packed_v NULL: no graph term, or use_weight = False (then C == D). */
int64_t dif_xwide_packed_bytes(int C, int D);
int dif_xwide_pack_f32(const float* src, int64_t ld, int transposed, int C, int D, void* packed, dif_stream_t stream);
int dif_simple_layer_xwide_f32(const float* x, int64_t ldx, int64_t n_rows, int C, int D, const void* packed_m,
                               const void* packed_v, const float* bmat, int dv, const float* bias, float attn_scale,
                               const float* ax, int64_t ldax, const float* bv, const float* row_sums, float gcn_scale,
                               const float* x0, int64_t ldx0, int residual, float alpha, const float* ln_weight,
                               const float* ln_bias, float ln_eps, int relu, float* out, int64_t ldo, dif_stream_t stream);
/* nn.Linear (-> LayerNorm) (-> ReLU) with a wide result, difformer.py:188-191 at image and text/run.sh:27 (512 -> 300), on the
 * same kernel: out = LN(x[:, :Ch] Wa^T + x[:, Ch:] Wb^T + bias) with Ch = C_in / 2 <= 416 and packed_a / packed_b =
 * dif_xwide_pack_f32(W + 0 / + Ch, ld = C_in, transposed = 0, Ch, D); packed_b NULL: one product, C_in <= 416.  D <= 416;
 * C_in / Ch and D multiples of 4; rows of x and out 16-byte aligned. */
int dif_linear_xwide_f32(const float* x, int64_t ldx, int64_t n_rows, int C_in, const void* packed_a, const void* packed_b,
                         const float* bias, int D, const float* ln_weight, const float* ln_bias, float ln_eps, int relu,
                         float* out, int64_t ldo, dif_stream_t stream);
/* Gram record of the closed form for 64 < C <= 128 (hidden 128): record [X^T X (C x C, ALL of it) | sum x (C) | ...] in the
 * layout of dif_gram_sym_f32, from ONE pass over x on the fp32 MFMA (csrc/simple_layer_wide.hip).  C % 4 == 0, 16-byte aligned
 * rows; workspace: dif_gram128_workspace_bytes, 16-byte aligned. */
size_t dif_gram128_workspace_bytes(int64_t n_rows, int C);
int dif_gram128_f32(const float* x, int64_t ldx, int64_t n_rows, int C, float* record, void* workspace,
                    size_t workspace_bytes, dif_stream_t stream);
/* Closed-form `simple` layer at hidden 65..128 (node classification/run.sh:42-44) in ONE pass over the rows
 * (csrc/simple_layer_wide.hip): out = LN(alpha (a_s (x Mn + cn) / (x.u + cd) + g_s ((A_hat x) Wv^T + (A_hat 1) bv^T) [+ x0]) +
 * (1 - alpha) x).  bmat [C][dv] / bias [dv] are dif_wide_scale_f64's outputs (columns [0, D) = Mn, column D = u; cn | cd);
 * ax = A_hat x unscaled or NULL (no graph); Wv / bv / row_sums NULL together for use_weight = False (then C == D).  C, D <= 128,
 * multiples of 4, 16-byte aligned rows.  Both products run on split-bfloat16 operands (fp32 accumulation, ~4e-6). */
int dif_simple_layer_wide_f32(const float* x, int64_t ldx, int64_t n_rows, int C, int D, const float* bmat, int dv,
                              const float* bias, float attn_scale, const float* ax, int64_t ldax, const float* Wv,
                              const float* bv, const float* row_sums, float gcn_scale, const float* x0, int64_t ldx0,
                              int residual, float alpha, const float* ln_weight, const float* ln_bias, float ln_eps,
                              int relu, float* out, int64_t ldo, dif_stream_t stream);
int dif_layer_tail_mix_f32(const float* conv, int64_t ldc, const float* den, int64_t ldden, float conv_scale,
                           const float* add, int64_t lda, float add_scale, const float* rs, const float* bv,
                           int64_t n_rows, int D, const float* x0, int64_t ldx0, const float* prev, int64_t ldp,
                           float alpha, const float* ln_weight, const float* ln_bias, float ln_eps, int relu,
                           float* out, int64_t ldo, dif_stream_t stream);

/* dif_gcn_spmm_f32 with the tail above fused into its epilogue (H == 1 layers: conv row = feature
 * row, F = D <= 256): out = tail(gcn_scale * A_hat x (+ attn_scale * attn)).  Saves the [n,D] round
 * trip between the two kernels.  relu = 1 appends max(., 0) (DIFFormer_v2: norm -> ReLU, difformer-v2.py:214-217).
 * Returns DIF_E_SHAPE when the row does not fit one lane group. */
int dif_gcn_spmm_tail_f32(const int32_t* rowptr, const int32_t* blkptr, int n_blocks,
                          const int32_t* src, const float* val, int64_t n_nodes, int64_t nnz,
                          const float* x, int64_t ldx, int64_t row_begin, int64_t n_rows, int F,
                          const float* attn, int64_t lda, float attn_scale, float gcn_scale,
                          const int32_t* row_order, int64_t n_split_rows, const float* x0, int64_t ldx0, const float* prev, int64_t ldp, float alpha,
                          const float* ln_weight, const float* ln_bias, float ln_eps, int relu,
                          float* out, int64_t ldo, dif_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * f3  backward of dif_layer_tail_f32 (what loss.backward(), main.py:130, asks of difformer.py:137-140, :200-203 and of
 *     the input layer's LayerNorm -> ReLU, :189-191; the reference leaves it to autograd).  fp32, D % 4 == 0, D <= 512,
 *     16-byte aligned rows (DIF_E_SHAPE / DIF_E_BADARG otherwise).  The row is re-derived from the forward's inputs
 *     (conv, x0, prev) and every gradient leaves in one pass:
 *       d_conv [n,H,D] (each head gets dz / H), d_x0 [n,D], d_prev [n,D]  (each may be NULL: not wanted),
 *       d_ln float[2 D] = {d ln_weight [D], d ln_bias [D]} (NULL iff ln_weight is NULL).
 *     Deterministic: per-workgroup partial sums in `workspace`, column sums in a fixed order.
 * ------------------------------------------------------------------------------------- */
size_t dif_layer_tail_bwd_workspace_bytes(int64_t n_rows, int D);
int dif_layer_tail_bwd_f32(const float* conv, int64_t ldc, int64_t n_rows, int H, int D, const float* x0,
                           int64_t ldx0, const float* prev, int64_t ldp, float alpha, const float* ln_weight,
                           const float* ln_bias, float ln_eps, int relu, const float* grad_out, int64_t ldg,
                           float* d_conv, int64_t lddc, float* d_x0, int64_t lddx0, float* d_prev, int64_t lddp,
                           float* d_ln, void* workspace, size_t workspace_bytes, dif_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a5 ends  input MLP  difformer.py:188-191 (Linear -> LayerNorm -> ReLU)  and output Linear :208 for the
 * narrow shapes of this model:  out = x W^T + b  [-> LayerNorm(ln_weight, ln_bias, eps)] [-> ReLU].
 * x [n_rows, C_in], W [C_out, C_in] (nn.Linear layout), b [C_out].  Covers C_in <= 128 (and C_out <= 64
 * when LayerNorm is fused) and long rows into a narrow layer (128 < C_in <= 8192 -> C_out <= 64: the input MLP on
 * bag-of-words / embedding features); DIF_E_SHAPE otherwise -- the host then uses the vendor GEMM.
 * Long rows, float32: the products run on split-bfloat16 operands (x = hi + lo, three bf16 MFMAs, fp32 accumulation,
 * ~4e-6 of the float64 result) unless DIFFORMER_EXACT_FP32=1 is set in the environment (fp32 MFMA = an fmaf chain).
 * dif_linear_packed_f32 is the form for FEW rows (< 16,384) or rows that are only 4-byte aligned (C_in % 4 != 0: Cora's
 * 1,433 features): one workgroup per 16-row tile, K split over its eight waves, W given PACKED -- split into bfloat16
 * hi / lo parts and laid out fragment by fragment ([ceil(C_in / 64)][16][64 lanes] x 16 bytes = dif_linear_packed_bytes)
 * by dif_linear_pack_f32, which the host runs once per parameter version.
 * ------------------------------------------------------------------------------------- */
int dif_linear_f32(const float* x, int64_t ldx, int64_t n_rows, int C_in, const float* W,
                   const float* bias, int C_out, const float* ln_weight, const float* ln_bias,
                   float ln_eps, int relu, float* out, int64_t ldo, dif_stream_t stream);
int64_t dif_linear_packed_bytes(int C_in);
int dif_linear_pack_f32(const float* W, int C_in, int C_out, void* packed, dif_stream_t stream);
int dif_linear_packed_f32(const float* x, int64_t ldx, int64_t n_rows, int C_in, const void* packed,
                          const float* bias, int C_out, const float* ln_weight, const float* ln_bias,
                          float ln_eps, int relu, float* out, int64_t ldo, dif_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a1 backward: gradient of full_attention_conv(..., 'simple') (difformer.py:18-39) w.r.t. q, k, v -- what
 * loss.backward() (main.py:130) needs; the reference leaves it to autograd.  With gn = g/den and
 * gd = -(g.out)/den per (row, head):  dif_simple_bwd_prep_f32 writes gn [n,H,D], gd [n,H] and
 * sums [H*M+1] = { sum_n q*gd per (h,m), sum gd };  dif_simple_reduce_f32(q, q, gn) then gives q^T gn and
 * sum gn;  dif_rowgemm_f32 (out = A Mat + bias + r (x) u + beta Cin per head, K <= 512, beta a DEVICE scalar)
 * forms dq, dk, dv (formulas in csrc/simple_attn_bwd.hip).
 * ------------------------------------------------------------------------------------- */
size_t dif_simple_bwd_workspace_bytes(int64_t n_rows, int H, int M, int D);
int dif_simple_bwd_prep_f32(const float* q, int64_t ldq, const float* g, int64_t ldg,
                            const float* out, int64_t ldo, const float* reduced, int64_t n_rows,
                            int64_t n_global, int H, int M, int D, float* gn, float* gd, float* sums,
                            void* workspace, size_t workspace_bytes, dif_stream_t stream);
int dif_rowgemm_f32(const float* A, int64_t lda, const float* Mat, int ldm, int mat_head_stride,
                    int mat_t, float mat_scale, const float* bias, const float* r, const float* u,
                    float u_scale, const float* Cin, int64_t ldc, const float* beta_dev,
                    int64_t n_rows, int H, int K, int C, float* out, int64_t ldo, dif_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * bfloat16 STORAGE variants (BASELINE config C5: Pokec mini-batches in bf16).  Same semantics as the
 * _f32 entry points; every tensor the reference would hold in its activation dtype (x, q, k, v, attn,
 * conv, out, Linear / LayerNorm parameters) is bf16 (uint16 bit patterns behind void*), while the CSR
 * values, the `reduced` record and all accumulation stay float32.  The reference itself cannot run
 * bf16 (difformer.py:27 raises), so parity is against the float64 oracle on the bf16-rounded inputs
 * at a bf16-appropriate tolerance.  dif_gcn_spmm_tail_bf16: tail_enabled = 0 gives the plain SpMM.
 * ------------------------------------------------------------------------------------- */
int dif_linear_bf16(const void* x, int64_t ldx, int64_t n_rows, int C_in, const void* W,
                    const void* bias, int C_out, const void* ln_weight, const void* ln_bias,
                    float ln_eps, int relu, void* out, int64_t ldo, dif_stream_t stream);
int dif_project_reduce_bf16(const void* x, int64_t ldx, int64_t n_rows, int C_in,
                            const void* Wq, const void* bq, const void* Wk, const void* bk,
                            const void* Wv, const void* bv, int H, int D,
                            void* q_out, int64_t ldq, void* v_out, int64_t ldv,
                            float* reduced, void* workspace, size_t workspace_bytes,
                            dif_stream_t stream);
int dif_simple_reduce_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk,
                           const void* v, int64_t ldv, int64_t n_rows, int H, int M, int D,
                           float* reduced, void* workspace, size_t workspace_bytes,
                           dif_stream_t stream);
int dif_simple_apply_bf16(const void* q, int64_t ldq, const float* reduced, int64_t n_rows,
                          int64_t n_global, int H, int M, int D, void* out, int64_t ldo,
                          dif_stream_t stream);
/* closed form of the simple layer with bfloat16 ACTIVATIONS (x, ax, x0, out); the record, the coefficients and the
 * parameters (Wv, bv, LayerNorm) are float32 -- the host keeps exact float32 copies of its bfloat16 parameters and calls
 * dif_simple_coeffs_f32 with them.  No products for a next layer (the sliced format is float32-only). */
int dif_gram_bf16(const void* x, int64_t ldx, int64_t n_rows, int C, float* record, void* workspace,
                  size_t workspace_bytes, dif_stream_t stream);
/* Input layer + Gram record + slice-major copy in one pass (difformer.py:188-191 feeding the first closed-form layer on a
 * dense graph): out = ReLU(LayerNorm(x W^T + b)) [n_rows, D] row-major for C_in <= 64 -> D <= 64 (D % 4 == 0), record =
 * [out^T out | column sums] as dif_gram_f32 leaves it, ys (nullable with rowptr / plan) = the deg^-1/2-scaled slice-major
 * copy the sliced product reads.  workspace: dif_gram_workspace_bytes(n_rows, D). */
int dif_input_gram_f32(const float* x, int64_t ldx, int64_t n_rows, int C_in, const float* W, const float* bias, int D,
                       const float* ln_weight, const float* ln_bias, float ln_eps, int relu, float* out, int64_t ldo,
                       const int32_t* rowptr, const int32_t* plan, float* ys, float* record, void* workspace,
                       size_t workspace_bytes, dif_stream_t stream);
/* ... and with the model's output Linear in the same pass (dif_simple_layer_head_f32 with bfloat16 activations and logits) */
int dif_simple_layer_head_bf16(const void* x, int64_t ldx, int64_t n_rows, int C, int D, const float* coef, const void* ax,
                               int64_t ldax, const float* Wv, const float* bv, const float* row_sums, float gcn_scale,
                               const void* x0, int64_t ldx0, int residual, float alpha, const float* ln_weight,
                               const float* ln_bias, float ln_eps, int relu, void* out, int64_t ldo, const float* Wo,
                               const float* bo, int Co, void* logits, int64_t ldl, dif_stream_t stream);
/* Backward of dif_simple_coeffs_f32 (training through the Gram record; difformer.py:18-38 under autograd without q, k, v).
   dcoef = gradient with respect to coef, in coef's layout [dMnT: D x C][dcn: D][du: C][dcd]; coef = the forward's output.
   out (dif_simple_coeffs_bwd_len floats) = [S: C x C][t: C][dWq: D x C][dbq: D][dWk: D x C][dbk: D][dWv: D x C][dbv: D]:
   the gradient of the rows through the record is  dx = x S + 1 t^T;  dWv, dbv are left untouched when Wv == NULL. */
size_t dif_simple_coeffs_bwd_len(int C, int D);
int dif_simple_coeffs_bwd_f32(const float* record, int64_t n_global, int C, int D, const float* Wq, const float* bq,
                              const float* Wk, const float* bk, const float* Wv, const float* bv, float attn_scale,
                              const float* coef, const float* dcoef, float* out, dif_stream_t stream);
/* Backward of the attention term of the closed-form layer, att = (x Mn + cn) / (x u + cd) (difformer.py:25-39 in closed
   form), in one pass over the rows: d [n, D] = gradient with respect to att (ldd), dx_in [n, C] (nullable, ldi) = what dx
   already holds.  -> d_num [n, D] (dense) = d / den, d_den [n] = -<d_num, att>, dx [n, C] (ldo) = dx_in + d_num Mn^T + d_den u^T.
   C, D <= 64 and multiples of 4, rows 16-byte aligned (DIF_E_SHAPE / DIF_E_BADARG otherwise). */
int dif_closed_form_attn_bwd_f32(const float* x, int64_t ldx, int64_t n_rows, int C, int D, const float* coef, const float* d,
                                 int64_t ldd, const float* dx_in, int64_t ldi, float* d_num, float* d_den, float* dx,
                                 int64_t ldo, const float* row_sums, float* sums, dif_stream_t stream);
/* sums (nullable): dif_closed_form_attn_bwd_groups(n_rows) records of 132 floats, one per workgroup, to be added up by the
   caller: [x^T d_den: 64][row_sums^T d: 64][sum d_den][3 unused] -- the gradients of u and cd and, with row_sums [n]
   (nullable) = the row sums of the adjacency, the bias gradient of the weighted graph branch (difformer.py:118, :130-134). */
int dif_closed_form_attn_bwd_groups(int64_t n_rows);
/* The closed-form layer with the AGGREGATION in the same pass, for graphs with a few entries per row on one GPU (replaces
   the dif_gcn_spmm_* launch + dif_simple_layer_*; reference: gcn_conv difformer.py:59-73 folded into DIFFormerConv.forward
   difformer.py:107-130): rowptr int32 [n_rows + 1] / src int32 / val float32 = dif_csr_build's CSR with n_blocks = 1 over the
   SAME n_rows nodes that x holds.  Wo != NULL: also the output Linear (Co <= 128) -> logits [n_rows, Co], `out` may be NULL. */
int dif_simple_layer_gather_f32(const float* x, int64_t ldx, int64_t n_rows, int C, int D, const float* coef,
                                const int32_t* rowptr, const int32_t* src, const float* val, const float* Wv, const float* bv,
                                float gcn_scale, const float* x0, int64_t ldx0, int residual, float alpha,
                                const float* ln_weight, const float* ln_bias, float ln_eps, int relu, float* out, int64_t ldo,
                                const float* Wo, const float* bo, int Co, float* logits, int64_t ldl, dif_stream_t stream);
int dif_simple_layer_gather_bf16(const void* x, int64_t ldx, int64_t n_rows, int C, int D, const float* coef,
                                 const int32_t* rowptr, const int32_t* src, const float* val, const float* Wv, const float* bv,
                                 float gcn_scale, const void* x0, int64_t ldx0, int residual, float alpha,
                                 const float* ln_weight, const float* ln_bias, float ln_eps, int relu, void* out, int64_t ldo,
                                 const float* Wo, const float* bo, int Co, void* logits, int64_t ldl, dif_stream_t stream);
int dif_simple_layer_bf16(const void* x, int64_t ldx, int64_t n_rows, int C, int D, const float* coef, const void* ax,
                          int64_t ldax, const float* Wv, const float* bv, const float* row_sums, float gcn_scale,
                          const void* x0, int64_t ldx0, int residual, float alpha, const float* ln_weight,
                          const float* ln_bias, float ln_eps, int relu, void* out, int64_t ldo, dif_stream_t stream);
int dif_sigmoid_attn_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                          int64_t N, int64_t L, int H, int M, int D, void* out, int64_t ldo, void* workspace,
                          size_t workspace_bytes, dif_stream_t stream);
int dif_gcn_spmm_tail_bf16(const int32_t* rowptr, const int32_t* blkptr, int n_blocks,
                           const int32_t* src, const float* val, int64_t n_nodes, int64_t nnz,
                           const void* x, int64_t ldx, int64_t row_begin, int64_t n_rows, int F,
                           const void* attn, int64_t lda, float attn_scale, float gcn_scale,
                           const int32_t* row_order, int64_t n_split_rows, int tail_enabled, const void* x0, int64_t ldx0, const void* prev,
                           int64_t ldp, float alpha, const void* ln_weight, const void* ln_bias,
                           float ln_eps, int relu, void* out, int64_t ldo, dif_stream_t stream);
int dif_layer_tail_bf16(const void* conv, int64_t ldc, int64_t n_rows, int H, int D,
                        const void* x0, int64_t ldx0, const void* prev, int64_t ldp,
                        float alpha, const void* ln_weight, const void* ln_bias, float ln_eps,
                        int relu, void* out, int64_t ldo, dif_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Whole model on TINY graphs (csrc/tiny_model.hip): the `spatial-temporal/` folder trains DIFFormer(d, 4, 1, num_layers=2,
 * num_heads=1, use_weight=False) on 20 / 129 / 1,068-node snapshots, hundreds of forwards per epoch on tensors fresh from
 * `snapshot.to(device)` (spatial-temporal/run.sh:5-40, main.py:94-121).  Three launches replace the ~110 of the
 * layer-by-layer path: the graph preparation, the forward, the backward.  One head, float32, n <= 4,096 nodes,
 * hidden <= 8, <= 64 input features, <= 8 outputs, <= 8 layers.  kernel = 1 ('sigmoid') above 64 nodes (run.sh:39-43:
 * wikimath, 1,068 nodes): the same two calls issue one launch per layer (two beyond 512 nodes) plus one, the layer's
 * O(n^2) pair loop spread over the chip (csrc/tiny_sigmoid_grid.hip); kernel = 0 ('simple') above 256 nodes likewise
 * (csrc/tiny_simple_grid.hip: the sums over nodes met across launches); launch_plan forces either form.
 *
 * dif_tiny_graph_build   replaces gcn_conv's graph side (node classification/difformer.py:64-75 =
 *   spatial-temporal/difformer.py:64-75: degree(col), d_norm_in / d_norm_out, value = w * d_in * d_out, nan_to_num,
 *   SparseTensor(row=col, col=row)) for E <= 65,535 entries: rowptr / src / val = destination-major CSR (a row's entries
 *   in edge order; val bit-identical to dif_csr_build's), rowptr_t / dst_t / val_t = its transpose (the backward's
 *   operand).  status int32 [2]: [0] != 0 an index outside [0, N) (the entry is then filed under node 0), [1] = longest
 *   destination row.  All buffers caller-owned; one workgroup per direction, no host synchronisation.
 * dif_tiny_forward_f32   replaces DIFFormer.forward (difformer.py:184-209) with everything below it (DIFFormerConv.forward
 *   :113-145, full_attention_conv :10-61 for 'simple' (kernel = 0) and 'sigmoid' (kernel = 1), gcn_conv :63-79).
 *   params: 6 + 8 * num_layers device pointers (host array) in the order fcs.0.weight, fcs.0.bias, bns.0.weight,
 *   bns.0.bias, fcs.1.weight, fcs.1.bias, then per layer Wk.weight, Wk.bias, Wq.weight, Wq.bias, Wv.weight, Wv.bias,
 *   bns.{l+1}.weight, bns.{l+1}.bias (NULL where the configuration has none: use_bn = 0, use_weight = 0), row-major as
 *   nn.Linear stores them.  attn_scale / gcn_scale: 1 / 1, or (1 - graph_weight) / graph_weight (:130-134), times a
 *   constant edge weight.  rnd: [(num_layers + 1), n, hidden] uniforms in [0, 1) for the dropouts of :192 and :204
 *   (an element is kept iff its uniform >= dropout and scaled by 1 / (1 - dropout)), or NULL (eval, dropout = 0).
 *   tape: dif_tiny_tape_floats(...) floats, 16-byte aligned: the layer inputs, pre-LayerNorm values, attention outputs
 *   and sums the backward reads.  y [n, out_channels].
 * dif_tiny_backward_f32  what autograd derives for `cost.backward()` (spatial-temporal/main.py:112,119) from that
 *   forward: grads = pointers in the order of params (each the size of its parameter), dx [n, in_channels] or NULL;
 *   rowptr_t / dst_t / val_t = the transposed CSR; scratch: dif_tiny_scratch_floats(...) floats.  The tape is only read:
 *   `backward(retain_graph=True)` may run again.  Bitwise reproducible (sums over nodes in a fixed order).
 */
typedef struct {
    int32_t n, in_channels, hidden, out_channels, num_layers;
    int32_t kernel;                 /* 0 = 'simple', 1 = 'sigmoid' */
    int32_t use_bn, use_residual, use_weight, use_graph, use_source, training;
    float alpha, attn_scale, gcn_scale, dropout, eps;
    int32_t launch_plan;            /* 0 = by size; 1 = one workgroup; 2 = one launch per layer stage over the chip */
    int64_t nnz;
} dif_tiny_cfg;
size_t dif_tiny_tape_floats(int n, int hidden, int num_layers);
size_t dif_tiny_scratch_floats(int n, int hidden, int num_layers);
size_t dif_tiny_graph_workspace_bytes(int64_t E, int64_t N);
int dif_tiny_graph_build(const int64_t* edge_index, const float* edge_weight, int64_t E, int64_t N, int32_t* rowptr,
                         int32_t* src, float* val, int32_t* rowptr_t, int32_t* dst_t, float* val_t, int32_t* status,
                         void* workspace, size_t workspace_bytes, dif_stream_t stream);
int dif_tiny_forward_f32(const dif_tiny_cfg* cfg, const float* x, int64_t ldx, const void* const* params,
                         const int32_t* rowptr, const int32_t* src, const float* val, const float* rnd, float* tape,
                         float* y, dif_stream_t stream);
int dif_tiny_backward_f32(const dif_tiny_cfg* cfg, const float* x, int64_t ldx, const void* const* params,
                          const int32_t* rowptr_t, const int32_t* dst_t, const float* val_t, const float* rnd,
                          float* tape, const float* grad_y, void* const* grads, float* dx, float* scratch,
                          dif_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFORMER_HIP_H */

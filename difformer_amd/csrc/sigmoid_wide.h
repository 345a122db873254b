// csrc/sigmoid_wide.hip: sigmoid attention for heads of 65 .. 512 columns on split-bfloat16 planes (forward + backward);
// reached through dif_sigmoid_attn_f32 / dif_sigmoid_attn_fwd_f32 / dif_sigmoid_attn_bwd_f32 (not part of the C ABI itself).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace dif {

bool sigw_covers(int M, int D);                         // 64 < max(M, D) <= 512
bool sigw_narrow_pays(int M, int D, int64_t N, int64_t L, bool training);   // 32 < max(M, D) <= 64 and enough pairs for the plane kernels
size_t sigw_fwd_workspace_bytes(int64_t N, int64_t L, int H, int M, int D);
int sigw_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, int64_t N, int64_t L, int H,
             int M, int D, float* out, int64_t ldo, float* den_or_null, void* workspace, size_t workspace_bytes, hipStream_t st);
size_t sigw_bwd_workspace_bytes(int64_t N, int64_t L, int H, int M, int D);
int sigw_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* out, int64_t ldo,
             const float* den, const float* g, int64_t ldg, int64_t N, int64_t L, int H, int M, int D, float* dq, int64_t lddq,
             float* dk, int64_t lddk, float* dv, int64_t lddv, void* workspace, size_t workspace_bytes, hipStream_t st);

}  // namespace dif

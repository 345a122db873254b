// Shared helpers for the gfx950 kernels of libdifformer_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/difformer_hip.h"

namespace dif {

// thread-local last-error text, returned by dif_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Post-launch check: hipGetLastError only (never synchronises).
int launch_status(const char* what);

constexpr int kWave = 64;         // CDNA wavefront
constexpr int kCUs = 256;         // MI355X

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace dif

#define DIF_REQUIRE(cond, code, ...) \
    do { if (!(cond)) return dif::fail((code), __VA_ARGS__); } while (0)

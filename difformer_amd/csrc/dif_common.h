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

// record_finalize.hip: column-sum the P partial records of the simple kernel's stage 1 into `reduced`
int launch_record_finalize(const float* ws, int P, int64_t ws_stride, int t_main, int tiles, float* reduced,
                           hipStream_t st);

// DIFFORMER_EXACT_FP32=1: products that normally run on split-bfloat16 operands (x = hi + lo, three bf16 MFMAs per product,
// ~4e-6 of the float64 result: the long-row input Linear, the output Linear inside the last layer kernel) stay on the fp32
// MFMA (bitwise an fmaf chain).  Read once per process.
bool exact_fp32();

constexpr int kWave = 64;         // CDNA wavefront
constexpr int kCUs = 256;         // MI355X

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), every lane gets the total: four v_add_f32 with DPP operands
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror) instead of four ds_bpermute round trips through the
// LDS crossbar.  Bitwise equal to the xor butterfly (1, 2, 4, 8): after each step all lanes of the merged group hold the
// same partial sum, so the mirrored partner carries exactly the value the xor partner would.
template <int CTRL>
__device__ __forceinline__ float dpp_take(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_take<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_take<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_take<0x141>(v);     // row_half_mirror
    v += dpp_take<0x140>(v);     // row_mirror
    return v;
}

// Sum over the four 16-lane rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48), every lane gets the total: gfx950's
// v_permlane16_swap / v_permlane32_swap (odd rows of one copy <-> even rows of the other; upper half <-> lower half) and two adds
// instead of two ds_bpermute round trips.  Bitwise equal to `v += shfl_xor(v, 16); v += shfl_xor(v, 32)`.  Inline assembly: the
// compiler's builtin returns both results in one register on ROCm 7.2 (v_add v1, v1, v1), and it needs the two wait states
// the compiler would insert after the VALU writes of the operands.
__device__ __forceinline__ float rows4_sum(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));      // (not volatile: a pure function of a, b)
    a += b;
    b = a;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace dif

#define DIF_REQUIRE(cond, code, ...) \
    do { if (!(cond)) return dif::fail((code), __VA_ARGS__); } while (0)

// ---- storage types: float32 (the reference's dtype) and bfloat16 (storage only; every accumulation is fp32) ----
namespace dif {

struct bf16 { uint16_t bits; };

__host__ __device__ __forceinline__ float bf16_to_f32(uint16_t b) {
    union { uint32_t u; float f; } c;
    c.u = static_cast<uint32_t>(b) << 16;
    return c.f;
}

// round-to-nearest-even, NaN preserved (quiet)
__host__ __device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    union { float f; uint32_t u; } c;
    c.f = f;
    if ((c.u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((c.u >> 16) | 0x40u);
    c.u += 0x7fffu + ((c.u >> 16) & 1u);
    return static_cast<uint16_t>(c.u >> 16);
}

// A pointer the compiler could not prove global (computed from loaded indices inside a loop, rebuilt from an integer, read out of
// a struct argument) is dereferenced with FLAT instructions: slower, and they count on the LDS counter too.  Every tensor this
// library touches is device-global memory: say so where it matters.
#define DIF_GLOBAL __attribute__((address_space(1)))
template <typename U>
__device__ __forceinline__ const U DIF_GLOBAL* as_global(const U* p) { return (const U DIF_GLOBAL*)p; }

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kBytes = 4;
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ f32x4 ld4g(const float* p) { return *as_global(reinterpret_cast<const f32x4*>(p)); }   // global_load, whatever the compiler knows
    static __device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
    // four elements from an address that is only ELEMENT-aligned (rows of 65 or 1,433 columns): still one global_load_dwordx4
    // (the target runs in unaligned access mode; the type's alignment tells the compiler not to assume more)
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    static __device__ __forceinline__ f32x4 ld4u(const float* p) { const f32x4u v = *reinterpret_cast<const f32x4u*>(p); return f32x4{v[0], v[1], v[2], v[3]}; }
};
template <> struct Elem<bf16> {
    static constexpr int kBytes = 2;
    static __device__ __forceinline__ float ld(const bf16* p) { return bf16_to_f32(p->bits); }
    static __device__ __forceinline__ void st(bf16* p, float v) { p->bits = f32_to_bf16(v); }
    static __device__ __forceinline__ f32x4 unpack4(uint32_t x, uint32_t y) {
        f32x4 v;
        v[0] = bf16_to_f32(static_cast<uint16_t>(x & 0xffffu)); v[1] = bf16_to_f32(static_cast<uint16_t>(x >> 16));
        v[2] = bf16_to_f32(static_cast<uint16_t>(y & 0xffffu)); v[3] = bf16_to_f32(static_cast<uint16_t>(y >> 16));
        return v;
    }
    static __device__ __forceinline__ f32x4 ld4(const bf16* p) {       // one 8-byte load
        const uint2 r = *reinterpret_cast<const uint2*>(p);
        return unpack4(r.x, r.y);
    }
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ f32x4 ld4g(const bf16* p) {      // ... as a global_load, whatever the compiler knows
        const u32x2 r = *as_global(reinterpret_cast<const u32x2*>(p));
        return unpack4(r[0], r[1]);
    }
    // four elements from an address that is only 2-byte aligned (rows of 65 bfloat16: every other row starts in the middle of a
    // dword).  A misaligned 8-byte load is split by the memory pipeline (skinny_linear_kernel<5, bf16>: 37 us against 26 us for
    // its float32 twin at 100,000 x 65); instead: the three ALIGNED dwords that cover the eight bytes, shifted into place by
    // v_alignbit (two VALU instructions).  Reads up to two bytes past the four elements -- inside the same allocation for any
    // tensor whose storage is a multiple of four bytes (the allocator's granularity is 512).
    typedef uint32_t u32x3a __attribute__((ext_vector_type(3), aligned(4)));
    static __device__ __forceinline__ f32x4 ld4u(const bf16* p) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        const u32x3a r = *(const u32x3a DIF_GLOBAL*)(a & ~static_cast<uintptr_t>(3));        // (an address rebuilt from an integer: flat without the cast)
        const uint32_t sh = static_cast<uint32_t>(a & 2u) * 8u;           // 0 or 16 bits
        const uint32_t lo = __builtin_amdgcn_alignbit(r[1], r[0], sh);    // ({r1, r0} >> sh) & 0xffffffff
        const uint32_t hi = __builtin_amdgcn_alignbit(r[2], r[1], sh);
        return f32x4{bf16_to_f32(static_cast<uint16_t>(lo & 0xffffu)), bf16_to_f32(static_cast<uint16_t>(lo >> 16)),
                     bf16_to_f32(static_cast<uint16_t>(hi & 0xffffu)), bf16_to_f32(static_cast<uint16_t>(hi >> 16))};
    }
    static __device__ __forceinline__ void st4(bf16* p, f32x4 v) {     // one 8-byte store
        uint2 r;
        r.x = static_cast<uint32_t>(f32_to_bf16(v[0])) | (static_cast<uint32_t>(f32_to_bf16(v[1])) << 16);
        r.y = static_cast<uint32_t>(f32_to_bf16(v[2])) | (static_cast<uint32_t>(f32_to_bf16(v[3])) << 16);
        *reinterpret_cast<uint2*>(p) = r;
    }
};

// pointer usable with Elem<T>::ld4 / st4 (4 elements wide)
template <typename T>
inline bool aligned_v4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & (4 * sizeof(T) - 1)) == 0; }

}  // namespace dif

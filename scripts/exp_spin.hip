// experiment: a grid of single-wave workgroups without LDS that spin on VALU work (co-residency probe)
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(64) void spin_kernel(float* out, int iters) {
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; }
    if (a == 12345.678f) out[blockIdx.x] = a + b;
}
extern "C" int spin_launch(float* out, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(64), 0, static_cast<hipStream_t>(stream), out, iters);
    return static_cast<int>(hipGetLastError());
}

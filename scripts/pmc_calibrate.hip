// Calibration kernels for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md section HBM):
// known byte counts in the access patterns the SpMM uses (4-byte coalesced streams, 16-byte gathers/streams,
// 16-byte stores).  Build on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/pmc_calibrate.hip -o /tmp/pmc_cal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void read_dword(const float* __restrict__ p, size_t n, float* out) {
    float a = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i];
    if (a == 123.456f) *out = a;
}
__global__ void read_dwordx4(const f32x4* __restrict__ p, size_t n4, float* out) {
    f32x4 a = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) a += p[i];
    if (a[0] + a[1] + a[2] + a[3] == 123.456f) *out = a[0];
}
__global__ void write_dwordx4(f32x4* __restrict__ p, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        p[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}
int main() {
    const size_t bytes = size_t(2) << 30;   // 2 GiB: far beyond the 256 MiB Infinity Cache
    float *buf, *out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4);
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_dword, dim3(2048), dim3(256), 0, 0, buf, bytes / 4, out);
        hipLaunchKernelGGL(read_dwordx4, dim3(2048), dim3(256), 0, 0, (const f32x4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(write_dwordx4, dim3(2048), dim3(256), 0, 0, (f32x4*)buf, bytes / 16);
    }
    hipDeviceSynchronize();
    printf("calibration bytes per launch: %zu\n", bytes);
    return 0;
}

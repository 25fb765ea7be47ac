#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) copy4(const uint4* a, uint4* b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) b[i] = a[i];
}
__global__ void __launch_bounds__(256) rmw4(uint4* a, size_t n) {   // read and write back in place
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { uint4 v = a[i]; v.x ^= 1; a[i] = v; }
}
int main() {
    const size_t bytes = 6ull << 30, n = bytes / 16;
    uint4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL(copy4, dim3(256 * 16), dim3(256), 0, 0, a, b, n); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("copy: %.3f ms, %.2f TB/s (read + write)\n", ms, 2.0 * bytes / (ms * 1e-3) / 1e12);
        hipEventRecord(e0); hipLaunchKernelGGL(rmw4, dim3(256 * 16), dim3(256), 0, 0, a, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("in-place rmw: %.3f ms, %.2f TB/s (read + write)\n", ms, 2.0 * bytes / (ms * 1e-3) / 1e12);
    }
}

// Plain streaming rates of this box: copy, in-place read-modify-write, read only, write only, each with ordinary
// and with non-temporal (streaming) accesses.  6 GiB buffers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void __launch_bounds__(256) copy4(const u4* a, u4* b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
        else b[i] = a[i];
    }
}
template <bool NT>
__global__ void __launch_bounds__(256) rmw4(u4* a, size_t n) {  // read and write back in place
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        u4 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        v.x ^= 1;
        if (NT) __builtin_nontemporal_store(v, a + i);
        else a[i] = v;
    }
}
template <bool NT>
__global__ void __launch_bounds__(256) read4(const u4* a, u4* sink, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u4 acc = {0, 0, 0, 0};
    for (; i < n; i += stride) acc ^= NT ? __builtin_nontemporal_load(a + i) : a[i];
    if (acc.x == 0x12345678u) sink[0] = acc;
}
template <bool NT>
__global__ void __launch_bounds__(256) write4(u4* a, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const u4 v = {1, 2, 3, (unsigned)i};
    for (; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(v, a + i);
        else a[i] = v;
    }
}
#define TIME(name, factor, ...)                                                                       \
    do {                                                                                              \
        (void)hipEventRecord(e0);                                                                     \
        __VA_ARGS__;                                                                                  \
        (void)hipEventRecord(e1);                                                                     \
        (void)hipEventSynchronize(e1);                                                                \
        float ms;                                                                                     \
        (void)hipEventElapsedTime(&ms, e0, e1);                                                       \
        printf("%-28s %.3f ms, %.2f TB/s\n", name, ms, (factor) * bytes / (ms * 1e-3) / 1e12);        \
    } while (0)
int main() {
    const size_t bytes = 6ull << 30, n = bytes / 16;
    u4 *a, *b;
    (void)hipMalloc(&a, bytes), (void)hipMalloc(&b, bytes), (void)hipMemset(a, 1, bytes), (void)hipMemset(b, 2, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const dim3 g(256 * 16), t(256);
    for (int rep = 0; rep < 2; rep++) {
        TIME("copy (read + write)", 2.0, hipLaunchKernelGGL(copy4<false>, g, t, 0, 0, a, b, n));
        TIME("copy, non-temporal", 2.0, hipLaunchKernelGGL(copy4<true>, g, t, 0, 0, a, b, n));
        TIME("in-place rmw", 2.0, hipLaunchKernelGGL(rmw4<false>, g, t, 0, 0, a, n));
        TIME("in-place rmw, non-temporal", 2.0, hipLaunchKernelGGL(rmw4<true>, g, t, 0, 0, a, n));
        TIME("read only", 1.0, hipLaunchKernelGGL(read4<false>, g, t, 0, 0, a, b, n));
        TIME("read only, non-temporal", 1.0, hipLaunchKernelGGL(read4<true>, g, t, 0, 0, a, b, n));
        TIME("write only", 1.0, hipLaunchKernelGGL(write4<false>, g, t, 0, 0, a, n));
        TIME("write only, non-temporal", 1.0, hipLaunchKernelGGL(write4<true>, g, t, 0, 0, a, n));
    }
}

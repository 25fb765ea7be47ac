// Throughput of 64-bit atomicMin (and of a plain 8-byte read-compare, and a 4-byte atomicMin) on scattered, mostly
// distinct addresses of a large array -- what a per-(ray, leaf group) kernel would do to the ray's closest t.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/atomic_rate scripts/ubench/atomic_rate.hip
// usage: atomic_rate [slots (millions), default 112] [ops per slot x100, default 190]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// mode 0: atomicMin u64 on slot[idx].t; 1: load, compare, atomicMin only if smaller; 2: u32 atomicMin; 3: load only
template <int MODE>
__global__ void k(unsigned long long* t, uint32_t* p, uint64_t n_slots, uint64_t n_ops, uint32_t spread, unsigned long long* sink) {
    unsigned long long acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_ops; i += (uint64_t)gridDim.x * blockDim.x) {
        // pair i belongs to a ray near i * n_slots / n_ops (emission order ~ slot order), jittered by `spread`
        const uint64_t h = mix(i);
        uint64_t s = (uint64_t)((double)i * (double)n_slots / (double)n_ops) + (h % spread);
        if (s >= n_slots) s = n_slots - 1;
        const unsigned long long v = (h >> 12) | 0x3ff0000000000000ull;
        unsigned long long* a = t + s * 16;  // the t field of a 128-byte slot
        if (MODE == 0) atomicMin(a, v);
        else if (MODE == 1) { if (v < __builtin_nontemporal_load(a)) atomicMin(a, v); }
        else if (MODE == 2) atomicMin(p + s * 32, (uint32_t)h);
        else acc += *a;
    }
    if (acc == 0x1234567) *sink = acc;
}

int main(int argc, char** argv) {
    const uint64_t n_slots = (uint64_t)(argc > 1 ? atoi(argv[1]) : 112) * 1000000ull;
    const uint64_t n_ops = n_slots * (uint64_t)(argc > 2 ? atoi(argv[2]) : 190) / 100ull;
    unsigned long long *t, *sink;
    CK(hipMalloc(&t, n_slots * 128));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(t, 0x7f, n_slots * 128));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (uint32_t spread : {1u, 512u, 65536u}) {
        for (int mode = 0; mode < 4; mode++) {
            CK(hipMemset(t, 0x7f, n_slots * 128));
            CK(hipEventRecord(e0));
            const dim3 g(256 * 20), b(256);
            if (mode == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, t, (uint32_t*)t, n_slots, n_ops, spread, sink);
            if (mode == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, t, (uint32_t*)t, n_slots, n_ops, spread, sink);
            if (mode == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, t, (uint32_t*)t, n_slots, n_ops, spread, sink);
            if (mode == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, t, (uint32_t*)t, n_slots, n_ops, spread, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("spread %6u mode %d (%s): %8.2f ms for %.0f M ops = %7.1f G ops/s\n", spread, mode,
                   mode == 0 ? "atomicMin u64" : mode == 1 ? "load, then atomicMin u64 if smaller" : mode == 2 ? "atomicMin u32" : "load u64",
                   ms, n_ops / 1e6, n_ops / ms / 1e6);
        }
    }
    return 0;
}

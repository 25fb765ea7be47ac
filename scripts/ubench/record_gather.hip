// What the memory system delivers for the traversal kernel's fetch pattern: every lane reads its own random 128-byte record
// (seven dwordx4 loads, 64 different lines per instruction) from a table of a given size, does a little arithmetic on it
// and takes the next record from what it read (a dependent chain, as a walk is) -- at 8, 5 and 4 workgroups per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/record_gather scripts/ubench/record_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ inline uint64_t mix(uint64_t h) { h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; return h; }

// WORK: dependent f32 fma's between two fetches (0: none; 100 ~ the cheap box test; 165 ~ the f64 step's issue time / 4 cycles)
template <int WORK>
__global__ void __launch_bounds__(256) k(const uint4* table, uint32_t* out, uint32_t n_rec, int iters) {
    extern __shared__ uint32_t pad[];
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t rec = (uint32_t)(mix(tid + 1) % n_rec);
    float acc = 1.0f;
    for (int it = 0; it < iters; it++) {
        const uint4* p = table + (size_t)rec * 8;
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6];
        uint32_t x = a.x ^ b.y ^ c.z ^ d.w ^ e.x ^ f.y ^ g.z;
        float v = __uint_as_float((x & 0x007fffffu) | 0x3f800000u);
#pragma unroll
        for (int w = 0; w < WORK; w++) v = __builtin_fmaf(v, 0.999f, acc * 1e-9f);
        acc += v;
        rec = (uint32_t)(mix(x + __float_as_uint(v) + it + (uint64_t)tid * 0x9E3779B9ull) % n_rec);   // depends on this record, differs by lane
    }
    out[tid] = __float_as_uint(acc) + pad[0] * 0u;
}

template <int WORK>
static void run(const uint4* t, uint32_t* out, uint32_t n_rec, int wg_per_cu) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256 * wg_per_cu, iters = 2000;
    const size_t lds = wg_per_cu >= 8 ? 0 : (size_t)(160 * 1024 / wg_per_cu) - 1024;   // caps the workgroups a CU holds
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<WORK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k<WORK>, dim3(blocks), dim3(256), lds, 0, t, out, n_rec, iters);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<WORK>, dim3(blocks), dim3(256), lds, 0, t, out, n_rec, iters); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double recs = (double)blocks * 256 * iters;
    printf("  %d workgroups/CU, %3d fma between fetches: %8.3f ms  %6.1f G records/s  %5.2f TB/s  (%.0f ns per dependent fetch per lane)\n",
           wg_per_cu, WORK, ms, recs / ms / 1e6, recs * 128 / ms / 1e9, ms * 1e6 / iters);
}

int main() {
    const size_t max_bytes = (size_t)512 << 20;
    uint4* t; uint32_t* out;
    CK(hipMalloc(&t, max_bytes)); CK(hipMalloc(&out, (size_t)256 * 8 * 256 * 4));
    CK(hipMemset(t, 0x5a, max_bytes));
    for (size_t kb : {16ull, 1024ull, 21ull * 1024, 166ull * 1024, 512ull * 1024}) {
        const uint32_t n_rec = (uint32_t)(kb * 1024 / 128);
        printf("table %6zu KB (%u records)\n", kb, n_rec);
        for (int wg : {8, 5, 4}) {
            run<0>(t, out, n_rec, wg);
            run<100>(t, out, n_rec, wg);
            run<165>(t, out, n_rec, wg);
        }
    }
    return 0;
}

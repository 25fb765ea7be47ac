// Does the vector L1 care how a lane's 64-byte record is fetched?  A: every lane reads its own
// record with four dwordx4 loads (64 different lines per instruction).  B: the four lanes of a
// quad read the four 16-byte pieces of one record per instruction (16 different lines per
// instruction, each read whole), four instructions for the quad's four records.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint64_t mix(uint64_t h) { h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32; return h; }
template <int MODE>
__global__ void __launch_bounds__(256) k(const uint4* table, uint32_t* out, uint32_t n_rec, int iters) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, q = lane & 3u;
    uint32_t acc = 0;
    uint64_t h = mix(tid + 1);
    for (int it = 0; it < iters; it++) {
        h = mix(h + it);
        const uint32_t rec = (uint32_t)(h % n_rec);
        if (MODE == 0) {
            const uint4* p = table + (size_t)rec * 4;
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x ^ b.y ^ c.z ^ d.w;
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const uint32_t r = (uint32_t)__shfl((int)rec, (int)((lane & ~3u) | (uint32_t)kk));
                const uint4 v = table[(size_t)r * 4 + q];
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    out[tid] = acc;
}
template <int MODE> void run(const char* name, const uint4* t, uint32_t* out, uint32_t n_rec) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, t, out, n_rec, iters);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, t, out, n_rec, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    const double recs = (double)blocks * 256 * iters;
    printf("%-28s table %6u KB: %.3f ms, %.1f G records/s, %.2f TB/s\n", name, (unsigned)((uint64_t)n_rec * 64 >> 10), ms, recs / ms / 1e6, recs * 64 / ms / 1e9);
}
int main() {
    for (uint32_t n_rec : {1u << 8, 1u << 11, 1u << 14, 1u << 19, 1u << 21}) {   // 16 KB (L1), 128 KB, 1 MB (L2), 32 MB (L2/MALL), 128 MB (MALL)
        uint4* t; uint32_t* out; hipMalloc(&t, (size_t)n_rec * 64); hipMalloc(&out, 256 * 8 * 256 * 4);
        hipMemset(t, 1, (size_t)n_rec * 64);
        run<0>("own record, 4 x dwordx4", t, out, n_rec);
        run<1>("quad-cooperative", t, out, n_rec);
        hipFree(t); hipFree(out);
    }
    return 0;
}

// Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS code's access pattern: every lane
// reads one whole 64-byte record (4 x dwordx4) at a random index of a table much larger than
// the Infinity Cache, and writes 16 bytes of another.  Known bytes: n * 64 read, n * 16 written.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(256) gather64(const uint4* table, uint4* out, uint64_t n_rec, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t h = i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const uint64_t r = h % n_rec;
    const uint4 a = table[r * 4], b = table[r * 4 + 1], c = table[r * 4 + 2], d = table[r * 4 + 3];
    uint4 s; s.x = a.x ^ b.y ^ c.z ^ d.w; s.y = a.y + b.z; s.z = c.x + d.y; s.w = a.w;
    out[i] = s;
}
int main() {
    const uint64_t n_rec = 1ull << 26;   // 64 M records x 64 B = 4 GiB
    const uint64_t n = 1ull << 26;       // 64 M lanes
    uint4 *table, *out;
    hipMalloc(&table, n_rec * 64); hipMalloc(&out, n * 16);
    hipMemset(table, 1, n_rec * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(gather64, dim3((unsigned)(n / 256)), dim3(256), 0, 0, table, out, n_rec, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("gather64: %.3f ms, read %.3f GB (%.1f GB/s), written %.3f GB\n", ms, n * 64 / 1e9, n * 64 / 1e9 / (ms * 1e-3), n * 16 / 1e9);
    }
    return 0;
}

// Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS code's access patterns: every lane reads one
// whole record of 64, 128 or 192 bytes (4, 8 or 12 x dwordx4 -- a RaySlot line, a compact wide record,
// a whole pool slot) at a random index of a 4-8 GiB table, far larger than the 256 MiB Infinity Cache,
// and writes 16 bytes of another buffer in order.  Known bytes per kernel: n * REC read, n * 16 written.
// Run each size under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) and compare:
// scripts/fetch_calib.sh does that and writes profiles/rNN_fetch_calibration.json.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int Q>  // 16-byte pieces per record
__global__ void __launch_bounds__(256) gather(const uint4* table, uint4* out, uint64_t n_rec, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t h = i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const uint64_t r = h % n_rec;
    uint4 s = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const uint4 a = table[r * Q + q];
        s.x ^= a.x, s.y += a.y, s.z ^= a.z, s.w += a.w;
    }
    out[i] = s;
}
template <int Q>
void run(uint4* table, uint4* out, uint64_t table_bytes, uint64_t n) {
    const uint64_t n_rec = table_bytes / (16ull * Q);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(gather<Q>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, table, out, n_rec, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("gather%d: %.3f ms, read %llu bytes (%.1f GB/s), written %llu bytes\n", 16 * Q, ms,
           (unsigned long long)(n * 16 * Q), n * 16.0 * Q / 1e9 / (ms * 1e-3), (unsigned long long)(n * 16));
}
int main() {
    const uint64_t table_bytes = 6ull << 30;  // 6 GiB: divisible by 64, 128 and 192
    const uint64_t n = 1ull << 25;            // 32 M lanes
    uint4 *table, *out;
    if (hipMalloc(&table, table_bytes) != hipSuccess || hipMalloc(&out, n * 16) != hipSuccess) return 1;
    hipMemset(table, 1, table_bytes);
    run<4>(table, out, table_bytes, n);
    run<8>(table, out, table_bytes, n);
    run<12>(table, out, table_bytes, n);
    return 0;
}

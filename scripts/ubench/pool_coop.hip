// The same bytes as pool_rw.hip at 100 % density, moved cooperatively: a wave reads 64 consecutive slots (12 KiB) with
// twelve perfectly coalesced dwordx4 loads, stages them in LDS, every lane reads ITS slot from LDS (12 x ds_read_b128),
// and the first 128 bytes of each slot go back the same way.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(256, 2) pool_coop(uint4* pool, uint32_t n_chunks) {
    __shared__ uint4 stage[4][64 * 12 + 12];
    uint4* st = stage[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t c = wave; c < n_chunks; c += n_waves) {
        uint4* base = pool + (size_t)c * 64 * 12;
        uint4 v[12];
#pragma unroll
        for (int q = 0; q < 12; q++) v[q] = base[q * 64 + lane];          // coalesced: 1 KiB per instruction
#pragma unroll
        for (int q = 0; q < 12; q++) st[q * 64 + lane] = v[q];            // granule g of the chunk at st[g]
        // lane l's slot = granules 12 l .. 12 l + 11
        uint4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 12; q++) {
            const uint4 r = st[lane * 12 + q];
            acc.x ^= r.x, acc.y += r.y, acc.z ^= r.z, acc.w += r.w;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) st[lane * 12 + q] = acc;              // new first 128 bytes of the slot
#pragma unroll
        for (int q = 0; q < 12; q++) {                                     // write back coalesced, only granules of the first 128 B
            const uint32_t g = q * 64 + lane;
            if (g % 12 < 8) base[g] = st[g];
        }
    }
}
int main() {
    const uint32_t np = 1u << 25, n_chunks = np / 64;
    uint4* pool;
    if (hipMalloc(&pool, (size_t)np * 192) != hipSuccess) return 1;
    (void)hipMemset(pool, 1, (size_t)np * 192);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks_per_cu = 2; blocks_per_cu <= 4; blocks_per_cu *= 2)
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(pool_coop, dim3(256 * blocks_per_cu), dim3(256), 0, 0, pool, n_chunks);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("cooperative, grid %d blocks/CU: %.3f ms, %.2f TB/s useful (192 B read + 128 B written per slot)\n",
                            blocks_per_cu, ms, (double)np * 320.0 / (ms * 1e-3) / 1e12);
        }
    return 0;
}

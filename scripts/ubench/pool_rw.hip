// What rate does the hit kernel's MEMORY pattern reach by itself?  32 M slots of 192 bytes, a random 58 % of them
// "HIT"; every wave takes windows of 512 slots, compacts the HIT ones (ballot + rank, as wavefront.hip does) and
// for each of them reads the whole slot (12 x dwordx4) and writes its first 128 bytes back (8 x dwordx4) -- no
// arithmetic to speak of.  Compare with wf_hit_kernel's 3.6 TB/s of fabric traffic at 40 % VALU occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr uint32_t WINDOW = 512;
__global__ void __launch_bounds__(256, 2) pool_rw(uint4* pool, const uint8_t* state, uint32_t n_windows, int depth) {
    __shared__ uint16_t lists[4][WINDOW];
    uint16_t* list = lists[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t win = wave; win < n_windows; win += n_waves) {
        const uint32_t* sw = reinterpret_cast<const uint32_t*>(state + (size_t)win * WINDOW) + lane * 2;
        const uint32_t w0 = sw[0], w1 = sw[1];
        uint32_t count = 0;
        for (int j = 0; j < 8; j++) {
            const uint32_t s = ((j < 4 ? w0 : w1) >> ((j & 3) * 8)) & 0xffu;
            const unsigned long long mask = __ballot(s == 2u);
            if (s == 2u) list[count + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = (uint16_t)(lane * 8 + j);
            count += (uint32_t)__popcll(mask);
        }
        for (uint32_t k = 0; k < count; k += 64u * depth) {
            uint4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
            uint32_t slot[2];
            for (int d = 0; d < depth; d++) {
                const uint32_t idx = k + d * 64u + lane;
                slot[d] = idx < count ? win * WINDOW + list[idx] : 0xffffffffu;
                if (slot[d] != 0xffffffffu) {
                    const uint4* p = pool + (size_t)slot[d] * 12;
#pragma unroll
                    for (int q = 0; q < 12; q++) {
                        const uint4 v = p[q];
                        acc[d].x ^= v.x, acc[d].y += v.y, acc[d].z ^= v.z, acc[d].w += v.w;
                    }
                }
            }
            for (int d = 0; d < depth; d++)
                if (slot[d] != 0xffffffffu) {
                    uint4* p = pool + (size_t)slot[d] * 12;
#pragma unroll
                    for (int q = 0; q < 8; q++) p[q] = acc[d];
                }
        }
    }
}
int main() {
    const uint32_t np = 1u << 25, n_windows = np / WINDOW;
    uint4* pool; uint8_t* state;
    if (hipMalloc(&pool, (size_t)np * 192) != hipSuccess || hipMalloc(&state, np) != hipSuccess) return 1;
    (void)hipMemset(pool, 1, (size_t)np * 192);
    std::vector<uint8_t> h(np);
    uint64_t x = 88172645463325252ull; uint64_t hits = 0;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int density = 40; density <= 100; density += 20) {
    hits = 0;
    for (uint32_t i = 0; i < np; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (int)(x % 100) < density ? 2 : 3; hits += h[i] == 2; }
    (void)hipMemcpy(state, h.data(), np, hipMemcpyHostToDevice);
    printf("== density %d %%\n", density);
    for (int depth = 1; depth <= 1; depth++)
        for (int blocks_per_cu = 2; blocks_per_cu <= 2; blocks_per_cu *= 2)
            for (int rep = 0; rep < 2; rep++) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(pool_rw, dim3(256 * blocks_per_cu), dim3(256), 0, 0, pool, state, n_windows, depth);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("depth %d, grid %d blocks/CU: %.3f ms, %.2f TB/s useful (192 B read + 128 B written per hit, %llu hits)\n",
                                depth, blocks_per_cu, ms, hits * 320.0 / (ms * 1e-3) / 1e12, (unsigned long long)hits);
            }
    }
    return 0;
}

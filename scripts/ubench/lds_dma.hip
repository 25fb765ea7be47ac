// Does global_load_lds_dwordx4 put lane l's 16 bytes at lds_base + l * 16 (per wave), for scattered per-lane sources?
// hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/lds_dma scripts/ubench/lds_dma.hip && scripts/ubench/lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint4* src, const uint32_t* idx, uint4* dst) {
    __shared__ uint4 buf[4][2][64];  // wave, granule, lane
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint4* p = src + (size_t)idx[blockIdx.x * 256 + threadIdx.x] * 8;   // a 128-byte record per lane
    __builtin_amdgcn_global_load_lds(p, &buf[wave][0][0], 16, 0, 0);
    __builtin_amdgcn_global_load_lds(p + 3, &buf[wave][1][0], 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint4 a = buf[wave][0][lane], b = buf[wave][1][lane];
    dst[(blockIdx.x * 256 + threadIdx.x) * 2] = a;
    dst[(blockIdx.x * 256 + threadIdx.x) * 2 + 1] = b;
}
int main() {
    const int n = 1 << 16, blocks = 64;
    std::vector<uint4> h(n * 8);
    for (int i = 0; i < n * 8; i++) h[i] = make_uint4(i, i ^ 0x5555, i * 3, ~i);
    std::vector<uint32_t> idx(blocks * 256);
    for (size_t i = 0; i < idx.size(); i++) idx[i] = (uint32_t)((i * 2654435761u) % n);
    uint4 *d_src, *d_dst; uint32_t* d_idx;
    hipMalloc(&d_src, h.size() * 16); hipMalloc(&d_dst, idx.size() * 32); hipMalloc(&d_idx, idx.size() * 4);
    hipMemcpy(d_src, h.data(), h.size() * 16, hipMemcpyHostToDevice);
    hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_src, d_idx, d_dst);
    std::vector<uint4> out(idx.size() * 2);
    hipMemcpy(out.data(), d_dst, out.size() * 16, hipMemcpyDeviceToHost);
    long bad = 0;
    for (size_t i = 0; i < idx.size(); i++) {
        const uint4 a = h[(size_t)idx[i] * 8], b = h[(size_t)idx[i] * 8 + 3];
        bad += out[2 * i].x != a.x || out[2 * i].w != a.w || out[2 * i + 1].y != b.y || out[2 * i + 1].z != b.z;
    }
    printf("lds dma: %ld of %zu lanes wrong\n", bad, idx.size());
    return bad != 0;
}

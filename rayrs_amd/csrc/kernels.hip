// kernels.hip -- the resolve kernel (per-pixel sum of the item sums) and the
// single-function self-test kernels.  The path pipeline itself is wavefront.hip.
#include <hip/hip_runtime.h>

#include "device_path.h"
#include "kernels.h"

namespace rayrs {

// Adds the chunk sums of each pixel in chunk order, applies pixel / spp
// (main.rs:89; Div<f64> = multiply by 1/spp) and writes the framebuffer.
__global__ void __launch_bounds__(256) resolve_kernel(CameraDev cam, RenderDev rp, uint32_t lt0, uint32_t n_lt) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = (uint64_t)n_lt * 64u;  // the rank's tiles lt0 .. lt0 + n_lt - 1
    if (idx >= n) return;
    const uint32_t pit = (uint32_t)(idx & 63u);
    const uint32_t lt = lt0 + (uint32_t)(idx >> 6);
    const uint32_t tile = lt * rp.tile_ranks + rp.tile_rank;
    const uint32_t row = (tile / rp.tiles_x) * 8u + (pit >> 3);
    const uint32_t col = (tile % rp.tiles_x) * 8u + (pit & 7u);
    if (row >= cam.H || col >= cam.W) return;
    double x = 0.0, y = 0.0, z = 0.0;
    for (uint32_t k = 0; k < rp.nchunks; k++) {
        const double* src = rp.partial + ((((size_t)lt * rp.nchunks + k) * 64u + pit) - rp.partial_item0) * 3;
        if (k == 0) {
            x = src[0], y = src[1], z = src[2];
        } else {
            x += src[0], y += src[1], z += src[2];
        }
    }
    if (x != x || y != y || z != z) atomicAdd(&rp.counters->nan_pixels, 1ull);           // main.rs:81
    if (x < 0.0 || y < 0.0 || z < 0.0) atomicAdd(&rp.counters->neg_pixels, 1ull);        // main.rs:85
    const double inv = 1.0 / (double)rp.spp;
    x *= inv, y *= inv, z *= inv;
    const size_t pix = (size_t)row * cam.W + col;
    if (rp.out_format == RAYRS_OUT_F64) {
        double* dst = reinterpret_cast<double*>(rp.out) + pix * 3;
        dst[0] = x, dst[1] = y, dst[2] = z;
    } else {
        float* dst = reinterpret_cast<float*>(rp.out) + pix * 3;  // image.rs:224-229
        dst[0] = (float)x, dst[1] = (float)y, dst[2] = (float)z;
    }
}

// dst += src, element by element (rayrs_render_multi: ranks rehearsed on one device; every element is
// non-zero in at most one of the two, so the sum is exact)
template <typename T>
__global__ void __launch_bounds__(256) accumulate_kernel(T* dst, const T* src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

hipError_t launch_accumulate(void* dst, const void* src, size_t n, bool f64, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (f64)
        hipLaunchKernelGGL(accumulate_kernel<double>, dim3(blocks), dim3(256), 0, stream, (double*)dst, (const double*)src, n);
    else
        hipLaunchKernelGGL(accumulate_kernel<float>, dim3(blocks), dim3(256), 0, stream, (float*)dst, (const float*)src, n);
    return hipGetLastError();
}

// ------------------------------------------------------------ launch glue

static inline uint32_t lds_bytes_for(uint32_t stack_depth) { return 4u * 64u * (stack_depth + 1u) * 4u; }  // + the spare entry

hipError_t launch_resolve(const CameraDev& cam, const RenderDev& rp, uint32_t lt0, uint32_t n_lt, hipStream_t stream) {
    const uint64_t n = (uint64_t)n_lt * 64u;
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(resolve_kernel, dim3(blocks), dim3(256), 0, stream, cam, rp, lt0, n_lt);
    return hipGetLastError();
}

// ------------------------------------------------------- self-test kernels

__global__ void test_math_kernel(int fn, const double* x, const double* y, uint64_t n, double* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i], b = y ? y[i] : 0.0;
    double r;
    switch (fn) {
        case 0: r = rr_sin(a); break;
        case 1: r = rr_cos(a); break;
        case 2: r = rr_tan(a); break;
        case 3: r = rr_log(a); break;
        case 4: r = rr_exp(a); break;
        case 5: r = rr_acos(a); break;
        case 6: r = rr_atan2(a, b); break;
        case 7: r = rr_sqrt(a); break;
        case 8: r = a / b; break;
        case 9:    // div3_by (device_path.h): the three quotients a / b, (a * 0x1p-600) / b, (-3 a) / b, one per call
        case 10:
        case 11: {
            double q0, q1, q2;
            div3_by(a, a * 0x1p-600, -3.0 * a, b, q0, q1, q2);
            r = fn == 9 ? q0 : fn == 10 ? q1 : q2;
            break;
        }
        default: r = 0.0; break;
    }
    out[i] = r;
}

__global__ void test_rng_kernel(uint64_t seed, const uint64_t* pixel, const uint64_t* sample, const uint32_t* draw,
                                uint64_t n, uint64_t* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = rr_draw_bits(rr_path_key(seed, pixel[i], sample[i]), draw[i]);
}

template <bool COMPACT>
__global__ void __launch_bounds__(256) test_intersect_kernel(SceneDev sc, const double* o, const double* d, uint64_t n,
                                                             double* t_out, long long* prim_out, uint32_t* spill) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // sc.stack_lds entries of the stack in LDS, the rest in the strip `spill` (as in the traversal kernel)
    const LaneStack stack{lds_stack + (size_t)wave * (sc.stack_lds + 1u) * 64u + lane, spill + i, sc.stack_lds,
                          gridDim.x * blockDim.x};
    if (i >= n) return;
    double t = 0.0;
    uint32_t prim = 0;
    WorkCount wc{0, 0, 0, 0, 0};
    const V3 ro = mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    const bool hit = sc.exact ? bvh_intersect<COMPACT, false, true>(sc, ro, rd, stack, t, prim, wc)
                              : bvh_intersect<COMPACT, false, false>(sc, ro, rd, stack, t, prim, wc);
    t_out[i] = hit ? t : 0.0;
    prim_out[i] = hit ? (long long)prim : -1ll;
}

hipError_t launch_test_intersect(bool compact, const SceneDev& sc, const double* o, const double* d, uint64_t n,
                                 double* t_out, long long* prim_out, uint32_t* spill, hipStream_t stream) {
    const uint32_t lds = lds_bytes_for(sc.stack_lds);
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (compact) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&test_intersect_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(test_intersect_kernel<true>, dim3(blocks), dim3(256), lds, stream, sc, o, d, n, t_out,
                           prim_out, spill);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&test_intersect_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(test_intersect_kernel<false>, dim3(blocks), dim3(256), lds, stream, sc, o, d, n, t_out,
                           prim_out, spill);
    }
    return hipGetLastError();
}

// One sample exactly as the path kernels run it -- the path key, the primary ray (wavefront.hip next_sample), then the
// loop of lib.rs:521-560 with the device functions of device_path.h -- in ONE lane from the first ray to the last, with
// its trace: per loop iteration the primitive the query found (0xffffffff: none), its t, and the throughput and the
// RNG's draw index on leaving the iteration (rayrs_selftest.h rayrs_test_path_trace; the CPU checker keeps the same).
template <bool COMPACT>
__global__ void __launch_bounds__(256) test_path_trace_kernel(SceneDev sc, CameraDev cam, uint64_t seed, uint32_t max_bounces,
                                                              const uint32_t* pix, const uint32_t* sample, uint64_t n,
                                                              uint32_t cap, uint32_t* n_out, uint32_t* prim_out, double* t_out,
                                                              double* thr_out, uint32_t* draw_out, double* rgb_out,
                                                              uint32_t* spill) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const LaneStack stack{lds_stack + (size_t)wave * (sc.stack_lds + 1u) * 64u + lane, spill + i, sc.stack_lds,
                          gridDim.x * blockDim.x};
    if (i >= n) return;
    const uint32_t row = pix[i] >> 16, col = pix[i] & 0xffffu;
    Rng rng{rr_path_key(seed, (uint64_t)row * cam.W + col, (uint64_t)sample[i]), 0};
    V3 o, d;
    primary_ray(cam, cam.H - row, cam.W - col, rng, o, d);  // image origin is upper left, camera origin lower right (main.rs:74-75)
    V3 thr = mk(1.0, 1.0, 1.0), light = mk(0.0, 0.0, 0.0), result = mk(0.0, 0.0, 0.0);
    uint32_t b = 0;
    bool returned = false;
    auto put = [&](uint32_t prim, double t) {
        if (b < cap) {
            const size_t at = (size_t)i * cap + b;
            prim_out[at] = prim, t_out[at] = t, draw_out[at] = rng.draw;
            thr_out[3 * at] = thr.x, thr_out[3 * at + 1] = thr.y, thr_out[3 * at + 2] = thr.z;
        }
    };
    for (; b < max_bounces && !returned; b++) {
        double t = 0.0;
        uint32_t prim = 0xffffffffu;
        WorkCount wc{0, 0, 0, 0, 0};
        const bool hit = sc.exact ? bvh_intersect<COMPACT, false, true>(sc, o, d, stack, t, prim, wc)
                                  : bvh_intersect<COMPACT, false, false>(sc, o, d, stack, t, prim, wc);
        if (!hit) {
            put(0xffffffffu, 0.0);
            result = v_add(light, v_mul(thr, background(sc, d)));  // lib.rs:555
            returned = true;
            continue;
        }
        const PrimRec<COMPACT> rec = load_prim<COMPACT>(sc.prims, prim);
        const V3 position = v_add(o, v_scale(d, t));
        const V3 normal = prim_normal<COMPACT>(rec, position);
        const V3 view = v_unit(v_scale(d, -1.0));
        const SurfaceDev* surf = sc.surfaces + (rec.tag() >> 8);
        const Scatter ev = material_evaluate(surf, normal, view, rng);
        if (!ev.scatter) {  // lib.rs:550
            put(prim, t);
            result = light, returned = true;
            continue;
        }
        light = v_add(light, v_mul(thr, mk(surf->emit[0], surf->emit[1], surf->emit[2])));
        thr = v_mul(thr, ev.color);
        const double p = rr_max(rr_max(thr.x, thr.y), thr.z);
        if (rng.next() > p) {
            put(prim, t);
            result = light, returned = true;
            continue;
        }
        thr = mk(thr.x / p, thr.y / p, thr.z / p);  // DivAssign, vecmath.rs:708-714
        o = position, d = ev.dir;
        put(prim, t);
    }
    if (!returned) result = light;  // lib.rs:559
    n_out[i] = b;
    rgb_out[3 * i] = result.x, rgb_out[3 * i + 1] = result.y, rgb_out[3 * i + 2] = result.z;
}

hipError_t launch_test_path_trace(bool compact, const SceneDev& sc, const CameraDev& cam, uint64_t seed, uint32_t max_bounces,
                                  const uint32_t* pix, const uint32_t* sample, uint64_t n, uint32_t cap, uint32_t* n_out,
                                  uint32_t* prim_out, double* t_out, double* thr_out, uint32_t* draw_out, double* rgb_out,
                                  uint32_t* spill, hipStream_t stream) {
    const uint32_t lds = lds_bytes_for(sc.stack_lds);
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (compact) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&test_path_trace_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(test_path_trace_kernel<true>, dim3(blocks), dim3(256), lds, stream, sc, cam, seed, max_bounces, pix,
                           sample, n, cap, n_out, prim_out, t_out, thr_out, draw_out, rgb_out, spill);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&test_path_trace_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(test_path_trace_kernel<false>, dim3(blocks), dim3(256), lds, stream, sc, cam, seed, max_bounces, pix,
                           sample, n, cap, n_out, prim_out, t_out, thr_out, draw_out, rgb_out, spill);
    }
    return hipGetLastError();
}

__global__ void test_material_kernel(const SurfaceDev* surf, const double* normal, const double* view,
                                     const uint64_t* key, uint64_t n, int32_t* scattered, double* color, double* dir,
                                     uint32_t* draws) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng rng{key[i], 0};
    const Scatter ev = material_evaluate(surf, mk(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]),
                                         mk(view[3 * i], view[3 * i + 1], view[3 * i + 2]), rng);
    scattered[i] = ev.scatter ? 1 : 0;
    color[3 * i] = ev.color.x, color[3 * i + 1] = ev.color.y, color[3 * i + 2] = ev.color.z;
    dir[3 * i] = ev.dir.x, dir[3 * i + 1] = ev.dir.y, dir[3 * i + 2] = ev.dir.z;
    draws[i] = rng.draw;
}

__global__ void test_background_kernel(SceneDev sc, const double* dir, uint64_t n, double* rgb) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 c = background(sc, mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
    rgb[3 * i] = c.x, rgb[3 * i + 1] = c.y, rgb[3 * i + 2] = c.z;
}

hipError_t launch_test_math(int fn, const double* x, const double* y, uint64_t n, double* out, hipStream_t stream) {
    hipLaunchKernelGGL(test_math_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, fn, x, y, n, out);
    return hipGetLastError();
}
hipError_t launch_test_rng(uint64_t seed, const uint64_t* pixel, const uint64_t* sample, const uint32_t* draw,
                           uint64_t n, uint64_t* out, hipStream_t stream) {
    hipLaunchKernelGGL(test_rng_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, seed, pixel, sample,
                       draw, n, out);
    return hipGetLastError();
}
hipError_t launch_test_material(const SurfaceDev* surf, const double* normal, const double* view, const uint64_t* key,
                                uint64_t n, int32_t* scattered, double* color, double* dir, uint32_t* draws,
                                hipStream_t stream) {
    hipLaunchKernelGGL(test_material_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, surf, normal,
                       view, key, n, scattered, color, dir, draws);
    return hipGetLastError();
}
hipError_t launch_test_background(const SceneDev& sc, const double* dir, uint64_t n, double* rgb, hipStream_t stream) {
    hipLaunchKernelGGL(test_background_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, sc, dir, n,
                       rgb);
    return hipGetLastError();
}

}  // namespace rayrs

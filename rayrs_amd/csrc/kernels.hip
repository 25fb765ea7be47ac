// kernels.hip -- gfx950 kernels: the persistent path-tracing kernel, the resolve
// kernel and the single-function self-test kernels.
//
// trace_kernel: persistent waves.  A work item is (8x8 tile, sample chunk, pixel
// in tile); 64 consecutive items are the 64 pixels of one tile for one chunk of
// samples, so the lanes of a wave start from neighbouring pixels.  Each lane owns
// one item at a time and runs its samples one after the other (a finished path is
// replaced in place by the lane's next sample, so no lane waits for another
// lane's path), and a lane whose item is finished takes the next unclaimed item
// from the wave's pool: __ballot finds the idle lanes, the popcount of the lower
// lanes ranks them, and the pool is refilled 64 items at a time with one atomic
// on the device-wide queue head.  The per-lane traversal stack lives in LDS
// (entry k of lane l at word k*64 + l: conflict-free).  Path state is f64 in
// registers.
#include <hip/hip_runtime.h>

#include "device_path.h"
#include "kernels.h"

namespace rayrs {

// Lane states of the path state machine.
constexpr uint32_t ST_IDLE = 0;   // no path: needs its item's next sample, a new item, or is out of work
constexpr uint32_t ST_TRAV = 1;   // a BVH query is in progress (Trav holds its state)
constexpr uint32_t ST_SHADE = 2;  // the query has finished and waits for the shading phase

// Shading runs for the waiting lanes once fewer than TRAV_MIN lanes of the wave
// are still traversing; until then the traversing lanes keep taking macro steps.
constexpr int TRAV_MIN = 40;
// A leaf phase runs once this many lanes stand on a leaf (or none is on an interior record).
constexpr int LEAF_MIN = 16;

template <bool COMPACT, bool COUNT>
__global__ void __launch_bounds__(256, 2) trace_kernel(SceneDev sc, CameraDev cam, RenderDev rp) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t* stack = lds_stack + (size_t)wave * sc.stack_depth * 64u + lane;
    const unsigned long long lanemask_lt = (1ull << lane) - 1ull;

    // wave-uniform pool of claimed items
    unsigned long long pool_next = 0, pool_end = 0;

    // lane state
    uint32_t state = ST_IDLE;
    bool dead = false, has_item = false;
    uint32_t item = 0, row = 0, col = 0, s_cur = 0, s_end = 0, bounce = 0;
    double acc_x = 0, acc_y = 0, acc_z = 0;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1), thr = mk(1, 1, 1), light = mk(0, 0, 0);
    Rng rng{0, 0};
    Trav tv;
    tv.inv = mk(0, 0, 0), tv.best_t = 0, tv.best_prim = 0xffffffffu, tv.cur = TRAV_DONE, tv.sp = 0;
    unsigned long long n_rays = 0, n_paths = 0, n_escaped = 0;
    WorkCount wc{0, 0, 0, 0, 0};
    unsigned long long wc_int = 0, wc_tri = 0, wc_sph = 0, wc_pln = 0;
    unsigned long long u_int_wave = 0, u_int_lane = 0, u_leaf_wave = 0, u_leaf_lane = 0, u_shade_wave = 0,
                       u_shade_lane = 0;

    for (;;) {
        // wave-uniform scheduling decision
        const bool at_int = state == ST_TRAV && trav_at_interior(tv);
        const bool at_leaf = state == ST_TRAV && !trav_at_interior(tv);
        const int n_int = __popcll(__ballot(at_int));
        const int n_leaf = __popcll(__ballot(at_leaf));
        if (n_int + n_leaf >= TRAV_MIN) {
            if (n_leaf >= LEAF_MIN || n_int == 0) {
                // ---- leaf phase: every lane standing on a leaf tests its primitives
                if (COUNT) u_leaf_wave += 1, u_leaf_lane += at_leaf ? 1 : 0;
                if (at_leaf) {
                    trav_leaf_step<COMPACT, COUNT>(sc, o, d, stack, tv, wc);
                    if (tv.cur == TRAV_DONE) state = ST_SHADE;
                }
            } else {
                // ---- interior phase: one record for every lane standing on one
                if (COUNT) u_int_wave += 1, u_int_lane += at_int ? 1 : 0;
                if (at_int) {
                    trav_interior_step<COMPACT, COUNT>(sc, o, stack, tv, wc);
                    if (tv.cur == TRAV_DONE) state = ST_SHADE;
                }
            }
            continue;
        }
        {

            if (COUNT) {
                if (__ballot(state == ST_SHADE) != 0ull) u_shade_wave++;
                if (state == ST_SHADE) u_shade_lane++;
            }
            // ---- shade the finished queries: the rest of one radiance() iteration (lib.rs:526-556)
            if (state == ST_SHADE) {
                bool finished = false;
                V3 result = light;
                if (tv.best_prim != 0xffffffffu) {
                    const PrimRec<COMPACT> rec = load_prim<COMPACT>(sc.prims, tv.best_prim);
                    const V3 position = v_add(o, v_scale(d, tv.best_t));
                    const V3 normal = prim_normal<COMPACT>(rec, position);
                    const V3 view = v_unit(v_scale(d, -1.0));
                    const SurfaceDev* surf = sc.surfaces + (rec.tag() >> 8);
                    const Scatter ev = material_evaluate(surf, normal, view, rng);
                    if (ev.scatter) {
                        light = v_add(light, v_mul(thr, mk(surf->emit[0], surf->emit[1], surf->emit[2])));
                        thr = v_mul(thr, ev.color);
                        const double p = rr_max(rr_max(thr.x, thr.y), thr.z);
                        if (rng.next() > p) {
                            finished = true;
                            result = light;
                        } else {
                            thr = mk(thr.x / p, thr.y / p, thr.z / p);  // DivAssign, vecmath.rs:708-714
                            o = position;
                            d = ev.dir;
                        }
                    } else {
                        finished = true;  // lib.rs:550
                        result = light;
                    }
                } else {
                    n_escaped++;
                    finished = true;
                    result = v_add(light, v_mul(thr, background(sc, d)));  // lib.rs:555
                }
                if (!finished && bounce >= rp.max_bounces) {  // loop bound of lib.rs:525; lib.rs:559
                    finished = true;
                    result = light;
                }
                if (finished) {
                    acc_x += result.x;  // main.rs:69
                    acc_y += result.y;
                    acc_z += result.z;
                    state = ST_IDLE;
                } else {
                    bounce++;
                    n_rays++;
                    trav_init(sc, o, d, tv);
                    state = tv.cur == TRAV_DONE ? ST_SHADE : ST_TRAV;
                }
            }
            // ---- A: an item whose samples are all done is written out
            if (state == ST_IDLE && has_item && s_cur >= s_end) {
                double* dst = rp.partial + (size_t)item * 3;
                dst[0] = acc_x;
                dst[1] = acc_y;
                dst[2] = acc_z;
                has_item = false;
            }
            // ---- B: idle lanes take items from the wave's pool (ballot + rank)
            bool need = state == ST_IDLE && !has_item && !dead;
            unsigned long long need_mask = __ballot(need);
            while (need_mask != 0ull) {
                if (pool_next >= pool_end) {
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(&rp.counters->queue_head, 64ull);
                    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base);
                    const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
                    base = ((unsigned long long)bhi << 32) | blo;
                    if (base >= rp.total_items) {
                        if (need) dead = true;
                        break;
                    }
                    pool_next = base;
                    pool_end = base + 64ull < rp.total_items ? base + 64ull : rp.total_items;
                }
                const uint32_t avail = (uint32_t)(pool_end - pool_next);
                const uint32_t rank = (uint32_t)__popcll(need_mask & lanemask_lt);
                if (need && rank < avail) {
                    item = (uint32_t)(pool_next + rank);
                    need = false;
                    has_item = true;
                    const uint32_t pit = item & 63u;
                    const uint32_t tc = item >> 6;
                    const uint32_t chunk = tc % rp.nchunks;
                    const uint32_t tile = (tc / rp.nchunks) * rp.tile_ranks + rp.tile_rank;
                    row = (tile / rp.tiles_x) * 8u + (pit >> 3);
                    col = (tile % rp.tiles_x) * 8u + (pit & 7u);
                    s_cur = chunk * rp.chunk;
                    s_end = s_cur + rp.chunk < rp.spp ? s_cur + rp.chunk : rp.spp;
                    if (row >= cam.H || col >= cam.W) s_cur = s_end;  // padding pixel of an edge tile
                    acc_x = acc_y = acc_z = 0.0;
                }
                const uint32_t wanted = (uint32_t)__popcll(need_mask);
                pool_next += wanted < avail ? wanted : avail;
                need_mask = __ballot(need);
            }
            // ---- C: start the lane's next sample (main.rs:68-76)
            if (state == ST_IDLE && has_item && s_cur < s_end) {
                rng.key = rr_path_key(rp.seed, (uint64_t)row * cam.W + col, (uint64_t)s_cur);
                rng.draw = 0;
                // image origin is upper left, camera origin lower right (main.rs:74-75)
                primary_ray(cam, cam.H - row, cam.W - col, rng, o, d);
                thr = mk(1.0, 1.0, 1.0);
                light = mk(0.0, 0.0, 0.0);
                s_cur++;
                n_paths++;
                if (rp.max_bounces == 0) {  // radiance() with an empty loop returns zeros
                    state = ST_IDLE;
                } else {
                    bounce = 1;
                    n_rays++;
                    trav_init(sc, o, d, tv);
                    state = tv.cur == TRAV_DONE ? ST_SHADE : ST_TRAV;
                }
            }
            // all lanes out of work and nothing in flight
            if (__ballot(has_item || state != ST_IDLE) == 0ull) break;
        }
        // when few lanes are traversing, let them advance once per scheduling round as well
        if (state == ST_TRAV) {
            if (trav_at_interior(tv))
                trav_interior_step<COMPACT, COUNT>(sc, o, stack, tv, wc);
            else
                trav_leaf_step<COMPACT, COUNT>(sc, o, d, stack, tv, wc);
            if (tv.cur == TRAV_DONE) state = ST_SHADE;
        }
    }
    if (COUNT) wc_int += wc.interior, wc_tri += wc.tri, wc_sph += wc.sphere, wc_pln += wc.plane;

    Counters* c = rp.counters;
    if (n_rays) atomicAdd(&c->rays, n_rays);
    if (n_paths) atomicAdd(&c->paths, n_paths);
    if (COUNT) {
        if (n_escaped) atomicAdd(&c->escaped_paths, n_escaped);
        if (wc_int) atomicAdd(&c->interior_visits, wc_int);
        if (wc_tri) atomicAdd(&c->tri_tests, wc_tri);
        if (wc_sph) atomicAdd(&c->sphere_tests, wc_sph);
        if (wc_pln) atomicAdd(&c->plane_tests, wc_pln);
        atomicAdd(&c->step_wave, u_int_wave), atomicAdd(&c->step_lane, u_int_lane);
        atomicAdd(&c->inner_wave, u_leaf_lane), atomicAdd(&c->leaf_wave, u_leaf_wave);
        atomicAdd(&c->shade_wave, u_shade_wave), atomicAdd(&c->shade_lane, u_shade_lane);
    }
}

// Adds the chunk sums of each pixel in chunk order, applies pixel / spp
// (main.rs:89; Div<f64> = multiply by 1/spp) and writes the framebuffer.
__global__ void __launch_bounds__(256) resolve_kernel(CameraDev cam, RenderDev rp) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = (uint64_t)rp.n_local_tiles * 64u;
    if (idx >= n) return;
    const uint32_t pit = (uint32_t)(idx & 63u);
    const uint32_t lt = (uint32_t)(idx >> 6);
    const uint32_t tile = lt * rp.tile_ranks + rp.tile_rank;
    const uint32_t row = (tile / rp.tiles_x) * 8u + (pit >> 3);
    const uint32_t col = (tile % rp.tiles_x) * 8u + (pit & 7u);
    if (row >= cam.H || col >= cam.W) return;
    double x = 0.0, y = 0.0, z = 0.0;
    for (uint32_t k = 0; k < rp.nchunks; k++) {
        const double* src = rp.partial + (((size_t)lt * rp.nchunks + k) * 64u + pit) * 3;
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

// ------------------------------------------------------------ launch glue

static inline uint32_t lds_bytes_for(uint32_t stack_depth) { return 4u * 64u * stack_depth * 4u; }

template <bool COMPACT, bool COUNT>
static hipError_t launch_trace_t(const SceneDev& sc, const CameraDev& cam, const RenderDev& rp, uint32_t blocks,
                                 hipStream_t stream) {
    const uint32_t lds = lds_bytes_for(sc.stack_depth);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_kernel<COMPACT, COUNT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((trace_kernel<COMPACT, COUNT>), dim3(blocks), dim3(256), lds, stream, sc, cam, rp);
    return hipGetLastError();
}

hipError_t launch_trace(bool compact, bool count, const SceneDev& sc, const CameraDev& cam, const RenderDev& rp,
                        uint32_t blocks, hipStream_t stream) {
    if (compact) {
        return count ? launch_trace_t<true, true>(sc, cam, rp, blocks, stream)
                     : launch_trace_t<true, false>(sc, cam, rp, blocks, stream);
    }
    return count ? launch_trace_t<false, true>(sc, cam, rp, blocks, stream)
                 : launch_trace_t<false, false>(sc, cam, rp, blocks, stream);
}

uint32_t trace_lds_bytes(uint32_t stack_depth) { return lds_bytes_for(stack_depth); }

hipError_t trace_occupancy(bool compact, uint32_t stack_depth, int* blocks_per_cu) {
    const uint32_t lds = lds_bytes_for(stack_depth);
    if (compact) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_kernel<true, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, trace_kernel<true, false>, 256, lds);
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&trace_kernel<false, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, trace_kernel<false, false>, 256, lds);
}

hipError_t launch_resolve(const CameraDev& cam, const RenderDev& rp, hipStream_t stream) {
    const uint64_t n = (uint64_t)rp.n_local_tiles * 64u;
    if (n == 0) return hipSuccess;
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(resolve_kernel, dim3(blocks), dim3(256), 0, stream, cam, rp);
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
                                                             double* t_out, long long* prim_out) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t* stack = lds_stack + (size_t)wave * sc.stack_depth * 64u + lane;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double t = 0.0;
    uint32_t prim = 0;
    WorkCount wc{0, 0, 0, 0, 0};
    const bool hit = bvh_intersect<COMPACT, false>(sc, mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]),
                                                   mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]), stack, t, prim, wc);
    t_out[i] = hit ? t : 0.0;
    prim_out[i] = hit ? (long long)prim : -1ll;
}

hipError_t launch_test_intersect(bool compact, const SceneDev& sc, const double* o, const double* d, uint64_t n,
                                 double* t_out, long long* prim_out, hipStream_t stream) {
    const uint32_t lds = lds_bytes_for(sc.stack_depth);
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (compact) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&test_intersect_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(test_intersect_kernel<true>, dim3(blocks), dim3(256), lds, stream, sc, o, d, n, t_out,
                           prim_out);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&test_intersect_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(test_intersect_kernel<false>, dim3(blocks), dim3(256), lds, stream, sc, o, d, n, t_out,
                           prim_out);
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

// local_pool.hip -- the radiance integrator for small scenes: every path stays on its CU, in LDS, from
// the primary ray to the sample's end, and only the item sums reach HBM.
//
// wavefront.hip streams paths through a pool in HBM, three launches per bounce: that is what a deep BVH
// wants (traversal at five waves per SIMD, shading at two) and what a scene of a handful of primitives
// pays for without needing it.  When the walk tree is at most one record -- the reference's example
// scenes: a floor and one to seven spheres, test_scenes.rs:14-256 -- a BVH query is a loop over at most
// four gating boxes and sixteen primitives, all wave-uniform data.  So here one launch renders the frame:
//
//   * each wave owns a pool of LP_PATHS_PER_WAVE paths in LDS (structure of arrays: 12 doubles and
//     4 words per path, one state byte: 112 bytes, which is what lets three workgroups of four such
//     waves share a CU's 160 KiB) -- the reference keeps a path in locals for its whole life
//     (lib.rs:521-560); this is the same thing for 112 paths per wave;
//   * a wave repeatedly picks the phase most of its paths wait for, compacts up to 64 of them into its
//     lanes (__ballot + popcount rank, the list in LDS), loads their state, runs the phase on all lanes
//     and stores the state back:
//        GEN     finished item -> HBM, next sample or next item, primary ray      main.rs:67-79, lib.rs:202-210
//        BG      Scene::background, sample added to the item's sum, then GEN       lib.rs:254-285, :555
//        SHADE_k Material::evaluate of material kind k, emission, roulette        lib.rs:528-551
//     a wave in SHADE_k runs one arm of Material::evaluate (the kind is wave-uniform); GEN and SHADE_k end with the
//     new ray's query (ISECT: Bvh::intersect -- root box, gating boxes, primitives, bvh.rs:391-415) on the same lanes;
//   * an item (pixel, sample chunk) belongs to one path slot, which runs its samples one after the other,
//     so they are summed in order (main.rs:67-69); items come from a device-wide counter, a few dozen at
//     a time per wave.
// The arithmetic is device_path.h's, the same functions the streaming kernels call: frames are identical
// bit for bit on either route (tests/test_gpu_local_pool.py renders both and the oracle).
#include <hip/hip_runtime.h>

#include "device_path.h"
#include "local_pool.h"
#include "wavefront.h"

namespace rayrs {

namespace {

constexpr uint32_t P = LP_PATHS_PER_WAVE;
static_assert(P > 64 && P <= 128 && P % 16 == 0, "two state bytes per lane");

// a path's fields in LDS: field f of path p at [f * P + p]
// F_O: the ray's origin while the path waits for ISECT; the hit POSITION o + d * t once ISECT has found one
// (lib.rs:528: nothing else of o and t is used after the query).  What can be recomputed is not kept: the
// item's pixel and last sample follow from the item number, the sample's RNG key from pixel and sample.
enum { F_OX, F_OY, F_OZ, F_DX, F_DY, F_DZ, F_TX, F_TY, F_TZ, F_AX, F_AY, F_AZ, LP_NF64 };
enum { U_PRIM, U_BD, U_ITEM, U_SCUR, LP_NU32 };
constexpr uint32_t LP_WAVE_BYTES = P * (LP_NF64 * 8u + LP_NU32 * 4u) + P + 64u;  // fields, state bytes, list
static_assert(LP_WAVE_BYTES % 16u == 0, "pools stay 16-byte aligned");
constexpr uint32_t LP_PRIM_GRANULES = 5;  // 16-byte granules per primitive record in LDS (either layout)
// after the four pools: the LocalScene, n_prims + 4 primitive records (ISECT requests one record ahead, and never
// uses what lies beyond a group's end), n_surfaces surface rows -- sized per scene at launch
RR_DEV uint32_t lp_scene_bytes_dev(uint32_t n_prims) {
    return (uint32_t)sizeof(LocalScene) + (n_prims + 4u) * LP_PRIM_GRANULES * 16u;
}
static inline uint32_t lp_block_bytes(uint32_t n_prims, uint32_t n_surfaces) {
    return 4u * LP_WAVE_BYTES + (uint32_t)sizeof(LocalScene) + (n_prims + 4u) * LP_PRIM_GRANULES * 16u +
           n_surfaces * (uint32_t)sizeof(SurfaceDev);
}

// path states = the phase a path waits for
constexpr uint32_t LP_GEN = 0, LP_ISECT = 1, LP_BG = 2, LP_SHADE0 = 3;  // LP_SHADE0 + RAYRS_MAT_* (0..8)
constexpr uint32_t LP_NSTATE = 12, LP_DEAD = 12;

// U_BD: bounce (15 bits) | LP_DIRECT_BIT | draw << 16
constexpr uint32_t LP_BOUNCE_MASK = 0x7fffu;
constexpr uint32_t LP_DIRECT_BIT = 0x8000u;  // a primary ray that missed the root box: a query answered without a walk

// Items a wave takes from the device-wide counter at a time: LocalDev::reserve, between 8 and 256, about a
// sixteenth of a wave's share of the segment.  One counter word serves about 90 atomics per microsecond on this
// chip; at 32 items a grab the config-2 frame asked for 60 (each a round trip the wave waits for).

struct Pool {
    double* f64;
    uint32_t* u32;
    uint8_t* state;
    uint8_t* list;
    RR_DEV double& f(uint32_t field, uint32_t p) const { return f64[field * P + p]; }
    RR_DEV uint32_t& u(uint32_t field, uint32_t p) const { return u32[field * P + p]; }
    RR_DEV V3 v3(uint32_t field, uint32_t p) const { return mk(f(field, p), f(field + 1, p), f(field + 2, p)); }
    RR_DEV void set_v3(uint32_t field, uint32_t p, V3 v) const { f(field, p) = v.x, f(field + 1, p) = v.y, f(field + 2, p) = v.z; }
};

struct LpRange {  // a wave's reserved items of the segment [next, end), and whether the counter has run out
    uint32_t next, end;
    bool gone;
};

struct LpCount {  // per lane
    unsigned long long rays, paths, escaped, direct;
    uint32_t interior, tri, sphere, plane;
};

template <bool COMPACT>
RR_DEV PrimRec<COMPACT> load_prim_lds(const uint4* s_prims, uint32_t p) {
    PrimRec<COMPACT> r;
    const uint4* src = s_prims + p * LP_PRIM_GRANULES;
#pragma unroll
    for (int i = 0; i < (COMPACT ? 3 : 5); i++) r.q[i] = src[i];
    return r;
}

RR_DEV double* lp_light(const LocalDev& lp, uint32_t p) {
    const uint32_t wave_global = blockIdx.x * 4u + (threadIdx.x >> 6);
    return lp.light + ((size_t)wave_global * P + p) * 4u;
}

// ---- GEN: the path in this slot has ended (or the slot never had one).  Write the item out if its samples are
// all done, take the next sample -- or the next item -- and make its primary ray (main.rs:67-79, lib.rs:202-210).
// A primary ray that misses the box of the BVH's root Node is a Miss before anything else is looked at
// (bvh.rs:394): such a path goes straight to BG.
RR_DEV uint32_t lp_gen(const Pool& pl, bool valid, uint32_t p, const SceneDev& sc, const CameraDev& cam,
                       const RenderDev& rp, const LocalDev& lp, LpRange& range, LpCount& n, V3& ray_o, V3& ray_d,
                       V3& ray_inv) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = pl.u(U_SCUR, p);
    bool has_item = valid && (w >> 31) != 0u;
    uint32_t s_cur = w & SLOT_SAMPLE_MASK, item = pl.u(U_ITEM, p);
    uint32_t row, col, s_end, s_first;
    item_geometry(rp, item, row, col, s_first, s_end);  // (of no meaning without an item)
    if (has_item && s_cur >= s_end) {  // the item's sum goes to the resolve kernel
        double* dst = rp.partial + ((size_t)item - rp.partial_item0) * 3;
        dst[0] = pl.f(F_AX, p), dst[1] = pl.f(F_AY, p), dst[2] = pl.f(F_AZ, p);
        has_item = false;
    }
    bool need = valid && !has_item, fresh = false, dead = false;
    unsigned long long need_mask = __ballot(need);
    while (need_mask != 0ull) {
        if (range.next >= range.end) {  // wave-uniform
            uint32_t first = 0xffffffffu;
            if (!range.gone) {
                unsigned long long f64v = 0;
                if (lane == 0) f64v = atomicAdd(lp.next_item, (unsigned long long)lp.reserve);
                const uint32_t flo = __builtin_amdgcn_readfirstlane((uint32_t)f64v);
                const uint32_t fhi = __builtin_amdgcn_readfirstlane((uint32_t)(f64v >> 32));
                first = (fhi != 0u || (uint64_t)flo >= lp.item_count) ? 0xffffffffu : flo;
            }
            if (first == 0xffffffffu) {  // the segment's items are all taken: these slots are done
                range.gone = true;
                if (need) dead = true;
                break;
            }
            range.next = first;
            range.end = (uint64_t)first + lp.reserve < lp.item_count ? first + lp.reserve : (uint32_t)lp.item_count;
        }
        const uint32_t avail = range.end - range.next;
        const uint32_t rank = lanes_below(need_mask);
        if (need && rank < avail) {
            item = (uint32_t)(lp.item_base + range.next + rank);
            uint32_t s_begin;
            item_geometry(rp, item, row, col, s_begin, s_end);
            s_cur = s_begin;
            if (row >= cam.H || col >= cam.W || rp.max_bounces == 0u) {
                // padding pixel of an edge tile (never read) or radiance() with an empty loop: zeros
                double* dst = rp.partial + ((size_t)item - rp.partial_item0) * 3;
                dst[0] = dst[1] = dst[2] = 0.0;
                if (row < cam.H && col < cam.W) n.paths += s_end - s_begin;
            } else {
                has_item = true, fresh = true, need = false;
            }
        }
        const uint32_t wanted = (uint32_t)__popcll(need_mask);
        range.next += wanted < avail ? wanted : avail;
        need_mask = __ballot(need);
    }
    uint32_t ns = LP_DEAD;
    if (valid && dead) pl.u(U_SCUR, p) = 0u;
    if (valid && !dead) {
        Rng rng;
        rng.key = rr_path_key(rp.seed, (uint64_t)row * cam.W + col, (uint64_t)s_cur);
        rng.draw = 0;
        V3 o, d;
        // image origin is upper left, camera origin lower right (main.rs:74-75)
        primary_ray(cam, cam.H - row, cam.W - col, rng, o, d);
        n.paths++;
        s_cur++;
        const V3 inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
        const bool enters = root_box_hit(sc, o, inv);
        ray_o = o, ray_d = d, ray_inv = inv;  // for the query that follows at once (GEN is not short of registers)
        pl.set_v3(F_OX, p, o);
        pl.set_v3(F_DX, p, d);
        pl.set_v3(F_TX, p, mk(1.0, 1.0, 1.0));  // throughput 1, light 0 (lib.rs:522-523)
        pl.u(U_BD, p) = 1u | (enters ? 0u : LP_DIRECT_BIT) | (rng.draw << 16);
        pl.u(U_SCUR, p) = s_cur | SLOT_ITEM_BIT;
        if (fresh) {
            pl.u(U_ITEM, p) = item;
            pl.set_v3(F_AX, p, mk(0.0, 0.0, 0.0));
        }
        ns = enters ? LP_ISECT : LP_BG;
    }
    return ns;
}

// ---- ISECT: Bvh::intersect (bvh.rs:391-415) on a walk tree of at most one record.  What decides whether the
// reference reaches a primitive is one box, its gating box (scene_host.cpp build_walk_trees: the gate tree), after the root
// Node's; the closest hit is the smallest accepted t, the first primitive in depth-first order on exact ties
// (bvh.rs:62).  Gates and primitives are the same for every lane: the loops are wave-uniform and the records
// arrive through the scalar cache.
// FROM_GEN: the ray, 1 / d and the root box's verdict come from lp_gen in registers (a lane is only `valid` here if its
// ray enters the root box).  Behind SHADE_k the ray is read back from the pool: handed over in registers, it costs
// that phase more in spills than the six LDS reads cost here (config 2 +1.5 %).
template <bool COMPACT, bool COUNT, bool FROM_GEN>
RR_DEV uint32_t lp_isect(const Pool& pl, bool valid, uint32_t p, V3 gen_o, V3 gen_d, V3 gen_inv, const SceneDev& sc,
                         const LocalScene& ls, const SurfaceDev* s_surf, const uint4* s_prims, LpCount& n) {
    const V3 o = FROM_GEN ? gen_o : pl.v3(F_OX, p), d = FROM_GEN ? gen_d : pl.v3(F_DX, p);
    const V3 inv = FROM_GEN ? gen_inv : mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    const bool in = FROM_GEN ? valid : (valid && root_box_hit(sc, o, inv));
    const bool nx = inv.x < 0.0, ny = inv.y < 0.0, nz = inv.z < 0.0;
    if (valid) n.rays++;
    if (COUNT && in) n.interior += ls.n_records;
    double best_t = sc.t1;
    uint32_t best_prim = 0xffffffffu, best_tag = 0u;
    for (uint32_t g = 0; g < ls.n_gates; g++) {
        const double* b = ls.box[g];
        double entry;
        const bool pass = in && slab(nx ? b[1] : b[0], nx ? b[0] : b[1], ny ? b[3] : b[2], ny ? b[2] : b[3],
                                     nz ? b[5] : b[4], nz ? b[4] : b[5], o, inv, sc.t0, sc.t1, entry);
        if (__ballot(pass) == 0ull) continue;
        // the records wait in LDS (every lane reads the same address: a broadcast); the next one is requested
        // before this one is worked on -- at two waves per SIMD nothing else would hide the wait
        const uint32_t first = ls.first[g], count = ls.count[g];
        PrimRec<COMPACT> ahead = load_prim_lds<COMPACT>(s_prims, first);
        for (uint32_t k = 0; k < count; k++) {
            const uint32_t pr = first + k;
            const PrimRec<COMPACT> r = ahead;
            ahead = load_prim_lds<COMPACT>(s_prims, pr + 1u);
            if (pass) {
                if (COUNT) {
                    const uint32_t kind = r.tag() & 3u;
                    if (kind == PRIM_TRIANGLE) n.tri++;
                    else if (kind == PRIM_SPHERE) n.sphere++;
                    else n.plane++;
                }
                double t;
                if (prim_intersect<COMPACT>(r, o, d, t) && t > sc.t0 && t < sc.t1) {      // bvh.rs:406
                    if (t < best_t || (t == best_t && pr < best_prim)) {                  // bvh.rs:62
                        best_t = t, best_prim = pr, best_tag = r.tag();
                    }
                }
            }
        }
    }
    uint32_t ns = LP_BG;
    if (best_prim != 0xffffffffu) {  // (never for a lane without a path: `in` is false there)
        pl.set_v3(F_OX, p, v_add(o, v_scale(d, best_t)));  // position, lib.rs:528
        pl.u(U_PRIM, p) = best_prim;
        ns = LP_SHADE0 + (uint32_t)s_surf[best_tag >> 8].kind;
    }
    return ns;
}

// ---- BG: no hit.  radiance() returns light + throughput * background (lib.rs:555); main.rs:69 adds it to the pixel.
RR_DEV uint32_t lp_background(const Pool& pl, bool valid, uint32_t p, const SceneDev& sc, const LocalDev& lp,
                              LpCount& n) {
    const V3 d = pl.v3(F_DX, p), thr = pl.v3(F_TX, p);
    const uint32_t bd = pl.u(U_BD, p), w = pl.u(U_SCUR, p);
    V3 light = mk(0.0, 0.0, 0.0);
    if (valid && ((w >> 30) & 1u)) {
        const double* l = lp_light(lp, p);
        light = mk(l[0], l[1], l[2]);
    }
    const V3 result = v_add(light, v_mul(thr, background(sc, d)));
    if (valid) {
        pl.f(F_AX, p) += result.x, pl.f(F_AY, p) += result.y, pl.f(F_AZ, p) += result.z;
        n.escaped++;
        if (bd & LP_DIRECT_BIT) n.direct++, n.rays++;  // the root box test was this path's only query
    }
    return LP_GEN;
}

// ---- SHADE_k: lib.rs:528-551 for a closest hit on a surface of material kind k (wave-uniform).
// A path's fields wait in LDS; each is read where it is needed and not before -- position and direction for the normal
// and the view vector, the draw index for the RNG, and throughput, light, bounce and cursor only BEHIND
// Material::evaluate, whose arms need the registers (held across it, they cost the kernel 16 spilled registers at
// three workgroups per CU).
template <bool COMPACT>
RR_DEV uint32_t lp_shade(const Pool& pl, bool valid, uint32_t p, int kind, const CameraDev& cam, const RenderDev& rp,
                         const LocalDev& lp, const SurfaceDev* s_surf, const uint4* s_prims, uint32_t& hit_sid) {
    const uint32_t prim = valid ? pl.u(U_PRIM, p) : 0u;
    const PrimRec<COMPACT> rec = load_prim_lds<COMPACT>(s_prims, prim);
    const uint32_t sid = rec.tag() >> 8;
    hit_sid = valid ? (sid < 7u ? sid : 7u) : 8u;
    const SurfaceDev* surf = &s_surf[sid];
    Scatter ev;
    uint32_t draw;
    {
        const V3 normal = prim_normal<COMPACT>(rec, pl.v3(F_OX, p));   // F_OX holds the hit position (lp_isect)
        const V3 view = v_unit(v_scale(pl.v3(F_DX, p), -1.0));
        // the RNG key of the sample in flight: the item's pixel and the sample before the cursor
        uint32_t row, col, s_first, s_end;
        item_geometry(rp, pl.u(U_ITEM, p), row, col, s_first, s_end);
        Rng rng{rr_path_key(rp.seed, (uint64_t)row * cam.W + col, (uint64_t)((pl.u(U_SCUR, p) & SLOT_SAMPLE_MASK) - 1u)),
                pl.u(U_BD, p) >> 16};
        ev = material_evaluate_kind(kind, surf, normal, view, rng);
        draw = rng.draw;
        if (ev.scatter) {  // the roulette's draw (lib.rs:539), taken here so that the key need not outlive the block
            const double u = rng.next();
            draw = rng.draw;
            const uint32_t bd = pl.u(U_BD, p), w = pl.u(U_SCUR, p);
            const uint32_t bounce = bd & LP_BOUNCE_MASK;
            V3 thr = pl.v3(F_TX, p);
            V3 light = mk(0.0, 0.0, 0.0);
            if (valid && ((w >> 30) & 1u)) {
                const double* l = lp_light(lp, p);
                light = mk(l[0], l[1], l[2]);
            }
            light = v_add(light, v_mul(thr, mk(surf->emit[0], surf->emit[1], surf->emit[2])));
            thr = v_mul(thr, ev.color);
            const double pr = rr_max(rr_max(thr.x, thr.y), thr.z);
            if (!valid) return LP_DEAD;
            if (!(u > pr) && bounce < rp.max_bounces) {            // roulette lib.rs:539; loop bound lib.rs:525, :559
                thr = mk(thr.x / pr, thr.y / pr, thr.z / pr);     // DivAssign, vecmath.rs:708-714
                pl.set_v3(F_DX, p, ev.dir);                        // (F_OX keeps the position: the new ray's origin)
                pl.set_v3(F_TX, p, thr);
                pl.u(U_BD, p) = (bounce + 1u) | (draw << 16);
                const bool keep_light = !((rr_f64_bits(light.x) | rr_f64_bits(light.y) | rr_f64_bits(light.z)) == 0ull);
                if (keep_light) {
                    double* l = lp_light(lp, p);
                    l[0] = light.x, l[1] = light.y, l[2] = light.z;
                }
                pl.u(U_SCUR, p) = (w & ~SLOT_LIGHT_BIT) | (keep_light ? SLOT_LIGHT_BIT : 0u);
                return LP_ISECT;
            }
            // radiance() returns `light` (lib.rs:559, or the roulette's return); main.rs:69 adds it to the pixel
            pl.f(F_AX, p) += light.x, pl.f(F_AY, p) += light.y, pl.f(F_AZ, p) += light.z;
            return LP_GEN;
        }
    }
    // NoScatter: radiance() returns `light` without the surface's emission (lib.rs:550)
    if (!valid) return LP_DEAD;
    const uint32_t w = pl.u(U_SCUR, p);
    if ((w >> 30) & 1u) {
        const double* l = lp_light(lp, p);
        pl.f(F_AX, p) += l[0], pl.f(F_AY, p) += l[1], pl.f(F_AZ, p) += l[2];
    }
    return LP_GEN;
}

// the shader clock once everything issued so far has completed (count_work only: clock64() alone is read early,
// and a phase's time lands in the next phase's bucket)
RR_DEV unsigned long long lp_clock() {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}

RR_DEV unsigned long long lp_wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}
RR_DEV void lp_wave_add(unsigned long long* dst, unsigned long long v) {
    const unsigned long long s = lp_wave_sum(v);
    if ((threadIdx.x & 63u) == 0 && s) atomicAdd(dst, s);
}

}  // namespace

template <bool COMPACT, bool COUNT>
__global__ void __launch_bounds__(256, LP_WPS) lp_path_kernel(SceneDev sc, LocalScene ls, CameraDev cam, RenderDev rp,
                                                         LocalDev lp) {
    extern __shared__ __align__(16) unsigned char lp_lds[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    unsigned char* base = lp_lds + wave * LP_WAVE_BYTES;
    Pool pl;
    pl.f64 = reinterpret_cast<double*>(base);
    pl.u32 = reinterpret_cast<uint32_t*>(base + P * LP_NF64 * 8u);
    pl.state = reinterpret_cast<uint8_t*>(pl.u32 + P * LP_NU32);
    pl.list = pl.state + P;
    LocalScene* s_ls = reinterpret_cast<LocalScene*>(lp_lds + 4u * LP_WAVE_BYTES);
    uint4* s_prims = reinterpret_cast<uint4*>(lp_lds + 4u * LP_WAVE_BYTES + sizeof(LocalScene));
    SurfaceDev* s_surf = reinterpret_cast<SurfaceDev*>(lp_lds + 4u * LP_WAVE_BYTES + lp_scene_bytes_dev(ls.n_prims));
    {
        for (uint32_t i = threadIdx.x; i < (uint32_t)(sizeof(LocalScene) / 4); i += 256u)
            reinterpret_cast<uint32_t*>(s_ls)[i] = reinterpret_cast<const uint32_t*>(&ls)[i];
        const uint32_t n_surf = sc.n_surfaces < LP_MAX_PRIMS ? sc.n_surfaces : LP_MAX_PRIMS;
        for (uint32_t i = threadIdx.x; i < n_surf * (uint32_t)(sizeof(SurfaceDev) / 4); i += 256u)
            reinterpret_cast<uint32_t*>(s_surf)[i] = reinterpret_cast<const uint32_t*>(sc.surfaces)[i];
        constexpr uint32_t G = COMPACT ? 3u : 5u;
        const uint4* src = reinterpret_cast<const uint4*>(sc.prims);
        const uint32_t n_prims = ls.n_prims < LP_MAX_PRIMS ? ls.n_prims : LP_MAX_PRIMS;
        for (uint32_t i = threadIdx.x; i < n_prims * G; i += 256u) s_prims[(i / G) * LP_PRIM_GRANULES + i % G] = src[i];
        for (uint32_t q = lane; q < P; q += 64u) {
            pl.state[q] = (uint8_t)LP_GEN;
            pl.u(U_SCUR, q) = 0u;  // no item
        }
        __syncthreads();
    }

    // paths waiting per phase (wave-uniform; statically indexed)
    uint32_t cnt[LP_NSTATE];
#pragma unroll
    for (uint32_t s = 0; s < LP_NSTATE; s++) cnt[s] = 0;
    cnt[LP_GEN] = P;
    LpRange range{0u, 0u, false};
    LpCount n{0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long u_wave = 0, u_lane = 0;
    unsigned long long tk_isect = 0, tk_shade = 0, tk_other = 0, tk_last = COUNT ? lp_clock() : 0ull;
    unsigned long long n_isect = 0, n_shade = 0, t_mid = 0;

    for (;;) {
        // the phase most paths wait for
        uint32_t ph = LP_DEAD, most = 0;
#pragma unroll
        for (uint32_t s = 0; s < LP_NSTATE; s++)
            if (cnt[s] > most) most = cnt[s], ph = s;
        if (most == 0u) break;
        // up to 64 of its paths, lowest first, into the lanes
        const uint32_t st0 = pl.state[lane], st1 = lane + 64u < P ? pl.state[lane + 64u] : LP_DEAD;
        const bool m0 = st0 == ph, m1 = st1 == ph;
        const unsigned long long mask0 = __ballot(m0), mask1 = __ballot(m1);
        const uint32_t n0 = (uint32_t)__popcll(mask0);
        if (m0) pl.list[lanes_below(mask0)] = (uint8_t)lane;
        const uint32_t r1 = n0 + lanes_below(mask1);
        if (m1 && r1 < 64u) pl.list[r1] = (uint8_t)(lane + 64u);
        const uint32_t count = most < 64u ? most : 64u;
        const bool valid = lane < count;
        const uint32_t p = valid ? (uint32_t)pl.list[lane] : 0u;
        if (COUNT) u_wave += 1, u_lane += valid ? 1 : 0;

        uint32_t ns = LP_DEAD;
        // Phase executions chain where every lane goes the same way: a path that escapes (BG) has ended, so its slot
        // takes its next sample (GEN) at once; a path that has just got a ray (GEN, SHADE_k) asks its BVH query in the
        // same execution, on the lanes it is in -- the query is short and its loops are wave-uniform, so the lanes of
        // ended paths idle through little.  What is saved is a phase execution's overhead (state bytes, list, field
        // loads, counters): 2.64 -> 1.22 executions per query on config 2, 24.5 -> 22.0 ms.  LP_ISECT is only what
        // lp_gen / lp_shade answer for "has a ray"; no path waits in it.
        if (ph == LP_BG) (void)lp_background(pl, valid, p, sc, lp, n);
        if (ph == LP_GEN || ph == LP_BG) {
            V3 ro, rd, rinv;
            ro = rd = rinv = mk(0.0, 0.0, 1.0);
            ns = lp_gen(pl, valid, p, sc, cam, rp, lp, range, n, ro, rd, rinv);
            if (COUNT) t_mid = lp_clock();
            const bool q = valid && ns == LP_ISECT;
            if (__ballot(q) != 0ull) {
                const uint32_t nq = lp_isect<COMPACT, COUNT, true>(pl, q, p, ro, rd, rinv, sc, *s_ls, s_surf, s_prims, n);
                if (q) ns = nq;
            }
        } else {
            uint32_t hit_sid = 8u;
            ns = lp_shade<COMPACT>(pl, valid, p, (int)(ph - LP_SHADE0), cam, rp, lp, s_surf, s_prims, hit_sid);
            if (COUNT) {  // what the queries found, per surface row (bench.py: ray shares)
#pragma unroll
                for (uint32_t k = 0; k < 8u; k++) {
                    const uint32_t c = (uint32_t)__popcll(__ballot(hit_sid == k));
                    if (lane == 0 && c) atomicAdd(&rp.counters->surface_hits[k], (unsigned long long)c);
                }
            }
        }
        if (ph >= LP_SHADE0) {
            if (COUNT) t_mid = lp_clock();
            const bool q = valid && ns == LP_ISECT;
            if (__ballot(q) != 0ull) {
                const V3 none = mk(0.0, 0.0, 1.0);
                const uint32_t nq = lp_isect<COMPACT, COUNT, false>(pl, q, p, none, none, none, sc, *s_ls, s_surf, s_prims, n);
                if (q) ns = nq;
            }
        }
        if (valid) pl.state[p] = (uint8_t)ns;
        if (COUNT) {  // shader clock per phase kind: ISECT / SHADE_k / GEN and BG (Counters::*_ticks)
            const unsigned long long now = lp_clock();
            tk_isect += now - t_mid, n_isect++;  // the query at the end of every execution (t_mid: where it began)
            if (ph >= LP_SHADE0) tk_shade += t_mid - tk_last, n_shade++;
            else tk_other += t_mid - tk_last;
            tk_last = now;
        }
#pragma unroll
        for (uint32_t s = 0; s < LP_NSTATE; s++) {
            cnt[s] += (uint32_t)__popcll(__ballot(valid && ns == s));
            if (s == ph) cnt[s] -= count;
        }
    }

    Counters* c = rp.counters;
    lp_wave_add(&c->rays, n.rays);
    lp_wave_add(&c->paths, n.paths);
    lp_wave_add(&c->escaped_paths, n.escaped);
    lp_wave_add(&c->direct_rays, n.direct);
    if (COUNT) {
        lp_wave_add(&c->interior_visits, n.interior);
        lp_wave_add(&c->tri_tests, n.tri);
        lp_wave_add(&c->sphere_tests, n.sphere);
        lp_wave_add(&c->plane_tests, n.plane);
        if (lane == 0) {
            atomicAdd(&c->step_wave, u_wave * 64ull);
            atomicAdd(&c->interior_ticks, tk_isect), atomicAdd(&c->leaf_ticks, tk_shade);
            atomicAdd(&c->refill_ticks, tk_other);
            atomicAdd(&c->inner_wave, n_isect), atomicAdd(&c->leaf_wave, n_shade);  // phase executions (diagnostics)
        }
        lp_wave_add(&c->step_lane, u_lane);
    }
}

uint32_t lp_lds_bytes(uint32_t n_prims, uint32_t n_surfaces) { return lp_block_bytes(n_prims, n_surfaces); }

hipError_t lp_configure() {
    const int most = (int)lp_block_bytes(LP_MAX_PRIMS, LP_MAX_PRIMS);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lp_path_kernel<true, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, most);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lp_path_kernel<true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, most);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lp_path_kernel<false, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, most);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&lp_path_kernel<false, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, most);
}

// workgroups of the kernel a CU holds with this scene's LDS footprint (at most LP_WPS, its launch bound)
hipError_t lp_occupancy(bool compact, uint32_t n_prims, uint32_t n_surfaces, int* blocks_per_cu) {
    const uint32_t lds = lp_block_bytes(n_prims < LP_MAX_PRIMS ? n_prims : LP_MAX_PRIMS, n_surfaces < LP_MAX_PRIMS ? n_surfaces : LP_MAX_PRIMS);
    if (compact) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, lp_path_kernel<true, false>, 256, lds);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, lp_path_kernel<false, false>, 256, lds);
}

hipError_t lp_launch(bool compact, bool count, const SceneDev& sc, const LocalScene& ls, const CameraDev& cam,
                     const RenderDev& rp, const LocalDev& lp, uint32_t blocks, hipStream_t stream) {
    const uint32_t n_surf = sc.n_surfaces < LP_MAX_PRIMS ? sc.n_surfaces : LP_MAX_PRIMS;
    const uint32_t lds = lp_block_bytes(ls.n_prims < LP_MAX_PRIMS ? ls.n_prims : LP_MAX_PRIMS, n_surf);
    if (compact && count)
        hipLaunchKernelGGL((lp_path_kernel<true, true>), dim3(blocks), dim3(256), lds, stream, sc, ls, cam, rp, lp);
    else if (compact)
        hipLaunchKernelGGL((lp_path_kernel<true, false>), dim3(blocks), dim3(256), lds, stream, sc, ls, cam, rp, lp);
    else if (count)
        hipLaunchKernelGGL((lp_path_kernel<false, true>), dim3(blocks), dim3(256), lds, stream, sc, ls, cam, rp, lp);
    else
        hipLaunchKernelGGL((lp_path_kernel<false, false>), dim3(blocks), dim3(256), lds, stream, sc, ls, cam, rp, lp);
    return hipGetLastError();
}

}  // namespace rayrs

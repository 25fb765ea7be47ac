// stream_pool.hip -- the radiance integrator between two deep BVH walks: a path is taken out of its slot in the HBM
// pool when the traversal kernel has answered its query, stays on the CU, in LDS, for as long as its next queries can be
// answered there, and goes back to its slot only when one needs a walk below the root record.
//
// wavefront.hip's hit and miss kernels touch a path's 128-byte slot twice per BVH query (read, write) whatever the query
// costs, and the traversal kernel twice more.  But most queries of the reference's framings never come near the mesh:
// they enter the root Node's box and, of the walk tree's root record, only leaf slots -- on the mesh scenes the group
// that holds the floor rectangle.  Such a query is the root box, four slab tests and that group's primitives, all
// wave-uniform data (local_pool.hip does nothing else for the sphere scenes).  So:
//
//   wf_trav_kernel   unchanged: the READY slots' queries, walked                                  (bvh.rs:391-415)
//   sp_path_kernel   every slot that has an answer (HIT, MISS) or nothing at all (IDLE): IMPORT the path into the
//                    wave's pool in LDS; then, as in local_pool.hip, phases compacted into full waves with __ballot --
//                    SHADE_k (Material::evaluate of kind k, emission, roulette; lib.rs:528-551), BG (Scene::background,
//                    lib.rs:555), GEN (item bookkeeping, next sample, primary ray; main.rs:67-79), ISECT (the next
//                    query's first steps: root box, root record, the leaf groups entered) -- until the next query
//                    enters an interior slot of the root record: EXPORT to the slot, READY for the traversal kernel.
//
// A round is two launches instead of three, and a path crosses the HBM pool once per DEEP query instead of once per
// query.  Results are the streaming kernels': the same device_path.h functions, the reference's rule for every
// primitive tested (closest hit = smallest accepted t, first in depth-first order on ties), the same items in the same
// order per slot.  A query answered here is the walk's own first record: the work counters stay those of the oracle's walk.
#include <hip/hip_runtime.h>

#include "device_path.h"
#include "stream_pool.h"
#include "wavefront.h"

namespace rayrs {

namespace {

constexpr uint32_t P = SP_PATHS_PER_WAVE;
static_assert(P > 64 && P <= 128 && P % 8 == 0, "two state bytes per lane");
constexpr uint32_t SP_WINDOW = 256;  // HBM slots per import window

// a path's fields in LDS: field f of path p at [f * P + p]: 92 bytes.  F_O: the ray's origin while the path waits for
// ISECT, the hit position once a query has found one (lib.rs:528: nothing else of o and t is used afterwards).  What can
// be recomputed or fetched again is not kept: pixel, last sample and RNG key follow from the item number; the normal from
// the primitive's record (U_PRIM: in LDS for the root record's groups, else in HBM, where the import has just read it);
// the item's sum lives in the item-sum array itself (sp_add_sample).
enum { F_OX, F_OY, F_OZ, F_DX, F_DY, F_DZ, F_TX, F_TY, F_TZ, SP_NF64 };
enum { U_PRIM, U_BD, U_ITEM, U_SCUR, U_HSLOT, SP_NU32 };
// U_PRIM: queries answered here since the import (8 bits) | SP_PRIM_LDS (the record is s_prims[index]) | primitive or index
constexpr uint32_t SP_PRIM_LDS = 1u << 23;
constexpr uint32_t SP_PRIM_MASK = (1u << 23) - 1u;
constexpr uint32_t SP_WAVE_BYTES = (P * (SP_NF64 * 8u + SP_NU32 * 4u) + P + 64u + SP_WINDOW * 2u + 15u) & ~15u;  // fields, states, list, import list
static_assert(SP_WAVE_BYTES % 16u == 0, "pools stay 16-byte aligned");
constexpr uint32_t SP_PRIM_GRANULES = 5;
constexpr uint32_t SP_SURFACES_LDS = 16;
constexpr uint32_t SP_SHARED_BYTES = (uint32_t)sizeof(RootRecord) + (SP_ROOT_PRIMS + 4u) * SP_PRIM_GRANULES * 16u;

// path states = the phase a path waits for
constexpr uint32_t SP_FREE = 0, SP_GEN = 1, SP_ISECT = 2, SP_BG = 3, SP_SHADE0 = 4;  // SP_SHADE0 + RAYRS_MAT_*
constexpr uint32_t SP_NSTATE = 13;
constexpr uint32_t SP_IMPORT = 13;  // (a phase, not a state)

// A path goes back to its slot after SP_RESIDENCY queries answered here even if the next one needs no walk: paths that
// never need one (sky and floor) would otherwise fill a wave's pool for good, and the slots it still has to import --
// and the traversal kernel behind this launch -- would wait until the frame's items run out.
constexpr uint32_t SP_RESIDENCY = 32;
constexpr uint32_t SP_BOUNCE_MASK = 0x7fffu;
constexpr uint32_t SP_DIRECT_BIT = 0x8000u;  // a primary ray that missed the root box: a query answered without a walk

struct Pool {
    double* f64;
    uint32_t* u32;
    uint8_t* state;
    uint8_t* list;
    uint16_t* imports;
    RR_DEV double& f(uint32_t field, uint32_t p) const { return f64[field * P + p]; }
    RR_DEV uint32_t& u(uint32_t field, uint32_t p) const { return u32[field * P + p]; }
    RR_DEV V3 v3(uint32_t field, uint32_t p) const { return mk(f(field, p), f(field + 1, p), f(field + 2, p)); }
    RR_DEV void set_v3(uint32_t field, uint32_t p, V3 v) const { f(field, p) = v.x, f(field + 1, p) = v.y, f(field + 2, p) = v.z; }
};

struct SpRange {  // a wave's reserved items [next, end), kept in WfDev::wave_items between launches
    unsigned long long next, end;
};
constexpr unsigned long long SP_ITEMS_GONE = ~0ull;
constexpr uint32_t SP_RESERVE = 256;

struct SpCount {  // per lane
    unsigned long long rays, paths, escaped, direct;
    uint32_t interior, tri, sphere, plane, retired;
};

template <bool COMPACT>
RR_DEV PrimRec<COMPACT> sp_load_prim_lds(const uint4* s_prims, uint32_t p) {
    PrimRec<COMPACT> r;
    const uint4* src = s_prims + p * SP_PRIM_GRANULES;
#pragma unroll
    for (int i = 0; i < (COMPACT ? 3 : 5); i++) r.q[i] = src[i];
    return r;
}

RR_DEV uint32_t sp_udiv_by(uint32_t n, uint32_t d, double inv_d, uint32_t& rem) {  // wavefront.hip udiv_by
    uint32_t q = (uint32_t)((double)n * inv_d);
    int32_t r = (int32_t)(n - q * d);
    if (r < 0) q--, r += (int32_t)d;
    else if ((uint32_t)r >= d) q++, r -= (int32_t)d;
    rem = (uint32_t)r;
    return q;
}

// the item numbering of wavefront.hip item_geometry: 64 pixels of a tile x the tile's chunks
RR_DEV void sp_item_geometry(const RenderDev& rp, uint32_t item, uint32_t& row, uint32_t& col, uint32_t& s_begin,
                             uint32_t& s_end) {
    const uint32_t pit = item & 63u;
    uint32_t chunk, tx;
    const uint32_t tile = sp_udiv_by(item >> 6, rp.nchunks, rp.inv_nchunks, chunk) * rp.tile_ranks + rp.tile_rank;
    row = sp_udiv_by(tile, rp.tiles_x, rp.inv_tiles_x, tx) * 8u + (pit >> 3);
    col = tx * 8u + (pit & 7u);
    s_begin = chunk * rp.chunk;
    s_end = s_begin + rp.chunk < rp.spp ? s_begin + rp.chunk : rp.spp;
}

RR_DEV double* sp_light(const WfDev& wf, uint32_t hslot) { return wf.light + (size_t)hslot * 4u; }

// main.rs:67-69: pixel += radiance, in sample order.  An item's sum is kept where the resolve kernel will read it,
// rp.partial[item]: the item's first sample stores 0 + r (what the reference's zero-initialised sum holds after it),
// later ones add to it.  One slot works on an item, one sample after the other, so the order is the reference's.
// The sum so far is requested (sp_sum_so_far) at the start of the phase that may end the path, and used at its end.
RR_DEV V3 sp_sum_so_far(const RenderDev& rp, uint32_t item) {
    const double* src = rp.partial + (size_t)item * 3;
    return mk(src[0], src[1], src[2]);
}
RR_DEV void sp_add_sample(const RenderDev& rp, uint32_t item, bool first_sample, V3 so_far, V3 r) {
    double* dst = rp.partial + (size_t)item * 3;
    if (first_sample) so_far = mk(0.0, 0.0, 0.0);  // (what was read is another item's leftover, or nothing)
    dst[0] = so_far.x + r.x, dst[1] = so_far.y + r.y, dst[2] = so_far.z + r.z;
}

// EXPORT (inline, by the phase that finds a path in need of a walk): the path back to its slot, READY for the
// traversal kernel; its place in the pool is free.
RR_DEV void sp_export(const Pool& pl, uint32_t p, const WfDev& wf) {
    const uint32_t hslot = pl.u(U_HSLOT, p);
    RaySlot* rs = &wf.slots[hslot].ray;
    TailSlot* ts = &wf.slots[hslot].tail;
    const V3 o = pl.v3(F_OX, p), d = pl.v3(F_DX, p), thr = pl.v3(F_TX, p);
    rs->o[0] = o.x, rs->o[1] = o.y, rs->o[2] = o.z;
    rs->d[0] = d.x, rs->d[1] = d.y, rs->d[2] = d.z;
    rs->bd = pl.u(U_BD, p) & ~SP_DIRECT_BIT;
    ts->thr[0] = thr.x, ts->thr[1] = thr.y, ts->thr[2] = thr.z;
    ts->item = pl.u(U_ITEM, p);
    ts->s_cur = pl.u(U_SCUR, p);
    wf.state[hslot] = WF_READY;
}

// ---- IMPORT: the path of an HBM slot the traversal kernel has answered (or that never had one) into a free place of the pool
template <bool COMPACT>
RR_DEV uint32_t sp_import(const Pool& pl, bool valid, uint32_t p, uint32_t hslot, uint32_t hstate, const SceneDev& sc,
                          const WfDev& wf, const SurfaceDev* s_surf, uint32_t n_surf_lds) {
    // lanes without work read slot 0: harmless
    const RaySlot* rs = &wf.slots[hslot].ray;
    const TailSlot* ts = &wf.slots[hslot].tail;
    const V3 o = mk(rs->o[0], rs->o[1], rs->o[2]), d = mk(rs->d[0], rs->d[1], rs->d[2]);
    const double t = rs->t;
    const uint32_t prim = rs->prim, bd = rs->bd;
    const V3 thr = mk(ts->thr[0], ts->thr[1], ts->thr[2]);
    const uint32_t item = ts->item, scur = ts->s_cur;
    uint32_t ns = SP_GEN;
    if (hstate == WF_HIT) {  // the surface's material kind decides the phase the path waits for
        const uint32_t tag = reinterpret_cast<const uint32_t*>(sc.prims)[(size_t)(valid ? prim : 0u) * (COMPACT ? 12u : 20u) +
                                                                          (COMPACT ? 11u : 19u)];
        const uint32_t sid = tag >> 8;
        if (valid) {
            pl.set_v3(F_OX, p, v_add(o, v_scale(d, t)));  // position, lib.rs:528
            pl.u(U_PRIM, p) = prim;  // (no query answered here yet)
        }
        const SurfaceDev* surf = sid < n_surf_lds ? &s_surf[sid] : sc.surfaces + sid;
        ns = SP_SHADE0 + (uint32_t)surf->kind;
    } else {
        if (valid) pl.u(U_PRIM, p) = 0u;
        if (hstate == WF_MISS) ns = SP_BG;
    }
    if (valid) {
        pl.set_v3(F_DX, p, d);
        pl.set_v3(F_TX, p, thr);
        pl.u(U_BD, p) = bd & ~SP_DIRECT_BIT;  // (bounce counts stay below 2^15)
        pl.u(U_ITEM, p) = item;
        pl.u(U_SCUR, p) = hstate == WF_IDLE ? 0u : scur;
        pl.u(U_HSLOT, p) = hslot;
    }
    return ns;
}

// ---- GEN: local_pool.hip lp_gen on a slot of the HBM pool: a slot that finds no further item is DEAD.
// `winding`: the wave has imported all of its slots; new rays go straight back to their slots.
RR_DEV uint32_t sp_gen(const Pool& pl, bool valid, uint32_t p, bool winding, const SceneDev& sc, const CameraDev& cam,
                       const RenderDev& rp, const WfDev& wf, SpRange& range, SpCount& n) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lanemask_lt = (1ull << lane) - 1ull;
    const uint32_t w = pl.u(U_SCUR, p);
    bool has_item = valid && (w >> 31) != 0u;
    uint32_t s_cur = w & SLOT_SAMPLE_MASK, item = pl.u(U_ITEM, p);
    uint32_t row, col, s_end, s_first;
    sp_item_geometry(rp, item, row, col, s_first, s_end);  // (of no meaning without an item)
    if (has_item && s_cur >= s_end) has_item = false;  // its sum is where the resolve kernel reads it (sp_add_sample)
    bool need = valid && !has_item, fresh = false, dead = false;
    unsigned long long need_mask = __ballot(need);
    while (need_mask != 0ull) {
        if (range.next >= range.end) {  // wave-uniform
            unsigned long long first = SP_ITEMS_GONE;
            if (range.end != SP_ITEMS_GONE) {
                if (lane == 0) first = atomicAdd(rp.next_item, (unsigned long long)SP_RESERVE);
                const uint32_t flo = __builtin_amdgcn_readfirstlane((uint32_t)first);
                const uint32_t fhi = __builtin_amdgcn_readfirstlane((uint32_t)(first >> 32));
                first = ((unsigned long long)fhi << 32) | flo;
            }
            if (first >= rp.total_items) {  // the counter has run out: these slots are done
                range.next = range.end = SP_ITEMS_GONE;
                if (need) dead = true;
                break;
            }
            range.next = first;
            range.end = first + SP_RESERVE < rp.total_items ? first + SP_RESERVE : rp.total_items;
        }
        const uint32_t avail = (uint32_t)(range.end - range.next);
        const uint32_t rank = (uint32_t)__popcll(need_mask & lanemask_lt);
        if (need && rank < avail) {
            item = (uint32_t)(range.next + rank);
            uint32_t s_begin;
            sp_item_geometry(rp, item, row, col, s_begin, s_end);
            s_cur = s_begin;
            if (row >= cam.H || col >= cam.W || rp.max_bounces == 0u) {
                // padding pixel of an edge tile (never read) or radiance() with an empty loop: zeros
                double* dst = rp.partial + (size_t)item * 3;
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
    uint32_t ns = SP_FREE;
    if (valid && dead) {  // no further item: the slot is out of work
        const uint32_t hslot = pl.u(U_HSLOT, p);
        wf.slots[hslot].tail.s_cur = 0u;
        wf.state[hslot] = WF_DEAD;
        n.retired++;
    }
    if (valid && !dead) {
        Rng rng;
        rng.key = rr_path_key(rp.seed, (uint64_t)row * cam.W + col, (uint64_t)s_cur);
        rng.draw = 0;
        V3 o, d;
        // image origin is upper left, camera origin lower right (main.rs:74-75)
        primary_ray(cam, cam.H - row, cam.W - col, rng, o, d);
        n.paths++;
        s_cur++;
        // winding: the root box is the traversal kernel's first step anyway (trav_init)
        const bool enters = winding || root_box_hit(sc, o, mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z));
        pl.set_v3(F_OX, p, o);
        pl.set_v3(F_DX, p, d);
        pl.set_v3(F_TX, p, mk(1.0, 1.0, 1.0));  // throughput 1, light 0 (lib.rs:522-523)
        pl.u(U_BD, p) = 1u | (enters ? 0u : SP_DIRECT_BIT) | (rng.draw << 16);
        pl.u(U_SCUR, p) = s_cur | SLOT_ITEM_BIT;
        if (fresh) pl.u(U_ITEM, p) = item;
        pl.u(U_PRIM, p) = 0u;
        ns = enters ? SP_ISECT : SP_BG;
        if (winding) {
            sp_export(pl, p, wf);
            ns = SP_FREE;
        }
    }
    return ns;
}

// ---- ISECT: the next query's first steps (trav_init and the traversal kernel's visit of the root record): the root
// Node's box (bvh.rs:394), the root record's four boxes -- nothing is culled there: the closest hit is still t1 --, and,
// when no interior slot is entered, the primitives of the leaf groups entered, by the reference's rule (bvh.rs:62,
// :406).  A ray that enters an interior slot is the traversal kernel's (which starts again at the root box).
template <bool COMPACT, bool COUNT>
RR_DEV uint32_t sp_isect(const Pool& pl, bool valid, uint32_t p, const SceneDev& sc, const RootRecord& root,
                         const WfDev& wf, const SurfaceDev* s_surf, uint32_t n_surf_lds, const uint4* s_prims, SpCount& n) {
    const V3 o = pl.v3(F_OX, p), d = pl.v3(F_DX, p);
    const V3 inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    const bool in = valid && root_box_hit(sc, o, inv);
    const bool nx = inv.x < 0.0, ny = inv.y < 0.0, nz = inv.z < 0.0;
    bool pass[4];
    bool walk = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double* b = root.box[k];
        double entry;
        pass[k] = in && root.kind[k] != REF_NONE &&
                  slab(nx ? b[1] : b[0], nx ? b[0] : b[1], ny ? b[3] : b[2], ny ? b[2] : b[3], nz ? b[5] : b[4],
                       nz ? b[4] : b[5], o, inv, sc.t0, sc.t1, entry);
        walk = walk || (pass[k] && root.kind[k] == REF_INTERIOR);
    }
    // (the residency budget: the query goes to the traversal kernel whatever it needs)
    const uint32_t stay = valid ? pl.u(U_PRIM, p) >> 24 : 0u;
    if (stay >= SP_RESIDENCY) walk = true;
    const bool here = valid && !walk;  // answered here (a Miss when the root box is missed)
    if (here) n.rays++;
    if (COUNT && in && !walk) n.interior++;
    double best_t = sc.t1;
    uint32_t best_prim = 0xffffffffu, best_lds = 0u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (root.kind[k] != REF_RANGE) continue;  // wave-uniform
        const bool test = pass[k] && !walk;
        if (__ballot(test) == 0ull) continue;
        const uint32_t first = root.first[k], count = root.count[k], lfirst = root.lds_first[k];
        PrimRec<COMPACT> ahead = sp_load_prim_lds<COMPACT>(s_prims, lfirst);
        for (uint32_t j = 0; j < count; j++) {
            const PrimRec<COMPACT> r = ahead;
            ahead = sp_load_prim_lds<COMPACT>(s_prims, lfirst + j + 1u);
            if (test) {
                if (COUNT) {
                    const uint32_t kind = r.tag() & 3u;
                    if (kind == PRIM_TRIANGLE) n.tri++;
                    else if (kind == PRIM_SPHERE) n.sphere++;
                    else n.plane++;
                }
                double t;
                const uint32_t pr = first + j;
                if (prim_intersect<COMPACT>(r, o, d, t) && t > sc.t0 && t < sc.t1) {      // bvh.rs:406
                    if (t < best_t || (t == best_t && pr < best_prim)) best_t = t, best_prim = pr, best_lds = lfirst + j;  // bvh.rs:62
                }
            }
        }
    }
    uint32_t ns = SP_BG;
    if (valid && walk) {
        sp_export(pl, p, wf);
        ns = SP_FREE;
    }
    if (here) pl.u(U_PRIM, p) = (stay + 1u) << 24;
    if (here && best_prim != 0xffffffffu) {
        const uint32_t tag = s_prims[best_lds * SP_PRIM_GRANULES + (COMPACT ? 2u : 4u)].w;
        const uint32_t sid = tag >> 8;
        pl.set_v3(F_OX, p, v_add(o, v_scale(d, best_t)));  // position, lib.rs:528
        pl.u(U_PRIM, p) = best_lds | SP_PRIM_LDS | ((stay + 1u) << 24);
        const SurfaceDev* surf = sid < n_surf_lds ? &s_surf[sid] : sc.surfaces + sid;
        ns = SP_SHADE0 + (uint32_t)surf->kind;
    }
    return ns;
}

// ---- BG: no hit.  radiance() returns light + throughput * background (lib.rs:555); main.rs:69 adds it to the pixel.
RR_DEV uint32_t sp_background(const Pool& pl, bool valid, uint32_t p, const SceneDev& sc, const RenderDev& rp,
                              const WfDev& wf, SpCount& n) {
    const V3 d = pl.v3(F_DX, p), thr = pl.v3(F_TX, p);
    const uint32_t bd = pl.u(U_BD, p), w = pl.u(U_SCUR, p);
    const uint32_t item = pl.u(U_ITEM, p);
    const V3 so_far = sp_sum_so_far(rp, valid ? item : 0u);
    const bool first = (bd & SP_BOUNCE_MASK) <= 1u;  // throughput 1 and light 0 are implied (lib.rs:522-523)
    V3 light = mk(0.0, 0.0, 0.0);
    if (valid && !first && ((w >> 30) & 1u)) {
        const double* l = sp_light(wf, pl.u(U_HSLOT, p));
        light = mk(l[0], l[1], l[2]);
    }
    const V3 result = v_add(light, v_mul(first ? mk(1.0, 1.0, 1.0) : thr, background(sc, d)));
    if (valid) {
        uint32_t row, col, s_first, s_end;
        sp_item_geometry(rp, item, row, col, s_first, s_end);
        sp_add_sample(rp, item, (w & SLOT_SAMPLE_MASK) - 1u == s_first, so_far, result);
        n.escaped++;
        if (bd & SP_DIRECT_BIT) n.direct++, n.rays++;  // the root box test was this path's only query
    }
    return SP_GEN;
}

// ---- SHADE_k: lib.rs:528-551 for a closest hit on a surface of material kind k (wave-uniform).
template <bool COMPACT>
RR_DEV uint32_t sp_shade(const Pool& pl, bool valid, uint32_t p, int kind, const SceneDev& sc, const CameraDev& cam,
                         const RenderDev& rp, const WfDev& wf, const SurfaceDev* s_surf, uint32_t n_surf_lds,
                         const uint4* s_prims, uint32_t& hit_sid) {
    const V3 position = pl.v3(F_OX, p), d = pl.v3(F_DX, p);  // (the hit position stays where it is: the next origin)
    // the primitive's record once more, for the normal (lib.rs:529) and the surface row: from LDS when the query was
    // answered here, else from HBM, where the import read its tag a moment ago
    const uint32_t pw = valid ? pl.u(U_PRIM, p) : SP_PRIM_LDS;
    PrimRec<COMPACT> rec;
    if (pw & SP_PRIM_LDS) rec = sp_load_prim_lds<COMPACT>(s_prims, pw & 0xffu);
    else rec = load_prim<COMPACT>(sc.prims, pw & SP_PRIM_MASK);
    const V3 normal = prim_normal<COMPACT>(rec, position);
    const uint32_t bd = pl.u(U_BD, p), w = pl.u(U_SCUR, p), hslot = pl.u(U_HSLOT, p);
    const uint32_t bounce = bd & SP_BOUNCE_MASK;
    // throughput and light of a path's first query are 1 and 0 (lib.rs:522-523): slots of the streaming kernels' format
    // do not carry them then
    V3 thr = bounce > 1u ? pl.v3(F_TX, p) : mk(1.0, 1.0, 1.0);
    uint32_t row, col, s_first, s_end;
    const uint32_t item = pl.u(U_ITEM, p);
    const V3 so_far = sp_sum_so_far(rp, valid ? item : 0u);
    sp_item_geometry(rp, item, row, col, s_first, s_end);
    Rng rng{rr_path_key(rp.seed, (uint64_t)row * cam.W + col, (uint64_t)((w & SLOT_SAMPLE_MASK) - 1u)), bd >> 16};
    V3 light = mk(0.0, 0.0, 0.0);
    if (valid && bounce > 1u && ((w >> 30) & 1u)) {
        const double* l = sp_light(wf, hslot);
        light = mk(l[0], l[1], l[2]);
    }
    const V3 view = v_unit(v_scale(d, -1.0));
    const uint32_t sid = rec.tag() >> 8;
    hit_sid = valid ? (sid < 7u ? sid : 7u) : 8u;
    const SurfaceDev* surf = sid < n_surf_lds ? &s_surf[sid] : sc.surfaces + sid;
    const Scatter ev = material_evaluate_kind(kind, surf, normal, view, rng);
    bool goes_on = false;
    if (ev.scatter) {
        light = v_add(light, v_mul(thr, mk(surf->emit[0], surf->emit[1], surf->emit[2])));
        thr = v_mul(thr, ev.color);
        const double pr = rr_max(rr_max(thr.x, thr.y), thr.z);
        if (!(rng.next() > pr) && bounce < rp.max_bounces) {  // roulette lib.rs:539; loop bound lib.rs:525, :559
            thr = mk(thr.x / pr, thr.y / pr, thr.z / pr);     // DivAssign, vecmath.rs:708-714
            goes_on = true;
        }
    }
    if (!valid) return SP_FREE;
    if (goes_on) {
        pl.set_v3(F_DX, p, ev.dir);
        pl.set_v3(F_TX, p, thr);
        pl.u(U_BD, p) = (bounce + 1u) | (rng.draw << 16);
        const bool keep_light = !((rr_f64_bits(light.x) | rr_f64_bits(light.y) | rr_f64_bits(light.z)) == 0ull);
        if (keep_light) {
            double* l = sp_light(wf, hslot);
            l[0] = light.x, l[1] = light.y, l[2] = light.z;
        }
        pl.u(U_SCUR, p) = (w & ~SLOT_LIGHT_BIT) | (keep_light ? SLOT_LIGHT_BIT : 0u);
        return SP_ISECT;
    }
    // radiance() returns `light` (lib.rs:550, :559, or the roulette's return); main.rs:69 adds it to the pixel
    sp_add_sample(rp, item, (w & SLOT_SAMPLE_MASK) - 1u == s_first, so_far, light);
    return SP_GEN;
}

// the shader clock once everything issued so far has completed (count_work only)
RR_DEV unsigned long long sp_clock() {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}

RR_DEV unsigned long long sp_wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}
RR_DEV void sp_wave_add(unsigned long long* dst, unsigned long long v) {
    const unsigned long long s = sp_wave_sum(v);
    if ((threadIdx.x & 63u) == 0 && s) atomicAdd(dst, s);
}

}  // namespace

template <bool COMPACT, bool COUNT>
__global__ void __launch_bounds__(256, SP_WPS) sp_path_kernel(SceneDev sc, RootRecord root_arg, CameraDev cam, RenderDev rp,
                                                         WfDev wf) {
    extern __shared__ __align__(16) unsigned char sp_lds[];
    if (wf.ctl->live_slots == 0u) return;  // a round enqueued behind the frame's last one
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const unsigned long long lanemask_lt = (1ull << lane) - 1ull;
    unsigned char* base = sp_lds + wave * SP_WAVE_BYTES;
    Pool pl;
    pl.f64 = reinterpret_cast<double*>(base);
    pl.u32 = reinterpret_cast<uint32_t*>(base + P * SP_NF64 * 8u);
    pl.state = reinterpret_cast<uint8_t*>(pl.u32 + P * SP_NU32);
    pl.list = pl.state + P;
    pl.imports = reinterpret_cast<uint16_t*>(pl.list + 64u);
    RootRecord* s_root = reinterpret_cast<RootRecord*>(sp_lds + 4u * SP_WAVE_BYTES);
    uint4* s_prims = reinterpret_cast<uint4*>(sp_lds + 4u * SP_WAVE_BYTES + sizeof(RootRecord));
    SurfaceDev* s_surf = reinterpret_cast<SurfaceDev*>(sp_lds + 4u * SP_WAVE_BYTES + SP_SHARED_BYTES);
    const uint32_t n_surf_lds = sc.n_surfaces < SP_SURFACES_LDS ? sc.n_surfaces : SP_SURFACES_LDS;
    {
        for (uint32_t i = threadIdx.x; i < (uint32_t)(sizeof(RootRecord) / 4); i += 256u)
            reinterpret_cast<uint32_t*>(s_root)[i] = reinterpret_cast<const uint32_t*>(&root_arg)[i];
        for (uint32_t i = threadIdx.x; i < n_surf_lds * (uint32_t)(sizeof(SurfaceDev) / 4); i += 256u)
            reinterpret_cast<uint32_t*>(s_surf)[i] = reinterpret_cast<const uint32_t*>(sc.surfaces)[i];
        // the primitives of the root record's leaf groups, in the order of RootRecord::lds_first
        constexpr uint32_t G = COMPACT ? 3u : 5u;
        const uint4* src = reinterpret_cast<const uint4*>(sc.prims);
        for (uint32_t k = 0; k < 4u; k++) {
            if (root_arg.kind[k] != REF_RANGE) continue;
            for (uint32_t i = threadIdx.x; i < root_arg.count[k] * G; i += 256u)
                s_prims[(root_arg.lds_first[k] + i / G) * SP_PRIM_GRANULES + i % G] = src[(size_t)root_arg.first[k] * G + i];
        }
        for (uint32_t q = lane; q < P; q += 64u) pl.state[q] = (uint8_t)SP_FREE;
        if (blockIdx.x == 0 && threadIdx.x == 0) wf.ctl->next_window = 0;  // the traversal kernel's window cursor
        __syncthreads();
    }
    const RootRecord& root = *s_root;

    const uint32_t wave_global = blockIdx.x * 4u + wave;
    const uint32_t n_windows = wf.np / SP_WINDOW;
    uint32_t cnt[SP_NSTATE];
#pragma unroll
    for (uint32_t s = 0; s < SP_NSTATE; s++) cnt[s] = 0;
    cnt[SP_FREE] = P;
    uint32_t imp_pos = 0, imp_len = 0, imp_base = 0;  // the wave's import list (wave-uniform)
    bool winding = false;  // the wave's windows have all been imported: resident paths go back to their slots at their next ray
    SpRange range;
    range.next = wf.wave_items[2 * (size_t)wave_global];
    range.end = wf.wave_items[2 * (size_t)wave_global + 1];
    SpCount n{0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long u_wave = 0, u_lane = 0;
    unsigned long long tk[5] = {0, 0, 0, 0, 0}, nk[5] = {0, 0, 0, 0, 0}, tk_last = COUNT ? sp_clock() : 0ull;  // import gen isect bg shade

    // Windows of the pool are dealt round robin: wave g takes g, g + n_waves, ...  A wave's windows are spread over the
    // whole pool, so every wave sees the frame's mix of cheap and deep regions; and because a slot is always taken by the
    // same wave, the item range a wave has reserved is used up by its own slots (no item is ever left behind: a slot
    // dies only when its wave's range is empty and the counter has run out, and from then on the range stays empty).
    const uint32_t n_waves_total = gridDim.x * 4u;
    uint32_t next_win = wave_global;
    for (;;) {
        // slots to import: the rest of the current window's list, else the wave's next window that has any
        while (!winding && imp_pos >= imp_len) {
            const uint32_t w = next_win;
            if (w >= n_windows) {
                winding = true;  // everything imported: resident paths go back to their slots at their next ray
                break;
            }
            next_win += n_waves_total;
            // SP_WINDOW = 256 slots: four state bytes per lane
            const uint32_t word = reinterpret_cast<const uint32_t*>(wf.state + (size_t)w * SP_WINDOW)[lane];
            uint32_t count = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t s = (word >> (j * 8)) & 0xffu;
                const bool m = s == WF_HIT || s == WF_MISS || s == WF_IDLE;
                const unsigned long long mask = __ballot(m);
                if (m) pl.imports[count + (uint32_t)__popcll(mask & lanemask_lt)] = (uint16_t)((lane * 4u + (uint32_t)j) | (s << 8));
                count += (uint32_t)__popcll(mask);
            }
            imp_base = w * SP_WINDOW, imp_len = count, imp_pos = 0;
        }
        // the phase most paths wait for; importing counts as many as there are free places and slots to take.  With no
        // more windows anywhere, what is left of this wave's list goes first: its paths are the last the launch waits for.
        const uint32_t imp_avail = imp_len - imp_pos;
        const uint32_t n_import = cnt[SP_FREE] < imp_avail ? cnt[SP_FREE] : imp_avail;
        uint32_t ph = SP_NSTATE + 1u, most = 0;
#pragma unroll
        for (uint32_t s = 1; s < SP_NSTATE; s++)
            if (cnt[s] > most) most = cnt[s], ph = s;
        // (a pool with empty places has fewer paths to fill its phases with: import as soon as 16 places are free)
        if (n_import > most || (n_import >= 16u) || (winding && n_import > 0u)) most = n_import, ph = SP_IMPORT;
        if (most == 0u) break;  // (then winding: nothing resident, nothing left to take)
        // up to 64 of its paths (importing: free places), lowest first, into the lanes
        const uint32_t want = ph == SP_IMPORT ? SP_FREE : ph;
        const uint32_t st0 = pl.state[lane], st1 = lane + 64u < P ? pl.state[lane + 64u] : 0xffu;
        const bool m0 = st0 == want, m1 = st1 == want;
        const unsigned long long mask0 = __ballot(m0), mask1 = __ballot(m1);
        const uint32_t n0 = (uint32_t)__popcll(mask0);
        if (m0) pl.list[__popcll(mask0 & lanemask_lt)] = (uint8_t)lane;
        const uint32_t r1 = n0 + (uint32_t)__popcll(mask1 & lanemask_lt);
        if (m1 && r1 < 64u) pl.list[r1] = (uint8_t)(lane + 64u);
        const uint32_t count = most < 64u ? most : 64u;
        const bool valid = lane < count;
        const uint32_t p = valid ? (uint32_t)pl.list[lane] : 0u;
        if (COUNT) u_wave += 1, u_lane += valid ? 1 : 0;

        uint32_t ns = SP_FREE;
        if (ph == SP_IMPORT) {
            const uint32_t e = valid ? (uint32_t)pl.imports[imp_pos + lane] : 0u;
            ns = sp_import<COMPACT>(pl, valid, p, imp_base + (e & 0xffu), valid ? e >> 8 : (uint32_t)WF_IDLE, sc, wf, s_surf,
                                    n_surf_lds);
            imp_pos += count;
        } else if (ph == SP_GEN) {
            ns = sp_gen(pl, valid, p, winding, sc, cam, rp, wf, range, n);
        } else if (ph == SP_ISECT) {
            if (winding) {  // back to the slot as it is: the traversal kernel starts at the root box
                if (valid) sp_export(pl, p, wf);
                ns = SP_FREE;
            } else {
                ns = sp_isect<COMPACT, COUNT>(pl, valid, p, sc, root, wf, s_surf, n_surf_lds, s_prims, n);
            }
        } else if (ph == SP_BG) {
            ns = sp_background(pl, valid, p, sc, rp, wf, n);
        } else {
            uint32_t hit_sid = 8u;
            ns = sp_shade<COMPACT>(pl, valid, p, (int)(ph - SP_SHADE0), sc, cam, rp, wf, s_surf, n_surf_lds, s_prims, hit_sid);
            if (COUNT) {  // what the queries found, per surface row (bench.py: ray shares)
#pragma unroll
                for (uint32_t k = 0; k < 8u; k++) {
                    const uint32_t c = (uint32_t)__popcll(__ballot(hit_sid == k));
                    if (lane == 0 && c) atomicAdd(&rp.counters->surface_hits[k], (unsigned long long)c);
                }
            }
        }
        if (valid) pl.state[p] = (uint8_t)ns;
        if (COUNT) {
            const unsigned long long now = sp_clock();
            const int k = ph == SP_IMPORT ? 0 : ph == SP_GEN ? 1 : ph == SP_ISECT ? 2 : ph == SP_BG ? 3 : 4;
#pragma unroll
            for (int j = 0; j < 5; j++)
                if (j == k) tk[j] += now - tk_last, nk[j] += 1;
            tk_last = now;
        }
#pragma unroll
        for (uint32_t s = 0; s < SP_NSTATE; s++) {
            cnt[s] += (uint32_t)__popcll(__ballot(valid && ns == s));
            if (s == want) cnt[s] -= count;
        }
    }

    if (lane == 0) {
        wf.wave_items[2 * (size_t)wave_global] = range.next;
        wf.wave_items[2 * (size_t)wave_global + 1] = range.end;
    }
    Counters* c = rp.counters;
    sp_wave_add(&c->rays, n.rays);
    sp_wave_add(&c->paths, n.paths);
    sp_wave_add(&c->escaped_paths, n.escaped);
    sp_wave_add(&c->direct_rays, n.direct);
    {
        const uint32_t r = (uint32_t)sp_wave_sum(n.retired);
        if (lane == 0 && r) atomicSub(&wf.ctl->live_slots, r);
    }
    if (COUNT) {
        sp_wave_add(&c->interior_visits, n.interior);
        sp_wave_add(&c->tri_tests, n.tri);
        sp_wave_add(&c->sphere_tests, n.sphere);
        sp_wave_add(&c->plane_tests, n.plane);
        // (the traversal kernel owns step_wave / step_lane; this kernel's lane utilisation goes to the shade_* pair)
        if (lane == 0) {
            atomicAdd(&c->shade_wave, u_wave * 64ull);
#pragma unroll
            for (int j = 0; j < 5; j++) atomicAdd(&c->sp_ticks[j], tk[j]), atomicAdd(&c->sp_phases[j], nk[j]);
        }
        sp_wave_add(&c->shade_lane, u_lane);
    }
}

uint32_t sp_lds_bytes(uint32_t n_surfaces) {
    const uint32_t ns = n_surfaces < SP_SURFACES_LDS ? n_surfaces : SP_SURFACES_LDS;
    return 4u * SP_WAVE_BYTES + SP_SHARED_BYTES + ns * (uint32_t)sizeof(SurfaceDev);
}

hipError_t sp_configure() {
    const int most = (int)sp_lds_bytes(SP_SURFACES_LDS);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sp_path_kernel<true, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, most);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sp_path_kernel<true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, most);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sp_path_kernel<false, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, most);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&sp_path_kernel<false, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, most);
}

hipError_t sp_launch(bool compact, bool count, const SceneDev& sc, const RootRecord& root, const CameraDev& cam,
                     const RenderDev& rp, const WfDev& wf, uint32_t blocks, hipStream_t stream) {
    const uint32_t lds = sp_lds_bytes(sc.n_surfaces);
    if (compact && count)
        hipLaunchKernelGGL((sp_path_kernel<true, true>), dim3(blocks), dim3(256), lds, stream, sc, root, cam, rp, wf);
    else if (compact)
        hipLaunchKernelGGL((sp_path_kernel<true, false>), dim3(blocks), dim3(256), lds, stream, sc, root, cam, rp, wf);
    else if (count)
        hipLaunchKernelGGL((sp_path_kernel<false, true>), dim3(blocks), dim3(256), lds, stream, sc, root, cam, rp, wf);
    else
        hipLaunchKernelGGL((sp_path_kernel<false, false>), dim3(blocks), dim3(256), lds, stream, sc, root, cam, rp, wf);
    return hipGetLastError();
}

}  // namespace rayrs

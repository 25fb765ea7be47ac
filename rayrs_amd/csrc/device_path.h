// device_path.h -- gfx950 device functions of the radiance integrator.
//
// One lane traces one path.  All arithmetic is f64 in the reference's operator
// order (no contraction: the translation unit is built with -ffp-contract=off);
// elementary functions and the RNG come from include/rayrs_numeric.h so that the
// CPU checker, which compiles the same header, sees the same bits.
//
// Reference items restated here (paths relative to rayrs-lib/src):
//   vecmath.rs:341-352, :513-806   vector helpers
//   geometry.rs:106-136, :229-282, :359-379, :458-513   shapes, AABB slab test
//   bvh.rs:40-73, :391-415         closest hit (first leaf in DFS order wins ties)
//   material.rs:91-109, :259-593, :721-812, :903-1046, :1132-1161, :1189-1231,
//               :1233-1518         materials
//   lib.rs:202-210, :254-285, :521-560   primary ray, background, radiance
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/rayrs_hip.h"
#include "../../include/rayrs_numeric.h"
#include "layout.h"

namespace rayrs {

#define RR_DEV __device__ __forceinline__

struct V3 {
    double x, y, z;
};

// Number of set bits of a wave mask below this lane: the rank of a lane in a ballot (v_mbcnt: no per-lane
// "lanes below me" mask has to live in two registers for it).
RR_DEV uint32_t lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

RR_DEV V3 mk(double x, double y, double z) { return V3{x, y, z}; }
RR_DEV V3 v_add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RR_DEV V3 v_sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RR_DEV V3 v_mul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RR_DEV V3 v_scale(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
RR_DEV V3 v_div(V3 a, double s) {  // Div<f64>: multiply by the reciprocal, vecmath.rs:690-698
    const double inv = 1.0 / s;
    return v_scale(a, inv);
}
RR_DEV double v_dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RR_DEV V3 v_cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RR_DEV double v_mag2(V3 a) { return v_dot(a, a); }
RR_DEV V3 v_unit(V3 a) { return v_div(a, rr_sqrt(v_mag2(a))); }
RR_DEV bool v_is_zeros(V3 a) { return a.x == 0.0 && a.y == 0.0 && a.z == 0.0; }

RR_DEV void v_onb(V3 n, V3& e1, V3& e2) {  // vecmath.rs:341-352
    if (rr_fabs(n.x) > rr_fabs(n.y))
        e1 = v_unit(mk(n.z, 0.0, -n.x));
    else
        e1 = v_unit(mk(0.0, n.z, -n.y));
    e2 = v_unit(v_cross(n, e1));
}

// powi(4) / powi(5) as LLVM expands them for a constant exponent
RR_DEV double pow4(double x) {
    const double x2 = x * x;
    return x2 * x2;
}
RR_DEV double pow5(double x) {
    const double x2 = x * x;
    return x * (x2 * x2);
}

struct Rng {
    uint64_t key;
    uint32_t draw;
    RR_DEV double next() { return rr_uniform(key, draw++); }
};

// ------------------------------------------------------------------ records

RR_DEV double f64_from(uint32_t lo, uint32_t hi) { return rr_bits_f64(((uint64_t)hi << 32) | lo); }

template <bool COMPACT>
struct PrimRec {
    // enough dwords for either layout; only the first 12 (compact) or 20 (full) are loaded
    uint4 q[COMPACT ? 3 : 5];
    RR_DEV uint32_t tag() const { return COMPACT ? q[2].w : q[4].w; }
    RR_DEV uint32_t dw(int i) const {
        const uint4& v = q[i >> 2];
        switch (i & 3) {
            case 0: return v.x;
            case 1: return v.y;
            case 2: return v.z;
            default: return v.w;
        }
    }
    RR_DEV double f64_at(int i) const { return f64_from(dw(2 * i), dw(2 * i + 1)); }  // i-th double of the payload
    RR_DEV double tri_coord(int i) const {  // i-th of p1.xyz p2.xyz p3.xyz
        if (COMPACT) return (double)__uint_as_float(dw(i));
        return f64_at(i);
    }
};

template <bool COMPACT>
RR_DEV PrimRec<COMPACT> load_prim(const void* prims, uint32_t p) {
    PrimRec<COMPACT> r;
    const uint4* src = reinterpret_cast<const uint4*>(prims) + (size_t)p * (COMPACT ? 3 : 5);
#pragma unroll
    for (int i = 0; i < (COMPACT ? 3 : 5); i++) r.q[i] = src[i];
    return r;
}

// --------------------------------------------------------------- primitives

// Sphere::intersect, geometry.rs:106-132
RR_DEV bool sphere_intersect(double radius2, V3 c, V3 o, V3 d, double& t) {
    const V3 odiff = v_sub(o, c);
    const double a = v_mag2(d);
    const double b = 2.0 * v_dot(d, odiff);
    const double cc = v_mag2(odiff) - radius2;
    const double desc = b * b - 4.0 * a * cc;
    if (desc > 0.0) {
        const double sq = rr_sqrt(desc);
        const double t1 = (-b - sq) / (2.0 * a);
        const double t2 = (-b + sq) / (2.0 * a);
        if (t1 < 0.0) {
            if (t2 < 0.0) return false;
            t = t2;
            return true;
        }
        t = t1;
        return true;
    }
    return false;
}

RR_DEV bool range_contains(double start, double end, double x) { return start <= x && x < end; }

// Plane::intersect, geometry.rs:229-271
RR_DEV bool plane_intersect(uint32_t axis, double u0, double u1, double v0, double v1, double pos, V3 o, V3 d,
                            double& t) {
    const uint32_t ax = axis >> 1;  // 0 X, 1 Y, 2 Z
    const double dn = ax == 0 ? d.x : (ax == 1 ? d.y : d.z);
    const double on = ax == 0 ? o.x : (ax == 1 ? o.y : o.z);
    if (dn != 0.0) {
        const double tt = (pos - on) / dn;
        const V3 p = v_add(o, v_scale(d, tt));
        const double pu = ax == 0 ? p.y : p.x;
        const double pv = ax == 2 ? p.y : p.z;
        if (range_contains(u0, u1, pu) && range_contains(v0, v1, pv)) {
            t = tt;
            return true;
        }
    }
    return false;
}

// Three IEEE quotients by one denominator.  The compiler expands x / y into v_div_scale (of y, and of x),
// v_rcp, two Newton steps on the reciprocal of the scaled y, one multiply, one residual, v_div_fmas and
// v_div_fixup.  Everything up to the refined reciprocal depends on the numerator only through the scaling
// of y (exponents far apart or near the ends of the range); when the three numerators scale y alike -- they
// do, unless one of them is extreme -- that part is computed once and each quotient finishes with its own
// four instructions: the same operations on the same operands as three separate divisions, hence the same
// bits, for 29 instead of 42 issue slots.  Otherwise: three separate divisions.
RR_DEV void div3_by(double n0, double n1, double n2, double y, double& q0, double& q1, double& q2) {
    bool unused, f0, f1, f2;
    const double sy = __builtin_amdgcn_div_scale(n0, y, false, &unused);
    const double sy1 = __builtin_amdgcn_div_scale(n1, y, false, &unused);
    const double sy2 = __builtin_amdgcn_div_scale(n2, y, false, &unused);
    if (rr_f64_bits(sy) == rr_f64_bits(sy1) && rr_f64_bits(sy) == rr_f64_bits(sy2)) {
        const double nsy = -sy;
        const double r = __builtin_amdgcn_rcp(sy);
        const double a0 = __builtin_fma(nsy, r, 1.0);
        const double r1 = __builtin_fma(r, a0, r);
        const double a1 = __builtin_fma(nsy, r1, 1.0);
        const double r2 = __builtin_fma(r1, a1, r1);
        const double s0 = __builtin_amdgcn_div_scale(n0, y, true, &f0);
        const double s1 = __builtin_amdgcn_div_scale(n1, y, true, &f1);
        const double s2 = __builtin_amdgcn_div_scale(n2, y, true, &f2);
        const double m0 = s0 * r2, m1 = s1 * r2, m2 = s2 * r2;
        q0 = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(__builtin_fma(nsy, m0, s0), r2, m0, f0), y, n0);
        q1 = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(__builtin_fma(nsy, m1, s1), r2, m1, f1), y, n1);
        q2 = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(__builtin_fma(nsy, m2, s2), r2, m2, f2), y, n2);
    } else {
        q0 = n0 / y, q1 = n1 / y, q2 = n2 / y;
    }
}

// Triangle::intersect, geometry.rs:359-375, with e1/e2 formed as Triangle::new does (:342-343)
RR_DEV bool triangle_intersect(V3 p1, V3 p2, V3 p3, V3 o, V3 d, double& t) {
    const V3 e1 = v_sub(p2, p1);
    const V3 e2 = v_sub(p3, p1);
    const V3 tt = v_sub(o, p1);
    const V3 p = v_cross(d, e2);
    const V3 q = v_cross(tt, e1);
    const double den = v_dot(p, e1);
    double dd, u, v;  // three true divisions (geometry.rs:364-374), sharing what they can
    div3_by(v_dot(q, e2), v_dot(p, tt), v_dot(q, d), den, dd, u, v);
    // (four compares OR-ed as wave masks: given lane booleans, the compiler folds the three "< 0" into
    // min(dd, u, v) < 0 -- three canonicalisations, two minima and a compare for three compares)
    const unsigned long long out = __builtin_amdgcn_ballot_w64(dd < 0.0) | __builtin_amdgcn_ballot_w64(u < 0.0) |
                                   __builtin_amdgcn_ballot_w64(v < 0.0) | __builtin_amdgcn_ballot_w64(u + v > 1.0);
    if (__builtin_amdgcn_inverse_ballot_w64(out)) return false;
    t = dd;
    return true;
}

template <bool COMPACT>
RR_DEV bool prim_intersect(const PrimRec<COMPACT>& r, V3 o, V3 d, double& t) {
    const uint32_t tag = r.tag();
    const uint32_t kind = tag & 3u;
    if (kind == PRIM_TRIANGLE) {
        const V3 p1 = mk(r.tri_coord(0), r.tri_coord(1), r.tri_coord(2));
        const V3 p2 = mk(r.tri_coord(3), r.tri_coord(4), r.tri_coord(5));
        const V3 p3 = mk(r.tri_coord(6), r.tri_coord(7), r.tri_coord(8));
        return triangle_intersect(p1, p2, p3, o, d, t);
    } else if (kind == PRIM_SPHERE) {
        return sphere_intersect(r.f64_at(0), mk(r.f64_at(1), r.f64_at(2), r.f64_at(3)), o, d, t);
    } else {
        return plane_intersect((tag >> 2) & 7u, r.f64_at(0), r.f64_at(1), r.f64_at(2), r.f64_at(3), r.f64_at(4), o, d,
                               t);
    }
}

// Hittable::normal: geometry.rs:134-136, :273-282, :377-379 (+ Triangle::new :344-351)
template <bool COMPACT>
RR_DEV V3 prim_normal(const PrimRec<COMPACT>& r, V3 position) {
    const uint32_t tag = r.tag();
    const uint32_t kind = tag & 3u;
    if (kind == PRIM_TRIANGLE) {
        const V3 p1 = mk(r.tri_coord(0), r.tri_coord(1), r.tri_coord(2));
        const V3 p2 = mk(r.tri_coord(3), r.tri_coord(4), r.tri_coord(5));
        const V3 p3 = mk(r.tri_coord(6), r.tri_coord(7), r.tri_coord(8));
        return v_unit(v_cross(v_sub(p2, p1), v_sub(p3, p1)));
    } else if (kind == PRIM_SPHERE) {
        return v_unit(v_sub(position, mk(r.f64_at(1), r.f64_at(2), r.f64_at(3))));
    } else {
        const uint32_t axis = (tag >> 2) & 7u;
        const double s = (axis & 1u) ? -1.0 : 1.0;
        const uint32_t ax = axis >> 1;
        return mk(ax == 0 ? s : 0.0, ax == 1 ? s : 0.0, ax == 2 ? s : 0.0);
    }
}

// ---------------------------------------------------------------- traversal

// AxisAlignedBoundingBox::intersect (geometry.rs:458-513) with 1/dir hoisted
// (the reference recomputes the same quotient at every node) and the
// `if inv < 0 { (hi*inv, lo*inv) } else { (lo*inv, hi*inv) }` swap applied to the
// bounds before the subtraction: (near - o) * inv and (far - o) * inv are the
// same two products.  tmin only grows and tmax only shrinks, so the single final
// compare equals the reference's three early-outs.  `entry` is the slab entry
// parameter used for ordering/culling.
RR_DEV bool slab(double xn, double xf, double yn, double yf, double zn, double zf, V3 o, V3 inv, double tmin,
                 double tmax, double& entry) {
    tmin = rr_max(tmin, (xn - o.x) * inv.x);
    tmax = rr_min(tmax, (xf - o.x) * inv.x);
    tmin = rr_max(tmin, (yn - o.y) * inv.y);
    tmax = rr_min(tmax, (yf - o.y) * inv.y);
    tmin = rr_max(tmin, (zn - o.z) * inv.z);
    tmax = rr_min(tmax, (zf - o.z) * inv.z);
    entry = tmin;
    return !(tmax <= tmin);
}

struct WorkCount {
    uint32_t interior, tri, sphere, plane;
    uint32_t leaf_prims;  // primitives of the last leaf visited in a macro step
};

// What the hot-group step counts, per wave (scalar registers: ballots and popcounts, no lane counters).
struct HotTally {
    uint32_t owed;      // rays that were put to the group's gating box
    uint32_t entered;   // ... and entered it: each tests every primitive of the group
    uint32_t divided;   // triangle tests that went on to the three divisions (not settled before them)
};

RR_DEV double f32bits_to_f64(uint32_t u) { return (double)__uint_as_float(u); }

// Bvh::intersect (bvh.rs:212-214, :391-415) as a resumable per-lane state
// machine.  Children are tested at the parent, the nearer one is entered first
// and the farther one pushed on the lane's LDS stack (entry k of lane l lives at
// stack[k * 64]); boxes entered beyond the closest hit so far are skipped.  The
// answer is the reference's: smallest accepted t, lowest DFS index on exact ties.
//
// A query advances by interior steps (one record) and leaf steps (one leaf
// reference); the kernel decides per wave which kind to run next.  A record is
// fetched whole, with loads that do not depend on its contents, before anything
// is decided from it.
constexpr uint32_t TRAV_DONE = 0xffffffffu;
// Closest-hit culling is the one place where the walk departs from the reference's: BvhTree::intersect never culls
// (bvh.rs:391-415), the walk skips a slot whose box is entered beyond best_t * TRAV_CULL_MARGIN.  That loses a
// hit only if a primitive's COMPUTED t lies in front of the entry parameter of a box around it by more than the margin
// AND another accepted hit falls in between.  Moeller-Trumbore's t has a relative error of about eps * (distance / size)
// / (grazing angle) -- unbounded as the ray approaches the triangle's plane -- and which triangles lie behind a box is
// not known without reading them, so NO margin computed from the ray and the box alone is sound: culling is either off
// (EXACT, the default walk: the margin is +infinity, the reference's visit set by construction) or a bet
// (rayrs_render_params.fast_traversal; headline frame: 4.72 instead of 6.32 records per query on its own tree,
// profiles/r05_walks.txt).  The fast walk's margin is the bet measured with scripts/fuzz_traversal.py
// (profiles/r03_fuzz_traversal.txt: 10^8 rays on sliver meshes and nearly flat sheets, origins up to 10^6 scene sizes
// away): rays at 10^-7 rad and more off a triangle's plane put t at most 2^-11.1 in front of a box; between 10^-9 and
// 10^-7 rad one ray in 10^7 loses its hit, closer to the plane one in 10^7 again (tests/test_walk_tree.py pins one: 2 % in
// front).  2^-10 costs 0.4 % more record visits and 0.7 % more primitive tests on the headline scene than 2^-40.
// best_t > t0 >= 0 (lib.rs:234), so best_t * inf = inf: nothing is beyond it, NaN entries included (!(NaN > x)).
constexpr double TRAV_CULL_MARGIN = 1.0 + 0x1p-10;

// A lane's traversal stack.  Entry k lives in LDS at lds[k * 64] while k < cap; deeper
// entries, which only the worst-case visit order of a deep tree reaches, go to a per-lane
// strip of HBM (entry k at spill[(k - cap) * stride]).  Sizing LDS for the common case
// instead of the bound is what lets five workgroups share a CU on the 1M-triangle scene.
struct LaneStack {
    uint32_t* lds;  // cap entries and one spare (never read) that branch-free pushes may scribble on
    uint32_t* spill;
    uint32_t cap, stride;
    RR_DEV void put(int k, uint32_t v) const {
        if ((uint32_t)k < cap) lds[k * 64] = v;
        else spill[(size_t)((uint32_t)k - cap) * stride] = v;
    }
    RR_DEV uint32_t get(int k) const {
        return (uint32_t)k < cap ? lds[k * 64] : spill[(size_t)((uint32_t)k - cap) * stride];
    }
};

struct Trav {
    V3 inv;
    double best_t;
    uint32_t best_prim;  // 0xffffffff = no hit yet
    uint32_t cur;        // reference to visit next, TRAV_DONE when finished
    int sp;
};

// What BvhTree::intersect does first (bvh.rs:394): the box of the root Node.  A ray that misses it
// is a Miss without anything else being looked at.
RR_DEV bool root_box_hit(const SceneDev& sc, V3 o, V3 inv) {
    const bool nx = inv.x < 0.0, ny = inv.y < 0.0, nz = inv.z < 0.0;
    double entry;
    return slab(nx ? sc.root_box[1] : sc.root_box[0], nx ? sc.root_box[0] : sc.root_box[1],
                ny ? sc.root_box[3] : sc.root_box[2], ny ? sc.root_box[2] : sc.root_box[3],
                nz ? sc.root_box[5] : sc.root_box[4], nz ? sc.root_box[4] : sc.root_box[5], o, inv, sc.t0, sc.t1,
                entry);
}

RR_DEV void trav_init(const SceneDev& sc, V3 o, V3 d, Trav& tv) {
    tv.inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    tv.best_t = sc.t1;
    tv.best_prim = 0xffffffffu;
    tv.sp = 0;
    tv.cur = root_box_hit(sc, o, tv.inv) ? sc.root_ref : TRAV_DONE;
}

RR_DEV void trav_pop(const LaneStack& stack, Trav& tv) {
    if (tv.sp > 0) {
        tv.sp--;
        tv.cur = stack.get(tv.sp);
    } else {
        tv.cur = TRAV_DONE;
    }
}

// One wide record (layout.h): up to four boxes tested, the hit slots entered nearest first
// (ties by slot), the others pushed farthest first.
RR_DEV bool slab_f32(uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1, uint32_t z0, uint32_t z1, bool nx, bool ny,
                     bool nz, V3 o, V3 inv, double tmin, double tmax, double& entry) {
    return slab(f32bits_to_f64(nx ? x1 : x0), f32bits_to_f64(nx ? x0 : x1), f32bits_to_f64(ny ? y1 : y0),
                f32bits_to_f64(ny ? y0 : y1), f32bits_to_f64(nz ? z1 : z0), f32bits_to_f64(nz ? z0 : z1), o, inv, tmin,
                tmax, entry);
}

RR_DEV bool slab_f64(uint4 x, uint4 y, uint4 z, bool nx, bool ny, bool nz, V3 o, V3 inv, double tmin, double tmax,
                     double& entry) {
    const double x0 = f64_from(x.x, x.y), x1 = f64_from(x.z, x.w);
    const double y0 = f64_from(y.x, y.y), y1 = f64_from(y.z, y.w);
    const double z0 = f64_from(z.x, z.y), z1 = f64_from(z.z, z.w);
    return slab(nx ? x1 : x0, nx ? x0 : x1, ny ? y1 : y0, ny ? y0 : y1, nz ? z1 : z0, nz ? z0 : z1, o, inv, tmin, tmax,
                entry);
}

// The first `count` wide records, copied to LDS by the traversal kernel: the records with the
// largest boxes (scene_host.cpp front_largest), which most queries read.  Random 16-byte reads
// cost the vector L1 about a cycle per lane; LDS serves them several times faster.  Records are
// spaced one granule (16 bytes) more than their size apart, so that lanes reading the same
// piece of 16 different records use 16 different bank groups.
struct HotNodes {
    const uint4* lds;
    uint32_t count;
    template <bool COMPACT>
    RR_DEV static constexpr uint32_t stride() { return COMPACT ? 9u : 17u; }  // granules
};

template <bool COMPACT, bool COUNT, bool EXACT = false>
RR_DEV void trav_interior_step(const SceneDev& sc, V3 o, const LaneStack& stack, const HotNodes& hot, Trav& tv,
                               WorkCount& wc) {
    const double tmin = sc.t0, tmax = sc.t1;
    const V3 inv = tv.inv;
    const bool nx = inv.x < 0.0, ny = inv.y < 0.0, nz = inv.z < 0.0;
    const uint32_t rec = tv.cur & 0x3fffffffu;
    if (COUNT) wc.interior++;
    double e0, e1, e2, e3;
    bool h0, h1, h2, h3;
    uint32_t r0, r1, r2, r3;
    if (COMPACT) {
        uint4 a, b, c, d, f, g, r;
        if (rec < hot.count) {
            const uint4* src = hot.lds + rec * HotNodes::stride<true>();
            a = src[0], b = src[1], c = src[2], d = src[3], f = src[4], g = src[5], r = src[6];
        } else {
            const uint4* src = reinterpret_cast<const uint4*>(sc.nodes) + (size_t)rec * 8;
            a = src[0], b = src[1], c = src[2], d = src[3], f = src[4], g = src[5], r = src[6];
        }
        r0 = r.x, r1 = r.y, r2 = r.z, r3 = r.w;
        // slot k = dwords 6k .. 6k+5 (xmin xmax ymin ymax zmin zmax)
        h0 = slab_f32(a.x, a.y, a.z, a.w, b.x, b.y, nx, ny, nz, o, inv, tmin, tmax, e0);
        h1 = slab_f32(b.z, b.w, c.x, c.y, c.z, c.w, nx, ny, nz, o, inv, tmin, tmax, e1);
        h2 = slab_f32(d.x, d.y, d.z, d.w, f.x, f.y, nx, ny, nz, o, inv, tmin, tmax, e2);
        h3 = slab_f32(f.z, f.w, g.x, g.y, g.z, g.w, nx, ny, nz, o, inv, tmin, tmax, e3);
    } else {
        // slot by slot (a whole 208-byte record in registers would not fit five waves per SIMD);
        // the scenes that need this layout are the sphere rows, whose few records all sit in LDS
        const bool in_lds = rec < hot.count;
        const uint4* lsrc = hot.lds + (in_lds ? rec : 0u) * HotNodes::stride<false>();
        const uint4* gsrc = reinterpret_cast<const uint4*>(sc.nodes) + (size_t)rec * 16;
        uint4 x, y, z, r;
        if (in_lds) x = lsrc[0], y = lsrc[1], z = lsrc[2], r = lsrc[12];
        else x = gsrc[0], y = gsrc[1], z = gsrc[2], r = gsrc[12];
        r0 = r.x, r1 = r.y, r2 = r.z, r3 = r.w;
        h0 = slab_f64(x, y, z, nx, ny, nz, o, inv, tmin, tmax, e0);
        if (in_lds) x = lsrc[3], y = lsrc[4], z = lsrc[5];
        else x = gsrc[3], y = gsrc[4], z = gsrc[5];
        h1 = slab_f64(x, y, z, nx, ny, nz, o, inv, tmin, tmax, e1);
        if (in_lds) x = lsrc[6], y = lsrc[7], z = lsrc[8];
        else x = gsrc[6], y = gsrc[7], z = gsrc[8];
        h2 = slab_f64(x, y, z, nx, ny, nz, o, inv, tmin, tmax, e2);
        if (in_lds) x = lsrc[9], y = lsrc[10], z = lsrc[11];
        else x = gsrc[9], y = gsrc[10], z = gsrc[11];
        h3 = slab_f64(x, y, z, nx, ny, nz, o, inv, tmin, tmax, e3);
    }
    // keep the references' load with the boxes' (the compiler would otherwise sink it below the
    // "any slot hit" branch, a second memory round trip per step)
    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
    // Unused slots need no special case here: they carry the inverted box (scene_host.cpp), for
    // which the slab test above says "missed".  Boxes entered beyond the closest hit so far are
    // skipped -- beyond it by TRAV_CULL_MARGIN (unless EXACT): a primitive's computed t and the entry parameter of the
    // box around it are rounded independently (and t badly so on grazing rays), so a hit computed in
    // front of its own box must not be lost to a farther one (every primitive that is tested is judged
    // by the reference's rule, so a wider margin only costs visits, never the answer).
    // Which slots are entered, and the six comparisons of their entry parameters below, are taken as wave masks (a
    // compare writes its mask to a scalar register pair) and combined there: "b before a" and its complement are
    // one comparison and one scalar operation, where the compiler, given lane booleans, issues a second f64
    // compare for every complement.
    // (a compile-time choice: the margin as a kernel argument is one more scalar pair alive across the walk, which the
    // traversal kernel answers by re-loading arguments from memory inside its loop -- +27 % kernel time, measured)
    const double cull = EXACT ? (double)__builtin_inf() : tv.best_t * TRAV_CULL_MARGIN;
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(h0) & __builtin_amdgcn_ballot_w64(!(e0 > cull));
    const unsigned long long m1 = __builtin_amdgcn_ballot_w64(h1) & __builtin_amdgcn_ballot_w64(!(e1 > cull));
    const unsigned long long m2 = __builtin_amdgcn_ballot_w64(h2) & __builtin_amdgcn_ballot_w64(!(e2 > cull));
    const unsigned long long m3 = __builtin_amdgcn_ballot_w64(h3) & __builtin_amdgcn_ballot_w64(!(e3 > cull));
#define RR_LANE_BIT(mask) __builtin_amdgcn_inverse_ballot_w64(mask)
    h0 = RR_LANE_BIT(m0), h1 = RR_LANE_BIT(m1), h2 = RR_LANE_BIT(m2), h3 = RR_LANE_BIT(m3);
    const int n = (int)h0 + (int)h1 + (int)h2 + (int)h3;
    if (n == 0) {
        trav_pop(stack, tv);
        return;
    }
    // rank of a hit slot = number of hit slots visited before it (nearer entry, then lower slot);
    // for a < b, slot a goes first unless e_b < e_a.  (EXACT: every slot the ray enters is visited whatever the
    // closest hit, so the order buys nothing -- the closest hit is the smallest accepted t, ties by primitive
    // index, in any order: slot order, and the six comparisons are not made.)
    int k_0, k_1, k_2, k_3;
    if (EXACT) {
        k_0 = 0, k_1 = (int)h0, k_2 = (int)h0 + (int)h1, k_3 = (int)h0 + (int)h1 + (int)h2;
    } else {
        const unsigned long long c01 = __builtin_amdgcn_ballot_w64(e1 < e0), c02 = __builtin_amdgcn_ballot_w64(e2 < e0);
        const unsigned long long c03 = __builtin_amdgcn_ballot_w64(e3 < e0), c12 = __builtin_amdgcn_ballot_w64(e2 < e1);
        const unsigned long long c13 = __builtin_amdgcn_ballot_w64(e3 < e1), c23 = __builtin_amdgcn_ballot_w64(e3 < e2);
        k_0 = (int)RR_LANE_BIT(m1 & c01) + (int)RR_LANE_BIT(m2 & c02) + (int)RR_LANE_BIT(m3 & c03);
        k_1 = (int)RR_LANE_BIT(m0 & ~c01) + (int)RR_LANE_BIT(m2 & c12) + (int)RR_LANE_BIT(m3 & c13);
        k_2 = (int)RR_LANE_BIT(m0 & ~c02) + (int)RR_LANE_BIT(m1 & ~c12) + (int)RR_LANE_BIT(m3 & c23);
        k_3 = (int)RR_LANE_BIT(m0 & ~c03) + (int)RR_LANE_BIT(m1 & ~c13) + (int)RR_LANE_BIT(m2 & ~c23);
    }
#undef RR_LANE_BIT
    tv.cur = (h0 && k_0 == 0) ? r0 : (h1 && k_1 == 0) ? r1 : (h2 && k_2 == 0) ? r2 : r3;
    // rank k >= 1 goes to stack entry top - k; everything else to the lane's spare entry
    const int top = tv.sp + n - 1;
    tv.sp = top;
    if (__ballot((uint32_t)top > stack.cap) == 0ull) {  // all of the wave's entries are in LDS: no branches
        // (the rank-0 slot is written too, to entry `top`: the first free one above the new stack top, never read
        // before it is overwritten, and at most the spare entry -- one comparison per slot less)
        const int spare = (int)stack.cap;
        stack.lds[(h0 ? top - k_0 : spare) * 64] = r0;
        stack.lds[(h1 ? top - k_1 : spare) * 64] = r1;
        stack.lds[(h2 ? top - k_2 : spare) * 64] = r2;
        stack.lds[(h3 ? top - k_3 : spare) * 64] = r3;
    } else {
        if (h0 && k_0 > 0) stack.put(top - k_0, r0);
        if (h1 && k_1 > 0) stack.put(top - k_1, r1);
        if (h2 && k_2 > 0) stack.put(top - k_2, r2);
        if (h3 && k_3 > 0) stack.put(top - k_3, r3);
    }
}

// One leaf reference: its 1..4 primitives in DFS order, then pop.
template <bool COMPACT, bool COUNT>
RR_DEV void trav_leaf_step(const SceneDev& sc, V3 o, V3 d, const LaneStack& stack, Trav& tv, WorkCount& wc) {
    const double tmin = sc.t0, tmax = sc.t1;
    const uint32_t first = (tv.cur & 0x3fffffffu) >> 2;
    const uint32_t count = (tv.cur & 3u) + 1u;
    if (COUNT) wc.leaf_prims = count;
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t p = first + k;
        const PrimRec<COMPACT> r = load_prim<COMPACT>(sc.prims, p);
        if (COUNT) {
            const uint32_t kind = r.tag() & 3u;
            if (kind == PRIM_TRIANGLE) wc.tri++;
            else if (kind == PRIM_SPHERE) wc.sphere++;
            else wc.plane++;
        }
        double t;
        if (prim_intersect<COMPACT>(r, o, d, t) && t > tmin && t < tmax) {  // bvh.rs:406
            if (t < tv.best_t || (t == tv.best_t && p < tv.best_prim)) {    // bvh.rs:62
                tv.best_t = t;
                tv.best_prim = p;
            }
        }
    }
    trav_pop(stack, tv);
}

RR_DEV bool trav_at_interior(const Trav& tv) { return (tv.cur >> 30) == REF_INTERIOR; }

// ---- the default walk with a lane's leaf groups SET ASIDE (wavefront.hip wf_trav_kernel<.., EXACT>) ----
// The default walk culls nothing: which groups a ray's primitives are tested of is decided by the gating boxes alone, and
// the closest hit is the smallest accepted t, the first primitive in depth-first order on exact ties (bvh.rs:62), in ANY
// visiting order.  So a lane need not stop at a leaf slot: it puts the group's reference on a queue of its own (LEAFQ entries
// in LDS behind its stack) and walks on; a leaf phase of the wave serves one queued group per lane.  A lane then takes part
// in an interior phase whenever it has a record to visit and in a leaf phase whenever it has a group waiting, instead of
// standing idle through the phases of the other kind (lanes at work per step 0.59 -> 0.8, scripts/sim/walk_sched_sim.py).
// Trav::cur is an interior record or TRAV_DONE, the stack holds interior records only, and Trav::sp carries two counts:
// stack height in its low half, queued groups in its high half.  A lane whose queue could not take four more groups sits
// out interior phases until a leaf phase has served it (the queue never overflows).
constexpr uint32_t LEAFQ = TRAV_LEAFQ;                // queue entries per lane (layout.h)
constexpr uint32_t LEAFQ_ONE = 1u << 16;              // one queued group, in Trav::sp
constexpr uint32_t LEAFQ_ROOM = (LEAFQ - 3u) << 16;   // sp below this: four more groups fit
constexpr uint32_t REF_LEAF_BASE = REF_RANGE << 30;   // an entered slot's reference at or above this is a leaf group
RR_DEV bool defer_has_leaf(const Trav& tv) { return (uint32_t)tv.sp >= LEAFQ_ONE; }
RR_DEV bool defer_has_room(const Trav& tv) { return (uint32_t)tv.sp < LEAFQ_ROOM; }
RR_DEV bool defer_finished(const Trav& tv) { return tv.cur == TRAV_DONE && (uint32_t)tv.sp < LEAFQ_ONE; }

// The lane's queue lives behind its stack's spare entry: entry cap + 1 + k of the lane's LDS column.
RR_DEV uint32_t* defer_queue(const LaneStack& stack) { return stack.lds + (stack.cap + 1u) * 64u; }

// A reference the lane is handed from outside a record (the root of a tree, the first record's slots at a refill).
RR_DEV void defer_take_ref(const LaneStack& stack, Trav& tv, uint32_t ref) {
    if (ref >= REF_LEAF_BASE) {
        defer_queue(stack)[((uint32_t)tv.sp >> 16) * 64u] = ref;
        tv.sp += (int)LEAFQ_ONE;
    } else if (tv.cur == TRAV_DONE) {
        tv.cur = ref;
    } else {
        stack.put((int)((uint32_t)tv.sp & 0xffffu), ref);
        tv.sp += 1;
    }
}

template <bool COMPACT, bool COUNT>
RR_DEV void trav_interior_step_defer(const SceneDev& sc, V3 o, const LaneStack& stack, const HotNodes& hot, Trav& tv,
                                     WorkCount& wc) {
    const double tmin = sc.t0, tmax = sc.t1;
    const V3 inv = tv.inv;
    const bool nx = inv.x < 0.0, ny = inv.y < 0.0, nz = inv.z < 0.0;
    const uint32_t rec = tv.cur & 0x3fffffffu;
    if (COUNT) wc.interior++;
    double e0, e1, e2, e3;
    bool h0, h1, h2, h3;
    uint32_t r0, r1, r2, r3;
    if (COMPACT) {
        uint4 a, b, c, d, f, g, r;
        if (rec < hot.count) {
            const uint4* src = hot.lds + rec * HotNodes::stride<true>();
            a = src[0], b = src[1], c = src[2], d = src[3], f = src[4], g = src[5], r = src[6];
        } else {
            const uint4* src = reinterpret_cast<const uint4*>(sc.nodes) + (size_t)rec * 8;
            a = src[0], b = src[1], c = src[2], d = src[3], f = src[4], g = src[5], r = src[6];
        }
        r0 = r.x, r1 = r.y, r2 = r.z, r3 = r.w;
        h0 = slab_f32(a.x, a.y, a.z, a.w, b.x, b.y, nx, ny, nz, o, inv, tmin, tmax, e0);
        h1 = slab_f32(b.z, b.w, c.x, c.y, c.z, c.w, nx, ny, nz, o, inv, tmin, tmax, e1);
        h2 = slab_f32(d.x, d.y, d.z, d.w, f.x, f.y, nx, ny, nz, o, inv, tmin, tmax, e2);
        h3 = slab_f32(f.z, f.w, g.x, g.y, g.z, g.w, nx, ny, nz, o, inv, tmin, tmax, e3);
    } else {
        const bool in_lds = rec < hot.count;
        const uint4* lsrc = hot.lds + (in_lds ? rec : 0u) * HotNodes::stride<false>();
        const uint4* gsrc = reinterpret_cast<const uint4*>(sc.nodes) + (size_t)rec * 16;
        uint4 x, y, z, r;
        if (in_lds) x = lsrc[0], y = lsrc[1], z = lsrc[2], r = lsrc[12];
        else x = gsrc[0], y = gsrc[1], z = gsrc[2], r = gsrc[12];
        r0 = r.x, r1 = r.y, r2 = r.z, r3 = r.w;
        h0 = slab_f64(x, y, z, nx, ny, nz, o, inv, tmin, tmax, e0);
        if (in_lds) x = lsrc[3], y = lsrc[4], z = lsrc[5];
        else x = gsrc[3], y = gsrc[4], z = gsrc[5];
        h1 = slab_f64(x, y, z, nx, ny, nz, o, inv, tmin, tmax, e1);
        if (in_lds) x = lsrc[6], y = lsrc[7], z = lsrc[8];
        else x = gsrc[6], y = gsrc[7], z = gsrc[8];
        h2 = slab_f64(x, y, z, nx, ny, nz, o, inv, tmin, tmax, e2);
        if (in_lds) x = lsrc[9], y = lsrc[10], z = lsrc[11];
        else x = gsrc[9], y = gsrc[10], z = gsrc[11];
        h3 = slab_f64(x, y, z, nx, ny, nz, o, inv, tmin, tmax, e3);
    }
    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));  // (the references' load stays with the boxes': see trav_interior_step)
    // Entered slots, as wave masks (unused slots carry the inverted box: never entered), split by what the slot refers to.
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(h0), m1 = __builtin_amdgcn_ballot_w64(h1);
    const unsigned long long m2 = __builtin_amdgcn_ballot_w64(h2), m3 = __builtin_amdgcn_ballot_w64(h3);
    const unsigned long long q0 = __builtin_amdgcn_ballot_w64(r0 >= REF_LEAF_BASE), q1 = __builtin_amdgcn_ballot_w64(r1 >= REF_LEAF_BASE);
    const unsigned long long q2 = __builtin_amdgcn_ballot_w64(r2 >= REF_LEAF_BASE), q3 = __builtin_amdgcn_ballot_w64(r3 >= REF_LEAF_BASE);
#define RR_LANE_BIT(mask) __builtin_amdgcn_inverse_ballot_w64(mask)
    const bool i0 = RR_LANE_BIT(m0 & ~q0), i1 = RR_LANE_BIT(m1 & ~q1), i2 = RR_LANE_BIT(m2 & ~q2), i3 = RR_LANE_BIT(m3 & ~q3);
    const bool l0 = RR_LANE_BIT(m0 & q0), l1 = RR_LANE_BIT(m1 & q1), l2 = RR_LANE_BIT(m2 & q2), l3 = RR_LANE_BIT(m3 & q3);
#undef RR_LANE_BIT
    // rank of an entered slot among the entered slots of its kind, in slot order
    const int ki1 = (int)i0, ki2 = ki1 + (int)i1, ki3 = ki2 + (int)i2, n_int = ki3 + (int)i3;
    const int kl1 = (int)l0, kl2 = kl1 + (int)l1, kl3 = kl2 + (int)l2, n_leaf = kl3 + (int)l3;
    const int sp = (int)((uint32_t)tv.sp & 0xffffu);
    const int lq = (int)((uint32_t)tv.sp >> 16);
    // interior slots: the first becomes the lane's next record, rank k >= 1 goes to stack entry top - k (they come off
    // in slot order); leaf slots: queue entries lq + rank; everything else to the lane's spare entry
    const int top = sp + n_int - 1;  // the new stack height if n_int >= 1
    const int spare = (int)stack.cap;
    const int qbase = spare + 1 + lq;
    const int d0 = i0 ? top : (l0 ? qbase : spare);
    const int d1 = i1 ? top - ki1 : (l1 ? qbase + kl1 : spare);
    const int d2 = i2 ? top - ki2 : (l2 ? qbase + kl2 : spare);
    const int d3 = i3 ? top - ki3 : (l3 ? qbase + kl3 : spare);
    if (__ballot(n_int > 0 && (uint32_t)top > stack.cap) == 0ull) {  // all of the wave's stack entries are in LDS: no branches
        // (the rank-0 interior slot is written too, to entry `top`: the first free one above the new stack top)
        stack.lds[d0 * 64] = r0;
        stack.lds[d1 * 64] = r1;
        stack.lds[d2 * 64] = r2;
        stack.lds[d3 * 64] = r3;
    } else {
        if (l0) stack.lds[d0 * 64] = r0;  // (an interior slot 0 has rank 0: it becomes the lane's next record)
        if (l1) stack.lds[d1 * 64] = r1;
        else if (i1 && ki1 > 0) stack.put(top - ki1, r1);
        if (l2) stack.lds[d2 * 64] = r2;
        else if (i2 && ki2 > 0) stack.put(top - ki2, r2);
        if (l3) stack.lds[d3 * 64] = r3;
        else if (i3 && ki3 > 0) stack.put(top - ki3, r3);
    }
    if (n_int > 0) {
        tv.cur = i0 ? r0 : i1 ? r1 : i2 ? r2 : r3;
        tv.sp = (int)(((uint32_t)(lq + n_leaf) << 16) | (uint32_t)top);
    } else if (sp > 0) {
        tv.cur = stack.get(sp - 1);
        tv.sp = (int)(((uint32_t)(lq + n_leaf) << 16) | (uint32_t)(sp - 1));
    } else {
        tv.cur = TRAV_DONE;
        tv.sp = (int)((uint32_t)(lq + n_leaf) << 16);
    }
}

// One queued leaf group of the lane: its 1..4 primitives in DFS order.
template <bool COMPACT, bool COUNT>
RR_DEV void trav_leaf_step_defer(const SceneDev& sc, V3 o, V3 d, const LaneStack& stack, Trav& tv, WorkCount& wc) {
    const double tmin = sc.t0, tmax = sc.t1;
    tv.sp -= (int)LEAFQ_ONE;
    const uint32_t ref = defer_queue(stack)[((uint32_t)tv.sp >> 16) * 64u];
    const uint32_t first = (ref & 0x3fffffffu) >> 2;
    const uint32_t count = (ref & 3u) + 1u;
    if (COUNT) wc.leaf_prims = count;
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t p = first + k;
        const PrimRec<COMPACT> r = load_prim<COMPACT>(sc.prims, p);
        if (COUNT) {
            const uint32_t kind = r.tag() & 3u;
            if (kind == PRIM_TRIANGLE) wc.tri++;
            else if (kind == PRIM_SPHERE) wc.sphere++;
            else wc.plane++;
        }
        double t;
        if (prim_intersect<COMPACT>(r, o, d, t) && t > tmin && t < tmax) {  // bvh.rs:406
            if (t < tv.best_t || (t == tv.best_t && p < tv.best_prim)) {    // bvh.rs:62
                tv.best_t = t;
                tv.best_prim = p;
            }
        }
    }
}

// ------------------------------------------------------- the hot group (layout.h HotGroupDev)

// Read with scalar loads: the data is the same for every lane, and a load through the constant address space with a
// wave-uniform address is one s_load for the wave -- its values are scalar operands of the vector arithmetic below,
// no vector register holds them and nothing is converted.
typedef const __attribute__((address_space(4))) HotGroupDev* HotPtr;
RR_DEV HotPtr hot_ptr(const SceneDev& sc) { return (HotPtr)sc.hot; }

// Triangle::intersect (geometry.rs:359-375) on a wave-uniform triangle -- p1 and the e1 = p2 - p1, e2 = p3 - p1 that
// Triangle::new stores (geometry.rs:342-343) -- with the three divisions made only if some lane of the wave needs them.
//
// What the reference decides first is `d < 0. || u < 0. || v < 0. || u + v > 1.` on the three quotients
// d = n0 / den, u = n1 / den, v = n2 / den.  IEEE-754 division is correctly rounded (on this chip too:
// tests/test_gpu_functions.py), so two facts about a quotient q = fl(n / den) need no division:
//   (S) q < 0 holds if n and den have opposite sign bits, |n| >= 2^-500 (which a NaN is not, and an infinity is) and
//       |den| <= 2^500 (a zero is, a NaN or an infinity is not): for den = +-0 the quotient is the infinity of the
//       product's sign, -inf; for an infinite n and such a den likewise; otherwise the exact quotient is negative and at
//       least 2^-1000 in magnitude, and rounding to nearest keeps both;
//   (B) q >= 2 holds if n and den have equal sign bits, 2^-500 <= |den| <= 2^500 and |n| >= 2 |den| (2 |den| is exact): the
//       exact quotient is >= 2, 2 is a double, rounding is monotone (+inf included).
// (S) for any of the three numerators rejects the ray.  (B) for u rejects it if v is a number -- |n2| <= 2^500 with
// such a den makes it a finite one: either v < 0, or v >= 0 (a zero of either sign included) and then fl(u + v) >= 2 > 1
// by monotonicity again -- and the same with u and v exchanged.  A ray that passes a small triangle at a distance
// has barycentric coordinates far outside [0, 2), so whole waves leave here, after the two cross products and four
// dot products that the reference computes as well; a wave in which some lane is not settled goes on to the
// divisions with every lane, whose compares then reject the settled lanes again (same test, same verdict).
// The numbers on the hot path are the reference's: same operands, same operations, same order.
RR_DEV bool hot_triangle_intersect(V3 p1, V3 e1, V3 e2, V3 o, V3 d, bool live, double& t, HotTally& ht) {
    const V3 tt = v_sub(o, p1);
    const V3 p = v_cross(d, e2);
    const V3 q = v_cross(tt, e1);
    const double den = v_dot(p, e1);
    const double n0 = v_dot(q, e2), n1 = v_dot(p, tt), n2 = v_dot(q, d);
    {
        const uint32_t dh = (uint32_t)(rr_f64_bits(den) >> 32);
        const unsigned long long opp0 = __builtin_amdgcn_ballot_w64((int)((uint32_t)(rr_f64_bits(n0) >> 32) ^ dh) < 0);
        const unsigned long long opp1 = __builtin_amdgcn_ballot_w64((int)((uint32_t)(rr_f64_bits(n1) >> 32) ^ dh) < 0);
        const unsigned long long opp2 = __builtin_amdgcn_ballot_w64((int)((uint32_t)(rr_f64_bits(n2) >> 32) ^ dh) < 0);
        const double aden = rr_fabs(den), a0 = rr_fabs(n0), a1 = rr_fabs(n1), a2 = rr_fabs(n2);
        const unsigned long long den_le = __builtin_amdgcn_ballot_w64(aden <= 0x1p500);
        const unsigned long long den_ge = __builtin_amdgcn_ballot_w64(aden >= 0x1p-500);
        const unsigned long long big0 = __builtin_amdgcn_ballot_w64(a0 >= 0x1p-500);
        const unsigned long long big1 = __builtin_amdgcn_ballot_w64(a1 >= 0x1p-500);
        const unsigned long long big2 = __builtin_amdgcn_ballot_w64(a2 >= 0x1p-500);
        const unsigned long long fin1 = __builtin_amdgcn_ballot_w64(a1 <= 0x1p500);
        const unsigned long long fin2 = __builtin_amdgcn_ballot_w64(a2 <= 0x1p500);
        const double two_den = aden * 2.0;
        const unsigned long long two1 = __builtin_amdgcn_ballot_w64(a1 >= two_den);
        const unsigned long long two2 = __builtin_amdgcn_ballot_w64(a2 >= two_den);
        const unsigned long long settled = (den_le & ((opp0 & big0) | (opp1 & big1) | (opp2 & big2))) |
                                           (den_le & den_ge & ((~opp1 & two1 & fin2) | (~opp2 & two2 & fin1)));
        if ((__builtin_amdgcn_ballot_w64(live) & ~settled) == 0ull) return false;
    }
    ht.divided += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(live));
    double dd, u, v;
    div3_by(n0, n1, n2, den, dd, u, v);
    const unsigned long long out = __builtin_amdgcn_ballot_w64(dd < 0.0) | __builtin_amdgcn_ballot_w64(u < 0.0) |
                                   __builtin_amdgcn_ballot_w64(v < 0.0) | __builtin_amdgcn_ballot_w64(u + v > 1.0);
    if (__builtin_amdgcn_inverse_ballot_w64(out)) return false;
    t = dd;
    return true;
}

// AxisAlignedBoundingBox::intersect (geometry.rs:458-513) on a wave-uniform box, as slab() computes it: the near / far
// bound is chosen by the sign of 1 / d -- here after the two products instead of before (the same two products either way).
RR_DEV bool uniform_box_entered(double x0, double x1, double y0, double y1, double z0, double z1, V3 o, V3 inv, double tmin,
                                double tmax) {
    const bool nx = inv.x < 0.0, ny = inv.y < 0.0, nz = inv.z < 0.0;
    const double ax = (x0 - o.x) * inv.x, bx = (x1 - o.x) * inv.x;
    const double ay = (y0 - o.y) * inv.y, by = (y1 - o.y) * inv.y;
    const double az = (z0 - o.z) * inv.z, bz = (z1 - o.z) * inv.z;
    double lo = tmin, hi = tmax;
    lo = rr_max(lo, nx ? bx : ax), hi = rr_min(hi, nx ? ax : bx);
    lo = rr_max(lo, ny ? by : ay), hi = rr_min(hi, ny ? ay : by);
    lo = rr_max(lo, nz ? bz : az), hi = rr_min(hi, nz ? az : bz);
    return !(hi <= lo);
}

// What BvhTree::intersect does with the hot group (bvh.rs:396-410): the gating box, and if the ray enters it, the
// group's primitives in depth-first order.  Every lane of the wave that owes the test (`owe`; the others idle) runs it
// here, once per ray -- in the kernel that made the ray (wavefront.hip finish_rays) or at the start of a query
// (bvh_intersect below): the closest hit is the smallest accepted t, the first primitive in depth-first order on exact
// ties (bvh.rs:62), in any visiting order.
template <bool COUNT>
RR_DEV void hot_group_step(const SceneDev& sc, V3 o, V3 d, bool owe, Trav& tv, WorkCount& wc, HotTally& ht) {
    const HotPtr h = hot_ptr(sc);
    const double tmin = sc.t0, tmax = sc.t1;
    const V3 inv = tv.inv;
    const bool entered = owe && uniform_box_entered(h->box[0], h->box[1], h->box[2], h->box[3], h->box[4], h->box[5], o, inv, tmin, tmax);
    const unsigned long long entered_mask = __builtin_amdgcn_ballot_w64(entered);
    ht.owed += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(owe));
    ht.entered += (uint32_t)__popcll(entered_mask);
    if (entered_mask == 0ull) return;
    const uint32_t first = h->first, count = h->count;
#pragma nounroll
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t tag = h->prim[k].tag;
        const uint32_t kind = tag & 3u;
        double t = 0.0;
        bool hit;
        if (kind == PRIM_TRIANGLE) {
            if (COUNT && entered) wc.tri++;
            hit = hot_triangle_intersect(mk(h->prim[k].v[0], h->prim[k].v[1], h->prim[k].v[2]),
                                         mk(h->prim[k].v[3], h->prim[k].v[4], h->prim[k].v[5]),
                                         mk(h->prim[k].v[6], h->prim[k].v[7], h->prim[k].v[8]), o, d, entered, t, ht);
        } else if (kind == PRIM_SPHERE) {
            if (COUNT && entered) wc.sphere++;
            hit = sphere_intersect(h->prim[k].v[0], mk(h->prim[k].v[1], h->prim[k].v[2], h->prim[k].v[3]), o, d, t);
        } else {
            if (COUNT && entered) wc.plane++;
            hit = plane_intersect((tag >> 2) & 7u, h->prim[k].v[0], h->prim[k].v[1], h->prim[k].v[2], h->prim[k].v[3],
                                  h->prim[k].v[4], o, d, t);
        }
        const uint32_t p = first + k;
        if (entered && hit && t > tmin && t < tmax) {                     // bvh.rs:406
            if (t < tv.best_t || (t == tv.best_t && p < tv.best_prim)) {  // bvh.rs:62
                tv.best_t = t;
                tv.best_prim = p;
            }
        }
    }
}

// The first record of the tree without the hot group (HotGroupDev::root_box, the f64 values its record holds): does the ray
// enter which of its four slots?  The same test, on the same values, that trav_interior_step makes of that record.
RR_DEV uint32_t hot_root_record_entered(const SceneDev& sc, V3 o, V3 inv) {  // bit c: the ray enters slot c
    const HotPtr h = hot_ptr(sc);
    const double tmin = sc.t0, tmax = sc.t1;
    uint32_t mask = 0u;
#pragma unroll
    for (int c = 0; c < 4; c++)
        mask |= uniform_box_entered(h->root_box[c][0], h->root_box[c][1], h->root_box[c][2], h->root_box[c][3], h->root_box[c][4],
                                    h->root_box[c][5], o, inv, tmin, tmax) ? (1u << c) : 0u;
    return mask;
}

template <bool COMPACT, bool COUNT, bool EXACT = false>
RR_DEV bool bvh_intersect(const SceneDev& sc, V3 o, V3 d, const LaneStack& stack, double& t_hit, uint32_t& prim_hit,
                          WorkCount& wc) {
    Trav tv;
    trav_init(sc, o, d, tv);
    const HotNodes hot{nullptr, 0u};
    // (the default walk on a scene with a hot group: every ray that enters the root box owes the group its test; the
    // wave-level calls inside need every lane of the wave here, so the branch is on the wave-uniform pointer only)
    HotTally ht{0, 0, 0};
    if (EXACT && sc.hot != nullptr) hot_group_step<COUNT>(sc, o, d, tv.cur != TRAV_DONE, tv, wc, ht);
    while (tv.cur != TRAV_DONE) {
        if (trav_at_interior(tv))
            trav_interior_step<COMPACT, COUNT, EXACT>(sc, o, stack, hot, tv, wc);
        else
            trav_leaf_step<COMPACT, COUNT>(sc, o, d, stack, tv, wc);
    }
    t_hit = tv.best_t;
    prim_hit = tv.best_prim;
    return tv.best_prim != 0xffffffffu;
}

// ---------------------------------------------------------------- materials

struct CtLayer {  // struct CookTorrance, material.rs:194-200
    double alpha2;
    int metallic;
    double ior;
    V3 r0;
    V3 color;
};

RR_DEV CtLayer load_ct(const SurfaceDev* s) {
    CtLayer ct;
    ct.alpha2 = s->ct_alpha2;
    ct.metallic = s->metallic;
    ct.ior = s->ct_ior;
    ct.r0 = mk(s->ct_r0[0], s->ct_r0[1], s->ct_r0[2]);
    ct.color = mk(s->ct_color[0], s->ct_color[1], s->ct_color[2]);
    return ct;
}

RR_DEV double dir_ior_ratio(bool entering, double ior) { return entering ? 1.0 / ior : ior; }  // :1200-1205
RR_DEV V3 dir_normal(bool entering, V3 n) { return entering ? n : v_scale(n, -1.0); }          // :1215-1220

RR_DEV double schlick_scalar(double ior_curr, double ior_new, V3 n, V3 v) {  // :1472-1479
    double r0 = (ior_curr - ior_new) / (ior_curr + ior_new);
    r0 = r0 * r0;
    return r0 + (1.0 - r0) * pow5(1.0 - v_dot(n, v));
}
RR_DEV V3 schlick_vec(V3 r0, V3 n, V3 v) {  // :1484-1489
    const double p = pow5(1.0 - v_dot(n, v));
    return v_add(r0, v_scale(v_sub(mk(1.0, 1.0, 1.0), r0), p));
}
RR_DEV V3 reflect(V3 n, V3 v) { return v_sub(v_scale(n, 2.0 * v_dot(v, n)), v); }  // :1492-1496
RR_DEV bool refract(V3 n, V3 v, double ior_ratio, V3& out) {                         // :1502-1518
    const double cos_theta = v_dot(v, n);
    const double sin_theta = rr_sqrt(1.0 - cos_theta * cos_theta);
    if (ior_ratio * sin_theta > 1.0) return false;
    const V3 par = v_scale(v_sub(v_scale(n, cos_theta), v), ior_ratio);
    const V3 perp = v_scale(n, -rr_sqrt(1.0 - v_mag2(par)));
    out = v_add(perp, par);
    return true;
}

RR_DEV V3 fresnel_value(const CtLayer& ct, V3 n, V3 v, bool entering) {  // :1457-1469
    if (!ct.metallic) {
        const double f = entering ? schlick_scalar(1.0, ct.ior, n, v) : schlick_scalar(ct.ior, 1.0, n, v);
        return mk(f, f, f);
    }
    return schlick_vec(ct.r0, n, v);
}

RR_DEV double beckmann_from_tan(double tan_theta_h, double alpha2, double nh) {  // :940, :1310, :1414
    return rr_exp(-tan_theta_h * tan_theta_h / alpha2) / (RR_PI * alpha2 * pow4(nh));
}

// The reference evaluates the Beckmann term of the half vector twice for a Cook-Torrance reflection: once in
// the pdf (Pdf::Beckmann::value :915-941, on |n.h|) and once in the BRDF (:1276-1322, on n.h).  l + v and
// v + l are the same vector, so whenever n.h >= 0 the two evaluations have the same inputs bit for bit
// (acos, tan, exp, powi(4), two divisions): the pdf leaves what it computed here and the BRDF takes it.
struct CtHalf {
    bool valid;            // h is the unit half vector; tan_abs (and, unless it is infinite, beckmann_abs) come from fabs(nh)
    V3 h;
    double nh;             // n . h, signed
    double tan_abs, beckmann_abs;
};

RR_DEV V3 ct_brdf(const CtLayer& ct, V3 n, V3 l, V3 v, const CtHalf* half = nullptr) {  // :1276-1322
    const double nv = rr_fabs(v_dot(n, v));
    const double nl = rr_fabs(v_dot(n, l));
    if (nv == 0.0 || nl == 0.0) return mk(0, 0, 0);
    V3 h;
    double nh;
    if (half != nullptr && half->valid) {  // (l + v).unit() of the pdf: the same vector, and not zero
        h = half->h, nh = half->nh;
    } else {
        h = v_add(v, l);
        if (v_is_zeros(h)) return mk(0, 0, 0);
        h = v_unit(h);
        nh = v_dot(n, h);
    }
    double tan_theta_h, beckmann = 0.0;
    const bool shared = half != nullptr && half->valid && nh >= 0.0;  // then fabs(nh) and nh are the same number
    if (shared) {
        tan_theta_h = half->tan_abs;
    } else {
        tan_theta_h = rr_tan(rr_acos(nh));
    }
    if (__builtin_isinf(tan_theta_h)) return mk(0, 0, 0);
    if (shared) beckmann = half->beckmann_abs;
    else beckmann = beckmann_from_tan(tan_theta_h, ct.alpha2, nh);
    const double hv = v_dot(h, v);
    const double g = rr_min(2.0 * nh * nv / hv, rr_min(2.0 * nh * nl / hv, 1.0));
    const V3 f = fresnel_value(ct, h, v, true);
    V3 r = v_mul(ct.color, f);
    r = v_scale(r, beckmann);
    r = v_scale(r, g);
    return v_div(r, 4.0 * nv * nl);
}

RR_DEV V3 ct_btdf(const CtLayer& ct, V3 n, V3 l, V3 v, bool entering) {  // :1362-1442
    const double nv = rr_fabs(v_dot(n, v));
    const double nl = rr_fabs(v_dot(n, l));
    const double ior_ratio = dir_ior_ratio(entering, ct.ior);
    V3 h;
    if (ior_ratio > 1.0)
        h = v_add(l, v_scale(v, ior_ratio));
    else
        h = v_sub(v_scale(v, -ior_ratio), l);
    if (nv == 0.0 || nl == 0.0) return mk(0, 0, 0);
    if (v_is_zeros(h)) return mk(0, 0, 0);
    h = v_unit(h);
    const double nh = v_dot(n, h);
    const double tan_theta_h = rr_tan(rr_acos(nh));
    if (__builtin_isinf(tan_theta_h)) return mk(0, 0, 0);
    const double beckmann = beckmann_from_tan(tan_theta_h, ct.alpha2, nh);
    const double hl = rr_fabs(v_dot(h, l));
    const double hv = rr_fabs(v_dot(h, v));
    const double g = rr_min(2.0 * nh * nv / hv, rr_min(2.0 * nh * nl / hv, 1.0));
    double denom = ior_ratio * hv + hl;
    denom = denom * denom;
    const double norm_fac = hv * hl / (nv * nl);
    const V3 f = fresnel_value(ct, h, v, entering);
    V3 r = v_mul(ct.color, v_sub(mk(1.0, 1.0, 1.0), f));
    r = v_scale(r, beckmann);
    r = v_scale(r, g);
    r = v_scale(r, norm_fac);
    r = v_scale(r, ior_ratio);
    r = v_scale(r, ior_ratio);
    return v_div(r, denom);
}

RR_DEV double pdf_beckmann_reflect_value(double alpha2, V3 n, V3 l, V3 v, CtHalf* half = nullptr) {  // :915-941
    if (half) half->valid = false;
    V3 h = v_add(l, v);
    if (v_is_zeros(h)) return 1.0;
    h = v_unit(h);
    const double nh_signed = v_dot(n, h);
    const double nh = rr_fabs(nh_signed);
    const double tan_theta_h = rr_tan(rr_acos(nh));
    if (half) half->valid = true, half->h = h, half->nh = nh_signed, half->tan_abs = tan_theta_h, half->beckmann_abs = 0.0;
    if (__builtin_isinf(tan_theta_h)) return 1.0;
    const double b = beckmann_from_tan(tan_theta_h, alpha2, nh);
    if (half) half->beckmann_abs = b;
    return b;
}

// Pdf::Beckmann.generate (:1006-1020) / MicrofacetDistribution::generate (:1139-1161)
template <bool WITH_VALUE>
RR_DEV V3 beckmann_generate(double alpha2, V3 n, Rng& rng, double& value) {
    V3 e1, e2;
    v_onb(n, e1, e2);
    const double phi = 2.0 * RR_PI * rng.next();
    const double tan2theta = -alpha2 * rr_log(1.0 - rng.next());
    const double costheta = 1.0 / rr_sqrt(1.0 + tan2theta);
    const double sintheta = rr_sqrt(1.0 - costheta * costheta);
    const rr_sincos_t sc_phi = rr_sincos(phi);
    const double sp = sc_phi.s, cp = sc_phi.c;
    const double x = cp * sintheta;
    const double y = sp * sintheta;
    const V3 h = v_add(v_add(v_scale(e1, x), v_scale(e2, y)), v_scale(n, costheta));
    if (WITH_VALUE) {
        const double nh = v_dot(n, h);
        value = rr_exp(-tan2theta / alpha2) / (RR_PI * alpha2 * pow4(nh));
    }
    return h;
}

struct Scatter {
    bool scatter;
    V3 color;
    V3 dir;
};

RR_DEV Scatter no_scatter() { return Scatter{false, mk(0, 0, 0), mk(0, 0, 0)}; }

RR_DEV Scatter ct_evaluate_reflection(const CtLayer& ct, V3 n, V3 h, V3 v, V3 l, double pdf,
                                      const CtHalf* half = nullptr) {  // :721-758
    if (v_dot(h, v) < 0.0) return no_scatter();
    const double nl = v_dot(n, l);
    if (nl < 0.0) return no_scatter();
    const double frac_dwh_dwi = 4.0 * v_dot(h, l);
    V3 color = v_scale(ct_brdf(ct, n, l, v, half), nl);
    color = v_scale(v_div(color, pdf), frac_dwh_dwi);
    if (v_is_zeros(color)) return no_scatter();
    return Scatter{true, color, l};
}

RR_DEV Scatter ct_evaluate_refraction(const CtLayer& ct, V3 n, V3 h, V3 v, V3 l, double pdf, bool entering,
                                      double ior_ratio) {  // :764-812
    if (v_dot(h, v) < 0.0) return no_scatter();
    const double nl = v_dot(n, l);
    if (nl > 0.0) return no_scatter();
    const double hl = rr_fabs(v_dot(h, l));
    const double hv = rr_fabs(v_dot(h, v));
    double denom = ior_ratio * hv + hl;
    denom = denom * denom;
    const double dwh_dwi = hl / denom;
    V3 color = v_div(v_scale(ct_btdf(ct, n, l, v, entering), rr_fabs(nl)), ior_ratio * ior_ratio);
    color = v_div(color, pdf * dwh_dwi);
    if (v_is_zeros(color)) return no_scatter();
    return Scatter{true, color, l};
}

RR_DEV Scatter lambertian_scatter(V3 color_in, V3 n, Rng& rng) {  // :259-281, :982-993, :1233-1243
    V3 e1, e2;
    v_onb(n, e1, e2);
    const double u = rng.next();
    const double phi = 2.0 * RR_PI * rng.next();
    const double su = rr_sqrt(u);
    const rr_sincos_t sc_phi = rr_sincos(phi);
    const double sp = sc_phi.s, cp = sc_phi.c;
    const double x = cp * su;
    const double y = sp * su;
    const double z = rr_sqrt(1.0 - u);
    const V3 l = v_add(v_add(v_scale(e1, x), v_scale(e2, y)), v_scale(n, z));
    const double ndl = v_dot(n, l);
    const V3 brdf = v_scale(color_in, RR_FRAC_1_PI);
    const V3 color = v_div(v_scale(brdf, ndl), ndl * RR_FRAC_1_PI);
    return Scatter{true, color, l};
}

RR_DEV Scatter ct_scatter(const CtLayer& ct, V3 n, V3 v, Rng& rng) {  // :403-424
    double unused;
    const V3 h = beckmann_generate<false>(ct.alpha2, n, rng, unused);
    const V3 l = reflect(h, v);
    CtHalf half;
    const double pdf = pdf_beckmann_reflect_value(ct.alpha2, n, l, v, &half);
    return ct_evaluate_reflection(ct, n, h, v, l, pdf, &half);
}

RR_DEV V3 reflect_brdf(V3 color, V3 n, V3 l) { return v_div(color, rr_fabs(v_dot(n, l))); }  // :1254-1265
RR_DEV V3 refract_btdf(V3 color, V3 n, V3 l, V3 v) {                                          // :1333-1351
    if (v_dot(l, v) > 0.0) return mk(0, 0, 0);
    return v_div(color, rr_fabs(v_dot(n, l)));
}

// Material::evaluate, material.rs:91-109.  `kind` is s->kind, passed apart so that a caller whose wave holds
// one material kind only (local_pool.hip) can hand it over as a wave-uniform value: one arm, scalar branches.
RR_DEV Scatter material_evaluate_kind(int kind, const SurfaceDev* s, V3 n, V3 v, Rng& rng) {
    const V3 color = mk(s->color[0], s->color[1], s->color[2]);
    switch (kind) {
        case RAYRS_MAT_LAMBERTIAN: return lambertian_scatter(color, n, rng);
        case RAYRS_MAT_REFLECT: {  // :283-301 (pdf.value == 1: x / 1. is exact)
            const V3 l = reflect(n, v);
            return Scatter{true, v_scale(reflect_brdf(color, n, l), v_dot(n, l)), l};
        }
        case RAYRS_MAT_REFRACT: {  // :303-337
            const bool entering = v_dot(n, v) > 0.0;
            const V3 nf = dir_normal(entering, n);
            V3 l;
            if (!refract(nf, v, dir_ior_ratio(entering, s->ior), l)) return no_scatter();
            return Scatter{true, v_scale(refract_btdf(color, nf, l, v), rr_fabs(v_dot(nf, l))), l};
        }
        case RAYRS_MAT_GLASS: {  // :339-401
            const double cos_theta = v_dot(n, v);
            const bool entering = cos_theta > 0.0;
            const V3 nf = dir_normal(entering, n);
            const double sin2theta = 1.0 - cos_theta * cos_theta;
            const double ior_ratio = dir_ior_ratio(entering, s->ior);
            bool do_reflect = ior_ratio * ior_ratio * sin2theta >= 1.0;
            if (!do_reflect) {
                const double fresnel =
                    entering ? schlick_scalar(1.0, s->ior, nf, v) : schlick_scalar(s->ior, 1.0, nf, v);
                do_reflect = rng.next() < fresnel;
            }
            if (do_reflect) {
                const V3 l = reflect(nf, v);
                return Scatter{true, v_scale(reflect_brdf(color, nf, l), v_dot(nf, l)), l};
            }
            V3 l;
            if (!refract(nf, v, ior_ratio, l)) return no_scatter();  // .unwrap()
            return Scatter{true, v_scale(refract_btdf(color, nf, l, v), rr_fabs(v_dot(nf, l))), l};
        }
        case RAYRS_MAT_COOK_TORRANCE: return ct_scatter(load_ct(s), n, v, rng);
        case RAYRS_MAT_COOK_TORRANCE_REFRACT: {  // :426-467
            const CtLayer ct = load_ct(s);
            const bool entering = v_dot(n, v) > 0.0;
            const V3 nf = dir_normal(entering, n);
            const double ior_ratio = dir_ior_ratio(entering, s->ior);
            double value;
            V3 h = beckmann_generate<true>(ct.alpha2, nf, rng, value);
            h = dir_normal(entering, h);
            V3 l;
            if (!refract(h, v, ior_ratio, l)) return no_scatter();
            return ct_evaluate_refraction(ct, nf, h, v, l, value, entering, ior_ratio);
        }
        case RAYRS_MAT_COOK_TORRANCE_GLASS: {  // :469-565
            const CtLayer ct = load_ct(s);
            double value;
            V3 h = beckmann_generate<true>(ct.alpha2, n, rng, value);
            const bool entering = v_dot(n, v) > 0.0;
            h = dir_normal(entering, h);
            const V3 nf = dir_normal(entering, n);
            const double cos_theta = v_dot(h, v);
            const double ior_ratio = dir_ior_ratio(entering, s->ior);
            const double sin2_theta = 1.0 - cos_theta * cos_theta;
            if (ior_ratio * ior_ratio * sin2_theta >= 1.0) {
                const V3 l = reflect(h, v);
                return ct_evaluate_reflection(ct, nf, h, v, l, value);
            }
            const double fresnel = entering ? schlick_scalar(1.0, s->ior, h, v) : schlick_scalar(s->ior, 1.0, h, v);
            if (rng.next() < fresnel) {
                const V3 l = reflect(h, v);
                Scatter ev = ct_evaluate_reflection(ct, nf, h, v, l, value);
                if (ev.scatter) ev.color = v_div(ev.color, fresnel);
                return ev;
            }
            V3 l;
            if (!refract(h, v, ior_ratio, l)) return no_scatter();  // .expect()
            Scatter ev = ct_evaluate_refraction(ct, nf, h, v, l, value, entering, ior_ratio);
            if (ev.scatter) ev.color = v_div(ev.color, 1.0 - fresnel);
            return ev;
        }
        case RAYRS_MAT_PLASTIC: {  // :567-593
            const double fresnel = schlick_scalar(1.0, s->ior, n, v);
            if (rng.next() < fresnel) {
                Scatter ev = ct_scatter(load_ct(s), n, v, rng);
                if (ev.scatter) ev.color = v_div(ev.color, fresnel);
                return ev;
            }
            return lambertian_scatter(color, n, rng);
        }
        default: return no_scatter();  // NoReflect
    }
}

RR_DEV Scatter material_evaluate(const SurfaceDev* s, V3 n, V3 v, Rng& rng) {
    return material_evaluate_kind(s->kind, s, n, v, rng);
}

// --------------------------------------------------------------- background

RR_DEV uint32_t f64_as_index(double x) {  // Rust `as usize`: NaN and negatives -> 0, saturating
    if (!(x > 0.0)) return 0u;
    if (x >= 4294967295.0) return 0xffffffffu;
    return (uint32_t)x;
}

// The four texels of a lookup at (i, j): one 64-byte record (scene_host.cpp).  Image::pixel would panic
// out of range (image.rs:183-186); that only happens for phi == 2*pi or theta == pi, where the weights
// of the out-of-range texels are zero: the record holds the clamped neighbours.
struct HdriQuad {
    V3 f0, f1, f2, f3;  // (i, j), (i+1, j), (i, j+1), (i+1, j+1)
};
RR_DEV HdriQuad hdri_quad(const SceneDev& sc, uint32_t i, uint32_t j) {
    if (i >= sc.hdri_h) i = sc.hdri_h - 1;
    if (j >= sc.hdri_w) j = sc.hdri_w - 1;
    const float4* q = reinterpret_cast<const float4*>(sc.hdri) + ((size_t)i * sc.hdri_w + j) * 4;
    const float4 a = q[0], b = q[1], c = q[2], d = q[3];
    HdriQuad r;
    r.f0 = mk((double)a.x, (double)a.y, (double)a.z);
    r.f1 = mk((double)b.x, (double)b.y, (double)b.z);
    r.f2 = mk((double)c.x, (double)c.y, (double)c.z);
    r.f3 = mk((double)d.x, (double)d.y, (double)d.z);
    return r;
}

// Scene::background, lib.rs:254-285
RR_DEV V3 background(const SceneDev& sc, V3 dir) {
    dir = v_unit(dir);
    const double phi = rr_atan2(dir.z, dir.x) + RR_PI;
    const double theta = rr_acos(dir.y);
    const double x = phi / (2.0 * RR_PI) * sc.hdri_wm1;
    const double y = theta / RR_PI * sc.hdri_hm1;
    const double x_f = rr_floor(x), x_c = rr_ceil(x), y_f = rr_floor(y), y_c = rr_ceil(y);
    const uint32_t i = f64_as_index(y_f);
    const uint32_t j = f64_as_index(x_f);
    const HdriQuad q = hdri_quad(sc, i, j);
    const V3 a = v_scale(v_scale(q.f0, x_c - x), y_c - y);
    const V3 b = v_scale(v_scale(q.f1, x_c - x), y - y_f);
    const V3 c = v_scale(v_scale(q.f2, x - x_f), y_c - y);
    const V3 e = v_scale(v_scale(q.f3, x - x_f), y - y_f);
    return v_add(v_add(v_add(a, b), c), e);
}

// Camera::generate_primary_ray, lib.rs:202-210
RR_DEV void primary_ray(const CameraDev& cam, uint32_t i, uint32_t j, Rng& rng, V3& o, V3& d) {
    const double fi = (double)i, fj = (double)j;
    const double x = (fj + rng.next()) / cam.ppc - cam.width / 2.0;
    const double y = (fi + rng.next()) / cam.ppc - cam.height / 2.0;
    o = mk(cam.origin[0], cam.origin[1], cam.origin[2]);
    const V3 ex = mk(cam.e_x[0], cam.e_x[1], cam.e_x[2]);
    const V3 ey = mk(cam.e_y[0], cam.e_y[1], cam.e_y[2]);
    d = v_add(v_add(mk(cam.z[0], cam.z[1], cam.z[2]), v_scale(ex, x)), v_scale(ey, y));
}

}  // namespace rayrs

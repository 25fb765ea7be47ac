/*
 * rayrs_oracle.c -- CPU oracle: plain-C f64 restatement of the rayrs CPU path.
 *
 * TEST INFRASTRUCTURE ONLY (see rayrs_oracle.h).  Every function cites the
 * reference lines it restates; citations are relative to /root/reference.
 * Expression order follows the Rust source operator by operator (Rust never
 * contracts a*b+c, so this file must be compiled with -ffp-contract=off).
 *
 * PARITY STATUS: intersection/vecmath/bbox behaviour is pinned by the
 * reference's own unit tests (re-expressed in tests/); radiance and the
 * materials are "parity unpinned": the reference has no numeric test or
 * deterministic output for them and cannot be built here (Rust, no rustc).
 *
 * Deviation that is deliberate and build-defined: rand::random::<f64>() is
 * replaced by the counter RNG of include/rayrs_numeric.h, and the elementary
 * functions are that header's portable ones unless orc_set_math_mode(1).
 */
#define _GNU_SOURCE
#include "rayrs_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/rayrs_numeric.h"

/* ------------------------------------------------------------ math mode */

static int g_libm = 0;
void orc_set_math_mode(int libm) { g_libm = libm ? 1 : 0; }
int orc_get_math_mode(void) { return g_libm; }

static inline double m_sin(double x) { return g_libm ? sin(x) : rr_sin(x); }
static inline double m_cos(double x) { return g_libm ? cos(x) : rr_cos(x); }
static inline double m_tan(double x) { return g_libm ? tan(x) : rr_tan(x); }
static inline double m_log(double x) { return g_libm ? log(x) : rr_log(x); }
static inline double m_exp(double x) { return g_libm ? exp(x) : rr_exp(x); }
static inline double m_acos(double x) { return g_libm ? acos(x) : rr_acos(x); }
static inline double m_atan2(double y, double x) { return g_libm ? atan2(y, x) : rr_atan2(y, x); }

double orc_math(int fn, double x, double y) {
    switch (fn) {
        case 0: return m_sin(x);
        case 1: return m_cos(x);
        case 2: return m_tan(x);
        case 3: return m_log(x);
        case 4: return m_exp(x);
        case 5: return m_acos(x);
        case 6: return m_atan2(x, y);
        case 7: return rr_sqrt(x);
        default: return 0.0;
    }
}

uint64_t orc_rng_bits(uint64_t seed, uint64_t pixel, uint64_t sample, uint32_t draw) {
    return rr_draw_bits(rr_path_key(seed, pixel, sample), draw);
}

/* ------------------------------------------------- vecmath.rs:513-806 */

typedef struct {
    double x, y, z;
} v3;

static inline v3 V(double x, double y, double z) {
    v3 r = {x, y, z};
    return r;
}
static inline v3 v_from(const double* p) { return V(p[0], p[1], p[2]); }
static inline void v_to(v3 a, double* p) {
    p[0] = a.x;
    p[1] = a.y;
    p[2] = a.z;
}
static inline v3 v_add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); } /* :716-728 */
static inline v3 v_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); } /* :762-774 */
static inline v3 v_mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); } /* :606-618 */
static inline v3 v_scale(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); } /* :620-638 */
/* Div<f64> multiplies by the reciprocal, :690-698 */
static inline v3 v_div(v3 a, double s) {
    double inv = 1.0 / s;
    return v_scale(a, inv);
}
/* DivAssign<f64> truly divides, :708-714 */
static inline v3 v_div_assign(v3 a, double s) { return V(a.x / s, a.y / s, a.z / s); }
static inline double v_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* :539-545 */
static inline v3 v_cross(v3 a, v3 b) {                                                 /* :565-577 */
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline double v_mag2(v3 a) { return v_dot(a, a); }          /* :521-523 */
static inline double v_mag(v3 a) { return rr_sqrt(v_mag2(a)); }    /* :517-519 */
static inline v3 v_unit(v3 a) { return v_div(a, v_mag(a)); }       /* :525-527 */
static inline int v_is_zeros(v3 a) { return a.x == 0.0 && a.y == 0.0 && a.z == 0.0; } /* :299-301 */

/* Vec3::orthonormal_basis, vecmath.rs:341-352 */
static inline void v_onb(v3 n, v3* e1, v3* e2) {
    if (rr_fabs(n.x) > rr_fabs(n.y))
        *e1 = v_unit(V(n.z, 0.0, -n.x));
    else
        *e1 = v_unit(V(0.0, n.z, -n.y));
    *e2 = v_unit(v_cross(n, *e1));
}

/* f64::powi(4) / powi(5): LLVM expands a constant powi by binary
 * decomposition: x^4 = (x*x)*(x*x), x^5 = x*((x*x)*(x*x)). */
static inline double pow4(double x) {
    double x2 = x * x;
    return x2 * x2;
}
static inline double pow5(double x) {
    double x2 = x * x;
    return x * (x2 * x2);
}

/* The helpers above through one exported entry point, for the reference's own vector tests
 * (vecmath.rs:812-893, doc tests :339 and geometry.rs:672) re-expressed in
 * tests/test_oracle_reference_tests.py.  op: 0 a+b  1 a-b  2 a*b (componentwise)  3 a*s  4 cross(a,b)
 * 5 powf(a, s) (f64::powf is the platform pow, vecmath.rs:368-376)  6 clip(a, s, t) = min(max).max(min),
 * :389-397  7 unit(a)  8 a / s (Div<f64>: multiply by the reciprocal, :690-698)
 * 9 a /= s (DivAssign<f64>: true division, :708-714).  Scalars: orc_vec_scalar, op 0 dot(a,b)  1 mag2(a). */
void orc_vec_op(int op, const double a[3], const double b[3], double s, double t, double out[3]) {
    v3 x = v_from(a), y = b ? v_from(b) : V(0, 0, 0), r = V(0, 0, 0);
    switch (op) {
        case 0: r = v_add(x, y); break;
        case 1: r = v_sub(x, y); break;
        case 2: r = v_mul(x, y); break;
        case 3: r = v_scale(x, s); break;
        case 4: r = v_cross(x, y); break;
        case 5: r = V(pow(x.x, s), pow(x.y, s), pow(x.z, s)); break;
        case 6: r = V(rr_max(rr_min(x.x, t), s), rr_max(rr_min(x.y, t), s), rr_max(rr_min(x.z, t), s)); break;
        case 7: r = v_unit(x); break;
        case 8: r = v_div(x, s); break;
        case 9: r = v_div_assign(x, s); break;
        default: break;
    }
    v_to(r, out);
}
double orc_vec_scalar(int op, const double a[3], const double b[3]) {
    return op == 0 ? v_dot(v_from(a), v_from(b)) : v_mag2(v_from(a));
}
void orc_orthonormal_basis(const double n[3], double e1[3], double e2[3]) { /* vecmath.rs:341-352 */
    v3 a, b;
    v_onb(v_from(n), &a, &b);
    v_to(a, e1);
    v_to(b, e2);
}

/* ------------------------------------------------------ RNG draw stream */

typedef struct {
    uint64_t key;
    uint32_t draw;
} rng_t;

static inline double rng_next(rng_t* r) { return rr_uniform(r->key, r->draw++); }

/* ----------------------------------------------------------- ray, lib.rs */

typedef struct {
    v3 o, d;
} ray_t;

static inline v3 ray_point(ray_t r, double t) { return v_add(r.o, v_scale(r.d, t)); } /* lib.rs:41-43 */

/* --------------------------------------------------- shapes, geometry.rs */

typedef struct {
    int kind;
    /* sphere, geometry.rs:78-81 */
    double radius2;
    v3 origin;
    /* plane, :176-181 */
    int axis;
    double u0, u1, v0, v1, pos;
    /* triangle, :312-322 */
    v3 p1, p2, p3, e1, e2, normal;
    double area;
} shape_t;

typedef struct {
    double xmin, xmax, ymin, ymax, zmin, zmax;
} aabb_t;

/* AxisAlignedBoundingBox::intersect, geometry.rs:458-513 */
static int aabb_intersect(const aabb_t* b, ray_t ray, double tmin, double tmax) {
    double xmax = b->xmax - ray.o.x;
    double xmin = b->xmin - ray.o.x;
    double inv_x = 1.0 / ray.d.x;
    double t0, t1;
    if (inv_x < 0.0) {
        t0 = xmax * inv_x;
        t1 = xmin * inv_x;
    } else {
        t0 = xmin * inv_x;
        t1 = xmax * inv_x;
    }
    tmin = rr_max(tmin, t0);
    tmax = rr_min(tmax, t1);
    if (tmax <= tmin) return 0;

    double ymax = b->ymax - ray.o.y;
    double ymin = b->ymin - ray.o.y;
    double inv_y = 1.0 / ray.d.y;
    if (inv_y < 0.0) {
        t0 = ymax * inv_y;
        t1 = ymin * inv_y;
    } else {
        t0 = ymin * inv_y;
        t1 = ymax * inv_y;
    }
    tmin = rr_max(tmin, t0);
    tmax = rr_min(tmax, t1);
    if (tmax <= tmin) return 0;

    double zmax = b->zmax - ray.o.z;
    double zmin = b->zmin - ray.o.z;
    double inv_z = 1.0 / ray.d.z;
    if (inv_z < 0.0) {
        t0 = zmax * inv_z;
        t1 = zmin * inv_z;
    } else {
        t0 = zmin * inv_z;
        t1 = zmax * inv_z;
    }
    tmin = rr_max(tmin, t0);
    tmax = rr_min(tmax, t1);
    if (tmax <= tmin) return 0;
    return 1;
}

/* Same slab arithmetic, but also returns the entry parameter so that the
 * ordered traversal can sort children and cull by the closest hit.  The
 * boolean is identical to aabb_intersect (tmin only grows, tmax only
 * shrinks, so one final compare equals the three early-outs). */
static int aabb_intersect_entry(const double* bx, ray_t ray, v3 inv, double tmin, double tmax, double* entry) {
    double lo, hi, t0, t1;
    lo = bx[0] - ray.o.x;
    hi = bx[1] - ray.o.x;
    if (inv.x < 0.0) { t0 = hi * inv.x; t1 = lo * inv.x; } else { t0 = lo * inv.x; t1 = hi * inv.x; }
    tmin = rr_max(tmin, t0);
    tmax = rr_min(tmax, t1);
    lo = bx[2] - ray.o.y;
    hi = bx[3] - ray.o.y;
    if (inv.y < 0.0) { t0 = hi * inv.y; t1 = lo * inv.y; } else { t0 = lo * inv.y; t1 = hi * inv.y; }
    tmin = rr_max(tmin, t0);
    tmax = rr_min(tmax, t1);
    lo = bx[4] - ray.o.z;
    hi = bx[5] - ray.o.z;
    if (inv.z < 0.0) { t0 = hi * inv.z; t1 = lo * inv.z; } else { t0 = lo * inv.z; t1 = hi * inv.z; }
    tmin = rr_max(tmin, t0);
    tmax = rr_min(tmax, t1);
    *entry = tmin;
    return !(tmax <= tmin);
}

/* Sphere::intersect, geometry.rs:106-132 */
static int sphere_intersect(const shape_t* s, ray_t ray, double* t) {
    v3 odiff = v_sub(ray.o, s->origin);
    double a = v_mag2(ray.d);
    double b = 2.0 * v_dot(ray.d, odiff);
    double c = v_mag2(odiff) - s->radius2;
    double desc = b * b - 4.0 * a * c;
    if (desc > 0.0) {
        double sq = rr_sqrt(desc);
        double t1 = (-b - sq) / (2.0 * a);
        double t2 = (-b + sq) / (2.0 * a);
        if (t1 < 0.0) {
            if (t2 < 0.0) return 0;
            *t = t2;
            return 1;
        }
        *t = t1;
        return 1;
    }
    return 0;
}

/* Range::contains: start <= x < end */
static inline int range_contains(double start, double end, double x) { return start <= x && x < end; }

/* Plane::intersect, geometry.rs:229-271 */
static int plane_intersect(const shape_t* s, ray_t ray, double* t) {
    switch (s->axis) {
        case ORC_AXIS_X:
        case ORC_AXIS_XREV:
            if (ray.d.x != 0.0) {
                double tt = (s->pos - ray.o.x) / ray.d.x;
                v3 p = ray_point(ray, tt);
                if (range_contains(s->u0, s->u1, p.y) && range_contains(s->v0, s->v1, p.z)) {
                    *t = tt;
                    return 1;
                }
            }
            return 0;
        case ORC_AXIS_Y:
        case ORC_AXIS_YREV:
            if (ray.d.y != 0.0) {
                double tt = (s->pos - ray.o.y) / ray.d.y;
                v3 p = ray_point(ray, tt);
                if (range_contains(s->u0, s->u1, p.x) && range_contains(s->v0, s->v1, p.z)) {
                    *t = tt;
                    return 1;
                }
            }
            return 0;
        default:
            if (ray.d.z != 0.0) {
                double tt = (s->pos - ray.o.z) / ray.d.z;
                v3 p = ray_point(ray, tt);
                if (range_contains(s->u0, s->u1, p.x) && range_contains(s->v0, s->v1, p.y)) {
                    *t = tt;
                    return 1;
                }
            }
            return 0;
    }
}

/* Triangle::new, geometry.rs:341-354 */
static void triangle_init(shape_t* s, v3 p1, v3 p2, v3 p3) {
    memset(s, 0, sizeof(*s));
    s->kind = ORC_SHAPE_TRIANGLE;
    s->p1 = p1;
    s->p2 = p2;
    s->p3 = p3;
    s->e1 = v_sub(p2, p1);
    s->e2 = v_sub(p3, p1);
    v3 n = v_cross(s->e1, s->e2);
    s->normal = v_unit(n);
    s->area = v_mag(n) / 2.0;
}

/* Triangle::intersect (Moller-Trumbore), geometry.rs:359-375 */
static int triangle_intersect(const shape_t* s, ray_t ray, double* t) {
    v3 tt = v_sub(ray.o, s->p1);
    v3 p = v_cross(ray.d, s->e2);
    v3 q = v_cross(tt, s->e1);
    double den = v_dot(p, s->e1);
    double d = v_dot(q, s->e2) / den;
    double u = v_dot(p, tt) / den;
    double v = v_dot(q, ray.d) / den;
    if (d < 0.0 || u < 0.0 || v < 0.0 || u + v > 1.0) return 0;
    *t = d;
    return 1;
}

static int shape_intersect(const shape_t* s, ray_t ray, double* t) {
    switch (s->kind) {
        case ORC_SHAPE_SPHERE: return sphere_intersect(s, ray, t);
        case ORC_SHAPE_PLANE: return plane_intersect(s, ray, t);
        default: return triangle_intersect(s, ray, t);
    }
}

/* Hittable::normal: geometry.rs:134-136, :273-282, :377-379 */
static v3 shape_normal(const shape_t* s, v3 p) {
    switch (s->kind) {
        case ORC_SHAPE_SPHERE: return v_unit(v_sub(p, s->origin));
        case ORC_SHAPE_PLANE:
            switch (s->axis) {
                case ORC_AXIS_X: return V(1.0, 0.0, 0.0);
                case ORC_AXIS_XREV: return V(-1.0, 0.0, 0.0);
                case ORC_AXIS_Y: return V(0.0, 1.0, 0.0);
                case ORC_AXIS_YREV: return V(0.0, -1.0, 0.0);
                case ORC_AXIS_Z: return V(0.0, 0.0, 1.0);
                default: return V(0.0, 0.0, -1.0);
            }
        default: return s->normal;
    }
}

/* From<&Sphere/&Plane/&Triangle> for AxisAlignedBoundingBox, geometry.rs:686-733 */
static aabb_t shape_bbox(const shape_t* s) {
    aabb_t b;
    switch (s->kind) {
        case ORC_SHAPE_SPHERE: {
            double radius = rr_sqrt(s->radius2);
            b.xmin = s->origin.x - radius;
            b.xmax = s->origin.x + radius;
            b.ymin = s->origin.y - radius;
            b.ymax = s->origin.y + radius;
            b.zmin = s->origin.z - radius;
            b.zmax = s->origin.z + radius;
            return b;
        }
        case ORC_SHAPE_PLANE:
            switch (s->axis) {
                case ORC_AXIS_X:
                case ORC_AXIS_XREV:
                    b.xmin = s->pos; b.xmax = s->pos;
                    b.ymin = s->u0; b.ymax = s->u1;
                    b.zmin = s->v0; b.zmax = s->v1;
                    return b;
                case ORC_AXIS_Y:
                case ORC_AXIS_YREV:
                    b.xmin = s->u0; b.xmax = s->u1;
                    b.ymin = s->pos; b.ymax = s->pos;
                    b.zmin = s->v0; b.zmax = s->v1;
                    return b;
                default:
                    b.xmin = s->u0; b.xmax = s->u1;
                    b.ymin = s->v0; b.ymax = s->v1;
                    b.zmin = s->pos; b.zmax = s->pos;
                    return b;
            }
        default:
            b.xmin = rr_min(s->p1.x, rr_min(s->p2.x, s->p3.x));
            b.ymin = rr_min(s->p1.y, rr_min(s->p2.y, s->p3.y));
            b.zmin = rr_min(s->p1.z, rr_min(s->p2.z, s->p3.z));
            b.xmax = rr_max(s->p1.x, rr_max(s->p2.x, s->p3.x));
            b.ymax = rr_max(s->p1.y, rr_max(s->p2.y, s->p3.y));
            b.zmax = rr_max(s->p1.z, rr_max(s->p2.z, s->p3.z));
            return b;
    }
}

/* AxisAlignedBoundingBox::expand, geometry.rs:674-683 */
static inline aabb_t aabb_expand(aabb_t a, aabb_t o) {
    aabb_t r;
    r.xmin = rr_min(a.xmin, o.xmin);
    r.xmax = rr_max(a.xmax, o.xmax);
    r.ymin = rr_min(a.ymin, o.ymin);
    r.ymax = rr_max(a.ymax, o.ymax);
    r.zmin = rr_min(a.zmin, o.zmin);
    r.zmax = rr_max(a.zmax, o.zmax);
    return r;
}
void orc_aabb_expand(const double a[6], const double b[6], double out[6]) {
    aabb_t x = {a[0], a[1], a[2], a[3], a[4], a[5]}, y = {b[0], b[1], b[2], b[3], b[4], b[5]};
    aabb_t r = aabb_expand(x, y);
    out[0] = r.xmin, out[1] = r.xmax, out[2] = r.ymin, out[3] = r.ymax, out[4] = r.zmin, out[5] = r.zmax;
}
/* center, geometry.rs:577-582 */
static inline v3 aabb_center(aabb_t b) {
    return V((b.xmax - b.xmin) / 2.0 + b.xmin, (b.ymax - b.ymin) / 2.0 + b.ymin, (b.zmax - b.zmin) / 2.0 + b.zmin);
}
/* volume, :609-613 */
static inline double aabb_volume(aabb_t b) { return (b.xmax - b.xmin) * (b.ymax - b.ymin) * (b.zmax - b.zmin); }
/* surface_area, :640-645 */
static inline double aabb_surface_area(aabb_t b) {
    double x = b.xmax - b.xmin;
    double y = b.ymax - b.ymin;
    double z = b.zmax - b.zmin;
    return 2.0 * x * y + 2.0 * y * z + 2.0 * x * z;
}

/* ------------------------------------------------- materials, material.rs */

typedef struct {
    double alpha2;
    int metallic;
    double ior; /* SchlickDielectric(ior) */
    v3 r0;      /* SchlickMetallic(r0) */
    v3 color;
} ct_t;

typedef struct {
    int kind;
    v3 color;  /* Lambertian / Reflect / Refract / Glass colour */
    double ior;
    ct_t ct;   /* CookTorrance layer of kinds 4..7 */
} mat_t;

typedef struct {
    int emissive;
    double strength;
    v3 color;
} emis_t;

static int in01(const double* c) {
    return c[0] >= 0.0 && c[0] <= 1.0 && c[1] >= 0.0 && c[1] <= 1.0 && c[2] >= 0.0 && c[2] <= 1.0;
}

/* ctor checks: material.rs:608-611, :629-632, :650-655, :673-683, :705-715,
 * :832-843, :863-874, :887-900 */
static int mat_from_desc(const orc_material* d, mat_t* m) {
    memset(m, 0, sizeof(*m));
    m->kind = d->kind;
    if (d->kind == ORC_MAT_NO_REFLECT) return 0;
    if (d->kind < 0 || d->kind > ORC_MAT_NO_REFLECT) return -1;
    if (!in01(d->color)) return -1;
    m->color = v_from(d->color);
    int needs_ior = d->kind == ORC_MAT_REFRACT || d->kind == ORC_MAT_GLASS || d->kind == ORC_MAT_COOK_TORRANCE_REFRACT ||
                    d->kind == ORC_MAT_COOK_TORRANCE_GLASS || d->kind == ORC_MAT_PLASTIC;
    if (needs_ior && !(d->ior > 0.0 && isfinite(d->ior))) return -1;
    m->ior = d->ior;
    if (d->kind >= ORC_MAT_COOK_TORRANCE && d->kind <= ORC_MAT_PLASTIC) {
        if (!(d->alpha > 0.0 && isfinite(d->alpha))) return -1;
        m->ct.alpha2 = d->alpha * d->alpha;
        m->ct.color = m->color;
        if (d->kind == ORC_MAT_COOK_TORRANCE) {
            m->ct.metallic = d->metallic ? 1 : 0;
            if (m->ct.metallic)
                m->ct.r0 = v_from(d->r0);
            else
                m->ct.ior = d->ior;
        } else {
            m->ct.metallic = 0;
            m->ct.ior = d->ior;
            if (d->kind == ORC_MAT_PLASTIC) {
                if (!in01(d->spec_color)) return -1;
                m->ct.color = v_from(d->spec_color);
            }
        }
    }
    return 0;
}

/* Emission::new, material.rs:1067-1074 */
static int emis_from_desc(const orc_emission* d, emis_t* e) {
    memset(e, 0, sizeof(*e));
    if (!d || !d->emissive) return 0;
    if (!(d->strength >= 0.0) || !in01(d->color)) return -1;
    e->emissive = 1;
    e->strength = d->strength;
    e->color = v_from(d->color);
    return 0;
}

/* Emitting::emit, material.rs:1077-1084 */
static inline v3 emis_emit(const emis_t* e) {
    if (e->emissive) return V(e->strength * e->color.x, e->strength * e->color.y, e->strength * e->color.z);
    return V(0.0, 0.0, 0.0);
}

/* ScatteringDirection, material.rs:1191-1231 */
static inline double dir_ior_ratio(int entering, double ior) { return entering ? 1.0 / ior : ior; }
static inline v3 dir_normal(int entering, v3 n) { return entering ? n : v_scale(n, -1.0); }

/* schlick_scalar, material.rs:1472-1479 */
static inline double schlick_scalar(double ior_curr, double ior_new, v3 n, v3 v) {
    double r0 = (ior_curr - ior_new) / (ior_curr + ior_new);
    r0 = r0 * r0;
    return r0 + (1.0 - r0) * pow5(1.0 - v_dot(n, v));
}
/* schlick_vec, material.rs:1484-1489 */
static inline v3 schlick_vec(v3 r0, v3 n, v3 v) {
    double p = pow5(1.0 - v_dot(n, v));
    return v_add(r0, v_scale(v_sub(V(1.0, 1.0, 1.0), r0), p));
}
/* reflect, material.rs:1492-1496 */
static inline v3 reflect(v3 n, v3 v) { return v_sub(v_scale(n, 2.0 * v_dot(v, n)), v); }
/* refract, material.rs:1502-1518; returns 0 for None */
static inline int refract(v3 n, v3 v, double ior_ratio, v3* out) {
    double cos_theta = v_dot(v, n);
    double sin_theta = rr_sqrt(1.0 - cos_theta * cos_theta);
    if (ior_ratio * sin_theta > 1.0) return 0;
    v3 par = v_scale(v_sub(v_scale(n, cos_theta), v), ior_ratio);
    v3 perp = v_scale(n, -rr_sqrt(1.0 - v_mag2(par)));
    *out = v_add(perp, par);
    return 1;
}

/* Fresnel::value, material.rs:1457-1469 */
static inline v3 fresnel_value(const ct_t* ct, v3 n, v3 v, int entering) {
    if (!ct->metallic) {
        double f = entering ? schlick_scalar(1.0, ct->ior, n, v) : schlick_scalar(ct->ior, 1.0, n, v);
        return V(f, f, f);
    }
    return schlick_vec(ct->r0, n, v);
}

/* Beckmann D as written at material.rs:940, :1310-1311, :1414-1415 */
static inline double beckmann_from_tan(double tan_theta_h, double alpha2, double nh) {
    return m_exp(-tan_theta_h * tan_theta_h / alpha2) / (RR_PI * alpha2 * pow4(nh));
}

/* Brdf for CookTorrance, material.rs:1276-1322 */
static v3 ct_brdf(const ct_t* ct, v3 n, v3 l, v3 v) {
    double nv = rr_fabs(v_dot(n, v));
    double nl = rr_fabs(v_dot(n, l));
    v3 h = v_add(v, l);
    if (nv == 0.0 || nl == 0.0) return V(0, 0, 0);
    if (v_is_zeros(h)) return V(0, 0, 0);
    h = v_unit(h);
    double nh = v_dot(n, h);
    double theta_h = m_acos(nh);
    double tan_theta_h = m_tan(theta_h);
    if (isinf(tan_theta_h)) return V(0, 0, 0);
    double beckmann = beckmann_from_tan(tan_theta_h, ct->alpha2, nh);
    double hv = v_dot(h, v);
    double g = rr_min(2.0 * nh * nv / hv, rr_min(2.0 * nh * nl / hv, 1.0));
    v3 f = fresnel_value(ct, h, v, 1);
    v3 r = v_mul(ct->color, f);
    r = v_scale(r, beckmann);
    r = v_scale(r, g);
    return v_div(r, 4.0 * nv * nl);
}

/* Btdf for CookTorrance, material.rs:1362-1442 */
static v3 ct_btdf(const ct_t* ct, v3 n, v3 l, v3 v, int entering) {
    double nv = rr_fabs(v_dot(n, v));
    double nl = rr_fabs(v_dot(n, l));
    double ior_ratio = dir_ior_ratio(entering, ct->ior);
    v3 h;
    if (ior_ratio > 1.0)
        h = v_add(l, v_scale(v, ior_ratio));
    else
        h = v_sub(v_scale(v, -ior_ratio), l);
    if (nv == 0.0 || nl == 0.0) return V(0, 0, 0);
    if (v_is_zeros(h)) return V(0, 0, 0);
    h = v_unit(h);
    double nh = v_dot(n, h);
    double theta_h = m_acos(nh);
    double tan_theta_h = m_tan(theta_h);
    if (isinf(tan_theta_h)) return V(0, 0, 0);
    double beckmann = beckmann_from_tan(tan_theta_h, ct->alpha2, nh);
    double hl = rr_fabs(v_dot(h, l));
    double hv = rr_fabs(v_dot(h, v));
    double g = rr_min(2.0 * nh * nv / hv, rr_min(2.0 * nh * nl / hv, 1.0));
    double denom = ior_ratio * hv + hl;
    denom = denom * denom;
    double norm_fac = hv * hl / (nv * nl);
    v3 f = fresnel_value(ct, h, v, entering);
    v3 r = v_mul(ct->color, v_sub(V(1.0, 1.0, 1.0), f));
    r = v_scale(r, beckmann);
    r = v_scale(r, g);
    r = v_scale(r, norm_fac);
    r = v_scale(r, ior_ratio);
    r = v_scale(r, ior_ratio);
    return v_div(r, denom);
}

/* Pdf::Beckmann(alpha2, Reflect).value, material.rs:915-941 */
static double pdf_beckmann_reflect_value(double alpha2, v3 n, v3 l, v3 v) {
    v3 h = v_add(l, v);
    if (v_is_zeros(h)) return 1.0;
    h = v_unit(h);
    double nh = rr_fabs(v_dot(n, h));
    double theta_h = m_acos(nh);
    double tan_theta_h = m_tan(theta_h);
    if (isinf(tan_theta_h)) return 1.0;
    return beckmann_from_tan(tan_theta_h, alpha2, nh);
}

/* Beckmann half-vector sampling shared by Pdf::Beckmann.generate
 * (material.rs:1006-1020) and MicrofacetDistribution::generate (:1139-1161):
 * draws phi first, then u.  *value (if not NULL) is the :1157 expression. */
static v3 beckmann_generate(double alpha2, v3 n, rng_t* rng, double* value) {
    v3 e1, e2;
    v_onb(n, &e1, &e2);
    double phi = 2.0 * RR_PI * rng_next(rng);
    double tan2theta = -alpha2 * m_log(1.0 - rng_next(rng));
    double costheta = 1.0 / rr_sqrt(1.0 + tan2theta);
    double sintheta = rr_sqrt(1.0 - costheta * costheta);
    double x = m_cos(phi) * sintheta;
    double y = m_sin(phi) * sintheta;
    v3 h = v_add(v_add(v_scale(e1, x), v_scale(e2, y)), v_scale(n, costheta));
    if (value) {
        double nh = v_dot(n, h);
        *value = m_exp(-tan2theta / alpha2) / (RR_PI * alpha2 * pow4(nh));
    }
    return h;
}

typedef struct {
    int scatter;
    v3 color;
    v3 dir;
} scat_t;

static inline scat_t no_scatter(void) {
    scat_t s;
    memset(&s, 0, sizeof(s));
    return s;
}
static inline scat_t do_scatter(v3 color, v3 dir) {
    scat_t s;
    s.scatter = 1;
    s.color = color;
    s.dir = dir;
    return s;
}

/* CookTorrance::evaluate_reflection, material.rs:721-758 */
static scat_t ct_evaluate_reflection(const ct_t* ct, v3 n, v3 h, v3 v, v3 l, double pdf) {
    if (v_dot(h, v) < 0.0) return no_scatter();
    double nl = v_dot(n, l);
    if (nl < 0.0) return no_scatter();
    double frac_dwh_dwi = 4.0 * v_dot(h, l);
    v3 color = v_scale(ct_brdf(ct, n, l, v), nl);
    color = v_scale(v_div(color, pdf), frac_dwh_dwi);
    if (v_is_zeros(color)) return no_scatter();
    return do_scatter(color, l);
}

/* CookTorrance::evaluate_refraction, material.rs:764-812 */
static scat_t ct_evaluate_refraction(const ct_t* ct, v3 n, v3 h, v3 v, v3 l, double pdf, int entering,
                                     double ior_ratio) {
    if (v_dot(h, v) < 0.0) return no_scatter();
    double nl = v_dot(n, l);
    if (nl > 0.0) return no_scatter();
    double hl = rr_fabs(v_dot(h, l));
    double hv = rr_fabs(v_dot(h, v));
    double denom = ior_ratio * hv + hl;
    denom = denom * denom;
    double dwh_dwi = hl / denom;
    v3 color = v_div(v_scale(ct_btdf(ct, n, l, v, entering), rr_fabs(nl)), ior_ratio * ior_ratio);
    color = v_div(color, pdf * dwh_dwi);
    if (v_is_zeros(color)) return no_scatter();
    return do_scatter(color, l);
}

/* Bsdf for LambertianDiffuse, material.rs:259-281 with Pdf::Cosine
 * (:913, :982-993) and the brdf at :1233-1243 */
static scat_t lambertian_scatter(v3 color_in, v3 n, rng_t* rng) {
    v3 e1, e2;
    v_onb(n, &e1, &e2);
    double u = rng_next(rng);
    double phi = 2.0 * RR_PI * rng_next(rng);
    double su = rr_sqrt(u);
    double x = m_cos(phi) * su;
    double y = m_sin(phi) * su;
    double z = rr_sqrt(1.0 - u);
    v3 l = v_add(v_add(v_scale(e1, x), v_scale(e2, y)), v_scale(n, z));
    double ndl = v_dot(n, l);
    v3 brdf = v_scale(color_in, RR_FRAC_1_PI);
    v3 color = v_div(v_scale(brdf, ndl), ndl * RR_FRAC_1_PI);
    return do_scatter(color, l);
}

/* Bsdf for CookTorrance, material.rs:403-424 */
static scat_t ct_scatter(const ct_t* ct, v3 n, v3 v, rng_t* rng) {
    v3 h = beckmann_generate(ct->alpha2, n, rng, NULL);
    v3 l = reflect(h, v);
    return ct_evaluate_reflection(ct, n, h, v, l, pdf_beckmann_reflect_value(ct->alpha2, n, l, v));
}

/* Reflect brdf, material.rs:1254-1265 */
static inline v3 reflect_brdf(v3 color, v3 n, v3 l) { return v_div(color, rr_fabs(v_dot(n, l))); }
/* Refract btdf, material.rs:1333-1351 */
static inline v3 refract_btdf(v3 color, v3 n, v3 l, v3 v) {
    if (v_dot(l, v) > 0.0) return V(0, 0, 0);
    return v_div(color, rr_fabs(v_dot(n, l)));
}

/* Material::evaluate, material.rs:91-109 and the Bsdf impls :259-593 */
static scat_t mat_evaluate(const mat_t* m, v3 n, v3 v, rng_t* rng) {
    switch (m->kind) {
        case ORC_MAT_LAMBERTIAN: return lambertian_scatter(m->color, n, rng);
        case ORC_MAT_REFLECT: { /* :283-301; pdf.value == 1 */
            v3 l = reflect(n, v);
            v3 color = v_div(v_scale(reflect_brdf(m->color, n, l), v_dot(n, l)), 1.0);
            return do_scatter(color, l);
        }
        case ORC_MAT_REFRACT: { /* :303-337 */
            double cos_theta = v_dot(n, v);
            int entering = cos_theta > 0.0;
            v3 nf = dir_normal(entering, n);
            v3 l;
            if (!refract(nf, v, dir_ior_ratio(entering, m->ior), &l)) return no_scatter();
            v3 color = v_div(v_scale(refract_btdf(m->color, nf, l, v), rr_fabs(v_dot(nf, l))), 1.0);
            return do_scatter(color, l);
        }
        case ORC_MAT_GLASS: { /* :339-401 */
            double cos_theta = v_dot(n, v);
            int entering = cos_theta > 0.0;
            v3 nf = dir_normal(entering, n);
            double sin2theta = 1.0 - cos_theta * cos_theta;
            double ior_ratio = dir_ior_ratio(entering, m->ior);
            if (ior_ratio * ior_ratio * sin2theta >= 1.0) {
                v3 l = reflect(nf, v);
                return do_scatter(v_scale(reflect_brdf(m->color, nf, l), v_dot(nf, l)), l);
            }
            double fresnel = entering ? schlick_scalar(1.0, m->ior, nf, v) : schlick_scalar(m->ior, 1.0, nf, v);
            if (rng_next(rng) < fresnel) {
                v3 l = reflect(nf, v);
                return do_scatter(v_scale(reflect_brdf(m->color, nf, l), v_dot(nf, l)), l);
            }
            v3 l;
            if (!refract(nf, v, ior_ratio, &l)) return no_scatter(); /* .unwrap() */
            return do_scatter(v_scale(refract_btdf(m->color, nf, l, v), rr_fabs(v_dot(nf, l))), l);
        }
        case ORC_MAT_COOK_TORRANCE: return ct_scatter(&m->ct, n, v, rng);
        case ORC_MAT_COOK_TORRANCE_REFRACT: { /* :426-467 */
            int entering = v_dot(n, v) > 0.0;
            v3 nf = dir_normal(entering, n);
            double ior_ratio = dir_ior_ratio(entering, m->ior);
            double value;
            v3 h = beckmann_generate(m->ct.alpha2, nf, rng, &value);
            h = dir_normal(entering, h);
            v3 l;
            if (!refract(h, v, ior_ratio, &l)) return no_scatter();
            return ct_evaluate_refraction(&m->ct, nf, h, v, l, value, entering, ior_ratio);
        }
        case ORC_MAT_COOK_TORRANCE_GLASS: { /* :469-565 */
            double value;
            v3 h = beckmann_generate(m->ct.alpha2, n, rng, &value);
            int entering = v_dot(n, v) > 0.0;
            h = dir_normal(entering, h);
            v3 nf = dir_normal(entering, n);
            double cos_theta = v_dot(h, v);
            double ior_ratio = dir_ior_ratio(entering, m->ior);
            double sin2_theta = 1.0 - cos_theta * cos_theta;
            if (ior_ratio * ior_ratio * sin2_theta >= 1.0) {
                v3 l = reflect(h, v);
                return ct_evaluate_reflection(&m->ct, nf, h, v, l, value);
            }
            double fresnel = entering ? schlick_scalar(1.0, m->ior, h, v) : schlick_scalar(m->ior, 1.0, h, v);
            if (rng_next(rng) < fresnel) {
                v3 l = reflect(h, v);
                scat_t s = ct_evaluate_reflection(&m->ct, nf, h, v, l, value);
                if (s.scatter) s.color = v_div(s.color, fresnel);
                return s;
            }
            v3 l;
            if (!refract(h, v, ior_ratio, &l)) return no_scatter(); /* .expect() */
            scat_t s = ct_evaluate_refraction(&m->ct, nf, h, v, l, value, entering, ior_ratio);
            if (s.scatter) s.color = v_div(s.color, 1.0 - fresnel);
            return s;
        }
        case ORC_MAT_PLASTIC: { /* :567-593 */
            double fresnel = schlick_scalar(1.0, m->ior, n, v);
            if (rng_next(rng) < fresnel) {
                scat_t s = ct_scatter(&m->ct, n, v, rng);
                if (s.scatter) s.color = v_div(s.color, fresnel);
                return s;
            }
            return lambertian_scatter(m->color, n, rng);
        }
        default: return no_scatter(); /* NoReflect */
    }
}

/* -------------------------------------------------------- scene and BVH */

typedef struct {
    shape_t geom;
    int mat;  /* index into scene->mats */
    int emis; /* index into scene->emis */
} object_t;

typedef struct tnode {
    int is_leaf;
    int obj;
    aabb_t box;
    int nchild;
    struct tnode* child[4];
} tnode;

struct orc_scene {
    object_t* objs;
    size_t nobjs, cap_objs;
    mat_t* mats;
    emis_t* emis;
    size_t nmats, cap_mats;
    /* built */
    int built;
    double t0, t1;
    tnode* pool;
    size_t pool_used, pool_cap;
    tnode* root;
    uint32_t hdri_w, hdri_h;
    double* hdri; /* w*h*3 f64, clipped */
    /* flattened */
    orc_flat_info finfo;
    double* child_box;
    uint32_t* child_ref;
    uint32_t* prim_object;
    double* wide_box;   /* n_wide * 4 * 6 */
    uint32_t* wide_ref; /* n_wide * 4 */
    int have_wide;      /* orc_set_wide was called */
    int have_hot;       /* orc_set_hot_group was called after it: the product's hot group (include/rayrs_hip.h hot_*) */
    double hot_box[6];
    uint32_t hot_first, hot_count;
};

orc_scene* orc_scene_create(void) { return (orc_scene*)calloc(1, sizeof(orc_scene)); }

void orc_scene_destroy(orc_scene* s) {
    if (!s) return;
    free(s->objs);
    free(s->mats);
    free(s->emis);
    free(s->pool);
    free(s->hdri);
    free(s->child_box);
    free(s->child_ref);
    free(s->prim_object);
    free(s->wide_box);
    free(s->wide_ref);
    free(s);
}

static int scene_push_surface(orc_scene* s, const orc_material* m, const orc_emission* e) {
    mat_t mt;
    emis_t em;
    if (!m) return -1;
    if (mat_from_desc(m, &mt) != 0) return -1;
    if (emis_from_desc(e, &em) != 0) return -1;
    if (s->nmats == s->cap_mats) {
        s->cap_mats = s->cap_mats ? s->cap_mats * 2 : 16;
        s->mats = (mat_t*)realloc(s->mats, s->cap_mats * sizeof(mat_t));
        s->emis = (emis_t*)realloc(s->emis, s->cap_mats * sizeof(emis_t));
    }
    s->mats[s->nmats] = mt;
    s->emis[s->nmats] = em;
    return (int)(s->nmats++);
}

static object_t* scene_push_objects(orc_scene* s, size_t n) {
    if (s->nobjs + n > s->cap_objs) {
        size_t cap = s->cap_objs ? s->cap_objs : 64;
        while (cap < s->nobjs + n) cap *= 2;
        s->objs = (object_t*)realloc(s->objs, cap * sizeof(object_t));
        s->cap_objs = cap;
    }
    object_t* r = s->objs + s->nobjs;
    s->nobjs += n;
    return r;
}

/* Object::sphere lib.rs:321-327, Sphere::new geometry.rs:96-102 */
int orc_add_sphere(orc_scene* s, double radius, const double origin[3], const orc_material* m, const orc_emission* e) {
    if (!(radius > 0.0)) return -1;
    int surf = scene_push_surface(s, m, e);
    if (surf < 0) return -1;
    object_t* o = scene_push_objects(s, 1);
    memset(o, 0, sizeof(*o));
    o->geom.kind = ORC_SHAPE_SPHERE;
    o->geom.radius2 = radius * radius;
    o->geom.origin = v_from(origin);
    o->mat = o->emis = surf;
    return 0;
}

/* Object::plane lib.rs:342-357, Plane::new geometry.rs:204-225 */
int orc_add_plane(orc_scene* s, int axis, double umin, double umax, double vmin, double vmax, double pos,
                  const orc_material* m, const orc_emission* e) {
    if (!(umin < umax && vmin < vmax)) return -1;
    if (axis < 0 || axis > 5) return -1;
    int surf = scene_push_surface(s, m, e);
    if (surf < 0) return -1;
    object_t* o = scene_push_objects(s, 1);
    memset(o, 0, sizeof(*o));
    o->geom.kind = ORC_SHAPE_PLANE;
    o->geom.axis = axis;
    o->geom.u0 = umin;
    o->geom.u1 = umax;
    o->geom.v0 = vmin;
    o->geom.v1 = vmax;
    o->geom.pos = pos;
    o->mat = o->emis = surf;
    return 0;
}

/* Object::triangle lib.rs:380-386 */
int orc_add_triangle(orc_scene* s, const double p1[3], const double p2[3], const double p3[3], const orc_material* m,
                     const orc_emission* e) {
    int surf = scene_push_surface(s, m, e);
    if (surf < 0) return -1;
    object_t* o = scene_push_objects(s, 1);
    triangle_init(&o->geom, v_from(p1), v_from(p2), v_from(p3));
    o->mat = o->emis = surf;
    return 0;
}

/* Object::from_triangles lib.rs:407-415 (one material/emission cloned per triangle) */
int orc_add_triangles(orc_scene* s, const double* verts, uint32_t nverts, const uint32_t* idx, uint32_t ntris,
                      const orc_material* m, const orc_emission* e) {
    for (uint32_t i = 0; i < 3 * ntris; i++)
        if (idx[i] >= nverts) return -1;
    int surf = scene_push_surface(s, m, e);
    if (surf < 0) return -1;
    object_t* o = scene_push_objects(s, ntris);
    for (uint32_t i = 0; i < ntris; i++) {
        triangle_init(&o[i].geom, v_from(verts + 3 * (size_t)idx[3 * i]), v_from(verts + 3 * (size_t)idx[3 * i + 1]),
                      v_from(verts + 3 * (size_t)idx[3 * i + 2]));
        o[i].mat = o[i].emis = surf;
    }
    return 0;
}

/* ---- builder, bvh.rs:7-38, :81-185, :227-389 ---- */

typedef struct {
    const object_t* objs;
    aabb_t* boxes;  /* per object */
    v3* centers;    /* per object, BvhData::new bvh.rs:88-95 */
    uint32_t* ids;  /* current order of the objects */
    uint32_t* tmp;  /* merge-sort scratch */
    double* sa_pre; /* sweep builder scratch */
    double* sa_suf;
    orc_scene* scene;
    int sweep;
    uint32_t splits;
} build_t;

static inline double center_axis(const build_t* b, uint32_t id, int axis) {
    const v3* c = &b->centers[id];
    return axis == 0 ? c->x : (axis == 1 ? c->y : c->z);
}

/* slice::sort_by is a stable merge sort; BvhData::sort bvh.rs:97-136 */
static void stable_sort_ids(build_t* b, size_t lo, size_t hi, int axis) {
    size_t n = hi - lo;
    if (n < 2) return;
    uint32_t* a = b->ids + lo;
    /* already sorted? then a stable sort is the identity */
    int sorted = 1;
    for (size_t i = 1; i < n; i++)
        if (center_axis(b, a[i - 1], axis) > center_axis(b, a[i], axis)) {
            sorted = 0;
            break;
        }
    if (sorted) return;
    uint32_t* t = b->tmp + lo;
    /* bottom-up merge sort, insertion sort on runs of 8 */
    const size_t RUN = 8;
    for (size_t s = 0; s < n; s += RUN) {
        size_t e = s + RUN < n ? s + RUN : n;
        for (size_t i = s + 1; i < e; i++) {
            uint32_t key = a[i];
            double kv = center_axis(b, key, axis);
            size_t j = i;
            while (j > s && center_axis(b, a[j - 1], axis) > kv) {
                a[j] = a[j - 1];
                j--;
            }
            a[j] = key;
        }
    }
    uint32_t* src = a;
    uint32_t* dst = t;
    for (size_t w = RUN; w < n; w *= 2) {
        for (size_t s = 0; s < n; s += 2 * w) {
            size_t mid = s + w < n ? s + w : n;
            size_t e = s + 2 * w < n ? s + 2 * w : n;
            size_t i = s, j = mid, k = s;
            while (i < mid && j < e) {
                /* take from the right only if strictly smaller: stability */
                if (center_axis(b, src[j], axis) < center_axis(b, src[i], axis))
                    dst[k++] = src[j++];
                else
                    dst[k++] = src[i++];
            }
            while (i < mid) dst[k++] = src[i++];
            while (j < e) dst[k++] = src[j++];
        }
        uint32_t* sw = src;
        src = dst;
        dst = sw;
    }
    if (src != a) memcpy(a, src, n * sizeof(uint32_t));
}

/* AxisAlignedBoundingBox::from_object_list, geometry.rs:544-550 */
static aabb_t bbox_of_ids(const build_t* b, size_t lo, size_t hi) {
    aabb_t box = b->boxes[b->ids[lo]];
    for (size_t i = lo + 1; i < hi; i++) box = aabb_expand(box, b->boxes[b->ids[i]]);
    return box;
}

/* calculate_sah, bvh.rs:15-38 */
static double calculate_sah_naive(const build_t* b, double ct, double ci, double surface_area, size_t lo, size_t mid,
                                  size_t hi) {
    double p_left = 0.0, p_right = 0.0;
    if (mid > lo) p_left = aabb_surface_area(bbox_of_ids(b, lo, mid)) / surface_area;
    if (hi > mid) p_right = aabb_surface_area(bbox_of_ids(b, mid, hi)) / surface_area;
    return ct + ci * (p_left * (double)(mid - lo) + p_right * (double)(hi - mid));
}

static tnode* new_node(orc_scene* s) {
    tnode* n = &s->pool[s->pool_used++];
    memset(n, 0, sizeof(*n));
    return n;
}

static tnode* new_leaf(orc_scene* s, uint32_t obj) {
    tnode* n = new_node(s);
    n->is_leaf = 1;
    n->obj = (int)obj;
    return n;
}

/* split_ind / BvhData::split_index: first position whose centre > split, or
 * "None" (returned as n) -- bvh.rs:7-13, :138-143 */
static size_t split_index(const build_t* b, size_t lo, size_t hi, int axis, double split) {
    for (size_t i = lo; i < hi; i++)
        if (center_axis(b, b->ids[i], axis) > split) return i - lo;
    return hi - lo; /* None */
}

static tnode* build_rec(build_t* b, size_t lo, size_t hi, int sah) {
    size_t len = hi - lo;
    aabb_t bbox = bbox_of_ids(b, lo, hi); /* data.bbox(), bvh.rs:231 / :323 */
    tnode* node = new_node(b->scene);
    node->box = bbox;
    if (len > 4) {
        double x = bbox.xmax - bbox.xmin;
        double y = bbox.ymax - bbox.ymin;
        double z = bbox.zmax - bbox.zmin;
        int axis;
        double min, extent;
        if (x >= y && x >= z) { /* bvh.rs:248-257 / :337-346 */
            axis = 0;
            min = bbox.xmin;
            extent = x;
        } else if (y >= z) {
            axis = 1;
            min = bbox.ymin;
            extent = y;
        } else {
            axis = 2;
            min = bbox.zmin;
            extent = z;
        }
        stable_sort_ids(b, lo, hi, axis);
        size_t ind;
        int have = 0;
        if (sah) {
            double surface_area = aabb_surface_area(bbox);
            double split_dist = extent / (double)(b->splits - 1u); /* bvh.rs:259 */
            double min_sah = INFINITY;
            size_t min_ind = 0;
            if (!b->sweep) {
                for (uint32_t i = 1; i < b->splits + 1u; i++) { /* bvh.rs:262-271 */
                    size_t k = split_index(b, lo, hi, axis, min + (double)i * split_dist);
                    if (k < len) {
                        double c = calculate_sah_naive(b, 0.3, 1.0, surface_area, lo, lo + k, hi);
                        if (c < min_sah) {
                            min_sah = c;
                            min_ind = k;
                            have = 1;
                        }
                    }
                }
            } else {
                /* same candidates and same cost values via prefix/suffix
                 * boxes: min/max are exact, so the boxes are bit-identical
                 * to the from_object_list folds. */
                double* pre = b->sa_pre + lo;
                double* suf = b->sa_suf + lo;
                aabb_t acc = b->boxes[b->ids[lo]];
                pre[0] = 0.0; /* unused: left empty */
                for (size_t i = 1; i < len; i++) {
                    /* pre[i] = SA(bbox(ids[lo..lo+i])) */
                    pre[i] = aabb_surface_area(acc);
                    acc = aabb_expand(acc, b->boxes[b->ids[lo + i]]);
                }
                /* suffix boxes must be folded left-to-right in the
                 * reference; min/max folds are order independent */
                acc = b->boxes[b->ids[hi - 1]];
                suf[len - 1] = aabb_surface_area(acc);
                for (size_t i = len - 1; i-- > 0;) {
                    acc = aabb_expand(b->boxes[b->ids[lo + i]], acc);
                    suf[i] = aabb_surface_area(acc);
                }
                size_t k = 0;
                for (uint32_t i = 1; i < b->splits + 1u; i++) {
                    double split = min + (double)i * split_dist;
                    while (k < len && !(center_axis(b, b->ids[lo + k], axis) > split)) k++;
                    if (k < len) {
                        double p_left = k > 0 ? pre[k] / surface_area : 0.0;
                        double p_right = suf[k] / surface_area;
                        double c = 0.3 + 1.0 * (p_left * (double)k + p_right * (double)(len - k));
                        if (c < min_sah) {
                            min_sah = c;
                            min_ind = k;
                            have = 1;
                        }
                    } else {
                        break; /* larger splits are None as well */
                    }
                }
            }
            ind = min_ind;
        } else {
            /* build_midpoint, bvh.rs:337-348 */
            v3 c = aabb_center(bbox);
            double split = axis == 0 ? c.x : (axis == 1 ? c.y : c.z);
            size_t k = split_index(b, lo, hi, axis, split);
            have = k < len;
            ind = k;
        }
        if (have) { /* bvh.rs:279-287 / :352-360 */
            if (ind == 0 || ind == len - 1) ind = len / 2;
        } else {
            ind = len / 2;
        }
        size_t mid = lo + ind;
        node->nchild = 2;
        node->child[0] = (mid - lo > 1) ? build_rec(b, lo, mid, sah) : new_leaf(b->scene, b->ids[lo]);
        node->child[1] = (hi - mid > 1) ? build_rec(b, mid, hi, sah) : new_leaf(b->scene, b->ids[mid]);
    } else {
        node->nchild = (int)len; /* bvh.rs:306-316 */
        for (size_t i = 0; i < len; i++) node->child[i] = new_leaf(b->scene, b->ids[lo + i]);
    }
    return node;
}

/* ---- flatten ---- */

#define REF_KIND_INTERIOR 0u
#define REF_KIND_RANGE 1u
#define REF_KIND_SINGLE 2u
#define REF_KIND_NONE 3u

static int node_is_bottom(const tnode* n) {
    for (int i = 0; i < n->nchild; i++)
        if (!n->child[i]->is_leaf) return 0;
    return 1;
}

typedef struct {
    orc_scene* s;
    uint32_t n_int, n_prim, depth;
} flat_t;

/* A split node (len > 4) always has two children; a bottom node (len <= 4)
 * has 1..4 leaf children.  A split node whose two children are both single
 * leaves cannot occur (len > 4), so "two children, both leaves" is always a
 * bottom node of two objects. */
static int node_is_split(const tnode* n) { return !n->is_leaf && !node_is_bottom(n); }

static uint32_t emit_ref(flat_t* f, const tnode* n, uint32_t* next_int, uint32_t* next_prim);

static void emit_interior(flat_t* f, const tnode* n, uint32_t rec, uint32_t* next_int, uint32_t* next_prim) {
    orc_scene* s = f->s;
    for (int c = 0; c < 2; c++) {
        const tnode* ch = n->child[c];
        double* bx = s->child_box + ((size_t)rec * 2 + (size_t)c) * 6;
        if (ch->is_leaf) {
            /* the reference does not box-test a direct leaf child; keep
             * the object's box for reference only */
            aabb_t b = shape_bbox(&s->objs[ch->obj].geom);
            bx[0] = b.xmin; bx[1] = b.xmax; bx[2] = b.ymin; bx[3] = b.ymax; bx[4] = b.zmin; bx[5] = b.zmax;
        } else {
            bx[0] = ch->box.xmin; bx[1] = ch->box.xmax; bx[2] = ch->box.ymin;
            bx[3] = ch->box.ymax; bx[4] = ch->box.zmin; bx[5] = ch->box.zmax;
        }
        s->child_ref[(size_t)rec * 2 + (size_t)c] = emit_ref(f, ch, next_int, next_prim);
    }
}

static uint32_t emit_ref(flat_t* f, const tnode* n, uint32_t* next_int, uint32_t* next_prim) {
    orc_scene* s = f->s;
    if (n->is_leaf) {
        uint32_t p = (*next_prim)++;
        s->prim_object[p] = (uint32_t)n->obj;
        return (REF_KIND_SINGLE << 30) | (p << 2);
    }
    if (node_is_split(n)) {
        uint32_t rec = (*next_int)++;
        emit_interior(f, n, rec, next_int, next_prim);
        return (REF_KIND_INTERIOR << 30) | rec;
    }
    uint32_t first = *next_prim;
    for (int i = 0; i < n->nchild; i++) s->prim_object[(*next_prim)++] = (uint32_t)n->child[i]->obj;
    return (REF_KIND_RANGE << 30) | (first << 2) | (uint32_t)(n->nchild - 1);
}

static void count_nodes(const tnode* n, uint32_t d, uint32_t* n_int, uint32_t* n_prim, uint32_t* depth) {
    if (n->is_leaf) {
        (*n_prim)++;
        return;
    }
    if (node_is_split(n)) {
        (*n_int)++;
        if (d + 1 > *depth) *depth = d + 1;
        count_nodes(n->child[0], d + 1, n_int, n_prim, depth);
        count_nodes(n->child[1], d + 1, n_int, n_prim, depth);
    } else {
        *n_prim += (uint32_t)n->nchild;
    }
}

static void flatten(orc_scene* s) {
    uint32_t n_int = 0, n_prim = 0, depth = 0;
    count_nodes(s->root, 0, &n_int, &n_prim, &depth);
    s->child_box = (double*)calloc((size_t)(n_int ? n_int : 1) * 12, sizeof(double));
    s->child_ref = (uint32_t*)calloc((size_t)(n_int ? n_int : 1) * 2, sizeof(uint32_t));
    s->prim_object = (uint32_t*)calloc(n_prim ? n_prim : 1, sizeof(uint32_t));
    flat_t f;
    f.s = s;
    f.n_int = n_int;
    f.n_prim = n_prim;
    f.depth = depth;
    uint32_t ni = 0, np = 0;
    s->finfo.root_ref = emit_ref(&f, s->root, &ni, &np);
    s->finfo.n_interior = n_int;
    s->finfo.n_prims = n_prim;
    s->finfo.depth = depth;
    s->finfo.root_box[0] = s->root->box.xmin;
    s->finfo.root_box[1] = s->root->box.xmax;
    s->finfo.root_box[2] = s->root->box.ymin;
    s->finfo.root_box[3] = s->root->box.ymax;
    s->finfo.root_box[4] = s->root->box.zmin;
    s->finfo.root_box[5] = s->root->box.zmax;
}

/* The tree the HIP traversal kernel walks is NOT built here: it is the product's own
 * (rayrs_amd/csrc/scene_host.cpp build_walk_trees), handed over through
 * rayrs_scene_export_wide and orc_set_wide so that isect_wide below makes the kernel's walk
 * on the kernel's data.  What makes that tree legal -- every group of the reference's tree
 * appears exactly once behind its exact gating box, every interior box contains what is
 * below it -- is checked from the outside by tests/test_bvh_builder.py, and that the walk
 * returns the reference's hits by comparing it with isect_reference (traversal 0). */
int orc_set_wide(orc_scene* s, uint32_t n_wide, uint32_t wide_root_ref, uint32_t wide_depth, const double* wide_box,
                 const uint32_t* wide_ref) {
    if (!s || !s->built) return -1;
    free(s->wide_box);
    free(s->wide_ref);
    s->wide_box = (double*)malloc(((size_t)n_wide * 24 + 1) * sizeof(double));
    s->wide_ref = (uint32_t*)malloc(((size_t)n_wide * 4 + 1) * sizeof(uint32_t));
    if (n_wide) {
        memcpy(s->wide_box, wide_box, (size_t)n_wide * 24 * sizeof(double));
        memcpy(s->wide_ref, wide_ref, (size_t)n_wide * 4 * sizeof(uint32_t));
    }
    s->finfo.n_wide = n_wide;
    s->finfo.wide_root_ref = wide_root_ref;
    s->finfo.wide_depth = wide_depth;
    s->have_wide = 1;
    s->have_hot = 0;
    return 0;
}

/* The product's hot group (rayrs_scene_info_t.hot_box / hot_first / hot_count): a group of the gate tree that the
 * records handed over with orc_set_wide (rayrs_scene_export_hot_tree) leave out and the default walk's kernels test
 * once per ray beside the walk.  isect_wide then does the same: the gating box as the reference tests it, and the
 * group's primitives if the ray enters it.  Call after orc_set_wide (which forgets the group). */
int orc_set_hot_group(orc_scene* s, const double box[6], uint32_t first, uint32_t count) {
    if (!s || !s->built || !s->have_wide || count < 1 || count > 4 || (size_t)first + count > s->finfo.n_prims) return -1;
    memcpy(s->hot_box, box, 6 * sizeof(double));
    s->hot_first = first;
    s->hot_count = count;
    s->have_hot = 1;
    return 0;
}

/* Scene::new lib.rs:227-245, Bvh::build bvh.rs:199-210 */
int orc_scene_build(orc_scene* s, double z_near, double z_far, int heuristic, uint32_t splits, int builder,
                    uint32_t hdri_w, uint32_t hdri_h, const float* hdri_rgb) {
    if (!s || s->built) return -1;
    if (!(z_near >= 0.0) || !(z_far > z_near)) return -1; /* lib.rs:234-235 */
    if (s->nobjs == 0) return -1;                        /* bvh.rs:229 */
    if (heuristic == 1 && splits < 2) return -1;
    if (hdri_w < 2 || hdri_h < 2 || !hdri_rgb) return -1;
    s->t0 = z_near;
    s->t1 = z_far;

    build_t b;
    memset(&b, 0, sizeof(b));
    b.objs = s->objs;
    b.scene = s;
    b.sweep = builder;
    b.splits = splits;
    size_t n = s->nobjs;
    b.boxes = (aabb_t*)malloc(n * sizeof(aabb_t));
    b.centers = (v3*)malloc(n * sizeof(v3));
    b.ids = (uint32_t*)malloc(n * sizeof(uint32_t));
    b.tmp = (uint32_t*)malloc(n * sizeof(uint32_t));
    b.sa_pre = (double*)malloc(n * sizeof(double));
    b.sa_suf = (double*)malloc(n * sizeof(double));
    for (size_t i = 0; i < n; i++) {
        b.boxes[i] = shape_bbox(&s->objs[i].geom);
        b.centers[i] = aabb_center(b.boxes[i]);
        b.ids[i] = (uint32_t)i;
    }
    s->pool_cap = 2 * n + 8;
    s->pool = (tnode*)malloc(s->pool_cap * sizeof(tnode));
    s->pool_used = 0;
    s->root = build_rec(&b, 0, n, heuristic == 1);
    free(b.boxes);
    free(b.centers);
    free(b.ids);
    free(b.tmp);
    free(b.sa_pre);
    free(b.sa_suf);

    /* HDRI: f32 -> f64, clip(0, 3) (main.rs:42-43; clip = min(max).max(min), vecmath.rs:388-396) */
    s->hdri_w = hdri_w;
    s->hdri_h = hdri_h;
    size_t nt = (size_t)hdri_w * hdri_h * 3;
    s->hdri = (double*)malloc(nt * sizeof(double));
    for (size_t i = 0; i < nt; i++) s->hdri[i] = rr_max(rr_min((double)hdri_rgb[i], 3.0), 0.0);

    flatten(s);
    s->built = 1;
    return 0;
}

int orc_flatten_info(const orc_scene* s, orc_flat_info* info) {
    if (!s || !s->built) return -1;
    *info = s->finfo;
    return 0;
}

int orc_flatten_export(const orc_scene* s, double* child_box, uint32_t* child_ref, uint32_t* prim_object) {
    if (!s || !s->built) return -1;
    memcpy(child_box, s->child_box, (size_t)s->finfo.n_interior * 12 * sizeof(double));
    memcpy(child_ref, s->child_ref, (size_t)s->finfo.n_interior * 2 * sizeof(uint32_t));
    memcpy(prim_object, s->prim_object, (size_t)s->finfo.n_prims * sizeof(uint32_t));
    return 0;
}

int orc_flatten_export_wide(const orc_scene* s, double* wide_box, uint32_t* wide_ref) {
    if (!s || !s->built || !s->have_wide) return -1;
    memcpy(wide_box, s->wide_box, (size_t)s->finfo.n_wide * 24 * sizeof(double));
    memcpy(wide_ref, s->wide_ref, (size_t)s->finfo.n_wide * 4 * sizeof(uint32_t));
    return 0;
}

int orc_scene_bbox(const orc_scene* s, double box[6], double center[3], double* volume, double* surface_area) {
    if (!s || s->nobjs == 0) return -1;
    aabb_t b = shape_bbox(&s->objs[0].geom);
    for (size_t i = 1; i < s->nobjs; i++) b = aabb_expand(b, shape_bbox(&s->objs[i].geom));
    box[0] = b.xmin; box[1] = b.xmax; box[2] = b.ymin; box[3] = b.ymax; box[4] = b.zmin; box[5] = b.zmax;
    v_to(aabb_center(b), center);
    *volume = aabb_volume(b);
    *surface_area = aabb_surface_area(b);
    return 0;
}

/* The bounding boxes Bvh::build starts from (bvh.rs:212-226: object.bbox() of every object, geometry.rs), in
 * insertion order: boxes = nobjs * 6 (xmin xmax ymin ymax zmin zmax). */
int orc_object_boxes(const orc_scene* s, double* boxes) {
    if (!s) return -1;
    for (size_t i = 0; i < s->nobjs; i++) {
        aabb_t b = shape_bbox(&s->objs[i].geom);
        double* o = boxes + i * 6;
        o[0] = b.xmin; o[1] = b.xmax; o[2] = b.ymin; o[3] = b.ymax; o[4] = b.zmin; o[5] = b.zmax;
    }
    return 0;
}

/* ---- traversal ---- */

typedef struct {
    int hit;
    double t;
    int obj;
} isect_t;

/* RayIntersection::update, bvh.rs:50-72 */
static inline isect_t isect_update(isect_t self, isect_t other, double tmin) {
    if (!self.hit) return other;
    if (!other.hit) return self;
    if (other.t > tmin && other.t < self.t) return other;
    return self;
}

/* BvhTree::intersect, bvh.rs:391-415 */
static isect_t isect_reference(const orc_scene* s, const tnode* n, ray_t ray, double tmin, double tmax) {
    isect_t miss = {0, 0.0, -1};
    if (!n->is_leaf) {
        if (aabb_intersect(&n->box, ray, tmin, tmax)) {
            isect_t acc = miss;
            for (int i = 0; i < n->nchild; i++)
                acc = isect_update(acc, isect_reference(s, n->child[i], ray, tmin, tmax), tmin);
            return acc;
        }
        return miss;
    }
    double t;
    if (shape_intersect(&s->objs[n->obj].geom, ray, &t)) {
        if (t > tmin && t < tmax) {
            isect_t h = {1, t, n->obj};
            return h;
        }
    }
    return miss;
}

typedef struct {
    uint64_t interior_visits, tri_tests, sphere_tests, plane_tests;
} trav_counters;

/* Boxes entered beyond the closest hit so far are skipped -- beyond it by this relative margin:
 * a primitive's computed t and the entry parameter of the box around it are rounded
 * independently, so a hit an ulp in front of its own box must not be lost
 * (rayrs_amd/csrc/device_path.h has the same constant; the counters only agree if both do). */
/* the kernel's margin (device_path.h TRAV_CULL_MARGIN); orc_set_cull_margin swaps it for experiments
 * (scripts/fuzz_traversal.py measures what a margin protects and what it costs) */
static double g_cull_margin = 1.0 + 0x1p-10;
void orc_set_cull_margin(double rel) { g_cull_margin = 1.0 + rel; }
#define TRAV_CULL_MARGIN g_cull_margin

/* The kernel's traversal, restated on the flattened tree: children tested at
 * the parent, near child first, far child pushed, boxes whose entry lies
 * beyond the closest hit skipped.  Returns the reference's answer: closest
 * accepted t, the first primitive in DFS order on exact ties (bvh.rs:62). */
static isect_t isect_ordered(const orc_scene* s, ray_t ray, double tmin, double tmax, trav_counters* cnt) {
    isect_t best = {0, 0.0, -1};
    uint32_t best_prim = 0xffffffffu;
    double best_t = tmax; /* accept t < tmax */
    v3 inv = V(1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z);
    double entry;
    if (!aabb_intersect_entry(s->finfo.root_box, ray, inv, tmin, tmax, &entry)) return best;
    uint32_t stack[256];
    int sp = 0;
    uint32_t cur = s->finfo.root_ref;
    for (;;) {
        uint32_t kind = cur >> 30;
        if (kind == REF_KIND_INTERIOR) {
            uint32_t rec = cur & 0x3fffffffu;
            if (cnt) cnt->interior_visits++;
            int hit[2];
            double ent[2];
            for (int c = 0; c < 2; c++) {
                uint32_t r = s->child_ref[(size_t)rec * 2 + c];
                if ((r >> 30) == REF_KIND_SINGLE) {
                    hit[c] = 1;
                    ent[c] = tmin;
                } else {
                    hit[c] = aabb_intersect_entry(s->child_box + ((size_t)rec * 2 + c) * 6, ray, inv, tmin, tmax,
                                                  &ent[c]);
                    if (hit[c] && ent[c] > best_t * TRAV_CULL_MARGIN) hit[c] = 0;
                }
            }
            uint32_t r0 = s->child_ref[(size_t)rec * 2], r1 = s->child_ref[(size_t)rec * 2 + 1];
            if (hit[0] && hit[1]) {
                if (ent[1] < ent[0]) {
                    stack[sp++] = r0;
                    cur = r1;
                } else {
                    stack[sp++] = r1;
                    cur = r0;
                }
                continue;
            } else if (hit[0]) {
                cur = r0;
                continue;
            } else if (hit[1]) {
                cur = r1;
                continue;
            }
        } else {
            uint32_t first = (cur & 0x3fffffffu) >> 2;
            uint32_t count = (cur & 3u) + 1u;
            for (uint32_t k = 0; k < count; k++) {
                uint32_t p = first + k;
                int obj = (int)s->prim_object[p];
                const shape_t* g = &s->objs[obj].geom;
                if (cnt) {
                    if (g->kind == ORC_SHAPE_TRIANGLE) cnt->tri_tests++;
                    else if (g->kind == ORC_SHAPE_SPHERE) cnt->sphere_tests++;
                    else cnt->plane_tests++;
                }
                double t;
                if (shape_intersect(g, ray, &t) && t > tmin && t < tmax) {
                    if (!best.hit || t < best.t || (t == best.t && p < best_prim)) {
                        best.hit = 1;
                        best.t = t;
                        best.obj = obj;
                        best_prim = p;
                        best_t = t;
                    }
                }
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    return best;
}

/* Diagnostics: when set, isect_wide adds one to hist[record] per visit and to
 * hist[n_wide + min(sp, 63)] the stack height seen at each visit (single-threaded use). */
static uint64_t* g_visit_hist = NULL;
void orc_set_visit_histogram(uint64_t* hist) { g_visit_hist = hist; }

/* The walk the HIP traversal kernel makes, on the four-slot records of the product's walk tree
 * (orc_set_wide): up to four boxes per record, the hit slots entered nearest first (ties by
 * slot), the rest pushed farthest first.  Every leaf slot is a group of the reference's tree
 * behind its gating box, so the primitives tested are a subset of those the reference reaches
 * that contains the closest hit. */
static isect_t isect_wide(const orc_scene* s, ray_t ray, double tmin, double tmax, trav_counters* cnt) {
    if (!s->have_wide) {
        fprintf(stderr, "oracle: traversal 2 needs the product's walk tree (orc_set_wide)\n");
        abort();
    }
    isect_t best = {0, 0.0, -1};
    uint32_t best_prim = 0xffffffffu;
    double best_t = tmax;
    v3 inv = V(1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z);
    double entry;
    if (!aabb_intersect_entry(s->finfo.root_box, ray, inv, tmin, tmax, &entry)) return best;
    uint32_t stack_fixed[512];
    uint32_t* stack = stack_fixed;
    if (s->finfo.wide_depth + 4u > 512u) stack = (uint32_t*)malloc(((size_t)s->finfo.wide_depth + 4u) * sizeof(uint32_t));
    int sp = 0;
    uint32_t cur = s->finfo.wide_root_ref;
    if (s->have_hot) {
        /* the product's hot group: not in the records; behind its gating box, once per ray that enters the root box
         * (the order of visits does not matter: smallest accepted t, lowest DFS index on exact ties) */
        double e_hot;
        if (aabb_intersect_entry(s->hot_box, ray, inv, tmin, tmax, &e_hot))
            stack[sp++] = (REF_KIND_RANGE << 30) | (s->hot_first << 2) | (s->hot_count - 1u);
    }
    for (;;) {
        if ((cur >> 30) == REF_KIND_INTERIOR) {
            uint32_t rec = cur & 0x3fffffffu;
            if (cnt) cnt->interior_visits++;
            if (g_visit_hist) {
                g_visit_hist[rec]++;
                g_visit_hist[s->finfo.n_wide + (uint32_t)(sp < 63 ? sp : 63)]++;
            }
            int hit[4], n = 0;
            double ent[4];
            const uint32_t* refs = s->wide_ref + (size_t)rec * 4;
            for (int c = 0; c < 4; c++) {
                uint32_t kind = refs[c] >> 30;
                hit[c] = 0;
                ent[c] = 0.0;
                if (kind == REF_KIND_SINGLE) {
                    hit[c] = 1;
                    ent[c] = tmin;
                } else if (kind != REF_KIND_NONE) {
                    hit[c] = aabb_intersect_entry(s->wide_box + ((size_t)rec * 4 + c) * 6, ray, inv, tmin, tmax, &ent[c]);
                    if (hit[c] && ent[c] > best_t * TRAV_CULL_MARGIN) hit[c] = 0;
                }
                n += hit[c];
            }
            if (n > 0) {
                /* visit order: by entry, then by slot */
                int order[4], k = 0;
                for (int c = 0; c < 4; c++)
                    if (hit[c]) order[k++] = c;
                for (int a = 1; a < n; a++) { /* stable insertion sort */
                    int c = order[a], b = a;
                    while (b > 0 && ent[order[b - 1]] > ent[c]) {
                        order[b] = order[b - 1];
                        b--;
                    }
                    order[b] = c;
                }
                for (int a = n - 1; a >= 1; a--) stack[sp++] = refs[order[a]];
                cur = refs[order[0]];
                continue;
            }
        } else {
            uint32_t first = (cur & 0x3fffffffu) >> 2;
            uint32_t count = (cur & 3u) + 1u;
            for (uint32_t k = 0; k < count; k++) {
                uint32_t p = first + k;
                int obj = (int)s->prim_object[p];
                const shape_t* g = &s->objs[obj].geom;
                if (cnt) {
                    if (g->kind == ORC_SHAPE_TRIANGLE) cnt->tri_tests++;
                    else if (g->kind == ORC_SHAPE_SPHERE) cnt->sphere_tests++;
                    else cnt->plane_tests++;
                }
                double t;
                if (shape_intersect(g, ray, &t) && t > tmin && t < tmax) {
                    if (!best.hit || t < best.t || (t == best.t && p < best_prim)) {
                        best.hit = 1;
                        best.t = t;
                        best.obj = obj;
                        best_prim = p;
                        best_t = t;
                    }
                }
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    if (stack != stack_fixed) free(stack);
    return best;
}

static inline isect_t scene_intersect(const orc_scene* s, ray_t ray, int traversal, trav_counters* cnt) {
    if (traversal == 0) return isect_reference(s, s->root, ray, s->t0, s->t1);
    if (traversal == 2) return isect_wide(s, ray, s->t0, s->t1, cnt);
    return isect_ordered(s, ray, s->t0, s->t1, cnt);
}

/* ---- camera, lib.rs:99-133, :153-177, :202-210 ---- */

int orc_camera_new(const double origin[3], const double up[3], const double lookat[3], double fov, double width,
                   double height, uint32_t ppi, orc_camera* out) {
    if (!(fov > 0.0 && fov < 180.0)) return -1;
    if (!(width > 0.0) || !(height > 0.0)) return -1;
    v3 o = v_from(origin), u = v_from(up), la = v_from(lookat);
    if (o.x == la.x && o.y == la.y && o.z == la.z) return -1;
    uint32_t ppc = (uint32_t)round((double)ppi * 2.54);
    v3 z = v_unit(v_sub(la, o));
    v3 x = v_unit(v_cross(u, z));
    v3 y = v_unit(v_cross(z, x));
    /* f64::to_radians multiplies by pi/180 */
    double rad = fov * (RR_PI / 180.0);
    double f = width / m_tan(rad / 2.0);
    v3 zz = V(f * z.x, f * z.y, f * z.z);
    v_to(o, out->origin);
    v_to(x, out->e_x);
    v_to(y, out->e_y);
    v_to(zz, out->z);
    out->width = width;
    out->height = height;
    out->ppc = ppc;
    out->x_pixels = (uint32_t)round(width * (double)ppc);
    out->y_pixels = (uint32_t)round(height * (double)ppc);
    return 0;
}

static ray_t primary_ray(const orc_camera* c, uint32_t i, uint32_t j, rng_t* rng) {
    double fi = (double)i;
    double fj = (double)j;
    double x = (fj + rng_next(rng)) / (double)c->ppc - c->width / 2.0;
    double y = (fi + rng_next(rng)) / (double)c->ppc - c->height / 2.0;
    ray_t r;
    r.o = v_from(c->origin);
    r.d = v_add(v_add(v_from(c->z), v_scale(v_from(c->e_x), x)), v_scale(v_from(c->e_y), y));
    return r;
}

/* ---- Scene::background, lib.rs:254-285 ---- */

static inline v3 hdri_pixel(const orc_scene* s, size_t i, size_t j) {
    /* Image::pixel asserts bounds (image.rs:183-186); the only way out of
     * range is phi == 2*pi or theta == pi exactly, where every weight that
     * touches the out-of-range texel is zero.  Clamp instead of panicking. */
    if (i >= s->hdri_h) i = s->hdri_h - 1;
    if (j >= s->hdri_w) j = s->hdri_w - 1;
    const double* p = s->hdri + (i * (size_t)s->hdri_w + j) * 3;
    return V(p[0], p[1], p[2]);
}

static inline size_t f64_as_usize(double x) { /* Rust `as usize`: saturating, NaN -> 0 */
    if (!(x > 0.0)) return 0;
    if (x >= 18446744073709551615.0) return (size_t)-1;
    return (size_t)x;
}

static v3 background(const orc_scene* s, v3 dir) {
    dir = v_unit(dir);
    double phi = m_atan2(dir.z, dir.x) + RR_PI;
    double theta = m_acos(dir.y);
    double x = phi / (2.0 * RR_PI) * (double)(s->hdri_w - 1);
    double y = theta / RR_PI * (double)(s->hdri_h - 1);
    double x_f = rr_floor(x), x_c = rr_ceil(x), y_f = rr_floor(y), y_c = rr_ceil(y);
    size_t i = f64_as_usize(y_f);
    size_t j = f64_as_usize(x_f);
    v3 f0 = hdri_pixel(s, i, j), f1 = hdri_pixel(s, i + 1, j), f2 = hdri_pixel(s, i, j + 1),
       f3 = hdri_pixel(s, i + 1, j + 1);
    v3 a = v_scale(v_scale(f0, x_c - x), y_c - y);
    v3 b = v_scale(v_scale(f1, x_c - x), y - y_f);
    v3 c = v_scale(v_scale(f2, x - x_f), y_c - y);
    v3 d = v_scale(v_scale(f3, x - x_f), y - y_f);
    return v_add(v_add(v_add(a, b), c), d);
}

/* ---- radiance, lib.rs:521-560 ---- */

typedef struct {
    uint64_t rays, escaped;
    trav_counters trav;
} path_counters;

/* A path's trace (orc_path_trace): per loop iteration of radiance() the object the query found (-1: none), its t (0 for
 * a miss), and the throughput and the RNG's draw index on leaving the iteration. */
typedef struct {
    uint32_t cap, n;
    int64_t* obj;
    double* t;
    double* thr;
    uint32_t* draw;
} path_trace;

static void trace_put(path_trace* tr, uint32_t b, int64_t obj, double t, v3 thr, uint32_t draw) {
    if (!tr) return;
    if (b < tr->cap) {
        tr->obj[b] = obj, tr->t[b] = t, tr->draw[b] = draw;
        tr->thr[3 * b] = thr.x, tr->thr[3 * b + 1] = thr.y, tr->thr[3 * b + 2] = thr.z;
    }
    tr->n = b + 1;
}

static v3 radiance_traced(const orc_scene* s, ray_t r, uint32_t max_bounces, rng_t* rng, int traversal, path_counters* pc,
                          path_trace* tr);
static v3 radiance(const orc_scene* s, ray_t r, uint32_t max_bounces, rng_t* rng, int traversal, path_counters* pc) {
    return radiance_traced(s, r, max_bounces, rng, traversal, pc, NULL);
}

/* Diagnostics (scripts/sim/: the traversal kernel's scheduling is simulated on the rays a real frame makes): when set,
 * radiance() appends every ray it puts to the scene -- o, d and the loop iteration -- to the buffer (single-threaded use). */
static double* g_ray_dump = NULL;
static uint64_t g_ray_dump_cap = 0, g_ray_dump_n = 0;
void orc_set_ray_dump(double* rays7, uint64_t cap) { g_ray_dump = rays7, g_ray_dump_cap = cap, g_ray_dump_n = 0; }
uint64_t orc_ray_dump_count(void) { return g_ray_dump_n; }

static v3 radiance_traced(const orc_scene* s, ray_t r, uint32_t max_bounces, rng_t* rng, int traversal, path_counters* pc,
                          path_trace* tr) {
    v3 throughput = V(1.0, 1.0, 1.0);
    v3 light = V(0.0, 0.0, 0.0);
    for (uint32_t b = 0; b < max_bounces; b++) {
        if (pc) pc->rays++;
        if (g_ray_dump && g_ray_dump_n < g_ray_dump_cap) {
            double* q = g_ray_dump + 7 * g_ray_dump_n++;
            q[0] = r.o.x, q[1] = r.o.y, q[2] = r.o.z, q[3] = r.d.x, q[4] = r.d.y, q[5] = r.d.z, q[6] = (double)b;
        }
        isect_t h = scene_intersect(s, r, traversal, pc ? &pc->trav : NULL);
        if (h.hit) {
            const object_t* obj = &s->objs[h.obj];
            v3 position = ray_point(r, h.t);
            v3 normal = shape_normal(&obj->geom, position);
            v3 view = v_unit(v_scale(r.d, -1.0));
            scat_t ev = mat_evaluate(&s->mats[obj->mat], normal, view, rng);
            if (ev.scatter) {
                light = v_add(light, v_mul(throughput, emis_emit(&s->emis[obj->emis])));
                throughput = v_mul(throughput, ev.color);
                double p = rr_max(rr_max(throughput.x, throughput.y), throughput.z);
                if (rng_next(rng) > p) {
                    trace_put(tr, b, h.obj, h.t, throughput, rng->draw);
                    return light;
                }
                throughput = v_div_assign(throughput, p);
                r.o = position;
                r.d = ev.dir;
                trace_put(tr, b, h.obj, h.t, throughput, rng->draw);
            } else {
                trace_put(tr, b, h.obj, h.t, throughput, rng->draw);
                return light;
            }
        } else {
            if (pc) pc->escaped++;
            trace_put(tr, b, -1, 0.0, throughput, rng->draw);
            return v_add(light, v_mul(throughput, background(s, r.d)));
        }
    }
    return light;
}

/* ---- render: main.rs:57-101 ---- */

typedef struct {
    const orc_scene* s;
    const orc_camera* c;
    uint32_t spp, max_bounces, row0, row1, width, height, chunk;
    uint64_t seed;
    int traversal;
    double* out;
    uint32_t nblocks_x, nblocks_y;
    uint32_t next_block; /* atomic */
    pthread_mutex_t lock;
    orc_stats total;
} render_job;

static void* render_worker(void* arg) {
    render_job* job = (render_job*)arg;
    path_counters pc;
    memset(&pc, 0, sizeof(pc));
    uint64_t paths = 0, nan_px = 0, neg_px = 0;
    const uint32_t B = 16; /* main.rs:57 */
    for (;;) {
        uint32_t blk = __atomic_fetch_add(&job->next_block, 1u, __ATOMIC_RELAXED);
        if (blk >= job->nblocks_x * job->nblocks_y) break;
        uint32_t off_y = (blk / job->nblocks_x) * B, off_x = (blk % job->nblocks_x) * B;
        uint32_t bw = job->width - off_x < B ? job->width - off_x : B;
        uint32_t bh = job->height - off_y < B ? job->height - off_y : B;
        if (off_y + bh <= job->row0 || off_y >= job->row1) continue;
        for (uint32_t j = 0; j < bw; j++) {     /* main.rs:65 */
            for (uint32_t i = 0; i < bh; i++) { /* main.rs:66 */
                uint32_t row = i + off_y, col = j + off_x;
                if (row < job->row0 || row >= job->row1) continue;
                v3 pixel = V(0, 0, 0);
                uint64_t pix_index = (uint64_t)row * job->width + col;
                /* main.rs:67-79; with chunk < spp the samples are summed per
                 * chunk and the chunk sums added in order (build-defined
                 * variant used for load balancing on the GPU) */
                for (uint32_t s0 = 0; s0 < job->spp; s0 += job->chunk) {
                    v3 part = V(0, 0, 0);
                    uint32_t s1 = s0 + job->chunk < job->spp ? s0 + job->chunk : job->spp;
                    for (uint32_t sidx = s0; sidx < s1; sidx++) {
                        rng_t rng;
                        rng.key = rr_path_key(job->seed, pix_index, sidx);
                        rng.draw = 0;
                        /* main.rs:74-75: camera origin is lower right */
                        ray_t r = primary_ray(job->c, job->height - i - off_y, job->width - j - off_x, &rng);
                        v3 rad = radiance(job->s, r, job->max_bounces, &rng, job->traversal, &pc);
                        part = v_add(part, rad);
                        paths++;
                    }
                    pixel = s0 == 0 ? part : v_add(pixel, part);
                }
                if (rr_isnan(pixel.x) || rr_isnan(pixel.y) || rr_isnan(pixel.z)) nan_px++;
                if (pixel.x < 0.0 || pixel.y < 0.0 || pixel.z < 0.0) neg_px++;
                v3 px = v_div(pixel, (double)job->spp); /* main.rs:89 */
                v_to(px, job->out + pix_index * 3);
            }
        }
    }
    pthread_mutex_lock(&job->lock);
    job->total.rays += pc.rays;
    job->total.paths += paths;
    job->total.nan_pixels += nan_px;
    job->total.neg_pixels += neg_px;
    job->total.interior_visits += pc.trav.interior_visits;
    job->total.tri_tests += pc.trav.tri_tests;
    job->total.sphere_tests += pc.trav.sphere_tests;
    job->total.plane_tests += pc.trav.plane_tests;
    job->total.escaped_paths += pc.escaped;
    pthread_mutex_unlock(&job->lock);
    return NULL;
}

int orc_render(const orc_scene* s, const orc_camera* c, uint32_t spp, uint32_t max_bounces, uint64_t seed,
               uint32_t sample_chunk, uint32_t row0, uint32_t row1, int nthreads, int traversal, double* out_rgb,
               orc_stats* stats) {
    if (!s || !s->built || !c || !out_rgb || spp == 0) return -1;
    if (traversal == 1 && s->finfo.depth > 250) return -1;
    render_job job;
    memset(&job, 0, sizeof(job));
    job.s = s;
    job.c = c;
    job.spp = spp;
    job.max_bounces = max_bounces;
    job.seed = seed;
    job.chunk = (sample_chunk == 0 || sample_chunk >= spp) ? spp : sample_chunk;
    job.width = c->x_pixels;
    job.height = c->y_pixels;
    job.row0 = row0;
    job.row1 = row1 < job.height ? row1 : job.height;
    job.traversal = traversal;
    job.out = out_rgb;
    job.nblocks_x = (job.width + 15) / 16;
    job.nblocks_y = (job.height + 15) / 16;
    pthread_mutex_init(&job.lock, NULL);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    struct timespec ts0, ts1;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    pthread_t th[256];
    for (int i = 1; i < nthreads; i++) pthread_create(&th[i], NULL, render_worker, &job);
    render_worker(&job);
    for (int i = 1; i < nthreads; i++) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    pthread_mutex_destroy(&job.lock);
    job.total.seconds = (double)(ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double)(ts1.tv_nsec - ts0.tv_nsec);
    if (stats) *stats = job.total;
    return 0;
}

/* ---- unit-level entry points ---- */

int orc_aabb_intersect(const double box[6], const double o[3], const double d[3], double tmin, double tmax) {
    aabb_t b = {box[0], box[1], box[2], box[3], box[4], box[5]};
    ray_t r = {v_from(o), v_from(d)};
    return aabb_intersect(&b, r, tmin, tmax);
}

int orc_sphere_intersect(double radius, const double c[3], const double o[3], const double d[3], double* t) {
    shape_t s;
    memset(&s, 0, sizeof(s));
    s.kind = ORC_SHAPE_SPHERE;
    s.radius2 = radius * radius;
    s.origin = v_from(c);
    ray_t r = {v_from(o), v_from(d)};
    return sphere_intersect(&s, r, t);
}

int orc_plane_intersect(int axis, double umin, double umax, double vmin, double vmax, double pos, const double o[3],
                        const double d[3], double* t) {
    shape_t s;
    memset(&s, 0, sizeof(s));
    s.kind = ORC_SHAPE_PLANE;
    s.axis = axis;
    s.u0 = umin; s.u1 = umax; s.v0 = vmin; s.v1 = vmax; s.pos = pos;
    ray_t r = {v_from(o), v_from(d)};
    return plane_intersect(&s, r, t);
}

int orc_triangle_intersect(const double p1[3], const double p2[3], const double p3[3], const double o[3],
                           const double d[3], double* t) {
    shape_t s;
    triangle_init(&s, v_from(p1), v_from(p2), v_from(p3));
    ray_t r = {v_from(o), v_from(d)};
    return triangle_intersect(&s, r, t);
}

void orc_triangle_normal(const double p1[3], const double p2[3], const double p3[3], double n[3]) {
    shape_t s;
    triangle_init(&s, v_from(p1), v_from(p2), v_from(p3));
    v_to(s.normal, n);
}

int64_t orc_bvh_intersect(const orc_scene* s, const double o[3], const double d[3], double tmin, double tmax,
                          int traversal, double* t) {
    if (!s || !s->built) return -2;
    ray_t r = {v_from(o), v_from(d)};
    isect_t h = traversal == 0   ? isect_reference(s, s->root, r, tmin, tmax)
                : traversal == 2 ? isect_wide(s, r, tmin, tmax, NULL)
                                 : isect_ordered(s, r, tmin, tmax, NULL);
    if (!h.hit) return -1;
    *t = h.t;
    return h.obj;
}

/* ---- batches of queries (scripts/fuzz_traversal.py, tests): n rays, o and d as n*3 doubles ---- */

typedef struct {
    const orc_scene* s;
    uint64_t n;
    const double *o, *d;
    double tmin, tmax;
    int traversal, nthreads, index;
    double* t_out;
    int64_t* obj_out;
    double* worst;   /* per thread: 3 doubles (margin probe) */
} batch_job;

static void* intersect_batch_worker(void* arg) {
    batch_job* j = (batch_job*)arg;
    for (uint64_t i = (uint64_t)j->index; i < j->n; i += (uint64_t)j->nthreads) {
        double t = 0.0;
        j->obj_out[i] = orc_bvh_intersect(j->s, j->o + 3 * i, j->d + 3 * i, j->tmin, j->tmax, j->traversal, &t);
        j->t_out[i] = t;
    }
    return NULL;
}

/* orc_bvh_intersect for n rays on nthreads threads: t_out[i] and obj_out[i] (-1 = miss). */
int orc_bvh_intersect_batch(const orc_scene* s, uint64_t n, const double* o, const double* d, double tmin, double tmax,
                            int traversal, int nthreads, double* t_out, int64_t* obj_out) {
    if (!s || !s->built) return -2;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    batch_job jobs[256];
    pthread_t th[256];
    for (int k = 0; k < nthreads; k++) {
        batch_job j = {s, n, o, d, tmin, tmax, traversal, nthreads, k, t_out, obj_out, NULL};
        jobs[k] = j;
    }
    for (int k = 1; k < nthreads; k++) pthread_create(&th[k], NULL, intersect_batch_worker, &jobs[k]);
    intersect_batch_worker(&jobs[0]);
    for (int k = 1; k < nthreads; k++) pthread_join(th[k], NULL);
    return 0;
}

/* How far in front of the boxes around it can a primitive's computed t lie?  The traversal kernel skips a
 * slot whose box is entered beyond closest_t * (1 + 2^-10) (TRAV_CULL_MARGIN; the reference never culls), which
 * is exact as long as no accepted hit precedes the entry parameter of a box on its own root path by more than
 * that margin.  This probe walks the product's wide tree WITHOUT culling and, for every accepted hit t behind
 * boxes entered at e_1 <= ... (the slots on its path), records rel = (max e - t) / t:
 *   out[0] = largest rel seen (<= 0: every hit lies at or behind its boxes' entries)
 *   out[1] = number of accepted hits with rel > 0,  out[2] = number with rel > the margin in force */
static void margin_probe_rec(const orc_scene* s, ray_t ray, v3 inv, double tmin, double tmax, uint32_t ref,
                             double path_entry, double* out) {
    if ((ref >> 30) == REF_KIND_INTERIOR) {
        uint32_t rec = ref & 0x3fffffffu;
        const uint32_t* refs = s->wide_ref + (size_t)rec * 4;
        for (int c = 0; c < 4; c++) {
            uint32_t kind = refs[c] >> 30;
            double e;
            if (kind == REF_KIND_NONE) continue;
            if (kind != REF_KIND_SINGLE &&
                !aabb_intersect_entry(s->wide_box + ((size_t)rec * 4 + c) * 6, ray, inv, tmin, tmax, &e))
                continue;
            if (kind == REF_KIND_SINGLE) e = tmin;
            margin_probe_rec(s, ray, inv, tmin, tmax, refs[c], e > path_entry ? e : path_entry, out);
        }
        return;
    }
    uint32_t first = (ref & 0x3fffffffu) >> 2, count = (ref & 3u) + 1u;
    for (uint32_t k = 0; k < count; k++) {
        const shape_t* g = &s->objs[s->prim_object[first + k]].geom;
        double t;
        if (shape_intersect(g, ray, &t) && t > tmin && t < tmax) {
            double rel = (path_entry - t) / t;
            if (rel > out[0]) out[0] = rel;
            if (rel > 0.0) out[1] += 1.0;
            if (rel > g_cull_margin - 1.0) out[2] += 1.0;
        }
    }
}

static void* margin_probe_worker(void* arg) {
    batch_job* j = (batch_job*)arg;
    const orc_scene* s = j->s;
    for (uint64_t i = (uint64_t)j->index; i < j->n; i += (uint64_t)j->nthreads) {
        ray_t r = {v_from(j->o + 3 * i), v_from(j->d + 3 * i)};
        v3 inv = V(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
        double e;
        if (!aabb_intersect_entry(s->finfo.root_box, r, inv, j->tmin, j->tmax, &e)) continue;
        margin_probe_rec(s, r, inv, j->tmin, j->tmax, s->finfo.wide_root_ref, e, j->worst);
    }
    return NULL;
}

int orc_cull_margin_probe(const orc_scene* s, uint64_t n, const double* o, const double* d, double tmin, double tmax,
                          int nthreads, double out[3]) {
    if (!s || !s->built || !s->have_wide) return -2;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    batch_job jobs[256];
    pthread_t th[256];
    int started[256];
    double worst[256][3];  /* per call: concurrent probes do not share it */
    for (int k = 0; k < nthreads; k++) {
        worst[k][0] = -1.0, worst[k][1] = worst[k][2] = 0.0;
        batch_job j = {s, n, o, d, tmin, tmax, 2, nthreads, k, NULL, NULL, worst[k]};
        jobs[k] = j;
    }
    /* a worker that could not be started is run here: every share of the rays is probed */
    for (int k = 1; k < nthreads; k++) started[k] = pthread_create(&th[k], NULL, margin_probe_worker, &jobs[k]) == 0;
    margin_probe_worker(&jobs[0]);
    for (int k = 1; k < nthreads; k++) {
        if (started[k]) pthread_join(th[k], NULL);
        else margin_probe_worker(&jobs[k]);
    }
    out[0] = -1.0, out[1] = out[2] = 0.0;
    for (int k = 0; k < nthreads; k++) {
        if (worst[k][0] > out[0]) out[0] = worst[k][0];
        out[1] += worst[k][1], out[2] += worst[k][2];
    }
    return 0;
}

int orc_material_evaluate(const orc_material* m, const double position[3], const double normal[3],
                          const double view[3], uint64_t key, uint32_t* draw, double color[3], double dir[3]) {
    (void)position;
    mat_t mt;
    if (mat_from_desc(m, &mt) != 0) return -1;
    rng_t rng = {key, *draw};
    scat_t ev = mat_evaluate(&mt, v_from(normal), v_from(view), &rng);
    *draw = rng.draw;
    if (!ev.scatter) return 0;
    v_to(ev.color, color);
    v_to(ev.dir, dir);
    return 1;
}

void orc_background(const orc_scene* s, const double dir[3], double rgb[3]) { v_to(background(s, v_from(dir)), rgb); }

void orc_primary_ray(const orc_camera* c, uint32_t i, uint32_t j, uint64_t key, uint32_t* draw, double o[3],
                     double d[3]) {
    rng_t rng = {key, *draw};
    ray_t r = primary_ray(c, i, j, &rng);
    *draw = rng.draw;
    v_to(r.o, o);
    v_to(r.d, d);
}

/* The sample (pixel row, col of the image; sample index) exactly as orc_render starts it (path key, primary ray,
 * radiance), with its trace: returns the number of loop iterations; the first min(that, cap) entries of the arrays are
 * filled.  rgb: radiance()'s return value. */
uint32_t orc_path_trace(const orc_scene* s, const orc_camera* c, uint32_t row, uint32_t col, uint32_t sample, uint64_t seed,
                        uint32_t max_bounces, int traversal, uint32_t cap, int64_t* obj, double* t, double* thr,
                        uint32_t* draw, double rgb[3]) {
    rng_t rng;
    rng.key = rr_path_key(seed, (uint64_t)row * c->x_pixels + col, sample);
    rng.draw = 0;
    ray_t r = primary_ray(c, c->y_pixels - row, c->x_pixels - col, &rng); /* main.rs:74-75: camera origin is lower right */
    path_trace tr = {cap, 0, obj, t, thr, draw};
    v_to(radiance_traced(s, r, max_bounces, &rng, traversal, NULL, &tr), rgb);
    return tr.n;
}

uint32_t orc_radiance(const orc_scene* s, const double o[3], const double d[3], uint32_t max_bounces, uint64_t key,
                      uint32_t* draw, int traversal, double rgb[3]) {
    rng_t rng = {key, *draw};
    ray_t r = {v_from(o), v_from(d)};
    path_counters pc;
    memset(&pc, 0, sizeof(pc));
    v_to(radiance(s, r, max_bounces, &rng, traversal, &pc), rgb);
    *draw = rng.draw;
    return (uint32_t)pc.rays;
}

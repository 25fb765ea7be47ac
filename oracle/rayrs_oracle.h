/*
 * rayrs_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C, f64 restatement of the CPU
 * path of Frojdholm/rayrs (rayrs-lib + the block loop of rayrs/src/main.rs).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; the product library (rayrs_amd/csrc) never includes, links or
 * calls anything in this directory.
 *
 * PARITY STATUS: the reference is Rust-only (no rustc/cargo in this image),
 * non-deterministic (OS-seeded rand 0.7.3) and has no numeric test for
 * radiance or materials, so those parts are "parity unpinned".  What the
 * reference's own tests do pin (vecmath.rs:812-893, geometry.rs:735-888,
 * bvh.rs:543-559 and the doc-test scalars) is re-expressed against this
 * oracle in tests/test_oracle_reference_tests.py.
 */
#ifndef RAYRS_ORACLE_H
#define RAYRS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* material.rs:57-68 */
enum {
    ORC_MAT_LAMBERTIAN = 0,
    ORC_MAT_REFLECT = 1,
    ORC_MAT_REFRACT = 2,
    ORC_MAT_GLASS = 3,
    ORC_MAT_COOK_TORRANCE = 4,
    ORC_MAT_COOK_TORRANCE_REFRACT = 5,
    ORC_MAT_COOK_TORRANCE_GLASS = 6,
    ORC_MAT_PLASTIC = 7,
    ORC_MAT_NO_REFLECT = 8
};

/* geometry.rs:161-168 */
enum { ORC_AXIS_X = 0, ORC_AXIS_XREV = 1, ORC_AXIS_Y = 2, ORC_AXIS_YREV = 3, ORC_AXIS_Z = 4, ORC_AXIS_ZREV = 5 };

enum { ORC_SHAPE_SPHERE = 0, ORC_SHAPE_PLANE = 1, ORC_SHAPE_TRIANGLE = 2 };

typedef struct {
    int32_t kind;         /* ORC_MAT_* */
    int32_t metallic;     /* CookTorrance only: 1 = Fresnel::SchlickMetallic(r0), 0 = SchlickDielectric(ior) */
    double color[3];      /* diffuse / base colour */
    double spec_color[3]; /* Plastic: colour of the CookTorrance layer */
    double alpha;         /* roughness (the ctor squares it, material.rs:711) */
    double ior;
    double r0[3];         /* SchlickMetallic reflectance */
} orc_material;

typedef struct {
    int32_t emissive; /* 0 = Emission::Dark */
    int32_t pad;
    double strength;
    double color[3];
} orc_emission;

/* lib.rs:56-67 after Camera::new */
typedef struct {
    double origin[3];
    double e_x[3];
    double e_y[3];
    double z[3];
    double width, height;
    uint32_t ppc;
    uint32_t x_pixels, y_pixels;
} orc_camera;

typedef struct {
    uint64_t rays;          /* BVH queries = radiance loop iterations (lib.rs:525-526) */
    uint64_t paths;
    uint64_t nan_pixels;    /* main.rs:81-83 */
    uint64_t neg_pixels;    /* main.rs:85-87 */
    /* counters of the ordered (kernel-style) traversal, zero in reference mode */
    uint64_t interior_visits;
    uint64_t tri_tests;
    uint64_t sphere_tests;
    uint64_t plane_tests;
    uint64_t escaped_paths; /* paths that ended in Scene::background */
    double seconds;
} orc_stats;

typedef struct orc_scene orc_scene;

/* 0 = portable functions of include/rayrs_numeric.h (bit-identical to the
 * kernel), 1 = the platform libm (what the Rust reference would call). */
void orc_set_math_mode(int libm);
int orc_get_math_mode(void);

orc_scene* orc_scene_create(void);
void orc_scene_destroy(orc_scene* s);

/* Object ctors, lib.rs:321-415.  Return 0 or a negative error for the
 * parameter checks the reference asserts. */
int orc_add_sphere(orc_scene* s, double radius, const double origin[3], const orc_material* m, const orc_emission* e);
int orc_add_plane(orc_scene* s, int axis, double umin, double umax, double vmin, double vmax, double pos,
                  const orc_material* m, const orc_emission* e);
int orc_add_triangle(orc_scene* s, const double p1[3], const double p2[3], const double p3[3], const orc_material* m,
                     const orc_emission* e);
int orc_add_triangles(orc_scene* s, const double* verts, uint32_t nverts, const uint32_t* idx, uint32_t ntris,
                      const orc_material* m, const orc_emission* e);

/* Scene::new, lib.rs:227-245.  heuristic: 0 = Midpoint, 1 = Sah{splits}.
 * builder: 0 = literal restatement of bvh.rs:227-317 (O(splits*N) per node),
 *          1 = same decisions via sorted prefix/suffix boxes (O(N log N) per
 *              node), identical tree (tests/test_oracle_bvh.py).
 * hdri: hdri_w*hdri_h RGB f32 texels, row-major; clipped to [0,3] as
 * main.rs:43 does. */
int orc_scene_build(orc_scene* s, double z_near, double z_far, int heuristic, uint32_t splits, int builder,
                    uint32_t hdri_w, uint32_t hdri_h, const float* hdri_rgb);

/* Camera::new, lib.rs:99-133 */
int orc_camera_new(const double origin[3], const double up[3], const double lookat[3], double fov, double width,
                   double height, uint32_t ppi, orc_camera* out);

/* The block loop of main.rs:57-101 with the build-defined RNG.
 * traversal: 0 = recursive reference traversal (bvh.rs:391-415),
 *            1 = ordered traversal with closest-hit culling on the two-child records,
 *            2 = the same on the four-slot records of the product's walk tree, which must have
 *                been handed over with orc_set_wide (the kernel's walk on the kernel's data:
 *                same visit order, so its counters equal the kernel's).
 * sample_chunk: 0 or >= spp = one sequential sum per pixel (main.rs:67-79);
 * otherwise per-chunk sums added in chunk order (see include/rayrs_hip.h).
 * rows [row0,row1) of the image are rendered (whole image: 0,height); the
 * rest of out_rgb is left untouched.  out_rgb: height*width*3 f64. */
int orc_render(const orc_scene* s, const orc_camera* c, uint32_t spp, uint32_t max_bounces, uint64_t seed,
               uint32_t sample_chunk, uint32_t row0, uint32_t row1, int nthreads, int traversal, double* out_rgb,
               orc_stats* stats);

/* ---- unit-level entry points for the known-answer and parity tests ---- */

double orc_math(int fn, double x, double y); /* 0 sin 1 cos 2 tan 3 log 4 exp 5 acos 6 atan2(x=y_arg,y=x_arg) 7 sqrt */
uint64_t orc_rng_bits(uint64_t seed, uint64_t pixel, uint64_t sample, uint32_t draw);

/* vecmath.rs helpers (see rayrs_oracle.c for the op codes) */
void orc_vec_op(int op, const double a[3], const double b[3], double s, double t, double out[3]);
double orc_vec_scalar(int op, const double a[3], const double b[3]);
void orc_orthonormal_basis(const double n[3], double e1[3], double e2[3]);
/* AxisAlignedBoundingBox::expand (geometry.rs:674-683) of two boxes */
void orc_aabb_expand(const double a[6], const double b[6], double out[6]);

int orc_aabb_intersect(const double box[6] /* xmin,xmax,ymin,ymax,zmin,zmax */, const double o[3], const double d[3],
                       double tmin, double tmax);
/* returns 1 and *t if Some(t) */
int orc_sphere_intersect(double radius, const double c[3], const double o[3], const double d[3], double* t);
int orc_plane_intersect(int axis, double umin, double umax, double vmin, double vmax, double pos, const double o[3],
                        const double d[3], double* t);
int orc_triangle_intersect(const double p1[3], const double p2[3], const double p3[3], const double o[3],
                           const double d[3], double* t);
void orc_triangle_normal(const double p1[3], const double p2[3], const double p3[3], double n[3]);

/* AxisAlignedBoundingBox helpers for the doc-test scalars, geometry.rs:544-683.
 * Fills box[6], center[3], volume, surface_area of the scene's object list. */
int orc_scene_bbox(const orc_scene* s, double box[6], double center[3], double* volume, double* surface_area);
int orc_object_boxes(const orc_scene* s, double* boxes); /* nobjs * 6, insertion order */

/* Bvh::intersect; returns object index (insertion order) or -1, *t. */
int64_t orc_bvh_intersect(const orc_scene* s, const double o[3], const double d[3], double tmin, double tmax,
                          int traversal, double* t);

/* Material::evaluate with draws taken from (key, *draw); returns 1 for
 * Scatter (color, dir filled) or 0 for NoScatter. */
int orc_material_evaluate(const orc_material* m, const double position[3], const double normal[3],
                          const double view[3], uint64_t key, uint32_t* draw, double color[3], double dir[3]);

void orc_background(const orc_scene* s, const double dir[3], double rgb[3]);
void orc_primary_ray(const orc_camera* c, uint32_t i, uint32_t j, uint64_t key, uint32_t* draw, double o[3],
                     double d[3]);
/* One sample as orc_render starts it (path key from (seed, pixel, sample), primary ray, radiance) with its trace: per
 * iteration of radiance()'s loop the object found (-1 none), t (0 for a miss), the throughput and the RNG draw index
 * on leaving the iteration.  Returns the number of iterations; min(that, cap) entries are written. */
uint32_t orc_path_trace(const orc_scene* s, const orc_camera* c, uint32_t row, uint32_t col, uint32_t sample, uint64_t seed,
                        uint32_t max_bounces, int traversal, uint32_t cap, int64_t* obj, double* t, double* thr,
                        uint32_t* draw, double rgb[3]);
/* radiance() for one path; returns number of BVH queries. */
uint32_t orc_radiance(const orc_scene* s, const double o[3], const double d[3], uint32_t max_bounces, uint64_t key,
                      uint32_t* draw, int traversal, double rgb[3]);

/* ---- flattened tree, for comparing with the product's builder ---- */
typedef struct {
    uint32_t n_interior;
    uint32_t n_prims;
    uint32_t root_ref;
    uint32_t depth; /* max number of stack entries the ordered traversal can need */
    double root_box[6];
    uint32_t n_wide;        /* four-slot records of the product's walk tree (0 until orc_set_wide) */
    uint32_t wide_root_ref;
    uint32_t wide_depth;    /* stack entries the wide traversal can need */
    uint32_t reserved;
} orc_flat_info;

/* ref = kind << 30 | payload; kind 0 interior (payload = record index),
 * 1 = leaf range with box test (payload = first_prim << 2 | count-1),
 * 2 = single primitive, no box test (payload = prim << 2). */
int orc_flatten_info(const orc_scene* s, orc_flat_info* info);
/* wide_box: n_wide*4*6 f64, wide_ref: n_wide*4 u32 (kind 3 = unused slot) */
/* diagnostics for sizing the kernel's LDS: per-record visit counts and stack heights of the
 * wide traversal (hist: n_wide + 64 counters, NULL switches it off; single-threaded renders only) */
void orc_set_visit_histogram(uint64_t* hist);
int orc_flatten_export_wide(const orc_scene* s, double* wide_box, uint32_t* wide_ref);
/* Hands the oracle the four-slot records of the product's walk tree (rayrs_scene_export_wide):
 * traversal 2 walks them.  The oracle does not build that tree itself. */
int orc_set_wide(orc_scene* s, uint32_t n_wide, uint32_t wide_root_ref, uint32_t wide_depth, const double* wide_box,
                 const uint32_t* wide_ref);
/* The product's hot group (rayrs_scene_info_t.hot_*), which the records of rayrs_scene_export_hot_tree leave out:
 * traversal 2 then tests it beside the walk, as the default walk's kernels do.  After orc_set_wide, which forgets it. */
int orc_set_hot_group(orc_scene* s, const double box[6], uint32_t first, uint32_t count);
/* child_box: n_interior*2*6 f64; child_ref: n_interior*2; prim_object: n_prims
 * (object index, insertion order, of the DFS-ordered primitives). */
int orc_flatten_export(const orc_scene* s, double* child_box, uint32_t* child_ref, uint32_t* prim_object);

/* orc_bvh_intersect for n rays (o, d: n*3 doubles) on nthreads threads: t_out[i], obj_out[i] (-1 = miss). */
int orc_bvh_intersect_batch(const orc_scene* s, uint64_t n, const double* o, const double* d, double tmin, double tmax,
                            int traversal, int nthreads, double* t_out, int64_t* obj_out);
/* How far in front of the boxes around it a primitive's computed t lies, over n rays on the product's walk tree
 * (orc_set_wide), walked without culling: out[0] = largest (entry - t) / t over accepted hits and the boxes on
 * their root paths, out[1] = hits with a positive value, out[2] = hits beyond the cull margin in force (2^-10). */
/* experiments only: the relative margin of traversal 1 / 2's closest-hit culling (default: the kernel's) */
void orc_set_cull_margin(double rel);
int orc_cull_margin_probe(const orc_scene* s, uint64_t n, const double* o, const double* d, double tmin, double tmax,
                          int nthreads, double out[3]);

#ifdef __cplusplus
}
#endif
#endif

/*
 * rayrs_hip.h -- C ABI of the MI355X (gfx950) build of rayrs-lib's per-pixel
 * radiance integrator.
 *
 * The reference (Frojdholm/rayrs) has no FFI: its finest seam is the Rust
 * call pair Camera::generate_primary_ray (rayrs-lib/src/lib.rs:202) +
 * rayrs_lib::radiance (lib.rs:521) inside the rayon block loop of
 * rayrs/src/main.rs:57-101.  A per-ray FFI is useless for a GPU, so this
 * boundary is per image: rayrs_render replaces main.rs:57-101 wholesale, and
 * because Scene/Object/Camera keep their fields private (lib.rs:56-67,
 * :216-220, :302-306) the constructors are part of the boundary too.  Each
 * entry point names the reference item it stands for; INTEGRATION.md shows
 * the Rust `extern "C"` block a maintainer would add to rayrs-lib.
 *
 * Conventions: every function returns 0 (RAYRS_OK) or a negative
 * rayrs_status; nothing unwinds or aborts across the boundary.  Parameter
 * checks are the reference's assert!s turned into RAYRS_INVALID_ARG.  Plain
 * pointers and sizes only; device pointers are passed as void*.
 */
#ifndef RAYRS_HIP_H
#define RAYRS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RAYRS_OK = 0,
    RAYRS_INVALID_ARG = -1, /* a reference assert! would have fired */
    RAYRS_HIP_ERROR = -2,   /* a HIP runtime call failed (see rayrs_last_error) */
    RAYRS_OOM = -3,
    RAYRS_NO_DEVICE = -4,   /* scene was created host-only or no GPU present */
    RAYRS_UNSUPPORTED = -5, /* more than 2^32 (pixel, sample chunk) items on a rank; max_bounces > 8000 (8000 itself is
                               accepted: bounce count and draw index travel as 16 bits, a bounce draws at most four
                               numbers); an image side > 65535; a walk tree that needs more than 4096 stack entries */
    RAYRS_IO_ERROR = -6,    /* file missing or malformed (see rayrs_io_last_error) */
    RAYRS_RCCL_ERROR = -7   /* the RCCL library could not be loaded, or one of its calls failed (rayrs_render_multi) */
} rayrs_status;

const char* rayrs_strerror(int status);
/* Text of the last HIP error seen by the calling thread ("" if none). */
const char* rayrs_last_error(void);

/* ---- material.rs ---- */

/* enum Material, material.rs:57-68 */
enum {
    RAYRS_MAT_LAMBERTIAN = 0,            /* LambertianDiffuse::new(color)              :608 */
    RAYRS_MAT_REFLECT = 1,               /* Reflect::new(color)                        :629 */
    RAYRS_MAT_REFRACT = 2,               /* Refract::new(color, ior)                   :650 */
    RAYRS_MAT_GLASS = 3,                 /* Glass::new(color, ior)                     :673 */
    RAYRS_MAT_COOK_TORRANCE = 4,         /* CookTorrance::new(color, alpha, fresnel)   :705 */
    RAYRS_MAT_COOK_TORRANCE_REFRACT = 5, /* CookTorranceRefract::new(color, alpha, ior):832 */
    RAYRS_MAT_COOK_TORRANCE_GLASS = 6,   /* CookTorranceGlass::new(color, alpha, ior)  :863 */
    RAYRS_MAT_PLASTIC = 7,               /* Plastic::new(color, spec_color, alpha, ior):887 */
    RAYRS_MAT_NO_REFLECT = 8             /* Material::NoReflect                            */
};

typedef struct {
    int32_t kind;         /* RAYRS_MAT_* */
    int32_t metallic;     /* CookTorrance: 1 = Fresnel::SchlickMetallic(r0), 0 = SchlickDielectric(ior) (:128-131) */
    double color[3];
    double spec_color[3]; /* Plastic only */
    double alpha;         /* roughness; stored squared like the reference ctor (:711) */
    double ior;
    double r0[3];
} rayrs_material;

/* enum Emission, material.rs:1056-1075; emissive = 0 is Emission::Dark */
typedef struct {
    int32_t emissive;
    int32_t pad;
    double strength;
    double color[3];
} rayrs_emission;

/* enum Axis, geometry.rs:161-168 */
enum { RAYRS_AXIS_X = 0, RAYRS_AXIS_XREV = 1, RAYRS_AXIS_Y = 2, RAYRS_AXIS_YREV = 3, RAYRS_AXIS_Z = 4, RAYRS_AXIS_ZREV = 5 };

/* enum BvhHeuristic, bvh.rs:187-191 */
enum { RAYRS_BVH_MIDPOINT = 0, RAYRS_BVH_SAH = 1 };

/* ---- Vec<Object>: lib.rs:302-512 ---- */

typedef struct rayrs_objects rayrs_objects;

int rayrs_objects_create(rayrs_objects** out);
void rayrs_objects_destroy(rayrs_objects* objs);
uint64_t rayrs_objects_len(const rayrs_objects* objs);

/* Object::sphere(radius, origin, mat, emission)                     lib.rs:321 */
int rayrs_object_sphere(rayrs_objects* objs, double radius, const double origin[3], const rayrs_material* mat,
                        const rayrs_emission* emission);
/* Object::plane(axis, umin, umax, vmin, vmax, pos, mat, emission)   lib.rs:342 */
int rayrs_object_plane(rayrs_objects* objs, int axis, double umin, double umax, double vmin, double vmax, double pos,
                       const rayrs_material* mat, const rayrs_emission* emission);
/* Object::triangle(p1, p2, p3, mat, emission)                       lib.rs:380 */
int rayrs_object_triangle(rayrs_objects* objs, const double p1[3], const double p2[3], const double p3[3],
                          const rayrs_material* mat, const rayrs_emission* emission);
/* Object::from_triangles(tris, mat, emission)                       lib.rs:407
 * as an indexed mesh: verts = nverts*3 coordinates, idx = ntris*3 vertex
 * indices (0-based, counter-clockwise).  The f32 form is what a PLY file
 * holds; the values are widened to f64 exactly. */
int rayrs_object_from_triangles_f32(rayrs_objects* objs, const float* verts, uint32_t nverts, const uint32_t* idx,
                                    uint32_t ntris, const rayrs_material* mat, const rayrs_emission* emission);
int rayrs_object_from_triangles_f64(rayrs_objects* objs, const double* verts, uint32_t nverts, const uint32_t* idx,
                                    uint32_t ntris, const rayrs_material* mat, const rayrs_emission* emission);
/* Object::from_spheres(spheres, mat, emission)                      lib.rs:422
 * centers = n*3 */
int rayrs_object_from_spheres(rayrs_objects* objs, double radius, const double* centers, uint32_t n,
                              const rayrs_material* mat, const rayrs_emission* emission);
/* Object::box_geom(lower_left, upper_right, mat, emission)          lib.rs:438 */
int rayrs_object_box_geom(rayrs_objects* objs, const double lower_left[3], const double upper_right[3],
                          const rayrs_material* mat, const rayrs_emission* emission);

/* ---- Scene: lib.rs:216-296 ---- */

typedef struct rayrs_scene rayrs_scene;

/* Scene::new(objects, z_near, z_far, heuristic, hdri)               lib.rs:227
 * Consumes nothing: `objs` stays owned by the caller and may be destroyed
 * right after.  Builds the BVH exactly as Bvh::build does (bvh.rs:199-389,
 * same splits, same child order), derives the tree the kernels walk from it
 * (rayrs_scene_export_wide) and uploads that to HIP device `device`.  device = -1 builds a host-only scene (no GPU needed) that can
 * be inspected with rayrs_scene_info / rayrs_scene_export_bvh but not
 * rendered.  hdri_rgb: hdri_w*hdri_h RGB f32 texels, row-major, image origin
 * upper left; values are clipped to [0, 3] as main.rs:43 does. */
int rayrs_scene_new(const rayrs_objects* objs, double z_near, double z_far, int heuristic, uint32_t splits,
                    uint32_t hdri_w, uint32_t hdri_h, const float* hdri_rgb, int device, rayrs_scene** out);
void rayrs_scene_destroy(rayrs_scene* scene);

typedef struct {
    uint64_t n_objects;
    uint32_t n_interior;   /* two-child records */
    uint32_t n_prims;      /* leaf primitives in DFS order (== n_objects) */
    uint32_t root_ref;
    uint32_t depth;        /* levels of interior records */
    uint32_t compact;      /* 1 = f32 node boxes / f32 triangle vertices (exactly representable) */
    uint32_t n_surfaces;
    uint32_t node_bytes;   /* bytes of one wide interior record on the device */
    uint32_t prim_bytes;   /* bytes of one primitive record on the device */
    uint64_t device_bytes; /* total scene footprint in HBM */
    double root_box[6];    /* xmin,xmax,ymin,ymax,zmin,zmax */
    double build_seconds;
    uint32_t n_wide;        /* four-slot records of the tree the fast walk reads (rayrs_scene_export_wide) */
    uint32_t wide_root_ref;
    uint32_t wide_depth;    /* stack entries the traversal can need */
    uint32_t local_pool;    /* 1 = the gate tree is at most one record: renders keep every path in LDS
                               (rayrs_tuning.local_pool, local_pool.hip) */
    uint32_t gate_n_wide;   /* the same three for the gate tree (rayrs_scene_export_gate_tree), which */
    uint32_t gate_root_ref; /* the default walk reads */
    uint32_t gate_depth;
    /* The default walk's HOT GROUP (hot_count = 0: the scene has none): one group of the gate tree -- the one whose
     * gating box is the largest, if it covers at least a quarter of the root Node's box: on the reference's obj scenes the
     * floor's bottom Node, which nine rays in ten enter -- is left out of the records the default walk reads
     * (rayrs_scene_export_hot_tree: hot_n_wide records) and tested ONCE PER RAY outside that tree, by the kernel that
     * makes the ray, for a whole batch of rays together: its gating box hot_box exactly as
     * AxisAlignedBoundingBox::intersect tests it, then its hot_count primitives hot_first ... in depth-first order
     * exactly as the reference tests them -- on wave-uniform f64 values.  The groups of that tree and
     * the hot group together are the groups of the gate tree, each behind its gating box: the primitives tested are
     * still exactly those BvhTree::intersect tests (tests/test_bvh_builder.py checks it from the exports). */
    uint32_t hot_n_wide;
    uint32_t hot_root_ref;
    uint32_t hot_depth;
    uint32_t hot_first;
    uint32_t hot_count;
    uint32_t hot_pad;
    double hot_box[6];
} rayrs_scene_info_t;

int rayrs_scene_info(const rayrs_scene* scene, rayrs_scene_info_t* info);
/* The same scene (BVH, records, materials, HDRI, tuning) uploaded to another HIP device without
 * building it again: what rayrs_render_multi wants one of per GPU.  `scene` may be host-only. */
int rayrs_scene_clone_to_device(const rayrs_scene* scene, int device, rayrs_scene** out);
/* HIP device of the scene, -1 for a host-only scene. */
int rayrs_scene_device(const rayrs_scene* scene);
/* child_box: n_interior*2*6 f64, child_ref: n_interior*2 u32, prim_object:
 * n_prims u32 (index of the object, in insertion order, at each DFS slot).
 * ref = kind<<30 | payload; kind 0 = interior record (payload = index),
 * 1 = 1..4 primitives behind a box test (payload = first<<2 | count-1),
 * 2 = one primitive with no box test (payload = prim<<2). */
int rayrs_scene_export_bvh(const rayrs_scene* scene, double* child_box, uint32_t* child_ref, uint32_t* prim_object);
/* The records the kernels walk (wide_box: n*4*6 f64, wide_ref: n*4 u32, kind 3 = unused slot): NOT the
 * reference's topology but trees built for traversal speed.  What decides whether BvhTree::intersect reaches a
 * primitive is one box, its GATING box -- the box of the Node it hangs under; boxes on a root path nest exactly and
 * the slab test is monotone in the bounds, so passing it implies passing every box above.
 *   rayrs_scene_export_gate_tree (n = gate_n_wide): a leaf slot (kind 1) is one of the reference's groups -- the
 * 1..4 leaves that share a parent Node, contiguous in depth-first order -- behind exactly its gating box; an
 * interior slot (kind 0) carries the union of the boxes below it, so a ray that misses it misses every gating box
 * inside.  Every group appears exactly once: the primitives this tree reaches are the primitives the reference
 * reaches.  Walked by default, and the source of the local-pool route's gates.
 *   rayrs_scene_export_wide (n = n_wide), what rayrs_render_params.fast_traversal walks: a leaf slot is ONE primitive behind its own bounding box
 * (Bvh::build's, geometry.rs bbox) widened on every side by 1/64 of its largest extent, rounded outwards to f32 and
 * clipped to its gating box.  It reaches a subset of what the reference reaches (see fast_traversal for what the
 * subset leaves out: a bet).
 * tests/test_bvh_builder.py checks all of this from the exports alone. */
int rayrs_scene_export_wide(const rayrs_scene* scene, double* wide_box, uint32_t* wide_ref);
int rayrs_scene_export_gate_tree(const rayrs_scene* scene, double* wide_box, uint32_t* wide_ref);
/* The gate tree without the hot group (rayrs_scene_info_t.hot_*; n = hot_n_wide): what the default walk reads on a scene
 * that has one.  RAYRS_INVALID_ARG on a scene without a hot group. */
int rayrs_scene_export_hot_tree(const rayrs_scene* scene, double* wide_box, uint32_t* wide_ref);

/* ---- Camera: lib.rs:54-211 ---- */

typedef struct {
    double origin[3];
    double e_x[3];
    double e_y[3];
    double z[3];
    double width, height;
    uint32_t ppc;
    uint32_t x_pixels; /* Camera::x_pixels lib.rs:153 */
    uint32_t y_pixels; /* Camera::y_pixels lib.rs:175 */
} rayrs_camera;

/* Camera::new(origin, up, lookat, fov, width, height, ppi)          lib.rs:99 */
int rayrs_camera_new(const double origin[3], const double up[3], const double lookat[3], double fov, double width,
                     double height, uint32_t ppi, rayrs_camera* out);

/* ---- render: the block loop of rayrs/src/main.rs:57-101 ---- */

enum { RAYRS_OUT_F32 = 0, RAYRS_OUT_F64 = 1 };

typedef struct {
    uint32_t spp;          /* main.rs:68 */
    uint32_t max_bounces;  /* main.rs:77 passes 50 */
    uint64_t seed;         /* build-defined RNG, include/rayrs_numeric.h */
    /* Pixels are summed per pixel in chunks of `sample_chunk` consecutive
     * samples; chunk sums are added in chunk order.  0 or >= spp reproduces
     * the reference's single sequential sum (main.rs:67-79). */
    uint32_t sample_chunk;
    /* Image tiles (8x8 pixels, row-major tile index t) with
     * t % tile_ranks == tile_rank are rendered; other pixels of `out` are not
     * written.  1-GPU: tile_rank 0, tile_ranks 1. */
    uint32_t tile_rank, tile_ranks;
    uint32_t out_format;   /* RAYRS_OUT_F32: f32x3 (image.rs:224-229), RAYRS_OUT_F64: f64x3 */
    uint32_t count_work;   /* 1 = also count traversal work (slower; for the roofline figure) */
    /* Which walk answers the BVH queries.  BvhTree::intersect tests every primitive whose enclosing Node boxes the
     * ray enters and never compares a box with the closest hit so far (bvh.rs:391-415).
     * 0 (default): the walk over the reference's leaf groups behind their exact gating boxes
     *   (rayrs_scene_export_gate_tree) with nothing culled: it tests exactly the primitives the reference tests, so
     *   its closest hit (smallest accepted t, first primitive in depth-first order on ties, bvh.rs:62) is the
     *   reference's for EVERY ray BY CONSTRUCTION.  The local-pool route walks this way too.  Which kernel makes a
     *   test, and when, is the library's business: on a scene with a hot group (rayrs_scene_info_t.hot_count) the root
     *   box, that group and the first record of the tree without it are tested by the kernel that makes the ray, the
     *   rest by the traversal kernel -- the same tests on the same values, the same set of primitives.
     * 1: the fast walk, which makes two bets on the reference's arithmetic, each measured, neither a construction
     *   (it was the default until round 5; the headline frame renders 10 % faster with it -- 23 % before round 6 moved
     *   the default walk's cheap tests out of the traversal kernel, profiles/r05_walks.txt, r06_final_bench.json):
     *   - closest-hit culling: a box entered beyond best_t * (1 + 2^-10) is skipped -- the reference's answer
     *     unless a primitive's COMPUTED t lies more than that in front of a box around it;
     *   - tight leaf boxes (rayrs_scene_export_wide): a primitive is tested only if the ray enters its own bounding
     *     box widened by 1/64 of its size (inside the reference's gating box, so nothing extra is ever tested) --
     *     the reference's answer unless its own test accepts a hit on a primitive the ray passes beside by more
     *     than that.
     *   Both fail only for rays aimed nearly IN a primitive's plane, where Moeller-Trumbore's own result is rounding
     *   noise, and the failure envelope below is MEASURED, not proven (scripts/fuzz_traversal.py, 4 * 10^7 rays on sliver
     *   meshes and nearly flat sheets, profiles/r05_fuzz_traversal.txt; "near" = from within the camera rule below):
     *   - in-plane rays at 1e-7 rad and MORE off the plane, from nearby: 17 wrong hits in 3.97 M such rays, all on finely
     *     tessellated sheets of SLIVER quads (250-400 per side, aspect ratios up to 10^6: the culling bet fails there at
     *     any distance; 1 of the 17 is lost by the leaf boxes alone); none on the coarser sheets;
     *   - closer than 1e-7 rad to the plane, from nearby: 52 in 5.96 M (31 of them by the leaf boxes alone);
     *   - from far away (beyond the camera rule) the leaf boxes fail for about one in-plane ray in 10^4 at 1e-9 rad and
     *     more, seven in 10^4 closer to the plane.
     *   tests/test_walk_tree.py pins one failing ray of each kind, on which the default walk returns the reference's
     *   primitive.  No rendered frame, of any size, has differed in a bit -- pixels are not aimed along triangle planes.
     *   The camera rule: a frame whose camera stands farther from the scene's bounding box than 8 times that box's
     *   diagonal, or than 4096 times the scene's small-primitive extent (the 5th percentile of the primitives' largest
     *   extents), takes the default walk whatever it asked for (rayrs_render_stats.exact_walk reports the walk a frame
     *   took).  The rule looks at the camera, so it guards PRIMARY rays only: a bounced ray that travels from a large
     *   surface to a finely tessellated one covers thousands of small-primitive sizes and makes the leaf-box bet in the
     *   regime where it was seen to fail; and on a finely tessellated scene most real cameras count as far, so the field
     *   is then ignored (visible only in exact_walk).
     * What a sound walk costs, and what certificates (a forward error bound of Moeller-Trumbore deciding where "box
     * missed" provably means "rejected") could and could not buy back: profiles/r05_certified_walks.txt, DESIGN.md 2.
     * Zero-initialise the struct: values above 1 are refused (RAYRS_INVALID_ARG).  (Until round 5 this field was
     * `exact_traversal` with the opposite sense.) */
    uint32_t fast_traversal;
} rayrs_render_params;

typedef struct {
    uint64_t rays;          /* BVH queries = radiance loop iterations, lib.rs:525-526 */
    uint64_t paths;
    uint64_t nan_pixels;    /* main.rs:81-83 */
    uint64_t neg_pixels;    /* main.rs:85-87 */
    uint64_t interior_visits; /* the next five only with count_work */
    uint64_t tri_tests;
    uint64_t sphere_tests;
    uint64_t plane_tests;
    uint64_t escaped_paths;
    /* lane-utilisation diagnostics (count_work only); each *_wave value is summed over
     * all 64 lanes of the waves that executed the phase, *_lane over the active lanes */
    uint64_t step_wave, step_lane, inner_wave, leaf_wave;
    /* shader-clock ticks the traversal kernel's waves spent in interior steps, in leaf steps
     * and in retiring/refilling (count_work only), summed over waves */
    uint64_t interior_ticks, leaf_ticks;
    double kernel_ms;       /* summed HIP-event time of the traversal kernel's launches on its stream (the local-pool
                               kernel's when local_pool is set) */
    double total_ms;        /* all kernels of the render: path rounds + resolve */
    uint64_t kernel_launches; /* launches of the traversal kernel (= path rounds) */
    double trace_ms;        /* all path rounds (gen + traversal + hit + miss kernels) */
    uint64_t refill_ticks;
    /* closest hits by surface (count_work only): surface_hits[k] = BVH queries whose closest hit carries
     * the k-th distinct (Material, Emission) pair in object insertion order, pairs 7 and up together;
     * rays - sum(surface_hits) = queries that found nothing (Scene::background) */
    uint64_t surface_hits[8];
    /* primary rays that missed the root Node's box (bvh.rs:394: a Miss before anything else is looked at): their
     * samples are finished by the kernel that made the ray, without a trip through the traversal and miss
     * kernels.  They are part of `rays` and of `escaped_paths`. */
    uint64_t direct_rays;
    double hit_ms, miss_ms; /* summed HIP-event times of the hit and the miss kernel's launches (kernel_ms: the traversal
                               kernel's, or the local-pool kernel's, which is then the only one) */
    uint32_t local_pool;    /* 1 = this frame was rendered by the local-pool kernel (rayrs_tuning.local_pool) */
    uint32_t exact_walk;    /* 1 = this frame's queries were answered by the reference's visit set (the default); 0 = by the
                               fast walk (rayrs_render_params.fast_traversal = 1, the camera near enough, the streaming
                               route: the local-pool route never makes the bets) */
    uint32_t hot_group;     /* 1 = ... with the scene's hot group (rayrs_scene_info_t.hot_count) and the first record of the walk tree
                               tested by the kernels that MAKE the rays, for whole batches at once: a ray whose query ends
                               there never travels through the traversal kernel */
    uint32_t stats_pad;
    uint64_t pre_rays;        /* the next five only with count_work: queries (of `rays`) that were answered by the kernel that made
                                 the ray -- bounced rays that miss the root box, rays that enter no slot of the walk tree's first record */
    uint64_t pre_root_records;/* ... of them, the rays that entered the root box: each is one visit of the walk tree's first record
                                 (counted in interior_visits) that ended the query */
    uint64_t hot_lane;        /* rays put to the hot group's gating box */
    uint64_t hot_prim_tests;  /* of tri_tests + sphere_tests + plane_tests, the hot group's */
    uint64_t hot_tri_divided; /* ... of its triangle tests, those that went on to the three divisions (the others were settled
                                 before them by two exact facts about IEEE division: device_path.h hot_triangle_intersect) */
} rayrs_render_stats;

/* The sample chunk a frame is rendered with when the caller has no reason to choose another:
 * the smallest `requested` * 2^k (requested = 0 means 4) for which the WHOLE frame -- all of its
 * 8x8 tiles, whoever renders them -- has at most 2^30 (pixel, chunk) items.  It depends on the
 * frame only, never on the number of GPUs that share it, so that every rank count sums a pixel's
 * samples in the same order and produces the same bits.  (Each rank keeps 24 bytes of partial
 * sum per item; a rank refuses more than 2^32 items.) */
uint32_t rayrs_frame_sample_chunk(uint32_t x_pixels, uint32_t y_pixels, uint32_t spp, uint32_t requested);

/* Renders into a HOST buffer of y_pixels*x_pixels*3 elements (row-major,
 * origin upper left, RGB).  Synchronous. */
int rayrs_render(rayrs_scene* scene, const rayrs_camera* camera, const rayrs_render_params* params, void* out_host,
                 rayrs_render_stats* stats);

/* Enqueues the render on `hip_stream` (a hipStream_t, NULL = default stream) writing a DEVICE
 * buffer of the same shape.  The number of path rounds is data dependent (a frame ends when the
 * last path does), so rounds are enqueued in batches and the call reads the live-path count of
 * batch b back while batch b+1 is already queued: it returns once the LAST batch is enqueued --
 * the GPU is never idle waiting for the host, but the call itself lasts about as long as the
 * frame minus its last batch.  The resolve pass and everything the caller enqueues on the stream
 * afterwards (an RCCL reduce, a copy) is ordered behind the render without a host wait. */
int rayrs_render_launch(rayrs_scene* scene, const rayrs_camera* camera, const rayrs_render_params* params,
                        void* out_device, void* hip_stream);
/* Waits for the last rayrs_render_launch on this scene and returns its counters. */
int rayrs_render_finish(rayrs_scene* scene, rayrs_render_stats* stats);

/* The same block loop over several GPUs of one node, inside the library (the reference does all of
 * its parallelism -- rayon over image blocks, main.rs:57-101 -- inside the product too).  scenes[i]
 * are n handles of the same scene on the devices to use (rayrs_scene_clone_to_device); rank i renders
 * the 8x8 tiles t with t % n == i from its own host thread on its own stream into a zeroed
 * framebuffer on its device, and one RCCL reduce (sum, over xGMI) to scenes[0]'s device assembles
 * the frame, which is copied to out_host.  A pixel is non-zero on exactly one rank, so the frame
 * equals the one-GPU frame bit for bit.  params->tile_rank / tile_ranks are ignored.  Handles on the
 * same device are allowed (rehearsal on one GPU): their buffers are summed on that device first.
 * RCCL is loaded at the first call that spans more than one device; RAYRS_RCCL_ERROR if that or a collective
 * fails.  stats: every counter summed over the ranks; times and kernel_launches = the slowest rank's. */
int rayrs_render_multi(rayrs_scene* const* scenes, uint32_t n, const rayrs_camera* camera,
                       const rayrs_render_params* params, void* out_host, rayrs_render_stats* stats);

/* ---- tuning: the two scheduling choices a caller may legitimately make.  0 = the built-in default.  Neither changes
 * the arithmetic or the answer.  (Round 1 read such settings from RAYRS_* environment variables; a library must not.  The kernels'
 * development knobs -- thresholds, LDS budgets, test switches -- are not part of this boundary: rayrs_amd/csrc/rayrs_lab.h.) */
typedef struct {
    uint32_t pool_slots;    /* streaming route: paths in flight = slots of the pool in HBM, 128 + 33 bytes each
                               (default: min(items, 256 Mi, samples / 12)); ignored on the local-pool route */
    uint32_t local_pool;    /* a scene whose gate tree is at most one record (gate_n_wide <= 1: the reference's sphere
                               scenes) is rendered by ONE launch that keeps every path in LDS from its first ray
                               to its last (local_pool.hip) instead of three launches per bounce over a pool in
                               HBM; same arithmetic, same bits.  0 = do so, 1 = never (the streaming kernels) */
} rayrs_tuning;
/* Applies to the renders launched on this scene afterwards.  Waits for a render in flight. */
int rayrs_scene_set_tuning(rayrs_scene* scene, const rayrs_tuning* tuning);

/* The boundary's version: bumped whenever a struct of this header changes the meaning of a field or an entry point
 * its behaviour.  5 = round 5: rayrs_render_params.exact_traversal became fast_traversal (opposite sense: zero is now
 * the reference's visit set), the device self-test hooks left this header.  6 = round 6: rayrs_scene_info_t.hot_*,
 * rayrs_render_stats.hot_*, rayrs_scene_export_hot_tree, rayrs_obj_load_spheres; the layout table below begins with this
 * number, so a binding that checks itself against the table fails on a version change as well.  A binding MUST compare
 * rayrs_abi_version() with the RAYRS_ABI_VERSION it was written against when it loads the library.  Every struct a caller fills must be zero-initialised
 * first: fields are added where padding used to be, and values out of a field's range are refused. */
#define RAYRS_ABI_VERSION 6
uint32_t rayrs_abi_version(void);

/* ---- layout of the structs above as THIS library was compiled, for bindings in other languages
 * to check theirs against (tests/test_abi.py does it for rayrs_amd/_ffi.py, and INTEGRATION.md's
 * #[repr(C)] structs carry the same numbers).  Writes up to `cap` words to `out` and returns the
 * number of words the full table has: first RAYRS_ABI_VERSION, then for each struct, in the order rayrs_material,
 * rayrs_emission, rayrs_camera, rayrs_scene_info_t, rayrs_render_params, rayrs_render_stats,
 * rayrs_tuning: sizeof, number of fields, then offsetof of every field in declaration order. */
uint32_t rayrs_abi_layout(uint32_t* out, uint32_t cap);

/* ---- file formats either side of the path (host only) ---- */

const char* rayrs_io_last_error(void);
/* frees a buffer returned by one of the loaders below */
void rayrs_buffer_free(void* p);

/* PLY 1.0, ascii or binary_little_endian: element vertex {x,y,z (+ ignored properties)},
 * element face {list vertex_indices}; polygons are fan-triangulated, winding kept.
 * (The reference's `ply` crate is entirely commented out, ply/src/lib.rs:1-350, so this
 * follows the public format.)  verts: nverts*3 f32, idx: ntris*3 -- feed them to
 * rayrs_object_from_triangles_f32. */
int rayrs_ply_load(const char* path, float** verts, uint32_t* nverts, uint32_t** idx, uint32_t* ntris);
int rayrs_ply_save(const char* path, const float* verts, uint32_t nverts, const uint32_t* idx, uint32_t ntris,
                   int binary);
/* wavefront_obj::load_obj_file, wavefront_obj.rs:15-45: `v x y z` / `f i j k` lines only. */
int rayrs_obj_load(const char* path, double** verts, uint32_t* nverts, uint32_t** idx, uint32_t* ntris);
/* wavefront_obj::load_obj_file_spheres, wavefront_obj.rs:46-64: one sphere centre per `v x y z` line, every other line
 * skipped; centers: n*3 f64 -- with the radius, feed them to rayrs_object_from_spheres (Object::from_spheres, lib.rs:422). */
int rayrs_obj_load_spheres(const char* path, double** centers, uint32_t* n);
/* Radiance .hdr: what HdrDecoder / HDREncoder do at rayrs/src/main.rs:36-41 and :113-121.
 * rgb: w*h*3 f32, top row first. */
int rayrs_hdr_load(const char* path, float** rgb, uint32_t* w, uint32_t* h);
int rayrs_hdr_save(const char* path, const float* rgb, uint32_t w, uint32_t h);
/* Image::to_raw_bytes(gamma), image.rs:193-222: clip(0,1).powf(gamma), (255.99*x) as u8.
 * out: w*h*3 bytes; counts = {clamped, NaN, negative} pixels (image.rs:218-220). */
int rayrs_image_to_bytes(const float* rgb, uint32_t w, uint32_t h, double gamma, uint8_t* out, uint64_t counts[3]);
int rayrs_ppm_save(const char* path, const uint8_t* bytes, uint32_t w, uint32_t h); /* image.rs:248-251 */
int rayrs_png_save(const char* path, const uint8_t* bytes, uint32_t w, uint32_t h); /* main.rs:104-110 */

#ifdef __cplusplus
}
#endif
#endif

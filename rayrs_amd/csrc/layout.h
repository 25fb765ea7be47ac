// layout.h -- HBM data layout shared by the host flattener and the gfx950 kernels.
//
// Everything the traversal touches is one of two record arrays, each record a
// whole number of 16-byte words so that a lane fetches it with dwordx4 loads:
//
//   wide record      four boxes + four references              128 B (compact) / 256 B (full)
//   primitive record triangle | sphere | plane + tag, DFS order  48 B (compact) /  80 B (full)
//
// The wide records are the product's own trees (scene_host.cpp build_walk_trees): the gate tree
// over the reference's leaf groups behind exactly their gating boxes (the default walk), and the fast
// walk's tree over the reference's primitives, each behind a box of its own inside its gating box; the two-child
// records of the reference's tree exist on the host only (rayrs_scene_export_bvh).
//
// "compact" = every box bound and every triangle vertex is exactly
// representable in f32 (true for PLY meshes, whose vertices are f32), so the
// f32 storage widens back to the very f64 values the reference computes with.
// Spheres and planes always keep f64 parameters inside their record.
#pragma once
#include <stdint.h>

namespace rayrs {

// child reference: kind << 30 | payload
constexpr uint32_t REF_INTERIOR = 0u;  // payload = interior record index
constexpr uint32_t REF_RANGE = 1u;     // payload = first_prim << 2 | (count - 1): 1..4 primitives behind the slot's box
constexpr uint32_t REF_SINGLE = 2u;    // two-child export only: payload = prim << 2, a direct leaf (no box of its own,
                                       // bvh.rs:297, :302); the gate tree holds it as a one-primitive REF_RANGE behind
                                       // the box of the Node it hangs under
constexpr uint32_t REF_NONE = 3u;

// primitive tag: kind | axis << 2 | surface << 8
constexpr uint32_t PRIM_SPHERE = 0u;
constexpr uint32_t PRIM_PLANE = 1u;
constexpr uint32_t PRIM_TRIANGLE = 2u;

struct NodeF32 {  // 64 B
    float box[2][6];  // xmin,xmax,ymin,ymax,zmin,zmax per child
    uint32_t ref[2];
    uint32_t pad[2];
};
struct NodeF64 {  // 128 B
    double box[2][6];
    uint32_t ref[2];
    uint32_t pad[6];
};
static_assert(sizeof(NodeF32) == 64, "NodeF32");
static_assert(sizeof(NodeF64) == 128, "NodeF64");

// The records the traversal kernel reads are "wide": up to four slots, each a box and a reference.
// A leaf slot holds a group of the reference's tree (the 1..4 leaves under one Node) behind the box
// whose test gates the reference's access to it; an interior slot holds the union of the boxes below
// it and the record they are in.  Unused slots carry REF_NONE and the inverted box.
struct Node4F32 {  // 128 B
    float box[4][6];
    uint32_t ref[4];
    uint32_t pad[4];
};
struct Node4F64 {  // 256 B
    double box[4][6];
    uint32_t ref[4];
    uint32_t pad[12];
};
// records renumbered to the front, largest box first (scene_host.cpp front_largest)
constexpr uint32_t WIDE_FRONT = 256;
constexpr uint32_t TRAV_LEAFQ = 8;  // leaf groups a lane of the default walk may have waiting (device_path.h LEAFQ, abi.cpp's LDS sizing)
static_assert(sizeof(Node4F32) == 128, "Node4F32");
static_assert(sizeof(Node4F64) == 256, "Node4F64");

// Primitive records are raw dwords: the payload starts at dword 0 and the tag
// is the last dword.
//   triangle: p1,p2,p3 as 9 f32 (compact) or 9 f64 (full)
//   sphere  : radius2, cx, cy, cz as 4 f64
//   plane   : umin, umax, vmin, vmax, pos as 5 f64 (axis in the tag)
constexpr uint32_t PRIM_DWORDS_COMPACT = 12;  // 48 B
constexpr uint32_t PRIM_DWORDS_FULL = 20;     // 80 B

// One row per distinct (Material, Emission) pair; material.rs:148-238, :1056-1060.
struct SurfaceDev {
    int32_t kind;      // RAYRS_MAT_*
    int32_t metallic;  // CookTorrance layer: Fresnel::SchlickMetallic
    double color[3];   // Lambertian / Reflect / Refract / Glass colour; Plastic diffuse colour
    double ior;
    double ct_alpha2;  // alpha * alpha, material.rs:711
    double ct_ior;
    double ct_r0[3];
    double ct_color[3];
    double emit[3];    // Emission::emit(): strength * color, or zeros
    int32_t emissive;
    int32_t pad;
};

// The HOT GROUP of the default walk (scene_host.cpp pick_hot_group, device_path.h hot_group_step): one leaf group of the
// reference's tree whose gating box covers most of the root Node's box -- on the benchmark scenes the 50 x 50 floor and
// the three mesh triangles that share its bottom Node, which 89 % of all rays enter.  It is taken out of the tree the
// default walk reads and tested once per ray by the kernel that MAKES the ray, for a whole batch of rays together
// (wavefront.hip finish_rays): the same gating-box test and the same primitive tests on the same f64 values as
// BvhTree::intersect makes (bvh.rs:391-415), but on
// wave-uniform data -- read with scalar loads, nothing converted, e1 = p2 - p1 and e2 = p3 - p1 formed on the host as
// Triangle::new forms them (geometry.rs:342-343).
struct HotPrim {       // 80 B
    double v[9];       // triangle: p1, e1, e2 | sphere: radius^2, centre | rectangle: umin, umax, vmin, vmax, pos
    uint32_t tag;      // the primitive record's tag (kind | axis << 2 | surface << 8)
    uint32_t pad;
};
struct HotGroupDev {
    double box[6];     // the group's gating box, as the reference tests it
    uint32_t first;    // DFS index of the group's first primitive
    uint32_t count;    // 1..4
    uint32_t pad[2];
    HotPrim prim[4];
    // the root record of the tree without the group (FlatScene::gate_hot), as f64: the kernel that MAKES a ray tests the
    // group and these four boxes for it (wavefront.hip finish_rays); a ray that enters none of them has its answer
    // there and never travels through the traversal kernel.  Unused slots hold the inverted box.
    double root_box[4][6];
    uint32_t root_ref[4];
    uint32_t n_tri, n_sphere, n_plane, pad1;  // kinds among the group's primitives (work counters)
};
static_assert(sizeof(HotPrim) == 80 && sizeof(HotGroupDev) == 384 + 192 + 32, "HotGroupDev");

struct SceneDev {
    const void* nodes;
    const void* prims;
    const SurfaceDev* surfaces;
    const float* hdri;  // per texel (i, j): the 2x2 footprint of a lookup there, 4 x RGBA f32 (A unused), clipped to [0,3]
    uint32_t hdri_w, hdri_h;
    uint32_t root_ref;
    uint32_t stack_depth;  // entries a traversal can have pending (FlatScene::wide_depth)
    uint32_t stack_lds;    // how many of them live in LDS; the rest overflow to HBM (LaneStack)
    uint32_t hot_records;  // leading wide records the traversal kernel copies to LDS (HotNodes)
    uint32_t n_surfaces;
    uint32_t leafq;        // the default walk: leaf groups a lane may have waiting in LDS (device_path.h LEAFQ); 0 = the fast walk
    double root_box[6];
    double t0, t1;  // Scene::t_range, lib.rs:218
    double hdri_wm1, hdri_hm1;  // (hdri_w - 1) as f64 and (hdri_h - 1) as f64, lib.rs:262-263 (converted on the host: a kernel
                                // that converts them hoists the results into vector registers for its whole run)
    // the default walk (rayrs_render_params.fast_traversal == 0): nothing is culled by the closest hit so far, as BvhTree::intersect
    // (bvh.rs:391-415); selects the EXACT instances of the kernels (device_path.h trav_interior_step)
    uint32_t exact, pad1;
    // the default walk's hot group, or null: then `nodes` is the whole gate tree (HotGroupDev above)
    const HotGroupDev* hot;
};

struct CameraDev {
    double origin[3], e_x[3], e_y[3], z[3];
    double width, height, ppc;
    uint32_t W, H;
};

struct Counters {
    unsigned long long rays, paths, nan_pixels, neg_pixels;
    unsigned long long interior_visits, tri_tests, sphere_tests, plane_tests, escaped_paths;
    // lane-utilisation diagnostics (count_work only): *_wave counts executions of a
    // phase by a wave x 64, *_lane the lanes that were active in it
    unsigned long long step_wave, step_lane, inner_wave, leaf_wave;
    unsigned long long interior_ticks, leaf_ticks, refill_ticks;  // shader clock, summed over waves
    unsigned long long surface_hits[8];  // closest hits per surface row (rows 7 and up together), count_work only
    unsigned long long direct_rays;  // primary rays that missed the root box: answered by the kernel that made them
    // the pre-test of new rays by the kernels that make them (wavefront.hip finish_rays; count_work only):
    unsigned long long pre_rays;          // queries answered there: bounced rays that miss the root box, rays that enter no slot of the walk tree's first record
    unsigned long long pre_root_records;  // ... of them: rays that entered the root box and no slot of the first record (one record visit each)
    unsigned long long hot_lane;          // rays put to the hot group's gating box
    unsigned long long hot_prim_tests;    // of tri + sphere + plane tests: the hot group's
    unsigned long long hot_tri_divided;   // of its triangle tests: those that went on to the three divisions
#ifdef RAYRS_LAB_TICKS
    unsigned long long lab_ticks[16];  // development build only (make LAB=1): shader-clock shares of the hit / miss loops
#endif
};

struct RenderDev {
    uint32_t spp, max_bounces;
    uint64_t seed;
    uint32_t chunk, nchunks;
    uint32_t tile_rank, tile_ranks;
    uint32_t tiles_x, tiles_y;
    uint32_t n_local_tiles;
    uint32_t out_format;
    uint64_t total_items;  // n_local_tiles * nchunks * 64
    double inv_nchunks, inv_tiles_x;  // reciprocals for udiv_by() in the kernels
    uint32_t refill_min, leaf_min;  // traversal scheduling thresholds (lanes)
    uint32_t leaf_wait, pad_rd;     // ... the default walk: a leaf phase once this many lanes can do nothing but wait for one
    uint32_t static_windows;        // pool windows dealt to the traversal waves round robin (wavefront.hip)
    uint32_t count_work;            // also count closest hits per surface (hit kernel)
    double* partial;       // item sums, 3 doubles each, of the items partial_item0 .. (all of them, or one segment's)
    uint64_t partial_item0;
    unsigned long long* next_item;  // device-wide item counter
    Counters* counters;
    void* out;
};

// n / d and n % d for a launch-constant d with 1/d at hand: the quotient of the f64 product is
// within one of the true one (n < 2^32, relative error 2^-52), and the remainder says which.
// A dozen instructions against the ~35 of a 32-bit integer division.
#if defined(__HIPCC__)
#define RR_LAYOUT_FN __host__ __device__ static inline
#else
#define RR_LAYOUT_FN static inline
#endif
RR_LAYOUT_FN uint32_t udiv_by(uint32_t n, uint32_t d, double inv_d, uint32_t& rem) {
    uint32_t q = (uint32_t)((double)n * inv_d);
    int32_t r = (int32_t)(n - q * d);
    if (r < 0) q--, r += (int32_t)d;
    else if ((uint32_t)r >= d) q++, r -= (int32_t)d;
    rem = (uint32_t)r;
    return q;
}

// The pixel, and the sample range, of item `item`: 64 pixels of a tile x the tile's chunks (main.rs:65-68).
RR_LAYOUT_FN void item_geometry(const RenderDev& rp, uint32_t item, uint32_t& row, uint32_t& col, uint32_t& s_begin,
                                uint32_t& s_end) {
    const uint32_t pit = item & 63u;
    uint32_t chunk, tile_col;
    const uint32_t tile = udiv_by(item >> 6, rp.nchunks, rp.inv_nchunks, chunk) * rp.tile_ranks + rp.tile_rank;
    row = udiv_by(tile, rp.tiles_x, rp.inv_tiles_x, tile_col) * 8u + (pit >> 3);
    col = tile_col * 8u + (pit & 7u);
    s_begin = chunk * rp.chunk;
    s_end = s_begin + rp.chunk < rp.spp ? s_begin + rp.chunk : rp.spp;
}

}  // namespace rayrs

// scene_host.hpp -- host-side counterpart of rayrs-lib's Object / Scene / Camera
// constructors and of Bvh::build, producing the flattened arrays the kernels read.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rayrs_hip.h"
#include "layout.h"

namespace rayrs {

struct Vec3 {
    double x, y, z;
};

struct Aabb {
    double xmin, xmax, ymin, ymax, zmin, zmax;
};

// geometry.rs: Sphere :78-81, Plane :176-181, Triangle :312-322 (p1,p2,p3 only:
// e1, e2 and the normal are recomputed in f64 by the kernel with the same
// operations Triangle::new uses).
struct Shape {
    uint32_t kind;  // PRIM_*
    uint32_t axis;
    double radius2;
    Vec3 origin;
    double u0, u1, v0, v1, pos;
    Vec3 p1, p2, p3;
};

struct Object {  // lib.rs:302-306
    Shape geom;
    uint32_t surface;
};

// Vec<Object> under construction
struct ObjectList {
    std::vector<Object> objs;
    std::vector<SurfaceDev> surfaces;

    int add_surface(const rayrs_material* m, const rayrs_emission* e);
};

Aabb shape_bbox(const Shape& s);

struct WalkTree {
    std::vector<double> box;    // n * 24
    std::vector<uint32_t> ref;  // n * 4
    uint32_t root_ref = 0;
    uint32_t depth = 0;         // stack entries the traversal can need
    std::vector<uint8_t> node_bytes;  // the records as the kernels read them (Node4F32 / Node4F64)
    uint32_t n() const { return (uint32_t)(ref.size() / 4); }
};

struct FlatScene {
    // logical tree (always f64, used for export and as the source of the device records)
    std::vector<double> child_box;  // n_interior * 12
    std::vector<uint32_t> child_ref;
    std::vector<uint32_t> prim_object;
    uint32_t root_ref = 0;
    uint32_t depth = 0;
    // The trees the kernels traverse (scene_host.cpp "the walk trees"): four-slot records whose leaf slots are
    //   gate  the reference's leaf groups behind their exact gating boxes -- reaches exactly what the reference
    //         reaches; the default walk's tree, and where the local-pool route's gates come from;
    //   walk  single primitives behind their own widened boxes inside the gating box (rayrs_render_params.fast_traversal).
    //   gate_hot  the gate tree without the scene's HOT GROUP (layout.h HotGroupDev; has_hot), which the default walk's
    //         kernels test once per ray outside the tree -- gate_hot's groups + the hot group = gate's groups.
    WalkTree walk, gate, gate_hot;
    bool has_hot = false;
    HotGroupDev hot = {};
    double root_box[6] = {0, 0, 0, 0, 0, 0};
    bool compact = false;
    // device images (the trees' records are in WalkTree::node_bytes)
    std::vector<uint8_t> prim_bytes;
    std::vector<float> hdri_quads;  // 16 floats per texel: the 2x2 footprint of a lookup at (i, j), RGBA each
    uint32_t hdri_w = 0, hdri_h = 0;
    double t0 = 0, t1 = 0;
    double build_seconds = 0;
    double small_extent = 0;  // 5th percentile of the primitives' largest bounding-box extents (abi.cpp camera_is_far)
    uint32_t n_interior() const { return (uint32_t)(child_ref.size() / 2); }
    uint32_t n_prims() const { return (uint32_t)prim_object.size(); }
};

// Scene::new (lib.rs:227-245) minus the upload.  Returns RAYRS_* status.
int build_flat_scene(const ObjectList& objs, double z_near, double z_far, int heuristic, uint32_t splits,
                     uint32_t hdri_w, uint32_t hdri_h, const float* hdri_rgb, FlatScene* out);

// Camera::new (lib.rs:99-133)
int camera_new(const double origin[3], const double up[3], const double lookat[3], double fov, double width,
               double height, uint32_t ppi, rayrs_camera* out);

}  // namespace rayrs

// abi.cpp -- the extern "C" boundary declared in include/rayrs_hip.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rayrs_hip.h"
#include "kernels.h"
#include "local_pool.h"
#include "rayrs_lab.h"
#include "rayrs_selftest.h"
#include "scene_host.hpp"
#include "scene_internal.hpp"
#include "wavefront.h"

using namespace rayrs;

namespace {
thread_local std::string g_last_error;
constexpr uint32_t TRAV_STACK_LDS = 12;
constexpr uint32_t TRAV_HOT_BYTES = 14u * 1024u;
// The trees the default walk reads: a lane's leaf groups wait in a queue of LEAFQ entries behind its stack
// (device_path.h trav_interior_step_defer), so its stack holds interior records only -- 8 entries in LDS serve what 12
// served with the leaves among them -- and the records kept in LDS give up the other 4 KiB of the queue's 8.
constexpr uint32_t FLAT_BLOCKS_PER_CU_MIN = 8, FLAT_BLOCKS_PER_CU_MAX = 24;  // the gen / hit / miss kernels' common grid (rayrs_lab_tuning.flat_blocks_per_cu)
constexpr uint32_t TRAV_STACK_LDS_DEFER = 8;
constexpr uint32_t TRAV_HOT_BYTES_DEFER = 10u * 1024u;
}  // namespace

namespace rayrs {
void set_last_error(const std::string& text) { g_last_error = text; }
int hip_fail(hipError_t e, const char* what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? RAYRS_OOM : RAYRS_HIP_ERROR;
}
}  // namespace rayrs

static void push_triangle(ObjectList& l, Vec3 p1, Vec3 p2, Vec3 p3, uint32_t surf) {
    Object o;
    std::memset(&o, 0, sizeof(o));
    o.geom.kind = PRIM_TRIANGLE;
    o.geom.p1 = p1, o.geom.p2 = p2, o.geom.p3 = p3;
    o.surface = surf;
    l.objs.push_back(o);
}

template <typename T>
static int from_triangles(rayrs_objects* objs, const T* verts, uint32_t nverts, const uint32_t* idx, uint32_t ntris,
                          const rayrs_material* mat, const rayrs_emission* emission) {
    if (!objs || (!verts && nverts) || (!idx && ntris)) return RAYRS_INVALID_ARG;
    for (size_t i = 0; i < (size_t)ntris * 3; i++)
        if (idx[i] >= nverts) return RAYRS_INVALID_ARG;
    const int surf = objs->list.add_surface(mat, emission);
    if (surf < 0) return surf;
    objs->list.objs.reserve(objs->list.objs.size() + ntris);
    for (uint32_t t = 0; t < ntris; t++) {
        const T* a = verts + 3 * (size_t)idx[3 * t];
        const T* b = verts + 3 * (size_t)idx[3 * t + 1];
        const T* c = verts + 3 * (size_t)idx[3 * t + 2];
        push_triangle(objs->list, {(double)a[0], (double)a[1], (double)a[2]},
                      {(double)b[0], (double)b[1], (double)b[2]}, {(double)c[0], (double)c[1], (double)c[2]},
                      (uint32_t)surf);
    }
    return RAYRS_OK;
}

extern "C" {

const char* rayrs_strerror(int status) {
    switch (status) {
        case RAYRS_OK: return "ok";
        case RAYRS_INVALID_ARG: return "invalid argument (a reference assert! would have fired)";
        case RAYRS_HIP_ERROR: return "HIP runtime error";
        case RAYRS_OOM: return "out of memory";
        case RAYRS_NO_DEVICE: return "no HIP device for this scene";
        case RAYRS_UNSUPPORTED: return "unsupported (size or depth limit)";
        case RAYRS_IO_ERROR: return "file missing or malformed";
        case RAYRS_RCCL_ERROR: return "RCCL unavailable or failed";
        default: return "unknown status";
    }
}

const char* rayrs_last_error(void) { return g_last_error.c_str(); }

// ------------------------------------------------------------- Vec<Object>

int rayrs_objects_create(rayrs_objects** out) {
    RAYRS_GUARDED({
    if (!out) return RAYRS_INVALID_ARG;
    *out = new (std::nothrow) rayrs_objects();
    return *out ? RAYRS_OK : RAYRS_OOM;
    })
}

void rayrs_objects_destroy(rayrs_objects* objs) { delete objs; }

uint64_t rayrs_objects_len(const rayrs_objects* objs) { return objs ? objs->list.objs.size() : 0; }

int rayrs_object_sphere(rayrs_objects* objs, double radius, const double origin[3], const rayrs_material* mat,
                        const rayrs_emission* emission) {
    RAYRS_GUARDED({
    if (!objs || !origin) return RAYRS_INVALID_ARG;
    if (!(radius > 0.)) return RAYRS_INVALID_ARG;  // geometry.rs:97
    const int surf = objs->list.add_surface(mat, emission);
    if (surf < 0) return surf;
    Object o;
    std::memset(&o, 0, sizeof(o));
    o.geom.kind = PRIM_SPHERE;
    o.geom.radius2 = radius * radius;  // geometry.rs:99
    o.geom.origin = {origin[0], origin[1], origin[2]};
    o.surface = (uint32_t)surf;
    objs->list.objs.push_back(o);
    return RAYRS_OK;
    })
}

int rayrs_object_plane(rayrs_objects* objs, int axis, double umin, double umax, double vmin, double vmax, double pos,
                       const rayrs_material* mat, const rayrs_emission* emission) {
    RAYRS_GUARDED({
    if (!objs) return RAYRS_INVALID_ARG;
    if (!(umin < umax && vmin < vmax)) return RAYRS_INVALID_ARG;  // geometry.rs:205-212
    if (axis < 0 || axis > 5) return RAYRS_INVALID_ARG;
    const int surf = objs->list.add_surface(mat, emission);
    if (surf < 0) return surf;
    Object o;
    std::memset(&o, 0, sizeof(o));
    o.geom.kind = PRIM_PLANE;
    o.geom.axis = (uint32_t)axis;
    o.geom.u0 = umin, o.geom.u1 = umax, o.geom.v0 = vmin, o.geom.v1 = vmax, o.geom.pos = pos;
    o.surface = (uint32_t)surf;
    objs->list.objs.push_back(o);
    return RAYRS_OK;
    })
}

int rayrs_object_triangle(rayrs_objects* objs, const double p1[3], const double p2[3], const double p3[3],
                          const rayrs_material* mat, const rayrs_emission* emission) {
    RAYRS_GUARDED({
    if (!objs || !p1 || !p2 || !p3) return RAYRS_INVALID_ARG;
    const int surf = objs->list.add_surface(mat, emission);
    if (surf < 0) return surf;
    push_triangle(objs->list, {p1[0], p1[1], p1[2]}, {p2[0], p2[1], p2[2]}, {p3[0], p3[1], p3[2]}, (uint32_t)surf);
    return RAYRS_OK;
    })
}

int rayrs_object_from_triangles_f32(rayrs_objects* objs, const float* verts, uint32_t nverts, const uint32_t* idx,
                                    uint32_t ntris, const rayrs_material* mat, const rayrs_emission* emission) {
    RAYRS_GUARDED({
    return from_triangles<float>(objs, verts, nverts, idx, ntris, mat, emission);
    })
}

int rayrs_object_from_triangles_f64(rayrs_objects* objs, const double* verts, uint32_t nverts, const uint32_t* idx,
                                    uint32_t ntris, const rayrs_material* mat, const rayrs_emission* emission) {
    RAYRS_GUARDED({
    return from_triangles<double>(objs, verts, nverts, idx, ntris, mat, emission);
    })
}

int rayrs_object_from_spheres(rayrs_objects* objs, double radius, const double* centers, uint32_t n,
                              const rayrs_material* mat, const rayrs_emission* emission) {
    if (!objs || (!centers && n)) return RAYRS_INVALID_ARG;
    for (uint32_t i = 0; i < n; i++) {
        const int st = rayrs_object_sphere(objs, radius, centers + 3 * (size_t)i, mat, emission);
        if (st != RAYRS_OK) return st;
    }
    return RAYRS_OK;
}

int rayrs_object_box_geom(rayrs_objects* objs, const double ll[3], const double ur[3], const rayrs_material* mat,
                          const rayrs_emission* emission) {
    if (!objs || !ll || !ur) return RAYRS_INVALID_ARG;
    // lib.rs:444-505, same order (note: both Y faces sit at lower_left.y, as in the reference)
    const struct {
        int axis;
        double u0, u1, v0, v1, pos;
    } faces[6] = {
        {RAYRS_AXIS_X, ll[1], ur[1], ll[2], ur[2], ll[0]},    {RAYRS_AXIS_XREV, ll[1], ur[1], ll[2], ur[2], ur[0]},
        {RAYRS_AXIS_ZREV, ll[0], ur[0], ll[1], ur[1], ll[2]}, {RAYRS_AXIS_Z, ll[0], ur[0], ll[1], ur[1], ur[2]},
        {RAYRS_AXIS_YREV, ll[0], ur[0], ll[2], ur[2], ll[1]}, {RAYRS_AXIS_Y, ll[0], ur[0], ll[2], ur[2], ll[1]},
    };
    for (const auto& f : faces) {
        const int st = rayrs_object_plane(objs, f.axis, f.u0, f.u1, f.v0, f.v1, f.pos, mat, emission);
        if (st != RAYRS_OK) return st;
    }
    return RAYRS_OK;
}

// ------------------------------------------------------------------- Scene

extern "C++" void rayrs::scene_free_device(rayrs_scene* s) {
    if (s->device < 0) return;
    (void)hipSetDevice(s->device);
    if (s->pending && s->last_stream) (void)hipStreamSynchronize(s->last_stream);
    for (auto& w : s->trav)
        if (w.d_nodes) (void)hipFree(w.d_nodes);
    if (s->d_prims) (void)hipFree(s->d_prims);
    if (s->d_hot) (void)hipFree(s->d_hot);
    if (s->d_surfaces) (void)hipFree(s->d_surfaces);
    if (s->d_hdri) (void)hipFree(s->d_hdri);
    if (s->d_counters) (void)hipFree(s->d_counters);
    if (s->d_partial) (void)hipFree(s->d_partial);
    {
        rayrs_scene::Pool& pl = s->pool;
        if (pl.block) (void)hipFree(pl.block);
        if (pl.d_wave_items) (void)hipFree(pl.d_wave_items);
        if (pl.d_stack_spill) (void)hipFree(pl.d_stack_spill);
        if (pl.wf.ctl) (void)hipFree(pl.wf.ctl);
        if (pl.h_live) (void)hipHostFree(pl.h_live);
        for (auto& e : pl.ev_batch)
            if (e) (void)hipEventDestroy(e);
        for (auto& e : pl.ev_round)
            if (e) (void)hipEventDestroy(e);
    }
    if (s->d_next_item) (void)hipFree(s->d_next_item);
    if (s->multi_out) (void)hipFree(s->multi_out);
    if (s->multi_stream) (void)hipStreamDestroy(s->multi_stream);
    if (s->d_local_light) (void)hipFree(s->d_local_light);
    if (s->d_local_items) (void)hipFree(s->d_local_items);
    for (auto& e : s->ev)
        if (e) (void)hipEventDestroy(e);
}

void rayrs_scene_destroy(rayrs_scene* scene) {
    if (!scene) return;
    scene_free_device(scene);
    delete scene;
}

// Sizes the traversal workgroup's LDS from the tree and the scene's tuning, and asks the runtime how
// many such workgroups fit a CU.  A workgroup's LDS: the first stack_lds entries of each lane's stack
// (deeper entries overflow to HBM; on the 1M-triangle scene 99.4 % of visits happen with at most 7
// pending), 4 KiB of window lists, and the hot_records largest wide records.  13 + 4 + 14 KiB (the default walk's trees: 17 KiB with the lanes' leaf queues + 4 + 10) lets
// five workgroups (the kernel's launch bound) share a CU's 160 KiB.
static int scene_configure_traversal(rayrs_scene* s) {
    const FlatScene& f = s->flat;
    for (int x = 0; x < 3; x++) {
        const WalkTree& t = s->tree(x);
        rayrs_scene::Walk& w = s->trav[x];
        const uint32_t depth = t.depth ? t.depth : 1;
        w.leafq = x == 0 ? 0u : TRAV_LEAFQ;  // ([0] is the fast walk's tree; [1] is walked either way, [2] by the default walk only)
        uint32_t want = s->lab.stack_lds ? s->lab.stack_lds : (w.leafq ? TRAV_STACK_LDS_DEFER : TRAV_STACK_LDS);
        w.stack_lds = want < depth ? want : depth;
        const uint32_t rec_bytes = f.compact ? (uint32_t)sizeof(Node4F32) + 16u : (uint32_t)sizeof(Node4F64) + 16u;
        uint32_t hot = (w.leafq ? TRAV_HOT_BYTES_DEFER : TRAV_HOT_BYTES) / rec_bytes;
        if (s->lab.hot_records == 0xffffffffu) hot = 0;
        else if (s->lab.hot_records) hot = s->lab.hot_records < WIDE_FRONT ? s->lab.hot_records : WIDE_FRONT;
        w.hot_records = hot < t.n() ? hot : t.n();
        HIP_TRY(wf_trav_occupancy(f.compact, w.stack_lds, w.leafq, w.hot_records, &w.blocks_per_cu));
        if (w.blocks_per_cu < 1) w.blocks_per_cu = 1;
    }
    return RAYRS_OK;
}

// The gating boxes of a walk tree of at most one record, as kernel arguments of local_pool.hip (LocalScene).
constexpr uint32_t LOCAL_SEGMENT_ITEMS = 1u << 27;  // items per launch of the local-pool kernel
constexpr uint32_t LOCAL_MAX_SEGMENTS = 64;
static void scene_configure_local(rayrs_scene* s) {
    const FlatScene& f = s->flat;
    LocalScene& ls = s->local;
    std::memset(&ls, 0, sizeof(ls));
    s->local_ok = false;
    const WalkTree& t = f.gate;  // (the groups behind their gating boxes: this route makes neither of the default walk's bets)
    if (t.n() > 1 || f.n_prims() == 0 || f.n_prims() > LP_MAX_PRIMS || s->surfaces.size() > LP_MAX_PRIMS) return;
    auto add_gate = [&](const double* box, uint32_t ref) {
        if ((ref >> 30) != REF_RANGE) return false;
        const uint32_t g = ls.n_gates++;
        for (int i = 0; i < 6; i++) ls.box[g][i] = box[i];
        ls.first[g] = (ref & 0x3fffffffu) >> 2;
        ls.count[g] = (ref & 3u) + 1u;
        return true;
    };
    if (t.n() == 0) {  // the root group behind the root Node's box (trav_init)
        if (!add_gate(f.root_box, t.root_ref)) return;
    } else {
        if ((t.root_ref >> 30) != REF_INTERIOR) return;
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t ref = t.ref[k];
            if ((ref >> 30) == REF_NONE) continue;
            if (!add_gate(&t.box[(size_t)k * 6], ref)) return;  // an interior slot: not a one-record tree
        }
    }
    uint32_t covered = 0;
    for (uint32_t g = 0; g < ls.n_gates; g++) covered += ls.count[g];
    if (covered != f.n_prims()) return;
    ls.n_records = t.n();
    ls.n_prims = f.n_prims();
    for (const SurfaceDev& sf : s->surfaces) ls.kind_mask |= 1u << (uint32_t)sf.kind;
    s->local_ok = true;
}

extern "C++" int rayrs::scene_upload(rayrs_scene* s) {
    HIP_TRY(hipSetDevice(s->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, s->device));
    s->cu_count = prop.multiProcessorCount;
    const FlatScene& f = s->flat;
    for (int x = 0; x < 3; x++) {
        const WalkTree& t = s->tree(x);
        if (x == 2 && !f.has_hot) continue;
        HIP_TRY(hipMalloc(&s->trav[x].d_nodes, t.node_bytes.size()));
        HIP_TRY(hipMemcpy(s->trav[x].d_nodes, t.node_bytes.data(), t.node_bytes.size(), hipMemcpyHostToDevice));
    }
    if (f.has_hot) {
        HIP_TRY(hipMalloc((void**)&s->d_hot, sizeof(HotGroupDev)));
        HIP_TRY(hipMemcpy(s->d_hot, &f.hot, sizeof(HotGroupDev), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(&s->d_prims, f.prim_bytes.size()));
    HIP_TRY(hipMemcpy(s->d_prims, f.prim_bytes.data(), f.prim_bytes.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void**)&s->d_surfaces, s->surfaces.size() * sizeof(SurfaceDev)));
    HIP_TRY(hipMemcpy(s->d_surfaces, s->surfaces.data(), s->surfaces.size() * sizeof(SurfaceDev),
                      hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void**)&s->d_hdri, f.hdri_quads.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(s->d_hdri, f.hdri_quads.data(), f.hdri_quads.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void**)&s->d_counters, sizeof(Counters)));
    for (auto& e : s->ev) HIP_TRY(hipEventCreate(&e));
    s->device_bytes = f.walk.node_bytes.size() + f.gate.node_bytes.size() + (f.has_hot ? f.gate_hot.node_bytes.size() : 0) + f.prim_bytes.size() + s->surfaces.size() * sizeof(SurfaceDev) +
                      f.hdri_quads.size() * sizeof(float);
    {
        const int st = scene_configure_traversal(s);
        if (st != RAYRS_OK) return st;
    }
    {
        rayrs_scene::Pool& pl = s->pool;
        HIP_TRY(hipMalloc((void**)&pl.wf.ctl, sizeof(WfCtl)));
        HIP_TRY(hipHostMalloc((void**)&pl.h_live, 2 * sizeof(uint32_t), hipHostMallocDefault));
        for (auto& e : pl.ev_batch) HIP_TRY(hipEventCreate(&e));
    }
    HIP_TRY(hipMalloc((void**)&s->d_next_item, sizeof(unsigned long long)));
    if (s->local_ok) {
        HIP_TRY(lp_configure());
        // (from about 13 primitives and surface rows up three workgroups' LDS no longer fit a CU: ask, do not assume)
        HIP_TRY(lp_occupancy(s->flat.compact, s->local.n_prims, (uint32_t)s->surfaces.size(), &s->local_blocks_per_cu));
        if (s->local_blocks_per_cu < 1) s->local_blocks_per_cu = 1;
        if (s->local_blocks_per_cu > (int)LP_WPS) s->local_blocks_per_cu = (int)LP_WPS;
        HIP_TRY(hipMalloc((void**)&s->d_local_items, LOCAL_MAX_SEGMENTS * sizeof(unsigned long long)));
    }
    return RAYRS_OK;
}

int rayrs_scene_new(const rayrs_objects* objs, double z_near, double z_far, int heuristic, uint32_t splits,
                    uint32_t hdri_w, uint32_t hdri_h, const float* hdri_rgb, int device, rayrs_scene** out) {
    RAYRS_GUARDED({
    if (!objs || !out) return RAYRS_INVALID_ARG;
    *out = nullptr;
    std::unique_ptr<rayrs_scene> s(new rayrs_scene());
    int st = build_flat_scene(objs->list, z_near, z_far, heuristic, splits, hdri_w, hdri_h, hdri_rgb, &s->flat);
    if (st != RAYRS_OK) return st;
    // Every traversal lane gets a stack of WalkTree::depth entries (12 in LDS, the rest in HBM: 1.3 MB per entry on a
    // 256-CU device).  The reference recurses as deep as its tree; a tree that needs more than 4096 pending
    // entries (a chain of thousands of nested objects) is refused instead of allocating gigabytes for it.
    if (std::max(s->flat.walk.depth, s->flat.gate.depth) > 4096u) {
        g_last_error = "walk tree needs " + std::to_string(std::max(s->flat.walk.depth, s->flat.gate.depth)) + " stack entries (limit 4096)";
        return RAYRS_UNSUPPORTED;
    }
    s->surfaces = objs->list.surfaces;
    s->n_objects = objs->list.objs.size();
    s->device = device;
    scene_configure_local(s.get());
    if (device >= 0) {
        st = scene_upload(s.get());
        if (st != RAYRS_OK) {
            scene_free_device(s.get());
            return st;
        }
    }
    *out = s.release();
    return RAYRS_OK;
    })
}

int rayrs_scene_info(const rayrs_scene* scene, rayrs_scene_info_t* info) {
    if (!scene || !info) return RAYRS_INVALID_ARG;
    const FlatScene& f = scene->flat;
    std::memset(info, 0, sizeof(*info));
    info->n_objects = scene->n_objects;
    info->n_interior = f.n_interior();
    info->n_prims = f.n_prims();
    info->root_ref = f.root_ref;
    info->depth = f.depth;
    info->compact = f.compact ? 1u : 0u;
    info->n_surfaces = (uint32_t)scene->surfaces.size();
    info->node_bytes = f.compact ? (uint32_t)sizeof(Node4F32) : (uint32_t)sizeof(Node4F64);
    info->n_wide = f.walk.n();
    info->wide_root_ref = f.walk.root_ref;
    info->wide_depth = f.walk.depth;
    info->gate_n_wide = f.gate.n();
    info->gate_root_ref = f.gate.root_ref;
    info->gate_depth = f.gate.depth;
    if (f.has_hot) {
        info->hot_n_wide = f.gate_hot.n();
        info->hot_root_ref = f.gate_hot.root_ref;
        info->hot_depth = f.gate_hot.depth;
        info->hot_first = f.hot.first;
        info->hot_count = f.hot.count;
        for (int i = 0; i < 6; i++) info->hot_box[i] = f.hot.box[i];
    }
    info->local_pool = (scene->local_ok && scene->tuning.local_pool != 1u) ? 1u : 0u;
    info->prim_bytes = 4u * (f.compact ? PRIM_DWORDS_COMPACT : PRIM_DWORDS_FULL);
    info->device_bytes = scene->device_bytes;
    for (int i = 0; i < 6; i++) info->root_box[i] = f.root_box[i];
    info->build_seconds = f.build_seconds;
    return RAYRS_OK;
}

int rayrs_scene_export_bvh(const rayrs_scene* scene, double* child_box, uint32_t* child_ref, uint32_t* prim_object) {
    if (!scene) return RAYRS_INVALID_ARG;
    const FlatScene& f = scene->flat;
    if (child_box && !f.child_box.empty()) std::memcpy(child_box, f.child_box.data(), f.child_box.size() * 8);
    if (child_ref && !f.child_ref.empty()) std::memcpy(child_ref, f.child_ref.data(), f.child_ref.size() * 4);
    if (prim_object && !f.prim_object.empty())
        std::memcpy(prim_object, f.prim_object.data(), f.prim_object.size() * 4);
    return RAYRS_OK;
}

int rayrs_scene_export_wide(const rayrs_scene* scene, double* wide_box, uint32_t* wide_ref) {
    if (!scene) return RAYRS_INVALID_ARG;
    const WalkTree& t = scene->flat.walk;
    if (wide_box && !t.box.empty()) std::memcpy(wide_box, t.box.data(), t.box.size() * 8);
    if (wide_ref && !t.ref.empty()) std::memcpy(wide_ref, t.ref.data(), t.ref.size() * 4);
    return RAYRS_OK;
}

int rayrs_scene_export_gate_tree(const rayrs_scene* scene, double* wide_box, uint32_t* wide_ref) {
    if (!scene) return RAYRS_INVALID_ARG;
    const WalkTree& t = scene->flat.gate;
    if (wide_box && !t.box.empty()) std::memcpy(wide_box, t.box.data(), t.box.size() * 8);
    if (wide_ref && !t.ref.empty()) std::memcpy(wide_ref, t.ref.data(), t.ref.size() * 4);
    return RAYRS_OK;
}

int rayrs_scene_export_hot_tree(const rayrs_scene* scene, double* wide_box, uint32_t* wide_ref) {
    if (!scene || !scene->flat.has_hot) return RAYRS_INVALID_ARG;
    const WalkTree& t = scene->flat.gate_hot;
    if (wide_box && !t.box.empty()) std::memcpy(wide_box, t.box.data(), t.box.size() * 8);
    if (wide_ref && !t.ref.empty()) std::memcpy(wide_ref, t.ref.data(), t.ref.size() * 4);
    return RAYRS_OK;
}

int rayrs_scene_clone_to_device(const rayrs_scene* scene, int device, rayrs_scene** out) {
    RAYRS_GUARDED({
        if (!scene || !out || device < 0) return RAYRS_INVALID_ARG;
        *out = nullptr;
        std::unique_ptr<rayrs_scene> s(new rayrs_scene());
        s->flat = scene->flat;
        s->surfaces = scene->surfaces;
        s->n_objects = scene->n_objects;
        s->tuning = scene->tuning;
        s->lab = scene->lab;
        s->device = device;
        scene_configure_local(s.get());
        const int st = scene_upload(s.get());
        if (st != RAYRS_OK) {
            scene_free_device(s.get());
            return st;
        }
        *out = s.release();
        return RAYRS_OK;
    })
}

int rayrs_scene_device(const rayrs_scene* scene) { return scene ? scene->device : -1; }

static int scene_quiesce(rayrs_scene* scene) {  // settings change between renders, never under one
    if (scene->device >= 0) {
        HIP_TRY(hipSetDevice(scene->device));
        if (scene->pending && scene->last_stream) {
            HIP_TRY(hipStreamSynchronize(scene->last_stream));
            scene->pending = false;
        }
    }
    return RAYRS_OK;
}

int rayrs_scene_set_tuning(rayrs_scene* scene, const rayrs_tuning* tuning) {
    if (!scene || !tuning) return RAYRS_INVALID_ARG;
    if (tuning->local_pool > 1u) return RAYRS_INVALID_ARG;
    const int st = scene_quiesce(scene);
    if (st != RAYRS_OK) return st;
    scene->tuning = *tuning;
    return RAYRS_OK;
}

#ifdef RAYRS_LAB_TICKS
// development build only (make LAB=1): the tick counters of the last render
extern "C" int rayrs_lab_ticks(rayrs_scene* scene, uint64_t out[16]) {
    if (!scene || !out || scene->device < 0) return RAYRS_INVALID_ARG;
    HIP_TRY(hipSetDevice(scene->device));
    Counters c;
    HIP_TRY(hipMemcpy(&c, scene->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; i++) out[i] = c.lab_ticks[i];
    return RAYRS_OK;
}
#endif

// rayrs_lab.h: HIP-event times of the last render's path rounds, three per round (traversal, hit, miss kernel; the
// local-pool route: its launch, 0, 0).  Returns the number of rounds; writes at most cap_rounds of them.
extern "C" int rayrs_lab_round_ms(rayrs_scene* scene, float* out, uint32_t cap_rounds) {
    if (!scene || scene->device < 0 || scene->pending) return RAYRS_INVALID_ARG;
    HIP_TRY(hipSetDevice(scene->device));
    const rayrs_scene::Pool& pl = scene->pool;
    for (uint32_t r = 0; r < pl.timed_rounds && r < cap_rounds && out; r++) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pl.ev_round[4 * r], pl.ev_round[4 * r + 1]));
        out[3 * r] = ms, out[3 * r + 1] = out[3 * r + 2] = 0.f;
        if (scene->last_local) continue;
        HIP_TRY(hipEventElapsedTime(&ms, pl.ev_round[4 * r + 1], pl.ev_round[4 * r + 2]));
        out[3 * r + 1] = ms;
        HIP_TRY(hipEventElapsedTime(&ms, pl.ev_round[4 * r + 2], pl.ev_round[4 * r + 3]));
        out[3 * r + 2] = ms;
    }
    return (int)pl.timed_rounds;
}

// rayrs_lab.h: the kernels' development knobs (tests/ and scripts/ubench/ only)
int rayrs_lab_set(rayrs_scene* scene, const rayrs_lab_tuning* lab) {
    if (!scene || !lab) return RAYRS_INVALID_ARG;
    if (lab->stack_lds > 64u) return RAYRS_INVALID_ARG;  // 4 x 64 lanes x 65 entries x 4 B: what a workgroup's LDS can spare
    if (lab->hot_group != 0u && lab->hot_group != 0xffffffffu) return RAYRS_INVALID_ARG;
    if (lab->static_pct > 100u || lab->refill_min > 64u || lab->leaf_min > 64u || lab->leaf_wait > 64u || lab->flat_blocks_per_cu > 64u || lab->eager_light > 1u || lab->force_rccl > 1u || lab->gate_tree > 1u)
        return RAYRS_INVALID_ARG;
    if (lab->local_reserve != 0u && (lab->local_reserve < 8u || lab->local_reserve > 4096u)) return RAYRS_INVALID_ARG;
    if (lab->local_segment_items != 0u && lab->local_segment_items < 65536u) return RAYRS_INVALID_ARG;
    const int st = scene_quiesce(scene);
    if (st != RAYRS_OK) return st;
    scene->lab = *lab;
    if (scene->device >= 0) return scene_configure_traversal(scene);
    return RAYRS_OK;
}

uint32_t rayrs_frame_sample_chunk(uint32_t x_pixels, uint32_t y_pixels, uint32_t spp, uint32_t requested) {
    if (spp == 0) return 0;
    uint64_t chunk = requested ? requested : 4u;
    const uint64_t pixels = (uint64_t)((x_pixels + 7u) / 8u) * ((y_pixels + 7u) / 8u) * 64u;  // whole 8x8 tiles
    while (chunk < spp && pixels * ((spp + chunk - 1) / chunk) > (1ull << 30)) chunk *= 2;
    return chunk >= spp ? 0u : (uint32_t)chunk;  // 0 = one sequential sum per pixel (the reference's order)
}

uint32_t rayrs_abi_version(void) { return RAYRS_ABI_VERSION; }

uint32_t rayrs_abi_layout(uint32_t* out, uint32_t cap) {
    std::vector<uint32_t> t;
    t.push_back(RAYRS_ABI_VERSION);  // (a binding that validates itself against this table fails on a version change too)
#define RAYRS_STRUCT(T, N) t.push_back((uint32_t)sizeof(T)), t.push_back(N)
#define RAYRS_FIELD(T, F) t.push_back((uint32_t)offsetof(T, F))
    RAYRS_STRUCT(rayrs_material, 7);
    RAYRS_FIELD(rayrs_material, kind), RAYRS_FIELD(rayrs_material, metallic), RAYRS_FIELD(rayrs_material, color);
    RAYRS_FIELD(rayrs_material, spec_color), RAYRS_FIELD(rayrs_material, alpha), RAYRS_FIELD(rayrs_material, ior);
    RAYRS_FIELD(rayrs_material, r0);
    RAYRS_STRUCT(rayrs_emission, 4);
    RAYRS_FIELD(rayrs_emission, emissive), RAYRS_FIELD(rayrs_emission, pad), RAYRS_FIELD(rayrs_emission, strength);
    RAYRS_FIELD(rayrs_emission, color);
    RAYRS_STRUCT(rayrs_camera, 9);
    RAYRS_FIELD(rayrs_camera, origin), RAYRS_FIELD(rayrs_camera, e_x), RAYRS_FIELD(rayrs_camera, e_y);
    RAYRS_FIELD(rayrs_camera, z), RAYRS_FIELD(rayrs_camera, width), RAYRS_FIELD(rayrs_camera, height);
    RAYRS_FIELD(rayrs_camera, ppc), RAYRS_FIELD(rayrs_camera, x_pixels), RAYRS_FIELD(rayrs_camera, y_pixels);
    RAYRS_STRUCT(rayrs_scene_info_t, 26);
    RAYRS_FIELD(rayrs_scene_info_t, n_objects), RAYRS_FIELD(rayrs_scene_info_t, n_interior);
    RAYRS_FIELD(rayrs_scene_info_t, n_prims), RAYRS_FIELD(rayrs_scene_info_t, root_ref);
    RAYRS_FIELD(rayrs_scene_info_t, depth), RAYRS_FIELD(rayrs_scene_info_t, compact);
    RAYRS_FIELD(rayrs_scene_info_t, n_surfaces), RAYRS_FIELD(rayrs_scene_info_t, node_bytes);
    RAYRS_FIELD(rayrs_scene_info_t, prim_bytes), RAYRS_FIELD(rayrs_scene_info_t, device_bytes);
    RAYRS_FIELD(rayrs_scene_info_t, root_box), RAYRS_FIELD(rayrs_scene_info_t, build_seconds);
    RAYRS_FIELD(rayrs_scene_info_t, n_wide), RAYRS_FIELD(rayrs_scene_info_t, wide_root_ref);
    RAYRS_FIELD(rayrs_scene_info_t, wide_depth), RAYRS_FIELD(rayrs_scene_info_t, local_pool);
    RAYRS_FIELD(rayrs_scene_info_t, gate_n_wide), RAYRS_FIELD(rayrs_scene_info_t, gate_root_ref);
    RAYRS_FIELD(rayrs_scene_info_t, gate_depth);
    RAYRS_FIELD(rayrs_scene_info_t, hot_n_wide), RAYRS_FIELD(rayrs_scene_info_t, hot_root_ref);
    RAYRS_FIELD(rayrs_scene_info_t, hot_depth), RAYRS_FIELD(rayrs_scene_info_t, hot_first);
    RAYRS_FIELD(rayrs_scene_info_t, hot_count), RAYRS_FIELD(rayrs_scene_info_t, hot_pad);
    RAYRS_FIELD(rayrs_scene_info_t, hot_box);
    RAYRS_STRUCT(rayrs_render_params, 9);
    RAYRS_FIELD(rayrs_render_params, spp), RAYRS_FIELD(rayrs_render_params, max_bounces);
    RAYRS_FIELD(rayrs_render_params, seed), RAYRS_FIELD(rayrs_render_params, sample_chunk);
    RAYRS_FIELD(rayrs_render_params, tile_rank), RAYRS_FIELD(rayrs_render_params, tile_ranks);
    RAYRS_FIELD(rayrs_render_params, out_format), RAYRS_FIELD(rayrs_render_params, count_work);
    RAYRS_FIELD(rayrs_render_params, fast_traversal);
    RAYRS_STRUCT(rayrs_render_stats, 33);
    RAYRS_FIELD(rayrs_render_stats, rays), RAYRS_FIELD(rayrs_render_stats, paths);
    RAYRS_FIELD(rayrs_render_stats, nan_pixels), RAYRS_FIELD(rayrs_render_stats, neg_pixels);
    RAYRS_FIELD(rayrs_render_stats, interior_visits), RAYRS_FIELD(rayrs_render_stats, tri_tests);
    RAYRS_FIELD(rayrs_render_stats, sphere_tests), RAYRS_FIELD(rayrs_render_stats, plane_tests);
    RAYRS_FIELD(rayrs_render_stats, escaped_paths), RAYRS_FIELD(rayrs_render_stats, step_wave);
    RAYRS_FIELD(rayrs_render_stats, step_lane), RAYRS_FIELD(rayrs_render_stats, inner_wave);
    RAYRS_FIELD(rayrs_render_stats, leaf_wave), RAYRS_FIELD(rayrs_render_stats, interior_ticks);
    RAYRS_FIELD(rayrs_render_stats, leaf_ticks), RAYRS_FIELD(rayrs_render_stats, kernel_ms);
    RAYRS_FIELD(rayrs_render_stats, total_ms), RAYRS_FIELD(rayrs_render_stats, kernel_launches);
    RAYRS_FIELD(rayrs_render_stats, trace_ms), RAYRS_FIELD(rayrs_render_stats, refill_ticks);
    RAYRS_FIELD(rayrs_render_stats, surface_hits), RAYRS_FIELD(rayrs_render_stats, direct_rays);
    RAYRS_FIELD(rayrs_render_stats, hit_ms), RAYRS_FIELD(rayrs_render_stats, miss_ms);
    RAYRS_FIELD(rayrs_render_stats, local_pool), RAYRS_FIELD(rayrs_render_stats, exact_walk);
    RAYRS_FIELD(rayrs_render_stats, hot_group), RAYRS_FIELD(rayrs_render_stats, stats_pad);
    RAYRS_FIELD(rayrs_render_stats, pre_rays), RAYRS_FIELD(rayrs_render_stats, pre_root_records);
    RAYRS_FIELD(rayrs_render_stats, hot_lane);
    RAYRS_FIELD(rayrs_render_stats, hot_prim_tests), RAYRS_FIELD(rayrs_render_stats, hot_tri_divided);
    RAYRS_STRUCT(rayrs_tuning, 2);
    RAYRS_FIELD(rayrs_tuning, pool_slots), RAYRS_FIELD(rayrs_tuning, local_pool);
#undef RAYRS_STRUCT
#undef RAYRS_FIELD
    for (uint32_t i = 0; i < cap && i < t.size(); i++) out[i] = t[i];
    return (uint32_t)t.size();
}

// ------------------------------------------------------------------ Camera

int rayrs_camera_new(const double origin[3], const double up[3], const double lookat[3], double fov, double width,
                     double height, uint32_t ppi, rayrs_camera* out) {
    return camera_new(origin, up, lookat, fov, width, height, ppi, out);
}

// ------------------------------------------------------------------ render

static SceneDev make_scene_dev(const rayrs_scene* s, bool exact) {
    SceneDev sc;
    std::memset(&sc, 0, sizeof(sc));
    const int which = s->walk_index(exact);
    const WalkTree& t = s->tree(which);
    const rayrs_scene::Walk& w = s->trav[which];
    sc.hot = which == 2 ? s->d_hot : nullptr;
    sc.nodes = w.d_nodes;
    sc.prims = s->d_prims;
    sc.surfaces = s->d_surfaces;
    sc.hdri = s->d_hdri;
    sc.hdri_w = s->flat.hdri_w;
    sc.hdri_h = s->flat.hdri_h;
    sc.hdri_wm1 = (double)(s->flat.hdri_w - 1u);
    sc.hdri_hm1 = (double)(s->flat.hdri_h - 1u);
    sc.root_ref = t.root_ref;
    sc.stack_depth = t.depth ? t.depth : 1;
    sc.stack_lds = w.stack_lds;
    sc.hot_records = w.hot_records;
    sc.leafq = w.leafq;
    sc.n_surfaces = (uint32_t)s->surfaces.size();
    for (int i = 0; i < 6; i++) sc.root_box[i] = s->flat.root_box[i];
    sc.t0 = s->flat.t0;
    sc.t1 = s->flat.t1;
    sc.exact = exact ? 1u : 0u;
    return sc;
}

// The fast walk's leaf boxes are a bet on the reference's arithmetic that was measured to hold for rays from nearby and
// to fail, a few times in 10^4, for rays aimed along a primitive's plane from far away (include/rayrs_hip.h
// fast_traversal; profiles/r04_tight_leaves.txt).  What decides is the distance in PRIMITIVE sizes (the error of the
// computed hit point is about eps * distance / angle against a widening of 1/64 of the primitive): failures were seen from
// 6,000 primitive sizes up, none within 4,000 (sheets of 6 ... 400 quads per side: scripts/fuzz_traversal.py).
// A frame whose camera is farther from the root Node's box than RAYRS_FAR_DIAGONALS times that box's diagonal, or than
// RAYRS_FAR_PRIMITIVES times the scene's small primitives (the 5th percentile of their largest extents), takes the
// default walk whatever was asked (rayrs_render_stats.exact_walk says which walk a frame took).  This guards PRIMARY
// rays only: a bounced ray from a large surface to a finely tessellated one (the headline scene's floor to its mesh)
// is thousands of small-primitive sizes long and makes the bet all the same -- the fast walk stays a bet.
constexpr double RAYRS_FAR_DIAGONALS = 8.0;
constexpr double RAYRS_FAR_PRIMITIVES = 4096.0;
static bool camera_is_far(const rayrs_scene* s, const rayrs_camera* c) {
    const double* b = s->flat.root_box;
    double d2 = 0.0, e2 = 0.0;
    for (int a = 0; a < 3; a++) {
        const double lo = b[2 * a], hi = b[2 * a + 1], o = c->origin[a];
        const double out = o < lo ? lo - o : (o > hi ? o - hi : 0.0);
        d2 += out * out;
        e2 += (hi - lo) * (hi - lo);
    }
    // (a box or an origin that is not a number compares false: the walk that was asked for, as for any other frame)
    const double small = s->flat.small_extent;
    return d2 > RAYRS_FAR_DIAGONALS * RAYRS_FAR_DIAGONALS * e2 || (small > 0.0 && d2 > RAYRS_FAR_PRIMITIVES * RAYRS_FAR_PRIMITIVES * small * small);
}

static CameraDev make_camera_dev(const rayrs_camera* c) {
    CameraDev cam;
    std::memset(&cam, 0, sizeof(cam));
    for (int i = 0; i < 3; i++) {
        cam.origin[i] = c->origin[i];
        cam.e_x[i] = c->e_x[i];
        cam.e_y[i] = c->e_y[i];
        cam.z[i] = c->z[i];
    }
    cam.width = c->width;
    cam.height = c->height;
    cam.ppc = (double)c->ppc;  // `self.ppc as f64`, lib.rs:206
    cam.W = c->x_pixels;
    cam.H = c->y_pixels;
    return cam;
}

int rayrs_render_launch(rayrs_scene* scene, const rayrs_camera* camera, const rayrs_render_params* params,
                        void* out_device, void* hip_stream) {
    RAYRS_GUARDED({
    if (!scene || !camera || !params || !out_device) return RAYRS_INVALID_ARG;
    if (scene->device < 0) return RAYRS_NO_DEVICE;
    if (params->spp == 0 || camera->x_pixels == 0 || camera->y_pixels == 0) return RAYRS_INVALID_ARG;
    if (camera->x_pixels > 65535u || camera->y_pixels > 65535u) return RAYRS_UNSUPPORTED;  // TailSlot::pix is 16 + 16 bits
    // a path's bounce count and RNG draw index travel as 16 bits each (15 + 16 in the local pool); a bounce draws at
    // most four numbers (material.rs:579 + :1009-1011 + lib.rs:539), so 8000 bounces stay below 2^15 and 2^16
    if (params->max_bounces > 8000u) return RAYRS_UNSUPPORTED;
    if (params->spp > SLOT_SAMPLE_MASK) return RAYRS_UNSUPPORTED;     // a slot's sample cursor has 30 bits
    if (params->tile_ranks == 0 || params->tile_rank >= params->tile_ranks) return RAYRS_INVALID_ARG;
    if (params->out_format != RAYRS_OUT_F32 && params->out_format != RAYRS_OUT_F64) return RAYRS_INVALID_ARG;
    if (params->fast_traversal > 1u) return RAYRS_INVALID_ARG;
    HIP_TRY(hipSetDevice(scene->device));
    hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
    if (scene->pending) {  // one render in flight per scene: its counters and partial sums are shared
        HIP_TRY(hipStreamSynchronize(scene->last_stream));
        scene->pending = false;
    }
    const rayrs_lab_tuning& lab = scene->lab;

    RenderDev rp;
    std::memset(&rp, 0, sizeof(rp));
    rp.spp = params->spp;
    rp.max_bounces = params->max_bounces;
    rp.seed = params->seed;
    rp.chunk = (params->sample_chunk == 0 || params->sample_chunk >= params->spp) ? params->spp : params->sample_chunk;
    rp.nchunks = (rp.spp + rp.chunk - 1) / rp.chunk;
    rp.tile_rank = params->tile_rank;
    rp.tile_ranks = params->tile_ranks;
    rp.tiles_x = (camera->x_pixels + 7) / 8;
    rp.tiles_y = (camera->y_pixels + 7) / 8;
    const uint64_t n_tiles = (uint64_t)rp.tiles_x * rp.tiles_y;
    const uint64_t n_local = n_tiles > rp.tile_rank ? (n_tiles - rp.tile_rank + rp.tile_ranks - 1) / rp.tile_ranks : 0;
    rp.n_local_tiles = (uint32_t)n_local;
    rp.total_items = n_local * rp.nchunks * 64ull;
    rp.inv_nchunks = 1.0 / (double)rp.nchunks;
    rp.inv_tiles_x = 1.0 / (double)rp.tiles_x;
    if (rp.total_items >= (1ull << 32)) return RAYRS_UNSUPPORTED;
    rp.refill_min = lab.refill_min ? lab.refill_min : 52u;
    // (a leaf phase once this many lanes stand on a leaf: 24 where a leaf is one primitive -- the fast walk's tree: 612 -> 604 ms
    // of traversal on the headline frame against 32 --, 32 where it is a group of up to four -- the default walk: 988 -> 962 ms
    // against 24; it is set below, once the walk is known)
    rp.count_work = params->count_work ? 1u : 0u;
    rp.out_format = params->out_format;
    rp.out = out_device;
    rp.counters = scene->d_counters;

    // Item sums: 24 bytes per (pixel, chunk) item, added per pixel in chunk order by the resolve kernel.  The streaming
    // kernels finish items in no particular order, so the array covers the frame.  The local-pool route renders the frame
    // as a sequence of launches over segments of whole tiles and resolves each segment behind its launch: one segment's
    // worth is all it needs (config 4: 3.2 GB instead of 25.8).
    const bool use_local = scene->local_ok && scene->tuning.local_pool != 1u;
    const uint64_t tile_items = (uint64_t)rp.nchunks * 64u;
    uint64_t seg_want = lab.local_segment_items ? lab.local_segment_items : LOCAL_SEGMENT_ITEMS;
    // (a frame stays within LOCAL_MAX_SEGMENTS launches: larger segments rather than a refused frame)
    if ((rp.total_items + seg_want - 1) / seg_want > LOCAL_MAX_SEGMENTS) seg_want = (rp.total_items + LOCAL_MAX_SEGMENTS - 1) / LOCAL_MAX_SEGMENTS;
    const uint64_t seg_tiles = (seg_want + tile_items - 1) / tile_items > 0 ? (seg_want + tile_items - 1) / tile_items : 1;
    const uint64_t seg_items = seg_tiles * tile_items;
    const uint64_t partial_need = use_local && seg_items < rp.total_items ? seg_items : rp.total_items;
    if (use_local && seg_items >= (1ull << 32)) return RAYRS_UNSUPPORTED;  // (one tile's chunks alone: spp beyond 2^27)
    if (partial_need > scene->partial_items) {
        if (scene->d_partial) HIP_TRY(hipFree(scene->d_partial));
        scene->d_partial = nullptr;
        scene->partial_items = 0;
        HIP_TRY(hipMalloc((void**)&scene->d_partial, (size_t)partial_need * 3 * sizeof(double)));
        scene->partial_items = (size_t)partial_need;
    }
    rp.partial = scene->d_partial;
    rp.partial_item0 = 0;

    const bool exact = params->fast_traversal == 0u || camera_is_far(scene, camera);
    scene->last_exact = exact;
    rp.leaf_min = lab.leaf_min ? lab.leaf_min : (exact ? 48u : lab.gate_tree ? 32u : 24u);
    // (the default walk: its lanes walk on while their leaf groups wait, so a leaf phase may wait for more of them -- or for
    // leaf_wait lanes that can do nothing else; scripts/sim/walk_sched_sim.py, swept on the GPU: profiles/r06_leaf_queue.txt)
    rp.leaf_wait = lab.leaf_wait ? lab.leaf_wait : 16u;
    const SceneDev sc = make_scene_dev(scene, exact);
    // (pre-tested rays -- a scene with a hot group, wavefront.hip finish_rays -- wanted 56 while a lane stood idle on its
    // leaf: 664 -> 659 ms of traversal, profiles/r06_tuning_sweep.txt; with the leaf groups set aside 52 is best again:
    // 626 -> 619 ms, profiles/r06_leaf_queue.txt)
    const CameraDev cam = make_camera_dev(camera);

    // ---- path pool.  A traversal launch works through the whole pool, and its ramp-up
    // and drain are a fixed cost, so large pools win even when that leaves only one or two
    // items per slot -- up to the point where the hit and miss kernels lose more to the larger
    // footprint.  Swept on the headline frame with one-line slots: 24 M slots 1520 ms, 32 M 1516,
    // 48 M 1497, 64 M 1467, 96 M 1468, 128 M 1481, 192 M 1498 (round 1, 192-byte slots: 32 M).
    // Swept again on round 3's kernels (the traversal kernel faster, its fixed cost per launch the same): 64 M 1369 ms,
    // 80 M 1372, 96 M 1368, 112 M 1353, 128 M 1356, 160 M 1353, 192 M 1350: 112 M slots (18 GB of the 288 GB), 84 rounds.
    // And on round 6's (the hit kernel at three waves per SIMD, the cheap queries answered by the kernels that make the rays):
    // 80 M 1298 ms, 96 M 1290, 112 M 1276, 128 M 1278, 144 M 1267, 160 M 1250, 192 M 1250, 224 M 1246, 256 M 1242, 320 M 1242
    // (profiles/r06_leaf_queue.txt (12)): 256 M slots (43 GB), 52 rounds.
    // A frame should also last some tens of rounds, or filling and draining the pool is all it does: at most one
    // slot per 12 samples -- which is what a one-eighth tile share of the headline frame gets (44.7 M: 33.5 M 189 ms,
    // 48 M 184, 64 M 186).
    constexpr uint64_t POOL_MAX_SLOTS = 1ull << 28;
    uint64_t np64 = rp.total_items;
    if (np64 > POOL_MAX_SLOTS) np64 = POOL_MAX_SLOTS;
    {
        const uint64_t samples = n_local * 64ull * rp.spp;
        const uint64_t by_work = samples / 12u > (1ull << 20) ? samples / 12u : (1ull << 20);
        if (np64 > by_work) np64 = by_work;
    }
    if (scene->tuning.pool_slots) np64 = scene->tuning.pool_slots;
    if (np64 > rp.total_items) np64 = rp.total_items;
    const uint64_t live_total = np64;

    const bool compact = scene->flat.compact;
    const bool count = params->count_work != 0;
    uint32_t trav_bpc = (uint32_t)scene->trav[scene->walk_index(exact)].blocks_per_cu;
    if (lab.trav_blocks_per_cu && lab.trav_blocks_per_cu < trav_bpc) trav_bpc = lab.trav_blocks_per_cu;
    const uint32_t trav_blocks = (uint32_t)scene->cu_count * trav_bpc;
    uint32_t static_pct = lab.static_pct ? lab.static_pct : 50u;
    if (static_pct > 100) static_pct = 100;

    // A path's light lives in a side array and only while it is not +0 (wavefront.h PathSlot).  Where a surface
    // emits, paths do get light, and the hit and miss kernels request the side array's entry together with the
    // slot instead of after it.
    bool eager_light = lab.eager_light != 0u;
    for (const SurfaceDev& sf : scene->surfaces)
        if (sf.emit[0] != 0.0 || sf.emit[1] != 0.0 || sf.emit[2] != 0.0) eager_light = true;
    constexpr size_t slot_bytes = sizeof(PathSlot) + 4 * sizeof(double);  // slot + its entry of the light array
    rayrs_scene::Pool& pl = scene->pool;
    WfDev wf = pl.wf;
    uint32_t flat_blocks = 0;
    if (!use_local && rp.total_items > 0) {  // (the local-pool route keeps its paths in LDS)
        const uint32_t np = (uint32_t)((live_total + 1023ull) & ~1023ull);  // whole windows
        const size_t block_bytes = (size_t)np * (slot_bytes + 1u);
        if (block_bytes > pl.block_bytes || !pl.block) {
            if (pl.block) HIP_TRY(hipFree(pl.block));
            pl.block = nullptr;
            pl.block_bytes = 0;
            HIP_TRY(hipMalloc(&pl.block, block_bytes));
            pl.block_bytes = block_bytes;
        }
        pl.wf.slots = static_cast<PathSlot*>(pl.block);
        pl.wf.light = reinterpret_cast<double*>(pl.wf.slots + np);
        pl.wf.state = reinterpret_cast<uint8_t*>(pl.wf.light + (size_t)np * 4u);
        wf = pl.wf;
        wf.np = np;
        const uint32_t n_windows = np / wf_window_slots();
        {
            // whole round-robin rounds covering about static_pct % of the pool's windows
            const uint64_t n_waves = (uint64_t)trav_blocks * 4u;
            rp.static_windows = (uint32_t)((uint64_t)n_windows * static_pct / 100u / n_waves * n_waves);
        }

        // the gen, hit and miss kernels run with this one grid, so wave w means the same windows in all three: one
        // wave per window, at most 8 ... 24 workgroups per CU (below)
        uint32_t fb = (n_windows + 3u) / 4u;
        // (8 ... 24 workgroups per CU, so that a wave has about nine windows: with the hit kernel at three resident workgroups
        // per CU a finer grid evens out the end of a launch -- hit 411 -> 404 ms on the headline's 218 k windows at 24 -- but
        // a wave that gets three windows spends its time on its first and last batch: config 3's 87 k windows want 8 or 9
        // (198 against 200.5 ms at 24); profiles/r06_leaf_queue.txt (9))
        uint32_t per_cu = lab.flat_blocks_per_cu;
        if (!per_cu) {
            per_cu = n_windows / (36u * (uint32_t)scene->cu_count);
            per_cu = per_cu < FLAT_BLOCKS_PER_CU_MIN ? FLAT_BLOCKS_PER_CU_MIN : per_cu > FLAT_BLOCKS_PER_CU_MAX ? FLAT_BLOCKS_PER_CU_MAX : per_cu;
        }
        const uint32_t flat_cap = (uint32_t)scene->cu_count * per_cu;
        if (fb > flat_cap) fb = flat_cap;
        flat_blocks = fb;
        wf.n_flat_waves = fb * 4u;
        if (wf.n_flat_waves > pl.wave_items_cap) {
            if (pl.d_wave_items) HIP_TRY(hipFree(pl.d_wave_items));
            pl.d_wave_items = nullptr;
            pl.wave_items_cap = 0;
            HIP_TRY(hipMalloc((void**)&pl.d_wave_items, (size_t)wf.n_flat_waves * 2 * sizeof(unsigned long long)));
            pl.wave_items_cap = wf.n_flat_waves;
        }
        wf.wave_items = pl.d_wave_items;
        wf.trav_threads = trav_blocks * 256u;
        {
            const size_t words = (size_t)(sc.stack_depth - sc.stack_lds) * wf.trav_threads;
            if (words > pl.stack_spill_words) {
                if (pl.d_stack_spill) HIP_TRY(hipFree(pl.d_stack_spill));
                pl.d_stack_spill = nullptr;
                pl.stack_spill_words = 0;
                HIP_TRY(hipMalloc((void**)&pl.d_stack_spill, words * sizeof(uint32_t)));
                pl.stack_spill_words = words;
            }
            wf.stack_spill = pl.d_stack_spill;
        }
    }
    rp.next_item = scene->d_next_item;
    pl.timed_rounds = 0;

    HIP_TRY(hipMemsetAsync(scene->d_counters, 0, sizeof(Counters), stream));
    HIP_TRY(hipMemsetAsync(scene->d_next_item, 0, sizeof(unsigned long long), stream));
    if (use_local) HIP_TRY(hipMemsetAsync(scene->d_local_items, 0, LOCAL_MAX_SEGMENTS * sizeof(unsigned long long), stream));
    HIP_TRY(hipEventRecord(scene->ev[0], stream));
    scene->rounds = 0;
    scene->last_local = use_local;
    auto round_events = [&](size_t n) -> int {
        while (pl.ev_round.size() < n) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            pl.ev_round.push_back(e);
        }
        return RAYRS_OK;
    };
    if (use_local && rp.total_items > 0) {
        // ---- one launch per segment of the frame's items; a launch ends when its last path has (local_pool.hip)
        const uint64_t n_seg = (rp.total_items + seg_items - 1) / seg_items;
        // LP_WPS workgroups of four waves per CU; fewer when the frame has fewer items than resident paths
        uint32_t blocks = (uint32_t)scene->cu_count * (uint32_t)scene->local_blocks_per_cu;
        {
            const uint64_t most_items = rp.total_items < seg_items ? rp.total_items : seg_items;
            const uint64_t want = (most_items + 4u * LP_PATHS_PER_WAVE - 1) / (4u * LP_PATHS_PER_WAVE);
            if (want < blocks) blocks = (uint32_t)(want ? want : 1);
        }
        const size_t paths = (size_t)blocks * 4u * LP_PATHS_PER_WAVE;
        if (paths > scene->local_light_paths) {
            if (scene->d_local_light) HIP_TRY(hipFree(scene->d_local_light));
            scene->d_local_light = nullptr;
            scene->local_light_paths = 0;
            HIP_TRY(hipMalloc((void**)&scene->d_local_light, paths * 4 * sizeof(double)));
            scene->local_light_paths = paths;
        }
        for (uint64_t seg = 0; seg < n_seg; seg++) {
            LocalDev lp;
            lp.light = scene->d_local_light;
            lp.next_item = scene->d_local_items + seg;
            lp.item_base = seg * seg_items;
            lp.item_count = rp.total_items - lp.item_base < seg_items ? rp.total_items - lp.item_base : seg_items;
            RenderDev rseg = rp;
            rseg.partial_item0 = lp.item_base;
            {
                const uint64_t share = lp.item_count / ((uint64_t)blocks * 4u * 16u);  // a sixteenth of a wave's share
                lp.reserve = (uint32_t)(share < 8u ? 8u : share > 256u ? 256u : share);
                if (lab.local_reserve) lp.reserve = lab.local_reserve;
                lp.pad = 0;
            }
            {
                const int st = round_events(4 * (size_t)(seg + 1));
                if (st != RAYRS_OK) return st;
            }
            HIP_TRY(hipEventRecord(pl.ev_round[4 * seg], stream));
            HIP_TRY(lp_launch(compact, count, sc, scene->local, cam, rseg, lp, blocks, stream));
            HIP_TRY(hipEventRecord(pl.ev_round[4 * seg + 1], stream));
            // the segment's tiles, resolved behind its launch (the next segment reuses the item-sum array)
            HIP_TRY(launch_resolve(cam, rseg, (uint32_t)(seg * seg_tiles), (uint32_t)(lp.item_count / tile_items), stream));
            pl.timed_rounds = (uint32_t)seg + 1;
        }
        scene->rounds = (uint32_t)n_seg;
    } else if (rp.total_items > 0) {
        HIP_TRY(wf_launch_init(wf, (uint32_t)live_total, stream));
        HIP_TRY(wf_launch_gen(compact, sc, cam, rp, wf, flat_blocks, stream));  // initial fill; later samples start in hit/miss
        pl.h_live[0] = pl.h_live[1] = (uint32_t)live_total;
        // Rounds are enqueued in batches; the live-slot count of batch b is read back while batch b+1 is
        // already queued, so the GPU never waits for the host.  Rounds behind the frame's last one find
        // live_slots == 0 and return at once; batches shrink from 16 rounds to 4 once fewer than an eighth of
        // the slots have work, so that at most 7 such rounds are queued after the end.
        constexpr uint32_t MAX_TIMED = 8192;
        uint32_t it = 0;
        uint32_t batch = 16;
        for (uint32_t b = 0;; b++) {
            for (uint32_t k = 0; k < batch; k++, it++) {
                const bool timed = it < MAX_TIMED;
                // four events per round: before the traversal kernel, after it, after the hit kernel, after the miss kernel
                if (timed) {
                    const int st = round_events(4 * (size_t)(it + 1));
                    if (st != RAYRS_OK) return st;
                    HIP_TRY(hipEventRecord(pl.ev_round[4 * it], stream));
                }
                HIP_TRY(wf_launch_trav(compact, count, sc, rp, wf, trav_blocks, stream));
                if (timed) HIP_TRY(hipEventRecord(pl.ev_round[4 * it + 1], stream));
                HIP_TRY(wf_launch_hit(compact, eager_light, sc, cam, rp, wf, flat_blocks, stream));
                if (timed) HIP_TRY(hipEventRecord(pl.ev_round[4 * it + 2], stream));
                HIP_TRY(wf_launch_miss(compact, eager_light, sc, cam, rp, wf, flat_blocks, stream));
                if (timed) {
                    HIP_TRY(hipEventRecord(pl.ev_round[4 * it + 3], stream));
                    pl.timed_rounds = it + 1;
                }
            }
            HIP_TRY(hipMemcpyAsync(&pl.h_live[b & 1u], &wf.ctl->live_slots, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipEventRecord(pl.ev_batch[b & 1u], stream));
            if (b > 0) {
                HIP_TRY(hipEventSynchronize(pl.ev_batch[(b - 1u) & 1u]));
                const uint64_t seen = pl.h_live[(b - 1u) & 1u];
                if (seen == 0u) break;
                batch = seen * 8u < live_total ? 4u : 16u;
            }
            if (it > (1u << 26)) {
                g_last_error = "path rounds did not terminate";
                return RAYRS_HIP_ERROR;
            }
        }
        scene->rounds = it;
    }
    HIP_TRY(hipEventRecord(scene->ev[1], stream));
    if (!use_local) HIP_TRY(launch_resolve(cam, rp, 0u, rp.n_local_tiles, stream));
    HIP_TRY(hipEventRecord(scene->ev[2], stream));
    scene->last_stream = stream;
    scene->pending = true;
    scene->last_count = count;
    return RAYRS_OK;
    })
}

int rayrs_render_finish(rayrs_scene* scene, rayrs_render_stats* stats) {
    if (!scene) return RAYRS_INVALID_ARG;
    if (scene->device < 0) return RAYRS_NO_DEVICE;
    if (!scene->pending) return RAYRS_INVALID_ARG;
    HIP_TRY(hipSetDevice(scene->device));
    HIP_TRY(hipEventSynchronize(scene->ev[2]));
    scene->pending = false;
    if (stats) {
        Counters c;
        HIP_TRY(hipMemcpy(&c, scene->d_counters, sizeof(c), hipMemcpyDeviceToHost));
        std::memset(stats, 0, sizeof(*stats));
        stats->rays = c.rays;
        stats->paths = c.paths;
        stats->nan_pixels = c.nan_pixels;
        stats->neg_pixels = c.neg_pixels;
        stats->interior_visits = c.interior_visits;
        stats->tri_tests = c.tri_tests;
        stats->sphere_tests = c.sphere_tests;
        stats->plane_tests = c.plane_tests;
        stats->escaped_paths = c.escaped_paths;
        stats->step_wave = c.step_wave, stats->step_lane = c.step_lane, stats->inner_wave = c.inner_wave;
        stats->leaf_wave = c.leaf_wave, stats->interior_ticks = c.interior_ticks, stats->leaf_ticks = c.leaf_ticks;
        stats->refill_ticks = c.refill_ticks;
        for (int k = 0; k < 8; k++) stats->surface_hits[k] = c.surface_hits[k];
        stats->direct_rays = c.direct_rays;
        stats->pre_rays = c.pre_rays, stats->pre_root_records = c.pre_root_records, stats->hot_lane = c.hot_lane;
        stats->hot_prim_tests = c.hot_prim_tests, stats->hot_tri_divided = c.hot_tri_divided;
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, scene->ev[0], scene->ev[1]));
        stats->trace_ms = ms;
        HIP_TRY(hipEventElapsedTime(&ms, scene->ev[0], scene->ev[2]));
        stats->total_ms = ms;
        const rayrs_scene::Pool& pl = scene->pool;
        double t = 0.0, h = 0.0, m = 0.0;
        for (uint32_t r = 0; r < pl.timed_rounds; r++) {
            HIP_TRY(hipEventElapsedTime(&ms, pl.ev_round[4 * r], pl.ev_round[4 * r + 1]));
            t += ms;
            if (scene->last_local) continue;  // one kernel per segment
            HIP_TRY(hipEventElapsedTime(&ms, pl.ev_round[4 * r + 1], pl.ev_round[4 * r + 2]));
            h += ms;
            HIP_TRY(hipEventElapsedTime(&ms, pl.ev_round[4 * r + 2], pl.ev_round[4 * r + 3]));
            m += ms;
        }
        // rounds beyond the event pool (very long renders) are extrapolated from the timed ones
        if (pl.timed_rounds && scene->rounds > pl.timed_rounds) {
            const double f = (double)scene->rounds / (double)pl.timed_rounds;
            t *= f, h *= f, m *= f;
        }
        stats->kernel_ms = t;
        stats->hit_ms = h, stats->miss_ms = m;
        stats->local_pool = scene->last_local ? 1u : 0u;
        stats->exact_walk = (scene->last_exact || scene->last_local) ? 1u : 0u;
        stats->hot_group = (!scene->last_local && scene->last_exact && scene->walk_index(true) == 2) ? 1u : 0u;
        stats->kernel_launches = (uint64_t)scene->rounds;
    }
    return RAYRS_OK;
}

int rayrs_render(rayrs_scene* scene, const rayrs_camera* camera, const rayrs_render_params* params, void* out_host,
                 rayrs_render_stats* stats) {
    if (!scene || !camera || !params || !out_host) return RAYRS_INVALID_ARG;
    if (scene->device < 0) return RAYRS_NO_DEVICE;
    HIP_TRY(hipSetDevice(scene->device));
    const size_t elem = params->out_format == RAYRS_OUT_F64 ? 8 : 4;
    const size_t bytes = (size_t)camera->x_pixels * camera->y_pixels * 3 * elem;
    void* d_out = nullptr;
    HIP_TRY(hipMalloc(&d_out, bytes));
    // pixels of other ranks' tiles keep the caller's values
    hipError_t e = hipMemcpy(d_out, out_host, bytes, hipMemcpyHostToDevice);
    int st = e == hipSuccess ? RAYRS_OK : hip_fail(e, "hipMemcpy(out H2D)");
    if (st == RAYRS_OK) st = rayrs_render_launch(scene, camera, params, d_out, nullptr);
    if (st == RAYRS_OK) st = rayrs_render_finish(scene, stats);
    if (st == RAYRS_OK) {
        e = hipMemcpy(out_host, d_out, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) st = hip_fail(e, "hipMemcpy(out D2H)");
    }
    (void)hipFree(d_out);
    return st;
}

// ------------------------------------------------------------- self tests

namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
        return e == hipSuccess ? RAYRS_OK : hip_fail(e, "hipMalloc");
    }
    int upload(const void* src, size_t bytes) {
        int st = alloc(bytes);
        if (st != RAYRS_OK) return st;
        if (!bytes) return RAYRS_OK;
        hipError_t e = hipMemcpy(p, src, bytes, hipMemcpyHostToDevice);
        return e == hipSuccess ? RAYRS_OK : hip_fail(e, "hipMemcpy H2D");
    }
    int download(void* dst, size_t bytes) {
        if (!bytes) return RAYRS_OK;
        hipError_t e = hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost);
        return e == hipSuccess ? RAYRS_OK : hip_fail(e, "hipMemcpy D2H");
    }
};
#define ST_TRY(expr)                  \
    do {                              \
        int _s = (expr);              \
        if (_s != RAYRS_OK) return _s; \
    } while (0)
}  // namespace

int rayrs_test_math(int device, int fn, const double* x, const double* y, uint64_t n, double* out) {
    if (!x || !out) return RAYRS_INVALID_ARG;
    HIP_TRY(hipSetDevice(device));
    DevBuf dx, dy, dout;
    ST_TRY(dx.upload(x, n * 8));
    if (y) ST_TRY(dy.upload(y, n * 8));
    ST_TRY(dout.alloc(n * 8));
    if (n) HIP_TRY(launch_test_math(fn, (const double*)dx.p, y ? (const double*)dy.p : nullptr, n, (double*)dout.p, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    return dout.download(out, n * 8);
}

int rayrs_test_rng(int device, uint64_t seed, const uint64_t* pixel, const uint64_t* sample, const uint32_t* draw,
                   uint64_t n, uint64_t* out_bits) {
    if (!pixel || !sample || !draw || !out_bits) return RAYRS_INVALID_ARG;
    HIP_TRY(hipSetDevice(device));
    DevBuf dp, ds, dd, dout;
    ST_TRY(dp.upload(pixel, n * 8));
    ST_TRY(ds.upload(sample, n * 8));
    ST_TRY(dd.upload(draw, n * 4));
    ST_TRY(dout.alloc(n * 8));
    if (n)
        HIP_TRY(launch_test_rng(seed, (const uint64_t*)dp.p, (const uint64_t*)ds.p, (const uint32_t*)dd.p, n,
                                (uint64_t*)dout.p, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    return dout.download(out_bits, n * 8);
}

int rayrs_test_intersect(rayrs_scene* scene, const double* o, const double* d, uint64_t n, int exact, double* t,
                         int64_t* object) {
    if (!scene || !o || !d || !t || !object) return RAYRS_INVALID_ARG;
    if (scene->device < 0) return RAYRS_NO_DEVICE;
    HIP_TRY(hipSetDevice(scene->device));
    DevBuf dorg, ddir, dt, dprim;
    ST_TRY(dorg.upload(o, n * 24));
    ST_TRY(ddir.upload(d, n * 24));
    ST_TRY(dt.alloc(n * 8));
    ST_TRY(dprim.alloc(n * 8));
    const SceneDev sc = make_scene_dev(scene, exact != 0);
    DevBuf dspill;  // stack entries beyond the LDS part, one strip per thread of the launch
    const uint64_t threads = (n + 255) / 256 * 256;
    if (sc.stack_depth > sc.stack_lds) ST_TRY(dspill.alloc((size_t)(sc.stack_depth - sc.stack_lds) * threads * 4));
    if (n)
        HIP_TRY(launch_test_intersect(scene->flat.compact, sc, (const double*)dorg.p, (const double*)ddir.p, n,
                                      (double*)dt.p, (long long*)dprim.p, (uint32_t*)dspill.p, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    ST_TRY(dt.download(t, n * 8));
    ST_TRY(dprim.download(object, n * 8));
    for (uint64_t i = 0; i < n; i++)  // DFS slot -> object index in insertion order
        if (object[i] >= 0) object[i] = (int64_t)scene->flat.prim_object[(size_t)object[i]];
    return RAYRS_OK;
}

int rayrs_test_path_trace(rayrs_scene* scene, const rayrs_camera* camera, uint64_t seed, uint32_t max_bounces,
                          const uint32_t* pixel, const uint32_t* sample, uint64_t n, int exact, uint32_t cap,
                          uint32_t* n_queries, int64_t* object, double* t, double* throughput, uint32_t* draw, double* rgb) {
    if (!scene || !camera || !pixel || !sample || !n_queries || !object || !t || !throughput || !draw || !rgb || cap == 0)
        return RAYRS_INVALID_ARG;
    if (scene->device < 0) return RAYRS_NO_DEVICE;
    for (uint64_t i = 0; i < n; i++)
        if ((pixel[i] >> 16) >= camera->y_pixels || (pixel[i] & 0xffffu) >= camera->x_pixels) return RAYRS_INVALID_ARG;
    HIP_TRY(hipSetDevice(scene->device));
    DevBuf dpix, dsam, dn, dprim, dt, dthr, ddraw, drgb, dspill;
    ST_TRY(dpix.upload(pixel, n * 4));
    ST_TRY(dsam.upload(sample, n * 4));
    ST_TRY(dn.alloc(n * 4));
    ST_TRY(dprim.alloc(n * cap * 4));
    ST_TRY(dt.alloc(n * cap * 8));
    ST_TRY(dthr.alloc(n * cap * 24));
    ST_TRY(ddraw.alloc(n * cap * 4));
    ST_TRY(drgb.alloc(n * 24));
    HIP_TRY(hipMemset(dprim.p, 0xff, n * cap * 4));
    HIP_TRY(hipMemset(dt.p, 0, n * cap * 8));
    HIP_TRY(hipMemset(dthr.p, 0, n * cap * 24));
    HIP_TRY(hipMemset(ddraw.p, 0, n * cap * 4));
    const SceneDev sc = make_scene_dev(scene, exact != 0);
    const CameraDev cam = make_camera_dev(camera);
    const uint64_t threads = (n + 255) / 256 * 256;
    if (sc.stack_depth > sc.stack_lds) ST_TRY(dspill.alloc((size_t)(sc.stack_depth - sc.stack_lds) * threads * 4));
    if (n)
        HIP_TRY(launch_test_path_trace(scene->flat.compact, sc, cam, seed, max_bounces, (const uint32_t*)dpix.p,
                                       (const uint32_t*)dsam.p, n, cap, (uint32_t*)dn.p, (uint32_t*)dprim.p, (double*)dt.p,
                                       (double*)dthr.p, (uint32_t*)ddraw.p, (double*)drgb.p, (uint32_t*)dspill.p, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<uint32_t> prim(n * cap);
    ST_TRY(dn.download(n_queries, n * 4));
    ST_TRY(dprim.download(prim.data(), n * cap * 4));
    ST_TRY(dt.download(t, n * cap * 8));
    ST_TRY(dthr.download(throughput, n * cap * 24));
    ST_TRY(ddraw.download(draw, n * cap * 4));
    ST_TRY(drgb.download(rgb, n * 24));
    for (uint64_t k = 0; k < n * cap; k++)  // DFS slot -> object index in insertion order
        object[k] = prim[k] == 0xffffffffu ? -1 : (int64_t)scene->flat.prim_object[prim[k]];
    return RAYRS_OK;
}

int rayrs_test_material(int device, const rayrs_material* mat, const double* normal, const double* view,
                        const uint64_t* key, uint64_t n, int32_t* scattered, double* color, double* dir,
                        uint32_t* draws) {
    RAYRS_GUARDED({
    if (!mat || !normal || !view || !key || !scattered || !color || !dir || !draws) return RAYRS_INVALID_ARG;
    ObjectList tmp;
    const int surf = tmp.add_surface(mat, nullptr);
    if (surf < 0) return surf;
    HIP_TRY(hipSetDevice(device));
    DevBuf ds, dn, dv, dk, dsc, dc, dd, ddr;
    ST_TRY(ds.upload(&tmp.surfaces[0], sizeof(SurfaceDev)));
    ST_TRY(dn.upload(normal, n * 24));
    ST_TRY(dv.upload(view, n * 24));
    ST_TRY(dk.upload(key, n * 8));
    ST_TRY(dsc.alloc(n * 4));
    ST_TRY(dc.alloc(n * 24));
    ST_TRY(dd.alloc(n * 24));
    ST_TRY(ddr.alloc(n * 4));
    if (n)
        HIP_TRY(launch_test_material((const SurfaceDev*)ds.p, (const double*)dn.p, (const double*)dv.p,
                                     (const uint64_t*)dk.p, n, (int32_t*)dsc.p, (double*)dc.p, (double*)dd.p,
                                     (uint32_t*)ddr.p, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    ST_TRY(dsc.download(scattered, n * 4));
    ST_TRY(dc.download(color, n * 24));
    ST_TRY(dd.download(dir, n * 24));
    return ddr.download(draws, n * 4);
    })
}

int rayrs_test_background(rayrs_scene* scene, const double* dir, uint64_t n, double* rgb) {
    if (!scene || !dir || !rgb) return RAYRS_INVALID_ARG;
    if (scene->device < 0) return RAYRS_NO_DEVICE;
    HIP_TRY(hipSetDevice(scene->device));
    DevBuf dd, dout;
    ST_TRY(dd.upload(dir, n * 24));
    ST_TRY(dout.alloc(n * 24));
    const SceneDev sc = make_scene_dev(scene, false);
    if (n) HIP_TRY(launch_test_background(sc, (const double*)dd.p, n, (double*)dout.p, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    return dout.download(rgb, n * 24);
}

}  // extern "C"

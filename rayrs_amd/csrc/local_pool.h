// local_pool.h -- launch wrappers of local_pool.hip: the radiance integrator for scenes whose whole
// walk tree is one record (or none), with every path resident on the CU from its first ray to its last.
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"

namespace rayrs {

constexpr uint32_t LP_MAX_GATES = 4;   // leaf slots of the one wide record
constexpr uint32_t LP_MAX_PRIMS = 16;  // 4 gates x 4 primitives
#ifndef LP_P
#define LP_P 112   // paths per wave (64 < LP_P <= 128); with LP_WPS, an experiment's knob (-DLP_P=.. -DLP_WPS=..)
#endif
#ifndef LP_WPS
#define LP_WPS 3   // workgroups per CU the kernel is built for (= waves per SIMD)
#endif
constexpr uint32_t LP_PATHS_PER_WAVE = LP_P;

// The scene's gating boxes as kernel arguments: what is left of the BVH when the walk tree has at most one
// record.  Gate g is one leaf group of the reference's tree (primitives first .. first + count - 1 in
// depth-first order) behind exactly the box whose slab test gates the reference's access to it
// (scene_host.cpp build_walk_trees: the gate tree); with no record at all the only gate is the root group behind the root box.
struct LocalScene {
    double box[LP_MAX_GATES][6];  // xmin xmax ymin ymax zmin zmax
    uint32_t first[LP_MAX_GATES], count[LP_MAX_GATES];
    uint32_t n_gates;
    uint32_t n_records;   // 0 or 1: interior visits a query that enters the root box is charged with (work counters)
    uint32_t n_prims;
    uint32_t kind_mask;   // bit k: some surface has material kind k
};

struct LocalDev {
    double* light;                  // 4 doubles per resident path (3 used): the light of the paths that have any
    unsigned long long* next_item;  // item counter of this launch's segment, counting from 0
    uint64_t item_base;             // first item of the segment
    uint64_t item_count;            // items in the segment
    uint32_t reserve;               // items a wave takes from the counter at a time
    uint32_t pad;
};

uint32_t lp_lds_bytes(uint32_t n_prims, uint32_t n_surfaces);
hipError_t lp_configure();  // raises the kernels' dynamic LDS limit; once per device
hipError_t lp_occupancy(bool compact, uint32_t n_prims, uint32_t n_surfaces, int* blocks_per_cu);
hipError_t lp_launch(bool compact, bool count, const SceneDev& sc, const LocalScene& ls, const CameraDev& cam,
                     const RenderDev& rp, const LocalDev& lp, uint32_t blocks, hipStream_t stream);

}  // namespace rayrs

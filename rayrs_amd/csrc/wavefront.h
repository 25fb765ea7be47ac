// wavefront.h -- path pool and launch wrappers of wavefront.hip
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"

namespace rayrs {

// What the traversal kernel reads and writes: one 64-byte record per slot.
struct RaySlot {
    double o[3];
    double d[3];
    double t;        // closest hit (valid when prim != 0xffffffff)
    uint32_t prim;   // DFS slot of the closest primitive, 0xffffffff = miss
    uint32_t bounce; // number of the BVH query in flight, 1-based (loop counter of lib.rs:525)
};
static_assert(sizeof(RaySlot) == 64, "RaySlot");

// The rest of a path and of the item (pixel, sample chunk) it belongs to: 128 bytes.
struct PathSlot {
    double thr[3];    // throughput, lib.rs:522
    double light[3];  // lib.rs:523
    double acc[3];    // sum of the item's finished samples, main.rs:67-69
    uint64_t key;     // rr_path_key of the sample in flight
    uint32_t draw;    // next draw index
    uint32_t item;
    uint32_t s_cur;   // next sample of the item to start
    uint32_t s_end;
    uint32_t has_item;
    uint32_t pad[7];
};
static_assert(sizeof(PathSlot) == 128, "PathSlot");

// slot states
constexpr uint8_t WF_IDLE = 0;   // no path in flight: gen_kernel's input
constexpr uint8_t WF_READY = 1;  // ray written, waiting for the traversal kernel
constexpr uint8_t WF_HIT = 2;    // closest hit found: hit_kernel's input
constexpr uint8_t WF_MISS = 3;   // no hit: miss_kernel's input
constexpr uint8_t WF_DEAD = 4;   // out of work (or padding of the pool)

struct WfCtl {
    uint32_t next_window;  // window cursor of the traversal kernel
    uint32_t live_slots;   // slots that still have or can get work
    unsigned long long next_item;
    uint32_t pad[4];
};

struct WfDev {
    RaySlot* rays;
    PathSlot* paths;
    uint8_t* state;
    WfCtl* ctl;
    uint32_t np;  // slots in the pool, a multiple of 1024
};

hipError_t wf_launch_init(const WfDev& wf, uint32_t live, hipStream_t stream);
hipError_t wf_launch_gen(const CameraDev& cam, const RenderDev& rp, const WfDev& wf, uint32_t blocks,
                         hipStream_t stream);
hipError_t wf_launch_trav(bool compact, bool count, const SceneDev& sc, const RenderDev& rp, const WfDev& wf,
                          uint32_t blocks, hipStream_t stream);
hipError_t wf_trav_occupancy(bool compact, uint32_t stack_depth, int* blocks_per_cu);
hipError_t wf_launch_hit(bool compact, const SceneDev& sc, const CameraDev& cam, const RenderDev& rp, const WfDev& wf,
                         uint32_t blocks, hipStream_t stream);
hipError_t wf_launch_miss(const SceneDev& sc, const CameraDev& cam, const RenderDev& rp, const WfDev& wf,
                          uint32_t blocks, hipStream_t stream);

}  // namespace rayrs

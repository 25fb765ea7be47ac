// wavefront.h -- path pool and launch wrappers of wavefront.hip
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "layout.h"

namespace rayrs {

// Per-slot state, split by who touches it so that a kernel dirties only the lines it needs.

// What the traversal kernel reads and writes: one 64-byte record per slot.
struct RaySlot {
    double o[3];
    double d[3];
    double t;        // closest hit } written by the traversal kernel for a slot it leaves in state HIT;
    uint32_t prim;   // DFS slot of the closest primitive } stale in every other state
    uint32_t bd;     // bounce | draw << 16: number of the BVH query in flight, 1-based (loop counter of
                     // lib.rs:525), and the path's next RNG draw index
};
static_assert(sizeof(RaySlot) == 64, "RaySlot");

// The rest of a path's state, in the other half of the slot's line: throughput, the sum of the item's finished
// samples (main.rs:67-69) and the four words that say which item, sample and pixel the slot is working on.
// While bounce == 1 the throughput is (1,1,1) (lib.rs:522) and is not read.
struct TailSlot {
    double thr[3];
    double acc[3];
    uint32_t item;
    uint32_t s_cur;  // next sample of the item to start (30 bits) | SLOT_LIGHT_BIT | SLOT_ITEM_BIT
    uint32_t s_end;
    uint32_t pix;    // row << 16 | col of the item's pixel (image coordinates)
};
static_assert(sizeof(TailSlot) == 64, "TailSlot");

// One slot = ONE 128-byte line.  Memory is fetched in whole lines on this chip (profiles/r02_fetch_calibration.json),
// so what a kernel pays for a slot is the number of lines it touches, not the bytes it uses.
// A path's `light` (lib.rs:523, :534) is not in the slot.  It is exactly +0 until the path meets a surface that
// emits -- or until a component of the throughput stops being finite (+0 + NaN) -- and very few paths get
// there; those keep their light in a side array (WfDev::light, 32 bytes per slot) and say so with
// SLOT_LIGHT_BIT.  The sample's RNG key is recomputed from the pixel and the sample index.
// (Until late in round 2 the slot was 192 bytes with light and key in it: two lines per access, 10 % of the frame.)
struct PathSlot {
    RaySlot ray;
    TailSlot tail;
};
static_assert(sizeof(PathSlot) == 128, "PathSlot");
constexpr uint32_t SLOT_ITEM_BIT = 1u << 31;    // the slot has an item
constexpr uint32_t SLOT_LIGHT_BIT = 1u << 30;   // the path's light is in WfDev::light (else it is +0)
constexpr uint32_t SLOT_SAMPLE_MASK = (1u << 30) - 1u;  // samples per pixel a slot can count

// slot states
constexpr uint8_t WF_IDLE = 0;   // no path in flight: gen_kernel's input
constexpr uint8_t WF_READY = 1;  // ray written, waiting for the traversal kernel; | octant of its direction << 4 (wavefront.hip ready_state)
constexpr uint8_t WF_HIT = 2;    // closest hit found: hit_kernel's input
constexpr uint8_t WF_MISS = 3;   // no hit: miss_kernel's input
constexpr uint8_t WF_DEAD = 4;   // out of work (or padding of the pool)

// Two 128-byte lines, by who writes them: the window cursor takes the traversal kernel's atomics, live_slots the hit and
// miss kernels'.  A scalar load from a line that atomics are hammering pays the trip to memory every time: live_slots
// read in the cursor's line once per window fetch cost the traversal kernel 27 % (profiles/r04_xcd_streams.txt).
struct alignas(128) WfCtl {
    uint32_t next_window;  // cursor over the windows the traversal kernel hands out on demand
    uint32_t pad0[31];
    uint32_t live_slots;   // slots that still have or can get work
    uint32_t pad1[31];
};
static_assert(offsetof(WfCtl, live_slots) == 128 && sizeof(WfCtl) == 256, "WfCtl lines");

struct WfDev {
    PathSlot* slots;
    double* light;  // 4 doubles per slot (3 used): the light of the paths that have any (SLOT_LIGHT_BIT)
    uint8_t* state;
    WfCtl* ctl;
    uint32_t np;  // slots in the pool, a multiple of 1024
    // per-wave reserved item ranges [next, end) of the gen/hit/miss kernels, which all run with
    // the same grid (n_flat_waves waves) and give wave w the same windows
    unsigned long long* wave_items;
    uint32_t n_flat_waves;
    // overflow strips of the traversal stacks (LaneStack): one word per thread of the traversal
    // grid per entry beyond SceneDev::stack_lds
    uint32_t trav_threads;
    uint32_t* stack_spill;
};

uint32_t wf_window_slots();  // slots per window (a divisor of 1024)
hipError_t wf_launch_init(const WfDev& wf, uint32_t live, hipStream_t stream);
hipError_t wf_launch_gen(bool compact, const SceneDev& sc, const CameraDev& cam, const RenderDev& rp, const WfDev& wf,
                         uint32_t blocks, hipStream_t stream);
hipError_t wf_launch_trav(bool compact, bool count, const SceneDev& sc, const RenderDev& rp, const WfDev& wf,
                          uint32_t blocks, hipStream_t stream);
hipError_t wf_trav_occupancy(bool compact, uint32_t stack_lds, uint32_t leafq, uint32_t hot_records, int* blocks_per_cu);
// eager_light: request the side array's entry together with the slot (scenes in which a surface emits)
hipError_t wf_launch_hit(bool compact, bool eager_light, const SceneDev& sc, const CameraDev& cam, const RenderDev& rp, const WfDev& wf,
                         uint32_t blocks, hipStream_t stream);
hipError_t wf_launch_miss(bool compact, bool eager_light, const SceneDev& sc, const CameraDev& cam, const RenderDev& rp, const WfDev& wf,
                          uint32_t blocks, hipStream_t stream);

}  // namespace rayrs

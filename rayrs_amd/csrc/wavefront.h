// wavefront.h -- path pool and launch wrappers of wavefront.hip
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"

namespace rayrs {

// Per-slot state, split by who touches it so that a kernel dirties only the lines it needs.

// What the traversal kernel reads and writes: one 64-byte record per slot.
struct RaySlot {
    double o[3];
    double d[3];
    double t;        // closest hit (valid when prim != 0xffffffff)
    uint32_t prim;   // DFS slot of the closest primitive, 0xffffffff = miss
    uint32_t bd;     // bounce | draw << 16: number of the BVH query in flight, 1-based (loop counter of
                     // lib.rs:525), and the path's next RNG draw index
};
static_assert(sizeof(RaySlot) == 64, "RaySlot");

// Path state the hit/miss kernels carry from bounce to bounce.  While bounce == 1 the
// throughput is (1,1,1) and the light (0,0,0) (lib.rs:522-523) and are not stored.
struct HotSlot {
    double thr[3];    // throughput
    double light[3];
    uint64_t key;     // rr_path_key of the sample in flight
    uint64_t pad;
};
static_assert(sizeof(HotSlot) == 64, "HotSlot");

// The item (pixel, sample chunk) the slot is working on; touched only when a path ends.
struct ItemSlot {
    double acc[3];    // sum of the item's finished samples, main.rs:67-69
    uint32_t item;
    uint32_t s_cur;   // next sample of the item to start
    uint32_t s_end;
    uint32_t has_item;
    uint32_t pix;     // row << 16 | col of the item's pixel (image coordinates)
    uint32_t pad;
};
static_assert(sizeof(ItemSlot) == 48, "ItemSlot");

// The wide slot = 192 contiguous bytes.  Random 64-byte accesses to HBM run at ~0.9 TB/s on this
// chip against ~6 TB/s streamed (scripts/ubench/fetch_calib.hip), i.e. the number of separate
// DRAM rows a kernel opens per slot matters more than the bytes it moves: keeping the three
// records of a slot adjacent makes a slot one row activation per kernel.
struct Slot {
    RaySlot ray;
    HotSlot hot;
    ItemSlot item;
    uint64_t pad[2];
};
static_assert(sizeof(Slot) == 192, "Slot");

// The lean slot = ONE 128-byte line, for scenes in which nothing emits (every surface's Emission::emit() is
// exactly zero; an HDRI-lit scene).  There `light` (lib.rs:523, :534) is +0 + throughput * 0 at every hit: a
// component of it is +0, or NaN once the throughput's component stopped being finite -- one bit each.  The
// sample's RNG key is recomputed from the pixel and the sample index (in both layouts).  What is left fits
// the half line behind the ray: throughput, the item's sum and its four identifying words.
struct LeanTail {
    double thr[3];
    double acc[3];
    uint32_t item;
    uint32_t s_cur;  // next sample to start (28 bits) | LEAN_LIGHT_NAN << 28 (x, y, z) | has_item << 31
    uint32_t s_end;
    uint32_t pix;
};
struct LeanSlot {
    RaySlot ray;
    LeanTail tail;
};
static_assert(sizeof(LeanSlot) == 128, "LeanSlot");
constexpr uint32_t LEAN_SAMPLE_MASK = (1u << 28) - 1u;  // samples per pixel the lean layout can count

// slot states
constexpr uint8_t WF_IDLE = 0;   // no path in flight: gen_kernel's input
constexpr uint8_t WF_READY = 1;  // ray written, waiting for the traversal kernel
constexpr uint8_t WF_HIT = 2;    // closest hit found: hit_kernel's input
constexpr uint8_t WF_MISS = 3;   // no hit: miss_kernel's input
constexpr uint8_t WF_DEAD = 4;   // out of work (or padding of the pool)

struct WfCtl {
    uint32_t next_window;  // window cursor of the traversal kernel
    uint32_t live_slots;   // slots that still have or can get work
    uint32_t pad[6];
};

struct WfDev {
    unsigned char* slots;  // np slots of slot_bytes each; every layout starts with the RaySlot
    uint32_t slot_bytes;   // sizeof(Slot) or sizeof(LeanSlot)
    uint32_t pad_;
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
// the gen, hit and miss kernels exist for both slot layouts (wf.slot_bytes says which)
hipError_t wf_launch_gen(const SceneDev& sc, const CameraDev& cam, const RenderDev& rp, const WfDev& wf,
                         uint32_t blocks, hipStream_t stream);
hipError_t wf_launch_trav(bool compact, bool count, const SceneDev& sc, const RenderDev& rp, const WfDev& wf,
                          uint32_t blocks, hipStream_t stream);
hipError_t wf_trav_occupancy(bool compact, uint32_t stack_lds, uint32_t hot_records, int* blocks_per_cu);
hipError_t wf_launch_hit(bool compact, const SceneDev& sc, const CameraDev& cam, const RenderDev& rp, const WfDev& wf,
                         uint32_t blocks, hipStream_t stream);
hipError_t wf_launch_miss(const SceneDev& sc, const CameraDev& cam, const RenderDev& rp, const WfDev& wf,
                          uint32_t blocks, hipStream_t stream);

}  // namespace rayrs

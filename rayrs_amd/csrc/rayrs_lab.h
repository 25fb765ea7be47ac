/*
 * rayrs_lab.h -- PRIVATE scheduling knobs of librayrs_hip.so, for tests/ and scripts/ubench/ only.
 *
 * Not part of the boundary a rayrs-lib maintainer binds (include/rayrs_hip.h): nothing here has a
 * reference counterpart, none of it changes a frame, and any of it may go away.  The public
 * header keeps the two settings a caller may legitimately choose (rayrs_tuning: pool size, route).
 * 0 = the built-in default everywhere.
 */
#ifndef RAYRS_LAB_H
#define RAYRS_LAB_H

#include <stdint.h>

#include "../../include/rayrs_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint32_t refill_min;    /* traversal: refill a wave's idle lanes when fewer than this are traversing (52) */
    uint32_t leaf_min;      /* traversal: run a leaf phase once this many lanes stand on a leaf (the fast walk: 24) / have a leaf
                               group waiting (the default walk: 48) */
    uint32_t static_pct;    /* traversal: share of the pool's windows dealt round robin, 1..100 (50) */
    uint32_t stack_lds;     /* traversal: stack entries per lane kept in LDS, the rest in HBM (12; the default walk, whose
                               stack holds interior records only: 8) */
    uint32_t hot_records;   /* traversal: leading wide records copied to LDS, at most 256 (14 KiB worth; the default walk 10 KiB);
                               0xffffffff = none */
    uint32_t trav_blocks_per_cu; /* traversal workgroups per CU, at most what the occupancy query allows */
    uint32_t eager_light;   /* 1 = the hit and miss kernels request a path's entry of the light side array together
                               with its slot also where no surface emits (they do anyway where one does) */
    uint32_t local_reserve; /* local-pool route: items a wave takes from the counter at a time, 8..4096 */
    uint32_t local_segment_items; /* local-pool route: items per launch segment, >= 65536 (default 2^27); raised
                               as far as needed to keep a frame within 64 segments */
    uint32_t force_rccl;    /* rayrs_render_multi: run the RCCL reduce even when every handle sits on one device
                               (a one-device communicator: the call path of a multi-GPU node on a one-GPU box) */
    uint32_t gate_tree;     /* 1 = the fast walk (with closest-hit culling) over the gate tree instead of the tree
                               of single primitives: what rounds 2 and 3 walked, kept for the same-box A/B of
                               profiles/r04_tight_leaves.txt */
    uint32_t hot_group;     /* 0xffffffff = the default walk reads the whole gate tree also where the scene has a hot group
                               (layout.h HotGroupDev), and the kernels that make rays pre-test nothing: the walk of round 5,
                               kept for the same-box A/B */
    uint32_t leaf_wait;     /* traversal, default walk: run a leaf phase once this many lanes can do nothing but wait for one
                               (leaf groups queued, no record to visit) (16) */
    uint32_t flat_blocks_per_cu; /* gen / hit / miss kernels: workgroups per CU of their common grid, 1..64 (default: 8 .. 24 by the
                               pool's size, about nine windows a wave) */
} rayrs_lab_tuning;

/* Applies to the renders launched on this scene afterwards.  Waits for a render in flight. */
int rayrs_lab_set(rayrs_scene* scene, const rayrs_lab_tuning* lab);

/* HIP-event times (ms) of the last finished render's path rounds, three per round: traversal, hit, miss kernel (the
 * local-pool route: its launch, 0, 0).  Returns the number of rounds (negative: rayrs_status); writes min(rounds,
 * cap_rounds) * 3 floats. */
int rayrs_lab_round_ms(rayrs_scene* scene, float* out, uint32_t cap_rounds);

/* The calls rayrs_render_multi's reduce makes for ranks on these devices -- the plan (which rank leads a device, which
 * are summed on it first), ncclCommInitAll once per device list, the grouped in-place ncclReduce to root 0, the syncs --
 * `rounds` times against a RECORDING table instead of librccl and the HIP runtime: no GPU is touched, so the path a node
 * of N distinct devices takes can be checked on a box that has none (tests/test_multi_plan.py).  fail_reduce_at >= 0: that
 * ncclReduce (counted over all rounds) reports an error.  log: ';'-separated record.  Returns the first non-OK status. */
int rayrs_lab_multi_rehearse(const int* rank_devices, uint32_t n, uint32_t rounds, int fail_reduce_at, char* log, uint32_t cap);

#ifdef __cplusplus
}
#endif
#endif

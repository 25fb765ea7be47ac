// stream_pool.h -- launch wrappers of stream_pool.hip: everything of a path except its deep BVH walks, with the
// path resident in LDS between two walks (scenes with a walk tree of more than one record).
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"
#include "wavefront.h"

namespace rayrs {

#ifndef SP_P
#define SP_P 120  // paths per wave (64 < SP_P <= 128); with SP_WPS an experiment's knob
#endif
#ifndef SP_WPS
#define SP_WPS 3  // workgroups per CU the kernel is built for
#endif
constexpr uint32_t SP_PATHS_PER_WAVE = SP_P;
constexpr uint32_t SP_ROOT_PRIMS = 16;  // 4 leaf slots x 4 primitives of the walk tree's root record

// The walk tree's root record as kernel arguments: slot k is unused (kind REF_NONE), an interior slot (a ray that
// enters it needs the traversal kernel) or a leaf group of the reference's tree behind its gating box, whose
// primitives first .. first + count - 1 the shading kernel tests itself (lds_first: where their records wait in LDS).
struct RootRecord {
    double box[4][6];  // xmin xmax ymin ymax zmin zmax
    uint32_t kind[4], first[4], count[4], lds_first[4];
    uint32_t n_lds_prims;
    uint32_t pad;
};

uint32_t sp_lds_bytes(uint32_t n_surfaces);
hipError_t sp_configure();
// winding_only: the frame's first launch takes every IDLE slot; later launches the slots the traversal kernel answered
hipError_t sp_launch(bool compact, bool count, const SceneDev& sc, const RootRecord& root, const CameraDev& cam,
                     const RenderDev& rp, const WfDev& wf, uint32_t blocks, hipStream_t stream);

}  // namespace rayrs

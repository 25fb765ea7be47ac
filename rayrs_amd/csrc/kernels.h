// kernels.h -- host-callable launch wrappers of kernels.hip
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"

namespace rayrs {

// pixel sums of the rank's tiles lt0 .. lt0 + n_lt - 1 from their items' sums (rp.partial, which starts at item rp.partial_item0)
hipError_t launch_resolve(const CameraDev& cam, const RenderDev& rp, uint32_t lt0, uint32_t n_lt, hipStream_t stream);
hipError_t launch_accumulate(void* dst, const void* src, size_t n, bool f64, hipStream_t stream);

hipError_t launch_test_math(int fn, const double* x, const double* y, uint64_t n, double* out, hipStream_t stream);
hipError_t launch_test_rng(uint64_t seed, const uint64_t* pixel, const uint64_t* sample, const uint32_t* draw,
                           uint64_t n, uint64_t* out, hipStream_t stream);
hipError_t launch_test_intersect(bool compact, const SceneDev& sc, const double* o, const double* d, uint64_t n,
                                 double* t_out, long long* prim_out, uint32_t* spill, hipStream_t stream);
hipError_t launch_test_path_trace(bool compact, const SceneDev& sc, const CameraDev& cam, uint64_t seed, uint32_t max_bounces,
                                  const uint32_t* pix, const uint32_t* sample, uint64_t n, uint32_t cap, uint32_t* n_out,
                                  uint32_t* prim_out, double* t_out, double* thr_out, uint32_t* draw_out, double* rgb_out,
                                  uint32_t* spill, hipStream_t stream);
hipError_t launch_test_material(const SurfaceDev* surf, const double* normal, const double* view, const uint64_t* key,
                                uint64_t n, int32_t* scattered, double* color, double* dir, uint32_t* draws,
                                hipStream_t stream);
hipError_t launch_test_background(const SceneDev& sc, const double* dir, uint64_t n, double* rgb, hipStream_t stream);

}  // namespace rayrs

// scene_internal.hpp -- the handles behind include/rayrs_hip.h, shared by abi.cpp and multi_device.cpp.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rayrs_hip.h"
#include "local_pool.h"
#include "rayrs_lab.h"
#include "scene_host.hpp"
#include "wavefront.h"

struct rayrs_objects {
    rayrs::ObjectList list;
};

struct rayrs_scene {
    rayrs::FlatScene flat;
    std::vector<rayrs::SurfaceDev> surfaces;
    uint64_t n_objects = 0;
    int device = -1;
    void* d_prims = nullptr;
    rayrs::SurfaceDev* d_surfaces = nullptr;
    float* d_hdri = nullptr;
    rayrs::Counters* d_counters = nullptr;
    double* d_partial = nullptr;
    size_t partial_items = 0;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    hipStream_t last_stream = nullptr;
    bool pending = false;
    bool last_count = false;
    int cu_count = 0;
    // How the traversal kernel walks each of the scene's two trees: [0] FlatScene::walk (rayrs_render_params.fast_traversal),
    // [1] FlatScene::gate (the default walk).
    struct Walk {
        void* d_nodes = nullptr;
        int blocks_per_cu = 0;       // traversal kernel, from the occupancy query
        uint32_t stack_lds = 1;      // traversal stack entries kept in LDS
        uint32_t hot_records = 0;    // leading records kept in LDS
        uint32_t leafq = 0;          // leaf groups a lane of the default walk may have waiting in LDS (0: the fast walk's tree)
    };
    // [2] FlatScene::gate_hot (the default walk on a scene with a hot group: layout.h HotGroupDev).
    Walk trav[3];
    const rayrs::WalkTree& tree(int which) const { return which == 2 ? flat.gate_hot : which == 1 ? flat.gate : flat.walk; }
    rayrs::HotGroupDev* d_hot = nullptr;
    // which of the three a frame walks: the fast walk [0] (or [1] with rayrs_lab_tuning.gate_tree), the default walk [2] if the
    // scene has a hot group and the lab has not switched it off, else [1]
    int walk_index(bool exact) const {
        if (!exact) return lab.gate_tree ? 1 : 0;
        return (flat.has_hot && lab.hot_group != 0xffffffffu) ? 2 : 1;
    }
    uint64_t device_bytes = 0;
    // The path pool of the streaming route (abi.cpp rayrs_render_launch): slots, state bytes, control words,
    // per-wave item ranges and traversal-stack overflow strips, kept between renders.
    struct Pool {
        rayrs::WfDev wf = {};
        void* block = nullptr;       // one allocation holding the slot records and the state bytes
        size_t block_bytes = 0;
        unsigned long long* d_wave_items = nullptr;
        uint32_t wave_items_cap = 0;
        uint32_t* d_stack_spill = nullptr;
        size_t stack_spill_words = 0;
        uint32_t* h_live = nullptr;  // pinned: live_slots read-backs
        hipEvent_t ev_batch[2] = {nullptr, nullptr};
        std::vector<hipEvent_t> ev_round;  // four per round: around the traversal, hit and miss launches
        uint32_t timed_rounds = 0;
    };
    Pool pool;
    unsigned long long* d_next_item = nullptr;  // the device-wide item counter
    uint32_t rounds = 0;
    // Scenes whose walk tree is at most one record are rendered by local_pool.hip: every path resident in LDS.
    bool local_ok = false;
    bool last_local = false;          // the render in flight took that route
    bool last_exact = false;          // ... with the exact walk (asked for, or a far camera: abi.cpp camera_is_far)
    rayrs::LocalScene local = {};
    int local_blocks_per_cu = 1;      // local-pool kernel, from the occupancy query with the scene's LDS size
    double* d_local_light = nullptr;  // 4 doubles per resident path
    size_t local_light_paths = 0;
    unsigned long long* d_local_items = nullptr;  // one item counter per launch segment
    // rayrs_render_multi: this rank's stream and zeroed full-size framebuffer, kept between calls
    hipStream_t multi_stream = nullptr;
    void* multi_out = nullptr;
    size_t multi_out_bytes = 0;
    rayrs_tuning tuning = {};  // zeros = defaults (rayrs_scene_set_tuning)
    rayrs_lab_tuning lab = {};  // development knobs (rayrs_lab.h), zeros = defaults
};


namespace rayrs {
// thread-local text behind rayrs_last_error()
void set_last_error(const std::string& text);
int hip_fail(hipError_t e, const char* what);
// uploads s->flat to s->device and sizes the traversal kernel's LDS (abi.cpp)
int scene_upload(rayrs_scene* s);
void scene_free_device(rayrs_scene* s);
}  // namespace rayrs

#define HIP_TRY(expr)                                       \
    do {                                                    \
        hipError_t _e = (expr);                             \
        if (_e != hipSuccess) return rayrs::hip_fail(_e, #expr); \
    } while (0)

// No exception may cross the C boundary (a std::bad_alloc from a vector would otherwise end the host process).
#define RAYRS_GUARDED(...)                                   \
    try {                                                    \
        __VA_ARGS__                                          \
    } catch (const std::bad_alloc&) {                        \
        rayrs::set_last_error("out of host memory");         \
        return RAYRS_OOM;                                    \
    } catch (const std::exception& e) {                      \
        rayrs::set_last_error(e.what());                     \
        return RAYRS_INVALID_ARG;                            \
    } catch (...) {                                          \
        rayrs::set_last_error("unknown exception");          \
        return RAYRS_INVALID_ARG;                            \
    }

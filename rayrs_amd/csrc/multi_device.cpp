// multi_device.cpp -- rayrs_render_multi: the block loop of rayrs/src/main.rs:57-101 over the GPUs
// of one node, inside the library.
//
// The reference spreads 16x16 image blocks over rayon's worker threads (main.rs:61) and joins the
// blocks into one image (main.rs:101).  Here the workers are GPUs: scene i renders the 8x8 tiles
// t with t % n == i (paths never communicate and the RNG is keyed by (pixel, sample), so the
// partition does not change a pixel), each from its own host thread on its own stream into a
// zeroed full-size framebuffer on its device, and one RCCL reduce (sum, root = the first scene's
// device) over xGMI assembles the frame: a pixel is non-zero on exactly one rank, x + 0 is exact,
// so the result does not depend on the reduction order and equals the one-GPU frame bit for bit.
//
// RCCL is loaded with dlopen on first use -- a host that renders on one GPU never needs it -- and
// every failure of it comes back as RAYRS_RCCL_ERROR.  Ranks that share a device (rehearsing on
// one GPU) are first summed on that device by a small kernel, then the distinct devices reduce.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>

#include "kernels.h"
#include "scene_internal.hpp"

using namespace rayrs;

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
    // one communicator set per device list, kept for the life of the process (creating one takes ~1 s)
    std::map<std::vector<int>, std::vector<ncclComm_t>> comms;
};

std::mutex g_rccl_mutex;
Rccl g_rccl;

bool rccl_load(Rccl& r) {
    if (r.handle) return true;
    if (!r.error.empty()) return false;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) {
        r.error = std::string("dlopen(librccl.so.1): ") + dlerror();
        return false;
    }
    auto sym = [&](const char* name) {
        void* p = dlsym(r.handle, name);
        if (!p && r.error.empty()) r.error = std::string("librccl has no ") + name;
        return p;
    };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.Reduce = reinterpret_cast<decltype(r.Reduce)>(sym("ncclReduce"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!r.error.empty()) {
        dlclose(r.handle);
        r.handle = nullptr;
        return false;
    }
    return true;
}

int rccl_fail(Rccl& r, ncclResult_t e, const char* what) {
    set_last_error(std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error"));
    return RAYRS_RCCL_ERROR;
}

#define RCCL_TRY(r, expr)                                    \
    do {                                                     \
        ncclResult_t _e = (expr);                            \
        if (_e != ncclSuccess) return rccl_fail(r, _e, #expr); \
    } while (0)

// What the reduce needs of the HIP runtime, as a table: the real one below, a recording one in rayrs_lab_multi_rehearse
// (a CPU-side rehearsal of the calls an N-GPU node makes, on a box that has no N GPUs).
struct DeviceOps {
    hipError_t (*set_device)(int) = nullptr;
    hipError_t (*stream_sync)(hipStream_t) = nullptr;
};
const DeviceOps HIP_OPS{[](int d) { return hipSetDevice(d); }, [](hipStream_t s) { return hipStreamSynchronize(s); }};

// Which ranks' framebuffers are summed where: the ranks in order; the first rank on a device LEADS it (its buffer and
// stream take part in the collective), every later rank on that device is summed into its leader's buffer on the device.
struct ReducePlan {
    std::vector<int> devs;             // the distinct devices, in order of first appearance: devs[0] receives the frame
    std::vector<uint32_t> leader;      // leader[k] = the first rank on devs[k]
    std::vector<std::pair<uint32_t, uint32_t>> local;  // (k, rank): rank's buffer is added to devs[k]'s leader on the device
};

ReducePlan plan_reduce(const std::vector<int>& rank_device) {
    ReducePlan p;
    for (uint32_t i = 0; i < rank_device.size(); i++) {
        const auto at = std::find(p.devs.begin(), p.devs.end(), rank_device[i]);
        if (at == p.devs.end()) p.devs.push_back(rank_device[i]), p.leader.push_back(i);
        else p.local.emplace_back((uint32_t)(at - p.devs.begin()), i);
    }
    return p;
}

// Sum of the framebuffers of the distinct devices into bufs[0] (on devs[0]); one stream per device.  The communicators
// of a device list are created at its first use and kept (r.comms).  The caller holds g_rccl_mutex when r is g_rccl.
int rccl_reduce_to_first(Rccl& r, const DeviceOps& ops, const std::vector<int>& devs, const std::vector<void*>& bufs,
                         const std::vector<hipStream_t>& streams, size_t count, bool f64) {
    auto it = r.comms.find(devs);
    if (it == r.comms.end()) {
        std::vector<ncclComm_t> c(devs.size());
        RCCL_TRY(r, r.CommInitAll(c.data(), (int)devs.size(), devs.data()));
        it = r.comms.emplace(devs, c).first;
    }
    const std::vector<ncclComm_t>& comms = it->second;
    RCCL_TRY(r, r.GroupStart());
    for (size_t i = 0; i < devs.size(); i++) {
        HIP_TRY(ops.set_device(devs[i]));
        const ncclResult_t e = r.Reduce(bufs[i], bufs[i], count, f64 ? ncclFloat64 : ncclFloat32, ncclSum, 0, comms[i],
                                        streams[i]);
        if (e != ncclSuccess) {
            (void)r.GroupEnd();
            return rccl_fail(r, e, "ncclReduce");
        }
    }
    RCCL_TRY(r, r.GroupEnd());
    for (size_t i = 0; i < devs.size(); i++) {
        HIP_TRY(ops.set_device(devs[i]));
        HIP_TRY(ops.stream_sync(streams[i]));
    }
    return RAYRS_OK;
}

int rccl_reduce_to_first(const std::vector<int>& devs, const std::vector<void*>& bufs, const std::vector<hipStream_t>& streams,
                         size_t count, bool f64) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!rccl_load(g_rccl)) {
        set_last_error(g_rccl.error);
        return RAYRS_RCCL_ERROR;
    }
    return rccl_reduce_to_first(g_rccl, HIP_OPS, devs, bufs, streams, count, f64);
}

struct Rank {
    rayrs_scene* scene = nullptr;
    void* d_out = nullptr;
    hipStream_t stream = nullptr;
    rayrs_render_stats stats = {};
    int status = RAYRS_OK;
    std::string error;
};

void render_rank(Rank& r, const rayrs_camera* camera, rayrs_render_params params, size_t bytes) {
    auto fail = [&](int st) {
        r.status = st;
        r.error = rayrs_last_error();  // this thread's text
    };
    // the rank's stream and framebuffer live in the scene handle: a frame of a few milliseconds (the sphere scenes)
    // would otherwise spend as long creating and freeing them as rendering
    rayrs_scene* s = r.scene;
    hipError_t e = hipSetDevice(s->device);
    if (e == hipSuccess && !s->multi_stream) e = hipStreamCreateWithFlags(&s->multi_stream, hipStreamNonBlocking);
    if (e == hipSuccess && s->multi_out_bytes < bytes) {
        if (s->multi_out) (void)hipFree(s->multi_out);
        s->multi_out = nullptr, s->multi_out_bytes = 0;
        e = hipMalloc(&s->multi_out, bytes);
        if (e == hipSuccess) s->multi_out_bytes = bytes;
    }
    r.stream = s->multi_stream, r.d_out = s->multi_out;
    if (e == hipSuccess) e = hipMemsetAsync(r.d_out, 0, bytes, r.stream);  // the tiles of the other ranks stay exact zeros
    if (e != hipSuccess) return fail(hip_fail(e, "rayrs_render_multi: device setup"));
    int st = rayrs_render_launch(r.scene, camera, &params, r.d_out, r.stream);
    if (st == RAYRS_OK) st = rayrs_render_finish(r.scene, &r.stats);
    if (st != RAYRS_OK) fail(st);
}

}  // namespace

extern "C" int rayrs_render_multi(rayrs_scene* const* scenes, uint32_t n, const rayrs_camera* camera,
                                  const rayrs_render_params* params, void* out_host, rayrs_render_stats* stats) {
    RAYRS_GUARDED({
        if (!scenes || n == 0 || !camera || !params || !out_host) return RAYRS_INVALID_ARG;
        for (uint32_t i = 0; i < n; i++) {
            if (!scenes[i]) return RAYRS_INVALID_ARG;
            if (scenes[i]->device < 0) return RAYRS_NO_DEVICE;
            for (uint32_t j = 0; j < i; j++)
                if (scenes[j] == scenes[i]) return RAYRS_INVALID_ARG;  // one render in flight per scene handle
        }
        const bool f64 = params->out_format == RAYRS_OUT_F64;
        const size_t count = (size_t)camera->x_pixels * camera->y_pixels * 3;
        const size_t bytes = count * (f64 ? 8 : 4);

        // ---- every rank renders its tiles on its own host thread and stream
        std::vector<Rank> ranks(n);
        std::vector<std::thread> threads;
        threads.reserve(n);
        struct JoinAll {  // a std::system_error from a later emplace_back must not destroy joinable threads (std::terminate)
            std::vector<std::thread>& t;
            ~JoinAll() {
                for (auto& x : t)
                    if (x.joinable()) x.join();
            }
        } join_all{threads};
        for (uint32_t i = 0; i < n; i++) {
            ranks[i].scene = scenes[i];
            rayrs_render_params p = *params;
            p.tile_rank = i;
            p.tile_ranks = n;
            threads.emplace_back(render_rank, std::ref(ranks[i]), camera, p, bytes);
        }
        for (auto& t : threads) t.join();
        int st = RAYRS_OK;
        for (auto& r : ranks)
            if (r.status != RAYRS_OK && st == RAYRS_OK) {
                st = r.status;
                set_last_error(r.error);
            }

        // ---- ranks that share a device are summed there; then one RCCL reduce over the distinct devices
        std::vector<int> devs;
        std::vector<void*> bufs;
        std::vector<hipStream_t> streams;
        if (st == RAYRS_OK) {
            std::vector<int> rank_device(n);
            for (uint32_t i = 0; i < n; i++) rank_device[i] = ranks[i].scene->device;
            const ReducePlan plan = plan_reduce(rank_device);
            devs = plan.devs;
            for (const uint32_t l : plan.leader) bufs.push_back(ranks[l].d_out), streams.push_back(ranks[l].stream);
            for (const auto& [k, i] : plan.local) {
                if (st != RAYRS_OK) break;
                hipError_t e = hipSetDevice(devs[k]);
                if (e == hipSuccess) e = launch_accumulate(bufs[k], ranks[i].d_out, count, f64, streams[k]);
                if (e == hipSuccess) e = hipStreamSynchronize(streams[k]);
                if (e != hipSuccess) st = hip_fail(e, "rayrs_render_multi: same-device sum");
            }
        }
        // one device (a rehearsal with several handles on it): everything is summed already, no collective, no RCCL --
        // unless rayrs_lab.h force_rccl asks for the call path of a multi-GPU node anyway: a communicator of one
        // device, and the same grouped in-place ncclReduce to rank 0 (which then copies nothing and changes nothing)
        const bool force_rccl = scenes[0]->lab.force_rccl != 0u;
        if (st == RAYRS_OK && (devs.size() > 1 || force_rccl)) st = rccl_reduce_to_first(devs, bufs, streams, count, f64);
        if (st == RAYRS_OK) {
            hipError_t e = hipSetDevice(devs[0]);
            if (e == hipSuccess) e = hipMemcpy(out_host, bufs[0], bytes, hipMemcpyDeviceToHost);
            if (e != hipSuccess) st = hip_fail(e, "rayrs_render_multi: hipMemcpy(out D2H)");
        }
        if (st == RAYRS_OK && stats) {
            *stats = ranks[0].stats;  // every counter summed over the ranks; times and launches: the slowest rank's
            for (uint32_t i = 1; i < n; i++) {
                const rayrs_render_stats& s = ranks[i].stats;
                stats->rays += s.rays, stats->paths += s.paths, stats->nan_pixels += s.nan_pixels;
                stats->neg_pixels += s.neg_pixels, stats->escaped_paths += s.escaped_paths;
                stats->interior_visits += s.interior_visits, stats->tri_tests += s.tri_tests;
                stats->sphere_tests += s.sphere_tests, stats->plane_tests += s.plane_tests;
                stats->direct_rays += s.direct_rays;
                stats->step_wave += s.step_wave, stats->step_lane += s.step_lane;
                stats->inner_wave += s.inner_wave, stats->leaf_wave += s.leaf_wave;
                stats->interior_ticks += s.interior_ticks, stats->leaf_ticks += s.leaf_ticks;
                stats->refill_ticks += s.refill_ticks;
                stats->pre_rays += s.pre_rays, stats->pre_root_records += s.pre_root_records, stats->hot_lane += s.hot_lane;
                stats->hot_prim_tests += s.hot_prim_tests, stats->hot_tri_divided += s.hot_tri_divided;
                for (int k = 0; k < 8; k++) stats->surface_hits[k] += s.surface_hits[k];
                if (s.total_ms > stats->total_ms) stats->total_ms = s.total_ms;
                if (s.trace_ms > stats->trace_ms) stats->trace_ms = s.trace_ms;
                if (s.kernel_ms > stats->kernel_ms) stats->kernel_ms = s.kernel_ms;
                if (s.hit_ms > stats->hit_ms) stats->hit_ms = s.hit_ms;
                if (s.miss_ms > stats->miss_ms) stats->miss_ms = s.miss_ms;
                if (s.kernel_launches > stats->kernel_launches) stats->kernel_launches = s.kernel_launches;
            }
        }
        return st;
    })
}

// ---- rayrs_lab.h: the calls rayrs_render_multi's reduce makes on an N-GPU node, rehearsed without one ----
namespace {
struct Rehearsal {
    std::string log;
    int fail_reduce_at = -1;  // the n-th ncclReduce (0-based, over all rounds) reports an error
    int reduces = 0;
    int next_comm = 1;
};
thread_local Rehearsal* g_rehearsal = nullptr;
void say(const std::string& t) { g_rehearsal->log += t + ";"; }
}  // namespace

extern "C" int rayrs_lab_multi_rehearse(const int* rank_devices, uint32_t n, uint32_t rounds, int fail_reduce_at, char* log,
                                        uint32_t cap) {
    RAYRS_GUARDED({
        if (!rank_devices || n == 0 || !log || cap == 0) return RAYRS_INVALID_ARG;
        Rehearsal reh;
        reh.fail_reduce_at = fail_reduce_at;
        g_rehearsal = &reh;
        Rccl stub;  // records instead of calling librccl; communicators are small integers
        stub.CommInitAll = [](ncclComm_t* c, int nd, const int* d) {
            std::string t = "init[";
            for (int i = 0; i < nd; i++) {
                t += (i ? "," : "") + std::to_string(d[i]);
                c[i] = reinterpret_cast<ncclComm_t>((intptr_t)g_rehearsal->next_comm++);
            }
            say(t + "]");
            return ncclSuccess;
        };
        stub.GroupStart = []() { say("group_start"); return ncclSuccess; };
        stub.GroupEnd = []() { say("group_end"); return ncclSuccess; };
        stub.Reduce = [](const void* src, void* dst, size_t count, ncclDataType_t, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t) {
            say("reduce(comm=" + std::to_string((intptr_t)comm) + ",root=" + std::to_string(root) + ",count=" + std::to_string(count) +
                ",in_place=" + std::to_string(src == dst) + ",sum=" + std::to_string(op == ncclSum) + ")");
            return g_rehearsal->reduces++ == g_rehearsal->fail_reduce_at ? ncclInternalError : ncclSuccess;
        };
        stub.GetErrorString = [](ncclResult_t) { return "rehearsed failure"; };
        const DeviceOps ops{[](int d) { say("set_device(" + std::to_string(d) + ")"); return hipSuccess; },
                            [](hipStream_t) { say("sync"); return hipSuccess; }};
        std::vector<int> rank_device(rank_devices, rank_devices + n);
        int st = RAYRS_OK;
        for (uint32_t r = 0; r < rounds; r++) {
            const ReducePlan plan = plan_reduce(rank_device);
            std::string t = "plan devs[";
            for (size_t k = 0; k < plan.devs.size(); k++) t += (k ? "," : "") + std::to_string(plan.devs[k]) + ":" + std::to_string(plan.leader[k]);
            t += "] local[";
            for (size_t k = 0; k < plan.local.size(); k++) t += (k ? "," : "") + std::to_string(plan.local[k].first) + "<-" + std::to_string(plan.local[k].second);
            say(t + "]");
            // one fake buffer and stream per distinct device (the addresses are only compared)
            std::vector<void*> bufs;
            std::vector<hipStream_t> streams;
            for (size_t k = 0; k < plan.devs.size(); k++) bufs.push_back(reinterpret_cast<void*>((intptr_t)(0x1000 * (k + 1)))), streams.push_back(nullptr);
            if (plan.devs.size() > 1) {
                const int s1 = rccl_reduce_to_first(stub, ops, plan.devs, bufs, streams, 12, false);
                say("status=" + std::to_string(s1));
                if (s1 != RAYRS_OK && st == RAYRS_OK) st = s1;
            } else {
                say("one device: no collective");
            }
        }
        say("communicator_sets=" + std::to_string(stub.comms.size()));
        g_rehearsal = nullptr;
        const size_t m = std::min<size_t>(reh.log.size(), cap - 1);
        std::memcpy(log, reh.log.data(), m);
        log[m] = 0;
        return st;
    })
}

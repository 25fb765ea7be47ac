// rayrs_cli.cpp -- the reference's command line (rayrs/src/main.rs) on top of the C ABI:
//
//     rayrs hdri_path [spp] [--scene NAME] [--seed N] [--device N] [--max-bounces N] [--fast-traversal 0|1]
//                           [--gpus N | --devices a,b,...]
//
// --gpus N renders on HIP devices 0..N-1 at once (--devices names them; a device may be named more than
// once to rehearse on fewer GPUs): image tiles interleaved over the devices, one RCCL reduce of the
// framebuffer (rayrs_render_multi) -- the counterpart of the reference's rayon block loop, which also
// lives inside the binary.
//
// Same positional arguments and defaults as main.rs:125-138 (spp defaults to 2000 and a spp
// that does not parse silently becomes 2000), same default scene (material_test, main.rs:201),
// same outputs: <scene>.png (gamma 1/2.2) and <scene>.hdr, "Time taken" and the
// clamped/NaN/negative pixel counts.  The reference picks the scene by editing main();
// --scene selects among the same functions of test_scenes.rs.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rayrs_hip.h"

namespace {

rayrs_material cook_torrance_metal(double r, double g, double b, double alpha, double r0r, double r0g, double r0b) {
    rayrs_material m;
    std::memset(&m, 0, sizeof m);
    m.kind = RAYRS_MAT_COOK_TORRANCE;
    m.metallic = 1;
    m.color[0] = r, m.color[1] = g, m.color[2] = b;
    m.alpha = alpha;
    m.r0[0] = r0r, m.r0[1] = r0g, m.r0[2] = r0b;
    return m;
}

rayrs_material simple(int kind, double c, double alpha = 0., double ior = 0.) {
    rayrs_material m;
    std::memset(&m, 0, sizeof m);
    m.kind = kind;
    m.color[0] = m.color[1] = m.color[2] = c;
    m.spec_color[0] = m.spec_color[1] = m.spec_color[2] = 1.0;
    m.alpha = alpha;
    m.ior = ior;
    return m;
}

struct SceneDef {
    std::vector<rayrs_material> spheres;  // one unit sphere per material
    double cam_origin[3], cam_lookat[3], fov, width, height;
    uint32_t ppi;
    bool row;  // multiple_spheres layout (test_scenes.rs:169-211) or single sphere (:14-44)
};

bool make_scene(const std::string& name, SceneDef& s) {
    s.spheres.clear();
    auto row_cam = [&](double oy, double fov, double h) {
        s.cam_origin[0] = 0, s.cam_origin[1] = oy, s.cam_origin[2] = 20;
        s.cam_lookat[0] = 0, s.cam_lookat[1] = 1, s.cam_lookat[2] = 0;
        s.fov = fov, s.width = 1920. / 500., s.height = h, s.ppi = 125, s.row = true;
    };
    auto single_cam = [&]() {
        s.cam_origin[0] = 0, s.cam_origin[1] = 5, s.cam_origin[2] = 10;
        s.cam_lookat[0] = 0, s.cam_lookat[1] = 1, s.cam_lookat[2] = 0;
        s.fov = 50., s.width = 1920. / 500., s.height = 1080. / 500., s.ppi = 100, s.row = false;
    };
    if (name == "material_test") {  // test_scenes.rs:276-331
        s.spheres = {simple(RAYRS_MAT_LAMBERTIAN, 0.8), simple(RAYRS_MAT_PLASTIC, 0.8, 0.05, 1.45),
                     simple(RAYRS_MAT_REFLECT, 0.8), cook_torrance_metal(1, 1, 1, 0.05, 0.8, 0.8, 0.8),
                     simple(RAYRS_MAT_GLASS, 1.0, 0., 1.45), simple(RAYRS_MAT_COOK_TORRANCE_GLASS, 1.0, 0.05, 1.45),
                     simple(RAYRS_MAT_NO_REFLECT, 0.)};
        row_cam(3., 90., 250. / 500.);
    } else if (name == "spheres_metallic") {  // :213-224
        for (int i = 0; i < 7; i++) s.spheres.push_back(cook_torrance_metal(1, 1, 1, 0.01 * (double)(4 * i + 1), 0.8, 0.8, 0.8));
        row_cam(10., 72., 400. / 500.);
    } else if (name == "spheres_plastic") {  // :226-239
        for (int i = 0; i < 7; i++) s.spheres.push_back(simple(RAYRS_MAT_PLASTIC, 0.8, 0.01 * (double)(4 * i + 1), 1.45));
        row_cam(10., 72., 400. / 500.);
    } else if (name == "cook_torrance_spheres_frosted_glass") {  // :241-256
        for (int i = 0; i < 7; i++)
            s.spheres.push_back(simple(RAYRS_MAT_COOK_TORRANCE_GLASS, 1.0, 0.01 * (double)(4 * i + 1), 1.45));
        row_cam(10., 72., 400. / 500.);
    } else if (name == "diffuse_single_sphere") {  // :60-63
        s.spheres = {simple(RAYRS_MAT_LAMBERTIAN, 0.8)};
        single_cam();
    } else if (name == "copper_sphere") {  // :46-53
        s.spheres = {cook_torrance_metal(1, 1, 1, 0.05, 0.722, 0.451, 0.2)};
        single_cam();
    } else if (name == "glass_sphere") {  // :55-58
        s.spheres = {simple(RAYRS_MAT_GLASS, 0.8, 0., 1.45)};
        single_cam();
    } else if (name == "cook_torrance_glass_sphere") {  // :65-68
        s.spheres = {simple(RAYRS_MAT_COOK_TORRANCE_GLASS, 0.8, 0.05, 1.45)};
        single_cam();
    } else {
        return false;
    }
    return true;
}

int fail(const char* what, int status) {
    std::fprintf(stderr, "%s: %s (%s %s)\n", what, rayrs_strerror(status), rayrs_last_error(), rayrs_io_last_error());
    return 1;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) {  // main.rs:126-128
        std::fprintf(stderr, "Usage: rayrs hdri_path [spp]\n");
        return 1;
    }
    const char* hdri_path = argv[1];
    uint32_t spp = 2000;  // SPP, main.rs:20
    std::string scene_name = "material_test";
    uint64_t seed = 0x5EED;
    int device = 0;
    uint32_t max_bounces = 50;  // main.rs:77
    uint32_t fast_traversal = 0;  // rayrs_render_params.fast_traversal: 0 = the reference's visit set by construction
    std::vector<int> devices;
    int i = 2;
    if (i < argc && argv[i][0] != '-') {
        char* end = nullptr;
        const unsigned long v = std::strtoul(argv[i], &end, 10);
        if (end != argv[i] && *end == '\0' && v > 0) spp = (uint32_t)v;  // unwrap_or_else(|_| SPP), main.rs:133
        i++;
    }
    for (; i + 1 < argc; i += 2) {
        const std::string opt = argv[i];
        if (opt == "--scene") scene_name = argv[i + 1];
        else if (opt == "--seed") seed = std::strtoull(argv[i + 1], nullptr, 0);
        else if (opt == "--device") device = std::atoi(argv[i + 1]);
        else if (opt == "--max-bounces") max_bounces = (uint32_t)std::atoi(argv[i + 1]);
        else if (opt == "--fast-traversal") fast_traversal = std::atoi(argv[i + 1]) ? 1u : 0u;
        else if (opt == "--exact-traversal") fast_traversal = std::atoi(argv[i + 1]) ? 0u : 1u;  // (the pre-round-5 spelling)
        else if (opt == "--gpus") {
            devices.clear();
            for (int d = 0; d < std::atoi(argv[i + 1]); d++) devices.push_back(d);
        } else if (opt == "--devices") {
            devices.clear();
            for (const char* p = argv[i + 1]; *p;) {
                char* end = nullptr;
                devices.push_back((int)std::strtol(p, &end, 10));
                if (end == p) {
                    std::fprintf(stderr, "bad device list %s\n", argv[i + 1]);
                    return 1;
                }
                p = *end == ',' ? end + 1 : end;
            }
        }
        else {
            std::fprintf(stderr, "unknown option %s\n", opt.c_str());
            return 1;
        }
    }
    SceneDef def;
    if (!make_scene(scene_name, def)) {
        std::fprintf(stderr, "unknown scene %s\n", scene_name.c_str());
        return 1;
    }

    float* hdri = nullptr;
    uint32_t hw = 0, hh = 0;
    int st = rayrs_hdr_load(hdri_path, &hdri, &hw, &hh);  // main.rs:36-41; the clip(0,3) of :43 happens in rayrs_scene_new
    if (st != RAYRS_OK) return fail("hdri", st);

    rayrs_objects* objs = nullptr;
    if ((st = rayrs_objects_create(&objs)) != RAYRS_OK) return fail("objects", st);
    rayrs_emission dark;
    std::memset(&dark, 0, sizeof dark);
    const rayrs_material floor = cook_torrance_metal(1, 1, 1, 0.5, 0.8, 0.8, 0.8);  // test_scenes.rs:15-19
    if ((st = rayrs_object_plane(objs, RAYRS_AXIS_Y, -25., 25., -25., 25., 0., &floor, &dark)) != RAYRS_OK)
        return fail("floor", st);
    const long n = (long)def.spheres.size();
    for (long k = 0; k < n; k++) {
        const double origin[3] = {def.row ? 2.2 * (double)(k - n / 2) : 0.0, 1.0, 0.0};  // test_scenes.rs:184
        if ((st = rayrs_object_sphere(objs, 1., origin, &def.spheres[(size_t)k], &dark)) != RAYRS_OK)
            return fail("sphere", st);
    }
    if (devices.empty()) devices.push_back(device);
    rayrs_scene* scene = nullptr;
    st = rayrs_scene_new(objs, 0.000001, 1000000., RAYRS_BVH_SAH, 1000, hw, hh, hdri, devices[0], &scene);  // main.rs:52
    rayrs_objects_destroy(objs);
    rayrs_buffer_free(hdri);
    if (st != RAYRS_OK) return fail("scene", st);
    std::vector<rayrs_scene*> scenes{scene};  // one handle per device; the BVH is built once
    for (size_t d = 1; d < devices.size(); d++) {
        rayrs_scene* clone = nullptr;
        if ((st = rayrs_scene_clone_to_device(scene, devices[d], &clone)) != RAYRS_OK) return fail("scene clone", st);
        scenes.push_back(clone);
    }

    rayrs_camera cam;
    const double up[3] = {0., 1., 0.};
    if ((st = rayrs_camera_new(def.cam_origin, up, def.cam_lookat, def.fov, def.width, def.height, def.ppi, &cam)) != RAYRS_OK)
        return fail("camera", st);

    rayrs_render_params params;
    std::memset(&params, 0, sizeof params);
    params.spp = spp;
    params.max_bounces = max_bounces;
    params.seed = seed;
    params.sample_chunk = 0;  // the reference's single sequential sum per pixel
    params.tile_rank = 0;
    params.tile_ranks = 1;
    params.out_format = RAYRS_OUT_F32;
    params.fast_traversal = fast_traversal;
    std::vector<float> rgb((size_t)cam.x_pixels * cam.y_pixels * 3, 0.f);
    rayrs_render_stats stats;
    const auto t0 = std::chrono::steady_clock::now();
    if (scenes.size() == 1)
        st = rayrs_render(scene, &cam, &params, rgb.data(), &stats);
    else
        st = rayrs_render_multi(scenes.data(), (uint32_t)scenes.size(), &cam, &params, rgb.data(), &stats);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (st != RAYRS_OK) return fail("render", st);
    if (stats.nan_pixels) std::fprintf(stderr, "NaN pixel detected\n");       // main.rs:81-83
    if (stats.neg_pixels) std::fprintf(stderr, "Negative pixel detected\n");  // main.rs:85-87
    std::printf("Time taken %s: %.3f s\n", scene_name.c_str(), secs);         // main.rs:96-100
    std::printf("Rays: %llu (%.1f Mray/s)\n", (unsigned long long)stats.rays, (double)stats.rays / secs / 1e6);

    std::vector<uint8_t> bytes(rgb.size());
    uint64_t counts[3];
    rayrs_image_to_bytes(rgb.data(), cam.x_pixels, cam.y_pixels, 1. / 2.2, bytes.data(), counts);  // main.rs:106
    std::printf("Clamped pixels: %llu\nNaN pixels: %llu\nNegative pixels: %llu\n", (unsigned long long)counts[0],
                (unsigned long long)counts[1], (unsigned long long)counts[2]);  // image.rs:218-220
    if ((st = rayrs_png_save((scene_name + ".png").c_str(), bytes.data(), cam.x_pixels, cam.y_pixels)) != RAYRS_OK)
        return fail("png", st);
    if ((st = rayrs_hdr_save((scene_name + ".hdr").c_str(), rgb.data(), cam.x_pixels, cam.y_pixels)) != RAYRS_OK)
        return fail("hdr", st);
    for (rayrs_scene* sc : scenes) rayrs_scene_destroy(sc);
    return 0;
}

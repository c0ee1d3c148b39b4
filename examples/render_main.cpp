// render_main.cpp — the reference's offline path (main.cu:143-266, window code dropped) on top of the C ABI only:
//   parse_scene -> Scene (flatten + BVH) -> upload -> render -> save.   No Python, no torch.
//
//   build: g++ -std=c++17 -O2 -Iinclude examples/render_main.cpp -Lpathtracer_cuda_interactive_amd -lpt_host -lpt_hip \
//              -Wl,-rpath,'$ORIGIN/../pathtracer_cuda_interactive_amd' -o examples/render_main
//   run  : examples/render_main scene.xml|scene.pts out.pfm [width height spp [reference|lbvh|sah [nee]]]
//          (the last two are the extensions of include/pt_api.h: BVH built on the GPU, next-event estimation)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pt_api.h"
#include "pt_host.h"

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s scene.xml|scene.pts out.pfm|out.ppm [width height spp [reference|lbvh|sah [nee]]]\n", argv[0]);
        return 2;
    }
    const std::string path = argv[1], out = argv[2];
    const double t0 = now();
    pt_host_scene* hs = nullptr;
    const bool is_pts = path.size() > 4 && path.compare(path.size() - 4, 4, ".pts") == 0;
    int rc = is_pts ? pt_host_scene_load_pts(path.c_str(), &hs) : pt_host_scene_load_xml(path.c_str(), &hs);   // main.cu:172
    if (rc != PT_OK) { std::fprintf(stderr, "scene load failed (%d): %s\n", rc, pt_host_last_error()); return 1; }
    if ((rc = pt_host_scene_finalize(hs, PT_BVH_SORT_REFERENCE)) != PT_OK) {                                       // main.cu:176
        std::fprintf(stderr, "scene build failed (%d): %s\n", rc, pt_host_last_error());
        return 1;
    }
    pt_scene_desc desc;
    pt_camera cam;
    pt_host_scene_get_desc(hs, &desc);
    pt_host_scene_get_camera(hs, &cam);
    const int W = argc > 3 ? std::atoi(argv[3]) : cam.width;
    const int H = argc > 4 ? std::atoi(argv[4]) : cam.height;
    const int spp = argc > 5 ? std::atoi(argv[5]) : cam.spp;
    const std::string tree = argc > 6 ? argv[6] : "reference";
    const bool nee = argc > 7 && std::strcmp(argv[7], "nee") == 0;
    std::vector<pt_bvh_node> device_nodes;
    int depth = pt_host_scene_bvh_depth(hs);
    if (tree == "lbvh" || tree == "sah") {                // construct_bvh (bvh.cu:16-54) replaced by the device builder
        device_nodes.resize(2 * size_t(desc.num_shapes) - 1);
        int32_t root = 0, d = 0;
        double build_ms = 0;
        rc = pt_bvh_build_device(&desc, tree == "sah" ? PT_BVH_DEVICE_SAH : PT_BVH_DEVICE_LBVH, device_nodes.data(), &root, &d, &build_ms);
        if (rc != PT_OK) { std::fprintf(stderr, "device BVH build failed (%d): %s\n", rc, pt_last_error()); return 1; }
        desc.nodes = device_nodes.data(); desc.num_nodes = int32_t(device_nodes.size()); desc.root = root;
        depth = d;
        std::printf("device %s build: %.2f ms\n", tree.c_str(), build_ms);
    } else if (tree != "reference") {
        std::fprintf(stderr, "unknown tree '%s'\n", tree.c_str());
        return 2;
    }
    std::printf("Maximum BVH depth: %d\n", depth);                                                                 // scene.cpp:148-149
    const double t1 = now();

    pt_scene* scene = nullptr;
    if ((rc = pt_scene_create(&desc, &scene)) != PT_OK) {                                                           // main.cu:186-187
        std::fprintf(stderr, "upload failed (%d): %s\n", rc, pt_last_error());
        return 1;
    }
    pt_render_params rp;
    pt_host_default_params(&cam, W, H, spp, &rp);                                                                   // main.cu:237 + constants
    if (nee) rp.flags = PT_RENDER_NEE;
    std::vector<float> fb(size_t(W) * H * 3);
    const double t2 = now();
    if ((rc = pt_render(scene, &rp, fb.data(), 0)) != PT_OK) {                                                      // main.cu:234,258-260
        std::fprintf(stderr, "render failed (%d): %s\n", rc, pt_last_error());
        return 1;
    }
    const double t3 = now();
    pt_counters c;
    pt_get_counters(scene, &c);
    const bool ppm = out.size() > 4 && out.compare(out.size() - 4, 4, ".ppm") == 0;
    rc = ppm ? pt_host_write_ppm(out.c_str(), fb.data(), W, H) : pt_host_write_pfm(out.c_str(), fb.data(), W, H);
    if (rc != PT_OK) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
    std::printf("parse+build %.3f s, upload %.3f s, GPU rendering took %.3f s (kernel %.3f ms, %llu segments, %.1f Msamples/s)\n",
                t1 - t0, t2 - t1, t3 - t2, c.kernel_ms, (unsigned long long)c.segments, c.segments / c.kernel_ms / 1e3);
    pt_scene_destroy(scene);
    pt_host_scene_destroy(hs);
    return 0;
}

// interactive_main.cpp — the reference's interactive loop (main.cu:272-344: render_progressive + cudaDeviceSynchronize + UpdateTexture
// every frame) on top of the C ABI, window and ImGui dropped: every frame adds `spf` samples per pixel to a device accumulation
// buffer (pt_render_accumulate), the "display" takes a copy of that buffer on the same stream and the host waits for the copy of the
// frame `lag` frames back before it goes on (lag 0 = the reference's loop: synchronise, then show).  With lag >= 1 consecutive
// frames overlap on the GPU (frames in flight, include/pt_api.h at pt_render_async); the accumulated image is the same bit for bit.
//
//   build: hipcc -std=c++17 -O2 -Iinclude examples/interactive_main.cpp -Lpathtracer_cuda_interactive_amd -lpt_host -lpt_hip \
//                -Wl,-rpath,'$ORIGIN/../pathtracer_cuda_interactive_amd' -o examples/interactive_main
//   run  : examples/interactive_main scene.xml|scene.pts out.pfm [width height spf frames [lag]]
//          out.pfm = accumulation buffer / samples, what UpdateTexture would show after the last frame (linear radiance)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "pt_api.h"
#include "pt_host.h"

#define HIP_OK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); return 1; } \
    } while (0)

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s scene.xml|scene.pts out.pfm [width height spf frames [lag]]\n", argv[0]);
        return 2;
    }
    const std::string path = argv[1], out = argv[2];
    pt_host_scene* hs = nullptr;
    const bool is_pts = path.size() > 4 && path.compare(path.size() - 4, 4, ".pts") == 0;
    int rc = is_pts ? pt_host_scene_load_pts(path.c_str(), &hs) : pt_host_scene_load_xml(path.c_str(), &hs);
    if (rc != PT_OK) { std::fprintf(stderr, "scene load failed (%d): %s\n", rc, pt_host_last_error()); return 1; }
    if ((rc = pt_host_scene_finalize(hs, PT_BVH_SORT_REFERENCE)) != PT_OK) {
        std::fprintf(stderr, "scene build failed (%d): %s\n", rc, pt_host_last_error());
        return 1;
    }
    pt_scene_desc desc;
    pt_camera cam;
    pt_host_scene_get_desc(hs, &desc);
    pt_host_scene_get_camera(hs, &cam);
    const int W = argc > 3 ? std::atoi(argv[3]) : cam.width;
    const int H = argc > 4 ? std::atoi(argv[4]) : cam.height;
    const int spf = argc > 5 ? std::atoi(argv[5]) : 2;                       // main.cu:131: g_samples_per_frame = 2
    const int frames = argc > 6 ? std::atoi(argv[6]) : 64;
    const int lag = argc > 7 ? std::atoi(argv[7]) : 1;
    if (W <= 0 || H <= 0 || spf <= 0 || frames <= 0 || lag < 0 || lag > 8) { std::fprintf(stderr, "bad size / count\n"); return 2; }

    pt_scene* scene = nullptr;
    if ((rc = pt_scene_create(&desc, &scene)) != PT_OK) { std::fprintf(stderr, "upload failed (%d): %s\n", rc, pt_last_error()); return 1; }
    pt_scene_set_option(scene, "timing_frames", 0);                           // nobody reads per-frame kernel times here
    pt_render_params rp;
    pt_host_default_params(&cam, W, H, spf, &rp);
    rp.stream_stride = spf * frames;                                         // PCG streams: room for every sample of the run

    const size_t bytes = size_t(W) * H * 3 * sizeof(float);
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    float* accum = nullptr;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&accum), bytes));
    std::vector<float*> shown(size_t(lag) + 1, nullptr);                     // what the display reads: one copy per frame in flight
    std::vector<hipEvent_t> done(size_t(lag) + 1);
    for (int k = 0; k <= lag; k++) {
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&shown[k]), bytes));
        HIP_OK(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
    }
    for (int f = 0; f < 4; f++) {                                            // untimed: the handle allocates its per-frame scratch on first use
        rp.sample_offset = 0;
        if ((rc = pt_render_accumulate(scene, &rp, shown[0], stream)) != PT_OK) { std::fprintf(stderr, "warm-up failed (%d): %s\n", rc, pt_last_error()); return 1; }
    }
    HIP_OK(hipStreamSynchronize(stream));
    const double t0 = now();
    for (int f = 0; f < frames; f++) {
        rp.sample_offset = f * spf;                                          // accumulationSampleCount, main.cu:337
        if ((rc = pt_render_accumulate(scene, &rp, accum, stream)) != PT_OK) {   // render_progressive, main.cu:333-334
            std::fprintf(stderr, "frame %d failed (%d): %s\n", f, rc, pt_last_error());
            return 1;
        }
        const int slot = f % (lag + 1);
        HIP_OK(hipMemcpyAsync(shown[slot], accum, bytes, hipMemcpyDeviceToDevice, stream));   // UpdateTexture's read, main.cu:339
        HIP_OK(hipEventRecord(done[slot], stream));
        if (f >= lag) HIP_OK(hipEventSynchronize(done[(f - lag) % (lag + 1)]));              // the frame the display shows now
    }
    HIP_OK(hipStreamSynchronize(stream));
    const double t1 = now();

    std::vector<float> fb(size_t(W) * H * 3);
    HIP_OK(hipMemcpy(fb.data(), accum, bytes, hipMemcpyDeviceToHost));
    const float scale = 1.0f / float(spf * frames);                           // the display divides by the sample count
    for (float& v : fb) v *= scale;
    if (pt_host_write_pfm(out.c_str(), fb.data(), W, H) != PT_OK) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
    std::printf("%d frames of %d samples per pixel at %dx%d, display %d frame(s) behind: %.1f frames/s (%.3f ms per frame)\n", frames, spf, W, H,
                lag, frames / (t1 - t0), (t1 - t0) / frames * 1e3);
    for (int k = 0; k <= lag; k++) { (void)hipFree(shown[k]); (void)hipEventDestroy(done[k]); }
    (void)hipFree(accum);
    (void)hipStreamDestroy(stream);
    pt_scene_destroy(scene);
    pt_host_scene_destroy(hs);
    return 0;
}

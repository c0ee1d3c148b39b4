// ASan/UBSan driver for the host scene pipeline and the CPU oracle (sanitizers run on the CPU build only).
// Loads every kind of input, finalizes with both BVH sort modes, renders with both oracle flavours, round-trips the
// .pts container, and feeds ~1000 truncated / bit-flipped XML and .pts inputs: all must be rejected or parsed without
// a crash, leak or UB report.  Built and run by tests/test_sanitizers.py (REPO and TMP are defined on the command line).
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <fstream>
#include "pt_host.h"
extern "C" {
#include "pt_oracle.h"
}
static void write(const std::string& p, const std::string& t) { std::ofstream f(p, std::ios::binary); f << t; }
int main() {
    const char* good[] = {REPO "/tests/data/mixed.xml", REPO "/tests/data/quirk_point_light_first.xml",
                          REPO "/tests/golden/scenes/cbox.pts", REPO "/tests/golden/scenes/teapot.pts"};
    for (const char* g : good) {
        pt_host_scene* hs = nullptr;
        std::string path = g;
        int rc = path.find(".pts") != std::string::npos ? pt_host_scene_load_pts(g, &hs) : pt_host_scene_load_xml(g, &hs);
        if (rc) { printf("FAIL load %s: %s\n", g, pt_host_last_error()); return 1; }
        for (int mode = 0; mode < 2; mode++) {
            if (pt_host_scene_finalize(hs, mode)) { printf("FAIL finalize\n"); return 1; }
            pt_scene_desc d; pt_camera cam; pt_render_params p;
            pt_host_scene_get_desc(hs, &d); pt_host_scene_get_camera(hs, &cam);
            pt_host_default_params(&cam, 24, 18, 2, &p);
            std::vector<float> fb(24 * 18 * 3);
            pt_oracle_opts o{0, 0, 3, 0}; pt_oracle_counters c;
            if (pt_oracle_render(&d, &p, &o, fb.data(), &c)) { printf("FAIL oracle\n"); return 1; }
            o.math_mode = 1; o.rng_mode = 1;
            if (pt_oracle_render(&d, &p, &o, fb.data(), &c)) { printf("FAIL oracle libm\n"); return 1; }
        }
        pt_host_scene_save_pts(hs, TMP "/out.pts");
        pt_host_scene* h2 = nullptr;
        if (pt_host_scene_load_pts(TMP "/out.pts", &h2)) { printf("FAIL reload\n"); return 1; }
        pt_host_scene_destroy(h2);
        pt_host_scene_destroy(hs);
    }
    // malformed inputs: every prefix of a valid scene + byte flips must fail cleanly (or parse), never crash
    std::ifstream f(REPO "/tests/data/mixed.xml", std::ios::binary);
    std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    int ok = 0, bad = 0;
    for (size_t n = 0; n < text.size(); n += 7) {
        write(REPO "/tests/data/_san_tmp.xml", text.substr(0, n));
        pt_host_scene* hs = nullptr;
        int rc = pt_host_scene_load_xml(REPO "/tests/data/_san_tmp.xml", &hs);
        if (rc == 0) { ok++; if (pt_host_scene_finalize(hs, 0) == 0) {} pt_host_scene_destroy(hs); } else bad++;
    }
    unsigned seed = 12345;
    for (int k = 0; k < 400; k++) {
        std::string t = text;
        for (int j = 0; j < 3; j++) { seed = seed * 1664525u + 1013904223u; t[seed % t.size()] = char(32 + (seed >> 16) % 95); }
        write(REPO "/tests/data/_san_tmp.xml", t);
        pt_host_scene* hs = nullptr;
        int rc = pt_host_scene_load_xml(REPO "/tests/data/_san_tmp.xml", &hs);
        if (rc == 0) { ok++; pt_host_scene_finalize(hs, 0); pt_host_scene_destroy(hs); } else bad++;
    }
    // corrupt .pts: truncations and flips
    std::ifstream g(REPO "/tests/golden/scenes/cbox.pts", std::ios::binary);
    std::string pts((std::istreambuf_iterator<char>(g)), std::istreambuf_iterator<char>());
    for (size_t n = 0; n < pts.size(); n += 37) {
        write(TMP "/t.pts", pts.substr(0, n));
        pt_host_scene* hs = nullptr;
        if (pt_host_scene_load_pts(TMP "/t.pts", &hs) == 0) { ok++; pt_host_scene_finalize(hs, 0); pt_host_scene_destroy(hs); } else bad++;
    }
    for (int k = 0; k < 300; k++) {
        std::string t = pts;
        seed = seed * 1664525u + 1013904223u; t[seed % t.size()] ^= char(1 << ((seed >> 20) % 8));
        write(TMP "/t.pts", t);
        pt_host_scene* hs = nullptr;
        if (pt_host_scene_load_pts(TMP "/t.pts", &hs) == 0) { ok++; pt_host_scene_finalize(hs, 0); pt_host_scene_destroy(hs); } else bad++;
    }
    remove(REPO "/tests/data/_san_tmp.xml");
    printf("sanitizer driver done: %d parsed, %d rejected\n", ok, bad);
    return 0;
}

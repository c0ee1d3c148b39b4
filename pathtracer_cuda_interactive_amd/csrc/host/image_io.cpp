// Image writers for the offline path (the reference computes `fb` and never saves it, main.cu:258-266).
//   PFM: lossless linear float RGB (bottom-to-top scanlines, little-endian marker -1.0)
//   PPM: 8-bit display encoding of opengl_display.cpp:104-111 — sqrt gamma, clamp to [0,1], int(255.99*c)
#include <cmath>
#include <cstdio>
#include <vector>

#include "parsed_scene.h"

using namespace pth;

static thread_local std::string g_img_err;

extern "C" {

int pt_host_write_pfm(const char* path, const float* fb, int width, int height) {
    if (!path || !fb || width <= 0 || height <= 0) return PT_ERR_INVALID_ARG;
    FILE* f = std::fopen(path, "wb");
    if (!f) return PT_ERR_IO;
    std::fprintf(f, "PF\n%d %d\n-1.0\n", width, height);
    bool ok = true;
    for (int j = height - 1; j >= 0 && ok; j--)          // PFM stores the bottom row first; fb row 0 is the top
        ok = std::fwrite(fb + size_t(j) * width * 3, sizeof(float), size_t(width) * 3, f) == size_t(width) * 3;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? PT_OK : PT_ERR_IO;
}

int pt_host_write_ppm(const char* path, const float* fb, int width, int height) {
    if (!path || !fb || width <= 0 || height <= 0) return PT_ERR_INVALID_ARG;
    std::vector<unsigned char> px(size_t(width) * height * 3);
    for (size_t k = 0; k < px.size(); k++) {
        float c = sqrtf(fb[k]);
        c = fminf(fmaxf(c, 0.0f), 1.0f);                 // NaN -> 0 (fmaxf returns the non-NaN operand)
        px[k] = (unsigned char)int(255.99f * c);
    }
    FILE* f = std::fopen(path, "wb");
    if (!f) return PT_ERR_IO;
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    bool ok = std::fwrite(px.data(), 1, px.size(), f) == px.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok ? PT_OK : PT_ERR_IO;
}

}  // extern "C"

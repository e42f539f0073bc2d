// san_driver.cpp -- exercises the oracle (test infrastructure) under AddressSanitizer + UndefinedBehaviorSanitizer:
// a seeded random scene through compute_cov3d, sort, preprocess and both rasterisers (analytic and two-triangle) on
// several threads, a non-zero image to blend onto, degenerate inputs (NaN positions, zero covariances), and the PLY
// reader on a file it writes itself.  Built by `make -C oracle san`; run by tests/test_sanitizers.py.  Exit 0 = clean.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "splat_oracle.h"

static uint64_t s_state = 0x9e3779b97f4a7c15ull;
static float frand() {   // xorshift64*, [0,1)
    s_state ^= s_state >> 12; s_state ^= s_state << 25; s_state ^= s_state >> 27;
    return (float)((s_state * 2685821657736338717ull) >> 40) / 16777216.0f;
}
static float nrand() { return std::sqrt(-2.0f * std::log(frand() + 1e-7f)) * std::cos(6.2831853f * frand()); }

int main() {
    const uint64_t n = 3000;
    std::vector<float> pos(4 * n), sc(3 * n), op(n), rot(4 * n), sh(48 * n), cov(9 * n);
    for (uint64_t i = 0; i < n; ++i) {
        for (int a = 0; a < 3; ++a) { pos[4 * i + a] = nrand() * 1.5f; sc[3 * i + a] = std::exp(nrand() * 0.8f - 4.0f); }
        pos[4 * i + 3] = 1.0f;
        op[i] = 1.0f / (1.0f + std::exp(-nrand() * 2.5f));
        for (int a = 0; a < 4; ++a) rot[4 * i + a] = nrand();
        for (int a = 0; a < 48; ++a) sh[48 * i + a] = nrand() * (a < 3 ? 1.0f : 0.15f);
    }
    for (int k = 0; k < 20; ++k) pos[4 * (k * 7) + (k % 3)] = NAN;          // never visible
    orc_compute_cov3d(n, sc.data(), rot.data(), cov.data());
    for (int k = 0; k < 20; ++k) std::memset(&cov[9 * (k * 11 + 3)], 0, 36);  // cov2d = lowpass * I
    const int H = 120, W = 168;
    std::vector<uint32_t> img((size_t)H * W), img2((size_t)H * W);
    for (auto& p : img) p = (uint32_t)(frand() * 4294967295.0f);
    img2 = img;
    float eye[3] = {0.2f, -0.1f, 4.0f};
    orc_camera cam;
    orc_camera_make((float)H, (float)W, eye, 0.4f, -0.2f, 0.01f, 15, &cam);
    orc_conventions conv;
    orc_default_conventions(&conv);
    orc_stats st;
    if (orc_render(n, pos.data(), cov.data(), op.data(), sh.data(), &cam, &conv, img.data(), 0, H, 3, &st) != 0) return 2;
    conv.raster = 1;
    if (orc_render(n, pos.data(), cov.data(), op.data(), sh.data(), &cam, &conv, img2.data(), 0, H, 2, &st) != 0) return 3;
    conv.raster = 0; conv.y_up = 0; conv.sample_half = 0; conv.zclip = 0;
    cam.lowpass = 0.0f;                                                      // singular cov2d for the zero covariances
    if (orc_render(n, pos.data(), cov.data(), op.data(), sh.data(), &cam, &conv, img2.data(), 10, 60, 4, &st) != 0) return 4;
    std::vector<uint32_t> order(n);
    orc_sort(n, pos.data(), cam.view, order.data());
    std::vector<orc_record> rec(n);
    orc_preprocess(n, pos.data(), cov.data(), op.data(), sh.data(), &cam, &conv, rec.data());
    // PLY round trip: a 5-vertex file with the 62 INRIA properties in a shuffled order plus an int property to skip
    const char* path = "/tmp/splat_san_driver.ply";
    FILE* f = std::fopen(path, "wb");
    if (!f) return 5;
    std::fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex 5\nproperty int junk\n");
    const char* names[] = {"x", "y", "z", "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3",
                           "f_dc_0", "f_dc_1", "f_dc_2", "f_rest_0", "f_rest_44"};
    for (const char* nm : names) std::fprintf(f, "property float %s\n", nm);
    std::fprintf(f, "end_header\n");
    for (int v = 0; v < 5; ++v) {
        int junk = v; std::fwrite(&junk, 4, 1, f);
        for (int k = 0; k < 16; ++k) { float x = nrand(); std::fwrite(&x, 4, 1, f); }
    }
    std::fclose(f);
    std::vector<float> p4(20), s3(15), o1(5), r4(20), s48(240);
    if (orc_load_ply(path, p4.data(), s3.data(), o1.data(), r4.data(), s48.data()) != 5) return 6;
    std::remove(path);
    std::printf("san_driver ok: %llu visible, %llu fragments\n", (unsigned long long)st.n_visible, (unsigned long long)st.n_fragments);
    return 0;
}

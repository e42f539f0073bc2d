// splat_cli -- the reference's main loop (src/main.rs:41-80) without the window: load a PLY (or the
// naive scene), orbit the camera in 10-degree yaw steps, time "pose update + clear + render" exactly
// like src/main.rs:71-77 and print it; optionally write the last frame as a PPM.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../../include/splat_host.hpp"

int main(int argc, char** argv) {
    const char* ply = nullptr; const char* out = nullptr;
    int W = 800, H = 600, frames = 36;
    bool streaming = false, fast = false, one_call = false;
    int in_flight = 2;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--ply") && i + 1 < argc) ply = argv[++i];
        else if (!std::strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
        else if (!std::strcmp(argv[i], "--size") && i + 2 < argc) { W = std::atoi(argv[++i]); H = std::atoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--frames") && i + 1 < argc) frames = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--stream")) streaming = true;
        else if (!std::strcmp(argv[i], "--in-flight") && i + 1 < argc) in_flight = std::max(1, std::min(4, std::atoi(argv[++i])));
        else if (!std::strcmp(argv[i], "--frame")) one_call = true;   // clear + render_to_buffer as ONE call (render_frame_to_buffer) into a pinned `color`
        else if (!std::strcmp(argv[i], "--fast")) fast = true;      // SPLAT_MODE_FAST: every colour byte within 1 of the exact frame
        else { std::fprintf(stderr, "usage: splat_cli [--ply file] [--size W H] [--frames N] [--frame | --stream [--in-flight 1..4]] [--fast] [--out frame.ppm]\n"); return 2; }
    }
    try {
        std::printf("Loading gaussians from %s\n", ply ? ply : "naive_gaussians()");
        std::vector<splat::Gaussian> g = ply ? splat::load_from_ply(ply) : splat::naive_gaussians();
        std::printf("Computing cov3d for each gaussian\n");
        for (auto& x : g) x.compute_cov3d();
        splat::Vec3 pos{0.0f, 0.0f, 5.0f};                    // CAMERA_POSITION, src/main.rs:13
        splat::GaussianSplatPipeline01 pipeline(g, splat::Camera((float)H, (float)W, &pos));
        if (fast) pipeline.set_mode(SPLAT_MODE_FAST);
        std::vector<uint32_t> color((size_t)W * H, 0u);
        if (streaming) {
            // the same loop with the present step decoupled: `in_flight` pinned frames rotate (2 = a double-buffered window;
            // the library holds up to four), frame f is "presented" (waited for) while the frames behind it render and
            // cross PCIe
            std::vector<uint32_t*> buf;
            for (int k = 0; k < in_flight; ++k) buf.push_back(splat::GaussianSplatPipeline01::alloc_frame((size_t)W * H));
            auto t0 = std::chrono::steady_clock::now();
            for (int f = 0; f < frames; ++f) {
                pipeline.camera.update_camera_pose();
                pipeline.stream_frame(buf[f % in_flight]);
                if (f >= in_flight - 1) pipeline.wait_frame(buf[(f - (in_flight - 1)) % in_flight]);
                pipeline.camera.update_yaw_angle(10.0f * 3.14159265f / 180.0f);
            }
            for (int f = std::max(0, frames - (in_flight - 1)); f < frames; ++f) pipeline.wait_frame(buf[f % in_flight]);
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::printf("Streamed %d frames in %.3f ms: %.3f ms per frame, host-visible\n", frames, ms, ms / frames);
            std::memcpy(color.data(), buf[(frames - 1) % in_flight], color.size() * 4);
            for (auto* b : buf) splat::GaussianSplatPipeline01::free_frame(b);
            frames = 0;
        }
        if (one_call && !streaming) splat::GaussianSplatPipeline01::pin_frame(color.data(), color.size());
        for (int f = 0; f < frames; ++f) {
            auto t0 = std::chrono::steady_clock::now();
            pipeline.camera.update_camera_pose();
            if (one_call) pipeline.render_frame_to_buffer(color.data());      // src/main.rs:73-74 as one call: nothing uploaded
            else {
                std::fill(color.begin(), color.end(), 0u);    // Buffer2d::fill([W,H], 0)
                pipeline.render_to_buffer(color.data());
            }
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            // (the one-call frame asks for no statistics -- they cost a read-back per frame: last_stats is not this frame's)
            if (one_call) std::printf("Rendering took %.3f ms\n", ms);
            else std::printf("Rendering took %.3f ms (gpu %.3f ms, %llu visible, %llu pairs)\n", ms, pipeline.last_stats.ms_total,
                             (unsigned long long)pipeline.last_stats.n_visible, (unsigned long long)pipeline.last_stats.n_pairs);
            pipeline.camera.update_yaw_angle(10.0f * 3.14159265f / 180.0f);   // Key::Right
        }
        if (one_call && !streaming) splat::GaussianSplatPipeline01::unpin_frame(color.data());
        if (out) {
            FILE* fp = std::fopen(out, "wb");
            if (!fp) { std::perror(out); return 1; }
            std::fprintf(fp, "P6\n%d %d\n255\n", W, H);
            for (uint32_t p : color) { unsigned char rgb[3] = {(unsigned char)(p >> 16), (unsigned char)(p >> 8), (unsigned char)p}; std::fwrite(rgb, 1, 3, fp); }
            std::fclose(fp);
        }
    } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}

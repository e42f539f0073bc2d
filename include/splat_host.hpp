// splat_host.hpp -- C++ host-side mirror of the reference crate's public API for the hot path
// (src/lib.rs: camera, gaussians, pipelines), written over the C ABI of include/splat_hip.h.
// Same names, argument meaning and semantics as the Rust items they stand for; where the Rust
// code panics (unwrap) these throw std::runtime_error.
//
//   splat::Camera                    src/camera.rs:4-126
//   splat::Gaussian                  src/gaussians.rs:30-38, 101-113
//   splat::GaussianList              src/gaussians.rs:408-462
//   splat::naive_gaussians()         src/gaussians.rs:319-374
//   splat::load_from_ply()           src/gaussians.rs:246-283, 375-405
//   splat::GaussianSplatPipeline01   src/pipelines.rs:54-87   (AoS, low-pass 0.01)
//   splat::GaussianSplatPipeline02   src/pipelines.rs:172-175, 259-281 (SoA, low-pass 0.3)
#ifndef SPLAT_HOST_HPP
#define SPLAT_HOST_HPP
#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "splat_hip.h"

namespace splat {

using Vec3 = std::array<float, 3>;
using Mat4 = std::array<float, 16>;   // column-major, like nalgebra::Matrix4::as_slice()

class Camera {
public:
    // Camera::new(h, w, start_position) -- height first (src/camera.rs:22); default position (0,0,3)
    Camera(float h, float w, const Vec3* start_position = nullptr);
    void compute_matrices();                       // :41-68
    const Mat4& get_view_matrix() const { return view_matrix_; }
    const Mat4& get_project_matrix() const { return projection_matrix_; }
    void update_resolution(float height, float width);
    Vec3 get_htanfovxy_focal() const;              // :84-89
    float get_focal() const;
    void update_pitch_angle(float delta);
    void update_yaw_angle(float delta);
    void update_camera_pose();                     // :103-126
    // the per-frame constants the C ABI takes
    splat_camera constants(float lowpass, int32_t sh_dim = 15) const;

    float h, w;                 // pub
    Vec3 position;              // pub
    bool is_pose_dirty = true;  // pub

private:
    float znear_ = 0.01f, zfar_ = 100.0f, fovy_;
    Vec3 target_{0, 0, 0}, up_{0, -1, 0};
    float yaw_ = 0.0f, pitch_ = 0.0f;
    bool is_intrin_dirty_ = true;
    Mat4 view_matrix_, projection_matrix_;         // identity until compute_matrices (:36-37)
};

struct Gaussian {               // src/gaussians.rs:30-38
    Vec3 position{0, 0, 0};
    Vec3 scale{0, 0, 0};
    float opacity = 0.0f;
    std::array<float, 4> rotation{0, 0, 0, 1};   // nalgebra coords order (i, j, k, w); identity
    std::array<float, 48> sh{};
    std::array<float, 9> cov3d{};                // column-major 3x3, zero until compute_cov3d (:254)
    void compute_cov3d();                        // :101-113 (host arithmetic, one Gaussian)
};

std::vector<Gaussian> naive_gaussians();
std::vector<Gaussian> load_from_ply(const std::string& filename);
struct GaussianList;
// The loader on the fast path: the same decode, activations and recentring, straight into
// the SoA upload buffers, on `threads` host threads (0 = all).  cov3d is left zero (compute_cov3d, K0 on the GPU).
GaussianList load_from_ply_soa(const std::string& filename, int threads = 0);
long long ply_vertex_count(const std::string& filename);      // header only

struct GaussianList {           // src/gaussians.rs:408-416, SoA
    std::vector<float> positions;   // 4 x N (x,y,z,1)
    std::vector<float> scales;      // 3 x N
    std::vector<float> opacities;   // N
    std::vector<float> rotations;   // 4 x N (i,j,k,w)
    std::vector<float> sh;          // 48 x N
    std::vector<float> cov3d;       // 3 x 3N
    size_t num_gaussians = 0;
    // from_vec: gathers the arrays and computes cov3d (:419-440); compute = false keeps each
    // Gaussian's own cov3d (what Pipeline01 renders with)
    static GaussianList from_vec(const std::vector<Gaussian>& v, bool compute = true, splat_ctx* gpu = nullptr);
    void compute_cov3d(splat_ctx* gpu = nullptr);   // :446-462; on the GPU (K0) when a context is given
};

namespace detail {
class PipelineBase {
public:
    ~PipelineBase();
    PipelineBase(const PipelineBase&) = delete;
    PipelineBase& operator=(const PipelineBase&) = delete;
    splat_stats last_stats{};
    // Viewer loop (src/main.rs:69-78) without the stall: stream_frame() == clear + render_to_buffer,
    // queued; the pixels are in `color` after wait_frame(color).  Keep two buffers in flight
    // (pinned ones from alloc_frame make the copy truly asynchronous).
    void wait_frame(const uint32_t* color);
    // `color.clear(0); render_to_buffer(&mut color)` (src/main.rs:73-74) as one synchronous call (splat_render_frame): `color` is
    // written, never read.  pin_frame(color, pixels) once (the reference's buffer lives as long as the window, src/main.rs:62)
    // lets the compositor write straight into it; unpin_frame before the memory goes away.
    static void pin_frame(uint32_t* color, size_t pixels);
    static void unpin_frame(uint32_t* color);
    // The scene is uploaded to the GPU once, at the first frame, and cached (the reference re-reads its
    // public `gaussians` field on every render_to_buffer, src/pipelines.rs:67-79).  After mutating
    // `gaussians` (edits, compute_cov3d after the first frame, ...) call this: the next frame uploads again.
    void invalidate_scene() { uploaded_ = nullptr; }
    // SPLAT_MODE_* flags (include/splat_hip.h) for this pipeline's GPU context; the reference has no such switch and
    // the default (0) is its arithmetic.  Must be called before the first frame (the context is created there).
    void set_mode(int mode);
    static uint32_t* alloc_frame(size_t pixels);
    static void free_frame(uint32_t* p);
protected:
    PipelineBase() = default;
    void render(const GaussianList& g, const Camera& cam, float lowpass, uint32_t* color);
    void render_frame(const GaussianList& g, const Camera& cam, float lowpass, uint32_t* color);
    void stream(const GaussianList& g, const Camera& cam, float lowpass, uint32_t* color);
    void ensure(const GaussianList& g);
    splat_ctx* ctx_ = nullptr;
    const void* uploaded_ = nullptr;
    int mode_ = 0;
};
}  // namespace detail

class GaussianSplatPipeline01 : public detail::PipelineBase {
public:
    GaussianSplatPipeline01(std::vector<Gaussian> gaussians, Camera camera);
    // Blends onto `color` (w*h u32, 0xAARRGGBB) exactly as src/pipelines.rs:66-86 does.
    void render_to_buffer(uint32_t* color);
    void render_frame_to_buffer(uint32_t* color);   // clear + render_to_buffer, src/main.rs:73-74, in one call (see PipelineBase)
    void stream_frame(uint32_t* color);        // cleared frame, asynchronous (see PipelineBase)
    std::vector<Gaussian> gaussians;   // pub
    Camera camera;                     // pub
private:
    GaussianList soa_;
};

class GaussianSplatPipeline02 : public detail::PipelineBase {
public:
    GaussianSplatPipeline02(GaussianList gaussians, Camera camera);
    void render_to_buffer(uint32_t* color);    // src/pipelines.rs:260-280
    void render_frame_to_buffer(uint32_t* color);
    void stream_frame(uint32_t* color);
    GaussianList gaussians;            // pub
    Camera camera;                     // pub
};

}  // namespace splat
#endif

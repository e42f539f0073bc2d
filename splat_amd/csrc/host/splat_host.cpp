// splat_host.cpp -- C++ host-side mirror of the reference API (include/splat_host.hpp).
// Product code: it never touches oracle/.  f32 arithmetic follows the reference's order; the
// heavy lifting (cov3d for lists, the whole frame) goes through the C ABI to the GPU.
#include "../../../include/splat_host.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <stdexcept>

namespace splat {
namespace {

inline Vec3 sub(const Vec3& a, const Vec3& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline Vec3 add(const Vec3& a, const Vec3& b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
inline float dot(const Vec3& a, const Vec3& b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
inline Vec3 cross(const Vec3& a, const Vec3& b) {
    return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
}
inline Vec3 normalize(const Vec3& v) {
    float n = std::sqrt(dot(v, v));
    return {v[0] / n, v[1] / n, v[2] / n};
}
// glm::rotation(angle, axis): Rotation3::from_axis_angle(Unit::new_normalize(axis), angle); row-major 3x3
void rotation(float angle, const Vec3& axis, float R[9]) {
    Vec3 u = normalize(axis);
    if (angle == 0.0f) { const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; std::memcpy(R, I, sizeof I); return; }
    float sqx = u[0] * u[0], sqy = u[1] * u[1], sqz = u[2] * u[2];
    float s = std::sin(angle), c = std::cos(angle), omc = 1.0f - c;
    R[0] = sqx + (1.0f - sqx) * c;        R[1] = u[0] * u[1] * omc - u[2] * s;  R[2] = u[0] * u[2] * omc + u[1] * s;
    R[3] = u[0] * u[1] * omc + u[2] * s;  R[4] = sqy + (1.0f - sqy) * c;        R[5] = u[1] * u[2] * omc - u[0] * s;
    R[6] = u[0] * u[2] * omc - u[1] * s;  R[7] = u[1] * u[2] * omc + u[0] * s;  R[8] = sqz + (1.0f - sqz) * c;
}
Vec3 rotate3(const float R[9], const Vec3& v) {
    return {(R[0] * v[0] + R[1] * v[1]) + R[2] * v[2], (R[3] * v[0] + R[4] * v[1]) + R[5] * v[2],
            (R[6] * v[0] + R[7] * v[1]) + R[8] * v[2]};
}
void check(int rc, splat_ctx* ctx, const char* what) {
    if (rc != SPLAT_OK) throw std::runtime_error(std::string(what) + ": " + splat_last_error(ctx));
}
// 3x3 column-major product with nalgebra's accumulation order
void mul3(const float* a, const float* b, float* c) {
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) {
            float acc = a[0 * 3 + i] * b[j * 3 + 0];
            acc = a[1 * 3 + i] * b[j * 3 + 1] + acc;
            acc = a[2 * 3 + i] * b[j * 3 + 2] + acc;
            c[j * 3 + i] = acc;
        }
}
}  // namespace

// ---------------------------------------------------------------- Camera (src/camera.rs)
Camera::Camera(float h_, float w_, const Vec3* start_position)
    : h(h_), w(w_), position(start_position ? *start_position : Vec3{0.0f, 0.0f, 3.0f}) {
    fovy_ = 3.14159265358979323846f / 2.0f;
    view_matrix_.fill(0.0f); projection_matrix_.fill(0.0f);
    for (int i = 0; i < 4; ++i) view_matrix_[i * 4 + i] = projection_matrix_[i * 4 + i] = 1.0f;
}

void Camera::compute_matrices() {
    Vec3 viewdir = normalize(sub(position, target_));
    float cos_angle = dot(viewdir, up_);
    float sg = std::signbit(pitch_) ? -1.0f : 1.0f;            // f32::signum
    if (cos_angle * sg > 0.99f) pitch_ = 0.0f;
    float Rx[9], Ry[9];
    rotation(yaw_, up_, Rx);
    Vec3 p1 = add(rotate3(Rx, sub(position, target_)), target_);
    Vec3 right = cross(up_, position);                         // the FIELD, src/camera.rs:58
    rotation(pitch_, right, Ry);
    Vec3 eye = add(rotate3(Ry, sub(p1, target_)), target_);
    // glm::look_at (right-handed)
    Vec3 z = normalize(sub(eye, target_)), x = normalize(cross(up_, z)), y = normalize(cross(z, x));
    Mat4 V; V.fill(0.0f);
    for (int c = 0; c < 3; ++c) { V[c * 4 + 0] = x[c]; V[c * 4 + 1] = y[c]; V[c * 4 + 2] = z[c]; }
    V[12] = -dot(x, eye); V[13] = -dot(y, eye); V[14] = -dot(z, eye); V[15] = 1.0f;
    view_matrix_ = V;
    // glm::perspective(aspect, fovy, near, far) == nalgebra Perspective3::new
    Mat4 P; P.fill(0.0f);
    float aspect = w / h;
    P[5] = 1.0f / std::tan(fovy_ / 2.0f);
    P[0] = P[5] / aspect;
    P[10] = (zfar_ + znear_) / (znear_ - zfar_);
    P[14] = zfar_ * znear_ * 2.0f / (znear_ - zfar_);
    P[11] = -1.0f;
    projection_matrix_ = P;
}
void Camera::update_resolution(float height, float width) { h = height; w = width; is_intrin_dirty_ = true; }
Vec3 Camera::get_htanfovxy_focal() const {
    float htany = std::tan(fovy_ / 2.0f);
    return {htany / h * w, htany, h / (2.0f * htany)};
}
float Camera::get_focal() const { return h / (2.0f * std::tan(fovy_ / 2.0f)); }
void Camera::update_pitch_angle(float delta) { pitch_ += delta; is_pose_dirty = true; }
void Camera::update_yaw_angle(float delta) { yaw_ += delta; is_pose_dirty = true; }
void Camera::update_camera_pose() { compute_matrices(); is_pose_dirty = false; }
splat_camera Camera::constants(float lowpass, int32_t sh_dim) const {
    splat_camera c;
    std::memcpy(c.view, view_matrix_.data(), sizeof c.view);
    std::memcpy(c.proj, projection_matrix_.data(), sizeof c.proj);
    c.w = w; c.h = h;
    Vec3 ht = get_htanfovxy_focal();
    c.htanx = ht[0]; c.htany = ht[1]; c.focal = ht[2];
    c.cam_pos[0] = position[0]; c.cam_pos[1] = position[1]; c.cam_pos[2] = position[2];
    c.lowpass = lowpass; c.sh_dim = sh_dim;
    return c;
}

// ---------------------------------------------------------------- Gaussian (src/gaussians.rs)
void Gaussian::compute_cov3d() {
    const auto& q = rotation;
    float a = q[0] * q[0], b = q[1] * q[1], c = q[2] * q[2], d = q[3] * q[3];
    a += c; b += d;
    float n = std::sqrt(a + b);
    float i = q[0] / n, j = q[1] / n, k = q[2] / n, w = q[3] / n;
    float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f, ik = i * k * 2.0f, jk = j * k * 2.0f, wi = w * i * 2.0f;
    float R[9], Rt[9], S[9] = {0}, RS[9];
    R[0] = ww + ii - jj - kk; R[3] = ij - wk;           R[6] = wj + ik;
    R[1] = wk + ij;           R[4] = ww - ii + jj - kk; R[7] = jk - wi;
    R[2] = ik - wj;           R[5] = wi + jk;           R[8] = ww - ii - jj + kk;
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) Rt[cc * 3 + r] = R[r * 3 + cc];
    S[0] = scale[0] * scale[0]; S[4] = scale[1] * scale[1]; S[8] = scale[2] * scale[2];
    mul3(R, S, RS);
    mul3(RS, Rt, cov3d.data());
}

std::vector<Gaussian> naive_gaussians() {
    const float pos[4][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    const float sc[4][3] = {{0.03f, 0.03f, 0.03f}, {0.2f, 0.03f, 0.03f}, {0.03f, 0.2f, 0.03f}, {0.03f, 0.03f, 0.2f}};
    const float col[4][3] = {{1, 0, 1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    std::vector<Gaussian> out(4);
    for (int g = 0; g < 4; ++g) {
        for (int a = 0; a < 3; ++a) {
            out[g].position[a] = pos[g][a];
            out[g].scale[a] = sc[g][a];
            out[g].sh[a] = (col[g][a] - 0.5f) / 0.28209f;
        }
        out[g].opacity = 1.0f;
        out[g].rotation = {0, 0, 0, 1};
    }
    return out;
}

// ---- PLY (src/gaussians.rs:246-283 set_property, :375-405 load_from_ply) ---------------------------------------
// mmap + direct decode of the vertex block (binary little endian or ascii); only `float` properties are
// consumed, by name; a non-"vertex" element throws (the reference panics).  Binary files are decoded straight
// into the SoA upload buffers (GaussianList) by `threads` host threads -- every value goes through the same
// libm call whatever thread handles it, and the recentring stays one sequential f32 sum, so the result does not
// depend on the thread count and equals the reference's load (and the oracle's reader) bit for bit.
namespace {
struct PlyProp { int offset, size; int dst; };      // dst: slot of the scatter table below, -1 = ignored
struct PlyFile {
    const char* base = nullptr; size_t len = 0;
    int fmt = -1; long long n = -1; int stride = 0; size_t payload = 0;
    std::vector<PlyProp> props;
    ~PlyFile() { if (base) munmap((void*)base, len); }
};
// destination codes: 0-2 pos, 3-5 scale(exp), 6 opacity(sigmoid), 7-10 rot (i,j,k,w), 11.. sh[k]
int ply_dst_of(const std::string& s) {
    static const char* const names[] = {"x", "y", "z", "scale_0", "scale_1", "scale_2", "opacity",
                                        "rot_1", "rot_2", "rot_3", "rot_0", "f_dc_0", "f_dc_1", "f_dc_2"};
    for (int k = 0; k < 14; ++k)
        if (s == names[k]) return k;
    if (s.rfind("f_rest_", 0) == 0) {
        int idx = std::atoi(s.c_str() + 7);
        if (idx < 0 || idx > 44) throw std::runtime_error("f_rest index out of range");   // sh[3+index] would panic
        return 14 + idx;
    }
    return -1;
}
void ply_open(const std::string& filename, PlyFile& f) {
    int fd = ::open(filename.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + filename);
    struct stat st;
    if (fstat(fd, &st) != 0) { ::close(fd); throw std::runtime_error("cannot stat " + filename); }
    f.len = (size_t)st.st_size;
    void* m = f.len ? mmap(nullptr, f.len, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    ::close(fd);
    if (f.len && m == MAP_FAILED) throw std::runtime_error("cannot map " + filename);
    f.base = (const char*)m;
    auto type_size = [](const std::string& t) -> int {
        if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
        if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
        if (t == "int" || t == "uint" || t == "int32" || t == "uint32" || t == "float" || t == "float32") return 4;
        if (t == "double" || t == "float64") return 8;
        return 0;
    };
    size_t pos = 0;
    auto next_line = [&]() -> std::string {
        size_t e = pos;
        while (e < f.len && f.base[e] != '\n') ++e;
        std::string l(f.base + pos, e - pos);
        pos = e < f.len ? e + 1 : e;
        if (!l.empty() && l.back() == '\r') l.pop_back();
        return l;
    };
    if (next_line() != "ply") throw std::runtime_error(filename + ": not a PLY file");
    bool done = false;
    while (pos < f.len) {
        std::istringstream ls(next_line());
        std::string kw; ls >> kw;
        if (kw == "format") { std::string fm; ls >> fm; f.fmt = fm == "ascii" ? 0 : fm == "binary_little_endian" ? 1 : -1; }
        else if (kw == "element") {
            std::string name; ls >> name >> f.n;
            if (name != "vertex") throw std::runtime_error("Unexpected element!");
        } else if (kw == "property") {
            std::string ty, name; ls >> ty >> name;
            if (ty == "list") throw std::runtime_error("list properties are not supported in the vertex element");
            int sz = type_size(ty);
            if (!sz) throw std::runtime_error("unknown PLY type " + ty);
            bool isf = (ty == "float" || ty == "float32");
            f.props.push_back({f.stride, sz, isf ? ply_dst_of(name) : -1});
            f.stride += sz;
        } else if (kw == "end_header") { done = true; break; }
    }
    if (!done || f.fmt < 0 || f.n < 0) throw std::runtime_error(filename + ": unsupported or incomplete PLY header");
    f.payload = pos;
    if (f.fmt == 1 && f.payload + (size_t)f.n * f.stride > f.len) throw std::runtime_error(filename + ": truncated payload");
}
// one decoded value into the SoA arrays (set_property, :261-279)
inline void ply_store(GaussianList& l, size_t i, int dst, float v) {
    if (dst < 3) l.positions[4 * i + dst] = v;
    else if (dst < 6) l.scales[3 * i + dst - 3] = std::exp(v);                   // :264-266
    else if (dst == 6) l.opacities[i] = 1.0f / (1.0f + std::exp(-v));            // :267
    else if (dst < 11) l.rotations[4 * i + dst - 7] = v;                         // :268-271
    else l.sh[48 * i + dst - 11] = v;                                            // :272-279, no transpose
}
}  // namespace

GaussianList load_from_ply_soa(const std::string& filename, int threads) {
    PlyFile f;
    ply_open(filename, f);
    const size_t n = (size_t)f.n;
    GaussianList l;
    l.num_gaussians = n;
    l.positions.assign(4 * n, 0.0f); l.scales.assign(3 * n, 0.0f); l.opacities.assign(n, 0.0f);
    l.rotations.assign(4 * n, 0.0f); l.sh.assign(48 * n, 0.0f); l.cov3d.assign(9 * n, 0.0f);
    for (size_t i = 0; i < n; ++i) { l.positions[4 * i + 3] = 1.0f; l.rotations[4 * i + 3] = 1.0f; }   // Gaussian::new: identity quaternion
    if (threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
    threads = (int)std::min<size_t>((size_t)threads, std::max<size_t>(1, n / 4096));
    if (f.fmt == 1) {
        std::vector<PlyProp> used;
        for (const PlyProp& p : f.props) if (p.dst >= 0) used.push_back(p);
        auto work = [&](size_t a, size_t b) {
            const char* p = f.base + f.payload + a * (size_t)f.stride;
            for (size_t i = a; i < b; ++i, p += f.stride)
                for (const PlyProp& pr : used) { float v; std::memcpy(&v, p + pr.offset, 4); ply_store(l, i, pr.dst, v); }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < threads; ++t) th.emplace_back(work, n * t / threads, n * (t + 1) / threads);
        work(0, n / threads);
        for (auto& x : th) x.join();
    } else {
        const char* p = f.base + f.payload; const char* endp = f.base + f.len;
        for (size_t i = 0; i < n; ++i)
            for (const PlyProp& pr : f.props) {
                std::string tok;
                while (p < endp && std::isspace((unsigned char)*p)) ++p;
                while (p < endp && !std::isspace((unsigned char)*p)) tok.push_back(*p++);
                if (tok.empty()) throw std::runtime_error(filename + ": truncated payload");
                double v = std::strtod(tok.c_str(), nullptr);
                if (pr.dst >= 0) ply_store(l, i, pr.dst, (float)v);
            }
    }
    // recentre: ONE sequential f32 sum in index order (src/gaussians.rs:394-402), then the subtraction in parallel
    float ax = 0, ay = 0, az = 0;
    for (size_t i = 0; i < n; ++i) { ax += l.positions[4 * i]; ay += l.positions[4 * i + 1]; az += l.positions[4 * i + 2]; }
    const float nf = (float)n;
    ax /= nf; ay /= nf; az /= nf;
    auto sub = [&](size_t a, size_t b) {
        for (size_t i = a; i < b; ++i) { l.positions[4 * i] -= ax; l.positions[4 * i + 1] -= ay; l.positions[4 * i + 2] -= az; }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < threads; ++t) th.emplace_back(sub, n * t / threads, n * (t + 1) / threads);
        sub(0, n / threads);
        for (auto& x : th) x.join();
    }
    return l;
}

long long ply_vertex_count(const std::string& filename) { PlyFile f; ply_open(filename, f); return f.n; }

// load_from_ply (src/gaussians.rs:375-405): the AoS `Vec<Gaussian>` the reference returns, from the same decode
std::vector<Gaussian> load_from_ply(const std::string& filename) {
    GaussianList l = load_from_ply_soa(filename, 0);
    std::vector<Gaussian> out(l.num_gaussians);
    for (size_t i = 0; i < out.size(); ++i) {
        Gaussian& g = out[i];
        for (int a = 0; a < 3; ++a) { g.position[a] = l.positions[4 * i + a]; g.scale[a] = l.scales[3 * i + a]; }
        g.opacity = l.opacities[i];
        std::memcpy(g.rotation.data(), &l.rotations[4 * i], 16);
        std::memcpy(g.sh.data(), &l.sh[48 * i], 192);
    }
    return out;
}

// ---------------------------------------------------------------- GaussianList
GaussianList GaussianList::from_vec(const std::vector<Gaussian>& v, bool compute, splat_ctx* gpu) {
    GaussianList l;
    size_t n = v.size();
    l.num_gaussians = n;
    l.positions.resize(4 * n); l.scales.resize(3 * n); l.opacities.resize(n); l.rotations.resize(4 * n);
    l.sh.resize(48 * n); l.cov3d.assign(9 * n, 0.0f);
    for (size_t i = 0; i < n; ++i) {
        const Gaussian& g = v[i];
        for (int a = 0; a < 3; ++a) { l.positions[4 * i + a] = g.position[a]; l.scales[3 * i + a] = g.scale[a]; }
        l.positions[4 * i + 3] = 1.0f;
        l.opacities[i] = g.opacity;
        std::memcpy(&l.rotations[4 * i], g.rotation.data(), 16);
        std::memcpy(&l.sh[48 * i], g.sh.data(), 192);
        std::memcpy(&l.cov3d[9 * i], g.cov3d.data(), 36);
    }
    if (compute) l.compute_cov3d(gpu);
    return l;
}
void GaussianList::compute_cov3d(splat_ctx* gpu) {
    if (gpu) {   // kernel K0
        check(splat_compute_cov3d(gpu, num_gaussians, scales.data(), rotations.data(), cov3d.data()), gpu, "splat_compute_cov3d");
        return;
    }
    Gaussian g;
    for (size_t i = 0; i < num_gaussians; ++i) {
        std::memcpy(g.scale.data(), &scales[3 * i], 12);
        std::memcpy(g.rotation.data(), &rotations[4 * i], 16);
        g.compute_cov3d();
        std::memcpy(&cov3d[9 * i], g.cov3d.data(), 36);
    }
}

// ---------------------------------------------------------------- pipelines (src/pipelines.rs)
namespace detail {
PipelineBase::~PipelineBase() { if (ctx_) splat_destroy(ctx_); }
void PipelineBase::set_mode(int mode) {
    if (ctx_) throw std::logic_error("set_mode: the GPU context already exists (call before the first frame)");
    mode_ = mode;
}
void PipelineBase::ensure(const GaussianList& g) {
    if (!ctx_) {
        // (a libsplat_hip.so built from another header would write a splat_stats of another size into last_stats)
        if (splat_abi_version() != SPLAT_ABI_VERSION || splat_stats_size() != sizeof(splat_stats))
            throw std::runtime_error("libsplat_hip.so was built from another include/splat_hip.h (ABI version mismatch)");
        splat_config cfg;
        splat_default_config(&cfg);
        cfg.mode = mode_;
        if (splat_create(&cfg, &ctx_) != SPLAT_OK) throw std::runtime_error(std::string("splat_create: ") + splat_last_error(nullptr));
    }
    if (uploaded_ != (const void*)&g) {        // lazily, once per scene object
        check(splat_upload_scene(ctx_, g.num_gaussians, g.positions.data(), g.cov3d.data(), g.opacities.data(), g.sh.data()),
              ctx_, "splat_upload_scene");
        uploaded_ = &g;
    }
}
void PipelineBase::render(const GaussianList& g, const Camera& cam, float lowpass, uint32_t* color) {
    ensure(g);
    splat_camera c = cam.constants(lowpass, 15);   // the literal at src/pipelines.rs:100,189
    check(splat_render(ctx_, &c, color, &last_stats), ctx_, "splat_render");
}
void PipelineBase::render_frame(const GaussianList& g, const Camera& cam, float lowpass, uint32_t* color) {
    ensure(g);
    splat_camera c = cam.constants(lowpass, 15);
    check(splat_render_frame(ctx_, &c, color, nullptr), ctx_, "splat_render_frame");      // (no statistics: they are read back from the device)
}
void PipelineBase::pin_frame(uint32_t* color, size_t pixels) {
    if (splat_host_register(color, (uint64_t)pixels * 4) != SPLAT_OK) throw std::runtime_error("splat_host_register: the driver cannot page-lock this memory");
}
void PipelineBase::unpin_frame(uint32_t* color) { (void)splat_host_unregister(color); }
void PipelineBase::stream(const GaussianList& g, const Camera& cam, float lowpass, uint32_t* color) {
    ensure(g);
    splat_camera c = cam.constants(lowpass, 15);
    check(splat_render_stream(ctx_, &c, color), ctx_, "splat_render_stream");
}
void PipelineBase::wait_frame(const uint32_t* color) {
    if (!ctx_) throw std::runtime_error("wait_frame: nothing was streamed");
    check(splat_stream_wait(ctx_, color), ctx_, "splat_stream_wait");
}
uint32_t* PipelineBase::alloc_frame(size_t pixels) {
    void* p = splat_host_alloc((uint64_t)pixels * 4);
    if (!p) throw std::bad_alloc();
    return static_cast<uint32_t*>(p);
}
void PipelineBase::free_frame(uint32_t* p) { splat_host_free(p); }
}  // namespace detail

GaussianSplatPipeline01::GaussianSplatPipeline01(std::vector<Gaussian> g, Camera cam)
    : gaussians(std::move(g)), camera(std::move(cam)) {}
void GaussianSplatPipeline01::render_to_buffer(uint32_t* color) {
    // AoS: cov3d is whatever each Gaussian carries (zero unless the caller ran compute_cov3d, main.rs:24-26)
    if (soa_.num_gaussians != gaussians.size() || uploaded_ == nullptr) soa_ = GaussianList::from_vec(gaussians, false);
    render(soa_, camera, 0.01f, color);
}
void GaussianSplatPipeline01::render_frame_to_buffer(uint32_t* color) {
    if (soa_.num_gaussians != gaussians.size() || uploaded_ == nullptr) soa_ = GaussianList::from_vec(gaussians, false);
    render_frame(soa_, camera, 0.01f, color);
}
void GaussianSplatPipeline01::stream_frame(uint32_t* color) {
    if (soa_.num_gaussians != gaussians.size() || uploaded_ == nullptr) soa_ = GaussianList::from_vec(gaussians, false);
    stream(soa_, camera, 0.01f, color);
}
void GaussianSplatPipeline02::stream_frame(uint32_t* color) { stream(gaussians, camera, 0.3f, color); }
GaussianSplatPipeline02::GaussianSplatPipeline02(GaussianList g, Camera cam) : gaussians(std::move(g)), camera(std::move(cam)) {}
void GaussianSplatPipeline02::render_to_buffer(uint32_t* color) { render(gaussians, camera, 0.3f, color); }
void GaussianSplatPipeline02::render_frame_to_buffer(uint32_t* color) { render_frame(gaussians, camera, 0.3f, color); }

}  // namespace splat

// ---------------------------------------------------------------- C shim for the ctypes tests
extern "C" {
void splat_host_camera(float h, float w, const float* pos, float yaw, float pitch, int update, float lowpass,
                       splat_camera* out) {
    splat::Vec3 p{pos[0], pos[1], pos[2]};
    splat::Camera cam(h, w, &p);
    if (yaw != 0.0f) cam.update_yaw_angle(yaw);
    if (pitch != 0.0f) cam.update_pitch_angle(pitch);
    if (update) cam.update_camera_pose();
    *out = cam.constants(lowpass, 15);
}
// returns n (or -1 with the message in err); arrays may be NULL to query the count (header only)
long long splat_host_load_ply(const char* path, float* pos4, float* scales3, float* opacity, float* rot4, float* sh48,
                              char* err, int errlen) {
    try {
        if (!pos4) {
            return splat::ply_vertex_count(path);
        }
        splat::GaussianList l = splat::load_from_ply_soa(path, 0);
        std::memcpy(pos4, l.positions.data(), l.positions.size() * 4);
        std::memcpy(scales3, l.scales.data(), l.scales.size() * 4);
        std::memcpy(opacity, l.opacities.data(), l.opacities.size() * 4);
        std::memcpy(rot4, l.rotations.data(), l.rotations.size() * 4);
        std::memcpy(sh48, l.sh.data(), l.sh.size() * 4);
        return (long long)l.num_gaussians;
    } catch (const std::exception& e) {
        if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
        return -1;
    }
}
// the loader alone, timed: decode + activations + recentring into the SoA buffers with `threads` host threads
// (0 = all); returns seconds, or -1.  *n_out = Gaussians loaded.
double splat_host_time_load(const char* path, int threads, long long* n_out, char* err, int errlen) {
    try {
        auto t0 = std::chrono::steady_clock::now();
        splat::GaussianList l = splat::load_from_ply_soa(path, threads);
        auto t1 = std::chrono::steady_clock::now();
        if (n_out) *n_out = (long long)l.num_gaussians;
        return std::chrono::duration<double>(t1 - t0).count();
    } catch (const std::exception& e) {
        if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
        return -1.0;
    }
}
void splat_host_cov3d(unsigned long long n, const float* scales3, const float* rot4, float* cov3d) {
    splat::Gaussian g;
    for (unsigned long long i = 0; i < n; ++i) {
        std::memcpy(g.scale.data(), scales3 + 3 * i, 12);
        std::memcpy(g.rotation.data(), rot4 + 4 * i, 16);
        g.compute_cov3d();
        std::memcpy(cov3d + 9 * i, g.cov3d.data(), 36);
    }
}
// Pipeline01 / Pipeline02 render_to_buffer on the naive scene or a PLY (path may be NULL = naive_gaussians)
int splat_host_render(int pipeline, const char* ply_path, float h, float w, const float* pos, uint32_t* color,
                      char* err, int errlen) {
    try {
        std::vector<splat::Gaussian> v = ply_path ? splat::load_from_ply(ply_path) : splat::naive_gaussians();
        splat::Vec3 p{pos[0], pos[1], pos[2]};
        splat::Camera cam(h, w, &p);
        cam.update_camera_pose();
        // pipeline 1 / 2: render_to_buffer (blends onto `color`); 11 / 12: render_frame_to_buffer (`color` is written, never read)
        if (pipeline == 1 || pipeline == 11) {
            for (auto& g : v) g.compute_cov3d();               // src/main.rs:24-26
            splat::GaussianSplatPipeline01 pl(v, cam);
            if (pipeline == 1) pl.render_to_buffer(color); else pl.render_frame_to_buffer(color);
        } else {
            splat::GaussianSplatPipeline02 pl(splat::GaussianList::from_vec(v), cam);
            if (pipeline == 2) pl.render_to_buffer(color); else pl.render_frame_to_buffer(color);
        }
        return 0;
    } catch (const std::exception& e) {
        if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
        return -1;
    }
}
}

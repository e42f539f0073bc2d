// splat_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
// See splat_oracle.h for scope and parity status.  Every function cites the
// reference lines it restates (paths relative to /root/reference).
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile)
#include "splat_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// nalgebra 0.32 semantics (column-major storage; small products are a gemv per
// column, each gemv an axpy per input column, i.e. left-to-right accumulation
// y_i = ((A_i0*x_0 + A_i1*x_1) + A_i2*x_2) + ..., never fused).
// ---------------------------------------------------------------------------
struct M3 { float m[9]; float& at(int r, int c) { return m[c * 3 + r]; } float at(int r, int c) const { return m[c * 3 + r]; } };
struct M4 { float m[16]; float& at(int r, int c) { return m[c * 4 + r]; } float at(int r, int c) const { return m[c * 4 + r]; } };

inline M3 mul(const M3& a, const M3& b) {
    M3 c;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) {
            float acc = a.at(i, 0) * b.at(0, j);
            acc = a.at(i, 1) * b.at(1, j) + acc;
            acc = a.at(i, 2) * b.at(2, j) + acc;
            c.at(i, j) = acc;
        }
    return c;
}
inline M3 transpose(const M3& a) {
    M3 t;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t.at(i, j) = a.at(j, i);
    return t;
}
inline void mul4(const float* m /*col-major 4x4*/, const float v[4], float out[4]) {
    for (int i = 0; i < 4; ++i) {
        float acc = m[0 * 4 + i] * v[0];
        acc = m[1 * 4 + i] * v[1] + acc;
        acc = m[2 * 4 + i] * v[2] + acc;
        acc = m[3 * 4 + i] * v[3] + acc;
        out[i] = acc;
    }
}
// Rust `x as u8`: truncate toward zero, saturate, NaN -> 0.
inline uint32_t as_u8(float v) {
    if (!(v > 0.0f)) return 0u;
    if (v >= 255.0f) return 255u;
    return (uint32_t)(int)v;
}
// Rust f32::min / f32::max ignore a NaN operand, like fminf/fmaxf.
inline float rmin(float a, float b) { return fminf(a, b); }
inline float rmax(float a, float b) { return fmaxf(a, b); }

// SH basis constants, src/gaussians.rs:11-26
const float SH_C0 = 0.28209479177387814f;
const float SH_C1 = 0.4886025119029199f;
const float SH_C2_0 = 1.0925484305920792f;
const float SH_C2_1 = -1.0925484305920792f;
const float SH_C2_2 = 0.31539156525252005f;
const float SH_C2_3 = -1.0925484305920792f;
const float SH_C2_4 = 0.5462742152960396f;
const float SH_C3_0 = -0.5900435899266435f;
const float SH_C3_1 = 2.890611442640554f;
const float SH_C3_2 = -0.4570457994644658f;
const float SH_C3_3 = 0.3731763325901154f;
const float SH_C3_4 = -0.4570457994644658f;
const float SH_C3_5 = 1.445305721320277f;
const float SH_C3_6 = -0.5900435899266435f;

struct V3 { float x, y, z; };
inline V3 operator*(float s, V3 v) { return {s * v.x, s * v.y, s * v.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
// nalgebra dot for 3-vectors: (a+b)+c ; norm = sqrt ; normalize = component / norm
inline float norm3(V3 v) { return sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z); }
inline V3 normalize3(V3 v) { float n = norm3(v); return {v.x / n, v.y / n, v.z / n}; }
inline V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// src/gaussians.rs:40-99
V3 eval_sh(const float* sh, int sh_dim, V3 dir) {
    V3 color = SH_C0 * ld3(sh + 0);
    if (sh_dim > 3) {
        V3 c1 = ld3(sh + 3), c2 = ld3(sh + 6), c3 = ld3(sh + 9);
        float x = dir.x, y = dir.y, z = dir.z;
        color = color - (SH_C1 * y) * c1 + (SH_C1 * z) * c2 - (SH_C1 * x) * c3;
        if (sh_dim > 12) {
            V3 c4 = ld3(sh + 12), c5 = ld3(sh + 15), c6 = ld3(sh + 18), c7 = ld3(sh + 21), c8 = ld3(sh + 24);
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            color = color + (SH_C2_0 * xy) * c4 + (SH_C2_1 * yz) * c5 + (SH_C2_2 * (2.0f * zz - xx - yy)) * c6 +
                    (SH_C2_3 * xz) * c7 + (SH_C2_4 * (xx - yy)) * c8;
            if (sh_dim > 27) {
                V3 c9 = ld3(sh + 27), c10 = ld3(sh + 30), c11 = ld3(sh + 33), c12 = ld3(sh + 36);
                V3 c13 = ld3(sh + 39), c14 = ld3(sh + 42), c15 = ld3(sh + 45);
                color = color + (SH_C3_0 * y * (3.0f * xx - yy)) * c9 + (SH_C3_1 * xy * z) * c10 +
                        (SH_C3_2 * y * (4.0f * zz - xx - yy)) * c11 +
                        (SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy)) * c12 +
                        (SH_C3_4 * x * (4.0f * zz - xx - yy)) * c13 + (SH_C3_5 * z * (xx - yy)) * c14 +
                        (SH_C3_6 * x * (xx - 3.0f * yy)) * c15;
            }
        }
    }
    color = color + V3{0.5f, 0.5f, 0.5f};   // HALF, :28,:97 -- no clamp
    return color;
}

// src/gaussians.rs:101-113 (== :446-462).  rot = (i,j,k,w).
void cov3d_one(const float* s, const float* q, float* out9) {
    // UnitQuaternion::from_quaternion: q / |q| ; nalgebra's 4-vector dot is (q0q0+q2q2)+(q1q1+q3q3)
    float a = q[0] * q[0], b = q[1] * q[1], c = q[2] * q[2], d = q[3] * q[3];
    a += c; b += d;
    float n = sqrtf(a + b);
    float i = q[0] / n, j = q[1] / n, k = q[2] / n, w = q[3] / n;
    // to_rotation_matrix (nalgebra geometry/quaternion.rs)
    float ww = w * w, ii = i * i, jj = j * j, kk = k * k;
    float ij = i * j * 2.0f, wk = w * k * 2.0f, wj = w * j * 2.0f;
    float ik = i * k * 2.0f, jk = j * k * 2.0f, wi = w * i * 2.0f;
    M3 R;
    R.at(0, 0) = ww + ii - jj - kk; R.at(0, 1) = ij - wk;           R.at(0, 2) = wj + ik;
    R.at(1, 0) = wk + ij;           R.at(1, 1) = ww - ii + jj - kk; R.at(1, 2) = jk - wi;
    R.at(2, 0) = ik - wj;           R.at(2, 1) = wi + jk;           R.at(2, 2) = ww - ii - jj + kk;
    M3 S; std::memset(S.m, 0, sizeof S.m);
    S.at(0, 0) = s[0] * s[0]; S.at(1, 1) = s[1] * s[1]; S.at(2, 2) = s[2] * s[2];
    M3 cov = mul(mul(R, S), transpose(R));      // rotation * scale * rotation.transpose()
    std::memcpy(out9, cov.m, sizeof cov.m);
}

// src/gaussians.rs:114-161 (lowpass .01) / :473-522 (lowpass .3)
// corrected != 0: NOT the reference -- the EWA Jacobian the way the 3DGS paper has it (J enters
// transposed, so the perspective-shear terms -f*tx/tz^2, -f*ty/tz^2 reach the 2x2 block instead of
// the discarded third column: quirk Q9 / SURVEY section 8(f) rank 2).  Same products, same order.
void project_cov2d(const float* pos, const float* cov3d9, const orc_camera* cam, float out[4], float* depth_out,
                   int corrected = 0) {
    float pw[4] = {pos[0], pos[1], pos[2], 1.0f}, pc[4];
    mul4(cam->view, pw, pc);
    if (depth_out) *depth_out = pc[2];
    float tan_fovx = cam->htanx, tan_fovy = cam->htany, focal = cam->focal;
    float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    float txtz = pc[0] / pc[2], tytz = pc[1] / pc[2];
    float tx = rmin(limx, rmax(-limx, txtz)) * pc[2];
    float ty = rmin(limy, rmax(-limy, tytz)) * pc[2];
    float tz = pc[2];
    M3 J;                                       // Matrix3::new takes ROW-major arguments
    J.at(0, 0) = focal / tz; J.at(0, 1) = 0.0f;       J.at(0, 2) = -(focal * tx) / (tz * tz);
    J.at(1, 0) = 0.0f;       J.at(1, 1) = focal / tz; J.at(1, 2) = -(focal * ty) / (tz * tz);
    J.at(2, 0) = 0.0f;       J.at(2, 1) = 0.0f;       J.at(2, 2) = 0.0f;
    M3 V3x3;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) V3x3.at(r, c) = cam->view[c * 4 + r];
    M3 W = transpose(V3x3);
    M3 T = corrected ? mul(W, transpose(J)) : mul(W, J);
    M3 S; std::memcpy(S.m, cov3d9, sizeof S.m);
    M3 cov = mul(mul(transpose(T), transpose(S)), T);
    out[0] = cov.at(0, 0) + cam->lowpass;       // column-major 2x2: (0,0),(1,0),(0,1),(1,1)
    out[1] = cov.at(1, 0);
    out[2] = cov.at(0, 1);
    out[3] = cov.at(1, 1) + cam->lowpass;
}

// Exactly-covered pixel interval of one axis: {p in [0,size) : |p + off - c| <= h}.
// Returns false when empty.
bool covered_interval(float c, float h, float off, int size, int* lo, int* hi) {
    float flo = c - h - off, fhi = c + h - off;
    if (!(fhi >= -2.0f) || !(flo <= (float)size + 2.0f)) return false;
    int a = (int)rmax(floorf(flo) - 1.0f, 0.0f);
    int b = (int)rmin(ceilf(fhi) + 1.0f, (float)(size - 1));
    while (a <= b && !(fabsf(((float)a + off) - c) <= h)) ++a;
    while (b >= a && !(fabsf(((float)b + off) - c) <= h)) --b;
    if (a > b) return false;
    *lo = a; *hi = b;
    return true;
}

// Pipeline::vertex, once per Gaussian: src/pipelines.rs:96-125 + gaussian_vertex_shader :17-51
void preprocess_one(const float* pos4, const float* cov3d9, float opacity, const float* sh48,
                    const orc_camera* cam, const orc_conventions* conv, orc_record* r, int* singular) {
    std::memset(r, 0, sizeof *r);
    V3 p = ld3(pos4);
    V3 dir = normalize3(p - ld3(cam->cam_pos));                 // :99
    V3 col = eval_sh(sh48, cam->sh_dim, dir);                   // :100
    r->rgb[0] = col.x; r->rgb[1] = col.y; r->rgb[2] = col.z;
    r->opacity = opacity;
    project_cov2d(pos4, cov3d9, cam, r->cov2d, &r->depth, conv->corrected_projection);      // :102
    // try_inverse of a 2x2 (nalgebra linalg/inverse.rs): det = m11*m22 - m21*m12
    float m11 = r->cov2d[0], m21 = r->cov2d[1], m12 = r->cov2d[2], m22 = r->cov2d[3];
    float det = m11 * m22 - m21 * m12;
    float pw[4] = {pos4[0], pos4[1], pos4[2], 1.0f}, pv[4], q[4];
    mul4(cam->view, pw, pv);                                    // :39
    mul4(cam->proj, pv, q);                                     // :41
    for (int i = 0; i < 4; ++i) r->ndc[i] = q[i] / q[3];        // :42
    if (det == 0.0f) { *singular = 1; return; }                 // reference: unwrap() panic, :22
    r->conic[0] = m22 / det;                                    // inv(0,0)
    r->conic[1] = -m12 / det;                                   // inv(0,1)
    r->conic[2] = m11 / det;                                    // inv(1,1)
    r->hx = 3.0f * sqrtf(m11);                                  // :27
    r->hy = 3.0f * sqrtf(m22);
    // euc maps NDC to the target: x_px = W*(x*0.5+0.5); y_px = H*(y*-0.5+0.5) when +y is up.
    r->cx = (r->ndc[0] * 0.5f + 0.5f) * cam->w;
    r->cy = conv->y_up ? (r->ndc[1] * -0.5f + 0.5f) * cam->h : (r->ndc[1] * 0.5f + 0.5f) * cam->h;
    bool finite = std::isfinite(r->cx) && std::isfinite(r->cy) && std::isfinite(r->hx) && std::isfinite(r->hy) &&
                  std::isfinite(r->conic[0]) && std::isfinite(r->conic[1]) && std::isfinite(r->conic[2]) &&
                  std::isfinite(r->ndc[2]);
    if (!finite) return;
    if (conv->zclip && !(conv->zmin <= r->ndc[2] && r->ndc[2] <= conv->zmax)) return;   // euc z-clip, inclusive
    float off = conv->sample_half ? 0.5f : 0.0f;
    int x0, x1, y0, y1;
    if (!covered_interval(r->cx, r->hx, off, (int)cam->w, &x0, &x1)) return;
    if (!covered_interval(r->cy, r->hy, off, (int)cam->h, &y0, &y1)) return;
    r->px0 = x0; r->px1 = x1; r->py0 = y0; r->py1 = y1;
    r->visible = 1;
}

// Pipeline::fragment src/pipelines.rs:127-145 ; returns alpha (0 => fragment is (0,0,0,0))
inline float fragment_alpha(float a, float b, float c, float opacity, float x, float y) {
    float power = -0.5f * (a * x * x + c * y * y) - b * x * y;
    if (power > 0.0f) return 0.0f;
    float alpha = rmin(0.99f, opacity * expf(power));
    if (alpha < 1.0f / 255.0f) return 0.0f;
    return alpha;
}
// Pipeline::blend src/pipelines.rs:147-168
inline uint32_t blend_px(uint32_t old_px, float r, float g, float b, float alpha) {
    float o_r = (float)((old_px >> 16) & 0xff) / 255.0f;
    float o_g = (float)((old_px >> 8) & 0xff) / 255.0f;
    float o_b = (float)(old_px & 0xff) / 255.0f;
    float ia = 1.0f - alpha;
    float br = ia * o_r + alpha * r;
    float bg = ia * o_g + alpha * g;
    float bb = ia * o_b + alpha * b;
    uint32_t R = as_u8(br * 255.0f), G = as_u8(bg * 255.0f), B = as_u8(bb * 255.0f), A = as_u8(alpha * 255.0f);
    return B | (G << 8) | (R << 16) | (A << 24);   // u32::from_le_bytes([b,g,r,a])
}

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

void orc_default_conventions(orc_conventions* c) {
    c->y_up = 1; c->sample_half = 1; c->zclip = 1; c->zmin = 0.0f; c->zmax = 1.0f; c->raster = 0;
    c->corrected_projection = 0;
}

// src/camera.rs:22-39 (new) + :41-68 (compute_matrices) + :84-89 (htanfovxy_focal)
void orc_camera_make(float h, float w, const float pos[3], float yaw, float pitch, float lowpass,
                     int32_t sh_dim, orc_camera* out) {
    const float znear = 0.01f, zfar = 100.0f, fovy = 3.14159265358979323846f / 2.0f;   // f32 PI / 2.0
    V3 position = ld3(pos), target = {0, 0, 0}, up = {0, -1, 0};
    V3 viewdir = normalize3(position - target);
    float cos_angle = (viewdir.x * up.x + viewdir.y * up.y) + viewdir.z * up.z;
    float sg = std::isnan(pitch) ? pitch : (std::signbit(pitch) ? -1.0f : 1.0f);     // f32::signum
    if (cos_angle * sg > 0.99f) pitch = 0.0f;
    // glm::rotation(angle, axis) = Rotation3::from_axis_angle(Unit::new_normalize(axis), angle).to_homogeneous()
    auto rotation = [](float angle, V3 axis, float R[9] /*row-major*/) {
        V3 u = normalize3(axis);
        if (angle == 0.0f) { float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; std::memcpy(R, I, sizeof I); return; }
        float sqx = u.x * u.x, sqy = u.y * u.y, sqz = u.z * u.z;
        float s = sinf(angle), c = cosf(angle), omc = 1.0f - c;
        R[0] = sqx + (1.0f - sqx) * c;       R[1] = u.x * u.y * omc - u.z * s;  R[2] = u.x * u.z * omc + u.y * s;
        R[3] = u.x * u.y * omc + u.z * s;    R[4] = sqy + (1.0f - sqy) * c;     R[5] = u.y * u.z * omc - u.x * s;
        R[6] = u.x * u.z * omc - u.y * s;    R[7] = u.y * u.z * omc + u.x * s;  R[8] = sqz + (1.0f - sqz) * c;
    };
    auto apply = [](const float R[9], V3 v) {
        return V3{(R[0] * v.x + R[1] * v.y) + R[2] * v.z, (R[3] * v.x + R[4] * v.y) + R[5] * v.z,
                  (R[6] * v.x + R[7] * v.y) + R[8] * v.z};
    };
    float Rx[9], Ry[9];
    rotation(yaw, up, Rx);
    V3 p1 = apply(Rx, position - target) + target;
    V3 right = cross3(up, position);                            // uses the FIELD, :58
    rotation(pitch, right, Ry);
    V3 eye = apply(Ry, p1 - target) + target;
    // glm::look_at == right-handed look-at (direct construction; nalgebra goes through a
    // quaternion round trip that can differ by an ulp -- the matrices are INPUTS to the hot path)
    V3 zaxis = normalize3(eye - target);
    V3 xaxis = normalize3(cross3(up, zaxis));
    V3 yaxis = normalize3(cross3(zaxis, xaxis));
    M4 V; std::memset(V.m, 0, sizeof V.m);
    V.at(0, 0) = xaxis.x; V.at(0, 1) = xaxis.y; V.at(0, 2) = xaxis.z;
    V.at(1, 0) = yaxis.x; V.at(1, 1) = yaxis.y; V.at(1, 2) = yaxis.z;
    V.at(2, 0) = zaxis.x; V.at(2, 1) = zaxis.y; V.at(2, 2) = zaxis.z;
    V.at(0, 3) = -((xaxis.x * eye.x + xaxis.y * eye.y) + xaxis.z * eye.z);
    V.at(1, 3) = -((yaxis.x * eye.x + yaxis.y * eye.y) + yaxis.z * eye.z);
    V.at(2, 3) = -((zaxis.x * eye.x + zaxis.y * eye.y) + zaxis.z * eye.z);
    V.at(3, 3) = 1.0f;
    // glm::perspective(aspect, fovy, near, far) == nalgebra Perspective3::new
    M4 P; std::memset(P.m, 0, sizeof P.m);
    float aspect = w / h;
    P.at(1, 1) = 1.0f / tanf(fovy / 2.0f);
    P.at(0, 0) = P.at(1, 1) / aspect;
    P.at(2, 2) = (zfar + znear) / (znear - zfar);
    P.at(2, 3) = zfar * znear * 2.0f / (znear - zfar);
    P.at(3, 2) = -1.0f;
    std::memcpy(out->view, V.m, sizeof V.m);
    std::memcpy(out->proj, P.m, sizeof P.m);
    out->w = w; out->h = h;
    float htany = tanf(fovy / 2.0f);
    out->htany = htany;
    out->htanx = htany / h * w;
    out->focal = h / (2.0f * htany);
    out->cam_pos[0] = pos[0]; out->cam_pos[1] = pos[1]; out->cam_pos[2] = pos[2];
    out->lowpass = lowpass; out->sh_dim = sh_dim;
}

void orc_compute_cov3d(uint64_t n, const float* scales3, const float* rot4, float* cov3d_out) {
    for (uint64_t i = 0; i < n; ++i) cov3d_one(scales3 + 3 * i, rot4 + 4 * i, cov3d_out + 9 * i);
}

void orc_eval_sh(const float* sh48, int32_t sh_dim, const float dir[3], float out[3]) {
    V3 c = eval_sh(sh48, sh_dim, ld3(dir));
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

void orc_project_cov2d(const float pos[3], const float cov3d[9], const orc_camera* cam, float out[4]) {
    project_cov2d(pos, cov3d, cam, out, nullptr);
}
void orc_project_cov2d_corrected(const float pos[3], const float cov3d[9], const orc_camera* cam, float out[4]) {
    project_cov2d(pos, cov3d, cam, out, nullptr, 1);
}

// src/gaussians.rs:297-306: z = (view * positions)[2]; indices.sort_by(partial_cmp, NaN => Equal) -- stable
void orc_sort(uint64_t n, const float* pos4, const float view[16], uint32_t* order_out) {
    std::vector<float> z(n);
    for (uint64_t i = 0; i < n; ++i) { float pc[4]; mul4(view, pos4 + 4 * i, pc); z[i] = pc[2]; }
    std::iota(order_out, order_out + n, 0u);
    std::stable_sort(order_out, order_out + n, [&](uint32_t a, uint32_t b) { return z[a] < z[b]; });
}

void orc_preprocess(uint64_t n, const float* pos4, const float* cov3d, const float* opacity, const float* sh48,
                    const orc_camera* cam, const orc_conventions* conv, orc_record* out) {
    for (uint64_t i = 0; i < n; ++i) {
        int sing = 0;
        preprocess_one(pos4 + 4 * i, cov3d + 9 * i, opacity[i], sh48 + 48 * i, cam, conv, out + i, &sing);
    }
}

void orc_fragment(const float v[9], float out[4]) {
    float alpha = fragment_alpha(v[4], v[5], v[6], v[3], v[7], v[8]);
    if (alpha == 0.0f) { out[0] = out[1] = out[2] = out[3] = 0.0f; return; }
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = alpha;
}

uint32_t orc_blend(uint32_t old_pixel, const float frag[4]) {
    return blend_px(old_pixel, frag[0], frag[1], frag[2], frag[3]);
}

int orc_render(uint64_t n, const float* pos4, const float* cov3d, const float* opacity, const float* sh48,
               const orc_camera* cam, const orc_conventions* conv, uint32_t* argb, int32_t row0, int32_t row1,
               int32_t nthreads, orc_stats* stats) {
    const int W = (int)cam->w, H = (int)cam->h;
    if (row0 < 0) row0 = 0;
    if (row1 > H || row1 < 0) row1 = H;
    if (nthreads < 1) nthreads = 1;
    orc_stats st; std::memset(&st, 0, sizeof st);
    double t0 = now_ms();
    // vertex stage (once per Gaussian; the reference runs it 6x with identical results)
    std::vector<orc_record> rec(n);
    {
        std::vector<std::thread> th;
        std::vector<uint64_t> sing(nthreads, 0);
        for (int t = 0; t < nthreads; ++t)
            th.emplace_back([&, t] {
                uint64_t a = n * t / nthreads, b = n * (t + 1) / nthreads;
                for (uint64_t i = a; i < b; ++i) {
                    int s = 0;
                    preprocess_one(pos4 + 4 * i, cov3d + 9 * i, opacity[i], sh48 + 48 * i, cam, conv, &rec[i], &s);
                    sing[t] += s;
                }
            });
        for (auto& x : th) x.join();
        for (auto s : sing) st.n_singular += s;
    }
    double t1 = now_ms();
    // painter's order: stable ascending view z (far first), src/gaussians.rs:302-303
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return rec[a].depth < rec[b].depth; });
    double t2 = now_ms();
    for (uint64_t i = 0; i < n; ++i)
        if (rec[i].visible) {
            ++st.n_visible;
            st.n_tile_pairs += (uint64_t)(rec[i].px1 / 16 - rec[i].px0 / 16 + 1) * (uint64_t)(rec[i].py1 / 16 - rec[i].py0 / 16 + 1);
        }
    const float off = conv->sample_half ? 0.5f : 0.0f;
    std::vector<uint64_t> frags(nthreads, 0);
    // raster == 1: generic two-triangle rasterisation of the quad (sensitivity variant, see header)
    auto raster_triangles = [&](const orc_record& r, int a, int b, uint64_t& nf) {
        // quad corners in NDC as gaussian_vertex_shader builds them, then euc's NDC -> pixel map
        const float bndx = r.hx / cam->w * 2.0f, bndy = r.hy / cam->h * 2.0f;
        const float corner[4][2] = {{-1, 1}, {-1, -1}, {1, -1}, {1, 1}};
        float vx[4], vy[4], ux[4], uy[4];
        for (int k = 0; k < 4; ++k) {
            float nx = corner[k][0] * bndx + r.ndc[0], ny = corner[k][1] * bndy + r.ndc[1];
            vx[k] = (nx * 0.5f + 0.5f) * cam->w;
            vy[k] = conv->y_up ? (ny * -0.5f + 0.5f) * cam->h : (ny * 0.5f + 0.5f) * cam->h;
            ux[k] = corner[k][0] * r.hx; uy[k] = corner[k][1] * r.hy;      // coordxy at the corner
        }
        const int tri[2][3] = {{0, 1, 2}, {0, 2, 3}};
        for (int t = 0; t < 2; ++t) {
            const int i0 = tri[t][0], i1 = tri[t][1], i2 = tri[t][2];
            float area = (vx[i1] - vx[i0]) * (vy[i2] - vy[i0]) - (vy[i1] - vy[i0]) * (vx[i2] - vx[i0]);
            if (area == 0.0f || !std::isfinite(area)) continue;
            float xmin = std::min(vx[i0], std::min(vx[i1], vx[i2])), xmax = std::max(vx[i0], std::max(vx[i1], vx[i2]));
            float ymin = std::min(vy[i0], std::min(vy[i1], vy[i2])), ymax = std::max(vy[i0], std::max(vy[i1], vy[i2]));
            int x0 = (int)std::max(0.0f, std::floor(xmin) - 1.0f), x1 = (int)std::min((float)(W - 1), std::ceil(xmax) + 1.0f);
            int y0 = (int)std::max((float)a, std::floor(ymin) - 1.0f), y1 = (int)std::min((float)(b - 1), std::ceil(ymax) + 1.0f);
            for (int y = y0; y <= y1; ++y) {
                uint32_t* row = argb + (size_t)y * W;
                for (int x = x0; x <= x1; ++x) {
                    float px = (float)x + off, py = (float)y + off;
                    float w0 = ((vx[i1] - px) * (vy[i2] - py) - (vy[i1] - py) * (vx[i2] - px)) / area;
                    float w1 = ((vx[i2] - px) * (vy[i0] - py) - (vy[i2] - py) * (vx[i0] - px)) / area;
                    float w2 = 1.0f - w0 - w1;
                    if (!(w0 >= 0.0f && w1 >= 0.0f && w2 >= 0.0f)) continue;
                    float cxy = w0 * ux[i0] + w1 * ux[i1] + w2 * ux[i2];
                    float cyy = w0 * uy[i0] + w1 * uy[i1] + w2 * uy[i2];
                    float alpha = fragment_alpha(r.conic[0], r.conic[1], r.conic[2], r.opacity, cxy, cyy);
                    if (alpha == 0.0f) row[x] = blend_px(row[x], 0.0f, 0.0f, 0.0f, 0.0f);
                    else row[x] = blend_px(row[x], r.rgb[0], r.rgb[1], r.rgb[2], alpha);
                    ++nf;
                }
            }
        }
    };
    auto band = [&](int t) {
        int rows = row1 - row0;
        int a = row0 + (int)((int64_t)rows * t / nthreads), b = row0 + (int)((int64_t)rows * (t + 1) / nthreads);
        uint64_t nf = 0;
        for (uint64_t k = 0; k < n; ++k) {
            const orc_record& r = rec[order[k]];
            if (!r.visible) continue;
            if (conv->raster == 1) { raster_triangles(r, a, b, nf); continue; }
            int y0 = std::max(r.py0, a), y1 = std::min(r.py1, b - 1);
            for (int y = y0; y <= y1; ++y) {
                float sy = (float)y + off;
                float dy = conv->y_up ? (r.cy - sy) : (sy - r.cy);   // coordxy.y grows with NDC y
                uint32_t* row = argb + (size_t)y * W;
                for (int x = r.px0; x <= r.px1; ++x) {
                    float dx = ((float)x + off) - r.cx;
                    float alpha = fragment_alpha(r.conic[0], r.conic[1], r.conic[2], r.opacity, dx, dy);
                    // rejected fragments are (0,0,0,0) and are STILL blended: RGB unchanged, A := 0
                    if (alpha == 0.0f) row[x] = blend_px(row[x], 0.0f, 0.0f, 0.0f, 0.0f);
                    else row[x] = blend_px(row[x], r.rgb[0], r.rgb[1], r.rgb[2], alpha);
                    ++nf;
                }
            }
        }
        frags[t] = nf;
    };
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(band, t);
        for (auto& x : th) x.join();
    }
    double t3 = now_ms();
    for (auto f : frags) st.n_fragments += f;
    st.ms_preprocess = t1 - t0; st.ms_sort = t2 - t1; st.ms_raster = t3 - t2;
    if (stats) *stats = st;
    return 0;
}

// ---------------------------------------------------------------------------
// PLY loader: src/gaussians.rs:375-405 + set_property :258-282.  Only
// `float`-typed scalar properties are consumed (Property::Float); everything
// else is parsed and ignored.  A non-"vertex" element is an error (panic :390).
// ---------------------------------------------------------------------------
namespace {
struct Prop { std::string name; int type; bool is_list; int count_type; };   // type: size code below
int type_code(const std::string& t) {
    if (t == "char" || t == "int8") return 1;
    if (t == "uchar" || t == "uint8") return 2;
    if (t == "short" || t == "int16") return 3;
    if (t == "ushort" || t == "uint16") return 4;
    if (t == "int" || t == "int32") return 5;
    if (t == "uint" || t == "uint32") return 6;
    if (t == "float" || t == "float32") return 7;
    if (t == "double" || t == "float64") return 8;
    return 0;
}
int type_size(int c) { static const int s[] = {0, 1, 1, 2, 2, 4, 4, 4, 8}; return s[c]; }
}  // namespace

int64_t orc_load_ply(const char* path, float* pos4, float* scales3, float* opacity, float* rot4, float* sh48) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return -1;
    char line[1024];
    int fmt = -1;   // 0 ascii, 1 binary LE, 2 binary BE
    int64_t nvert = -1;
    std::vector<Prop> props;
    bool in_vertex = false, ok = false;
    if (!std::fgets(line, sizeof line, f) || std::strncmp(line, "ply", 3) != 0) { std::fclose(f); return -2; }
    while (std::fgets(line, sizeof line, f)) {
        char a[256], b[256], c[256], d[256];
        int k = std::sscanf(line, "%255s %255s %255s %255s", a, b, c, d);
        if (k < 1) continue;
        std::string kw = a;
        if (kw == "format" && k >= 2) {
            std::string s = b;
            fmt = s == "ascii" ? 0 : s == "binary_little_endian" ? 1 : s == "binary_big_endian" ? 2 : -1;
        } else if (kw == "element" && k >= 3) {
            if (std::string(b) != "vertex") { std::fclose(f); return -3; }   // panic!("Unexpected element!")
            nvert = std::atoll(c); in_vertex = true;
        } else if (kw == "property" && in_vertex) {
            Prop p;
            if (std::string(b) == "list" && k >= 4) {
                char e[256]; std::sscanf(line, "%*s %*s %255s %255s %255s", b, c, e);
                p.is_list = true; p.count_type = type_code(b); p.type = type_code(c); p.name = e;
            } else { p.is_list = false; p.count_type = 0; p.type = type_code(b); p.name = c; }
            if (!p.type) { std::fclose(f); return -4; }
            props.push_back(p);
        } else if (kw == "end_header") { ok = true; break; }
    }
    if (!ok || fmt < 0 || nvert < 0) { std::fclose(f); return -5; }
    if (!pos4) { std::fclose(f); return nvert; }

    // property name -> destination (set_property match arms)
    struct Dst { int arr; int idx; };   // arr: 0 none,1 pos,2 scale,3 opacity,4 rot,5 sh
    std::vector<Dst> dst(props.size(), Dst{0, 0});
    for (size_t i = 0; i < props.size(); ++i) {
        const std::string& s = props[i].name;
        if (props[i].is_list || props[i].type != 7) continue;   // only Property::Float arms exist
        if (s == "x") dst[i] = {1, 0}; else if (s == "y") dst[i] = {1, 1}; else if (s == "z") dst[i] = {1, 2};
        else if (s == "scale_0") dst[i] = {2, 0}; else if (s == "scale_1") dst[i] = {2, 1}; else if (s == "scale_2") dst[i] = {2, 2};
        else if (s == "opacity") dst[i] = {3, 0};
        else if (s == "rot_0") dst[i] = {4, 3}; else if (s == "rot_1") dst[i] = {4, 0};
        else if (s == "rot_2") dst[i] = {4, 1}; else if (s == "rot_3") dst[i] = {4, 2};
        else if (s == "f_dc_0") dst[i] = {5, 0}; else if (s == "f_dc_1") dst[i] = {5, 1}; else if (s == "f_dc_2") dst[i] = {5, 2};
        else if (s.rfind("f_rest_", 0) == 0) {
            int idx = std::atoi(s.c_str() + 7);
            if (idx < 0 || idx > 44) { std::fclose(f); return -6; }   // sh[3+index] out of bounds would panic
            dst[i] = {5, 3 + idx};
        }
    }
    auto read_scalar = [&](int type, double* out) -> bool {
        if (fmt == 0) { return std::fscanf(f, "%lf", out) == 1; }
        unsigned char buf[8]; int sz = type_size(type);
        if (std::fread(buf, 1, sz, f) != (size_t)sz) return false;
        if (fmt == 2) std::reverse(buf, buf + sz);
        switch (type) {
            case 1: *out = (int8_t)buf[0]; break;
            case 2: *out = buf[0]; break;
            case 3: { int16_t v; std::memcpy(&v, buf, 2); *out = v; break; }
            case 4: { uint16_t v; std::memcpy(&v, buf, 2); *out = v; break; }
            case 5: { int32_t v; std::memcpy(&v, buf, 4); *out = v; break; }
            case 6: { uint32_t v; std::memcpy(&v, buf, 4); *out = v; break; }
            case 7: { float v; std::memcpy(&v, buf, 4); *out = v; break; }
            case 8: { double v; std::memcpy(&v, buf, 8); *out = v; break; }
        }
        return true;
    };
    for (int64_t i = 0; i < nvert; ++i) {
        // PropertyAccess::new :247-256
        float* P = pos4 + 4 * i; P[0] = P[1] = P[2] = 0.0f; P[3] = 1.0f;
        float* S = scales3 + 3 * i; S[0] = S[1] = S[2] = 0.0f;
        opacity[i] = 0.0f;
        float* Q = rot4 + 4 * i; Q[0] = Q[1] = Q[2] = 0.0f; Q[3] = 1.0f;   // Quaternion::identity()
        float* SH = sh48 + 48 * i; std::memset(SH, 0, 48 * sizeof(float));
        for (size_t k = 0; k < props.size(); ++k) {
            if (props[k].is_list) {
                double cnt; if (!read_scalar(props[k].count_type, &cnt)) { std::fclose(f); return -7; }
                for (int j = 0; j < (int)cnt; ++j) { double v; if (!read_scalar(props[k].type, &v)) { std::fclose(f); return -7; } }
                continue;
            }
            double dv; if (!read_scalar(props[k].type, &dv)) { std::fclose(f); return -7; }
            float v = (float)dv;
            switch (dst[k].arr) {
                case 1: P[dst[k].idx] = v; break;
                case 2: S[dst[k].idx] = expf(v); break;                          // :264-266
                case 3: opacity[i] = 1.0f / (1.0f + expf(-v)); break;            // :267
                case 4: Q[dst[k].idx] = v; break;                                // :268-271
                case 5: SH[dst[k].idx] = v; break;                               // :272-279 (no transpose)
                default: break;
            }
        }
    }
    std::fclose(f);
    // recentre: sequential f32 sum, :394-402
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    for (int64_t i = 0; i < nvert; ++i) { ax += pos4[4 * i]; ay += pos4[4 * i + 1]; az += pos4[4 * i + 2]; }
    float nf = (float)nvert;
    ax /= nf; ay /= nf; az /= nf;
    for (int64_t i = 0; i < nvert; ++i) { pos4[4 * i] -= ax; pos4[4 * i + 1] -= ay; pos4[4 * i + 2] -= az; }
    return nvert;
}

}  // extern "C"

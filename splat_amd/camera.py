"""Host-side mirror of the reference's `Camera` (src/camera.rs): same fields, same method names,
same f32 arithmetic order.  It only PRODUCES the per-frame constants the hot path consumes
(view, projection, w, h, htan/focal, position) -- SURVEY.md section 8 row a13."""
import numpy as np

from . import _lib

f32 = np.float32


def _normalize(v):
    n = np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2], dtype=f32)
    return (v / n).astype(f32)


def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], f32)


def _dot(a, b):
    return f32((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2])


def _rotation(angle, axis):
    """glm::rotation(angle, &axis) -> 3x3 block (Rotation3::from_axis_angle of the normalised axis)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        u = _normalize(axis)
    if angle == 0.0:
        return np.eye(3, dtype=f32)
    s, c = f32(np.sin(f32(angle))), f32(np.cos(f32(angle)))
    omc = f32(1.0) - c
    sq = u * u
    one = f32(1.0)
    return np.array([
        [sq[0] + (one - sq[0]) * c, u[0] * u[1] * omc - u[2] * s, u[0] * u[2] * omc + u[1] * s],
        [u[0] * u[1] * omc + u[2] * s, sq[1] + (one - sq[1]) * c, u[1] * u[2] * omc - u[0] * s],
        [u[0] * u[2] * omc - u[1] * s, u[1] * u[2] * omc + u[0] * s, sq[2] + (one - sq[2]) * c]], f32)


def _apply(R, v):
    return np.array([(R[i, 0] * v[0] + R[i, 1] * v[1]) + R[i, 2] * v[2] for i in range(3)], f32)


class Camera:
    """src/camera.rs:4-19.  `Camera(h, w, start_position=None)` -- height first (:22)."""

    def __init__(self, h, w, start_position=None):
        self.znear = f32(0.01)
        self.zfar = f32(100.0)
        self.h = f32(h)
        self.w = f32(w)
        self.fovy = f32(np.pi) / f32(2.0)
        self.position = np.array(start_position if start_position is not None else (0.0, 0.0, 3.0), f32)
        self.target = np.zeros(3, f32)
        self.up = np.array([0.0, -1.0, 0.0], f32)
        self.yaw = f32(0.0)
        self.pitch = f32(0.0)
        self.is_pose_dirty = True
        self.is_intrin_dirty = True
        self.view_matrix = np.eye(4, dtype=f32)        # identity until compute_matrices (:36-37)
        self.projection_matrix = np.eye(4, dtype=f32)

    def compute_matrices(self):
        """src/camera.rs:41-68"""
        viewdir = _normalize(self.position - self.target)
        cos_angle = _dot(viewdir, self.up)
        sg = f32(-1.0) if np.signbit(self.pitch) else f32(1.0)
        if cos_angle * sg > f32(0.99):
            self.pitch = f32(0.0)
        rx = _rotation(self.yaw, self.up)
        position = _apply(rx, self.position - self.target) + self.target
        right = _cross(self.up, self.position)            # the field, not the rotated position (:58)
        ry = _rotation(self.pitch, right)
        eye = _apply(ry, position - self.target) + self.target
        # glm::look_at (right-handed)
        z = _normalize(eye - self.target)
        x = _normalize(_cross(self.up, z))
        y = _normalize(_cross(z, x))
        V = np.zeros((4, 4), f32)
        V[0, :3], V[1, :3], V[2, :3] = x, y, z
        V[0, 3], V[1, 3], V[2, 3] = -_dot(x, eye), -_dot(y, eye), -_dot(z, eye)
        V[3, 3] = 1.0
        self.view_matrix = V
        # glm::perspective(aspect, fovy, near, far) == nalgebra Perspective3::new
        P = np.zeros((4, 4), f32)
        aspect = self.w / self.h
        P[1, 1] = f32(1.0) / f32(np.tan(self.fovy / f32(2.0)))
        P[0, 0] = P[1, 1] / aspect
        P[2, 2] = (self.zfar + self.znear) / (self.znear - self.zfar)
        P[2, 3] = self.zfar * self.znear * f32(2.0) / (self.znear - self.zfar)
        P[3, 2] = -1.0
        self.projection_matrix = P

    def get_view_matrix(self):
        return self.view_matrix

    def get_project_matrix(self):
        return self.projection_matrix

    def update_resolution(self, height, width):
        self.h, self.w = f32(height), f32(width)
        self.is_intrin_dirty = True

    def get_htanfovxy_focal(self):
        htany = f32(np.tan(self.fovy / f32(2.0)))
        htanx = htany / self.h * self.w
        focal = self.h / (f32(2.0) * htany)
        return np.array([htanx, htany, focal], f32)

    def get_focal(self):
        return self.h / (f32(2.0) * f32(np.tan(self.fovy / f32(2.0))))

    def update_pitch_angle(self, delta):
        self.pitch = f32(self.pitch + f32(delta))
        self.is_pose_dirty = True

    def update_yaw_angle(self, delta):
        self.yaw = f32(self.yaw + f32(delta))
        self.is_pose_dirty = True

    def update_camera_pose(self):
        self.compute_matrices()
        self.is_pose_dirty = False

    # ---- boundary: the struct the C ABI takes -------------------------------------------
    def to_c(self, lowpass, sh_dim=15):
        c = _lib.CameraC()
        c.view[:] = np.asarray(self.view_matrix, f32).T.reshape(-1).tolist()     # column-major
        c.proj[:] = np.asarray(self.projection_matrix, f32).T.reshape(-1).tolist()
        c.w, c.h = float(self.w), float(self.h)
        ht = self.get_htanfovxy_focal()
        c.htanx, c.htany, c.focal = float(ht[0]), float(ht[1]), float(ht[2])
        c.cam_pos[:] = [float(v) for v in self.position]
        c.lowpass = float(lowpass)
        c.sh_dim = int(sh_dim)
        return c

"""Host-side mirror of src/pipelines.rs: `GaussianSplatPipeline01` / `GaussianSplatPipeline02` with
public `gaussians`, `camera` and `render_to_buffer(color)`, whose body is one call through the C ABI
instead of sort + euc (src/pipelines.rs:66-86, 260-280)."""
import numpy as np

from .renderer import Renderer


class _Pipeline:
    LOWPASS = None
    SH_DIM = 15          # the literal at src/pipelines.rs:100 and :189

    def __init__(self, gaussians, camera, renderer=None, **renderer_kw):
        self.gaussians = gaussians
        self.camera = camera
        self._renderer = renderer or Renderer(**renderer_kw)
        self._uploaded = None

    @property
    def renderer(self):
        return self._renderer

    def _ensure_uploaded(self):
        if self._uploaded is not self.gaussians:      # lazily, once per scene object
            self._renderer.upload(self.gaussians)
            self._uploaded = self.gaussians

    def invalidate_scene(self):
        """The GPU copy of `gaussians` is cached by object identity (the reference re-reads the field on
        every render_to_buffer, src/pipelines.rs:67-79): call this after mutating the scene in place."""
        self._uploaded = None

    def camera_constants(self):
        return self.camera.to_c(self.LOWPASS, self.SH_DIM)

    def render_to_buffer(self, color):
        """Blend the scene onto `color` (uint32 [h,w], 0xAARRGGBB, modified in place) exactly as the
        reference does; the camera matrices must already be computed (update_camera_pose), else they
        are identity as in the reference (src/camera.rs:36-37).  Returns the frame's stats."""
        self._ensure_uploaded()
        return self._renderer.render(self.camera_constants(), color)


class GaussianSplatPipeline01(_Pipeline):
    """AoS pipeline: low-pass 0.01 (src/gaussians.rs:156-157); cov3d is whatever each Gaussian
    carries -- zero unless the caller ran compute_cov3d (src/main.rs:24-26)."""
    LOWPASS = np.float32(0.01)


class GaussianSplatPipeline02(_Pipeline):
    """SoA pipeline: low-pass 0.3 (src/gaussians.rs:517-518); GaussianList::from_vec computes
    cov3d on construction (src/gaussians.rs:439)."""
    LOWPASS = np.float32(0.3)

    def __init__(self, gaussians, camera, renderer=None, **renderer_kw):
        super().__init__(gaussians, camera, renderer, **renderer_kw)
        self.gaussians.compute_cov3d(self._renderer)

//! render_to_buffer over libsplat_hip.so: the two pipeline structs of src/pipelines.rs:54-57 and :172-175 with the
//! same pub fields and the same `render_to_buffer(&self, &mut euc::Buffer<u32, 2>)`, whose body (sort + euc render,
//! src/pipelines.rs:66-86 and :260-280) becomes one call through the C ABI.  UNTESTED here (no rustc in the
//! authoring image).
//!
//! Needs four one-line accessors on GaussianList (its matrices are private, src/gaussians.rs:408-416):
//!     pub fn positions_slice(&self) -> &[f32] { self.positions.as_slice() }   // 4 x N, column = (x, y, z, 1)
//!     pub fn cov3d_slice(&self)     -> &[f32] { self.cov3d.as_slice() }       // 3 x 3N, 3 x 3 column-major blocks
//!     pub fn opacities_slice(&self) -> &[f32] { self.opacities.as_slice() }   // N
//!     pub fn sh_slice(&self)        -> &[f32] { self.sh.as_slice() }          // 48 x N
//!     pub fn len(&self) -> usize
use std::cell::OnceCell;
use std::ffi::CStr;

use crate::camera::Camera;
use crate::ffi;
use crate::gaussians::{Gaussian, GaussianList};

/// What Camera's getters return (src/camera.rs:70-93), in the ABI's layout.
fn camera_constants(camera: &Camera, lowpass: f32) -> ffi::SplatCamera {
    let h = camera.get_htanfovxy_focal();
    let mut c: ffi::SplatCamera = unsafe { std::mem::zeroed() };
    c.view.copy_from_slice(camera.get_view_matrix().as_slice()); // nalgebra storage is column-major
    c.proj.copy_from_slice(camera.get_project_matrix().as_slice());
    c.w = camera.w;
    c.h = camera.h;
    c.htanx = h[0];
    c.htany = h[1];
    c.focal = h[2];
    c.cam_pos = [camera.position.x, camera.position.y, camera.position.z]; // the FIELD (src/pipelines.rs:99)
    c.lowpass = lowpass; // 0.01 in Pipeline01 (src/gaussians.rs:156), 0.3 in Pipeline02 (:517)
    c.sh_dim = 15; // the literal at src/pipelines.rs:100 / :189
    c
}

struct Gpu(*mut ffi::SplatCtx);
impl Drop for Gpu {
    fn drop(&mut self) {
        unsafe { ffi::splat_destroy(self.0) }
    }
}

fn check(ctx: *mut ffi::SplatCtx, rc: i32, what: &str) {
    if rc != ffi::SPLAT_OK {
        // the reference's convention for this path is panic (`unwrap`, src/pipelines.rs:22)
        let msg = unsafe { CStr::from_ptr(ffi::splat_last_error(ctx)) };
        panic!("{}: {}", what, msg.to_string_lossy());
    }
}

fn create_and_upload(pos4: &[f32], cov3d: &[f32], opacity: &[f32], sh: &[f32]) -> Gpu {
    unsafe {
        // a libsplat_hip.so built from another header would write a splat_stats of another size into ours
        assert!(ffi::splat_abi_version() == ffi::SPLAT_ABI_VERSION && ffi::splat_stats_size() as usize == std::mem::size_of::<ffi::SplatStats>(),
                "libsplat_hip.so speaks ABI version {}, this crate {}", ffi::splat_abi_version(), ffi::SPLAT_ABI_VERSION);
        let mut cfg: ffi::SplatConfig = std::mem::zeroed();
        ffi::splat_default_config(&mut cfg);
        let mut ctx = std::ptr::null_mut();
        check(std::ptr::null_mut(), ffi::splat_create(&cfg, &mut ctx), "splat_create");
        let n = opacity.len() as u64;
        check(ctx, ffi::splat_upload_scene(ctx, n, pos4.as_ptr(), cov3d.as_ptr(), opacity.as_ptr(), sh.as_ptr()), "splat_upload_scene");
        Gpu(ctx)
    }
}

fn render(gpu: &Gpu, cam: &ffi::SplatCamera, color: &mut euc::Buffer<u32, 2>) {
    let rc = unsafe { ffi::splat_render(gpu.0, cam, color.raw_mut().as_mut_ptr(), std::ptr::null_mut()) };
    check(gpu.0, rc, "splat_render");
}

/// The viewer loop's pair `color.clear(0); pipeline.render_to_buffer(&mut color)` (src/main.rs:73-74) as one call: `color` is
/// written, never read -- no upload of the cleared image, the clear fused into the compositor.  Pin `color`'s storage once
/// (`ffi::splat_host_register(color.raw_mut().as_mut_ptr() as *mut _, 4 * w * h)` next to its creation, src/main.rs:62;
/// unregister before it is dropped) and the compositor writes the pixels straight into it.
fn render_frame(gpu: &Gpu, cam: &ffi::SplatCamera, color: &mut euc::Buffer<u32, 2>) {
    let rc = unsafe { ffi::splat_render_frame(gpu.0, cam, color.raw_mut().as_mut_ptr(), std::ptr::null_mut()) };
    check(gpu.0, rc, "splat_render_frame");
}

/// GaussianSplatPipeline02 (src/pipelines.rs:172-175): the SoA scene IS the ABI's layout.
pub struct GaussianSplatPipeline02Hip {
    pub gaussians: GaussianList,
    pub camera: Camera,
    gpu: OnceCell<Gpu>, // created and uploaded at the first frame
}

impl GaussianSplatPipeline02Hip {
    pub fn new(gaussians: GaussianList, camera: Camera) -> Self {
        Self { gaussians, camera, gpu: OnceCell::new() }
    }
    /// src/pipelines.rs:260-280
    pub fn render_to_buffer(&self, color: &mut euc::Buffer<u32, 2>) {
        let g = &self.gaussians;
        let gpu = self.gpu.get_or_init(|| create_and_upload(g.positions_slice(), g.cov3d_slice(), g.opacities_slice(), g.sh_slice()));
        render(gpu, &camera_constants(&self.camera, 0.3), color);
    }
    /// replaces the pair at src/main.rs:73-74 (clear + render_to_buffer); the result is the same image
    pub fn render_frame_to_buffer(&self, color: &mut euc::Buffer<u32, 2>) {
        let g = &self.gaussians;
        let gpu = self.gpu.get_or_init(|| create_and_upload(g.positions_slice(), g.cov3d_slice(), g.opacities_slice(), g.sh_slice()));
        render_frame(gpu, &camera_constants(&self.camera, 0.3), color);
    }
}

/// GaussianSplatPipeline01 (src/pipelines.rs:54-57): Vec<Gaussian> is AoS and not repr(C), so the four arrays are
/// gathered first -- GaussianList::from_vec WITHOUT its compute_cov3d call: Pipeline01 uses whatever each
/// Gaussian.cov3d holds (zero unless the caller ran compute_cov3d, src/main.rs:24-26).
pub struct GaussianSplatPipeline01Hip {
    pub gaussians: Vec<Gaussian>,
    pub camera: Camera,
    gpu: OnceCell<Gpu>,
}

impl GaussianSplatPipeline01Hip {
    pub fn new(gaussians: Vec<Gaussian>, camera: Camera) -> Self {
        Self { gaussians, camera, gpu: OnceCell::new() }
    }
    fn gpu(&self) -> &Gpu {
        self.gpu.get_or_init(|| {
            let n = self.gaussians.len();
            let (mut pos4, mut cov, mut op, mut sh) = (vec![0f32; 4 * n], vec![0f32; 9 * n], vec![0f32; n], vec![0f32; 48 * n]);
            for (i, g) in self.gaussians.iter().enumerate() {
                pos4[4 * i..4 * i + 3].copy_from_slice(g.position.as_slice());
                pos4[4 * i + 3] = 1.0;
                cov[9 * i..9 * i + 9].copy_from_slice(g.cov3d.as_slice()); // Matrix3 storage: column-major
                op[i] = g.opacity;
                sh[48 * i..48 * i + 48].copy_from_slice(g.sh.as_slice());
            }
            create_and_upload(&pos4, &cov, &op, &sh)
        })
    }
    /// src/pipelines.rs:66-86
    pub fn render_to_buffer(&self, color: &mut euc::Buffer<u32, 2>) {
        render(self.gpu(), &camera_constants(&self.camera, 0.01), color);
    }
    /// replaces the pair at src/main.rs:73-74 (clear + render_to_buffer); the result is the same image
    pub fn render_frame_to_buffer(&self, color: &mut euc::Buffer<u32, 2>) {
        render_frame(self.gpu(), &camera_constants(&self.camera, 0.01), color);
    }
}

//! `extern "C"` view of include/splat_hip.h -- field for field, in declaration order.
//! UNTESTED here (no rustc in the authoring image); tests/test_host.py checks the same header against the
//! ctypes binding, and examples/render_c.c compiles it as C99.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const SPLAT_OK: c_int = 0;
pub const SPLAT_ERR_INVALID: c_int = -1;
pub const SPLAT_ERR_HIP: c_int = -2;
pub const SPLAT_ERR_NO_SCENE: c_int = -3;
pub const SPLAT_ERR_CAPACITY: c_int = -4;
pub const SPLAT_MODE_EXACT: i32 = 0;
pub const SPLAT_MODE_CORRECTED_PROJECTION: i32 = 1;
pub const SPLAT_MODE_LIBM_EXP: i32 = 2;
pub const SPLAT_MODE_FAST: i32 = 4;
// tuning options (splat_set_option / splat_get_option): equivalent schedules and storage sizes, never pixels
pub const SPLAT_OPT_PIPELINE_DEPTH: i32 = 1;
pub const SPLAT_OPT_FUSED_SORT_MAX: i32 = 2;
pub const SPLAT_OPT_REGION_SPARE: i32 = 3;
pub const SPLAT_OPT_EARLY_OUT_EPS: i32 = 4;
pub const SPLAT_OPT_EARLY_OUT_MIN_LIST: i32 = 5;
pub const SPLAT_OPT_EARLY_OUT_SCAN_EIGHTHS: i32 = 6;
pub const SPLAT_OPT_SORT_IN_COMPOSITOR: i32 = 7;
pub const SPLAT_OPT_PAIR_WALK: i32 = 8;
pub const SPLAT_OPT_TIMING_EVERY: i32 = 9;
pub const SPLAT_OPT_BLOCK_CULLING: i32 = 10;
pub const SPLAT_OPT_ONE_PASS_BINNING: i32 = 11;
pub const SPLAT_OPT_KEY_BUFFER_BYTES: i32 = 12;
pub const SPLAT_OPT_FAST_CLOSE_WIDTH: i32 = 13;
pub const SPLAT_OPT_PRIORITY_LIST_LEN: i32 = 14;
pub const SPLAT_OPT_FRAME_OVERLAP: i32 = 15;
pub const SPLAT_OPT_NEAR_SELECT_KEYS: i32 = 16;
pub const SPLAT_OPT_OVERFLOW_REDO: i32 = 17;
pub const SPLAT_OPT_START_HINTS: i32 = 18;
pub const SPLAT_OPT_HOST_ZERO_COPY: i32 = 19;
pub const SPLAT_OPT_KEYS_PER_GAUSSIAN: i32 = 20;
pub const SPLAT_OPT_COUNT_FIRST: i32 = 21;
pub const SPLAT_OPT_LARGE_SPLAT_TILES: i32 = 22;
pub const SPLAT_OPT_LARGE_LIST_MIN: i32 = 23;
/// SPLAT_ABI_VERSION of the header this file mirrors; compared with splat_abi_version() before the first call
pub const SPLAT_ABI_VERSION: u32 = 6;

#[repr(C)] pub struct SplatCtx { _private: [u8; 0] }
#[repr(C)] pub struct SplatMulti { _private: [u8; 0] }

#[repr(C)] #[derive(Clone, Copy)]
pub struct SplatConfig {
    pub device: i32, pub mode: i32,
    pub y_up: i32, pub sample_half: i32, pub zclip: i32, pub zmin: f32, pub zmax: f32,
    pub pair_capacity: u64,
}
#[repr(C)] #[derive(Clone, Copy)]
pub struct SplatCamera {
    pub view: [f32; 16], pub proj: [f32; 16],      // nalgebra as_slice(): column-major
    pub w: f32, pub h: f32,
    pub htanx: f32, pub htany: f32, pub focal: f32,
    pub cam_pos: [f32; 3],
    pub lowpass: f32, pub sh_dim: i32,
}
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct SplatStats {
    pub n_gaussians: u64, pub n_visible: u64, pub n_singular: u64, pub n_pairs: u64,
    pub max_tile_len: u64, pub bytes_algorithmic: u64,
    pub ms_preprocess: f32, pub ms_scan: f32, pub ms_emit: f32, pub ms_sort: f32,
    pub ms_composite: f32, pub ms_total: f32,
    pub n_fallback: u64, pub n_sort_fallback: u64, pub n_iter_scan: u64, pub n_iter_blend: u64,
    pub n_blocks_culled: u64, pub flops_algorithmic: u64,
    pub n_near_tiles: u64, pub n_near_fallback: u64,
}

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct SplatRecord {
    pub cx: f32, pub cy: f32, pub hx: f32, pub hy: f32,
    pub conic_a: f32, pub conic_b: f32, pub conic_c: f32, pub opacity: f32,
    pub r: f32, pub g: f32, pub b: f32, pub depth: f32,
    pub px0: i32, pub px1: i32, pub py0: i32, pub py1: i32,
}

extern "C" {
    pub fn splat_abi_version() -> u32;
    pub fn splat_stats_size() -> u64;
    pub fn splat_default_config(cfg: *mut SplatConfig);
    pub fn splat_create(cfg: *const SplatConfig, out: *mut *mut SplatCtx) -> c_int;
    pub fn splat_destroy(ctx: *mut SplatCtx);
    pub fn splat_last_error(ctx: *const SplatCtx) -> *const c_char;
    pub fn splat_upload_scene(ctx: *mut SplatCtx, n: u64, pos4: *const f32, cov3d: *const f32,
                              opacity: *const f32, sh: *const f32) -> c_int;
    pub fn splat_compute_cov3d(ctx: *mut SplatCtx, n: u64, scales3: *const f32, rot4: *const f32,
                               cov3d_out: *mut f32) -> c_int;
    pub fn splat_set_slab(ctx: *mut SplatCtx, tile_row0: i32, tile_row1: i32) -> c_int;
    pub fn splat_tile_row_loads(ctx: *mut SplatCtx, cam: *const SplatCamera, row_pairs: *mut u64, n_rows: i32) -> c_int;
    pub fn splat_render(ctx: *mut SplatCtx, cam: *const SplatCamera, argb: *mut u32, stats: *mut SplatStats) -> c_int;
    // `color.clear(0); render_to_buffer(&mut color)` (src/main.rs:73-74) in one call: argb_out is written, never read
    pub fn splat_render_frame(ctx: *mut SplatCtx, cam: *const SplatCamera, argb_out: *mut u32, stats: *mut SplatStats) -> c_int;
    pub fn splat_render_device(ctx: *mut SplatCtx, cam: *const SplatCamera, d_argb: *mut c_void,
                               sync: i32, stats: *mut SplatStats) -> c_int;
    pub fn splat_render_frame_device(ctx: *mut SplatCtx, cam: *const SplatCamera, d_argb: *mut c_void,
                                     sync: i32, stats: *mut SplatStats) -> c_int;
    pub fn splat_sync(ctx: *mut SplatCtx) -> c_int;
    pub fn splat_frames_dropped(ctx: *const SplatCtx) -> u64;
    pub fn splat_device_bytes(ctx: *const SplatCtx, peak: *mut u64) -> u64;
    pub fn splat_binning_mode(ctx: *mut SplatCtx) -> i64;
    pub fn splat_set_frame_overlap(ctx: *mut SplatCtx, n: i32) -> c_int;   // 2: frames to different images composite side by side
    pub fn splat_set_option(ctx: *mut SplatCtx, option: i32, value: f64) -> c_int;   // SPLAT_OPT_*: what the SPLAT_* environment variables set, from code
    pub fn splat_get_option(ctx: *const SplatCtx, option: i32, value: *mut f64) -> c_int;
    pub fn splat_stream(ctx: *mut SplatCtx) -> *mut c_void;               // the hipStream_t the kernels run on
    pub fn splat_set_stream(ctx: *mut SplatCtx, hip_stream: *mut c_void) -> c_int;
    pub fn splat_get_timing(ctx: *mut SplatCtx, ms: *mut f64 /* [6] */, frames: *mut u64, reset: i32) -> c_int;
    pub fn splat_get_records(ctx: *mut SplatCtx, out: *mut SplatRecord, n: u64) -> c_int;
    pub fn splat_get_tile_lists(ctx: *mut SplatCtx, tile_offsets: *mut u32, n_offsets: u64, order: *mut u32, n_order: u64) -> c_int;
    // viewer loop (src/main.rs:69-78): cleared frame, asynchronous copy-out into pinned frames
    pub fn splat_render_stream(ctx: *mut SplatCtx, cam: *const SplatCamera, argb_out: *mut u32) -> c_int;
    pub fn splat_stream_wait(ctx: *mut SplatCtx, argb_out: *const u32) -> c_int;
    pub fn splat_host_alloc(bytes: u64) -> *mut c_void;
    pub fn splat_host_free(p: *mut c_void);
    pub fn splat_host_register(p: *mut c_void, bytes: u64) -> c_int;      // pin a buffer the caller owns (e.g. Buffer2d's storage), once
    pub fn splat_host_unregister(p: *mut c_void) -> c_int;
    // device images for a host without a HIP toolchain of its own
    pub fn splat_device_alloc(ctx: *mut SplatCtx, bytes: u64) -> *mut c_void;
    pub fn splat_device_free(ctx: *mut SplatCtx, d_ptr: *mut c_void);
    pub fn splat_device_upload(ctx: *mut SplatCtx, d_dst: *mut c_void, h_src: *const c_void, bytes: u64) -> c_int;
    pub fn splat_device_download(ctx: *mut SplatCtx, h_dst: *mut c_void, d_src: *const c_void, bytes: u64) -> c_int;
    // multi-GPU, one process (B): what a multi-GPU render_to_buffer binds
    pub fn splat_multi_create(cfg: *const SplatConfig, devices: *const i32, n_devices: i32, out: *mut *mut SplatMulti) -> c_int;
    pub fn splat_multi_destroy(m: *mut SplatMulti);
    pub fn splat_multi_last_error(m: *const SplatMulti) -> *const c_char;
    pub fn splat_multi_upload_scene(m: *mut SplatMulti, n: u64, pos4: *const f32, cov3d: *const f32,
                                    opacity: *const f32, sh: *const f32) -> c_int;
    pub fn splat_multi_balance(m: *mut SplatMulti, cam: *const SplatCamera) -> c_int;
    pub fn splat_multi_set_frame_overlap(m: *mut SplatMulti, n: i32) -> c_int;
    pub fn splat_multi_get_slabs(m: *const SplatMulti, slabs_out: *mut i32) -> c_int;
    pub fn splat_multi_image(m: *mut SplatMulti) -> *mut c_void;
    pub fn splat_multi_ctx(m: *mut SplatMulti, rank: i32) -> *mut SplatCtx;
    pub fn splat_multi_render(m: *mut SplatMulti, cam: *const SplatCamera, argb: *mut u32, stats: *mut SplatStats) -> c_int;
    pub fn splat_multi_render_frame(m: *mut SplatMulti, cam: *const SplatCamera) -> c_int;
    pub fn splat_multi_sync(m: *mut SplatMulti) -> c_int;
    pub fn splat_multi_download(m: *mut SplatMulti, argb_out: *mut u32, w: i32, h: i32) -> c_int;
    // multi-GPU, one process per GPU (A)
    pub fn splat_comm_unique_id(id: *mut u8 /* [128] */) -> c_int;
    pub fn splat_comm_init_rank(ctx: *mut SplatCtx, id: *const u8, n_ranks: i32, rank: i32) -> c_int;
    pub fn splat_slab_partition(row_loads: *const u64, n_rows: i32, n_ranks: i32, row_overhead: f64,
                                slabs_out: *mut i32 /* n_ranks x {row0,row1} */) -> c_int;
    pub fn splat_comm_set_slabs(ctx: *mut SplatCtx, slabs: *const i32) -> c_int;
    pub fn splat_comm_gather(ctx: *mut SplatCtx, d_argb: *mut c_void, w: i32, h: i32, root: i32) -> c_int;
    pub fn splat_comm_loopback(ctx: *mut SplatCtx, on: i32) -> c_int;
    pub fn splat_comm_destroy(ctx: *mut SplatCtx);
}

/*
 * splat_hip.h -- C ABI of libsplat_hip.so, the MI355X (gfx950) drop-in for the
 * rasterisation hot path of thomasantony/splat.
 *
 * The reference has no FFI; its operator surface for this path is the Rust API
 *     GaussianSplatPipeline01::render_to_buffer(&self, &mut euc::Buffer<u32,2>)   src/pipelines.rs:66-86
 *     GaussianSplatPipeline02::render_to_buffer(...)                              src/pipelines.rs:260-280
 * which internally runs sort (src/gaussians.rs:297-306, 464-471), the euc vertex
 * stage (src/pipelines.rs:96-125 / 184-213 -> src/gaussians.rs:40-99, 114-161,
 * 473-522; src/pipelines.rs:17-51), euc's triangle rasteriser, fragment
 * (src/pipelines.rs:127-145) and blend (src/pipelines.rs:147-168).
 * The entry points below are what a Rust `extern "C"` block in that crate binds
 * (INTEGRATION.md shows the stub); every signature uses plain pointers and sizes.
 *
 * Conventions: all matrices column-major f32 (nalgebra storage); pixels are
 * 0xAARRGGBB u32, row-major, `w*h` of them; every function returns SPLAT_OK or
 * a negative error code and never throws/unwinds across the boundary;
 * splat_last_error() gives the message.  One splat_ctx per host thread/GPU.
 */
#ifndef SPLAT_HIP_H
#define SPLAT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SPLAT_OK 0
#define SPLAT_ERR_INVALID (-1)   /* bad argument                                          */
#define SPLAT_ERR_HIP (-2)       /* HIP runtime error (no device, OOM, launch failure)    */
#define SPLAT_ERR_NO_SCENE (-3)  /* render before upload                                  */
#define SPLAT_ERR_CAPACITY (-4)  /* (Gaussian,tile) pair buffer could not be grown        */

#define SPLAT_TILE 16            /* 16x16 pixel tiles                                     */

/* Version of this header's struct layouts and entry points.  splat_abi_version() returns the one the LIBRARY was built
 * with: a binding compiled against another header (splat_stats grew by two fields between versions 4 and 5; a caller that
 * passes the shorter struct has 16 bytes written past it) compares the two before its first call -- the ctypes binding,
 * rust/src/pipelines_hip.rs and the C++ host mirror all do -- and splat_stats_size() tells the byte count it will write. */
#define SPLAT_ABI_VERSION 6

/* modes: bit flags, 0 = the default */
#define SPLAT_MODE_EXACT 0       /* back-to-front, 8-bit truncation per splat as blend() does it; the exponential of
                                    fragment() is the device's (~1.5 ulp): within 1 LSB of libm's on ~1e-5 of the pixels */
#define SPLAT_MODE_CORRECTED_PROJECTION 1 /* same compositing, but cov2d = J W S W^T J^T with the perspective-shear
                                    terms of J (3DGS paper) that the reference drops (its Matrix3::new is
                                    row-major, src/gaussians.rs:141-151).  NOT parity with the reference:
                                    SURVEY section 8(f) rank 2, off by default.                                  */
#define SPLAT_MODE_LIBM_EXP 2    /* fragment()'s exp computed exactly as glibc's expf does (in double, table + cubic):
                                    the frame is then the CPU restatement's frame bit for bit.  A verification mode:
                                    ~1.5x the compositor's time.  May be combined with the flag above.               */
#define SPLAT_MODE_FAST 4        /* the compositor's early-out stops proving the frame EXACT and proves it within one
                                    count instead: the layers behind the walk's start are bracketed as in the exact
                                    mode ([lo,hi] from 0 and from 255, blend() is monotone), but the bracket counts as
                                    closed at hi - lo <= 2 and the walk continues from its MIDDLE (within 1 of the
                                    exact state; blend() never expands a difference of integer states), which half
                                    the optical depth achieves.  Every R, G, B byte is within 1 of the
                                    SPLAT_MODE_EXACT frame's (the alpha byte is the same); about a third less
                                    compositor time.  (SPLAT_FAST_WIDTH=1: closed at hi - lo <= 1, continue from lo.)  The north_star's "front-to-back ... early-out
                                    on saturated alpha", with the error bound proven instead of hoped for.        */

typedef struct splat_ctx splat_ctx;

typedef struct {
    int32_t device;        /* HIP device ordinal                                          */
    int32_t mode;          /* SPLAT_MODE_*                                                */
    /* euc conventions; splat_default_config() documents defaults  */
    int32_t y_up;          /* 1: NDC +y is row 0                                          */
    int32_t sample_half;   /* 1: sample at pixel centres                                  */
    int32_t zclip;         /* 1: cull quads with ndc.z outside [zmin,zmax]                */
    float zmin, zmax;
    uint64_t pair_capacity;/* initial (Gaussian,tile) pair capacity; 0 = auto; grows      */
} splat_config;

/* Per-frame constants = what Camera's getters return (src/camera.rs:70-93). */
typedef struct {
    float view[16];        /* Camera::get_view_matrix                                     */
    float proj[16];        /* Camera::get_project_matrix                                  */
    float w, h;            /* Camera::w, Camera::h                                        */
    float htanx, htany, focal; /* Camera::get_htanfovxy_focal                             */
    float cam_pos[3];      /* Camera::position (the field, src/pipelines.rs:99)           */
    float lowpass;         /* 0.01 for Pipeline01 (gaussians.rs:156), 0.3 for Pipeline02 (:517) */
    int32_t sh_dim;        /* 15 at both call sites (src/pipelines.rs:100,189)            */
} splat_camera;

typedef struct {
    uint64_t n_gaussians;
    uint64_t n_visible;    /* Gaussians with >= 1 covered sample in this context's slab    */
    uint64_t n_singular;   /* det(cov2d) == 0: skipped (the reference panics, pipelines.rs:22) */
    uint64_t n_pairs;      /* D = (visible Gaussian, tile) overlaps                        */
    uint64_t max_tile_len; /* longest per-tile list                                        */
    uint64_t bytes_algorithmic; /* N*148 + n_visible*48 + D*60 + w*h*4   */
    /* device time of the last frame per kernel, ms (HIP events on the context's stream)  */
    float ms_preprocess, ms_scan, ms_emit, ms_sort, ms_composite, ms_total;
    uint64_t n_fallback;   /* compositor waves whose early-out could not be proven exact and were redone in full */
    uint64_t n_sort_fallback; /* tiles re-sorted by the exact bitonic network because depth ties were out of index order */
    uint64_t n_iter_scan;  /* compositor (wave, record) iterations spent in the front-to-back scan           */
    uint64_t n_iter_blend; /* ... and in exact blending (both measured on the frame the stats belong to)     */
    uint64_t n_blocks_culled; /* 256-Gaussian blocks K1 skipped: bounds cannot reach the slab / target      */
    uint64_t flops_algorithmic; /* sum over pixels of (its tile's list length) * 25  */
    uint64_t n_near_tiles;    /* tiles whose list of more than 2048 keys was served by its selected nearest keys       */
    uint64_t n_near_fallback; /* ... of which the whole list had to be sorted after all (a walk reached the selection's far end) */
} splat_stats;

/* Projected per-Gaussian record as the kernels keep it (debug / stage parity). */
typedef struct {
    float cx, cy, hx, hy;           /* quad centre and 3-sigma half extents, pixels        */
    float conic_a, conic_b, conic_c, opacity;
    float r, g, b, depth;           /* SH colour + 0.5 (unclamped); view-space z           */
    int32_t px0, px1, py0, py1;     /* exactly covered inclusive pixel range; px0>px1 = culled */
} splat_record;

uint32_t splat_abi_version(void);    /* SPLAT_ABI_VERSION of the library */
uint64_t splat_stats_size(void);     /* sizeof(splat_stats) as the library writes it */
void splat_default_config(splat_config* cfg);
int splat_create(const splat_config* cfg, splat_ctx** out);
void splat_destroy(splat_ctx* ctx);
const char* splat_last_error(const splat_ctx* ctx);   /* ctx may be NULL: last create() error */

/* Scene upload: exactly GaussianList's buffers (src/gaussians.rs:408-416):
 * pos4 = positions.as_slice() (4n: x,y,z,1), cov3d = 3x3 column-major blocks (9n),
 * opacity (n), sh (48n, f_dc then f_rest un-transposed).  Host pointers; copied. */
int splat_upload_scene(splat_ctx* ctx, uint64_t n, const float* pos4, const float* cov3d,
                       const float* opacity, const float* sh);

/* compute_cov3d (src/gaussians.rs:101-113, 446-462) on the GPU.  scales3 (3n, already exp'd),
 * rot4 (4n, nalgebra coords order i,j,k,w, un-normalised), cov3d_out (9n).  Host pointers. */
int splat_compute_cov3d(splat_ctx* ctx, uint64_t n, const float* scales3, const float* rot4,
                        float* cov3d_out);

/* Restrict rendering to tile rows [tile_row0, tile_row1) (multi-GPU slabs); (0,-1) = all. */
int splat_set_slab(splat_ctx* ctx, int32_t tile_row0, int32_t tile_row1);

/* Load estimate for balancing slabs: (Gaussian,tile) pair count of every tile row of the FULL
 * frame for this camera (the slab setting is ignored and left unchanged).  Runs only the
 * preprocess + scan kernels.  n_rows must be ceil(h / SPLAT_TILE). */
int splat_tile_row_loads(splat_ctx* ctx, const splat_camera* cam, uint64_t* row_pairs, int32_t n_rows);

/* Viewer-loop variant (src/main.rs:69-78: clear, render, present): the frame is rendered onto a
 * CLEARED device image -- what the loop's `color.clear(0)` + `render_to_buffer` amount to -- and
 * copied to `argb_out` (host, w*h u32) asynchronously; the call returns once the work is queued.
 * Frames rotate through four device images, so up to four may be in flight: frames N+1.. render while frame N
 * crosses PCIe (two in flight is a double-buffered window; three or more keep the device's frame pipeline full).
 * `argb_out` is complete after splat_stream_wait(ctx, argb_out) (or splat_sync); give each frame
 * in flight its own buffer, ideally pinned (splat_host_alloc) -- a pageable one works but the
 * copy then blocks the calling thread.  A streamed frame that outgrew its storage on the device (tile
 * bucket, pair buffer, sort launch sizes: all sized from earlier frames) is detected by the wait, which
 * grows the storage and renders that frame again into `argb_out` before it returns: a viewer loop never
 * sees the miss (splat_frames_dropped() counts them).  On a context restricted to a slab only the slab's rows are
 * rendered; the other rows of `argb_out` read as zeros. */
int splat_render_stream(splat_ctx* ctx, const splat_camera* cam, uint32_t* argb_out);
int splat_stream_wait(splat_ctx* ctx, const uint32_t* argb_out);
void* splat_host_alloc(uint64_t bytes);      /* page-locked host memory (hipHostMalloc); NULL on failure */
void splat_host_free(void* p);
/* Page-lock a host image the CALLER owns (hipHostRegister) -- the reference's `color` buffer lives as long as the window
 * (src/main.rs:62), so it is pinned once: splat_render's copies of a pageable image go through the driver's staging
 * buffers at a fraction of the PCIe rate (C3, synchronous host in/out frame: 850 -> 990 frames/s).  Unregister before the
 * memory is freed.  SPLAT_ERR_HIP when the driver refuses (memory it cannot map). */
int splat_host_register(void* p, uint64_t bytes);
int splat_host_unregister(void* p);

/* Device images for callers without a HIP toolchain of their own (a Rust/C host using
 * splat_render_device): plain allocations on the context's GPU, and copies ordered on the context's
 * stream (the copy functions return when the data has arrived). */
void* splat_device_alloc(splat_ctx* ctx, uint64_t bytes);          /* NULL on failure (splat_last_error) */
void splat_device_free(splat_ctx* ctx, void* d_ptr);
int splat_device_upload(splat_ctx* ctx, void* d_dst, const void* h_src, uint64_t bytes);
int splat_device_download(splat_ctx* ctx, void* h_dst, const void* d_src, uint64_t bytes);

/* render_to_buffer: blends the scene onto `argb` (in/out, host, w*h u32).  stats may be NULL. */
int splat_render(splat_ctx* ctx, const splat_camera* cam, uint32_t* argb, splat_stats* stats);

/* The viewer loop's frame, host-visible: `color.clear(0); pipeline.render_to_buffer(&mut color)` (src/main.rs:73-74) as ONE
 * synchronous call.  The clear is fused into the compositor and the pixels cross PCIe once, device -> host -- splat_render
 * (in/out blending, the literal render_to_buffer) ships the caller's zeros up first: 8.3 MB each way at 1080p.  `argb_out`
 * (w*h u32) is WRITTEN, never read.  A page-locked `argb_out` (splat_host_alloc, or the caller's long-lived buffer after
 * splat_host_register: the reference's `color`, src/main.rs:62) that the device can address is written by the compositor
 * itself while the frame is being composited (zero copy, SPLAT_OPT_HOST_ZERO_COPY); any other one is filled by a copy
 * behind the compositor.  With a slab set only the slab's rows of `argb_out` are written.  stats may be NULL (and should be
 * in a loop: a statistics frame reads counters back). */
int splat_render_frame(splat_ctx* ctx, const splat_camera* cam, uint32_t* argb_out, splat_stats* stats);

/* Same with a device-resident image (in/out, w*h u32 in HBM).  Work is enqueued on the
 * context's stream; with sync != 0 (or stats != NULL) the call waits for completion. */
int splat_render_device(splat_ctx* ctx, const splat_camera* cam, void* d_argb, int32_t sync,
                        splat_stats* stats);
/* Wait for everything enqueued.  SPLAT_ERR_CAPACITY: an asynchronous frame (splat_render_device with
 * sync == 0) was skipped on the device because it outgrew its storage -- its image was left untouched and
 * the storage has been grown: render it again.  Reported once.  A synchronous render redoes its OWN frame
 * internally and never returns this code for a frame that composited. */
int splat_sync(splat_ctx* ctx);
/* The viewer loop's frame on a device image: `color.clear(0); render_to_buffer(&mut color)` (src/main.rs:73-74) in one
 * call -- the clear is fused into the compositor (old pixels are not read; tiles nothing covers are zeroed), so the
 * frame costs no separate pass over the image.  Same result, byte for byte, as a memset followed by
 * splat_render_device.  With a slab set, only the slab's rows are cleared and rendered. */
int splat_render_frame_device(splat_ctx* ctx, const splat_camera* cam, void* d_argb, int32_t sync, splat_stats* stats);
uint64_t splat_frames_dropped(const splat_ctx* ctx);  /* frames skipped on the device since splat_create (redone or reported) */
/* Device memory this context holds right now (scene planes, per-frame buffers of its frame slots, key buffers, images it
 * allocated); *peak (nullable) = the most it has held since splat_create. */
uint64_t splat_device_bytes(const splat_ctx* ctx, uint64_t* peak);
/* Frame overlap for SWAP CHAINS (default 1; SPLAT_FRAME_OVERLAP=2 sets it at splat_create).  The reference's loop clears and
 * renders ONE buffer per frame (src/main.rs:71-75), and so does every entry point above: the compositors of consecutive
 * frames run one after the other, in call order, on the context's stream.  A frame that is bound by the latency of its
 * densest tile's lone wavefront -- a small scene, a tile-row slab of a multi-GPU frame -- then leaves most of the chip idle.
 * With n = 2 an ASYNCHRONOUS frame (sync == 0, stats == NULL) to an image that no frame in flight touches composites on
 * a second internal stream, beside the previous frame's compositor: alternate two device images and two frames share
 * the chip (C2: 5.7 k -> 7.9 k frames/s, an eighth-of-a-frame slab of C3: 0.13 -> 0.08 ms per frame).  Frames to the SAME image
 * keep their call order (stream order on one lane: in/out blending and clear+render behave as before), synchronous
 * frames order themselves behind everything in flight, splat_comm_gather follows the frame rendered last.  What changes:
 * an overlapped frame is no longer ordered against work the CALLER enqueues on splat_stream() -- use splat_sync (or a
 * synchronous frame) before touching a target image from a stream of your own.  splat_render_stream's frames are not
 * overlapped (the second compositor stream is the one their copies to the host travel on: a stream more would share a
 * hardware queue with a busy one). */
int splat_set_frame_overlap(splat_ctx* ctx, int32_t n);
/* Tuning options of a context, for hosts that cannot (or should not) reach them through the environment -- a Rust or C
 * application sets them after splat_create.  None of them changes a pixel of an exact-mode frame: they choose between equivalent schedules
 * and storage sizes (SPLAT_MODE_FAST frames stay within 1 of the exact frame per colour byte whatever
 * SPLAT_OPT_EARLY_OUT_EPS / SPLAT_OPT_FAST_CLOSE_WIDTH say, but which of those frames it is depends on them).  The SPLAT_* environment variable of the same purpose, when set, is read at splat_create and
 * PINS the option: a later splat_set_option on it leaves the operator's value in force and returns SPLAT_OK
 * (splat_get_option tells what is in force).  splat_set_option waits for the frames in flight; options that size
 * storage (pipeline depth, key buffer bytes, one-pass binning) take effect with the next frame, which re-allocates.
 * SPLAT_ERR_INVALID: unknown option or value out of range (the range is in the comment of each). */
#define SPLAT_OPT_PIPELINE_DEPTH 1       /* frames in flight on the device, 1..6 (default 6; SPLAT_PIPELINE): 1 = one stream, no
                                            cross-frame overlap; 2 = binning + sort of frame N+1 under the compositor of frame N;
                                            6 = four frame slots, two binning chains in flight                                   */
#define SPLAT_OPT_FUSED_SORT_MAX 2       /* lists up to this many keys are sorted by their tile's compositor workgroup, 0..2048
                                            (default 2048; SPLAT_FUSED_SORT); 0 = every list goes through the sort launches      */
#define SPLAT_OPT_REGION_SPARE 3         /* one-pass binning: how far a tile's key region may grow into the buffer's spare room,
                                            >= 1 (default 4; SPLAT_REGION_SPARE): larger tolerates larger camera jumps           */
#define SPLAT_OPT_EARLY_OUT_EPS 4        /* transmittance below which the compositor's near-to-far scan stops, 0..1 (default
                                            1e-6, 2e-3 in SPLAT_MODE_FAST; SPLAT_EARLY_EPS); 0 switches the early-out off       */
#define SPLAT_OPT_EARLY_OUT_MIN_LIST 5   /* shortest tile list that gets the early-out scan, >= 0 (default 768, 384 in
                                            SPLAT_MODE_FAST; SPLAT_EARLY_MIN)                                                    */
#define SPLAT_OPT_EARLY_OUT_SCAN_EIGHTHS 6 /* the scan gives up after this many eighths of the list, 1..8 (default 4;
                                            SPLAT_EARLY_SCAN8)                                                                   */
#define SPLAT_OPT_SORT_IN_COMPOSITOR 7   /* who sorts lists of more than 2048 keys: 0 = sort launches, 1 = the tile's compositor
                                            workgroup, -1 = chosen per frame from the previous frame (default; SPLAT_SORT_IN_COMP) */
#define SPLAT_OPT_PAIR_WALK 8            /* the exact walk takes two records per step with packed math: 0 / 1, -1 = per frame
                                            from the previous frame's statistics (default; SPLAT_PAIR_BLEND)                     */
#define SPLAT_OPT_TIMING_EVERY 9         /* per-kernel timing events ride on every n-th asynchronous frame, >= 1 (default 8;
                                            SPLAT_TIMING_EVERY)                                                                  */
#define SPLAT_OPT_BLOCK_CULLING 10       /* K1 skips 256-Gaussian blocks whose bounds cannot reach the slab / target: 0 / 1
                                            (default 1; SPLAT_CULL)                                                              */
#define SPLAT_OPT_ONE_PASS_BINNING 11    /* per-tile key regions filled by K1 itself (no count / emit passes): 0 / 1 (default 1;
                                            SPLAT_BUCKETS)                                                                       */
#define SPLAT_OPT_KEY_BUFFER_BYTES 12    /* ceiling on the key buffers of all frame slots together for one-pass binning, bytes
                                            (default 128 GiB; SPLAT_BUCKET_BYTES): beyond it the two-pass path is used          */
#define SPLAT_OPT_FAST_CLOSE_WIDTH 13    /* SPLAT_MODE_FAST: the bracket counts as closed at hi - lo <= 1 or 2 (default 2;
                                            SPLAT_FAST_WIDTH)                                                                    */
#define SPLAT_OPT_PRIORITY_LIST_LEN 14   /* compositor waves of lists at least this long (x2, x4) run at raised priority, >= 1
                                            (default: off; SPLAT_PRIO_LEN)                                                       */
#define SPLAT_OPT_FRAME_OVERLAP 15       /* = splat_set_frame_overlap: 1 / 2 (default 1; SPLAT_FRAME_OVERLAP)                     */
#define SPLAT_OPT_NEAR_SELECT_KEYS 16    /* near selection: of a tile list of more than 2048 keys only the nearest <= this many are
                                            selected (by depth, no sort) and put in order -- the exact early-out never looks
                                            farther on all but a few tiles, which sort their whole list after all (stats:
                                            n_near_tiles, n_near_fallback); 64..2048, 0 = off: every long list is sorted in full
                                            (default 2048; SPLAT_NEAR_KEYS)                                                      */
#define SPLAT_OPT_OVERFLOW_REDO 17       /* a frame whose camera differs from the one its tile regions were sized for carries a
                                            second binning behind its scan -- launches that leave at once unless a list outgrew its
                                            region -- so that such a frame is binned again ON THE DEVICE instead of being skipped and
                                            reported (SPLAT_ERR_CAPACITY at the next splat_sync).  0 = off; 1 = adaptive (default):
                                            only while a list has outgrown its region within the last 256 frames -- a scene that
                                            never does pays nothing; the frame of a camera JUMP carries it too; the first frame of a
                                            smooth path that outgrows a region after a quiet stretch is skipped and reported as
                                            before and arms the redo; 2 = on every frame of a moving camera (five
                                            near-empty launches per frame on the binning stream).  SPLAT_OVERFLOW_REDO           */
#define SPLAT_OPT_START_HINTS 18         /* where a compositor wave's exact walk starts is normally found by a scan of its tile's list
                                            from the near end (the early-out).  With a camera at rest the lists are the previous
                                            frame's lists: the walk starts where that frame's did (one word per wave, kept from
                                            frame to frame) and the scan is skipped; with a camera that moved by less than about
                                            half a degree since the last frame, where it did plus an eighth, three frames of four.
                                            A start that turns out too shallow is retried deeper, as after any scan: exactness
                                            never rests on the hint.  0 = scan every frame; 1 = camera at rest only; 2 = at rest
                                            and in slow motion (default 2; SPLAT_START_HINTS)                                    */
#define SPLAT_OPT_HOST_ZERO_COPY 19      /* splat_render_frame into a page-locked image the device can address: 1 = the compositor
                                            stores its pixels straight into host memory (the image crosses PCIe under the frame),
                                            0 = device image + a copy behind the frame (default 1; SPLAT_HOST_ZERO_COPY)           */
#define SPLAT_OPT_KEYS_PER_GAUSSIAN 20   /* one-pass binning: entries of a frame slot's key buffer per Gaussian of the scene, 4..256,
                                            0 = by the scene's size (default; SPLAT_KEYS_PER_GAUSSIAN).  What the tiles' regions do
                                            not ask for is their room to grow under a moving camera (SPLAT_OPT_REGION_SPARE)       */
#define SPLAT_OPT_COUNT_FIRST 21         /* one-pass binning: which frames count their pairs per tile first (K1's count flavour: geometry
                                            only, a third of a K1) and bin into regions that fit exactly their own camera, instead of
                                            into regions sized from the frame two back: 0 = only a frame slot without regions (first
                                            frames, a new scene / target / slab); 1 (default) = also the next 64 frames of a moving
                                            camera once three in four of the recent frames binned into another camera's regions
                                            outgrew them (were binned twice by the overflow redo, or skipped) -- under every
                                            SPLAT_OPT_OVERFLOW_REDO setting; the frame of a camera jump does NOT count first (it
                                            carries the redo launches instead); 2 = also every frame whose camera moved by more than
                                            half a degree.  The rule is include/splat_policy.h (SPLAT_COUNT_FIRST)                    */
#define SPLAT_OPT_LARGE_SPLAT_TILES 22   /* one-pass binning: a splat of more tiles than this -- and every splat wider or taller than the
                                            projection kernel's 32 x 32-tile window -- is not expanded pair by pair with global atomics by
                                            its K1 block; it goes to the frame's large list, which a second kernel bins tile by tile
                                            (a camera inside the scene: K1 1.1 -> 0.18 ms on 1.5 M Gaussians).  0 = the window alone
                                            decides, -1 = no list (default 128; SPLAT_LARGE_TILES)                                  */
#define SPLAT_OPT_LARGE_LIST_MIN 23      /* ... and frames keep such a list only while the last frame the host has heard from had at
                                            least this many large splats (half of it to let go again; always when nothing is known
                                            or behind a camera jump): the list's kernel is one more launch on a small frame's chain.
                                            0 = always, -1 = never (default 256; SPLAT_LARGE_LIST_MIN; include/splat_policy.h)        */
int splat_set_option(splat_ctx* ctx, int32_t option, double value);
int splat_get_option(const splat_ctx* ctx, int32_t option, double* value);
void* splat_stream(splat_ctx* ctx);                   /* the hipStream_t the kernels run on */
/* Run on a caller-owned hipStream_t (e.g. the stream a device image / RCCL gather lives on). */
int splat_set_stream(splat_ctx* ctx, void* hip_stream);
/* Accumulated per-kernel device time since the last reset, from HIP events recorded on the
 * kernels' own streams around the launches: ms[0..5] = preprocess, scan, emit, sort, composite,
 * status read-back; *frames = frames that carried events.  Event records cost queue bubbles, so
 * only every SPLAT_TIMING_EVERY-th frame (default 8) of an asynchronous run carries them, plus every
 * frame rendered with a stats pointer; averages = ms[k] / *frames.  Waits for outstanding frames. */
int splat_get_timing(splat_ctx* ctx, double ms[6], uint64_t* frames, int32_t reset);

/* Debug / stage parity: results of the last frame.  splat_get_tile_lists needs a frame rendered WITH a stats pointer
 * (ordinary frames keep the short lists they sort on chip and never write them back: SPLAT_ERR_INVALID then). */
int splat_get_records(splat_ctx* ctx, splat_record* out, uint64_t n);
/* tile_offsets: n_tiles+1 entries (slab-local tiles, row-major); order: n_pairs Gaussian indices,
 * each tile's list in blend order (far -> near). Pass NULL to query sizes via stats. */
int splat_get_tile_lists(splat_ctx* ctx, uint32_t* tile_offsets, uint64_t n_offsets, uint32_t* order,
                         uint64_t n_order);
/* How the last frame was binned: > 0 = one-pass binning (the default), the value being the entries of the key buffer
 * its tiles' regions live in -- every tile owns a region sized from the list it had two frames earlier (x 1.5 + 512 keys),
 * so the buffer holds ~1.6 x the frame's (Gaussian, tile) pairs; a frame whose list outgrows its region is skipped and
 * redone / reported like any frame that outgrows its storage (splat_sync).  0 = two-pass binning (count, scan, emit into
 * exactly sized lists): SPLAT_BUCKETS=0, a caller-fixed pair_capacity, or key buffers that would not fit
 * SPLAT_BUCKET_BYTES (default 128 GiB for all frame slots together).  < 0 without a frame. */
int64_t splat_binning_mode(splat_ctx* ctx);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU: one frame = disjoint TILE-ROW SLABS, one per GPU, every GPU holding the
 * whole scene; the only exchange is one gather of slab pixel rows to the root per frame -- a grouped
 * ncclSend / ncclRecv over xGMI, contiguous rows straight out of each peer's image into the root's.
 * Pixels are independent given the ordered splat list (src/pipelines.rs:147-168), so the gathered frame
 * equals the single-GPU frame byte for byte.  RCCL (librccl.so.1) is loaded on first use.
 * Two forms, same kernels, same gather:
 *   (A) one process per GPU (torchrun / MPI style): splat_comm_* on an ordinary context;
 *   (B) one process, one host thread + context per device: splat_multi_* (ncclCommInitAll).
 * What replaces what: the reference renders the whole frame in one render_to_buffer call
 * (src/pipelines.rs:66-86); euc fans rows out to CPU threads -- these entry points fan tile rows out to GPUs.
 * ------------------------------------------------------------------------------------------------ */
#define SPLAT_UNIQUE_ID_BYTES 128

/* Contiguous tile-row slabs minimising the heaviest slab: row_loads = splat_tile_row_loads() output,
 * row_overhead = fixed cost per tile row in the same unit (pairs).  slabs_out: n_ranks x {row0, row1}.
 * row_loads == NULL: equal split of n_rows.  Deterministic: every rank derives the same partition. */
int splat_slab_partition(const uint64_t* row_loads, int32_t n_rows, int32_t n_ranks, double row_overhead,
                         int32_t* slabs_out);

/* (A) one process per GPU.  Rank 0 calls splat_comm_unique_id and hands the 128 bytes to the other ranks by
 * its own means (a file, a socket, torch.distributed ...); every rank then calls splat_comm_init_rank on its
 * context (collective: ncclCommInitRank on the context's device). */
int splat_comm_unique_id(uint8_t id[SPLAT_UNIQUE_ID_BYTES]);
int splat_comm_init_rank(splat_ctx* ctx, const uint8_t id[SPLAT_UNIQUE_ID_BYTES], int32_t n_ranks, int32_t rank);
/* The partition (n_ranks x {row0,row1}, identical on every rank); also sets this context's own slab. */
int splat_comm_set_slabs(splat_ctx* ctx, const int32_t* slabs);
/* Gather: enqueued on the context's stream behind the frames rendered so far.  d_argb = this rank's w x h
 * device image; afterwards (stream order) the root's image holds every rank's rows. */
int splat_comm_gather(splat_ctx* ctx, void* d_argb, int32_t w, int32_t h, int32_t root);
/* Test hook for one-GPU boxes: with `on`, a SINGLE-rank communicator's gather moves this rank's slab rows through
 * RCCL to itself (ncclSend + ncclRecv to the own rank in one group, on the context's stream) and restores them from
 * what arrived -- the code path, RCCL kernel included, of the multi-rank gather.  The frame is unchanged if RCCL
 * delivered the rows intact.  n_ranks > 1: SPLAT_ERR_INVALID. */
int splat_comm_loopback(splat_ctx* ctx, int32_t on);
void splat_comm_destroy(splat_ctx* ctx);              /* also done by splat_destroy */

/* (B) one process.  devices[i] = HIP ordinal of rank i (rank 0 is the root).  A device may be listed more than
 * once (slabs then share that GPU; RCCL refuses duplicate devices, so rows travel as device-to-device copies
 * ordered by events -- meant for testing the decomposition on a single-GPU box); SPLAT_MULTI_TRANSPORT=peer
 * selects those copies for distinct devices too (hipMemcpyPeerAsync over xGMI). */
typedef struct splat_multi splat_multi;
int splat_multi_create(const splat_config* cfg /* device field ignored; NULL = defaults */, const int32_t* devices,
                       int32_t n_devices, splat_multi** out);
void splat_multi_destroy(splat_multi* m);
const char* splat_multi_last_error(const splat_multi* m);   /* m may be NULL: last create() error */
int splat_multi_upload_scene(splat_multi* m, uint64_t n, const float* pos4, const float* cov3d, const float* opacity,
                             const float* sh);               /* replicated on every device, in parallel */
/* Load-balanced slabs for this camera (count-only pass on the root + splat_slab_partition).  cam == NULL: equal
 * slabs for the target size of the last partition (SPLAT_ERR_INVALID before any frame / balance: the height fixes
 * the tile rows).  The partition stays until the next call or a frame of another target size. */
int splat_multi_balance(splat_multi* m, const splat_camera* cam);
int splat_multi_get_slabs(const splat_multi* m, int32_t* slabs_out /* n_devices x {row0,row1} */);
/* render_to_buffer across the devices: blends the scene onto `argb` (host, in/out, w*h u32).  stats (nullable)
 * = sums over the slabs (a Gaussian that reaches two slabs counts in both). */
int splat_multi_render(splat_multi* m, const splat_camera* cam, uint32_t* argb, splat_stats* stats);
/* Viewer-loop frame (src/main.rs:69-78), device resident: every device clears its slab rows, renders them and
 * sends them to the root's image.  Asynchronous: returns once the frame is queued on every device's thread;
 * splat_multi_sync waits.  The root's image (device memory on devices[0]) is splat_multi_image(). */
int splat_multi_render_frame(splat_multi* m, const splat_camera* cam);
/* Waits for every rank.  SPLAT_ERR_CAPACITY: an asynchronous slab frame outgrew storage sized from earlier frames and
 * was skipped on its device -- the root's image then holds that slab's rows as the gather found them (cleared or
 * stale); the storage has been grown: render the frame again (splat_multi_render_frame may report the same for a loss
 * found while it re-partitions).  Any other error: the first one a rank recorded since the last sync
 * (splat_multi_last_error names the rank). */
int splat_multi_sync(splat_multi* m);
/* The root's image that holds the most recent frame (read it after splat_multi_sync).  With frame overlap 2 the frames of
 * splat_multi_render_frame alternate between two images on every device -- ask again after each frame. */
void* splat_multi_image(splat_multi* m);
/* splat_set_frame_overlap for every rank (default 1).  2: every device keeps two slab images and the frames of
 * splat_multi_render_frame use them in turn, so the compositor (and the row gather) of frame N+1 runs beside frame N's on
 * each device; the partition weighs a tile row's fixed cost for that.  Waits for the frames in flight; the next frame
 * partitions again. */
int splat_multi_set_frame_overlap(splat_multi* m, int32_t n);
int splat_multi_download(splat_multi* m, uint32_t* argb_out, int32_t w, int32_t h);   /* root image -> host (after a sync) */
splat_ctx* splat_multi_ctx(splat_multi* m, int32_t rank);  /* the rank's context (statistics, timing); not to be rendered on directly */

#ifdef __cplusplus
}
#endif
#endif

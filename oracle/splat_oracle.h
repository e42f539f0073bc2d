/*
 * splat_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain C++ restatement of the rasterisation hot path of thomasantony/splat
 * (/root/reference/src/{gaussians,pipelines,camera}.rs + the euc rasteriser it
 * calls).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (splat_amd/) never does.
 *
 * PARITY STATUS
 *   pinned   : per-Gaussian math (focal, cov2d scaling law, conic, 3-sigma box,
 *              NDC corner) against the outputs recorded in
 *              notes/00_Gaussian_Projection.ipynb:212-215 and values re-derived
 *              from that notebook's class (tests/golden/notebook_kat.json);
 *              screen y direction (NDC +y -> row 0) by comparing the committed
 *              Rust/euc screenshot notes/screenshot.png with the notebook's
 *              recorded render (same orientation; the notebook maps NDC +y to
 *              row 0, notes/util.py:105-113).
 *   UNPINNED : everything that lives inside the third-party `euc` crate
 *              (git rev 290e14c, Cargo.lock:221-223), whose source is absent:
 *              sample position, inside test, z-clip range, barycentric
 *              interpolation rounding.  They are runtime switches here
 *              (orc_conventions) with documented defaults.
 *
 * All arithmetic is IEEE f32, evaluated in the reference's operation order,
 * compiled with -ffp-contract=off (Rust never contracts a*b+c).
 */
#ifndef SPLAT_ORACLE_H
#define SPLAT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* euc conventions that decide pixels. */
typedef struct {
    int32_t y_up;          /* 1: NDC +y is row 0 (default: INFERRED from notes/screenshot.png and notes/util.py:101-113, not pinned) */
    int32_t sample_half;   /* 1: sample at pixel centre (x+.5,y+.5) (default)   */
    int32_t zclip;         /* 1: cull quads whose ndc.z is outside [zmin,zmax]  */
    float zmin, zmax;      /* default 0..1 (euc CoordinateMode::VULKAN)         */
    int32_t raster;        /* 0: analytic rectangle, coordxy = sample - centre (default).
                              1: SENSITIVITY VARIANT -- the quad as two triangles (0-1-2, 0-2-3 of
                                 src/pipelines.rs:7-14) rasterised separately with f32 barycentric
                                 weights, inclusive edges (a sample on the shared diagonal is blended
                                 by both), coordxy interpolated from the corner values.  It is what
                                 a triangle rasteriser does in general, NOT a transcription of euc
                                 (absent); it measures how far euc-internal rounding can move pixels. */
    int32_t corrected_projection; /* 0 (default): the reference's cov2d (no perspective-shear term, quirk Q9).
                              1: NOT the reference -- J enters transposed, as in the 3DGS paper; the
                                 counterpart of splat_config.mode = SPLAT_MODE_CORRECTED_PROJECTION. */
} orc_conventions;

/* Per-frame camera constants, i.e. what Camera's getters return
 * (src/camera.rs:70-93).  Matrices are column-major like nalgebra. */
typedef struct {
    float view[16];
    float proj[16];
    float w, h;
    float htanx, htany, focal;
    float cam_pos[3];      /* Camera::position FIELD (src/pipelines.rs:99)      */
    float lowpass;         /* 0.01 Pipeline01 / 0.3 Pipeline02                  */
    int32_t sh_dim;        /* 15 at both call sites (src/pipelines.rs:100,189)  */
} orc_camera;

/* The projected per-Gaussian record = what Pipeline::vertex computes, once
 * instead of 6x (src/pipelines.rs:96-125, 17-51). */
typedef struct {
    float cx, cy;          /* quad centre in pixel coordinates                 */
    float hx, hy;          /* 3-sigma half extents in pixels (bboxsize_cam)    */
    float conic[3];        /* inv(cov2d) (0,0),(0,1),(1,1)                     */
    float opacity;
    float rgb[3];          /* SH colour + 0.5, unclamped                       */
    float depth;           /* view-space z (sort key)                          */
    float ndc[4];          /* P*V*p / w                                        */
    float cov2d[4];        /* column-major 2x2 incl. low-pass                  */
    int32_t visible;       /* 0: culled (z-clip, non-finite, singular, empty)  */
    int32_t px0, px1, py0, py1; /* conservative inclusive pixel range (clamped) */
} orc_record;

typedef struct {
    uint64_t n_visible;
    uint64_t n_singular;   /* det == 0: the reference would panic (pipelines.rs:22) */
    uint64_t n_fragments;  /* covered samples, i.e. blend() calls              */
    uint64_t n_tile_pairs; /* (visible Gaussian, 16x16 tile) overlaps = D      */
    double ms_preprocess, ms_sort, ms_raster;
} orc_stats;

void orc_default_conventions(orc_conventions* c);

/* ---- camera (src/camera.rs) ------------------------------------------- */
/* Camera::new + compute_matrices + getters, for yaw/pitch given. */
void orc_camera_make(float h, float w, const float pos[3], float yaw, float pitch,
                     float lowpass, int32_t sh_dim, orc_camera* out);

/* ---- scene math (src/gaussians.rs) ------------------------------------ */
/* compute_cov3d :101-113 / :446-462. rot = (i,j,k,w) nalgebra coords order;
 * cov3d_out = 9 floats column-major per Gaussian. */
void orc_compute_cov3d(uint64_t n, const float* scales3, const float* rot4, float* cov3d_out);
/* eval_spherical_harmonics :40-99 (+0.5, no clamp). */
void orc_eval_sh(const float* sh48, int32_t sh_dim, const float dir[3], float out[3]);
/* project_cov3d_to_screen :114-161 / :473-522; out = 2x2 column-major. */
void orc_project_cov2d(const float pos[3], const float cov3d[9], const orc_camera* cam, float out[4]);
void orc_project_cov2d_corrected(const float pos[3], const float cov3d[9], const orc_camera* cam, float out[4]);
/* stable ascending view-z argsort :297-306 / :464-471. */
void orc_sort(uint64_t n, const float* pos4, const float view[16], uint32_t* order_out);
/* full vertex stage for every Gaussian. */
void orc_preprocess(uint64_t n, const float* pos4, const float* cov3d, const float* opacity,
                    const float* sh48, const orc_camera* cam, const orc_conventions* conv,
                    orc_record* out);

/* ---- fragment / blend (src/pipelines.rs:127-168) ---------------------- */
void orc_fragment(const float vdata[9], float out[4]);
uint32_t orc_blend(uint32_t old_pixel, const float frag[4]);

/* ---- whole frame: render_to_buffer (src/pipelines.rs:66-86) ------------ */
/* argb is in/out (the reference blends onto the caller's buffer), w*h u32
 * 0xAARRGGBB row-major.  Rows [row0,row1) only are touched (slab rendering);
 * nthreads > 1 splits the rows into bands (per-pixel order is unchanged). */
int orc_render(uint64_t n, const float* pos4, const float* cov3d, const float* opacity,
               const float* sh48, const orc_camera* cam, const orc_conventions* conv,
               uint32_t* argb, int32_t row0, int32_t row1, int32_t nthreads, orc_stats* stats);

/* ---- PLY loader (src/gaussians.rs:246-283, 375-405) -------------------- */
/* Returns vertex count (or <0 on error) when outputs are NULL; otherwise fills
 * pos4 (4n, w=1), scales (3n, exp applied), opacity (n, sigmoid applied),
 * rot4 (4n as i,j,k,w, NOT normalised), sh (48n), after mean-recentring. */
int64_t orc_load_ply(const char* path, float* pos4, float* scales3, float* opacity,
                     float* rot4, float* sh48);

#ifdef __cplusplus
}
#endif
#endif

// splat_internal.h -- shared between splat_api.hip (host) and splat_kernels.hip (gfx950 kernels).
#ifndef SPLAT_INTERNAL_H
#define SPLAT_INTERNAL_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/splat_hip.h"

#ifndef SPLAT_K1X
#define SPLAT_K1X 0      // K1 timing experiments (tools/k1_ab.py); 0 = the product
#endif
// Round-7 timing experiments (tools/lab/build_variant.sh name -DSPLAT_EXP_...=...; profiles/r07_*): 0 = the product
#ifndef SPLAT_EXP_STAGE2
#define SPLAT_EXP_STAGE2 0     // compositor: the staging verdicts computed twice (what they cost)
#endif
#ifndef SPLAT_EXP_K1DUMMY
#define SPLAT_EXP_K1DUMMY 0    // K1: this many dummy VALU instructions per (Gaussian, tile) hand-out step (what per-pair verdicts would cost)
#endif
#ifndef SPLAT_EXP_SKIPFULL
#define SPLAT_EXP_SKIPFULL 0   // near selection: lists of this many keys or more that need a full sort are left unsorted (what their sort costs a frame)
#endif
#ifndef SPLAT_EXP_SORT2
#define SPLAT_EXP_SORT2 0      // compositor: the short lists' in-LDS sort run twice (what it costs)
#endif
#ifndef SPLAT_EXP_K1DROP
#define SPLAT_EXP_K1DROP 0     // K1: 1 = close-up rectangles dropped from binning, n >= 2 = every rectangle of more than n tiles (invalid frames)
#endif

namespace splat {

constexpr int TILE = SPLAT_TILE;
constexpr int SCENE_PLANES = 16;  // float4 planes per Gaussian (see pack_scene_kernel)
constexpr int LIVE_PLANES = 10;   // planes read per frame at sh_dim <= 27 (160 B / Gaussian)


// Everything a frame's kernels need, passed by value (kernarg -> SGPRs).
struct FrameConst {
    float view[16];
    float proj[16];
    float w, h;
    float htanx, htany, focal;
    float cam[3];
    float lowpass;
    int sh_dim;
    int y_up, sample_half, zclip;
    float zmin, zmax;
    int W, H;              // integer target size
    int tiles_x;           // tiles per row
    int tile_row0;         // slab: first tile row
    int n_tile_rows;       // slab: tile rows in this context
    int row_px0, row_px1;  // slab pixel rows [row_px0, row_px1)
    float early_eps;       // compositor early-out: transmittance below which a pixel stops needing layers; 0 = off
    float close_width;     // the skipped layers' [lo,hi] bracket counts as closed when hi - lo <= this on every channel:
                           // 0 = the exact frame, 2 = SPLAT_MODE_FAST (the walk continues from the bracket's middle: within 1 of exact)
    int early_min;         // shortest list the early-out is tried on
    int early_scan8;       // the transmittance scan gives up after this many eighths of the list
    int prio_len;          // lists >= prio_len / 2x / 4x run at wave priority 1 / 2 / 3
    unsigned int bucket_cap; // one-pass binning: entries of the key buffer the tiles' regions live in (0: two-pass binning with exact lists)
    int corrected;         // SPLAT_MODE_CORRECTED_PROJECTION: J enters transposed (perspective-shear terms kept)
    int cull_blocks;       // K1 skips 256-Gaussian blocks whose bounds cannot reach the slab (needs lowpass > 0)
    int start_hints;       // compositor: the camera is at rest -- a wave's exact walk may start where the previous frame's did (start_hint) instead of scanning for it
    int start_light;       // ... in very slow motion: half the margin on a hinted start, the scan every eighth frame instead of every fourth
    int redo_only;         // K1 as a REDO launch: leaves at once unless the frame's scan flagged a tile that outgrew its region
    int large_tiles;       // one-pass binning: a splat of more tiles than this goes to the frame's large list (bin_large_kernel); 0 = only
                           // the splats wider or taller than K1's 32 x 32-tile window
};

// Upload-time bounds of one K1 block (256 consecutive slots of the Morton-ordered scene): the AABB
// of the finite positions and the largest Frobenius norm of a cov3d (>= its spectral norm).
struct BlockBounds { float lo[3], hi[3], fmax, pad; };

// Device-side frame status, read back once per frame.
struct FrameStatus {
    unsigned long long n_visible;
    unsigned long long n_singular;
    unsigned long long n_pairs;
    unsigned int max_tile_len;
    unsigned int overflow;   // 1: n_pairs > capacity; 2: a tile outgrew its bucket; 3: more long lists than the
                             // sort grids were sized for; 4: the long lists outgrew the second key buffer.  Emit/sort/composite skipped
    unsigned long long n_fallback;   // waves whose early-out bracket did not close (redone in full)
    unsigned long long n_sort_fallback; // tiles whose radix-by-depth order failed the 64-bit check (depth ties): bitonic redo
    unsigned int n_near_tiles;       // tiles whose long list (> 2048 keys) was served by its selected nearest keys (select_near) ...
    unsigned int n_near_fallback;    // ... and those among them that needed the whole list sorted after all
    unsigned int n_large;            // one-pass binning: splats K1 found large this frame (listed for bin_large_kernel, or only counted)
    unsigned int n_window;           // ... and those among them wider or taller than K1's 32 x 32-tile window (one atomic per pair without a list)
    unsigned int n_ge8192, n_ge2048;    // tiles whose list has >= 8192 / >= 2048 keys: in `order` they are a prefix
    unsigned int n_ge16384;             // likewise >= 16384 (the lists sorted as several runs and merged)
    unsigned int redone;                // 1: the frame outgrew its regions and was binned again on the device (overflow redo); written by every scan
    unsigned int n_long_keys;           // one-pass binning: entries of the second key buffer the frame's lists of more than 2048 keys ask for
    unsigned int arrived;               // 1: the frame's scan has written this status (the host zeroes its copy when it enqueues the frame)
    unsigned long long n_blocks_culled; // K1 blocks skipped by the bounds test (filled on the host from the block flags)
    unsigned long long layout_total;    // one-pass binning: key-buffer entries the regions built from this frame ask for (layout_kernel)
};

// 48-byte projected record (3 x float4), stored in SLOT order (the Morton order of the scene planes: K1 writes them
// coalesced, the compositor's gathers are local) and gathered by the compositor through the keys' slot halves.
struct Rec {
    float4 a;   // cx, cy, hx, hy
    float4 b;   // conic a, b, c, opacity
    float4 c;   // r, g, b, power below which alpha < 1/255 for certain
};

// Experiment switches of the launch wrappers (SPLAT_SORT_RADIX_MIN, SPLAT_SCAN_THREADS, SPLAT_DBG_NTILES, SPLAT_COMP_LDS_PAD):
// per context, read at splat_create; use_launch_knobs() makes a context's set current for the calling thread.
struct LaunchKnobs {
    unsigned int sort_radix_min = 128;     // lists up to this length use the bitonic network
    int scan_threads = 0;                  // 0: by the tile count
    unsigned int dbg_ntiles = 0;           // != 0: composite only the N longest tiles
    unsigned int comp_lds_pad = 0;         // extra dynamic LDS per compositor workgroup (an occupancy cap)
    unsigned int k1_lds_pad = 0;           // SPLAT_K1_LDS_PAD: the same for K1 (lab: how K1's time depends on the blocks resident per CU)
    unsigned int dbg_select_stride = 0;    // SPLAT_DBG_SELECT_STRIDE: slots of the tile order per workgroup of the near selection's launch (default 8)
    unsigned long long dbg_keys2_entries = 0; // SPLAT_DBG_KEYS2_ENTRIES: initial size of a frame slot's second key buffer (tests force its growth)
    int dbg_hint_radius = -1;              // SPLAT_DBG_HINT_RADIUS: the near selection's neighbourhood, in tiles (default: by the camera's motion)
    unsigned int dbg_starts = 0;           // SPLAT_DBG_STARTS: statistics frames record (list length, nearest keys the walk needed) per wave
};
void use_launch_knobs(const LaunchKnobs* k);

void launch_pack_scene(hipStream_t s, uint64_t n, const float* pos4, const float* cov3d, const float* opacity,
                       const float* sh, const unsigned int* perm, float4* planes);
void launch_cov3d(hipStream_t s, uint64_t n, const float* scales3, const float* rot4, float* cov3d);
void launch_preprocess(hipStream_t s, uint64_t n, const float4* planes, const unsigned int* orig, FrameConst fc, Rec* recs,
                       float* depth, ushort4* rect, unsigned int* counts, unsigned int* vislist, unsigned long long* keys,
                       const BlockBounds* bounds,
                       unsigned int* blockinfo /* per block: bit 31 = skipped by culling; one-pass binning: visible | singular << 9 */,
                       FrameStatus* status,
                       const unsigned int* layout = nullptr /* one-pass binning (fc.bucket_cap != 0): counts[t] is the cursor of tile t's
                                                               region keys[layout[t] .. layout[t+1]); nullptr: two-pass counting */,
                       bool count_only = false /* one-pass binning's COUNT flavour: counts[t] += the tile's pairs and nothing else (no SH,
                                                  no record, no key) -- the pass in front of a layout that fits exactly this camera */,
                       uint4* large_list = nullptr, unsigned int* large_count = nullptr /* one-pass binning: the frame's list of large
                                                  splats (n entries) and its counter -- launch_bin_large behind this launch bins them */);
// the large splats K1 listed, tile by tile (bin_large_kernel): same stream, right behind launch_preprocess; `cursors` as given to it
void launch_bin_large(hipStream_t s, const FrameConst& fc, const uint4* large_list, const unsigned int* large_count, unsigned int large_cap, unsigned int* cursors,
                      unsigned long long* keys, const FrameStatus* status, bool count_only);
void launch_scan(hipStream_t s, unsigned int m, unsigned int* counts, unsigned int* offsets, unsigned int* cursor,
                 unsigned int* order, unsigned int* lens, FrameStatus* status, unsigned long long capacity,
                 unsigned int bucket_cap, unsigned int grid_big, unsigned int grid_mid, unsigned int grid_long,
                 FrameStatus* host_status = nullptr /* pinned, device-visible: the scan also delivers the status there */,
                 const unsigned int* layout = nullptr /* one-pass binning: the regions the frame was binned into */,
                 unsigned int* next_layout = nullptr, unsigned int* next_counts = nullptr /* both given: a second workgroup of
                     the launch builds the regions + cursors of the next frame on this stream (see launch_layout) */,
                 float spare_max = 4.0f /* how far a region may grow into the buffer's spare room */,
                 bool redo_only = false /* the second scan of a frame binned again on the device: nothing unless status->overflow == 2 */,
                 unsigned int* off2 = nullptr /* one-pass binning: per tile, where its room in the SECOND key buffer starts -- handed out
                                                 by this scan to the lists of more than 2048 keys */,
                 unsigned int cap2 = 0 /* entries of the second key buffer: beyond it the frame is flagged (overflow 4) */,
                 unsigned int* large_count = nullptr /* the frame's large-splat counter: reset here for the slot's next K1 */,
                 unsigned int tiles_x = 0, unsigned int motion_radius = 0 /* != 0 (a moving camera): the next frame's regions are sized from
                                                 the longest list within this many tiles of each tile (build_layout) */);
// the regions (and cursors) of the slot's next one-pass frame from this frame's lists; an all-zero `layout` with cursors
// counted from zero is the bootstrap
void launch_layout(hipStream_t s, unsigned int m, const unsigned int* counts, const unsigned int* layout, unsigned int* next_layout,
                   unsigned int* next_counts, unsigned int key_entries, FrameStatus* status, FrameStatus* host_status, float spare_max = 4.0f,
                   const FrameStatus* redo_gate = nullptr /* != nullptr: a redo launch -- does nothing unless redo_gate->overflow == 2 */,
                   unsigned int* large_count = nullptr /* see launch_scan */);
void launch_emit(hipStream_t s, uint64_t n, FrameConst fc, const float* depth, const ushort4* rect, const unsigned int* orig,
                 const unsigned int* vislist, unsigned int* cursor, unsigned long long* keys, const FrameStatus* status);
// grid_big / grid_mid: how many entries of `order` (longest lists first) the 1024- and 512-thread
// sort launches cover; the scan validates them against the frame's actual list lengths.
void launch_sort(hipStream_t s, unsigned int n_tiles, unsigned int grid_big, unsigned int grid_mid, unsigned int grid_long, const unsigned int* offsets,
                 const unsigned int* order, const unsigned int* lens, unsigned long long* keys, unsigned long long* keys2,
                 FrameStatus* status, const unsigned int* orig /* slot -> original index: the order among equal depths */,
                 unsigned int fused_sort_max = 0,
                 const unsigned int* off2 = nullptr /* where a tile's room in keys2 starts; nullptr: at offsets[tile], like its list in keys */);
// fused_sort_max: lists of up to this many keys (<= 2048) are sorted by the compositor's workgroups
// themselves (launch_sort must be given the same value and then leaves them alone); 0 = off.
// near selection instead of the sort launches: the nearest keys of every list of more than 2048 keys, by last frame's need
void launch_select(hipStream_t s, unsigned int n_tiles, const unsigned int* offsets, const unsigned int* order, const unsigned int* lens,
                   unsigned long long* keys, unsigned long long* keys2, FrameStatus* status, const unsigned int* orig, unsigned int near_cap,
                   const unsigned int* need_hint, unsigned int* near_m /* per tile: how many of the nearest keys are in order */,
                   unsigned int tiles_x, unsigned int tile_rows /* the tile grid: a tile's selection also looks at its neighbours' hints */,
                   unsigned int* near_thr = nullptr /* one word per tile, kept from frame to frame: the depth its last selection began at */,
                   unsigned int grid = 0 /* workgroups (each strides over the tile order); 0 = an eighth of the tiles */,
                   bool at_rest = false /* the camera of the last frames: selections sized tightly */,
                   const unsigned int* off2 = nullptr /* see launch_sort */,
                   int hint_radius = 2 /* a tile's selection is sized from its own walks' need and its neighbours' within this many tiles */);
void launch_composite(hipStream_t s, unsigned int n_tiles, FrameConst fc, const unsigned int* offsets,
                      const unsigned int* order, const unsigned int* lens, unsigned long long* keys, const Rec* recs,
                      uint32_t* argb, FrameStatus* status, const unsigned int* orig, unsigned int fused_sort_max = 0,
                      uint2* iters = nullptr /* per wave (scan, blend) iteration counts, statistics frames only */,
                      bool keep_keys = true /* lists sorted inside the compositor are also written back to the bucket
                                               (the debug getters read them there); off on ordinary frames */,
                      bool pair_walk = false /* the two-records-per-step flavour of the exact walk (same pixels) */,
                      bool libm_exp = false /* SPLAT_MODE_LIBM_EXP: expf as the host libm computes it */,
                      bool clear_first = false /* the frame starts from a cleared image: old pixels are not read, tiles
                                                  nothing covers are zeroed (color.clear(0) of src/main.rs:73, fused) */,
                      unsigned long long* keys2 = nullptr /* != nullptr: no sort launch ran; lists of more than 2048 keys
                                                             are sorted by their tile's workgroup through this buffer */,
                      const unsigned int* near_m = nullptr /* != nullptr (with keys2): near selection -- launch_select ran in front: of a
                                                   list of more than 2048 keys only the nearest near_m[tile] are in order (composite_tile) */,
                      unsigned int* need_hint = nullptr /* 4 words per tile, kept from frame to frame: how many of its list's nearest
                                                           keys each wave's walk needed (sizes the next frame's selection) */,
                      unsigned int* start_hint = nullptr /* 4 words per tile, kept from frame to frame: where each wave's exact walk
                                                            started, in keys from the list's near end (fc.start_hints) */,
                      const unsigned int* off2 = nullptr /* see launch_sort */);
hipError_t init_device_kernels();   // per-device kernel attributes; call with the device current

// ---- splat_multi.hip: the multi-GPU layer's hooks into a context (splat_ctx itself stays private to splat_api.hip)
struct CommState;                              // RCCL communicator + partition of one context
CommState** ctx_comm_slot(splat_ctx* c);
hipStream_t ctx_stream(splat_ctx* c);
hipStream_t frame_stream(splat_ctx* c);    // the stream the most recent frame's compositor is on (compositor lanes, splat_api.hip)
int ctx_device(const splat_ctx* c);
int frame_tail(splat_ctx* c);                // re-record the most recent frame's "ended" event behind work enqueued on its lane (a row gather)
int ctx_quiesce(splat_ctx* c);               // wait for everything enqueued; a skipped frame stays pending for splat_sync
int ctx_fail(splat_ctx* c, int code, const char* msg);
void comm_release(CommState* s);               // splat_destroy -> here

}  // namespace splat
#endif

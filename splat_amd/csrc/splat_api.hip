// splat_api.hip -- C ABI (include/splat_hip.h) over the gfx950 kernels.  Host side only:
// buffer ownership, per-frame constants, launch sequence, HIP-event timing, error reporting.
//
// Frames overlap on the device (SPLAT_PIPELINE): per-frame buffers live in SLOTS used in rotation; preprocess +
// scan + sort of later frames run on internal high-priority streams into the next slots while frame N composites
// on the caller's stream (fork/join = two events per frame; nothing is skipped or reused between frames).
//   1  one stream, no overlap
//   2  the chain (K1, scan, sort) of frame N+1 on the bin stream under the compositor of frame N
//   3  the sort on a stream of its own (it starves beside the other two: no gain)
//   4, 5  three / four slots on the two streams of mode 2 (+0..2 %)
//   6  (default) four slots, the chains of consecutive frames on alternating streams: with one chain at a time the
//      frame time is the chain's length under contention; with two the compositor is the bound (C3 +4 %, C1 +10 %)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "splat_internal.h"
#include "../../include/splat_policy.h"

using namespace splat;

namespace {
thread_local std::string g_create_error;
constexpr int N_EV = 9;        // e0..e2 on the bin stream (start, K1, scan), e8, e3, e4 on the sort stream (start, K2, K3), e5..e7 on the caller's (K4 start, K4 end, status)
constexpr int N_TIMES = 6;     // preprocess, scan, emit, sort, composite, status read-back
constexpr int EV_RING = 32;
static_assert(EV_RING == SPLAT_POLICY_RING, "the frame policy sees the whole status ring");
constexpr int N_SLOTS = 4;

struct EvSet {
    hipEvent_t e[N_EV];
    bool used = false;
    bool timed = false;        // the per-kernel events e0..e6, e8 were recorded for this frame
};

struct Slot {                  // everything one frame writes before the image
    Rec* recs = nullptr;
    float* depth = nullptr;
    ushort4* rect = nullptr;
    unsigned int* vislist = nullptr;
    unsigned int* counts = nullptr;        // per tile: pair count (two-pass binning, zero between frames) / cursor of the tile's region (one-pass)
    unsigned int* counts_b = nullptr;      // one-pass binning: cursors and regions exist twice per slot -- the layout of the slot's NEXT frame
    unsigned int* lay_a = nullptr;         // is written (layout_kernel of an earlier frame on the same stream) while nothing reads that copy
    unsigned int* lay_b = nullptr;
    int flip = 0;                          // which copy the slot's next frame uses (0: counts / lay_a, 1: counts_b / lay_b)
    bool layout_valid = false;             // ... and whether it holds regions for the current scene / target / slab
    unsigned int* offsets = nullptr;
    unsigned int* cursor = nullptr;
    unsigned int* order = nullptr;
    unsigned int* lens = nullptr;           // list length per tile (the list starts at offsets[tile])
    // Binning again on the device (overflow redo): counters, regions and cursors of a frame whose lists outgrew the regions
    // it was given -- its second binning pass writes here, not into the copies the pipeline hands on
    unsigned int* redo_layout = nullptr;
    unsigned int* redo_cursors = nullptr;
    uint64_t layout_cam[2] = {0, 0};        // per copy of the layout: a hash of the camera whose lists sized it
    unsigned int* near_m = nullptr;         // near selection: per tile, how many of its list's nearest keys launch_select put in order
    unsigned long long* keys = nullptr;
    unsigned long long* keys2 = nullptr;   // the second key buffer: sorted near selections, scatter space of the long lists' sorts and merges
    unsigned int* off2 = nullptr;          // one-pass binning: per tile, where its room in keys2 starts (handed out by the frame's scan to the
                                           // lists of more than 2048 keys; two-pass binning mirrors the first buffer instead)
    unsigned int* blockinfo = nullptr;     // per K1 block: the info word this slot's last K1 wrote (see launch_preprocess);
                                           // per slot, because the K1s of consecutive frames run concurrently
    uint4* large_list = nullptr;           // one-pass binning: the frame's large splats (key, tile rectangle), n entries -- K1 lists them,
    unsigned int* large_count = nullptr;   // bin_large_kernel bins them tile by tile; the counter is zero between frames (scan / layout reset it)
    FrameStatus* d_status = nullptr;
    hipEvent_t ev_binned = nullptr;        // bin stream -> sort stream: buckets and lengths are final
    hipEvent_t ev_ready = nullptr;         // sort stream -> caller's stream: lists are sorted
    int free_ring = -1;                    // compositor's stream -> bin stream: the slot is free when the frame that used it last has
                                           // ended -- that frame's ring event (an event of the slot's own would be a second barrier
                                           // packet behind every compositor: ~3 us of queue drain per frame)
    bool used = false;
};
}  // namespace

struct splat_ctx {
    splat_config cfg{};
    hipStream_t stream = nullptr;          // compositor + image: the caller-visible stream
    bool own_stream = false;
    hipStream_t bin_stream = nullptr;      // K1 + scan of a later frame
    hipStream_t sort_stream = nullptr;     // K2 + K3 (depth 2: the bin stream itself)
    // scene
    uint64_t n = 0;
    float4* planes = nullptr;
    unsigned int* orig = nullptr;          // slot -> original Gaussian index (Morton order of position)
    BlockBounds* bounds = nullptr;         // per K1 block of 256 slots (block culling)
    bool cull_blocks = true;               // SPLAT_CULL=0 disables
    std::vector<unsigned int> h_orig;
    // per-frame buffers
    Slot slots[N_SLOTS];
    unsigned int m_alloc = 0;
    uint64_t cap = 0;                      // entries in each used slot's keys buffer
    uint64_t cap2 = 0;                     // entries in each used slot's second key buffer (0: none).  Two-pass binning: a mirror of the first
                                           // (cap2 == cap); one-pass: room for the lists of more than 2048 keys only (default_keys2_capacity)
    uint64_t keys2_want = 0;               // a harvested frame's long lists outgrew the second key buffer: grow to this
    // one-pass binning (per-tile buckets): on unless SPLAT_BUCKETS=0, the caller fixed pair_capacity,
    // the buckets would not fit bucket_bytes, or a tile outgrew the largest LDS-sortable bucket
    // one-pass binning (per-tile regions of the key buffer, sized from earlier frames' lists): on unless SPLAT_BUCKETS=0, the
    // caller fixed pair_capacity, or the key buffers would not fit bucket_bytes
    bool use_buckets = true;
    bool bucket_failed = false;            // sticky until the scene changes
    uint64_t bucket_bytes = 128ull << 30;  // SPLAT_BUCKET_BYTES: all key buffers of all slots together (288 GB of HBM per GPU)
    uint64_t layout_want = 0;              // entries the regions of a harvested frame asked for and did not get (grow to this)
    unsigned int layout_m = 0;             // tile count the slots' layouts were built for
    bool last_one_pass = false;            // what the previous frame's binning was (the cursors must be zero for two-pass counting)
    unsigned int* zero_layout = nullptr;   // m_alloc zeros: the empty layout of the bootstrap (every key dropped, every pair counted)
    uint64_t dev_bytes = 0, dev_bytes_peak = 0;   // device memory held by this context
    LaunchKnobs knobs;                     // experiment switches of the launch wrappers (this context's)
    float region_spare = 4.0f;             // SPLAT_REGION_SPARE: how far a tile's region may grow into the key buffer's spare room (1: not at all)
    uint64_t frame_idx = 0;
    int last_slot = -1;                    // buffer slot of the most recent frame (debug getters)
    bool last_lists_in_memory = false;     // ... and whether its compositor wrote the lists it sorted back to the buckets
    FrameStatus* h_status = nullptr;       // pinned, one per event-ring entry
    // Every frame in flight has a device status of its own (one per event-ring entry).  The scan kernel -- where a
    // frame's pair count, longest list and every overflow verdict are decided -- initialises it and writes the same
    // words straight into the pinned host copy, so an asynchronous frame needs no read-back and no reset on any
    // stream: the caller's stream carries nothing but the compositors back to back (a copy and two fills used to sit
    // between them, ~30 us per frame; a side stream for them shares a hardware queue with a binning stream and
    // serialises the frames).  Frames rendered with statistics still copy the final status (late counters).
    FrameStatus* d_status_ring = nullptr;
    bool clear_first = false;              // the frame being enqueued starts from a cleared image (fused into the compositor)
    // host-image path
    uint32_t* d_img = nullptr;
    size_t img_cap = 0;
    // streaming path (splat_render_stream): two device images, a copy stream, per-image events
    static constexpr int S_IMGS = 4;       // streamed frames in flight (device images): the pipeline wants three (two chains + a compositor)
    uint32_t* s_img[S_IMGS] = {};
    size_t s_cap = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t s_rendered[S_IMGS] = {}, s_copied[S_IMGS] = {};
    const uint32_t* s_dst[S_IMGS] = {};
    bool s_used[S_IMGS] = {};
    uint64_t s_idx = 0;
    int s_ring[S_IMGS] = {-1, -1, -1, -1}; // event-ring entry of the frame each streaming image holds
    splat_camera s_cam[S_IMGS] = {};       // ... and its camera (a frame skipped on the device is redone by splat_stream_wait)
    // slab
    int slab0 = 0, slab1 = -1;
    // timing
    EvSet ring[EV_RING];
    int ring_next = 0;
    int last_ring = -1;
    double acc_ms[N_TIMES] = {0, 0, 0, 0, 0, 0};
    uint64_t acc_frames = 0;
    // last frame
    FrameConst fc{};
    unsigned int n_tiles = 0;
    uint64_t overflow_want = 0;            // a harvested frame overflowed the pair buffer: grow to this
    bool bucket_overflow = false;          // a harvested frame overflowed a tile bucket: leave one-pass binning
    // sort launch sizes: the long-list sort launches cover a prefix of the longest-first tile order,
    // sized from the most recent harvested frame (+25 % + slack); the device validates, a miss redoes the frame
    bool sort_hint = false, sort_grid_miss = false;
    unsigned int hint_ge8192 = 0, hint_ge2048 = 0, hint_ge16384 = 0;
    // which flavour of the exact walk the compositor runs (the pixels are the same): 0 = one record per step, 1 = two
    // records per step with packed math (fewer issue slots: for frames whose compositor is bound by its longest
    // list's single wave, not by throughput), -1 = by the last harvested frame's pairs per key of the longest list
    int pair_mode = -1;                    // SPLAT_PAIR_BLEND
    uint64_t hint_pairs = 0; unsigned int hint_maxlen = 0; unsigned int hint_large = 0, hint_window = 0;
    unsigned int grid_big = 0, grid_mid = 0, grid_long = 0;      // what the frame being enqueued uses
    FrameStatus last{};
    // frames skipped on the device (their storage outgrown: see finish_frame).  A synchronous call redoes its own
    // frame; a lost ASYNCHRONOUS frame is reported once, by the next splat_sync / splat_stream_wait
    uint64_t frames_dropped = 0, frames_drop_reported = 0;
    bool deferred_drop = false;
    bool tight_grids = false;              // SPLAT_DBG_TIGHT_GRIDS: sort launches sized with no margin (tests force a miss)
    uint2* d_iters = nullptr;              // per compositor wave: (scan, blend) iterations of a frame rendered with stats
    unsigned int iters_alloc = 0;
    bool iters_valid = false;
    // Who sorts the lists of more than 2048 keys: 0 = the sort launches (74 / 147 KB workgroups, starved beside a compositor
    // in flight, but the cheaper code), 1 = the tile's own compositor workgroup (no launches, no starvation, more
    // work), -1 = by the previous frame: the compositor when the AVERAGE list is longer than 2048 keys, i.e. when
    // the sort launches would carry most of the frame's keys, and the frame is throughput-bound (C5: +7 %; C3 -5 %, C2 -13 %
    // if forced).  SPLAT_SORT_IN_COMP.
    int sort_in_comp = -1;
    float fast_width = 2.0f;               // SPLAT_MODE_FAST: bracket width that counts as closed (SPLAT_FAST_WIDTH: 1 or 2)
    float early_eps = 1e-6f;               // SPLAT_EARLY_EPS overrides (0 disables the early-out)
    int early_min = 768;                   // SPLAT_EARLY_MIN
    int early_scan8 = 4;                   // SPLAT_EARLY_SCAN8
    int prio_len = 0x3fffffff;             // SPLAT_PRIO_LEN
    unsigned int fused_sort_max = 2048;    // SPLAT_FUSED_SORT: lists up to this length are sorted inside the compositor (0: off)
    // Near selection (SPLAT_NEAR_KEYS / SPLAT_OPT_NEAR_SELECT_KEYS; 0 = off): a list of more than 2048 keys is not sorted; its
    // tile's compositor workgroup selects the nearest <= near_cap keys by depth and sorts those -- the exact early-out never
    // looks farther on all but a few tiles, which then sort their whole list after all.  No sort launches in such frames.
    unsigned int near_cap = 2048;
    // Overflow redo (SPLAT_OVERFLOW_REDO / SPLAT_OPT_OVERFLOW_REDO, default on): a frame whose camera differs from the one its
    // tile regions were sized for carries a second binning (count pass, exact regions, K1, scan) behind its scan, as launches
    // that leave at once unless that scan found a tile beyond its region -- such a frame is binned again on the device
    // instead of being skipped, reported and rendered again by the caller.
    // 0 = off; 1 = ADAPTIVE (default): the redo launches ride on moving frames only while a list has outgrown its region within
    // the last 256 frames (a scene that never does -- most -- pays nothing; the first such frame after a quiet stretch is
    // skipped and reported as before, and arms the redo); 2 = on every moving frame.
    int overflow_redo = 1;
    // What the frame policy (include/splat_policy.h, splat_policy.cpp: a pure function, tested without a GPU) carries from frame
    // to frame: the previous camera and how long it has been the same, the count-first / overflow-redo runs left, how the frames
    // in the status ring were binned.  reset_policy() where the lists it speaks of stop existing (scene, target, slab, options).
    splat_policy_state pol{};
    // One-pass binning: splats of more tiles than this (and every splat wider or taller than K1's 32 x 32-tile window) go to the
    // frame's large list and are binned tile by tile behind K1 (bin_large_kernel).  SPLAT_LARGE_TILES: 0 = the window alone
    // decides, < 0 = no list at all (K1's blocks expand close-ups themselves, one atomic per pair: the round-5 path).
    int layout_motion = 1;               // SPLAT_LAYOUT_MOTION=0: regions always sized from each tile's own list (round 6)
    int large_list_min = 256;            // SPLAT_LARGE_LIST_MIN: large splats a recent frame must have had for frames to keep the list (splat_policy.h)
    int large_tiles = 128;               // (C2 / C3 / C5, bench pose and from inside: 96-128 best of 0..1024, profiles/r07_large_splats.txt)
    int count_first = 1;                   // SPLAT_OPT_COUNT_FIRST: 0 only slots without a layout; 1 + the 64 moving frames behind a run of frames that outgrew
                                           // their regions (three in four of the recent ones); 2 + every frame whose camera moved by more than half a degree
    bool idle = false;                     // nothing of this context is in flight (set by the waits that drain every stream, cleared by every enqueue)
    int start_hints = 2;                   // SPLAT_OPT_START_HINTS / SPLAT_START_HINTS: 0 the compositor scans for its walks' starts on every frame; 1 not with
                                           // a camera at rest; 2 nor, three frames of four, with one in slow motion (see enqueue_frame)
    bool one_pass_select = true;           // SPLAT_DBG_ONE_PASS_SELECT=0: near selection always takes its two passes (histogram, compaction)
    unsigned int* need_hint = nullptr;     // 4 x m_alloc words: per tile and wave, the nearest keys its walk needed in the most recent frame
    bool last_near = false;                // the most recent frame ran with near selection: its long lists are unordered in memory
    int timing_every = 8;                  // SPLAT_TIMING_EVERY: per-kernel events on every n-th frame (and whenever stats are asked for)
    int pipeline = 6;                      // frames in flight on the device (SPLAT_PIPELINE = 1..6, see enqueue_frame)
    // Compositor LANES (splat_set_frame_overlap): the compositors of consecutive frames run one after the other on the
    // context's stream (lane 0) -- unless overlap is on and the frames go to DIFFERENT images (a swap chain), in which case
    // the second lane (the copy stream: ensure_lane) takes every other one and two compositors share the chip.  A frame that is
    // bound by the latency of its densest tile's lone wave (small scenes, multi-GPU slabs) leaves the chip mostly idle:
    // with two lanes C2 runs at 7.9 k instead of 5.7 k frames/s, an eighth-of-a-frame slab at 0.08 instead of 0.13 ms.
    // lane[q]: its most recent frame (sequence number, event-ring entry).  img_tab: the most recent frame of every image
    // seen lately (address range, lane, ring entry) -- hazards are decided per IMAGE: with three images in rotation the
    // earlier frame to an image is not its lane's last one.
    struct Lane { uint64_t seq = 0; int ring = -1; };
    Lane lane[2];
    struct ImgRec { const char* lo = nullptr; const char* hi = nullptr; uint64_t seq = 0; int ring = -1; int lane = 0; };
    static constexpr int N_IMG_TAB = 8;
    ImgRec img_tab[N_IMG_TAB];
    hipStream_t comp2 = nullptr;           // lane 1 (lane 0 is `stream`)
    int overlap = 1;                       // splat_set_frame_overlap / SPLAT_FRAME_OVERLAP: 2 = asynchronous frames may use lane 1
    int last_lane = 0;                     // the lane of the most recent frame
    uint64_t lane_seq = 0;
    bool streamed_call = false;            // splat_render_stream is rendering: its frames stay on lane 0 (lane 1's stream carries their copies)
    hipEvent_t pre_wait = nullptr;         // one-shot: the next frame's compositor waits for it (splat_render_stream: its image is still crossing PCIe)
    uint32_t env_pinned = 0;               // bit k: SPLAT_OPT_k was set from the environment at splat_create (splat_set_option leaves it alone)
    int host_zero_copy = 1;                // SPLAT_OPT_HOST_ZERO_COPY: splat_render_frame's compositor stores into a device-addressable host image
    uint64_t region_mult = 0;              // the key buffer's entries per Gaussian chosen for this scene (0: not yet)
    unsigned int keys_per_gaussian = 0;    // SPLAT_OPT_KEYS_PER_GAUSSIAN: 0 = default_region_capacity decides
    splat::CommState* comm = nullptr;      // multi-GPU: RCCL communicator + partition (splat_multi.hip)
    std::string err;
};

namespace splat {
CommState** ctx_comm_slot(splat_ctx* c) { return &c->comm; }
hipStream_t ctx_stream(splat_ctx* c) { return c->stream; }
// the stream the most recent frame's compositor was enqueued on: work that must follow THAT frame (the gather of its rows,
// the copy of its pixels) goes there -- and the next frame to the same image follows it on the same lane
hipStream_t frame_stream(splat_ctx* c) { return (c->last_lane && c->comp2) ? c->comp2 : c->stream; }
// Work enqueued BEHIND the most recent frame on its lane that later frames / copies must also follow (the gather of its
// rows): the frame's ring event -- the one every cross-lane hazard wait, join_lanes, the slot-reuse wait and the image
// table use -- is recorded again behind it, so "the frame has ended" includes that work from now on.
int frame_tail(splat_ctx* c) {
    if (c->last_ring < 0 || !c->ring[c->last_ring].used) return SPLAT_OK;
    hipError_t e = hipEventRecord(c->ring[c->last_ring].e[7], frame_stream(c));
    if (e != hipSuccess) { c->err = std::string("hipEventRecord(frame tail): ") + hipGetErrorString(e); return SPLAT_ERR_HIP; }
    return SPLAT_OK;
}
int ctx_device(const splat_ctx* c) { return c->cfg.device; }
int ctx_fail(splat_ctx* c, int code, const char* msg) { if (c) c->err = msg; return code; }
}  // namespace splat

namespace {

#define HIP_TRY(ctx, expr)                                                                             \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                            \
            return SPLAT_ERR_HIP;                                                                      \
        }                                                                                              \
    } while (0)

int fail(splat_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

// Upload-time ordering: 30-bit Morton code of the position inside the scene's bounding box.
// order[j] = original index stored in slot j.  Ties keep index order; non-finite positions go first.
inline uint32_t spread3(uint32_t v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
void morton_order(uint64_t n, const float* pos4, std::vector<unsigned int>& order) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint64_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            float v = pos4[4 * i + a];
            if (std::isfinite(v)) { lo[a] = std::min(lo[a], v); hi[a] = std::max(hi[a], v); }
        }
    float sc[3];
    for (int a = 0; a < 3; ++a) sc[a] = (hi[a] > lo[a]) ? 1023.0f / (hi[a] - lo[a]) : 0.0f;
    std::vector<uint64_t> keyed(n);
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t code = 0;
        for (int a = 0; a < 3; ++a) {
            float v = pos4[4 * i + a];
            uint32_t q = std::isfinite(v) ? (uint32_t)std::min(1023.0f, std::max(0.0f, (v - lo[a]) * sc[a])) : 0u;
            code |= spread3(q) << a;
        }
        keyed[i] = ((uint64_t)code << 32) | (uint64_t)i;
    }
    std::sort(keyed.begin(), keyed.end());
    order.resize(n);
    for (uint64_t j = 0; j < n; ++j) order[j] = (unsigned int)keyed[j];
}

// Bounds of every K1 block (256 consecutive slots): AABB of the finite centres, largest ||cov3d||_F.
void block_bounds(uint64_t n, const float* pos4, const float* cov3d, const std::vector<unsigned int>& order,
                  std::vector<BlockBounds>& out) {
    const uint64_t nb = (n + 255) / 256;
    out.resize(nb);
    for (uint64_t b = 0; b < nb; ++b) {
        BlockBounds bb;
        for (int a = 0; a < 3; ++a) { bb.lo[a] = INFINITY; bb.hi[a] = -INFINITY; }
        bb.fmax = 0.0f; bb.pad = 0.0f;
        const uint64_t j1 = std::min<uint64_t>(n, (b + 1) * 256);
        for (uint64_t j = b * 256; j < j1; ++j) {
            const uint64_t i = order[j];
            const float* p = pos4 + 4 * i;
            if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))) continue;   // never visible
            for (int a = 0; a < 3; ++a) { bb.lo[a] = std::min(bb.lo[a], p[a]); bb.hi[a] = std::max(bb.hi[a], p[a]); }
            double f2 = 0.0;
            for (int e = 0; e < 9; ++e) f2 += (double)cov3d[9 * i + e] * (double)cov3d[9 * i + e];
            float f = (float)std::sqrt(f2) * 1.0001f;
            if (!(f >= 0.0f)) f = INFINITY;                    // NaN: unbounded extent
            bb.fmax = std::max(bb.fmax, f);
        }
        if (!(bb.lo[0] <= bb.hi[0]))                            // no finite centre at all: NaN bounds answer "maybe"
            for (int a = 0; a < 3; ++a) { bb.lo[a] = NAN; bb.hi[a] = NAN; }
        out[b] = bb;
    }
}

// Device allocations of a context go through these two, so that splat_device_bytes() can say what it holds (sizes are
// kept in a side table: hipFree does not tell).
void ledger_add(splat_ctx* c, void* p, size_t bytes);
void ledger_del(void* p);
template <typename T>
hipError_t dmalloc(splat_ctx* c, T** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) ledger_add(c, (void*)*p, bytes);
    return e;
}
template <typename T>
void dfree(T*& p) {
    if (p) { ledger_del((void*)p); (void)hipFree(p); p = nullptr; }
}

// hipMemset on device memory is ENQUEUED (on the legacy default stream) and may return before the fill has run -- and the
// context's streams are non-blocking: nothing on them waits for that stream.  A fill that the next launches depend on is
// waited for here.  (Found with eight processes on one GPU, where the default stream's fill arrives late: the first
// splat_tile_row_loads of two ranks in eight counted into counters that were zeroed afterwards -- tools/row_loads_stress.py.)
hipError_t fill_now(void* p, int value, size_t bytes) {
    hipError_t e = hipMemset(p, value, bytes);
    return e == hipSuccess ? hipStreamSynchronize(nullptr) : e;
}

const float* ev_times(const EvSet& s, float t[N_TIMES]) {
    static const int pairs[N_TIMES][2] = {{0, 1}, {1, 2}, {8, 3}, {3, 4}, {5, 6}, {6, 7}};
    for (int k = 0; k < N_TIMES; ++k) {
        t[k] = 0.f;
        (void)hipEventElapsedTime(&t[k], s.e[pairs[k][0]], s.e[pairs[k][1]]);
    }
    return t;
}

void harvest(splat_ctx* c, int r) {
    EvSet& s = c->ring[r];
    if (!s.used) return;
    (void)hipEventSynchronize(s.e[7]);
    if (s.timed) {
        float t[N_TIMES];
        ev_times(s, t);
        for (int k = 0; k < N_TIMES; ++k) c->acc_ms[k] += t[k];
        c->acc_frames++;
    }
    const FrameStatus& st = c->h_status[r];
    if (st.overflow) c->frames_dropped++;
    if (st.overflow == 1) c->overflow_want = std::max<uint64_t>(c->overflow_want, st.n_pairs);
    if (st.overflow == 2) c->bucket_overflow = true;           // a tile's list outgrew its region: the layouts are stale
    if (st.overflow == 2 || st.redone == 1u) c->pol.redo_armed = SPLAT_POLICY_REDO_RUN;  // ... or did and was binned again on the device: keep the redo launches on
    if (st.layout_total > c->cap) c->layout_want = std::max<uint64_t>(c->layout_want, st.layout_total);   // the regions were cut off
    if (st.overflow == 3) c->sort_grid_miss = true;
    if (st.overflow == 4) c->keys2_want = std::max<uint64_t>(c->keys2_want, st.n_long_keys);
    if (st.overflow == 0 || st.overflow == 3) { c->sort_hint = true; c->hint_ge8192 = st.n_ge8192; c->hint_ge2048 = st.n_ge2048; c->hint_ge16384 = st.n_ge16384; }
    if (st.overflow == 0) { c->hint_pairs = st.n_pairs; c->hint_maxlen = st.max_tile_len; c->hint_large = st.n_large; c->hint_window = st.n_window; }
    s.used = false;
}

std::mutex g_ledger_mu;
std::unordered_map<void*, std::pair<splat_ctx*, size_t>> g_ledger;
void ledger_add(splat_ctx* c, void* p, size_t bytes) {
    std::lock_guard<std::mutex> g(g_ledger_mu);
    g_ledger[p] = {c, bytes};
    c->dev_bytes += bytes;
    c->dev_bytes_peak = std::max(c->dev_bytes_peak, c->dev_bytes);
}
void ledger_del(void* p) {
    std::lock_guard<std::mutex> g(g_ledger_mu);
    auto it = g_ledger.find(p);
    if (it == g_ledger.end()) return;
    it->second.first->dev_bytes -= it->second.second;
    g_ledger.erase(it);
}

// Page-locked host ranges this library made (splat_host_alloc / splat_host_register): start -> bytes.  The zero-copy frame
// stores W*H*4 bytes through the device mapping of `argb_out`; the mapping must cover all of them (a caller that
// registered a slab's worth of a larger image got a GPU page fault instead of the copy path).  A range the table does not
// know (locked by the caller's own hipHostMalloc / hipHostRegister, a framework's pinned tensor) is asked of the runtime.
static std::mutex g_host_mu;
static std::unordered_map<const char*, size_t> g_host_ranges;
static void host_range_add(const void* p, size_t bytes) { std::lock_guard<std::mutex> g(g_host_mu); g_host_ranges[(const char*)p] = bytes; }
static void host_range_del(const void* p) { std::lock_guard<std::mutex> g(g_host_mu); g_host_ranges.erase((const char*)p); }
static bool host_range_is_locked(const void* host, void* dev, size_t bytes) {
    {
        std::lock_guard<std::mutex> g(g_host_mu);
        const char* lo = (const char*)host;
        for (const auto& r : g_host_ranges)
            if (r.first <= lo && lo < r.first + r.second) return lo + bytes <= r.first + r.second;      // (ours: the table decides)
    }
    hipDeviceptr_t base = nullptr; size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)dev) == hipSuccess && base != nullptr &&
        (const char*)dev >= (const char*)base && (const char*)dev + bytes <= (const char*)base + size)
        return true;
    (void)hipGetLastError();
    return false;
}

// Device -> host copy of image rows behind a frame.  A destination of which only a PART is page-locked (ADVICE r5's case: the
// caller registered a slab's worth of a larger image) is refused whole by hipMemcpyAsync (invalid argument): it is copied in
// pieces cut at the borders of the locked ranges this library knows, and a piece the runtime still refuses goes through a
// page-locked staging buffer.  The usual destination -- wholly locked or wholly pageable -- is one call as before.
static hipError_t copy_to_host_image(uint32_t* dst, const uint32_t* src, size_t bytes, hipStream_t stream) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
    if (e != hipErrorInvalidValue) return e;
    (void)hipGetLastError();
    std::vector<size_t> cuts{0, bytes};
    {
        std::lock_guard<std::mutex> g(g_host_mu);
        const char* lo = (const char*)dst;
        for (const auto& r : g_host_ranges)
            for (const char* edge : {r.first, r.first + r.second})
                if (edge > lo && edge < lo + bytes) cuts.push_back((size_t)(edge - lo));
    }
    std::sort(cuts.begin(), cuts.end());
    for (size_t k = 0; k + 1 < cuts.size(); ++k) {
        const size_t off = cuts[k], len = cuts[k + 1] - cuts[k];
        if (!len) continue;
        e = hipMemcpyAsync((char*)dst + off, (const char*)src + off, len, hipMemcpyDeviceToHost, stream);
        if (e == hipErrorInvalidValue) {
            (void)hipGetLastError();
            void* stage = nullptr;
            if ((e = hipHostMalloc(&stage, len)) != hipSuccess) return e;
            e = hipMemcpyAsync(stage, (const char*)src + off, len, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e == hipSuccess) std::memcpy((char*)dst + off, stage, len);
            (void)hipHostFree(stage);
        }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// The frame policy's memory speaks of lists that no longer exist (another scene, target, slab, key-buffer layout) or of
// thresholds that changed (an option): forget it.  Statuses of frames still in the ring stay where they are; ring_kind = 0
// makes the policy ignore them ('binned twice' words of an earlier scene could arm 64 count-first frames on the next).
void reset_policy(splat_ctx* c) {
    c->pol.still_frames = 0; c->pol.last_cam_hash = 0;
    c->pol.count_first_left = 0; c->pol.redo_armed = 0;
    std::memset(c->pol.ring_kind, 0, sizeof c->pol.ring_kind);
}

int sync_all(splat_ctx* c) {
    if (c->bin_stream) HIP_TRY(c, hipStreamSynchronize(c->bin_stream));
    if (c->sort_stream) HIP_TRY(c, hipStreamSynchronize(c->sort_stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->copy_stream) HIP_TRY(c, hipStreamSynchronize(c->copy_stream));      // (lane 1 is this stream too)
    return SPLAT_OK;
}

// the stream (and events) of splat_render_stream's device -> host copies
hipError_t ensure_copy_stream(splat_ctx* c) {
    if (c->copy_stream) return hipSuccess;
    hipStream_t st = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int k = 0; k < splat_ctx::S_IMGS && e == hipSuccess; ++k) {
        if (!c->s_rendered[k]) e = hipEventCreateWithFlags(&c->s_rendered[k], hipEventDisableTiming);
        if (e == hipSuccess && !c->s_copied[k]) e = hipEventCreateWithFlags(&c->s_copied[k], hipEventDisableTiming);
    }
    if (e != hipSuccess) { if (st) (void)hipStreamDestroy(st); return e; }      // (events made so far are reused by the next attempt, destroyed with the context)
    c->copy_stream = st;
    return hipSuccess;
}

// The copy entry points run on the context's stream: they follow what the second lane still holds as well.
hipError_t join_lanes(splat_ctx* c) {
    if (!c->comp2 || c->lane[1].seq == 0 || c->lane[1].ring < 0) return hipSuccess;
    return hipStreamWaitEvent(c->stream, c->ring[c->lane[1].ring].e[7], 0);
}

// the second compositor lane exists from the first time something may use it
int ensure_lane(splat_ctx* c) {
    if (c->comp2 || c->pipeline == 0) return SPLAT_OK;        // (one stream for everything: no lanes)
    // Lane 1 IS the copy stream of splat_render_stream.  HIP multiplexes its streams onto four hardware queues, and the
    // context already has four (the caller's, two binning chains, the copy stream): a fifth shares a queue with a busy
    // one -- measured both ways round, whichever of the two was created later lost a third of its rate (streamed C3
    // 2570 -> 1840 frames/s, or the swap chain on C2 8100 -> 6060).  The two are never busy together: streamed frames
    // stay on lane 0 (render_device_impl), so the copy stream carries either copies or lane-1 compositors.
    HIP_TRY(c, ensure_copy_stream(c));
    c->comp2 = c->copy_stream;
    return SPLAT_OK;
}

int ensure_bins(splat_ctx* c, unsigned int m) {
    if (m + 1 <= c->m_alloc) return SPLAT_OK;
    int rc = sync_all(c);
    if (rc != SPLAT_OK) return rc;
    c->m_alloc = 0;
    c->sort_hint = false;                  // another target geometry: the list-length profile is unknown again
    dfree(c->zero_layout);
    HIP_TRY(c, dmalloc(c, &c->zero_layout, sizeof(unsigned int) * (size_t)(m + 1)));
    HIP_TRY(c, fill_now(c->zero_layout, 0, sizeof(unsigned int) * (size_t)(m + 1)));
    dfree(c->need_hint);
    HIP_TRY(c, dmalloc(c, &c->need_hint, sizeof(unsigned int) * 9u * (size_t)(m + 1)));      // (+ one word per tile behind them: the depth its last selection began at; + four: where its waves' walks started)
    HIP_TRY(c, fill_now(c->need_hint, 0, sizeof(unsigned int) * 9u * (size_t)(m + 1)));
    for (Slot& s : c->slots) {
        dfree(s.counts); dfree(s.offsets); dfree(s.cursor); dfree(s.order); dfree(s.lens); dfree(s.counts_b); dfree(s.lay_a); dfree(s.lay_b);
        dfree(s.near_m); dfree(s.redo_layout); dfree(s.redo_cursors); dfree(s.off2);
        s.layout_valid = false; s.flip = 0;
        HIP_TRY(c, dmalloc(c, &s.off2, sizeof(unsigned int) * (size_t)(m + 1)));
        HIP_TRY(c, dmalloc(c, &s.near_m, sizeof(unsigned int) * (size_t)(m + 1)));
        HIP_TRY(c, dmalloc(c, &s.redo_layout, sizeof(unsigned int) * (size_t)(m + 1)));
        HIP_TRY(c, dmalloc(c, &s.redo_cursors, sizeof(unsigned int) * (size_t)(m + 1)));
        HIP_TRY(c, dmalloc(c, &s.counts_b, sizeof(unsigned int) * (size_t)(m + 1)));
        HIP_TRY(c, dmalloc(c, &s.lay_a, sizeof(unsigned int) * (size_t)(m + 1)));
        HIP_TRY(c, dmalloc(c, &s.lay_b, sizeof(unsigned int) * (size_t)(m + 1)));
        HIP_TRY(c, dmalloc(c, &s.lens, sizeof(unsigned int) * (size_t)(m + 1)));
        HIP_TRY(c, dmalloc(c, &s.counts, sizeof(unsigned int) * (size_t)(m + 1)));
        HIP_TRY(c, dmalloc(c, &s.offsets, sizeof(unsigned int) * (size_t)(m + 1)));
        HIP_TRY(c, dmalloc(c, &s.cursor, sizeof(unsigned int) * (size_t)(m + 1)));
        HIP_TRY(c, dmalloc(c, &s.order, sizeof(unsigned int) * (size_t)(m + 1)));
        HIP_TRY(c, fill_now(s.counts, 0, sizeof(unsigned int) * (size_t)(m + 1)));
    }
    c->m_alloc = m + 1;
    return SPLAT_OK;
}

int slots_in_use(const splat_ctx* c);

// depth / pixel rectangle / visible list per Gaussian and frame slot (16 bytes x N x slots): what the COUNTING flavour of K1
// hands to K2 -- two-pass binning, splat_tile_row_loads, splat_get_records.  A one-pass frame keeps all three in registers,
// so they exist from the first call that needs them (the slots the caller names), not from the upload.
int ensure_two_pass_buffers(splat_ctx* c, Slot& s) {
    if (s.depth || c->n == 0) return SPLAT_OK;
    hipError_t e = dmalloc(c, &s.depth, sizeof(float) * c->n);
    if (e == hipSuccess) e = dmalloc(c, &s.rect, sizeof(ushort4) * c->n);
    if (e == hipSuccess) e = dmalloc(c, &s.vislist, sizeof(unsigned int) * c->n);
    if (e != hipSuccess) { dfree(s.depth); dfree(s.rect); dfree(s.vislist); return fail(c, SPLAT_ERR_HIP, std::string("hipMalloc(two-pass buffers): ") + hipGetErrorString(e)); }
    return SPLAT_OK;
}

// keys of at least `want` entries, and a second key buffer of at least `want2`, in every slot that frames rotate through
// (a buffer that is large enough is kept)
int ensure_keys(splat_ctx* c, uint64_t want, uint64_t want2) {
    if (want <= c->cap && want2 <= c->cap2) return SPLAT_OK;
    if (want >= 0xFFFFFFF0ull || want2 >= 0xFFFFFFF0ull) return fail(c, SPLAT_ERR_CAPACITY, "pair count exceeds 2^32");
    int rc = sync_all(c);
    if (rc != SPLAT_OK) return rc;
    want = std::max(want, c->cap); want2 = std::max(want2, c->cap2);
    const bool grow1 = want > c->cap, grow2 = want2 > c->cap2;
    if (grow1) { c->cap = 0; for (Slot& s : c->slots) dfree(s.keys); }
    if (grow2) { c->cap2 = 0; for (Slot& s : c->slots) dfree(s.keys2); }
    for (int k = 0; k < slots_in_use(c); ++k) {
        Slot& s = c->slots[k];
        hipError_t e = hipSuccess;
        if (grow1) e = dmalloc(c, &s.keys, sizeof(unsigned long long) * want);
        if (e == hipSuccess && grow2) e = dmalloc(c, &s.keys2, sizeof(unsigned long long) * want2);
        if (e != hipSuccess) {
            // (all or nothing: a half-made set would leave slots without buffers behind capacities that say otherwise)
            for (Slot& t : c->slots) { dfree(t.keys); dfree(t.keys2); }
            c->cap = 0; c->cap2 = 0;
            return fail(c, SPLAT_ERR_CAPACITY, std::string("cannot allocate pair buffer: ") + hipGetErrorString(e));
        }
    }
    c->cap = want; c->cap2 = want2;
    return SPLAT_OK;
}

uint64_t default_pair_capacity(const splat_ctx* c) {
    return c->cfg.pair_capacity ? c->cfg.pair_capacity : std::max<uint64_t>(1ull << 22, 16 * c->n);
}
// One-pass binning: four times that while the key buffers of all frame slots together stay under 8 GiB, twice under 64 GiB (C5 --
// 6 M Gaussians at 4K -- would take 24 GB at 64 N for nothing measurable at rest).
// What the tiles' regions do not ask for is handed out to them as room to grow (build_layout), and that room is what a moving
// camera lives on: with 16 N entries the trained-like surface scene's lists (10-13.5 M pairs, regions asking for 1.5 x + 512 a
// tile) left none.  C3s at 3 / 10 degrees a frame: 1450 / 1340 frames/s with 16 N, 1650 / 1340 with 32 N, 1725 / 1460 with 64 N
// (frames no longer binned twice); 36 uncorrelated synchronous poses 602 -> 811 -> 825 frames/s.  2.3 -> 6.1 GB of device
// memory on C3, 14.8 -> 15.4 GB on C5 (32 N), of 288.
int slots_in_use(const splat_ctx* c);
uint64_t region_capacity_for(const splat_ctx* c, uint64_t mult) { return std::max<uint64_t>(1ull << 22, mult * c->n); }
// The second key buffer of a one-pass frame slot: room for the lists of more than 2048 keys (the scan hands it out).  12
// entries per Gaussian hold every frame measured (C3's bench pose asks for 5.3 M of 18 M, the surface scene from inside
// for 15 M); a frame that asks for more is skipped once and the buffer grown (finish_frame).
uint64_t default_keys2_capacity(const splat_ctx* c) {
    if (c->knobs.dbg_keys2_entries) return c->knobs.dbg_keys2_entries;      // (tests force the growth path)
    return std::max<uint64_t>(1ull << 22, 12 * c->n);
}
uint64_t default_region_multiplier(splat_ctx* c) {
    if (c->keys_per_gaussian) return c->keys_per_gaussian;
    if (c->region_mult) return c->region_mult;             // (decided once per scene: the query below is a driver call)
    const uint64_t per_entry = 8ull * (uint64_t)slots_in_use(c);               // a key buffer in every frame slot
    const uint64_t GiB = 1ull << 30;
    // 32 entries per Gaussian (64 for scenes of up to half a million Gaussians, where it is a gigabyte: their frames are
    // short, and a frame binned twice shows), 16 beyond 32 GiB.  What the 64 of round 5 bought a camera in motion -- fewer
    // frames binned twice -- the count-first frames of round 6 give it without the room (profiles/r06_motion_probe.txt),
    // for 1.5 GB less on C3.
    uint64_t mult = 16;
    if (64 * c->n * per_entry <= 1 * GiB) mult = 64;
    else if (32 * c->n * per_entry <= 32 * GiB) mult = 32;
    // ... and never more than a quarter of what the device has free right now: a GPU shared with other contexts (eight slab
    // ranks on one device, a host application's own allocations) takes the smaller buffer at once instead of failing the large one
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b != 0)
        while (mult > 16 && region_capacity_for(c, mult) * per_entry > (uint64_t)free_b / 4u) mult >>= 1;
    else (void)hipGetLastError();
    c->region_mult = mult;
    return mult;
}

// One-pass binning (per-tile regions of the key buffer)?  Not when switched off, when the caller fixed the pair capacity
// (exactly sized lists then), or after the key buffers failed to fit.
bool one_pass_wanted(const splat_ctx* c, unsigned int m) {
    return c->use_buckets && !c->cfg.pair_capacity && m != 0 && !c->bucket_failed;
}
constexpr uint64_t KEY_ENTRIES_MAX = 0xfff00000ull - 65536ull;      // (cursors are 32 bits and run past their regions' ends on a frame that overflows: headroom)

int build_frame_const(splat_ctx* c, const splat_camera* cam, FrameConst* fc, unsigned int* n_tiles) {
    if (!cam) return fail(c, SPLAT_ERR_INVALID, "camera is NULL");
    if (!(cam->w >= 1.0f) || !(cam->h >= 1.0f) || cam->w > 65535.0f || cam->h > 65535.0f ||
        cam->w != std::floor(cam->w) || cam->h != std::floor(cam->h))
        return fail(c, SPLAT_ERR_INVALID, "camera w/h must be integers in [1, 65535]");
    std::memcpy(fc->view, cam->view, sizeof fc->view);
    std::memcpy(fc->proj, cam->proj, sizeof fc->proj);
    fc->w = cam->w; fc->h = cam->h;
    fc->htanx = cam->htanx; fc->htany = cam->htany; fc->focal = cam->focal;
    fc->cam[0] = cam->cam_pos[0]; fc->cam[1] = cam->cam_pos[1]; fc->cam[2] = cam->cam_pos[2];
    fc->lowpass = cam->lowpass; fc->sh_dim = cam->sh_dim;
    fc->y_up = c->cfg.y_up; fc->sample_half = c->cfg.sample_half; fc->zclip = c->cfg.zclip;
    fc->zmin = c->cfg.zmin; fc->zmax = c->cfg.zmax;
    fc->early_eps = c->early_eps; fc->early_min = c->early_min; fc->prio_len = c->prio_len;
    fc->close_width = (c->cfg.mode & SPLAT_MODE_FAST) ? c->fast_width : 0.0f;
    fc->early_scan8 = c->early_scan8;
    fc->bucket_cap = 0;
    fc->corrected = (c->cfg.mode & SPLAT_MODE_CORRECTED_PROJECTION) ? 1 : 0;
    fc->redo_only = 0;
    fc->large_tiles = std::max(0, c->large_tiles);
    fc->start_hints = 0;
    fc->start_light = 0;
    // (a singular cov2d needs lowpass == 0 or a non-PSD cov3d; with lowpass == 0 every Gaussian is
    // looked at so that n_singular stays what the reference would have panicked on)
    fc->cull_blocks = (c->cull_blocks && c->bounds && cam->lowpass > 0.0f) ? 1 : 0;
    fc->W = (int)cam->w; fc->H = (int)cam->h;
    fc->tiles_x = (fc->W + TILE - 1) / TILE;
    int tiles_y = (fc->H + TILE - 1) / TILE;
    int r0 = std::max(0, c->slab0), r1 = (c->slab1 < 0) ? tiles_y : std::min(c->slab1, tiles_y);
    if (r1 < r0) r1 = r0;
    fc->tile_row0 = r0; fc->n_tile_rows = r1 - r0;
    fc->row_px0 = r0 * TILE; fc->row_px1 = std::min(r1 * TILE, fc->H);
    *n_tiles = (unsigned int)(fc->tiles_x * fc->n_tile_rows);
    return SPLAT_OK;
}

// Enqueue one frame.  Never blocks the host unless the event ring wraps onto a frame that is
// still running (32 frames behind).
// `timed`: record the per-kernel timing events.  Every hipEventRecord is a barrier packet that
// drains its queue for a few microseconds -- nine of them per frame were ~25 us of bubbles in a
// 590 us frame -- so untimed frames (all but every `timing_every`-th of an asynchronous run) record
// only the one event that tells the host the frame's status has arrived.
int enqueue_frame(splat_ctx* c, uint32_t* d_argb, bool timed, bool want_iters = false, bool may_overlap = false, bool awaited = false) {
    use_launch_knobs(&c->knobs);
    const int r = c->ring_next;
    EvSet& ev = c->ring[r];
    auto mark = [&](int k, hipStream_t st) -> hipError_t { return timed ? hipEventRecord(ev.e[k], st) : hipSuccess; };
    c->ring_next = (c->ring_next + 1) % EV_RING;
    harvest(c, r);
    std::memset(&c->h_status[r], 0, sizeof(FrameStatus));      // (this frame's scan fills it; until then it says nothing: see the peek below)
    const int si = (int)(c->frame_idx++ % (uint64_t)slots_in_use(c));
    Slot& s = c->slots[si];
    FrameStatus* const d_st = c->d_status_ring + r;       // (initialised by this frame's scan)
    // pipeline depth 1: everything on the caller's stream.  2: K1..K3 of frame N+1 on the bin stream
    // under the compositor of frame N.  3: K1 + scan of frame N+2 on the bin stream, K2 + K3 of
    // frame N+1 on the sort stream, compositor of frame N on the caller's stream -- the bin chain is
    // the longest of the three under contention, so splitting it raises the frame rate.
    hipStream_t bs = c->pipeline ? c->bin_stream : c->stream;
    // 6 (default): four slots, and the bin + sort chains of consecutive frames alternate between two streams, so
    // the chain of frame N+2 (a latency chain: K1 -> scan -> sort) runs beside the chain of frame N+1 and the
    // compositor of frame N.  With one chain at a time the frame time IS the chain's length under contention
    // (C3: 0.28 + 0.03 + 0.11 = 0.42 ms against a compositor of 0.37); two in flight leave the compositor as the
    // bound: C3 2241 -> 2334 fps, C1 17.4 k -> 19.2 k, C5 +1 %, C2 -1 %
    if (c->pipeline >= 6 && (c->frame_idx & 1ull)) bs = c->sort_stream;
    hipStream_t ss = c->pipeline == 3 ? c->sort_stream : bs;     // (4, 5: three / four slots on two streams)
    // A frame the caller waits for, with nothing else in flight (the reference's loop: one synchronous frame per pose,
    // src/main.rs:69-78), has nothing to overlap with: its whole chain goes on the caller's stream, in order -- no event
    // recorded on one stream and waited for on another between its binning and its compositor (two barrier packets and a
    // cross-queue hand-over: ~15 us of a 0.6-ms frame).
    const unsigned int m = c->n_tiles;
    // THE FRAME'S DECISIONS (include/splat_policy.h): everything below this call only launches what it says.
    splat_policy_decision pd;
    {
        splat_policy_knobs pk;
        pk.start_hints = c->start_hints; pk.count_first = c->count_first; pk.overflow_redo = c->overflow_redo; pk.early_min = c->early_min;
        pk.early_eps = c->early_eps; pk.near_cap = c->near_cap; pk.fused_sort_max = c->fused_sort_max; pk.sort_in_comp = c->sort_in_comp;
        pk.pair_mode = c->pair_mode; pk.pipeline = c->pipeline; pk.tight_grids = c->tight_grids ? 1 : 0;
        pk.large_list_min = c->large_tiles < 0 ? -1 : c->large_list_min; pk.layout_motion = c->layout_motion;
        splat_policy_input pi;
        std::memset(&pi, 0, sizeof pi);
        std::memcpy(pi.view, c->fc.view, sizeof pi.view); std::memcpy(pi.proj, c->fc.proj, sizeof pi.proj);
        pi.w = c->fc.w; pi.h = c->fc.h; pi.htanx = c->fc.htanx; pi.htany = c->fc.htany; pi.focal = c->fc.focal;
        std::memcpy(pi.cam, c->fc.cam, sizeof pi.cam); pi.lowpass = c->fc.lowpass;
        pi.tile_row0 = c->fc.tile_row0; pi.n_tile_rows = c->fc.n_tile_rows;
        pi.frame_idx = c->frame_idx; pi.ring_entry = r; pi.one_pass = c->fc.bucket_cap ? 1 : 0;
        pi.layout_valid = s.layout_valid ? 1 : 0; pi.layout_cam = s.layout_cam[s.flip];
        pi.awaited = awaited ? 1 : 0; pi.idle = c->idle ? 1 : 0; pi.has_keys2 = s.keys2 != nullptr ? 1 : 0; pi.n_tiles = m;
        pi.sort_hint = c->sort_hint ? 1 : 0; pi.hint_maxlen = c->hint_maxlen; pi.hint_ge2048 = c->hint_ge2048; pi.hint_ge8192 = c->hint_ge8192;
        pi.hint_ge16384 = c->hint_ge16384; pi.hint_pairs = c->hint_pairs; pi.hint_large = c->hint_large; pi.hint_window = c->hint_window;
        for (int q = 0; q < EV_RING; ++q) {      // (the scans of frames in flight write these words to the host: a peek, no wait)
            const volatile FrameStatus* hs = &c->h_status[q];
            pi.status[q].in_flight = c->ring[q].used ? 1u : 0u; pi.status[q].arrived = hs->arrived;
            pi.status[q].overflow = hs->overflow; pi.status[q].redone = hs->redone;
        }
        if (splat_policy_decide(&pk, &c->pol, &pi, &pd) != 0) return fail(c, SPLAT_ERR_INVALID, "frame policy refused its input");
        c->pol = pd.next;
    }
    // A frame the caller waits for, with nothing else in flight, has nothing to overlap with: its whole chain goes on the
    // caller's stream, in order (pd.solo)
    const bool solo = pd.solo != 0;
    if (solo) { bs = c->stream; ss = c->stream; }
    c->idle = false;
    if (c->pipeline) {
        // order this frame's binning after whatever the caller queued before the call (it may have
        // written the scene-independent inputs we read? no -- but it keeps stream semantics intact
        // for a caller that interleaves uploads), and after the compositor that last used the slot
        if (s.used && s.free_ring >= 0) HIP_TRY(c, hipStreamWaitEvent(bs, c->ring[s.free_ring].e[7], 0));
    }
#if SPLAT_K1X == 30 || SPLAT_K1X == 31
    { int rck = ensure_two_pass_buffers(c, s); if (rck != SPLAT_OK) return rck; }      // (the timeline builds stamp into depth / rect)
#endif
    if (!c->fc.bucket_cap) {                               // two-pass binning: K1 counts visible Gaussians into the status before the scan
        int rc2 = ensure_two_pass_buffers(c, s);
        if (rc2 != SPLAT_OK) return rc2;
        HIP_TRY(c, hipMemsetAsync(d_st, 0, sizeof(FrameStatus), bs));
    }
    // One-pass binning: the tiles' regions of the key buffer and their cursors.  Normally the layout_kernel of the frame
    // before this one ON THIS STREAM has left them in the slot's other copy (below); a slot without a layout (first
    // frames, a new scene / target / slab, after a frame outgrew a region) counts its pairs first -- K1 against the
    // empty layout drops every key and counts every pair -- and builds regions that fit exactly this camera.
    unsigned int *cursors = s.counts, *layout = nullptr;
    const uint64_t cam_hash = pd.cam_hash;      // (what places the Gaussians on the target: the camera and the slab)
    uint4* const large_list = pd.use_large_list ? s.large_list : nullptr;       // (no list: K1's blocks expand their close-ups themselves, and only count the large splats)
    c->fc.start_hints = pd.start_hints_mode; c->fc.start_light = pd.start_light; c->fc.early_min = pd.early_min;
    if (c->fc.bucket_cap) {
        // COUNT FIRST (pd.count_first): the frame counts its pairs per tile (K1's count flavour: geometry planes only, no SH, no
        // record, no key -- a third of a K1) and bins into regions that fit exactly ITS camera.
        const bool count_first = pd.count_first != 0;
        if (count_first) {
            const int into = s.layout_valid ? s.flip : 1;
            HIP_TRY(c, hipMemsetAsync(s.redo_cursors, 0, sizeof(unsigned int) * ((size_t)m + 1), bs));     // (the redo's buffer: a count-first frame has no redo)
            launch_preprocess(bs, c->n, c->planes, c->orig, c->fc, s.recs, s.depth, s.rect, s.redo_cursors, s.vislist, s.keys, c->bounds, s.blockinfo, d_st, c->zero_layout, true,
                              large_list, s.large_count);
            launch_bin_large(bs, c->fc, large_list, s.large_count, (unsigned int)c->n, s.redo_cursors, s.keys, d_st, true);
            launch_layout(bs, m, s.redo_cursors, c->zero_layout, into ? s.lay_b : s.lay_a, into ? s.counts_b : s.counts, c->fc.bucket_cap, nullptr, nullptr, c->region_spare,
                          nullptr, s.large_count);
            s.flip = into; s.layout_valid = true;
            s.layout_cam[into] = cam_hash;
        }
        cursors = s.flip ? s.counts_b : s.counts;
        layout = s.flip ? s.lay_b : s.lay_a;
    }
    HIP_TRY(c, mark(0, bs));
    launch_preprocess(bs, c->n, c->planes, c->orig, c->fc, s.recs, s.depth, s.rect, cursors, s.vislist, s.keys, c->bounds, s.blockinfo, d_st, layout, false,
                      large_list, s.large_count);
    if (c->fc.bucket_cap) launch_bin_large(bs, c->fc, large_list, s.large_count, (unsigned int)c->n, cursors, s.keys, d_st, false);     // (the large splats K1 listed, tile by tile)
    HIP_TRY(c, mark(1, bs));
    c->grid_big = pd.grid_big; c->grid_mid = pd.grid_mid; c->grid_long = pd.grid_long;      // (what the sort launches cover; the scan validates)
    // (one-pass binning: a second workgroup of the scan's launch builds the regions of the NEXT frame on this binning
    // stream -- two frames on with two chains in flight -- from this frame's lists, into that slot's idle copy)
    unsigned int *next_layout = nullptr, *next_counts = nullptr;
    if (c->fc.bucket_cap) {
        const int stride = (c->pipeline >= 6) ? 2 : 1;
        Slot& nx = c->slots[(si + stride) % slots_in_use(c)];
        const int into = (&nx == &s) ? (s.flip ^ 1) : (nx.layout_valid ? (nx.flip ^ 1) : 1);
        next_layout = into ? nx.lay_b : nx.lay_a; next_counts = into ? nx.counts_b : nx.counts;
        nx.flip = into; nx.layout_valid = true;
        nx.layout_cam[into] = cam_hash;
    }
    const bool comp_sorts_frame = pd.comp_sorts != 0;
    const unsigned int near_cap = pd.near_cap;
    unsigned int* const off2 = c->fc.bucket_cap ? s.off2 : nullptr;        // (two-pass binning: the second buffer mirrors the first)
    launch_scan(bs, m, cursors, s.offsets, s.cursor, s.order, s.lens, d_st, c->cap, c->fc.bucket_cap, c->grid_big, c->grid_mid, c->grid_long, &c->h_status[r], layout,
                next_layout, next_counts, c->region_spare, false, off2, (unsigned int)std::min<uint64_t>(c->cap2, 0xffffffffull), s.large_count,
                (unsigned int)c->fc.tiles_x, (unsigned int)pd.layout_radius);
    const bool redo = pd.redo != 0;
    if (redo) {
        // OVERFLOW REDO.  The regions this frame was binned into were sized for another camera (two frames back on a moving
        // path): if the scan above found a list beyond its region, the frame is binned again right here -- regions that
        // fit exactly this camera (from the counts the first pass left), K1, scan -- into copies of their own; if not
        // (the usual case) every one of these launches reads one word and leaves.  Either way the kernels behind see a
        // complete frame: nothing is skipped, nothing to report, nothing for the caller to render again.
        // (No count pass: K1's reservations keep counting past a region's end -- cursor minus region start IS the tile's exact
        // pair count, overflowed or not (the scan's `raw`) -- and the scan leaves cursors and regions alone: the regions that
        // fit this camera are built straight from them.  One K1 where round 5 ran two.)
        FrameConst fr = c->fc;
        fr.redo_only = 1;
        launch_layout(bs, m, cursors, layout, s.redo_layout, s.redo_cursors, c->fc.bucket_cap, nullptr, nullptr, c->region_spare, d_st);
        launch_preprocess(bs, c->n, c->planes, c->orig, fr, s.recs, s.depth, s.rect, s.redo_cursors, s.vislist, s.keys, c->bounds, s.blockinfo, d_st, s.redo_layout, false,
                          large_list, s.large_count);
        launch_bin_large(bs, fr, large_list, s.large_count, (unsigned int)c->n, s.redo_cursors, s.keys, d_st, false);
        launch_scan(bs, m, s.redo_cursors, s.offsets, s.cursor, s.order, s.lens, d_st, c->cap, c->fc.bucket_cap, c->grid_big, c->grid_mid, c->grid_long, &c->h_status[r], s.redo_layout,
                    nullptr, nullptr, c->region_spare, true, off2, (unsigned int)std::min<uint64_t>(c->cap2, 0xffffffffull), s.large_count);
    }
    HIP_TRY(c, mark(2, bs));
    if (ss != bs) {
        HIP_TRY(c, hipEventRecord(s.ev_binned, bs));
        HIP_TRY(c, hipStreamWaitEvent(ss, s.ev_binned, 0));
    }
    HIP_TRY(c, mark(8, ss));
    if (!c->fc.bucket_cap)      // one-pass binning placed the keys in K1
        launch_emit(ss, c->n, c->fc, s.depth, s.rect, c->orig, s.vislist, s.cursor, s.keys, d_st);
    HIP_TRY(c, mark(3, ss));
    const bool comp_sorts = comp_sorts_frame;
    if (near_cap) {     // near selection: the nearest keys of the long lists instead of the sort launches
        // Its workgroups: an eighth as many as tiles, each walking the longest-first order with that stride until the lists get
        // short -- a frame in the pipeline has 0.3 ms of slack in front of its compositor, and fewer resident selections leave
        // the previous frame's compositor its LDS (C3 +2 %, C3s +4 %; a sixteenth: C3 +1 %, C3s -1 %; a 64th: -15 %).  A frame
        // the caller waits for gets a workgroup per long list (their number a frame ago, plus an eighth).
        launch_select(ss, m, s.offsets, s.order, s.lens, s.keys, s.keys2, d_st, c->orig, near_cap, c->need_hint, s.near_m,
                      (unsigned int)c->fc.tiles_x, (unsigned int)c->fc.n_tile_rows, c->one_pass_select ? c->need_hint + 4u * (size_t)c->m_alloc : nullptr,
                      pd.select_grid, c->fc.start_hints == 1, off2, pd.hint_radius);
    }
    else if (!comp_sorts)
        launch_sort(ss, m, c->grid_big, c->grid_mid, c->grid_long, s.offsets, s.order, s.lens, s.keys, s.keys2, d_st, c->orig, c->fused_sort_max, off2);
    HIP_TRY(c, mark(4, ss));
    // Which lane composites?  Frames to one image stay on one lane (stream order is their write-after-write / in-out
    // order, as always); a frame to another image takes the other lane when it may.  A frame that must not overlap
    // (synchronous, statistics, overlap off) goes to lane 0 behind everything lane 1 still holds.
    int li = 0;
    const char* const img_lo = reinterpret_cast<const char*>(d_argb);
    const char* const img_hi = img_lo + (size_t)c->fc.W * (size_t)c->fc.H * 4u;
    if (c->comp2) {
        auto touches = [&](const splat_ctx::ImgRec& e) { return e.seq != 0 && img_lo < e.hi && e.lo < img_hi; };
        const splat_ctx::ImgRec* newest = nullptr;        // the most recent frame to (a part of) this image
        for (const auto& e : c->img_tab)
            if (touches(e) && (!newest || e.seq > newest->seq)) newest = &e;
        if (!may_overlap) li = 0;
        else if (newest) li = newest->lane;
        else li = c->lane[0].seq <= c->lane[1].seq ? 0 : 1;            // an image nobody holds: the lane idle for longer
        hipStream_t mine = li ? c->comp2 : c->stream;
        // Frames to this image on the OTHER lane come first.  (Ring events: when the ring wraps onto an entry the host
        // has waited for that frame -- harvest -- so a re-recorded event only ever stands for a LATER frame.)
        for (const auto& e : c->img_tab)
            if (touches(e) && e.lane != li && e.ring >= 0) HIP_TRY(c, hipStreamWaitEvent(mine, c->ring[e.ring].e[7], 0));
        if (!may_overlap && li == 0 && c->lane[1].seq != 0 && c->lane[1].ring >= 0)
            HIP_TRY(c, hipStreamWaitEvent(mine, c->ring[c->lane[1].ring].e[7], 0));
        // this frame's entry: the image's own one, else a free one, else the oldest -- whose frame this one then follows
        // (an image that drops out of the table must not be in flight any more)
        splat_ctx::ImgRec* slot = nullptr;
        for (auto& e : c->img_tab) if (e.seq != 0 && e.lo == img_lo && e.hi == img_hi) { slot = &e; break; }
        if (!slot) for (auto& e : c->img_tab) if (e.seq == 0) { slot = &e; break; }
        if (!slot) {
            for (auto& e : c->img_tab) if (!slot || e.seq < slot->seq) slot = &e;
            if (slot->ring >= 0) {      // (both lanes: whichever lane the forgotten image's next frame takes, it follows)
                HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ring[slot->ring].e[7], 0));
                HIP_TRY(c, hipStreamWaitEvent(c->comp2, c->ring[slot->ring].e[7], 0));
            }
        }
        slot->lo = img_lo; slot->hi = img_hi; slot->seq = c->lane_seq + 1; slot->ring = r; slot->lane = li;
    }
    hipStream_t cs = li ? c->comp2 : c->stream;
    if (c->pre_wait) { HIP_TRY(c, hipStreamWaitEvent(cs, c->pre_wait, 0)); c->pre_wait = nullptr; }
    if (c->pipeline && ss != cs) {
        HIP_TRY(c, hipEventRecord(s.ev_ready, ss));
        HIP_TRY(c, hipStreamWaitEvent(cs, s.ev_ready, 0));
    }
    HIP_TRY(c, mark(5, cs));
    uint2* iters = nullptr;
    c->iters_valid = false;
    if (want_iters) {       // statistics frame: every compositor wave leaves its iteration counts (plain stores)
        if (c->iters_alloc < m * 4u) {
            dfree(c->d_iters); c->iters_alloc = 0;
            HIP_TRY(c, dmalloc(c, &c->d_iters, sizeof(uint2) * (size_t)m * 4u));
            c->iters_alloc = m * 4u;
        }
        HIP_TRY(c, hipMemsetAsync(c->d_iters, 0, sizeof(uint2) * (size_t)m * 4u, cs));
        iters = c->d_iters;
        c->iters_valid = true;
    }
    // throughput-bound frames (many pairs per key of the longest list: C3 737, C5 3100) keep the one-record walk, the
    // others (C2 316, an eighth-of-a-frame slab 92, C1 36) take the paired one; measured crossover between 316 and 737
    const bool pair_walk = pd.pair_walk != 0;
    launch_composite(cs, m, c->fc, s.offsets, s.order, s.lens, s.keys, s.recs, d_argb, d_st, c->orig, c->fused_sort_max, iters, want_iters,
                     pair_walk, (c->cfg.mode & SPLAT_MODE_LIBM_EXP) != 0, c->clear_first, comp_sorts ? s.keys2 : nullptr, near_cap ? s.near_m : nullptr,
                     c->need_hint, c->need_hint ? c->need_hint + 5u * (size_t)c->m_alloc : nullptr, off2);
    c->last_near = near_cap != 0u;
    HIP_TRY(c, mark(6, cs));
    // the scan has already delivered this frame's status to h_status[r]; a statistics frame refreshes it with the late
    // counters (compositor retries, sort fallbacks)
    if (want_iters) HIP_TRY(c, hipMemcpyAsync(&c->h_status[r], d_st, sizeof(FrameStatus), hipMemcpyDeviceToHost, cs));
    HIP_TRY(c, hipEventRecord(ev.e[7], cs));
    s.free_ring = r;
    HIP_TRY(c, hipGetLastError());
    s.used = true;
    ev.used = true;
    ev.timed = timed;
    c->last_ring = r;
    c->last_slot = si;
    c->lane[li].seq = ++c->lane_seq; c->lane[li].ring = r;
    c->last_lane = li;
    c->last_lists_in_memory = want_iters || c->fused_sort_max == 0;
    return SPLAT_OK;
}

// Wait for everything enqueued and harvest every frame's status.  Returns SPLAT_ERR_CAPACITY when at
// least one harvested frame was skipped on the device because it outgrew its storage (pair buffer, tile
// bucket, sort launch sizes): the storage / hints have then been grown for the next attempt.
// *last_skipped tells whether the MOST RECENT frame is among the skipped ones -- only that one may be
// redone by a synchronous caller (an older asynchronous frame that was lost must not make the current,
// complete frame blend a second time onto the in/out image).
int finish_frame(splat_ctx* c, bool* last_skipped = nullptr) {
    if (last_skipped) *last_skipped = false;
    int rc = sync_all(c);
    if (rc != SPLAT_OK) return rc;
    c->idle = true;                          // every stream of the context has drained
    if (c->last_ring >= 0) c->last = c->h_status[c->last_ring];
    if (last_skipped) *last_skipped = c->last_ring >= 0 && c->last.overflow != 0;
    for (int k = 0; k < EV_RING; ++k) harvest(c, k);
    const char* msg = nullptr;
    if (c->sort_grid_miss) {            // the hint has been refreshed from the frame that missed
        c->sort_grid_miss = false;
        msg = "more long tile lists than the sort launches covered; frame must be re-rendered";
    }
    if (c->bucket_overflow || c->layout_want) {
        // A tile's list outgrew the region it had been given (sized from a frame two back: the camera moved, or nothing
        // was known yet), or the regions did not fit the key buffer.  The slots' layouts are dropped -- the next frame
        // counts its pairs first (bootstrap, enqueue_frame) and gets regions that fit it -- and the buffer grows to what
        // the layout asked for.
        const bool overflowed = c->bucket_overflow;
        c->bucket_overflow = false;
        for (Slot& sl : c->slots) sl.layout_valid = false;
        if (c->layout_want > c->cap) {
            const uint64_t asked = c->layout_want + c->layout_want / 4 + 1024, want = std::min<uint64_t>(asked, KEY_ENTRIES_MAX);
            const uint64_t bytes = (want + c->cap2) * 8ull * (uint64_t)slots_in_use(c);
            if (bytes > c->bucket_bytes || asked > KEY_ENTRIES_MAX) c->bucket_failed = true;    // no room: exactly sized lists instead
            else {
                rc = ensure_keys(c, want, c->cap2);
                if (rc == SPLAT_ERR_CAPACITY) { c->bucket_failed = true; rc = SPLAT_OK; }
                else if (rc != SPLAT_OK) return rc;
            }
        }
        c->layout_want = 0;
        if (overflowed) msg = "a tile outgrew its region of the key buffer; regions rebuilt, frame must be re-rendered";
    }
    if (c->overflow_want) {
        uint64_t want = (uint64_t)((double)c->overflow_want * 1.25) + 1024;
        c->overflow_want = 0;
        rc = ensure_keys(c, want, want);
        if (rc != SPLAT_OK) return rc;
        msg = "pair buffer overflowed; capacity grown, frame must be re-rendered";
    }
    if (c->keys2_want) {
        // the lists of more than 2048 keys asked for more room in the second key buffer than it has (a camera deep inside a
        // dense scene): a quarter more than they asked for
        const uint64_t want2 = std::min<uint64_t>(c->keys2_want + c->keys2_want / 4 + 1024, KEY_ENTRIES_MAX);
        c->keys2_want = 0;
        rc = ensure_keys(c, c->cap, want2);
        if (rc != SPLAT_OK) return rc;
        msg = "the long tile lists outgrew the second key buffer; buffer grown, frame must be re-rendered";
    }
    return msg ? fail(c, SPLAT_ERR_CAPACITY, msg) : SPLAT_OK;
}

// For the entry points that only wait (splat_sync, splat_get_*, ...): a skipped frame -- found now or left
// pending by a synchronous render that was itself complete -- is reported once.
int finish_and_report(splat_ctx* c) {
    int rc = finish_frame(c);
    c->frames_drop_reported = c->frames_dropped;
    if (rc == SPLAT_OK && c->deferred_drop) {
        c->deferred_drop = false;
        return fail(c, SPLAT_ERR_CAPACITY, "an earlier asynchronous frame was skipped on the device (storage has been grown); render it again");
    }
    if (rc == SPLAT_ERR_CAPACITY) c->deferred_drop = false;
    return rc;
}

// ... and for the ones that must quiesce the device on the way to something else (re-allocation, a count-only
// pass): a loss found here stays pending for the next report.
int finish_quiet(splat_ctx* c) {
    int rc = finish_frame(c);
    if (rc == SPLAT_ERR_CAPACITY) { c->deferred_drop = true; rc = SPLAT_OK; }
    return rc;
}

int slots_in_use(const splat_ctx* c) { return c->pipeline >= 5 ? 4 : (c->pipeline >= 3 ? 3 : (c->pipeline ? 2 : 1)); }

// Pick the binning path for a frame over m tiles and make sure its key storage exists.
int prepare_binning(splat_ctx* c, unsigned int m, FrameConst* fc) {
    const bool one_pass = one_pass_wanted(c, m);
    if (one_pass != c->last_one_pass || m != c->layout_m) {
        // another path or another tile grid: the counters hold the other path's state (cursors / zeros), the regions
        // describe another grid
        int rc = finish_quiet(c);
        if (rc != SPLAT_OK) return rc;
        for (Slot& sl : c->slots) {
            sl.layout_valid = false; sl.flip = 0;
            if (sl.counts) HIP_TRY(c, hipMemsetAsync(sl.counts, 0, sizeof(unsigned int) * (size_t)(m + 1), c->stream));
        }
        if (c->need_hint) HIP_TRY(c, hipMemsetAsync(c->need_hint, 0, sizeof(unsigned int) * 9u * (size_t)c->m_alloc, c->stream));   // another grid: another tile under every index
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->last_one_pass = one_pass; c->layout_m = m;
        reset_policy(c);
    }
    fc->bucket_cap = 0;
    if (one_pass) {
        // key buffer: what the two-pass path would start with, unless a layout has asked for more (finish_frame grows it)
        // (always with the second buffer: a region may hold a list of any length, and a list beyond 16384 keys sorts as
        // runs merged through it -- fixed-stride buckets could cap the lists at what the buffers at hand could sort)
        // The buffer that does not fit (the byte budget, or the allocation itself fails: a device with less free memory than
        // the default assumes) is tried again at half the size, down to the 16 entries per Gaussian the two-pass path would
        // start with, before one-pass binning is given up for the scene.
        int rc = SPLAT_ERR_CAPACITY;
        for (uint64_t mult = default_region_multiplier(c); ; mult >>= 1) {
            const uint64_t want = std::min<uint64_t>(std::max<uint64_t>(c->cap, region_capacity_for(c, mult)), KEY_ENTRIES_MAX);
            const uint64_t want2 = std::min<uint64_t>(std::max<uint64_t>(c->cap2, default_keys2_capacity(c)), KEY_ENTRIES_MAX);
            rc = ((want + want2) * 8ull * (uint64_t)slots_in_use(c) > c->bucket_bytes) ? SPLAT_ERR_CAPACITY : ensure_keys(c, want, want2);
            if (rc != SPLAT_ERR_CAPACITY || mult <= 16 || want <= c->cap) break;
            (void)hipGetLastError();
        }
        if (rc == SPLAT_OK) { fc->bucket_cap = (unsigned int)std::min<uint64_t>(c->cap, KEY_ENTRIES_MAX); return rc; }
        if (rc != SPLAT_ERR_CAPACITY) return rc;
        c->bucket_failed = true;                          // no room: exactly sized lists instead
        return prepare_binning(c, m, fc);
    }
    // two-pass binning: exactly sized lists, and a second buffer that mirrors the first (the merges and global-memory radix
    // passes of its longest lists address it through the lists' own offsets)
    const uint64_t want = std::max<uint64_t>(default_pair_capacity(c), c->cap);
    return ensure_keys(c, want, want);
}

void fill_stats(splat_ctx* c, splat_stats* st) {
    std::memset(st, 0, sizeof *st);
    st->n_gaussians = c->n;
    st->n_visible = c->last.n_visible;
    st->n_singular = c->last.n_singular;
    st->n_pairs = c->last.n_pairs;
    st->max_tile_len = c->last.max_tile_len;
    st->n_fallback = c->last.n_fallback;
    st->n_sort_fallback = c->last.n_sort_fallback;
    st->n_iter_scan = 0; st->n_iter_blend = 0;
    st->n_near_tiles = c->last.n_near_tiles; st->n_near_fallback = c->last.n_near_fallback;
    st->n_blocks_culled = 0;
    const unsigned int* binfo = c->last_slot >= 0 ? c->slots[c->last_slot].blockinfo : nullptr;
    if (binfo && c->last_ring >= 0 && (c->fc.cull_blocks || c->fc.bucket_cap)) {   // the frame has finished: sum its block words
        std::vector<unsigned int> f((c->n + 255) / 256);
        if (hipMemcpy(f.data(), binfo, f.size() * sizeof(unsigned int), hipMemcpyDeviceToHost) == hipSuccess) {
            uint64_t vis = 0, sing = 0;
            for (unsigned int v : f) {
                if (v & 0x80000000u) { st->n_blocks_culled++; continue; }
                vis += v & 0x1ffu; sing += (v >> 9) & 0x1ffu;
            }
            if (c->fc.bucket_cap) { st->n_visible = vis; st->n_singular = sing; }     // (two-pass binning counts on the device)
        }
    }
    // (one-pass binning counts the visible Gaussians in the block words summed above, not on the device)
    st->bytes_algorithmic = c->n * 148ull + st->n_visible * 48ull + c->last.n_pairs * 60ull +
                            (uint64_t)c->fc.W * (uint64_t)(c->fc.row_px1 - c->fc.row_px0) * 4ull;
    // flops_alg = sum over pixels of (its tile's list length) * 25, and the compositor's
    // measured (wave, record) iterations -- left by every wave of a frame rendered with a stats pointer
    if (c->last_slot >= 0 && c->last_ring >= 0 && c->n_tiles) {
        const Slot& sl = c->slots[c->last_slot];
        const size_t m = c->n_tiles;
        std::vector<unsigned int> len(m);
        if (c->last.overflow == 0 && hipMemcpy(len.data(), sl.lens, sizeof(unsigned int) * m, hipMemcpyDeviceToHost) == hipSuccess) {
            uint64_t f = 0;
            for (size_t t = 0; t < m; ++t) {
                const int tx = (int)(t % (size_t)c->fc.tiles_x), ty = (int)(t / (size_t)c->fc.tiles_x) + c->fc.tile_row0;
                const int pw = std::min(TILE, c->fc.W - tx * TILE), ph = std::min(TILE, std::min(c->fc.H, c->fc.row_px1) - ty * TILE);
                if (pw > 0 && ph > 0) f += (uint64_t)len[t] * (uint64_t)(pw * ph);
            }
            st->flops_algorithmic = f * 25ull;
        }
        if (c->iters_valid && c->d_iters) {
            std::vector<uint2> it(m * 4u);
            if (hipMemcpy(it.data(), c->d_iters, sizeof(uint2) * it.size(), hipMemcpyDeviceToHost) == hipSuccess)
            {
                if (c->knobs.dbg_starts) {
                    // (debug: the words are (list length, nearest keys the wave's walk needed) -- how much of every list had to
                    // be in order at all, by list length class)
                    const unsigned int edges[6] = {0u, 768u, 2048u, 8192u, 16384u, 0xffffffffu};
                    for (int cls = 0; cls < 5; ++cls) {
                        uint64_t tiles = 0, keys = 0, need = 0, need2k = 0, full = 0;
                        for (size_t t = 0; t < m; ++t) {
                            const unsigned int len = std::max(std::max(it[4 * t].x, it[4 * t + 1].x), std::max(it[4 * t + 2].x, it[4 * t + 3].x));
                            if (len == 0 || len <= edges[cls] || len > edges[cls + 1]) continue;
                            unsigned int nd = 0;
                            for (int w = 0; w < 4; ++w) nd = std::max(nd, it[4 * t + w].y);
                            ++tiles; keys += len; need += nd; need2k += nd <= 2048u ? 1u : 0u; full += nd >= len ? 1u : 0u;
                        }
                        std::fprintf(stderr, "lists of %u < len <= %u keys: %llu tiles, %llu keys, deepest wave start needs %llu of them (%.1f %%); "
                                     "%llu tiles need <= 2048 keys, %llu need the whole list\n", edges[cls], edges[cls + 1],
                                     (unsigned long long)tiles, (unsigned long long)keys, (unsigned long long)need, keys ? 100.0 * need / keys : 0.0,
                                     (unsigned long long)need2k, (unsigned long long)full);
                    }
                } else
                for (const uint2& v : it) { st->n_iter_scan += v.x; st->n_iter_blend += v.y; }
                if (std::getenv("SPLAT_DBG_IMBALANCE")) {      // how much of a tile's four wave slots its slowest wave leaves idle
                    double sum = 0, held = 0;
                    for (size_t t = 0; t < m; ++t) {
                        double mx = 0;
                        for (int w = 0; w < 4; ++w) { const double cst = 14.0 * it[4 * t + w].x + 50.0 * it[4 * t + w].y + 300.0; sum += cst; mx = std::max(mx, cst); }
                        held += 4.0 * mx;
                    }
                    std::fprintf(stderr, "compositor wave imbalance: sum of wave costs / (4 x slowest wave per tile) = %.3f\n", held > 0 ? sum / held : 1.0);
                    // ... and what the early-out scan bought: waves that scanned, split by whether the exact walk then took
                    // about as many steps as the scan (it found its start) or at least twice as many (it gave up half way)
                    uint64_t sc_ok = 0, bl_ok = 0, sc_no = 0, bl_no = 0, w_ok = 0, w_no = 0, bl_none = 0;
                    for (const uint2& v : it) {
                        if (v.x == 0) { bl_none += v.y; continue; }
                        if ((uint64_t)v.y * 10u >= (uint64_t)v.x * 18u) { sc_no += v.x; bl_no += v.y; ++w_no; } else { sc_ok += v.x; bl_ok += v.y; ++w_ok; }
                    }
                    std::fprintf(stderr, "early-out scan: %llu waves found a start (scan %llu, blend %llu steps), %llu did not (scan %llu, blend %llu); waves without a scan: blend %llu\n",
                                 (unsigned long long)w_ok, (unsigned long long)sc_ok, (unsigned long long)bl_ok, (unsigned long long)w_no, (unsigned long long)sc_no,
                                 (unsigned long long)bl_no, (unsigned long long)bl_none);
                }
            }
        }
    }
    float t[N_TIMES] = {0};
    if (c->last_ring >= 0 && c->ring[c->last_ring].timed) ev_times(c->ring[c->last_ring], t);   // events stay valid after harvest
    st->ms_preprocess = t[0]; st->ms_scan = t[1]; st->ms_emit = t[2]; st->ms_sort = t[3]; st->ms_composite = t[4];
    st->ms_total = t[0] + t[1] + t[2] + t[3] + t[4];
}

void free_scene(splat_ctx* c) {
    dfree(c->planes); dfree(c->orig); dfree(c->bounds);
    for (Slot& s : c->slots) { dfree(s.recs); dfree(s.depth); dfree(s.rect); dfree(s.vislist); dfree(s.blockinfo); dfree(s.large_list); dfree(s.large_count); s.used = false; }
    c->n = 0;
    c->h_orig.clear();
    c->last_slot = -1;
}

}  // namespace

namespace {
// One option, validated and stored (include/splat_hip.h SPLAT_OPT_*).  The caller has quiesced the context where that
// matters.  false: unknown option or value out of range.
// dry: validate only (splat_set_option checks the value before it touches anything, pinned options included)
bool store_option(splat_ctx* c, int opt, double v, bool dry = false) {
    if (!(v == v)) return false;
#define SPLAT_DRY_ if (dry) return true
    switch (opt) {
        case SPLAT_OPT_PIPELINE_DEPTH: {
            if (v < 1.0 || v > 6.0) return false;
            SPLAT_DRY_;
            const int p = (int)v <= 1 ? 0 : (int)v;
            if (p != c->pipeline) {
                // another number of frame slots: the key buffers exist per slot in use and are made again by the next frame
                for (Slot& sl : c->slots) { dfree(sl.keys); dfree(sl.keys2); sl.layout_valid = false; sl.flip = 0; sl.used = false; sl.free_ring = -1; }
                c->cap = 0; c->cap2 = 0; c->frame_idx = 0;
                c->pipeline = p;
                // One stream for everything has no compositor lanes: the lane state goes as splat_set_frame_overlap(1) leaves it
                // (unless the operator pinned the overlap from the environment: then only the second lane's stream is dropped, and
                // the lanes come back -- ensure_lane, at the next frame that may overlap -- when the depth goes up again).  A depth
                // raised again does NOT restore an overlap of 2 set from code: set it again.
                if (p == 0) {
                    c->comp2 = nullptr;
                    if (!(c->env_pinned & (1u << SPLAT_OPT_FRAME_OVERLAP))) c->overlap = 1;
                    c->lane[0] = splat_ctx::Lane{}; c->lane[1] = splat_ctx::Lane{};
                    for (auto& e : c->img_tab) e = splat_ctx::ImgRec{};
                    c->last_lane = 0;
                }
            }
            return true;
        }
        case SPLAT_OPT_FUSED_SORT_MAX: if (v < 0.0 || v > 2048.0) return false; SPLAT_DRY_; c->fused_sort_max = (unsigned int)v; return true;
        case SPLAT_OPT_REGION_SPARE: if (v < 1.0) return false; SPLAT_DRY_; c->region_spare = (float)v; return true;
        case SPLAT_OPT_EARLY_OUT_EPS: if (v < 0.0 || v > 1.0) return false; SPLAT_DRY_; c->early_eps = (float)v; return true;
        case SPLAT_OPT_EARLY_OUT_MIN_LIST: if (v < 0.0 || v > 1e9) return false; SPLAT_DRY_; c->early_min = (int)v; return true;
        case SPLAT_OPT_EARLY_OUT_SCAN_EIGHTHS: if (v < 1.0 || v > 8.0) return false; SPLAT_DRY_; c->early_scan8 = (int)v; return true;
        case SPLAT_OPT_SORT_IN_COMPOSITOR: if (v < -1.0 || v > 1.0) return false; SPLAT_DRY_; c->sort_in_comp = v < 0.0 ? -1 : (v != 0.0 ? 1 : 0); return true;
        case SPLAT_OPT_PAIR_WALK: if (v < -1.0 || v > 1.0) return false; SPLAT_DRY_; c->pair_mode = v < 0.0 ? -1 : (v != 0.0 ? 1 : 0); return true;
        case SPLAT_OPT_TIMING_EVERY: if (v < 1.0 || v > 1e9) return false; SPLAT_DRY_; c->timing_every = (int)v; return true;
        case SPLAT_OPT_BLOCK_CULLING: if (v != 0.0 && v != 1.0) return false; SPLAT_DRY_; c->cull_blocks = v != 0.0; return true;
        case SPLAT_OPT_ONE_PASS_BINNING: if (v != 0.0 && v != 1.0) return false; SPLAT_DRY_; c->use_buckets = v != 0.0; return true;
        case SPLAT_OPT_KEY_BUFFER_BYTES: if (v < 0.0 || v > 1.8e19) return false; SPLAT_DRY_; c->bucket_bytes = (uint64_t)v; c->bucket_failed = false; return true;
        case SPLAT_OPT_FAST_CLOSE_WIDTH: if (v != 1.0 && v != 2.0) return false; SPLAT_DRY_; c->fast_width = (float)v; return true;
        case SPLAT_OPT_PRIORITY_LIST_LEN: if (v < 1.0 || v > 1073741823.0) return false; SPLAT_DRY_; c->prio_len = (int)v; return true;
        case SPLAT_OPT_FRAME_OVERLAP: if (v != 1.0 && v != 2.0) return false; SPLAT_DRY_; c->overlap = (int)v; return true;    // (lanes: splat_set_option / splat_create make them)
        case SPLAT_OPT_NEAR_SELECT_KEYS: if (v != 0.0 && (v < 64.0 || v > 2048.0)) return false; SPLAT_DRY_; c->near_cap = (unsigned int)v; return true;
        case SPLAT_OPT_OVERFLOW_REDO: if (v != 0.0 && v != 1.0 && v != 2.0) return false; SPLAT_DRY_; c->overflow_redo = (int)v; return true;
        case SPLAT_OPT_START_HINTS: if (v != 0.0 && v != 1.0 && v != 2.0) return false; SPLAT_DRY_; c->start_hints = (int)v; return true;
        case SPLAT_OPT_HOST_ZERO_COPY: if (v != 0.0 && v != 1.0) return false; SPLAT_DRY_; c->host_zero_copy = (int)v; return true;
        case SPLAT_OPT_COUNT_FIRST: if (v != 0.0 && v != 1.0 && v != 2.0) return false; SPLAT_DRY_; c->count_first = (int)v; return true;
        case SPLAT_OPT_KEYS_PER_GAUSSIAN: if (v != 0.0 && (v < 4.0 || v > 256.0)) return false; SPLAT_DRY_; c->keys_per_gaussian = (unsigned int)v; return true;
        case SPLAT_OPT_LARGE_SPLAT_TILES: if (v < -1.0 || v > 1048576.0 || v != std::floor(v)) return false; SPLAT_DRY_; c->large_tiles = (int)v; return true;
        case SPLAT_OPT_LARGE_LIST_MIN: if (v < -1.0 || v > 1e9 || v != std::floor(v)) return false; SPLAT_DRY_; c->large_list_min = (int)v; return true;
        default: return false;
    }
#undef SPLAT_DRY_
}
bool load_option(const splat_ctx* c, int opt, double* v) {
    switch (opt) {
        case SPLAT_OPT_PIPELINE_DEPTH: *v = c->pipeline ? c->pipeline : 1; return true;
        case SPLAT_OPT_FUSED_SORT_MAX: *v = c->fused_sort_max; return true;
        case SPLAT_OPT_REGION_SPARE: *v = c->region_spare; return true;
        case SPLAT_OPT_EARLY_OUT_EPS: *v = c->early_eps; return true;
        case SPLAT_OPT_EARLY_OUT_MIN_LIST: *v = c->early_min; return true;
        case SPLAT_OPT_EARLY_OUT_SCAN_EIGHTHS: *v = c->early_scan8; return true;
        case SPLAT_OPT_SORT_IN_COMPOSITOR: *v = c->sort_in_comp; return true;
        case SPLAT_OPT_PAIR_WALK: *v = c->pair_mode; return true;
        case SPLAT_OPT_TIMING_EVERY: *v = c->timing_every; return true;
        case SPLAT_OPT_BLOCK_CULLING: *v = c->cull_blocks ? 1 : 0; return true;
        case SPLAT_OPT_ONE_PASS_BINNING: *v = c->use_buckets ? 1 : 0; return true;
        case SPLAT_OPT_KEY_BUFFER_BYTES: *v = (double)c->bucket_bytes; return true;
        case SPLAT_OPT_FAST_CLOSE_WIDTH: *v = c->fast_width; return true;
        case SPLAT_OPT_PRIORITY_LIST_LEN: *v = c->prio_len; return true;
        case SPLAT_OPT_FRAME_OVERLAP: *v = c->overlap; return true;
        case SPLAT_OPT_NEAR_SELECT_KEYS: *v = c->near_cap; return true;
        case SPLAT_OPT_OVERFLOW_REDO: *v = c->overflow_redo; return true;
        case SPLAT_OPT_START_HINTS: *v = c->start_hints; return true;
        case SPLAT_OPT_HOST_ZERO_COPY: *v = c->host_zero_copy; return true;
        case SPLAT_OPT_KEYS_PER_GAUSSIAN: *v = c->keys_per_gaussian; return true;
        case SPLAT_OPT_COUNT_FIRST: *v = c->count_first; return true;
        case SPLAT_OPT_LARGE_SPLAT_TILES: *v = c->large_tiles; return true;
        case SPLAT_OPT_LARGE_LIST_MIN: *v = c->large_list_min; return true;
        default: return false;
    }
}
// The environment's say, at splat_create: the variable of an option, when set, stores it (clamped into the option's
// range as the variables always were) and pins it.
void option_from_env(splat_ctx* c, int opt, const char* name, double lo, double hi) {
    const char* e = std::getenv(name);
    if (!e || !*e) return;
    double v = std::atof(e);
    if (!(v == v)) return;
    v = std::min(hi, std::max(lo, v));
    if (store_option(c, opt, v)) c->env_pinned |= 1u << opt;
}
}  // namespace

namespace splat {
// wait for everything this context has enqueued; a frame found skipped stays pending for the next splat_sync to report
int ctx_quiesce(splat_ctx* c) { return finish_quiet(c); }
}  // namespace splat

extern "C" {

uint32_t splat_abi_version(void) { return SPLAT_ABI_VERSION; }
uint64_t splat_stats_size(void) { return sizeof(splat_stats); }

void splat_default_config(splat_config* cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof *cfg);
    cfg->device = 0;
    cfg->mode = SPLAT_MODE_EXACT;
    cfg->y_up = 1;          // pinned: notes/screenshot.png has the orientation of the notebook render
    cfg->sample_half = 1;   // euc samples pixel centres (unpinned)
    cfg->zclip = 1; cfg->zmin = 0.0f; cfg->zmax = 1.0f;   // euc CoordinateMode::VULKAN default (unpinned)
    cfg->pair_capacity = 0;
}

int splat_create(const splat_config* cfg, splat_ctx** out) {
    if (!out) return fail(nullptr, SPLAT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    splat_config def;
    splat_default_config(&def);
    if (!cfg) cfg = &def;
    if (cfg->mode & ~(SPLAT_MODE_CORRECTED_PROJECTION | SPLAT_MODE_LIBM_EXP | SPLAT_MODE_FAST))
        return fail(nullptr, SPLAT_ERR_INVALID, "unknown mode");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, SPLAT_ERR_HIP, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count is 0"));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, SPLAT_ERR_INVALID, "device ordinal out of range");
    splat_ctx* c = new (std::nothrow) splat_ctx();
    if (!c) return fail(nullptr, SPLAT_ERR_INVALID, "out of host memory");
    c->cfg = *cfg;
    if (cfg->mode & SPLAT_MODE_FAST) {
        c->early_eps = 2e-3f;      // hi - lo <= 2 needs a contraction of ~1/128, not of ~1e-5 ...
        c->early_min = 384;        // ... which lists of a few hundred keys reach too (C3: 2960 -> 3125 frames/s; exact mode: 768 is best)
    }
    // the options a host reaches through splat_set_option; their environment variables pin them (include/splat_hip.h)
    option_from_env(c, SPLAT_OPT_SORT_IN_COMPOSITOR, "SPLAT_SORT_IN_COMP", -1, 1);
    if (const char* e0 = std::getenv("SPLAT_FAST_WIDTH")) { c->fast_width = std::atoi(e0) <= 1 ? 1.0f : 2.0f; c->env_pinned |= 1u << SPLAT_OPT_FAST_CLOSE_WIDTH; }
    option_from_env(c, SPLAT_OPT_EARLY_OUT_EPS, "SPLAT_EARLY_EPS", 0, 1);
    option_from_env(c, SPLAT_OPT_EARLY_OUT_MIN_LIST, "SPLAT_EARLY_MIN", 0, 1e9);
    option_from_env(c, SPLAT_OPT_EARLY_OUT_SCAN_EIGHTHS, "SPLAT_EARLY_SCAN8", 1, 8);
    option_from_env(c, SPLAT_OPT_PRIORITY_LIST_LEN, "SPLAT_PRIO_LEN", 1, 1073741823.0);
    option_from_env(c, SPLAT_OPT_FRAME_OVERLAP, "SPLAT_FRAME_OVERLAP", 1, 2);
    option_from_env(c, SPLAT_OPT_PIPELINE_DEPTH, "SPLAT_PIPELINE", 1, 6);
    option_from_env(c, SPLAT_OPT_TIMING_EVERY, "SPLAT_TIMING_EVERY", 1, 1e9);
    option_from_env(c, SPLAT_OPT_FUSED_SORT_MAX, "SPLAT_FUSED_SORT", 0, 2048);
    if (const char* e5 = std::getenv("SPLAT_BUCKETS")) { c->use_buckets = std::atoi(e5) != 0; c->env_pinned |= 1u << SPLAT_OPT_ONE_PASS_BINNING; }
    if (const char* e7 = std::getenv("SPLAT_CULL")) { c->cull_blocks = std::atoi(e7) != 0; c->env_pinned |= 1u << SPLAT_OPT_BLOCK_CULLING; }
    if (const char* e8 = std::getenv("SPLAT_DBG_TIGHT_GRIDS")) c->tight_grids = std::atoi(e8) != 0;
    option_from_env(c, SPLAT_OPT_PAIR_WALK, "SPLAT_PAIR_BLEND", -1, 1);
    if (const char* e6 = std::getenv("SPLAT_BUCKET_BYTES")) { c->bucket_bytes = std::strtoull(e6, nullptr, 10); c->env_pinned |= 1u << SPLAT_OPT_KEY_BUFFER_BYTES; }
    option_from_env(c, SPLAT_OPT_REGION_SPARE, "SPLAT_REGION_SPARE", 1, 1e9);
    if (const char* er = std::getenv("SPLAT_OVERFLOW_REDO")) { c->overflow_redo = std::min(2, std::max(0, std::atoi(er))); c->env_pinned |= 1u << SPLAT_OPT_OVERFLOW_REDO; }
    if (const char* en = std::getenv("SPLAT_NEAR_KEYS")) {
        const int v = std::atoi(en);
        c->near_cap = v <= 0 ? 0u : (unsigned int)std::min(2048, std::max(64, v));
        c->env_pinned |= 1u << SPLAT_OPT_NEAR_SELECT_KEYS;
    }
    option_from_env(c, SPLAT_OPT_HOST_ZERO_COPY, "SPLAT_HOST_ZERO_COPY", 0, 1);
    option_from_env(c, SPLAT_OPT_COUNT_FIRST, "SPLAT_COUNT_FIRST", 0, 2);
    if (const char* kg = std::getenv("SPLAT_KEYS_PER_GAUSSIAN")) {
        const int v = std::atoi(kg);
        c->keys_per_gaussian = v <= 0 ? 0u : (unsigned int)std::min(256, std::max(4, v));
        c->env_pinned |= 1u << SPLAT_OPT_KEYS_PER_GAUSSIAN;
    }
    option_from_env(c, SPLAT_OPT_LARGE_SPLAT_TILES, "SPLAT_LARGE_TILES", -1, 1048576);
    option_from_env(c, SPLAT_OPT_LARGE_LIST_MIN, "SPLAT_LARGE_LIST_MIN", -1, 1e9);
    if (const char* lm = std::getenv("SPLAT_LAYOUT_MOTION")) c->layout_motion = std::atoi(lm) != 0 ? 1 : 0;
    if (const char* k1 = std::getenv("SPLAT_SORT_RADIX_MIN")) c->knobs.sort_radix_min = (unsigned int)std::max(0, std::atoi(k1));
    if (const char* k2 = std::getenv("SPLAT_SCAN_THREADS")) c->knobs.scan_threads = std::atoi(k2);
    if (const char* k3 = std::getenv("SPLAT_DBG_NTILES")) c->knobs.dbg_ntiles = (unsigned int)std::max(0, std::atoi(k3));
    if (const char* k8 = std::getenv("SPLAT_DBG_SELECT_STRIDE")) c->knobs.dbg_select_stride = (unsigned int)std::max(0, std::atoi(k8));
    if (const char* k10 = std::getenv("SPLAT_START_HINTS")) { c->start_hints = std::min(2, std::max(0, std::atoi(k10))); c->env_pinned |= 1u << SPLAT_OPT_START_HINTS; }
    if (const char* k7 = std::getenv("SPLAT_DBG_ONE_PASS_SELECT")) c->one_pass_select = std::atoi(k7) != 0;
    if (const char* k11 = std::getenv("SPLAT_DBG_KEYS2_ENTRIES")) c->knobs.dbg_keys2_entries = std::strtoull(k11, nullptr, 10);
    if (const char* k9 = std::getenv("SPLAT_DBG_HINT_RADIUS")) c->knobs.dbg_hint_radius = std::atoi(k9);
    if (const char* k5 = std::getenv("SPLAT_DBG_STARTS")) c->knobs.dbg_starts = std::atoi(k5) != 0 ? 1u : 0u;
    if (const char* k4 = std::getenv("SPLAT_COMP_LDS_PAD")) c->knobs.comp_lds_pad = (unsigned int)std::max(0, std::atoi(k4));
    if (const char* k1p = std::getenv("SPLAT_K1_LDS_PAD")) c->knobs.k1_lds_pad = (unsigned int)std::min(48 * 1024, std::max(0, std::atoi(k1p)));
    auto bail = [&](const char* what, hipError_t err) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(err);
        splat_destroy(c);
        return SPLAT_ERR_HIP;
    };
    if ((e = hipSetDevice(cfg->device)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = init_device_kernels()) != hipSuccess) return bail("hipFuncSetAttribute", e);
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    c->own_stream = true;
    {   // The binning streams are separate hardware queues from the caller's stream.  With ONE chain in flight
        // (SPLAT_PIPELINE <= 5) they get the highest priority: the chain is the frame's critical path and its
        // short kernels should not queue behind a frame-long compositor (normal priority: 2174 -> 1969 fps).
        // With TWO chains in flight (6, the default) the compositor is the critical path and the chains run at
        // its priority (2349 -> 2384 fps on C3, C5 +1 %, C2 unchanged).
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        int prio = c->pipeline >= 6 ? (lo + hi) / 2 : hi;
        if (const char* ep = std::getenv("SPLAT_BIN_PRIO")) prio = std::atoi(ep) > 0 ? hi : (std::atoi(ep) < 0 ? lo : (lo + hi) / 2);
        if ((e = hipStreamCreateWithPriority(&c->bin_stream, hipStreamNonBlocking, prio)) != hipSuccess) return bail("hipStreamCreate", e);
        if ((e = hipStreamCreateWithPriority(&c->sort_stream, hipStreamNonBlocking, prio)) != hipSuccess) return bail("hipStreamCreate", e);
    }
    if (c->overlap >= 2 && c->pipeline) { if ((e = ensure_copy_stream(c)) != hipSuccess) return bail("hipStreamCreate", e); c->comp2 = c->copy_stream; }
    for (Slot& s : c->slots) {
        if ((e = dmalloc(c, &s.d_status, sizeof(FrameStatus))) != hipSuccess) return bail("hipMalloc(status)", e);
        if ((e = fill_now(s.d_status, 0, sizeof(FrameStatus))) != hipSuccess) return bail("fill_now(status)", e);
        if ((e = hipEventCreateWithFlags(&s.ev_ready, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&s.ev_binned, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    }
    if ((e = dmalloc(c, &c->d_status_ring, sizeof(FrameStatus) * EV_RING)) != hipSuccess) return bail("hipMalloc(status ring)", e);
    if ((e = fill_now(c->d_status_ring, 0, sizeof(FrameStatus) * EV_RING)) != hipSuccess) return bail("fill_now(status ring)", e);
    if ((e = hipHostMalloc(&c->h_status, sizeof(FrameStatus) * EV_RING, hipHostMallocCoherent)) != hipSuccess) return bail("hipHostMalloc(status)", e);
    std::memset(c->h_status, 0, sizeof(FrameStatus) * EV_RING);
    for (auto& s : c->ring)
        for (auto& ev : s.e)
            if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    *out = c;
    return SPLAT_OK;
}

void splat_destroy(splat_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->bin_stream) (void)hipStreamSynchronize(c->bin_stream);
    if (c->sort_stream) (void)hipStreamSynchronize(c->sort_stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);      // lane 1: compositors and gathers that use the buffers / the communicator released below
    if (c->comm) { comm_release(c->comm); c->comm = nullptr; }
    free_scene(c);
    dfree(c->zero_layout); dfree(c->need_hint);
    for (Slot& s : c->slots) {
        dfree(s.counts); dfree(s.offsets); dfree(s.cursor); dfree(s.order); dfree(s.lens); dfree(s.counts_b); dfree(s.lay_a); dfree(s.lay_b);
        dfree(s.near_m); dfree(s.redo_layout); dfree(s.redo_cursors); dfree(s.off2);
        dfree(s.keys); dfree(s.keys2); dfree(s.d_status);
        if (s.ev_ready) (void)hipEventDestroy(s.ev_ready);
        if (s.ev_binned) (void)hipEventDestroy(s.ev_binned);
    }
    dfree(c->d_img); dfree(c->d_iters);
    for (int k = 0; k < splat_ctx::S_IMGS; ++k) {
        dfree(c->s_img[k]);
        if (c->s_rendered[k]) (void)hipEventDestroy(c->s_rendered[k]);
        if (c->s_copied[k]) (void)hipEventDestroy(c->s_copied[k]);
    }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->h_status) (void)hipHostFree(c->h_status);
    dfree(c->d_status_ring);
    for (auto& s : c->ring)
        for (auto& ev : s.e)
            if (ev) (void)hipEventDestroy(ev);
    if (c->bin_stream) (void)hipStreamDestroy(c->bin_stream);
    if (c->sort_stream) (void)hipStreamDestroy(c->sort_stream);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    {   // (nothing of this context may stay in the allocation table: an entry outliving it would be a dangling owner)
        std::lock_guard<std::mutex> g(g_ledger_mu);
        for (auto it = g_ledger.begin(); it != g_ledger.end();) it = (it->second.first == c) ? g_ledger.erase(it) : std::next(it);
    }
    delete c;
}

const char* splat_last_error(const splat_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

void* splat_stream(splat_ctx* c) { return c ? (void*)c->stream : nullptr; }

int splat_set_frame_overlap(splat_ctx* c, int32_t n) {
    if (!c) return SPLAT_ERR_INVALID;
    if (n < 1 || n > 2) return fail(c, SPLAT_ERR_INVALID, "frame overlap is 1 or 2");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    int rc = finish_quiet(c);
    if (rc != SPLAT_OK) return rc;
    c->overlap = n;
    if (n >= 2) return ensure_lane(c);
    c->comp2 = nullptr;                    // (the stream stays: it is the copy stream)
    c->lane[0] = splat_ctx::Lane{}; c->lane[1] = splat_ctx::Lane{};
    for (auto& e : c->img_tab) e = splat_ctx::ImgRec{};
    c->last_lane = 0;
    return SPLAT_OK;
}

int splat_set_option(splat_ctx* c, int32_t option, double value) {
    if (!c) return SPLAT_ERR_INVALID;
    if (option < 1 || option > 31) return fail(c, SPLAT_ERR_INVALID, "unknown option");
    double cur;
    if (!load_option(c, option, &cur)) return fail(c, SPLAT_ERR_INVALID, "unknown option");
    // the value is checked before anything else happens: an invalid one is an error whether or not the option is pinned, and
    // leaves the context as it was
    if (option == SPLAT_OPT_FRAME_OVERLAP ? (value != 1.0 && value != 2.0) : !store_option(c, option, value, true))
        return fail(c, SPLAT_ERR_INVALID, "option value out of range");
    if (c->env_pinned & (1u << option)) return SPLAT_OK;      // the operator's environment variable stays in force
    reset_policy(c);      // (the next frames scan for their walks' starts again, nothing stays armed: thresholds may have changed)
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    int rc = finish_quiet(c);
    if (rc != SPLAT_OK) return rc;
    if (option == SPLAT_OPT_FRAME_OVERLAP) return splat_set_frame_overlap(c, (int32_t)value);
    if (option == SPLAT_OPT_KEYS_PER_GAUSSIAN) {
        // another key buffer size: the buffers are made again by the next frame (prepare_binning), the regions with them
        for (Slot& sl : c->slots) { dfree(sl.keys); dfree(sl.keys2); sl.layout_valid = false; sl.flip = 0; }
        c->cap = 0; c->cap2 = 0; c->bucket_failed = false;
    }
    if (!store_option(c, option, value)) return fail(c, SPLAT_ERR_INVALID, "option value out of range");
    // (another selection size: what the tiles' walks needed under the old one is forgotten)
    if (option == SPLAT_OPT_NEAR_SELECT_KEYS && c->need_hint && c->m_alloc)
        HIP_TRY(c, fill_now(c->need_hint, 0, sizeof(unsigned int) * 9u * (size_t)c->m_alloc));
    // (the large list switched on with a scene in place: its buffers exist from now on -- splat_upload_scene makes them otherwise)
    if (option == SPLAT_OPT_LARGE_SPLAT_TILES && c->large_tiles >= 0 && c->n != 0)
        for (Slot& sl : c->slots)
            if (!sl.large_list) {
                HIP_TRY(c, dmalloc(c, &sl.large_list, sizeof(uint4) * c->n));
                if (!sl.large_count) HIP_TRY(c, dmalloc(c, &sl.large_count, sizeof(unsigned int) * 4));
                HIP_TRY(c, fill_now(sl.large_count, 0, sizeof(unsigned int) * 4));
            }
    return SPLAT_OK;
}

int splat_get_option(const splat_ctx* c, int32_t option, double* value) {
    if (!c || !value) return SPLAT_ERR_INVALID;
    return load_option(c, option, value) ? SPLAT_OK : SPLAT_ERR_INVALID;
}

int splat_set_stream(splat_ctx* c, void* stream) {
    if (!c) return SPLAT_ERR_INVALID;
    int rc = finish_quiet(c);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)stream;
    c->own_stream = false;
    return rc;
}

int splat_upload_scene(splat_ctx* c, uint64_t n, const float* pos4, const float* cov3d, const float* opacity,
                       const float* sh) {
    if (!c) return SPLAT_ERR_INVALID;
    if (n && (!pos4 || !cov3d || !opacity || !sh)) return fail(c, SPLAT_ERR_INVALID, "NULL scene pointer");
    if (n >= 0xFFFFFFFFull) return fail(c, SPLAT_ERR_INVALID, "too many Gaussians (index is 32-bit)");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    (void)finish_frame(c);          // (frames of the scene being replaced: nothing left to redo)
    c->deferred_drop = false; c->frames_drop_reported = c->frames_dropped;
    int rc = sync_all(c);
    if (rc != SPLAT_OK) return rc;
    free_scene(c);
    if (n == 0) return SPLAT_OK;
    morton_order(n, pos4, c->h_orig);
    float *d_pos = nullptr, *d_cov = nullptr, *d_op = nullptr, *d_sh = nullptr;
    auto cleanup = [&] { dfree(d_pos); dfree(d_cov); dfree(d_op); dfree(d_sh); };
    hipError_t e;
#define UP_TRY(expr)                                                            \
    if ((e = (expr)) != hipSuccess) {                                            \
        cleanup();                                                               \
        return fail(c, SPLAT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e)); \
    }
    UP_TRY(dmalloc(c, &c->planes, sizeof(float4) * SCENE_PLANES * n));
    UP_TRY(dmalloc(c, &c->orig, sizeof(unsigned int) * n));
    for (Slot& s : c->slots) {
        UP_TRY(dmalloc(c, &s.recs, sizeof(Rec) * n));
        UP_TRY(dmalloc(c, &s.blockinfo, sizeof(unsigned int) * ((n + 255) / 256)));
        UP_TRY(hipMemsetAsync(s.blockinfo, 0, sizeof(unsigned int) * ((n + 255) / 256), c->stream));
        if (c->large_tiles >= 0) {          // (SPLAT_LARGE_TILES < 0: no list, K1's blocks expand their close-ups themselves)
            UP_TRY(dmalloc(c, &s.large_list, sizeof(uint4) * n));
            UP_TRY(dmalloc(c, &s.large_count, sizeof(unsigned int) * 4));
            UP_TRY(hipMemsetAsync(s.large_count, 0, sizeof(unsigned int) * 4, c->stream));
        }
    }
    UP_TRY(hipMemcpyAsync(c->orig, c->h_orig.data(), sizeof(unsigned int) * n, hipMemcpyHostToDevice, c->stream));
    std::vector<BlockBounds> hb;
    block_bounds(n, pos4, cov3d, c->h_orig, hb);
    UP_TRY(dmalloc(c, &c->bounds, sizeof(BlockBounds) * hb.size()));
    UP_TRY(hipMemcpyAsync(c->bounds, hb.data(), sizeof(BlockBounds) * hb.size(), hipMemcpyHostToDevice, c->stream));
    UP_TRY(dmalloc(c, &d_pos, sizeof(float) * 4 * n));
    UP_TRY(dmalloc(c, &d_cov, sizeof(float) * 9 * n));
    UP_TRY(dmalloc(c, &d_op, sizeof(float) * n));
    UP_TRY(dmalloc(c, &d_sh, sizeof(float) * 48 * n));
    UP_TRY(hipMemcpyAsync(d_pos, pos4, sizeof(float) * 4 * n, hipMemcpyHostToDevice, c->stream));
    UP_TRY(hipMemcpyAsync(d_cov, cov3d, sizeof(float) * 9 * n, hipMemcpyHostToDevice, c->stream));
    UP_TRY(hipMemcpyAsync(d_op, opacity, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    UP_TRY(hipMemcpyAsync(d_sh, sh, sizeof(float) * 48 * n, hipMemcpyHostToDevice, c->stream));
    launch_pack_scene(c->stream, n, d_pos, d_cov, d_op, d_sh, c->orig, c->planes);
    UP_TRY(hipGetLastError());
    UP_TRY(hipStreamSynchronize(c->stream));
#undef UP_TRY
    cleanup();
    c->n = n;
    c->region_mult = 0;
    c->bucket_failed = false;              // key storage is sized at the first frame (prepare_binning)
    for (Slot& sl : c->slots) sl.layout_valid = false;
    c->sort_hint = false;
    // another scene under every tile: what the walks of the old one needed says nothing (near selection, start hints)
    if (c->need_hint && c->m_alloc) (void)fill_now(c->need_hint, 0, sizeof(unsigned int) * 9u * (size_t)c->m_alloc);
    reset_policy(c);
    // The per-tile arrays (a few hundred KB per frame slot at 4K) exist before the first frame as well: fifty small allocations
    // and six synchronous fills were a third of its call.  A larger target than 3840 x 2160 makes them again, as always.
    if (c->m_alloc == 0 && ensure_bins(c, 240u * 135u) != SPLAT_OK) (void)hipGetLastError();
    // The key buffers of one-pass binning depend on the scene's size only: made here, not inside the first frame's call (1.5 GB
    // of hipMalloc on C3).  A failure is left to the first frame, which retries smaller sizes (prepare_binning).
    if (c->use_buckets && !c->cfg.pair_capacity && !c->bucket_failed) {
        const uint64_t want = std::min<uint64_t>(std::max<uint64_t>(c->cap, region_capacity_for(c, default_region_multiplier(c))), KEY_ENTRIES_MAX);
        const uint64_t want2 = std::min<uint64_t>(std::max<uint64_t>(c->cap2, default_keys2_capacity(c)), KEY_ENTRIES_MAX);
        if ((want + want2) * 8ull * (uint64_t)slots_in_use(c) <= c->bucket_bytes && ensure_keys(c, want, want2) != SPLAT_OK) (void)hipGetLastError();
    }
    return SPLAT_OK;
}

int splat_compute_cov3d(splat_ctx* c, uint64_t n, const float* scales3, const float* rot4, float* cov3d_out) {
    if (!c) return SPLAT_ERR_INVALID;
    if (n == 0) return SPLAT_OK;
    if (!scales3 || !rot4 || !cov3d_out) return fail(c, SPLAT_ERR_INVALID, "NULL pointer");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    float *d_s = nullptr, *d_r = nullptr, *d_o = nullptr;
    hipError_t e = hipSuccess;
    int rc = SPLAT_OK;
    do {
        if ((e = dmalloc(c, &d_s, sizeof(float) * 3 * n)) != hipSuccess) break;
        if ((e = dmalloc(c, &d_r, sizeof(float) * 4 * n)) != hipSuccess) break;
        if ((e = dmalloc(c, &d_o, sizeof(float) * 9 * n)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d_s, scales3, sizeof(float) * 3 * n, hipMemcpyHostToDevice, c->stream)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d_r, rot4, sizeof(float) * 4 * n, hipMemcpyHostToDevice, c->stream)) != hipSuccess) break;
        launch_cov3d(c->stream, n, d_s, d_r, d_o);
        if ((e = hipGetLastError()) != hipSuccess) break;
        if ((e = hipMemcpyAsync(cov3d_out, d_o, sizeof(float) * 9 * n, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) break;
        e = hipStreamSynchronize(c->stream);
    } while (0);
    if (e != hipSuccess) rc = fail(c, SPLAT_ERR_HIP, std::string("splat_compute_cov3d: ") + hipGetErrorString(e));
    dfree(d_s); dfree(d_r); dfree(d_o);
    return rc;
}

int splat_set_slab(splat_ctx* c, int32_t tile_row0, int32_t tile_row1) {
    if (!c) return SPLAT_ERR_INVALID;
    if (tile_row0 < 0 || (tile_row1 >= 0 && tile_row1 < tile_row0)) return fail(c, SPLAT_ERR_INVALID, "bad slab");
    if (tile_row0 == c->slab0 && tile_row1 == c->slab1) return SPLAT_OK;
    // another slab: the slots' region layouts (sized from the lists of the OLD slab's tile rows -- a partition that keeps
    // the tile count would keep them otherwise) and the sort launch sizes describe other tiles.  The next frame counts its
    // pairs first (bootstrap) instead of being binned against them, overflowing and getting dropped on the device.
    int rc = SPLAT_OK;
    if (c->frame_idx != 0) {
        (void)hipSetDevice(c->cfg.device);
        rc = finish_quiet(c);
    }
    for (Slot& sl : c->slots) sl.layout_valid = false;
    c->sort_hint = false;
    c->hint_pairs = 0; c->hint_maxlen = 0;
    if (c->need_hint && c->m_alloc) (void)hipMemsetAsync(c->need_hint, 0, sizeof(unsigned int) * 9u * (size_t)c->m_alloc, c->stream);
    c->slab0 = tile_row0; c->slab1 = tile_row1;
    reset_policy(c);
    return rc;
}

int splat_tile_row_loads(splat_ctx* c, const splat_camera* cam, uint64_t* row_pairs, int32_t n_rows) {
    if (!c) return SPLAT_ERR_INVALID;
    if (!row_pairs) return fail(c, SPLAT_ERR_INVALID, "row_pairs is NULL");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    int rc = finish_quiet(c);
    if (rc != SPLAT_OK) return rc;
    const int s0 = c->slab0, s1 = c->slab1;
    c->slab0 = 0; c->slab1 = -1;
    FrameConst fc; unsigned int nt = 0;
    rc = build_frame_const(c, cam, &fc, &nt);
    c->slab0 = s0; c->slab1 = s1;
    if (rc != SPLAT_OK) return rc;
    if (n_rows != fc.n_tile_rows) return fail(c, SPLAT_ERR_INVALID, "n_rows must be ceil(h/16)");
    for (int r = 0; r < n_rows; ++r) row_pairs[r] = 0;
    if (c->n == 0 || nt == 0) return SPLAT_OK;
    rc = ensure_bins(c, nt);
    if (rc != SPLAT_OK) return rc;
    Slot& s = c->slots[0];
    rc = ensure_two_pass_buffers(c, s);
    if (rc != SPLAT_OK) return rc;
    fc.bucket_cap = 0;          // count only
    HIP_TRY(c, hipMemsetAsync(s.counts, 0, sizeof(unsigned int) * ((size_t)nt + 1), c->stream));   // (one-pass frames leave cursors there)
    s.layout_valid = false; s.flip = 0;                                                          // ... and will find them gone
    HIP_TRY(c, hipMemsetAsync(s.d_status, 0, sizeof(FrameStatus), c->stream));
    launch_preprocess(c->stream, c->n, c->planes, c->orig, fc, s.recs, s.depth, s.rect, s.counts, s.vislist, nullptr, c->bounds, s.blockinfo, s.d_status);
    launch_scan(c->stream, nt, s.counts, s.offsets, s.cursor, s.order, s.lens, s.d_status, ~0ull, 0u, nt, nt, nt);
    HIP_TRY(c, hipGetLastError());
    std::vector<unsigned int> off((size_t)nt + 1);
    HIP_TRY(c, hipMemcpyAsync(off.data(), s.offsets, sizeof(unsigned int) * off.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemsetAsync(s.d_status, 0, sizeof(FrameStatus), c->stream));     // frames expect a zeroed status
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int r = 0; r < n_rows; ++r)
        row_pairs[r] = (uint64_t)off[(size_t)(r + 1) * fc.tiles_x] - (uint64_t)off[(size_t)r * fc.tiles_x];
    return SPLAT_OK;
}

}  // extern "C"

namespace {
// host_out != nullptr (splat_render_frame): the rendered rows also travel to host memory behind the compositor, on its stream,
// BEFORE the host waits -- one wait per frame; a frame the device skipped is redone and copied again
int render_device_impl(splat_ctx* c, const splat_camera* cam, void* d_argb, int32_t sync, splat_stats* stats, bool clear_first,
                       uint32_t* host_out = nullptr);
}

extern "C" {

int splat_render_device(splat_ctx* c, const splat_camera* cam, void* d_argb, int32_t sync, splat_stats* stats) {
    return render_device_impl(c, cam, d_argb, sync, stats, false);
}
int splat_render_frame_device(splat_ctx* c, const splat_camera* cam, void* d_argb, int32_t sync, splat_stats* stats) {
    return render_device_impl(c, cam, d_argb, sync, stats, true);
}

}  // extern "C"

namespace {
int render_device_impl(splat_ctx* c, const splat_camera* cam, void* d_argb, int32_t sync, splat_stats* stats, bool clear_first,
                       uint32_t* host_out) {
    if (!c) return SPLAT_ERR_INVALID;
    if (!d_argb) return fail(c, SPLAT_ERR_INVALID, "d_argb is NULL");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    int rc = SPLAT_OK;
    if (c->n == 0 && !c->planes) {
        // an empty scene renders nothing (the reference's loop body never runs); a viewer-loop frame is still cleared
        unsigned int nt;
        rc = build_frame_const(c, cam, &c->fc, &nt);
        if (rc) return rc;
        if (c->pre_wait) { HIP_TRY(c, hipStreamWaitEvent(c->stream, c->pre_wait, 0)); c->pre_wait = nullptr; }
        c->last_lane = 0;
        if (clear_first && c->fc.row_px1 > c->fc.row_px0) {
            HIP_TRY(c, hipMemsetAsync((uint32_t*)d_argb + (size_t)c->fc.row_px0 * c->fc.W, 0,
                                      (size_t)(c->fc.row_px1 - c->fc.row_px0) * c->fc.W * 4, c->stream));
            if (sync || stats) HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (host_out) std::memset(host_out + (size_t)c->fc.row_px0 * c->fc.W, 0, (size_t)(c->fc.row_px1 - c->fc.row_px0) * c->fc.W * 4);
        }
        if (stats) { c->last = FrameStatus{}; c->last_ring = -1; fill_stats(c, stats); }
        return SPLAT_OK;
    }
    rc = build_frame_const(c, cam, &c->fc, &c->n_tiles);
    if (rc != SPLAT_OK) return rc;
    if (c->n_tiles == 0) { if (stats) { c->last = FrameStatus{}; c->last_ring = -1; fill_stats(c, stats); } return SPLAT_OK; }
    rc = ensure_bins(c, c->n_tiles);
    if (rc != SPLAT_OK) return rc;
    for (int attempt = 0; attempt < 4; ++attempt) {
        rc = prepare_binning(c, c->n_tiles, &c->fc);
        if (rc != SPLAT_OK) return rc;
        const bool timed = stats != nullptr || c->timing_every <= 1 || (c->frame_idx % (uint64_t)c->timing_every) == 0;
        c->clear_first = clear_first;
        rc = enqueue_frame(c, (uint32_t*)d_argb, timed, stats != nullptr, !sync && !stats && c->comp2 != nullptr && c->overlap >= 2 && !c->streamed_call,
                           sync || stats != nullptr);
        c->clear_first = false;
        if (rc != SPLAT_OK) return rc;
        if (host_out && c->fc.row_px1 > c->fc.row_px0) {
            const size_t first = (size_t)c->fc.row_px0 * (size_t)c->fc.W, count = (size_t)(c->fc.row_px1 - c->fc.row_px0) * (size_t)c->fc.W;
            HIP_TRY(c, copy_to_host_image(host_out + first, (const uint32_t*)d_argb + first, count * 4u, frame_stream(c)));
        }
        if (!sync && !stats) return SPLAT_OK;
        bool skipped = false;
        rc = finish_frame(c, &skipped);
        {   // frames lost besides this one (older asynchronous frames) are reported by the next splat_sync
            const uint64_t pending = c->frames_dropped - c->frames_drop_reported;
            if (pending > (skipped ? 1u : 0u)) c->deferred_drop = true;
            c->frames_drop_reported = c->frames_dropped;
        }
        if (rc != SPLAT_ERR_CAPACITY) break;
        if (!skipped) {
            // THIS frame composited; what was lost is an older asynchronous frame.  Redoing the frame would
            // blend it onto the in/out image a second time: report the loss at the next splat_sync instead.
            c->deferred_drop = true;
            rc = SPLAT_OK;
            break;
        }
        // this frame was skipped on the device; its storage has been grown / its path switched: redo
        if (!(c->fc.bucket_cap || c->last.overflow == 3 || c->cap > c->last.n_pairs)) break;
    }
    if (rc != SPLAT_OK) return rc;
    if (stats) fill_stats(c, stats);
    return SPLAT_OK;
}
}  // namespace

extern "C" {

int splat_render(splat_ctx* c, const splat_camera* cam, uint32_t* argb, splat_stats* stats) {
    if (!c) return SPLAT_ERR_INVALID;
    if (!argb || !cam) return fail(c, SPLAT_ERR_INVALID, "NULL argument");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    FrameConst fc; unsigned int nt;
    int rc = build_frame_const(c, cam, &fc, &nt);
    if (rc != SPLAT_OK) return rc;
    size_t bytes = (size_t)fc.W * fc.H * 4;
    if (bytes > c->img_cap) {
        (void)finish_quiet(c);
        dfree(c->d_img); c->img_cap = 0;
        HIP_TRY(c, dmalloc(c, &c->d_img, bytes));
        c->img_cap = bytes;
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_img, argb, bytes, hipMemcpyHostToDevice, c->stream));
    rc = splat_render_device(c, cam, c->d_img, 1, stats);
    if (rc != SPLAT_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(argb, c->d_img, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPLAT_OK;
}

// The viewer loop's frame, host-visible and synchronous: `color.clear(0); render_to_buffer(&mut color)` (src/main.rs:73-74)
// as ONE call that ships no zeros.  splat_render uploads the caller's image (8.3 MB of zeros at 1080p), blends, downloads;
// here the clear is fused into the compositor and the pixels cross PCIe once, device -> host:
//   * a page-locked `argb_out` (splat_host_alloc / splat_host_register) whose memory the device can address is written by the
//     COMPOSITOR ITSELF (zero copy: the image crosses PCIe while the frame is still being composited; SPLAT_OPT_HOST_ZERO_COPY),
//   * any other `argb_out` gets a copy behind the compositor on its stream (page-locked: DMA at the PCIe rate; pageable: the
//     driver's staged copy).
int splat_render_frame(splat_ctx* c, const splat_camera* cam, uint32_t* argb_out, splat_stats* stats) {
    if (!c) return SPLAT_ERR_INVALID;
    if (!argb_out || !cam) return fail(c, SPLAT_ERR_INVALID, "NULL argument");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    FrameConst fc; unsigned int nt;
    int rc = build_frame_const(c, cam, &fc, &nt);
    if (rc != SPLAT_OK) return rc;
    const size_t bytes = (size_t)fc.W * fc.H * 4;
    if (c->host_zero_copy) {
        // is this image page-locked and mapped into the device's address space?  (pageable memory: the query fails)
        hipPointerAttribute_t at;
        std::memset(&at, 0, sizeof at);
        void* dev = nullptr;
        if (hipPointerGetAttributes(&at, argb_out) == hipSuccess && at.type == hipMemoryTypeHost &&
            hipHostGetDevicePointer(&dev, argb_out, 0) == hipSuccess && dev != nullptr &&
            host_range_is_locked(argb_out, dev, bytes))      // (the WHOLE image, not its first byte: a store past a mapping is a GPU page fault)
            return render_device_impl(c, cam, dev, 1, stats, true);      // the sync at the end of the frame makes the stores visible
        (void)hipGetLastError();
    }
    if (bytes > c->img_cap) {
        (void)finish_quiet(c);
        dfree(c->d_img); c->img_cap = 0;
        HIP_TRY(c, dmalloc(c, &c->d_img, bytes));
        c->img_cap = bytes;
    }
    return render_device_impl(c, cam, c->d_img, 1, stats, true, argb_out);
}

int splat_render_stream(splat_ctx* c, const splat_camera* cam, uint32_t* argb_out) {
    if (!c) return SPLAT_ERR_INVALID;
    if (!argb_out || !cam) return fail(c, SPLAT_ERR_INVALID, "NULL argument");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    FrameConst fc; unsigned int nt;
    int rc = build_frame_const(c, cam, &fc, &nt);
    if (rc != SPLAT_OK) return rc;
    const size_t bytes = (size_t)fc.W * fc.H * 4;
    HIP_TRY(c, ensure_copy_stream(c));
    if (bytes > c->s_cap) {
        rc = sync_all(c);
        if (rc != SPLAT_OK) return rc;
        for (int k = 0; k < splat_ctx::S_IMGS; ++k) {
            dfree(c->s_img[k]); c->s_used[k] = false;
            HIP_TRY(c, dmalloc(c, &c->s_img[k], bytes));
            HIP_TRY(c, fill_now(c->s_img[k], 0, bytes));      // a slab context renders its own rows only: the rest reads as zeros
        }
        c->s_cap = bytes;
    }
    const int k = (int)(c->s_idx++ % (uint64_t)splat_ctx::S_IMGS);
    // the image must not be cleared while its previous frame is still crossing PCIe: the compositor waits for that copy.
    // (Streamed frames stay on lane 0 whatever the frame overlap: lane 1's stream is the one their copies travel on.)
    c->pre_wait = c->s_used[k] ? c->s_copied[k] : nullptr;
    c->streamed_call = true;
    rc = splat_render_frame_device(c, cam, c->s_img[k], 0, nullptr);      // clear + render (the clear is fused into the compositor)
    c->streamed_call = false;
    if (c->pre_wait) {      // no frame was enqueued (an error, a target without tiles): nothing of this call may run ahead of the copy
        (void)hipStreamWaitEvent(c->stream, c->pre_wait, 0);
        c->pre_wait = nullptr; c->last_lane = 0;
    }
    if (rc != SPLAT_OK) return rc;
    c->s_ring[k] = c->last_ring; c->s_cam[k] = *cam;
    HIP_TRY(c, hipEventRecord(c->s_rendered[k], frame_stream(c)));
    HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->s_rendered[k], 0));
    HIP_TRY(c, hipMemcpyAsync(argb_out, c->s_img[k], bytes, hipMemcpyDeviceToHost, c->copy_stream));
    HIP_TRY(c, hipEventRecord(c->s_copied[k], c->copy_stream));
    c->s_dst[k] = argb_out; c->s_used[k] = true;
    return SPLAT_OK;
}

int splat_stream_wait(splat_ctx* c, const uint32_t* argb_out) {
    if (!c) return SPLAT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    // the most recent frame that went to this buffer (the other image may hold an older one)
    for (int j = 1; j <= splat_ctx::S_IMGS; ++j) {
        const int k = (int)((c->s_idx + (uint64_t)(splat_ctx::S_IMGS - j)) % (uint64_t)splat_ctx::S_IMGS);    // newest first
        if (c->s_used[k] && c->s_dst[k] == argb_out) {
            HIP_TRY(c, hipEventSynchronize(c->s_copied[k]));
            // The frame's status arrived before its pixels left.  If it was skipped on the device (it outgrew
            // its tile buckets / the sort launches sized from an earlier frame), grow the storage and render
            // it again, synchronously, into the same buffer: a viewer loop never sees the miss.
            const int r = c->s_ring[k];
            const bool skipped = r >= 0 && c->h_status[r].overflow != 0;
            if (!skipped) return SPLAT_OK;
            int rc = finish_quiet(c);                 // every frame in flight lands; storage grown
            if (rc != SPLAT_OK) return rc;
            {   // Streamed losses are redone by their own waits (this one below, the other images' by theirs: their status
                // words say so) and are therefore settled here; only a loss that belongs to NO streaming image -- a
                // plain asynchronous splat_render_device frame in flight beside the stream -- stays pending for the
                // next splat_sync.  (Without this the synchronous redo below found `dropped - reported >= 1` with its
                // own frame complete and left a spurious SPLAT_ERR_CAPACITY behind for the next splat_sync.)
                uint64_t streamed = 0;
                for (int q = 0; q < splat_ctx::S_IMGS; ++q)
                    if (c->s_used[q] && c->s_ring[q] >= 0 && c->h_status[c->s_ring[q]].overflow != 0) ++streamed;
                const uint64_t pending = c->frames_dropped - c->frames_drop_reported;
                c->deferred_drop = pending > streamed;
                c->frames_drop_reported = c->frames_dropped;
            }
            const size_t bytes = (size_t)c->s_cam[k].w * (size_t)c->s_cam[k].h * 4;
            const bool other_pending = c->deferred_drop;
            rc = splat_render_frame_device(c, &c->s_cam[k], c->s_img[k], 1, nullptr);
            if (rc != SPLAT_OK) return rc;
            c->deferred_drop = other_pending;          // (the redo's own bookkeeping saw only settled streamed losses)
            c->s_ring[k] = c->last_ring;
            HIP_TRY(c, hipMemcpyAsync(const_cast<uint32_t*>(argb_out), c->s_img[k], bytes, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            return SPLAT_OK;
        }
    }
    return fail(c, SPLAT_ERR_INVALID, "no streamed frame is bound to this buffer");
}

void* splat_device_alloc(splat_ctx* c, uint64_t bytes) {
    if (!c) return nullptr;
    void* p = nullptr;
    hipError_t e = hipSetDevice(c->cfg.device);
    if (e == hipSuccess) e = dmalloc(c, &p, (size_t)std::max<uint64_t>(bytes, 1));
    if (e != hipSuccess) { c->err = std::string("splat_device_alloc: ") + hipGetErrorString(e); return nullptr; }
    return p;
}
void splat_device_free(splat_ctx* c, void* p) {
    if (!c || !p) return;
    (void)hipSetDevice(c->cfg.device);
    (void)sync_all(c);                      // frames in flight may still be writing the image
    ledger_del(p);
    (void)hipFree(p);
}
int splat_device_upload(splat_ctx* c, void* d_dst, const void* h_src, uint64_t bytes) {
    if (!c) return SPLAT_ERR_INVALID;
    if (bytes && (!d_dst || !h_src)) return fail(c, SPLAT_ERR_INVALID, "NULL pointer");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, join_lanes(c));
    HIP_TRY(c, hipMemcpyAsync(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPLAT_OK;
}
int splat_device_download(splat_ctx* c, void* h_dst, const void* d_src, uint64_t bytes) {
    if (!c) return SPLAT_ERR_INVALID;
    if (bytes && (!h_dst || !d_src)) return fail(c, SPLAT_ERR_INVALID, "NULL pointer");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, join_lanes(c));
    HIP_TRY(c, hipMemcpyAsync(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPLAT_OK;
}

void* splat_host_alloc(uint64_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, (size_t)bytes) != hipSuccess) return nullptr;
    host_range_add(p, (size_t)bytes);
    return p;
}
void splat_host_free(void* p) { if (p) { host_range_del(p); (void)hipHostFree(p); } }
int splat_host_register(void* p, uint64_t bytes) {
    if (!p || !bytes) return SPLAT_ERR_INVALID;
    if (hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault) != hipSuccess) return SPLAT_ERR_HIP;
    host_range_add(p, (size_t)bytes);
    return SPLAT_OK;
}
int splat_host_unregister(void* p) {
    if (!p) return SPLAT_ERR_INVALID;
    host_range_del(p);
    return hipHostUnregister(p) == hipSuccess ? SPLAT_OK : SPLAT_ERR_HIP;
}

int splat_sync(splat_ctx* c) {
    if (!c) return SPLAT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    return finish_and_report(c);
}

uint64_t splat_frames_dropped(const splat_ctx* c) { return c ? c->frames_dropped : 0; }

uint64_t splat_device_bytes(const splat_ctx* c, uint64_t* peak) {
    if (!c) return 0;
    if (peak) *peak = c->dev_bytes_peak;
    return c->dev_bytes;
}

int splat_get_timing(splat_ctx* c, double ms_out[6], uint64_t* frames, int32_t reset) {
    if (!c) return SPLAT_ERR_INVALID;
    int rc = splat_sync(c);
    if (rc != SPLAT_OK) return rc;
    if (ms_out) for (int k = 0; k < N_TIMES; ++k) ms_out[k] = c->acc_ms[k];
    if (frames) *frames = c->acc_frames;
    if (reset) { for (auto& v : c->acc_ms) v = 0; c->acc_frames = 0; }
    return SPLAT_OK;
}

int splat_get_records(splat_ctx* c, splat_record* out, uint64_t n) {
    if (!c || !out) return SPLAT_ERR_INVALID;
    if (n != c->n) return fail(c, SPLAT_ERR_INVALID, "record count mismatch");
    if (n == 0) return SPLAT_OK;
    int rc = splat_sync(c);
    if (rc != SPLAT_OK) return rc;
    if (c->last_slot < 0) return fail(c, SPLAT_ERR_INVALID, "no frame rendered yet");
    rc = ensure_two_pass_buffers(c, c->slots[c->last_slot]);
    if (rc != SPLAT_OK) return rc;
    const Slot& s = c->slots[c->last_slot];
    // The 48-byte records are the ones the frame itself was composited from (written by whichever flavour of K1
    // rendered it -- preprocess_kernel<true> on the default one-pass path); they are read FIRST.
    std::vector<Rec> r(n); std::vector<float> d(n); std::vector<ushort4> q(n);
    HIP_TRY(c, hipMemcpy(r.data(), s.recs, sizeof(Rec) * n, hipMemcpyDeviceToHost));
    {
        // Depth and pixel rectangle exist only in registers on the one-pass path, and block culling skips whole
        // blocks (whose records above are then stale, for Gaussians that reach nothing): recompute those two with
        // the counting flavour of K1, culling off (same code, same values), then clear its counts again.
        FrameConst fc = c->fc;
        fc.bucket_cap = 0; fc.cull_blocks = 0;
        Slot& sl = c->slots[c->last_slot];
        HIP_TRY(c, hipMemsetAsync(sl.counts, 0, sizeof(unsigned int) * ((size_t)c->n_tiles + 1), c->stream));   // (cursors of a one-pass frame)
        sl.layout_valid = false; sl.flip = 0;
        HIP_TRY(c, hipMemsetAsync(s.d_status, 0, sizeof(FrameStatus), c->stream));
        launch_preprocess(c->stream, c->n, c->planes, c->orig, fc, s.recs, s.depth, s.rect, s.counts, s.vislist, nullptr, nullptr, nullptr, s.d_status);
        HIP_TRY(c, hipMemsetAsync(s.counts, 0, sizeof(unsigned int) * ((size_t)c->n_tiles + 1), c->stream));
        HIP_TRY(c, hipMemsetAsync(s.d_status, 0, sizeof(FrameStatus), c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    HIP_TRY(c, hipMemcpy(d.data(), s.depth, sizeof(float) * n, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(q.data(), s.rect, sizeof(ushort4) * n, hipMemcpyDeviceToHost));
    for (uint64_t j = 0; j < n; ++j) {          // everything lives in slot order; the caller gets original order
        const uint64_t i = c->h_orig[j];
        splat_record& o = out[i];
        o.cx = r[j].a.x; o.cy = r[j].a.y; o.hx = r[j].a.z; o.hy = r[j].a.w;
        o.conic_a = r[j].b.x; o.conic_b = c->fc.y_up ? r[j].b.y : -r[j].b.y; o.conic_c = r[j].b.z; o.opacity = r[j].b.w;
        o.r = r[j].c.x; o.g = r[j].c.y; o.b = r[j].c.z; o.depth = d[j];
        o.px0 = q[j].x; o.px1 = q[j].y; o.py0 = q[j].z; o.py1 = q[j].w;
    }
    return SPLAT_OK;
}

#if SPLAT_K1X == 30 || SPLAT_K1X == 31
int splat_debug_k1_stamps(splat_ctx* c, unsigned long long* out, unsigned long long n_words) {
    if (!c || c->last_slot < 0) return SPLAT_ERR_INVALID;
    (void)sync_all(c);
    return hipMemcpy(out, c->slots[c->last_slot].depth, n_words * 8, hipMemcpyDeviceToHost) == hipSuccess ? SPLAT_OK : SPLAT_ERR_HIP;
}
int splat_debug_k1_hwid(splat_ctx* c, unsigned long long* out, unsigned long long n_words) {
    if (!c || c->last_slot < 0) return SPLAT_ERR_INVALID;
    (void)sync_all(c);
    return hipMemcpy(out, c->slots[c->last_slot].rect, n_words * 8, hipMemcpyDeviceToHost) == hipSuccess ? SPLAT_OK : SPLAT_ERR_HIP;
}
#endif

// (debug, not part of the ABI)  The near selection's per-tile state of the most recent frame: lens[m], near_m[m], need_hint[4 m].
int splat_debug_near_state(splat_ctx* c, unsigned int* lens, unsigned int* near_m, unsigned int* need_hint, unsigned int m) {
    if (!c || c->last_slot < 0 || m != c->n_tiles) return SPLAT_ERR_INVALID;
    (void)sync_all(c);
    const Slot& s = c->slots[c->last_slot];
    bool ok = hipMemcpy(lens, s.lens, sizeof(unsigned int) * m, hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipMemcpy(near_m, s.near_m, sizeof(unsigned int) * m, hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipMemcpy(need_hint, c->need_hint, sizeof(unsigned int) * 4u * m, hipMemcpyDeviceToHost) == hipSuccess;
    return ok ? SPLAT_OK : SPLAT_ERR_HIP;
}

int64_t splat_binning_mode(splat_ctx* c) {
    if (!c || c->last_slot < 0) return -1;
    return (int64_t)c->fc.bucket_cap;
}

int splat_get_tile_lists(splat_ctx* c, uint32_t* tile_offsets, uint64_t n_offsets, uint32_t* order, uint64_t n_order) {
    if (!c) return SPLAT_ERR_INVALID;
    int rc = splat_sync(c);
    if (rc != SPLAT_OK) return rc;
    if (c->last_slot < 0) return fail(c, SPLAT_ERR_INVALID, "no frame rendered yet");
    if (!c->last_lists_in_memory)
        return fail(c, SPLAT_ERR_INVALID, "the last frame kept its sorted short lists on chip: render it with a stats pointer to read the lists");
    const Slot& s = c->slots[c->last_slot];
    if (n_offsets != (uint64_t)c->n_tiles + 1 || n_order != c->last.n_pairs)
        return fail(c, SPLAT_ERR_INVALID, "tile list size mismatch");
    if (c->last_near && c->last.overflow == 0) {
        // the frame's compositor selected the nearest keys of every list of more than 2048 keys and left the list itself
        // as K1 had written it: put those lists in order now (the sort launches, every tile in their grids)
        use_launch_knobs(&c->knobs);
        launch_sort(c->stream, c->n_tiles, c->n_tiles, c->n_tiles, c->n_tiles, s.offsets, s.order, s.lens, s.keys, s.keys2, c->slots[c->last_slot].d_status, c->orig, 2048u,
                    c->fc.bucket_cap ? s.off2 : nullptr);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->last_near = false;
    }
    const size_t m = c->n_tiles;
    std::vector<unsigned int> beg(m + 1), len(m + 1);
    HIP_TRY(c, hipMemcpy(beg.data(), s.offsets, sizeof(unsigned int) * (m + 1), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(len.data(), s.lens, sizeof(unsigned int) * m, hipMemcpyDeviceToHost));
    uint64_t run = 0;
    for (size_t t = 0; t < m; ++t) { tile_offsets[t] = (uint32_t)run; run += len[t]; }
    tile_offsets[m] = (uint32_t)run;
    if (run != n_order) return fail(c, SPLAT_ERR_INVALID, "tile list size mismatch");
    if (n_order) {
        std::vector<unsigned long long> k(n_order);
        if (!c->fc.bucket_cap) {
            HIP_TRY(c, hipMemcpy(k.data(), s.keys, sizeof(unsigned long long) * n_order, hipMemcpyDeviceToHost));
        } else {                    // lists live in fixed-stride buckets: gather them
            for (size_t t = 0; t < m; ++t)
                if (len[t])
                    HIP_TRY(c, hipMemcpy(k.data() + tile_offsets[t], s.keys + beg[t], sizeof(unsigned long long) * len[t],
                                         hipMemcpyDeviceToHost));
        }
        for (uint64_t i = 0; i < n_order; ++i) order[i] = c->h_orig[(uint32_t)k[i]];      // keys carry slots
    }
    return SPLAT_OK;
}

}  // extern "C"

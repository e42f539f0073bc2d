// splat_policy.cpp -- the frame scheduler's decisions (include/splat_policy.h).  Plain host C++: no HIP header, no context,
// nothing allocated; enqueue_frame (splat_api.hip) calls it once per frame and then only launches.  Every number in here was
// measured: DESIGN.md section 3 ("Frame scheduler") and profiles/README.md name the files.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "../../include/splat_policy.h"

namespace {

constexpr int TILE_PX = 16;

// A hash of what places the Gaussians on the target: the camera and the slab.  A frame binned into regions sized under the SAME
// hash cannot outgrow them -- the lists are the same lists.  (FNV-1a over the words; 0 is kept for "no camera yet".)
uint64_t camera_hash(const splat_policy_input& in) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t nbytes) {
        const unsigned char* b = (const unsigned char*)p;
        for (size_t q = 0; q < nbytes; ++q) { h ^= b[q]; h *= 1099511628211ull; }
    };
    mix(in.view, sizeof in.view); mix(in.proj, sizeof in.proj);
    const float five[5] = {in.w, in.h, in.htanx, in.htany, in.focal};
    mix(five, sizeof five); mix(in.cam, sizeof in.cam);
    mix(&in.lowpass, sizeof(float));
    const int32_t rows[2] = {in.tile_row0, in.n_tile_rows};
    mix(rows, sizeof rows);
    return h ? h : 1;
}

// How far the camera moved since the last frame: the largest change of a view / projection entry, translations relative to
// their size (a zoom is motion too).  A rotation by a small angle changes the entries by about the angle in radians.
float camera_delta(const splat_policy_input& in, const splat_policy_state& st) {
    float delta = 0.0f;
    for (int q = 0; q < 32; ++q) {
        const float a = q < 16 ? in.view[q] : in.proj[q - 16], b = st.last_view[q];
        const float d = std::fabs(a - b) / std::max(1.0f, std::max(std::fabs(a), std::fabs(b)));
        delta = (d == d) ? std::max(delta, d) : 1.0f;
    }
    return delta;
}

// Near selection needs the early-out (a walk that must start at the list's first key needs the whole list in order) and the
// compositor's own sort of the lists up to 2048 keys ...
bool near_selection(const splat_policy_knobs& k) {
    return k.near_cap != 0u && k.fused_sort_max >= 2048u && k.early_eps > 0.0f && k.sort_in_comp != 0;
}
// ... and is only worth its launch on frames that have such lists at all (the last harvested frame's longest list; nothing known
// yet: assume so) -- a list of 8192 keys, or a few hundred above 2048: on C2 (300 k Gaussians at 720p, a few dozen lists barely
// above 2048 keys, a 0.12-ms frame) the selection's launch cost more than the sort launches it replaces.
bool near_selection_for_frame(const splat_policy_knobs& k, const splat_policy_input& in) {
    if (!near_selection(k)) return false;
    if (!in.sort_hint || in.hint_maxlen == 0u) return true;
    return in.hint_maxlen > 2048u && (in.hint_ge8192 != 0u || in.hint_ge2048 >= 256u);
}
// Who orders the lists of more than 2048 keys: the tile's own compositor workgroup (no sort launches) with near selection, when
// forced, or -- auto -- when the average list is longer than 2048 keys (the sort launches would carry most of the frame) and the
// frame is not the chain of its longest list (more than 1500 pairs per key of that list: C5 3100; the four centre tile rows of
// C3 as a slab: average 2300 keys but 100 pairs per key of the 10 892-key list, whose sort must not move in front of its walk).
bool compositor_sorts_long_lists(const splat_policy_knobs& k, const splat_policy_input& in) {
    if (k.fused_sort_max < 2048u) return false;
    if (near_selection_for_frame(k, in)) return true;
    return k.sort_in_comp > 0 ||
           (k.sort_in_comp < 0 && in.hint_pairs > 2048ull * (uint64_t)in.n_tiles && in.hint_pairs > 1500ull * (uint64_t)in.hint_maxlen);
}

}  // namespace

extern "C" {

void splat_policy_default_knobs(splat_policy_knobs* k) {
    if (!k) return;
    std::memset(k, 0, sizeof *k);
    k->start_hints = 2; k->count_first = 1; k->overflow_redo = 1; k->early_min = 768; k->early_eps = 1e-6f;
    k->near_cap = 2048u; k->fused_sort_max = 2048u; k->sort_in_comp = -1; k->pair_mode = -1; k->pipeline = 6; k->tight_grids = 0;
    k->large_list_min = 256; k->layout_motion = 1;
}

void splat_policy_struct_sizes(uint64_t sizes[4]) {
    if (!sizes) return;
    sizes[0] = sizeof(splat_policy_knobs); sizes[1] = sizeof(splat_policy_state);
    sizes[2] = sizeof(splat_policy_input); sizes[3] = sizeof(splat_policy_decision);
}

int splat_policy_decide(const splat_policy_knobs* kp, const splat_policy_state* sp, const splat_policy_input* ip, splat_policy_decision* out) {
    if (!kp || !sp || !ip || !out) return -1;
    const splat_policy_knobs& k = *kp;
    const splat_policy_input& in = *ip;
    if (in.ring_entry < 0 || in.ring_entry >= SPLAT_POLICY_RING) return -1;
    const int r = in.ring_entry;
    splat_policy_decision d;
    std::memset(&d, 0, sizeof d);
    d.next = *sp;
    splat_policy_state& st = d.next;
    st.ring_kind[r] = 0;                         // (the entry is this frame's from here on)

    // ---- the camera: the same one, a creeping one, a moving one, a cut
    const uint64_t hash = camera_hash(in);
    d.cam_hash = hash;
    // (a camera at rest for a few frames -- frames overlap on the device: the hints a frame reads must come from the same camera
    // whichever of the frames before it wrote them last: the compositor's walks start where they started then)
    st.still_frames = (hash == sp->last_cam_hash) ? std::min(sp->still_frames + 1u, 1000u) : 0u;
    st.last_cam_hash = hash;
    const float delta = camera_delta(in, *sp);
    std::memcpy(st.last_view, in.view, sizeof in.view);
    std::memcpy(st.last_view + 16, in.proj, sizeof in.proj);
    d.cam_delta = delta;
    d.cam_jumped = delta >= SPLAT_POLICY_DELTA_JUMP ? 1 : 0;          // (a cut, not a pan: ~12 degrees or more since the last frame)

    // ---- START HINTS.  At rest for three frames: every hint in the table comes from this camera -> the walks start exactly where
    // they did (1).  The first frames at rest, and a camera that moves fast: scan (0), which also refreshes the hints.  A camera
    // that moved by less than ~half a degree: where they did plus a margin, and every fourth frame the scan, tiles taking turns
    // (>= 2: the frame number rides along).
    int mode = 0;
    if (k.start_hints >= 1 && st.still_frames >= (uint32_t)SPLAT_POLICY_STILL_FRAMES) mode = 1;
    else if (k.start_hints >= 2 && st.still_frames == 0u && delta < SPLAT_POLICY_DELTA_SLOW) mode = 2 + (int)(in.frame_idx & 0xffffull);
    d.start_hints_mode = mode;
    d.start_light = delta < SPLAT_POLICY_DELTA_CREEP ? 1 : 0;
    // how far the image moved since the last frame, in tiles: a rotation by delta radians shifts the centre by focal * delta
    // pixels.  The near selection looks that far around a tile for what its walks may need: 2 tiles for a camera at rest, 7 for a
    // 10-degree step.
    // (std::min(64.0f, x) is 64 for a NaN or infinite x -- a camera whose focal length is not a number: no undefined cast)
    d.hint_radius = std::min(7, std::max(2, (int)std::min(64.0f, std::ceil(delta * in.focal / (float)TILE_PX)) + 1));
    // (at rest the scan is paid once, in the first frames after the camera stopped: lists from half the usual length take the
    // early-out then -- 384 instead of 768 keys: C3 3060 -> 3120 frames/s, below that nothing more)
    d.early_min = k.early_min;
    if (k.start_hints >= 1 && st.still_frames >= 1u) d.early_min = std::min(d.early_min, std::max(k.early_min / 2, 1));

    // ---- COUNT FIRST: the frame counts its pairs per tile (K1's count flavour: a third of a K1) and bins into regions that fit
    // exactly ITS camera.  A slot without a layout does; and, by SPLAT_OPT_COUNT_FIRST, (1) the moving frames behind a run of frames
    // that outgrew regions sized two frames back (armed below), or (2) every frame whose camera moved by more than half a degree.
    bool count_first = false, moved = false;
    if (in.one_pass) {
        count_first = !in.layout_valid;
        if (!count_first && in.layout_cam != hash && k.count_first != 0) {
            if (k.count_first >= 2 && delta >= SPLAT_POLICY_DELTA_SLOW) count_first = true;
            else if (st.count_first_left > 0) { count_first = true; --st.count_first_left; }
        }
        if (count_first) st.ring_kind[r] = 2;
        moved = !count_first && in.layout_cam != hash;       // (a frame that counted first is binned into regions of its own camera)
    }

    // ---- what the frames in flight have reported (their scans write the status words to the host: a peek, no wait)
    if (in.one_pass && moved) {
        // OVERFLOW REDO, adaptive: a frame that outgrew a region (or was binned again) arms the redo launches from here on, not
        // from 32 lost frames later
        if (k.overflow_redo == 1 && (st.redo_armed < SPLAT_POLICY_REDO_RUN / 2 || st.count_first_left == 0)) {
            for (int q = 0; q < SPLAT_POLICY_RING; ++q) {
                if (!in.status[q].in_flight || q == r) continue;
                if (in.status[q].overflow == 2u || in.status[q].redone == 1u) { st.redo_armed = SPLAT_POLICY_REDO_RUN; break; }
            }
        }
        // ... and whether counting first pays on this path: of the frames binned into another camera's regions whose status has
        // arrived (in flight or harvested: a ring entry keeps both until it is used again), did three in four outgrow them?  A
        // frame binned twice costs two K1s, one that counts first 1.4; with half of them binned twice the optimistic frames still
        // won (C3's uncorrelated poses, +10 %), and one such frame arming this cost 11 % there (profiles/r06_knob_matrix.json).
        if (k.count_first != 0 && st.count_first_left == 0) {
            int known = 0, twice = 0;
            for (int q = 0; q < SPLAT_POLICY_RING; ++q) {
                if (q == r || st.ring_kind[q] != 1) continue;
                if (!in.status[q].arrived) continue;
                ++known; twice += (in.status[q].overflow == 2u || in.status[q].redone == 1u) ? 1 : 0;
            }
            if (known >= 2 && 4 * twice >= 3 * known) st.count_first_left = SPLAT_POLICY_COUNT_FIRST_RUN;
        }
        st.ring_kind[r] = 1;
    }
    // (the frame of a camera JUMP, whose lists have nothing to do with the ones its regions were sized from, and the frames right
    // behind it -- the slots' regions are sized two frames ahead, from lists of before the jump)
    if (d.cam_jumped && k.overflow_redo == 1) st.redo_armed = std::max(st.redo_armed, SPLAT_POLICY_REDO_JUMP_RUN);
    const bool redo = in.one_pass && moved && (k.overflow_redo >= 2 || (k.overflow_redo == 1 && (st.redo_armed > 0 || d.cam_jumped)));
    if (redo && st.redo_armed > 0) --st.redo_armed;
    d.count_first = count_first ? 1 : 0; d.moved = moved ? 1 : 0; d.redo = redo ? 1 : 0; d.ring_kind = st.ring_kind[r];

    // ---- a frame the caller waits for, with nothing else in flight (the reference's loop: one synchronous frame per pose,
    // src/main.rs:69-78), has nothing to overlap with: its whole chain goes on the caller's stream, in order
    d.solo = (in.awaited && in.idle && k.pipeline != 0) ? 1 : 0;

    // ---- the lists of more than 2048 keys, the launches' sizes, the walk's flavour
    const uint32_t m = in.n_tiles;
    const bool comp_sorts = compositor_sorts_long_lists(k, in) && in.has_keys2;
    d.comp_sorts = comp_sorts ? 1 : 0;
    d.near_cap = (comp_sorts && near_selection_for_frame(k, in)) ? k.near_cap : 0u;
    if (comp_sorts) { d.grid_big = m; d.grid_mid = m; d.grid_long = m; }          // no sort launches to size
    else if (in.sort_hint && k.tight_grids) { d.grid_big = in.hint_ge8192; d.grid_mid = in.hint_ge2048; d.grid_long = in.hint_ge16384; }
    else if (in.sort_hint) {
        // generous: an asynchronous frame that misses is lost (reported at the next sync), idle extra workgroups of a launch
        // that has the chip to itself cost next to nothing
        d.grid_big = (uint32_t)std::min<uint64_t>(m, 2ull * in.hint_ge8192 + 32);
        d.grid_mid = (uint32_t)std::min<uint64_t>(m, (uint64_t)in.hint_ge2048 + in.hint_ge2048 / 2 + 128);
        d.grid_long = (uint32_t)std::min<uint64_t>(m, 2ull * in.hint_ge16384 + 8);
    } else { d.grid_big = m; d.grid_mid = m; d.grid_long = m; }
    // The near selection's workgroups: an eighth as many as tiles, each walking the longest-first order with that stride until
    // the lists get short -- a frame in the pipeline has 0.3 ms of slack in front of its compositor, and fewer resident selections
    // leave the previous frame's compositor its LDS.  A frame the caller waits for gets a workgroup per long list (their number
    // a frame ago, plus an eighth).
    {
        uint32_t grid = (m + 7u) / 8u;
        if (in.sort_hint) grid = std::max<uint32_t>(grid, in.hint_ge2048 / 2u + 16u);       // (at most ~two long lists per workgroup: a close-up has thousands)
        if (in.awaited) grid = in.sort_hint ? std::max<uint32_t>(grid, in.hint_ge2048 + in.hint_ge2048 / 8u + 16u) : m;
        d.select_grid = std::min(grid, m);
    }
    // throughput-bound frames (many pairs per key of the longest list: C3 737, C5 3100) keep the one-record walk, the others
    // (C2 316, an eighth-of-a-frame slab 92, C1 36) take the paired one; measured crossover between 316 and 737
    d.pair_walk = k.pair_mode >= 0 ? (k.pair_mode != 0 ? 1 : 0)
                                   : ((in.hint_maxlen != 0 && in.hint_pairs < (uint64_t)SPLAT_POLICY_PAIR_WALK_RATIO * (uint64_t)in.hint_maxlen) ? 1 : 0);
    // ---- LARGE LIST: K1 lists the splats of hundreds of tiles and bin_large_kernel bins them tile by tile -- one more launch on
    // the frame's chain (~1 us when the list is empty, ~5 us of dependent round trips for a few dozen entries).  Worth it from a
    // few hundred large splats up (C3's bench pose has 900: K1 0.152 -> 0.134 ms; from inside the cloud thousands: 1.1 -> 0.18),
    // not for the 33 of C2's bench pose (8390 -> 8170 frames/s with the list).  K1 counts its large splats either way; nothing
    // known yet, or a camera jump: keep the list.  Half the threshold to let go again.
    // A camera AT REST whose frames nobody waits for is the one case where the list can cost: the walks start from hints, the
    // compositor is at its fastest, and the frame rate is the binning chain's length -- which the list's launch adds to (the
    // trained-like surface scene at rest: 3209 frames/s without, 2958 with, 7000 splats over the threshold but only 500 outside
    // the window; profiles/r07_knob_matrix.json).  There the list stays only for what a block cannot do well itself: a
    // thousand splats outside the window (a camera parked inside the scene has thousands).
    const bool rest_async = st.still_frames >= (uint32_t)SPLAT_POLICY_STILL_FRAMES && !in.awaited;
    const uint32_t have = rest_async ? in.hint_window : in.hint_large;
    const uint32_t need = (uint32_t)k.large_list_min * (rest_async ? 4u : 1u);
    if (k.large_list_min < 0 || !in.one_pass) st.large_on = 0u;
    else if (k.large_list_min == 0 || !in.sort_hint || d.cam_jumped) st.large_on = 1u;
    else if (sp->large_on) st.large_on = have >= need / 2u ? 1u : 0u;
    else st.large_on = have >= need ? 1u : 0u;
    d.use_large_list = (int32_t)st.large_on;
    // ---- the regions this frame's scan sizes are used two frames on (the chains of consecutive frames alternate between two
    // streams; one frame on with a shallower pipeline): by then a moving camera has shifted the image by about that many steps of
    // delta * focal pixels.  A tile's region is sized from the longest list within that distance (build_layout), so that the list
    // that slides over it finds room -- not after a jump (the lists to come have nothing to do with these) and not at rest.
    {
        const float ahead = k.pipeline >= 6 ? 2.0f : 1.0f;
        const float shift = ahead * delta * in.focal / (float)TILE_PX;            // tiles
        d.layout_radius = (k.layout_motion != 0 && in.one_pass && !d.cam_jumped && shift >= 0.5f) ? std::min(12, (int)std::min(64.0f, std::ceil(shift)) + 1) : 0;
        // (a frame the caller waits for with nothing else in flight pays the filter on its own chain -- 15-35 us of the scan's
        // launch, the reference's loop on C3 1670 -> 1565 frames/s when it ran unasked: there only while lists have been
        // outgrowing their regions lately; under frames in flight it is hidden)
        if (d.solo && st.redo_armed <= 0 && st.count_first_left <= 0) d.layout_radius = 0;
    }
    *out = d;
    return 0;
}

}  // extern "C"

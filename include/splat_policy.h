/* splat_policy.h -- the frame scheduler's DECISIONS as a pure function (no HIP call, no context, no allocation).
 *
 * splat_render_* enqueue a frame as a fixed sequence of launches (K1, scan, near selection, compositor: DESIGN.md section 3);
 * what varies from frame to frame is decided here, from the camera, what the frames before it reported, and a handful of words
 * of state carried from call to call:
 *   - start hints     may the compositor's walks start where the previous frame's did, and with what margin
 *   - hint radius     how far around a tile the near selection looks for what its walks may need
 *   - count first     does the frame count its (Gaussian, tile) pairs before it bins them (regions that fit ITS camera)
 *   - overflow redo   does the frame carry the launches that bin it again on the device if a list outgrew its region
 *   - large list      do the frame's large splats go through the list and bin_large_kernel (one more launch on the chain)
 *   - who orders the lists of more than 2048 keys, the near selection's launch size, the sort launches' sizes, the walk flavour
 * The library's enqueue_frame() calls splat_policy_decide() and then only launches; tests/test_frame_policy.py drives the same
 * function through scripted camera paths with injected frame statuses on a box without a GPU.
 * Not part of the reference's operator surface (include/splat_hip.h is): a diagnostic interface, versioned by struct size.
 * No reference counterpart: src/main.rs:69 renders when the pose is dirty and that is all the scheduling it has. */
#ifndef SPLAT_POLICY_H
#define SPLAT_POLICY_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SPLAT_POLICY_RING 32             /* frames whose status words the host keeps (the event ring of a context) */

/* thresholds of the decisions (named so that the tests cross each from both sides) */
#define SPLAT_POLICY_DELTA_SLOW 0.009f   /* largest relative change of a view / projection entry: below = slow motion (~half a degree) */
#define SPLAT_POLICY_DELTA_CREEP 0.003f  /* below = very slow motion (~0.15 degrees): lighter margins on hinted starts */
#define SPLAT_POLICY_DELTA_JUMP 0.2f     /* at or above = a cut, not a pan (~12 degrees) */
#define SPLAT_POLICY_STILL_FRAMES 3      /* frames at rest before every start hint in flight comes from this camera */
#define SPLAT_POLICY_COUNT_FIRST_RUN 64  /* moving frames that count first once counting first has been armed */
#define SPLAT_POLICY_REDO_RUN 256        /* moving frames that carry the redo launches once a list has outgrown its region */
#define SPLAT_POLICY_REDO_JUMP_RUN 8     /* ... and behind a camera jump */
#define SPLAT_POLICY_PAIR_WALK_RATIO 500 /* pairs per key of the longest list below which the paired walk is taken */

typedef struct splat_policy_knobs {      /* the context's options that decisions depend on (SPLAT_OPT_*) */
    int32_t start_hints;                 /* SPLAT_OPT_START_HINTS 0..2 */
    int32_t count_first;                 /* SPLAT_OPT_COUNT_FIRST 0..2 */
    int32_t overflow_redo;               /* SPLAT_OPT_OVERFLOW_REDO 0..2 */
    int32_t early_min;                   /* SPLAT_OPT_EARLY_OUT_MIN_LIST */
    float early_eps;                     /* SPLAT_OPT_EARLY_OUT_EPS (0: no early-out) */
    uint32_t near_cap;                   /* SPLAT_OPT_NEAR_SELECT_KEYS (0: off) */
    uint32_t fused_sort_max;             /* SPLAT_OPT_FUSED_SORT_MAX */
    int32_t sort_in_comp;                /* SPLAT_OPT_SORT_IN_COMPOSITOR: -1 auto, 0, 1 */
    int32_t pair_mode;                   /* SPLAT_OPT_PAIR_WALK: -1 auto, 0, 1 */
    int32_t pipeline;                    /* SPLAT_OPT_PIPELINE_DEPTH (0: everything on one stream) */
    int32_t tight_grids;                 /* debug: sort launches sized with no margin */
    int32_t layout_motion;               /* 1: a moving camera's regions are sized from the lists around each tile (layout_radius); 0: from its own */
    int32_t large_list_min;              /* large splats (SPLAT_LARGE_TILES) a recent frame must have had for frames to keep a large list:
                                            0 = always, < 0 = never (K1's blocks expand close-ups themselves).  An asynchronous frame
                                            of a camera AT REST keeps it only from four times as many splats outside K1's window */
} splat_policy_knobs;

typedef struct splat_policy_state {      /* carried from frame to frame; all zeros = a fresh context / scene / target */
    uint64_t last_cam_hash;
    uint32_t still_frames;               /* frames in a row with the same camera hash */
    int32_t count_first_left;            /* moving frames left that count first */
    int32_t redo_armed;                  /* moving frames left that carry the redo launches (adaptive mode) */
    uint32_t large_on;                   /* the frames keep a large-splat list (hysteresis: off below half of large_list_min) */
    float last_view[32];                 /* the previous frame's view and projection */
    uint8_t ring_kind[SPLAT_POLICY_RING];/* per ring entry, how its frame was binned: 1 into another camera's regions, 2 counted first, 0 neither */
} splat_policy_state;

typedef struct splat_policy_frame_status {   /* what a frame's scan has delivered to the host so far */
    uint32_t in_flight;                  /* the ring entry holds a frame that has not been harvested */
    uint32_t arrived;                    /* its scan has written the words below */
    uint32_t overflow;                   /* FrameStatus::overflow (2: a list outgrew its region) */
    uint32_t redone;                     /* 1: binned again on the device */
} splat_policy_frame_status;

typedef struct splat_policy_input {
    /* the camera and the slab (what places the Gaussians on the target) */
    float view[16], proj[16];
    float w, h, htanx, htany, focal;
    float cam[3];
    float lowpass;
    int32_t tile_row0, n_tile_rows;
    uint64_t frame_idx;                  /* this frame's number in the context, from 1 */
    int32_t ring_entry;                  /* the status ring entry this frame takes */
    int32_t one_pass;                    /* one-pass binning (per-tile regions of the key buffer) */
    int32_t layout_valid;                /* the frame's slot has regions ... */
    int32_t awaited;                     /* the caller waits for this frame */
    uint64_t layout_cam;                 /* ... sized under this camera hash */
    int32_t idle;                        /* nothing of the context is in flight */
    int32_t has_keys2;                   /* the slot has a second key buffer */
    uint32_t n_tiles;
    int32_t sort_hint;                   /* the list-length profile of an earlier frame is known: */
    uint32_t hint_maxlen, hint_ge2048, hint_ge8192, hint_ge16384;
    uint64_t hint_pairs;
    uint32_t hint_large;                 /* large splats of the last harvested frame (listed or only counted) */
    uint32_t hint_window;                /* ... those among them outside K1's 32 x 32-tile window: without a list, one global atomic per pair */
    splat_policy_frame_status status[SPLAT_POLICY_RING];
} splat_policy_input;

typedef struct splat_policy_decision {
    uint64_t cam_hash;
    float cam_delta;                     /* largest relative change of a view / projection entry since the last frame */
    int32_t cam_jumped;
    int32_t start_hints_mode;            /* FrameConst::start_hints: 0 scan, 1 at rest, >= 2 slow motion (the frame number rides along) */
    int32_t start_light;
    int32_t early_min;                   /* shortest list the early-out is tried on, this frame */
    int32_t hint_radius;                 /* 2..7 tiles */
    int32_t count_first;
    int32_t moved;                       /* binned into regions sized for another camera */
    int32_t redo;                        /* carries the overflow-redo launches */
    int32_t ring_kind;                   /* what next.ring_kind[ring_entry] became */
    int32_t solo;                        /* whole chain on the caller's stream */
    int32_t comp_sorts;                  /* the compositor's workgroups order the lists of more than 2048 keys (no sort launches) */
    uint32_t near_cap;                   /* != 0: near selection, with this many keys */
    uint32_t select_grid;                /* workgroups of the near selection's launch */
    uint32_t grid_big, grid_mid, grid_long;   /* prefixes of the longest-first order the sort launches cover */
    int32_t pair_walk;
    int32_t use_large_list;              /* K1 lists its large splats and bin_large_kernel bins them tile by tile (else K1 expands them itself) */
    int32_t layout_radius;               /* tiles: the regions this frame's scan builds (for the frame two on) are sized from the longest
                                            list within this distance of each tile; 0 = from the tile's own list (camera at rest, a jump) */
    int32_t reserved;
    splat_policy_state next;
} splat_policy_decision;

/* SPLAT_OK (0) or SPLAT_ERR_INVALID (-1: NULL argument, ring_entry out of range).  Pure: the same arguments give the same decision. */
int splat_policy_decide(const splat_policy_knobs* knobs, const splat_policy_state* state, const splat_policy_input* in,
                        splat_policy_decision* out);
/* the knobs of a context created with no SPLAT_* environment variable set */
void splat_policy_default_knobs(splat_policy_knobs* knobs);
/* sizeof the four structs above, in declaration order, as the LIBRARY was built (a binding compares before the first call) */
void splat_policy_struct_sizes(uint64_t sizes[4]);

#ifdef __cplusplus
}
#endif
#endif

"""The frame scheduler's decisions, driven without a GPU.

enqueue_frame (splat_amd/csrc/splat_api.hip) asks splat_policy_decide (include/splat_policy.h, splat_policy.cpp: a pure
function, no HIP call) what a frame does -- start hints, near-selection neighbourhood, count first, overflow redo, who orders
the long lists, launch sizes, the walk's flavour -- and then only launches.  These tests script camera paths rest -> creep ->
pan -> jump -> rest, inject the statuses the frames' scans would have reported, and assert the decisions; every threshold is
crossed from both sides.  The reference has no counterpart (src/main.rs:69: render when the pose is dirty)."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

from splat_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RING = 32
DELTA_SLOW, DELTA_CREEP, DELTA_JUMP = 0.009, 0.003, 0.2


class Knobs(C.Structure):
    _fields_ = [("start_hints", C.c_int32), ("count_first", C.c_int32), ("overflow_redo", C.c_int32), ("early_min", C.c_int32),
                ("early_eps", C.c_float), ("near_cap", C.c_uint32), ("fused_sort_max", C.c_uint32), ("sort_in_comp", C.c_int32),
                ("pair_mode", C.c_int32), ("pipeline", C.c_int32), ("tight_grids", C.c_int32), ("layout_motion", C.c_int32), ("large_list_min", C.c_int32)]


class State(C.Structure):
    _fields_ = [("last_cam_hash", C.c_uint64), ("still_frames", C.c_uint32), ("count_first_left", C.c_int32),
                ("redo_armed", C.c_int32), ("large_on", C.c_uint32), ("last_view", C.c_float * 32), ("ring_kind", C.c_uint8 * RING)]


class FrameStatus(C.Structure):
    _fields_ = [("in_flight", C.c_uint32), ("arrived", C.c_uint32), ("overflow", C.c_uint32), ("redone", C.c_uint32)]


class Input(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("proj", C.c_float * 16), ("w", C.c_float), ("h", C.c_float), ("htanx", C.c_float),
                ("htany", C.c_float), ("focal", C.c_float), ("cam", C.c_float * 3), ("lowpass", C.c_float),
                ("tile_row0", C.c_int32), ("n_tile_rows", C.c_int32), ("frame_idx", C.c_uint64), ("ring_entry", C.c_int32),
                ("one_pass", C.c_int32), ("layout_valid", C.c_int32), ("awaited", C.c_int32), ("layout_cam", C.c_uint64),
                ("idle", C.c_int32), ("has_keys2", C.c_int32), ("n_tiles", C.c_uint32), ("sort_hint", C.c_int32),
                ("hint_maxlen", C.c_uint32), ("hint_ge2048", C.c_uint32), ("hint_ge8192", C.c_uint32), ("hint_ge16384", C.c_uint32),
                ("hint_pairs", C.c_uint64), ("hint_large", C.c_uint32), ("hint_window", C.c_uint32), ("status", FrameStatus * RING)]


class Decision(C.Structure):
    _fields_ = [("cam_hash", C.c_uint64), ("cam_delta", C.c_float), ("cam_jumped", C.c_int32), ("start_hints_mode", C.c_int32),
                ("start_light", C.c_int32), ("early_min", C.c_int32), ("hint_radius", C.c_int32), ("count_first", C.c_int32),
                ("moved", C.c_int32), ("redo", C.c_int32), ("ring_kind", C.c_int32), ("solo", C.c_int32), ("comp_sorts", C.c_int32),
                ("near_cap", C.c_uint32), ("select_grid", C.c_uint32), ("grid_big", C.c_uint32), ("grid_mid", C.c_uint32),
                ("grid_long", C.c_uint32), ("pair_walk", C.c_int32), ("use_large_list", C.c_int32), ("layout_radius", C.c_int32), ("reserved", C.c_int32), ("next", State)]


@pytest.fixture(scope="module")
def L():
    lib = C.CDLL(_lib.LIB_PATH)          # loads without a GPU
    lib.splat_policy_decide.restype = C.c_int
    lib.splat_policy_decide.argtypes = [C.POINTER(Knobs), C.POINTER(State), C.POINTER(Input), C.POINTER(Decision)]
    lib.splat_policy_default_knobs.restype = None
    lib.splat_policy_default_knobs.argtypes = [C.POINTER(Knobs)]
    lib.splat_policy_struct_sizes.restype = None
    lib.splat_policy_struct_sizes.argtypes = [C.POINTER(C.c_uint64)]
    return lib


def test_header_symbols_are_exported_and_the_structs_are_the_librarys(L):
    hdr = open(os.path.join(ROOT, "include", "splat_policy.h")).read()
    declared = set(re.findall(r"\b(splat_policy_[a-z_]+)\s*\(", hdr))
    assert declared == {"splat_policy_decide", "splat_policy_default_knobs", "splat_policy_struct_sizes"}
    for n in declared:
        assert hasattr(L, n)
    sizes = (C.c_uint64 * 4)()
    L.splat_policy_struct_sizes(sizes)
    assert list(sizes) == [C.sizeof(Knobs), C.sizeof(State), C.sizeof(Input), C.sizeof(Decision)]
    for name, val in re.findall(r"#define SPLAT_POLICY_(DELTA_[A-Z]+) ([0-9.]+)f", hdr):
        assert float(val) == {"DELTA_SLOW": DELTA_SLOW, "DELTA_CREEP": DELTA_CREEP, "DELTA_JUMP": DELTA_JUMP}[name]


def test_null_arguments_and_a_bad_ring_entry_are_refused(L):
    k, st, i, d = Knobs(), State(), Input(), Decision()
    L.splat_policy_default_knobs(k)
    assert L.splat_policy_decide(None, st, i, d) == -1
    assert L.splat_policy_decide(k, st, i, None) == -1
    i.ring_entry = RING
    assert L.splat_policy_decide(k, st, i, d) == -1
    i.ring_entry = -1
    assert L.splat_policy_decide(k, st, i, d) == -1


def yaw_view(angle, dist=5.0):
    """a camera orbiting the origin about y (column-major 4x4, like splat_camera.view)"""
    c, s = math.cos(angle), math.sin(angle)
    m = np.array([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, -dist], [0, 0, 0, 1]], np.float32)
    return m.T.reshape(-1)


PROJ = np.array([[1 / 1.7778, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1.0002, -0.020002], [0, 0, -1, 0]], np.float32).T.reshape(-1)


class Driver:
    """The host side of enqueue_frame around the policy: the ring, the slots' layouts (four slots, regions written two frames
    ahead like the scan's second workgroup does), and the statuses the test injects."""

    def __init__(self, L, **knobs):
        self.L = L
        self.k = Knobs()
        L.splat_policy_default_knobs(self.k)
        for n, v in knobs.items():
            setattr(self.k, n, v)
        self.st = State()
        self.frame = 0
        self.slots = [dict(valid=False, cam=0) for _ in range(4)]
        self.status = [dict(in_flight=0, arrived=0, overflow=0, redone=0) for _ in range(RING)]
        self.hints = dict(sort_hint=0, hint_maxlen=0, hint_ge2048=0, hint_ge8192=0, hint_ge16384=0, hint_pairs=0, hint_large=0, hint_window=0)
        self.n_tiles = 8160
        self.log = []

    def reset(self):
        """what reset_policy() does in the library (scene / target / slab / option change)"""
        self.st.still_frames = 0
        self.st.last_cam_hash = 0
        self.st.count_first_left = 0
        self.st.redo_armed = 0
        for q in range(RING):
            self.st.ring_kind[q] = 0
        for s in self.slots:
            s["valid"] = False

    def settle(self, angle=0.0, frames=6):
        """a context that has been at `angle` for a while: regions everywhere, and the run of redo launches its very first
        frame armed (a jump away from 'no camera') spent long ago"""
        for _ in range(frames):
            self.step(angle)
        self.st.redo_armed = 0

    def step(self, angle, report=None, awaited=0, idle=0, one_pass=1, has_keys2=1, focal=540.0):
        """one frame at yaw `angle`; report = (overflow, redone) its scan will deliver (visible to the NEXT frames)"""
        self.frame += 1
        r = (self.frame - 1) % RING
        si = (self.frame - 1) % 4
        i = Input()
        i.view[:] = list(yaw_view(angle))
        i.proj[:] = list(PROJ)
        i.w, i.h, i.htanx, i.htany, i.focal = 1920.0, 1080.0, 1.7778, 1.0, focal
        i.cam[:] = [0.0, 0.0, 5.0]
        i.lowpass = 0.01
        i.tile_row0, i.n_tile_rows = 0, 68
        i.frame_idx, i.ring_entry, i.one_pass = self.frame, r, one_pass
        i.layout_valid, i.layout_cam = int(self.slots[si]["valid"]), self.slots[si]["cam"]
        i.awaited, i.idle, i.has_keys2, i.n_tiles = awaited, idle, has_keys2, self.n_tiles
        for n, v in self.hints.items():
            setattr(i, n, v)
        self.status[r] = dict(in_flight=0, arrived=0, overflow=0, redone=0)      # (the host zeroes its copy when it enqueues the frame)
        for q in range(RING):
            for n, v in self.status[q].items():
                setattr(i.status[q], n, v)
        d = Decision()
        assert self.L.splat_policy_decide(self.k, self.st, i, d) == 0
        # purity: the same arguments give the same decision
        d2 = Decision()
        assert self.L.splat_policy_decide(self.k, self.st, i, d2) == 0
        assert bytes(d) == bytes(d2)
        self.st = State.from_buffer_copy(d.next)
        # what enqueue_frame does with the decision: the slot's regions, the next-but-one slot's regions
        if one_pass:
            if d.count_first:
                self.slots[si] = dict(valid=True, cam=d.cam_hash)
            self.slots[(si + 2) % 4] = dict(valid=True, cam=d.cam_hash)
        ov, rd = report if report is not None else (0, 0)
        self.status[r] = dict(in_flight=1, arrived=1, overflow=ov, redone=rd)
        self.log.append(d)
        return d


def test_rest_creep_pan_jump_rest(L):
    D = Driver(L)
    a = 0.0
    # ---- first frames at one pose: slots without regions count first; hints only after three frames at rest
    modes = []
    for f in range(8):
        d = D.step(a)
        modes.append(d.start_hints_mode)
        assert d.count_first == (1 if f < 2 else 0), f        # slots 0 and 1 have no regions yet; 2 and 3 got theirs from frames 1 and 2
        assert d.moved == 0 and d.redo == 0
        assert d.hint_radius == (7 if f == 0 else 2)      # (the first frame of a context differs from 'no camera' by everything: a jump)
    assert modes == [0, 0, 0, 0, 1, 1, 1, 1] or modes == [0, 0, 0, 1, 1, 1, 1, 1]
    first_hinted = modes.index(1)
    assert D.log[first_hinted].next.still_frames == 3
    assert D.log[1].early_min == 384 and D.log[0].early_min == 768      # at rest the early-out takes lists from half the usual length
    # (the first frame of a context is a jump away from 'no camera': the first eight MOVING frames carry the redo launches)
    assert D.log[0].cam_jumped == 1 and D.st.redo_armed == 8
    moving = 0
    # ---- creeping: under DELTA_CREEP a frame -> hinted starts with the light margin, frame number riding along
    for f in range(6):
        a += 0.002
        d = D.step(a)
        moving += 1
        assert 0 < d.cam_delta < DELTA_CREEP
        assert d.start_hints_mode == 2 + (D.frame & 0xFFFF) and d.start_light == 1
        assert d.moved == 1 and d.count_first == 0 and d.cam_jumped == 0
        assert d.redo == 1 and d.next.redo_armed == 8 - moving
        assert d.ring_kind == 1
    # ---- slow pan: between the two thresholds -> hinted, normal margin; the jump's run ends, and nothing outgrew a region:
    # the adaptive redo stays off from here on
    for f in range(4):
        a += 0.006
        d = D.step(a)
        moving += 1
        assert DELTA_CREEP <= d.cam_delta < DELTA_SLOW
        assert d.start_hints_mode >= 2 and d.start_light == 0
        assert d.redo == (1 if moving <= 8 else 0)
    # ---- fast pan (10 degrees a frame): scan every frame, wide neighbourhood, still optimistic binning
    for f in range(6):
        a += math.radians(10.0)
        d = D.step(a)
        assert DELTA_SLOW <= d.cam_delta < DELTA_JUMP
        assert d.start_hints_mode == 0 and d.cam_jumped == 0
        assert d.hint_radius in (6, 7)
        assert d.count_first == 0 and d.moved == 1 and d.redo == 0
    # ---- a cut: the frame of the jump and the ones right behind it carry the redo launches
    a += math.radians(90.0)
    d = D.step(a)
    assert d.cam_jumped == 1 and d.cam_delta >= DELTA_JUMP and d.redo == 1 and d.start_hints_mode == 0
    assert d.next.redo_armed == 7
    # ---- and rest again: same camera, but the frame right behind the jump still has regions sized before it (the scan sizes
    # the slot two frames ahead): it carries the redo; the one after it has the jump frame's regions
    seen_redo = 0
    for f in range(10):
        d = D.step(a)
        seen_redo += d.redo
        assert d.moved == (1 if f == 0 else 0) and d.redo == d.moved
        if f >= 3:
            assert d.start_hints_mode == 1
    assert seen_redo == 1


def test_delta_thresholds_from_both_sides(L):
    """one view entry nudged to just under / exactly at each threshold"""
    def decide_with_delta(delta, **kn):
        D = Driver(L, **kn)
        D.settle()
        i_view = yaw_view(0.0).copy()
        # entry 12 (tx) is 0: |a - b| / max(1, ...) = delta exactly
        D2 = D
        D2.frame += 1
        r = (D2.frame - 1) % RING
        i = Input()
        i_view[12] = np.float32(delta)
        i.view[:] = list(i_view)
        i.proj[:] = list(PROJ)
        i.w, i.h, i.htanx, i.htany, i.focal = 1920.0, 1080.0, 1.7778, 1.0, 540.0
        i.cam[:] = [0.0, 0.0, 5.0]
        i.lowpass = 0.01
        i.tile_row0, i.n_tile_rows, i.frame_idx, i.ring_entry, i.one_pass = 0, 68, D2.frame, r, 1
        i.layout_valid, i.layout_cam, i.has_keys2, i.n_tiles = 1, D.log[-1].cam_hash, 1, 8160
        d = Decision()
        assert L.splat_policy_decide(D.k, D.st, i, d) == 0
        return d
    below = lambda t: float(np.nextafter(np.float32(t), np.float32(0)))
    d = decide_with_delta(below(DELTA_CREEP)); assert d.start_light == 1 and d.start_hints_mode >= 2
    d = decide_with_delta(DELTA_CREEP);        assert d.start_light == 0 and d.start_hints_mode >= 2
    d = decide_with_delta(below(DELTA_SLOW));  assert d.start_hints_mode >= 2
    d = decide_with_delta(DELTA_SLOW);         assert d.start_hints_mode == 0
    d = decide_with_delta(below(DELTA_JUMP));  assert d.cam_jumped == 0 and d.redo == 0
    d = decide_with_delta(DELTA_JUMP);         assert d.cam_jumped == 1 and d.redo == 1
    # COUNT_FIRST = 2: every frame whose camera moved by half a degree or more counts first; below it does not
    d = decide_with_delta(below(DELTA_SLOW), count_first=2); assert d.count_first == 0 and d.moved == 1
    d = decide_with_delta(DELTA_SLOW, count_first=2);        assert d.count_first == 1 and d.moved == 0 and d.ring_kind == 2
    # START_HINTS = 1: hints at rest only; 0: never
    d = decide_with_delta(below(DELTA_CREEP), start_hints=1); assert d.start_hints_mode == 0
    # the neighbourhood: ceil(delta * focal / 16) + 1 tiles, clamped to 2..7
    assert decide_with_delta(0.0296).hint_radius == 2 and decide_with_delta(0.0297).hint_radius == 3
    assert decide_with_delta(0.19).hint_radius == 7


def test_still_frames_threshold_and_start_hints_knob(L):
    for knob, expect in ((2, [0, 0, 0, 1, 1]), (1, [0, 0, 0, 1, 1]), (0, [0, 0, 0, 0, 0])):
        D = Driver(L, start_hints=knob)
        got = [D.step(0.3).start_hints_mode for _ in range(5)]
        assert got == expect, (knob, got)
        assert [d.early_min for d in D.log] == ([768, 384, 384, 384, 384] if knob else [768] * 5)


def test_adaptive_redo_arms_on_a_reported_overflow_and_runs_out(L):
    D = Driver(L)
    a = 0.0
    D.settle(a)
    # a pan; the third moving frame's scan reports a list beyond its region (overflow 2: the frame was skipped)
    for f in range(3):
        a += 0.05
        d = D.step(a, report=(2, 0) if f == 2 else None)
        assert d.redo == 0
    a += 0.05
    d = D.step(a)
    assert d.redo == 1 and d.next.redo_armed == 255           # armed by the peek at the frame in flight, this frame included
    # (the overflowed frame would stay in the ring for 32 frames and re-arm on every peek: take it out, as the harvest does
    # when the ring wraps, to watch the run end)
    for s in D.status:
        s["overflow"] = 0
    n = 1
    while True:
        a += 0.05
        d = D.step(a, report=None)
        if not d.redo:
            break
        n += 1
        assert n < 400
    assert n == 256
    # a camera at rest carries no redo however armed (once its slots' regions are its own: two frames)
    D.step(a); D.step(a)
    D.st.redo_armed = 100
    for _ in range(4):
        d = D.step(a)
        assert d.moved == 0 and d.redo == 0 and d.next.redo_armed == 100


@pytest.mark.parametrize("mode,expect_moving_redo", [(0, 0), (2, 1)])
def test_overflow_redo_off_and_always(L, mode, expect_moving_redo):
    D = Driver(L, overflow_redo=mode)
    a = 0.0
    D.settle(a)
    for _ in range(5):
        a += 0.05
        d = D.step(a)
        assert d.redo == expect_moving_redo
        assert d.next.redo_armed == 0                 # (the adaptive run belongs to mode 1)
    a += 1.0
    assert D.step(a).redo == expect_moving_redo       # a jump changes nothing in the fixed modes
    # evidence arms counting first in every redo mode (frames skipped -- mode 0 -- or binned twice -- mode 2; ADVICE r5: it used
    # to be armed under mode 1 only), and a frame that counts first needs no redo
    # (the six frames above fitted their regions: three in four of the known ones must have outgrown them -- 18 more)
    firsts = []
    for f in range(24):
        a += 0.05
        d = D.step(a, report=(2, 0) if mode == 0 else (0, 1))
        firsts.append(d.count_first)
        assert d.redo == (0 if d.count_first else expect_moving_redo)
    assert firsts[:18] == [0] * 18 and firsts[19:] == [1] * 5, firsts
    d = D.step(a, one_pass=0)
    assert d.redo == 0 and d.count_first == 0 and d.moved == 0      # two-pass binning: exactly sized lists, nothing to outgrow


def test_count_first_arms_at_three_in_four_binned_twice_not_below(L):
    def run(reports, **kn):
        D = Driver(L, **kn)
        a = 0.0
        D.settle(a)
        out = []
        for rep in reports:
            a += 0.05
            out.append(D.step(a, report=rep))
        return D, out
    twice, fine = (0, 1), (0, 0)
    # 3 of 4 known moving frames binned twice -> the fifth arms (it still bins optimistically), the sixth counts first
    D, out = run([twice, twice, twice, fine, fine, fine, fine])
    assert [d.count_first for d in out[:5]] == [0, 0, 0, 1, 1]
    assert out[2].next.count_first_left == 64          # the third frame knows two, both binned twice: 2 of 2 >= 3/4 -> armed
    assert out[2].moved == 1                           # ... and is itself still binned optimistically
    # exactly at the boundary: 3 of 4 arms, 2 of 4 (and 5 of 7) does not
    for reports, armed in (([twice, fine, twice, twice, fine], True), ([twice, fine, twice, fine, fine], False),
                           ([fine, fine, twice, twice, twice, fine], False)):
        D, out = run(reports)
        # the decision that sees all of `reports[:-1]` is the last one
        assert (out[-1].next.count_first_left > 0 or out[-1].count_first == 1) == armed, reports
    # one known frame is not evidence
    D, out = run([twice, fine])
    assert out[1].next.count_first_left == 0
    # COUNT_FIRST = 0: never armed; a skipped frame (overflow 2, OVERFLOW_REDO = 0) is evidence like one binned twice
    D, out = run([twice] * 6, count_first=0)
    assert all(d.count_first == 0 and d.next.count_first_left == 0 for d in out)
    D, out = run([(2, 0)] * 4, overflow_redo=0)
    assert out[-1].next.count_first_left > 0 or out[-1].count_first == 1


def test_a_count_first_run_is_64_moving_frames_and_frames_at_rest_do_not_spend_it(L):
    D = Driver(L)
    a = 0.0
    D.settle(a)
    for _ in range(3):
        a += 0.05
        D.step(a, report=(0, 1))
    counted = 0
    for f in range(200):
        a += 0.05
        d = D.step(a, report=(0, 0))
        if f == 10:       # a pause in the middle: the first two frames at rest still have regions sized in motion (two frames ahead)
            for g in range(5):           # and count first; after that the regions are this camera's and nothing is spent
                dd = D.step(a)
                counted += dd.count_first
                if g == 1:
                    left = dd.next.count_first_left
                if g >= 2:
                    assert dd.count_first == 0 and dd.moved == 0
            assert D.st.count_first_left == left
        counted += d.count_first
        if d.count_first:
            assert d.moved == 0 and d.redo == 0 and d.ring_kind == 2
    # 64 frames, then the run is over; the frames binned twice have left the evidence window by then or are outvoted
    assert counted == 64


def test_reset_forgets_armed_runs_and_ring_evidence(L):
    D = Driver(L)
    a = 0.0
    D.settle(a)
    for _ in range(4):
        a += 0.05
        D.step(a, report=(0, 1))
    assert D.st.count_first_left > 0 and D.st.redo_armed > 0
    D.reset()                       # upload_scene / set_slab / set_option / another tile grid
    assert D.st.count_first_left == 0 and D.st.redo_armed == 0 and not any(D.st.ring_kind)
    # the stale 'binned twice' words are still in the ring; nothing may arm on them
    for f in range(6):
        a += 0.05
        d = D.step(a, report=(0, 0))
        if f < 2:
            assert d.count_first == 1              # slots without regions (bootstrap), not an armed run
        assert d.next.count_first_left == 0
    assert D.log[-1].count_first == 0


def test_solo_chain_only_when_awaited_and_idle_and_pipelined(L):
    D = Driver(L)
    assert D.step(0.0, awaited=1, idle=1).solo == 1
    assert D.step(0.0, awaited=1, idle=0).solo == 0
    assert D.step(0.0, awaited=0, idle=1).solo == 0
    D = Driver(L, pipeline=0)
    assert D.step(0.0, awaited=1, idle=1).solo == 0


def test_long_lists_launch_sizes_and_walk_flavour(L):
    D = Driver(L)
    # nothing known yet: near selection on (assume long lists), the selection's grid an eighth of the tiles, one-record walk
    d = D.step(0.0)
    assert d.comp_sorts == 1 and d.near_cap == 2048 and d.select_grid == 1020 and d.pair_walk == 0
    assert (d.grid_big, d.grid_mid, d.grid_long) == (8160, 8160, 8160)
    d = D.step(0.0, awaited=1)
    assert d.select_grid == 8160                       # a frame the caller waits for, nothing known: a workgroup per tile
    # C3-like profile: longest list 10 892, 470 lists >= 2048 keys -> near selection, one-record walk (737 pairs per key)
    D.hints = dict(sort_hint=1, hint_maxlen=10892, hint_ge2048=470, hint_ge8192=3, hint_ge16384=0, hint_pairs=8025623)
    d = D.step(0.0)
    assert d.near_cap == 2048 and d.comp_sorts == 1 and d.pair_walk == 0 and d.select_grid == 1020
    assert D.step(0.0, awaited=1).select_grid == 1020   # max(1020, 470 + 58 + 16)
    D.hints["hint_ge2048"] = 4000
    assert D.step(0.0).select_grid == 2016 and D.step(0.0, awaited=1).select_grid == 4516
    # C2-like: a few dozen lists barely above 2048 keys -> sort launches sized from the hints, paired walk (316 pairs per key)
    D.hints = dict(sort_hint=1, hint_maxlen=2300, hint_ge2048=40, hint_ge8192=0, hint_ge16384=0, hint_pairs=726800)
    D.n_tiles = 3600
    d = D.step(0.0)
    assert d.near_cap == 0 and d.comp_sorts == 0 and d.pair_walk == 1
    assert (d.grid_big, d.grid_mid, d.grid_long) == (32, 188, 8)
    # the two sides of both near-selection thresholds (256 lists >= 2048 keys, or one of 8192) and of the walk's (500 pairs per key)
    D.hints["hint_ge2048"] = 255
    assert D.step(0.0).near_cap == 0
    D.hints["hint_ge2048"] = 256
    assert D.step(0.0).near_cap == 2048
    D.hints.update(hint_ge2048=40, hint_ge8192=1)
    assert D.step(0.0).near_cap == 2048
    D.hints.update(hint_ge8192=0, hint_maxlen=2048)
    assert D.step(0.0).near_cap == 0                 # no list above 2048 keys at all
    D.hints.update(hint_maxlen=1000, hint_pairs=499999)
    assert D.step(0.0).pair_walk == 1
    D.hints.update(hint_pairs=500000)
    assert D.step(0.0).pair_walk == 0
    # forced either way, and without a second key buffer the compositor cannot order long lists
    D.k.pair_mode = 1
    assert D.step(0.0).pair_walk == 1
    D.k.pair_mode = -1
    D.hints = dict(sort_hint=0, hint_maxlen=0, hint_ge2048=0, hint_ge8192=0, hint_ge16384=0, hint_pairs=0)
    d = D.step(0.0, has_keys2=0)
    assert d.comp_sorts == 0 and d.near_cap == 0
    for kn in (dict(near_cap=0), dict(early_eps=0.0), dict(fused_sort_max=1024), dict(sort_in_comp=0)):
        D2 = Driver(L, **kn)
        d = D2.step(0.0)
        assert d.near_cap == 0, kn
    # C5-like, near selection off: the compositor still orders the long lists when they carry most of the frame
    D2 = Driver(L, near_cap=0)
    D2.n_tiles = 32400
    D2.hints = dict(sort_hint=1, hint_maxlen=27043, hint_ge2048=20000, hint_ge8192=300, hint_ge16384=20, hint_pairs=83800000)
    assert D2.step(0.0).comp_sorts == 1
    D2.hints["hint_pairs"] = 2048 * 32400
    assert D2.step(0.0).comp_sorts == 0
    D2.k.tight_grids = 1
    d = D2.step(0.0)
    assert (d.grid_big, d.grid_mid, d.grid_long) == (300, 20000, 20)


def test_slab_and_target_are_part_of_the_camera(L):
    D = Driver(L)
    h0 = D.step(0.0).cam_hash
    assert D.step(0.0).cam_hash == h0
    i_frame = D.frame
    # same matrices, another slab -> another hash (regions sized for other tile rows must not count as this camera's)
    k, st, d = D.k, D.st, Decision()
    i = Input()
    i.view[:] = list(yaw_view(0.0)); i.proj[:] = list(PROJ)
    i.w, i.h, i.htanx, i.htany, i.focal = 1920.0, 1080.0, 1.7778, 1.0, 540.0
    i.cam[:] = [0.0, 0.0, 5.0]; i.lowpass = 0.01
    i.tile_row0, i.n_tile_rows, i.frame_idx, i.ring_entry, i.one_pass, i.n_tiles = 9, 9, i_frame + 1, 2, 1, 1080
    assert L.splat_policy_decide(k, st, i, d) == 0
    assert d.cam_hash != h0 and d.next.still_frames == 0
    i.tile_row0, i.n_tile_rows = 0, 68
    i.lowpass = 0.3
    assert L.splat_policy_decide(k, st, i, d) == 0 and d.cam_hash != h0
    # a NaN in the matrices counts as a jump (delta 1), never as rest
    i.lowpass = 0.01
    i.view[3] = float("nan")
    assert L.splat_policy_decide(k, st, i, d) == 0 and d.cam_jumped == 1 and d.cam_delta == 1.0
    # ... and a focal length that is not a number decides nothing undefined
    for bad in (float("nan"), float("inf"), -float("inf")):
        i.view[3] = 0.0
        i.focal = bad
        assert L.splat_policy_decide(k, st, i, d) == 0 and 2 <= d.hint_radius <= 7 and 0 <= d.layout_radius <= 12


def test_large_list_is_kept_from_a_few_hundred_large_splats_with_hysteresis(L):
    """K1 lists its large splats for bin_large_kernel only while recent frames had enough of them to pay for the launch: 256
    large splats (128 to let go again) for a frame the caller waits for or a camera in motion; for the asynchronous frames of a
    camera AT REST -- where the frame rate is the binning chain's length -- 1024 splats OUTSIDE K1's window (512 to let go).
    Nothing known yet, or a camera jump, keeps the list; the knob's 0 / negative force it on / off."""
    D = Driver(L)
    assert D.step(0.0).use_large_list == 1                 # nothing known
    D.hints.update(sort_hint=1, hint_large=33, hint_window=5)      # C2's bench pose
    D.settle()
    assert D.log[-1].use_large_list == 0
    # synchronous frames at rest: the count of large splats decides
    seen = []
    for n in (255, 256, 200, 128, 127, 255, 256):
        D.hints["hint_large"] = n
        seen.append(D.step(0.0, awaited=1).use_large_list)
    assert seen == [0, 1, 1, 1, 0, 0, 1]
    # asynchronous frames at rest: the surface scene's bench pose (7000 over the threshold, 500 outside the window) keeps none,
    # a camera parked inside the scene (thousands outside the window) does
    D.hints.update(hint_large=7065, hint_window=513)
    assert D.step(0.0).use_large_list == 1                 # (still on from the frame before: 513 >= 512, the letting-go threshold)
    D.hints["hint_window"] = 511
    assert D.step(0.0).use_large_list == 0
    seen = []
    for n in (1023, 1024, 600, 512, 511, 1023, 3065):
        D.hints["hint_window"] = n
        seen.append(D.step(0.0).use_large_list)
    assert seen == [0, 1, 1, 1, 0, 0, 1]
    # the same scene in motion (asynchronous): the count of large splats decides again
    D.hints.update(hint_large=7065, hint_window=513)
    a = 0.0
    for _ in range(4):
        a += 0.05
        assert D.step(a).use_large_list == 1
    D.hints.update(hint_large=0, hint_window=0)
    assert D.step(a + 0.05).use_large_list == 0
    assert D.step(a + 1.5).use_large_list == 1             # a jump (into the cloud, for all the host knows)
    assert D.step(a + 1.5).use_large_list == 0             # ... and the frame behind it goes by what was counted again
    d = D.step(a + 1.5, one_pass=0)
    assert d.use_large_list == 0                           # two-pass binning has no list
    for knob, want in ((0, 1), (-1, 0)):
        D2 = Driver(L, large_list_min=knob)
        D2.hints.update(sort_hint=1, hint_large=100000 if knob < 0 else 0, hint_window=100000 if knob < 0 else 0)
        D2.settle()
        assert D2.step(0.0).use_large_list == want and D2.step(0.3).use_large_list == want and D2.step(2.0).use_large_list == want


def test_layout_radius_follows_the_cameras_motion(L):
    """the regions a frame's scan builds are used two frames on: a moving camera's are sized from the longest list within the
    distance the image shifts until then (2 x delta x focal / 16 tiles, + 1, at most 12); at rest, after a jump, with a shallow
    shift (under half a tile) or switched off: from each tile's own list"""
    D = Driver(L)
    D.settle()
    assert D.log[-1].layout_radius == 0
    d = D.step(0.0005)                       # 0.0005 rad x 540 px x 2 / 16 = 0.03 tiles
    assert d.layout_radius == 0
    d = D.step(0.0005 + math.radians(1.0))   # a degree: 2 x 0.01745 x 540 / 16 = 1.18 tiles -> 2 + 1
    assert d.layout_radius == 3, d.layout_radius
    a = 0.02
    seen = []
    for _ in range(3):
        a += math.radians(3.0)
        seen.append(D.step(a).layout_radius)
    assert all(4 <= r <= 5 for r in seen), seen       # 3 degrees: 3.5 tiles -> 4 + 1
    a += math.radians(10.0)
    assert D.step(a).layout_radius == 12              # capped
    a += 2.0
    d = D.step(a)
    assert d.cam_jumped == 1 and d.layout_radius == 0
    assert D.step(a).layout_radius == 0               # at rest again
    D1 = Driver(L, pipeline=2)
    D1.settle()
    assert D1.step(math.radians(1.0)).layout_radius == 2      # one frame ahead: 0.59 tiles -> 1 + 1
    D0 = Driver(L, layout_motion=0)
    D0.settle()
    assert D0.step(math.radians(3.0)).layout_radius == 0
    assert D.step(a + 0.05, one_pass=0).layout_radius == 0


def test_a_solo_frame_sizes_for_motion_only_while_lists_have_been_outgrowing_their_regions(L):
    """the motion filter costs a synchronous frame with nothing in flight 15-35 us of its own chain: it runs there only while the
    redo launches are armed or a count-first run is on (lists HAVE outgrown regions lately); under frames in flight always"""
    D = Driver(L)
    D.settle()
    a = math.radians(3.0)
    assert D.step(a, awaited=1, idle=1).layout_radius == 0
    a += math.radians(3.0)
    assert D.step(a, awaited=1, idle=0).layout_radius >= 4        # not solo: something is in flight
    a += math.radians(3.0)
    assert D.step(a).layout_radius >= 4                           # asynchronous
    D.st.redo_armed = 5
    a += math.radians(3.0)
    assert D.step(a, awaited=1, idle=1).layout_radius >= 4
    D.st.redo_armed = 0
    D.st.count_first_left = 3
    a += math.radians(3.0)
    assert D.step(a, awaited=1, idle=1).layout_radius >= 4

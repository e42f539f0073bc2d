"""CPU tests of the pin kit (rust/, tools/pin_euc.py): the Rust sources cannot be compiled here (no rustc), so what
can be held is (i) that rust/src/ffi.rs declares exactly the functions include/splat_hip.h declares, with the struct
fields in the header's order, and (ii) that the dump comparison identifies a hidden convention setting."""
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def test_rust_ffi_declares_the_header():
    hdr = open(os.path.join(ROOT, "include", "splat_hip.h")).read()
    ffi = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    declared = set(re.findall(r"\b(splat_[a-z0-9_]+)\s*\(", hdr))
    bound = set(re.findall(r"pub fn (splat_[a-z0-9_]+)\s*\(", ffi))
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))
    # struct fields, in order (names only; the types are checked by eye against the ctypes binding's sizes)
    def c_fields(name):
        body = re.search(r"typedef struct \{([^}]*)\} %s;" % name, hdr).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                out.append(re.sub(r"\[.*?\]", "", part.strip().split()[-1]))
        return out
    def rs_fields(name):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % name, ffi, re.S).group(1)
        body = re.sub(r"//.*", "", body)
        return re.findall(r"pub ([a-z0-9_]+):", body)
    for c, r in (("splat_config", "SplatConfig"), ("splat_camera", "SplatCamera"), ("splat_stats", "SplatStats"),
                 ("splat_record", "SplatRecord")):
        assert c_fields(c) == rs_fields(r), (c, c_fields(c), rs_fields(r))


def test_rust_files_exist_and_do_not_carry_reference_code():
    for f in ("build.rs", "src/ffi.rs", "src/pipelines_hip.rs", "examples/dump_frames.rs", "README.md"):
        assert os.path.exists(os.path.join(ROOT, "rust", f)), f
    src = open(os.path.join(ROOT, "rust", "src", "pipelines_hip.rs")).read()
    assert "impl Pipeline" not in src and "fn fragment" not in src        # the euc pipeline stays in the reference
    assert "splat_render" in src and "render_to_buffer" in src


def test_pin_euc_selftest_identifies_a_hidden_setting():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pin_euc.py"), "--selftest"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "selftest ok" in r.stdout


def _pin():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pin_euc
    return pin_euc


def test_committed_candidates_are_what_the_oracle_renders():
    """tests/golden/pin_candidates.npz = the four dump frames of rust/examples/dump_frames.rs as the oracle renders them
    under all 16 convention settings (tools/pin_euc.py --write-candidates).  The moment ONE run of the reference exists,
    `tools/pin_euc.py --against-candidates DIR` names euc's setting from this file alone."""
    import tempfile
    import numpy as np
    P = _pin()
    cand, st = P.load_candidates()
    assert len(st) == 16 and [P.label(a) for a in st] == [P.label(b) for b in P.SETTINGS]
    assert sorted(cand) == ["c1_256x256_p01.raw", "naive_1280x720_p01_identity.raw", "naive_1280x720_p02.raw", "naive_800x600_p01.raw"]
    with tempfile.TemporaryDirectory() as d:
        fr = P.frames(P.c1_scene_ply(os.path.join(d, "c1.ply")))
        for name, (scene, cam, lowpass, (h, w)) in fr.items():
            for si, k in enumerate(P.SETTINGS):
                assert np.array_equal(P.render(scene, cam, lowpass, k), cand[name][si]), (name, P.label(k))
        # ... and dumps written under a hidden setting come back as that setting, from the file alone
        hidden = 6
        for name in cand:
            cand[name][hidden].astype("<u4").tofile(os.path.join(d, name))
        best = P.against_candidates(d, out=open(os.devnull, "w"))
        for name, si in best.items():
            assert np.array_equal(cand[name][si], cand[name][hidden]), (name, si)
    # the settings are not all the same picture: the file can tell them apart where the scene lets it
    c1 = cand["c1_256x256_p01.raw"]
    assert len({c.tobytes() for c in c1}) >= 8

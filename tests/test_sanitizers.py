"""SURVEY section 5 (race / memory checking): the oracle -- the checker every parity claim rests on -- runs clean under
AddressSanitizer + UndefinedBehaviorSanitizer: threaded rasterisers, degenerate inputs, the PLY reader.  CPU only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "san"], capture_output=True, text=True)
    if r.returncode != 0 and ("cannot find -lasan" in r.stderr or "libasan" in r.stderr or "libubsan" in r.stderr):
        pytest.skip("sanitizer runtimes are not installed in this image")
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    p = subprocess.run([os.path.join(ROOT, "oracle", "_san", "san_driver")], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-3000:])
    assert "runtime error" not in p.stderr and "ERROR: AddressSanitizer" not in p.stderr, p.stderr[-3000:]
    assert "san_driver ok" in p.stdout

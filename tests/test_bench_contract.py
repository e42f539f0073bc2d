"""The bench line as the task statement's contract reads it, checked on the COMMITTED line of the round's last build
(profiles/r07_bench.json: `python bench.py` on one MI355X) and on bench.py's argument defaults -- no GPU needed.  The cross-checks
are the ones a reviewer runs by hand: value x ms_per_step, the dominant kernel inside the frame time, frac = bytes / time / peak,
B_alg / ms_per_step below the HBM peak, BASELINE.json's metric and the configuration it is quoted on."""
import ast
import json
import os

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
LINE = os.path.join(ROOT, "profiles", "r07_bench.json")


@pytest.fixture(scope="module")
def line():
    with open(LINE) as f:
        return json.loads(f.read().strip().splitlines()[-1])


@pytest.fixture(scope="module")
def baseline():
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        return json.load(f)


def test_the_keys_of_the_contract_are_there(line):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True
    assert line["dtype"] == "f32"                      # the arithmetic type of the path, not a precision claim
    assert "synthetic" in line["data"]
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["vs_baseline"] is None                 # BASELINE.md holds no published number for this metric on this hardware


def test_metric_and_workload_are_baselines(line, baseline):
    # BASELINE.json: "frames/sec + HBM GB/s, 1.5M Gaussians @1920x1080, 1/2/4/8 MI355X" -- the line's metric is the frames/sec of it, its
    # roofline object the GB/s
    assert baseline["metric"].startswith("frames/sec") and line["metric"] == "frames_per_sec" and line["unit"] == "frames/s"
    assert "1.5M Gaussians" in baseline["metric"] and "1920" in baseline["metric"]
    # the configuration the metric is quoted on: 1.5 M Gaussians at 1920 x 1080 (C3)
    wl = line["config"]["workload"]
    assert wl.startswith("C3") and "1500000" in wl.replace(" ", "").replace(",", "") and "1920x1080" in wl, wl


def test_value_is_steps_over_time(line):
    assert line["value"] * line["ms_per_step"] == pytest.approx(1000.0, rel=1e-6)
    assert line["steps"] >= 1 and line["warmup"] >= 0
    assert line["config"]["frames_dropped"] == 0
    assert sum(line["extra_legs"]["frames_dropped"].values()) == 0       # a leg that lost a frame is not a measurement


def test_roofline_follows_from_its_own_numbers(line):
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    # achieved = algorithmic bytes per launch / the kernel's average launch duration (HIP events on its own stream)
    assert r["achieved"] == pytest.approx(r["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-6)
    # D x 48 B + W x H x 4 B
    assert r["bytes_per_launch"] == line["config"]["n_pairs"] * 48 + 1920 * 1080 * 4
    # the dominant kernel fits inside a frame, and the frame's algorithmic bytes do not exceed what HBM can move in it
    assert r["avg_launch_ms"] <= line["ms_per_step"] * 1.02
    f = line["roofline_frame"]
    assert f["bytes_algorithmic"] / (line["ms_per_step"] * 1e-3) / 1e9 <= 8000.0
    assert f["frac"] == pytest.approx(f["achieved"] / f["peak"], rel=1e-9)
    # counter traffic per launch is there, and not far above the algorithmic bytes (no wasted re-reads)
    assert r["traffic"] is not None and 0 < r["traffic"] < 1.5 * r["bytes_per_launch"]


def test_the_frame_on_its_own_axis(line):
    v = line["roofline_frame"]["valu"]
    assert v is not None and v["simds"] == 1024
    assert v["wave_instructions_per_frame"] == pytest.approx(sum(v["by_kernel"].values()), rel=1e-9)
    cyc = (line["ms_per_step"] * 1e-3) * v["clock_ghz"] * 1e9 * v["simds"] / v["wave_instructions_per_frame"]
    assert v["cycles_per_wave_instruction_per_simd"] == pytest.approx(cyc, rel=1e-6)
    assert 2.4 <= cyc <= 4.5          # between a full-rate and a half-rate instruction's issue cost: the chip's VALU rate, not idle time


def test_cpu_baseline_is_labelled(line):
    c = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert c["kind"] == "port"        # no Rust toolchain in the image: the oracle, not the reference binary


def test_parity_is_reported_with_what_it_rests_on(line):
    p = line["parity"]
    assert p["max_channel_diff_lsb"] <= 1 and p["pixels_differing"] <= 1e-4 * 1920 * 1080
    assert "notes/util.py" in p["against"] and "ASSUMED" in p["against"].upper()


def test_bench_defaults_finish_in_minutes_and_default_to_one_gpu():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    defaults = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument" and node.args:
            name = node.args[0].value if isinstance(node.args[0], ast.Constant) else None
            for kw in node.keywords:
                if kw.arg == "default" and isinstance(kw.value, ast.Constant):
                    defaults[name] = kw.value.value
    assert defaults.get("--gpus") == 1
    assert 1 <= defaults.get("--steps") <= 1000 and 0 <= defaults.get("--warmup") <= 100

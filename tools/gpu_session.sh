#!/bin/bash
# scratch: the command list of the current gpurun call (rewritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 300 python tools/motion_probe.py --steps 1,3,10 C3s C3 > $O/motion_probe.txt 2>&1
cat $O/motion_probe.txt
timeout 900 python tools/knob_matrix.py --out $O/knob_matrix.json > $O/knob_matrix.txt 2>&1
cat $O/knob_matrix.txt | tail -25
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06f/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        L=d["extra_legs"]
        print(f, "value %.0f orbit %.0f scanning %.0f slow %.0f rand %.0f" % (d["value"], d.get("value_orbit") or 0, L["fixed_pose_scanning_every_frame_fps"], L["slow_pan_0p1_deg_per_frame_fps"], L["random_pose_sync_fps"]))
        print("   ", {k:(round(v,1) if isinstance(v,float) else v) for k,v in L.items() if "reference_loop" in k or "host_visible" in k})
        print("    c3s", {k:(round(v["frames_per_sec"]) if isinstance(v,dict) else None) for k,v in L.get("c3s_surface_scene",{}).items() if isinstance(v,dict)})
        print("    peak", d["config"]["device_bytes_peak"], "kern", {k:round(v,4) for k,v in d["kernel_ms"].items()}, "iso", {k:round(v,4) for k,v in (d["kernel_ms_isolated"] or {}).items()})
        print("    parity", d.get("parity",{}).get("max_channel_diff_lsb"), d.get("parity",{}).get("pixels_differing"), "libm", d.get("parity",{}).get("libm_exp_mode",{}).get("pixels_differing"), "dropped", L.get("frames_dropped"))
    except Exception as e: print(f, "ERR", e)
PY

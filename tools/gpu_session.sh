#!/bin/bash
# scratch: the command list of the current gpurun call (rewritten per session; results land in gpurun_out/)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
SPLAT_HOST_ZERO_COPY=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc > $O/bench_nozc.json 2> $O/bench_nozc.err; echo "bench rc $?"
for rep in 1 2; do
  SPLAT_AMD_LIB=$PWD/splat_amd/libsplat_hip_unfused.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc > $O/bench_unfused_$rep.json 2> $O/bench_unfused_$rep.err; echo "bench rc $?"
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc > $O/bench_fused_$rep.json 2> $O/bench_fused_$rep.err; echo "bench rc $?"
done
for k in 16 32 64; do
  SPLAT_KEYS_PER_GAUSSIAN=$k timeout 300 python tools/motion_probe.py --steps 3,10 --caps 2048 C3s > $O/motion_k$k.txt 2>&1
done
cat $O/motion_k*.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06a/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        L=d["extra_legs"]
        print(f, "value %.0f orbit %.0f scanning %.0f slow %.0f rand %.0f" % (d["value"], d.get("value_orbit") or 0, L["fixed_pose_scanning_every_frame_fps"], L["slow_pan_0p1_deg_per_frame_fps"], L["random_pose_sync_fps"]))
        print("   ", {k:(round(v,1) if isinstance(v,float) else v) for k,v in L.items() if "reference_loop" in k or "host_visible" in k})
        print("    c3s", {k:(round(v["frames_per_sec"]) if isinstance(v,dict) else None) for k,v in L.get("c3s_surface_scene",{}).items() if isinstance(v,dict)})
        print("    peak", d["config"]["device_bytes_peak"], "kern", {k:round(v,4) for k,v in d["kernel_ms"].items()}, "iso", {k:round(v,4) for k,v in (d["kernel_ms_isolated"] or {}).items()})
    except Exception as e: print(f, "ERR", e)
PY

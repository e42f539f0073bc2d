#!/bin/bash
# scratch: the command list of the current gpurun call (rewritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n; mkdir -p $O
for wl in C1 C2; do timeout 900 python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc $?"; done
timeout 600 python bench.py --mode fast --steps 100 > $O/bench_fast_mode.json 2> $O/bench_fast.err; echo "fast rc $?"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 600 python tools/slab_scaling.py C3 swap > $O/slab_scaling_C3_swap.txt 2>&1
timeout 600 python tools/slab_scaling.py C3 > $O/slab_scaling_C3.txt 2>&1
timeout 900 python tools/slab_scaling.py C5 swap > $O/slab_scaling_C5_swap.txt 2>&1
tail -4 $O/slab_scaling_C3_swap.txt $O/slab_scaling_C3.txt $O/slab_scaling_C5_swap.txt
for wl in C2 C3 C3s; do timeout 900 python tools/parity_sweep.py $wl $O/parity_sweep_$wl.json > $O/ps_$wl.log 2>&1; timeout 900 python tools/parity_sweep.py $wl $O/parity_sweep_libm_$wl.json libm > $O/psl_$wl.log 2>&1; tail -2 $O/ps_$wl.log $O/psl_$wl.log; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06n/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        L=d.get("extra_legs",{})
        print(f.split('/')[-1], "value %.0f" % d["value"], "orbit", round(d.get("value_orbit") or 0), "refloop", round(d.get("value_reference_loop") or 0), "rf", round(L.get("host_visible_splat_render_frame_fps",0)), "peak %.2f" % (d["config"]["device_bytes_peak"]/1e9), "par", d.get("parity",{}).get("pixels_differing"), d.get("parity",{}).get("libm_exp_mode",{}).get("pixels_differing"), "drops", {k:v for k,v in L.get("frames_dropped",{}).items() if v})
    except Exception as e: print(f, "ERR", e)
PY

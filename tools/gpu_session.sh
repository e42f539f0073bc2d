#!/bin/bash
# scratch: the command list of the current gpurun call (rewritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
for rep in 1 2; do for v in "" _sel4 _sel5; do
  SPLAT_AMD_LIB=$PWD/splat_amd/libsplat_hip$v.so timeout 300 python bench.py --orbit --steps 216 --warmup 36 --no-cpu-baseline --no-live-pmc --no-extra-legs > $O/o$v.$rep.json 2> $O/o$v.$rep.err
done; done
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
for wl in C1 C2 C5; do timeout 900 python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc $?"; done
timeout 600 python bench.py --mode fast --steps 100 > $O/bench_fast_mode.json 2> $O/bench_fast.err; echo "fast rc $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06m/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        L=d.get("extra_legs",{})
        print(f.split('/')[-1], "value %.0f" % d["value"], "orbit", round(d.get("value_orbit") or 0), "refloop", round(d.get("value_reference_loop") or 0), "rf", round(L.get("host_visible_splat_render_frame_fps",0)), "peak %.2f" % (d["config"]["device_bytes_peak"]/1e9), "iso", {k:round(v,4) for k,v in (d["kernel_ms_isolated"] or {}).items() if k in ("preprocess","scan","sort","composite")}, "par", d.get("parity",{}).get("pixels_differing"), d.get("parity",{}).get("libm_exp_mode",{}).get("pixels_differing"))
    except Exception as e: print(f, "ERR", e)
PY

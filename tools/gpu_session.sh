#!/bin/bash
# scratch: the command list of the current gpurun call (rewritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
for rep in 1 2; do
for v in "" _ab3; do
  SPLAT_AMD_LIB=$PWD/splat_amd/libsplat_hip$v.so timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-live-pmc --no-extra-legs > $O/b$v.$rep.json 2> $O/b$v.$rep.err
  SPLAT_AMD_LIB=$PWD/splat_amd/libsplat_hip$v.so timeout 300 python bench.py --orbit --steps 216 --warmup 36 --no-cpu-baseline --no-live-pmc --no-extra-legs > $O/o$v.$rep.json 2> $O/o$v.$rep.err
done; done
for k in 32 64; do
  SPLAT_KEYS_PER_GAUSSIAN=$k timeout 300 python tools/motion_probe.py --steps 3,10 --caps 2048 C3s > $O/motion_k$k.txt 2>&1
done
cat $O/motion_k*.txt
timeout 600 python tools/fuzz_parity.py 400 78000 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
timeout 600 python tools/fuzz_async.py > $O/fuzz_async.txt 2>&1; tail -4 $O/fuzz_async.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06d/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "value %.0f" % d["value"], "peak %.2f GB" % (d["config"]["device_bytes_peak"]/1e9), "kern", {k:round(v,4) for k,v in d["kernel_ms"].items() if k in("preprocess","sort","composite")}, "iso", {k:round(v,4) for k,v in (d["kernel_ms_isolated"] or {}).items() if k in("preprocess","sort","composite")})
    except Exception as e: print(f, "ERR", e)
PY

#!/bin/bash
# usage: tools/bench_env.sh "VAR=a VAR2=b" "VAR=c" ...   -- one bench.py run (no CPU baseline) per environment,
# one summary line each: frames/s, per-kernel ms inside the timed region and isolated
for envs in "$@"; do
  env $envs python bench.py --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null > /tmp/bench_env.json
  python - "$envs" <<'PY'
import json,sys
d=json.load(open("/tmp/bench_env.json"))
k=d["kernel_ms"]; i=d.get("kernel_ms_isolated") or {}
print("%-44s %8.1f fps  %.4f ms | K1 %.3f scan %.3f sort %.3f K4 %.3f | iso K1 %.3f scan %.3f sort %.3f K4 %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], k["preprocess"], k["scan"], k["sort"], k["composite"], i.get("preprocess",0), i.get("scan",0), i.get("sort",0), i.get("composite",0)))
PY
done

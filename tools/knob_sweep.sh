for env in "" "SPLAT_PIPELINE=5" "SPLAT_PIPELINE=4" "SPLAT_PIPELINE=2" "SPLAT_BIN_PRIO=1" "SPLAT_BIN_PRIO=-1" "SPLAT_EARLY_MIN=512" "SPLAT_EARLY_MIN=1024" "SPLAT_FUSED_SORT=1536"; do
  r=$(env $env timeout 300 python bench.py --no-cpu-baseline --no-extra-legs 2>>gpurun_out/b.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['frames_dropped'])")
  echo "$env -> $r"
done

#!/bin/bash
# Collect the profiles a round commits under profiles/: kernel-trace stats, three separate PMC passes
# (never combined with trace domains), the bench line.  Run on the GPU box from the repo root:
#   bash tools/profile_round.sh r02a
# Everything is wrapped in `timeout`: a rocprofv3 that hangs at exit must not eat the box.
tag=${1:-rXX}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# (--no-extra-legs: the legs render other poses and another scene -- C3s -- with the same kernels; without them every launch of
#  a kernel in these files is the headline workload's, and the counter files stay under gpurun's 64 MiB)
B="python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra-legs"
P="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-include-regex splat --output-format csv -d $out/pmc_$c -- $P > $out/pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES \
  --kernel-include-regex splat --output-format csv -d $out/pmc_SQ -- $P > $out/pmc_SQ.log 2>&1
# second SQ pass: where the parked half of K1's wave time goes (LDS issue stalls, bank conflicts, memory instructions)
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM \
  --kernel-include-regex splat --output-format csv -d $out/pmc_SQ2 -- $P > $out/pmc_SQ2.log 2>&1
# the same trace with nothing overlapping (one stream): every kernel alone on the chip
SPLAT_PIPELINE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_isolated -- $B > $out/trace_isolated.log 2>&1
timeout 300 python bench.py > $out/bench.json 2> $out/bench.err
find $out -name "*.csv" | head -40
tail -c 600 $out/bench.json

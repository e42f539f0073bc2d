#!/bin/bash
# scratch: the command list of the current gpurun call (rewritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O
for rep in 1 2; do
  timeout 300 python tools/motion_probe.py --steps 0.6,1,2,3 --caps 2048 C3 C3s > $O/motion_base.$rep.txt 2>&1
  SPLAT_DBG_PAN_HINTS=0.07 timeout 300 python tools/motion_probe.py --steps 0.6,1,2,3 --caps 2048 C3 C3s > $O/motion_pan.$rep.txt 2>&1
done
grep -h "frames/s" $O/motion_base.1.txt | cut -c1-75; echo; grep -h "frames/s" $O/motion_pan.1.txt | cut -c1-75; echo; grep -h "frames/s" $O/motion_base.2.txt | cut -c1-75; echo; grep -h "frames/s" $O/motion_pan.2.txt | cut -c1-75
SPLAT_DBG_PAN_HINTS=0.07 timeout 600 python tools/fuzz_async.py 30 5100 > $O/fuzz_async_pan.txt 2>&1; tail -n 2 $O/fuzz_async_pan.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06s; mkdir -p $O
for rep in 1 2; do
  timeout 300 python tools/motion_probe.py --steps 1,3,10 --caps 2048 C3 C3s > $O/base.$rep.txt 2>&1
  SPLAT_DBG_SELECT_TIGHT=1 timeout 300 python tools/motion_probe.py --steps 1,3,10 --caps 2048 C3 C3s > $O/tight.$rep.txt 2>&1
done
for f in base.1 tight.1 base.2 tight.2; do echo $f; grep -h "frames/s" $O/$f.txt | cut -c1-180; done

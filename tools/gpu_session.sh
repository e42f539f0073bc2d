#!/bin/bash
# scratch: the command list of the current gpurun call (rewritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
timeout 1200 python tools/knob_matrix.py --scenes C3,C3s,C5 --motions 1deg,10deg,random --out $O/knob_matrix_part.json > $O/knob_matrix.txt 2>&1
cat $O/knob_matrix.txt | tail -12

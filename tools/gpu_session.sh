#!/bin/bash
# scratch: the command list of the current gpurun call (rewritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06l; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 1200 python tools/knob_matrix.py --out $O/knob_matrix.json > $O/knob_matrix.txt 2>&1
tail -22 $O/knob_matrix.txt
timeout 300 python tools/motion_probe.py --steps 1,3,10 C3s C3 > $O/motion_probe.txt 2>&1
bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1
tail -c 400 $O/profile_round.log

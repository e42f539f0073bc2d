#!/bin/bash
# scratch: the command list of the current gpurun call (rewritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06t; mkdir -p $O
timeout 1500 python tools/parity_sweep.py C5 $O/parity_sweep_C5.json > $O/ps_C5.log 2>&1; tail -n 1 $O/ps_C5.log | cut -c1-300
timeout 1500 python tools/parity_sweep.py C5 $O/parity_sweep_libm_C5.json libm > $O/psl_C5.log 2>&1; tail -n 1 $O/psl_C5.log | cut -c1-300
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -n 2 $O/pytest.log

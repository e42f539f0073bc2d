#!/bin/bash
# scratch: the command list of the current gpurun call (rewritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06q; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 3 $O/pytest.log
for rep in 1 2; do for sp in 0 1; do
  SPLAT_DBG_SOLO_SPLIT=$sp timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc > $O/b_sp$sp.$rep.json 2> $O/b_sp$sp.$rep.err
done; done
timeout 300 python tools/fuzz_parity.py 300 950000 > $O/fuzz.txt 2>&1; tail -n 1 $O/fuzz.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06q/b_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); L=d["extra_legs"]
    print(f.split('/')[-1], "value %.0f refloop %.0f pageable %.0f literal %.0f frame %.0f render %.0f rand %.0f first %.2f jump %.2f" % (d["value"], L["reference_loop_fps"], L["reference_loop_pageable_image_fps"], L["reference_loop_literal_clear_and_render_to_buffer_fps"], L["host_visible_splat_render_frame_fps"], L["host_visible_splat_render_fps"], L["random_pose_sync_fps"], L["first_frame_ms"], L["pose_jump_ms"]), L["reference_loop_frame_equals_device_frame"])
PY

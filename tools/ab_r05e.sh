#!/bin/bash
# Round 5: projection with and without the per-Gaussian hoist (option pre_hoist), C2 and the real merged set, alternating, three rounds
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_hoist; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_raster_forward_gpu.py -x -q -m gpu -k "hoist" 2>&1 | tail -2
B="--no-cpu-baseline --no-exact --no-d2h --steps 10 --warmup 3"
for i in 1 2 3; do for h in 0 1; do
  F3DG_PRE_HOIST=$h timeout 200 python bench.py $B > $O/c2_h${h}_$i.log 2>&1
  F3DG_PRE_HOIST=$h timeout 200 python bench.py $B --data real > $O/real_h${h}_$i.log 2>&1
done; done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r05_hoist")
for f in sorted(glob.glob(O + "/*.log")):
    l = [x for x in open(f) if x.startswith("{")]
    if l:
        j = json.loads(l[-1]); st = j["roofline"]["stage_ms_per_step"]
        print(os.path.basename(f)[:-4], "value %.0f" % j["value"], {k: round(v, 3) for k, v in st.items()})
PY

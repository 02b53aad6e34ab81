#!/bin/bash
# Round 5, first contact of the rank-packed compositing kernel (render_kernel = 4) with the GPU: its tests, then C2 and the real merged
# set with render3s and with render4 at several packing thresholds (compositing ms from the HIP-event stage times of bench.py).
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_raster_forward_gpu.py -x -q -m gpu -k "packed or pretest_is_conservative" > $O/pytest_packed.log 2>&1
tail -3 $O/pytest_packed.log
B="--no-cpu-baseline --no-exact --no-d2h --steps 10 --warmup 3"
for cfg in "3 32" "4 64" "4 48" "4 32" "4 24" "4 16" "4 8" "4 0"; do
  set -- $cfg
  F3DG_RENDER_KERNEL=$1 F3DG_RENDER_PACK_TH=$2 python bench.py $B > $O/c2_k$1_th$2.log 2>&1
  F3DG_RENDER_KERNEL=$1 F3DG_RENDER_PACK_TH=$2 python bench.py $B --data real > $O/real_k$1_th$2.log 2>&1
  F3DG_RENDER_KERNEL=$1 F3DG_RENDER_PACK_TH=$2 python bench.py $B --render-mode exact > $O/c2x_k$1_th$2.log 2>&1
  F3DG_RENDER_KERNEL=$1 F3DG_RENDER_PACK_TH=$2 python bench.py $B --render-mode exact --data real > $O/realx_k$1_th$2.log 2>&1
done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r05a")
for f in sorted(glob.glob(O + "/*_k*.log")):
    line = [l for l in open(f) if l.startswith("{")]
    if not line:
        print(os.path.basename(f), "NO LINE"); continue
    j = json.loads(line[-1]); rf = j["roofline"]
    kc = rf.get("kernel_counters") or {}
    print(os.path.basename(f), "value %.0f" % j["value"], "compositing ms %.3f" % rf["ms_per_launch"], "frac %.3f" % rf["frac"],
          {k: (round(v, 3) if isinstance(v, float) else v) for k, v in kc.items() if k != "note"})
PY

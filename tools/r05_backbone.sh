#!/bin/bash
# Round 5 backbone evidence on a fresh box: the fp16 / deterministic-GroupNorm tests, C3's loops with their durations (the first 64-image
# pass is the first thing MIOpen sees of that size), the 16-bit options' effect on frames, C4-shaped steps
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_backbone}; mkdir -p $O; cd $R
( time python -m pytest tests/test_baseline_configs_gpu.py -x -q -m gpu --durations=8 -k "c3" ) > $O/c3_tests.log 2>&1
tail -15 $O/c3_tests.log
python -m pytest tests/test_python_ops_gpu.py -x -q -m gpu -k "backbone or groupnorm or group_norm" > $O/backbone_tests.log 2>&1
tail -4 $O/backbone_tests.log
python tools/backbone_frame_psnr.py bf16 fp16 > $O/frames.log 2>&1; tail -9 $O/frames.log
for cfg in "16 fp32" "16 bf16" "16 fp16" "64 bf16" "64 fp16" "64 fp32"; do
  set -- $cfg
  python bench.py --workload c4 --images $1 --steps 2 --warmup 1 --backbone $2 > $O/c4_$1_$2.log 2>&1
  python - <<PY
import json
l=[x for x in open("$O/c4_$1_$2.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); print("c4 $1 images $2:", round(j["value"],1), "views/s", round(j["ms_per_step"],1), "ms/step")
else:
    print("c4 $1 $2: no line"); print(open("$O/c4_$1_$2.log").read()[-600:])
PY
done

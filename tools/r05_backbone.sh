#!/bin/bash
# Round 5 backbone evidence (every step under its own timeout): the fp16 / deterministic-GroupNorm tests, the 16-bit options' effect on
# frames of the real image, C4-shaped steps per backbone precision, GroupNorm pass times
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_backbone}; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_python_ops_gpu.py -x -q -m gpu -s -k "backbone or groupnorm or group_norm" > $O/backbone_tests.log 2>&1
grep -E "SongUNet|passed|failed|Error" $O/backbone_tests.log | tail -8
timeout 900 python tools/backbone_frame_psnr.py bf16 fp16 > $O/frames.log 2>&1; tail -9 $O/frames.log
for cfg in "16 fp32" "16 bf16" "16 fp16" "64 bf16" "64 fp16"; do
  set -- $cfg
  timeout 900 python bench.py --workload c4 --images $1 --steps 2 --warmup 1 --backbone $2 > $O/c4_$1_$2.log 2>&1
  python - <<PY
import json
l=[x for x in open("$O/c4_$1_$2.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); print("c4 $1 images $2:", round(j["value"],1), "views/s", round(j["ms_per_step"],1), "ms/step")
else:
    print("c4 $1 $2: no line"); print(open("$O/c4_$1_$2.log").read()[-600:])
PY
done

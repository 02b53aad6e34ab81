#!/bin/bash
# Cost of keeping the distortion channel in the reference's operations inside the fast compositing arithmetic:
# builds the library with -DF3DG_FAST_EXACT_DIST, runs the C2 bench and the parity report's distortion column, restores the product build.
cd "$(dirname "$0")/.."
F3DG_EXTRA_F3DG_RENDER="-fno-slp-vectorize -DF3DG_FAST_EXACT_DIST" python -c "import importlib; importlib.import_module('f3d-gaus_amd.build').build(force=True)"
timeout 2400 /usr/local/graft/bin/gpurun --timeout 1200 -- 'python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-d2h --no-exact 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"exact distortion in fast mode:\", d[\"value_in_hbm\"], d[\"roofline\"][\"stage_ms_per_step\"])"; python tests/tools/parity_report.py 2>&1 | tail -16'
python -c "import importlib; importlib.import_module('f3d-gaus_amd.build').build(force=True)"

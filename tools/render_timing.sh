#!/bin/bash
# Builds the library with -DF3DG_TIMING, runs tools/render_timing.py on the GPU box, restores the product build.
cd "$(dirname "$0")/.."
F3DG_EXTRA_F3DG_RENDER="-fno-slp-vectorize -DF3DG_TIMING" python -c "import importlib; importlib.import_module('f3d-gaus_amd.build').build(force=True)"
timeout 1800 /usr/local/graft/bin/gpurun --timeout 900 -- "python tools/render_timing.py $* 2>&1 | tail -8"
python -c "import importlib; importlib.import_module('f3d-gaus_amd.build').build(force=True)"

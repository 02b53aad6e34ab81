#!/bin/bash
# A/B of the backward kernel's register budget (F3DG_BWD_OCC) at C5; run from the repo root in the build container
for occ in 5 6; do
F3DG_EXTRA_F3DG_BACKWARD="-fno-slp-vectorize -DF3DG_BWD_OCC=$occ" python -c "import importlib; importlib.import_module('f3d-gaus_amd.build').build(force=True)"
echo "occ $occ"; timeout 1200 /usr/local/graft/bin/gpurun --timeout 600 -- 'python tools/stress_c5.py 2>&1 | tail -2 | head -1' 2>&1 | tail -1
done
python -c "import importlib; importlib.import_module('f3d-gaus_amd.build').build(force=True)"

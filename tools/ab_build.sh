#!/bin/bash
# A/B of compile-time variants ON THE GPU BOX: every argument is "label|file.hip|extra hipcc flags|command"; the named translation unit
# is rebuilt with the extra flags (the others as they are), the library relinked, the command run. The product build is restored at
# the end.   tools/ab_build.sh "base|f3dg_render.hip||python tools/ab_render.py --label base" "x|f3dg_render.hip|-DX|python ..."
cd "$(dirname "$0")/.."
for spec in "$@"; do
    IFS='|' read -r label file flags cmd <<< "$spec"
    up=$(echo "${file%.hip}" | tr a-z A-Z)
    base=$(python - <<PY
import importlib
b = importlib.import_module("f3d-gaus_amd.build")
print(" ".join(b.EXTRA_FLAGS.get("$file", [])))
PY
)
    touch "f3d-gaus_amd/csrc/$file"
    env "F3DG_EXTRA_$up=$base $flags" python -c "import importlib; importlib.import_module('f3d-gaus_amd.build').build()" || { echo "BUILD FAILED: $label"; continue; }
    echo "== $label ($file: $flags)"
    eval "$cmd"
    touch "f3d-gaus_amd/csrc/$file"
done
python -c "import importlib; importlib.import_module('f3d-gaus_amd.build').build()"

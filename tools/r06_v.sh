#!/bin/bash
cd /root/repo; export TMPDIR=/tmp F3DG_BENCH_PMC=0; ulimit -c 0
for o in 0 1 0 1; do echo "== tile_split=$o"; F3DG_OPTIONS=tile_split=$o timeout 300 python bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), {k:round(v,3) for k,v in d['stage_ms_per_step'].items()})"; done
for o in 0 1; do echo "== tile_split=$o 512^2 x 60 views"; F3DG_OPTIONS=tile_split=$o timeout 300 python bench.py --res 512 --views 60 --gaussians 589824 --no-cpu-baseline --no-exact --no-d2h --steps 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['roofline']['stage_ms_per_step'])"; done
F3DG_OPTIONS=tile_split=1 timeout 900 python -m pytest tests/test_baseline_configs_gpu.py tests/test_fullsize_properties_gpu.py -m gpu -x -q -k "512 or c5 or C5 or large or full" 2>&1 | tail -2

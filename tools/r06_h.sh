#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=/root/repo/gpurun_out/r06h; mkdir -p $O
timeout 300 python -m pytest tests/test_raster_backward_gpu.py -m gpu -x -q 2>&1 | tail -2
c5() { grep '^{' $1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value']), {k:round(v,3) for k,v in d['stage_ms_per_step'].items()})"; }
timeout 400 python bench.py --workload c5 --steps 3 --warmup 1 > $O/c5_dense.log 2>&1; echo "c5 dense: $(c5 $O/c5_dense.log)"
timeout 400 python tools/bench_real_train.py 32 2>&1 | grep -v amdgpu.ids | tail -1
timeout 700 bash tools/pmc_kernel.sh r06h/pmc_c5_bwd5 render5_bwd --workload c5 > /dev/null 2>&1
python - <<'PY'
import json
j=json.load(open('/root/repo/gpurun_out/r06h/pmc_c5_bwd5/summary.txt'))
for k,e in j.items():
    c=e['counters']
    print(k[5:30], 'us', e.get('kernel_stats'), {x:'%.3e'%c[x] for x in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_WAVE_CYCLES','SQ_WAIT_INST_ANY','SQ_WAIT_INST_LDS','SQ_ACTIVE_INST_LDS','SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE') if x in c}, 'issue', e.get('valu_issue_frac'), 'lane', e.get('lane_utilisation'), 'lds', e.get('lds_busy_frac'), e.get('hbm_bytes'))
PY

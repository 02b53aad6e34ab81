#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=gpurun_out/r05o; mkdir -p $O; rm -rf $O/*
timeout 300 python -m pytest tests/test_small_path_gpu.py tests/test_python_ops_gpu.py -m gpu -x -q -k "deferred" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 200 python bench.py --workload dropin --steps 5 --warmup 2 2>&1 | tail -1 > $O/dropin.json
python -c "
import json; d=json.load(open('$O/dropin.json')); print({k:round(v) for k,v in d.items() if k.startswith('value')}); print(d['us_per_call'])"

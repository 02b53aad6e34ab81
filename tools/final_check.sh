#!/bin/bash
# end-of-round check on a fresh box: build, smoke, the whole GPU suite, the default bench line (what the driver runs)
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=gpurun_out/final_check; mkdir -p $O; rm -rf $O/*
timeout 600 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
timeout 400 python bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log | tail -1 | cut -c1-260

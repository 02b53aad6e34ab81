#!/bin/bash
# final check of the round: build from scratch, smoke, the whole GPU suite, the default bench line
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=gpurun_out/r05_last; mkdir -p $O; rm -rf $O/*
timeout 600 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time timeout 1200 python -m pytest tests -m gpu -q --durations=5 ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log | tail -1 | cut -c1-400

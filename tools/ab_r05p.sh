#!/bin/bash
# forwards with auxiliary planes on the small-call path: gradient tests, the one-view forward + backward step
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=gpurun_out/r05p; mkdir -p $O; rm -rf $O/*
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 200 python tools/bench_one_view_train.py 2>&1 | grep -v amdgpu.ids > $O/train1.log; cat $O/train1.log

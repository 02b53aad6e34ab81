#!/bin/bash
# GPU suite on the library that travelled (default or lab build)
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=/root/repo/gpurun_out/r06e_$1; mkdir -p $O; rm -rf $O/*
python -c "import f3dgaus_amd; from f3dgaus_amd import _lib; print(_lib.lib().f3dg_version())" 2>&1 | grep -v amdgpu.ids
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 ) > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log

#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0
F3DG_OPTIONS="bwd_dense=1" timeout 700 bash tools/pmc_kernel.sh r06h/pmc_c5_bwd5 render5_bwd --workload c5 > /dev/null 2>&1
cat gpurun_out/r06h/pmc_c5_bwd5/summary.txt

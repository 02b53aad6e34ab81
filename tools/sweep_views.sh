#!/bin/bash
# bench.py at several views-per-call settings (run on the GPU box)
for v in "$@"; do
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --views-per-call $v 2>/dev/null | tail -1 | V=$v python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ['V'], round(d['value']), round(d['ms_per_step'],2), d['roofline']['stage_ms_per_step'])"
done

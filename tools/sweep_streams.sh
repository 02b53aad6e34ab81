#!/bin/bash
for cfg in "120 1" "60 2" "40 3" "30 2" "30 4" "24 5"; do
  set -- $cfg
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --views-per-call $1 --streams $2 2>/dev/null | tail -1 | V="$cfg" python -c "import sys,json,os; d=json.loads(sys.stdin.read()); print(os.environ['V'], round(d['value']), round(d['ms_per_step'],2), d['roofline']['stage_ms_per_step'])"
done

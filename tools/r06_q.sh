#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0
S=$(date +%s); python bench.py --workload c5 --steps 3 --warmup 1 2> gpurun_out/r06q_err.txt | tail -1 > gpurun_out/r06q_bench.json; echo "wall $(( $(date +%s) - S )) s"; tail -3 gpurun_out/r06q_err.txt

python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r06q_bench.json').read())
r=d['roofline']
print('value',d['value'],'traffic',r['traffic'],'frac',r['frac'],'frac_on_counter',r.get('frac_on_counter_traffic'))
print(json.dumps(r.get('traffic_live'),indent=1))
print(json.dumps(r.get('traffic_from_profiles'),indent=1)[:400])
PY

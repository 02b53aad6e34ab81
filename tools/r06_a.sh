#!/bin/bash
# round 6, call A (lab build): baseline lines on this box, who issues the frame copy, staging-only replay, FETCH_SIZE calibration
cd /root/repo; export TMPDIR=/tmp; ulimit -c 0; O=gpurun_out/r06a; mkdir -p $O; rm -rf $O/*
T="timeout 400"
$T python bench.py --no-cpu-baseline --steps 20 --d2h-issue main > $O/bench_c2_main.log 2>&1
$T python bench.py --no-cpu-baseline --steps 20 --d2h-issue thread > $O/bench_c2_thread.log 2>&1
$T python bench.py --no-cpu-baseline --steps 20 --d2h-issue main > $O/bench_c2_main2.log 2>&1
$T python bench.py --no-cpu-baseline --data real > $O/bench_real.log 2>&1
$T python tools/replay_staging.py --data real > $O/replay_real.log 2>&1
$T python tools/replay_staging.py --data real --channels rgb_depth_alpha > $O/replay_real_lean.log 2>&1
$T python tools/replay_staging.py --data synthetic --views 120 > $O/replay_c2.log 2>&1
cd /tmp
timeout 120 /root/repo/tools/micro/gather64 > $O/gather64.log 2>&1
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  d=$O/pmc_$(echo $c | tr ' ' '_'); timeout 200 rocprofv3 --pmc $c --output-format csv -d $d -o g -- /root/repo/tools/micro/gather64 > /dev/null 2>&1
done
cd /root/repo
python - <<'PY' > $O/gather64_pmc.txt 2>&1
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r06a/pmc_*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print(f.split('/')[2], k, ['%.4e' % x for x in v])
PY
for f in bench_c2_main bench_c2_thread bench_c2_main2 bench_real; do echo "$f: $(grep '^{' $O/$f.log | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['value']), round(d['value_in_hbm']), round(d['ms_per_step'],3), round(d['ms_per_step_in_hbm'],3), d['call_ms_spread']['all_stages'], d['call_ms_spread']['slowest_call_index'], (d.get('with_d2h') or {}).get('uint8_rgb',{}).get('leg_alone_ms'), r['stage_ms_per_step'], round(r['frac'],3))")"; done
cat $O/replay_real.log | tail -40; tail -30 $O/replay_real_lean.log; tail -30 $O/replay_c2.log; cat $O/gather64.log; cat $O/gather64_pmc.txt

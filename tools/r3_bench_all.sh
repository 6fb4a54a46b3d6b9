#!/bin/bash
# bench lines of every config at HEAD (run through gpurun from the repo root): gpurun_out/<tag>/bench_<config>.json
TAG=${1:-r3/all}
mkdir -p gpurun_out/$TAG
for c in C1 C2 C3 C4; do
  timeout 900 python bench.py --config $c > gpurun_out/$TAG/bench_$c.json 2> gpurun_out/$TAG/bench_$c.err
done
timeout 1500 python bench.py --config C5 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/$TAG/bench_C5.json 2> gpurun_out/$TAG/bench_C5.err
python - <<PY
import json
for c in ('C1','C2','C3','C4','C5'):
    try:
        p=json.loads(open('gpurun_out/$TAG/bench_%s.json' % c).read().strip().splitlines()[-1])
        r=p['roofline'] or {}
        print(c,'ms',p['ms_per_step'],'value',p['value'],'recorded',p.get('recorded_step_ms_per_step'),'fwd',p.get('fwd_only_ms'),'| roofline',r.get('kernel','')[:24],r.get('avg_us'),'frac',r.get('frac'),'traffic',r.get('traffic'),r.get('traffic_frac'))
    except Exception as e: print(c,'failed',e)
PY

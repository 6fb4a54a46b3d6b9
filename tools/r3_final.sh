#!/bin/bash
# round-3 closing run: full GPU suite, the profile recipe at HEAD, bench lines of the other configs, C5 kernel stats
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
( time timeout 2400 python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/r3/t_final.log 2>&1
tail -12 gpurun_out/r3/t_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 1500 python bench.py > gpurun_out/r3/bench_default_timed.json 2> gpurun_out/r3/bench_default_timed.err ) 2>&1 | grep real
bash tools/profile_round.sh r3/p2 > /dev/null 2>&1
for c in C1 C2 C4; do
  timeout 900 python bench.py --config $c --no-cpu-baseline --steps 20 > gpurun_out/r3/bench_${c}_final.json 2> gpurun_out/r3/bench_${c}_final.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/c5_stats -- python3 bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3/bench_C5_under_rocprof.json 2> gpurun_out/r3/c5_stats.log
find gpurun_out/r3/c5_stats -name '*kernel_trace.csv' -delete
timeout 1500 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3/bench_C5_final.json 2> gpurun_out/r3/bench_C5_final.err
python - <<'PY'
import json
for c in ('C1','C2','C4','C5'):
    try:
        p=json.load(open(f'gpurun_out/r3/bench_{c}_final.json'))
        print(c,'ms',p['ms_per_step'],'value',p['value'],'recorded',p.get('recorded_step_ms_per_step'),'fwd',p.get('fwd_only_ms'))
    except Exception as e: print(c,'failed',e)
p=json.load(open('gpurun_out/r3/p2/bench_default.json'))
print('C3 ms',p['ms_per_step'],p['value'],'traffic',p['roofline'].get('traffic'),p['roofline'].get('traffic_refused'))
PY

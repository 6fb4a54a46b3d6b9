#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/stats1 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r3/bench_under_rocprof1.json 2> gpurun_out/r3/stats1.log
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r3/stats1/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:45]:
    print(f"{r['Name'][:110]:110s} {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
find gpurun_out/r3/stats1 -name '*kernel_trace.csv' -delete
timeout 1500 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3/bench_c5_a.json 2> gpurun_out/r3/bench_c5_a.err
python - <<'PY'
import json
p=json.load(open('gpurun_out/r3/bench_c5_a.json'))
print('C5 ms',p['ms_per_step'],p['value'])
for k,v in p['kernels_us'].items(): print(f"{k:32s} {v['avg_us']:11.1f} x{v['launches_per_step']:.1f}")
PY

#!/bin/bash
# round-3 GPU call 1: full gpu test suite, priority A/B, default bench
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
( time timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 ) > gpurun_out/r3/t1.log 2>&1
tail -5 gpurun_out/r3/t1.log
OPS=interact,layer0 ROUNDS=4 bash tools/ab_run.sh split_ws base s1 s3 m1 > gpurun_out/r3/ab_prio.txt 2>&1
cat gpurun_out/r3/ab_prio.txt
timeout 900 python bench.py > gpurun_out/r3/bench1.json 2> gpurun_out/r3/bench1.err
python - <<'PY'
import json
p=json.load(open('gpurun_out/r3/bench1.json'))
print(p['ms_per_step'], p['value'], p.get('fwd_only_ms'))
for k,v in p['kernels_us'].items(): print(f"{k:32s} {v['avg_us']:9.1f} x{v['launches_per_step']:.1f}")
PY

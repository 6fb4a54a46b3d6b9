#!/bin/bash
# scratch of round 4: what one gpurun call runs while a change is being tried (this state: the GPU suite and the default bench line)
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q --durations=3 ) > $O/t_full.log 2>&1
tail -7 $O/t_full.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_C3_now.json 2>/dev/null
python -c "
import json; p=json.load(open('$O/bench_C3_now.json')); print('C3', p['ms_per_step'], {n: v['avg_us'] for n, v in p['kernels_us'].items() if v['avg_us'] > 80})"

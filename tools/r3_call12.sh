#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
( timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "chunks or node_segment_sum or split_rows or bench_workload or two_hop or f3_model or f9" ) 2>&1 | tail -3
timeout 1500 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3/bench_c5_acc.json 2> gpurun_out/r3/bench_c5_acc.err
python - <<'PY'
import json
p=json.load(open('gpurun_out/r3/bench_c5_acc.json'))
print('C5 accumulate ms',p['ms_per_step'],p['value'])
PY

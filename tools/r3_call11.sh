#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "split_rows or node_segment_sum or bench_workload or two_hop or general_hypergraph" ) 2>&1 | tail -3
for m in 100000000 4096 1024 256; do
IHG_HEAVY_MAX_SEGMENTS=$m timeout 1500 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3/bench_c5_seg$m.json 2> gpurun_out/r3/bench_c5_seg$m.err
python - $m <<'PY'
import json,sys
p=json.load(open(f'gpurun_out/r3/bench_c5_seg{sys.argv[1]}.json'))
print('C5 max_segments',sys.argv[1],'ms',p['ms_per_step'],p['value'])
PY
done

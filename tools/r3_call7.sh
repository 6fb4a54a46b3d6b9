#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
( timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_passes or interact_forward_backward or persistent_tiles or tiny_hypergraphs or split_arithmetic or f8_wide or worst_case or chunks or bench_workload or recorded or driver_end" ) > gpurun_out/r3/t7.log 2>&1
tail -8 gpurun_out/r3/t7.log
for v in 0 1; do
IHG_FWD_KPASS=$v timeout 1500 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r3/bench_c5_kp$v.json 2> gpurun_out/r3/bench_c5_kp$v.err
python - $v <<'PY'
import json,sys
p=json.load(open(f'gpurun_out/r3/bench_c5_kp{sys.argv[1]}.json'))
print('C5 kpass',sys.argv[1],'ms',p['ms_per_step'],p['value'])
PY
done

#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_passes or interact_forward_backward or persistent_tiles or tiny_hypergraphs or split_arithmetic or f8_wide or f2_layers or layer0_path or worst_case or user_slot or hyperedge_cotangents or chunks or fp32_mfma" ) > gpurun_out/r3/t4.log 2>&1
tail -8 gpurun_out/r3/t4.log
OPS=interact,layer0 ROUNDS=4 bash tools/ab_run.sh split_ nopin pin > gpurun_out/r3/ab_pin.txt 2>&1
cat gpurun_out/r3/ab_pin.txt
timeout 600 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r3/bench_pin.json 2> gpurun_out/r3/bench_pin.err
python - <<'PY'
import json,sys
p=json.load(open('gpurun_out/r3/bench_pin.json'))
k=p['kernels_us']
print('ms',p['ms_per_step'],'fwd_only',p.get('fwd_only_ms'),'ifwd',k['interact_fwd']['avg_us'],'e2n',k['k7.edges_to_nodes']['avg_us'],'ibwd',k['interact_bwd']['avg_us'])
PY

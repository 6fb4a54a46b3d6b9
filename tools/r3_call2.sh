#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
( time timeout 2400 python -m pytest tests -m gpu -q --durations=12 ) > gpurun_out/r3/t2.log 2>&1
tail -25 gpurun_out/r3/t2.log
for v in 0 1 0 1; do
  IHG_LAYER0_ONE_NODE=$v timeout 600 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r3/bench_onenode_$v.json 2> gpurun_out/r3/bench_onenode_$v.err
  python - $v <<'PY'
import json,sys
p=json.load(open(f'gpurun_out/r3/bench_onenode_{sys.argv[1]}.json'))
k=p['kernels_us']
print('one_node',sys.argv[1],'ms',p['ms_per_step'],'fwd_only',p.get('fwd_only_ms'),'dense_cot',p.get('dense_last_cotangent_ms_per_step'),'f32',p.get('fp32_mfma_kernels_ms_per_step'),
      'nl_bwd',k['node_linear_bwd'],'ifwd',k['interact_fwd']['avg_us'],'ibwd',k['interact_bwd']['avg_us'])
PY
done

#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
( time timeout 2400 python -m pytest tests -m gpu -q --durations=8 ) > gpurun_out/r3/t8.log 2>&1
tail -16 gpurun_out/r3/t8.log
( time timeout 1200 python bench.py > gpurun_out/r3/bench_default_a.json 2> gpurun_out/r3/bench_default_a.err ) 2>&1 | tail -4
python - <<'PY'
import json
p=json.load(open('gpurun_out/r3/bench_default_a.json'))
print('ms',p['ms_per_step'],'value',p['value'],'recorded',p.get('recorded_step_ms_per_step'),p.get('recorded_step_error'),'f32',p.get('fp32_mfma_kernels_ms_per_step'),'dense',p.get('dense_last_cotangent_ms_per_step'),'rows',p.get('batch_rows_last_layer_ms_per_step'),'fwd',p.get('fwd_only_ms'))
print('roofline',{k:p['roofline'][k] for k in ('achieved','frac','frac_algorithmic','traffic','traffic_refused','avg_us')})
print('cpu',p['cpu_baseline']['value'],p['cpu_baseline']['cores'],p['cpu_baseline']['ms_per_step'])
print('c1',p['cpu_baseline_c1_full_size']['value'],p['cpu_baseline_c1_full_size']['gpu_same_input'])
print('interaction',p['roofline_interaction']['forward'],p['roofline_interaction']['backward'])
PY
timeout 600 python bench.py --config C2 --no-cpu-baseline --steps 20 > gpurun_out/r3/bench_c2_a.json 2> gpurun_out/r3/bench_c2_a.err
timeout 600 python bench.py --config C4 --no-cpu-baseline --steps 20 > gpurun_out/r3/bench_c4_a.json 2> gpurun_out/r3/bench_c4_a.err
python - <<'PY'
import json
for c in ('c2','c4'):
    p=json.load(open(f'gpurun_out/r3/bench_{c}_a.json'))
    print(c,'ms',p['ms_per_step'],'value',p['value'],'recorded',p.get('recorded_step_ms_per_step'),'fwd',p.get('fwd_only_ms'))
PY

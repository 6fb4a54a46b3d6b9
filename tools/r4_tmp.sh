export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "linear or f8_ or f3_ or tables_in_place or without_hyperedge or compose or framework" 2>&1 | tail -3
python bench.py --no-cpu-baseline > /tmp/b.json 2>/dev/null; python - <<'PY'
import json
p=json.load(open('/tmp/b.json')); print('C3', p['ms_per_step'], 'rec', p.get('recorded_step_ms_per_step'), 'fwd', p.get('fwd_only_ms'))
for k,v in p['kernels_us'].items():
    if 'linear' in k or 'interact' in k: print('   ',k,v)
PY

export TMPDIR=/tmp
python bench.py --no-cpu-baseline > /tmp/b.json 2>/dev/null; python - <<'PY'
import json
p=json.load(open('/tmp/b.json')); print('C3', p['ms_per_step'], 'rec', p.get('recorded_step_ms_per_step'), 'fwd', p.get('fwd_only_ms'))
for k,v in p['kernels_us'].items(): print('   ',k,v)
PY

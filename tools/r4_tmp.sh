export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "score_topk or top_items or batched_evaluation or f6_ or driver_end" 2>&1 | tail -8
python3 - <<'PY'
import json, subprocess, sys
out = subprocess.run([sys.executable, 'bench.py', '--no-cpu-baseline', '--steps', '5'], capture_output=True, text=True)
try:
    p = json.loads(out.stdout.strip().splitlines()[-1]); print('eval', p['evaluation_top10']); print('step', p['ms_per_step'])
except Exception as e:
    print('bench failed', e, out.stderr[-1500:])
PY

export TMPDIR=/tmp
O=gpurun_out/r6/step6; mkdir -p $O
( time timeout 2400 python -m pytest tests/test_gpu_parity.py -q -x -k "isolated_nodes or f10 or tables_in_place or two_ranks or recorded" ) > $O/t_new.log 2>&1
tail -8 $O/t_new.log
( time python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_C5.json 2> $O/bench_C5.err ) 2> $O/time_c5.txt
tail -3 $O/bench_C5.err
python - <<'PY'
import json
for f in ['bench_C5']:
    try:
        d=json.load(open(f'gpurun_out/r6/step6/{f}.json'))
        print(f, d['ms_per_step'], d.get('fwd_only_ms'), d.get('recorded_step_ms_per_step'), d.get('recorded_step_error'), d['config'].get('nodes_in_hyperedges'))
    except Exception as e: print(f, 'ERR', e)
PY

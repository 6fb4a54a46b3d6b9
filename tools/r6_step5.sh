export TMPDIR=/tmp
O=gpurun_out/r6/step5; mkdir -p $O
( time timeout 2400 python -m pytest tests/test_gpu_parity.py -q -x -k "isolated_nodes or f10 or multiplicities_training or tables_in_place or last_layer_backward or fused_bce or two_ranks or batch_combine" ) > $O/t_new.log 2>&1
tail -12 $O/t_new.log
( time python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_C5.json 2> $O/bench_C5.err ) 2> $O/time_c5.txt
tail -3 $O/bench_C5.err
( time IHG_COMPACT_NODES=0 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_C5_nocompact.json 2> $O/bench_C5_nocompact.err ) 2>> $O/time_c5.txt
python - <<'PY'
import json
for f in ['bench_C5','bench_C5_nocompact']:
    try:
        d=json.load(open(f'gpurun_out/r6/step5/{f}.json'))
        print(f, d['ms_per_step'], d['config'].get('nodes_in_hyperedges'), d['config'].get('compact_nodes'), {k:(v['avg_us'],v['launches_per_step']) for k,v in (d.get('kernels_us') or {}).items()})
    except Exception as e: print(f, 'ERR', e)
PY

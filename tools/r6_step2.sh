export TMPDIR=/tmp
O=gpurun_out/r6/step2; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "multiplicities or merged_list or planes or pair_sums or test_edge_gather_sum or integration_md" ) > $O/t_new.log 2>&1
tail -15 $O/t_new.log
( time python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_C5.json 2> $O/bench_C5.err ) 2> $O/time_c5.txt
tail -3 $O/bench_C5.err
python bench.py --no-cpu-baseline --no-extras > $O/bench_C3.json 2> $O/bench_C3.err
python - <<'PY'
import json
for f in ['bench_C3','bench_C5']:
    try:
        d=json.load(open(f'gpurun_out/r6/step2/{f}.json'))
        print(f, d['ms_per_step'], d['config'].get('distinct_hyperedges'), d['config'].get('two_hop_merged'), {k:(v['avg_us'],v['launches_per_step']) for k,v in (d.get('kernels_us') or {}).items()})
    except Exception as e: print(f, 'ERR', e)
PY
cat $O/time_c5.txt

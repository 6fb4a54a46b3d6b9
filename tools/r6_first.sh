export TMPDIR=/tmp
O=gpurun_out/r6/first; mkdir -p $O
python bench.py --no-cpu-baseline --no-extras > $O/bench_C3_start.json 2> $O/bench_C3_start.err
( time python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_C5_merged0.json 2> $O/bench_C5_merged0.err ) 2> $O/time_c5_0.txt
( time IHG_TWO_HOP_MERGED=1 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_C5_merged1.json 2> $O/bench_C5_merged1.err ) 2> $O/time_c5_1.txt
python - <<'PY'
import json
for f in ['bench_C3_start','bench_C5_merged0','bench_C5_merged1']:
    try:
        d=json.load(open(f'gpurun_out/r6/first/{f}.json'))
        print(f, d['ms_per_step'], {k:(v['avg_us'],v['launches_per_step']) for k,v in (d.get('kernels_us') or {}).items()})
    except Exception as e: print(f, 'ERR', e)
PY
cat $O/time_c5_*.txt

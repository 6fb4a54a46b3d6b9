export TMPDIR=/tmp
O=gpurun_out/r6/probe; mkdir -p $O
for k in 1 2 4 8; do python bench.py --union-of-ranks $k --no-cpu-baseline --no-extras > $O/bench_C3_union$k.json 2> $O/bench_C3_union$k.err; done
python bench.py --config C5 --union-of-ranks 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_C5_union8.json 2> $O/bench_C5_union8.err
python bench.py --config C2 --union-of-ranks 8 --no-cpu-baseline --no-extras > $O/bench_C2_union8.json 2> $O/bench_C2_union8.err
python bench.py --config C4 --union-of-ranks 8 --no-cpu-baseline --no-extras > $O/bench_C4_union8.json 2> $O/bench_C4_union8.err
IHG_COMPACT_NODES=1 IHG_EDGE_MULTIPLICITY=1 IHG_TWO_HOP_MERGED=1 python bench.py --no-cpu-baseline --no-extras > $O/bench_C3_forced_collapse.json 2> $O/bench_C3_forced_collapse.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6/probe/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['ms_per_step'], d['config'].get('nodes_in_hyperedges'), d['config'].get('distinct_hyperedges'))
    except Exception as e: print(f,'ERR',e)
PY

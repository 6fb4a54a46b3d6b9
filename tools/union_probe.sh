#!/bin/bash
# tools/union_probe.sh: what a rank's training step costs under the K-rank cotangent exchange, measured on ONE GPU (bench.py --union-of-ranks K: batches of 1,100 x K rows - the
# union the propagation backward runs on), with the per-kernel table; and the layout's collapses forced on at C3, where `auto` leaves them off.  Output: gpurun_out/r6/probe/
export TMPDIR=/tmp
O=gpurun_out/r6/probe; mkdir -p $O
for k in 1 8; do python bench.py --union-of-ranks $k --no-cpu-baseline > $O/table_C3_union$k.json 2> $O/table_C3_union$k.err; done
for k in 1 8; do python bench.py --config C5 --union-of-ranks $k --steps 3 --warmup 1 --no-cpu-baseline > $O/table_C5_union$k.json 2> $O/table_C5_union$k.err; done
python - <<'PY'
import json
for c in ('C3','C5'):
    a=json.load(open(f'gpurun_out/r6/probe/table_{c}_union1.json')); b=json.load(open(f'gpurun_out/r6/probe/table_{c}_union8.json'))
    print(c, a['ms_per_step'], b['ms_per_step'])
    for k,v in b['kernels_us'].items():
        u=a['kernels_us'].get(k,{'avg_us':0,'launches_per_step':0})
        d=(v['avg_us']*v['launches_per_step']-u['avg_us']*u['launches_per_step'])
        if abs(d)>20: print(f'   {k:38s} {u["avg_us"]*u["launches_per_step"]:10.1f} -> {v["avg_us"]*v["launches_per_step"]:10.1f} us')
PY

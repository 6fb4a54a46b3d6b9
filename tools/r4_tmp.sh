export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "without_hyperedge_rows or full_size_c5_node_level or c5_scaled or heaviest or f8" 2>&1 | tail -4
for v in new q; do
  if [ $v = q ]; then export IHG_NODE_FWD_Q256=1; fi
  python bench.py --config C5 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_C5_grouped_$v.json 2>/dev/null; python -c "
import json; p=json.load(open('$O/bench_C5_grouped_$v.json')); print('C5 $v', p['ms_per_step'])"
done

export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "without_hyperedge_rows or node_level or interactive or training_step or heaviest or full_size" 2>&1 | tail -5
for c in C2 C5; do
python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_${c}_q_h2.json 2>/dev/null; python -c "
import json; p=json.load(open('$O/bench_${c}_q_h2.json')); print('$c', p['ms_per_step'])"
done

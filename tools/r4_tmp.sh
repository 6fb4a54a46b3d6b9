export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 2000 python -m pytest tests/test_gpu_parity.py -q -x -k "without_hyperedge_rows or split_arithmetic or c5_scaled or full_size or f8 or f3 or f5 or training_step or worst_case or heaviest" 2>&1 | tail -4
python bench.py --config C5 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_C5_now.json 2>/dev/null; python -c "
import json; p=json.load(open('$O/bench_C5_now.json')); print('C5', p['ms_per_step'])"

export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "without_hyperedge_rows or f8_ or full_size_c5 or linear or heaviest" 2>&1 | tail -3
python bench.py --config C5 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r4/bench_C5_now.json 2>/dev/null; python -c "
import json; p=json.load(open('gpurun_out/r4/bench_C5_now.json')); print('C5', p['ms_per_step'])"

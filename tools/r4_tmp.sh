export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "member or backward or gradient or training_step or heaviest or split" 2>&1 | tail -5
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4/bench_C3_members_h2.json 2>gpurun_out/r4/bench_C3_members_h2.err; python -c "
import json; p=json.load(open('gpurun_out/r4/bench_C3_members_h2.json')); print('C3', p['ms_per_step'], p['roofline'])"

export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "segment or two_hop or pair_sums or split_rows or heavy or powerlaw or bag or gcn or full_size_properties or hyper or f3 or f8" 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_C3_now.json 2>/dev/null; python -c "
import json; p=json.load(open('$O/bench_C3_now.json')); print('C3', p['ms_per_step'], {n:v['avg_us'] for n,v in p['kernels_us'].items() if v['avg_us']>200})"
python bench.py --config C5 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_C5_now.json 2>/dev/null; python -c "
import json; p=json.load(open('$O/bench_C5_now.json')); print('C5', p['ms_per_step'])"

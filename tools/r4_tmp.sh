export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
OPS=linear ROUNDS=10 bash tools/ab_run.sh dense_weight_grad base dwgfp16 base dwgfp16 > $O/ab_dwgfp16.txt 2>&1
cat $O/ab_dwgfp16.txt
( time timeout 2400 python -m pytest tests -m gpu -q --durations=3 ) > $O/t_full.log 2>&1
tail -7 $O/t_full.log
python bench.py --config C5 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_C5_now.json 2>/dev/null; python -c "
import json; p=json.load(open('$O/bench_C5_now.json')); print('C5', p['ms_per_step'])"

export TMPDIR=/tmp
O=gpurun_out/r6/step4; mkdir -p $O
( time timeout 2400 python -m pytest tests/test_gpu_parity.py -q -k "multiplicities_training or c5_scaled or full_size_properties_bench_workload or heaviest_rows or full_size_c5 or last_layer_backward_skips or bench_launches or widths_between or without_hyperedge_rows" ) > $O/t_new.log 2>&1
tail -12 $O/t_new.log
python bench.py --config C2 --dim 96 --no-cpu-baseline --no-extras --steps 20 > $O/bench_C2_d96.json 2> $O/bench_C2_d96.err
IHG_PAD_WIDTHS=0 python bench.py --config C2 --dim 96 --no-cpu-baseline --no-extras --steps 10 > $O/bench_C2_d96_nopad.json 2> $O/bench_C2_d96_nopad.err
python bench.py --config C2 --dim 64 --no-cpu-baseline --no-extras --steps 20 > $O/bench_C2_d64.json 2> $O/bench_C2_d64.err
python bench.py --config C2 --dim 128 --no-cpu-baseline --no-extras --steps 20 > $O/bench_C2_d128.json 2> $O/bench_C2_d128.err
python - <<'PY'
import json
for f in ['bench_C2_d96','bench_C2_d96_nopad','bench_C2_d64','bench_C2_d128']:
    try:
        d=json.load(open(f'gpurun_out/r6/step4/{f}.json')); print(f, d['ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY

export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q --durations=3 ) > $O/t_full.log 2>&1
tail -6 $O/t_full.log
bash tools/r4_final.sh

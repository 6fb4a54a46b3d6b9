export TMPDIR=/tmp
O=gpurun_out/r6/step8; mkdir -p $O
( time timeout 2400 python -m pytest tests/test_gpu_parity.py -q -x -k "driver_with_two_ranks or eight_ranks or two_ranks" ) > $O/t_new.log 2>&1
tail -30 $O/t_new.log

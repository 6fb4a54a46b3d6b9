export TMPDIR=/tmp
O=gpurun_out/r6/step3; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "multiplicities or batch_scatter or batch_combine or scatter_rows or two_ranks or one_rank_rccl or bench_launches" ) > $O/t_new.log 2>&1
tail -8 $O/t_new.log
( time timeout 3000 python -m pytest tests -m gpu -q --durations=8 ) > $O/t_full.log 2>&1
tail -25 $O/t_full.log

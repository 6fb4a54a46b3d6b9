export TMPDIR=/tmp
O=gpurun_out/r6/step7; mkdir -p $O
( time timeout 2400 python -m pytest tests/test_gpu_parity.py -q -k "f3_model or top_items_at_a_width or widths_between or isolated_nodes or f5_config or f2_layers or batched_evaluation or driver_end_to_end or f9 or f7" ) > $O/t_new.log 2>&1
tail -8 $O/t_new.log

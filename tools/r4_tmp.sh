export TMPDIR=/tmp
mkdir -p gpurun_out/r4
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "c5_scaled_weight_gradients" ) 2>&1 | tail -15

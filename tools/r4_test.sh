#!/bin/bash
# GPU test pass of the round: the new tests first (fast feedback), then the whole GPU suite
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
( time timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "${1:-tables_in_place or top_items or recorded or score_topk}" ) > gpurun_out/r4/t_new.log 2>&1
tail -15 gpurun_out/r4/t_new.log
if [ "${2:-full}" = "full" ]; then
  ( time timeout 2400 python -m pytest tests -m gpu -q --durations=8 ) > gpurun_out/r4/t_full.log 2>&1
  tail -25 gpurun_out/r4/t_full.log
fi

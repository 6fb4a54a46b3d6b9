#!/bin/bash
# tools/gpu_tests.sh OUT_DIR [PYTEST_K_EXPRESSION] [full]: on the GPU box (through gpurun, from anywhere): the tests matching the expression first (fast feedback on
# what a change touched), then - with "full" - the whole GPU suite.  Logs: OUT_DIR/t_new.log, OUT_DIR/t_full.log (OUT_DIR under gpurun_out/ to get them back).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=${1:?usage: tools/gpu_tests.sh OUT_DIR [K_EXPRESSION] [full]}
mkdir -p "$O"
if [ -n "${2:-}" ]; then
  ( time timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -s -k "$2" ) > "$O/t_new.log" 2>&1
  grep -E "^F10|^F[0-9] |split arithmetic|passed|failed|error" "$O/t_new.log" | tail -40
fi
if [ "${3:-}" = "full" ]; then
  ( time timeout 3000 python -m pytest tests -m gpu -q --durations=8 ) > "$O/t_full.log" 2>&1
  tail -25 "$O/t_full.log"
fi

#!/bin/bash
# Collect the profile artefacts of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> [config]   -> gpurun_out/<tag>/...      (config: C3 by default; C5: fewer steps, every counter pass over bench.py itself)
# 1. rocprofv3 --kernel-trace --stats of the default bench (10 steps)      2. the default bench line, unprofiled
# 3. PMC passes (separate runs): FETCH_SIZE, WRITE_SIZE over 3 bench steps and over K5 / K7 launched alone (tools/kbench.py); SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE over the interact kernels
set -u
TAG=${1:-prof}
CONFIG=${2:-C3}
OUT=gpurun_out/$TAG
if [ "$CONFIG" = "C5" ]; then STEPS="--steps 3 --warmup 1"; PSTEPS="--steps 2 --warmup 1"; else STEPS="--steps 10 --warmup 3"; PSTEPS="--steps 3 --warmup 2"; fi
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --config $CONFIG $STEPS --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
if [ "$CONFIG" = "C3" ]; then python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
else python3 bench.py --config $CONFIG $STEPS --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; fi
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 bench.py --config $CONFIG $PSTEPS --no-cpu-baseline --no-extras --no-kernel-events > $OUT/pmc_$c.json 2> $OUT/pmc_$c.log
done
if [ "$CONFIG" = "C3" ]; then
for c in FETCH_SIZE WRITE_SIZE; do     # K5 and K7's hyperedge -> node launch on their own (K7's seven in-situ launches per step share one kernel name and grid)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_k57_$c -- python3 tools/kbench.py --config C3 --rounds 4 --ops k5,k7 > $OUT/pmc_k57_$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 tools/kbench.py --config C3 --rounds 3 --ops layer > $OUT/pmc_mfma.log 2>&1
else
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 bench.py --config $CONFIG $PSTEPS --no-cpu-baseline --no-extras --no-kernel-events > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.log
fi
python3 tools/k7_roles_from_trace.py $OUT/stats > $OUT/k7_by_role.json 2>> $OUT/stats.log     # K7's launches by role (one kernel name, one grid: --stats averages them)
find $OUT -name '*kernel_trace.csv' -delete        # large, and the stats / counter files carry what is quoted
ls -R $OUT | head -40

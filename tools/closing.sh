#!/bin/bash
# tools/closing.sh ROUND_TAG [notests]: the closing run of a round on the GPU box (through gpurun): the whole GPU suite, smoke, the profile recipe at HEAD for C3 and C5
# (tools/profile_round.sh: kernel stats, the default bench line, FETCH / WRITE and MFMA-busy passes), bench lines of the other configs / layers / widths, and the kernel
# sequence of one steady-state step.  Everything lands under gpurun_out/ROUND_TAG/; tools/collect_profiles.py copies what is quoted into profiles/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:?usage: tools/closing.sh ROUND_TAG [notests]}
O=gpurun_out/$TAG
mkdir -p "$O"
if [ "${2:-}" != "notests" ]; then
  ( time timeout 3000 python -m pytest tests -m gpu -q --durations=5 ) > "$O/t_full.log" 2>&1 || echo "GPU SUITE FAILED (rc $?) - profiles below are still taken"
  tail -8 "$O/t_full.log"
fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh "$TAG/final" C3 > /dev/null 2>&1
bash tools/profile_round.sh "$TAG/final_c5" C5 > /dev/null 2>&1
for c in C1 C2 C4; do
  timeout 900 python bench.py --config $c --no-cpu-baseline --steps 20 > "$O/bench_${c}_final.json" 2> "$O/bench_${c}_final.err"
done
timeout 900 python bench.py --config C2 --dim 32 --no-cpu-baseline --steps 20 > "$O/bench_C2_emb32_final.json" 2> "$O/bench_C2_emb32_final.err"
timeout 900 python bench.py --layer hgcn --no-cpu-baseline > "$O/bench_C3_hgcn_final.json" 2> "$O/bench_C3_hgcn_final.err"
timeout 900 python bench.py --layer hgcn --config C5 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > "$O/bench_C5_hgcn_final.json" 2> "$O/bench_C5_hgcn_final.err"
REPO=$(pwd)
(cd /tmp && rm -rf /tmp/steptrace && rocprofv3 --kernel-trace --output-format csv -d /tmp/steptrace -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events > /dev/null 2>&1)
python3 tools/step_kernels.py /tmp/steptrace > "$O/step_kernels_final.txt" 2>&1
python3 tools/bench_table.py "$O"/bench_*_final.json "$O/final/bench_default.json" "$O/final_c5/bench_default.json"
du -sh "$O"

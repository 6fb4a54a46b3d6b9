#!/bin/bash
# round-4 closing run: the profile recipe at HEAD for C3 and C5, bench lines of the other configs / layers / widths, smoke
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh r4/final C3 > /dev/null 2>&1
bash tools/profile_round.sh r4/final_c5 C5 > /dev/null 2>&1
for c in C1 C2 C4; do
  timeout 900 python bench.py --config $c --no-cpu-baseline --steps 20 > $O/bench_${c}_final.json 2> $O/bench_${c}_final.err
done
timeout 900 python bench.py --config C2 --dim 32 --no-cpu-baseline --steps 20 > $O/bench_C2_emb32_final.json 2> $O/bench_C2_emb32_final.err
timeout 900 python bench.py --layer hgcn --no-cpu-baseline > $O/bench_C3_hgcn_final.json 2> $O/bench_C3_hgcn_final.err
timeout 900 python bench.py --layer hgcn --config C5 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_C5_hgcn_final.json 2> $O/bench_C5_hgcn_final.err
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/steptrace -- python3 /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events > /dev/null 2>&1)
python3 tools/step_kernels.py /tmp/steptrace > $O/step_kernels_final.txt 2>&1
python - <<'PY'
import json
O='gpurun_out/r4'
for c in ('C1','C2','C4','C2_emb32','C3_hgcn','C5_hgcn'):
    try:
        p=json.load(open(f'{O}/bench_{c}_final.json'))
        print(c,'ms',p['ms_per_step'],'value',p['value'],'recorded',p.get('recorded_step_ms_per_step'),'fwd',p.get('fwd_only_ms'))
    except Exception as e: print(c,'failed',e)
for d in ('final','final_c5'):
    try:
        p=json.load(open(f'{O}/{d}/bench_default.json'))
        print(d,'ms',p['ms_per_step'],p['value'],'traffic',p['roofline'].get('traffic'),p['roofline'].get('traffic_refused'))
    except Exception as e: print(d,'failed',e)
PY
du -sh $O

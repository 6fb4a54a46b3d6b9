#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
for combo in "256 128" "256 256" "512 256" "512 512" "1024 512" "2048 512" "256 128"; do
  set -- $combo
  IHG_HEAVY_THRESHOLD=$1 IHG_HEAVY_CHUNK=$2 timeout 600 python bench.py --no-cpu-baseline --steps 20 --no-extras > gpurun_out/r3/bench_heavy_$1_$2.json 2> /dev/null
  python - $1 $2 <<'PY'
import json,sys
p=json.load(open(f'gpurun_out/r3/bench_heavy_{sys.argv[1]}_{sys.argv[2]}.json'))
print('threshold',sys.argv[1],'chunk',sys.argv[2],'ms',p['ms_per_step'],'e2n',p['roofline']['avg_us'])
PY
done

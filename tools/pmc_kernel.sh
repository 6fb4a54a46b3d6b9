#!/bin/bash
# tools/pmc_kernel.sh KERNEL_SUBSTRING "COUNTER COUNTER ..." ["more counters" ...]: per-dispatch averages of hardware counters of the kernels whose
# name contains KERNEL_SUBSTRING (one rocprofv3 --pmc pass per quoted group, around tools/kbench.py --ops interact; CONFIG / SCALE / DIM / KBENCH_OPS from the environment)
REPO=$(cd "$(dirname "$0")/.." && pwd)
pat=$1; shift
export TMPDIR=/tmp
i=0
for group in "$@"; do
  out=/tmp/pmc_$i; rm -rf $out; i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $group --kernel-trace --output-format csv -d $out -- python3 $REPO/tools/kbench.py --config ${CONFIG:-C3} --scale ${SCALE:-1} --dim ${DIM:-0} --rounds 2 --ops ${KBENCH_OPS:-interact} > /dev/null 2>&1)
  python3 - "$out" "$pat" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])) if f else []:
    if sys.argv[2] in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(f'{k:32s} {sum(v) / len(v):16.0f}   ({len(v)} dispatches)')
PY
done

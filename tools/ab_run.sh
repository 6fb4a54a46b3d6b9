#!/bin/bash
# tools/ab_run.sh KERNEL_SUBSTRING NAME...: average duration of the kernels whose name contains KERNEL_SUBSTRING under each
# build_ab/lib_NAME.so (rocprofv3 kernel trace around tools/kbench.py --ops interact; CONFIG / SCALE / DIM / ROUNDS / OPS from the environment, C3 at full scale by default)
REPO=$(cd "$(dirname "$0")/.." && pwd)
pat=$1; shift
export TMPDIR=/tmp
for name in "$@"; do
  out=/tmp/ab_$name
  rm -rf $out
  (cd /tmp && IHG_ALLOW_ABLATION_BUILD=1 IHGNN_HIP_LIBRARY=$REPO/build_ab/lib_$name.so rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $REPO/tools/kbench.py --config ${CONFIG:-C3} --scale ${SCALE:-1} --dim ${DIM:-0} --rounds ${ROUNDS:-4} --ops ${OPS:-interact} > /dev/null 2>&1)
  python3 - "$out" "$name" "$pat" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])) if f else []:
    if sys.argv[3] in r['Name']:
        print(f"{sys.argv[2]:12s} {r['Name'][28:80]:52s} {int(r['Calls']):3d} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done

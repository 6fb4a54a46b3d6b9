#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_passes or interact_forward_backward or persistent_tiles or tiny_hypergraphs or f8_wide" ) 2>&1 | tail -3
for c in 0 98304 196608 393216 786432; do
  out=/tmp/ch_$c; rm -rf $out
  (cd /tmp && IHG_FWD_CHUNK=$c rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 /root/repo/tools/kbench.py --config C3 --rounds 6 --ops ifwd,k7 > /dev/null 2>&1)
  python3 - "$out" "chunk=$c" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
tot=0
for r in csv.DictReader(open(f[0])) if f else []:
    if 'interact_fwd' in r['Name'] or 'node_segment_sum' in r['Name']:
        print(f"{sys.argv[2]:14s} {r['Name'][28:90]:62s} {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:9.1f} us total {float(r['TotalDurationNs'])/1e3/8:9.1f} us/round")
PY
done

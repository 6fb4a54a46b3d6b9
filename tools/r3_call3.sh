#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_passes or interact_forward_backward or persistent_tiles or tiny_hypergraphs or split_arithmetic or f8_wide or f2_layers or layer0_path or worst_case" ) > gpurun_out/r3/t3.log 2>&1
tail -15 gpurun_out/r3/t3.log
for v in 0 1; do
  out=/tmp/kp_$v; rm -rf $out
  (cd /tmp && IHG_FWD_KPASS=$v rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 /root/repo/tools/kbench.py --config C3 --rounds 4 --ops interact > /dev/null 2>&1)
  python3 - "$out" "kpass=$v" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])) if f else []:
    if 'interact_fwd' in r['Name'] or 'pack_planes_fwd' in r['Name']:
        print(f"{sys.argv[2]:10s} {r['Name'][28:100]:72s} {int(r['Calls']):3d} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
for v in 0 1; do
  IHG_FWD_KPASS=$v timeout 600 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r3/bench_kpass_$v.json 2> gpurun_out/r3/bench_kpass_$v.err
  python - $v <<'PY'
import json,sys
p=json.load(open(f'gpurun_out/r3/bench_kpass_{sys.argv[1]}.json'))
k=p['kernels_us']
print('kpass',sys.argv[1],'ms',p['ms_per_step'],'fwd_only',p.get('fwd_only_ms'),'ifwd',k['interact_fwd']['avg_us'],'e2n',k['k7.edges_to_nodes']['avg_us'],'ibwd',k['interact_bwd']['avg_us'])
PY
done

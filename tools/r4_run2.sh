#!/bin/bash
# round-4 run 2: ablation ladder of the gathering member-gradient kernel, counters of the node-level dense kernels on their own, the row-order (locality) probe
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
OPS=layer ROUNDS=6 bash tools/ab_run.sh interact_bwd_members base m_nogst m_nodst m_nostores m_nomem m_nody m_noloads m_nomfma m_nosplit m_nour m_noprod m_nomemory m_svconly m_mfmaonly base > $O/abl_members.txt 2>&1
cat $O/abl_members.txt
# node-level linear maps (typed row GEMM forward, weight / input gradient backward) and the node-level contraction, alone
for pat in row_gemm_split dense_weight_grad_split node_interact_fwd node_interact_weight; do
  echo "== $pat" >> $O/pmc_dense.txt
  KBENCH_OPS=linear,layer bash tools/pmc_kernel.sh $pat \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
    "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" >> $O/pmc_dense.txt 2>&1
done
cat $O/pmc_dense.txt
python3 tools/kbench.py --config C3 --rounds 8 --ops linear,layer,pairs,twohop > $O/kbench_C3_base.txt 2>&1
tail -20 $O/kbench_C3_base.txt
for c in C3 C4; do
  python3 tools/locality_probe.py --config $c > $O/locality_$c.txt 2>&1
  cat $O/locality_$c.txt
  (cd /tmp && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/loc_$c -- python3 /root/repo/tools/locality_probe.py --config $c --rounds 2 > /dev/null 2>&1)
  python3 - /tmp/loc_$c >> $O/locality_$c.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
rows = [r for r in csv.DictReader(open(f[0]))] if f else []
rows.sort(key=lambda r: int(r['Dispatch_Id']))
acc = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    k = 'two_hop' if 'node_segment_sum_kernel' in n else ('pair_sums' if 'node_pair_sums' in n else None)
    if k: acc.setdefault((k, r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
# launches in program order: per variant 2 warm-up + 2 timed launches of each kernel
seq = collections.defaultdict(list)
for (k, _), v in acc.items():
    seq[k].append(v)
for k, v in seq.items():
    for i in range(0, len(v), 4):
        grp = v[i:i + 4]
        hit = sum(x.get('TCC_HIT_sum', 0) for x in grp); miss = sum(x.get('TCC_MISS_sum', 0) for x in grp)
        print(f'{k} variant#{i // 4}: TCC hit rate {hit / max(hit + miss, 1):.3f} (hit {hit / len(grp):.3e} miss {miss / len(grp):.3e} per launch)')
PY
done
cat $O/locality_C3.txt $O/locality_C4.txt

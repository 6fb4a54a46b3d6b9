#!/bin/bash
# round-4 opening run: baselines at HEAD, the pure aggregation layer's bench lines (HGCN: C3, C5), counters of the member-gradient kernel, C5 counters
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
( time python bench.py --no-cpu-baseline > $O/bench_C3_base.json 2> $O/bench_C3_base.err ) 2>&1 | grep real
( time python bench.py --layer hgcn --no-cpu-baseline > $O/bench_C3_hgcn.json 2> $O/bench_C3_hgcn.err ) 2>&1 | grep real
KBENCH_OPS=layer bash tools/pmc_kernel.sh interact_bwd_members \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" \
  "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
  "SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE" > $O/pmc_members_base.txt 2>&1
cat $O/pmc_members_base.txt
( time python bench.py --layer hgcn --config C5 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_C5_hgcn.json 2> $O/bench_C5_hgcn.err ) 2>&1 | grep real
# C5 under the profiler: kernel stats, FETCH_SIZE / WRITE_SIZE, matrix-pipe busy (separate passes)
( time rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -- python3 bench.py --config C5 --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_C5_under_rocprof.json 2> $O/c5_stats.log ) 2>&1 | grep real
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/c5_pmc_$c -- python3 bench.py --config C5 --steps 2 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events > $O/c5_pmc_$c.json 2> $O/c5_pmc_$c.log
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/c5_pmc_mfma -- python3 bench.py --config C5 --steps 2 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-events > $O/c5_pmc_mfma.json 2> $O/c5_pmc_mfma.log
python3 tools/pmc_summary.py $O/c5_pmc_FETCH_SIZE $O/c5_pmc_WRITE_SIZE > $O/c5_pmc_summary.txt 2>&1
find $O -name '*kernel_trace.csv' -size +20M -delete
du -sh $O

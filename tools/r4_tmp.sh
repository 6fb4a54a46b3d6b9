export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
OPS=layer ROUNDS=10 bash tools/ab_run.sh node_interact_fwd_grouped base wkb wkb2 base wkb wkb2 > $O/ab_wkb2.txt 2>&1
cat $O/ab_wkb2.txt

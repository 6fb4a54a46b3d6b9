export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
OPS=linear ROUNDS=10 bash tools/ab_run.sh dense_weight_grad base dxpipe base dxpipe > $O/ab_dxpipe.txt 2>&1
cat $O/ab_dxpipe.txt

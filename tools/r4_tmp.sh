export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
OPS=linear ROUNDS=8 bash tools/ab_run.sh dense_weight_grad base d_nodw d_nodx d_nomfma d_nosplit d_noloads d_nostores d_nomem base > $O/abl_dense_weight_grad.txt 2>&1
cat $O/abl_dense_weight_grad.txt

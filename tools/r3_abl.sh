#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
OPS=ifwd ROUNDS=6 bash tools/ab_run.sh kpass pin a_nomfma a_nomem a_nofirst a_noloads a_noimg a_nosplit a_nosvc a_nostore a_svconly > gpurun_out/r3/abl_kpass.txt 2>&1
cat gpurun_out/r3/abl_kpass.txt

#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02c
mkdir -p $OUT
cd $ROOT
TOAD_GEMM_H2=1 timeout 300 python tools/h2_diag.py > $OUT/diag_h2.txt 2>&1
TOAD_GEMM_H2=0 timeout 300 python tools/h2_diag.py > $OUT/diag_bf16x3.txt 2>&1
cat $OUT/diag_h2.txt; echo; cat $OUT/diag_bf16x3.txt

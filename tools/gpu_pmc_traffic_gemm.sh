#!/bin/bash
# HBM traffic of three GEMMs of the step (per-op entry points): separate FETCH_SIZE / WRITE_SIZE passes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/${1:-traffic_gemm}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- python $ROOT/tools/pmc_gemm.py > $OUT/$c.log 2>&1
  python $ROOT/tools/pmc_table2.py $OUT/$c 3 > $OUT/$c.txt 2>&1
  rm -rf $OUT/$c/*/*.db
  grep -A3 "gemm_\|absmax\|fixup\|slab_reduce" $OUT/$c.txt | cut -c1-200
done

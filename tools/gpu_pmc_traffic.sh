#!/bin/bash
# HBM traffic of the pool kernels: separate FETCH_SIZE / WRITE_SIZE passes (KB; combined in one pass rocprofv3 aborts on gfx950)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/${1:-traffic}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- python $ROOT/tools/pmc_pool.py 100000 > $OUT/$c.log 2>&1
  python $ROOT/tools/pmc_table2.py $OUT/$c 3 > $OUT/$c.txt 2>&1
  rm -rf $OUT/$c/*/*.db
  grep -A3 "gated_pool" $OUT/$c.txt | cut -c1-200
done

#!/bin/bash
# per-kernel times of the per-slide step at the given sizes on the SHIPPED library (rocprofv3 --kernel-trace --stats): exp_stats.sh TAG "sizes"
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp
for n in ${2:-10000}; do
  python tools/exp_step_variant.py $n 300 2>&1 | grep -v amdgpu.ids | tee -a $OUT/step_stats.txt
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sprof_$n -o p -- python $ROOT/tools/exp_step_variant.py $n 60 > $OUT/sprof_$n.log 2>&1)
  python tools/summarize_rocprof.py $(find $OUT/sprof_$n -name "*kernel_stats.csv" | head -1) "N=$n" 2>&1 | grep "toad::" | tee -a $OUT/step_stats.txt
  find $OUT/sprof_$n -name "*.db" -delete; find $OUT/sprof_$n -name "*trace.csv" -delete
done

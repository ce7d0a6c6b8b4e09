#!/bin/bash
# kernel-trace stats of the fused step at small bag sizes. usage: gpu_small.sh TAG N1 N2 ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT; shift
export TMPDIR=/tmp
for n in "$@"; do
  python tools/small_bag_prof.py $n 200 | tee $OUT/wall_$n.txt
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$n -o p -- python $ROOT/tools/pmc_step.py $n 20 > $OUT/prof_$n.log 2>&1)
  python tools/summarize_rocprof.py $(find $OUT/prof_$n -name "*kernel_stats.csv" | head -1) "fused step N=$n (20 steps)" > $OUT/kernel_stats_$n.md 2>&1
  head -30 $OUT/kernel_stats_$n.md
done

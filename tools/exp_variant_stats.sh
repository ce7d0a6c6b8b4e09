#!/bin/bash
# per-kernel times of the per-slide step on the shipped library and on a variant built with the given defines: exp_variant_stats.sh TAG "DEF1 DEF2" "sizes" "kernel regex"
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp
defs=$(python -c "import sys; print(tuple(sys.argv[1].split()))" "$2")
python -c "import toad_amd.build as b; b.build(defines=$defs, tag='_var', verbose=False)" > $OUT/build_var.log 2>&1
for lib in libtoad_hip.so libtoad_hip_var.so; do
  for n in ${3:-10000}; do
    TOAD_HIP_LIB=$ROOT/toad_amd/$lib python tools/exp_step_variant.py $n 300 2>&1 | grep -v amdgpu.ids | tee -a $OUT/variant_stats.txt
    (cd /tmp && TOAD_HIP_LIB=$ROOT/toad_amd/$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vprof_${lib}_$n -o p -- python $ROOT/tools/exp_step_variant.py $n 60 > $OUT/vprof_${lib}_$n.log 2>&1)
    python tools/summarize_rocprof.py $(find $OUT/vprof_${lib}_$n -name "*kernel_stats.csv" | head -1) "$lib N=$n" 2>&1 | grep -E "${4:-toad::}" | sed "s/^/$lib N=$n /" | tee -a $OUT/variant_stats.txt
    find $OUT/vprof_${lib}_$n -name "*.db" -delete; find $OUT/vprof_${lib}_$n -name "*trace.csv" -delete
  done
done

#!/bin/bash
# A/B of library variants at several bag sizes: ab_n.sh OUTTAG "N1 N2 .." tag1 tag2 ... ("-" = the shipped library)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT; NS=$2; shift 2
for n in $NS; do
  for round in 1 2; do
    for t in "$@"; do
      if [ "$t" = "-" ]; then python tools/ab_step.py $n 100; else TOAD_HIP_LIB=$ROOT/toad_amd/libtoad_hip_$t.so python tools/ab_step.py $n 100; fi
    done
  done
done 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt

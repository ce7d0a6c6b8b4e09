#!/bin/bash
# A/B of library variants on one box: ab.sh OUTTAG N tag1 tag2 ... ("-" = the shipped library); two interleaved rounds
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT; N=$2; shift 2
for round in 1 2; do
  for t in "$@"; do
    if [ "$t" = "-" ]; then python tools/ab_step.py $N 30; else TOAD_HIP_LIB=$ROOT/toad_amd/libtoad_hip_$t.so python tools/ab_step.py $N 30; fi
  done
done 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt

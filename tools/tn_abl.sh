#!/bin/bash
# Cycle-level ablation of gemm_tn_h2_big_kernel (profiles/r02at): phase stamps of the tracing build with one piece of the step removed at a time.
# Cycles, not microseconds - removing work lowers the power draw and raises the clock, so a time ablation mostly measures power.
# Build the variants first (on the CPU box, they travel with gpurun):
#   python - <<'PY'
#   from toad_amd.build import build
#   build(defines=("TOAD_H2_TRACE=1",), tag="_trace")
#   for tag, d in (("_tr_noconv", ("TOAD_ABL_TN_NO_CONVA",)), ("_tr_nowrite", ("TOAD_ABL_TN_NO_WRITE",)), ("_tr_noread", ("TOAD_ABL_TN_NO_READ2",)),
#                  ("_tr_nodma", ("TOAD_ABL_TN_NO_DMA",)), ("_tr_nocw", ("TOAD_ABL_TN_NO_CONVA", "TOAD_ABL_TN_NO_WRITE", "TOAD_ABL_TN_NO_DMA"))):
#       build(defines=("TOAD_H2_TRACE=1",) + d, tag=tag)
#   PY
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for t in trace tr_noconv tr_nowrite tr_noread tr_nodma tr_nocw; do
  [ -f toad_amd/libtoad_hip_$t.so ] || continue
  echo "== $t"
  TOAD_TN_TRACE=1 TOAD_HIP_LIB=$ROOT/toad_amd/libtoad_hip_$t.so timeout 120 python tools/ab_step.py 100000 2 2>&1 | grep "tn trace" | tail -3 | cut -c1-330 \
    | sed 's/workgroup cycles min [0-9]* mean/mean/; s/| us min.*| step period/| step period/; s/| epilogue.*//'
done

#!/bin/bash
# second counter pass over the fused step: LDS / VALU / VMEM issue activity. usage: gpu_pmc2.sh TAG
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/${1:-pmc2}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc2 -o p -- python $ROOT/tools/pmc_step.py 100000 3 > $OUT/pmc2.log 2>&1
python $ROOT/tools/pmc_table2.py $OUT/pmc2 3 > $OUT/pmc2.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc3 -o p -- python $ROOT/tools/pmc_step.py 100000 3 > $OUT/pmc3.log 2>&1
python $ROOT/tools/pmc_table2.py $OUT/pmc3 3 > $OUT/pmc3.txt 2>&1
rm -rf $OUT/pmc2/*/*.db $OUT/pmc3/*/*.db
grep -A4 "gemm_nt_h2_big_kernel<false, false, 0>\|gemm_tn_h2" $OUT/pmc2.txt | cut -c1-400; grep -A4 "gemm_nt_h2_big_kernel<false, false, 0>\|gemm_tn_h2" $OUT/pmc3.txt | cut -c1-400; tail -3 $OUT/pmc3.log

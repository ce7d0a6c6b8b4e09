#!/bin/bash
# round-2 GPU pass B: after removing the memsets / contended atomics: suite, benches at three bag sizes, kernel stats (csv), PMC passes
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02b
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_dp_two_ranks.py > $OUT/pytest_all.log 2>&1
echo "rc=$?" >> $OUT/pytest_all.log
timeout 600 python -m pytest tests/test_gpu_dp_two_ranks.py -q -m gpu > $OUT/pytest_two_ranks.log 2>&1
echo "rc=$?" >> $OUT/pytest_two_ranks.log
timeout 400 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --config 3 --steps 50 --warmup 5 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
timeout 300 python bench.py --steps 100 --warmup 5 --patches 256 --no-cpu-baseline > $OUT/bench_256.json 2> $OUT/bench_256.err
timeout 300 python bench.py --config 2 --steps 50 --warmup 5 > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
timeout 400 python bench.py --config 4 --steps 3 --warmup 1 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
cd /tmp
for n in 100000 10000 256; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$n -o p -- python $ROOT/tools/pmc_step.py $n 8 > $OUT/prof_$n.log 2>&1
  python $ROOT/tools/summarize_rocprof.py $(find $OUT/prof_$n -name "*kernel_stats.csv" | head -1) "r02b fused step N=$n (8 steps)" > $OUT/kernel_stats_$n.md 2>&1
done
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc1 -o p -- python $ROOT/tools/pmc_step.py 100000 3 > $OUT/pmc1.log 2>&1
python $ROOT/tools/pmc_table2.py $OUT/pmc1 3 > $OUT/pmc1.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc2 -o p -- python $ROOT/tools/pmc_step.py 100000 3 > $OUT/pmc2.log 2>&1
python $ROOT/tools/pmc_table2.py $OUT/pmc2 3 > $OUT/pmc2.txt 2>&1
rm -rf $OUT/pmc1/*/*.db $OUT/pmc2/*/*.db 2>/dev/null
cd $ROOT
tail -4 $OUT/pytest_all.log; tail -4 $OUT/pytest_two_ranks.log; cut -c1-300 $OUT/bench.json; cat $OUT/kernel_stats_256.md | head -40

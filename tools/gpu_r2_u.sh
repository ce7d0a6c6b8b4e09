#!/bin/bash
# round-2 GPU pass U: ping-pong NT + convert-once TN kernels: suite, benches, kernel stats, PMC
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02u
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest_all.log 2>&1
echo "rc=$?" >> $OUT/pytest_all.log
timeout 400 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --config 3 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
timeout 300 python bench.py --steps 100 --warmup 5 --patches 256 --no-cpu-baseline > $OUT/bench_256.json 2> $OUT/bench_256.err
cd /tmp
for n in 100000 256; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$n -o p -- python $ROOT/tools/pmc_step.py $n 8 > $OUT/prof_$n.log 2>&1
  python $ROOT/tools/summarize_rocprof.py $(find $OUT/prof_$n -name "*kernel_stats.csv" | head -1) "r02u fused step N=$n (8 steps)" > $OUT/kernel_stats_$n.md 2>&1
done
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc1 -o p -- python $ROOT/tools/pmc_step.py 100000 3 > $OUT/pmc1.log 2>&1
python $ROOT/tools/pmc_table2.py $OUT/pmc1 3 > $OUT/pmc1.txt 2>&1
rm -rf $OUT/pmc1/*/*.db 2>/dev/null
cd $ROOT
tail -6 $OUT/pytest_all.log; cut -c1-200 $OUT/bench.json; cat $OUT/kernel_stats_100000.md | head -12

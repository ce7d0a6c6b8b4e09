#!/bin/bash
# round-2 closing GPU pass: whole suite, every bench configuration, kernel stats, counter pass. usage: gpu_r2_final.sh TAG
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02final}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest_all.log 2>&1
echo "rc=$?" >> $OUT/pytest_all.log
timeout 500 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --config 2 --steps 50 --warmup 5 > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
timeout 300 python bench.py --config 3 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
timeout 400 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
timeout 300 python bench.py --steps 100 --warmup 5 --patches 256 --no-cpu-baseline > $OUT/bench_256.json 2> $OUT/bench_256.err
cd /tmp
for n in 100000 10000 256; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$n -o p -- python $ROOT/tools/pmc_step.py $n 8 > $OUT/prof_$n.log 2>&1
  python $ROOT/tools/summarize_rocprof.py $(find $OUT/prof_$n -name "*kernel_stats.csv" | head -1) "$TAG fused step N=$n (8 steps)" > $OUT/kernel_stats_$n.md 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o p -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-dropin > $OUT/prof_bench.log 2>&1
python $ROOT/tools/summarize_rocprof.py $(find $OUT/prof_bench -name "*kernel_stats.csv" | head -1) "$TAG rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-dropin" > $OUT/bench_kernel_stats.md 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc1 -o p -- python $ROOT/tools/pmc_step.py 100000 3 > $OUT/pmc1.log 2>&1
python $ROOT/tools/pmc_table2.py $OUT/pmc1 3 > $OUT/pmc1.txt 2>&1
rm -rf $OUT/pmc1/*/*.db $OUT/prof_*/*/*.db 2>/dev/null
find $OUT -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
cd $ROOT
tail -4 $OUT/pytest_all.log
for f in bench bench_cfg2 bench_cfg3 bench_cfg4 bench_256; do cut -c1-160 $OUT/$f.json | tail -1; done

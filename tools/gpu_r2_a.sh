#!/bin/bash
# round-2 GPU pass A: new fp16 two-piece kernels + whole-slide calls: kernel tests (asm split, plain-C split as the fallback arm),
# the full GPU suite, the headline bench (h2 on / off on the same box), rocprofv3 kernel stats.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r02a
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
rocm-smi --showclocks > $OUT/clocks.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_kernels.py -q -m gpu -x > $OUT/pytest_kernels.log 2>&1
echo "rc=$?" >> $OUT/pytest_kernels.log
if ! grep -q "rc=0" $OUT/pytest_kernels.log; then
  TOAD_HIP_LIB=$ROOT/toad_amd/libtoad_hip_csplit.so timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_kernels.py -q -m gpu -x > $OUT/pytest_kernels_csplit.log 2>&1
  echo "rc=$?" >> $OUT/pytest_kernels_csplit.log
fi
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest_all.log 2>&1
echo "rc=$?" >> $OUT/pytest_all.log
timeout 400 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
TOAD_GEMM_H2=0 timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_h2off.json 2> $OUT/bench_h2off.err
timeout 400 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > $OUT/bench_200.json 2> $OUT/bench_200.err
timeout 300 python bench.py --steps 50 --warmup 5 --patches 10000 --no-cpu-baseline > $OUT/bench_10k.json 2> $OUT/bench_10k.err
timeout 300 python bench.py --steps 100 --warmup 5 --patches 256 --no-cpu-baseline > $OUT/bench_256.json 2> $OUT/bench_256.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
cd $ROOT && python tools/summarize_rocprof.py $(find $OUT/prof -name "*kernel_stats.csv" | head -1) "r02a bench N=100k" > $OUT/kernel_stats.md 2>&1 || true
ls -R $OUT | head -50
tail -5 $OUT/pytest_kernels.log; tail -15 $OUT/pytest_all.log; cat $OUT/bench.json | cut -c1-600

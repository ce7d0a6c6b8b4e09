#!/bin/bash
# experiment: stem kernel variants (shipped / -DTOAD_STEM_PREFETCH), stem kernel time from rocprofv3 + extractor tiles/s
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import toad_amd.build as b; b.build(defines=('TOAD_STEM_PREFETCH',), tag='_pf', verbose=False)" > $OUT/build_pf.log 2>&1
for lib in libtoad_hip.so libtoad_hip_pf.so; do
  run="import sys, runpy; sys.path.insert(0, '$ROOT'); import tools.ab.select_lib; sys.argv = ['extractor_bench.py', '512', '4']; runpy.run_path('$ROOT/tools/extractor_bench.py', run_name='__main__')"
  TOAD_HIP_LIB=$ROOT/toad_amd/$lib python -c "$run" 2>&1 | grep -v amdgpu.ids | sed "s/^/$lib /" | tee -a $OUT/stem.txt
  (cd /tmp && TOAD_HIP_LIB=$ROOT/toad_amd/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/xprof_$lib -o p -- python -c "$run" > $OUT/xprof_$lib.log 2>&1)
  python tools/summarize_rocprof.py $(find $OUT/xprof_$lib -name "*kernel_stats.csv" | head -1) "$lib" 2>&1 | grep -E "stem|stream_kernel<2, 4, 0>|halo" | sed "s/^/$lib /" | tee -a $OUT/stem.txt
  find $OUT/xprof_$lib -name "*.db" -delete; find $OUT/xprof_$lib -name "*trace.csv" -delete
done

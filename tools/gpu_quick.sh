#!/bin/bash
# quick GPU check: kernel-level + whole-slide tests, headline bench, kernel stats of the fused step. usage: gpu_quick.sh TAG
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-quick}
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_kernels.py "tests/test_gpu_model.py::test_backward_chain_tight_with_identical_relu_masks" "tests/test_gpu_model.py::test_fused_step_entry_is_bitwise_the_per_op_path" -q -m gpu > $OUT/pytest.log 2>&1
echo "rc=$?" >> $OUT/pytest.log
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $ROOT/tools/pmc_step.py 100000 8 > $OUT/prof.log 2>&1
python $ROOT/tools/summarize_rocprof.py $(find $OUT/prof -name "*kernel_stats.csv" | head -1) "${1:-quick} fused step N=100000 (8 steps)" > $OUT/kernel_stats.md 2>&1
cd $ROOT
tail -3 $OUT/pytest.log; python - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'sust', d.get('sustained',{}).get('value'), 'dropin', (d.get('dropin') or {}).get('ms_per_step'), 'mfma', d.get('roofline_mfma',{}).get('achieved'), d.get('op_us_per_slide'))
PY
head -14 $OUT/kernel_stats.md

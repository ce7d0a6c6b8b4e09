#!/bin/bash
# wgrad-related tests + same-box A/B against named variants. usage: gpu_tn.sh TAG variant...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT; shift
timeout 120 python tools/ab_step.py 20000 3 || { echo "PRECHECK FAILED"; exit 1; }
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_kernels.py "tests/test_gpu_model.py::test_backward_chain_tight_with_identical_relu_masks" "tests/test_gpu_model.py::test_fused_step_entry_is_bitwise_the_per_op_path" -q -m gpu -x > $OUT/pytest.log 2>&1
echo "rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
bash tools/ab.sh $(basename $OUT) 100000 - "$@"

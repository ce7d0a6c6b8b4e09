#!/bin/bash
# the driver's bench command, its rocprofv3 summary and the other BASELINE configs in one gpurun call: bench_all.sh TAG
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
T=$1; O=gpurun_out/$T; mkdir -p $O
bash tools/gpu_run.sh $T bench:"--gpus 1 --steps 20 --warmup 5"; cp $O/bench.json $O/bench_default.json
bash tools/gpu_run.sh $T bstats:"--steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0"
for c in 2 3 4 5; do bash tools/gpu_run.sh $T bench:"--config $c"; cp $O/bench.json $O/bench_config$c.json; done

#!/bin/bash
# Board power and shader clock while the fused step runs back to back (rocm-smi sampled every ~0.25 s). usage: power_sample.sh TAG [N] [command ...]
# (a command after N replaces the default load `python tools/ab_step.py N 6000`, e.g. `python tools/extractor_bench.py 512 300`)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/${1:-power}; mkdir -p $OUT; N=${2:-100000}
rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk" > $OUT/idle.txt
if [ $# -gt 2 ]; then shift 2; timeout 120 "$@" > $OUT/run.log 2>&1 & else timeout 120 python tools/ab_step.py $N 6000 > $OUT/run.log 2>&1 & fi
PID=$!
sleep 6
for i in $(seq 1 24); do
  rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk" | tr '\n' ' '; echo
  sleep 0.25
  kill -0 $PID 2>/dev/null || break
done > $OUT/samples.txt
wait $PID
echo "--- idle / caps"; cat $OUT/idle.txt; echo "--- under load"; cat $OUT/samples.txt | cut -c1-300 | head -24; tail -2 $OUT/run.log | cut -c1-200

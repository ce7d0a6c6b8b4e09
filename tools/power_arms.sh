#!/bin/bash
# Board power of every arm the closing tables are built from: idle, the register-only MFMA loop, the streaming kernels
# (zeros AND random data), and the fused step itself - rocm-smi sampled every ~0.3 s while each runs alone for ~12 s.
#   tools/power_arms.sh TAG      -> gpurun_out/TAG/power_arms.txt  (one line per arm: samples, W min / mean / max, sclk | the rate the tool printed)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/${1:-power_arms}; mkdir -p $OUT
(cd tools/ubench && hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip 2>/dev/null; hipcc --offload-arch=gfx950 -O3 -o hbm_mix hbm_mix.hip 2>/dev/null)
summ() {     # summ NAME: one line from samples_NAME.txt and arm_NAME.log
  python3 - "$1" $OUT/samples_$1.txt $OUT/arm_$1.log <<'PY'
import re, sys
name, sf, lf = sys.argv[1:4]
pw, ck = [], []
for ln in open(sf):
    m = re.search(r"Power \(W\):\s*([\d.]+)", ln)
    if m: pw.append(float(m.group(1)))
    m = re.search(r"sclk[^(]*\((\d+)Mhz\)", ln)
    if m: ck.append(int(m.group(1)))
rate = " ".join(l.strip() for l in open(lf) if re.search(r"TB/s|TFLOP/s|step\s+[\d.]+ ms", l))[:170]
if pw:
    print(f"{name:<18} samples {len(pw):2d}  W min {min(pw):6.0f} mean {sum(pw)/len(pw):6.0f} max {max(pw):6.0f}   sclk MHz mean {sum(ck)/max(len(ck),1):5.0f}   | {rate}")
else:
    print(f"{name:<18} no samples | {rate}")
PY
}
smi() { rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk" | tr '\n' ' '; echo; }
sample() {   # sample NAME cmd...: run cmd in the background, sample power while it runs
  local name=$1; shift
  timeout 90 "$@" > $OUT/arm_$name.log 2>&1 &
  local pid=$!
  sleep 4
  : > $OUT/samples_$name.txt
  for i in $(seq 1 20); do
    kill -0 $pid 2>/dev/null || break
    smi >> $OUT/samples_$name.txt
    sleep 0.2
  done
  wait $pid
  summ $name
}
{
echo "# board power per arm (rocm-smi, ~0.3 s period, each arm alone for ~12 s; tools/power_arms.sh)"
rocm-smi --showmaxpower 2>&1 | grep -i "max" | head -2
sleep 5
: > $OUT/samples_idle.txt; : > $OUT/arm_idle.log
for i in $(seq 1 8); do smi >> $OUT/samples_idle.txt; sleep 0.2; done
summ idle
sample mfma_only        tools/ubench/mfma_power 12 0 0
sample mfma_lds         tools/ubench/mfma_power 12 0 1
sample read2_zeros      tools/ubench/hbm_mix 12 1 0
sample read2_random     tools/ubench/hbm_mix 12 1 1
sample copy_zeros       tools/ubench/hbm_mix 12 2 0
sample copy_random      tools/ubench/hbm_mix 12 2 1
sample r2w1_zeros       tools/ubench/hbm_mix 12 3 0
sample r2w1_random      tools/ubench/hbm_mix 12 3 1
sample fused_step_100k  python tools/ab_step.py 100000 5000
} 2>&1 | tee $OUT/power_arms.txt

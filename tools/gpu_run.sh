#!/bin/bash
# One parameterised GPU job for `gpurun`: gpu_run.sh TAG [what ...]; results under gpurun_out/TAG/.
#   tests[:expr]  pytest -m gpu (optionally -k expr or a file list after ':')      bench[:args]  bench.py (no CPU baseline unless args say so)
#   ab:N          tools/ab_step.py on prepared and fp32 bags                         stats[:N]     rocprofv3 --kernel-trace --stats of the fused step
#   pmc[:N]       SQ counter pass of the fused step                                  smoke         __graft_entry__.smoke()
#   traffic[:N]   FETCH_SIZE / WRITE_SIZE passes of the pool kernels -> pool_traffic.json (copy to profiles/rNN_pool_traffic.json)
#   xstats[:B] / xlayers[:B] / xtraffic[:B]   extractor alone: kernel stats, per-convolution table, HBM bytes per call (PMC)
#   bstats[:args] rocprofv3 --kernel-trace --stats of `python bench.py args`             ctraffic:"B H W Cin Cout k s p"  FETCH_SIZE of one convolution layer
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-run}; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for what in "$@"; do
  key=${what%%:*}; arg=""; [[ "$what" == *:* ]] && arg=${what#*:}
  case $key in
    tests)
      if [ -z "$arg" ]; then timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1
      elif [[ "$arg" == tests/* ]]; then timeout 900 python -m pytest $arg -x -q -m gpu > $OUT/pytest.log 2>&1
      else timeout 900 python -m pytest tests -x -q -m gpu -k "$arg" > $OUT/pytest.log 2>&1; fi
      echo "rc=$?" >> $OUT/pytest.log; tail -15 $OUT/pytest.log ;;
    testsall)     # the whole GPU suite without -x: every failure of one lease in one log
      timeout ${arg:-1800} python -m pytest tests -q -m gpu > $OUT/pytest_all.log 2>&1
      echo "rc=$?" >> $OUT/pytest_all.log; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_all.log | tail -30 ;;
    bench)
      timeout 600 python bench.py ${arg:---steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 3} > $OUT/bench.json 2> $OUT/bench.err
      tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err ;;
    ab)
      for r in 1 2; do for b in prepared fp32; do TOAD_BAG=$b timeout 300 python tools/ab_step.py ${arg:-100000} 30; done; done 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt ;;
    stats)
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $ROOT/tools/pmc_step.py ${arg:-100000} 8 > $OUT/prof.log 2>&1)
      python tools/summarize_rocprof.py $(find $OUT/prof -name "*kernel_stats.csv" | head -1) "$TAG fused step N=${arg:-100000} (8 steps, raw fp32 bag)" > $OUT/kernel_stats.md 2>&1
      head -24 $OUT/kernel_stats.md ;;
    bstats)         # rocprofv3 --kernel-trace --stats of the bench command itself (the summary the bench line's kernel times are checked against)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bprof -o p -- python $ROOT/bench.py ${arg:---steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0} > $OUT/bprof.json 2> $OUT/bprof.err)
      python tools/summarize_rocprof.py $(find $OUT/bprof -name "*kernel_stats.csv" | head -1) "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py ${arg:---steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 0}" > $OUT/bench_kernel_stats.md 2>&1
      head -16 $OUT/bench_kernel_stats.md; tail -c 300 $OUT/bprof.json ;;
    pmc)
      (cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE \
          --kernel-trace --output-format csv -d $OUT/pmc -o p -- python $ROOT/tools/pmc_step.py ${arg:-100000} 3 > $OUT/pmc.log 2>&1)
      python tools/pmc_table.py $(dirname $(find $OUT/pmc -name "*counter_collection.csv" | head -1)) > $OUT/pmc_step.txt 2>&1; head -30 $OUT/pmc_step.txt ;;
    pmc2)           # LDS / issue-stall view of the same step: bank conflicts, LDS issue stalls, instruction mix
      (cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
          --kernel-trace --output-format csv -d $OUT/pmc2 -o p -- python $ROOT/tools/pmc_step.py ${arg:-100000} 3 > $OUT/pmc2.log 2>&1)
      python tools/pmc_table.py $(dirname $(find $OUT/pmc2 -name "*counter_collection.csv" | head -1)) > $OUT/pmc2_step.txt 2>&1; head -60 $OUT/pmc2_step.txt ;;
    traffic)        # HBM traffic of the pool kernels: separate FETCH_SIZE / WRITE_SIZE passes (one pass with both aborts on gfx950)
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/traffic/$c -o p -- python $ROOT/tools/pmc_pool.py ${arg:-100000} > $OUT/traffic_$c.log 2>&1)
        find $OUT/traffic/$c -name "*.db" -delete
      done
      python tools/pool_traffic_json.py $OUT/traffic ${arg:-100000} > $OUT/pool_traffic.json 2> $OUT/pool_traffic.err; cat $OUT/pool_traffic.json | head -40; tail -3 $OUT/pool_traffic.err ;;
    xstats)         # rocprofv3 kernel stats of the extractor alone (B tiles per call)
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/xprof -o p -- python $ROOT/tools/extractor_bench.py ${arg:-512} 4 > $OUT/xprof.log 2>&1)
      python tools/summarize_rocprof.py $(find $OUT/xprof -name "*kernel_stats.csv" | head -1) "$TAG extractor, ${arg:-512} tiles per call" > $OUT/extractor_kernel_stats.md 2>&1
      head -24 $OUT/extractor_kernel_stats.md; tail -2 $OUT/xprof.log ;;
    xlayers)        # per-convolution table from the xstats kernel trace (run xstats first)
      python tools/extractor_layer_times.py $(find $OUT/xprof -name "*kernel_trace.csv" | head -1) ${arg:-512} > $OUT/extractor_layer_times.txt 2>&1; cat $OUT/extractor_layer_times.txt | head -30 ;;
    xtraffic)       # HBM bytes of one extractor call: separate FETCH_SIZE / WRITE_SIZE passes -> extractor_traffic.json (copy to profiles/rNN_extractor_traffic.json)
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/xtraffic/$c -o p -- python $ROOT/tools/extractor_bench.py ${arg:-512} 1 > $OUT/xtraffic_$c.log 2>&1)
        find $OUT/xtraffic/$c -name "*.db" -delete
      done
      python tools/extractor_traffic_json.py $OUT/xtraffic ${arg:-512} 3 > $OUT/extractor_traffic.json 2> $OUT/extractor_traffic.err; head -12 $OUT/extractor_traffic.json; tail -3 $OUT/extractor_traffic.err
      find $OUT/xtraffic -name "*counter_collection.csv" -size +20M -delete ;;
    ctraffic)       # HBM fetch bytes of one convolution layer: ctraffic:"B H W Cin Cout k s p"
      (cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/ctraffic -o p -- python $ROOT/tools/conv_traffic.py $arg 3 > $OUT/ctraffic.log 2>&1)
      find $OUT/ctraffic -name "*.db" -delete
      python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/ctraffic/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE": agg[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{k:<70} n={len(v)} FETCH_SIZE {sum(v)/len(v)/1024:.1f} MB raw ({2*sum(v)/len(v)/1024:.1f} MB with the gfx950 wide-read x2)")
PY
      tail -2 $OUT/ctraffic.log ;;
    steptraffic)    # HBM bytes of every kernel of the fused step (raw fp32 bag): separate FETCH_SIZE / WRITE_SIZE passes over tools/pmc_step.py
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/steptraffic/$c -o p -- python $ROOT/tools/pmc_step.py ${arg:-100000} 3 > $OUT/steptraffic_$c.log 2>&1)
        find $OUT/steptraffic/$c -name "*.db" -delete
      done; ls $OUT/steptraffic/* | head ;;
    mfma)           # what the power cap leaves of the matrix pipe (tools/ubench/mfma_power: arms 0-3, random operands)
      (cd tools/ubench && [ -x mfma_power ] || hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip) ; timeout 120 tools/ubench/mfma_power ${arg:-2} > $OUT/mfma_power.txt 2>&1; cat $OUT/mfma_power.txt ;;
    hbm)            # streaming rates of read-only / copy / 2R:1W / 3R:1W kernels (tools/ubench/hbm_mix)
      (cd tools/ubench && [ -x hbm_mix ] || hipcc --offload-arch=gfx950 -O3 -o hbm_mix hbm_mix.hip) ; timeout 120 tools/ubench/hbm_mix 0 -1 1 > $OUT/hbm_mix.txt 2>&1; cat $OUT/hbm_mix.txt ;;     # (random data since round 6: zeros draw ~190 W less)
    closing)        # the GEMM chain's closing table from this call's ab / mfma / hbm / steptraffic outputs
      python tools/gemm_closing_table.py $OUT/ab.txt $OUT/mfma_power.txt $OUT/hbm_mix.txt $OUT/steptraffic ${arg:-100000} $OUT/power_arms.txt > $OUT/gemm_closing_table.md 2> $OUT/closing.err; cat $OUT/gemm_closing_table.md; tail -3 $OUT/closing.err ;;
    power)          # board power of every arm of the closing tables (tools/power_arms.sh) -> $OUT/power_arms.txt
      bash tools/power_arms.sh $TAG > $OUT/power_arms.log 2>&1; cat $OUT/power_arms.txt ;;
    xclosing)       # the extractor's closing table from this call's xstats trace + mfma / hbm outputs
      python tools/extractor_closing_table.py $(find $OUT/xprof -name "*kernel_trace.csv" | head -1) $OUT/mfma_power.txt $OUT/hbm_mix.txt ${arg:-512} $OUT/power_arms.txt > $OUT/extractor_closing_table.md 2> $OUT/xclosing.err; cat $OUT/extractor_closing_table.md; tail -3 $OUT/xclosing.err ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log ;;
    *) echo "unknown job $what" ;;
  esac
done

set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06i; mkdir -p $O
bash tools/gpu_run.sh r06i testsall
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sustain-seconds 3 --no-prepared-legs --no-dropin --no-ingest > $O/bench_A.json 2> $O/bench_A.err
python bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench4_A.json 2>> $O/bench_A.err
sed -i 's/constexpr int64_t kTnBatchMaxRows = 32768;/constexpr int64_t kTnBatchMaxRows = 1ll << 30;/' toad_amd/csrc/common.h
python -m toad_amd.build > $O/build.log 2>&1; tail -1 $O/build.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sustain-seconds 3 --no-prepared-legs --no-dropin --no-ingest > $O/bench_B.json 2> $O/bench_B.err
python bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench4_B.json 2>> $O/bench_B.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sustain-seconds 3 --no-prepared-legs --no-dropin --no-ingest > $O/bench_B2.json 2>> $O/bench_B.err
python - <<'PY'
import json
for f in ("bench_A","bench_B","bench_B2","bench4_A","bench4_B"):
    try:
        d=json.loads(open(f"gpurun_out/r06i/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("sustained",{}).get("value"), d.get("op_us_per_slide"), d.get("batched",{}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY

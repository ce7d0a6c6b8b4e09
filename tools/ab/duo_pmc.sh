#!/bin/bash
# PMC pass over tools/ab/duo_bench.py: matrix-pipe busy and effective clock of the 8-wave and the duo NT kernels on the same operands.
# (counters in their own rocprofv3 run with --kernel-trace only, as the pool's gpurun requires)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/${1:-r04c}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $OUT/pmc -o p -- python $ROOT/tools/ab/duo_bench.py ${2:-98304} 3 > $OUT/pmc.log 2>&1
python - <<PY
import csv, glob, collections, re
d = glob.glob("$OUT/pmc/**/p_counter_collection.csv", recursive=True)[0]
kt = glob.glob("$OUT/pmc/**/p_kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r.get("Dispatch_Id", r.get("Correlation_Id"))] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = collections.OrderedDict()
for r in csv.DictReader(open(d)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if "gemm_nt_h2" not in name: continue
    key = (r["Dispatch_Id"], name)
    rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
print("| dispatch | kernel | us | MFMA pipe busy | waves parked | issue-stalled | clock GHz | VALU insts |")
print("|---|---|---:|---:|---:|---:|---:|---:|")
for (disp, name), c in rows.items():
    us = dur.get(disp, float("nan"))
    gui = c.get("GRBM_GUI_ACTIVE", float("nan"))
    print(f"| {disp} | {name.replace('toad::','')[:60]} | {us:.1f} | {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(gui*128):.3f} | {c.get('SQ_WAIT_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1):.3f} | "
          f"{c.get('SQ_WAIT_INST_ANY',0)/max(c.get('SQ_WAVE_CYCLES',1),1):.3f} | {gui/8/us/1e3:.2f} | {c.get('SQ_INSTS_VALU',0):.3g} |")
PY

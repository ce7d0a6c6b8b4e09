"""Join a rocprofv3 --pmc ... --kernel-trace --output-format csv pass (files <dir>/**/p_counter_collection.csv and
p_kernel_trace.csv) into per-kernel means, grouped by kernel name AND grid/shape order of appearance within a step."""
import csv, sys, collections, re, glob
d = sys.argv[1]
cc = list(csv.DictReader(open(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0])))
agg = collections.OrderedDict()
for r in cc:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if not name.startswith("toad::"):
        continue
    key = (name, r["Dispatch_Id"])
    agg.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    agg[key]["_dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 if "End_Timestamp" in r else 0.0
names = collections.OrderedDict()
for (name, did), cs in agg.items():
    names.setdefault(name, []).append(cs)
for name, lst in names.items():
    # dispatches of one kernel repeat with the step period: print each position of the LAST step separately
    per_step = len(lst) // max(1, int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    last = lst[-per_step:] if per_step else lst
    print(f"== {name}: {len(lst)} dispatches, {per_step} per step; last step:")
    for i, cs in enumerate(last):
        print("   #%d " % i + "  ".join(f"{k}={v:.4g}" for k, v in sorted(cs.items())))

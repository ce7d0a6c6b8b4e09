"""Join rocprofv3 counter_collection.csv with kernel_trace.csv -> per-kernel mean counters."""
import csv, sys, collections, re
d = sys.argv[1]
cc = list(csv.DictReader(open(f"{d}/p_counter_collection.csv")))
kt = {r["Dispatch_Id"] if "Dispatch_Id" in r else r.get("Correlation_Id"): r for r in csv.DictReader(open(f"{d}/p_kernel_trace.csv"))}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in cc:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if not name.startswith("toad::"):
        continue
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in kt.values():
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for name, cs in agg.items():
    print(f"== {name}   calls {len(next(iter(cs.values())))}  mean dur {sum(dur[name])/max(len(dur[name]),1):.1f} us")
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} mean {sum(v)/len(v):16.1f}   min {min(v):14.1f} max {max(v):14.1f}")

"""Compact a rocprofv3 --kernel-trace --stats CSV into a short markdown table (for profiles/)."""
import csv
import re
import sys

src, title = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "rocprofv3 kernel stats"
rows = list(csv.DictReader(open(src)))
print(f"# {title}\n")
print("| kernel | calls | avg us | min us | max us | total ms | % |")
print("|---|---:|---:|---:|---:|---:|---:|")
for r in rows:
    name = r["Name"]
    name = re.sub(r"\(.*", "", name)                      # drop the argument list
    name = re.sub(r"^void ", "", name)
    if len(name) > 90:
        name = name[:87] + "..."
    print(f"| `{name}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
          f"{float(r['MaxNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['Percentage']):.2f} |")

"""Combine the separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over tools/pmc_pool.py into profiles-style JSON
(the file bench.py's roofline.traffic reads): HBM bytes per launch of the fused pool forward / backward at N patches.
FETCH_SIZE is doubled (gfx950: a wide coalesced streaming read is tallied at half its bytes - MI355X_MICROARCH.md, HBM)."""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import POOL_KERNEL_SOURCES, pool_kernel_sha
out_dir, n = sys.argv[1], int(sys.argv[2])
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out_dir}/{c}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = re.sub(r"[<(].*", "", r["Kernel_Name"]).replace("void ", "").replace("toad::", "")
        if r["Counter_Name"] == c:
            agg[name].append(float(r["Counter_Value"]))
    res[c] = {k: sum(v) / len(v) for k, v in agg.items() if k.startswith("gated_pool")}
D, L, T = 384, 512, 2
fwd = (2 * res["FETCH_SIZE"].get("gated_pool_fwd_kernel", 0) + res["WRITE_SIZE"].get("gated_pool_fwd_kernel", 0)
       + 2 * res["FETCH_SIZE"].get("gated_pool_combine_kernel", 0) + res["WRITE_SIZE"].get("gated_pool_combine_kernel", 0)) * 1024
bwd = (2 * res["FETCH_SIZE"].get("gated_pool_bwd_kernel", 0) + res["WRITE_SIZE"].get("gated_pool_bwd_kernel", 0)) * 1024
print(json.dumps({"patches": n, "kernel_source_sha256": pool_kernel_sha(), "kernel_sources": list(POOL_KERNEL_SOURCES), "pool_fwd_hbm_bytes_per_launch": fwd, "pool_bwd_hbm_bytes_per_launch": bwd,
                  "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/pmc_pool.py; KB units; FETCH_SIZE x2 (gfx950 wide-load "
                            "correction); forward = gated_pool_fwd_kernel + gated_pool_combine_kernel; backward without the dH_pool output (whole-slide path)",
                  "fetch_size_kb": res["FETCH_SIZE"], "write_size_kb": res["WRITE_SIZE"],
                  "algorithmic_bytes": {"pool_fwd": 4 * (n * (2 * D + L + T) + T * D + T + T * L), "pool_bwd_whole_slide": 4 * n * (L + 4 * D + T)}}, indent=1))

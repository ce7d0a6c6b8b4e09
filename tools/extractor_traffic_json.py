"""Combine the separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over tools/extractor_bench.py into the JSON bench_extract.py reads for
`roofline.traffic`: HBM bytes of ONE extractor call of B tiles (all kernels of the call), plus the per-kernel split.
FETCH_SIZE is doubled for kernels whose reads are wide coalesced streams (gfx950 tallies a 128-byte request at half its bytes: MI355X_MICROARCH.md, HBM /
rocprofv3 section); the raw figures are kept beside the corrected ones.
usage: python tools/extractor_traffic_json.py <dir with FETCH_SIZE/ and WRITE_SIZE/> <B> <calls in the trace>"""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_extract import EXTRACTOR_KERNEL_SOURCES, extractor_kernel_sha
out_dir, B, calls = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out_dir}/{c}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"]).replace("toad::", "")
        name = re.sub(r"\(.*", "", name)
        if name.startswith(("at::", "void at::")) or "elementwise" in name or "distribution" in name:
            continue                                         # torch.randn of the input tiles, not part of the call
        agg[name] += float(r["Counter_Value"])
    res[c] = {k: v / calls for k, v in agg.items()}          # KB per call
kernels = sorted(set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"]))
per = {k: {"fetch_kb_raw": round(res["FETCH_SIZE"].get(k, 0.0), 1), "write_kb": round(res["WRITE_SIZE"].get(k, 0.0), 1)} for k in kernels}
fetch = sum(v["fetch_kb_raw"] for v in per.values()) * 1024
write = sum(v["write_kb"] for v in per.values()) * 1024
# algorithmic bytes of the call: every activation tensor written once and read once by each consumer (fp32 NHWC), input tiles read once
def algorithmic(B):
    t = B * 3 * 256 * 256 * 4                                # input tiles (read by the space-to-depth gather)
    # (256-wide tiles: the stem reads the NCHW tiles themselves - no space-to-depth image)
    t += B * 64 * 64 * 64 * 4                                # pooled stem output written (the pool rides in the stem's epilogue: the stem's own output never exists)
    inpl, hh = 64, 64
    for pl, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2)):
        for b in range(blocks):
            s = stride if b == 0 else 1
            ho = hh // s
            x_in = B * hh * hh * inpl * 4
            t += x_in + B * hh * hh * pl * 4                  # conv1: read x, write [hh, hh, pl]
            t += B * hh * hh * pl * 4 + B * ho * ho * pl * 4  # conv2: read, write
            if b == 0:
                t += x_in // (s * s) + B * ho * ho * 4 * pl * 4                # downsample: strided read, write
            t += B * ho * ho * pl * 4 + 2 * B * ho * ho * 4 * pl * 4          # conv3: read conv2 out + residual, write block output
            inpl, hh = 4 * pl, ho
    t += B * 16 * 16 * 1024 * 4 + B * 1024 * 4               # average pool
    return t
print(json.dumps({"tiles_per_call": B, "calls_in_trace": calls, "kernel_source_sha256": extractor_kernel_sha(), "kernel_sources": list(EXTRACTOR_KERNEL_SOURCES),
                  "hbm_bytes_per_call": {"fetch_raw": fetch, "fetch_x2": 2 * fetch, "write": write, "total_with_fetch_x2": 2 * fetch + write},
                  "algorithmic_bytes_per_call": algorithmic(B),
                  "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) on tools/extractor_bench.py; KB units; "
                            "fetch_x2 applies the gfx950 wide-load correction to every kernel (all of them read 16 bytes per lane); the weights (< 40 MB per "
                            "call) are L2-resident and not in the algorithmic figure",
                  "per_kernel_kb_per_call": per}, indent=1))

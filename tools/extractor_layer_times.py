"""Per-convolution GEMM time of one extractor forward from a rocprofv3 kernel trace (B tiles of 256x256).
usage: python tools/extractor_layer_times.py <kernel_trace.csv> [B]"""
import csv, sys
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
g = [((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name']) for r in rows
     if 'gemm_nt_h2' in r['Kernel_Name'] or 'gemm_nt_split' in r['Kernel_Name'] or 'conv3x3_h2' in r['Kernel_Name'] or 'stem_halo_pool' in r['Kernel_Name']]
last = g[-43:]
order = [("stem", B * 128 * 128, 147, 64)]   # algorithmic K (the kernel runs the 192-column space-to-depth operand)
inpl, hh = 64, 64
for li, (pl, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2)), 1):
    for b in range(blocks):
        s = stride if b == 0 else 1
        ho = (hh + 2 - 3) // s + 1
        order.append((f"l{li}.{b}.c1", B * hh * hh, inpl, pl))
        order.append((f"l{li}.{b}.c2", B * ho * ho, 9 * pl, pl))
        if b == 0:
            order.append((f"l{li}.{b}.down", B * ho * ho, inpl, 4 * pl))
        order.append((f"l{li}.{b}.c3", B * ho * ho, pl, 4 * pl))
        inpl, hh = 4 * pl, ho
agg, tot, totf = {}, 0.0, 0.0
for (name, M, K, N), (us, kn) in zip(order, last):
    fl = 2 * M * K * N
    tot += us; totf += fl
    kind = ("stem from NCHW + pool" if "stem_halo_pool" in kn else "halo" + kn.split("halo_kernel")[1].split("(")[0] if "halo" in kn else "stream" + kn.split("stream_kernel")[1].split("(")[0] if "stream" in kn else
            "narrow" + kn.split("narrow_kernel")[1].split("(")[0] if "narrow" in kn else "big" + ("+res" if "false, true" in kn else ""))
    a = agg.setdefault((M, K, N, kind), [0, 0.0, 0.0]); a[0] += 1; a[1] += us; a[2] += fl
print(f"{'M':>8} {'K':>5} {'N':>5} {'kernel':<20} {'n':>2} {'us each':>8} {'TF-eq':>6} {'total us':>9} {'HBM floor us (each)':>10}")
for (M, K, N, kind), (c, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    floor = (M * min(K, 1 << 30) + M * N) * 4 / 8e12 * 1e6
    print(f"{M:8d} {K:5d} {N:5d} {kind:<20} {c:2d} {us / c:8.1f} {fl / us / 1e6:6.1f} {us:9.1f} {floor:10.1f}")
print(f"sum {tot:.0f} us   {totf / tot / 1e6:.1f} TF-eq over the 43 GEMM launches")

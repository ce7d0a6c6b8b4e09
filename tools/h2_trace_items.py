"""Item-level timing inside gemm_nt_h2_big_kernel (library built with -DTOAD_H2_TRACE=3): for workgroups 0 and 9, waves 0 and 4, the shader-clock
stamps around each item's epilogue of the 1024 -> 512 forward on a 100k-patch bag: main-loop end, epilogue start, epilogue end, re-sync barrier."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toad_amd import ops, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
NO = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, K, device=dev, generator=g); w = torch.randn(NO, K, device=dev, generator=g) * 0.03; b = torch.zeros(NO, device=dev)
lib = _lib.load()
ws = ops._ws(lib.toad_linear_ws_bytes(N, NO, K), dev)
off = 256 * 256 * 256 * 4 + 65536 * 4
for _ in range(3):
    ws[off: off + 8192].zero_()
    ops.linear_act_fwd(x, w, b, 1)
torch.cuda.synchronize()
tr = ws[off: off + 2 * 8 * 2 * 9 * 8].view(torch.int64).cpu().view(2, 8, 2, 9)
for blk in (0, 1):
    for wv in (0, 1):
        print(f"workgroup {blk * 9} wave {wv * 4}:")
        prev = None
        for it in range(8):
            r = tr[blk, it, wv]
            if int(r[1]) == 0:
                break
            t0 = int(r[0])
            start = t0 if prev is None else prev
            print(f"   item {it}: ends at step {int(r[5])}: main loop {int(r[1]) - start} cyc | wait {int(r[2] - r[1])} | epilogue {int(r[3] - r[2])} | resync {int(r[4] - r[3])}   (t = {int(r[4]) - t0})")
            prev = int(r[4])

wg4 = ws[off + 2 * 8 * 2 * 9 * 8: off + 2 * 8 * 2 * 9 * 8 + 256 * 4 * 8].view(torch.int64).cpu().view(256, 4)
wg = wg4[:, :2]
dur = (wg[:, 1] - wg[:, 0]).double()
rt = wg4[:, 2:].double() / 100.0                       # s_memrealtime: 100 MHz -> us
print("workgroup durations (cycles): min %.0f  median %.0f  max %.0f" % (dur.min(), dur.median(), dur.max()))
print("workgroup durations (us, s_memrealtime): min %.1f median %.1f max %.1f  -> shader clock %.0f MHz" % ((rt[:, 1] - rt[:, 0]).min(), (rt[:, 1] - rt[:, 0]).median(), (rt[:, 1] - rt[:, 0]).max(), float((dur / (rt[:, 1] - rt[:, 0])).median())))
print("first workgroup start -> last workgroup end: %.1f us; start skew (last start - first start) %.1f us; end skew %.1f us" % (rt[:, 1].max() - rt[:, 0].min(), rt[:, 0].max() - rt[:, 0].min(), rt[:, 1].max() - rt[:, 1].min()))
for xcd in range(8):
    d = dur[xcd::8]
    print(f"   XCD {xcd}: cycles min {d.min():.0f} max {d.max():.0f} | start {rt[xcd::8, 0].min() - rt[:, 0].min():.1f}..{rt[xcd::8, 0].max() - rt[:, 0].min():.1f} us | end {rt[xcd::8, 1].min() - rt[:, 0].min():.1f}..{rt[xcd::8, 1].max() - rt[:, 0].min():.1f} us")
srt = torch.sort(dur).values
print("sorted durations, every 16th:", [int(v) for v in srt[::16].tolist()], "last 8:", [int(v) for v in srt[-8:].tolist()])
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); ops.linear_act_fwd(x, w, b, 1); e1.record(); torch.cuda.synchronize()
print("event time of the whole call (absmax + split + GEMM + fix-up): %.1f us" % (e0.elapsed_time(e1) * 1e3))

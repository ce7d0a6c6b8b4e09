"""A/B of the 8-wave NT kernel against the duo kernel on the extractor's WIDE GEMM shapes (ext_linear's "big" branch; per 512 tiles).
    python tools/ab/duo_ext_bench.py [tiles] [reps]
Same operands, tensor-wide scalar scales, optional residual; checks bitwise equality, then times interleaved with HIP events."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from toad_amd import ops, _lib  # noqa: E402

TILES = int(sys.argv[1]) if len(sys.argv) > 1 else 512
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
duo = ctypes.CDLL(os.path.join(HERE, "libtoad_duo.so"))
P, I64, SZ = ctypes.c_void_p, ctypes.c_int64, ctypes.c_size_t
duo.toad_exp_ext_f32.restype = ctypes.c_int
duo.toad_exp_ext_f32.argtypes = [ctypes.c_int, P, P, P, P, P, P, P, I64, I64, I64, ctypes.c_int, P, SZ, P]
duo.toad_last_error.restype = ctypes.c_char_p
lib = _lib.load()


def p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def run(which, x, gx, w, b, res, y, gy, ws):
    m, k = x.shape
    n = w.shape[0]
    gy.zero_()
    rc = duo.toad_exp_ext_f32(which, p(x), p(gx), p(w), p(b), p(res), p(y), p(gy), m, k, n, 1, p(ws), ws.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, duo.toad_last_error()


g = torch.Generator(device=dev).manual_seed(7)
tot = [0.0, 0.0]
# (pixels per tile, K, N, residual, calls per extractor pass)
for px, k, n, has_res, calls in ((1024, 128, 512, True, 4), (256, 256, 1024, True, 6), (4096, 64, 256, False, 1), (1024, 256, 512, False, 1), (1024, 512, 256, False, 1),
                                 (256, 512, 1024, False, 1), (256, 1024, 256, False, 5)):
    m = TILES * px
    x = torch.randn(m, k, device=dev, generator=g).relu_()
    w = torch.randn(n, k, device=dev, generator=g) * (2.0 / (n + k)) ** 0.5
    b = torch.randn(n, device=dev, generator=g) * 0.05
    res = torch.randn(m, n, device=dev, generator=g).relu_() if has_res else None
    gx = x.abs().max().reshape(1)
    ws = torch.empty(lib.toad_linear_ws_bytes(m, n, k), dtype=torch.uint8, device=dev)
    y0, y1 = torch.empty(m, n, device=dev), torch.empty(m, n, device=dev)
    g0, g1 = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    stag = [int(v) for v in os.environ.get("STAGGER", "1").split(",")]        # arm 1: 1 = the duo kernel, >= 2: the shipped kernel staggered over that many 10 ns ticks
    run(0, x, gx, w, b, res, y0, g0, ws); run(stag[0], x, gx, w, b, res, y1, g1, ws)
    torch.cuda.synchronize()
    same = torch.equal(y0, y1) and torch.equal(g0, g1)
    arms = [0] + stag
    t = [[] for _ in arms]
    for _ in range(REPS):
        for i, which in enumerate(arms):
            y, gy = (y0, g0) if i == 0 else (y1, g1)
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(which, x, gx, w, b, res, y, gy, ws); e.record()
            t[i].append((a, e))
    torch.cuda.synchronize()
    med = [sorted(a.elapsed_time(e) * 1e3 for a, e in tt)[len(tt) // 2] for tt in t]
    if len(arms) > 2:
        print("   arms " + "  ".join(f"{w}: {m:.1f}" for w, m in zip(arms, med)))
    tot[0] += med[0] * calls; tot[1] += med[1] * calls
    byt = 4.0 * m * (k + n * (2 if has_res else 1))
    print(f"M={m:8d} K={k:5d} N={n:5d} {'res' if has_res else '   '} x{calls}: bitwise {'EQUAL' if same else 'DIFFERENT %.3e' % (y0 - y1).abs().max().item()} | 8-wave {med[0]:7.1f} us "
          f"({byt / med[0] * 1e-6:5.2f} TB/s, {2.0 * m * n * k / med[0] * 1e-6:6.1f} TF-eq) | duo {med[1]:7.1f} us ({byt / med[1] * 1e-6:5.2f} TB/s) | ratio {med[1] / med[0]:.3f}", flush=True)
    del x, res, y0, y1, ws
print(f"per extractor pass of {TILES} tiles (incl. the weight split + gmax memset): 8-wave {tot[0]:.0f} us, duo {tot[1]:.0f} us")

"""A/B of the 8-wave NT kernel (libtoad_hip.so) against the duo experiment (tools/ab/libtoad_duo.so) on the MIL step's NT shapes.
    python tools/ab/duo_bench.py [rows] [reps]
Checks that the two give bitwise the same outputs (same arithmetic, same MFMA order), then times them interleaved with HIP events."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from toad_amd import ops, _lib  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
duo = ctypes.CDLL(os.path.join(HERE, "libtoad_duo.so"))
P, I64, SZ = ctypes.c_void_p, ctypes.c_int64, ctypes.c_size_t
duo.toad_exp_nt_duo_f32.restype = ctypes.c_int
duo.toad_exp_nt_duo_f32.argtypes = [P, P, I64, I64, P, P, I64, I64, I64, ctypes.c_int, P, P, P, P, P, P, P, P, ctypes.c_int, P, SZ, P, P]
duo.toad_last_error.restype = ctypes.c_char_p
lib = _lib.load()


def p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


TRACE = None


def run_duo(x, w, b, act, x_amax, want_amax, want_bits, relu_src=None, bits_in=None, pool=None):
    m, k = x.shape
    n = w.shape[0]
    y = torch.empty((m, n), device=dev)
    y_amax = torch.empty((ops.amax_floats(m),), device=dev) if want_amax else None
    bits = torch.empty((ops.relu_bits_bytes(m, n),), dtype=torch.uint8, device=dev) if want_bits else None
    ws = ops._ws(lib.toad_linear_ws_bytes(m, n, k), dev, "duo")
    pa = ps = pd = None; pt = 0
    if pool is not None:
        pa, ps, pd = pool; pt = pa.shape[1]
    rc = duo.toad_exp_nt_duo_f32(p(x), p(w), k, 1, p(b), p(y), m, k, n, act, p(x_amax), p(y_amax), p(bits), p(relu_src), p(bits_in), p(pa), p(ps), p(pd), pt,
                                 p(ws), ws.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), p(TRACE))
    assert rc == 0, duo.toad_last_error()
    return y, y_amax, bits


def timeit(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record()
    return a, b


g = torch.Generator(device=dev).manual_seed(5)
rows = []
for name, k, n, kind in (("fwd2  512->512 relu+bits", 512, 512, "fwd_bits"), ("fwd_ab 512->768", 512, 768, "fwd"), ("fwd1 1024->512 relu+bits", 1024, 512, "fwd_bits"),
                         ("dgrad2 512->512 bitmask", 512, 512, "dgrad"), ("dgrad_ab 768->512 pool+bitmask", 768, 512, "dgrad_pool")):
    x = torch.randn(M, k, device=dev, generator=g)
    w = torch.randn(n, k, device=dev, generator=g) * (2.0 / (n + k)) ** 0.5
    b = torch.randn(n, device=dev, generator=g) * 0.05
    xa = ops.absmax_rows256(x)
    if kind == "fwd_bits":
        ref = lambda: ops.linear_act_fwd(x, w, b, 1, x_amax=xa, want_bits=True)
        new = lambda: run_duo(x, w, b, 1, xa, True, True)
    elif kind == "fwd":
        ref = lambda: (ops.linear_act_fwd(x, w, b, 0, x_amax=xa), None, None)
        new = lambda: run_duo(x, w, b, 0, xa, False, False)
    else:
        src = torch.randn(M, n, device=dev, generator=g).relu_()                       # the saved activation whose ReLU mask the dgrad applies
        _, _, bits_in = ops.linear_act_fwd(torch.randn(M, 32, device=dev, generator=g), torch.randn(n, 32, device=dev, generator=g), None, 1, want_bits=True)
        src = None                                                                        # (the mask of THAT product; relu_src must match it for remainder tiles)
        h = ops.linear_act_fwd(torch.randn(M, 32, device=dev, generator=g), torch.randn(n, 32, device=dev, generator=g), None, 1, want_bits=True)
        src, bits_in = h[0], h[2]
        pool = None
        if kind == "dgrad_pool":
            pool = (torch.randn(M, 2, device=dev, generator=g), torch.tensor([[3.0, 1e4], [2.5, 2e4]], device=dev), torch.randn(2, n, device=dev, generator=g) * 1e-3)
        # per-op dgrad API: dX[M,n] = dY[M,k] . WT[n,k]^T with WT = w (already [n_out, k_red] row-major)
        ref = lambda: ops.linear_dgrad(x, w, relu_src=src, pool=pool, dy_amax=xa, want_amax=True, relu_bits=bits_in) + (None,)
        new = lambda: run_duo(x, w, None, 0, xa, True, False, relu_src=src, bits_in=bits_in, pool=pool)
    r, d = ref(), new()
    torch.cuda.synchronize()
    same = torch.equal(r[0], d[0]) and (r[1] is None or torch.equal(r[1], d[1])) and (r[2] is None or d[2] is None or torch.equal(r[2], d[2]))
    maxdiff = (r[0] - d[0]).abs().max().item()
    for _ in range(3):
        ref(); new()
    evs_r, evs_d = [], []
    for _ in range(REPS):
        evs_r.append(timeit(ref)); evs_d.append(timeit(new))
    torch.cuda.synchronize()
    tr = sorted(a.elapsed_time(b) * 1e3 for a, b in evs_r); td = sorted(a.elapsed_time(b) * 1e3 for a, b in evs_d)
    flop = 2.0 * M * n * k
    rows.append((name, same, maxdiff, tr[len(tr) // 2], td[len(td) // 2]))
    print(f"{name:34s} M={M}: bitwise {'EQUAL' if same else 'DIFFERENT (max abs diff %.3e)' % maxdiff} | 8-wave {tr[len(tr)//2]:7.1f} us ({flop/tr[len(tr)//2]*1e-6:6.1f} TF-eq)"
          f" | duo {td[len(td)//2]:7.1f} us ({flop/td[len(td)//2]*1e-6:6.1f} TF-eq) | ratio {td[len(td)//2]/tr[len(tr)//2]:.3f}", flush=True)
    if os.environ.get("DUO_TRACE"):
        TRACE = torch.zeros(512 * 4 * 10, dtype=torch.int64, device=dev)
        new(); torch.cuda.synchronize()
        t = TRACE.reshape(512, 4, 10).double()
        ph = t[:, :, :8].reshape(-1, 8)
        names = ["load a (12 frag reads issued, octet 0 split+written, 2 loads)", "k16#0: wait frags + 24 MFMAs", "load b (12 reads, octet 1 split+written)", "barrier 1",
                 "DMA issue + 2 loads + 24 MFMAs", "vmcnt(2) + barrier 2", "epilogue + zero", "loop top"]
        steps = (M // 128) * (n // 256 if n % 256 == 0 else n // 256 + 1) * (k // 32) / 512
        print("   per step (cycles, median over 2048 waves; %d steps per workgroup): " % steps + "; ".join(f"{nm}: {ph[:, i].median().item() / steps:.0f}" for i, nm in enumerate(names)))
        dur = (t[:, :, 9] - t[:, :, 8]).reshape(-1)
        print(f"   wave lifetime cycles: min {dur.min().item():.0f} median {dur.median().item():.0f} max {dur.max().item():.0f}; starts spread {(t[:, :, 8].max() - t[:, :, 8].min()).item():.0f}")
        w0 = t[:, 0, :]                                      # wave 0 of every workgroup; stamps 8 / 9 = s_memrealtime (100 MHz, chip-wide) at start / end
        t0_ = w0[:, 8].min()
        st = (w0[:, 8] - t0_) * 10e-3; en = (w0[:, 9] - t0_) * 10e-3          # us
        mid = 0.5 * (st.min() + en.max())
        print(f"   workgroup start us: min {st.min():.1f} median {st.median():.1f} max {st.max():.1f}; end us: min {en.min():.1f} median {en.median():.1f} max {en.max():.1f}; "
              f"alive at {mid:.1f} us: {int(((st <= mid) & (en >= mid)).sum())} of 512; started within 5 us: {int((st < 5).sum())}")
        TRACE = None
print("sum 8-wave %.1f us, duo %.1f us" % (sum(r[3] for r in rows), sum(r[4] for r in rows)))

"""Thin tensor-level wrappers over the C ABI (pointers + sizes + the current HIP stream).

PyTorch is used for device memory and streams only; all arithmetic happens in libtoad_hip.so.
Shape/dtype/device/contiguity are validated here (the checks PyTorch's own ops would do for the
reference), the C side validates again and reports through toad_last_error().
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

ACT_NONE, ACT_RELU = 0, 1

# ---- optional per-op HIP-event timing (used by bench.py for the roofline figures) -------------
# Events are recorded on the stream the kernels are launched on (the current torch stream), so the
# elapsed time is device time of exactly the launches made by that C-ABI call.
_TIMING = None
_TIMING_LEVEL = 2          # 1: only the fused pool forward is bracketed (2 events per slide); 2: pool + the eight GEMM calls (18)
_EVENT_POOL = []           # events created AND recorded once ahead of time, so a timed region pays no event creation
_TIMING_STRIDE = 1         # whole-slide calls: bracket every k-th call only (an event packet costs ~5 us of stream time: 2 % of a 10k-patch step)
_TIMING_CALLS = 0


def enable_timing(on: bool = True, level: int = 2, prealloc: int = 0, stride: int = 1) -> None:
    """Switch per-op HIP-event timing on / off. Event packets are not free (18 per slide cost ~0.1 ms of stream time), so a
    throughput measurement uses level 1 (two events around the dominant kernel) and a separate instrumented loop level 2.
    `prealloc` events are created and materialised now, outside any timed region. `stride`: the whole-slide calls (mil_step,
    mil_multi_step) record their events on every stride-th call only (the first one included)."""
    global _TIMING, _TIMING_LEVEL, _TIMING_STRIDE, _TIMING_CALLS
    _TIMING = {} if on else None
    _TIMING_LEVEL = level
    _TIMING_STRIDE = max(1, int(stride))
    _TIMING_CALLS = 0
    if on and prealloc > 0 and torch.cuda.is_available():
        for _ in range(prealloc):
            e = torch.cuda.Event(enable_timing=True)
            e.record()                     # materialises the underlying hipEvent_t
            _EVENT_POOL.append(e)
        torch.cuda.synchronize()


def _timing_due() -> bool:
    global _TIMING_CALLS
    _TIMING_CALLS += 1
    return (_TIMING_CALLS - 1) % _TIMING_STRIDE == 0


def timing_call_count() -> int:
    """Whole-slide calls made since enable_timing (bracketed or not)."""
    return _TIMING_CALLS


def _take_event():
    if _EVENT_POOL:
        return _EVENT_POOL.pop()
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def collect_timing():
    """-> {op name: (calls, total milliseconds)}; synchronises the device."""
    if _TIMING is None:
        return {}
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in _TIMING.items()}


class _timed:
    __slots__ = ("name", "e0")

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _TIMING is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if _TIMING is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _TIMING.setdefault(self.name, []).append((self.e0, e1))


def _chk(t: Optional[torch.Tensor], name: str, dtype=torch.float32, allow_none: bool = False):
    if t is None:
        if allow_none:
            return
        raise ValueError(f"{name} is None")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA(HIP) tensor: toad_amd has no CPU path (got {t.device})")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Per-op workspaces are scratch: one growing buffer per (device, stream) instead of one allocation per call. Work on a
# stream is ordered, so consecutive ops of that stream can share it; different streams (two models training side by side)
# get different buffers.
_WS_CACHE = {}


def _ws(nbytes: int, device, slot: str = "op") -> torch.Tensor:
    nbytes = max(int(nbytes), 16)
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream(), slot)
    buf = _WS_CACHE.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes + (nbytes >> 3), dtype=torch.uint8, device=device)      # 12 % headroom: bags vary in length
        _WS_CACHE[key] = buf
    return buf


def release_workspaces() -> None:
    """Drop every cached workspace (they are re-created on demand)."""
    _WS_CACHE.clear()


def amax_floats(rows: int) -> int:
    return int(_lib.load().toad_amax_floats(int(rows)))


def absmax_rows256(x: torch.Tensor) -> torch.Tensor:
    """abs-max array of x [M,K]: one float per block of 256 rows (the operand scales of the fp16 two-piece GEMMs)."""
    _chk(x, "x")
    m, k = x.shape
    out = torch.empty((amax_floats(m),), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().toad_absmax_rows256_f32(_p(x), m, k, _p(out), _stream()), "toad_absmax_rows256_f32")
    return out


def h2_ok(m: int, n: int, k: int) -> bool:
    """True when the persistent fp16 two-piece kernel serves an [m,k] x [n,k]^T product."""
    return bool(_lib.load().toad_linear_h2_ok(int(m), int(n), int(k)))


def x16_ok(n: int) -> bool:
    """True when a [n, 1024] fp16 bag can go through the whole-slide fp16 entry points (toad_mil_*_x16_f32)."""
    return bool(_lib.load().toad_mil_x16_ok(int(n)))


def relu_bits_bytes(m: int, n: int) -> int:
    return int(_lib.load().toad_relu_bits_bytes(int(m), int(n)))


# ---- ONE arithmetic (round 6) -------------------------------------------------------------------------------------------------------------
# The fp16 two-piece GEMMs take reductions in whole 32-deep stages, 16-byte output rows and operands they can address with 32-bit row offsets
# (toad_linear_h2_ok). Every shape TOAD itself builds qualifies; a standalone Attn_Net_Gated of another (L, D) (models/model_toad.py:19 takes any)
# used to fall through to the exact-fp32 MFMA kernels inside the library (a 5x lower ceiling and another rounding). The per-op wrappers below
# ZERO-PAD such operands instead - padded products are exactly zero, padded output columns are sliced off - and cut operands of 2^32 bytes and
# more into row chunks, so every public Python entry point runs on the two-piece kernels (tests/test_gpu_h2.py counts the library's fallback
# launches: toad_fallback_launches). The exact-fp32 kernels remain in the library only for raw C-ABI callers that pass such shapes unpadded.
_K_STAGE = 32
_CHUNK_ROWS = 4092 * 256                        # csrc/step.hip kChunkRows: rows of one NT launch on a 1024-wide fp32 operand (32-bit byte offsets)
_CHUNK_SEED_STEP = 0xD1B54A32D192ED03          # csrc/step.hip kChunkSeedStep: train-mode dropout stream of row chunk j = seed + j * this
_M64 = 0xFFFFFFFFFFFFFFFF


def _pad_cols(t: Optional[torch.Tensor], to: int) -> Optional[torch.Tensor]:
    """t [..., c] -> [..., to] with zero columns appended (a copy; only shapes outside the kernels' envelope pay it)."""
    if t is None or t.shape[-1] == to:
        return t
    return torch.nn.functional.pad(t, (0, to - t.shape[-1])).contiguous()


def _up(v: int, q: int) -> int:
    return (v + q - 1) // q * q


def _row_chunks(m: int, row_bytes: int):
    """Row ranges of one NT launch each: whole operand when its bytes fit 32-bit offsets, else multiples of 256 rows."""
    if m * row_bytes < (1 << 32):
        return [(0, m)]
    step = max(256, min(_CHUNK_ROWS, ((1 << 32) - 1) // row_bytes // 256 * 256))
    return [(r0, min(m, r0 + step)) for r0 in range(0, m, step)]


def linear_act_fwd(x, w, b, act: int, out: Optional[torch.Tensor] = None, drop_p: float = 0.0, drop_seed: int = 0,
                   x_amax: Optional[torch.Tensor] = None, want_amax: bool = False, want_bits: bool = False):
    """Y = dropout_p(act(X W^T + b)); X [M,K], W [N,K], b [N] or None. drop_p = 0 disables dropout.
    x_amax: abs-max array of X (else measured inside). want_amax: also return the abs-max array of Y -> (Y, y_amax).
    want_bits (act = RELU, h2_ok shapes): also return the one-bit image of Y for the dgrad of this layer -> (Y, y_amax, bits).
    Shapes outside the two-piece kernels' envelope are zero-padded (K to a multiple of 32, N to a multiple of 4) and operands of >= 2^32 bytes
    run as row chunks (train-mode dropout then draws chunk j's masks from drop_seed + j * kChunkSeedStep, like the whole-slide calls)."""
    _chk(x, "x"); _chk(w, "w"); _chk(b, "b", allow_none=True); _chk(x_amax, "x_amax", allow_none=True)
    m, k = x.shape
    n, k2 = w.shape
    if k != k2 or (b is not None and b.numel() != n):
        raise ValueError(f"linear_act_fwd: shape mismatch x{tuple(x.shape)} w{tuple(w.shape)}")
    kp, np_ = _up(k, _K_STAGE), _up(n, 4)
    chunks = _row_chunks(m, kp * 4)
    if kp != k or np_ != n or len(chunks) > 1:
        if drop_p > 0 and np_ != n:
            raise NotImplementedError("linear_act_fwd: dropout with an output width that is not a multiple of 4")
        xp, wp = _pad_cols(x, kp), _pad_cols(w, kp)
        if np_ != n:
            wp = torch.nn.functional.pad(wp, (0, 0, 0, np_ - n)).contiguous()
            bp = None if b is None else torch.nn.functional.pad(b, (0, np_ - n)).contiguous()
        else:
            bp = b
        yp = out if (out is not None and np_ == n) else torch.empty((m, np_), dtype=torch.float32, device=x.device)
        amaxs, bitss = [], []
        for j, (r0, r1) in enumerate(chunks):
            xa = None if x_amax is None else x_amax[r0 // 256:(r1 + 255) // 256].contiguous()
            res = linear_act_fwd(xp[r0:r1], wp, bp, act, out=yp[r0:r1], drop_p=drop_p, drop_seed=(drop_seed + j * _CHUNK_SEED_STEP) & _M64,
                                 x_amax=xa, want_amax=want_amax or want_bits, want_bits=want_bits and np_ == n)
            if want_amax or want_bits:
                amaxs.append(res[1])
                if want_bits:
                    bitss.append(res[2] if np_ == n else None)
        y = yp if np_ == n else yp[:, :n].contiguous()
        if out is not None and y is not out:
            out.copy_(y); y = out
        if want_bits:
            return y, torch.cat(amaxs), (torch.cat(bitss) if all(t is not None for t in bitss) else None)
        return (y, torch.cat(amaxs)) if want_amax else y
    y = out if out is not None else torch.empty((m, n), dtype=torch.float32, device=x.device)
    _chk(y, "out")
    lib = _lib.load()
    y_amax = torch.empty((amax_floats(m),), dtype=torch.float32, device=x.device) if (want_amax or want_bits) else None
    bits = torch.empty((relu_bits_bytes(m, n),), dtype=torch.uint8, device=x.device) if (want_bits and h2_ok(m, n, k)) else None
    ws = _ws(lib.toad_linear_ws_bytes(m, n, k), x.device)
    with _timed("gemm_fwd"):
        _lib.check(lib.toad_linear_act_fwd_f32(_p(x), _p(w), _p(b), _p(y), m, k, n, act, float(drop_p), int(drop_seed),
                                               _p(x_amax), _p(y_amax), _p(bits), _p(ws), ws.numel(), _stream()), "toad_linear_act_fwd_f32")
    if want_bits:
        return y, y_amax, bits
    return (y, y_amax) if want_amax else y


def transpose(w: torch.Tensor) -> torch.Tensor:
    _chk(w, "w")
    r, c = w.shape
    out = torch.empty((c, r), dtype=torch.float32, device=w.device)
    _lib.check(_lib.load().toad_transpose_f32(_p(w), _p(out), r, c, _stream()), "toad_transpose_f32")
    return out


def linear_dgrad(dy, wt, addend=None, relu_src=None, out: Optional[torch.Tensor] = None, mask_scale: float = 1.0,
                 pool=None, dy_amax: Optional[torch.Tensor] = None, want_amax: bool = False, relu_bits: Optional[torch.Tensor] = None):
    """dX = (dY W + addend + pool) * (relu_src > 0) * mask_scale; dY [M,N], wt = W^T [K,N].
    pool = (A_raw [M,T], stats [T,2], dM [T,K]): the attention-pooling gradient sum_t softmax(A_raw)[:,t] dM[t,:], recomputed in
    the epilogue instead of read from an [M,K] buffer (needs h2_ok(M, K, N)). relu_bits: the one-bit image of relu_src
    (linear_act_fwd(..., want_bits=True) of the same layer)."""
    _chk(dy, "dy"); _chk(wt, "wt"); _chk(addend, "addend", allow_none=True); _chk(relu_src, "relu_src", allow_none=True)
    _chk(relu_bits, "relu_bits", dtype=torch.uint8, allow_none=True)
    _chk(dy_amax, "dy_amax", allow_none=True)
    m, n = dy.shape
    k, n2 = wt.shape
    if n != n2:
        raise ValueError("linear_dgrad: shape mismatch")
    np_, kp = _up(n, _K_STAGE), _up(k, 4)
    chunks = _row_chunks(m, np_ * 4)
    if np_ != n or kp != k or len(chunks) > 1:               # outside the two-piece kernels' envelope: zero padding / row chunks (see linear_act_fwd)
        dyp, wtp = _pad_cols(dy, np_), _pad_cols(wt, np_)
        if kp != k:
            wtp = torch.nn.functional.pad(wtp, (0, 0, 0, kp - k)).contiguous()
            if relu_bits is not None:
                raise NotImplementedError("linear_dgrad: a bit image with an output width that is not a multiple of 4")
        addp, srcp = _pad_cols(addend, kp), _pad_cols(relu_src, kp)
        poolp = None if pool is None else (pool[0], pool[1], _pad_cols(pool[2], kp))
        dxp = out if (out is not None and kp == k) else torch.empty((m, kp), dtype=torch.float32, device=dy.device)
        amaxs = []
        for (r0, r1) in chunks:
            b0, b1 = r0 // 256, (r1 + 255) // 256
            bits_c = None
            if relu_bits is not None:
                per_blk = relu_bits.numel() // max(amax_floats(m), 1)
                bits_c = relu_bits[b0 * per_blk:b1 * per_blk]
            res = linear_dgrad(dyp[r0:r1], wtp, None if addp is None else addp[r0:r1], None if srcp is None else srcp[r0:r1], out=dxp[r0:r1],
                               mask_scale=mask_scale, pool=None if poolp is None else (poolp[0][r0:r1], poolp[1], poolp[2]),
                               dy_amax=None if dy_amax is None else dy_amax[b0:b1].contiguous(), want_amax=want_amax, relu_bits=bits_c)
            if want_amax:
                amaxs.append(res[1])
        dxr = dxp if kp == k else dxp[:, :k].contiguous()
        if out is not None and dxr is not out:
            out.copy_(dxr); dxr = out
        return (dxr, torch.cat(amaxs)) if want_amax else dxr
    dx = out if out is not None else torch.empty((m, k), dtype=torch.float32, device=dy.device)
    _chk(dx, "out")
    for t, nm in ((addend, "addend"), (relu_src, "relu_src"), (dx, "out")):
        if t is not None and tuple(t.shape) != (m, k):
            raise ValueError(f"linear_dgrad: {nm} must be [{m},{k}]")
    pa = ps = pd = None
    pt = 0
    if pool is not None:
        pa, ps, pd = pool
        _chk(pa, "pool A_raw"); _chk(ps, "pool stats"); _chk(pd, "pool dM")
        pt = pa.shape[1]
        if pa.shape[0] != m or tuple(ps.shape) != (pt, 2) or tuple(pd.shape) != (pt, k):
            raise ValueError("linear_dgrad: pooling-addend shapes must be A_raw [M,T], stats [T,2], dM [T,K]")
    lib = _lib.load()
    dx_amax = torch.empty((amax_floats(m),), dtype=torch.float32, device=dy.device) if want_amax else None
    ws = _ws(lib.toad_linear_ws_bytes(m, k, n), dy.device)
    with _timed("gemm_dgrad"):
        _lib.check(lib.toad_linear_dgrad_f32(_p(dy), _p(wt), _p(addend), _p(relu_src), float(mask_scale), _p(dx), m, n, k,
                                             _p(pa), _p(ps), _p(pd), pt, _p(dy_amax), _p(dx_amax), _p(relu_bits), _p(ws), ws.numel(),
                                             _stream()), "toad_linear_dgrad_f32")
    return (dx, dx_amax) if want_amax else dx


def linear_wgrad(dy, x, dw: Optional[torch.Tensor] = None, db: Optional[torch.Tensor] = None,
                 beta: float = 0.0, want_db: bool = True, dy_amax: Optional[torch.Tensor] = None,
                 x_amax: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """dW = beta*dW + dY^T X; db = beta*db + colsum(dY). dy_amax / x_amax: abs-max arrays of the operands (else measured).
    ``x`` may be a PreparedBag (plane-tiled input operand: toad_linear_wgrad_xp_f32)."""
    prepared = getattr(x, "is_prepared_bag", False)
    _chk(dy, "dy"); _chk(dy_amax, "dy_amax", allow_none=True)
    if not prepared:
        _chk(x, "x"); _chk(x_amax, "x_amax", allow_none=True)
    m, n = dy.shape
    m2, k = x.shape
    if m != m2:
        raise ValueError("linear_wgrad: shape mismatch")
    if not prepared and (n % 4 or k % 4):                    # 16-byte rows for the TN kernels: zero columns, sliced off the result
        np_, kp = _up(n, 4), _up(k, 4)
        dwp, dbp = linear_wgrad(_pad_cols(dy, np_), _pad_cols(x, kp), None, None, 0.0, want_db or db is not None, dy_amax, x_amax)
        dwn = dwp[:n, :k]
        if dw is None:
            dw = dwn.contiguous()
        else:
            dw.mul_(beta).add_(dwn) if beta != 0.0 else dw.copy_(dwn)
        if dbp is not None:
            if db is None:
                db = dbp[:n].contiguous()
            else:
                db.mul_(beta).add_(dbp[:n]) if beta != 0.0 else db.copy_(dbp[:n])
        return dw, db
    if dw is None:
        dw = torch.empty((n, k), dtype=torch.float32, device=dy.device); beta = 0.0
    if db is None and want_db:
        db = torch.empty((n,), dtype=torch.float32, device=dy.device)
    _chk(dw, "dw"); _chk(db, "db", allow_none=True)
    lib = _lib.load()
    nbytes = lib.toad_linear_wgrad_ws_bytes(m, n, k)
    ws = _ws(nbytes, dy.device, "wgrad")
    with _timed("gemm_wgrad"):
        if prepared:
            _lib.check(lib.toad_linear_wgrad_xp_f32(_p(dy), _p(x.planes), _p(x.amax), _p(dw), _p(db), m, n, k, float(beta), _p(dy_amax),
                                                    _p(ws), ws.numel(), _stream()), "toad_linear_wgrad_xp_f32")
        else:
            _lib.check(lib.toad_linear_wgrad_f32(_p(dy), _p(x), _p(dw), _p(db), m, n, k, float(beta), _p(dy_amax), _p(x_amax),
                                                 _p(ws), ws.numel(), _stream()), "toad_linear_wgrad_f32")
    return dw, db


def dropout_mask(n: int, drop_p: float, drop_seed: int, device) -> torch.Tensor:
    """The multiplier (0 or 1/(1-p)) the kernels apply to flat element e under ``drop_seed`` (never stored by them)."""
    out = torch.empty((n,), dtype=torch.float32, device=device)
    _lib.check(_lib.load().toad_dropout_mask_f32(_p(out), n, float(drop_p), int(drop_seed), _stream()), "toad_dropout_mask_f32")
    return out


def gated_pool_fwd(p: torch.Tensor, d: int, h: Optional[torch.Tensor], wc, bc, drop_p: float = 0.0, seed_a: int = 0, seed_b: int = 0):
    """p = [N, 2D] stacked pre-activations (Pa | Pb). Returns (A_raw [N,T], M [T,L], stats [T,2]);
    with h=None only A_raw is computed (attention_only)."""
    _chk(p, "p"); _chk(h, "h", allow_none=True); _chk(wc, "wc"); _chk(bc, "bc")
    n, ldp = p.shape
    t = wc.shape[0]
    if ldp != 2 * d or wc.shape[1] != d or bc.numel() != t:
        raise ValueError("gated_pool_fwd: shape mismatch")
    lib = _lib.load()
    a_raw = torch.empty((n, t), dtype=torch.float32, device=p.device)
    pb_ptr = p.data_ptr() + 4 * d
    if h is None:
        _lib.check(lib.toad_gated_pool_fwd_f32(_p(p), pb_ptr, ldp, None, _p(wc), _p(bc), _p(a_raw), None, None, None, 0,
                                               n, 512, d, t, float(drop_p), int(seed_a), int(seed_b), _stream()),
                   "toad_gated_pool_fwd_f32")
        return a_raw, None, None
    l = h.shape[1]
    if h.shape[0] != n:
        raise ValueError("gated_pool_fwd: h rows != p rows")
    m = torch.empty((t, l), dtype=torch.float32, device=p.device)
    stats = torch.empty((t, 2), dtype=torch.float32, device=p.device)
    ws = _ws(lib.toad_gated_pool_ws_bytes(n, l, d, t), p.device, "pool")
    with _timed("pool_fwd"):
        _lib.check(lib.toad_gated_pool_fwd_f32(_p(p), pb_ptr, ldp, _p(h), _p(wc), _p(bc), _p(a_raw), _p(m), _p(stats),
                                               _p(ws), ws.numel(), n, l, d, t, float(drop_p), int(seed_a), int(seed_b),
                                               _stream()), "toad_gated_pool_fwd_f32")
    return a_raw, m, stats


def gated_pool_bwd(p, d: int, h, wc, a_raw, stats, m, dm, da_ext=None, dwc=None, dbc=None, beta: float = 0.0,
                   dp: Optional[torch.Tensor] = None, dh: Optional[torch.Tensor] = None,
                   drop_p: float = 0.0, seed_a: int = 0, seed_b: int = 0, want_dh: bool = True, want_amax: bool = False):
    """Returns (dP [N,2D], dH_pool [N,L] | None, dWc [T,D], dbc [T]) (+ the abs-max array of dP with want_amax).
    want_dh=False skips the [N,L] pooling gradient (linear_dgrad(pool=...) recomputes it in its epilogue)."""
    for t_, nm in ((p, "p"), (h, "h"), (wc, "wc"), (a_raw, "a_raw"), (stats, "stats"), (m, "m"), (dm, "dm")):
        _chk(t_, nm)
    _chk(da_ext, "da_ext", allow_none=True)
    n, ldp = p.shape
    l = h.shape[1]
    t = wc.shape[0]
    if ldp != 2 * d or tuple(a_raw.shape) != (n, t) or tuple(dm.shape) != (t, l) or tuple(m.shape) != (t, l):
        raise ValueError("gated_pool_bwd: shape mismatch")
    if dp is None:
        dp = torch.empty_like(p)
    if dh is None and want_dh:
        dh = torch.empty_like(h)
    if dwc is None:
        dwc = torch.empty_like(wc); dbc = torch.empty((t,), dtype=torch.float32, device=p.device); beta = 0.0
    _chk(dp, "dp"); _chk(dh, "dh", allow_none=True); _chk(dwc, "dwc"); _chk(dbc, "dbc")
    lib = _lib.load()
    dp_amax = torch.empty((amax_floats(n),), dtype=torch.float32, device=p.device) if want_amax else None
    ws = _ws(lib.toad_gated_pool_bwd_ws_bytes(n, l, d, t), p.device, "pool")
    with _timed("pool_bwd"):
        _lib.check(lib.toad_gated_pool_bwd_f32(_p(p), p.data_ptr() + 4 * d, ldp, _p(h), _p(wc), _p(a_raw), _p(stats), _p(m),
                                               _p(dm), _p(da_ext), _p(dp), dp.data_ptr() + 4 * d, ldp, _p(dh), _p(dwc),
                                               _p(dbc), float(beta), _p(dp_amax), _p(ws), ws.numel(), n, l, d, t, float(drop_p),
                                               int(seed_a), int(seed_b), _stream()),
                   "toad_gated_pool_bwd_f32")
    if want_amax:
        return dp, dh, dwc, dbc, dp_amax
    return dp, dh, dwc, dbc


# ---- scores only, over a column block of the stacked pre-activations (standalone Attn_Net_Gated of any shape) -------------------------
# The pool kernels' covering instantiation takes D <= 512 (multiple of 4) and n_tasks <= 4. The scores are linear in the gate columns and
# independent per task, so any (D, n_tasks) is a sum over column blocks of <= 512 and a concatenation over task blocks of <= 4
# (toad_amd.model_toad._ScoresFn drives the blocks). ``p`` is the [N, 2*dtot] stacked pre-activation matrix; the block is columns
# [d0, d0 + wc.shape[1]) of its tanh half and of its sigmoid half.
def gate_scores_block_fwd(p: torch.Tensor, dtot: int, d0: int, wc: torch.Tensor, bc: torch.Tensor, drop_p: float = 0.0, seed_a: int = 0, seed_b: int = 0):
    """A_block [N, T_blk] = (tanh(Pa[:, blk]) * sigmoid(Pb[:, blk])) wc^T + bc for one (task block, column block)."""
    _chk(p, "p"); _chk(wc, "wc"); _chk(bc, "bc")
    n, ldp = p.shape
    t, d = wc.shape
    if ldp != 2 * dtot or d0 % 4 or d % 4 or dtot % 4 or d0 + d > dtot or d > 512 or t > 4 or bc.numel() != t:
        raise ValueError("gate_scores_block_fwd: bad block")
    a_raw = torch.empty((n, t), dtype=torch.float32, device=p.device)
    _lib.check(_lib.load().toad_gated_pool_fwd_f32(p.data_ptr() + 4 * d0, p.data_ptr() + 4 * (dtot + d0), ldp, None, _p(wc), _p(bc), _p(a_raw), None, None,
                                                   None, 0, n, 8, d, t, float(drop_p), int(seed_a), int(seed_b), _stream()), "toad_gated_pool_fwd_f32")
    return a_raw


def gate_scores_block_bwd(p: torch.Tensor, dtot: int, d0: int, wc: torch.Tensor, da: torch.Tensor, drop_p: float = 0.0, seed_a: int = 0, seed_b: int = 0):
    """Backward of gate_scores_block_fwd for the external gradient ``da`` [N, T_blk]: returns (dPa_blk [N, d], dPb_blk [N, d], dWc_blk, dbc_blk).
    The pooled backward is reused with softmax weights of zero (stats = (0, inf)) and a zero pooled gradient: dS = dA exactly; its H operand is
    then multiplied by zeros only, so an 8-column dummy stands in for it (no limit on the layer's L)."""
    _chk(p, "p"); _chk(wc, "wc"); _chk(da, "da")
    n, ldp = p.shape
    t, d = wc.shape
    if ldp != 2 * dtot or d0 % 4 or d % 4 or dtot % 4 or d0 + d > dtot or d > 512 or t > 4 or tuple(da.shape) != (n, t):
        raise ValueError("gate_scores_block_bwd: bad block")
    dev = p.device
    lib = _lib.load()
    h = torch.zeros((n, 8), dtype=torch.float32, device=dev)
    zeros_tl = torch.zeros((t, 8), dtype=torch.float32, device=dev)
    a0 = torch.zeros((n, t), dtype=torch.float32, device=dev)
    stats = torch.tensor([[0.0, float("inf")]] * t, dtype=torch.float32, device=dev)
    dp = torch.empty((n, 2 * d), dtype=torch.float32, device=dev)
    dwc = torch.empty_like(wc); dbc = torch.empty((t,), dtype=torch.float32, device=dev)
    ws = _ws(lib.toad_gated_pool_bwd_ws_bytes(n, 8, d, t), dev, "pool")
    _lib.check(lib.toad_gated_pool_bwd_f32(p.data_ptr() + 4 * d0, p.data_ptr() + 4 * (dtot + d0), ldp, _p(h), _p(wc), _p(a0), _p(stats), _p(zeros_tl), _p(zeros_tl),
                                           _p(da), _p(dp), dp.data_ptr() + 4 * d, 2 * d, None, _p(dwc), _p(dbc), 0.0, None, _p(ws), ws.numel(), n, 8, d, t,
                                           float(drop_p), int(seed_a), int(seed_b), _stream()), "toad_gated_pool_bwd_f32")
    return dp[:, :d], dp[:, d:], dwc, dbc


def heads_fwd(m, sex, wcls, bcls, wsite, bsite):
    """Returns (Mcat [2,L+1], logits [1,C], Y_prob, Y_hat [1,1] int64, site_logits [1,2], site_prob, site_hat)."""
    for t_, nm in ((m, "m"), (sex, "sex"), (wcls, "wcls"), (bcls, "bcls"), (wsite, "wsite"), (bsite, "bsite")):
        _chk(t_, nm)
    l = m.shape[1]
    c = wcls.shape[0]
    if m.shape[0] != 2 or wcls.shape[1] != l + 1 or tuple(wsite.shape) != (2, l + 1) or sex.numel() != 1:
        raise ValueError("heads_fwd: shape mismatch")
    dev = m.device
    mcat = torch.empty((2, l + 1), dtype=torch.float32, device=dev)
    logits = torch.empty((1, c), dtype=torch.float32, device=dev)
    y_prob = torch.empty((1, c), dtype=torch.float32, device=dev)
    y_hat = torch.empty((1, 1), dtype=torch.int64, device=dev)
    site_logits = torch.empty((1, 2), dtype=torch.float32, device=dev)
    site_prob = torch.empty((1, 2), dtype=torch.float32, device=dev)
    site_hat = torch.empty((1, 1), dtype=torch.int64, device=dev)
    _lib.check(_lib.load().toad_heads_fwd_f32(_p(m), _p(sex), _p(wcls), _p(bcls), _p(wsite), _p(bsite), _p(mcat), _p(logits),
                                              _p(y_prob), _p(y_hat), _p(site_logits), _p(site_prob), _p(site_hat), l, c,
                                              _stream()), "toad_heads_fwd_f32")
    return mcat, logits, y_prob, y_hat, site_logits, site_prob, site_hat


def heads_bwd(mcat, dlogits, dsite, wcls, wsite, dmcat_ext=None, grads=None, beta: float = 0.0, want_dsex: bool = False):
    """Returns (dWcls, dbcls, dWsite, dbsite, dM [2,L]) (+ dsex [1] with want_dsex: the gradient of the `sex` scalar
    appended to both pooled rows, models/model_toad.py:99). grads = optional tuple of 4 destination tensors."""
    for t_, nm in ((mcat, "mcat"), (dlogits, "dlogits"), (dsite, "dsite"), (wcls, "wcls"), (wsite, "wsite")):
        _chk(t_, nm)
    _chk(dmcat_ext, "dmcat_ext", allow_none=True)
    l = mcat.shape[1] - 1
    c = wcls.shape[0]
    dev = mcat.device
    if dlogits.numel() != c or dsite.numel() != 2:
        raise ValueError("heads_bwd: shape mismatch")
    if grads is None:
        grads = (torch.empty_like(wcls), torch.empty((c,), dtype=torch.float32, device=dev),
                 torch.empty_like(wsite), torch.empty((2,), dtype=torch.float32, device=dev))
        beta = 0.0
    dwcls, dbcls, dwsite, dbsite = grads
    for t_, nm in ((dwcls, "dwcls"), (dbcls, "dbcls"), (dwsite, "dwsite"), (dbsite, "dbsite")):
        _chk(t_, nm)
    dm = torch.empty((2, l), dtype=torch.float32, device=dev)
    dsex = torch.empty((1,), dtype=torch.float32, device=dev) if want_dsex else None
    _lib.check(_lib.load().toad_heads_bwd_f32(_p(mcat), _p(dlogits), _p(dsite), _p(wcls), _p(wsite), _p(dmcat_ext),
                                              _p(dwcls), _p(dbcls), _p(dwsite), _p(dbsite), _p(dm), _p(dsex), float(beta), l, c,
                                              _stream()), "toad_heads_bwd_f32")
    if want_dsex:
        return dwcls, dbcls, dwsite, dbsite, dm, dsex
    return dwcls, dbcls, dwsite, dbsite, dm


def mtl_ce_fwd_bwd(logits, site_logits, label, site, w_cls: float = 0.75, w_site: float = 0.25):
    """Weighted two-task CE and its gradient. Returns (loss_vec[3] = loss, cls_loss, site_loss; dlogits; dsite)."""
    _chk(logits, "logits"); _chk(site_logits, "site_logits")
    _chk(label, "label", dtype=torch.int64); _chk(site, "site", dtype=torch.int64)
    c = logits.numel()
    dev = logits.device
    loss = torch.empty((3,), dtype=torch.float32, device=dev)
    dlogits = torch.empty((1, c), dtype=torch.float32, device=dev)
    dsite = torch.empty((1, 2), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().toad_mtl_ce_fwd_bwd_f32(_p(logits), _p(site_logits), _p(label), _p(site), float(w_cls),
                                                   float(w_site), _p(loss), _p(dlogits), _p(dsite), c, _stream()),
               "toad_mtl_ce_fwd_bwd_f32")
    return loss, dlogits, dsite


# ---- whole per-slide step in one C call (toad_mil_step_f32) -------------------------------------------
STEP_SLOTS = ("w1", "b1", "w2", "b2", "wab", "bab", "wc", "bc", "wcls", "bcls", "wsite", "bsite")
_GEMM_EVENT_NAMES = ("gemm_fwd", "gemm_fwd", "gemm_fwd", "gemm_wgrad", "gemm_dgrad", "gemm_wgrad", "gemm_dgrad", "gemm_wgrad")


def _ptr_array(tensors):
    import ctypes
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


class PreparedBag:
    """A slide's bag in the form the first Linear and its weight gradient consume directly (toad_bag_prepare_f32, ABI 9): the two
    fp16 pieces of every element, plane-tiled in the GEMMs' LDS stage order, plus the bag's abs-max array. Same 4 bytes per
    element as the fp32 bag it was made from (which the caller may drop). Made once per slide - at ingest - because a bag is an
    input that stays the same across epochs (datasets/dataset_mtl_concat.py:369-373). Quacks like the [N, 1024] tensor it stands
    for where the host code only asks for shape / device; it is not differentiable."""

    is_prepared_bag = True
    requires_grad = False
    is_cuda = True
    dtype = torch.float32            # the values it represents (fp32 precision)

    def __init__(self, planes: torch.Tensor, amax: torch.Tensor, n: int, k: int):
        self.planes, self.amax, self.shape = planes, amax, torch.Size((n, k))

    @property
    def device(self):
        return self.planes.device

    def dim(self) -> int:
        return 2

    def contiguous(self):
        return self

    def to(self, *args, **kwargs):           # already resident in its final form (train._to_device calls data.to(device))
        dev = kwargs.get("device", args[0] if args and isinstance(args[0], (str, torch.device)) else None)
        if dev is not None and torch.device(dev).type == "cuda" and torch.device(dev).index not in (None, self.planes.device.index):
            raise RuntimeError(f"PreparedBag lives on {self.planes.device}; prepare it on the device that trains on it")
        if dev is not None and torch.device(dev).type != "cuda":
            raise RuntimeError("PreparedBag is a device-side format (toad_amd has no CPU path)")
        return self

    def nbytes(self) -> int:
        return self.planes.numel() + 4 * self.amax.numel()


def prepare_bag(x: torch.Tensor, out: Optional[PreparedBag] = None) -> PreparedBag:
    """fp32 (or fp16 / bf16: up-cast first) bag [N, K] on the device -> PreparedBag. One pass for the abs-max array, one for the
    split (reads the bag twice, writes it once: ~0.2 ms per 100,000 x 1024 bag, off the training stream when done at ingest).
    ``out``: a PreparedBag of the same shape whose buffers are overwritten (a double-buffered ingest loop allocates two and
    alternates; the caller orders the overwrite behind the last step that read them)."""
    if x.dtype != torch.float32:
        x = x.float()
    x = x.contiguous()
    _chk(x, "bag")
    n, k = x.shape
    if n == 0 or k % 8 != 0:
        raise ValueError("prepare_bag: needs a non-empty [N, K] bag with K a multiple of 8")
    lib = _lib.load()
    if out is not None:
        if tuple(out.shape) != (n, k) or out.planes.device != x.device:
            raise ValueError("prepare_bag: `out` must be a PreparedBag of the same shape on the same device")
        planes, amax = out.planes, out.amax
    else:
        planes = torch.empty(int(lib.toad_bag_planes_bytes(n, k)), dtype=torch.uint8, device=x.device)
        amax = torch.empty(amax_floats(n), dtype=torch.float32, device=x.device)
    _lib.check(lib.toad_bag_prepare_f32(_p(x), n, k, _p(planes), _p(amax), _stream()), "toad_bag_prepare_f32")
    return out if out is not None else PreparedBag(planes, amax, n, k)


def _chk_bag(bag) -> int:
    """The bag of a whole-slide call: fp32 (-> 0), fp16 (features stored in half precision: toad_mil_*_x16_f32, -> 1) or a
    PreparedBag (toad_mil_*_xp_f32, -> 2)."""
    if getattr(bag, "is_prepared_bag", False):
        _chk(bag.planes, "bag planes", dtype=torch.uint8); _chk(bag.amax, "bag amax")
        return 2
    if bag.dtype == torch.float16:
        _chk(bag, "bag", dtype=torch.float16)
        return 1
    _chk(bag, "bag")
    return 0


def _step_dims(w, bag):
    c = w["wcls"].shape[0]
    d = w["wc"].shape[1]
    if bag.shape[1] != 1024 or w["w1"].shape != (512, 1024) or w["wab"].shape != (2 * d, 512):
        raise ValueError("whole-slide calls need the TOAD trunk shapes (1024 -> 512 -> 2 x 384|256)")
    return bag.shape[0], c, d


def mil_step(w, grads, beta: float, bag, sex, label, site, w_cls: float = 0.75, w_site: float = 0.25,
             drop_p: float = 0.0, seed: int = 0, want_logits: bool = False, x_amax: Optional[torch.Tensor] = None):
    """forward + weighted CE + backward for one slide in ONE library call. ``w`` / ``grads`` map the STEP_SLOTS
    to tensors (grads = beta*grads + gradient). Returns (loss[3], logits [1,C] | None, site_logits [1,2] | None)."""
    import ctypes
    half = _chk_bag(bag)
    _chk(sex, "sex"); _chk(label, "label", dtype=torch.int64); _chk(site, "site", dtype=torch.int64)
    _chk(x_amax, "x_amax", allow_none=True)
    ws_t = [w[k] for k in STEP_SLOTS]
    gs_t = [grads[k] for k in STEP_SLOTS]
    for k, t in zip(STEP_SLOTS, ws_t):
        _chk(t, k)
    for k, t in zip(STEP_SLOTS, gs_t):
        _chk(t, "grad " + k)
    n, c, d = _step_dims(w, bag)
    lib = _lib.load()
    dev = bag.device
    ws = _ws(lib.toad_mil_step_ws_bytes(n, c, d), dev, "step")
    loss = torch.empty((3,), dtype=torch.float32, device=dev)
    logits = torch.empty((1, c), dtype=torch.float32, device=dev) if want_logits else None
    slog = torch.empty((1, 2), dtype=torch.float32, device=dev) if want_logits else None
    events = None
    ev_objs = None
    if _TIMING is not None and _timing_due():
        nev = 18 if _TIMING_LEVEL >= 2 else 2
        ev_objs = [_take_event() for _ in range(nev)]
        events = (ctypes.c_void_p * 18)(*([e.cuda_event for e in ev_objs] + [None] * (18 - nev)))
    if half == 2:
        _lib.check(lib.toad_mil_step_xp_f32(_ptr_array(ws_t), _ptr_array(gs_t), float(beta), _p(bag.planes), _p(bag.amax), _p(sex), _p(label),
                                            _p(site), float(w_cls), float(w_site), n, c, d, float(drop_p), int(seed), _p(loss), _p(logits),
                                            _p(slog), _p(ws), ws.numel(), events, _stream()), "toad_mil_step_xp_f32")
    elif half:
        _lib.check(lib.toad_mil_step_x16_f32(_ptr_array(ws_t), _ptr_array(gs_t), float(beta), _p(bag), _p(sex), _p(label), _p(site),
                                             float(w_cls), float(w_site), n, c, d, float(drop_p), int(seed), _p(loss), _p(logits),
                                             _p(slog), _p(ws), ws.numel(), events, _stream()), "toad_mil_step_x16_f32")
    else:
        _lib.check(lib.toad_mil_step_f32(_ptr_array(ws_t), _ptr_array(gs_t), float(beta), _p(bag), _p(sex), _p(label), _p(site),
                                         float(w_cls), float(w_site), n, c, d, float(drop_p), int(seed), _p(x_amax), _p(loss), _p(logits),
                                         _p(slog), _p(ws), ws.numel(), events, _stream()), "toad_mil_step_f32")
    if ev_objs is not None:
        _TIMING.setdefault("pool_fwd", []).append((ev_objs[0], ev_objs[1]))
        if len(ev_objs) == 18:
            for i, name in enumerate(_GEMM_EVENT_NAMES):
                _TIMING.setdefault(name, []).append((ev_objs[2 + 2 * i], ev_objs[3 + 2 * i]))
    return loss, logits, slog


def _adjacent_rows(bags):
    """Bags that already lie back to back in ONE allocation (an ingest buffer filled slide after slide, or views cut from a concatenated
    tensor) are their own concatenation: return it as a view, or None. Saves the copy of every bag row that torch.cat would make."""
    first = bags[0]
    if first.dtype != torch.float32 or first.dim() != 2:
        return None
    store = first.untyped_storage().data_ptr()
    nxt, rows = first.data_ptr() + first.numel() * 4, first.shape[0]
    for b in bags[1:]:
        if b.dtype != torch.float32 or b.dim() != 2 or b.shape[1] != first.shape[1] or not b.is_contiguous() \
                or b.untyped_storage().data_ptr() != store or b.data_ptr() != nxt:
            return None
        nxt += b.numel() * 4
        rows += b.shape[0]
    return torch.as_strided(first, (rows, first.shape[1]), (first.shape[1], 1))


def mil_multi_step(w, grads, beta: float, bags, sex, label, site, w_cls: float = 0.75, w_site: float = 0.25,
                   drop_p: float = 0.0, seed: int = 0, want_logits: bool = False, offsets=None):
    """forward + weighted CE + backward for a BATCH of slides in ONE library call (toad_mil_multi_step_f32): the trunk / attention GEMMs
    run once over the concatenated bags, pooling + heads + loss per slide. ``bags``: a list of fp32 [N_b, 1024] device tensors
    (concatenated here), or ONE already concatenated [sum N_b, 1024] tensor together with ``offsets`` (list of B + 1 row offsets).
    ``sex`` [B] float32, ``label`` / ``site`` [B] int64 device tensors. grads = beta*grads + sum over the batch.
    Returns (loss [B,3], logits [B,C] | None, site_logits [B,2] | None)."""
    import ctypes
    if offsets is None:
        bags = [b.contiguous() for b in bags]
        offsets = [0]
        for b in bags:
            offsets.append(offsets[-1] + int(b.shape[0]))
        xcat = bags[0] if len(bags) == 1 else _adjacent_rows(bags)
        if xcat is None:
            xcat = torch.cat(bags, 0)
    else:
        xcat = bags
        offsets = [int(o) for o in offsets]
    nb = len(offsets) - 1
    _chk(xcat, "bags"); _chk(sex, "sex"); _chk(label, "label", dtype=torch.int64); _chk(site, "site", dtype=torch.int64)
    if xcat.dim() != 2 or xcat.shape[0] != offsets[-1] or sex.numel() != nb or label.numel() != nb or site.numel() != nb:
        raise ValueError("mil_multi_step: one sex / label / site entry per slide and offsets[-1] == rows of the concatenation")
    ws_t = [w[k] for k in STEP_SLOTS]
    gs_t = [grads[k] for k in STEP_SLOTS]
    for k, t in zip(STEP_SLOTS, ws_t):
        _chk(t, k)
    for k, t in zip(STEP_SLOTS, gs_t):
        _chk(t, "grad " + k)
    n, c, d = _step_dims(w, xcat)
    lib = _lib.load()
    dev = xcat.device
    nbytes = int(lib.toad_mil_multi_ws_bytes(n, nb, c, d))
    if nbytes == 0:
        raise ValueError(f"mil_multi_step: unsupported batch (rows {n}, slides {nb})")
    ws = _ws(nbytes, dev, "step")
    loss = torch.empty((nb, 3), dtype=torch.float32, device=dev)
    logits = torch.empty((nb, c), dtype=torch.float32, device=dev) if want_logits else None
    slog = torch.empty((nb, 2), dtype=torch.float32, device=dev) if want_logits else None
    offs = (ctypes.c_int64 * (nb + 1))(*offsets)
    events = None
    ev_objs = None
    if _TIMING is not None and _timing_due():                # same 18-event layout as mil_step: pool forward, then the eight GEMM calls
        nev = 18 if _TIMING_LEVEL >= 2 else 2
        ev_objs = [_take_event() for _ in range(nev)]
        events = (ctypes.c_void_p * 18)(*([e.cuda_event for e in ev_objs] + [None] * (18 - nev)))
    _lib.check(lib.toad_mil_multi_step_f32(_ptr_array(ws_t), _ptr_array(gs_t), float(beta), _p(xcat), offs, nb, _p(sex), _p(label), _p(site),
                                           float(w_cls), float(w_site), c, d, float(drop_p), int(seed), _p(loss), _p(logits), _p(slog),
                                           _p(ws), ws.numel(), events, _stream()), "toad_mil_multi_step_f32")
    if ev_objs is not None:
        _TIMING.setdefault("pool_fwd", []).append((ev_objs[0], ev_objs[1]))
        if len(ev_objs) == 18:
            for i, name in enumerate(_GEMM_EVENT_NAMES):
                _TIMING.setdefault(name, []).append((ev_objs[2 + 2 * i], ev_objs[3 + 2 * i]))
            # whole-call bucket from the events the library already records: first GEMM's start .. the deferred weight-gradient reduction's end
            # (the weight split launch in front of the first GEMM, ~6 us, is outside it; no extra event packets on the stream)
            _TIMING.setdefault("mil_multi_step", []).append((ev_objs[2], ev_objs[17]))
    return loss, logits, slog


# ---- model(data, sex) and loss.backward() as one C call each (toad_mil_fwd_f32 / toad_mil_bwd_f32) ---------------------
ARENA_SLOTS = ("h1", "h", "p", "a_raw", "stats", "m", "mcat", "logits", "y_prob", "y_hat", "site_logits", "site_prob", "site_hat",
               "x_amax", "h1_amax", "h_amax", "h1_bits", "h_bits")


class MilArena:
    """The forward arena of one slide: saved activations + outputs, as typed views of ONE allocation."""

    def __init__(self, n: int, c: int, d: int, device, cached: bool = False):
        """``cached``: carve the arena out of the per-(device, stream) workspace cache instead of a fresh allocation - for forwards
        whose activations nobody will read back (no_grad / eval): the caller must CLONE the outputs it keeps, since the next
        cached forward overwrites them (a view of a fresh arena would pin ~7 KB per patch for as long as any output lives)."""
        self.n, self.c, self.d = n, c, d
        nbytes, align, offs = _arena_layout(n, c, d)          # (three library queries, cached per shape: this runs once per forward)
        self.buf = _ws(nbytes, device, "eval_arena")[:nbytes] if cached else torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.base = (-self.buf.data_ptr()) % align
        self.off = offs

    def view(self, name: str, shape, dtype=torch.float32) -> torch.Tensor:
        nbytes = _ELEM_BYTES[dtype]
        for s_ in shape:
            nbytes *= int(s_)
        o = self.base + self.off[name]
        return self.buf[o:o + nbytes].view(dtype).view(*shape)


_ELEM_BYTES = {torch.float32: 4, torch.int64: 8, torch.uint8: 1, torch.float16: 2, torch.int32: 4}
_ARENA_LAYOUTS = {}
_SCRATCH_BYTES = {}


def _arena_layout(n: int, c: int, d: int):
    """(arena bytes, base alignment, slot -> offset from the aligned base) of toad_mil_arena_layout, cached per (N, C, D)."""
    key = (n, c, d)
    lay = _ARENA_LAYOUTS.get(key)
    if lay is None:
        import ctypes
        lib = _lib.load()
        offs = (ctypes.c_int64 * len(ARENA_SLOTS))()
        _lib.check(lib.toad_mil_arena_layout(n, c, d, offs), "toad_mil_arena_layout")
        lay = (int(lib.toad_mil_arena_bytes(n, c, d)), int(lib.toad_mil_buffer_align(n)), {k: int(o) for k, o in zip(ARENA_SLOTS, offs)})
        if len(_ARENA_LAYOUTS) > 4096:
            _ARENA_LAYOUTS.clear()
        _ARENA_LAYOUTS[key] = lay
    return lay


def _scratch_bytes(n: int, c: int, d: int) -> int:
    key = (n, c, d)
    b = _SCRATCH_BYTES.get(key)
    if b is None:
        b = int(_lib.load().toad_mil_scratch_bytes(n, c, d))
        if len(_SCRATCH_BYTES) > 4096:
            _SCRATCH_BYTES.clear()
        _SCRATCH_BYTES[key] = b
    return b


def mil_fwd(w, bag, sex, drop_p: float = 0.0, seed: int = 0, attention_only: bool = False,
            x_amax: Optional[torch.Tensor] = None, cached_arena: bool = False) -> MilArena:
    """models/model_toad.py:90-116 for one bag in ONE library call; returns the arena (outputs + what backward needs).
    ``cached_arena``: forward-only use (see MilArena): clone what you keep."""
    half = _chk_bag(bag)
    _chk(sex, "sex", allow_none=attention_only); _chk(x_amax, "x_amax", allow_none=True)
    ws_t = [w[k] for k in STEP_SLOTS]
    for k, t in zip(STEP_SLOTS, ws_t):
        _chk(t, k)
    n, c, d = _step_dims(w, bag)
    lib = _lib.load()
    arena = MilArena(n, c, d, bag.device, cached=cached_arena)
    scratch = _ws(_scratch_bytes(n, c, d), bag.device, "mil")
    with _timed("mil_fwd"):
        if half == 2:
            _lib.check(lib.toad_mil_fwd_xp_f32(_ptr_array(ws_t), _p(bag.planes), _p(bag.amax), _p(sex), n, c, d, float(drop_p), int(seed),
                                               1 if attention_only else 0, _p(arena.buf), arena.buf.numel(), _p(scratch), scratch.numel(),
                                               _stream()), "toad_mil_fwd_xp_f32")
        elif half:
            _lib.check(lib.toad_mil_fwd_x16_f32(_ptr_array(ws_t), _p(bag), _p(sex), n, c, d, float(drop_p), int(seed),
                                                1 if attention_only else 0, _p(arena.buf), arena.buf.numel(), _p(scratch), scratch.numel(),
                                                _stream()), "toad_mil_fwd_x16_f32")
        else:
            _lib.check(lib.toad_mil_fwd_f32(_ptr_array(ws_t), _p(bag), _p(sex), n, c, d, float(drop_p), int(seed), _p(x_amax),
                                            1 if attention_only else 0, _p(arena.buf), arena.buf.numel(), _p(scratch), scratch.numel(),
                                            _stream()), "toad_mil_fwd_f32")
    return arena


def mil_bwd(w, grads, beta: float, bag, arena: MilArena, dlogits, dsite, da_ext=None, dmcat_ext=None, drop_p: float = 0.0,
            seed: int = 0, need_dx: bool = False, need_dsex: bool = False):
    """Backward of mil_fwd in ONE library call: grads[slot] = beta*grads[slot] + gradient. Returns (dX | None, dsex | None)."""
    half = _chk_bag(bag)
    if half and need_dx:
        raise ValueError("mil_bwd: no gradient with respect to an fp16 / prepared bag (pass the float32 bag if the bag itself is trained)")
    _chk(dlogits, "dlogits"); _chk(dsite, "dsite")
    _chk(da_ext, "da_ext", allow_none=True); _chk(dmcat_ext, "dmcat_ext", allow_none=True)
    ws_t = [w[k] for k in STEP_SLOTS]
    gs_t = [grads[k] for k in STEP_SLOTS]
    for k, t in zip(STEP_SLOTS, gs_t):
        _chk(t, "grad " + k)
    n, c, d = arena.n, arena.c, arena.d
    if dlogits.numel() != c or dsite.numel() != 2 or bag.shape[0] != n:
        raise ValueError("mil_bwd: shape mismatch")
    if da_ext is not None and tuple(da_ext.shape) != (n, 2):
        raise ValueError("mil_bwd: da_ext must be [N,2]")
    if dmcat_ext is not None and tuple(dmcat_ext.shape) != (2, 513):
        raise ValueError("mil_bwd: dmcat_ext must be [2,513]")
    lib = _lib.load()
    dev = bag.device
    dx = torch.empty_like(bag) if need_dx else None
    dsex = torch.empty((1,), dtype=torch.float32, device=dev) if need_dsex else None
    scratch = _ws(_scratch_bytes(n, c, d), dev, "mil")
    buf = arena.buf
    with _timed("mil_bwd"):
        if half == 2:
            _lib.check(lib.toad_mil_bwd_xp_f32(_ptr_array(ws_t), _ptr_array(gs_t), float(beta), _p(bag.planes), _p(bag.amax), n, c, d, float(drop_p),
                                               int(seed), _p(buf), buf.numel(), _p(dlogits), _p(dsite), _p(da_ext), _p(dmcat_ext), _p(dsex),
                                               _p(scratch), scratch.numel(), _stream()), "toad_mil_bwd_xp_f32")
        elif half:
            _lib.check(lib.toad_mil_bwd_x16_f32(_ptr_array(ws_t), _ptr_array(gs_t), float(beta), _p(bag), n, c, d, float(drop_p), int(seed),
                                                _p(buf), buf.numel(), _p(dlogits), _p(dsite), _p(da_ext), _p(dmcat_ext), _p(dsex),
                                                _p(scratch), scratch.numel(), _stream()), "toad_mil_bwd_x16_f32")
        else:
            _lib.check(lib.toad_mil_bwd_f32(_ptr_array(ws_t), _ptr_array(gs_t), float(beta), _p(bag), n, c, d, float(drop_p), int(seed),
                                            _p(buf), buf.numel(), _p(dlogits), _p(dsite), _p(da_ext), _p(dmcat_ext), _p(dx), _p(dsex),
                                            _p(scratch), scratch.numel(), _stream()), "toad_mil_bwd_f32")
    return dx, dsex


# ---- feature-extractor pieces (conv.hip) --------------------------------------------------------------------------
def linear_act_res_fwd(x, w, b, residual, act: int) -> torch.Tensor:
    """Y = act(X W^T + b + residual): a convolution-as-GEMM with folded BN, the bottleneck's skip add and ReLU."""
    _chk(x, "x"); _chk(w, "w"); _chk(b, "b", allow_none=True); _chk(residual, "residual", allow_none=True)
    m, k = x.shape
    n, k2 = w.shape
    if k != k2 or (b is not None and b.numel() != n) or (residual is not None and tuple(residual.shape) != (m, n)):
        raise ValueError(f"linear_act_res_fwd: shape mismatch x{tuple(x.shape)} w{tuple(w.shape)}")
    y = torch.empty((m, n), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    ws = _ws(lib.toad_linear_ws_bytes(m, n, k), x.device)
    _lib.check(lib.toad_linear_act_res_fwd_f32(_p(x), _p(w), _p(b), _p(residual), _p(y), m, k, n, act, _p(ws), ws.numel(),
                                               _stream()), "toad_linear_act_res_fwd_f32")
    return y


def conv_nhwc(x: torch.Tensor, wf: torch.Tensor, b, residual, kh: int, kw: int, stride: int, pad: int, act: int) -> torch.Tensor:
    """Implicit-GEMM convolution: x [B,H,W,Cin] NHWC, wf [Cout, kh*kw*Cin] in (ky,kx,c) order -> [B,Ho,Wo,Cout]."""
    _chk(x, "x"); _chk(wf, "wf"); _chk(b, "b", allow_none=True); _chk(residual, "residual", allow_none=True)
    bb, h, w, c = x.shape
    cout = wf.shape[0]
    if wf.shape[1] != kh * kw * c:
        raise ValueError("conv_nhwc: weight shape mismatch")
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    y = torch.empty((bb, ho, wo, cout), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    ws = _ws(lib.toad_linear_ws_bytes(bb * ho * wo, cout, kh * kw * c), x.device)
    _lib.check(lib.toad_conv_nhwc_f32(_p(x), _p(wf), _p(b), _p(residual), _p(y), bb, h, w, c, kh, kw, stride, pad, cout, act,
                                      _p(ws), ws.numel(), _stream()), "toad_conv_nhwc_f32")
    return y


def im2col_nhwc(x: torch.Tensor, kh: int, kw: int, stride: int, pad: int) -> torch.Tensor:
    """x [B,H,W,C] -> cols [B*Ho*Wo, kh*kw*C], column order (ky, kx, c)."""
    _chk(x, "x")
    b, h, w, c = x.shape
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    cols = torch.empty((b * ho * wo, kh * kw * c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().toad_im2col_nhwc_f32(_p(x), _p(cols), b, h, w, c, kh, kw, stride, pad, _stream()), "toad_im2col_nhwc_f32")
    return cols


def im2col_stem_nchw(x: torch.Tensor) -> torch.Tensor:
    """x [B,3,H,W] -> cols [B*Ho*Wo, 160] for the 7x7/2 pad-3 stem, column order (c, ky, kx), columns 147.. zero."""
    _chk(x, "x")
    b, c, h, w = x.shape
    if c != 3:
        raise ValueError("im2col_stem_nchw: expected 3 channels")
    ho, wo = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    cols = torch.empty((b * ho * wo, 160), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().toad_im2col_stem_nchw_f32(_p(x), _p(cols), b, h, w, _stream()), "toad_im2col_stem_nchw_f32")
    return cols


def stem_conv(x: torch.Tensor, wf: torch.Tensor, b, act: int) -> torch.Tensor:
    """7x7/2 pad-3 stem on NCHW tiles via the space-to-depth image: x [B,3,H,W], wf [64,192] -> [B,Ho,Wo,64] NHWC."""
    _chk(x, "x"); _chk(wf, "wf"); _chk(b, "b", allow_none=True)
    bb, c, h, w = x.shape
    if c != 3 or tuple(wf.shape) != (64, 192):
        raise ValueError("stem_conv: expected [B,3,H,W] tiles and a [64,192] space-to-depth weight")
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    xs = torch.empty((bb, ho + 3, wo + 3, 12), dtype=torch.float32, device=x.device)
    y = torch.empty((bb, ho, wo, 64), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.toad_stem_s2d_nchw_f32(_p(x), _p(xs), bb, h, w, _stream()), "toad_stem_s2d_nchw_f32")
    ws = _ws(lib.toad_linear_ws_bytes(bb * ho * wo, 64, 192), x.device)
    _lib.check(lib.toad_stem_conv_s2d_f32(_p(xs), _p(wf), _p(b), _p(y), bb, ho, wo, act, _p(ws), ws.numel(), _stream()), "toad_stem_conv_s2d_f32")
    return y


def stem_conv_pool(x: torch.Tensor, wf: torch.Tensor, b) -> torch.Tensor:
    """Stem + ReLU + 3x3/2 max-pool as one kernel (tiles 256 wide: Wo == 128, Ho even): x [B,3,H,W] -> [B,Ho/2,Wo/2,64] NHWC,
    bit-identical to maxpool3x3s2_nhwc(stem_conv(x, wf, b, ACT_RELU))."""
    _chk(x, "x"); _chk(wf, "wf"); _chk(b, "b", allow_none=True)
    bb, c, h, w = x.shape
    if c != 3 or tuple(wf.shape) != (64, 192):
        raise ValueError("stem_conv_pool: expected [B,3,H,W] tiles and a [64,192] space-to-depth weight")
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    xs = torch.empty((bb, ho + 3, wo + 3, 12), dtype=torch.float32, device=x.device)
    y = torch.empty((bb, ho // 2, wo // 2, 64), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.toad_stem_s2d_nchw_f32(_p(x), _p(xs), bb, h, w, _stream()), "toad_stem_s2d_nchw_f32")
    ws = _ws(lib.toad_linear_ws_bytes(bb * ho * wo, 64, 192), x.device)
    _lib.check(lib.toad_stem_conv_pool_s2d_f32(_p(xs), _p(wf), _p(b), _p(y), bb, ho, wo, _p(ws), ws.numel(), _stream()), "toad_stem_conv_pool_s2d_f32")
    return y


def stem_pool_nchw(x: torch.Tensor, wf: torch.Tensor, b) -> torch.Tensor:
    """Stem + ReLU + 3x3/2 max-pool straight from NCHW tiles (W == 256, H % 4 == 0): x [B,3,H,W] -> [B,H/4,64,64] NHWC; the values of
    maxpool3x3s2_nhwc(stem_conv(x, wf, b, ACT_RELU)) to fp32 round-off (per-tile operand scales)."""
    _chk(x, "x"); _chk(wf, "wf"); _chk(b, "b", allow_none=True)
    bb, c, h, w = x.shape
    if c != 3 or tuple(wf.shape) != (64, 192):
        raise ValueError("stem_pool_nchw: expected [B,3,H,W] tiles and a [64,192] space-to-depth weight")
    y = torch.empty((bb, h // 4, w // 4, 64), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    ws = _ws(lib.toad_linear_ws_bytes(bb * (h // 2) * (w // 2), 64, 192), x.device)
    _lib.check(lib.toad_stem_pool_nchw_f32(_p(x), _p(wf), _p(b), _p(y), bb, h, w, _p(ws), ws.numel(), _stream()), "toad_stem_pool_nchw_f32")
    return y


def maxpool3x3s2_nhwc(x: torch.Tensor) -> torch.Tensor:
    _chk(x, "x")
    b, h, w, c = x.shape
    y = torch.empty((b, (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1, c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().toad_maxpool3x3s2_nhwc_f32(_p(x), _p(y), b, h, w, c, _stream()), "toad_maxpool3x3s2_nhwc_f32")
    return y


def avgpool_nhwc(x: torch.Tensor) -> torch.Tensor:
    """x [B,HW,C] -> [B,C] mean over HW."""
    _chk(x, "x")
    b, hw, c = x.shape
    y = torch.empty((b, c), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().toad_avgpool_nhwc_f32(_p(x), _p(y), b, hw, c, _stream()), "toad_avgpool_nhwc_f32")
    return y

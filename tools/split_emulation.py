"""CPU emulation of the split-operand GEMM arithmetics against fp64 (no GPU needed): exact-fp32 fma chain, three-piece bf16 split
(six MFMA terms, the extractor path) and two-piece fp16 split (three terms, the MIL path, csrc/gemm_h2.inc). Each MFMA is modelled as an
exact 16-deep dot product added to an fp32 accumulator. Prints max / rms error, error relative to sum|a.b| and the mean signed error."""
import numpy as np
rng = np.random.default_rng(0)
M, N, K = 256, 128, 1024
def emulate(A, B, split, terms):
    # A [M,K], B [N,K] fp32. returns fp32 result emulating MFMA: 16-deep exact-product groups summed in fp64, then fp32 accumulate
    Ap = split(A); Bp = split(B)
    acc = np.zeros((M, N), np.float32)
    for k0 in range(0, K, 16):
        s = np.zeros((M, N), np.float64)
        for (i, j) in terms:
            # each term is its own MFMA: acc = acc + sum16 (rounded to fp32 after each)
            t = Ap[i][:, k0:k0+16].astype(np.float64) @ Bp[j][:, k0:k0+16].astype(np.float64).T
            acc = (acc.astype(np.float64) + t).astype(np.float32)
    return acc
def split_bf16_trunc(X):
    def tr(x):
        u = x.view(np.uint32) & np.uint32(0xFFFF0000); return u.view(np.float32)
    h = tr(X.copy()); r1 = X - h; m = tr(r1.copy()); r2 = r1 - m; l = tr(r2.copy())
    return [h, m, l]
def make_split_f16(rn=True):
    def sp(X):
        amax = np.abs(X).max()
        e = np.ceil(np.log2(amax)) if amax > 0 else 0
        s = np.float32(2.0 ** (14 - e))
        v = (X * s).astype(np.float32)
        h = v.astype(np.float16).astype(np.float32)
        r = v - h
        m = r.astype(np.float16).astype(np.float32)
        return [h / s, m / s]   # exact power-of-2 division: keeps emulation in original units
    return sp
def fp32_chain(A, B):
    acc = np.zeros((M, N), np.float32)
    for k in range(K):
        acc = np.float32(acc + (A[:, k:k+1] * B[:, k].reshape(1, -1)).astype(np.float32))  # product rounding too (fma would not)
    return acc
def fp32_fma_chain(A, B):
    acc = np.zeros((M, N), np.float64)
    a64 = A.astype(np.float64); b64 = B.astype(np.float64)
    acc32 = np.zeros((M, N), np.float32)
    for k in range(K):
        acc32 = (acc32.astype(np.float64) + a64[:, k:k+1] * b64[:, k].reshape(1, -1)).astype(np.float32)
    return acc32
for name, gen in (("normal", lambda s: rng.standard_normal(s).astype(np.float32)),
                  ("relu*tiny", lambda s: (np.maximum(rng.standard_normal(s), 0) * 1e-6).astype(np.float32)),
                  ("lognormal wide", lambda s: (rng.standard_normal(s) * np.exp(3 * rng.standard_normal(s))).astype(np.float32))):
    A = gen((M, K)); B = (gen((N, K)) if name != "relu*tiny" else rng.standard_normal((N, K)).astype(np.float32) * 0.03)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    sab = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
    res = {
        "fp32 fma chain": fp32_fma_chain(A, B),
        "bf16x3 trunc 6 terms": emulate(A, B, split_bf16_trunc, [(2,0),(0,2),(1,1),(1,0),(0,1),(0,0)]),
        "fp16x2 RN 3 terms": emulate(A, B, make_split_f16(), [(1,0),(0,1),(0,0)]),
        "fp16x2 RN 4 terms": emulate(A, B, make_split_f16(), [(1,1),(1,0),(0,1),(0,0)]),
    }
    print(f"== {name}: rms|ref|={np.sqrt((ref**2).mean()):.3e}")
    for k, v in res.items():
        err = v.astype(np.float64) - ref
        print(f"  {k:24s} max|err|={np.abs(err).max():.3e} rms={np.sqrt((err**2).mean()):.3e} max err/sum|ab|={np.abs(err/sab).max():.3e} mean signed={err.mean():+.2e}")

"""GPU: short fixed-seed runs of the random-shape parity sweeps (tests/fuzz_*.py; the long sweeps are run by hand, results in
profiles/r03zu_fuzz_sweeps.txt). Each sweep is its own process: it prints one line per case and exits non-zero on the first mismatch class.
The first long sweep of the convolution kernels found a workspace overrun at K = 32 that no hand-picked shape had hit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("script,cases,seed", [("fuzz_conv.py", 60, 11), ("fuzz_gemm.py", 24, 11), ("fuzz_pool.py", 40, 11),
                                               ("fuzz_eval.py", 8, 11), ("fuzz_multi.py", 6, 11), ("fuzz_mil.py", 8, 11), ("fuzz_attn.py", 24, 11)])
def test_random_shape_sweep(cuda, script, cases, seed):
    r = subprocess.run([sys.executable, os.path.join(HERE, script), str(cases), str(seed)], capture_output=True, text=True, timeout=850)
    tail = "\n".join((r.stdout + r.stderr).strip().splitlines()[-25:])
    assert r.returncode == 0, tail
    assert "0 failures" in r.stdout or "worst" in r.stdout, tail

"""Build recipe for libtoad_hip.so (hipcc, gfx950 only, in-tree output).

    python -m toad_amd.build            # build if stale
    python -m toad_amd.build --force
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtoad_hip.so")
SOURCES = ["capi.hip", "gemm_f32.hip", "gated_pool.hip", "heads.hip", "step.hip", "conv.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(REPO, "include", "toad_hip.h")] + \
    [os.path.join(CSRC, f) for f in ("gemm_nt_f32.inc", "gemm_tn.inc", "gemm_h2.inc", "gemm_h2_epilogue.inc", "gemm_pt.inc", "gemm_narrow.inc", "gemm_stream.inc", "stem_halo.inc")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
         "-I" + os.path.join(REPO, "include"), "-I" + CSRC, "-Wall", "-Wno-unused-function"]


LAST_BUILD = {"compiled": [], "linked": False, "reused": False}      # what the last build() call did (reported by __graft_entry__.build)


def _stale(obj: str, deps) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, defines=(), tag: str = "") -> str:
    """Build libtoad_hip{tag}.so. `defines` / `tag` build an experiment variant next to the shipped library (tools/ab/README.md; the measurement tools select it
    through tools/ab/select_lib.py, the product's loader never does); the sources under csrc/ carry no variant switches of their own."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    lib = LIB.replace(".so", tag + ".so")
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", tag + ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            cmd = [hipcc] + FLAGS + ["-D" + d for d in defines] + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    linked = False
    if force or procs or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        linked = True
    global LAST_BUILD
    LAST_BUILD = {"compiled": [src for src, _ in procs], "linked": linked, "reused": not procs and not linked}
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

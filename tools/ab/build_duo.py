"""Build tools/ab/libtoad_duo.so: the shipped sources + the duo-kernel experiment (tools/ab/exp_duo.hip)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "toad_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-I" + os.path.join(REPO, "include"), "-I" + CSRC, "-I" + HERE,
         "-Wall", "-Wno-unused-function"]


def build(extra=()):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src, out in ((os.path.join(HERE, "exp_duo.hip"), "exp_duo.o"), (os.path.join(CSRC, "capi.hip"), "capi_duo.o")):
        o = os.path.join(HERE, out)
        subprocess.check_call([hipcc] + FLAGS + list(extra) + ["-c", src, "-o", o])
        objs.append(o)
    lib = os.path.join(HERE, "libtoad_duo.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build(sys.argv[1:]))

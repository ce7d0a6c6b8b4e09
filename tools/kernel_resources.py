"""Per-kernel register / spill report of one csrc/*.hip file (device pass only).
    python tools/kernel_resources.py gemm_f32 [name regex]"""
import os, re, subprocess, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "."
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-I" + REPO + "/include",
       "-I" + REPO + "/toad_amd/csrc", "-Wno-unused-function", "--offload-device-only", "-Rpass-analysis=kernel-resource-usage", "-c",
       f"{REPO}/toad_amd/csrc/{src}.hip", "-o", f"/tmp/{src}.dev.o"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, []
for l in out.splitlines():
    m = re.search(r"remark: +(.*?)( \[-Rpass.*)?$", l.rstrip())
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    if re.search(flt, name):
        print(f"{name[:84]:<84} VGPR {r.get('VGPRs'):>3} AGPR {r.get('AGPRs'):>3} scratch {r.get('ScratchSize [bytes/lane]'):>4} "
              f"sspill {r.get('SGPRs Spill'):>3} vspill {r.get('VGPRs Spill'):>3} occ {r.get('Occupancy [waves/SIMD]')}")

"""VGPRs / spills / occupancy of every kernel in one csrc/*.hip file, from hipcc's -Rpass-analysis=kernel-resource-usage remarks
(cross-compiles for gfx950, no GPU needed).  Usage: python tools/kernel_resources.py nvalchemi-toolkit-ops_amd/csrc/d3.hip [name-filter]"""
import os
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nvalchemi-toolkit-ops_amd"))
import build_native  # noqa: E402  (the product's per-file flags: the numbers must be those of the shipped objects)

flags = [f for f in build_native.SOURCES.get(os.path.basename(src), []) if f]
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *flags, "-x", "hip", "--cuda-device-only", "-c", src,
                        "-o", os.path.join(d, "k.co"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
rows, cur = [], {}
for line in r.stderr.splitlines():
    m = re.search(r":\d+:\d+: (?:remark: )?(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith(("Function Name:", "Name:")):
        if cur:
            rows.append(cur)
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
if cur:
    rows.append(cur)
for r_ in rows:
    if flt in r_["name"]:
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", r_["name"])[:70]
        print(f"{short:70s} VGPR {r_.get('VGPRs'):>4s} AGPR {r_.get('AGPRs', '-'):>3s} spill {r_.get('VGPRs Spill'):>3s} "
              f"LDS {r_.get('LDS Size [bytes/block]', '-'):>6s} occ {r_.get('Occupancy [waves/SIMD]')}")

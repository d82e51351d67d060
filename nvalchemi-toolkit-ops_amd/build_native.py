"""Builds libnvalchemiops_hip.so (gfx950) in-tree with hipcc.  No CPU fallback is produced: the Python
package refuses to run its ops when this library is missing.

    python nvalchemi-toolkit-ops_amd/build_native.py [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "nvalchemiops", "lib")
LIB = os.path.join(OUT_DIR, "libnvalchemiops_hip.so")
LIB_D3_IEEE = os.path.join(OUT_DIR, "libnvalchemiops_d3_ieee.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")  # the one ROCm installation headers, compiler and link-time libraries come from
HIPCC = os.environ.get("HIPCC", os.path.join(ROCM, "bin", "hipcc"))
ARCH = "gfx950"

# per-file extra flags.  nlist.hip must evaluate the cutoff test exactly like the oracle: no FMA contraction.
SOURCES = {
    "capi.cpp": [],
    "fft.cpp": [],  # hipFFT plans (mi_fft_plan_*): linked with -lhipfft below
    "comm.cpp": [],  # mi_comm_*: RCCL bound at run time (dlopen), nothing to link but libdl
    "nlist.hip": ["-ffp-contract=off"] + os.environ.get("MI_NLIST_EXTRA_FLAGS", "").split(),
    # D3 pair math is fp32 with 1/x and sqrt on every pair: hardware v_rcp/v_sqrt (1 ulp) instead of the IEEE-exact expansions
    # (~10 instructions each); energies/forces stay inside the stated 2e-6 / 1e-5 tolerances
    # -fno-slp-vectorize: the SLP pass fuses adjacent fp32 FMAs into v_pk_fma_f32, which on gfx950 issues as two passes (no gain) and
    # costs v_mov's to pair the operands -- the energy pass is VALU-issue-bound
    "d3.hip": ["-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize"] + os.environ.get("MI_D3_EXTRA_FLAGS", "").split(),
    "ewald.hip": os.environ.get("MI_EWALD_EXTRA_FLAGS", "").split(),
    "pme.hip": os.environ.get("MI_PME_EXTRA_FLAGS", "").split(),
    "dft.hip": [],  # dense DFT of any mesh size: the transform of last resort behind the self-tested hipFFT plans
    "calib.hip": [],
}
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    build_dir = os.path.join(HERE, "build")
    os.makedirs(build_dir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "nvalchemiops_hip.h"))
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(build_dir, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [path] + headers):
            lang = ["-x", "hip"] if src.endswith(".hip") else []
            jobs.append([HIPCC] + COMMON + extra + lang + ["-c", path, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    # tests only: d3.hip with correctly rounded sqrt / divide and libm expf (-DMI_D3_IEEE, no fast-math flag) for the D3 error
    # budget (tests/test_d3_gpu.py).  The product never loads it.
    ieee_obj = os.path.join(build_dir, "d3_ieee.o")
    d3_src = os.path.join(CSRC, "d3.hip")
    if force or _stale(ieee_obj, [d3_src] + headers):
        jobs.append([HIPCC] + COMMON + ["-DMI_D3_IEEE", "-x", "hip", "-c", d3_src, "-o", ieee_obj])

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        # libhipfft.so.0: inside a Python process torch has loaded its own copy of that SONAME already (same rocFFT the torch.fft path
        # used); a plain C caller resolves it from the ROCm installation
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-L" + os.path.join(ROCM, "lib"), "-lhipfft", "-ldl"])
    if force or _stale(LIB_D3_IEEE, [ieee_obj, os.path.join(build_dir, "capi.o")]):
        run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_D3_IEEE, ieee_obj, os.path.join(build_dir, "capi.o")])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

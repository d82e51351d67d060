"""Probe (written at the end of round 4, NOT yet run on a GPU: the round's budget was spent): towards a stand-alone reproducer of the hipFFT
drift described in DESIGN.md 3.7.

Observation to reproduce: late in a process that has created many FFT plans of assorted shapes (torch.fft on the composed autograd path and the
library's own hipFFT plans), a NEWLY created library plan for mesh (16, 8, 32) returns a transform that is a few per cent off numpy, the
same in fp32 and fp64, deterministically.  Hypothesis: rocFFT's per-length twiddle cache hands a new plan entries whose device memory was
released with an earlier plan.

What this does: (1) reference transform of a fixed mesh with a fresh plan, (2) churn -- rfftn / irfftn through torch.fft and through library
plans over a list of shapes sharing the 1-D lengths 8, 16, 32, dropping torch's plan cache in between, (3) NEW library plans for (16, 8, 32)
and friends, compared with numpy.  Prints the relative error per step; a jump at step (3) is the reproducer."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from nvalchemiops import _capi as C  # noqa: E402

dev = torch.device("cuda:0")


def lib_rfftn(mesh):
    nx, ny, nz = mesh.shape
    code = C.dtype_code(mesh.dtype)
    h = ctypes.c_void_p()
    C.check(C.lib().mi_fft_plan_create(nx, ny, nz, 1, code, 0, ctypes.byref(h)), "mi_fft_plan_create")
    out = torch.empty((nx, ny, nz // 2 + 1), dtype=torch.complex128 if mesh.dtype == torch.float64 else torch.complex64, device=dev)
    C.check(C.lib().mi_fft_plan_exec(h, C.ptr(mesh.clone()), C.ptr(out), C.stream_of(mesh)), "mi_fft_plan_exec")
    torch.cuda.synchronize()
    return out, h   # the plan is deliberately kept alive, as the library's cache keeps it


def err(got, mesh):
    want = np.fft.rfftn(mesh.cpu().numpy().astype(np.float64))
    return float(np.abs(got.cpu().numpy() - want).max() / np.abs(want).max())


g = torch.Generator(device="cpu").manual_seed(1)
keep = []
probe_shapes = [(16, 8, 32), (8, 16, 32), (32, 8, 16), (16, 16, 16)]
for dt in (torch.float64, torch.float32):
    m = torch.randn((16, 16, 16), generator=g, dtype=dt).to(dev)
    out, h = lib_rfftn(m)
    keep.append(h)
    print("fresh process", dt, (16, 16, 16), f"{err(out, m):.2e}", flush=True)
churn = [(8, 8, 8), (16, 8, 24), (30, 36, 45), (12, 10, 14), (31, 9, 6), (8, 64, 16), (32, 32, 32), (64, 16, 128), (16, 256, 8), (20, 20, 20), (24, 16, 8), (32, 16, 8)]
for rep in range(int(os.environ.get('REPRO_REPS', '3'))):
    for shp in churn:
        for dt in (torch.float64, torch.float32):
            m = torch.randn(shp, generator=g, dtype=dt).to(dev)
            s = torch.fft.rfftn(m)
            torch.fft.irfftn(s * 1.0, s=shp)
            out, h = lib_rfftn(m)
            keep.append(h)
    try:
        torch.backends.cuda.cufft_plan_cache.clear()
    except Exception as exc:  # not every ROCm build of torch exposes the cache
        print("plan cache clear:", type(exc).__name__)
    for shp in probe_shapes:
        for dt in (torch.float64, torch.float32):
            m = torch.randn(shp, generator=g, dtype=dt).to(dev)
            out, h = lib_rfftn(m)
            keep.append(h)
            t = torch.fft.rfftn(m)
            print(f"after churn {rep}", dt, shp, f"library plan {err(out, m):.2e}   torch.fft {err(t, m):.2e}", flush=True)

"""Probe (round 5): the dense DFT (`mi_dft3d`) against self-tested hipFFT plans, forward + inverse over 4 channels as the reciprocal-space step
uses them, for meshes the fused solve does not take.  Median of 20 event-bracketed calls each, fp64 and fp32.

    python tools/probe/dft_vs_plan.py"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nvalchemi-toolkit-ops_amd")]
from nvalchemiops import _capi as C  # noqa: E402
from nvalchemiops.interactions.electrostatics import pme as P  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)


shapes = [((12, 10, 14), 1), ((20, 18, 24), 1), ((30, 36, 45), 1), ((48, 48, 48), 1), ((60, 60, 60), 1), ((96, 96, 96), 1), ((100, 100, 100), 1),
          ((32, 32, 32), 128), ((24, 24, 24), 128), ((16, 16, 16), 512)]
print("# mesh, batch, dtype: R2C + C2R(4 channels) ms through hipFFT plans | dense DFT | ratio")
for dims, b in shapes:
    nx, ny, nz = dims
    for dt, cdt in ((torch.float64, torch.complex128), (torch.float32, torch.complex64)):
        code = C.dtype_code(dt)
        x = torch.randn((b, nx, ny, nz), dtype=dt, device=dev)
        spec = torch.empty((b, nx, ny, nz // 2 + 1), dtype=cdt, device=dev)
        conv = torch.randn((b, 4, nx, ny, nz // 2 + 1), dtype=dt, device=dev).to(cdt)
        real = torch.empty((b, 4, nx, ny, nz), dtype=dt, device=dev)
        out = {}
        for mode in ("plan", "dft"):
            P._FFT_PLANS.clear()
            P._FORCE_DFT = mode == "dft"
            f = P._fft_plan(dev, dims, b, code, False)
            i = P._fft_plan(dev, dims, b * 4, code, True)
            kind = type(f).__name__ + "/" + type(i).__name__

            def step():
                f(x, spec)
                i(conv, real)
            out[mode] = (timed(step), kind)
        print(f"{dims} x {b} {str(dt).split('.')[-1]}: {out['plan'][0]:.4f} ({out['plan'][1]}) | {out['dft'][0]:.4f} | x{out['dft'][0] / out['plan'][0]:.2f}", flush=True)

"""Probe (round 4; needs the `real_interleaved` plan option this probe was run with -- see git history of csrc/fft.cpp): hipFFT C2R over 4 channels of a 128^3 fp64 mesh, planar output [4][nx][ny][nz] vs channel-interleaved output
[nx][ny][nz][4] (the layout a gather kernel would read as one 32-byte record per mesh point).  Timing + equality of the two results."""
import ctypes, statistics, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "nvalchemi-toolkit-ops_amd"))
import torch
from nvalchemiops import _capi as C
from nvalchemiops.interactions.electrostatics.pme import _FftPlan

dev = torch.device("cuda:0")
for n in (128, 64, 32):
    nz = n // 2 + 1
    for batch in (4, 1):
        spec = torch.randn((batch, n, n, nz), dtype=torch.complex128, device=dev)
        ref = torch.fft.irfftn(spec, s=(n, n, n), dim=(1, 2, 3), norm="forward")
        res = {}
        for inter in (0, 1):
            plan = _FftPlan((n, n, n), batch, 1, True, interleaved=bool(inter))
            out = torch.empty((n, n, n, batch) if inter else (batch, n, n, n), dtype=torch.float64, device=dev)
            ts = []
            for it in range(25):
                src = spec.clone()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); plan(src, out); b.record(); b.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            got = out.permute(3, 0, 1, 2) if inter else out
            res[inter] = (statistics.median(ts[5:]), float((got - ref).abs().max() / ref.abs().max()))
        print(f"n={n} batch={batch}: planar {res[0][0]:.1f} us (err {res[0][1]:.1e})   interleaved {res[1][0]:.1f} us (err {res[1][1]:.1e})")

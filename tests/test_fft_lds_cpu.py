"""Index arithmetic of the fused PME mesh solve (csrc/fft_lds.h) against numpy, without a GPU.

The kernels of `mi_pme_solve` are thin launch wrappers around per-item bodies; tests/native/fft_host_harness.cpp runs those bodies as one
host thread per block.  Checked here: the slot maps, every 1-D plan, the packed R2C / C2R rows, the forward planes against numpy.fft and
the whole k-space step (forward, Green function / B-spline moduli / -i k_d, inverse) against the oracle's numpy restatement of
pme.py:1398-1440.  TEST INFRASTRUCTURE: the product has no CPU path (tests/test_host_cpu.py::test_no_cpu_fallback)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def H(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fft_harness") / "fft_host_harness.so")
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "nvalchemi-toolkit-ops_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "fft_host_harness.cpp"), "-o", so])
    return ctypes.CDLL(so)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _slots(H, n, max_lr=3):
    radix = np.zeros(8, np.int32)
    s_of_f, f_of_s = np.zeros(n, np.int32), np.zeros(n, np.int32)
    nst = H.h_plan(n, max_lr, _p(radix), _p(s_of_f), _p(f_of_s))
    assert nst >= 1 and int(np.prod(radix[:nst])) == n and set(radix[:nst].tolist()) <= {2, 3, 4, 5, 8, 16} and radix[:nst].max() <= max(2 ** max_lr, 5)
    if n & (n - 1) == 0:
        assert nst == -(-int(np.log2(n)) // max_lr)  # powers of two: as few stages as the radix allows
    return s_of_f, f_of_s


# every power of two up to 512, and products of 2, 3 and 5 (round 6: mixed radix): 96 / 100 / 120 are what `mesh_spacing=` callers get
@pytest.mark.parametrize("max_lr", [3, 4])
@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 256, 512, 3, 5, 6, 9, 10, 12, 15, 18, 20, 24, 25, 27, 30, 36, 40, 45, 48, 50, 60, 72, 75, 80, 90, 96,
                               100, 108, 120, 125, 135, 144, 150, 160, 180, 192, 200, 225, 240, 243, 250, 256, 270, 288, 300, 320, 360, 375, 384, 400, 405,
                               432, 450, 480, 486, 500])
def test_line_forward_slots_and_inverse(H, n, max_lr):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    s_of_f, f_of_s = _slots(H, n, max_lr)
    assert np.array_equal(f_of_s[s_of_f], np.arange(n)) and np.array_equal(np.sort(s_of_f), np.arange(n))
    d = np.ascontiguousarray(np.stack([x.real, x.imag], -1))
    assert H.h_line(_p(d), n, max_lr, 0) == 0
    got = d[:, 0] + 1j * d[:, 1]
    assert np.allclose(got[s_of_f], np.fft.fft(x), rtol=0, atol=1e-12 * n)
    # the inverse takes the slots and returns n * x in natural order
    assert H.h_line(_p(d), n, max_lr, 1) == 0
    assert np.allclose(d[:, 0] + 1j * d[:, 1], n * x, rtol=0, atol=1e-12 * n * n)


def _unscramble(H, spec, ny, nz):
    """[..., ny, P] in (y slot, z slot) order -> natural order along y and z"""
    M = nz // 2
    sy, _ = _slots(H, ny)
    sz, _ = _slots(H, M)
    zs = np.concatenate([sz, [M]])
    return spec[..., sy, :][..., zs]


@pytest.mark.parametrize("dims", [(8, 8, 8), (8, 16, 32), (16, 8, 64), (32, 32, 32), (8, 64, 16), (16, 16, 128),
                                  (8, 12, 20), (10, 15, 18), (9, 25, 30), (8, 96, 100), (12, 50, 90), (8, 24, 250)])  # mixed radix; M = nz / 2 odd as well
def test_forward_planes_equal_numpy_over_y_and_z(H, dims):
    nx, ny, nz = dims
    B = 2
    rng = np.random.default_rng(sum(dims))
    mesh = np.ascontiguousarray(rng.standard_normal((B, nx, ny, nz)))
    spec = np.zeros((B, nx, ny, nz // 2 + 1, 2))
    assert H.h_forward(_p(mesh), _p(spec), B, nx, ny, nz) == 0
    got = _unscramble(H, spec[..., 0] + 1j * spec[..., 1], ny, nz)
    want = np.fft.fft(np.fft.rfft(mesh, axis=-1), axis=-2)
    assert np.allclose(got, want, rtol=0, atol=1e-11 * ny * nz)


@pytest.mark.parametrize("dims,order,nch", [((8, 8, 8), 4, 4), ((16, 8, 32), 5, 4), ((8, 32, 16), 3, 1), ((32, 16, 8), 6, 4), ((16, 16, 16), 2, 4),
                                            # every plan shape the kernels can meet: x 128 = 16 x 8 and 256 = 16 x 16 (radix 16), y 256 = 8 x 8 x 4,
                                            # packed z rows of 128 = 8 x 8 x 2 and 64 = 8 x 8 points
                                            ((128, 8, 16), 4, 4), ((256, 8, 8), 4, 1), ((8, 256, 8), 4, 1), ((8, 8, 256), 5, 4), ((64, 16, 128), 4, 4),
                                            # mixed radix (round 6): the sizes a `mesh_spacing=` caller gets, odd packed rows (nz / 2 = 45, 25), three / four stages per axis
                                            ((12, 10, 18), 4, 4), ((20, 18, 24), 5, 4), ((96, 8, 10), 4, 4), ((10, 100, 12), 4, 1), ((8, 9, 120), 5, 4), ((15, 25, 90), 3, 4),
                                            ((48, 20, 50), 4, 4), ((100, 12, 8), 6, 4), ((120, 8, 10), 4, 4), ((240, 8, 8), 4, 1), ((8, 8, 500), 4, 1)])
def test_whole_kspace_step_equals_the_numpy_restatement(H, dims, order, nch):
    """mesh -> rfftn -> (spec / sf2) G, (-i k_d) conv -> irfftn * N for the potential + three field components (the middle of
    oracle.pme_reciprocal_space: pme.py:1398-1440); triclinic cells, two systems with their own alpha."""
    nx, ny, nz = dims
    B = 2
    rng = np.random.default_rng(7 + sum(dims))
    mesh = np.ascontiguousarray(rng.standard_normal((B, nx, ny, nz)))
    cells = np.stack([np.diag([9.0, 10.0, 11.0]) + 0.6 * rng.standard_normal((3, 3)) for _ in range(B)])
    alpha = np.array([0.35, 0.42])
    recip = np.ascontiguousarray(2.0 * np.pi * np.linalg.inv(cells))
    vol = np.ascontiguousarray(np.abs(np.linalg.det(cells)))
    out = np.zeros((B, nch, nx, ny, nz))
    expo = min(order, 4)   # the reference's structure-factor exponent
    nat = np.zeros((B, nx, ny, nz // 2 + 1, 2))
    assert H.h_solve(_p(mesh), _p(out), B, nx, ny, nz, _p(recip), _p(alpha), _p(vol), expo, nch, _p(nat)) == 0
    axes = (-3, -2, -1)
    # the by-product for callers that need the charge spectrum: unfactored, natural frequency order = numpy.fft.rfftn
    assert np.allclose(nat[..., 0] + 1j * nat[..., 1], np.fft.rfftn(mesh, axes=axes), rtol=0, atol=1e-11 * nx * ny * nz)
    kvec, k2 = O.generate_k_vectors_pme(cells, dims)
    g, sf2 = O.pme_green_structure_factor(k2, dims, alpha, cells, order)   # default oracle mode: exponent min(order, 4)
    conv = (np.fft.rfftn(mesh, axes=axes) / sf2) * g
    ntot = float(nx * ny * nz)
    want = [np.fft.irfftn(conv, dims, axes=axes) * ntot]
    for d in range(3):
        want.append(np.fft.irfftn(-1j * kvec[..., d] * conv, dims, axes=axes) * ntot)
    want = np.stack(want[:nch], 1)
    scale = np.abs(want).max()
    assert np.allclose(out, want, rtol=0, atol=1e-11 * scale), np.abs(out - want).max() / scale


@pytest.mark.parametrize("dims", [(8, 8, 8), (16, 8, 32), (8, 32, 16), (128, 8, 16), (8, 8, 256), (12, 10, 18), (20, 18, 24), (96, 8, 10), (10, 100, 12), (8, 9, 120),
                                  (15, 25, 90), (240, 8, 8)])
def test_transforms_on_their_own_equal_numpy(H, dims):
    """`mi_fft_lds` (round 6): the plane + column bodies as an unscaled rfftn / irfftn in natural frequency order -- what the backward of the
    autograd node and every other former hipFFT-plan caller now runs (pme.py:1398, :1422, :1455-1457)."""
    nx, ny, nz = dims
    B = 3
    rng = np.random.default_rng(11 + sum(dims))
    mesh = np.ascontiguousarray(rng.standard_normal((B, nx, ny, nz)))
    axes = (-3, -2, -1)
    nat = np.zeros((B, nx, ny, nz // 2 + 1, 2))
    assert H.h_r2c(_p(mesh), _p(nat), B, nx, ny, nz) == 0
    want = np.fft.rfftn(mesh, axes=axes)
    assert np.allclose(nat[..., 0] + 1j * nat[..., 1], want, rtol=0, atol=1e-11 * nx * ny * nz)
    sp = np.ascontiguousarray(np.stack([want.real, want.imag], -1))
    back = np.zeros((B, nx, ny, nz))
    assert H.h_c2r(_p(sp), _p(back), B, nx, ny, nz) == 0
    assert np.allclose(back, mesh * float(nx * ny * nz), rtol=0, atol=1e-10 * nx * ny * nz)
    # ... and of an arbitrary half spectrum (not the transform of a real mesh): the same numbers as numpy's irfftn, which -- after the x and y
    # transforms -- does not read the imaginary parts of the DC and Nyquist bins along z
    spec = want + (rng.standard_normal(want.shape) + 1j * rng.standard_normal(want.shape))
    sp = np.ascontiguousarray(np.stack([spec.real, spec.imag], -1))
    assert H.h_c2r(_p(sp), _p(back), B, nx, ny, nz) == 0
    ref = np.fft.irfftn(spec, dims, axes=axes) * float(nx * ny * nz)
    assert np.allclose(back, ref, rtol=0, atol=1e-10 * nx * ny * nz)
    assert np.array_equal(sp[..., 0] + 1j * sp[..., 1], spec)  # the input is left alone (hipFFT's multi-dimensional C2R overwrites it)


def test_barrier_placement_under_thread_sanitizer(tmp_path):
    """tests/native/fft_race_check.cpp: the same bodies (fp64 and fp32) with 3 and 16 host threads per block, MI_FFT_SYNC() = a pthread barrier, under
    ThreadSanitizer.  Any two threads touching one LDS / global element between two barriers would be a missing __syncthreads on the GPU:
    none reported, and every thread count gives the single-thread result bit for bit.  Negative control: with the barriers dropped the
    detector does report races (so a clean run means something)."""
    exe = str(tmp_path / "fft_race_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I" + os.path.join(ROOT, "nvalchemi-toolkit-ops_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "fft_race_check.cpp"), "-o", exe])
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert "ThreadSanitizer" not in run.stderr, run.stderr[-3000:]
    lines = [l for l in run.stdout.splitlines() if " mesh " in l]
    assert len(lines) == 9 and all(l.endswith("bit-identical") for l in lines), run.stdout   # fp64: 3 meshes x 2 thread counts, fp32: 3 x 1
    control = subprocess.run([exe, "drop"], capture_output=True, text=True, timeout=600)
    assert "ThreadSanitizer: data race" in control.stderr


def test_kernel_bodies_stay_inside_their_buffers(tmp_path):
    """The same program under AddressSanitizer, single-threaded, with every buffer -- the LDS of each kernel as `*_lds_bytes` sizes it for the
    launch, the spectra, the tables, the outputs -- allocated at exactly its size: no body reads or writes past an end."""
    exe = str(tmp_path / "fft_asan_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-pthread", "-I" + os.path.join(ROOT, "nvalchemi-toolkit-ops_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "fft_race_check.cpp"), "-o", exe])
    run = subprocess.run([exe, "asan"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "AddressSanitizer" not in run.stderr, run.stderr[-3000:]
    assert sum(l.endswith("in bounds") for l in run.stdout.splitlines()) == 6, run.stdout   # 3 meshes x (fp64, fp32)


def test_standalone_hipfft_reproducer_compiles(tmp_path):
    """tests/native/hipfft_repro.cpp is the stand-alone reproducer of the wrong-transform hipFFT plans: hipFFT and the HIP
    runtime only, no line of this repository.  Here: it still compiles against the installed hipFFT headers (it runs in
    tests/test_pme_gpu.py::test_hipfft_defect_reproduces_without_this_library)."""
    import shutil

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "hipfft_repro")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", os.path.join(ROOT, "tests", "native", "hipfft_repro.cpp"), "-o", exe, "-lhipfft"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]

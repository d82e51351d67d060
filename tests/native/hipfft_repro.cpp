// hipfft_repro.cpp -- stand-alone reproducer of the wrong-transform hipFFT plans that the library's plan cache guards against
// (csrc/fft.cpp).  NOTHING of this repository is linked or loaded: hipFFT + the HIP runtime only.
//
//   hipcc --offload-arch=gfx950 -O2 tests/native/hipfft_repro.cpp -o /tmp/hipfft_repro -lhipfft && /tmp/hipfft_repro
//   LD_LIBRARY_PATH=<torch>/lib /tmp/hipfft_repro        # the same binary against the hipFFT / rocFFT that ships inside the PyTorch wheel
//
// What it does: 3-D real-to-complex (and back) transforms of a batch of meshes through hipfftMakePlanMany -- exactly the call
// csrc/fft.cpp makes -- for a list of shapes built from the 1-D lengths 8, 16, 32, each checked against the transform evaluated from its
// definition on the host (long double accumulation).  Plans are kept alive like a plan cache keeps them.  Modes:
//   argv[1] = "fresh"   every shape is the first and only plan of its process (the program re-executes itself per shape)
//   argv[1] = "churn"   (default) one process, all plans alive, in the order the library's test-suite meets them
//   argv[1] = "seq"     seq <embeds> nx ny nz [nx ny nz ...]: the given shapes in the given order, one process, fp64 only (for narrowing down)
//   argv[2] = "null" | "explicit"   embeds passed as NULL or spelled out (default explicit)
// Exit code 0: every transform within 1e-9 (fp64) / 1e-4 (fp32) of the definition; 1: at least one plan computed something else -- the
// line of that plan says which.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <complex>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    auto _r = (x);                                                              \
    if (_r != 0) { fprintf(stderr, "%s failed: %d (line %d)\n", #x, (int)_r, __LINE__); exit(2); } \
  } while (0)

struct Shape { int nx, ny, nz, batch; };

static double rnd(unsigned long long& s) {  // splitmix64 -> (-1, 1)
  s += 0x9e3779b97f4a7c15ull;
  unsigned long long z = s;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z ^= z >> 31;
  return (double)(z >> 11) / 9007199254740992.0 * 2.0 - 1.0;
}

// the transform from its definition, separable: z (real -> half spectrum), then y, then x
static void host_rfftn(const std::vector<double>& in, std::vector<std::complex<double>>& out, const Shape& s) {
  const int nzr = s.nz / 2 + 1;
  const double pi2 = 6.283185307179586476925286766559;
  std::vector<std::complex<long double>> a((size_t)s.nx * s.ny * nzr), b(a.size());
  for (int x = 0; x < s.nx; ++x)
    for (int y = 0; y < s.ny; ++y)
      for (int k = 0; k < nzr; ++k) {
        std::complex<long double> acc = 0;
        for (int z = 0; z < s.nz; ++z) {
          const long double ph = -pi2 * (long double)((long long)k * z % s.nz) / s.nz;
          acc += (long double)in[((size_t)x * s.ny + y) * s.nz + z] * std::complex<long double>(cosl(ph), sinl(ph));
        }
        a[((size_t)x * s.ny + y) * nzr + k] = acc;
      }
  for (int x = 0; x < s.nx; ++x)
    for (int k = 0; k < s.ny; ++k)
      for (int z = 0; z < nzr; ++z) {
        std::complex<long double> acc = 0;
        for (int y = 0; y < s.ny; ++y) {
          const long double ph = -pi2 * (long double)((long long)k * y % s.ny) / s.ny;
          acc += a[((size_t)x * s.ny + y) * nzr + z] * std::complex<long double>(cosl(ph), sinl(ph));
        }
        b[((size_t)x * s.ny + k) * nzr + z] = acc;
      }
  out.resize(a.size());
  for (int k = 0; k < s.nx; ++k)
    for (int y = 0; y < s.ny; ++y)
      for (int z = 0; z < nzr; ++z) {
        std::complex<long double> acc = 0;
        for (int x = 0; x < s.nx; ++x) {
          const long double ph = -pi2 * (long double)((long long)k * x % s.nx) / s.nx;
          acc += b[((size_t)x * s.ny + y) * nzr + z] * std::complex<long double>(cosl(ph), sinl(ph));
        }
        out[((size_t)k * s.ny + y) * nzr + z] = std::complex<double>((double)acc.real(), (double)acc.imag());
      }
}

template <class R>
static int run_shape(const Shape& s, bool null_embeds, bool keep_alive, const char* tag) {
  const bool f64 = sizeof(R) == 8;
  const int nzr = s.nz / 2 + 1;
  const size_t nreal = (size_t)s.nx * s.ny * s.nz, nhalf = (size_t)s.nx * s.ny * nzr;
  unsigned long long seed = 1234 + 7 * s.nx + 13 * s.ny + 17 * s.nz + s.batch;
  std::vector<double> h((size_t)s.batch * nreal);
  for (auto& v : h) v = rnd(seed);
  std::vector<R> hr(h.begin(), h.end());
  R* d_in;
  void* d_out;
  R* d_back;
  CK(hipMalloc(&d_in, sizeof(R) * hr.size()));
  CK(hipMalloc(&d_out, 2 * sizeof(R) * s.batch * nhalf));
  CK(hipMalloc(&d_back, sizeof(R) * hr.size()));
  CK(hipMemcpy(d_in, hr.data(), sizeof(R) * hr.size(), hipMemcpyHostToDevice));
  int n[3] = {s.nx, s.ny, s.nz}, real_dims[3] = {s.nx, s.ny, s.nz}, half_dims[3] = {s.nx, s.ny, nzr};
  hipfftHandle fwd, inv;
  size_t ws = 0;
  CK(hipfftCreate(&fwd));
  CK(hipfftCreate(&inv));
  if (null_embeds) {
    CK(hipfftMakePlanMany(fwd, 3, n, nullptr, 1, 0, nullptr, 1, 0, f64 ? HIPFFT_D2Z : HIPFFT_R2C, s.batch, &ws));
    CK(hipfftMakePlanMany(inv, 3, n, nullptr, 1, 0, nullptr, 1, 0, f64 ? HIPFFT_Z2D : HIPFFT_C2R, s.batch, &ws));
  } else {
    CK(hipfftMakePlanMany(fwd, 3, n, real_dims, 1, (int)nreal, half_dims, 1, (int)nhalf, f64 ? HIPFFT_D2Z : HIPFFT_R2C, s.batch, &ws));
    CK(hipfftMakePlanMany(inv, 3, n, half_dims, 1, (int)nhalf, real_dims, 1, (int)nreal, f64 ? HIPFFT_Z2D : HIPFFT_C2R, s.batch, &ws));
  }
  if (f64) {
    CK(hipfftExecD2Z(fwd, (hipfftDoubleReal*)d_in, (hipfftDoubleComplex*)d_out));
  } else {
    CK(hipfftExecR2C(fwd, (hipfftReal*)d_in, (hipfftComplex*)d_out));
  }
  CK(hipDeviceSynchronize());
  std::vector<R> spec(2 * (size_t)s.batch * nhalf);
  CK(hipMemcpy(spec.data(), d_out, sizeof(R) * spec.size(), hipMemcpyDeviceToHost));
  if (f64) {
    CK(hipfftExecZ2D(inv, (hipfftDoubleComplex*)d_out, (hipfftDoubleReal*)d_back));
  } else {
    CK(hipfftExecC2R(inv, (hipfftComplex*)d_out, (hipfftReal*)d_back));
  }
  CK(hipDeviceSynchronize());
  std::vector<R> back(hr.size());
  CK(hipMemcpy(back.data(), d_back, sizeof(R) * back.size(), hipMemcpyDeviceToHost));
  double err_f = 0, err_b = 0, scale = 0;
  for (int b = 0; b < s.batch; ++b) {
    std::vector<double> one(h.begin() + (size_t)b * nreal, h.begin() + (size_t)(b + 1) * nreal);
    std::vector<std::complex<double>> want;
    host_rfftn(one, want, s);
    for (size_t k = 0; k < nhalf; ++k) {
      const double re = spec[2 * ((size_t)b * nhalf + k)], im = spec[2 * ((size_t)b * nhalf + k) + 1];
      err_f = fmax(err_f, hypot(re - want[k].real(), im - want[k].imag()));
      scale = fmax(scale, std::abs(want[k]));
    }
    for (size_t k = 0; k < nreal; ++k) err_b = fmax(err_b, fabs((double)back[(size_t)b * nreal + k] / (double)nreal - one[k]));
  }
  const double tol = f64 ? 1e-9 : 1e-4;
  const bool bad = !(err_f / scale < tol) || !(err_b < tol);
  printf("%-6s %s  (%3d,%3d,%3d) x %d  %s embeds   R2C rel err %.2e   R2C->C2R round trip %.2e   %s\n", tag, f64 ? "f64" : "f32", s.nx, s.ny, s.nz,
         s.batch, null_embeds ? "NULL    " : "explicit", err_f / scale, err_b, bad ? "WRONG" : "ok");
  fflush(stdout);
  CK(hipFree(d_in));
  CK(hipFree(d_out));
  CK(hipFree(d_back));
  if (!keep_alive) { CK(hipfftDestroy(fwd)); CK(hipfftDestroy(inv)); }
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "churn";
  const bool null_embeds = argc > 2 && strcmp(argv[2], "null") == 0;
  const Shape shapes[] = {{16, 16, 16, 1}, {8, 8, 8, 1},  {16, 8, 24, 1}, {12, 10, 14, 1}, {8, 64, 16, 1}, {32, 32, 32, 1}, {24, 16, 8, 1},
                          {32, 16, 8, 1},  {16, 8, 32, 1}, {8, 16, 32, 1}, {32, 8, 16, 1},  {16, 16, 16, 1}, {16, 16, 16, 4}, {16, 16, 16, 3},
                          {32, 8, 16, 2},  {16, 32, 8, 1}, {8, 32, 16, 1}};
  const int ns = (int)(sizeof(shapes) / sizeof(shapes[0]));
  int ver = 0;
  hipfftGetVersion(&ver);
  if (strcmp(mode, "one") == 0 && argc > 6) {  // child of "fresh": one shape, first plan of the process
    const Shape s{atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6])};
    return run_shape<double>(s, null_embeds, true, "fresh") | run_shape<float>(s, null_embeds, true, "fresh");
  }
  if (strcmp(mode, "seq") == 0) {  // seq <embeds> nx ny nz [nx ny nz ...]: these shapes in this order, one process, all plans alive (batch 1)
    int bad = 0;
    for (int k = 3; k + 2 < argc; k += 3) {
      const Shape s{atoi(argv[k]), atoi(argv[k + 1]), atoi(argv[k + 2]), 1};
      bad |= run_shape<double>(s, null_embeds, true, "seq");
    }
    return bad;
  }
  printf("hipFFT version %d, mode %s\n", ver, mode);
  fflush(stdout);
  int bad = 0;
  if (strcmp(mode, "fresh") == 0) {
    for (int k = 0; k < ns; ++k) {
      char cmd[512];
      snprintf(cmd, sizeof(cmd), "%s one %s %d %d %d %d", argv[0], null_embeds ? "null" : "explicit", shapes[k].nx, shapes[k].ny, shapes[k].nz, shapes[k].batch);
      bad |= system(cmd) != 0;
    }
  } else {
    for (int k = 0; k < ns; ++k) bad |= run_shape<double>(shapes[k], null_embeds, true, "churn") | run_shape<float>(shapes[k], null_embeds, true, "churn");
  }
  printf("%s\n", bad ? "RESULT: at least one hipFFT plan computed a transform other than its definition" : "RESULT: all plans correct");
  return bad ? 1 : 0;
}

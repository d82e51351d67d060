// dft.hip -- 3-D real <-> complex discrete Fourier transforms of ANY mesh size, evaluated from the definition, behind mi_dft3d.
//
// Why it exists (round 5): the reciprocal-space PME pipeline (reference: interactions/electrostatics/pme.py:1398 rfftn, :1422 / :1455-1457
// irfftn) needs FFTs for meshes the library's own in-LDS power-of-two transforms (fft_lds.h) do not cover.  Those went through hipFFT plans,
// and rocFFT on this stack can return a WRONG transform for some shapes depending on what else the process planned before -- (32, 8, 16) and
// (16, 8, 32) were seen 60 % / 5 % off numpy, deterministically, through hipFFT plans AND through torch.fft in the same process
// (profiles/r05_rocfft_drift_*.log).  A plan is therefore self-tested at creation (pme.py `_fft_plan`); when it fails, the
// transform is computed HERE: three passes of 1-D DFTs evaluated from the definition -- split once, n = n1 n2, into dense DFTs of length n1
// and n2 with a twiddle between them (n1 + n2 multiply-adds per output; a prime length is summed directly) -- twiddles from one sincospi
// table per block in double precision.  O(sqrt n) work per output instead of O(log n): 2 - 3 x a tuned FFT at PME's mesh sizes, but no plan,
// no library state, nothing that can be wrong in one process and right in the next.  NVALCHEMIOPS_PME_FFT=dft routes every mesh through
// it (the parity tests do).
//
// Layout (as mi_fft_plan_*): real [batch][nx][ny][nz], complex [batch][nx][ny][nz/2+1] interleaved; both directions UNSCALED
// (forward = rfftn norm="backward", inverse = irfftn norm="forward").  The inverse transforms its complex input in place (x and y passes)
// before the z pass writes the real output: the input is scratch, as it is for hipFFT's multi-dimensional C2R.
#include "common.h"

namespace {

template <class T> struct Cx { T re, im; };

// tw[t] = exp(sign * 2 pi i t / n), t in [0, n): evaluated in double whatever T is (sincospi: exact at the multiples of 1/2)
template <class T> __device__ __forceinline__ void dft_table(Cx<T>* tw, int n, int sign) {
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    double s, c;
    sincospi(2.0 * (double)t / (double)n, &s, &c);
    tw[t] = Cx<T>{(T)c, (T)(sign < 0 ? -s : s)};
  }
}

// ---- the transform of one tile, in LDS ---------------------------------------------------------------------------------------------------
// A tile is n x ti complex values, element (j, ii) at buf[j * ti + ii]: ti independent length-n transforms side by side.
// n = n1 * n2 (n1 = the largest divisor <= sqrt(n); 1 for a prime): with j = j1 n2 + j2 and k = k1 + n1 k2
//     X[k1 + n1 k2] = sum_j2 w^(n1 j2 k2) * [ w^(j2 k1) * sum_j1 x[j1 n2 + j2] w^(n2 j1 k1) ],        w = exp(sign 2 pi i / n),
// i.e. n2 dense DFTs of length n1, a twiddle, n1 dense DFTs of length n2: n1 + n2 multiply-adds per output instead of n (16 + 8 for 128,
// 10 + 10 for 100, 7 + 5 for 35), every factor from the ONE table tw[t] = w^t with incremental indices (no modulo in the loops).  Not an
// FFT -- two levels, any composite n, primes fall back to the definition -- but within a small factor of one for the mesh sizes PME uses.
struct DftSplit { int n1, n2; };
inline DftSplit dft_split(int n) {
  int best = 1;
  for (int d = 1; (long long)d * d <= n; ++d) if (n % d == 0) best = d;
  return DftSplit{best, n / best};
}

// stage 1: a -> b, b[(j2 * n1 + k1) * ti + ii] = w^(j2 k1) sum_j1 a[(j1 n2 + j2) * ti + ii] w^(n2 j1 k1)
template <class T>
__device__ __forceinline__ void dft_stage1(const Cx<T>* __restrict__ a, Cx<T>* __restrict__ b, const Cx<T>* __restrict__ tw, int n, int n1, int n2, int ti) {
  const int total = n * ti;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int m = idx / ti, ii = idx - m * ti;
    const int j2 = m / n1, k1 = m - j2 * n1;
    int step = (int)(((long long)n2 * k1) % n), t = 0;
    T re = 0, im = 0;
    const Cx<T>* x = a + j2 * ti + ii;
    for (int j1 = 0; j1 < n1; ++j1) {
      const Cx<T> v = x[(size_t)j1 * n2 * ti], w = tw[t];
      re += v.re * w.re - v.im * w.im;
      im += v.re * w.im + v.im * w.re;
      t += step;
      if (t >= n) t -= n;
    }
    const Cx<T> w = tw[(int)(((long long)j2 * k1) % n)];
    b[idx] = Cx<T>{re * w.re - im * w.im, re * w.im + im * w.re};  // idx == (j2 * n1 + k1) * ti + ii
  }
}
// stage 2 for output slot idx -> (k, ii): X[k1 + n1 k2] = sum_j2 b[(j2 n1 + k1) * ti + ii] w^(n1 j2 k2)
template <class T>
__device__ __forceinline__ Cx<T> dft_stage2(const Cx<T>* __restrict__ b, const Cx<T>* __restrict__ tw, int n, int n1, int n2, int ti, int k, int ii) {
  const int k2 = k / n1, k1 = k - k2 * n1;
  const int step = (int)(((long long)n1 * k2) % n);
  int t = 0;
  T re = 0, im = 0;
  const Cx<T>* x = b + k1 * ti + ii;
  for (int j2 = 0; j2 < n2; ++j2) {
    const Cx<T> v = x[(size_t)j2 * n1 * ti], w = tw[t];
    re += v.re * w.re - v.im * w.im;
    im += v.re * w.im + v.im * w.re;
    t += step;
    if (t >= n) t -= n;
  }
  return Cx<T>{re, im};
}

// C2C along an axis of stride `inner` (complex elements), in place: data viewed as [outer][n][inner]; a block owns `ti` consecutive inner
// positions of one `outer` slab
template <class T>
__global__ __launch_bounds__(256) void dft_axis_kernel(Cx<T>* __restrict__ data, int n, int n1, int n2, long long inner, int ti, long long tiles_per_slab, int sign) {
  extern __shared__ __align__(16) unsigned char dft_lds[];
  Cx<T>* tw = reinterpret_cast<Cx<T>*>(dft_lds);
  Cx<T>* a = tw + n;
  Cx<T>* b = a + (size_t)n * ti;
  const long long slab = blockIdx.x / tiles_per_slab, it = blockIdx.x - slab * tiles_per_slab;
  const long long i0 = it * ti;
  Cx<T>* base = data + slab * n * inner + i0;
  const int width = (inner - i0 < ti) ? (int)(inner - i0) : ti;
  dft_table(tw, n, sign);
  const int total = n * ti;
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int j = idx / ti, ii = idx - j * ti;
    a[idx] = ii < width ? base[(long long)j * inner + ii] : Cx<T>{T(0), T(0)};
  }
  __syncthreads();
  dft_stage1(a, b, tw, n, n1, n2, ti);
  __syncthreads();
  // in place: every value of the tile was read into LDS before the first write-back, and no other block touches these columns
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int k = idx / ti, ii = idx - k * ti;
    if (ii < width) base[(long long)k * inner + ii] = dft_stage2(b, tw, n, n1, n2, ti, k, ii);
  }
}

// R2C along z: in [lines][nz] real -> out [lines][nz/2+1] complex; a block owns `ti` consecutive lines (tile element (z, line))
template <class T>
__global__ __launch_bounds__(256) void dft_r2c_z_kernel(const T* __restrict__ in, Cx<T>* __restrict__ out, int nz, int n1, int n2, long long lines, int ti) {
  extern __shared__ __align__(16) unsigned char dft_lds[];
  Cx<T>* tw = reinterpret_cast<Cx<T>*>(dft_lds);
  Cx<T>* a = tw + nz;
  Cx<T>* b = a + (size_t)nz * ti;
  const int nzr = nz / 2 + 1;
  const long long l0 = (long long)blockIdx.x * ti;
  const int nl = (lines - l0 < ti) ? (int)(lines - l0) : ti;
  dft_table(tw, nz, -1);
  for (int idx = threadIdx.x; idx < nz * ti; idx += 256) {  // coalesced along z, transposed into the tile
    const int l = idx / nz, z = idx - l * nz;
    a[z * ti + l] = Cx<T>{l < nl ? in[(l0 + l) * nz + z] : T(0), T(0)};
  }
  __syncthreads();
  dft_stage1(a, b, tw, nz, n1, n2, ti);
  __syncthreads();
  for (int idx = threadIdx.x; idx < nzr * ti; idx += 256) {
    const int l = idx / nzr, k = idx - l * nzr;
    if (l < nl) out[(l0 + l) * nzr + k] = dft_stage2(b, tw, nz, n1, n2, ti, k, l);
  }
}

// C2R along z: in [lines][nz/2+1] complex (Hermitian half) -> out [lines][nz] real, unscaled.  The half spectrum is completed in LDS
// (conjugate mirror; the imaginary parts of the DC and Nyquist terms dropped, as every C2R transform does) and transformed as complex data.
template <class T>
__global__ __launch_bounds__(256) void dft_c2r_z_kernel(const Cx<T>* __restrict__ in, T* __restrict__ out, int nz, int n1, int n2, long long lines, int ti) {
  extern __shared__ __align__(16) unsigned char dft_lds[];
  Cx<T>* tw = reinterpret_cast<Cx<T>*>(dft_lds);
  Cx<T>* a = tw + nz;
  Cx<T>* b = a + (size_t)nz * ti;
  const int nzr = nz / 2 + 1;
  const long long l0 = (long long)blockIdx.x * ti;
  const int nl = (lines - l0 < ti) ? (int)(lines - l0) : ti;
  dft_table(tw, nz, +1);
  for (int idx = threadIdx.x; idx < nzr * ti; idx += 256) {
    const int l = idx / nzr, k = idx - l * nzr;
    Cx<T> v = l < nl ? in[(l0 + l) * nzr + k] : Cx<T>{T(0), T(0)};
    if (k == 0 || 2 * k == nz) v.im = T(0);
    a[k * ti + l] = v;
    if (k != 0 && 2 * k != nz) a[(nz - k) * ti + l] = Cx<T>{v.re, -v.im};
  }
  __syncthreads();
  dft_stage1(a, b, tw, nz, n1, n2, ti);
  __syncthreads();
  for (int idx = threadIdx.x; idx < nz * ti; idx += 256) {
    const int l = idx / nz, z = idx - l * nz;
    if (l < nl) out[(l0 + l) * nz + z] = dft_stage2(b, tw, nz, n1, n2, ti, z, l).re;
  }
}

template <class T> inline int dft_tile(int n) {  // largest power of two <= 32 with two n x tile buffers (+ the table) inside 64 KB of LDS
  const long long cap = sizeof(T) == 8 ? 1024 : 2048;
  int t = 32;
  while (t > 1 && (long long)n * t > cap) t >>= 1;
  return t;
}
template <class T> inline size_t dft_lds_bytes(int n, int ti) { return sizeof(Cx<T>) * ((size_t)n + 2 * (size_t)n * ti); }

template <class T>
int dft_axis(Cx<T>* data, long long outer, int n, long long inner, int sign, hipStream_t st) {
  if (n == 1) return MI_OK;
  const int ti = dft_tile<T>(n);
  const DftSplit sp = dft_split(n);
  const long long tiles = (inner + ti - 1) / ti, blocks = outer * tiles;
  MI_REQUIRE(blocks < (1ll << 31), "mesh too large for one launch");
  dft_axis_kernel<T><<<(unsigned)blocks, 256, dft_lds_bytes<T>(n, ti), st>>>(data, n, sp.n1, sp.n2, inner, ti, tiles, sign);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

template <class T>
int dft3d_impl(const void* in, void* out, int nx, int ny, int nz, int batch, int inverse, hipStream_t st) {
  const int nzr = nz / 2 + 1;
  const long long lines = (long long)batch * nx * ny;
  const int tl = dft_tile<T>(nz);
  const DftSplit sz = dft_split(nz);
  const long long zblocks = (lines + tl - 1) / tl;
  MI_REQUIRE(zblocks < (1ll << 31), "mesh too large for one launch");
  int rc;
  if (!inverse) {
    MI_TIMED("dft_r2c", st, (dft_r2c_z_kernel<T><<<(unsigned)zblocks, 256, dft_lds_bytes<T>(nz, tl), st>>>((const T*)in, (Cx<T>*)out, nz, sz.n1, sz.n2, lines, tl)));
    MI_LAUNCH_CHECK();
    if ((rc = dft_axis<T>((Cx<T>*)out, (long long)batch * nx, ny, nzr, -1, st)) != MI_OK) return rc;
    return dft_axis<T>((Cx<T>*)out, batch, nx, (long long)ny * nzr, -1, st);
  }
  Cx<T>* spec = (Cx<T>*)const_cast<void*>(in);
  if ((rc = dft_axis<T>(spec, batch, nx, (long long)ny * nzr, +1, st)) != MI_OK) return rc;
  if ((rc = dft_axis<T>(spec, (long long)batch * nx, ny, nzr, +1, st)) != MI_OK) return rc;
  MI_TIMED("dft_c2r", st, (dft_c2r_z_kernel<T><<<(unsigned)zblocks, 256, dft_lds_bytes<T>(nz, tl), st>>>(spec, (T*)out, nz, sz.n1, sz.n2, lines, tl)));
  MI_LAUNCH_CHECK();
  return MI_OK;
}

}  // namespace

extern "C" {

int mi_dft3d(void* in, void* out, int nx, int ny, int nz, int batch, int dtype, int inverse, void* stream) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype must be MI_F32 or MI_F64");
  MI_REQUIRE(nx >= 1 && ny >= 1 && nz >= 1 && batch >= 1 && nx <= MI_DFT_MAX_N && ny <= MI_DFT_MAX_N && nz <= MI_DFT_MAX_N,
             "mesh dimensions in [1, MI_DFT_MAX_N], batch >= 1");
  MI_REQUIRE(in && out, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_F32) return dft3d_impl<float>(in, out, nx, ny, nz, batch, inverse, st);
  return dft3d_impl<double>(in, out, nx, ny, nz, batch, inverse, st);
}

}  // extern "C"

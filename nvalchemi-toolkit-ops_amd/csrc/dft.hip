// dft.hip -- 3-D real <-> complex discrete Fourier transforms of ANY mesh size, evaluated from the definition, behind mi_dft3d.
//
// Why it exists (round 5): the reciprocal-space PME pipeline (reference: interactions/electrostatics/pme.py:1398 rfftn, :1422 / :1455-1457
// irfftn) needs FFTs for meshes the library's own in-LDS power-of-two transforms (fft_lds.h) do not cover.  Those went through hipFFT plans,
// and rocFFT on this stack can return a WRONG transform for some shapes depending on what else the process planned before -- (32, 8, 16) and
// (16, 8, 32) were seen 60 % / 5 % off numpy, deterministically, through hipFFT plans AND through torch.fft in the same process
// (profiles/r05_rocfft_drift_*.log).  A plan is therefore self-tested at creation (pme.py `_fft_plan`); when it fails, the
// transform is computed HERE: three passes of dense 1-D DFTs, X_k = sum_j x_j w^(jk), twiddles from one sincospi table per block in
// double precision.  O(n) work per output instead of O(log n) -- a 128^3 x 4 fp64 inverse is ~1.3e10 flops, a millisecond; the odd little
// meshes that actually end up here cost microseconds -- but no plan, no library state, nothing that can be wrong in one process and right
// in the next.  NVALCHEMIOPS_PME_FFT=dft routes every mesh through it (the parity tests do).
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

#define DFT_MAX_OUT 8  // outputs per thread: n * tile <= 256 * DFT_MAX_OUT

// C2C along an axis of stride `inner` (complex elements), in place: data viewed as [outer][n][inner]; a block owns `ti` consecutive inner
// positions of one `outer` slab: n x ti values staged in LDS, every thread produces up to DFT_MAX_OUT outputs (k, ii) with ii fixed.
template <class T>
__global__ __launch_bounds__(256) void dft_axis_kernel(Cx<T>* __restrict__ data, int n, long long inner, int ti, long long tiles_per_slab, int sign) {
  extern __shared__ __align__(16) unsigned char dft_lds[];
  Cx<T>* tw = reinterpret_cast<Cx<T>*>(dft_lds);
  Cx<T>* tile = tw + n;
  const long long slab = blockIdx.x / tiles_per_slab, it = blockIdx.x - slab * tiles_per_slab;
  const long long i0 = it * ti;
  Cx<T>* base = data + slab * n * inner + i0;
  const int width = (inner - i0 < ti) ? (int)(inner - i0) : ti;
  dft_table(tw, n, sign);
  const int total = n * ti;
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int j = idx / ti, ii = idx - j * ti;
    tile[idx] = ii < width ? base[(long long)j * inner + ii] : Cx<T>{T(0), T(0)};
  }
  __syncthreads();
  Cx<T> acc[DFT_MAX_OUT];
  int kk[DFT_MAX_OUT], tix[DFT_MAX_OUT];
  const int ii = threadIdx.x % ti;  // 256 % ti == 0: the same inner position for all of this thread's outputs
#pragma unroll
  for (int r = 0; r < DFT_MAX_OUT; ++r) { acc[r] = Cx<T>{T(0), T(0)}; kk[r] = (threadIdx.x + r * 256) / ti; tix[r] = 0; }
  for (int j = 0; j < n; ++j) {
    const Cx<T> x = tile[j * ti + ii];
#pragma unroll
    for (int r = 0; r < DFT_MAX_OUT; ++r) {
      if (kk[r] < n) {  // (thread-uniform per r beyond the tail)
        const Cx<T> w = tw[tix[r]];
        acc[r].re += x.re * w.re - x.im * w.im;
        acc[r].im += x.re * w.im + x.im * w.re;
        tix[r] += kk[r];
        if (tix[r] >= n) tix[r] -= n;
      }
    }
  }
  // in place: every value of the tile has been read into LDS before the first write-back (the barrier above), and no other block touches it
  if (ii < width) {
#pragma unroll
    for (int r = 0; r < DFT_MAX_OUT; ++r)
      if (kk[r] < n) base[(long long)kk[r] * inner + ii] = acc[r];
  }
}

// R2C along z: in [lines][nz] real -> out [lines][nz/2+1] complex; a block owns `tl` consecutive lines
template <class T>
__global__ __launch_bounds__(256) void dft_r2c_z_kernel(const T* __restrict__ in, Cx<T>* __restrict__ out, int nz, long long lines, int tl) {
  extern __shared__ __align__(16) unsigned char dft_lds[];
  Cx<T>* tw = reinterpret_cast<Cx<T>*>(dft_lds);
  T* tile = reinterpret_cast<T*>(tw + nz);
  const int nzr = nz / 2 + 1;
  const long long l0 = (long long)blockIdx.x * tl;
  const int nl = (lines - l0 < tl) ? (int)(lines - l0) : tl;
  dft_table(tw, nz, -1);
  for (int idx = threadIdx.x; idx < nl * nz; idx += 256) tile[idx] = in[l0 * nz + idx];
  __syncthreads();
  for (int idx = threadIdx.x; idx < nl * nzr; idx += 256) {
    const int l = idx / nzr, k = idx - l * nzr;
    const T* x = tile + l * nz;
    T re = 0, im = 0;
    int t = 0;
    for (int z = 0; z < nz; ++z) {
      const Cx<T> w = tw[t];
      re += x[z] * w.re;
      im += x[z] * w.im;
      t += k;
      if (t >= nz) t -= nz;
    }
    out[(l0 + l) * nzr + k] = Cx<T>{re, im};
  }
}

// C2R along z: in [lines][nz/2+1] complex (Hermitian half) -> out [lines][nz] real, unscaled; imaginary parts of the DC and Nyquist terms
// do not enter (their twiddles are exactly real), as in every C2R transform
template <class T>
__global__ __launch_bounds__(256) void dft_c2r_z_kernel(const Cx<T>* __restrict__ in, T* __restrict__ out, int nz, long long lines, int tl) {
  extern __shared__ __align__(16) unsigned char dft_lds[];
  Cx<T>* tw = reinterpret_cast<Cx<T>*>(dft_lds);
  Cx<T>* tile = tw + nz;
  const int nzr = nz / 2 + 1;
  const long long l0 = (long long)blockIdx.x * tl;
  const int nl = (lines - l0 < tl) ? (int)(lines - l0) : tl;
  dft_table(tw, nz, +1);
  for (int idx = threadIdx.x; idx < nl * nzr; idx += 256) {
    const int k = idx % nzr;
    Cx<T> v = in[l0 * nzr + idx];
    const T c = (k == 0 || 2 * k == nz) ? T(1) : T(2);  // the mirrored half of the spectrum
    tile[idx] = Cx<T>{c * v.re, c * v.im};
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < nl * nz; idx += 256) {
    const int l = idx / nz, z = idx - l * nz;
    const Cx<T>* x = tile + l * nzr;
    T acc = 0;
    int t = 0;
    for (int k = 0; k < nzr; ++k) {
      const Cx<T> w = tw[t];
      acc += x[k].re * w.re - x[k].im * w.im;
      t += z;
      if (t >= nz) t -= nz;
    }
    out[l0 * nz + idx] = acc;
  }
}

inline int dft_tile(int n) {  // largest power of two <= 32 with n * tile <= 256 * DFT_MAX_OUT
  int t = 32;
  while (t > 1 && (long long)n * t > 256ll * DFT_MAX_OUT) t >>= 1;
  return t;
}

template <class T>
int dft_axis(Cx<T>* data, long long outer, int n, long long inner, int sign, hipStream_t st) {
  if (n == 1) return MI_OK;
  const int ti = dft_tile(n);
  const long long tiles = (inner + ti - 1) / ti, blocks = outer * tiles;
  MI_REQUIRE(blocks < (1ll << 31), "mesh too large for one launch");
  const size_t lds = sizeof(Cx<T>) * ((size_t)n + (size_t)n * ti);
  dft_axis_kernel<T><<<(unsigned)blocks, 256, lds, st>>>(data, n, inner, ti, tiles, sign);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

template <class T>
int dft3d_impl(const void* in, void* out, int nx, int ny, int nz, int batch, int inverse, hipStream_t st) {
  const int nzr = nz / 2 + 1;
  const long long lines = (long long)batch * nx * ny;
  const int tl = dft_tile(nz);
  const long long zblocks = (lines + tl - 1) / tl;
  MI_REQUIRE(zblocks < (1ll << 31), "mesh too large for one launch");
  int rc;
  if (!inverse) {
    const size_t lds = sizeof(Cx<T>) * (size_t)nz + sizeof(T) * (size_t)nz * tl;
    MI_TIMED("dft_r2c", st, (dft_r2c_z_kernel<T><<<(unsigned)zblocks, 256, lds, st>>>((const T*)in, (Cx<T>*)out, nz, lines, tl)));
    MI_LAUNCH_CHECK();
    if ((rc = dft_axis<T>((Cx<T>*)out, (long long)batch * nx, ny, nzr, -1, st)) != MI_OK) return rc;
    return dft_axis<T>((Cx<T>*)out, batch, nx, (long long)ny * nzr, -1, st);
  }
  Cx<T>* spec = (Cx<T>*)const_cast<void*>(in);
  if ((rc = dft_axis<T>(spec, batch, nx, (long long)ny * nzr, +1, st)) != MI_OK) return rc;
  if ((rc = dft_axis<T>(spec, (long long)batch * nx, ny, nzr, +1, st)) != MI_OK) return rc;
  const size_t lds = sizeof(Cx<T>) * ((size_t)nz + (size_t)nzr * tl);
  MI_TIMED("dft_c2r", st, (dft_c2r_z_kernel<T><<<(unsigned)zblocks, 256, lds, st>>>(spec, (T*)out, nz, lines, tl)));
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

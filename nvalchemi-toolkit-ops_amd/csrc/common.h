// common.h -- shared host/device helpers for libnvalchemiops_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nvalchemiops_hip.h"

#define MI_WAVE 64

void mi_set_error(const char* fmt, ...);
void mi_timing_begin(const char* name, void* stream);
void mi_timing_end(void* stream);
// brackets a kernel launch with HIP events when timing is enabled (mi_timing_enable); free otherwise
#define MI_TIMED(name, st, ...)       \
  do {                                \
    mi_timing_begin(name, (void*)st); \
    __VA_ARGS__;                      \
    mi_timing_end((void*)st);         \
  } while (0)

#define MI_HIP_CHECK(expr)                                                                     \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      mi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return MI_EHIP;                                                                          \
    }                                                                                          \
  } while (0)

#define MI_LAUNCH_CHECK() MI_HIP_CHECK(hipGetLastError())

#define MI_REQUIRE(cond, msg)                    \
  do {                                           \
    if (!(cond)) {                               \
      mi_set_error("invalid argument: %s", msg); \
      return MI_EINVAL;                          \
    }                                            \
  } while (0)

static inline size_t mi_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int mi_blocks(long long n, int per_block) { return (int)((n + per_block - 1) / per_block); }

// ---- device helpers -----------------------------------------------------------------------------
template <class T> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<double> { using type = double4; };

__device__ __forceinline__ int lane_id() { return threadIdx.x & (MI_WAVE - 1); }
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// floor division / modulo on int32 (reference: math/math.py:41-50)
__host__ __device__ __forceinline__ void floor_divmod(int a, int b, int& q, int& r) {
  q = a / b;
  r = a - q * b;
  if (r < 0) { q -= 1; r += b; }
}

// wave-wide sum via cross-lane shuffles (all 64 lanes must be active)
template <class T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = MI_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, MI_WAVE);
  return v;
}

// row-vector x 3x3 (row-major) with the reference's summation order: r = row0*v0; r += row1*v1; r += row2*v2
template <class T> __host__ __device__ __forceinline__ void rowvec_mat3(const T v[3], const T* m, T out[3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    T r = m[c] * v[0];
    r = r + m[3 + c] * v[1];
    r = r + m[6 + c] * v[2];
    out[c] = r;
  }
}
// 3x3 (row-major) x column vector: r = col0*v0; r += col1*v1; r += col2*v2
template <class T> __host__ __device__ __forceinline__ void mat3_colvec(const T* m, const T v[3], T out[3]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    T s = m[3 * r] * v[0];
    s = s + m[3 * r + 1] * v[1];
    s = s + m[3 * r + 2] * v[2];
    out[r] = s;
  }
}
template <class T> __host__ __device__ inline void inverse3(const T* a, T* b) {
  b[0] = a[4] * a[8] - a[5] * a[7];
  b[1] = a[2] * a[7] - a[1] * a[8];
  b[2] = a[1] * a[5] - a[2] * a[4];
  b[3] = a[5] * a[6] - a[3] * a[8];
  b[4] = a[0] * a[8] - a[2] * a[6];
  b[5] = a[2] * a[3] - a[0] * a[5];
  b[6] = a[3] * a[7] - a[4] * a[6];
  b[7] = a[1] * a[6] - a[0] * a[7];
  b[8] = a[0] * a[4] - a[1] * a[3];
  T det = a[0] * b[0] + a[1] * b[3] + a[2] * b[6];
  T s = T(1) / det;
  for (int i = 0; i < 9; ++i) b[i] *= s;
}

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
// ---- coordination-number block of a neighbour search (round 6; csrc/nlist.hip writes it, csrc/d3.hip adopts it) ------------------------
// Layout of the buffer (mi_nl_cn_bytes): a 1 KiB header, then float cn[n_atoms] in the caller's atom order.
//   int32  [0]        raised (non-zero) when cn[] must not be used (a row overflowed its slots)
//   float  [2], [3]   the search's cutoff; k1 * log2(e) the terms were evaluated with
//   uint64 [32 .. 95] checksum slots (byte offset 256): the wrapping sum over all slots fingerprints what the numbers were computed FROM --
//                     every atom's index, system, position bits and scaled covalent radius, the cell entries and k1 -- so the consumer can
//                     verify, on the device, that it is being asked about the very same atoms (no reliance on host-side bookkeeping).
#define MI_CN_HEADER_BYTES 1024
#define MI_CN_SLOTS 64
#define MI_CN_SLOT_OFFSET_U64 32
__host__ __device__ __forceinline__ float mi_cn_scale(float k1) { return k1 * 1.44269504f; }  // exp(-k1 (rr - 1)) = 2^(K - K rr), K = k1 log2(e)
__host__ __device__ __forceinline__ unsigned long long mi_mix64(unsigned long long x) {  // splitmix64 finaliser
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27; x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return x;
}
__device__ __forceinline__ unsigned long long mi_bits64(float v) { return (unsigned long long)(unsigned)__float_as_int(v); }
__device__ __forceinline__ unsigned long long mi_bits64(double v) { return (unsigned long long)__double_as_longlong(v); }
// scaled covalent radius of an atom as the CN terms use it: rcov[Z] * K, -inf for an atom outside the tables (every term with it is 0)
__device__ __forceinline__ float mi_cn_rk(int z, int nz, const float* __restrict__ rcov, float K) {
  return (z > 0 && z < nz) ? rcov[z] * K : -INFINITY;
}
template <class T> __device__ __forceinline__ unsigned long long mi_cn_atom_hash(int i, int sys, T x, T y, T z, float rk) {
  unsigned long long h = mi_mix64(((unsigned long long)(unsigned)i << 32 | (unsigned)sys) + 0x9e3779b97f4a7c15ull);
  h = mi_mix64(h ^ mi_bits64(x));
  h = mi_mix64(h ^ mi_bits64(y));
  h = mi_mix64(h ^ mi_bits64(z));
  return mi_mix64(h ^ mi_bits64(rk));
}
// terms that do not belong to an atom: cell entry k of the [n_systems,3,3] array, and the scale K (thread 0)
template <class T> __device__ __forceinline__ unsigned long long mi_cn_cell_hash(int k, T v) { return mi_mix64(mi_mix64(0xce11ull + (unsigned long long)k) ^ mi_bits64(v)); }
__device__ __forceinline__ unsigned long long mi_cn_scale_hash(float K) { return mi_mix64(0x4b31ull ^ mi_bits64(K)); }
// adds every lane's value into slots[slot] with one atomic per wave (inactive lanes pass 0; all 64 lanes must call)
__device__ __forceinline__ void mi_cn_slot_add(unsigned long long* slots, int slot, unsigned long long v) {
#pragma unroll
  for (int o = MI_WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, MI_WAVE);
  if ((threadIdx.x & (MI_WAVE - 1)) == 0 && v != 0) atomicAdd(&slots[slot & (MI_CN_SLOTS - 1)], v);
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

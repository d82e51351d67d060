// Tuning probe (not part of the library): variants of the wave64-per-atom CSR row walk, to see what bounds d3_cn / d3_chain.
#include <hip/hip_runtime.h>
#include <stdint.h>
struct Int3 { int x, y, z; };
__device__ __forceinline__ float wsum(float v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64); return v; }

template <int V>
__global__ __launch_bounds__(256) void walk(const int* __restrict__ idx, const Int3* __restrict__ sh, const int* __restrict__ nptr,
                                            const float4* __restrict__ apos, int N, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const long long beg = nptr[i], end = nptr[i + 1];
  const float4 pi = apos[i];
  float acc = 0.f;
  for (long long e = beg + lane; e < end; e += 64) {
    const int j = idx[e];
    float v = (float)j;
    if (V >= 1) { const Int3 s = sh[e]; v += (float)(s.x + s.y + s.z); 
      if (V >= 2) { const float4 pj = apos[j];
        if (V == 2) v += pj.x;
        if (V >= 3) { const float dx = pj.x - pi.x + 40.f * s.x, dy = pj.y - pi.y + 40.f * s.y, dz = pj.z - pi.z + 40.f * s.z;
          const float r2 = dx * dx + dy * dy + dz * dz; const float rinv = rsqrtf(r2);
          v = 1.0f / (1.0f + __expf(-16.0f * ((pi.w + pj.w) * rinv - 1.0f))); } } }
    acc += v;
  }
  acc = wsum(acc);
  if (lane == 0) out[i] = acc;
}
// variant 10: unrolled x2 with independent loads issued first
__global__ __launch_bounds__(256) void walk_u2(const int* __restrict__ idx, const Int3* __restrict__ sh, const int* __restrict__ nptr,
                                               const float4* __restrict__ apos, int N, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const long long beg = nptr[i], end = nptr[i + 1];
  const float4 pi = apos[i];
  float acc = 0.f;
  for (long long e = beg + lane; e < end; e += 256) {
    int j[4]; Int3 s[4]; bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const long long ee = e + 64 * u; ok[u] = ee < end; j[u] = ok[u] ? idx[ee] : i; s[u] = ok[u] ? sh[ee] : Int3{0, 0, 0}; }
    float4 pj[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pj[u] = apos[j[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float dx = pj[u].x - pi.x + 40.f * s[u].x, dy = pj[u].y - pi.y + 40.f * s[u].y, dz = pj[u].z - pi.z + 40.f * s[u].z;
      const float r2 = dx * dx + dy * dy + dz * dz; const float rinv = rsqrtf(r2);
      const float v = 1.0f / (1.0f + __expf(-16.0f * ((pi.w + pj[u].w) * rinv - 1.0f)));
      acc += ok[u] ? v : 0.f;
    }
  }
  acc = wsum(acc);
  if (lane == 0) out[i] = acc;
}
// variants 30-32: cost model of the gather.  30: ideal locality (contiguous j), float4; 31: real j, 4-byte gather; 32: real j, 8-byte
template <int V>
__global__ __launch_bounds__(256) void gath(const int* __restrict__ idx, const Int3* __restrict__ sh, const int* __restrict__ nptr,
                                            const float4* __restrict__ apos, int N, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const long long beg = nptr[i], end = nptr[i + 1];
  float acc = 0.f;
  for (long long e = beg + lane; e < end; e += 64) {
    int j = idx[e];
    const Int3 s = sh[e];
    float v = (float)(s.x + s.y + s.z);
    if (V == 30) { j = (int)((e - beg + i) % N); v += apos[j].x + apos[j].w; }
    if (V == 31) v += reinterpret_cast<const float*>(apos)[j];
    if (V == 32) { const float2 t = reinterpret_cast<const float2*>(apos)[j]; v += t.x + t.y; }
    if (V == 33 || V == 34) { const float4 t = apos[j]; v += t.x + t.w; }
    if (V == 34) __syncthreads();
    acc += v;
  }
  acc = wsum(acc);
  if (lane == 0) out[i] = acc;
}
// variant 40: the cn walk with the four waves of a block kept in lock-step (one barrier per trip, common trip count)
__global__ __launch_bounds__(256) void walk_lockstep(const int* __restrict__ idx, const Int3* __restrict__ sh, const int* __restrict__ nptr,
                                                     const float4* __restrict__ apos, int N, float* __restrict__ out) {
  __shared__ int trips[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i0 = blockIdx.x * 4 + w;
  const int i = i0 < N ? i0 : N - 1;
  const long long beg = nptr[i], end = i0 < N ? nptr[i + 1] : beg;
  if (lane == 0) trips[w] = (int)((end - beg + 63) / 64);
  __syncthreads();
  const int T = max(max(trips[0], trips[1]), max(trips[2], trips[3]));
  const float4 pi = apos[i];
  float acc = 0.f;
  long long e = beg + lane;
  for (int t = 0; t < T; ++t, e += 64) {
    if (e < end) {
      const int j = idx[e];
      const Int3 s = sh[e];
      const float4 pj = apos[j];
      const float dx = pj.x - pi.x + 40.f * s.x, dy = pj.y - pi.y + 40.f * s.y, dz = pj.z - pi.z + 40.f * s.z;
      const float r2 = dx * dx + dy * dy + dz * dz; const float rinv = rsqrtf(r2);
      acc += 1.0f / (1.0f + __expf(-16.0f * ((pi.w + pj.w) * rinv - 1.0f)));
    }
    __syncthreads();
  }
  acc = wsum(acc);
  if (lane == 0 && i0 < N) out[i0] = acc;
}
// variants 41/42: W waves per block the cn walk with the four waves of a block kept in lock-step (one barrier per trip, common trip count)
template <int W>
__global__ __launch_bounds__(W * 64) void walk_lockstep_w(const int* __restrict__ idx, const Int3* __restrict__ sh, const int* __restrict__ nptr,
                                                     const float4* __restrict__ apos, int N, float* __restrict__ out) {
  __shared__ int trips[W];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i0 = blockIdx.x * W + w;
  const int i = i0 < N ? i0 : N - 1;
  const long long beg = nptr[i], end = i0 < N ? nptr[i + 1] : beg;
  if (lane == 0) trips[w] = (int)((end - beg + 63) / 64);
  __syncthreads();
  int T = 0;
  for (int k = 0; k < W; ++k) T = max(T, trips[k]);
  const float4 pi = apos[i];
  float acc = 0.f;
  long long e = beg + lane;
  for (int t = 0; t < T; ++t, e += 64) {
    if (e < end) {
      const int j = idx[e];
      const Int3 s = sh[e];
      const float4 pj = apos[j];
      const float dx = pj.x - pi.x + 40.f * s.x, dy = pj.y - pi.y + 40.f * s.y, dz = pj.z - pi.z + 40.f * s.z;
      const float r2 = dx * dx + dy * dy + dz * dz; const float rinv = rsqrtf(r2);
      acc += 1.0f / (1.0f + __expf(-16.0f * ((pi.w + pj.w) * rinv - 1.0f)));
    }
    __syncthreads();
  }
  acc = wsum(acc);
  if (lane == 0 && i0 < N) out[i0] = acc;
}
// variant 20: flat streaming of the arrays (no rows): upper bound for idx+shift streaming
__global__ __launch_bounds__(256) void flat(const int* __restrict__ idx, const Int3* __restrict__ sh, long long P, float* __restrict__ out) {
  float acc = 0.f;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < P; e += (long long)gridDim.x * 256) { const Int3 s = sh[e]; acc += (float)(idx[e] + s.x + s.y + s.z); }
  acc = wsum(acc);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
extern "C" int probe_walk(int variant, const int* idx, const int* sh, const int* nptr, const void* apos, int N, long long P, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (N + 3) / 4;
  const Int3* s3 = (const Int3*)sh; const float4* ap = (const float4*)apos;
  switch (variant) {
    case 0: walk<0><<<blocks, 256, 0, st>>>(idx, s3, nptr, ap, N, out); break;
    case 1: walk<1><<<blocks, 256, 0, st>>>(idx, s3, nptr, ap, N, out); break;
    case 2: walk<2><<<blocks, 256, 0, st>>>(idx, s3, nptr, ap, N, out); break;
    case 3: walk<3><<<blocks, 256, 0, st>>>(idx, s3, nptr, ap, N, out); break;
    case 30: gath<30><<<blocks, 256, 0, st>>>(idx, s3, nptr, ap, N, out); break;
    case 31: gath<31><<<blocks, 256, 0, st>>>(idx, s3, nptr, ap, N, out); break;
    case 32: gath<32><<<blocks, 256, 0, st>>>(idx, s3, nptr, ap, N, out); break;
    case 34: gath<34><<<blocks, 256, 0, st>>>(idx, s3, nptr, ap, N, out); break;
    case 33: gath<33><<<blocks, 256, 0, st>>>(idx, s3, nptr, ap, N, out); break;
    case 40: walk_lockstep<<<blocks, 256, 0, st>>>(idx, s3, nptr, ap, N, out); break;
    case 41: walk_lockstep_w<8><<<(N + 7) / 8, 512, 0, st>>>(idx, s3, nptr, ap, N, out); break;
    case 42: walk_lockstep_w<16><<<(N + 15) / 16, 1024, 0, st>>>(idx, s3, nptr, ap, N, out); break;
    case 10: walk_u2<<<blocks, 256, 0, st>>>(idx, s3, nptr, ap, N, out); break;
    case 20: flat<<<256 * 16, 256, 0, st>>>(idx, s3, P, out); break;
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

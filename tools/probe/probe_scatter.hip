// Tuning probe: pure full-width streaming writes (16 B per lane, 1 KiB per wave instruction) whose 1..64 KiB chunks are visited either in
// address order or in a scattered order over a footprint of F bytes -- does the write rate depend on how many pages are live at once
// (address translation reach) rather than on the bytes written?  chunk k of wave-slot w goes to base + perm(k) * chunk, perm(k) = k * odd mod nchunks.
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void scatter_fill(f4* __restrict__ base, long long nchunks, int chunk_f4, long long mult, long long total_chunks) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (long long k = wave; k < total_chunks; k += nwaves) {
    const long long p = (k * mult) & (nchunks - 1);  // nchunks is a power of two, mult odd: a bijection of the footprint's chunks
    f4* dst = base + p * chunk_f4;
    for (int o = lane; o < chunk_f4; o += 64) dst[o] = v;
  }
}
extern "C" int probe_scatter(void* base, long long footprint_bytes, int chunk_bytes, long long mult, long long total_bytes, int blocks, void* stream) {
  const long long nchunks = footprint_bytes / chunk_bytes;
  scatter_fill<<<blocks, 256, 0, (hipStream_t)stream>>>((f4*)base, nchunks, chunk_bytes / 16, mult, total_bytes / chunk_bytes);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

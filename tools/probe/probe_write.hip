// Tuning probe: row-owner write patterns of the neighbour-matrix fill (4-byte index + 12-byte shift per slot).
#include <hip/hip_runtime.h>
struct Int3 { int a, b, c; };
// each wave owns one row of M slots and writes it in chunks of CH consecutive slots (CH = 22: what a 37 %-hit group of 64
// candidates emits; CH = 64: LDS-buffered full groups)
template <int CH>
__global__ __launch_bounds__(256) void rows(int* __restrict__ nm, Int3* __restrict__ sh, int N, int M) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const long long base = (long long)i * M;
  for (int t = 0; t < M; t += CH) {
    if (lane < CH && t + lane < M) { nm[base + t + lane] = lane + t; sh[base + t + lane] = Int3{lane, t, i}; }
  }
}
extern "C" int probe_write(int ch, int* nm, int* sh, int N, int M, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (ch == 22) rows<22><<<(N + 3) / 4, 256, 0, st>>>(nm, (Int3*)sh, N, M);
  else if (ch == 64) rows<64><<<(N + 3) / 4, 256, 0, st>>>(nm, (Int3*)sh, N, M);
  else return 1;
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

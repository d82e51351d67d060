// binsort.h -- counting sort of N items by an integer bin key (search-grid cell, mesh tile) with atomic counters.  gfx950.
//
// Replaces a general radix sort of (key, index) pairs (~20 library kernels per call) by what the problem needs:
//   count    bs_wave_add on count[key]: one atomic per distinct key per wave               (in the caller's key kernel)
//   scan     exclusive prefix over the bins: BS_CHUNK bins per block in registers + LDS, then one tiny block over the block sums
//   scatter  slot = start[key] + bs_wave_add(fill[key]): items land inside their bin's segment in arrival order
// Bin starts are produced as absolute offsets (`start_abs`, with the end sentinel start_abs[nbins] = N) by the scatter launch
// itself, so the pipeline is 1 memset + 2 kernels behind the key kernel (3 above 4 M bins).  Order INSIDE a bin is arrival order; callers that
// need a deterministic order (neighbour rows: ascending atom index) rank the few items of a bin afterwards (nlist.hip).
#pragma once
#include "common.h"

#define BS_CHUNK 4096  // bins per scan block: 256 threads x 16

namespace {

// One increment of counter[key] per active lane, with the lanes of a wave that share a key COMBINED into one atomic: neighbouring
// atoms usually sit in the same cell, and 64 same-address atomics from one wave serialise in the L2 (the reference's <= 1000-cell
// cache grids put hundreds of atoms on each counter: 0.13 ms per pass at 131k atoms before this).  Returns, when RETURN, the lane's
// slot in the counter's sequence (lanes of a wave get consecutive slots in lane order).  Must be called by all lanes of the wave.
template <bool RETURN>
__device__ __forceinline__ int bs_wave_add(int* __restrict__ counter, int key, bool active) {
  const int lane = threadIdx.x & (MI_WAVE - 1);
  // group the lanes by key first (registers and shuffles only: one trip per distinct key among the remaining lanes) ...
  int leader_of = lane, rank = 0, size = 0;
  unsigned long long todo = __ballot(active);
  while (todo) {  // wave-uniform loop
    const int leader = __ffsll((long long)todo) - 1;
    const int k0 = __shfl(key, leader, MI_WAVE);
    const unsigned long long same = __ballot(active && key == k0);
    if (active && key == k0) { leader_of = leader; rank = __popcll(same & ((1ull << lane) - 1ull)); size = __popcll(same); }
    todo &= ~same;
  }
  // ... then ONE atomic instruction for all groups at once (their leaders, different addresses): the memory round trip is paid once
  // per wave, not once per distinct key (the scatter pass of a 100k-atom binning: 16-18 -> ~7 us)
  int base = 0;
  if (active && leader_of == lane) {
    if (RETURN) base = atomicAdd(&counter[key], size);
    else atomicAdd(&counter[key], size);
  }
  if (!RETURN) return 0;
  base = __shfl(base, leader_of, MI_WAVE);
  return base + rank;
}

// count[] -> in-place exclusive prefix inside each BS_CHUNK block; block_sum[b] = total of block b.
// nbins_dev (optional): device-side number of bins in use (the scan then covers nbins_dev + 1 <= cap entries, the extra one
// being the end sentinel); blocks beyond it only publish a zero sum.
__global__ __launch_bounds__(256) void bs_scan_partial_kernel(int* __restrict__ count, const int* __restrict__ nbins_dev, long long cap,
                                                             int* __restrict__ block_sum) {
  __shared__ int wave_tot[4];
  long long n = cap;
  if (nbins_dev) { const long long want = (long long)(*nbins_dev) + 1; n = want < cap ? want : cap; }
  const long long base = (long long)blockIdx.x * BS_CHUNK + (long long)threadIdx.x * 16;
  int v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = (base + k < n) ? count[base + k] : 0;
  int tsum = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { const int c = v[k]; v[k] = tsum; tsum += c; }
  // exclusive scan of the 256 thread sums: inclusive wave scan by shuffles, wave totals through LDS
  const int lane = threadIdx.x & (MI_WAVE - 1), wave = threadIdx.x / MI_WAVE;
  int inc = tsum;
#pragma unroll
  for (int o = 1; o < MI_WAVE; o <<= 1) { const int up = __shfl_up(inc, o, MI_WAVE); if (lane >= o) inc += up; }
  if (lane == MI_WAVE - 1) wave_tot[wave] = inc;
  __syncthreads();
  int woff = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) woff += (w < wave) ? wave_tot[w] : 0;
  const int excl = woff + inc - tsum;
#pragma unroll
  for (int k = 0; k < 16; ++k) if (base + k < n) count[base + k] = excl + v[k];
  if (threadIdx.x == 255) block_sum[blockIdx.x] = woff + inc;
}

// exclusive prefix of the block sums (one block; nblocks is small: cap / 4096)
__global__ __launch_bounds__(256) void bs_scan_blocks_kernel(const int* __restrict__ block_sum, int nblocks, int* __restrict__ block_off) {
  __shared__ int wave_tot[4];
  __shared__ int carry_sh;
  if (threadIdx.x == 0) carry_sh = 0;
  __syncthreads();
  const int lane = threadIdx.x & (MI_WAVE - 1), wave = threadIdx.x / MI_WAVE;
  for (int b0 = 0; b0 < nblocks; b0 += 256) {
    const int b = b0 + threadIdx.x;
    const int x = b < nblocks ? block_sum[b] : 0;
    int inc = x;
#pragma unroll
    for (int o = 1; o < MI_WAVE; o <<= 1) { const int up = __shfl_up(inc, o, MI_WAVE); if (lane >= o) inc += up; }
    if (lane == MI_WAVE - 1) wave_tot[wave] = inc;
    __syncthreads();
    int woff = carry_sh;
#pragma unroll
    for (int w = 0; w < 4; ++w) woff += (w < wave) ? wave_tot[w] : 0;
    if (b < nblocks) block_off[b] = woff + inc - x;
    __syncthreads();
    if (threadIdx.x == 255) carry_sh = woff + inc;
    __syncthreads();
  }
}

#define BS_INLINE_BLOCKS 1024  // up to this many scan blocks (4 M bins) the scatter launch prefixes the block sums itself

// thread t < N: item t goes to slot start[key] + (arrival rank in its bin); thread t <= nbins: absolute bin start t
// (start_abs[nbins] = N is the end sentinel).  `local` is read-only here.  INLINE (round 4): every block first turns the <= 1024 block
// sums of the partial scan into their exclusive prefix in LDS (4 loads + one shuffle scan per thread) -- the one-block
// bs_scan_blocks_kernel between the two launches, ~5 us of pure launch latency three times per step, is gone; with more scan blocks
// than that the prefix comes from that kernel through `block_off` as before.
template <bool INLINE>
__global__ __launch_bounds__(256) void bs_scatter_kernel(const int* __restrict__ keys, int N, const int* __restrict__ local,
                                                        const int* __restrict__ block_sum, int nblocks, const int* __restrict__ block_off,
                                                        const int* __restrict__ nbins_dev, long long cap, int* __restrict__ fill,
                                                        int* __restrict__ items_out, int* __restrict__ start_abs) {
  __shared__ int s_off[INLINE ? BS_INLINE_BLOCKS : 1];
  __shared__ int wave_tot[4];
  if (INLINE) {
    const int lane = threadIdx.x & (MI_WAVE - 1), wave = threadIdx.x / MI_WAVE;
    int v[4], tsum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int b = threadIdx.x * 4 + k; v[k] = b < nblocks ? block_sum[b] : 0; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int c = v[k]; v[k] = tsum; tsum += c; }
    int inc = tsum;
#pragma unroll
    for (int o = 1; o < MI_WAVE; o <<= 1) { const int up = __shfl_up(inc, o, MI_WAVE); if (lane >= o) inc += up; }
    if (lane == MI_WAVE - 1) wave_tot[wave] = inc;
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) woff += (w < wave) ? wave_tot[w] : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_off[threadIdx.x * 4 + k] = woff + inc - tsum + v[k];
    __syncthreads();
  }
  auto start_of = [&](int key) { return local[key] + (INLINE ? s_off[key / BS_CHUNK] : block_off[key / BS_CHUNK]); };
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = cap;
  if (nbins_dev) { const long long want = (long long)(*nbins_dev) + 1; n = want < cap ? want : cap; }
  if (start_abs && t < n) start_abs[t] = start_of((int)t);
  const bool in = t < N;
  const int key = in ? keys[t] : 0;
  const int rank = bs_wave_add<true>(fill, key, in);
  if (in) items_out[start_of(key) + rank] = (int)t;
}

// host side: scratch = {count[cap], fill[cap], block_sum[nb], block_off[nb]} (ints); count and fill are contiguous so that one
// memset clears both.  The caller's key kernel runs between bs_clear and bs_sort and does atomicAdd(&count[key], 1) per item.
struct BsScratch { int* count; int* fill; int* block_sum; int* block_off; long long cap; int nblocks; };
inline size_t bs_scratch_ints(long long cap) { return (size_t)(2 * cap + 2 * ((cap + BS_CHUNK - 1) / BS_CHUNK) + 8); }
inline BsScratch bs_carve(int* base, long long cap) {
  BsScratch s;
  s.cap = cap;
  s.nblocks = (int)((cap + BS_CHUNK - 1) / BS_CHUNK);
  s.count = base;
  s.fill = base + cap;
  s.block_sum = base + 2 * cap;
  s.block_off = s.block_sum + s.nblocks + 4;
  return s;
}
// (cleared in whole 16-byte words: a byte count that is not a multiple of 16 makes the runtime split the memset into two fill kernels; the up
// to three ints a round-up adds belong to block_sum, which the scan writes before anything reads it)
inline hipError_t bs_clear(const BsScratch& s, hipStream_t st) { return hipMemsetAsync(s.count, 0, (sizeof(int) * 2 * (size_t)s.cap + 15) / 16 * 16, st); }
// after the key kernel: scan + scatter.  items_out[N] receives the item indices grouped by bin, start_abs[<= cap] the bin starts.
inline hipError_t bs_sort(const BsScratch& s, const int* keys, int N, const int* nbins_dev, int* items_out, int* start_abs, hipStream_t st) {
  bs_scan_partial_kernel<<<s.nblocks, 256, 0, st>>>(s.count, nbins_dev, s.cap, s.block_sum);
  const long long threads = (long long)N > s.cap ? (long long)N : s.cap;
  if (s.nblocks <= BS_INLINE_BLOCKS) {
    bs_scatter_kernel<true><<<mi_blocks(threads, 256), 256, 0, st>>>(keys, N, s.count, s.block_sum, s.nblocks, nullptr, nbins_dev, s.cap, s.fill,
                                                                     items_out, start_abs);
  } else {
    bs_scan_blocks_kernel<<<1, 256, 0, st>>>(s.block_sum, s.nblocks, s.block_off);
    bs_scatter_kernel<false><<<mi_blocks(threads, 256), 256, 0, st>>>(keys, N, s.count, s.block_sum, s.nblocks, s.block_off, nbins_dev, s.cap, s.fill,
                                                                      items_out, start_abs);
  }
  return hipGetLastError();
}

}  // namespace

// nlist.hip -- batched cell-list neighbour search for MI355X (gfx950, wave64).
//
// What it computes is what the reference's cell-list path computes (neighborlist/cell_list.py:372-556,
// batch_cell_list.py:381-569, neighbor_utils.py:106-147): for every atom i the set of (j, S) with
// |r_j - r_i + S.cell|^2 < rc^2, S the integer lattice shift, evaluated with the reference's floating-point
// expression order (file compiled with -ffp-contract=off).  How it computes it is MI355X-first:
//
//   setup    one thread per system: own binning (cell edge ~ rc/k, k from the local density; NOT the
//            reference's 1000-cell cap) -- the result set does not depend on the binning
//   assign   one thread per atom: cell key + integer wrap of atoms outside the box + one atomic on the cell's counter
//   sort     counting sort by cell (binsort.h): block scan of the counters -> cell starts, scatter of the atom ids into their
//            cell's segment through a second set of atomic counters (3 small kernels; a general radix sort of (key, index)
//            pairs was ~20 library launches per list)
//   gather   rank of every atom inside its cell by atom index (the few ids of a cell are compared from L1; => ascending index in
//            a cell, rows are deterministic) and, in the same kernel, the cell-ordered float4/double4 record {x,y,z,index} the
//            query streams, the wraps and the per-slot cell key
//   query    ONE WAVE64 PER ATOM walks the full shell of cells; x-adjacent cells are contiguous in the sorted
//            order, so each (dy,dz) row is one coalesced run; 64 candidates are tested per step, hits are
//            compacted with __ballot/__popcll and written to the row owner's slots (coalesced, no atomics,
//            deterministic order); the owner also writes the padding => no separate torch.full pass.
//
//   query'   dense cells (large cutoffs): ONE BLOCK PER CELL, cells handed out through an atomic counter; the candidate stream
//            of the cell is staged tile-wise in LDS, two centre atoms per wave share each candidate fetched, runs are ordered
//            by periodic image so that S.cell is a per-64-group constant.  Both query kernels are launched, a device flag
//            written by `setup` decides which one works (no host sync).
//
// The same query kernels serve the padded matrix, the count pass and the direct CSR/COO fill.
#include <stdlib.h>

#include "binsort.h"
#include "common.h"

namespace {

#define NL_PRUNE_R 8
#ifndef NL_TILE
#define NL_TILE 1024      // candidates staged in LDS per tile (fp64: 34 KB per block)
#endif
#ifndef NL_TILE_F64
#define NL_TILE_F64 768
#endif
#ifndef NL_TILED_WAVES
#define NL_TILED_WAVES 0   // > 0: register cap of the tiled query for this many waves per SIMD (tuning aid)
#endif
#ifndef NL_CCHUNK
#define NL_CCHUNK 64     // centre atoms of a cell handled per sweep over its candidate tiles (cells above this are swept again); 8 / 16 / 32 (rows
                         // written in fewer, longer stretches at the price of re-staging the tiles): -3 % into badly placed buffers, +3 % into well
                         // placed ones (profiles/r05_ab_nl_cchunk.log)
#endif
#define NL_MAXROWS 96     // (2Ry+1)(2Rz+1) rows of cells a block can describe (R <= 4)
#define NL_MIXED 0x7fffffff
#define NL_TILED_GRID 1536   // persistent blocks (6 per CU); cells are handed out dynamically
// tuning aid: NVALCHEMIOPS_NL_TILED_GRID in the environment overrides the number of persistent blocks (read once)
static int nl_tiled_grid() {
  static const int g = [] { const char* e = getenv("NVALCHEMIOPS_NL_TILED_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : NL_TILED_GRID; }();
  return g;
}
template <class T> struct NlSys {
  T cell[9];
  T inv[9];
  T origin[3];  // subtracted before binning only
  int cpd[3];
  int R[3];
  int pbc[3];
  int nrange[3];  // image range of the naive method (neighbor_utils.py:150-211)
  int cell_off;
  int ncells;
  int tiled;      // dense cells (k >= 2) with a small row table: eligible for the block-per-cell LDS-tiled query
  int prune;      // orthorhombic cell with R <= NL_PRUNE_R: dxlim[|dz|][|dy|] = largest |dx| worth visiting (-1: skip the row)
  signed char dxlim[NL_PRUNE_R + 1][NL_PRUNE_R + 1];
};
struct NlGlobal {
  int total_cells; int any_wrap; int use_tiled; int pad1;
  int work[4];  // next cell to hand out, per query mode (dynamic scheduling of the tiled kernel)
  int done[4];  // blocks that ran out of work; the last one re-arms both counters for the next launch
};

struct NlLayout {
  size_t sys, glob, natoms, keys_in, keys_out, vals_in, vals_out, wrap, swrap, spos, srk, cell_start, bins, total;
  long long cell_cap;
};

inline long long nl_cell_cap(int N, int B) { return 4ll * N + 8ll * B; }

inline int nl_key_bits(long long cap) {
  int bits = 1;
  while ((1ll << bits) <= cap && bits < 31) ++bits;
  return bits;
}

NlLayout nl_layout(int N, int B, int dtype) {
  NlLayout L;
  size_t o = 0;
  const size_t sysz = dtype == MI_F32 ? sizeof(NlSys<float>) : sizeof(NlSys<double>);
  const size_t esz = dtype == MI_F32 ? 4 : 8;
  L.cell_cap = nl_cell_cap(N, B);
  auto take = [&](size_t bytes) { size_t at = o; o += mi_align(bytes); return at; };
  L.sys = take(sysz * (size_t)B);
  L.glob = take(sizeof(NlGlobal));
  L.natoms = take(sizeof(int) * (size_t)B);
  L.keys_in = take(sizeof(int) * (size_t)N);
  L.keys_out = take(sizeof(int) * (size_t)N);
  L.vals_in = take(sizeof(int) * (size_t)N);
  L.vals_out = take(sizeof(int) * (size_t)N);
  L.wrap = take(8 * (size_t)N);
  L.swrap = take(8 * (size_t)N);
  L.spos = take(4 * esz * (size_t)N);
  L.srk = take(sizeof(float) * (size_t)N);  // cell-ordered scaled covalent radii (searches that also sum coordination numbers, NlCn)
  L.cell_start = take(sizeof(int) * (size_t)(L.cell_cap + 2));
  L.bins = take(sizeof(int) * bs_scratch_ints(L.cell_cap + 2));  // counting-sort counters (binsort.h)
  L.total = o;
  return L;
}

__device__ __forceinline__ float4 pack4(float x, float y, float z, int idx) { return make_float4(x, y, z, __int_as_float(idx)); }
__device__ __forceinline__ double4 pack4(double x, double y, double z, int idx) { return make_double4(x, y, z, (double)idx); }
__device__ __forceinline__ int idx_of(const float4& v) { return __float_as_int(v.w); }
__device__ __forceinline__ int idx_of(const double4& v) { return (int)v.w; }

__global__ void nl_count_atoms_kernel(const int* __restrict__ batch_idx, int N, int* __restrict__ natoms) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < N;
  const int s = in ? batch_idx[i] : -1;
  // atoms of a system are normally contiguous: one atomic per wave instead of 64 colliding on the same address
  const int s0 = __shfl(s, 0, MI_WAVE);
  const unsigned long long same = __ballot(in && s == s0);
  if ((threadIdx.x & (MI_WAVE - 1)) == 0 && s0 >= 0) atomicAdd(&natoms[s0], __popcll(same));
  if (in && s != s0) atomicAdd(&natoms[s], 1);
}

// Grid description of one system (cells per dimension from the density, search radius, image range, pruning eligibility): everything of
// NlSys but the pruning table and the cell offset.  Deterministic in its inputs: the table blocks of nl_setup_kernel recompute it.
template <class T>
__device__ void nl_describe_system(const T* __restrict__ cell, const uint8_t* __restrict__ pbc, const int* __restrict__ natoms, int N, T cutoff,
                                   const T* __restrict__ origin, int s, NlSys<T>& S) {
  for (int k = 0; k < 9; ++k) S.cell[k] = cell[9 * (size_t)s + k];
  for (int k = 0; k < 3; ++k) S.origin[k] = origin ? origin[3 * (size_t)s + k] : T(0);
  inverse3(S.cell, S.inv);
  const T* a = S.cell;
  double det = (double)a[0] * ((double)a[4] * a[8] - (double)a[5] * a[7]) - (double)a[1] * ((double)a[3] * a[8] - (double)a[5] * a[6]) +
               (double)a[2] * ((double)a[3] * a[7] - (double)a[4] * a[6]);
  double vol = fabs(det);
  int ns = natoms ? natoms[s] : N;
  double rc = (double)cutoff;
  double apc = vol > 0 ? (double)ns / vol * rc * rc * rc : 0.0;  // atoms per rc^3 cube
#ifndef NL_K2_APC
#define NL_K2_APC 64.0
#endif
  int k = apc < NL_K2_APC ? 1 : (apc < 512.0 ? 2 : 3);
  long long cap = 4ll * ns + 8;
  double face[3];
  for (int d = 0; d < 3; ++d) {
    T col[3] = {S.inv[d], S.inv[3 + d], S.inv[6 + d]};
    T ln = sqrt(col[0] * col[0] + col[1] * col[1] + col[2] * col[2]);
    face[d] = 1.0 / (double)ln;
    S.pbc[d] = pbc[3 * (size_t)s + d] ? 1 : 0;
    double want = face[d] * k / (rc * (1.0 + 2e-6));  // cell edge strictly above rc/k: a box that is an exact multiple of rc/k must not round the search radius up to k+1
    S.cpd[d] = want >= 1048576.0 ? 1048576 : (want >= 1.0 ? (int)want : 1);
    S.nrange[d] = S.pbc[d] ? (int)ceil(ln * cutoff) : 0;
  }
  long long tot = (long long)S.cpd[0] * S.cpd[1] * S.cpd[2];
  while (tot > cap) {
    for (int d = 0; d < 3; ++d) S.cpd[d] = S.cpd[d] / 2 > 1 ? S.cpd[d] / 2 : 1;
    tot = (long long)S.cpd[0] * S.cpd[1] * S.cpd[2];
  }
  for (int d = 0; d < 3; ++d) {
    if (S.cpd[d] == 1 && !S.pbc[d]) S.R[d] = 0;
    else S.R[d] = (int)ceil(rc * S.cpd[d] / face[d] * (1.0 + 1e-6));
  }
  S.ncells = (int)tot;
  S.cell_off = 0;
  S.tiled = ((k >= 2 || apc >= 32.0) && (2 * S.R[1] + 1) * (2 * S.R[2] + 1) <= NL_MAXROWS && (!S.pbc[0] || S.R[0] <= S.cpd[0])) ? 1 : 0;
  // Orthorhombic cells: atoms in cells offset by d cells along a periodic axis are at least (|d|-1) cell edges apart along it,
  // so cells/rows provably beyond the cutoff are never visited.  Non-periodic axes contribute 0 (clamped atoms may sit
  // outside their cell).  Evaluated once here; the query kernel only does integer look-ups.
  const bool ortho = a[1] == T(0) && a[2] == T(0) && a[3] == T(0) && a[5] == T(0) && a[6] == T(0) && a[7] == T(0);
  S.prune = (ortho && S.R[0] <= NL_PRUNE_R && S.R[1] <= NL_PRUNE_R && S.R[2] <= NL_PRUNE_R) ? 1 : 0;
}

// One launch, three kinds of blocks (round 4):
//   block 0                      every system's grid description (all NlSys fields but the pruning table) + the cell offsets (block scan);
//   blocks 1 .. table_blocks     the pruning tables, one thread per (system, |dz|, |dy|) entry: the thread re-derives its system's description
//                                in registers (cheap, deterministic) and writes ONE byte -- 81 dependent fp64 sqrt / divide chains per
//                                system used to run in block 0 alone (0.08 ms for the 128 / 256 systems of BASELINE configs 5 / 3);
//   the remaining blocks         clear the counting-sort counters of the same call (`zero`, 16-byte words): no memset node in front of the
//                                binning.  The assign kernel behind this launch needs all three.
// Block 0 and the table blocks write disjoint bytes of sys[s] (block 0 never stores dxlim).
template <class T>
__global__ __launch_bounds__(256) void nl_setup_kernel(const T* __restrict__ cell, const uint8_t* __restrict__ pbc, const int* __restrict__ natoms, int N,
                                int B, T cutoff, const T* __restrict__ origin, NlSys<T>* __restrict__ sys, NlGlobal* __restrict__ glob,
                                int table_blocks, int4* __restrict__ zero, long long zero_words, int* __restrict__ pk_flag,
                                int* __restrict__ cn_hdr, float cn_scale) {
  constexpr int NE = (NL_PRUNE_R + 1) * (NL_PRUNE_R + 1);
  if (pk_flag && blockIdx.x == 0 && threadIdx.x == 0) *pk_flag = 0;  // the packed companion's "unusable" flag starts clear (see NlPacked)
  if (cn_hdr && blockIdx.x == 0) {  // header of the coordination-number block (common.h): flag clear, cutoff, scale, checksum slots zero
    if (threadIdx.x == 0) { cn_hdr[0] = 0; cn_hdr[1] = 0; cn_hdr[2] = __float_as_int((float)cutoff); cn_hdr[3] = __float_as_int(cn_scale); }
    if (threadIdx.x < MI_CN_SLOTS) reinterpret_cast<unsigned long long*>(cn_hdr)[MI_CN_SLOT_OFFSET_U64 + threadIdx.x] = 0ull;
  }
  if ((int)blockIdx.x > table_blocks) {
    const long long zb = (long long)blockIdx.x - table_blocks - 1, nzb = (long long)gridDim.x - table_blocks - 1;
    for (long long k = zb * blockDim.x + threadIdx.x; k < zero_words; k += nzb * blockDim.x) zero[k] = make_int4(0, 0, 0, 0);
    return;
  }
  if (blockIdx.x > 0) {
    const int t = ((int)blockIdx.x - 1) * blockDim.x + threadIdx.x;
    if (t >= B * NE) return;
    const int s = t / NE, az = (t - s * NE) / (NL_PRUNE_R + 1), ay = t - s * NE - az * (NL_PRUNE_R + 1);
    NlSys<T> S;
    nl_describe_system<T>(cell, pbc, natoms, N, cutoff, origin, s, S);
    if (!S.prune) return;
    const double rc = (double)cutoff;
    // cell edge per axis (0 for a non-periodic one): an orthorhombic cell has a diagonal inverse, so the norm of column d of cell^-1 is |inv[d][d]|
    double w[3];
    for (int d = 0; d < 3; ++d) {
      const double ln = fabs((double)S.inv[4 * d]);
      w[d] = (S.pbc[d] && ln > 0.0) ? (1.0 / ln) / S.cpd[d] : 0.0;
    }
    const double gz = (az > 1 ? az - 1 : 0) * w[2], gy = (ay > 1 ? ay - 1 : 0) * w[1];
    const double rem = rc * rc - gz * gz - gy * gy;
    int lim;
    if (rem < -1e-6 * rc * rc) lim = -1;
    else if (w[0] > 0.0) { lim = (int)(sqrt(rem > 0.0 ? rem : 0.0) / w[0] * (1.0 + 1e-6)) + 1; if (lim > S.R[0]) lim = S.R[0]; }
    else lim = S.R[0];
    sys[s].dxlim[az][ay] = (signed char)lim;
    return;
  }
  for (int s = threadIdx.x; s < B; s += blockDim.x) {
    NlSys<T> S;
    nl_describe_system<T>(cell, pbc, natoms, N, cutoff, origin, s, S);
    NlSys<T>* dst = sys + s;  // field by field: the pruning table of this struct belongs to the table blocks
    for (int k = 0; k < 9; ++k) { dst->cell[k] = S.cell[k]; dst->inv[k] = S.inv[k]; }
    for (int k = 0; k < 3; ++k) { dst->origin[k] = S.origin[k]; dst->cpd[k] = S.cpd[k]; dst->R[k] = S.R[k]; dst->pbc[k] = S.pbc[k]; dst->nrange[k] = S.nrange[k]; }
    dst->cell_off = 0; dst->ncells = S.ncells; dst->tiled = S.tiled; dst->prune = S.prune;
  }
  __syncthreads();
  // cell offsets = exclusive prefix of the per-system cell counts, 256 systems per trip by a block scan (round 4: one thread walking the
  // systems one dependent global read-modify-write at a time cost 0.12 ms for the 256 molecules of BASELINE config 3)
  __shared__ int wave_tot[4];
  __shared__ int carry_sh, tiled_sh;
  if (threadIdx.x == 0) { carry_sh = 0; tiled_sh = 1; }
  __syncthreads();
  const int lane = threadIdx.x & (MI_WAVE - 1), wave = threadIdx.x / MI_WAVE;
  for (int s0 = 0; s0 < B; s0 += blockDim.x) {
    const int s = s0 + threadIdx.x;
    const int x = s < B ? sys[s].ncells : 0;
    if (s < B && !sys[s].tiled) tiled_sh = 0;  // benign race: every writer stores 0
    int inc = x;
#pragma unroll
    for (int o = 1; o < MI_WAVE; o <<= 1) { const int up = __shfl_up(inc, o, MI_WAVE); if (lane >= o) inc += up; }
    if (lane == MI_WAVE - 1) wave_tot[wave] = inc;
    __syncthreads();
    int woff = carry_sh;
    for (int w = 0; w < 4; ++w) woff += (w < wave) ? wave_tot[w] : 0;
    if (s < B) sys[s].cell_off = woff + inc - x;
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) carry_sh = woff + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    glob->total_cells = carry_sh;
    glob->any_wrap = 0;
    glob->use_tiled = tiled_sh;
    for (int k = 0; k < 4; ++k) { glob->work[k] = 0; glob->done[k] = 0; }
  }
}

template <class T>
__global__ void nl_assign_kernel(const T* __restrict__ pos, const int* __restrict__ batch_idx, int N, const NlSys<T>* __restrict__ sys,
                                 int* __restrict__ keys, int* __restrict__ count, short4* __restrict__ wrap, NlGlobal* __restrict__ glob) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < N;
  int key = 0;
  if (in) {
    const int s = batch_idx ? batch_idx[i] : 0;
    const NlSys<T>& S = sys[s];
    T p[3] = {pos[3 * (size_t)i] - S.origin[0], pos[3 * (size_t)i + 1] - S.origin[1], pos[3 * (size_t)i + 2] - S.origin[2]}, frac[3];
    rowvec_mat3(p, S.inv, frac);
    int c[3], w[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      int v = (int)floor(frac[d] * (T)S.cpd[d]);
      if (S.pbc[d]) floor_divmod(v, S.cpd[d], w[d], c[d]);
      else { w[d] = 0; c[d] = v < 0 ? 0 : (v >= S.cpd[d] ? S.cpd[d] - 1 : v); }
    }
    key = S.cell_off + c[0] + S.cpd[0] * (c[1] + S.cpd[1] * c[2]);
    keys[i] = key;
    wrap[i] = make_short4((short)w[0], (short)w[1], (short)w[2], 0);
    if (w[0] | w[1] | w[2]) glob->any_wrap = 1;
  }
  bs_wave_add<false>(count, key, in);  // the cell's atom counter, one atomic per distinct cell per wave
}

// Slot p of the counting sort holds some atom of cell c = keys[id]; its final place is start[c] + (number of atoms of the cell
// with a smaller index): rows come out in ascending atom index inside a cell whatever order the atomics ran in.  The ids of a
// cell are adjacent, the lanes of a wave sit in one or two cells and read the same addresses (L1 broadcasts).  Same kernel:
// the cell-ordered record the query streams.
// What a search that also sums DFT-D3 coordination numbers (NlCn below) needs from the binning stage: the atoms' scaled covalent radii in
// cell order, and the fingerprint of everything the sums are computed from (common.h: positions, radii, cell, k1) in the block's header.
struct NlCnBuild { const int* numbers; const float* rcov; int nz; float K; float* srk; unsigned long long* slots; const void* cell; int n_cell; const int* batch_idx; };
template <class T>
__global__ void nl_rank_gather_kernel(const T* __restrict__ pos, const int* __restrict__ ids, const int* __restrict__ keys,
                                      const int* __restrict__ cell_start, const short4* __restrict__ wrap, int N,
                                      typename Vec4<T>::type* __restrict__ spos, short4* __restrict__ swrap, int* __restrict__ keys_sorted,
                                      NlCnBuild C) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long h = 0;
  if (p < N) {
    const int i = ids[p];
    const int key = keys[i];
    const int b = cell_start[key], e = cell_start[key + 1];
    int rank = 0;
    for (int q = b; q < e; ++q) rank += ids[q] < i ? 1 : 0;
    const int dst = b + rank;
    const T x = pos[3 * (size_t)i], y = pos[3 * (size_t)i + 1], z = pos[3 * (size_t)i + 2];
    spos[dst] = pack4(x, y, z, i);
    swrap[dst] = wrap[i];
    keys_sorted[dst] = key;
    if (C.srk) {
      const float rk = mi_cn_rk(C.numbers[i], C.nz, C.rcov, C.K);
      C.srk[dst] = rk;
      h = mi_cn_atom_hash<T>(i, C.batch_idx ? C.batch_idx[i] : 0, x, y, z, rk);
    }
  }
  if (C.srk) {  // (kernel-uniform) every lane takes part in the wave-level sum
    if (p < C.n_cell) h += mi_cn_cell_hash<T>(p, reinterpret_cast<const T*>(C.cell)[p]);
    if (p == 0) h += mi_cn_scale_hash(C.K);
    mi_cn_slot_add(C.slots, p / MI_WAVE, h);
  }
}

struct NlInt3 { int a, b, c; };  // one 12-byte store per hit for the unit shift

// coalesced wave-wide fill of dst[begin,end) with `value`; 16-byte stores in the aligned body
template <bool NT = true>
__device__ __forceinline__ void wave_fill(int* __restrict__ dst, long long begin, long long end, int value, int lane) {
  long long n = end - begin;
  if (n <= 0) return;
  int* p = dst + begin;
  int head = (int)(((16 - ((uintptr_t)p & 15)) & 15) >> 2);
  if (head > n) head = (int)n;
  if (lane < head) p[lane] = value;
  long long body = (n - head) >> 2;
  int4* p4 = reinterpret_cast<int4*>(p + head);
  // padding is never read back by this library: non-temporal stores (ref-nlist 524 288 atoms, 78 % padding: 1.09 -> 1.01 ms)
  typedef int nl_i4 __attribute__((ext_vector_type(4)));
  const nl_i4 w4 = {value, value, value, value};
  for (long long t = lane; t < body; t += MI_WAVE) {
    if constexpr (NT) __builtin_nontemporal_store(w4, reinterpret_cast<nl_i4*>(p4 + t));
    else *reinterpret_cast<nl_i4*>(p4 + t) = w4;
  }
  long long done = head + (body << 2);
  if (done + lane < n) p[done + lane] = value;
}

// inclusive prefix of the run lengths run_pre[1..n] (run_pre[0] = 0), by one wave: a shuffle scan per 64 runs (the serial loop was up to 243
// dependent LDS round trips per cell).  Not inlined: the scan's registers must not count towards the query kernel's budget (5 waves / SIMD).
__device__ __noinline__ void nl_prefix_runs(int* run_pre, int n, int lane) {
  int carry = 0;
  if (lane == 0) run_pre[0] = 0;
  for (int r0 = 0; r0 < n; r0 += MI_WAVE) {
    const int r = r0 + lane;
    int inc = r < n ? run_pre[r + 1] : 0;
#pragma unroll
    for (int o = 1; o < MI_WAVE; o <<= 1) { const int up = __shfl_up(inc, o, MI_WAVE); if (lane >= o) inc += up; }
    if (r < n) run_pre[r + 1] = carry + inc;
    carry += __shfl(inc, MI_WAVE - 1, MI_WAVE);
  }
}

// a 64-bit value every lane of the wave agrees on, moved to scalar registers (so that pointers built from it are wave-uniform)
__device__ __forceinline__ long long nl_uniform64(long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}

// a wave-uniform float / double pinned into scalar registers (the compiler cannot prove uniformity of a value loaded from LDS)
__device__ __forceinline__ float nl_uni(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ double nl_uni(double x) { return __longlong_as_double(nl_uniform64(__double_as_longlong(x))); }

// The pair test shared by both query kernels: reference expression order (cell_list.py:531-544), or the naive method's
// orientation / image range (naive.py:163-172), self-pair exclusion, canonical half-fill rule.
// Second output set of the single-sweep dual-cutoff search (naive_dual_cutoff.py:115-290: one walk over the pair set, `dist_sq < cutoff2_sq`
// fills list 2 and, nested in it, `dist_sq < cutoff1_sq` fills list 1 -- same image range for both).  The primary outputs of the query
// kernels take the LONG cutoff, this struct the short one.
template <class T> struct NlSecond { T rc2; int* nm; int* nsh; int* num; int M; };
// Optional by-product of a matrix-mode search (round 5): a 4-byte-per-slot companion of the padded matrix in the format mi_d3's passes
// stream (`d3_fetch_pk`, csrc/d3.hip): neighbour index in bits 0-25, unit shift + 1 in three 2-bit fields, all ones = padding.  The search has
// index and shift of every hit in registers when it stores the 16 bytes of the API format; emitting the packed word there costs one more
// 4-byte store per hit and saves the consumer the 16 B/slot read + 4 B/slot write of the pass that used to derive it (mi_d3's CN pass:
// 5.1 GB -> 1.0 GB on the headline list).  `flag` is raised when a stored shift lies outside {-1, 0, 1} (the companion is then unusable
// and the consumer reads the API arrays, bit-identical results).  words == nullptr: nothing is written.
struct NlPacked { unsigned* words; int* flag; };
// Optional by-product #2 of a matrix-mode search (round 6): the DFT-D3 coordination numbers CN_i = sum_j 1 / (1 + exp(-k1 ((rcov_i + rcov_j) / r_ij - 1)))
// over exactly the pairs the search stores (`_cn_kernel_nm`, dftd3.py:833-941, sums over the row; `_cn_counting` :608-645).  The search has
// d^2 of every hit in registers and both atoms' records in LDS, so the sum costs one rsq, one exp2, one rcp and three FMAs per tested pair
// inside a kernel that waits for its stores to drain -- and mi_d3's CN pass (a walk over the whole list with one 16-byte gather per
// neighbour) disappears.  fp32 terms, lane partials of <= 16 terms summed in fp64 across lanes and passes; the exponential is a plain
// v_exp_f32 of K - K rr (K = k1 log2 e): df <= f (1 - f) ln2 |t| 2^-23 <= 3e-7 where f is not tiny, the size of the rsq's own error.
// `flag` is raised when a row overflowed its M slots (the reference's sum stops at the stored entries): the consumer then runs its own pass.
struct NlCn { const float* rk; float* cn; int* flag; float K; };
__device__ __forceinline__ float nl_cn_term(float d2, float rki, float rkj, float K) {
  const float rinv = __builtin_amdgcn_rsqf(d2);
  const float t = fmaf(-(rki + rkj), rinv, K);  // -k1 (rr - 1) log2(e); an atom outside the tables carries rk = -inf: t = +inf, the term is 0
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
#define NL_PK_PAD 0xffffffffu
// NL_PREZERO_SHIFTS (round 5): in matrix mode with padding, the tiled kernel's owner wave zero-fills the whole shifts row with wide
// non-temporal streaming stores when it first meets a centre (30 KB in one go for the headline rows: DRAM-page-friendly), and the hits of
// un-shifted image groups -- 3 of 4 hits in a large box -- then store no shift at all; the row's padding needs no second pass.  Same bytes
// in the end, fewer of them in short runs: the fill into badly placed buffers drops from 1.30 - 1.55 to 1.17 - 1.39 ms,
// into well placed ones it is 1.07 - 1.17 either way (profiles/r05_ab_prezero_shifts.log; plain instead of non-temporal zero stores: no
// gain).  The wave waits for its zero stores (vmcnt) before the first hit store of the row, so program order is memory order.
#ifndef NL_PREZERO_SHIFTS
#define NL_PREZERO_SHIFTS 1
#endif
#ifndef NL_PREZERO_NT
#define NL_PREZERO_NT 1
#endif
#define NL_PK_ZERO_SHIFT 0x54000000u  // (0 + 1) in each of the three fields
__device__ __forceinline__ unsigned nl_pk_code(int Sx, int Sy, int Sz, bool& bad) {
  const unsigned cx = (unsigned)(Sx + 1), cy = (unsigned)(Sy + 1), cz = (unsigned)(Sz + 1);
  bad = (cx > 2u) | (cy > 2u) | (cz > 2u);
  return ((cx & 3u) << 26) | ((cy & 3u) << 28) | (cz << 30);
}

template <class T>
__device__ __forceinline__ bool nl_pair_hit(T pix, T piy, T piz, int i, T cjx, T cjy, T cjz, int j, int Sx, int Sy, int Sz, const T* cart,
                                            T rc2, bool naive, bool half, int nr0, int nr1, int nr2, T* d2_out = nullptr) {
  T dr0, dr1, dr2;
  bool ok = true;
  if (!naive) {
    dr0 = (cjx - pix) + cart[0];
    dr1 = (cjy - piy) + cart[1];
    dr2 = (cjz - piz) + cart[2];
  } else {
    const bool upper = Sx > 0 || (Sx == 0 && Sy > 0) || (Sx == 0 && Sy == 0 && Sz >= 0);
    if (upper) { dr0 = (cart[0] + cjx) - pix; dr1 = (cart[1] + cjy) - piy; dr2 = (cart[2] + cjz) - piz; }
    else { dr0 = ((-cart[0]) + pix) - cjx; dr1 = ((-cart[1]) + piy) - cjy; dr2 = ((-cart[2]) + piz) - cjz; }
    ok = (Sx <= nr0 && -Sx <= nr0) && (Sy <= nr1 && -Sy <= nr1) && (Sz <= nr2 && -Sz <= nr2);
  }
  const T d2 = dr0 * dr0 + dr1 * dr1 + dr2 * dr2;
  if (d2_out) *d2_out = d2;
  const bool zeroS = (Sx | Sy | Sz) == 0;
  bool hit = ok && (d2 < rc2) && !(j == i && zeroS);
  if (half && hit) {
    const bool pos_shift = Sx > 0 || (Sx == 0 && Sy > 0) || (Sx == 0 && Sy == 0 && Sz > 0);
    hit = pos_shift || (zeroS && j > i);
  }
  return hit;
}

// ---- block-per-cell, LDS-tiled query for dense cells (large cutoffs) -----------------------------------------------------
// All atoms of a cell share one candidate set.  A block describes the rows of neighbouring cells once, flattens them into
// one candidate stream, stages it tile by tile in LDS with coalesced loads (positions, neighbour index, image shift), and
// its four waves then test their centre atoms against the tile from LDS: no per-run bookkeeping and no global-memory
// latency inside the 64-candidates-per-step loop, and every candidate record is fetched from L2 once per cell, not once
// per atom.  Output path (ballot/popcount compaction, owner-written padding) is the same as the wave-per-atom kernel's.
// FAST = neither naive-expression nor half-fill requested: the per-candidate test is straight-line code.
// CNF: also sum the DFT-D3 coordination numbers of the centres over their hits (NlCn; matrix mode, FAST path only).
template <class T, int MODE, bool FAST, bool DUAL = false, bool CNF = false>
#if NL_TILED_WAVES > 0
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NL_TILED_WAVES, NL_TILED_WAVES))) void nl_query_tiled_kernel(
#else
// (the CN-summing float variant is held to the plain variant's 5 waves / SIMD: left alone the compiler takes 113 VGPRs = 4 waves)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((CNF && sizeof(T) == 4) ? 5 : 1, (CNF && sizeof(T) == 4) ? 5 : 8))) void nl_query_tiled_kernel(
#endif
    const typename Vec4<T>::type* __restrict__ spos, const short4* __restrict__ swrap, const int* __restrict__ cell_start,
    const NlSys<T>* __restrict__ sys, const NlGlobal* __restrict__ glob, int B, T rc2, int flags, int* __restrict__ nm,
    int* __restrict__ nsh, int* __restrict__ num, int M, int fill_value, const int* __restrict__ ptr, int* __restrict__ list_ij,
    int* __restrict__ list_sh, long long P, NlSecond<T> D, NlPacked K, NlCn CN = NlCn{nullptr, nullptr, nullptr, 0.0f}) {
  static_assert(!DUAL || (MODE == MI_NL_MODE_MATRIX && !FAST), "the dual-cutoff sweep is instantiated for the general matrix kernel only");
  static_assert(!CNF || (MODE == MI_NL_MODE_MATRIX && FAST && !DUAL), "coordination numbers ride with the plain full-list matrix search");
  // candidates staged in LDS per tile: fp64 records are twice as large and the fp64 kernel needs 112 VGPRs (4 waves / SIMD), so the smaller
  // tile is what lets a fourth block fit the LDS of a CU (9 A headline list: 0.254 -> 0.226 ms, profiles/r03_ab_nl_segfill.log)
  constexpr int TILE = sizeof(T) == 8 ? NL_TILE_F64 : NL_TILE;
  if (!glob->use_tiled) return;
  if (!DUAL && FAST != ((flags & (MI_NL_HALF_FILL | MI_NL_NAIVE_EXPR)) == 0)) return;  // the other instantiation handles this call
  __shared__ float trk[CNF ? TILE : 1], cen_rk[CNF ? NL_CCHUNK : 1];  // scaled covalent radii of the staged candidates / of the chunk's centres
  __shared__ double ccn[CNF ? NL_CCHUNK : 1];                           // running coordination number of each centre of the chunk
  __shared__ int ccnt2[DUAL ? NL_CCHUNK : 1];
  __shared__ T tx[TILE], ty[TILE], tz[TILE];
  __shared__ int tj[TILE];
  // per-candidate image shift: three shorts -- or, in the CN-summing variant, one word of 10-bit fields (its extra LDS arrays must not cost
  // the fifth block per CU: 160 KB / 5 = 32 KB per block)
  __shared__ short tsx[CNF ? 1 : TILE], tsy[CNF ? 1 : TILE], tsz[CNF ? 1 : TILE];
  __shared__ int tsp[CNF ? TILE : 1];
  __shared__ int run_beg[3 * NL_MAXROWS], run_pre[3 * NL_MAXROWS + 1], run_cs[3 * NL_MAXROWS];
  __shared__ int ccnt[NL_CCHUNK];
  // the centre atoms of the current chunk, staged once per chunk: a wave must not LOAD from global memory inside the tile loop, because
  // loads and stores share one in-order counter (vmcnt) on this ISA -- waiting for a 16-byte centre record would first wait for every row
  // store the wave has in flight, once per (tile, pass), and the drain of the row stores to HBM would stop overlapping the distance tests
  __shared__ T cen_x[NL_CCHUNK], cen_y[NL_CCHUNK], cen_z[NL_CCHUNK];
  __shared__ int cen_i[NL_CCHUNK];
  __shared__ short4 cen_w[NL_CCHUNK];
  __shared__ int grp_run[TILE / MI_WAVE];  // run containing the first candidate of each 64-candidate group of the current tile
  __shared__ int grp_shift[TILE / MI_WAVE];  // packed image shift common to all 64 candidates of the group, or NL_MIXED
  // run table before ordering by image (CN-summing variant: it lives in the candidate tile, which is staged only after the table is final)
  __shared__ int u_store[CNF ? 1 : 9 * NL_MAXROWS];
  static_assert(sizeof(T) * TILE >= sizeof(int) * 9 * NL_MAXROWS, "the run table must fit the x array of a tile");
  int* const u_beg = CNF ? reinterpret_cast<int*>(tx) : u_store;
  int* const u_len = u_beg + 3 * NL_MAXROWS;
  int* const u_cs = u_beg + 6 * NL_MAXROWS;
  const int tid = threadIdx.x, lane = tid & (MI_WAVE - 1), wave = tid / MI_WAVE;
  const int total_cells = glob->total_cells;
  const bool anyw = glob->any_wrap != 0;
  const bool half = (flags & MI_NL_HALF_FILL) != 0, naive = (flags & MI_NL_NAIVE_EXPR) != 0;
  const unsigned long long lt = (1ull << lane) - 1ull;
  __shared__ int next_cell;
  NlGlobal* gw = const_cast<NlGlobal*>(glob);
  for (;;) {
    // cells differ in cost (occupancy, boundary images): hand them out one at a time instead of a fixed stride
    __syncthreads();  // the previous cell is fully consumed (run table, tiles, next_cell)
    if (tid == 0) next_cell = atomicAdd(&gw->work[MODE], 1);
    __syncthreads();
    const int c = next_cell;
    if (c >= total_cells) break;
    const int c_beg = cell_start[c], n_c = cell_start[c + 1] - c_beg;
    if (n_c == 0) continue;  // block-uniform
    int s = 0;
    {  // system owning this cell (cell_off is increasing)
      int lo = 0, hi = B - 1;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (sys[mid].cell_off <= c) lo = mid; else hi = mid - 1; }
      s = lo;
    }
    const NlSys<T>* S = sys + s;
    const int nx = S->cpd[0], ny = S->cpd[1], nz = S->cpd[2];
    const int Rx = S->R[0], Ry = S->R[1], Rz = S->R[2];
    const bool pbx = S->pbc[0] != 0, pby = S->pbc[1] != 0, pbz = S->pbc[2] != 0;
    const int coff = S->cell_off;
    const int nr0 = S->nrange[0], nr1 = S->nrange[1], nr2 = S->nrange[2];
    T cm[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) cm[k] = S->cell[k];
    const int local = c - coff;
    const int cx = local % nx, cyz = local / nx, cy = cyz % ny, cz = cyz / ny;
    const int nyr = 2 * Ry + 1, nrows = (2 * Rz + 1) * nyr;
    __syncthreads();  // previous cell fully consumed before the run table is overwritten
    // ---- run table: one thread per (dy,dz) row, up to three x-images each
    if (tid < nrows) {
      const int dzr = tid / nyr, dz = dzr - Rz, dy = tid - dzr * nyr - Ry;
      const int tzc = cz + dz, tyc = cy + dy;
      bool ok = (pbz || (tzc >= 0 && tzc < nz)) && (pby || (tyc >= 0 && tyc < ny));
      int csz = 0, wz = tzc, csy = 0, wy = tyc;
      if (ok) {
        while (wz < 0) { wz += nz; --csz; }
        while (wz >= nz) { wz -= nz; ++csz; }
        while (wy < 0) { wy += ny; --csy; }
        while (wy >= ny) { wy -= ny; ++csy; }
      }
      int dxm = Rx;
      if (S->prune) { dxm = S->dxlim[dz < 0 ? -dz : dz][dy < 0 ? -dy : dy]; ok = ok && dxm >= 0; }
      int xlo = cx - dxm, xhi = cx + dxm;
      if (!pbx) { xlo = xlo < 0 ? 0 : xlo; xhi = xhi > nx - 1 ? nx - 1 : xhi; }
      const int rowbase = coff + nx * (wy + ny * wz);
      for (int g = 0; g < 3; ++g) {
        const int img0 = (g - 1) * nx;
        const int xa = (xlo > img0 ? xlo : img0), xb = (xhi < img0 + nx - 1 ? xhi : img0 + nx - 1);
        int b = 0, len = 0;
        if (ok && xa <= xb && (pbx || g == 1)) { b = cell_start[rowbase + xa - img0]; len = cell_start[rowbase + xb - img0 + 1] - b; }
        u_beg[3 * tid + g] = b;
        u_len[3 * tid + g] = len;
        u_cs[3 * tid + g] = ((g - 1) & 0x3ff) | ((csy & 0x3ff) << 10) | ((csz & 0x3ff) << 20);
      }
    }
    __syncthreads();
    // Runs are ordered by periodic image (the un-shifted image first, each image's runs in their row order): groups of 64
    // candidates then carry one common shift almost everywhere, S.cell becomes a per-group constant and the per-candidate
    // shift is only consulted for the few groups that straddle two images.  Any fixed candidate order is a valid row order.
    for (int r = tid; r < 3 * nrows; r += 256) {
      const int key = u_cs[r];
      int rank = 0;
      for (int q = 0; q < 3 * nrows; ++q) {
        const int kq = u_cs[q];
        rank += ((unsigned)kq < (unsigned)key) || (kq == key && q < r);
      }
      run_beg[rank] = u_beg[r];
      run_pre[rank + 1] = u_len[r];  // lengths first, turned into an inclusive prefix below
      run_cs[rank] = key;
    }
    __syncthreads();
    if (wave == 0) nl_prefix_runs(run_pre, 3 * nrows, lane);
    __syncthreads();
    const int nruns = 3 * nrows, total = run_pre[nruns];
    for (int cbase = 0; cbase < n_c; cbase += NL_CCHUNK) {
      const int cend = (cbase + NL_CCHUNK < n_c) ? cbase + NL_CCHUNK : n_c;
      if (tid < NL_CCHUNK) {
        ccnt[tid] = 0;
        if (DUAL) ccnt2[tid] = 0;
        if (cbase + tid < cend) {
          const auto cr = spos[c_beg + cbase + tid];
          cen_x[tid] = cr.x; cen_y[tid] = cr.y; cen_z[tid] = cr.z;
          cen_i[tid] = idx_of(cr);
          cen_w[tid] = anyw ? swrap[c_beg + cbase + tid] : make_short4(0, 0, 0, 0);
          if (CNF) { cen_rk[tid] = CN.rk[c_beg + cbase + tid]; ccn[tid] = 0.0; }
        }
      }
      for (int tile0 = 0; tile0 < total; tile0 += TILE) {
        const int tile_n = (total - tile0 < TILE) ? total - tile0 : TILE;
        __syncthreads();  // the previous tile has been consumed (and ccnt initialised)
        if (tid < TILE / MI_WAVE && tile0 + tid * MI_WAVE < total) {
          const int v = tile0 + tid * MI_WAVE;
          int lo = 0, hi = nruns - 1;  // last run with run_pre[run] <= v (zero-length runs share a prefix value: take the last)
          while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (run_pre[mid] <= v) lo = mid; else hi = mid - 1; }
          grp_run[tid] = lo;
        }
        __syncthreads();
        for (int t = tid; t < tile_n; t += 256) {
          const int v = tile0 + t;
          int lo = grp_run[t / MI_WAVE];  // 64 consecutive candidates span a few runs at most: short linear advance
          while (lo + 1 < nruns && run_pre[lo + 1] <= v) ++lo;
          const int q = run_beg[lo] + (v - run_pre[lo]);
          const auto rec = spos[q];
          tx[t] = rec.x; ty[t] = rec.y; tz[t] = rec.z;
          tj[t] = idx_of(rec);
          if (CNF) trk[t] = CN.rk[q];
          const int cs = run_cs[lo];
          int sx = (cs << 22) >> 22, sy = (cs << 12) >> 22, sz = (cs << 2) >> 22;  // sign-extend the 10-bit fields
          if (anyw) {
            const short4 wj = swrap[q];
            if (pbx) sx -= wj.x;
            if (pby) sy -= wj.y;
            if (pbz) sz -= wj.z;
          }
          if (CNF) tsp[t] = (sx & 0x3ff) | ((sy & 0x3ff) << 10) | ((sz & 0x3ff) << 20);
          else { tsx[t] = (short)sx; tsy[t] = (short)sy; tsz[t] = (short)sz; }
          // t = tid + 256 k: one wave stages exactly one 64-candidate group per trip (lanes past tile_n have left the loop)
          const int cs0 = __builtin_amdgcn_readfirstlane(cs);
          const bool same = __all(cs == cs0) && !anyw;  // per-atom wraps make the shift lane-dependent
          if (lane == 0) grp_shift[t / MI_WAVE] = same ? cs0 : NL_MIXED;
        }
        __syncthreads();
        // NC centre atoms per wave at a time share every candidate fetched from LDS (one set of ds_reads and address math per
        // 64 candidates instead of one per centre); groups of 64 candidates whose image shift is zero for the whole group
        // (the interior of the box) skip the S.cell term -- adding an exact zero cannot change the reference expression.
        constexpr int NC = 2;
        for (int ci0 = cbase + wave; ci0 < cend; ci0 += 4 * NC) {
          T ccx[NC], ccy[NC], ccz[NC];
          int ii[NC], cap_row[NC], cnt[NC], cnt2[NC];
          float rki[NC], cnp[NC];  // CNF: the centre's scaled radius (wave-uniform) and this pass's lane partial of its coordination number
          long long out_base[NC];
          short4 wi[NC];
          const int nc = (cend - ci0 + 3) / 4 < NC ? (cend - ci0 + 3) / 4 : NC;  // centres of this wave in this pass (uniform)
#pragma unroll
          for (int u = 0; u < NC; ++u) {
            const bool live = u < nc;
            const int ci = live ? ci0 + 4 * u : ci0;
            // a missing centre (last pass of a chunk) gets NaN coordinates: every `d2 < rc2` is false, so the candidate loop
            // needs no per-centre predicate
            ccx[u] = live ? cen_x[ci - cbase] : (T)NAN; ccy[u] = cen_y[ci - cbase]; ccz[u] = cen_z[ci - cbase];
            if (CNF) { ccx[u] = nl_uni(ccx[u]); ccy[u] = nl_uni(ccy[u]); ccz[u] = nl_uni(ccz[u]); }  // centre coordinates in SGPRs: room for the CN partials at 5 waves / SIMD
            ii[u] = __builtin_amdgcn_readfirstlane(cen_i[ci - cbase]);
            wi[u] = cen_w[ci - cbase];
            if (MODE == MI_NL_MODE_CSR) { out_base[u] = ptr[ii[u]]; cap_row[u] = ptr[ii[u] + 1] - ptr[ii[u]]; }
            else { out_base[u] = (long long)ii[u] * M; cap_row[u] = M; }
            cnt[u] = ccnt[ci - cbase];
            cnt2[u] = DUAL ? ccnt2[ci - cbase] : 0;
            if (CNF) { rki[u] = nl_uni(cen_rk[ci - cbase]); cnp[u] = 0.0f; }
          }
          const bool ortho = S->prune != 0;  // orthorhombic: S.cell has one non-zero term per component (adding exact zeros changes nothing)
          // Row bases as wave-uniform pointers and slots as 32-bit byte offsets: the hit stores become `global_store v_off, v_data, s[base]`
          // (one shift / one multiply per store instead of a 64-bit address per lane); the wave-level prefix count goes through
          // v_mbcnt (two instructions) and the hit mask straight from the compare (no 0/1 round trip).  The per-hit part of the
          // loop went from ~14 to ~7 vector instructions (profiles/r03_ab_nl_valu.log).
          int* row_j[NC];
          NlInt3* row_s[NC];
          unsigned* row_p[NC];  // packed companion row (matrix mode, optional)
#pragma unroll
          for (int u = 0; u < NC; ++u) {
            const long long ob = nl_uniform64(out_base[u]);
            row_p[u] = (MODE == MI_NL_MODE_MATRIX && !DUAL && K.words) ? K.words + ob : nullptr;
            if (MODE == MI_NL_MODE_MATRIX) { row_j[u] = nm + ob; row_s[u] = nsh ? reinterpret_cast<NlInt3*>(nsh) + ob : nullptr; }
            else if (MODE == MI_NL_MODE_CSR) { row_j[u] = list_ij + P + ob; row_s[u] = list_sh ? reinterpret_cast<NlInt3*>(list_sh) + ob : nullptr; }
            else { row_j[u] = nullptr; row_s[u] = nullptr; }
          }
          constexpr bool PREZERO = NL_PREZERO_SHIFTS && MODE == MI_NL_MODE_MATRIX && !DUAL;
          const bool prezero = PREZERO && nsh && !(flags & MI_NL_NO_PAD);  // kernel-uniform
          if (PREZERO && prezero && tile0 == 0) {
#pragma unroll
            for (int u = 0; u < NC; ++u)
              if (u < nc) wave_fill<NL_PREZERO_NT != 0>(nsh, out_base[u] * 3, (out_base[u] + M) * 3, 0, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the zeros are at the L2 before any shift of these rows is stored
          }
          auto emit_m = [&](int u, unsigned long long mask, bool hit, int j, int Sx, int Sy, int Sz, unsigned code, bool zero_shift = false) {
            if (MODE == MI_NL_MODE_COUNT) { cnt[u] += __popcll(mask); return; }
            if (mask) {
              // slot = entries already in the row + hits in lower lanes
              const unsigned slot = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, (unsigned)cnt[u]));
              if (hit && slot < (unsigned)cap_row[u]) {
                // (plain stores: the hits of a row arrive in short runs that the L2 merges into full lines; non-temporal stores here
                // cost +20-60 % on the headline list, profiles/r02_ab_nt.log.  CSR: the source row (constant i) is written in bulk
                // when the centre is finished)
                *reinterpret_cast<int*>(reinterpret_cast<char*>(row_j[u]) + slot * 4u) = j;
                if (row_s[u] && !(PREZERO && prezero && zero_shift))
                  *reinterpret_cast<NlInt3*>(reinterpret_cast<char*>(row_s[u]) + __umul24(slot, 12u)) = NlInt3{Sx, Sy, Sz};
                // (plain store like the two above: a non-temporal companion store was measured at +0.25 - 0.3 ms on the headline fill,
                // profiles/r05_ab_companion_nt_store.log)
                if (MODE == MI_NL_MODE_MATRIX && !DUAL && row_p[u]) *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(row_p[u]) + slot * 4u) = (unsigned)j | code;
              }
              cnt[u] += __popcll(mask);
            }
          };
          auto emit = [&](int u, bool hit, int j, int Sx, int Sy, int Sz, unsigned code) { emit_m(u, __builtin_amdgcn_ballot_w64(hit), hit, j, Sx, Sy, Sz, code); };
          // software-pipelined by one group: the LDS reads of group g+1 are issued before group g is tested, so the wave
          // does not sit out an LDS round trip at the top of every trip
          struct Cand { int j, tt, zg; T x, y, z; };
          auto fetch = [&](int t0) {
            Cand c;
            const int t = t0 + lane;
            const bool in = t < tile_n;
            c.tt = in ? t : 0;
            c.j = tj[c.tt];
            c.x = in ? tx[c.tt] : (T)NAN;  // lanes past the tile: NaN -> never a hit
            c.y = ty[c.tt]; c.z = tz[c.tt];
            c.zg = grp_shift[(t0 < tile_n ? t0 : 0) / MI_WAVE];
            return c;
          };
          Cand nxt = fetch(0);
          for (int t0 = 0; t0 < tile_n; t0 += MI_WAVE) {
            const Cand cur = nxt;
            nxt = fetch(t0 + MI_WAVE);
            const int j = cur.j, tt = cur.tt;
            // (read where it is used, not a trip ahead like the coordinates: one loop-carried register less, and the term is only formed
            // after d^2 -- the LDS round trip hides behind the distance arithmetic)
            const float rkj = CNF ? trk[tt] : 0.0f;
            const T cjx = cur.x, cjy = cur.y, cjz = cur.z;
            const int gs = __builtin_amdgcn_readfirstlane(cur.zg);
            if (FAST && gs == 0) {  // un-shifted image: adding S.cell = 0 cannot change the reference expression
#pragma unroll
              for (int u = 0; u < NC; ++u) {
                const T dr0 = cjx - ccx[u], dr1 = cjy - ccy[u], dr2 = cjz - ccz[u];
                const T d2 = dr0 * dr0 + dr1 * dr1 + dr2 * dr2;
                const bool in = d2 < rc2, other = j != ii[u];
                if (CNF) cnp[u] += (in & other & ((float)d2 >= 1e-24f)) ? nl_cn_term((float)d2, rki[u], rkj, CN.K) : 0.0f;  // (r < 1e-12: no term, dftd3.py:597)
                emit_m(u, __builtin_amdgcn_ballot_w64(in) & __builtin_amdgcn_ballot_w64(other), in & other, j, 0, 0, 0, NL_PK_ZERO_SHIFT, true);
              }
            } else if (FAST && gs != NL_MIXED) {  // one common non-zero shift: S.cell once per group, no self-pair possible
              const int Sx = (gs << 22) >> 22, Sy = (gs << 12) >> 22, Sz = (gs << 2) >> 22;
              T cart[3];
              if (ortho) { cart[0] = cm[0] * (T)Sx; cart[1] = cm[4] * (T)Sy; cart[2] = cm[8] * (T)Sz; }
              else { const T fs[3] = {(T)Sx, (T)Sy, (T)Sz}; rowvec_mat3(fs, cm, cart); }
              bool bad;  // group-uniform: a shift that does not fit the packed word makes the companion unusable
              const unsigned code = nl_pk_code(Sx, Sy, Sz, bad);
#pragma unroll
              for (int u = 0; u < NC; ++u) {
                const T dr0 = (cjx - ccx[u]) + cart[0], dr1 = (cjy - ccy[u]) + cart[1], dr2 = (cjz - ccz[u]) + cart[2];
                const T d2 = dr0 * dr0 + dr1 * dr1 + dr2 * dr2;
                const bool h = d2 < rc2;
                if (CNF) cnp[u] += (h & ((float)d2 >= 1e-24f)) ? nl_cn_term((float)d2, rki[u], rkj, CN.K) : 0.0f;
                if (MODE == MI_NL_MODE_MATRIX && bad && row_p[u] && h) *K.flag = 1;  // benign race: every writer stores 1
                emit(u, h, j, Sx, Sy, Sz, code);
              }
            } else {
              int Sx0, Sy0, Sz0;
              if (CNF) { const int w = tsp[tt]; Sx0 = (w << 22) >> 22; Sy0 = (w << 12) >> 22; Sz0 = (w << 2) >> 22; }
              else { Sx0 = tsx[tt]; Sy0 = tsy[tt]; Sz0 = tsz[tt]; }
#pragma unroll
              for (int u = 0; u < NC; ++u) {
                int Sx = Sx0, Sy = Sy0, Sz = Sz0;
                if (anyw) { if (pbx) Sx += wi[u].x; if (pby) Sy += wi[u].y; if (pbz) Sz += wi[u].z; }
                T cart[3];
                if (ortho) { cart[0] = cm[0] * (T)Sx; cart[1] = cm[4] * (T)Sy; cart[2] = cm[8] * (T)Sz; }
                else { const T fs[3] = {(T)Sx, (T)Sy, (T)Sz}; rowvec_mat3(fs, cm, cart); }
                bool h;
                if (FAST) {
                  const T dr0 = (cjx - ccx[u]) + cart[0], dr1 = (cjy - ccy[u]) + cart[1], dr2 = (cjz - ccz[u]) + cart[2];
                  const T d2 = dr0 * dr0 + dr1 * dr1 + dr2 * dr2;
                  h = (d2 < rc2) & !((j == ii[u]) & ((Sx | Sy | Sz) == 0));
                  if (CNF) cnp[u] += (h & ((float)d2 >= 1e-24f)) ? nl_cn_term((float)d2, rki[u], rkj, CN.K) : 0.0f;
                } else {
                  T d2;
                  h = nl_pair_hit<T>(ccx[u], ccy[u], ccz[u], ii[u], cjx, cjy, cjz, j, Sx, Sy, Sz, cart, rc2, naive, half, nr0, nr1, nr2, &d2);
                  if (DUAL) {  // the short list, nested in the long one (same walk, same image range)
                    const bool h2 = h && (d2 < D.rc2);
                    const unsigned long long m2 = __ballot(h2);
                    if (m2) {
                      const int slot = cnt2[u] + __popcll(m2 & lt);
                      if (h2 && slot < D.M) {
                        const long long o = (long long)ii[u] * D.M + slot;
                        D.nm[o] = j;
                        if (D.nsh) reinterpret_cast<NlInt3*>(D.nsh)[o] = NlInt3{Sx, Sy, Sz};
                      }
                      cnt2[u] += __popcll(m2);
                    }
                  }
                }
                bool bad;
                const unsigned code = nl_pk_code(Sx, Sy, Sz, bad);
                if (MODE == MI_NL_MODE_MATRIX && !DUAL && bad && row_p[u] && h) *K.flag = 1;
                emit(u, h, j, Sx, Sy, Sz, code);
              }
            }
          }
          if (CNF) {  // <= TILE / 64 fp32 terms per lane and pass; across lanes and passes the sum is fp64 (one wave owns a centre: no race on ccn)
#pragma unroll
            for (int u = 0; u < NC; ++u) {
              const double part = wave_sum((double)cnp[u]);
              if (lane == 0 && u < nc) ccn[ci0 + 4 * u - cbase] += part;
            }
          }
          if (lane == 0) {
#pragma unroll
            for (int u = 0; u < NC; ++u) if (u < nc) { ccnt[ci0 + 4 * u - cbase] = cnt[u]; if (DUAL) ccnt2[ci0 + 4 * u - cbase] = cnt2[u]; }
          }
        }
      }
      __syncthreads();
      // counts and (matrix mode) padding of this chunk of centres
      for (int ci = cbase + wave; ci < cend; ci += 4) {
        const int i = __builtin_amdgcn_readfirstlane(cen_i[ci - cbase]);
        const int cnt = ccnt[ci - cbase];
        if (MODE != MI_NL_MODE_CSR) { if (lane == 0) num[i] = cnt; }
        if (CNF && lane == 0) {
          CN.cn[i] = (float)ccn[ci - cbase];
          if (cnt > M) *CN.flag = 1;  // entries past the row width are counted but not stored: the list's own sum is smaller
        }
        if (MODE == MI_NL_MODE_CSR) {
          const long long rb = ptr[i];
          const int room = ptr[i + 1] - ptr[i];
          wave_fill(list_ij, rb, rb + (cnt < room ? cnt : room), i, lane);
        }
        if (MODE == MI_NL_MODE_MATRIX && !(flags & MI_NL_NO_PAD)) {
          const long long out_base = (long long)i * M;
          const int used = cnt < M ? cnt : M;
          wave_fill(nm, out_base + used, out_base + M, fill_value, lane);
          if (nsh && !(NL_PREZERO_SHIFTS && !DUAL)) wave_fill(nsh, (out_base + used) * 3, (out_base + M) * 3, 0, lane);
          if (!DUAL && K.words) wave_fill(reinterpret_cast<int*>(K.words), out_base + used, out_base + M, (int)NL_PK_PAD, lane);
        }
        if (DUAL) {
          const int c2 = ccnt2[ci - cbase];
          if (lane == 0) D.num[i] = c2;
          if (!(flags & MI_NL_NO_PAD)) {
            const long long ob = (long long)i * D.M;
            const int used = c2 < D.M ? c2 : D.M;
            wave_fill(D.nm, ob + used, ob + D.M, fill_value, lane);
            if (D.nsh) wave_fill(D.nsh, (ob + used) * 3, (ob + D.M) * 3, 0, lane);
          }
        }
      }
      __syncthreads();
    }
  }
  if (tid == 0 && atomicAdd(&gw->done[MODE], 1) == (int)gridDim.x - 1) {  // last block out re-arms the counters
    gw->work[MODE] = 0;
    gw->done[MODE] = 0;
  }
}

// Blocks of 4 waves, one centre per wave.  Both query kernels are launched for every search and the device-side grid description decides which
// one works; the idle one costs 6 - 12 us of empty blocks in front of a tiled search (25 000 blocks for the headline box).  Three cheaper shapes
// of this kernel were measured in round 6 and rejected because of what they do when it IS the working kernel (sparse cells: BASELINE config 2,
// the reference's cell_list benchmark rows): a fixed grid of 2048 blocks with a stride loop over the centres (idle launch 3 us; config 2
// 0.136 -> 0.154 ms); 16-wave blocks (idle launch 3 us; the reference's 524 288-atom row 1.02 -> 1.27 ms: a block's wave slots are only handed
// back when its slowest wave is done); and even the loop form launched with the old grid (one trip per wave) costs the reference's rows 5 - 6 %
// (0.961 -> 1.024 ms, same box).  profiles/r06_ab_atom_kernel_grid.log, r06_ab_atom_kernel_block_size.log, r06_ab_atom_kernel_loop_form.log.
template <class T, int MODE, bool DUAL = false, bool CNF = false>
__global__ __launch_bounds__(256) void nl_query_kernel(
    const typename Vec4<T>::type* __restrict__ spos, const short4* __restrict__ swrap, const int* __restrict__ keys_sorted,
    const int* __restrict__ cell_start, const int* __restrict__ batch_idx, const NlSys<T>* __restrict__ sys,
    const NlGlobal* __restrict__ glob, int N, T rc2, int flags, int* __restrict__ nm, int* __restrict__ nsh, int* __restrict__ num,
    int M, int fill_value, const int* __restrict__ ptr, int* __restrict__ list_ij, int* __restrict__ list_sh, long long P, NlSecond<T> D,
    NlPacked K, NlCn CN = NlCn{nullptr, nullptr, nullptr, 0.0f}) {
  static_assert(!DUAL || MODE == MI_NL_MODE_MATRIX, "the dual-cutoff sweep fills two padded matrices");
  static_assert(!CNF || (MODE == MI_NL_MODE_MATRIX && !DUAL), "coordination numbers ride with the plain matrix search");
  if (glob->use_tiled) return;  // dense cells: the block-per-cell LDS-tiled kernel (launched next to this one) does the work
  const int lane = threadIdx.x & (MI_WAVE - 1);
  const int p = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / MI_WAVE) + (threadIdx.x / MI_WAVE));
  if (p >= N) return;
  const auto ci = spos[p];
  const int i = __builtin_amdgcn_readfirstlane(idx_of(ci));
  const T pix = ci.x, piy = ci.y, piz = ci.z;
  const int s = batch_idx ? __builtin_amdgcn_readfirstlane(batch_idx[i]) : 0;
  const NlSys<T>* S = sys + s;
  const int nx = S->cpd[0], ny = S->cpd[1], nz = S->cpd[2];
  const int Rx = S->R[0], Ry = S->R[1], Rz = S->R[2];
  const bool pbx = S->pbc[0] != 0, pby = S->pbc[1] != 0, pbz = S->pbc[2] != 0;
  const int coff = S->cell_off;
  T cm[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) cm[k] = S->cell[k];
  const int local = __builtin_amdgcn_readfirstlane(keys_sorted[p]) - coff;
  const int cx = local % nx, cyz = local / nx, cy = cyz % ny, cz = cyz / ny;
  const bool anyw = glob->any_wrap != 0;
  short4 wi = make_short4(0, 0, 0, 0);
  if (anyw) wi = swrap[p];
  const bool half = (flags & MI_NL_HALF_FILL) != 0;
  const bool naive = (flags & MI_NL_NAIVE_EXPR) != 0;
  const int nr0 = S->nrange[0], nr1 = S->nrange[1], nr2 = S->nrange[2];
  const unsigned long long lt = (1ull << lane) - 1ull;
  long long out_base;
  int cap_row;
  if (MODE == MI_NL_MODE_CSR) { out_base = ptr[i]; cap_row = ptr[i + 1] - ptr[i]; }
  else { out_base = (long long)i * M; cap_row = M; }
  int cnt = 0, cnt2 = 0;
  // CNF: DFT-D3 coordination number of this centre over its hits (NlCn): the wave owns the row, so the lane partials live in registers
  // for the whole walk (fp64, as mi_d3's own pass keeps them)
  const float rki = CNF ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(CN.rk[p]))) : 0.0f;
  double cnacc = 0.0;

  const bool prune = S->prune != 0;

  // one contiguous run of candidates [beg,end) of the sorted order, all with the same cell image shift (sx, csy, csz)
  auto scan_run = [&](int beg, int end, int sx, int csy, int csz) {
    // without per-atom wraps the integer shift (hence S.cell) is the same for the whole run: evaluate it once
    T cart_u[3];
    {
      const T fsu[3] = {(T)sx, (T)csy, (T)csz};
      rowvec_mat3(fsu, cm, cart_u);
    }
    for (int q0 = beg; q0 < end; q0 += MI_WAVE) {
      const int q = q0 + lane;
      bool hit = false, hit2 = false;
      int j = 0, Sx = sx, Sy = csy, Sz = csz;
      if (q < end) {
        const auto cj = spos[q];
        j = idx_of(cj);
        if (anyw) {
          const short4 wj = swrap[q];
          if (pbx) Sx += (int)wi.x - (int)wj.x;
          if (pby) Sy += (int)wi.y - (int)wj.y;
          if (pbz) Sz += (int)wi.z - (int)wj.z;
        }
        T cart[3] = {cart_u[0], cart_u[1], cart_u[2]};
        if (anyw) {
          const T fs[3] = {(T)Sx, (T)Sy, (T)Sz};
          rowvec_mat3(fs, cm, cart);
        }
        T d2;
        hit = nl_pair_hit<T>(pix, piy, piz, i, cj.x, cj.y, cj.z, j, Sx, Sy, Sz, cart, rc2, naive, half, nr0, nr1, nr2, &d2);
        hit2 = DUAL && hit && (d2 < D.rc2);
        if (CNF) cnacc += (hit && (float)d2 >= 1e-24f) ? (double)nl_cn_term((float)d2, rki, CN.rk[q], CN.K) : 0.0;
      }
      if (DUAL) {  // the short list, nested in the long one
        const unsigned long long m2 = __ballot(hit2);
        if (m2) {
          const int slot = cnt2 + __popcll(m2 & lt);
          if (hit2 && slot < D.M) {
            const long long o = (long long)i * D.M + slot;
            D.nm[o] = j;
            if (D.nsh) reinterpret_cast<NlInt3*>(D.nsh)[o] = NlInt3{Sx, Sy, Sz};
          }
          cnt2 += __popcll(m2);
        }
      }
      const unsigned long long mask = __ballot(hit);
      if (mask) {
        const int slot = cnt + __popcll(mask & lt);
        if (MODE == MI_NL_MODE_MATRIX) {
          if (hit && slot < cap_row) {
            nm[out_base + slot] = j;
            if (nsh) reinterpret_cast<NlInt3*>(nsh)[out_base + slot] = NlInt3{Sx, Sy, Sz};
            if (K.words) {  // packed companion (see NlPacked)
              bool bad;
              const unsigned code = nl_pk_code(Sx, Sy, Sz, bad);
              if (bad) *K.flag = 1;
              K.words[out_base + slot] = (unsigned)j | code;
            }
          }
        } else if (MODE == MI_NL_MODE_CSR) {
          if (hit && slot < cap_row) {  // the source row (constant i) is written in bulk at the end
            list_ij[P + out_base + slot] = j;
            if (list_sh) reinterpret_cast<NlInt3*>(list_sh)[out_base + slot] = NlInt3{Sx, Sy, Sz};
          }
        }
        cnt += __popcll(mask);
      }
    }
  };

  if (!pbx || Rx <= nx) {
    // Common case (at most the images -1, 0, +1 along x): the (dy,dz) rows of cells are described by the lanes IN PARALLEL --
    // wrap, pruning look-up and the up-to-six cell_start reads of 64 rows cost one memory latency -- and then consumed one by
    // one through wave-uniform readlanes.  (A serial walk pays two dependent global loads per row before any candidate is seen.)
    const int nyr = 2 * Ry + 1, nrows = (2 * Rz + 1) * nyr;
    for (int rbase = 0; rbase < nrows; rbase += MI_WAVE) {
      const int row = rbase + lane;
      int rb[3] = {0, 0, 0}, re[3] = {0, 0, 0}, cs = 0;
      if (row < nrows) {
        const int dzr = row / nyr, dz = dzr - Rz, dy = row - dzr * nyr - Ry;
        const int tz = cz + dz, ty = cy + dy;
        bool ok = (pbz || (tz >= 0 && tz < nz)) && (pby || (ty >= 0 && ty < ny));
        int csz = 0, wz = tz, csy = 0, wy = ty;
        if (ok) {
          while (wz < 0) { wz += nz; --csz; }
          while (wz >= nz) { wz -= nz; ++csz; }
          while (wy < 0) { wy += ny; --csy; }
          while (wy >= ny) { wy -= ny; ++csy; }
        }
        int dxm = Rx;
        if (prune) { dxm = S->dxlim[dz < 0 ? -dz : dz][dy < 0 ? -dy : dy]; ok = ok && dxm >= 0; }
        if (ok) {
          int xlo = cx - dxm, xhi = cx + dxm;
          if (!pbx) { xlo = xlo < 0 ? 0 : xlo; xhi = xhi > nx - 1 ? nx - 1 : xhi; }
          const int rowbase = coff + nx * (wy + ny * wz);
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            const int img0 = (g - 1) * nx;
            const int xa = (xlo > img0 ? xlo : img0), xb = (xhi < img0 + nx - 1 ? xhi : img0 + nx - 1);
            if (xa <= xb && (pbx || g == 1)) { rb[g] = cell_start[rowbase + xa - img0]; re[g] = cell_start[rowbase + xb - img0 + 1]; }
          }
          cs = (csy & 0xffff) | (csz << 16);
        }
      }
      const int rmax = nrows - rbase < MI_WAVE ? nrows - rbase : MI_WAVE;
      for (int r = 0; r < rmax; ++r) {
        const int csr = __builtin_amdgcn_readlane(cs, r);
        const int csy = (int)(short)(csr & 0xffff), csz = csr >> 16;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const int beg = __builtin_amdgcn_readlane(rb[g], r), end = __builtin_amdgcn_readlane(re[g], r);
          if (beg < end) scan_run(beg, end, g - 1, csy, csz);
        }
      }
    }
  } else {
    // tiny periodic boxes (search radius larger than the box along x: many images per row): plain serial walk
    for (int dz = -Rz; dz <= Rz; ++dz) {
      const int tz = cz + dz;
      if (!pbz && (tz < 0 || tz >= nz)) continue;
      int csz = 0, wz = tz;
      while (wz < 0) { wz += nz; --csz; }
      while (wz >= nz) { wz -= nz; ++csz; }
      for (int dy = -Ry; dy <= Ry; ++dy) {
        const int ty = cy + dy;
        if (!pby && (ty < 0 || ty >= ny)) continue;
        int csy = 0, wy = ty;
        while (wy < 0) { wy += ny; --csy; }
        while (wy >= ny) { wy -= ny; ++csy; }
        int dxm = Rx;
        if (prune) {
          dxm = S->dxlim[dz < 0 ? -dz : dz][dy < 0 ? -dy : dy];
          if (dxm < 0) continue;
        }
        int xlo = cx - dxm, xhi = cx + dxm, sx_lo = 0, sx_hi = 0;
        for (int t = xlo; t < 0; t += nx) --sx_lo;
        for (int t = xhi; t >= nx; t -= nx) ++sx_hi;
        const int rowbase = coff + nx * (wy + ny * wz);
        for (int sx = sx_lo; sx <= sx_hi; ++sx) {
          const int img0 = sx * nx;
          const int xa = (xlo > img0 ? xlo : img0) - img0;
          const int xb = (xhi < img0 + nx - 1 ? xhi : img0 + nx - 1) - img0;
          scan_run(cell_start[rowbase + xa], cell_start[rowbase + xb + 1], sx, csy, csz);
        }
      }
    }
  }
  if (MODE != MI_NL_MODE_CSR) {
    if (lane == 0) num[i] = cnt;
  } else {
    wave_fill(list_ij, out_base, out_base + (cnt < cap_row ? cnt : cap_row), i, lane);
  }
  if (CNF) {
    cnacc = wave_sum(cnacc);
    if (lane == 0) {
      CN.cn[i] = (float)cnacc;
      if (cnt > M || half) *CN.flag = 1;  // truncated rows / half lists: the stored list's own sum differs
    }
  }
  if (MODE == MI_NL_MODE_MATRIX && !(flags & MI_NL_NO_PAD)) {
    // the row owner also writes the padding (replaces torch.full + zeros of cell_list.py:1358-1373)
    const int used = cnt < M ? cnt : M;
    wave_fill(nm, out_base + used, out_base + M, fill_value, lane);
    if (nsh) wave_fill(nsh, (out_base + used) * 3, (out_base + M) * 3, 0, lane);
    if (K.words) wave_fill(reinterpret_cast<int*>(K.words), out_base + used, out_base + M, (int)NL_PK_PAD, lane);
  }
  if (DUAL) {
    if (lane == 0) D.num[i] = cnt2;
    if (!(flags & MI_NL_NO_PAD)) {
      const long long ob = (long long)i * D.M;
      const int used = cnt2 < D.M ? cnt2 : D.M;
      wave_fill(D.nm, ob + used, ob + D.M, fill_value, lane);
      if (D.nsh) wave_fill(D.nsh, (ob + used) * 3, (ob + D.M) * 3, 0, lane);
    }
  }
}

// padded matrix -> COO (neighbor_utils.py:428-438: mask = matrix != fill_value, row-major order)
__global__ __launch_bounds__(256) void nl_matrix_to_coo_kernel(const int* __restrict__ nm, const int* __restrict__ nsh,
                                                               const int* __restrict__ ptr, int N, int M, int fill_value,
                                                               int* __restrict__ ij, int* __restrict__ sh, long long P) {
  const int lane = threadIdx.x & (MI_WAVE - 1);
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / MI_WAVE) + (threadIdx.x / MI_WAVE));
  if (i >= N) return;
  const long long base = ptr[i];
  const int room = ptr[i + 1] - ptr[i];
  const unsigned long long lt = (1ull << lane) - 1ull;
  int cnt = 0;
  for (int t0 = 0; t0 < M; t0 += MI_WAVE) {
    const int t = t0 + lane;
    int j = fill_value;
    if (t < M) j = nm[(long long)i * M + t];
    const bool keep = (t < M) && (j != fill_value);
    const unsigned long long mask = __ballot(keep);
    const int slot = cnt + __popcll(mask & lt);
    if (keep && slot < room) {
      ij[base + slot] = i;
      ij[P + base + slot] = j;
      if (sh) {
        const int* sp = nsh + ((long long)i * M + t) * 3;
        int* dp = sh + (base + slot) * 3;
        dp[0] = sp[0]; dp[1] = sp[1]; dp[2] = sp[2];
      }
    }
    cnt += __popcll(mask);
  }
}

// ---- reference-format sizing (cell_list.py:35-99, batch_cell_list.py:36-99) -----------------------
template <class T>
__device__ void ref_cells_per_dim(const T* cell9, const uint8_t* pbc3, T cutoff, int cpd[3], int* radius) {
  T inv[9];
  inverse3(cell9, inv);
  for (int d = 0; d < 3; ++d) {
    T col[3] = {inv[d], inv[3 + d], inv[6 + d]};
    T face = T(1) / sqrt(col[0] * col[0] + col[1] * col[1] + col[2] * col[2]);
    int c = (int)(face / cutoff);
    cpd[d] = c > 1 ? c : 1;
    if (radius) radius[d] = (cpd[d] == 1 && !pbc3[d]) ? 0 : (int)ceil(cutoff * (T)cpd[d] / face);
  }
}
template <class T>
__global__ void nl_estimate_sizes_kernel(const T* __restrict__ cell, const uint8_t* __restrict__ pbc, int B, T cutoff, int max_nbins,
                                         int* __restrict__ ncells, int* __restrict__ radius) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= B) return;
  int cpd[3];
  ref_cells_per_dim(cell + 9 * (size_t)s, pbc + 3 * (size_t)s, cutoff, cpd, radius + 3 * (size_t)s);
  long long tot = (long long)cpd[0] * cpd[1] * cpd[2];
  while (tot > max_nbins) {
    for (int d = 0; d < 3; ++d) cpd[d] = cpd[d] / 2 > 1 ? cpd[d] / 2 : 1;
    tot = (long long)cpd[0] * cpd[1] * cpd[2];
  }
  ncells[s] = (int)tot;
}

// reference-format cache: bin size (cell_list.py:102-163 / batch_cell_list.py:103-177) + per-atom cell/shift
template <class T>
__global__ void nl_cache_binsize_kernel(const T* __restrict__ cell, const uint8_t* __restrict__ pbc, int B, T cutoff, int max_total_cells,
                                        int batched, int* __restrict__ cpd_out, int* __restrict__ cell_off /*[B+1]*/) {
  for (int s = threadIdx.x; s < B; s += blockDim.x) {
    int cpd[3];
    ref_cells_per_dim<T>(cell + 9 * (size_t)s, pbc + 3 * (size_t)s, cutoff, cpd, nullptr);
    long long tot = (long long)cpd[0] * cpd[1] * cpd[2];
    const long long mult = batched ? B : 1;
    while (tot * mult > max_total_cells) {
      for (int d = 0; d < 3; ++d) cpd[d] = cpd[d] / 2 > 1 ? cpd[d] / 2 : 1;
      tot = (long long)cpd[0] * cpd[1] * cpd[2];
    }
    for (int d = 0; d < 3; ++d) cpd_out[3 * (size_t)s + d] = cpd[d];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int off = 0;
    for (int s = 0; s < B; ++s) { cell_off[s] = off; off += cpd_out[3 * s] * cpd_out[3 * s + 1] * cpd_out[3 * s + 2]; }
    cell_off[B] = off;
  }
}
template <class T>
__global__ void nl_cache_assign_kernel(const T* __restrict__ pos, const T* __restrict__ cell, const uint8_t* __restrict__ pbc,
                                       const int* __restrict__ batch_idx, int N, const int* __restrict__ cpd, const int* __restrict__ cell_off,
                                       int* __restrict__ keys, int* __restrict__ counts, int* __restrict__ atom_shift, int* __restrict__ atom_cell) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < N;
  int key = 0;
  if (in) {
    const int s = batch_idx ? batch_idx[i] : 0;
    T inv[9];
    inverse3(cell + 9 * (size_t)s, inv);
    T p[3] = {pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2]}, frac[3];
    rowvec_mat3(p, inv, frac);
    int c[3];
    for (int d = 0; d < 3; ++d) {
      const int n = cpd[3 * s + d];
      int v = (int)floor(frac[d] * (T)n), w = 0;
      if (pbc[3 * s + d]) floor_divmod(v, n, w, c[d]);
      else c[d] = v < 0 ? 0 : (v >= n ? n - 1 : v);
      atom_shift[3 * (size_t)i + d] = w;
      atom_cell[3 * (size_t)i + d] = c[d];
    }
    key = cell_off[s] + c[0] + cpd[3 * s] * (c[1] + cpd[3 * s + 1] * c[2]);
    keys[i] = key;
  }
  bs_wave_add<false>(counts, key, in);  // atoms_per_cell_count, as the reference's count kernel does (cell_list.py:166-276)
}
// The reference-format cache is the same counting sort written into the caller's arrays: atoms_per_cell_count by atomics,
// cell_atom_start_indices = its exclusive cumsum (cell_list.py:869-871), cell_atom_list by a scatter through the start array used as
// a running cursor, then ranked by atom index inside each cell (the reference's order is whatever its atomics produce; ascending
// here) and the cursor rewound.
__global__ void nl_cache_add_offsets_kernel(int* __restrict__ starts, const int* __restrict__ block_off, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) starts[c] += block_off[c / BS_CHUNK];
}
__global__ void nl_cache_scatter_kernel(const int* __restrict__ keys, int N, int* __restrict__ cursor, int* __restrict__ ids) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < N;
  const int slot = bs_wave_add<true>(cursor, in ? keys[i] : 0, in);
  if (in) ids[slot] = i;
}
__global__ void nl_cache_rank_kernel(const int* __restrict__ ids, const int* __restrict__ keys, const int* __restrict__ cursor_end,
                                     const int* __restrict__ counts, int N, int* __restrict__ cell_atoms) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const int i = ids[p], key = keys[i];
  const int e = cursor_end[key], b = e - counts[key];
  int rank = 0;
  for (int q = b; q < e; ++q) rank += ids[q] < i ? 1 : 0;
  cell_atoms[b + rank] = i;
}
__global__ void nl_cache_rewind_kernel(int* __restrict__ starts, const int* __restrict__ counts, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) starts[c] -= counts[c];
}

// ---- bounding boxes of non-periodic systems: the "cell" a free-space search bins in (no lattice is given) ---------------------
// order-preserving unsigned keys so min / max run on integer atomics
__device__ __forceinline__ unsigned long long nl_key(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double nl_unkey(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}
template <class T>
__global__ void nl_bbox_kernel(const T* __restrict__ pos, const int* __restrict__ batch_idx, int N, unsigned long long* __restrict__ box /*[B][6]*/) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < N;
  const int s = in ? (batch_idx ? batch_idx[i] : 0) : -1;
  double v[3] = {0, 0, 0};
  if (in) { v[0] = pos[3 * (size_t)i]; v[1] = pos[3 * (size_t)i + 1]; v[2] = pos[3 * (size_t)i + 2]; }
  const int s0 = __shfl(s, 0, MI_WAVE);
  if (__all(!in || s == s0) && s0 >= 0) {  // the usual case: a wave lies inside one system -> one atomic pair per axis per wave
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      double lo = in ? v[d] : INFINITY, hi = in ? v[d] : -INFINITY;
      for (int o = MI_WAVE / 2; o; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o, MI_WAVE)); hi = fmax(hi, __shfl_xor(hi, o, MI_WAVE)); }
      if ((threadIdx.x & (MI_WAVE - 1)) == 0) { atomicMin(&box[6 * (size_t)s0 + d], nl_key(lo)); atomicMax(&box[6 * (size_t)s0 + 3 + d], nl_key(hi)); }
    }
  } else if (in) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { atomicMin(&box[6 * (size_t)s + d], nl_key(v[d])); atomicMax(&box[6 * (size_t)s + 3 + d], nl_key(v[d])); }
  }
}
// cell = diag(max(hi - lo, 1) * 1.001), origin = lo (an empty system keeps a unit box at the origin)
template <class T>
__global__ void nl_bbox_finish_kernel(const unsigned long long* __restrict__ box, int B, T* __restrict__ cell, T* __restrict__ origin) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= B) return;
  for (int k = 0; k < 9; ++k) cell[9 * (size_t)s + k] = T(0);
  for (int d = 0; d < 3; ++d) {
    const unsigned long long klo = box[6 * (size_t)s + d], khi = box[6 * (size_t)s + 3 + d];
    const bool empty = klo == ~0ull;
    const double lo = empty ? 0.0 : nl_unkey(klo), hi = empty ? 0.0 : nl_unkey(khi);
    const T span = (T)lo == (T)lo ? (T)(hi - lo) : T(0);
    cell[9 * (size_t)s + 4 * d] = (span > T(1) ? span : T(1)) * (T)1.001;
    origin[3 * (size_t)s + d] = (T)lo;
  }
}

// ---- rebuild detection (neighborlist/rebuild_detection.py:37-170): one flag, set when any atom left its cell / moved past the skin
template <class T>
__global__ void nl_cells_changed_kernel(const T* __restrict__ pos, const T* __restrict__ cell, const int* __restrict__ atom_cell,
                                        const int* __restrict__ cpd, const uint8_t* __restrict__ pbc, int N, uint8_t* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  T inv[9];
  inverse3(cell, inv);
  const T p[3] = {pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2]};
  T frac[3];
  rowvec_mat3(p, inv, frac);  // == transpose(inverse(cell)) * r, same summation order
  bool changed = false;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int n = cpd[d];
    int c = (int)floor(frac[d] * (T)n);
    if (pbc[d]) { c = c % n; if (c < 0) c += n; }
    else c = c < 0 ? 0 : (c >= n ? n - 1 : c);
    changed = changed || (c != atom_cell[3 * (size_t)i + d]);
  }
  if (changed) *flag = 1;
}
template <class T>
__global__ void nl_moved_kernel(const T* __restrict__ ref, const T* __restrict__ cur, T threshold, int N, uint8_t* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const T dx = cur[3 * (size_t)i] - ref[3 * (size_t)i], dy = cur[3 * (size_t)i + 1] - ref[3 * (size_t)i + 1], dz = cur[3 * (size_t)i + 2] - ref[3 * (size_t)i + 2];
  if (sqrt(dx * dx + dy * dy + dz * dz) > threshold) *flag = 1;
}

template <class T>
int nl_neighbors_impl(const T* pos, int N, const T* cell, const uint8_t* pbc, const int* batch_idx, int B, double cutoff, int mode, int flags,
                      int* nm, int* nsh, int* num, int M, int fill_value, const int* ptr, int* list_ij, int* list_sh, long long P,
                      const T* origin, char* ws, const NlLayout& L, hipStream_t st, const NlSecond<T>* second = nullptr,
                      NlPacked K = NlPacked{nullptr, nullptr}, const mi_nl_cn_request* cnreq = nullptr, void* cn_block = nullptr) {
  auto* sys = reinterpret_cast<NlSys<T>*>(ws + L.sys);
  auto* glob = reinterpret_cast<NlGlobal*>(ws + L.glob);
  int* natoms = reinterpret_cast<int*>(ws + L.natoms);
  int* keys_in = reinterpret_cast<int*>(ws + L.keys_in);
  int* keys_out = reinterpret_cast<int*>(ws + L.keys_out);
  int* vals_in = reinterpret_cast<int*>(ws + L.vals_in);
  int* vals_out = reinterpret_cast<int*>(ws + L.vals_out);
  auto* wrap = reinterpret_cast<short4*>(ws + L.wrap);
  auto* swrap = reinterpret_cast<short4*>(ws + L.swrap);
  auto* spos = reinterpret_cast<typename Vec4<T>::type*>(ws + L.spos);
  int* cell_start = reinterpret_cast<int*>(ws + L.cell_start);
  const T rc = (T)cutoff;
  // coordination numbers as a by-product (NlCn): header words + checksum slots + cn[] of the caller's block, scaled radii in the workspace
  NlCn CN{nullptr, nullptr, nullptr, 0.0f};
  NlCnBuild CB{nullptr, nullptr, 0, 0.0f, nullptr, nullptr, nullptr, 0, nullptr};
  if (cnreq && cn_block) {
    const float K1 = mi_cn_scale(cnreq->k1);
    int* hdr = reinterpret_cast<int*>(cn_block);
    CN = NlCn{reinterpret_cast<const float*>(ws + L.srk), reinterpret_cast<float*>((char*)cn_block + MI_CN_HEADER_BYTES), hdr, K1};
    CB = NlCnBuild{cnreq->numbers, cnreq->covalent_radii, cnreq->nz, K1, reinterpret_cast<float*>(ws + L.srk),
                   reinterpret_cast<unsigned long long*>(cn_block) + MI_CN_SLOT_OFFSET_U64, cell, 9 * B, batch_idx};
    // (the header -- {unusable flag = 0, 0, cutoff, K} -- and the cleared checksum slots are written by nl_setup_kernel: no extra launch)
    MI_REQUIRE(!(flags & MI_NL_REUSE_GRID), "a search that sums coordination numbers bins its atoms itself (no MI_NL_REUSE_GRID)");
  }
  // cell list: cutoff cast to the positions dtype, then squared (cell_list.py:444,1020);
  // naive: squared in double, then cast (naive.py:290,388)
  const T rc2 = (flags & MI_NL_NAIVE_EXPR) ? (T)(cutoff * cutoff) : rc * rc;

  if (!(flags & MI_NL_REUSE_GRID)) {
    mi_timing_begin("nl_build(setup+assign+binsort+gather)", (void*)st);
    const int* nat = nullptr;
    if (batch_idx && B > 1) {
      MI_HIP_CHECK(hipMemsetAsync(natoms, 0, sizeof(int) * (size_t)B, st));
      nl_count_atoms_kernel<<<mi_blocks(N, 256), 256, 0, st>>>(batch_idx, N, natoms);
      MI_LAUNCH_CHECK();
      nat = natoms;
    }
    const BsScratch bins = bs_carve(reinterpret_cast<int*>(ws + L.bins), L.cell_cap + 2);
    // count[] and fill[] are contiguous (binsort.h) and the workspace offsets are 256-byte aligned: cleared as 16-byte words (the few
    // ints a round-up adds belong to block_sum, which the scan writes before anything reads it)
    const long long zero_words = (2 * bins.cap + 3) / 4;
    const int zero_blocks = (int)(zero_words / 2048 < 1 ? 1 : (zero_words / 2048 > 255 ? 255 : zero_words / 2048));
    const int table_blocks = mi_blocks((long long)B * (NL_PRUNE_R + 1) * (NL_PRUNE_R + 1), 256);
    nl_setup_kernel<T><<<1 + table_blocks + zero_blocks, 256, 0, st>>>(cell, pbc, nat, N, B, rc, origin, sys, glob, table_blocks,
                                                                       reinterpret_cast<int4*>(bins.count), zero_words, K.words ? K.flag : nullptr,
                                                                       CN.flag, CN.K);
    MI_LAUNCH_CHECK();
    nl_assign_kernel<T><<<mi_blocks(N, 256), 256, 0, st>>>(pos, batch_idx, N, sys, keys_in, bins.count, wrap, glob);
    MI_LAUNCH_CHECK();
    MI_HIP_CHECK(bs_sort(bins, keys_in, N, &glob->total_cells, vals_out, cell_start, st));
    nl_rank_gather_kernel<T><<<mi_blocks(CB.srk && 9 * B > N ? 9 * B : N, 256), 256, 0, st>>>(pos, vals_out, keys_in, cell_start, wrap, N, spos, swrap, keys_out, CB);
    MI_LAUNCH_CHECK();
    mi_timing_end((void*)st);
  }
  const int blocks = mi_blocks(N, 4);
  const NlSecond<T> none{T(0), nullptr, nullptr, nullptr, 0};
  const NlPacked nopk{nullptr, nullptr};
  // the companion's flag is cleared by nl_setup_kernel (one launch less); a search that reuses the grid has no set-up stage
  if (K.words && (flags & MI_NL_REUSE_GRID)) MI_HIP_CHECK(hipMemsetAsync(K.flag, 0, sizeof(int), st));
  if (second) {  // single-sweep dual cutoff: the primary outputs take the long cutoff, `second` the short one
    MI_TIMED("nl_query_dual", st,
             (nl_query_kernel<T, MI_NL_MODE_MATRIX, true><<<blocks, 256, 0, st>>>(spos, swrap, keys_out, cell_start, batch_idx, sys, glob, N, rc2, flags, nm, nsh,
                                                                                 num, M, fill_value, ptr, list_ij, list_sh, P, *second, nopk),
              nl_query_tiled_kernel<T, MI_NL_MODE_MATRIX, false, true><<<nl_tiled_grid(), 256, 0, st>>>(spos, swrap, cell_start, sys, glob, B, rc2, flags, nm, nsh,
                                                                                                       num, M, fill_value, ptr, list_ij, list_sh, P, *second, nopk)));
    MI_LAUNCH_CHECK();
    return MI_OK;
  }
#define MI_NLQ(MODE_)                                                                                                              \
  nl_query_kernel<T, MODE_><<<blocks, 256, 0, st>>>(spos, swrap, keys_out, cell_start, batch_idx, sys, glob, N, rc2, flags, nm, nsh, num, M, \
                                                    fill_value, ptr, list_ij, list_sh, P, none, K)
  // both query kernels are launched; the device-side grid description (glob->use_tiled) decides which one does the work
  // and the other returns at once -- no host synchronisation to pick a variant
#define MI_NLT(MODE_)                                                                                                                    \
  do {                                                                                                                                   \
    if ((flags & (MI_NL_HALF_FILL | MI_NL_NAIVE_EXPR)) == 0)                                                                             \
      nl_query_tiled_kernel<T, MODE_, true><<<nl_tiled_grid(), 256, 0, st>>>(spos, swrap, cell_start, sys, glob, B, rc2, flags, nm, nsh, num, \
                                                                          M, fill_value, ptr, list_ij, list_sh, P, none, K);             \
    else                                                                                                                                 \
      nl_query_tiled_kernel<T, MODE_, false><<<nl_tiled_grid(), 256, 0, st>>>(spos, swrap, cell_start, sys, glob, B, rc2, flags, nm, nsh, num, \
                                                                           M, fill_value, ptr, list_ij, list_sh, P, none, K);            \
  } while (0)
  if (mode == MI_NL_MODE_MATRIX && CN.cn) {
    // the search that also sums the coordination numbers: FAST tiled kernel + wave-per-atom kernel, as above (the device picks one)
    MI_TIMED(sizeof(T) == 4 ? "nl_query_matrix_f32" : "nl_query_matrix_f64", st,
             (nl_query_kernel<T, MI_NL_MODE_MATRIX, false, true><<<blocks, 256, 0, st>>>(spos, swrap, keys_out, cell_start, batch_idx, sys, glob, N, rc2, flags, nm,
                                                                                        nsh, num, M, fill_value, ptr, list_ij, list_sh, P, none, K, CN),
              nl_query_tiled_kernel<T, MI_NL_MODE_MATRIX, true, false, true><<<nl_tiled_grid(), 256, 0, st>>>(
                  spos, swrap, cell_start, sys, glob, B, rc2, flags, nm, nsh, num, M, fill_value, ptr, list_ij, list_sh, P, none, K, CN)));
  } else if (mode == MI_NL_MODE_MATRIX) MI_TIMED(sizeof(T) == 4 ? "nl_query_matrix_f32" : "nl_query_matrix_f64", st, MI_NLQ(MI_NL_MODE_MATRIX); MI_NLT(MI_NL_MODE_MATRIX));
  else if (mode == MI_NL_MODE_COUNT) MI_TIMED("nl_query_count", st, MI_NLQ(MI_NL_MODE_COUNT); MI_NLT(MI_NL_MODE_COUNT));
  else MI_TIMED("nl_query_csr", st, MI_NLQ(MI_NL_MODE_CSR); MI_NLT(MI_NL_MODE_CSR));
#undef MI_NLT
#undef MI_NLQ
  MI_LAUNCH_CHECK();
  return MI_OK;
}

template <class T>
int nl_cache_impl(const T* pos, int N, const T* cell, const uint8_t* pbc, const int* batch_idx, int B, double cutoff, int C, int* cpd,
                  int* atom_shift, int* atom_cell, int* counts, int* starts, int* cell_atoms, char* ws, const NlLayout& L, hipStream_t st) {
  int* keys_in = reinterpret_cast<int*>(ws + L.keys_in);
  int* ids = reinterpret_cast<int*>(ws + L.keys_out);
  int* cell_off = reinterpret_cast<int*>(ws + L.cell_start);  // [B+1] scratch (capacity 4N+8B+2 ints)
  int* block_sum = reinterpret_cast<int*>(ws + L.bins);        // scan scratch: 2 x (C / 4096 + 1) ints
  const int nblocks = (C + BS_CHUNK - 1) / BS_CHUNK;
  MI_REQUIRE((size_t)(2 * nblocks + 8) <= bs_scratch_ints(L.cell_cap + 2), "max_total_cells too large for this workspace");
  int* block_off = block_sum + nblocks + 4;
  nl_cache_binsize_kernel<T><<<1, 256, 0, st>>>(cell, pbc, B, (T)cutoff, C, batch_idx != nullptr, cpd, cell_off);
  MI_LAUNCH_CHECK();
  MI_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(int) * (size_t)C, st));
  nl_cache_assign_kernel<T><<<mi_blocks(N, 256), 256, 0, st>>>(pos, cell, pbc, batch_idx, N, cpd, cell_off, keys_in, counts, atom_shift, atom_cell);
  MI_LAUNCH_CHECK();
  MI_HIP_CHECK(hipMemcpyAsync(starts, counts, sizeof(int) * (size_t)C, hipMemcpyDeviceToDevice, st));
  bs_scan_partial_kernel<<<nblocks, 256, 0, st>>>(starts, nullptr, C, block_sum);
  bs_scan_blocks_kernel<<<1, 256, 0, st>>>(block_sum, nblocks, block_off);
  nl_cache_add_offsets_kernel<<<mi_blocks(C, 256), 256, 0, st>>>(starts, block_off, C);
  nl_cache_scatter_kernel<<<mi_blocks(N, 256), 256, 0, st>>>(keys_in, N, starts, ids);
  nl_cache_rank_kernel<<<mi_blocks(N, 256), 256, 0, st>>>(ids, keys_in, starts, counts, N, cell_atoms);
  nl_cache_rewind_kernel<<<mi_blocks(C, 256), 256, 0, st>>>(starts, counts, C);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

}  // namespace

extern "C" {

size_t mi_nl_workspace_bytes(int n_atoms, int n_systems, int dtype) {
  if (n_atoms < 0 || n_systems < 1) return 0;
  return nl_layout(n_atoms, n_systems, dtype).total;
}

}  // extern "C" (helper below has internal linkage)
static int nl_neighbors_entry(const void* positions, int n_atoms, const void* cell, const uint8_t* pbc, const int32_t* batch_idx, int n_systems,
                    double cutoff, int dtype, int mode, int flags, int32_t* neighbor_matrix, int32_t* neighbor_matrix_shifts,
                    int32_t* num_neighbors, int max_neighbors, int fill_value, const int32_t* neighbor_ptr, int32_t* list_ij,
                    int32_t* list_shifts, long long n_pairs, const void* bin_origin, void* workspace, size_t workspace_bytes, void* stream, void* packed_out,
                    const mi_nl_cn_request* cnreq = nullptr, void* cn_block = nullptr) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype must be MI_F32 or MI_F64");
  MI_REQUIRE(n_atoms >= 0 && n_systems >= 1, "n_atoms >= 0 and n_systems >= 1");
  MI_REQUIRE(cutoff > 0, "cutoff must be positive");
  MI_REQUIRE(mode >= 0 && mode <= 2, "mode");
  if (n_atoms == 0) return MI_OK;
  MI_REQUIRE(positions && cell && pbc && workspace, "null pointer");
  if (mode == MI_NL_MODE_MATRIX) MI_REQUIRE(neighbor_matrix && num_neighbors && max_neighbors >= 0, "matrix outputs");
  if (mode == MI_NL_MODE_COUNT) MI_REQUIRE(num_neighbors != nullptr, "num_neighbors");
  if (mode == MI_NL_MODE_CSR) MI_REQUIRE(neighbor_ptr && list_ij, "csr outputs");
  if (flags & MI_NL_NO_SHIFTS) neighbor_matrix_shifts = nullptr;
  NlLayout L = nl_layout(n_atoms, n_systems, dtype);
  if (workspace_bytes < L.total) { mi_set_error("workspace too small: %zu < %zu", workspace_bytes, L.total); return MI_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  NlPacked K{nullptr, nullptr};
  if (packed_out) {  // [64 words: flag line][n_atoms * max_neighbors words], the layout mi_d3_packed reads
    MI_REQUIRE(mode == MI_NL_MODE_MATRIX && neighbor_matrix_shifts && !(flags & (MI_NL_NO_PAD | MI_NL_HALF_FILL)) && n_atoms < (1 << 26),
               "packed companion: full (not half-filled) padded matrix with shifts, n_atoms < 2^26");
    K.flag = reinterpret_cast<int*>(packed_out);
    K.words = reinterpret_cast<unsigned*>(packed_out) + 64;
  }
  if (cnreq) MI_REQUIRE(packed_out && cn_block && cnreq->numbers && cnreq->covalent_radii && cnreq->nz >= 2, "coordination numbers: companion, output block, numbers and radii");
  if (dtype == MI_F32)
    return nl_neighbors_impl<float>((const float*)positions, n_atoms, (const float*)cell, pbc, batch_idx, n_systems, cutoff, mode, flags,
                                    neighbor_matrix, neighbor_matrix_shifts, num_neighbors, max_neighbors, fill_value, neighbor_ptr, list_ij,
                                    list_shifts, n_pairs, (const float*)bin_origin, (char*)workspace, L, st, nullptr, K, cnreq, cn_block);
  return nl_neighbors_impl<double>((const double*)positions, n_atoms, (const double*)cell, pbc, batch_idx, n_systems, cutoff, mode, flags,
                                   neighbor_matrix, neighbor_matrix_shifts, num_neighbors, max_neighbors, fill_value, neighbor_ptr, list_ij,
                                   list_shifts, n_pairs, (const double*)bin_origin, (char*)workspace, L, st, nullptr, K, cnreq, cn_block);
}

extern "C" {
int mi_nl_neighbors(const void* positions, int n_atoms, const void* cell, const uint8_t* pbc, const int32_t* batch_idx, int n_systems,
                    double cutoff, int dtype, int mode, int flags, int32_t* neighbor_matrix, int32_t* neighbor_matrix_shifts,
                    int32_t* num_neighbors, int max_neighbors, int fill_value, const int32_t* neighbor_ptr, int32_t* list_ij,
                    int32_t* list_shifts, long long n_pairs, const void* bin_origin, void* workspace, size_t workspace_bytes, void* stream) {
  return nl_neighbors_entry(positions, n_atoms, cell, pbc, batch_idx, n_systems, cutoff, dtype, mode, flags, neighbor_matrix, neighbor_matrix_shifts,
                            num_neighbors, max_neighbors, fill_value, neighbor_ptr, list_ij, list_shifts, n_pairs, bin_origin, workspace,
                            workspace_bytes, stream, nullptr);
}

size_t mi_nl_packed_bytes(int n_atoms, int max_neighbors) {
  if (n_atoms <= 0 || max_neighbors <= 0 || n_atoms >= (1 << 26)) return 0;
  return 256 + sizeof(unsigned) * (size_t)n_atoms * (size_t)max_neighbors;
}

int mi_nl_neighbors_packed(const void* positions, int n_atoms, const void* cell, const uint8_t* pbc, const int32_t* batch_idx, int n_systems,
                           double cutoff, int dtype, int flags, int32_t* neighbor_matrix, int32_t* neighbor_matrix_shifts,
                           int32_t* num_neighbors, int max_neighbors, int fill_value, const void* bin_origin, void* workspace,
                           size_t workspace_bytes, void* packed_out, size_t packed_bytes, void* stream) {
  MI_REQUIRE(packed_out != nullptr && packed_bytes >= mi_nl_packed_bytes(n_atoms, max_neighbors) && mi_nl_packed_bytes(n_atoms, max_neighbors) > 0,
             "packed_out: mi_nl_packed_bytes(n_atoms, max_neighbors) bytes");
  return nl_neighbors_entry(positions, n_atoms, cell, pbc, batch_idx, n_systems, cutoff, dtype, MI_NL_MODE_MATRIX, flags, neighbor_matrix,
                            neighbor_matrix_shifts, num_neighbors, max_neighbors, fill_value, nullptr, nullptr, nullptr, 0, bin_origin, workspace,
                            workspace_bytes, stream, packed_out);
}

size_t mi_nl_cn_bytes(int n_atoms) { return n_atoms > 0 ? MI_CN_HEADER_BYTES + sizeof(float) * (size_t)n_atoms : 0; }

int mi_nl_neighbors_packed_cn(const void* positions, int n_atoms, const void* cell, const uint8_t* pbc, const int32_t* batch_idx, int n_systems,
                              double cutoff, int dtype, int flags, int32_t* neighbor_matrix, int32_t* neighbor_matrix_shifts,
                              int32_t* num_neighbors, int max_neighbors, int fill_value, const void* bin_origin, void* workspace,
                              size_t workspace_bytes, void* packed_out, size_t packed_bytes, const mi_nl_cn_request* request, void* cn_out,
                              size_t cn_bytes, void* stream) {
  MI_REQUIRE(packed_out != nullptr && packed_bytes >= mi_nl_packed_bytes(n_atoms, max_neighbors) && mi_nl_packed_bytes(n_atoms, max_neighbors) > 0,
             "packed_out: mi_nl_packed_bytes(n_atoms, max_neighbors) bytes");
  MI_REQUIRE(request != nullptr && cn_out != nullptr && cn_bytes >= mi_nl_cn_bytes(n_atoms), "cn_out: mi_nl_cn_bytes(n_atoms) bytes and a request");
  return nl_neighbors_entry(positions, n_atoms, cell, pbc, batch_idx, n_systems, cutoff, dtype, MI_NL_MODE_MATRIX, flags, neighbor_matrix,
                            neighbor_matrix_shifts, num_neighbors, max_neighbors, fill_value, nullptr, nullptr, nullptr, 0, bin_origin, workspace,
                            workspace_bytes, stream, packed_out, request, cn_out);
}

int mi_nl_neighbors_dual(const void* positions, int n_atoms, const void* cell, const uint8_t* pbc, const int32_t* batch_idx, int n_systems,
                         double cutoff_short, double cutoff_long, int dtype, int flags, int32_t* neighbor_matrix_short,
                         int32_t* neighbor_matrix_shifts_short, int32_t* num_neighbors_short, int max_neighbors_short,
                         int32_t* neighbor_matrix_long, int32_t* neighbor_matrix_shifts_long, int32_t* num_neighbors_long, int max_neighbors_long,
                         int fill_value, const void* bin_origin, void* workspace, size_t workspace_bytes, void* stream) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype must be MI_F32 or MI_F64");
  MI_REQUIRE(n_atoms >= 0 && n_systems >= 1, "n_atoms >= 0 and n_systems >= 1");
  MI_REQUIRE(cutoff_short > 0 && cutoff_long >= cutoff_short, "0 < cutoff_short <= cutoff_long");
  if (n_atoms == 0) return MI_OK;
  MI_REQUIRE(positions && cell && pbc && workspace, "null pointer");
  MI_REQUIRE(neighbor_matrix_short && num_neighbors_short && max_neighbors_short >= 0 && neighbor_matrix_long && num_neighbors_long &&
                 max_neighbors_long >= 0, "matrix outputs");
  if (flags & MI_NL_NO_SHIFTS) neighbor_matrix_shifts_short = neighbor_matrix_shifts_long = nullptr;
  NlLayout L = nl_layout(n_atoms, n_systems, dtype);
  if (workspace_bytes < L.total) { mi_set_error("workspace too small: %zu < %zu", workspace_bytes, L.total); return MI_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  // squared cutoffs exactly as the single-cutoff entry point forms them (naive: squared in double, then cast; cell list: cast, then squared)
  if (dtype == MI_F32) {
    const float r1 = (float)cutoff_short;
    const NlSecond<float> sec{(flags & MI_NL_NAIVE_EXPR) ? (float)(cutoff_short * cutoff_short) : r1 * r1, neighbor_matrix_short,
                              neighbor_matrix_shifts_short, num_neighbors_short, max_neighbors_short};
    return nl_neighbors_impl<float>((const float*)positions, n_atoms, (const float*)cell, pbc, batch_idx, n_systems, cutoff_long, MI_NL_MODE_MATRIX,
                                    flags, neighbor_matrix_long, neighbor_matrix_shifts_long, num_neighbors_long, max_neighbors_long, fill_value,
                                    nullptr, nullptr, nullptr, 0, (const float*)bin_origin, (char*)workspace, L, st, &sec);
  }
  const double r1 = cutoff_short;
  const NlSecond<double> sec{(flags & MI_NL_NAIVE_EXPR) ? cutoff_short * cutoff_short : r1 * r1, neighbor_matrix_short, neighbor_matrix_shifts_short,
                             num_neighbors_short, max_neighbors_short};
  return nl_neighbors_impl<double>((const double*)positions, n_atoms, (const double*)cell, pbc, batch_idx, n_systems, cutoff_long, MI_NL_MODE_MATRIX,
                                   flags, neighbor_matrix_long, neighbor_matrix_shifts_long, num_neighbors_long, max_neighbors_long, fill_value,
                                   nullptr, nullptr, nullptr, 0, (const double*)bin_origin, (char*)workspace, L, st, &sec);
}

int mi_nl_matrix_to_coo(const int32_t* neighbor_matrix, const int32_t* neighbor_matrix_shifts, const int32_t* neighbor_ptr, int n_atoms,
                             int max_neighbors, int fill_value, int32_t* list_ij, int32_t* list_shifts, long long n_pairs, void* stream) {
  if (n_atoms == 0 || n_pairs == 0) return MI_OK;
  MI_REQUIRE(neighbor_matrix && neighbor_ptr && list_ij, "null pointer");
  nl_matrix_to_coo_kernel<<<mi_blocks(n_atoms, 4), 256, 0, (hipStream_t)stream>>>(neighbor_matrix, neighbor_matrix_shifts, neighbor_ptr, n_atoms,
                                                                                  max_neighbors, fill_value, list_ij,
                                                                                  neighbor_matrix_shifts ? list_shifts : nullptr, n_pairs);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

int mi_nl_bounding_cells(const void* positions, const int32_t* batch_idx, int n_atoms, int n_systems, int dtype, void* cell_out,
                         void* origin_out, void* scratch, void* stream) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype");
  MI_REQUIRE(n_systems >= 1 && cell_out && origin_out && scratch, "null pointer");
  MI_REQUIRE(n_atoms == 0 || positions, "positions");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* box = (unsigned long long*)scratch;
  // min keys start at all-ones, max keys at zero: one memset of 0xff over the rows, one of 0 over the max halves is avoided by
  // storing {lo x3 | hi x3} and initialising with two strided fills in the kernel-free way below
  MI_HIP_CHECK(hipMemsetAsync(box, 0xff, sizeof(unsigned long long) * 6 * (size_t)n_systems, st));
  MI_HIP_CHECK(hipMemset2DAsync(box + 3, sizeof(unsigned long long) * 6, 0, sizeof(unsigned long long) * 3, (size_t)n_systems, st));
  if (n_atoms > 0) {
    if (dtype == MI_F32) nl_bbox_kernel<float><<<mi_blocks(n_atoms, 256), 256, 0, st>>>((const float*)positions, batch_idx, n_atoms, box);
    else nl_bbox_kernel<double><<<mi_blocks(n_atoms, 256), 256, 0, st>>>((const double*)positions, batch_idx, n_atoms, box);
    MI_LAUNCH_CHECK();
  }
  if (dtype == MI_F32) nl_bbox_finish_kernel<float><<<mi_blocks(n_systems, 256), 256, 0, st>>>(box, n_systems, (float*)cell_out, (float*)origin_out);
  else nl_bbox_finish_kernel<double><<<mi_blocks(n_systems, 256), 256, 0, st>>>(box, n_systems, (double*)cell_out, (double*)origin_out);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

int mi_nl_cells_changed(const void* positions, const void* cell, const int32_t* atom_to_cell_mapping, const int32_t* cells_per_dimension,
                        const uint8_t* pbc, int n_atoms, int dtype, uint8_t* flag, void* stream) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype");
  if (n_atoms <= 0) return MI_OK;
  MI_REQUIRE(positions && cell && atom_to_cell_mapping && cells_per_dimension && pbc && flag, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_F32)
    nl_cells_changed_kernel<float><<<mi_blocks(n_atoms, 256), 256, 0, st>>>((const float*)positions, (const float*)cell, atom_to_cell_mapping,
                                                                           cells_per_dimension, pbc, n_atoms, flag);
  else
    nl_cells_changed_kernel<double><<<mi_blocks(n_atoms, 256), 256, 0, st>>>((const double*)positions, (const double*)cell, atom_to_cell_mapping,
                                                                            cells_per_dimension, pbc, n_atoms, flag);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

int mi_nl_moved_beyond_skin(const void* reference_positions, const void* current_positions, double threshold, int n_atoms, int dtype,
                            uint8_t* flag, void* stream) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype");
  if (n_atoms <= 0) return MI_OK;
  MI_REQUIRE(reference_positions && current_positions && flag, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_F32)
    nl_moved_kernel<float><<<mi_blocks(n_atoms, 256), 256, 0, st>>>((const float*)reference_positions, (const float*)current_positions,
                                                                   (float)threshold, n_atoms, flag);
  else
    nl_moved_kernel<double><<<mi_blocks(n_atoms, 256), 256, 0, st>>>((const double*)reference_positions, (const double*)current_positions,
                                                                    threshold, n_atoms, flag);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

int mi_nl_estimate_sizes(const void* cell, const uint8_t* pbc, int n_systems, double cutoff, int max_nbins, int dtype, int32_t* number_of_cells,
                         int32_t* neighbor_search_radius, void* stream) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype");
  MI_REQUIRE(cell && pbc && number_of_cells && neighbor_search_radius && n_systems >= 1, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_F32)
    nl_estimate_sizes_kernel<float><<<mi_blocks(n_systems, 128), 128, 0, st>>>((const float*)cell, pbc, n_systems, (float)cutoff, max_nbins,
                                                                               number_of_cells, neighbor_search_radius);
  else
    nl_estimate_sizes_kernel<double><<<mi_blocks(n_systems, 128), 128, 0, st>>>((const double*)cell, pbc, n_systems, cutoff, max_nbins,
                                                                                number_of_cells, neighbor_search_radius);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

int mi_nl_build_cell_cache(const void* positions, int n_atoms, const void* cell, const uint8_t* pbc, const int32_t* batch_idx, int n_systems,
                           double cutoff, int dtype, int max_total_cells, int32_t* cells_per_dimension, int32_t* atom_periodic_shifts,
                           int32_t* atom_to_cell_mapping, int32_t* atoms_per_cell_count, int32_t* cell_atom_start_indices,
                           int32_t* cell_atom_list, void* workspace, size_t workspace_bytes, void* stream) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype");
  if (n_atoms == 0) return MI_OK;
  MI_REQUIRE(positions && cell && pbc && workspace && cells_per_dimension && atom_periodic_shifts && atom_to_cell_mapping &&
                 atoms_per_cell_count && cell_atom_start_indices && cell_atom_list,
             "null pointer");
  MI_REQUIRE(max_total_cells >= n_systems, "max_total_cells");
  NlLayout L = nl_layout(n_atoms, n_systems, dtype);
  if (workspace_bytes < L.total) { mi_set_error("workspace too small: %zu < %zu", workspace_bytes, L.total); return MI_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_F32)
    return nl_cache_impl<float>((const float*)positions, n_atoms, (const float*)cell, pbc, batch_idx, n_systems, cutoff, max_total_cells,
                                cells_per_dimension, atom_periodic_shifts, atom_to_cell_mapping, atoms_per_cell_count, cell_atom_start_indices,
                                cell_atom_list, (char*)workspace, L, st);
  return nl_cache_impl<double>((const double*)positions, n_atoms, (const double*)cell, pbc, batch_idx, n_systems, cutoff, max_total_cells,
                               cells_per_dimension, atom_periodic_shifts, atom_to_cell_mapping, atoms_per_cell_count, cell_atom_start_indices,
                               cell_atom_list, (char*)workspace, L, st);
}

}  // extern "C"

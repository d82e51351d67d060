// ewald.hip -- real-space (erfc-damped) part of Ewald / PME.  gfx950, wave64.
//
// Arithmetic of the reference kernels `_[batch_]ewald_real_space_energy[_forces][_charge_grad]_{,neighbor_matrix_}kernel`
// (interactions/electrostatics/ewald_kernels.py:266-1495) and their fp64 helpers (:150-258): separation vector and
// |r| in the input dtype, then fp64: E_i += 1/2 q_i q_j erfc_AS(a r)/r with the Abramowitz-Stegun 7.1.26 polynomial
// (math/math.py:52-93, NOT libm erfc: SURVEY F4), force magnitude 1/2 q_i q_j (erfc/r^3 + 2a/sqrt(pi) e^{-a^2 r^2}/r^2).
// Matrix entries equal to mask_value are padding (exact equality, ewald_kernels.py:304); pairs with r <= 1e-8 are skipped.
//
// Execution shape: one wave64 per atom, lanes stride the row / CSR range; fp64 wave reductions.  The reference adds
// -f to atom i and atomically +f to atom j for every stored entry (ewald_kernels.py:518-544, :864-873).  Over a symmetric (full)
// list -- what its 1/2 prefactor assumes -- that equals 2x the row owner's sum, so the fast path writes
// forces and charge gradients by the owner only: no atomics, deterministic.  Whether the list IS symmetric is checked in the same
// pass: every stored entry adds hash(i, j, S) to one 64-bit sum and hash(j, i, -S) to another; the sums agree iff the entries pair
// up (up to a 2^-32-ish hash collision).  When they differ (half lists, rows truncated by overflow, one-sided lists) two fix-up
// launches -- which exit at once otherwise -- redo forces / charge gradients with the reference's i/j scatter (atomics), so any list
// gives the reference's result.  Same scheme in the adjoint.
#include "common.h"

namespace {

// 1/x and 1/sqrt(x) in fp64 from the hardware seeds (v_rcp_f64 / v_rsq_f64, ~26 bits) + two Newton steps: <= 1 ulp, ~7 / ~10 instructions
// against ~25 for an IEEE divide and ~30 + 25 for sqrt followed by a divide.  The real-space pass is fp64-VALU-bound; results move in the
// last bit only (the parity bar of this path is 1e-10 relative).  EW_IEEE_DIV=1 restores the IEEE forms (A/B).
#ifndef EW_IEEE_DIV
#define EW_IEEE_DIV 0
#endif
__device__ __forceinline__ double ew_rcp(double x) {
#if EW_IEEE_DIV
  return 1.0 / x;
#else
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
#endif
}
__device__ __forceinline__ double ew_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}
// distance and reciprocal distance of a pair vector held in the positions dtype (see ewald_real_kernel)
template <class T> __device__ __forceinline__ void ew_dist(T sx, T sy, T sz, double& dist, double& rinv) {
  if (sizeof(T) == 8 && !EW_IEEE_DIV) {
    const double r2 = (double)(sx * sx + sy * sy + sz * sz);
    rinv = ew_rsqrt(r2);
    dist = r2 * rinv;
  } else {
    dist = (double)sqrt(sx * sx + sy * sy + sz * sz);
    rinv = ew_rcp(dist);
  }
}
__device__ __forceinline__ double erfc_as_poly(double x, double e_neg_x2) {
  // x >= 0 here (alpha * distance); constants verbatim from math/math.py:73-78
  const double p = 0.3275911, a1 = 0.254829592, a2 = -0.284496736, a3 = 1.421413741, a4 = -1.453152027, a5 = 1.061405429;
  const double t = ew_rcp(1.0 + p * x);
  const double t2 = t * t, t3 = t2 * t, t4 = t3 * t, t5 = t4 * t;
  const double poly = a1 * t + a2 * t2 + a3 * t3 + a4 * t4 + a5 * t5;
  return poly * e_neg_x2;
}

// Order-sensitive 64-bit mix of one stored entry: two independent 32-bit mixes of (a, b, S) side by side, summed per direction in 64 bits.
// The list counts as symmetric iff the forward sum (entries as stored) equals the reverse sum (entries mirrored): a false "symmetric"
// needs a collision of 64-bit sums, ~2^-64 per call (one 32-bit mix summed in 64 bits gave ~2^-32, round-2 ADVICE).  Shift components
// beyond +-1024 alias in the packed shift word; that only weakens the mix, it cannot make a symmetric list look asymmetric.
__device__ __forceinline__ unsigned long long ew_entry_hash(unsigned a, unsigned b, int s0, int s1, int s2) {
  const unsigned sw = ((unsigned)(s0 + 1024) | ((unsigned)(s1 + 1024) << 11) | ((unsigned)(s2 + 1024) << 22));
  unsigned h = a * 0x9E3779B1u;
  h ^= __builtin_rotateleft32(b * 0x85EBCA77u, 13);
  h ^= sw * 0xC2B2AE3Du;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  // second half: other multipliers on the raw inputs, folded with the finished first half (7 instructions instead of a second full mix:
  // the check costs the real-space pass 0.195 -> 0.205 ms instead of 0.222)
  unsigned g = (b * 0x27D4EB2Fu) ^ (a * 0x165667B1u) ^ __builtin_rotateleft32(h, 16) ^ sw;
  g ^= g >> 15; g *= 0x846CA68Bu; g ^= g >> 16;
  return ((unsigned long long)g << 32) | h;
}
// Checksum scratch: sym[0..1] = final {forward, reverse} sums (written by block 0 of ewald_fixup_zero_kernel), then EW_SYM_SLOTS pairs of
// partial sums.  A wave adds its partials to the slot pair of its atom index: 100k waves on ONE address would serialise at
// ~12 ns per device-scope atomic (2.4 ms measured); spread over 1024 slot pairs they cost nothing measurable.
#define EW_SYM_SLOTS 1024
#define EW_SYM_WORDS (2 + 2 * EW_SYM_SLOTS)
#define EW_BWD_SLOTS 64  // partial rows per system of the backward pass's per-system sums (one wave64 folds them)
__device__ __forceinline__ void ew_sym_flush(unsigned long long hf, unsigned long long hr, unsigned long long* __restrict__ sym, int lane, int i) {
  hf = wave_sum(hf); hr = wave_sum(hr);
  if (lane == 0 && (hf | hr)) {
    unsigned long long* slot = sym + 2 + 2 * (size_t)(i & (EW_SYM_SLOTS - 1));
    atomicAdd(&slot[0], hf); atomicAdd(&slot[1], hr);
  }
}

// {x, y, z, q} per atom in one 16 / 32-byte record: the pair loop gathers ONE record per neighbour (fp64: two 16-byte loads of one
// line) instead of three coordinate loads plus the charge from a second array.  Same values, same arithmetic: results are bit-identical
// with and without the records; measured 0.205 -> 0.195 ms on the headline list (pack kernel included).
// `sym` (may be null): the checksum words of the same call, cleared here instead of by a memset node of their own (round 4).
template <class T>
__global__ void ewald_pack_kernel(const T* __restrict__ pos, const T* __restrict__ q, int N, typename Vec4<T>::type* __restrict__ rec,
                                  unsigned long long* __restrict__ sym) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (sym) for (int k = i; k < EW_SYM_WORDS; k += gridDim.x * blockDim.x) sym[k] = 0ull;
  if (i >= N) return;
  typename Vec4<T>::type r;
  r.x = pos[3 * (size_t)i]; r.y = pos[3 * (size_t)i + 1]; r.z = pos[3 * (size_t)i + 2]; r.w = q[i];
  rec[i] = r;
}

// TRUST (round 6, padded matrices only): `snum` = the num_neighbors array of the full-list search that wrote idx / ush -- the host passes it
// only while both are provably unchanged since (version counters: neighborlist/_engine.py::FullListRecord).  A full list is symmetric by
// construction unless a row overflowed its M slots, so the two 64-bit hashes per stored entry (a fifth of this kernel's instructions:
// 0.172 -> 0.138 ms on the 9 A headline list) are replaced by: (1) snum[i] > M raises the "not symmetric" mark, (2) every vstride-th row
// (vstride a power of two, rotating with vphase) has one of its entries (j, S) looked up in row j as (i, -S); a miss raises the mark -- a
// bulk edit behind torch's back is caught like a stale D3 companion is, a single edited entry is the caller's to announce
// (neighborlist.invalidate).  The mark is a unit in a forward checksum slot: forward != reverse sends the call down the general scatter
// path, as a failed hash would.  Rows are walked up to snum[i] instead of M (hits come first, padding behind them).
// The look-ups run in `vblocks` blocks of their own at the FRONT of the grid, 16 lanes per sampled row: done by the sampled row's own wave
// they cost 0.020 ms at stride 64 (four dependent loads in front of the pair loop keep the wave's whole block resident 2.5 x as long, and
// this kernel lives off its occupancy); as 1 / 64 of the rows x 1 / 16 of a block each they are not measurable.
template <class T, bool CSR, bool REC, bool TRUST = false>
__global__ __launch_bounds__(256) void ewald_real_kernel(const T* __restrict__ pos, const T* __restrict__ q, const T* __restrict__ cell,
                                                         const T* __restrict__ alpha, const int* __restrict__ batch_idx, int N,
                                                         const int* __restrict__ idx, const int* __restrict__ ush, const int* __restrict__ nptr,
                                                         int M, int mask_value, int flags, double* __restrict__ energies,
                                                         T* __restrict__ forces, double* __restrict__ cgrad,
                                                         unsigned long long* __restrict__ sym, const typename Vec4<T>::type* __restrict__ rec,
                                                         const int* __restrict__ snum = nullptr, int vblocks = 0, int vshift = 0, int vphase = 0) {
  static_assert(!(TRUST && CSR), "the trusted form describes a padded matrix");
  const int lane = threadIdx.x & (MI_WAVE - 1);
  if (TRUST && (int)blockIdx.x < vblocks) {  // ---- sampled mirror look-ups: sample k is row (k << vshift) + ((-vphase) mod stride)
    const int sub = lane >> 4, sl = lane & 15;
    const long long k = ((long long)blockIdx.x * (blockDim.x / MI_WAVE) + threadIdx.x / MI_WAVE) * 4 + sub;
    const long long il = (k << vshift) + ((-(long long)vphase) & ((1ll << vshift) - 1));
    bool found = true;  // rows outside the list and empty rows have nothing to look up
    int i = 0;
    if (il < N) {
      i = (int)il;
      const int cnt_i = snum[i];
      const int used = cnt_i < M ? cnt_i : M;
      if (used > 0) {
        unsigned h = ((unsigned)i * 0x9E3779B1u) ^ ((unsigned)vphase * 0x85EBCA77u);
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13;
        const long long ek = (long long)i * M + (long long)(h % (unsigned)used);
        const int j = idx[ek];
        const int S0 = ush[3 * ek], S1 = ush[3 * ek + 1], S2 = ush[3 * ek + 2];
        found = false;
        if ((unsigned)j < (unsigned)N) {
          const int cnt_j = snum[j];
          const long long jb = (long long)j * M, je = jb + (cnt_j < M ? cnt_j : M);
          for (long long e = jb + sl; e < je; e += 16)
            found = found || (idx[e] == i && ush[3 * e] == -S0 && ush[3 * e + 1] == -S1 && ush[3 * e + 2] == -S2);
        }
      }
    }
    const unsigned long long hits = __ballot(found);
    if (sl == 0 && ((hits >> (16 * sub)) & 0xFFFFull) == 0ull && sym) atomicAdd(&sym[2 + 2 * (size_t)(i & (EW_SYM_SLOTS - 1))], 1ull);
    return;
  }
  const int i = __builtin_amdgcn_readfirstlane((blockIdx.x - (TRUST ? vblocks : 0)) * (blockDim.x / MI_WAVE) + threadIdx.x / MI_WAVE);
  if (i >= N) return;
  const int s = batch_idx ? batch_idx[i] : 0;
  const double qi = (double)q[i], al = (double)alpha[s];
  T cm[9];
  for (int k = 0; k < 9; ++k) cm[k] = cell[9 * (size_t)s + k];
  // orthorhombic cell (wave-uniform): S . cell has one non-zero product per component; adding the exact zeros of the other six changes nothing
  const bool ortho = cm[1] == T(0) && cm[2] == T(0) && cm[3] == T(0) && cm[5] == T(0) && cm[6] == T(0) && cm[7] == T(0);
  const T pix = pos[3 * (size_t)i], piy = pos[3 * (size_t)i + 1], piz = pos[3 * (size_t)i + 2];
  const bool wf = (flags & MI_EW_FORCES) != 0, wc = (flags & MI_EW_CHARGE_GRAD) != 0;
  const double two_over_sqrt_pi = 2.0 / 1.7724538509055159;
  long long beg, end;
  if (CSR) { beg = nptr[i]; end = nptr[i + 1]; } else { beg = (long long)i * M; end = beg + M; }
  double eacc = 0.0, cgi = 0.0;
  T fx = 0, fy = 0, fz = 0;
  unsigned long long hf = 0, hr = 0;
  if (TRUST) {
    const int cnt_i = __builtin_amdgcn_readfirstlane(snum[i]);
    end = beg + (cnt_i < M ? cnt_i : M);
    // cnt_i > M: the search dropped entries of this row, their mirrors in other rows have no partner
    if (cnt_i > M && lane == 0 && sym) atomicAdd(&sym[2 + 2 * (size_t)(i & (EW_SYM_SLOTS - 1))], 1ull);
  }
  for (long long e = beg + lane; e < end; e += MI_WAVE) {
    const int j = idx[e];
    if ((!CSR && j == mask_value) || (unsigned)j >= (unsigned)N) continue;  // out-of-range indices (e.g. -1 padding with another mask_value) are padding
    T pjx, pjy, pjz;
    double qj;
    if (REC) { const typename Vec4<T>::type r = rec[j]; pjx = r.x; pjy = r.y; pjz = r.z; qj = (double)r.w; }
    else { pjx = pos[3 * (size_t)j]; pjy = pos[3 * (size_t)j + 1]; pjz = pos[3 * (size_t)j + 2]; qj = (double)q[j]; }
    const int S0 = ush[3 * e], S1 = ush[3 * e + 1], S2 = ush[3 * e + 2];
    if (!TRUST && sym) { hf += ew_entry_hash((unsigned)i, (unsigned)j, S0, S1, S2); hr += ew_entry_hash((unsigned)j, (unsigned)i, -S0, -S1, -S2); }
    const T fs[3] = {(T)S0, (T)S1, (T)S2};
    T sh[3];
    if (ortho) { sh[0] = cm[0] * fs[0]; sh[1] = cm[4] * fs[1]; sh[2] = cm[8] * fs[2]; }  // the other six products are exact zeros
    else rowvec_mat3(fs, cm, sh);  // == transpose(cell) * S with the same summation order
    const T sx = (pjx - pix) + sh[0], sy = (pjy - piy) + sh[1], sz = (pjz - piz) + sh[2];
    // distance and its reciprocal: fp64 positions -> one reciprocal square root (dist = r2 * rinv, <= 1 ulp from sqrt); fp32 positions keep
    // the reference's fp32 sqrt (the distance is an fp32 quantity there) and take the reciprocal of its fp64 cast
    double dist, rinv;
    ew_dist<T>(sx, sy, sz, dist, rinv);
    if (!(dist > 1e-8)) continue;  // (r2 == 0: rsq = inf, dist = NaN -> skipped as well)
    const double ar = al * dist;
    const double ex = exp(-(ar * ar));
    const double ec = erfc_as_poly(ar, ex);
    // products instead of the reference's quotients ec / d, ec / d^3, ex / d^2: the same numbers to the last bit or two
    const double pot = ec * rinv;
    eacc += 0.5 * qi * qj * pot;
    if (wf) {
      const double rinv2 = rinv * rinv;
      const double fm = (0.5 * qi * qj) * (pot * rinv2 + two_over_sqrt_pi * al * ex * rinv2);
      const T fmt = (T)fm;
      fx -= fmt * sx; fy -= fmt * sy; fz -= fmt * sz;
    }
    if (wc) cgi += qj * (0.5 * pot);
  }
  eacc = wave_sum(eacc);
  if (lane == 0) energies[i] = eacc;
  if (wf) {
    fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
    if (lane == 0) { forces[3 * (size_t)i] = T(2) * fx; forces[3 * (size_t)i + 1] = T(2) * fy; forces[3 * (size_t)i + 2] = T(2) * fz; }
  }
  if (wc) {
    cgi = wave_sum(cgi);
    if (lane == 0) cgrad[i] = 2.0 * cgi;
  }
  if (!TRUST && sym) ew_sym_flush(hf, hr, sym, lane, i);
}

// ---- fix-up for lists that are not symmetric: the reference's scatter (ewald_kernels.py:518-544, :864-873) ---------------------
// Round 4: this kernel also folds the EW_SYM_SLOTS partial checksums (every block for its own decision -- 16 KB from L2 -- and block 0
// publishes {forward, reverse} in sym[0..1] for the scatter kernel behind it): the separate one-block reduce launch is gone.
template <class T>
__global__ __launch_bounds__(256) void ewald_fixup_zero_kernel(unsigned long long* __restrict__ sym, T* __restrict__ a3, T* __restrict__ a1, double* __restrict__ d1, int N) {
  __shared__ unsigned long long part[2][256 / MI_WAVE];
  unsigned long long f = 0, r = 0;
  for (int k = threadIdx.x; k < EW_SYM_SLOTS; k += 256) { f += sym[2 + 2 * k]; r += sym[3 + 2 * k]; }
  f = wave_sum(f); r = wave_sum(r);
  if ((threadIdx.x & (MI_WAVE - 1)) == 0) { part[0][threadIdx.x / MI_WAVE] = f; part[1][threadIdx.x / MI_WAVE] = r; }
  __syncthreads();
  f = 0; r = 0;
#pragma unroll
  for (int k = 0; k < 256 / MI_WAVE; ++k) { f += part[0][k]; r += part[1][k]; }
  if (blockIdx.x == 0 && threadIdx.x == 0) { sym[0] = f; sym[1] = r; }
  if (f == r) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (a3) { a3[3 * (size_t)i] = T(0); a3[3 * (size_t)i + 1] = T(0); a3[3 * (size_t)i + 2] = T(0); }
  if (a1) a1[i] = T(0);
  if (d1) d1[i] = 0.0;
}
template <class T, bool CSR>
__global__ __launch_bounds__(256) void ewald_real_scatter_kernel(const T* __restrict__ pos, const T* __restrict__ q, const T* __restrict__ cell,
                                                                 const T* __restrict__ alpha, const int* __restrict__ batch_idx, int N,
                                                                 const int* __restrict__ idx, const int* __restrict__ ush, const int* __restrict__ nptr,
                                                                 int M, int mask_value, int flags, T* __restrict__ forces, double* __restrict__ cgrad,
                                                                 const unsigned long long* __restrict__ sym) {
  if (sym[0] == sym[1]) return;  // symmetric list: the owner-only results of ewald_real_kernel stand
  const int lane = threadIdx.x & (MI_WAVE - 1);
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / MI_WAVE) + threadIdx.x / MI_WAVE);
  if (i >= N) return;
  const int s = batch_idx ? batch_idx[i] : 0;
  const double qi = (double)q[i], al = (double)alpha[s];
  T cm[9];
  for (int k = 0; k < 9; ++k) cm[k] = cell[9 * (size_t)s + k];
  const T pix = pos[3 * (size_t)i], piy = pos[3 * (size_t)i + 1], piz = pos[3 * (size_t)i + 2];
  const bool wf = (flags & MI_EW_FORCES) != 0, wc = (flags & MI_EW_CHARGE_GRAD) != 0;
  const double two_over_sqrt_pi = 2.0 / 1.7724538509055159;
  long long beg, end;
  if (CSR) { beg = nptr[i]; end = nptr[i + 1]; } else { beg = (long long)i * M; end = beg + M; }
  double cgi = 0.0;
  T fx = 0, fy = 0, fz = 0;
  for (long long e = beg + lane; e < end; e += MI_WAVE) {
    const int j = idx[e];
    if ((!CSR && j == mask_value) || (unsigned)j >= (unsigned)N) continue;
    const double qj = (double)q[j];
    const T fs[3] = {(T)ush[3 * e], (T)ush[3 * e + 1], (T)ush[3 * e + 2]};
    T sh[3];
    rowvec_mat3(fs, cm, sh);
    const T sx = (pos[3 * (size_t)j] - pix) + sh[0], sy = (pos[3 * (size_t)j + 1] - piy) + sh[1], sz = (pos[3 * (size_t)j + 2] - piz) + sh[2];
    const double dist = (double)sqrt(sx * sx + sy * sy + sz * sz);
    if (!(dist > 1e-8)) continue;
    const double ar = al * dist;
    const double ex = exp(-(ar * ar));
    const double ec = erfc_as_poly(ar, ex);
    if (wf) {
      const double fm = (0.5 * qi * qj) * (ec / (dist * dist * dist) + two_over_sqrt_pi * al * ex / (dist * dist));
      const T fmt = (T)fm;
      const T gx = fmt * sx, gy = fmt * sy, gz = fmt * sz;
      fx -= gx; fy -= gy; fz -= gz;
      atomicAdd(&forces[3 * (size_t)j], gx); atomicAdd(&forces[3 * (size_t)j + 1], gy); atomicAdd(&forces[3 * (size_t)j + 2], gz);
    }
    if (wc) {
      const double pot = 0.5 * ec / dist;
      cgi += qj * pot;
      atomicAdd(&cgrad[j], qi * pot);
    }
  }
  if (wf) {
    fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
    if (lane == 0) { atomicAdd(&forces[3 * (size_t)i], fx); atomicAdd(&forces[3 * (size_t)i + 1], fy); atomicAdd(&forces[3 * (size_t)i + 2], fz); }
  }
  if (wc) {
    cgi = wave_sum(cgi);
    if (lane == 0) atomicAdd(&cgrad[i], cgi);
  }
}

// adjoint for L = sum_i g_i E_i (see header): same walk, same skips, weights (g_i + g_j)
template <class T, bool CSR>
__global__ __launch_bounds__(256) void ewald_real_bwd_kernel(const T* __restrict__ pos, const T* __restrict__ q, const T* __restrict__ cell,
                                                             const T* __restrict__ alpha, const int* __restrict__ batch_idx, int N,
                                                             const int* __restrict__ idx, const int* __restrict__ ush,
                                                             const int* __restrict__ nptr, int M, int mask_value, const T* __restrict__ gE,
                                                             T* __restrict__ gpos, T* __restrict__ gq, double* __restrict__ gcell,
                                                             double* __restrict__ galpha, unsigned long long* __restrict__ sym,
                                                             double* __restrict__ part) {
  const int lane = threadIdx.x & (MI_WAVE - 1);
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / MI_WAVE) + threadIdx.x / MI_WAVE);
  if (i >= N) return;
  const int s = batch_idx ? batch_idx[i] : 0;
  const double qi = (double)q[i], al = (double)alpha[s], gi = (double)gE[i];
  T cm[9];
  for (int k = 0; k < 9; ++k) cm[k] = cell[9 * (size_t)s + k];
  const T pix = pos[3 * (size_t)i], piy = pos[3 * (size_t)i + 1], piz = pos[3 * (size_t)i + 2];
  const double two_over_sqrt_pi = 2.0 / 1.7724538509055159;
  long long beg, end;
  if (CSR) { beg = nptr[i]; end = nptr[i + 1]; } else { beg = (long long)i * M; end = beg + M; }
  double gx = 0, gy = 0, gz = 0, gqi = 0, ga = 0;
  double gc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long hf = 0, hr = 0;
  for (long long e = beg + lane; e < end; e += MI_WAVE) {
    const int j = idx[e];
    if ((!CSR && j == mask_value) || (unsigned)j >= (unsigned)N) continue;  // out-of-range indices (e.g. -1 padding with another mask_value) are padding
    const double qj = (double)q[j], gj = (double)gE[j];
    const int S0 = ush[3 * e], S1 = ush[3 * e + 1], S2 = ush[3 * e + 2];
    if (sym) { hf += ew_entry_hash((unsigned)i, (unsigned)j, S0, S1, S2); hr += ew_entry_hash((unsigned)j, (unsigned)i, -S0, -S1, -S2); }
    const T fs[3] = {(T)S0, (T)S1, (T)S2};
    T sh[3];
    rowvec_mat3(fs, cm, sh);
    const T sx = (pos[3 * (size_t)j] - pix) + sh[0], sy = (pos[3 * (size_t)j + 1] - piy) + sh[1], sz = (pos[3 * (size_t)j + 2] - piz) + sh[2];
    double dist, rinv;
    ew_dist<T>(sx, sy, sz, dist, rinv);
    if (!(dist > 1e-8)) continue;
    const double ar = al * dist;
    const double ex = exp(-(ar * ar));
    const double ec = erfc_as_poly(ar, ex);
    const double rinv2 = rinv * rinv, pot = ec * rinv;
    const double fm = (0.5 * qi * qj) * (pot * rinv2 + two_over_sqrt_pi * al * ex * rinv2);
    const double wsum = gi + gj;
    gx += wsum * fm * (double)sx; gy += wsum * fm * (double)sy; gz += wsum * fm * (double)sz;
    gqi += 0.5 * wsum * qj * pot;
    if (galpha) ga += gi * (0.5 * qi * qj) * (-two_over_sqrt_pi * ex);
    if (gcell) {
      const double f = -gi * fm;
      const double sv[3] = {(double)sx, (double)sy, (double)sz};
      const double Sv[3] = {(double)S0, (double)S1, (double)S2};
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) gc[3 * a + b] += f * Sv[a] * sv[b];
    }
  }
  gx = wave_sum(gx); gy = wave_sum(gy); gz = wave_sum(gz); gqi = wave_sum(gqi);
  if (lane == 0) {
    gpos[3 * (size_t)i] = (T)gx; gpos[3 * (size_t)i + 1] = (T)gy; gpos[3 * (size_t)i + 2] = (T)gz;
    gq[i] = (T)gqi;
  }
  // per-system sums: with the partial buffer a wave adds to slot (i mod EW_BWD_SLOTS) of its system and ew_bwd_reduce_kernel folds the slots;
  // one device-scope fp64 atomic per wave on ONE address serialised 100k waves into 1.3 ms on the headline box (0.16 ms forward pass)
  double* pa = part ? part + ((size_t)s * EW_BWD_SLOTS + (i & (EW_BWD_SLOTS - 1))) * 10 : nullptr;
  if (galpha) { ga = wave_sum(ga); if (lane == 0 && ga != 0.0) atomicAdd(pa ? pa + 9 : &galpha[s], ga); }
  if (gcell) {
#pragma unroll
    for (int k = 0; k < 9; ++k) { const double v = wave_sum(gc[k]); if (lane == 0 && v != 0.0) atomicAdd(pa ? pa + k : &gcell[9 * (size_t)s + k], v); }
  }
  if (sym) ew_sym_flush(hf, hr, sym, lane, i);
}
// one block per system: fold the EW_BWD_SLOTS partial rows {cell[9], alpha} into grad_cell / grad_alpha (added: the caller zeroed them)
__global__ __launch_bounds__(EW_BWD_SLOTS) void ew_bwd_reduce_kernel(const double* __restrict__ part, double* __restrict__ gcell, double* __restrict__ galpha) {
  const int s = blockIdx.x;
  const double* row = part + ((size_t)s * EW_BWD_SLOTS + threadIdx.x) * 10;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const double v = wave_sum(row[k]);
    if (threadIdx.x == 0) {
      if (k < 9) { if (gcell) gcell[9 * (size_t)s + k] += v; }
      else if (galpha) galpha[s] += v;
    }
  }
}

// general adjoint for lists that are not symmetric: entry (i -> j) belongs to E_i only, so it carries weight g_i to BOTH ends
// (d/dr_i = +g_i fm sep, d/dr_j = -g_i fm sep; d/dq_i = g_i q_j pot, d/dq_j = g_i q_i pot); alpha / cell terms are per-owner already.
template <class T, bool CSR>
__global__ __launch_bounds__(256) void ewald_real_bwd_scatter_kernel(const T* __restrict__ pos, const T* __restrict__ q, const T* __restrict__ cell,
                                                                     const T* __restrict__ alpha, const int* __restrict__ batch_idx, int N,
                                                                     const int* __restrict__ idx, const int* __restrict__ ush,
                                                                     const int* __restrict__ nptr, int M, int mask_value, const T* __restrict__ gE,
                                                                     T* __restrict__ gpos, T* __restrict__ gq, const unsigned long long* __restrict__ sym) {
  if (sym[0] == sym[1]) return;
  const int lane = threadIdx.x & (MI_WAVE - 1);
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / MI_WAVE) + threadIdx.x / MI_WAVE);
  if (i >= N) return;
  const int s = batch_idx ? batch_idx[i] : 0;
  const double qi = (double)q[i], al = (double)alpha[s], gi = (double)gE[i];
  T cm[9];
  for (int k = 0; k < 9; ++k) cm[k] = cell[9 * (size_t)s + k];
  const T pix = pos[3 * (size_t)i], piy = pos[3 * (size_t)i + 1], piz = pos[3 * (size_t)i + 2];
  const double two_over_sqrt_pi = 2.0 / 1.7724538509055159;
  long long beg, end;
  if (CSR) { beg = nptr[i]; end = nptr[i + 1]; } else { beg = (long long)i * M; end = beg + M; }
  double gx = 0, gy = 0, gz = 0, gqi = 0;
  for (long long e = beg + lane; e < end; e += MI_WAVE) {
    const int j = idx[e];
    if ((!CSR && j == mask_value) || (unsigned)j >= (unsigned)N) continue;
    const double qj = (double)q[j];
    const T fs[3] = {(T)ush[3 * e], (T)ush[3 * e + 1], (T)ush[3 * e + 2]};
    T sh[3];
    rowvec_mat3(fs, cm, sh);
    const T sx = (pos[3 * (size_t)j] - pix) + sh[0], sy = (pos[3 * (size_t)j + 1] - piy) + sh[1], sz = (pos[3 * (size_t)j + 2] - piz) + sh[2];
    const double dist = (double)sqrt(sx * sx + sy * sy + sz * sz);
    if (!(dist > 1e-8)) continue;
    const double ar = al * dist;
    const double ex = exp(-(ar * ar));
    const double ec = erfc_as_poly(ar, ex);
    const double fm = (0.5 * qi * qj) * (ec / (dist * dist * dist) + two_over_sqrt_pi * al * ex / (dist * dist));
    const double wx = gi * fm * (double)sx, wy = gi * fm * (double)sy, wz = gi * fm * (double)sz;
    gx += wx; gy += wy; gz += wz;
    atomicAdd(&gpos[3 * (size_t)j], (T)(-wx)); atomicAdd(&gpos[3 * (size_t)j + 1], (T)(-wy)); atomicAdd(&gpos[3 * (size_t)j + 2], (T)(-wz));
    const double pot = 0.5 * ec / dist;
    gqi += gi * qj * pot;
    atomicAdd(&gq[j], (T)(gi * qi * pot));
  }
  gx = wave_sum(gx); gy = wave_sum(gy); gz = wave_sum(gz); gqi = wave_sum(gqi);
  if (lane == 0) {
    atomicAdd(&gpos[3 * (size_t)i], (T)gx); atomicAdd(&gpos[3 * (size_t)i + 1], (T)gy); atomicAdd(&gpos[3 * (size_t)i + 2], (T)gz);
    atomicAdd(&gq[i], (T)gqi);
  }
}

// ---- adjoint of the explicit FORCES (second derivatives of the pair sum) -------------------------------------------------------------
// L = sum_k w_k . F_k with the reference's force definition (every stored entry (i -> j): F_i -= fm sep, F_j += fm sep;
// ewald_kernels.py:518-544), so force-matching losses on real-space / PME forces can be differentiated (the reference gets this from
// its Warp tape: "forces" is in the grad_arrays of the `_energy_forces*` ops).  Per entry, with dw = w_j - w_i, u = dw . sep,
// g(d) = erfc(a d)/d^3 + c e^{-a^2 d^2}/d^2, c = 2a/sqrt(pi), analytic erfc' = -c e^{-a^2 d^2} (as in the energy adjoint):
//   L_e = 1/2 q_i q_j g u ;   G = dL_e/dsep = 1/2 q_i q_j (g' u sep/d + g dw),   g' = -3 erfc/d^4 - 3 c e/d^3 - 2 a^2 c e/d
//   dL/dr_j += G, dL/dr_i -= G ;  dL/dq_i += 1/2 q_j g u, dL/dq_j += 1/2 q_i g u ;  dL/dcell[a][b] += S_a G_b ;
//   dL/dalpha += 1/2 q_i q_j u dg/dalpha,  dg/dalpha = -(4 a^2/sqrt(pi)) e^{-a^2 d^2}   (the erfc and prefactor terms cancel).
// OWNER = false: entry-wise scatter with atomics, valid for every list.  OWNER = true (round 3): over a SYMMETRIC list the mirrored entry
// (j -> i, -S) has sep' = -sep, dw' = -dw, u' = u, hence G' = -G and the same charge terms with the roles swapped: atom i receives -G from
// its own entry and -G again from the mirrored one, so the row owner writes -2 sum G and 2 sum dq_i and nothing is scattered (76 M fp64
// atomics on the headline 9 A list: 1.86 -> see profiles/r03_bench_pme_train.json).  The same pass checksums the list (as mi_ewald_real
// does); if it is not symmetric the scatter variant runs behind it and overwrites positions / charges gradients; the per-system sums
// (cell, alpha) are sums over the stored entries and identical in both variants.
template <class T, bool CSR, bool OWNER>
__global__ __launch_bounds__(256) void ewald_real_force_bwd_kernel(const T* __restrict__ pos, const T* __restrict__ q, const T* __restrict__ cell,
                                                                   const T* __restrict__ alpha, const int* __restrict__ batch_idx, int N,
                                                                   const int* __restrict__ idx, const int* __restrict__ ush,
                                                                   const int* __restrict__ nptr, int M, int mask_value, const T* __restrict__ gF,
                                                                   const T* __restrict__ gC, double* __restrict__ gpos, double* __restrict__ gq,
                                                                   double* __restrict__ gcell, double* __restrict__ galpha, double* __restrict__ part,
                                                                   unsigned long long* __restrict__ sym) {
  if (!OWNER && sym && sym[0] == sym[1]) return;  // scatter variant behind the owner pass: only for lists that are not symmetric
  const int lane = threadIdx.x & (MI_WAVE - 1);
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / MI_WAVE) + threadIdx.x / MI_WAVE);
  if (i >= N) return;
  const int s = batch_idx ? batch_idx[i] : 0;
  const double qi = (double)q[i], al = (double)alpha[s];
  T cm[9];
  for (int k = 0; k < 9; ++k) cm[k] = cell[9 * (size_t)s + k];
  const T pix = pos[3 * (size_t)i], piy = pos[3 * (size_t)i + 1], piz = pos[3 * (size_t)i + 2];
  unsigned long long hf = 0, hr = 0;
  const double wix = gF ? (double)gF[3 * (size_t)i] : 0.0, wiy = gF ? (double)gF[3 * (size_t)i + 1] : 0.0, wiz = gF ? (double)gF[3 * (size_t)i + 2] : 0.0;
  const double vi = gC ? (double)gC[i] : 0.0;
  const double c = 2.0 / 1.7724538509055159 * al;
  long long beg, end;
  if (CSR) { beg = nptr[i]; end = nptr[i + 1]; } else { beg = (long long)i * M; end = beg + M; }
  double gx = 0, gy = 0, gz = 0, gqi = 0, ga = 0;
  double gc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (long long e = beg + lane; e < end; e += MI_WAVE) {
    const int j = idx[e];
    if ((!CSR && j == mask_value) || (unsigned)j >= (unsigned)N) continue;
    const double qj = (double)q[j];
    const int S0 = ush[3 * e], S1 = ush[3 * e + 1], S2 = ush[3 * e + 2];
    if (OWNER && sym) { hf += ew_entry_hash((unsigned)i, (unsigned)j, S0, S1, S2); hr += ew_entry_hash((unsigned)j, (unsigned)i, -S0, -S1, -S2); }
    const T fs[3] = {(T)S0, (T)S1, (T)S2};
    T sh[3];
    rowvec_mat3(fs, cm, sh);
    const T sxT = (pos[3 * (size_t)j] - pix) + sh[0], syT = (pos[3 * (size_t)j + 1] - piy) + sh[1], szT = (pos[3 * (size_t)j + 2] - piz) + sh[2];
    double d, ri;
    ew_dist<T>(sxT, syT, szT, d, ri);
    if (!(d > 1e-8)) continue;
    const double sx = (double)sxT, sy = (double)syT, sz = (double)szT;
    const double ar = al * d, ex = exp(-(ar * ar)), ec = erfc_as_poly(ar, ex);
    const double ri2 = ri * ri, ri3 = ri2 * ri;
    const double g = ec * ri3 + c * ex * ri2;
    const double gp = -3.0 * ec * (ri2 * ri2) - 3.0 * c * ex * ri3 - 2.0 * al * al * c * ex * ri;
    double Gx = 0.0, Gy = 0.0, Gz = 0.0, dqi = 0.0, dqj = 0.0;
    if (gF) {
      const double dwx = (double)gF[3 * (size_t)j] - wix, dwy = (double)gF[3 * (size_t)j + 1] - wiy, dwz = (double)gF[3 * (size_t)j + 2] - wiz;
      const double u = dwx * sx + dwy * sy + dwz * sz;
      const double hq = 0.5 * qi * qj;
      const double k1 = hq * gp * u * ri;
      Gx = k1 * sx + hq * g * dwx; Gy = k1 * sy + hq * g * dwy; Gz = k1 * sz + hq * g * dwz;
      dqi = 0.5 * qj * g * u;
      dqj = 0.5 * qi * g * u;
      if (galpha) ga += hq * u * (-(4.0 * al * al / 1.7724538509055159) * ex);
    }
    if (gC) {
      // charge-gradient outputs: the entry adds 1/2 q_j phi to cg_i and 1/2 q_i phi to cg_j, phi = erfc(a d)/d, so for L = sum_k v_k cg_k
      //   L_e = 1/2 phi A, A = v_i q_j + v_j q_i ;  dL_e/dsep = 1/2 A phi' sep/d = -1/2 A g sep ;  dL_e/dalpha = -1/2 A (2/sqrt(pi)) e^{-a^2 d^2}
      const double vj = (double)gC[j], phi = ec * ri, A = vi * qj + vj * qi;
      const double k2 = -0.5 * A * g;
      Gx += k2 * sx; Gy += k2 * sy; Gz += k2 * sz;
      dqi += 0.5 * phi * vj;
      dqj += 0.5 * phi * vi;
      if (galpha) ga += -0.5 * A * (2.0 / 1.7724538509055159) * ex;
    }
    gx -= Gx; gy -= Gy; gz -= Gz;
    gqi += dqi;
    if (!OWNER) {
      atomicAdd(&gpos[3 * (size_t)j], Gx); atomicAdd(&gpos[3 * (size_t)j + 1], Gy); atomicAdd(&gpos[3 * (size_t)j + 2], Gz);
      atomicAdd(&gq[j], dqj);
    }
    if (gcell) {
      const double Sv[3] = {(double)S0, (double)S1, (double)S2}, Gv[3] = {Gx, Gy, Gz};
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) gc[3 * a + b] += Sv[a] * Gv[b];
    }
  }
  gx = wave_sum(gx); gy = wave_sum(gy); gz = wave_sum(gz); gqi = wave_sum(gqi);
  if (lane == 0) {
    if (OWNER) {
      gpos[3 * (size_t)i] = 2.0 * gx; gpos[3 * (size_t)i + 1] = 2.0 * gy; gpos[3 * (size_t)i + 2] = 2.0 * gz;
      gq[i] = 2.0 * gqi;
    } else {
      atomicAdd(&gpos[3 * (size_t)i], gx); atomicAdd(&gpos[3 * (size_t)i + 1], gy); atomicAdd(&gpos[3 * (size_t)i + 2], gz);
      atomicAdd(&gq[i], gqi);
    }
  }
  if (OWNER && sym) ew_sym_flush(hf, hr, sym, lane, i);
  double* pa = part ? part + ((size_t)s * EW_BWD_SLOTS + (i & (EW_BWD_SLOTS - 1))) * 10 : nullptr;  // slotted per-system sums (see ewald_real_bwd_kernel)
  if (galpha) { ga = wave_sum(ga); if (lane == 0 && ga != 0.0) atomicAdd(pa ? pa + 9 : &galpha[s], ga); }
  if (gcell) {
#pragma unroll
    for (int k = 0; k < 9; ++k) { const double v = wave_sum(gc[k]); if (lane == 0 && v != 0.0) atomicAdd(pa ? pa + k : &gcell[9 * (size_t)s + k], v); }
  }
}

// ---- explicit-k reciprocal space (ewald_kernels.py:1496-2480) -----------------------------------------------
// The reference stores cos/sin(k.r) for every (k, atom) in two float64 [K,N] tables between its two kernels; here both passes
// recompute the phases (fp64 sincos from registers) so HBM sees only positions, k-vectors and the [B,K] structure factors.
constexpr int EK_TILE = 16;       // 16 k-vectors (or atoms) x 16 lanes per 256-thread block
constexpr int EK_ATOM_CHUNK = 4096;

__device__ __forceinline__ double sum16(double v) {
  v += __shfl_xor(v, 8, 16); v += __shfl_xor(v, 4, 16); v += __shfl_xor(v, 2, 16); v += __shfl_xor(v, 1, 16);
  return v;
}

// S[b][k] = G_k * sum_j w_j exp(i k.r_j),  G_k = 8 pi / V * exp(-k^2 / 4 alpha^2) / k^2   (half-space k set, hence 8 pi)
template <class T>
__global__ __launch_bounds__(256) void ewald_sf_kernel(const T* __restrict__ pos, const T* __restrict__ w, const T* __restrict__ kvec,
                                                       const T* __restrict__ cell, const T* __restrict__ alpha,
                                                       const int* __restrict__ system_ptr, int n_atoms, int K,
                                                       double* __restrict__ sf, double* __restrict__ total_charge) {
  const int b = blockIdx.y, k = blockIdx.x * EK_TILE + (threadIdx.x >> 4), lane = threadIdx.x & 15;
  const int a0 = system_ptr ? system_ptr[b] : 0, a1 = system_ptr ? system_ptr[b + 1] : n_atoms;
  const int c0 = a0 + blockIdx.z * EK_ATOM_CHUNK, c1 = min(c0 + EK_ATOM_CHUNK, a1);
  if (c0 >= a1) return;
  const T* cm = cell + 9 * (size_t)b;
  const double vol = fabs((double)(cm[0] * (cm[4] * cm[8] - cm[5] * cm[7]) - cm[1] * (cm[3] * cm[8] - cm[5] * cm[6]) +
                                   cm[2] * (cm[3] * cm[7] - cm[4] * cm[6])));
  const bool charge_owner = total_charge && blockIdx.x == 0 && (threadIdx.x >> 4) == 1 && K > 1;  // the reference's k_idx == 1 thread
  const bool live = k < K;
  double kx = 0, ky = 0, kz = 0;
  if (live) { const T* kv = kvec + 3 * ((size_t)b * K + k); kx = kv[0]; ky = kv[1]; kz = kv[2]; }
  const double k2 = kx * kx + ky * ky + kz * kz;
  double re = 0, im = 0, qs = 0;
  for (int a = c0 + lane; a < c1; a += 16) {
    const double q = w[a];
    const double ph = kx * (double)pos[3 * (size_t)a] + ky * (double)pos[3 * (size_t)a + 1] + kz * (double)pos[3 * (size_t)a + 2];
    double sn, cs;
    sincos(ph, &sn, &cs);
    re += q * cs; im += q * sn; qs += q;
  }
  re = sum16(re); im = sum16(im); qs = sum16(qs);
  if (lane != 0) return;
  if (charge_owner) atomicAdd(total_charge + b, qs / vol);
  if (!live || k2 < 1e-10) return;
  const double al = alpha[b];
  const double green = exp(-k2 * (0.25 / (al * al))) / k2 * (8.0 * M_PI) / vol;
  atomicAdd(sf + 2 * ((size_t)b * K + k), re * green);
  atomicAdd(sf + 2 * ((size_t)b * K + k) + 1, im * green);
}

// per atom: phi_i = sum_k (S_re cos + S_im sin)(k.r_i),  kf_i = sum_k (S_re sin - S_im cos) k ; then
// E_i = q phi/2 - alpha q^2/sqrt(pi) - pi q Q / (2 alpha^2),  F_i = q kf_i,  dE/dq_i = phi - 2 alpha q/sqrt(pi) - pi Q/alpha^2
template <class T>
__global__ __launch_bounds__(256) void ewald_recip_gather_kernel(const T* __restrict__ pos, const T* __restrict__ q, const T* __restrict__ kvec,
                                                                 const T* __restrict__ alpha, const int* __restrict__ batch_idx,
                                                                 const double* __restrict__ sf, const double* __restrict__ total_charge,
                                                                 int n_atoms, int K, double* __restrict__ potential,
                                                                 double* __restrict__ kforce, double* __restrict__ energies,
                                                                 T* __restrict__ forces, double* __restrict__ charge_grads) {
  const int i = blockIdx.x * EK_TILE + (threadIdx.x >> 4), lane = threadIdx.x & 15;
  const bool live = i < n_atoms;
  const int ii = live ? i : n_atoms - 1;
  const int b = batch_idx ? batch_idx[ii] : 0;
  const double x = pos[3 * (size_t)ii], y = pos[3 * (size_t)ii + 1], z = pos[3 * (size_t)ii + 2];
  const T* kv = kvec + 3 * (size_t)b * K;
  const double* s = sf + 2 * (size_t)b * K;
  double phi = 0, fx = 0, fy = 0, fz = 0;
  for (int k = lane; k < K; k += 16) {
    const double kx = kv[3 * k], ky = kv[3 * k + 1], kz = kv[3 * k + 2];
    const double sre = s[2 * k], sim = s[2 * k + 1];
    double sn, cs;
    sincos(kx * x + ky * y + kz * z, &sn, &cs);
    phi += sre * cs + sim * sn;
    const double fs = sre * sn - sim * cs;
    fx += fs * kx; fy += fs * ky; fz += fs * kz;
  }
  phi = sum16(phi); fx = sum16(fx); fy = sum16(fy); fz = sum16(fz);
  if (lane != 0 || !live) return;
  if (potential) potential[i] = phi;
  if (kforce) { kforce[3 * (size_t)i] = fx; kforce[3 * (size_t)i + 1] = fy; kforce[3 * (size_t)i + 2] = fz; }
  const double qi = q[i], al = alpha[b], Q = total_charge ? total_charge[b] : 0.0;
  if (energies) energies[i] = 0.5 * qi * phi - al * qi * qi / sqrt(M_PI) - M_PI * qi * Q / (2.0 * al * al);
  if (forces) { forces[3 * (size_t)i] = (T)(qi * fx); forces[3 * (size_t)i + 1] = (T)(qi * fy); forces[3 * (size_t)i + 2] = (T)(qi * fz); }
  if (charge_grads) charge_grads[i] = phi - 2.0 * al / sqrt(M_PI) * qi - M_PI / (al * al) * Q;
}


// kk_i = sum_k k (w_i . k) (S_re cos + S_im sin)(k.r_i): the position derivative of sum_i w_i . kf_i[S] at fixed S -- one of the terms of
// the adjoint of the explicit FORCES (nvalchemiops/interactions/electrostatics/ewald.py `_recip_outputs_adjoint`)
template <class T>
__global__ __launch_bounds__(256) void ewald_recip_gather_kk_kernel(const T* __restrict__ pos, const T* __restrict__ kvec, const int* __restrict__ batch_idx,
                                                                    const double* __restrict__ sf, const double* __restrict__ wvec, int n_atoms, int K,
                                                                    double* __restrict__ out) {
  const int i = blockIdx.x * EK_TILE + (threadIdx.x >> 4), lane = threadIdx.x & 15;
  const bool live = i < n_atoms;
  const int ii = live ? i : n_atoms - 1;
  const int b = batch_idx ? batch_idx[ii] : 0;
  const double x = pos[3 * (size_t)ii], y = pos[3 * (size_t)ii + 1], z = pos[3 * (size_t)ii + 2];
  const double wx = wvec[3 * (size_t)ii], wy = wvec[3 * (size_t)ii + 1], wz = wvec[3 * (size_t)ii + 2];
  const T* kv = kvec + 3 * (size_t)b * K;
  const double* s = sf + 2 * (size_t)b * K;
  double ox = 0, oy = 0, oz = 0;
  for (int k = lane; k < K; k += 16) {
    const double kx = kv[3 * k], ky = kv[3 * k + 1], kz = kv[3 * k + 2];
    double sn, cs;
    sincos(kx * x + ky * y + kz * z, &sn, &cs);
    const double t = (wx * kx + wy * ky + wz * kz) * (s[2 * k] * cs + s[2 * k + 1] * sn);
    ox += t * kx; oy += t * ky; oz += t * kz;
  }
  ox = sum16(ox); oy = sum16(oy); oz = sum16(oz);
  if (lane != 0 || !live) return;
  out[3 * (size_t)i] = ox; out[3 * (size_t)i + 1] = oy; out[3 * (size_t)i + 2] = oz;
}


// ---- cut-off Coulomb (coulomb.py:133-708): fp64 throughout, arbitrary (full, half or asymmetric) lists ----------------------------
// Pair walk of the eight reference kernels: r_ij = r_i - r_j - cell^T S, skip r >= cutoff or r < 1e-10, matrix entries
// j >= fill_value or j >= N are padding (:325, :385), alpha > 0 selects the erfc_AS-damped form.  Unlike the Ewald kernel above the
// reference op is used with lists that need not be symmetric, so forces follow its scatter exactly: +f_ij to the row owner and
// -f_ij to atom j (fp64 global atomics; only the owner's wave sum and the j-scatter touch memory).  `energy_prefactor` is 1/2
// everywhere except the reference's energy-only MATRIX kernels, which use q_i q_j without the 1/2 (:340, :623) -- the host passes
// 1.0 there so that results stay identical to the reference's on the same inputs.
__device__ __forceinline__ bool coulomb_pair(const double* __restrict__ pos, const double* cm, double pix, double piy, double piz, int j,
                                             const int* __restrict__ ush, long long e, double cutoff, double al, double& rx, double& ry,
                                             double& rz, double& phi, double& fmr) {
  const double s0 = (double)ush[3 * e], s1 = (double)ush[3 * e + 1], s2 = (double)ush[3 * e + 2];
  rx = pix - pos[3 * (size_t)j] - (cm[0] * s0 + cm[3] * s1 + cm[6] * s2);
  ry = piy - pos[3 * (size_t)j + 1] - (cm[1] * s0 + cm[4] * s1 + cm[7] * s2);
  rz = piz - pos[3 * (size_t)j + 2] - (cm[2] * s0 + cm[5] * s1 + cm[8] * s2);
  const double r = sqrt(rx * rx + ry * ry + rz * rz);
  if (r >= cutoff || r < 1e-10) return false;
  if (al > 0.0) {
    const double ar = al * r, ex = exp(-(ar * ar)), ec = erfc_as_poly(ar, ex);
    phi = ec / r;
    fmr = ec / (r * r * r) + 1.1283791670955126 * al * ex / (r * r);
  } else {
    phi = 1.0 / r;
    fmr = 1.0 / (r * r * r);
  }
  return true;
}

template <bool CSR>
__global__ __launch_bounds__(256) void coulomb_kernel(const double* __restrict__ pos, const double* __restrict__ q, const double* __restrict__ cell,
                                                      const int* __restrict__ batch_idx, int N, const int* __restrict__ idx,
                                                      const int* __restrict__ ush, const int* __restrict__ nptr, int M, int fill_value,
                                                      double cutoff, double al, double epref, double* __restrict__ energies,
                                                      double* __restrict__ forces) {
  const int lane = threadIdx.x & (MI_WAVE - 1);
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / MI_WAVE) + threadIdx.x / MI_WAVE);
  if (i >= N) return;
  const int s = batch_idx ? batch_idx[i] : 0;
  double cm[9];
  for (int k = 0; k < 9; ++k) cm[k] = cell[9 * (size_t)s + k];
  const double qi = q[i], pix = pos[3 * (size_t)i], piy = pos[3 * (size_t)i + 1], piz = pos[3 * (size_t)i + 2];
  long long beg, end;
  if (CSR) { beg = nptr[i]; end = nptr[i + 1]; } else { beg = (long long)i * M; end = beg + M; }
  double eacc = 0.0, fx = 0.0, fy = 0.0, fz = 0.0;
  for (long long e = beg + lane; e < end; e += MI_WAVE) {
    const int j = idx[e];
    if (j < 0 || j >= N || (!CSR && j >= fill_value)) continue;
    double rx, ry, rz, phi, fmr;
    if (!coulomb_pair(pos, cm, pix, piy, piz, j, ush, e, cutoff, al, rx, ry, rz, phi, fmr)) continue;
    const double qq = qi * q[j];
    eacc += epref * qq * phi;
    if (forces) {
      const double fm = 0.5 * qq * fmr;
      const double gx = fm * rx, gy = fm * ry, gz = fm * rz;
      fx += gx; fy += gy; fz += gz;
      atomicAdd(&forces[3 * (size_t)j], -gx);
      atomicAdd(&forces[3 * (size_t)j + 1], -gy);
      atomicAdd(&forces[3 * (size_t)j + 2], -gz);
    }
  }
  eacc = wave_sum(eacc);
  if (lane == 0) energies[i] = eacc;
  if (forces) {
    fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
    if (lane == 0) { atomicAdd(&forces[3 * (size_t)i], fx); atomicAdd(&forces[3 * (size_t)i + 1], fy); atomicAdd(&forces[3 * (size_t)i + 2], fz); }
  }
}

// adjoint of the per-atom energies for L = sum_i g_i E_i: every directed entry (i, j) contributes g_i * epref * q_i q_j phi(r_ij)
template <bool CSR>
__global__ __launch_bounds__(256) void coulomb_bwd_kernel(const double* __restrict__ pos, const double* __restrict__ q, const double* __restrict__ cell,
                                                          const int* __restrict__ batch_idx, int N, const int* __restrict__ idx,
                                                          const int* __restrict__ ush, const int* __restrict__ nptr, int M, int fill_value,
                                                          double cutoff, double al, double epref, const double* __restrict__ gE,
                                                          double* __restrict__ gpos, double* __restrict__ gq, double* __restrict__ gcell) {
  const int lane = threadIdx.x & (MI_WAVE - 1);
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / MI_WAVE) + threadIdx.x / MI_WAVE);
  if (i >= N) return;
  const int s = batch_idx ? batch_idx[i] : 0;
  double cm[9];
  for (int k = 0; k < 9; ++k) cm[k] = cell[9 * (size_t)s + k];
  const double qi = q[i], gi = gE[i] * epref, pix = pos[3 * (size_t)i], piy = pos[3 * (size_t)i + 1], piz = pos[3 * (size_t)i + 2];
  long long beg, end;
  if (CSR) { beg = nptr[i]; end = nptr[i + 1]; } else { beg = (long long)i * M; end = beg + M; }
  double px = 0.0, py = 0.0, pz = 0.0, cq = 0.0, gc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (long long e = beg + lane; e < end; e += MI_WAVE) {
    const int j = idx[e];
    if (j < 0 || j >= N || (!CSR && j >= fill_value)) continue;
    double rx, ry, rz, phi, fmr;
    if (!coulomb_pair(pos, cm, pix, piy, piz, j, ush, e, cutoff, al, rx, ry, rz, phi, fmr)) continue;
    const double qj = q[j];
    cq += gi * qj * phi;
    atomicAdd(&gq[j], gi * qi * phi);
    const double w = -gi * qi * qj * fmr;  // dL/dr_ij = w * r_ij
    const double wx = w * rx, wy = w * ry, wz = w * rz;
    px += wx; py += wy; pz += wz;
    atomicAdd(&gpos[3 * (size_t)j], -wx);
    atomicAdd(&gpos[3 * (size_t)j + 1], -wy);
    atomicAdd(&gpos[3 * (size_t)j + 2], -wz);
    const double sv[3] = {(double)ush[3 * e], (double)ush[3 * e + 1], (double)ush[3 * e + 2]};
    for (int a = 0; a < 3; ++a) { gc[3 * a] -= sv[a] * wx; gc[3 * a + 1] -= sv[a] * wy; gc[3 * a + 2] -= sv[a] * wz; }
  }
  px = wave_sum(px); py = wave_sum(py); pz = wave_sum(pz); cq = wave_sum(cq);
  bool any_c = false;
  for (int k = 0; k < 9; ++k) { gc[k] = wave_sum(gc[k]); any_c |= gc[k] != 0.0; }
  if (lane == 0) {
    atomicAdd(&gpos[3 * (size_t)i], px); atomicAdd(&gpos[3 * (size_t)i + 1], py); atomicAdd(&gpos[3 * (size_t)i + 2], pz);
    atomicAdd(&gq[i], cq);
    if (any_c) for (int k = 0; k < 9; ++k) atomicAdd(&gcell[9 * (size_t)s + k], gc[k]);
  }
}

// Adjoint of the explicit FORCES (force-matching on `coulomb_energy_forces` / `coulomb_forces`): L = sum_k w_k . F_k.  Entry (i -> j)
// adds f = 1/2 q_i q_j g(r) r_ij to F_i and -f to F_j, g = fmr of `coulomb_pair`, so with dw = w_i - w_j and u = r_ij . dw
//   L_e = 1/2 q_i q_j g u ;  G = dL_e/dr_ij = 1/2 q_i q_j (g' u r_ij / r + g dw),  g' = -3 erfc/r^4 - 3 c e/r^3 - 2 a^2 c e/r  (-3/r^4 undamped)
//   dL/dr_i += G, dL/dr_j -= G ;  dL/dq_i += 1/2 q_j g u, dL/dq_j += 1/2 q_i g u ;  dL/dcell[a][b] -= S_a G_b   (r_ij = r_i - r_j - cell^T S).
// The cutoff is a step: it has no derivative.  Entry-wise scatter, any list.
template <bool CSR>
__global__ __launch_bounds__(256) void coulomb_force_bwd_kernel(const double* __restrict__ pos, const double* __restrict__ q,
                                                                const double* __restrict__ cell, const int* __restrict__ batch_idx, int N,
                                                                const int* __restrict__ idx, const int* __restrict__ ush,
                                                                const int* __restrict__ nptr, int M, int fill_value, double cutoff, double al,
                                                                const double* __restrict__ gF, double* __restrict__ gpos,
                                                                double* __restrict__ gq, double* __restrict__ gcell) {
  const int lane = threadIdx.x & (MI_WAVE - 1);
  const int i = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / MI_WAVE) + threadIdx.x / MI_WAVE);
  if (i >= N) return;
  const int s = batch_idx ? batch_idx[i] : 0;
  double cm[9];
  for (int k = 0; k < 9; ++k) cm[k] = cell[9 * (size_t)s + k];
  const double qi = q[i], pix = pos[3 * (size_t)i], piy = pos[3 * (size_t)i + 1], piz = pos[3 * (size_t)i + 2];
  const double wix = gF[3 * (size_t)i], wiy = gF[3 * (size_t)i + 1], wiz = gF[3 * (size_t)i + 2];
  const double c = 1.1283791670955126 * al;
  long long beg, end;
  if (CSR) { beg = nptr[i]; end = nptr[i + 1]; } else { beg = (long long)i * M; end = beg + M; }
  double px = 0.0, py = 0.0, pz = 0.0, cq = 0.0, gc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (long long e = beg + lane; e < end; e += MI_WAVE) {
    const int j = idx[e];
    if (j < 0 || j >= N || (!CSR && j >= fill_value)) continue;
    double rx, ry, rz, phi, g;
    if (!coulomb_pair(pos, cm, pix, piy, piz, j, ush, e, cutoff, al, rx, ry, rz, phi, g)) continue;
    const double r2 = rx * rx + ry * ry + rz * rz, r = sqrt(r2);
    double gp;
    if (al > 0.0) {
      const double ar = al * r, ex = exp(-(ar * ar)), ec = erfc_as_poly(ar, ex);
      gp = -3.0 * ec / (r2 * r2) - 3.0 * c * ex / (r2 * r) - 2.0 * al * al * c * ex / r;
    } else {
      gp = -3.0 / (r2 * r2);
    }
    const double qj = q[j];
    const double dwx = wix - gF[3 * (size_t)j], dwy = wiy - gF[3 * (size_t)j + 1], dwz = wiz - gF[3 * (size_t)j + 2];
    const double u = rx * dwx + ry * dwy + rz * dwz;
    const double hq = 0.5 * qi * qj, k1 = hq * gp * u / r;
    const double Gx = k1 * rx + hq * g * dwx, Gy = k1 * ry + hq * g * dwy, Gz = k1 * rz + hq * g * dwz;
    px += Gx; py += Gy; pz += Gz;
    atomicAdd(&gpos[3 * (size_t)j], -Gx); atomicAdd(&gpos[3 * (size_t)j + 1], -Gy); atomicAdd(&gpos[3 * (size_t)j + 2], -Gz);
    cq += 0.5 * qj * g * u;
    atomicAdd(&gq[j], 0.5 * qi * g * u);
    const double sv[3] = {(double)ush[3 * e], (double)ush[3 * e + 1], (double)ush[3 * e + 2]};
    for (int a = 0; a < 3; ++a) { gc[3 * a] -= sv[a] * Gx; gc[3 * a + 1] -= sv[a] * Gy; gc[3 * a + 2] -= sv[a] * Gz; }
  }
  px = wave_sum(px); py = wave_sum(py); pz = wave_sum(pz); cq = wave_sum(cq);
  bool any_c = false;
  for (int k = 0; k < 9; ++k) { gc[k] = wave_sum(gc[k]); any_c |= gc[k] != 0.0; }
  if (lane == 0) {
    atomicAdd(&gpos[3 * (size_t)i], px); atomicAdd(&gpos[3 * (size_t)i + 1], py); atomicAdd(&gpos[3 * (size_t)i + 2], pz);
    atomicAdd(&gq[i], cq);
    if (any_c) for (int k = 0; k < 9; ++k) atomicAdd(&gcell[9 * (size_t)s + k], gc[k]);
  }
}

}  // namespace

extern "C" size_t mi_ewald_real_bwd_scratch_bytes(int n_systems);
extern "C" int mi_ewald_real_forces_bwd(const void* positions, const void* charges, const void* cell, const void* alpha, const int32_t* batch_idx,
                                        int n_atoms, int n_systems, int dtype, const int32_t* idx_j, const int32_t* unit_shifts,
                                        const int32_t* neighbor_ptr, int max_neighbors, int mask_value, const void* grad_forces,
                                        const void* grad_charge_grads, double* grad_positions, double* grad_charges, double* grad_cell,
                                        double* grad_alpha, void* scratch, size_t scratch_bytes, void* stream) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype");
  MI_REQUIRE(n_systems >= 1 && grad_positions && grad_charges, "null gradient outputs");
  hipStream_t st = (hipStream_t)stream;
  if (n_atoms > 0) {
    MI_HIP_CHECK(hipMemsetAsync(grad_positions, 0, sizeof(double) * 3 * (size_t)n_atoms, st));
    MI_HIP_CHECK(hipMemsetAsync(grad_charges, 0, sizeof(double) * (size_t)n_atoms, st));
  }
  if (grad_cell) MI_HIP_CHECK(hipMemsetAsync(grad_cell, 0, sizeof(double) * 9 * (size_t)n_systems, st));
  if (grad_alpha) MI_HIP_CHECK(hipMemsetAsync(grad_alpha, 0, sizeof(double) * (size_t)n_systems, st));
  if (n_atoms <= 0) return MI_OK;
  MI_REQUIRE(positions && charges && cell && alpha && idx_j && unit_shifts && (grad_forces || grad_charge_grads), "null pointer");
  const int blocks = mi_blocks(n_atoms, 4);
  const bool csr = neighbor_ptr != nullptr;
#define MI_EFB_ARGS(T_) (const T_*)positions, (const T_*)charges, (const T_*)cell, (const T_*)alpha, batch_idx, n_atoms, idx_j, unit_shifts, neighbor_ptr, \
                        max_neighbors, mask_value, (const T_*)grad_forces, (const T_*)grad_charge_grads, grad_positions, grad_charges
#define MI_EFB(T_, CSR_)                                                                                                                       \
  do {                                                                                                                                         \
    if (sym) {  /* owner pass + checksums; the scatter variant behind it only works when the list turned out not to be symmetric */          \
      ewald_real_force_bwd_kernel<T_, CSR_, true><<<blocks, 256, 0, st>>>(MI_EFB_ARGS(T_), grad_cell, grad_alpha, part, sym);                  \
      ewald_fixup_zero_kernel<double><<<mi_blocks(n_atoms, 256), 256, 0, st>>>(sym, grad_positions, grad_charges, nullptr, n_atoms);            \
      ewald_real_force_bwd_kernel<T_, CSR_, false><<<blocks, 256, 0, st>>>(MI_EFB_ARGS(T_), nullptr, nullptr, nullptr, sym);                    \
    } else {                                                                                                                                   \
      ewald_real_force_bwd_kernel<T_, CSR_, false><<<blocks, 256, 0, st>>>(MI_EFB_ARGS(T_), grad_cell, grad_alpha, part, nullptr);              \
    }                                                                                                                                          \
  } while (0)
  // optional scratch (mi_ewald_real_bwd_scratch_bytes(n_systems); only its partial-sum part is used here): slotted per-system sums
  double* part = nullptr;
  unsigned long long* sym = nullptr;
  if (scratch && scratch_bytes >= sizeof(unsigned long long) * EW_SYM_WORDS) {
    sym = reinterpret_cast<unsigned long long*>(scratch);
    MI_HIP_CHECK(hipMemsetAsync(sym, 0, sizeof(unsigned long long) * EW_SYM_WORDS, st));
  }
  if (sym && (grad_cell || grad_alpha) && scratch_bytes >= mi_ewald_real_bwd_scratch_bytes(n_systems)) {
    part = reinterpret_cast<double*>(sym + EW_SYM_WORDS);
    MI_HIP_CHECK(hipMemsetAsync(part, 0, sizeof(double) * 10 * EW_BWD_SLOTS * (size_t)n_systems, st));
  }
  mi_timing_begin("ewald_real_forces_bwd", stream);
  if (dtype == MI_F32) { if (csr) MI_EFB(float, true); else MI_EFB(float, false); }
  else { if (csr) MI_EFB(double, true); else MI_EFB(double, false); }
  if (part) ew_bwd_reduce_kernel<<<n_systems, EW_BWD_SLOTS, 0, st>>>(part, grad_cell, grad_alpha);
  mi_timing_end(stream);
#undef MI_EFB
#undef MI_EFB_ARGS
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" size_t mi_ewald_symmetry_scratch_bytes(void) { return sizeof(unsigned long long) * EW_SYM_WORDS; }
extern "C" size_t mi_ewald_real_scratch_bytes(int n_atoms, int dtype) {
  return sizeof(unsigned long long) * EW_SYM_WORDS + (dtype == MI_F32 ? 16 : 32) * (size_t)(n_atoms > 0 ? n_atoms : 0);
}

extern "C" size_t mi_ewald_real_bwd_scratch_bytes(int n_systems) {
  return sizeof(unsigned long long) * EW_SYM_WORDS + sizeof(double) * 10 * EW_BWD_SLOTS * (size_t)(n_systems > 0 ? n_systems : 1);
}

extern "C" int mi_ewald_real_bwd(const void* positions, const void* charges, const void* cell, const void* alpha, const int32_t* batch_idx,
                                 int n_atoms, int n_systems, int dtype, const int32_t* idx_j, const int32_t* unit_shifts, const int32_t* neighbor_ptr,
                                 int max_neighbors, int mask_value, const void* grad_energies, void* grad_positions, void* grad_charges,
                                 double* grad_cell, double* grad_alpha, void* symmetry_scratch, size_t scratch_bytes, void* stream) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype");
  if (n_atoms <= 0) return MI_OK;
  MI_REQUIRE(positions && charges && cell && alpha && idx_j && unit_shifts && grad_energies && grad_positions && grad_charges, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = mi_blocks(n_atoms, 4);
  const bool csr = neighbor_ptr != nullptr;
#define MI_EWB(T_, CSR_)                                                                                                                       \
  ewald_real_bwd_kernel<T_, CSR_><<<blocks, 256, 0, st>>>((const T_*)positions, (const T_*)charges, (const T_*)cell, (const T_*)alpha, batch_idx, \
                                                          n_atoms, idx_j, unit_shifts, neighbor_ptr, max_neighbors, mask_value,                  \
                                                          (const T_*)grad_energies, (T_*)grad_positions, (T_*)grad_charges, grad_cell, grad_alpha, sym, part)
#define MI_EWBS(T_, CSR_)                                                                                                                          \
  do {                                                                                                                                             \
    ewald_fixup_zero_kernel<T_><<<mi_blocks(n_atoms, 256), 256, 0, st>>>(sym, (T_*)grad_positions, (T_*)grad_charges, nullptr, n_atoms);          \
    ewald_real_bwd_scatter_kernel<T_, CSR_><<<blocks, 256, 0, st>>>((const T_*)positions, (const T_*)charges, (const T_*)cell, (const T_*)alpha,    \
                                                                    batch_idx, n_atoms, idx_j, unit_shifts, neighbor_ptr, max_neighbors, mask_value, \
                                                                    (const T_*)grad_energies, (T_*)grad_positions, (T_*)grad_charges, sym);        \
  } while (0)
  unsigned long long* sym = (unsigned long long*)symmetry_scratch;
  MI_REQUIRE(!sym || scratch_bytes >= sizeof(unsigned long long) * EW_SYM_WORDS, "symmetry scratch smaller than mi_ewald_symmetry_scratch_bytes()");
  if (sym) MI_HIP_CHECK(hipMemsetAsync(sym, 0, sizeof(unsigned long long) * EW_SYM_WORDS, st));
  // second part of the scratch (when the caller sized it with mi_ewald_real_bwd_scratch_bytes): slotted partials of the per-system sums
  double* part = nullptr;
  if (sym && n_systems >= 1 && (grad_cell || grad_alpha) && scratch_bytes >= mi_ewald_real_bwd_scratch_bytes(n_systems)) {
    part = reinterpret_cast<double*>(sym + EW_SYM_WORDS);
    MI_HIP_CHECK(hipMemsetAsync(part, 0, sizeof(double) * 10 * EW_BWD_SLOTS * (size_t)n_systems, st));
  }
  mi_timing_begin("ewald_real_bwd", stream);
  if (dtype == MI_F32) { if (csr) MI_EWB(float, true); else MI_EWB(float, false); }
  else { if (csr) MI_EWB(double, true); else MI_EWB(double, false); }
  if (part) ew_bwd_reduce_kernel<<<n_systems, EW_BWD_SLOTS, 0, st>>>(part, grad_cell, grad_alpha);
  mi_timing_end(stream);
  MI_LAUNCH_CHECK();
  if (sym) {
    if (dtype == MI_F32) { if (csr) MI_EWBS(float, true); else MI_EWBS(float, false); }
    else { if (csr) MI_EWBS(double, true); else MI_EWBS(double, false); }
    MI_LAUNCH_CHECK();
  }
#undef MI_EWB
#undef MI_EWBS
  return MI_OK;
}

extern "C" int mi_ewald_real(const void* positions, const void* charges, const void* cell, const void* alpha, const int32_t* batch_idx,
                             int n_atoms, int dtype, const int32_t* idx_j, const int32_t* unit_shifts, const int32_t* neighbor_ptr,
                             int max_neighbors, int mask_value, int flags, double* energies, void* forces, double* charge_grads,
                             void* scratch, size_t scratch_bytes, void* stream) {
  return mi_ewald_real_listed(positions, charges, cell, alpha, batch_idx, n_atoms, dtype, idx_j, unit_shifts, neighbor_ptr, max_neighbors, mask_value, flags,
                              energies, forces, charge_grads, scratch, scratch_bytes, nullptr, 0, 0, stream);
}
extern "C" int mi_ewald_real_listed(const void* positions, const void* charges, const void* cell, const void* alpha, const int32_t* batch_idx,
                                    int n_atoms, int dtype, const int32_t* idx_j, const int32_t* unit_shifts, const int32_t* neighbor_ptr,
                                    int max_neighbors, int mask_value, int flags, double* energies, void* forces, double* charge_grads,
                                    void* scratch, size_t scratch_bytes, const int32_t* search_num_neighbors, int verify_stride, int verify_phase,
                                    void* stream) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype");
  if (n_atoms <= 0) return MI_OK;
  MI_REQUIRE(positions && charges && cell && alpha && idx_j && unit_shifts && energies, "null pointer");
  MI_REQUIRE(!(flags & MI_EW_FORCES) || forces, "forces output");
  MI_REQUIRE(!(flags & MI_EW_CHARGE_GRAD) || charge_grads, "charge gradient output");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = mi_blocks(n_atoms, 4);
  const bool csr = neighbor_ptr != nullptr;
#define MI_EW(T_, CSR_)                                                                                                                        \
  do {                                                                                                                                         \
    if (rec) {                                                                                                                                 \
      ewald_pack_kernel<T_><<<mi_blocks(n_atoms, 256), 256, 0, st>>>((const T_*)positions, (const T_*)charges, n_atoms, (Vec4<T_>::type*)rec, sym);  \
      if (!CSR_ && trusted)                                                                                                                    \
        ewald_real_kernel<T_, false, true, true><<<blocks + vblocks, 256, 0, st>>>((const T_*)positions, (const T_*)charges, (const T_*)cell, (const T_*)alpha, \
                                                                batch_idx, n_atoms, idx_j, unit_shifts, neighbor_ptr, max_neighbors, mask_value,  \
                                                                flags, energies, (T_*)forces, charge_grads, sym, (const Vec4<T_>::type*)rec,   \
                                                                search_num_neighbors, vblocks, vshift, verify_phase);                           \
      else                                                                                                                                     \
      ewald_real_kernel<T_, CSR_, true><<<blocks, 256, 0, st>>>((const T_*)positions, (const T_*)charges, (const T_*)cell, (const T_*)alpha,   \
                                                                batch_idx, n_atoms, idx_j, unit_shifts, neighbor_ptr, max_neighbors, mask_value,  \
                                                                flags, energies, (T_*)forces, charge_grads, sym, (const Vec4<T_>::type*)rec);  \
    } else {                                                                                                                                   \
      ewald_real_kernel<T_, CSR_, false><<<blocks, 256, 0, st>>>((const T_*)positions, (const T_*)charges, (const T_*)cell, (const T_*)alpha,  \
                                                                 batch_idx, n_atoms, idx_j, unit_shifts, neighbor_ptr, max_neighbors, mask_value, \
                                                                 flags, energies, (T_*)forces, charge_grads, sym, nullptr);                    \
    }                                                                                                                                          \
  } while (0)
#define MI_EWS(T_, CSR_)                                                                                                                         \
  do {                                                                                                                                           \
    ewald_fixup_zero_kernel<T_><<<mi_blocks(n_atoms, 256), 256, 0, st>>>(sym, (T_*)forces, nullptr, charge_grads, n_atoms);                      \
    ewald_real_scatter_kernel<T_, CSR_><<<blocks, 256, 0, st>>>((const T_*)positions, (const T_*)charges, (const T_*)cell, (const T_*)alpha,      \
                                                                batch_idx, n_atoms, idx_j, unit_shifts, neighbor_ptr, max_neighbors, mask_value,  \
                                                                flags, (T_*)forces, charge_grads, sym);                                           \
  } while (0)
  // scratch = [symmetry checksums | {x,y,z,q} records]: either part is used only when the buffer is large enough for it
  const size_t sym_bytes = sizeof(unsigned long long) * EW_SYM_WORDS;
  const size_t rec_bytes = (dtype == MI_F32 ? 16 : 32) * (size_t)n_atoms;
  if (!scratch) scratch_bytes = 0;
  // the symmetry check only matters for outputs that are scattered in the reference (forces, charge gradients); energies are per owner
  unsigned long long* sym = ((flags & (MI_EW_FORCES | MI_EW_CHARGE_GRAD)) && scratch_bytes >= sym_bytes) ? (unsigned long long*)scratch : nullptr;
  void* rec = scratch_bytes >= sym_bytes + rec_bytes ? (void*)((char*)scratch + sym_bytes) : nullptr;
  // the trusted form needs the checksum words (its marks live there) and the record path (the pack kernel clears the words)
  const bool trusted = search_num_neighbors != nullptr && !csr && sym != nullptr && rec != nullptr && max_neighbors > 0;
  int vblocks = 0, vshift = 0;  // the stride rounded up to a power of two; 16 sampled rows per leading block
  if (trusted && verify_stride > 0) {
    while ((1ll << vshift) < (long long)verify_stride && vshift < 30) ++vshift;
    const long long off = (-(long long)verify_phase) & ((1ll << vshift) - 1);
    const long long samples = off < n_atoms ? (((long long)n_atoms - off + (1ll << vshift) - 1) >> vshift) : 0;
    vblocks = (int)((samples + 15) / 16);
  }
  if (sym && !rec) MI_HIP_CHECK(hipMemsetAsync(sym, 0, sizeof(unsigned long long) * EW_SYM_WORDS, st));  // with records: cleared by the pack kernel
  mi_timing_begin("ewald_real", stream);
  if (dtype == MI_F32) { if (csr) MI_EW(float, true); else MI_EW(float, false); }
  else { if (csr) MI_EW(double, true); else MI_EW(double, false); }
  mi_timing_end(stream);
  MI_LAUNCH_CHECK();
  if (sym) {
    if (dtype == MI_F32) { if (csr) MI_EWS(float, true); else MI_EWS(float, false); }
    else { if (csr) MI_EWS(double, true); else MI_EWS(double, false); }
    MI_LAUNCH_CHECK();
  }
#undef MI_EW
#undef MI_EWS
  return MI_OK;
}

extern "C" int mi_ewald_structure_factors(const void* positions, const void* weights, const void* k_vectors, const void* cell, const void* alpha,
                                          const int32_t* system_ptr, int n_atoms, int n_systems, int n_k, int max_atoms_per_system, int dtype,
                                          double* structure_factors, double* total_charge, void* stream) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype");
  MI_REQUIRE(n_systems >= 1 && (n_systems == 1 || system_ptr), "system_ptr is required for batches");
  hipStream_t st = (hipStream_t)stream;
  if (structure_factors && n_k > 0) MI_HIP_CHECK(hipMemsetAsync(structure_factors, 0, sizeof(double) * 2 * (size_t)n_systems * n_k, st));
  if (total_charge) MI_HIP_CHECK(hipMemsetAsync(total_charge, 0, sizeof(double) * n_systems, st));
  if (n_atoms <= 0 || n_k <= 0) return MI_OK;
  MI_REQUIRE(positions && weights && k_vectors && cell && alpha && structure_factors, "null pointer");
  if (max_atoms_per_system <= 0) max_atoms_per_system = n_atoms;
  dim3 grid(mi_blocks(n_k, EK_TILE), n_systems, mi_blocks(max_atoms_per_system, EK_ATOM_CHUNK));
  mi_timing_begin("ewald_structure_factors", stream);
  if (dtype == MI_F32)
    ewald_sf_kernel<float><<<grid, 256, 0, st>>>((const float*)positions, (const float*)weights, (const float*)k_vectors, (const float*)cell,
                                                 (const float*)alpha, system_ptr, n_atoms, n_k, structure_factors, total_charge);
  else
    ewald_sf_kernel<double><<<grid, 256, 0, st>>>((const double*)positions, (const double*)weights, (const double*)k_vectors, (const double*)cell,
                                                  (const double*)alpha, system_ptr, n_atoms, n_k, structure_factors, total_charge);
  mi_timing_end(stream);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_ewald_recip_gather(const void* positions, const void* charges, const void* k_vectors, const void* alpha, const int32_t* batch_idx,
                                     const double* structure_factors, const double* total_charge, int n_atoms, int n_k, int dtype,
                                     double* potential, double* kforce, double* energies, void* forces, double* charge_grads, void* stream) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype");
  if (n_atoms <= 0) return MI_OK;
  MI_REQUIRE(positions && charges && alpha && (n_k == 0 || (k_vectors && structure_factors)), "null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = mi_blocks(n_atoms, EK_TILE);
  mi_timing_begin("ewald_recip_gather", stream);
  if (dtype == MI_F32)
    ewald_recip_gather_kernel<float><<<blocks, 256, 0, st>>>((const float*)positions, (const float*)charges, (const float*)k_vectors,
                                                             (const float*)alpha, batch_idx, structure_factors, total_charge, n_atoms, n_k,
                                                             potential, kforce, energies, (float*)forces, charge_grads);
  else
    ewald_recip_gather_kernel<double><<<blocks, 256, 0, st>>>((const double*)positions, (const double*)charges, (const double*)k_vectors,
                                                              (const double*)alpha, batch_idx, structure_factors, total_charge, n_atoms, n_k,
                                                              potential, kforce, energies, (double*)forces, charge_grads);
  mi_timing_end(stream);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_coulomb(const double* positions, const double* charges, const double* cell, const int32_t* batch_idx, int n_atoms,
                          const int32_t* idx_j, const int32_t* unit_shifts, const int32_t* neighbor_ptr, int max_neighbors, int fill_value,
                          double cutoff, double alpha, double energy_prefactor, double* energies, double* forces, void* stream) {
  if (n_atoms <= 0) return MI_OK;
  MI_REQUIRE(positions && charges && cell && energies, "null pointer");
  MI_REQUIRE(neighbor_ptr || max_neighbors >= 0, "max_neighbors");
  hipStream_t st = (hipStream_t)stream;
  if (forces) MI_HIP_CHECK(hipMemsetAsync(forces, 0, sizeof(double) * 3 * (size_t)n_atoms, st));
  const bool csr = neighbor_ptr != nullptr;
  if (!csr && max_neighbors == 0) { MI_HIP_CHECK(hipMemsetAsync(energies, 0, sizeof(double) * (size_t)n_atoms, st)); return MI_OK; }
  MI_REQUIRE(idx_j && unit_shifts, "null neighbour arrays");
  const int blocks = mi_blocks(n_atoms, 4);
  mi_timing_begin("coulomb", stream);
  if (csr)
    coulomb_kernel<true><<<blocks, 256, 0, st>>>(positions, charges, cell, batch_idx, n_atoms, idx_j, unit_shifts, neighbor_ptr, 0, 0, cutoff, alpha,
                                                 energy_prefactor, energies, forces);
  else
    coulomb_kernel<false><<<blocks, 256, 0, st>>>(positions, charges, cell, batch_idx, n_atoms, idx_j, unit_shifts, nullptr, max_neighbors,
                                                  fill_value, cutoff, alpha, energy_prefactor, energies, forces);
  mi_timing_end(stream);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_coulomb_bwd(const double* positions, const double* charges, const double* cell, const int32_t* batch_idx, int n_atoms,
                              int n_systems, const int32_t* idx_j, const int32_t* unit_shifts, const int32_t* neighbor_ptr, int max_neighbors,
                              int fill_value, double cutoff, double alpha, double energy_prefactor, const double* grad_energies,
                              double* grad_positions, double* grad_charges, double* grad_cell, void* stream) {
  MI_REQUIRE(n_systems >= 1, "n_systems");
  MI_REQUIRE(grad_positions && grad_charges && grad_cell, "null gradient outputs");
  hipStream_t st = (hipStream_t)stream;
  MI_HIP_CHECK(hipMemsetAsync(grad_cell, 0, sizeof(double) * 9 * (size_t)n_systems, st));
  if (n_atoms <= 0) return MI_OK;
  MI_HIP_CHECK(hipMemsetAsync(grad_positions, 0, sizeof(double) * 3 * (size_t)n_atoms, st));
  MI_HIP_CHECK(hipMemsetAsync(grad_charges, 0, sizeof(double) * (size_t)n_atoms, st));
  const bool csr = neighbor_ptr != nullptr;
  if (!csr && max_neighbors <= 0) return MI_OK;
  MI_REQUIRE(positions && charges && cell && idx_j && unit_shifts && grad_energies, "null pointer");
  const int blocks = mi_blocks(n_atoms, 4);
  if (csr)
    coulomb_bwd_kernel<true><<<blocks, 256, 0, st>>>(positions, charges, cell, batch_idx, n_atoms, idx_j, unit_shifts, neighbor_ptr, 0, 0, cutoff,
                                                     alpha, energy_prefactor, grad_energies, grad_positions, grad_charges, grad_cell);
  else
    coulomb_bwd_kernel<false><<<blocks, 256, 0, st>>>(positions, charges, cell, batch_idx, n_atoms, idx_j, unit_shifts, nullptr, max_neighbors,
                                                      fill_value, cutoff, alpha, energy_prefactor, grad_energies, grad_positions, grad_charges,
                                                      grad_cell);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_coulomb_forces_bwd(const double* positions, const double* charges, const double* cell, const int32_t* batch_idx, int n_atoms,
                                     int n_systems, const int32_t* idx_j, const int32_t* unit_shifts, const int32_t* neighbor_ptr,
                                     int max_neighbors, int fill_value, double cutoff, double alpha, const double* grad_forces,
                                     double* grad_positions, double* grad_charges, double* grad_cell, void* stream) {
  MI_REQUIRE(n_systems >= 1, "n_systems");
  MI_REQUIRE(grad_positions && grad_charges && grad_cell, "null gradient outputs");
  hipStream_t st = (hipStream_t)stream;
  MI_HIP_CHECK(hipMemsetAsync(grad_cell, 0, sizeof(double) * 9 * (size_t)n_systems, st));
  if (n_atoms <= 0) return MI_OK;
  MI_HIP_CHECK(hipMemsetAsync(grad_positions, 0, sizeof(double) * 3 * (size_t)n_atoms, st));
  MI_HIP_CHECK(hipMemsetAsync(grad_charges, 0, sizeof(double) * (size_t)n_atoms, st));
  const bool csr = neighbor_ptr != nullptr;
  if (!csr && max_neighbors <= 0) return MI_OK;
  MI_REQUIRE(positions && charges && cell && idx_j && unit_shifts && grad_forces, "null pointer");
  const int blocks = mi_blocks(n_atoms, 4);
  if (csr)
    coulomb_force_bwd_kernel<true><<<blocks, 256, 0, st>>>(positions, charges, cell, batch_idx, n_atoms, idx_j, unit_shifts, neighbor_ptr, 0, 0, cutoff,
                                                           alpha, grad_forces, grad_positions, grad_charges, grad_cell);
  else
    coulomb_force_bwd_kernel<false><<<blocks, 256, 0, st>>>(positions, charges, cell, batch_idx, n_atoms, idx_j, unit_shifts, nullptr, max_neighbors,
                                                            fill_value, cutoff, alpha, grad_forces, grad_positions, grad_charges, grad_cell);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_ewald_recip_gather_kk(const void* positions, const void* k_vectors, const int32_t* batch_idx, const double* structure_factors,
                                        const double* weights, int n_atoms, int n_k, int dtype, double* out, void* stream) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype");
  if (n_atoms <= 0) return MI_OK;
  MI_REQUIRE(positions && weights && out && (n_k == 0 || (k_vectors && structure_factors)), "null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = mi_blocks(n_atoms, EK_TILE);
  if (dtype == MI_F32)
    ewald_recip_gather_kk_kernel<float><<<blocks, 256, 0, st>>>((const float*)positions, (const float*)k_vectors, batch_idx, structure_factors, weights,
                                                                n_atoms, n_k, out);
  else
    ewald_recip_gather_kk_kernel<double><<<blocks, 256, 0, st>>>((const double*)positions, (const double*)k_vectors, batch_idx, structure_factors,
                                                                 weights, n_atoms, n_k, out);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

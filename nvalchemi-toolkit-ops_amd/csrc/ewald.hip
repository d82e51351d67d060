// ewald.hip -- real-space (erfc-damped) part of Ewald / PME.  gfx950, wave64.
//
// Arithmetic of the reference kernels `_[batch_]ewald_real_space_energy[_forces][_charge_grad]_{,neighbor_matrix_}kernel`
// (interactions/electrostatics/ewald_kernels.py:266-1495) and their fp64 helpers (:150-258): separation vector and
// |r| in the input dtype, then fp64: E_i += 1/2 q_i q_j erfc_AS(a r)/r with the Abramowitz-Stegun 7.1.26 polynomial
// (math/math.py:52-93, NOT libm erfc: SURVEY F4), force magnitude 1/2 q_i q_j (erfc/r^3 + 2a/sqrt(pi) e^{-a^2 r^2}/r^2).
// Matrix entries equal to mask_value are padding (exact equality, ewald_kernels.py:304); pairs with r <= 1e-8 are skipped.
//
// Execution shape: one wave64 per atom, lanes stride the row / CSR range; fp64 wave reductions.  The reference adds
// -f to atom i and atomically +f to atom j for every directed pair; over the symmetric (full) list this kernel requires
// -- as the reference's own 1/2 prefactor already does -- that equals 2x the row owner's sum, so
// forces and charge gradients are written by the owner only: no atomics, deterministic.
#include "common.h"

namespace {

__device__ __forceinline__ double erfc_as_poly(double x, double e_neg_x2) {
  // x >= 0 here (alpha * distance); constants verbatim from math/math.py:73-78
  const double p = 0.3275911, a1 = 0.254829592, a2 = -0.284496736, a3 = 1.421413741, a4 = -1.453152027, a5 = 1.061405429;
  const double t = 1.0 / (1.0 + p * x);
  const double t2 = t * t, t3 = t2 * t, t4 = t3 * t, t5 = t4 * t;
  const double poly = a1 * t + a2 * t2 + a3 * t3 + a4 * t4 + a5 * t5;
  return poly * e_neg_x2;
}

template <class T, bool CSR>
__global__ __launch_bounds__(256) void ewald_real_kernel(const T* __restrict__ pos, const T* __restrict__ q, const T* __restrict__ cell,
                                                         const T* __restrict__ alpha, const int* __restrict__ batch_idx, int N,
                                                         const int* __restrict__ idx, const int* __restrict__ ush, const int* __restrict__ nptr,
                                                         int M, int mask_value, int flags, double* __restrict__ energies,
                                                         T* __restrict__ forces, double* __restrict__ cgrad) {
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
  double eacc = 0.0, cgi = 0.0;
  T fx = 0, fy = 0, fz = 0;
  for (long long e = beg + lane; e < end; e += MI_WAVE) {
    const int j = idx[e];
    if (!CSR && j == mask_value) continue;
    const double qj = (double)q[j];
    const T fs[3] = {(T)ush[3 * e], (T)ush[3 * e + 1], (T)ush[3 * e + 2]};
    T sh[3];
    rowvec_mat3(fs, cm, sh);  // == transpose(cell) * S with the same summation order
    const T sx = (pos[3 * (size_t)j] - pix) + sh[0], sy = (pos[3 * (size_t)j + 1] - piy) + sh[1], sz = (pos[3 * (size_t)j + 2] - piz) + sh[2];
    const double dist = (double)sqrt(sx * sx + sy * sy + sz * sz);
    if (!(dist > 1e-8)) continue;
    const double ar = al * dist;
    const double ex = exp(-(ar * ar));
    const double ec = erfc_as_poly(ar, ex);
    eacc += 0.5 * qi * qj * ec / dist;
    if (wf) {
      const double fm = (0.5 * qi * qj) * (ec / (dist * dist * dist) + two_over_sqrt_pi * al * ex / (dist * dist));
      const T fmt = (T)fm;
      fx -= fmt * sx; fy -= fmt * sy; fz -= fmt * sz;
    }
    if (wc) cgi += qj * (0.5 * ec / dist);
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
}

// adjoint for L = sum_i g_i E_i (see header): same walk, same skips, weights (g_i + g_j)
template <class T, bool CSR>
__global__ __launch_bounds__(256) void ewald_real_bwd_kernel(const T* __restrict__ pos, const T* __restrict__ q, const T* __restrict__ cell,
                                                             const T* __restrict__ alpha, const int* __restrict__ batch_idx, int N,
                                                             const int* __restrict__ idx, const int* __restrict__ ush,
                                                             const int* __restrict__ nptr, int M, int mask_value, const T* __restrict__ gE,
                                                             T* __restrict__ gpos, T* __restrict__ gq, double* __restrict__ gcell,
                                                             double* __restrict__ galpha) {
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
  for (long long e = beg + lane; e < end; e += MI_WAVE) {
    const int j = idx[e];
    if (!CSR && j == mask_value) continue;
    const double qj = (double)q[j], gj = (double)gE[j];
    const int S0 = ush[3 * e], S1 = ush[3 * e + 1], S2 = ush[3 * e + 2];
    const T fs[3] = {(T)S0, (T)S1, (T)S2};
    T sh[3];
    rowvec_mat3(fs, cm, sh);
    const T sx = (pos[3 * (size_t)j] - pix) + sh[0], sy = (pos[3 * (size_t)j + 1] - piy) + sh[1], sz = (pos[3 * (size_t)j + 2] - piz) + sh[2];
    const double dist = (double)sqrt(sx * sx + sy * sy + sz * sz);
    if (!(dist > 1e-8)) continue;
    const double ar = al * dist;
    const double ex = exp(-(ar * ar));
    const double ec = erfc_as_poly(ar, ex);
    const double fm = (0.5 * qi * qj) * (ec / (dist * dist * dist) + two_over_sqrt_pi * al * ex / (dist * dist));
    const double wsum = gi + gj;
    gx += wsum * fm * (double)sx; gy += wsum * fm * (double)sy; gz += wsum * fm * (double)sz;
    gqi += 0.5 * wsum * qj * ec / dist;
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
  if (galpha) { ga = wave_sum(ga); if (lane == 0) atomicAdd(&galpha[s], ga); }
  if (gcell) {
#pragma unroll
    for (int k = 0; k < 9; ++k) { const double v = wave_sum(gc[k]); if (lane == 0 && v != 0.0) atomicAdd(&gcell[9 * (size_t)s + k], v); }
  }
}

}  // namespace

extern "C" int mi_ewald_real_bwd(const void* positions, const void* charges, const void* cell, const void* alpha, const int32_t* batch_idx,
                                 int n_atoms, int dtype, const int32_t* idx_j, const int32_t* unit_shifts, const int32_t* neighbor_ptr,
                                 int max_neighbors, int mask_value, const void* grad_energies, void* grad_positions, void* grad_charges,
                                 double* grad_cell, double* grad_alpha, void* stream) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype");
  if (n_atoms <= 0) return MI_OK;
  MI_REQUIRE(positions && charges && cell && alpha && idx_j && unit_shifts && grad_energies && grad_positions && grad_charges, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = mi_blocks(n_atoms, 4);
  const bool csr = neighbor_ptr != nullptr;
#define MI_EWB(T_, CSR_)                                                                                                                       \
  ewald_real_bwd_kernel<T_, CSR_><<<blocks, 256, 0, st>>>((const T_*)positions, (const T_*)charges, (const T_*)cell, (const T_*)alpha, batch_idx, \
                                                          n_atoms, idx_j, unit_shifts, neighbor_ptr, max_neighbors, mask_value,                  \
                                                          (const T_*)grad_energies, (T_*)grad_positions, (T_*)grad_charges, grad_cell, grad_alpha)
  if (dtype == MI_F32) { if (csr) MI_EWB(float, true); else MI_EWB(float, false); }
  else { if (csr) MI_EWB(double, true); else MI_EWB(double, false); }
#undef MI_EWB
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_ewald_real(const void* positions, const void* charges, const void* cell, const void* alpha, const int32_t* batch_idx,
                             int n_atoms, int dtype, const int32_t* idx_j, const int32_t* unit_shifts, const int32_t* neighbor_ptr,
                             int max_neighbors, int mask_value, int flags, double* energies, void* forces, double* charge_grads, void* stream) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype");
  if (n_atoms <= 0) return MI_OK;
  MI_REQUIRE(positions && charges && cell && alpha && idx_j && unit_shifts && energies, "null pointer");
  MI_REQUIRE(!(flags & MI_EW_FORCES) || forces, "forces output");
  MI_REQUIRE(!(flags & MI_EW_CHARGE_GRAD) || charge_grads, "charge gradient output");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = mi_blocks(n_atoms, 4);
  const bool csr = neighbor_ptr != nullptr;
#define MI_EW(T_, CSR_)                                                                                                                     \
  ewald_real_kernel<T_, CSR_><<<blocks, 256, 0, st>>>((const T_*)positions, (const T_*)charges, (const T_*)cell, (const T_*)alpha, batch_idx, \
                                                      n_atoms, idx_j, unit_shifts, neighbor_ptr, max_neighbors, mask_value, flags, energies, \
                                                      (T_*)forces, charge_grads)
  mi_timing_begin("ewald_real", stream);
  if (dtype == MI_F32) { if (csr) MI_EW(float, true); else MI_EW(float, false); }
  else { if (csr) MI_EW(double, true); else MI_EW(double, false); }
  mi_timing_end(stream);
#undef MI_EW
  MI_LAUNCH_CHECK();
  return MI_OK;
}

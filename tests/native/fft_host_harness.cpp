// TEST HARNESS (never part of the product): runs the per-item bodies of csrc/fft_lds.h as ONE host "thread" per block (tid 0 of 1; the
// barriers between phases are no-ops then), so that the slot / twiddle / Hermitian index arithmetic of the fused PME mesh solve can be
// checked against numpy in the CPU suite.  Built by tests/test_fft_lds_cpu.py with g++.
#include <stdlib.h>
#include <vector>

#include "fft_lds.h"

using namespace mifft;

extern "C" {

int h_plan(int n, int max_lr, int* radix /*[8]*/, int* slot_of_freq /*[n]*/, int* freq_of_slot /*[n]*/) {
  if (!plan_ok(n)) return -1;
  const Plan p = make_plan(n, max_lr);
  for (int s = 0; s < MI_FFT_MAX_STAGES; ++s) radix[s] = p.radix[s];
  for (int k = 0; k < n; ++k) { slot_of_freq[k] = freq_slot(p, k); freq_of_slot[k] = slot_freq(p, k); }
  return p.nst;
}

// one 1-D line: forward (slots out) or inverse (slots in), unnormalised
int h_line(double* data /*[n][2]*/, int n, int max_lr, int inverse) {
  if (!plan_ok(n)) return -1;
  const Plan p = make_plan(n, max_lr);
  std::vector<Cx<double>> W(n);
  for (int t = 0; t < n; ++t) W[t] = unit_root<double>(t, n);
  // one stage after the other through the dispatcher that knows radix 16 (lines_fft is the radix <= 8 form the plane kernels use)
  Cx<double>* d = (Cx<double>*)data;
  for (int si = 0; si < p.nst; ++si) {
    const int s = inverse ? p.nst - 1 - si : si;
    for (int j = 0; j < n / p.radix[s]; ++j) {
      if (inverse) butterfly_r<+1, 4>(p, s, W.data(), 1, j, [=](int q) { return d[q]; }, [=](int q, Cx<double> v) { d[q] = v; });
      else butterfly_r<-1, 4>(p, s, W.data(), 1, j, [=](int q) { return d[q]; }, [=](int q, Cx<double> v) { d[q] = v; });
    }
  }
  return 0;
}

// kernel A over all (system, x) planes: mesh [B][nx][ny][nz] -> spectrum [B][nx][ny][P] in (y slot, z slot) order
int h_forward(const double* mesh, double* spec, int B, int nx, int ny, int nz) {
  if (!geom_ok(nx, ny, nz)) return -1;
  const Geom g = make_geom(B, nx, ny, nz);
  std::vector<char> tab(tables_bytes<double>(g));
  tables_body<double>(tab.data(), g, 0, 1);
  const Tables<double> tb = tables_at<double>(tab.data(), g);
  std::vector<char> lds(plane_lds_bytes<double>(g));
  for (int bx = 0; bx < B * nx; ++bx)
    fwd_plane_body<double>(mesh + (size_t)bx * ny * nz, (Cx<double>*)spec + (size_t)bx * ny * g.P, (Cx<double>*)lds.data(), g, tb, 0, 1);
  return 0;
}

// the transforms on their own (mi_fft_lds): R2C = kernel A + the PLAIN forward column kernel (natural-order half spectrum out);
// C2R = the PLAIN inverse column kernel (natural-order half spectrum in) + kernel C.  Unscaled both ways.
int h_r2c(const double* mesh, double* spec_nat /*[B][nx][ny][P][2]*/, int B, int nx, int ny, int nz) {
  if (!geom_ok(nx, ny, nz)) return -1;
  const Geom g = make_geom(B, nx, ny, nz);
  std::vector<char> tab(tables_bytes<double>(g));
  tables_body<double>(tab.data(), g, 0, 1);
  const Tables<double> tb = tables_at<double>(tab.data(), g);
  const size_t ncol = (size_t)ny * g.P;
  std::vector<Cx<double>> work((size_t)B * nx * ncol);
  size_t need = plane_lds_bytes_nat<double>(g);
  if (fwd_cols_lds_bytes<double>(g) > need) need = fwd_cols_lds_bytes<double>(g);
  std::vector<char> lds(need);
  for (int bx = 0; bx < B * nx; ++bx)
    fwd_plane_body<double, true, true>(mesh + (size_t)bx * ny * nz, work.data() + (size_t)bx * ncol, (Cx<double>*)lds.data(), g, tb, 0, 1);
  const int blocks = (int)((ncol + MI_SOLVE_COLS - 1) / MI_SOLVE_COLS);
  for (int b = 0; b < B; ++b)
    for (int blk = 0; blk < blocks; ++blk)
      fwd_cols_body<double, true, true>(work.data() + (size_t)b * nx * ncol, (Cx<double>*)lds.data(), g, tb, nullptr, 1.0, 1.0, 1, blk * MI_SOLVE_COLS, 0, 1,
                                        (Cx<double>*)spec_nat + (size_t)b * nx * ncol);
  return 0;
}
int h_c2r(const double* spec_nat, double* mesh, int B, int nx, int ny, int nz) {
  if (!geom_ok(nx, ny, nz)) return -1;
  const Geom g = make_geom(B, nx, ny, nz);
  std::vector<char> tab(tables_bytes<double>(g));
  tables_body<double>(tab.data(), g, 0, 1);
  const Tables<double> tb = tables_at<double>(tab.data(), g);
  const size_t ncol = (size_t)ny * g.P;
  std::vector<Cx<double>> work((size_t)B * nx * ncol);
  size_t need = plane_lds_bytes_nat<double>(g);
  if (inv_cols_lds_bytes<double>(g) > need) need = inv_cols_lds_bytes<double>(g);
  std::vector<char> lds(need);
  const int blocks = (int)((ncol + MI_SOLVE_COLS - 1) / MI_SOLVE_COLS);
  for (int b = 0; b < B; ++b)
    for (int blk = 0; blk < blocks; ++blk)
      inv_cols_body<double, true, true>((const Cx<double>*)spec_nat + (size_t)b * nx * ncol, work.data() + (size_t)b * nx * ncol, (Cx<double>*)lds.data(), g, tb,
                                        nullptr, 0, blk * MI_SOLVE_COLS, 0, 1);
  for (int p = 0; p < B * nx; ++p)
    inv_plane_body<double, true, true>(work.data() + (size_t)p * ncol, mesh + (size_t)p * ny * nz, (Cx<double>*)lds.data(), g, tb, 0, 1);
  return 0;
}

// kernels A, B1, B2, C: mesh [B][nx][ny][nz] -> real meshes [B][C][nx][ny][nz]  (unnormalised both ways, like hipFFT / the reference's norm='forward' inverse)
int h_solve(const double* mesh, double* out, int B, int nx, int ny, int nz, const double* recip /*[B][9]*/, const double* alpha, const double* volume,
            int sf_expo, int n_channels, double* spec_nat /*NULL or [B][nx][ny][P][2]: numpy.fft.rfftn(mesh)*/) {
  if (!geom_ok(nx, ny, nz)) return -1;
  const Geom g = make_geom(B, nx, ny, nz);
  std::vector<char> tab(tables_bytes<double>(g));
  tables_body<double>(tab.data(), g, 0, 1);
  const Tables<double> tb = tables_at<double>(tab.data(), g);
  const size_t ncol = (size_t)ny * g.P;
  std::vector<Cx<double>> spec((size_t)B * nx * ncol), conv((size_t)B * n_channels * nx * ncol);
  size_t need = plane_lds_bytes_nat<double>(g);
  if (inv_cols_lds_bytes<double>(g) > need) need = inv_cols_lds_bytes<double>(g);
  if (fwd_cols_lds_bytes<double>(g) > need) need = fwd_cols_lds_bytes<double>(g);
  std::vector<char> lds(need);
  for (int bx = 0; bx < B * nx; ++bx)
    fwd_plane_body<double>(mesh + (size_t)bx * ny * nz, spec.data() + (size_t)bx * ncol, (Cx<double>*)lds.data(), g, tb, 0, 1);
  const int blocks = (int)((ncol + MI_SOLVE_COLS - 1) / MI_SOLVE_COLS);
  for (int b = 0; b < B; ++b)
    for (int blk = 0; blk < blocks; ++blk)
      fwd_cols_body<double>(spec.data() + (size_t)b * nx * ncol, (Cx<double>*)lds.data(), g, tb, recip + 9 * b, alpha[b], volume[b], sf_expo,
                            blk * MI_SOLVE_COLS, 0, 1, spec_nat ? (Cx<double>*)spec_nat + (size_t)b * nx * ncol : nullptr);
  for (int b = 0; b < B; ++b)
    for (int ch = 0; ch < n_channels; ++ch)
      for (int blk = 0; blk < blocks; ++blk)
        inv_cols_body<double>(spec.data() + (size_t)b * nx * ncol, conv.data() + ((size_t)b * n_channels + ch) * nx * ncol, (Cx<double>*)lds.data(), g, tb,
                              recip + 9 * b, ch, blk * MI_SOLVE_COLS, 0, 1);
  for (int p = 0; p < B * n_channels * nx; ++p)
    inv_plane_body<double>(conv.data() + (size_t)p * ncol, out + (size_t)p * ny * nz, (Cx<double>*)lds.data(), g, tb, 0, 1);
  return 0;
}
}

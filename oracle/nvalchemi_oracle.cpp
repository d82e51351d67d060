// =====================================================================================
// oracle/nvalchemi_oracle.cpp  --  TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
//
// Single-threaded CPU restatement of the reference's hot path (NVIDIA/nvalchemi-toolkit-ops
// v0.2.0, Python + Warp kernels), written from the reference's kernel *semantics*:
// one "thread" of a Warp launch == one iteration of a serial loop here (which is also what
// Warp's own CPU backend does), atomics == sequential read-modify-write.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library.  The product (nvalchemi-toolkit-ops_amd/) never links, imports or calls it.
//
// Parity status: PINNED by the reference tests' own known answers
//   * neighbor counts of HoTlPd / SiCu at rc = 1, 4, 6 A   (test/neighborlist/test_cell_list.py:391-419,
//     test_batch_cell_list.py:516-541; structures test/neighborlist/test_utils.py:252-301)
//   * DFT-D3 Ne2 / HCl-dimer CN, energies, dE/dCN, forces  (test/interactions/dispersion/conftest.py:641-730)
//   * PME has no hard-coded numbers in the reference; it is pinned through an independent
//     explicit Ewald sum + Madelung constants in tests/ (see DESIGN.md "oracle").
// The reference itself (Warp) cannot be imported in the build container, so floating-point
// contraction/rounding of Warp's own builtins is unverifiable: compiled with -ffp-contract=off.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -shared -fPIC).
// =====================================================================================
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

// OpenMP build (libnvalchemi_oracle_omp.so, -fopenmp): the same loops with the outer atom loop shared between threads and every
// cross-row update (the reference's atomic_add) as an `omp atomic` -- used ONLY for the all-core leg of bench.py's cpu_baseline.
// The serial library (no -fopenmp: the pragmas vanish) stays the parity oracle; a CPU test checks both agree.
#ifdef _OPENMP
#include <omp.h>
#define ORC_PARALLEL_FOR _Pragma("omp parallel for schedule(dynamic, 32)")
#define ORC_ATOMIC _Pragma("omp atomic")
#define ORC_CAPTURE _Pragma("omp atomic capture")
#else
#define ORC_PARALLEL_FOR
#define ORC_ATOMIC
#define ORC_CAPTURE
#endif

namespace {

template <class T> struct Vec3 { T v[3]; };
template <class T> struct Mat3 { T m[3][3]; };

template <class T> inline T vlen(const T a[3]) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

template <class T> inline void load_mat(const T* p, Mat3<T>& M) {
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M.m[r][c] = p[3 * r + c];
}

// 3x3 inverse as adjugate * (1/det); only used for binning (result sets do not depend on it).
template <class T> inline Mat3<T> inverse3(const Mat3<T>& A) {
  const T (*a)[3] = A.m;
  Mat3<T> B;
  B.m[0][0] = a[1][1] * a[2][2] - a[1][2] * a[2][1];
  B.m[0][1] = a[0][2] * a[2][1] - a[0][1] * a[2][2];
  B.m[0][2] = a[0][1] * a[1][2] - a[0][2] * a[1][1];
  B.m[1][0] = a[1][2] * a[2][0] - a[1][0] * a[2][2];
  B.m[1][1] = a[0][0] * a[2][2] - a[0][2] * a[2][0];
  B.m[1][2] = a[0][2] * a[1][0] - a[0][0] * a[1][2];
  B.m[2][0] = a[1][0] * a[2][1] - a[1][1] * a[2][0];
  B.m[2][1] = a[0][1] * a[2][0] - a[0][0] * a[2][1];
  B.m[2][2] = a[0][0] * a[1][1] - a[0][1] * a[1][0];
  T det = a[0][0] * B.m[0][0] + a[0][1] * B.m[1][0] + a[0][2] * B.m[2][0];
  T s = T(1) / det;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) B.m[r][c] *= s;
  return B;
}

// row-vector * matrix:  r = M.row0*v0; r += M.row1*v1; r += M.row2*v2   (Warp mul(vec,mat))
template <class T> inline void rowvec_mat(const T v[3], const Mat3<T>& M, T out[3]) {
  for (int c = 0; c < 3; ++c) {
    T r = M.m[0][c] * v[0];
    r = r + M.m[1][c] * v[1];
    r = r + M.m[2][c] * v[2];
    out[c] = r;
  }
}
// matrix * column-vector: r = M.col0*v0; r += M.col1*v1; r += M.col2*v2  (Warp mul(mat,vec))
template <class T> inline void mat_colvec(const Mat3<T>& M, const T v[3], T out[3]) {
  for (int r = 0; r < 3; ++r) {
    T s = M.m[r][0] * v[0];
    s = s + M.m[r][1] * v[1];
    s = s + M.m[r][2] * v[2];
    out[r] = s;
  }
}

// floor divmod on int32  (math/math.py:41-50)
inline void floor_divmod(int a, int b, int& q, int& r) {
  q = a / b;  // C truncation == int(a / b)
  r = a % b;
  if (r < 0) { q -= 1; r = b + r; }
}

// ---------------------------------------------------------------------------------------
// Neighbor list: cell-list sizing  (cell_list.py:35-163, batch_cell_list.py:36-177)
// ---------------------------------------------------------------------------------------
template <class T>
void cells_per_dim_and_radius(const T* cell9, const uint8_t* pbc3, T cutoff, int cpd[3], int radius[3]) {
  Mat3<T> C; load_mat(cell9, C);
  Mat3<T> inv = inverse3(C);
  for (int i = 0; i < 3; ++i) {
    // row i of transpose(inverse) == column i of inverse
    T col[3] = {inv.m[0][i], inv.m[1][i], inv.m[2][i]};
    T face = T(1) / vlen(col);
    cpd[i] = std::max(int(face / cutoff), 1);
    if (radius) {
      if (cpd[i] == 1 && !pbc3[i]) radius[i] = 0;
      else radius[i] = int(std::ceil(cutoff * T(cpd[i]) / face));  // computed BEFORE halving
    }
  }
}
inline void halve_until(int cpd[3], long long cap, long long mult) {
  long long total = (long long)cpd[0] * cpd[1] * cpd[2];
  while (total * mult > cap) {
    for (int d = 0; d < 3; ++d) cpd[d] = std::max(cpd[d] / 2, 1);
    total = (long long)cpd[0] * cpd[1] * cpd[2];
  }
}

template <class T>
void atom_cell(const T* pos3, const Mat3<T>& inv, const uint8_t* pbc3, const int cpd[3], int cc[3], int wrap[3]) {
  T frac[3];
  rowvec_mat(pos3, inv, frac);
  for (int d = 0; d < 3; ++d) {
    int c = int(std::floor(frac[d] * T(cpd[d])));
    if (pbc3[d]) { int q, r; floor_divmod(c, cpd[d], q, r); wrap[d] = q; cc[d] = r; }
    else { wrap[d] = 0; cc[d] = std::min(std::max(c, 0), cpd[d] - 1); }
  }
}

// Full reference pipeline: estimate -> bin size -> count -> cumsum -> bin -> half-shell query.
// batch_idx == nullptr: single-system kernels (cell_list.py); else batch kernels (batch_cell_list.py).
// Returns the number of cells used (for reporting).
template <class T>
int cell_list_reference(const T* pos, int N, const T* cell, const uint8_t* pbc, const int* batch_idx, int B,
                        T cutoff, int max_nbins, int M, int fill_value, int half_fill,
                        int* nm, int* nsh, int* num,
                        int* out_cpd /*B*3 or null*/, int* out_radius /*B*3 or null*/) {
  const bool batch = batch_idx != nullptr;
  std::vector<int> cpd(3 * B), rad(3 * B), ncell(B);
  long long max_total = 0;
  for (int s = 0; s < B; ++s) {
    int c[3];
    cells_per_dim_and_radius(cell + 9 * s, pbc + 3 * s, cutoff, c, &rad[3 * s]);
    halve_until(c, max_nbins, 1);
    ncell[s] = c[0] * c[1] * c[2];
    max_total += ncell[s];
  }
  // construct_bin_size against the allocated capacity
  std::vector<int> cell_off(B + 1, 0);
  for (int s = 0; s < B; ++s) {
    int c[3];
    cells_per_dim_and_radius<T>(cell + 9 * s, pbc + 3 * s, cutoff, c, nullptr);
    halve_until(c, max_total, batch ? B : 1);
    for (int d = 0; d < 3; ++d) cpd[3 * s + d] = c[d];
    cell_off[s + 1] = cell_off[s] + c[0] * c[1] * c[2];
  }
  const int C = (int)max_total;
  std::vector<int> count(C, 0), start(C, 0), cell_atoms(N), amap(3 * (size_t)N), awrap(3 * (size_t)N);
  std::vector<Mat3<T>> inv(B), cm(B);
  for (int s = 0; s < B; ++s) { load_mat(cell + 9 * s, cm[s]); inv[s] = inverse3(cm[s]); }
  auto lin = [&](int s, const int cc[3]) {
    return cell_off[s] + cc[0] + cpd[3 * s] * (cc[1] + cpd[3 * s + 1] * cc[2]);
  };
  for (int i = 0; i < N; ++i) {
    int s = batch ? batch_idx[i] : 0;
    atom_cell(pos + 3 * (size_t)i, inv[s], pbc + 3 * s, &cpd[3 * s], &amap[3 * (size_t)i], &awrap[3 * (size_t)i]);
    count[lin(s, &amap[3 * (size_t)i])] += 1;
  }
  for (int c = 1; c < C; ++c) start[c] = start[c - 1] + count[c - 1];
  std::fill(count.begin(), count.end(), 0);
  for (int i = 0; i < N; ++i) {
    int s = batch ? batch_idx[i] : 0;
    int l = lin(s, &amap[3 * (size_t)i]);
    cell_atoms[start[l] + count[l]++] = i;
  }
  // outputs pre-filled as the wrappers do (cell_list.py:1358-1373)
  for (size_t k = 0; k < (size_t)N * M; ++k) nm[k] = fill_value;
  std::memset(nsh, 0, sizeof(int) * 3 * (size_t)N * M);
  std::memset(num, 0, sizeof(int) * (size_t)N);

  const T rc2 = cutoff * cutoff;
  auto visit = [&](int i, int s, int dx, int dy, int dz) {
    if (!(dx > 0 || (dx == 0 && dy > 0) || (dx == 0 && dy == 0 && dz >= 0))) return;
    const int* c3 = &cpd[3 * s];
    const uint8_t* p3 = pbc + 3 * s;
    int t[3] = {amap[3 * (size_t)i] + dx, amap[3 * (size_t)i + 1] + dy, amap[3 * (size_t)i + 2] + dz};
    for (int d = 0; d < 3; ++d) if (!p3[d] && (t[d] < 0 || t[d] >= c3[d])) return;
    int cs[3], wc[3];
    for (int d = 0; d < 3; ++d) floor_divmod(t[d], c3[d], cs[d], wc[d]);
    int l = lin(s, wc);
    const T* pi = pos + 3 * (size_t)i;
    for (int k = 0; k < count[l]; ++k) {
      int j = cell_atoms[start[l] + k];
      int S[3];
      for (int d = 0; d < 3; ++d) S[d] = p3[d] ? cs[d] + awrap[3 * (size_t)i + d] - awrap[3 * (size_t)j + d] : 0;
      if (dx == 0 && dy == 0 && dz == 0 && j <= i) continue;
      T fs[3] = {T(S[0]), T(S[1]), T(S[2])}, cart[3];
      rowvec_mat(fs, cm[s], cart);
      const T* pj = pos + 3 * (size_t)j;
      T dr[3];
      for (int d = 0; d < 3; ++d) dr[d] = (pj[d] - pi[d]) + cart[d];
      T d2 = dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2];
      if (d2 < rc2) {
        // _update_neighbor_matrix_pbc (neighbor_utils.py:106-147)
        int p;
        ORC_CAPTURE
        p = num[i]++;
        if (p < M) { nm[(size_t)i * M + p] = j; for (int d = 0; d < 3; ++d) nsh[((size_t)i * M + p) * 3 + d] = S[d]; }
        if (!half_fill) {
          int q;
          ORC_CAPTURE
          q = num[j]++;
          if (q < M) { nm[(size_t)j * M + q] = i; for (int d = 0; d < 3; ++d) nsh[((size_t)j * M + q) * 3 + d] = -S[d]; }
        }
      }
    }
  };
  ORC_PARALLEL_FOR
  for (int i = 0; i < N; ++i) {
    int s = batch ? batch_idx[i] : 0;
    const int* R = &rad[3 * s];
    if (!batch) {  // cell_list.py:464-466 loop nest
      for (int dx = 0; dx <= R[0]; ++dx) for (int dy = -R[1]; dy <= R[1]; ++dy) for (int dz = -R[2]; dz <= R[2]; ++dz) visit(i, s, dx, dy, dz);
    } else {       // batch_cell_list.py:468-470 loop nest
      for (int dz = -R[2]; dz <= R[2]; ++dz) for (int dy = -R[1]; dy <= R[1]; ++dy) for (int dx = 0; dx <= R[0]; ++dx) visit(i, s, dx, dy, dz);
    }
  }
  if (out_cpd) std::copy(cpd.begin(), cpd.end(), out_cpd);
  if (out_radius) std::copy(rad.begin(), rad.end(), out_radius);
  return C;
}

// Reference-format cell-list cache = what build_cell_list / batch_build_cell_list leave in the caller's tensors
// (cell_list.py:102-163 bin size against the ALLOCATED capacity, :166-276 count, :869-871 cumsum, :279-369 bin;
// batch_cell_list.py:103-373).  cell_atom_list is filled in ascending atom order here (thread ≙ loop iteration; the
// reference's order inside a cell is whatever its atomics produce).
template <class T>
void cell_cache_reference(const T* pos, int N, const T* cell, const uint8_t* pbc, const int* batch_idx, int B, T cutoff, int C,
                          int* cpd_out, int* atom_shift, int* atom_cc, int* counts, int* starts, int* cell_atoms) {
  const bool batch = batch_idx != nullptr;
  std::vector<int> cell_off(B + 1, 0);
  for (int s = 0; s < B; ++s) {
    int c[3];
    cells_per_dim_and_radius<T>(cell + 9 * s, pbc + 3 * s, cutoff, c, nullptr);
    halve_until(c, C, batch ? B : 1);
    for (int d = 0; d < 3; ++d) cpd_out[3 * s + d] = c[d];
    cell_off[s + 1] = cell_off[s] + c[0] * c[1] * c[2];
  }
  std::vector<Mat3<T>> inv(B);
  for (int s = 0; s < B; ++s) { Mat3<T> cm; load_mat(cell + 9 * s, cm); inv[s] = inverse3(cm); }
  std::fill(counts, counts + C, 0);
  auto lin = [&](int s, const int* cc) { return cell_off[s] + cc[0] + cpd_out[3 * s] * (cc[1] + cpd_out[3 * s + 1] * cc[2]); };
  for (int i = 0; i < N; ++i) {
    int s = batch ? batch_idx[i] : 0;
    atom_cell(pos + 3 * (size_t)i, inv[s], pbc + 3 * s, &cpd_out[3 * s], &atom_cc[3 * (size_t)i], &atom_shift[3 * (size_t)i]);
    counts[lin(s, &atom_cc[3 * (size_t)i])] += 1;
  }
  starts[0] = 0;
  for (int c = 1; c < C; ++c) starts[c] = starts[c - 1] + counts[c - 1];
  std::vector<int> fill(C, 0);
  for (int i = 0; i < N; ++i) {
    int s = batch ? batch_idx[i] : 0;
    int l = lin(s, &atom_cc[3 * (size_t)i]);
    cell_atoms[starts[l] + fill[l]++] = i;
  }
}

// rebuild_detection.py:37-170: any atom whose current cell differs from atom_to_cell_mapping / that moved farther than the skin
template <class T>
int cells_changed_reference(const T* pos, int N, const T* cell, const int* atom_to_cell, const int* cpd, const uint8_t* pbc) {
  Mat3<T> C; load_mat(cell, C);
  Mat3<T> inv = inverse3(C);
  for (int i = 0; i < N; ++i) {
    T frac[3];
    rowvec_mat(pos + 3 * (size_t)i, inv, frac);  // == transpose(inverse(cell)) * r
    for (int d = 0; d < 3; ++d) {
      int c = int(std::floor(frac[d] * T(cpd[d])));
      if (pbc[d]) { c = c % cpd[d]; if (c < 0) c += cpd[d]; }
      else c = std::min(std::max(c, 0), cpd[d] - 1);
      if (c != atom_to_cell[3 * (size_t)i + d]) return 1;
    }
  }
  return 0;
}
template <class T>
int moved_beyond_skin_reference(const T* ref, const T* cur, int N, T threshold) {
  for (int i = 0; i < N; ++i) {
    T d[3] = {cur[3 * (size_t)i] - ref[3 * (size_t)i], cur[3 * (size_t)i + 1] - ref[3 * (size_t)i + 1], cur[3 * (size_t)i + 2] - ref[3 * (size_t)i + 2]};
    if (vlen(d) > threshold) return 1;
  }
  return 0;
}

// Naive O(N^2)  (naive.py:37-182, neighbor_utils.py:26-67,150-211)
template <class T>
void naive_reference(const T* pos, int N, const T* cell /*null => no pbc*/, const uint8_t* pbc, T cutoff_sq,
                     double cutoff, int M, int fill_value, int half_fill, int* nm, int* nsh, int* num) {
  for (size_t k = 0; k < (size_t)N * M; ++k) nm[k] = fill_value;
  std::memset(num, 0, sizeof(int) * (size_t)N);
  auto upd = [&](int i, int j) {
    int p = num[i]++;
    if (p < M) nm[(size_t)i * M + p] = j;
    if (!half_fill && i < j) { int q = num[j]++; if (q < M) nm[(size_t)j * M + q] = i; }
  };
  if (!cell) {
    for (int i = 0; i < N; ++i)
      for (int j = i + 1; j < N; ++j) {
        T d[3]; for (int k = 0; k < 3; ++k) d[k] = pos[3 * (size_t)i + k] - pos[3 * (size_t)j + k];
        if (d[0] * d[0] + d[1] * d[1] + d[2] * d[2] < cutoff_sq) upd(i, j);
      }
    return;
  }
  std::memset(nsh, 0, sizeof(int) * 3 * (size_t)N * M);
  Mat3<T> C; load_mat(cell, C);
  Mat3<T> inv = inverse3(C);
  int s[3];
  for (int d = 0; d < 3; ++d) {
    T col[3] = {inv.m[0][d], inv.m[1][d], inv.m[2][d]};
    T dinv = pbc[d] ? vlen(col) : T(0);
    s[d] = int(std::ceil(dinv * T(cutoff)));
  }
  auto updp = [&](int i, int j, const int S[3]) {
    int p = num[i]++;
    if (p < M) { nm[(size_t)i * M + p] = j; for (int d = 0; d < 3; ++d) nsh[((size_t)i * M + p) * 3 + d] = S[d]; }
    if (!half_fill) { int q = num[j]++; if (q < M) { nm[(size_t)j * M + q] = i; for (int d = 0; d < 3; ++d) nsh[((size_t)j * M + q) * 3 + d] = -S[d]; } }
  };
  for (int k0 = 0; k0 <= s[0]; ++k0) for (int k1 = -s[1]; k1 <= s[1]; ++k1) for (int k2 = -s[2]; k2 <= s[2]; ++k2) {
    if (!(k0 > 0 || (k0 == 0 && k1 > 0) || (k0 == 0 && k1 == 0 && k2 >= 0))) continue;
    int S[3] = {k0, k1, k2};
    bool zero = (k0 == 0 && k1 == 0 && k2 == 0);
    T fs[3] = {T(k0), T(k1), T(k2)}, sc[3];
    rowvec_mat(fs, C, sc);
    for (int ia = 0; ia < N; ++ia) {
      T ps[3]; for (int d = 0; d < 3; ++d) ps[d] = sc[d] + pos[3 * (size_t)ia + d];
      int jend = zero ? ia : N;
      for (int ja = 0; ja < jend; ++ja) {
        T d[3]; for (int k = 0; k < 3; ++k) d[k] = ps[k] - pos[3 * (size_t)ja + k];
        if (d[0] * d[0] + d[1] * d[1] + d[2] * d[2] < cutoff_sq) updp(ja, ia, S);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// DFT-D3(BJ)   (dftd3.py:341-1615; pass sequencing :1911-2122 / :2306-2465)
// ---------------------------------------------------------------------------------------
// WIDE-SUM MODE (off by default; orc_set_d3_wide_sums): the same fp32 pair arithmetic, but every accumulation the reference does in
// fp32 (CN, dE/dCN per atom; energy and virial per system, which it adds with fp32 atomics in arbitrary atom order) is carried in
// double and rounded once.  Not the reference's result -- the reference's result MINUS its own summation-order noise; the D3 error
// budget measures the HIP kernels against this and reports the reference-order noise next to it.
static int g_d3_wide = 0;

struct D3Par {
  const float* rcov; const float* r4r2; const float* c6ab; const float* cnref; int nz;  // nz = maxZ+1
  float k1, k3, a1, a2, s6, s8, s5_on, s5_off, inv_w;
};

inline void s5_switch(float r, float on, float off, float inv_w, float& sw, float& dsw) {
  if (off <= on || r <= on) { sw = 1.0f; dsw = 0.0f; return; }
  if (r >= off) { sw = 0.0f; dsw = 0.0f; return; }
  float t = (r - on) * inv_w, t2 = t * t, t3 = t2 * t, t4 = t3 * t, t5 = t4 * t;
  sw = 1.0f - (10.0f * t3 - 15.0f * t4 + 6.0f * t5);
  dsw = (-30.0f * t2 + 60.0f * t3 - 30.0f * t4) * inv_w;
}

inline void c6_interp(float cn_i, float cn_j, const float* c6m, const float* cri, const float* crj, float k3,
                      float& c6, float& dci, float& dcj) {
  float mx = -1e20f;
  for (int p = 0; p < 5; ++p) for (int q = 0; q < 5; ++q) {
    float c = c6m[5 * p + q];
    if (c == 0.0f) continue;
    float di = cn_i - cri[5 * p + q], dj = cn_j - crj[5 * q + p];
    float a = k3 * (di * di + dj * dj);
    if (a > mx) mx = a;
  }
  float w = 0, z = 0, wdi = 0, wdj = 0, zdi = 0, zdj = 0;
  for (int p = 0; p < 5; ++p) for (int q = 0; q < 5; ++q) {
    float c = c6m[5 * p + q];
    if (c == 0.0f) continue;
    float di = cn_i - cri[5 * p + q], dj = cn_j - crj[5 * q + p];
    float a = k3 * (di * di + dj * dj) - mx;
    if (a < -12.0f) continue;
    float L = std::exp(a);
    w += L; z += c * L; wdi += L * di; wdj += L * dj; zdi += c * L * di; zdj += c * L * dj;
  }
  if (w > 1e-12f) {
    float wi = 1.0f / w;
    c6 = z * wi;
    float si = zdi - c6 * wdi, sj = zdj - c6 * wdj;
    float f = (2.0f * k3) * wi;
    dci = f * si; dcj = f * sj;
  } else { c6 = dci = dcj = 0.0f; }
}

// geometry: returns false when the pair is skipped (r < 1e-12)
template <class T>
inline bool d3_geom(const T* pi, const T* pj, const T* cart, bool periodic, float& r, float& rinv, float rij[3]) {
  for (int d = 0; d < 3; ++d) {
    T n = periodic ? (pj[d] - pi[d]) + cart[d] : pj[d] - pi[d];
    rij[d] = float(n);
  }
  r = std::sqrt(rij[0] * rij[0] + rij[1] * rij[1] + rij[2] * rij[2]);
  if (r < 1e-12f) return false;
  rinv = 1.0f / r;
  return true;
}

inline float cn_count(float rinv, float rci, float rcj, float k1, float* dcn) {
  float rr = (rci + rcj) * rinv;
  float f = 1.0f / (1.0f + std::exp(-k1 * (rr - 1.0f)));
  if (dcn) *dcn = -f * (1.0f - f) * k1 * rr * rinv;
  return f;
}

// One implementation for both layouts: row i owns entries [beg,end) of (jidx, shifts);
// matrix layout => beg=i*M, end=(i+1)*M and entries with j >= fill_value are padding.
template <class T>
void dftd3_reference(const T* pos, const int* numbers, int N, const int* jidx, const int* ushift /*null => non periodic*/,
                     const int* ptr /*null => matrix*/, int M, int fill_value, const T* cell, const int* batch_idx,
                     const D3Par& P, int compute_virial, float* energy, float* forces, float* cn, float* virial,
                     int B) {
  const bool periodic = (cell != nullptr && ushift != nullptr);
  std::memset(energy, 0, sizeof(float) * B);
  std::memset(forces, 0, sizeof(float) * 3 * (size_t)N);
  std::memset(cn, 0, sizeof(float) * (size_t)N);
  if (compute_virial) std::memset(virial, 0, sizeof(float) * 9 * B);
  std::vector<float> dEdCN(N, 0.0f);
  std::vector<double> wide(10 * (size_t)B, 0.0);
  auto range = [&](int i, size_t& b, size_t& e) {
    if (ptr) { b = ptr[i]; e = ptr[i + 1]; } else { b = (size_t)i * M; e = b + M; }
  };
  auto valid = [&](size_t k, int& j) { j = jidx[k]; if (!ptr && j >= fill_value) return false; return numbers[j] != 0; };
  auto cartshift = [&](int i, size_t k, T out[3]) {
    if (!periodic) { out[0] = out[1] = out[2] = T(0); return; }
    Mat3<T> C; load_mat(cell + 9 * (batch_idx ? batch_idx[i] : 0), C);
    T fs[3] = {T(ushift[3 * k]), T(ushift[3 * k + 1]), T(ushift[3 * k + 2])};
    rowvec_mat(fs, C, out);
  };
  // pass 1: coordination numbers
  ORC_PARALLEL_FOR
  for (int i = 0; i < N; ++i) {
    if (numbers[i] == 0) continue;
    double acc = 0.0;
    float rci = P.rcov[numbers[i]];
    size_t b, e; range(i, b, e);
    for (size_t k = b; k < e; ++k) {
      int j; if (!valid(k, j)) continue;
      T cs[3]; cartshift(i, k, cs);
      float r, rinv, rij[3];
      if (!d3_geom(pos + 3 * (size_t)i, pos + 3 * (size_t)j, cs, periodic, r, rinv, rij)) continue;
      const float f = cn_count(rinv, rci, P.rcov[numbers[j]], P.k1, nullptr);
      acc = g_d3_wide ? acc + double(f) : double(float(acc) + f);  // reference: sequential fp32 (dftd3.py:911)
    }
    cn[i] = float(acc);
  }
  // pass 2: energy, direct force, dE/dCN
  ORC_PARALLEL_FOR
  for (int i = 0; i < N; ++i) {
    if (numbers[i] == 0) continue;
    int zi = numbers[i];
    double F[3] = {0, 0, 0}, E = 0, V[9] = {0};
    double dacc = 0.0;
    size_t b, e; range(i, b, e);
    for (size_t k = b; k < e; ++k) {
      int j; if (!valid(k, j)) continue;
      T cs[3]; cartshift(i, k, cs);
      float r, rinv, rij[3];
      if (!d3_geom(pos + 3 * (size_t)i, pos + 3 * (size_t)j, cs, periodic, r, rinv, rij)) continue;
      float rhat[3] = {rij[0] * rinv, rij[1] * rinv, rij[2] * rinv};
      int zj = numbers[j];
      float c6, dci, dcj;
      c6_interp(cn[i], cn[j], P.c6ab + ((size_t)zi * P.nz + zj) * 25, P.cnref + ((size_t)zi * P.nz + zj) * 25,
                P.cnref + ((size_t)zj * P.nz + zi) * 25, P.k3, c6, dci, dcj);
      if (c6 < 1e-12f) continue;
      // _bj_damping (dftd3.py:648-687)
      float q = 3.0f * P.r4r2[zi] * P.r4r2[zj];
      float r0 = P.a1 * std::sqrt(q) + P.a2;
      float r2 = r * r, r4 = r2 * r2, r6 = r4 * r2, r8 = r4 * r4;
      float r02 = r0 * r0, r04 = r02 * r02, r06 = r04 * r02, r08 = r04 * r04;
      float i6 = 1.0f / (r6 + r06), i8 = 1.0f / (r8 + r08);
      float damp = P.s6 * i6 + P.s8 * q * i8;
      // _dispersion_energy_force (dftd3.py:690-731)
      float eij = -c6 * damp;
      float r5 = r4 * r, r7 = r6 * r;
      float d6 = -6.0f * P.s6 * r5 * i6 * i6;
      float d8 = -8.0f * P.s8 * q * r7 * i8 * i8;
      float dEdr = -c6 * (d6 + d8);
      float sw, dsw; s5_switch(r, P.s5_on, P.s5_off, P.inv_w, sw, dsw);
      float esw = eij * sw;
      float dEsw = sw * dEdr + eij * dsw;
      float Fd[3] = {dEsw * rhat[0], dEsw * rhat[1], dEsw * rhat[2]};
      for (int d = 0; d < 3; ++d) F[d] += double(Fd[d]);
      E += double(esw);
      const float dterm = -damp * dci;
      dacc = g_d3_wide ? dacc + double(dterm) : double(float(dacc) + dterm);
      if (compute_virial) for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) V[3 * a + c] += double(Fd[a] * rij[c]);
    }
    for (int d = 0; d < 3; ++d) forces[3 * (size_t)i + d] = float(F[d]);
    dEdCN[i] = float(dacc);
    int s = batch_idx ? batch_idx[i] : 0;
    if (g_d3_wide) {
      ORC_ATOMIC
      wide[10 * (size_t)s] += 0.5 * E;
      if (compute_virial) for (int a = 0; a < 9; ++a) {
        ORC_ATOMIC
        wide[10 * (size_t)s + 1 + a] += -0.5 * V[a];
      }
    } else {
      ORC_ATOMIC
      energy[s] += 0.5f * float(E);
      if (compute_virial) for (int a = 0; a < 9; ++a) {
        ORC_ATOMIC
        virial[9 * s + a] += -0.5f * float(V[a]);
      }
    }
  }
  // pass 3: chain-rule force through CN
  ORC_PARALLEL_FOR
  for (int i = 0; i < N; ++i) {
    if (numbers[i] == 0) continue;
    double F[3] = {0, 0, 0}, V[9] = {0};
    float rci = P.rcov[numbers[i]];
    size_t b, e; range(i, b, e);
    for (size_t k = b; k < e; ++k) {
      int j; if (!valid(k, j)) continue;
      T cs[3]; cartshift(i, k, cs);
      float r, rinv, rij[3];
      if (!d3_geom(pos + 3 * (size_t)i, pos + 3 * (size_t)j, cs, periodic, r, rinv, rij)) continue;
      float rhat[3] = {rij[0] * rinv, rij[1] * rinv, rij[2] * rinv};
      float dcn; cn_count(rinv, rci, P.rcov[numbers[j]], P.k1, &dcn);
      float dEdr = (dEdCN[i] + dEdCN[j]) * dcn;
      float Fc[3] = {dEdr * rhat[0], dEdr * rhat[1], dEdr * rhat[2]};
      for (int d = 0; d < 3; ++d) F[d] += double(Fc[d]);
      if (compute_virial) for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) V[3 * a + c] += double(Fc[a] * rij[c]);
    }
    for (int d = 0; d < 3; ++d) forces[3 * (size_t)i + d] = forces[3 * (size_t)i + d] + float(F[d]);
    if (compute_virial) {
      int s = batch_idx ? batch_idx[i] : 0;
      for (int a = 0; a < 9; ++a) {
        if (g_d3_wide) {
          ORC_ATOMIC
          wide[10 * (size_t)s + 1 + a] += -0.5 * V[a];
        } else {
          ORC_ATOMIC
          virial[9 * s + a] += -0.5f * float(V[a]);
        }
      }
    }
  }
  if (g_d3_wide)
    for (int s = 0; s < B; ++s) {
      energy[s] = float(wide[10 * (size_t)s]);
      if (compute_virial) for (int a = 0; a < 9; ++a) virial[9 * s + a] = float(wide[10 * (size_t)s + 1 + a]);
    }
}

// ---------------------------------------------------------------------------------------
// erfc (Abramowitz-Stegun 7.1.26; math/math.py:52-93) and Ewald real space
// (ewald_kernels.py:150-258 helpers; :266-1495 kernels)
// ---------------------------------------------------------------------------------------
template <class T> inline T erfc_as(T x) {
  T ax = std::fabs(x);
  const T p = T(0.3275911), a1 = T(0.254829592), a2 = T(-0.284496736), a3 = T(1.421413741), a4 = T(-1.453152027),
          a5 = T(1.061405429);
  T t = T(1) / (T(1) + p * ax);
  T t2 = t * t, t3 = t2 * t, t4 = t3 * t, t5 = t4 * t;
  T poly = a1 * t + a2 * t2 + a3 * t3 + a4 * t4 + a5 * t5;
  T e = poly * std::exp(-ax * ax);
  return x >= T(0) ? e : T(2) - e;
}

template <class T>
void ewald_real_reference(const T* pos, const T* q, const T* cell, const T* alpha, const int* batch_idx, int N,
                          const int* jidx, const int* ushift, const int* ptr, int M, int mask_value,
                          int want_forces, int want_cg, double* energies, T* forces, double* cgrad) {
  std::memset(energies, 0, sizeof(double) * (size_t)N);
  if (want_forces) std::memset(forces, 0, sizeof(T) * 3 * (size_t)N);
  if (want_cg) std::memset(cgrad, 0, sizeof(double) * (size_t)N);
  const double two_over_sqrt_pi = 2.0 / 1.7724538509055159;
  ORC_PARALLEL_FOR
  for (int i = 0; i < N; ++i) {
    int s = batch_idx ? batch_idx[i] : 0;
    double qi = double(q[i]), al = double(alpha[s]);
    Mat3<T> C; load_mat(cell + 9 * s, C);
    double eacc = 0.0, cgi = 0.0;
    T fi[3] = {T(0), T(0), T(0)};
    size_t b, e;
    if (ptr) { b = ptr[i]; e = ptr[i + 1]; } else { b = (size_t)i * M; e = b + M; }
    for (size_t k = b; k < e; ++k) {
      int j = jidx[k];
      if (!ptr && j == mask_value) continue;
      double qj = double(q[j]);
      T fs[3] = {T(ushift[3 * k]), T(ushift[3 * k + 1]), T(ushift[3 * k + 2])}, sh[3];
      rowvec_mat(fs, C, sh);  // == transpose(cell) * S, same summation order
      T sep[3];
      for (int d = 0; d < 3; ++d) sep[d] = (pos[3 * (size_t)j + d] - pos[3 * (size_t)i + d]) + sh[d];
      double dist = double(vlen(sep));
      if (!(dist > 1e-8)) continue;
      double ar = al * dist;
      double ec = erfc_as<double>(ar);
      eacc += 0.5 * qi * qj * ec / dist;
      if (want_forces) {
        double ex = std::exp(-(ar * ar));
        double fm = (0.5 * qi * qj) * (ec / (dist * dist * dist) + two_over_sqrt_pi * al * ex / (dist * dist));
        T f[3] = {T(fm) * sep[0], T(fm) * sep[1], T(fm) * sep[2]};
        for (int d = 0; d < 3; ++d) {
          fi[d] -= f[d];
          ORC_ATOMIC
          forces[3 * (size_t)j + d] += f[d];
        }
      }
      if (want_cg) {
        double pot = 0.5 * ec / dist;
        cgi += qj * pot;
        ORC_ATOMIC
        cgrad[j] += qi * pot;
      }
    }
    energies[i] += eacc;
    if (want_forces) for (int d = 0; d < 3; ++d) {
      ORC_ATOMIC
      forces[3 * (size_t)i + d] += fi[d];
    }
    if (want_cg) {
      ORC_ATOMIC
      cgrad[i] += cgi;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Cardinal B-splines, spread / gather  (spline.py:127-488 functions; :497-676, :763-959 kernels)
// ---------------------------------------------------------------------------------------
// EXTENDED MODE ("beyond reference", off by default; orc_set_extended): the reference has no order-5/6 branch and caps the
// structure-factor exponent at 4, so its order-5/6 PME is identically zero.  The product evaluates true cardinal
// B-splines there; to have something to compare them with, extended mode evaluates M_n(u) for n >= 5 by the closed-form
// truncated-power sum  M_n(u) = 1/(n-1)! sum_k (-1)^k C(n,k) (u-k)_+^(n-1)  in long double (a formulation that shares nothing
// with the product's bottom-up Cox-de Boor recursion) and uses exponent = order in the structure factor.  Orders <= 4 are untouched.
static int g_extended = 0;
template <class T> inline T bspline_truncated_power(T u, int n) {
  if (!(u >= T(0) && u < T(n))) return T(0);
  long double acc = 0.0L, binom = 1.0L, fact = 1.0L;
  for (int k = 2; k < n; ++k) fact *= (long double)k;
  for (int k = 0; k <= n; ++k) {
    long double x = (long double)u - (long double)k;
    if (x > 0.0L) { long double pw = 1.0L; for (int e = 0; e < n - 1; ++e) pw *= x; acc += ((k & 1) ? -binom : binom) * pw; }
    binom = binom * (long double)(n - k) / (long double)(k + 1);
  }
  return T(acc / fact);
}

template <class T> inline T bspline_w(T u, int order) {
  const T zero = 0, one = 1, two = 2, three = 3, four = 4, six = 6;
  if (order >= 5 && g_extended) return bspline_truncated_power(u, order);
  if (order == 4) {
    if (u >= zero && u < one) return u * u * u / six;
    if (u >= one && u < two) { T u2 = u * u, u3 = u2 * u; return (T(-3) * u3 + T(12) * u2 - T(12) * u + four) / six; }
    if (u >= two && u < three) { T u2 = u * u, u3 = u2 * u; return (three * u3 - T(24) * u2 + T(60) * u - T(44)) / six; }
    if (u >= three && u < four) { T v = four - u; return v * v * v / six; }
    return zero;
  } else if (order == 3) {
    if (u >= zero && u < one) return u * u / two;
    if (u >= one && u < two) return T(0.75) - (u - T(1.5)) * (u - T(1.5));
    if (u >= two && u < three) { T v = three - u; return v * v / two; }
    return zero;
  } else if (order == 2) {
    if (u >= zero && u < one) return u;
    if (u >= one && u < two) return two - u;
    return zero;
  } else if (order == 1) {
    return (u >= zero && u < one) ? one : zero;
  }
  return zero;  // order 5/6: the reference has no branch
}

template <class T> struct Stencil { int base[3]; T theta[3]; int off0[3]; };

template <class T>
inline Stencil<T> stencil_for(const T* p, const Mat3<T>& cit, const int dims[3], int order) {
  Stencil<T> s;
  T frac[3];
  mat_colvec(cit, p, frac);
  for (int d = 0; d < 3; ++d) {
    T mc = frac[d] * T(dims[d]);
    s.base[d] = int(std::floor(mc));
    s.theta[d] = mc - T(s.base[d]);
    s.off0[d] = int(std::floor(s.theta[d] - T(order - 2) * T(0.5)));
  }
  return s;
}
template <class T> inline T weight3(const Stencil<T>& s, const int off[3], int order) {
  T half = T(order) * T(0.5), u[3];
  for (int d = 0; d < 3; ++d) { u[d] = half + s.theta[d] - T(off[d]); if (u[d] < T(0) || u[d] >= T(order)) return T(0); }
  return bspline_w(u[0], order) * bspline_w(u[1], order) * bspline_w(u[2], order);
}
inline int wrapi(int i, int n) { return ((i % n) + n) % n; }

// mode: 0 spread (values -> mesh), 1 gather scalar (mesh -> out[N]), 2 gather vec3 (mesh[...,3] -> out[N,3], times charge)
template <class T>
void spline_reference(int mode, const T* pos, const T* values, const int* batch_idx, const T* cell_inv_t, int N, int B,
                      const int dims[3], int order, T* mesh, T* out) {
  const bool batch = batch_idx != nullptr;
  const size_t msz = (size_t)dims[0] * dims[1] * dims[2];
  const int P = order * order * order;
  ORC_PARALLEL_FOR
  for (int i = 0; i < N; ++i) {
    int s = batch ? batch_idx[i] : 0;
    Mat3<T> cit; load_mat(cell_inv_t + 9 * s, cit);
    Stencil<T> st = stencil_for(pos + 3 * (size_t)i, cit, dims, order);
    for (int pt = 0; pt < P; ++pt) {
      int off[3] = {pt / (order * order) + st.off0[0], (pt % (order * order)) / order + st.off0[1], pt % order + st.off0[2]};
      T w = weight3(st, off, order);
      // thresholds: single-system spread w>0, everything else w>1e-8 (spline.py:548,608,670,820,885,953)
      bool take = (mode == 0 && !batch) ? (w > T(0)) : (w > T(1e-8));
      if (!take) continue;
      size_t g = ((size_t)wrapi(st.base[0] + off[0], dims[0]) * dims[1] + wrapi(st.base[1] + off[1], dims[1])) * dims[2] +
                 wrapi(st.base[2] + off[2], dims[2]);
      if (mode == 0) {
        ORC_ATOMIC
        mesh[s * msz + g] += values[i] * w;
      }
      else if (mode == 1) out[i] += mesh[s * msz + g] * w;
      else for (int d = 0; d < 3; ++d) out[3 * (size_t)i + d] += (values[i] * mesh[(s * msz + g) * 3 + d]) * w;
    }
  }
}

// ---------------------------------------------------------------------------------------
// PME Green function / structure factor / corrections  (pme_kernels.py:93-657)
// ---------------------------------------------------------------------------------------
template <class T> inline T sinc_pi(T x) {
  if (std::fabs(x) < T(1e-6)) return T(1);
  T px = T(3.14159265358979323846) * x;
  return std::sin(px) / px;
}
template <class T>
void green_sf_reference(const T* k2, const T* alpha, const T* volume, int B, int nx, int ny, int nz, int order, T* G, T* sf2) {
  const int nzr = nz / 2 + 1;
  auto miller = [](int i, int n) { return i < (n + 1) / 2 ? i : i - n; };  // fftfreq(n, 1/n)
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < nx; ++i) for (int j = 0; j < ny; ++j) for (int k = 0; k < nzr; ++k) {
      size_t g = (((size_t)b * nx + i) * ny + j) * nzr + k;
      T ks = k2[g], a = alpha[b];
      if (ks < T(1e-10)) G[g] = T(0);
      else G[g] = T(6.283185307179586) * (std::exp(-(T(1) / (T(4) * a * a)) * ks) / ks) / volume[b];
      if (i == 0 && j == 0 && k == 0) G[g] = T(0);
      if (b == 0) {
        T sp = sinc_pi(T(miller(i, nx)) / T(nx)) * sinc_pi(T(miller(j, ny)) / T(ny)) * sinc_pi(T(k) / T(nz));
        T sf = sp;
        const int cap = g_extended ? order : 4;  // exponent capped at 4 unless extended mode is on
        for (int t = 1; t < cap; ++t) if (t < order) sf = sf * sp;
        if (sf < T(1e-10)) sf = T(1e-10);
        sf2[((size_t)i * ny + j) * nzr + k] = sf * sf;
      }
    }
}
template <class T>
void corrections_reference(const T* raw, const T* q, const int* batch_idx, const T* vol, const T* alpha, const T* qtot, int N,
                           T* E, T* dEdq) {
  const T pi = T(3.14159265358979323846), two = 2;
  for (int i = 0; i < N; ++i) {
    int s = batch_idx ? batch_idx[i] : 0;
    T c = q[i], a = alpha[s];
    E[i] = c * raw[i] - c * c * a / std::sqrt(pi) - c * pi * qtot[s] / (two * a * a * vol[s]);
    if (dEdq) dEdq[i] = two * raw[i] - two * a * c / std::sqrt(pi) - pi * qtot[s] / (a * a * vol[s]);
  }
}

}  // namespace

// =====================================================================================
// C entry points (ctypes).  dtype: 0 = float32, 1 = float64.
// =====================================================================================
#define DISPATCH(dtype, CALL32, CALL64) do { if ((dtype) == 0) { CALL32; } else { CALL64; } } while (0)

extern "C" {

int orc_cell_list(int dtype, const void* pos, int N, const void* cell, const uint8_t* pbc, const int* batch_idx, int B,
                  double cutoff, int max_nbins, int M, int fill_value, int half_fill, int* nm, int* nsh, int* num,
                  int* out_cpd, int* out_radius) {
  int C = 0;
  DISPATCH(dtype,
           C = cell_list_reference<float>((const float*)pos, N, (const float*)cell, pbc, batch_idx, B, float(cutoff), max_nbins, M, fill_value, half_fill, nm, nsh, num, out_cpd, out_radius),
           C = cell_list_reference<double>((const double*)pos, N, (const double*)cell, pbc, batch_idx, B, cutoff, max_nbins, M, fill_value, half_fill, nm, nsh, num, out_cpd, out_radius));
  return C;
}

void orc_cell_cache(int dtype, const void* pos, int N, const void* cell, const uint8_t* pbc, const int* batch_idx, int B, double cutoff, int C,
                    int* cpd, int* atom_shift, int* atom_cell, int* counts, int* starts, int* cell_atoms) {
  DISPATCH(dtype,
           cell_cache_reference<float>((const float*)pos, N, (const float*)cell, pbc, batch_idx, B, float(cutoff), C, cpd, atom_shift, atom_cell, counts, starts, cell_atoms),
           cell_cache_reference<double>((const double*)pos, N, (const double*)cell, pbc, batch_idx, B, cutoff, C, cpd, atom_shift, atom_cell, counts, starts, cell_atoms));
}
int orc_cells_changed(int dtype, const void* pos, int N, const void* cell, const int* atom_to_cell, const int* cpd, const uint8_t* pbc) {
  int r = 0;
  DISPATCH(dtype, r = cells_changed_reference<float>((const float*)pos, N, (const float*)cell, atom_to_cell, cpd, pbc),
           r = cells_changed_reference<double>((const double*)pos, N, (const double*)cell, atom_to_cell, cpd, pbc));
  return r;
}
int orc_moved_beyond_skin(int dtype, const void* ref, const void* cur, int N, double threshold) {
  int r = 0;
  DISPATCH(dtype, r = moved_beyond_skin_reference<float>((const float*)ref, (const float*)cur, N, float(threshold)),
           r = moved_beyond_skin_reference<double>((const double*)ref, (const double*)cur, N, threshold));
  return r;
}

void orc_naive(int dtype, const void* pos, int N, const void* cell, const uint8_t* pbc, double cutoff, int M, int fill_value,
               int half_fill, int* nm, int* nsh, int* num) {
  // naive.py:290,388: cutoff squared in Python double, THEN cast to the positions dtype
  DISPATCH(dtype,
           naive_reference<float>((const float*)pos, N, (const float*)cell, pbc, float(cutoff * cutoff), cutoff, M, fill_value, half_fill, nm, nsh, num),
           naive_reference<double>((const double*)pos, N, (const double*)cell, pbc, cutoff * cutoff, cutoff, M, fill_value, half_fill, nm, nsh, num));
}

// The dual-cutoff kernels (naive_dual_cutoff.py:115-290) walk ONE shift table, built for cutoff2 (:835), and nest the cutoff1 test inside
// the cutoff2 test: list 1 of a dual call = the naive search for cutoff1 with the IMAGE RANGE of cutoff2.
void orc_naive_range(int dtype, const void* pos, int N, const void* cell, const uint8_t* pbc, double cutoff, double range_cutoff, int M,
                     int fill_value, int half_fill, int* nm, int* nsh, int* num) {
  DISPATCH(dtype,
           naive_reference<float>((const float*)pos, N, (const float*)cell, pbc, float(cutoff * cutoff), range_cutoff, M, fill_value, half_fill, nm, nsh, num),
           naive_reference<double>((const double*)pos, N, (const double*)cell, pbc, cutoff * cutoff, range_cutoff, M, fill_value, half_fill, nm, nsh, num));
}

void orc_dftd3(int dtype, const void* pos, const int* numbers, int N, const int* jidx, const int* ushift, const int* ptr, int M,
               int fill_value, const void* cell, const int* batch_idx, int B, const float* rcov, const float* r4r2,
               const float* c6ab, const float* cnref, int nz, double a1, double a2, double s6, double s8, double k1, double k3,
               double s5_on, double s5_off, int compute_virial, float* energy, float* forces, float* cn, float* virial) {
  D3Par P{rcov, r4r2, c6ab, cnref, nz, float(k1), float(k3), float(a1), float(a2), float(s6), float(s8), float(s5_on), float(s5_off), 0.0f};
  P.inv_w = (s5_off > s5_on) ? float(1.0 / (s5_off - s5_on)) : 0.0f;  // dftd3.py:1983-1986 (double on host, then cast)
  DISPATCH(dtype,
           dftd3_reference<float>((const float*)pos, numbers, N, jidx, ushift, ptr, M, fill_value, (const float*)cell, batch_idx, P, compute_virial, energy, forces, cn, virial, B),
           dftd3_reference<double>((const double*)pos, numbers, N, jidx, ushift, ptr, M, fill_value, (const double*)cell, batch_idx, P, compute_virial, energy, forces, cn, virial, B));
}

void orc_ewald_real(int dtype, const void* pos, const void* q, const void* cell, const void* alpha, const int* batch_idx, int N,
                    const int* jidx, const int* ushift, const int* ptr, int M, int mask_value, int want_forces, int want_cg,
                    double* energies, void* forces, double* cgrad) {
  DISPATCH(dtype,
           ewald_real_reference<float>((const float*)pos, (const float*)q, (const float*)cell, (const float*)alpha, batch_idx, N, jidx, ushift, ptr, M, mask_value, want_forces, want_cg, energies, (float*)forces, cgrad),
           ewald_real_reference<double>((const double*)pos, (const double*)q, (const double*)cell, (const double*)alpha, batch_idx, N, jidx, ushift, ptr, M, mask_value, want_forces, want_cg, energies, (double*)forces, cgrad));
}

void orc_spline(int dtype, int mode, const void* pos, const void* values, const int* batch_idx, const void* cell_inv_t, int N,
                int B, const int* dims, int order, void* mesh, void* out) {
  DISPATCH(dtype,
           spline_reference<float>(mode, (const float*)pos, (const float*)values, batch_idx, (const float*)cell_inv_t, N, B, dims, order, (float*)mesh, (float*)out),
           spline_reference<double>(mode, (const double*)pos, (const double*)values, batch_idx, (const double*)cell_inv_t, N, B, dims, order, (double*)mesh, (double*)out));
}

void orc_green_sf(int dtype, const void* k2, const void* alpha, const void* volume, int B, int nx, int ny, int nz, int order,
                  void* G, void* sf2) {
  DISPATCH(dtype,
           green_sf_reference<float>((const float*)k2, (const float*)alpha, (const float*)volume, B, nx, ny, nz, order, (float*)G, (float*)sf2),
           green_sf_reference<double>((const double*)k2, (const double*)alpha, (const double*)volume, B, nx, ny, nz, order, (double*)G, (double*)sf2));
}

void orc_corrections(int dtype, const void* raw, const void* q, const int* batch_idx, const void* vol, const void* alpha,
                     const void* qtot, int N, void* E, void* dEdq) {
  DISPATCH(dtype,
           corrections_reference<float>((const float*)raw, (const float*)q, batch_idx, (const float*)vol, (const float*)alpha, (const float*)qtot, N, (float*)E, (float*)dEdq),
           corrections_reference<double>((const double*)raw, (const double*)q, batch_idx, (const double*)vol, (const double*)alpha, (const double*)qtot, N, (double*)E, (double*)dEdq));
}

double orc_erfc(int dtype, double x) { return dtype == 0 ? double(erfc_as<float>(float(x))) : erfc_as<double>(x); }

int orc_set_d3_wide_sums(int on) { int old = g_d3_wide; g_d3_wide = on != 0; return old; }

int orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

int orc_set_extended(int on) { int old = g_extended; g_extended = on != 0; return old; }

int orc_version() { return 2; }

}  // extern "C"

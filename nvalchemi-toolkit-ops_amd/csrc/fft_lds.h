// In-LDS 3-D FFT pieces of the fused PME mesh solve (csrc/pme.hip: pme_solve_*_kernel).
//
// The reference runs torch.fft.rfftn -> three elementwise spectrum kernels -> four torch.fft.irfftn (pme.py:1398-1440).  A library 3-D
// FFT is three passes over HBM per transform; here the whole k-space step of a mesh whose sizes are products of 2, 3 and 5 (round 6: mixed
// radix; rounds 4 - 5: powers of two only) is four kernels that touch their data once:
//   A   one block per (system, x) plane:        real rows -> packed R2C along z -> FFT along y, the plane never leaves LDS
//   B1  one block per 16 (y,z) columns:         FFT along x (radix 16 x 8), times the Green function / B-spline moduli on the way out, in place
//   B2  one block per (16 columns, channel):    (1 | -i k_d) on the way in, inverse FFT along x; the channels of a tile share one XCD's L2
//   C   persistent block per CU over the (system, channel, x) planes:  inverse FFT along y -> packed C2R along z -> real rows
// The 1-D transforms are in-place decimation-in-frequency forward (natural order in, digit-reversed "slots" out) and the mirrored
// decimation-in-time inverse (slots in, natural order out): no reordering pass, every butterfly reads and writes the same R addresses, so
// a stage needs no second buffer and one barrier.  Between the kernels the spectra stay in slot order along x, y and z (the k-space factor
// looks its Miller indices up in a small per-call table).  The first / last stage of a transform that touches HBM reads / writes it directly
// (no staging copy) wherever consecutive lanes then touch consecutive addresses.
//
// Everything in here is written as per-item bodies over (tid, nthreads) with MI_FFT_SYNC() between phases, so that the same code runs as
// one "thread" on the host (tests/native/fft_host_harness.cpp: index arithmetic checked against numpy without a GPU).  TEST-ONLY host
// use: the product never runs these bodies on the CPU.
#pragma once
#include <math.h>
#include <stddef.h>

#ifdef __HIPCC__
#define MI_HD __host__ __device__ __forceinline__
#else
#define MI_HD inline
#endif

namespace mifft {

template <class T> struct alignas(2 * sizeof(T)) Cx { T re, im; };

template <class T> MI_HD Cx<T> cadd(Cx<T> a, Cx<T> b) { return Cx<T>{a.re + b.re, a.im + b.im}; }
template <class T> MI_HD Cx<T> csub(Cx<T> a, Cx<T> b) { return Cx<T>{a.re - b.re, a.im - b.im}; }
template <class T> MI_HD Cx<T> cmul(Cx<T> a, Cx<T> b) { return Cx<T>{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
template <class T> MI_HD Cx<T> cconj(Cx<T> a) { return Cx<T>{a.re, -a.im}; }
// a * (-i) for SIGN = -1 (forward), a * (+i) for SIGN = +1
template <int SIGN, class T> MI_HD Cx<T> crot(Cx<T> a) { return SIGN < 0 ? Cx<T>{a.im, -a.re} : Cx<T>{-a.im, a.re}; }

// division of an index by a small runtime constant: a shift for powers of two, else multiply-high by a magic number,
// exact for x * d < 2^32 (items stay below 2^21, d below 2^10)
struct FastDiv {
  unsigned magic;
  unsigned short d;
  signed char shift;  // >= 0: d is 2^shift
};  // (8 bytes: the plans travel as kernel arguments)
MI_HD FastDiv make_div(unsigned d) {
  FastDiv f;
  f.d = (unsigned short)d;
  f.shift = -1;
  for (int s = 0; s < 16; ++s)
    if (d == (1u << s)) f.shift = (signed char)s;
  f.magic = (unsigned)(0x100000000ull / d) + 1u;
  return f;
}
MI_HD unsigned fdiv(const FastDiv& f, unsigned x) { return f.shift >= 0 ? x >> f.shift : (unsigned)(((unsigned long long)x * f.magic) >> 32); }

// ---- 1-D plan: n = product of radices out of {16, 8, 4, 2, 5, 3}, n <= 512 -------------------------------------------------------------------
// Stage s transforms sub-sequences of length L_s = n / (R_0 ... R_{s-1}) with radix R_s; sub_s = L_s / R_s.  Powers of two keep their shifts
// (FastDiv degenerates to one); other lengths pay a multiply-high per index.
#define MI_FFT_MAX_STAGES 6  // n <= 512: 486 = 2 x 3^5 is the longest chain
struct Plan {
  int n, nst;
  short radix[MI_FFT_MAX_STAGES], len[MI_FFT_MAX_STAGES] /* L_s */, sub[MI_FFT_MAX_STAGES] /* L_s / R_s */;
  short per[MI_FFT_MAX_STAGES];       // n / R_s: butterflies of one line in stage s
  short wstep[MI_FFT_MAX_STAGES];     // n / L_s: stride of stage s in a table of n unit roots
  FastDiv dsub[MI_FFT_MAX_STAGES];  // division by sub_s
  FastDiv dper[MI_FFT_MAX_STAGES];  // division by per_s
  FastDiv drad[MI_FFT_MAX_STAGES];  // division by R_s
  // powers of two keep the index arithmetic of rounds 4 - 5 (shifts and masks: the generic multiply-high / multiply form cost the headline
  // mesh's kernels 8 - 19 %): p2 != 0 <=> every radix is a power of two; then lgr = log2 R_s, lgsub = log2 sub_s, lglen = log2 L_s
  int p2;
  signed char lgr[MI_FFT_MAX_STAGES], lgsub[MI_FFT_MAX_STAGES], lglen[MI_FFT_MAX_STAGES], lgw[MI_FFT_MAX_STAGES] /* log2 (n / L_s) */;
  signed char lgper[MI_FFT_MAX_STAGES];  // log2 (n / R_s)
  // GEN (template parameter of the bodies below): false = the caller guarantees p2 for every plan it passes and only the shift / mask forms
  // and the power-of-two butterflies are compiled (the kernels of power-of-two meshes: same code as rounds 4 - 5, no radix-3 / radix-5 paths
  // in their instruction stream); true = both forms, chosen per plan at run time (mixed-radix meshes, and the host test harness)
};  // (every quotient the kernels need per item is a FastDiv or a stored integer: no runtime integer division in the bodies)
MI_HD bool plan_ok(int n) {
  if (n < 2 || n > 512) return false;
  int m = n;
  while (m % 2 == 0) m /= 2;
  while (m % 3 == 0) m /= 3;
  while (m % 5 == 0) m /= 5;
  return m == 1;
}
// max_lr: log2 of the largest power-of-two radix (4: radix 16 -- 64 VGPRs of fp64 points per butterfly, for kernels that can afford them; 3: radix 8)
MI_HD Plan make_plan(int n, int max_lr = 3) {
  Plan p;
  p.n = n;
  p.nst = 0;
  for (int s = 0; s < MI_FFT_MAX_STAGES; ++s) {
    p.radix[s] = 1; p.len[s] = 1; p.sub[s] = 1; p.per[s] = (short)n; p.wstep[s] = 1;
    p.dsub[s] = make_div(1); p.dper[s] = make_div(1); p.drad[s] = make_div(1);
    p.lgr[s] = p.lgsub[s] = p.lglen[s] = p.lgw[s] = p.lgper[s] = 0;
  }
  p.p2 = (n & (n - 1)) == 0 ? 1 : 0;
  int lg = 0, m = n;
  while (m % 2 == 0) { m /= 2; ++lg; }
  // the power-of-two part: as few stages as the radix allows, the large radices first
  const int nst2 = (lg + max_lr - 1) / max_lr;
  int rem = lg;
  for (int s = 0; s < nst2; ++s) {
    const int left = nst2 - s;
    const int l = (rem + left - 1) / left;  // ceil: 7 -> 4 + 3, 9 -> 3 + 3 + 3, 5 -> 3 + 2
    p.radix[p.nst++] = (short)(1 << l);
    rem -= l;
  }
  // then the fives and the threes, one radix each (the last stage walks adjacent elements: keep it a small one)
  while (m % 5 == 0) { m /= 5; p.radix[p.nst++] = 5; }
  while (m % 3 == 0) { m /= 3; p.radix[p.nst++] = 3; }
  int L = n;
  for (int s = 0; s < p.nst; ++s) {
    p.len[s] = (short)L;
    p.sub[s] = (short)(L / p.radix[s]);
    p.per[s] = (short)(n / p.radix[s]);
    p.wstep[s] = (short)(n / L);
    p.dsub[s] = make_div((unsigned)p.sub[s]);
    p.dper[s] = make_div((unsigned)p.per[s]);
    p.drad[s] = make_div((unsigned)p.radix[s]);
    if (p.p2) {
      auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
      p.lgr[s] = (signed char)lg2(p.radix[s]); p.lgsub[s] = (signed char)lg2(p.sub[s]); p.lglen[s] = (signed char)lg2(L); p.lgw[s] = (signed char)lg2(n / L);
      p.lgper[s] = (signed char)lg2(n / p.radix[s]);
    }
    L = p.sub[s];
  }
  return p;
}
// slot p of the forward output holds frequency k:  p = q0 n/R0 + q1 n/(R0 R1) + ...,  k = q0 + R0 q1 + R0 R1 q2 + ...
MI_HD int slot_freq(const Plan& p, int slot) {
  int k = 0, w = 1;
  for (int s = 0; s < p.nst; ++s) {
    const int q = (int)fdiv(p.dsub[s], (unsigned)slot);
    slot -= q * p.sub[s];
    k += q * w;
    w *= p.radix[s];
  }
  return k;
}
template <bool GEN = true> MI_HD int freq_slot(const Plan& p, int k) {
  int slot = 0;
  if (!GEN || p.p2) {
    for (int s = 0; s < p.nst; ++s) {
      slot += (k & (p.radix[s] - 1)) << p.lgsub[s];
      k >>= p.lgr[s];
    }
    return slot;
  }
  for (int s = 0; s < p.nst; ++s) {
    const int kq = (int)fdiv(p.drad[s], (unsigned)k);
    slot += (k - kq * p.radix[s]) * p.sub[s];
    k = kq;
  }
  return slot;
}

// ---- register butterflies: y_q = sum_r x_r exp(SIGN 2 pi i r q / R) --------------------------------------------------------------------
template <int SIGN, class T> MI_HD void dft2(Cx<T>* v) {
  const Cx<T> a = v[0], b = v[1];
  v[0] = cadd(a, b);
  v[1] = csub(a, b);
}
template <int SIGN, class T> MI_HD void dft4(Cx<T>* v) {
  const Cx<T> a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]), a2 = cadd(v[1], v[3]), a3 = crot<SIGN>(csub(v[1], v[3]));
  v[0] = cadd(a0, a2);
  v[2] = csub(a0, a2);
  v[1] = cadd(a1, a3);
  v[3] = csub(a1, a3);
}
template <int SIGN, class T> MI_HD void dft8(Cx<T>* v) {
  const T h = T(0.70710678118654752440);
  Cx<T> u[4], w[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    u[r] = cadd(v[r], v[r + 4]);
    w[r] = csub(v[r], v[r + 4]);
  }
  // w_r *= exp(SIGN i pi r / 4)
  w[1] = SIGN < 0 ? Cx<T>{h * (w[1].re + w[1].im), h * (w[1].im - w[1].re)} : Cx<T>{h * (w[1].re - w[1].im), h * (w[1].im + w[1].re)};
  w[2] = crot<SIGN>(w[2]);
  w[3] = SIGN < 0 ? Cx<T>{h * (w[3].im - w[3].re), -h * (w[3].re + w[3].im)} : Cx<T>{-h * (w[3].re + w[3].im), h * (w[3].re - w[3].im)};
  dft4<SIGN>(u);
  dft4<SIGN>(w);
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    v[2 * m] = u[m];
    v[2 * m + 1] = w[m];
  }
}
template <int SIGN, class T> MI_HD void dft16(Cx<T>* v) {
  const T c1 = T(0.92387953251128675613), s1 = T(0.38268343236508977173), h = T(0.70710678118654752440);
  Cx<T> u[8], w[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    u[r] = cadd(v[r], v[r + 8]);
    w[r] = csub(v[r], v[r + 8]);
  }
  // w_r *= exp(SIGN i pi r / 8)
  const T sg = SIGN < 0 ? T(-1) : T(1);
  w[1] = cmul(w[1], Cx<T>{c1, sg * s1});
  w[2] = cmul(w[2], Cx<T>{h, sg * h});
  w[3] = cmul(w[3], Cx<T>{s1, sg * c1});
  w[4] = crot<SIGN>(w[4]);
  w[5] = cmul(w[5], Cx<T>{-s1, sg * c1});
  w[6] = cmul(w[6], Cx<T>{-h, sg * h});
  w[7] = cmul(w[7], Cx<T>{-c1, sg * s1});
  dft8<SIGN>(u);
  dft8<SIGN>(w);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    v[2 * m] = u[m];
    v[2 * m + 1] = w[m];
  }
}
template <int SIGN, class T> MI_HD void dft3(Cx<T>* v) {
  const T c = T(-0.5), sn = (SIGN < 0 ? T(-1) : T(1)) * T(0.86602540378443864676);  // exp(SIGN 2 pi i / 3) = c + i sn
  const Cx<T> a = v[0], t = cadd(v[1], v[2]), d = csub(v[1], v[2]);
  const Cx<T> m{a.re + c * t.re, a.im + c * t.im};
  const Cx<T> r{-sn * d.im, sn * d.re};  // i sn d
  v[0] = cadd(a, t);
  v[1] = cadd(m, r);
  v[2] = csub(m, r);
}
template <int SIGN, class T> MI_HD void dft5(Cx<T>* v) {
  const T sg = SIGN < 0 ? T(-1) : T(1);
  const T c1 = T(0.30901699437494742410), c2 = T(-0.80901699437494742410);           // cos(2 pi / 5), cos(4 pi / 5)
  const T s1 = sg * T(0.95105651629515357212), s2 = sg * T(0.58778525229247312917);  // SIGN sin(2 pi / 5), SIGN sin(4 pi / 5)
  const Cx<T> a = v[0], t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), d1 = csub(v[1], v[4]), d2 = csub(v[2], v[3]);
  const Cx<T> m1{a.re + c1 * t1.re + c2 * t2.re, a.im + c1 * t1.im + c2 * t2.im};
  const Cx<T> m2{a.re + c2 * t1.re + c1 * t2.re, a.im + c2 * t1.im + c1 * t2.im};
  const Cx<T> r1{-(s1 * d1.im + s2 * d2.im), s1 * d1.re + s2 * d2.re};  // i (s1 d1 + s2 d2)
  const Cx<T> r2{-(s2 * d1.im - s1 * d2.im), s2 * d1.re - s1 * d2.re};  // i (s2 d1 - s1 d2)
  v[0] = cadd(a, cadd(t1, t2));
  v[1] = cadd(m1, r1);
  v[4] = csub(m1, r1);
  v[2] = cadd(m2, r2);
  v[3] = csub(m2, r2);
}
template <int SIGN, int R, class T> MI_HD void dftR(Cx<T>* v) {
  if (R == 2) dft2<SIGN>(v);
  if (R == 3) dft3<SIGN>(v);
  if (R == 4) dft4<SIGN>(v);
  if (R == 5) dft5<SIGN>(v);
  if (R == 8) dft8<SIGN>(v);
  if (R == 16) dft16<SIGN>(v);
}

// One butterfly of one stage.  Stage s of plan `pl` (sub-transform length L = pl.len[s], radix R = pl.radix[s], sub = L / R): butterfly j in
// [0, n / R) works on the points blk * L + o + r * sub (r < R; blk = j / sub, o = j % sub) -- `ld(point)` fetches them, `st(point, value)`
// stores the results to the same point numbers.  W: table of exp(-2 pi i t / NW), NW = wmul * n (wmul = 1, or 2 for the packed real rows,
// whose M-point transforms read the table of nz = 2 M roots).
// Forward (decimation in frequency): butterfly, then twiddle exp(-2 pi i o q / L).  Inverse (decimation in time): conjugate twiddle, then
// butterfly -- R times the exact inverse of the forward stage, so forward stages 0..S-1 followed by inverse stages S-1..0 give n * identity.
template <int R, bool GEN = true, class T, class Ld> MI_HD void butterfly_load(Cx<T>* v, const Plan& pl, int s, int j, Ld ld) {
  if (!GEN || pl.p2) {
    const int lsub = pl.lgsub[s];
    const int blk = j >> lsub, o = j & ((1 << lsub) - 1);
    const int base = (blk << pl.lglen[s]) + o;
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = ld(base + (r << lsub));
    return;
  }
  const int sub = pl.sub[s];
  const int blk = (int)fdiv(pl.dsub[s], (unsigned)j), o = j - blk * sub;
  const int base = blk * pl.len[s] + o;
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = ld(base + r * sub);
}
template <int SIGN, int R, bool GEN = true, class T, class St> MI_HD void butterfly_finish(Cx<T>* v, const Cx<T>* W, int wmul, const Plan& pl, int s, int j, St st) {
  int sub, base, wl;  // wl = o * (NW / L), NW = wmul * n the size of the table; q * wl < NW for q < R: o < sub = L / R
  if (!GEN || pl.p2) {
    const int lsub = pl.lgsub[s];
    const int blk = j >> lsub, o = j & ((1 << lsub) - 1);
    sub = 1 << lsub;
    base = (blk << pl.lglen[s]) + o;
    wl = (o << pl.lgw[s]) * wmul;
  } else {
    sub = pl.sub[s];
    const int blk = (int)fdiv(pl.dsub[s], (unsigned)j), o = j - blk * sub;
    base = blk * pl.len[s] + o;
    wl = o * pl.wstep[s] * wmul;
  }
  if (SIGN > 0) {
#pragma unroll
    for (int q = 1; q < R; ++q) v[q] = cmul(v[q], cconj(W[q * wl]));
  }
  dftR<SIGN, R>(v);
  if (SIGN < 0) {
#pragma unroll
    for (int q = 1; q < R; ++q) v[q] = cmul(v[q], W[q * wl]);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) st(base + r * sub, v[r]);
}
template <int SIGN, int R, bool GEN = true, class T, class Ld, class St>
MI_HD void butterfly(const Cx<T>* W, int wmul, const Plan& pl, int s, int j, Ld ld, St st) {
  Cx<T> v[R];
  butterfly_load<R, GEN>(v, pl, s, j, ld);
  butterfly_finish<SIGN, R, GEN>(v, W, wmul, pl, s, j, st);
}
// MAXLR: the largest power-of-two radix the caller's plans contain (make_plan's max_lr) -- radix 16 is only instantiated where it can occur
template <int SIGN, int MAXLR = 3, bool GEN = true, class T, class Ld, class St>
MI_HD void butterfly_r(const Plan& pl, int s, const Cx<T>* W, int wmul, int j, Ld ld, St st) {
  const int R = pl.radix[s];
  if (MAXLR >= 4 && R == 16) butterfly<SIGN, MAXLR >= 4 ? 16 : 8, GEN>(W, wmul, pl, s, j, ld, st);
  else if (R == 8) butterfly<SIGN, 8, GEN>(W, wmul, pl, s, j, ld, st);
  else if (R == 4) butterfly<SIGN, 4, GEN>(W, wmul, pl, s, j, ld, st);
  else if (!GEN || R == 2) butterfly<SIGN, 2, GEN>(W, wmul, pl, s, j, ld, st);
  else if (R == 5) butterfly<SIGN, GEN ? 5 : 2, GEN>(W, wmul, pl, s, j, ld, st);
  else butterfly<SIGN, GEN ? 3 : 2, GEN>(W, wmul, pl, s, j, ld, st);
}

// exp(-2 pi i t / n), evaluated in double whatever T is
template <class T> MI_HD Cx<T> unit_root(int t, int n) {
  double s, c;
#ifdef __HIP_DEVICE_COMPILE__
  sincospi(-2.0 * (double)t / (double)n, &s, &c);
#else
  const double a = -2.0 * 3.14159265358979323846 * (double)t / (double)n;
  s = sin(a);
  c = cos(a);
#endif
  return Cx<T>{(T)c, (T)s};
}

#if defined(__HIP_DEVICE_COMPILE__)
#define MI_FFT_SYNC() __syncthreads()
#elif defined(MI_FFT_HOST_BARRIER)
#define MI_FFT_SYNC() MI_FFT_HOST_BARRIER()  // tests/native/fft_race_check.cpp: one host thread per "lane group", a pthread barrier, ThreadSanitizer
#else
#define MI_FFT_SYNC() ((void)0)
#endif

// item -> (line, butterfly) of a stage over n_lines lines of per = n / R butterflies.  Consecutive items (lanes) take consecutive addresses
// wherever that is conflict-free: along the line while the butterfly's own points are far apart (rows, sub >= 16), across the lines
// otherwise (rows: the pitch is odd in 16-byte units, conflict-free for b128; columns: the lines are adjacent elements).
struct ItemMap {
  bool along;
  int per, lper;
  FastDiv dper, lines;
};
template <bool GEN = true> MI_HD void item_of(const ItemMap& m, int it, int& line, int& j) {
  if (m.along) {
    if (!GEN) { line = it >> m.lper; j = it & ((1 << m.lper) - 1); }
    else { line = (int)fdiv(m.dper, (unsigned)it); j = it - line * m.per; }
  }
  else { j = (int)fdiv(m.lines, (unsigned)it); line = it - j * (int)m.lines.d; }
}

// one stage, in place, over lines of an LDS array: element (line, point) at a[line * line_stride + point * point_stride]
template <int SIGN, bool GEN = true, class T>
MI_HD void lds_stage(Cx<T>* a, const Plan& pl, int s, int n_lines, const FastDiv& lines_div, int line_stride, int point_stride, const Cx<T>* W,
                     int wmul, int tid, int nth) {
  ItemMap m;
  m.per = pl.per[s];
  m.lper = pl.lgper[s];
  m.dper = pl.dper[s];
  m.along = point_stride == 1 && pl.sub[s] >= 16;
  m.lines = lines_div;
  const int items = n_lines * m.per;
  for (int it = tid; it < items; it += nth) {
    int line, j;
    item_of<GEN>(m, it, line, j);
    Cx<T>* d = a + line * line_stride;
    butterfly_r<SIGN, 3, GEN>(pl, s, W, wmul, j, [=](int p) { return d[p * point_stride]; }, [=](int p, Cx<T> v) { d[p * point_stride] = v; });
  }
}
// all stages of a batch of 1-D transforms held in LDS, forward or inverse, a barrier after each
template <int SIGN, bool GEN = true, class T>
MI_HD void lines_fft(Cx<T>* a, const Plan& pl, int n_lines, const FastDiv& lines_div, int line_stride, int point_stride, const Cx<T>* W, int wmul,
                     int tid, int nth) {
  for (int si = 0; si < pl.nst; ++si) {
    lds_stage<SIGN, GEN>(a, pl, SIGN < 0 ? si : pl.nst - 1 - si, n_lines, lines_div, line_stride, point_stride, W, wmul, tid, nth);
    MI_FFT_SYNC();
  }
}

// ---- R2C / C2R of a row held as M = n/2 complex points z_j = (x_2j, x_2j+1) ---------------------------------------------------------------
// forward: after the M-point forward FFT (slots), X_k = 1/2 [(Z_k + conj Z_{M-k}) - i w_k (Z_k - conj Z_{M-k})], w_k = exp(-2 pi i k / n);
// X_k lands in the slot Z_k had, X_M (the Nyquist bin) in element M.  item k in [0, M/2].
template <bool GEN = true, class T> MI_HD void r2c_post_item(Cx<T>* row, const Plan& pz, const Cx<T>* Wn /*exp(-2 pi i t / n)*/, int M, int k) {
  if (k == 0) {
    const Cx<T> z = row[0];
    row[0] = Cx<T>{z.re + z.im, T(0)};
    row[M] = Cx<T>{z.re - z.im, T(0)};
    return;
  }
  const int sa = freq_slot<GEN>(pz, k), sb = freq_slot<GEN>(pz, M - k);
  const Cx<T> a = row[sa], b = row[sb];
  const T h = T(0.5);
  {
    const Cx<T> e = cadd(a, cconj(b)), o = cmul(Wn[k], csub(a, cconj(b)));  // -i * o: (o.im, -o.re)
    row[sa] = Cx<T>{h * (e.re + o.im), h * (e.im - o.re)};
  }
  if (sa != sb) {
    const Cx<T> e = cadd(b, cconj(a)), o = cmul(Wn[M - k], csub(b, cconj(a)));
    row[sb] = Cx<T>{h * (e.re + o.im), h * (e.im - o.re)};
  }
}
// inverse (unnormalised, numpy.fft.irfft convention: the imaginary parts of the DC and Nyquist bins are not read):
// Z_k = (X_k + conj X_{M-k}) + i conj(w_k) (X_k - conj X_{M-k}), then the M-point inverse FFT gives z_j = (x_2j, x_2j+1) (sum over all n bins)
template <bool GEN = true, class T> MI_HD void c2r_pre_item(Cx<T>* row, const Plan& pz, const Cx<T>* Wn, int M, int k) {
  if (k == 0) {
    const T a = row[0].re, c = row[M].re;
    row[0] = Cx<T>{a + c, a - c};
    return;
  }
  const int sa = freq_slot<GEN>(pz, k), sb = freq_slot<GEN>(pz, M - k);
  const Cx<T> a = row[sa], b = row[sb];
  {
    const Cx<T> e = cadd(a, cconj(b)), o = cmul(cconj(Wn[k]), csub(a, cconj(b)));  // +i * o: (-o.im, o.re)
    row[sa] = Cx<T>{e.re - o.im, e.im + o.re};
  }
  if (sa != sb) {
    const Cx<T> e = cadd(b, cconj(a)), o = cmul(cconj(Wn[M - k]), csub(b, cconj(a)));
    row[sb] = Cx<T>{e.re - o.im, e.im + o.re};
  }
}

// ---- geometry shared by the kernels ---------------------------------------------------------------------------------------------------
struct Geom {
  int B, nx, ny, nz, M, P;  // M = nz / 2 complex points per packed real row, P = M + 1 bins per row (and the LDS row pitch)
  int p2, lgM;              // every axis a power of two (the GEN = false kernels); log2 M then
  Plan px, py, pz;          // pz: the M-point transform of the packed rows
  FastDiv divP, divNy, divM;
};
// sizes: products of 2, 3 and 5; nz even (the real rows are packed as nz / 2 complex points)
MI_HD bool geom_ok(int nx, int ny, int nz) {
  return plan_ok(nx) && plan_ok(ny) && nz % 2 == 0 && plan_ok(nz / 2) && nx >= 8 && ny >= 8 && nz >= 8 && nx <= 256 && ny <= 256 && nz <= 512;
}
MI_HD Geom make_geom(int B, int nx, int ny, int nz) {
  Geom g;
  g.B = B; g.nx = nx; g.ny = ny; g.nz = nz; g.M = nz / 2; g.P = nz / 2 + 1;
  g.px = make_plan(nx, 4); g.py = make_plan(ny); g.pz = make_plan(nz / 2);  // x columns: one wave per block, radix 16 affordable
  g.divP = make_div((unsigned)g.P);
  g.divNy = make_div((unsigned)ny);
  g.divM = make_div((unsigned)g.M);
  g.p2 = (g.px.p2 && g.py.p2 && g.pz.p2) ? 1 : 0;
  g.lgM = 0;
  while ((1 << g.lgM) < g.M) ++g.lgM;
  return g;
}

// ---- tables of one mesh shape (computed once per shape by the library, kept on the device) ---------------------------------------------
// unit roots exp(-2 pi i t / n) of the three axes; sinc(m / n) and the Miller index m of every SLOT of the three axes (z: P bins)
MI_HD int miller_index(int i, int n) { return i < (n + 1) / 2 ? i : i - n; }  // fftfreq(n, 1/n), as the k-grid of the reference (k_vectors.py:270-282)
template <class T> MI_HD T sinc_of(T x) {  // sin(pi x) / (pi x), pme_kernels.py:208-225
  if (fabs(x) < T(1e-6)) return T(1);
  const T px = T(3.14159265358979323846) * x;
  return sin(px) / px;
}
template <class T> struct Tables {
  const Cx<T>*Wx, *Wy, *Wz;
  const T *sx, *sy, *sz, *mx, *my, *mz;
};
template <class T> MI_HD size_t tables_bytes(const Geom& g) {
  return (size_t)(g.nx + g.ny + g.nz) * sizeof(Cx<T>) + (size_t)2 * (g.nx + g.ny + g.P) * sizeof(T);
}
template <class T> MI_HD Tables<T> tables_at(void* base, const Geom& g) {
  Tables<T> t;
  Cx<T>* w = (Cx<T>*)base;
  t.Wx = w; t.Wy = w + g.nx; t.Wz = w + g.nx + g.ny;
  T* r = (T*)(w + g.nx + g.ny + g.nz);
  t.sx = r; t.sy = r + g.nx; t.sz = r + g.nx + g.ny;
  r += g.nx + g.ny + g.P;
  t.mx = r; t.my = r + g.nx; t.mz = r + g.nx + g.ny;
  return t;
}
template <class T> MI_HD void tables_body(void* base, const Geom& g, int tid, int nth) {
  const Tables<T> t = tables_at<T>(base, g);
  for (int i = tid; i < g.nx; i += nth) {
    const int m = miller_index(slot_freq(g.px, i), g.nx);
    ((Cx<T>*)t.Wx)[i] = unit_root<T>(i, g.nx);
    ((T*)t.mx)[i] = (T)m;
    ((T*)t.sx)[i] = sinc_of((T)m / (T)g.nx);
  }
  for (int i = tid; i < g.ny; i += nth) {
    const int m = miller_index(slot_freq(g.py, i), g.ny);
    ((Cx<T>*)t.Wy)[i] = unit_root<T>(i, g.ny);
    ((T*)t.my)[i] = (T)m;
    ((T*)t.sy)[i] = sinc_of((T)m / (T)g.ny);
  }
  for (int i = tid; i < g.nz; i += nth) ((Cx<T>*)t.Wz)[i] = unit_root<T>(i, g.nz);
  for (int i = tid; i < g.P; i += nth) {
    const int m = i < g.M ? slot_freq(g.pz, i) : g.M;
    ((T*)t.mz)[i] = (T)m;
    ((T*)t.sz)[i] = sinc_of((T)m / (T)g.nz);
  }
}

// LDS bytes of the plane kernels: the plane + exp(-2 pi i t / nz) + exp(-2 pi i t / ny)
template <class T> MI_HD size_t plane_lds_bytes(const Geom& g) { return ((size_t)g.ny * g.P + g.nz + g.ny) * sizeof(Cx<T>); }
// ... of their NAT forms (transforms on their own, mi_fft_lds): + the frequency of every y slot and z slot
template <class T> MI_HD size_t plane_lds_bytes_nat(const Geom& g) { return plane_lds_bytes<T>(g) + ((size_t)g.ny + g.P) * sizeof(int); }
template <bool GEN> MI_HD void plane_nat_tables(int* NATY, int* NATZ, const Geom& g, int tid, int nth) {
  for (int t = tid; t < g.ny; t += nth) NATY[t] = slot_freq(g.py, t);
  for (int t = tid; t < g.P; t += nth) NATZ[t] = t < g.M ? slot_freq(g.pz, t) : g.M;
}

template <class T> MI_HD void plane_tables(Cx<T>* Wz, Cx<T>* Wy, const Tables<T>& tb, const Geom& g, int tid, int nth) {
  for (int t = tid; t < g.nz; t += nth) Wz[t] = tb.Wz[t];
  for (int t = tid; t < g.ny; t += nth) Wy[t] = tb.Wy[t];
}

// ---- kernel A body: one (system, x) plane -------------------------------------------------------------------------------------------
// in: real plane [ny][nz]; out: [ny][P] complex in (y slot, z slot) order; lds: plane_lds_bytes
// NAT (round 6, mi_fft_lds): out in NATURAL frequency order along y and z -- the last stage's stores go to the row / bin the slot stands for
// (a wave still writes whole rows: the same lines, its lanes in another order); lds: plane_lds_bytes_nat
template <class T, bool GEN = true, bool NAT = false> MI_HD void fwd_plane_body(const T* in, Cx<T>* out, Cx<T>* lds, const Geom& g, const Tables<T>& tb, int tid, int nth) {
  Cx<T>* plane = lds;
  Cx<T>* Wz = lds + g.ny * g.P;
  Cx<T>* Wy = Wz + g.nz;
  int* NATY = (int*)(Wy + g.ny);
  int* NATZ = NATY + g.ny;
  plane_tables(Wz, Wy, tb, g, tid, nth);
  if (NAT) plane_nat_tables<GEN>(NATY, NATZ, g, tid, nth);
  const Cx<T>* src = (const Cx<T>*)in;
  for (int e = tid; e < g.ny * g.M; e += nth) {
    const int y = GEN ? (int)fdiv(g.divM, (unsigned)e) : e >> g.lgM, j = GEN ? e - y * g.M : e & (g.M - 1);
    plane[y * g.P + j] = src[e];
  }
  MI_FFT_SYNC();
  // rows: M-point forward FFT; exp(-2 pi i t / M) = Wz[2 t]: the table of nz entries serves both (NW = nz)
  lines_fft<-1, GEN>(plane, g.pz, g.ny, g.divNy, g.P, 1, Wz, 2, tid, nth);
  const int half = g.M / 2 + 1;
  for (int it = tid; it < g.ny * half; it += nth) {
    const int k = (int)fdiv(g.divNy, (unsigned)it), y = it - k * g.ny;
    r2c_post_item<GEN>(plane + y * g.P, g.pz, Wz, g.M, k);
  }
  MI_FFT_SYNC();
  // columns: ny-point forward FFT for each of the P bins; the last stage stores straight to HBM (lanes along the row: coalesced)
  for (int s = 0; s + 1 < g.py.nst; ++s) {
    lds_stage<-1, GEN>(plane, g.py, s, g.P, g.divP, 1, g.P, Wy, 1, tid, nth);
    MI_FFT_SYNC();
  }
  {
    const int s = g.py.nst - 1;
    const int items = g.P * g.py.per[s];
    const int P = g.P;
    for (int it = tid; it < items; it += nth) {
      const int j = (int)fdiv(g.divP, (unsigned)it), c = it - j * P;
      const int cz = NAT ? NATZ[c] : c;
      butterfly_r<-1, 3, GEN>(g.py, s, Wy, 1, j, [=](int p) { return plane[p * P + c]; }, [=](int p, Cx<T> v) { out[(NAT ? NATY[p] : p) * P + cz] = v; });
    }
  }
}

// ---- kernel C body: one (system, channel, x) plane ----------------------------------------------------------------------------------
// in: [ny][P] complex in (y slot, z slot) order; out: real plane [ny][nz]
// NAT (round 6, mi_fft_lds): in is in NATURAL frequency order along y and z; lds: plane_lds_bytes_nat
template <class T, bool GEN = true, bool NAT = false> MI_HD void inv_plane_body(const Cx<T>* in, T* out, Cx<T>* lds, const Geom& g, const Tables<T>& tb, int tid, int nth) {
  Cx<T>* plane = lds;
  Cx<T>* Wz = lds + g.ny * g.P;
  Cx<T>* Wy = Wz + g.nz;
  int* NATY = (int*)(Wy + g.ny);
  int* NATZ = NATY + g.ny;
  plane_tables(Wz, Wy, tb, g, tid, nth);
  if (NAT) plane_nat_tables<GEN>(NATY, NATZ, g, tid, nth);
  MI_FFT_SYNC();
  const int P = g.P;
  {  // first inverse stage of the columns reads HBM directly
    const int s = g.py.nst - 1;
    const int items = P * g.py.per[s];
    for (int it = tid; it < items; it += nth) {
      const int j = (int)fdiv(g.divP, (unsigned)it), c = it - j * P;
      const int cz = NAT ? NATZ[c] : c;
      butterfly_r<+1, 3, GEN>(g.py, s, Wy, 1, j, [=](int p) { return in[(NAT ? NATY[p] : p) * P + cz]; }, [=](int p, Cx<T> v) { plane[p * P + c] = v; });
    }
    MI_FFT_SYNC();
  }
  for (int s = g.py.nst - 2; s >= 0; --s) {
    lds_stage<+1, GEN>(plane, g.py, s, P, g.divP, 1, P, Wy, 1, tid, nth);
    MI_FFT_SYNC();
  }
  const int half = g.M / 2 + 1;
  for (int it = tid; it < g.ny * half; it += nth) {
    const int k = (int)fdiv(g.divNy, (unsigned)it), y = it - k * g.ny;
    c2r_pre_item<GEN>(plane + y * P, g.pz, Wz, g.M, k);
  }
  MI_FFT_SYNC();
  lines_fft<+1, GEN>(plane, g.pz, g.ny, g.divNy, P, 1, Wz, 2, tid, nth);
  Cx<T>* dst = (Cx<T>*)out;
  for (int e = tid; e < g.ny * g.M; e += nth) {
    const int y = GEN ? (int)fdiv(g.divM, (unsigned)e) : e >> g.lgM, j = GEN ? e - y * g.M : e & (g.M - 1);
    dst[e] = plane[y * P + j];
  }
}

// ---- kernels B1 / B2: MI_SOLVE_COLS columns (y slot, z slot) of one system, all x ---------------------------------------------------
// Sixteen 16-byte columns = 256 contiguous bytes per x; two waves per block (radix 16 x 8 for 128 points: 16 points per lane, one exchange
// through LDS), several blocks per CU for the loads in flight.  B1 transforms the columns forward along x
// and multiplies by the k-space factor G / sf^2 on its way out, in place (the spectrum is then in slot order along all three axes);
// B2 -- one block per (columns, channel) -- reads them back with the channel factor (1 | -i k_d) and transforms back.
#define MI_SOLVE_COLS 16
#define MI_SOLVE_LGCOLS 4
// LDS of B1: tile + unit roots + KX[3][nx] + sinc_x[nx] + per column KC[3][8], sinc_y sinc_z [8], origin flag [8]
template <class T> MI_HD size_t fwd_cols_lds_bytes(const Geom& g) {
  return ((size_t)g.nx * MI_SOLVE_COLS + g.nx) * sizeof(Cx<T>) + ((size_t)4 * g.nx + 5 * MI_SOLVE_COLS) * sizeof(T) + ((size_t)g.nx + MI_SOLVE_COLS) * sizeof(int);
}
// LDS of B2: tile + unit roots + KX[nx] + KC[8] of the block's channel
template <class T> MI_HD size_t inv_cols_lds_bytes(const Geom& g) {
  return ((size_t)g.nx * MI_SOLVE_COLS + g.nx) * sizeof(Cx<T>) + ((size_t)g.nx + MI_SOLVE_COLS) * sizeof(T) + ((size_t)g.nx + MI_SOLVE_COLS) * sizeof(int);
}
// spec: [nx][ny*P] complex of system b (slot order in y, z), transformed in place into conv = (FFT_x spec / sf2) * G in slot order along x
// (pme.py:1418-1419, pme_kernels.py:194-225); recip: 2 pi cell^-1 of the system (row-major 3x3, k_d = sum_e m_e recip[d][e],
// k_vectors.py:270-282); sf_expo: exponent of the B-spline modulus (decode_order().sf_exponent)
// spec_nat (NULL or [nx][ny][P] of system b): the UNFACTORED spectrum in natural frequency order -- numpy.fft.rfftn(mesh) -- for a caller that
// needs the charge spectrum itself (the backward of the autograd node); 16-byte scattered stores, one per element
// PLAIN (round 6): a transform on its own (mi_fft_lds, R2C): `spec` comes from the NAT plane kernel -- columns in natural (y, z) order -- no
// k-space factor, nothing stored back; spec_nat is the output (row f_x of every column: 256 contiguous bytes per tile row); recip / alpha /
// volume are not read
template <class T, bool GEN = true, bool PLAIN = false>
MI_HD void fwd_cols_body(Cx<T>* spec, Cx<T>* lds, const Geom& g, const Tables<T>& tb, const T* recip, T alpha, T volume, int sf_expo, int col0, int tid,
                         int nth, Cx<T>* spec_nat = nullptr) {
  const int COLS = MI_SOLVE_COLS, LGC = MI_SOLVE_LGCOLS;
  const int ncol = g.ny * g.P, nx = g.nx;
  Cx<T>* S = lds;
  Cx<T>* Wx = S + nx * COLS;
  T* KX = (T*)(Wx + nx);  // [3][nx]
  T* SX = KX + 3 * nx;    // [nx]
  T* KC = SX + nx;        // [3][COLS]
  T* SC = KC + 3 * COLS;  // [COLS] sinc_y * sinc_z
  T* OC = SC + COLS;      // [COLS] 1 where m_y = m_z = 0
  int* NATC = (int*)(OC + COLS);  // [COLS] natural-order offset f_y * P + f_z of the column
  int* NATX = NATC + COLS;        // [nx]   natural-order index f_x of the slot
  for (int t = tid; t < nx; t += nth) {
    NATX[t] = slot_freq(g.px, t);
    Wx[t] = tb.Wx[t];
    if (!PLAIN) {
      SX[t] = tb.sx[t];
      const T m = tb.mx[t];
      KX[t] = m * recip[0]; KX[nx + t] = m * recip[3]; KX[2 * nx + t] = m * recip[6];
    }
  }
  for (int c = tid; c < COLS; c += nth) {
    const int col = col0 + c < ncol ? col0 + c : ncol - 1;
    const int ys = (int)fdiv(g.divP, (unsigned)col), zs = col - ys * g.P;
    if (!PLAIN) {
      const T my = tb.my[ys], mz = tb.mz[zs];
      for (int d = 0; d < 3; ++d) KC[d * COLS + c] = my * recip[3 * d + 1] + mz * recip[3 * d + 2];
      SC[c] = tb.sy[ys] * tb.sz[zs];
      OC[c] = (my == T(0) && mz == T(0)) ? T(1) : T(0);
    }
    NATC[c] = slot_freq(g.py, ys) * g.P + (zs < g.M ? slot_freq(g.pz, zs) : g.M);
  }
  MI_FFT_SYNC();
  const T inv4a2 = PLAIN ? T(0) : T(1) / (T(4) * alpha * alpha);
  auto factor_of = [=](int xs, int c) {
    const T k0 = KX[xs] + KC[c], k1 = KX[nx + xs] + KC[COLS + c], k2v = KX[2 * nx + xs] + KC[2 * COLS + c];
    T k2 = k0 * k0 + k1 * k1 + k2v * k2v;
    if (!(k2 > T(1e-12))) k2 = T(1e-12);
    const bool origin = OC[c] != T(0) && KX[xs] == T(0) && KX[nx + xs] == T(0) && KX[2 * nx + xs] == T(0) && SX[xs] == T(1);
    const T sp = SX[xs] * SC[c];
    T sf = sp;
    for (int t = 1; t < sf_expo; ++t) sf = sf * sp;
    if (sf < T(1e-10)) sf = T(1e-10);
    return (origin || k2 < T(1e-10)) ? T(0) : T(6.283185307179586) * exp(-inv4a2 * k2) / (k2 * volume * (sf * sf));
  };
  const Plan& px = g.px;
  for (int s = 0; s < px.nst; ++s) {
    const int items = COLS * px.per[s];
    const bool first = s == 0;
    for (int it = tid; it < items; it += nth) {
      const int j = it >> LGC, c = it & (COLS - 1);
      const bool live = col0 + c < ncol;
      const Cx<T>* col = spec + col0 + c;
      butterfly_r<-1, 4, GEN>(px, s, Wx, 1, j,
                      [=](int p) { return first ? (live ? col[(size_t)p * ncol] : Cx<T>{T(0), T(0)}) : S[(p << LGC) + c]; },
                      [=](int p, Cx<T> v) { S[(p << LGC) + c] = v; });
    }
    MI_FFT_SYNC();
  }
  // k-space factor on the way out (its own pass over the tile: the butterflies above keep their registers to themselves)
  for (int e = tid; e < nx * COLS; e += nth) {
    const int xs = e >> LGC, c = e & (COLS - 1);
    if (col0 + c < ncol) {
      const Cx<T> v = S[e];
      if (!PLAIN) {
        const T f = factor_of(xs, c);
        spec[(size_t)xs * ncol + col0 + c] = Cx<T>{v.re * f, v.im * f};
      }
      if (PLAIN) spec_nat[(size_t)NATX[xs] * ncol + col0 + c] = v;
      else if (spec_nat) spec_nat[(size_t)NATX[xs] * ncol + NATC[c]] = v;
    }
  }
}
// conv: [nx][ny*P] complex of system b in slot order along x, y, z; out: [nx][ny*P] of (system b, channel ch), natural order along x again:
// the inverse x transform of conv (ch = 0, the potential) or of (-i k_d) conv (ch = 1 + d, the field components; pme.py:1455-1457)
// PLAIN (round 6): a transform on its own (mi_fft_lds, C2R): `conv` is a half spectrum in NATURAL frequency order (numpy.fft.rfftn layout,
// [nx][ny][P]) -- the first stage reads row f_x of the slot it needs, the columns stay in natural (y, z) order for the NAT plane kernel
// behind it -- ch = 0, recip is not read
template <class T, bool GEN = true, bool PLAIN = false>
MI_HD void inv_cols_body(const Cx<T>* conv, Cx<T>* out, Cx<T>* lds, const Geom& g, const Tables<T>& tb, const T* recip, int ch, int col0, int tid, int nth) {
  const int COLS = MI_SOLVE_COLS, LGC = MI_SOLVE_LGCOLS;
  const int ncol = g.ny * g.P, nx = g.nx;
  Cx<T>* S = lds;
  Cx<T>* Wx = S + nx * COLS;
  T* KX = (T*)(Wx + nx);  // [nx] of this channel's d
  T* KC = KX + nx;        // [COLS]
  int* NATC = (int*)(KC + COLS);  // PLAIN: [COLS] natural-order offset f_y * P + f_z of the column
  int* NATX = NATC + COLS;        // PLAIN: [nx]   natural-order index f_x of the slot
  const int d = ch > 0 ? ch - 1 : 0;
  for (int t = tid; t < nx; t += nth) {
    Wx[t] = tb.Wx[t];
    if (PLAIN) NATX[t] = slot_freq(g.px, t);
    else KX[t] = tb.mx[t] * recip[3 * d];
  }
  for (int c = tid; c < COLS; c += nth) {
    const int col = col0 + c < ncol ? col0 + c : ncol - 1;
    const int ys = (int)fdiv(g.divP, (unsigned)col), zs = col - ys * g.P;
    if (!PLAIN) KC[c] = tb.my[ys] * recip[3 * d + 1] + tb.mz[zs] * recip[3 * d + 2];
  }
  MI_FFT_SYNC();
  const Plan& px = g.px;
  const bool field = !PLAIN && ch > 0;
  for (int s = px.nst - 1; s >= 0; --s) {
    const int items = COLS * px.per[s];
    const bool first = s == px.nst - 1, last = s == 0;
    auto ld_of = [=](int c, bool live) {
      const Cx<T>* src = conv + col0 + c;
      return [=](int p) {
        if (!first) return S[(p << LGC) + c];
        if (!live) return Cx<T>{T(0), T(0)};
        if (PLAIN) return src[(size_t)NATX[p] * ncol];
        const Cx<T> v = src[(size_t)p * ncol];
        if (!field) return v;
        const T kd = KX[p] + KC[c];
        return Cx<T>{kd * v.im, -(kd * v.re)};
      };
    };
    auto st_of = [=](int c, bool live) {
      Cx<T>* dst = out + col0 + c;
      return [=](int p, Cx<T> v) {
        if (last) { if (live) dst[(size_t)p * ncol] = v; }
        else S[(p << LGC) + c] = v;
      };
    };
    if (first && px.radix[s] == 8 && items == 2 * nth) {
      // the stage that reads HBM: both butterflies of the lane load first (16 lines in flight per lane instead of 8 + 8 one after the other)
      const int ia = tid, ib = tid + nth;
      const int ja = ia >> LGC, ca = ia & (COLS - 1), jb = ib >> LGC, cb = ib & (COLS - 1);
      const bool la = col0 + ca < ncol, lb = col0 + cb < ncol;
      Cx<T> va[8], vb[8];
      butterfly_load<8, GEN>(va, px, s, ja, ld_of(ca, la));
      butterfly_load<8, GEN>(vb, px, s, jb, ld_of(cb, lb));
      butterfly_finish<+1, 8, GEN>(va, Wx, 1, px, s, ja, st_of(ca, la));
      butterfly_finish<+1, 8, GEN>(vb, Wx, 1, px, s, jb, st_of(cb, lb));
    } else {
      for (int it = tid; it < items; it += nth) {
        const int j = it >> LGC, c = it & (COLS - 1);
        const bool live = col0 + c < ncol;
        butterfly_r<+1, 4, GEN>(px, s, Wx, 1, j, ld_of(c, live), st_of(c, live));
      }
    }
    MI_FFT_SYNC();
  }
}

}  // namespace mifft

// d3.hip -- DFT-D3(BJ) two-body dispersion: energies, forces, coordination numbers, virial.  gfx950 / wave64.
//
// Same arithmetic as the reference kernels (interactions/dispersion/dftd3.py): `_cn_kernel_nm/_nl` (:833,1321),
// `_direct_forces_and_dE_dCN_kernel_nm/_nl` (:944,1403) with `_c6ab_interpolate` (:427), `_bj_damping` (:648),
// `_dispersion_energy_force` (:690), `_s5_switch` (:341), and `_cn_forces_contrib_kernel_nm/_nl` (:1134,1535):
// fp32 pair math, fp64 accumulation of E / F / virial, thresholds (c6 == 0 skip, exp_arg - max < -12 skip,
// w <= 1e-12, c6 < 1e-12, r < 1e-12) kept verbatim.  Different execution shape:
//   * ONE WAVE64 PER ATOM, lanes stride the atom's row / CSR range (coalesced 256 B index reads) instead of one
//     thread looping over a row; per-lane fp64 partial sums are combined with cross-lane shuffles at the end.
//   * pass 0 of the reference (materialised cartesian_shifts, 12 B per slot written then read 3x) is gone:
//     S.cell is evaluated in registers in the positions dtype (`_unit_shift_to_cartesian`, :734).
//   * the species actually present are compacted on the device (no host sync).  If their tables have Grimme's structure
//     (reference CN a property of (Z, ref index), rectangular validity -- checked bit by bit on the device) the Gaussian
//     weight factorises into per-atom halves u_a(i) v_b(j): u is wave-uniform, v is evaluated once per atom after the CN pass
//     and gathered with the neighbour record, and the pair loop holds no exponential at all; the c6 rows and BJ constants of
//     the wave's element sit in a 2.8 KB LDS slice.  Other tables run the general 25-term form from an LDS-staged
//     {c6, cn_ref_i, cn_ref_j^T} float4 table (branch-free, selects), or from the global table above 16 species.
//   * cn / chain walks: 8 waves per block in lock-step (one barrier per trip) so that consecutive atoms share the cache lines
//     their gathers pull into L1; all walks are software-pipelined three deep with predicated (not branched) validity.
//   * per-system energy / virial: per-atom values, fp64 slab sums, one atomic per system change per wave, rounded once.
#include <type_traits>
#include <mutex>

#include "common.h"
#include "binsort.h"

// Pair-math primitives.  Product build: hardware v_rsq / v_rcp / v_sqrt / v_exp (1 ulp each; the exponential with a compensated
// argument, see d3_exp).  -DMI_D3_IEEE (tests only: libnvalchemiops_d3_ieee.so, built WITHOUT -fno-hip-fp32-correctly-rounded-divide-sqrt)
// swaps in correctly rounded sqrt / divide and libm expf, so the error budget of DESIGN.md section 5 can separate "fast ops" from
// "summation order".
#ifdef MI_D3_IEEE
#define D3_RCP(x) (1.0f / (x))
#define D3_SQRT(x) sqrtf(x)
#else
#define D3_RCP(x) __builtin_amdgcn_rcpf(x)
#define D3_SQRT(x) __builtin_amdgcn_sqrtf(x)
#endif

namespace {

struct D3Dev {
  const float* rcov;
  const float* r4r2;
  const float4* tab;  // [nz,nz,25] {c6ab[zi,zj,p,q], cn_ref[zi,zj,p,q], cn_ref[zj,zi,q,p], 0}
  int nz;
  float a1, a2, s6, s8, k1, k3, s5_on, s5_off, inv_w;
  // chain records (round 6, fp32 positions only; see D3CRec): one 16-byte record per atom, the species' radii by compact id, "do not use" flag
  float4* crec;
  const float* crc;
  int* cflag;
};

// ---- chain records (round 6; the pass is `_cn_forces_contrib_kernel_nm` / `_nl`, dftd3.py:1133-1258 / :1534-1660) ----------------------------
// The chain pass gathered TWO things per neighbour: the 16-byte record {x, y, z, rcov_j} and the 4-byte dE/dCN_j of another array.  A gather
// costs by the cache lines its 64 lanes touch, whatever it loads: without the second one the pass runs in 0.61 instead of 0.74 ms and the
// headline step in 3.26 instead of 3.38 (profiles/r06_ab_chain_gathers.log).  So the energy pass, which produces dE/dCN_i, leaves ONE record
// {x, y, z, w} per atom with w = dE/dCN_i AND the atom's compact species id (<= 15 species + one code for padding atoms): the radius comes
// from a 16-entry LDS table.  The id rides in the top `eb` EXPONENT bits of w: dE/dCN is scaled by a power of two so that everything below
// 256 Ha has zeros there -- scaling by 2^k and back is exact, a zero stays a zero, values below the window's lower end (2^-54 for <= 3
// species, 2^-22 otherwise) become denormals with an absolute resolution of 2^-45 or better -- so for <= 7 species the chain pass adds
// bit for bit the numbers it added before.  8 - 15 species: one more id bit replaces the LOWEST mantissa bit (w rounded to 23 bits: half an
// fp32 ulp).  More species, |dE/dCN| >= 256 Ha or a NaN (the energy pass raises P.cflag then), fp64 positions, or the packed list found
// unusable: the two-gather walk as before.
struct D3CRec {
  bool on; int eb, eshift; unsigned emask, mbit, clear; float down, up; int pad_code;
};
__device__ __forceinline__ D3CRec d3_crec_format(int S) {
  D3CRec f;
  const int codes = S + 1;  // + the padding atoms' code (its table radius is negative, like apos.w of such an atom)
  f.on = codes <= 16;
  f.eb = codes <= 4 ? 2 : 3;
  f.mbit = codes <= 8 ? 0u : 1u;
  f.eshift = 31 - f.eb;
  f.emask = ((1u << f.eb) - 1u) << f.eshift;
  f.clear = ~(f.emask | f.mbit);
  // |d| < 2^9 has exponent field <= 135; it must land on <= 2^(8 - eb) - 1 (63 or 31)
  const int k = 135 - ((1 << (8 - f.eb)) - 1);
  f.down = __int_as_float((127 - k) << 23);
  f.up = __int_as_float((127 + k) << 23);
  f.pad_code = S;
  return f;
}
__device__ __forceinline__ float d3_crec_encode(float d, int id, const D3CRec& f, bool* bad) {
  *bad = !(fabsf(d) < 256.0f);  // (one binade below the window's top: the rounding below may carry)
  unsigned u = (unsigned)__float_as_int(d * f.down);
  if (f.mbit) u = (u + 1u) & ~1u;  // nearest on the magnitude
  u |= ((unsigned)id << f.eshift) & f.emask;
  u |= ((unsigned)id >> f.eb) & f.mbit;
  return __int_as_float((int)u);
}
__device__ __forceinline__ int d3_crec_id(float w, const D3CRec& f) {
  const unsigned u = (unsigned)__float_as_int(w);
  return (int)(((u & f.emask) >> f.eshift) | ((u & f.mbit) << f.eb));
}
__device__ __forceinline__ float d3_crec_value(float w, const D3CRec& f) { return __int_as_float((int)((unsigned)__float_as_int(w) & f.clear)); }  // x f.up = dE/dCN

struct D3Species;
__device__ __forceinline__ int d3_species_count(const D3Species* info);
// (round 6: the blocks behind the atom and companion-check blocks of d3_pack_atoms_kernel; a launch of its own before)
struct D3Tables { const float* c6ab; const float* cnref; int nz; const D3Species* info; float4* tab; int first_block; };
__device__ __forceinline__ void d3_pack_tables_body(const float* __restrict__ c6ab, const float* __restrict__ cnref, int nz, const D3Species* __restrict__ info,
                                      float4* __restrict__ tab, long long vblock) {
  if (d3_species_count(info) <= 16) return;  // the global table is only read by the > 16 species variant of the energy pass
  const long long t = vblock * blockDim.x + threadIdx.x;
  const long long total = (long long)nz * nz * 25;
  if (t >= total) return;
  const int pq = (int)(t % 25), p = pq / 5, q = pq % 5;
  const long long zz = t / 25;
  const int zj = (int)(zz % nz), zi = (int)(zz / nz);
  const float c6 = c6ab[t];
  const float ci = cnref[t];
  const float cj = cnref[(((long long)zj * nz + zi) * 5 + q) * 5 + p];
  tab[t] = make_float4(c6, ci, cj, 0.0f);
}

template <class T> struct PairGeom { float r, rinv, rx, ry, rz; bool ok; };
struct Int3 { int a, b, c; };  // one 12-byte (dwordx3) load per pair for the unit shift

// ---- spatial order (round 3) ----------------------------------------------------------------------------------------------------
// The three passes gather one or two 16..32-byte records per neighbour, and a 64-lane gather costs by the cache LINES it touches
// (profiles/README.md 3.2).  With the atom records stored in the caller's atom order that cost depends on the caller: a lattice-ordered
// box touches ~22 lines per gather, the same box with its atoms in random order 64 (step 4.98 -> 8.82 ms).  So, when the packed-list
// path is taken, the atoms are binned on a coarse grid of the periodic cell (~16 per bin, x fastest, the order the cell-list search
// emits its rows in), the records are ALSO stored in that order, the CN pass translates each neighbour index into its place in the
// order while it packs the list (the place rides in the 4th word of the record the CN pass gathers anyway, so the translation is free),
// and the energy / chain passes gather from the ordered copies.  Rows are processed in the spatial order as well (waves of a block then
// walk nearly the same neighbours whatever the caller's order is).  Results do not depend on any of this: every sum keeps its order.
#define D3_SORT_MIN_ATOMS 2048
#define D3_SORT_PER_BIN 16.0
template <class T> __device__ __forceinline__ T d3_code_pack(int place, int z);
template <> __device__ __forceinline__ float d3_code_pack<float>(int place, int z) { return __int_as_float((place << 7) | z); }
template <> __device__ __forceinline__ double d3_code_pack<double>(int place, int z) { return (double)(((long long)place << 7) | z); }
__device__ __forceinline__ int d3_code_of(float w) { return __float_as_int(w); }
__device__ __forceinline__ int d3_code_of(double w) { return (int)(long long)w; }

// The grid of the spatial order.  With an estimate of the list's cutoff (the largest pair distance an earlier call saw, D3OrderState) it
// is the cell-list search's own grid (csrc/nlist.hip, nl_setup_kernel: k = 1 / 2 / 3 cells per cutoff by density, cells per dimension
// = floor(face * k / rc)): the rows of a list built by that search then visit the ordered records in long contiguous runs.  Without an
// estimate (first calls) a cubic grid with ~16 atoms per bin.  One thread per system; systems are assumed equally large (N / B atoms).
struct D3Grid { int cpd[3]; int off; };
template <class T>
__global__ void d3_sort_setup_kernel(const T* __restrict__ cell, int B, int N, float rc_est, D3Grid* __restrict__ grid, long long cap) {
  for (int s0 = 0; s0 < B; s0 += blockDim.x) {
    const int sidx = s0 + threadIdx.x;
    if (sidx < B) {
      T c[9], ci[9];
      for (int k = 0; k < 9; ++k) c[k] = cell[9 * (size_t)sidx + k];
      inverse3(c, ci);
      const double det = (double)c[0] * ((double)c[4] * c[8] - (double)c[5] * c[7]) - (double)c[1] * ((double)c[3] * c[8] - (double)c[5] * c[6]) +
                         (double)c[2] * ((double)c[3] * c[7] - (double)c[4] * c[6]);
      const double vol = fabs(det), ns = (double)N / (double)B;
      D3Grid g;
      if (rc_est > 0.0f && vol > 0.0) {
        const double rc = (double)rc_est;
        const double apc = ns / vol * rc * rc * rc;
        const int k = apc < 64.0 ? 1 : (apc < 512.0 ? 2 : 3);
        for (int d = 0; d < 3; ++d) {
          const T col[3] = {ci[d], ci[3 + d], ci[6 + d]};
          const T ln = sqrt(col[0] * col[0] + col[1] * col[1] + col[2] * col[2]);
          const double want = (1.0 / (double)ln) * k / (rc * (1.0 + 2e-6));
          g.cpd[d] = want >= 1024.0 ? 1024 : (want >= 1.0 ? (int)want : 1);
        }
      } else {
        int G = (int)cbrt(ns / D3_SORT_PER_BIN);
        G = G < 1 ? 1 : (G > 256 ? 256 : G);
        g.cpd[0] = g.cpd[1] = g.cpd[2] = G;
      }
      const long long per = cap / (B > 0 ? B : 1);  // bins this system may use
      while ((long long)g.cpd[0] * g.cpd[1] * g.cpd[2] > per && (g.cpd[0] > 1 || g.cpd[1] > 1 || g.cpd[2] > 1))
        for (int d = 0; d < 3; ++d) g.cpd[d] = g.cpd[d] / 2 > 1 ? g.cpd[d] / 2 : 1;
      g.off = g.cpd[0] * g.cpd[1] * g.cpd[2];  // turned into an offset below
      grid[sidx] = g;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // exclusive prefix over the systems (B is small next to N; one serial pass)
    int off = 0;
    for (int q = 0; q < B; ++q) { const int n = grid[q].off; grid[q].off = off; off += n; }
  }
}

template <class T>
__global__ void d3_sort_key_kernel(const T* __restrict__ pos, const T* __restrict__ cell, const int* __restrict__ batch_idx, int N,
                                   const D3Grid* __restrict__ grid, int* __restrict__ keys, int* __restrict__ count, int* __restrict__ incoherent) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = i < N;
  int key = 0, g[3] = {0, 0, 0}, gs = 0, n[3] = {1, 1, 1};
  if (in) {
    const int sidx = batch_idx ? batch_idx[i] : 0;
    const D3Grid G = grid[sidx];
    T c[9], ci[9];
    for (int k = 0; k < 9; ++k) c[k] = cell[9 * (size_t)sidx + k];
    inverse3(c, ci);
    const T x = pos[3 * (size_t)i], y = pos[3 * (size_t)i + 1], z = pos[3 * (size_t)i + 2];
    for (int d = 0; d < 3; ++d) {
      T f = x * ci[d] + y * ci[3 + d] + z * ci[6 + d];  // fractional coordinate (row vectors: pos = frac . cell)
      f = f - floor(f);
      const int q = (int)(f * (T)G.cpd[d]);
      n[d] = G.cpd[d];
      g[d] = q < 0 ? 0 : (q >= n[d] ? n[d] - 1 : q);  // (NaN / inf positions land in bin 0)
    }
    gs = sidx;
    key = G.off + (g[2] * n[1] + g[1]) * n[0] + g[0];  // x fastest, as the search's cell order
  }
  if (count) {  // (NULL: the launch only measures the order, see D3OrderState)
    if (in) keys[i] = key;
    bs_wave_add<false>(count, key, in);
  }
  // how spatially coherent is the caller's atom order?  consecutive atoms (i, i + 1) in neither the same nor a neighbouring bin
  const int nx_ = __shfl_down(g[0], 1, MI_WAVE), ny_ = __shfl_down(g[1], 1, MI_WAVE), nz_ = __shfl_down(g[2], 1, MI_WAVE), ns_ = __shfl_down(gs, 1, MI_WAVE);
  auto far = [](int a, int b, int m) { int d = a > b ? a - b : b - a; d = d < m - d ? d : m - d; return d > 1; };
  const bool pair = in && i + 1 < N && (threadIdx.x & (MI_WAVE - 1)) != MI_WAVE - 1;
  const bool bad = pair && (ns_ != gs || far(g[0], nx_, n[0]) || far(g[1], ny_, n[1]) || far(g[2], nz_, n[2]));
  const unsigned long long m = __ballot(bad);
  if (m && (threadIdx.x & (MI_WAVE - 1)) == 0) atomicAdd(incoherent, (int)__popcll(m));
}

// Per-atom records gathered by neighbour index j: ONE 16/32-byte load instead of x, y, z, Z, rcov as five gathers.
//   apos[j] = {x, y, z, rcov[Z_j]}   (w < 0 flags a padding atom, Z_j == 0)      in the positions dtype
//   aaux[j] = {CN_j, r4r2[Z_j], bits(Z_j << 8 | compact species id), 0}
// What mi_d3 checks ON THE DEVICE before it trusts a by-product of the neighbour search (round 6).  Both checks ride in the pack kernel:
//  * `slots`: the fingerprint of this call's inputs (common.h: every atom's index, system, position bits and scaled covalent radius, the
//    cell, k1), summed into MI_CN_SLOTS words of the workspace; the CN stage compares it with the one in the search's coordination-number
//    block and adopts those numbers only on a match (d3_cn_pre_kernel).
//  * `words`: a sampled re-derivation of the packed companion from the arrays it claims to describe (extra blocks behind the atom
//    blocks: one wave per sampled row); a mismatch, or the search's own "unusable" header flag, raises `flag`, and every pass then walks
//    the caller's arrays.  Arrays edited in bulk behind the companion's back are caught here; a single edited entry in an unsampled row
//    is not (include/nvalchemiops_hip.h says so).
#define D3_VERIFY_CHUNK 512
struct D3Guard {
  unsigned long long* slots; float K; const void* cell; int n_cell; const int* batch_idx;
  const int* nm; const int* nsh; const unsigned* words; const int* hdr_flag; int* flag; int M; int stride, phase; int atom_blocks;
};
template <class T>
__global__ void d3_pack_atoms_kernel(const T* __restrict__ pos, const int* __restrict__ numbers, int N, const float* __restrict__ rcov,
                                     const float* __restrict__ r4r2, const int* __restrict__ smap, int nz,
                                     typename Vec4<T>::type* __restrict__ apos, float4* __restrict__ aaux, float* __restrict__ forces,
                                     float* __restrict__ cn, float* __restrict__ dEdCN, float* __restrict__ e_atom, double* __restrict__ v_atom,
                                     const int* __restrict__ inv, typename Vec4<T>::type* __restrict__ apos_s, float4* __restrict__ aaux_s,
                                     typename Vec4<T>::type* __restrict__ acn, D3Guard G, D3Tables TB, float4* __restrict__ crec) {
  if ((int)blockIdx.x >= TB.first_block) {  // ---- the global species-pair table (only built for > 16 species)
    d3_pack_tables_body(TB.c6ab, TB.cnref, TB.nz, TB.info, TB.tab, (long long)blockIdx.x - TB.first_block);
    return;
  }
  if ((int)blockIdx.x >= G.atom_blocks) {  // ---- sampled check of the packed companion (block-uniform branch)
    const int lane = threadIdx.x & (MI_WAVE - 1);
    const int w = ((int)blockIdx.x - G.atom_blocks) * (blockDim.x / MI_WAVE) + threadIdx.x / MI_WAVE;
    if (w == 0 && lane == 0 && *G.hdr_flag != 0) *G.flag = 1;  // the search itself found the companion unusable (a shift outside {-1, 0, 1})
    if (G.stride <= 0) return;
    // one wave per sampled row, which compares ONE D3_VERIFY_CHUNK-slot stretch of it (chosen by hash, like the row): whole rows were 80 MB
    // per call on the headline list -- 30 us; a stretch per row is 16 MB and still 1 563 x 512 slots spread over the whole matrix
    const int chunks = 1, all_chunks = (G.M + D3_VERIFY_CHUNK - 1) / D3_VERIFY_CHUNK;
    // the sample: ONE row in every block of `stride` consecutive rows, at offset (phase + hash(block)) mod stride -- over `stride` calls with
    // consecutive phases every row is visited once, and within a call the offsets differ from block to block, so an edit with a regular
    // row pattern (every other row, every 64th) cannot sit between the samples
    const long long blk = w / chunks;
    const long long row = blk * G.stride + (long long)(((unsigned long long)G.phase + mi_mix64(0x51ull + (unsigned long long)blk)) % (unsigned long long)G.stride);
    if (row >= N) return;
    const int pick = G.stride == 1 ? -1 : (int)(mi_mix64(0x77ull + (unsigned long long)row * 1315423911ull + (unsigned long long)G.phase / (unsigned long long)G.stride) % (unsigned long long)all_chunks);
    // (stride 1 = "compare everything": the whole row)
    const int t_beg = pick < 0 ? 0 : pick * D3_VERIFY_CHUNK, t_end = pick < 0 ? G.M : (t_beg + D3_VERIFY_CHUNK < G.M ? t_beg + D3_VERIFY_CHUNK : G.M);
    bool bad = false;
#pragma unroll 4
    for (int t = t_beg + lane; t < t_end; t += MI_WAVE) {
      const long long e = row * G.M + t;
      const int j = G.nm[e];
      const Int3 sh = reinterpret_cast<const Int3*>(G.nsh)[e];
      const unsigned cx = (unsigned)(sh.a + 1), cy = (unsigned)(sh.b + 1), cz = (unsigned)(sh.c + 1);
      const bool entry = (unsigned)j < (unsigned)N;  // (a companion is only built for fill_value >= N: everything else is padding)
      const unsigned want = entry ? ((unsigned)j | ((cx & 3u) << 26) | ((cy & 3u) << 28) | (cz << 30)) : 0xffffffffu;
      bad |= G.words[e] != want || (entry && (cx > 2u || cy > 2u || cz > 2u));
    }
    if (bad) *G.flag = 1;  // benign race: every writer stores 1
    return;
  }
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (G.slots) {  // (kernel-uniform) fingerprint of the inputs; every lane of the wave takes part in the sum
    unsigned long long h = 0;
    if (k < N) {
      const int a = inv ? inv[k] : k;
      h = mi_cn_atom_hash<T>(a, G.batch_idx ? G.batch_idx[a] : 0, pos[3 * (size_t)a], pos[3 * (size_t)a + 1], pos[3 * (size_t)a + 2],
                             mi_cn_rk(numbers[a], nz, rcov, G.K));
    }
    if (k < G.n_cell) h += mi_cn_cell_hash<T>(k, reinterpret_cast<const T*>(G.cell)[k]);
    if (k == 0) h += mi_cn_scale_hash(G.K);
    mi_cn_slot_add(G.slots, k / MI_WAVE, h);
  }
  if (k >= N) return;
  const int i = inv ? inv[k] : k;  // inv: atom at position k of the spatial order (d3_sort_key_kernel), or NULL
  forces[3 * (size_t)i] = forces[3 * (size_t)i + 1] = forces[3 * (size_t)i + 2] = 0.0f;  // outputs of atoms the passes skip (Z == 0)
  cn[i] = dEdCN[i] = e_atom[i] = 0.0f;
  if (v_atom)
    for (int k = 0; k < 9; ++k) v_atom[9 * (size_t)i + k] = 0.0;
  const int z = numbers[i];
  const bool real = z > 0 && z < nz;
  typename Vec4<T>::type r;
  r.x = pos[3 * (size_t)i]; r.y = pos[3 * (size_t)i + 1]; r.z = pos[3 * (size_t)i + 2];
  r.w = real ? (T)rcov[z] : (T)-1;
  apos[i] = r;
  const int sj = real ? smap[z] : -1;
  const float4 ax = make_float4(0.0f, real ? r4r2[z] : 0.0f, __int_as_float((z << 8) | (sj & 0xff)), 0.0f);
  aaux[i] = ax;
  if (crec) {  // the chain record of an atom the energy pass skips (Z == 0): dE/dCN = 0 under the padding code; real atoms: overwritten there
    const D3CRec f = d3_crec_format(d3_species_count(TB.info));
    bool bad;
    if (f.on) crec[k] = make_float4((float)r.x, (float)r.y, (float)r.z, d3_crec_encode(0.0f, sj >= 0 ? sj : f.pad_code, f, &bad));
  }
  if (inv) {  // the same records at the atom's place in the spatial order, and the CN pass's record: position + {place, Z}
    apos_s[k] = r;
    aaux_s[k] = ax;
    r.w = d3_code_pack<T>(k, real ? z : 0);
    acn[i] = r;
  }
}

// `_compute_distance_vector_pbc` (dftd3.py:551-604): native-dtype difference (+ shift), cast to fp32, length, r<1e-12 skip
template <class T, class V4>
__device__ __forceinline__ PairGeom<T> d3_geom(const V4& pj, T pix, T piy, T piz, const Int3& sh, const T* __restrict__ cm, bool periodic) {
  PairGeom<T> g;
  T dx = pj.x - pix, dy = pj.y - piy, dz = pj.z - piz;
  if (periodic) {
    const T fs[3] = {(T)sh.a, (T)sh.b, (T)sh.c};
    T cart[3];
    // orthorhombic cell (wave-uniform test on the scalar cell entries): the six off-diagonal products are exact zeros, leave them out
    const bool ortho = cm[1] == T(0) && cm[2] == T(0) && cm[3] == T(0) && cm[5] == T(0) && cm[6] == T(0) && cm[7] == T(0);
    if (ortho) { cart[0] = fs[0] * cm[0]; cart[1] = fs[1] * cm[4]; cart[2] = fs[2] * cm[8]; }
    else rowvec_mat3(fs, cm, cart);
    dx = dx + cart[0]; dy = dy + cart[1]; dz = dz + cart[2];
  }
  g.rx = (float)dx; g.ry = (float)dy; g.rz = (float)dz;
  // r and 1/r from one hardware reciprocal square root (1 ulp) instead of an IEEE sqrt plus an IEEE divide (~25 VALU
  // instructions per pair in every pass); r < 1e-12 of the reference is r^2 < 1e-24 here
  const float r2 = g.rx * g.rx + g.ry * g.ry + g.rz * g.rz;
  g.ok = !(r2 < 1e-24f);
#ifdef MI_D3_IEEE
  g.r = sqrtf(g.ok ? r2 : 1.0f);
  const float rinv = 1.0f / g.r;
#else
  const float rinv = __builtin_amdgcn_rsqf(g.ok ? r2 : 1.0f);
  g.r = r2 * rinv;
#endif
  g.rinv = g.ok ? rinv : 0.0f;
  return g;
}

// One step of a row walk: neighbour index + unit shift of 64 consecutive entries (one per lane).
struct D3Step { int j; Int3 sh; bool in; };
// NT: non-temporal loads.  The caller's list is streamed once per pass while the 1.6 MB of atom records are gathered over and over; with the
// hint the stream does not evict the records (headline list, 4.1 GB: d3_cn 1.18 -> 1.115 ms, same-box A/B profiles/r02_ab_nt.log).  A list
// of about the size of the 256 MB Infinity Cache is better left in it: the reference's 54 000-atom configuration (1.0 GB list) LOSES 14 %
// with the hint (0.685 -> 0.785 ms), 85 750 atoms (1.65 GB) is neutral -- mi_d3 picks the CN-pass instantiation by list size (> 2 GB).
template <bool NT>
__device__ __forceinline__ D3Step d3_fetch(const int* __restrict__ idx, const Int3* __restrict__ ush3, long long e, long long end, bool periodic) {
  D3Step s;
  s.in = e < end;
  s.j = 0;
  s.sh = Int3{0, 0, 0};
  if (s.in) {
    if constexpr (NT) {
      s.j = __builtin_nontemporal_load(idx + e);
      if (periodic) { const int* u = reinterpret_cast<const int*>(ush3 + e); s.sh = Int3{__builtin_nontemporal_load(u), __builtin_nontemporal_load(u + 1), __builtin_nontemporal_load(u + 2)}; }
    } else {
      s.j = idx[e];
      if (periodic) s.sh = ush3[e];
    }
  }
  return s;
}
// Packed copy of a periodic padded list, written by the CN pass for the two passes after it: 4 B per slot instead of 16
// (index in bits 0-25, unit shift + 1 in three 2-bit fields; all ones = padding).  Only used when every shift is in {-1, 0, 1}
// and N < 2^26 -- otherwise `pk_flag` is raised and the passes read the caller's arrays as before.
#define D3_PK_INVALID 0xffffffffu
#define D3_PK_MAX_ATOMS (1 << 26)
__device__ __forceinline__ D3Step d3_fetch_pk(const unsigned* __restrict__ pk, long long e, long long end) {
  D3Step s;
  s.in = e < end;
  const unsigned w = s.in ? __builtin_nontemporal_load(pk + e) : D3_PK_INVALID;  // streamed once per pass: keep it out of the way of the gathered records
  s.j = (int)(w & 0x3ffffffu);  // padding decodes to 2^26 - 1 >= N: fails the index test below
  s.sh = Int3{(int)((w >> 26) & 3u) - 1, (int)((w >> 28) & 3u) - 1, (int)(w >> 30) - 1};
  return s;
}
// PK kernels read the packed copy unless the CN pass raised `pk_flag` (a shift outside {-1, 0, 1}): then they walk the caller's
// arrays like the plain kernels -- a wave-uniform choice, so one launch serves both cases.
template <bool PK>
__device__ __forceinline__ D3Step d3_fetch_any(const int* __restrict__ idx, const Int3* __restrict__ ush3, const unsigned* __restrict__ pk,
                                               bool use_pk, long long e, long long end, bool periodic) {
  if constexpr (PK) {
    if (use_pk) return d3_fetch_pk(pk, e, end);
  }
  return d3_fetch<false>(idx, ush3, e, end, periodic);
}
// Energy pass: its packed-list steps stay ONE register (the raw word) while they wait in the software pipeline and are decoded where
// they are used -- the decoded form (index + three shift ints + flag) in three pipeline stages was 10 loop-carried VGPRs, and
// that pass is latency-bound: registers are occupancy.
// wave-uniform values (properties of the row atom i, loaded by every lane from one address) pinned into SGPRs: the compiler cannot
// prove uniformity of a loaded value and would keep ~35 of them in VGPRs across the pair loop
__device__ __forceinline__ float d3_uni(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ double d3_uni(double x) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readfirstlane((int)b), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <bool RAW> struct D3Lazy;
template <> struct D3Lazy<false> { D3Step s; };
template <> struct D3Lazy<true> { unsigned w; };
__device__ __forceinline__ bool d3_lazy_in(const D3Lazy<false>& l) { return l.s.in; }
__device__ __forceinline__ int d3_lazy_j(const D3Lazy<false>& l) { return l.s.j; }
__device__ __forceinline__ Int3 d3_lazy_sh(const D3Lazy<false>& l) { return l.s.sh; }
__device__ __forceinline__ bool d3_lazy_in(const D3Lazy<true>& l) { return l.w != D3_PK_INVALID; }
__device__ __forceinline__ int d3_lazy_j(const D3Lazy<true>& l) { return (int)(l.w & 0x3ffffffu); }  // padding decodes to 2^26 - 1 >= N
__device__ __forceinline__ Int3 d3_lazy_sh(const D3Lazy<true>& l) {
  return Int3{(int)((l.w >> 26) & 3u) - 1, (int)((l.w >> 28) & 3u) - 1, (int)(l.w >> 30) - 1};
}
template <bool PK>
__device__ __forceinline__ D3Lazy<PK> d3_fetch_lazy(const int* __restrict__ idx, const Int3* __restrict__ ush3, const unsigned* __restrict__ pk,
                                                    long long e, long long end, bool periodic) {
  D3Lazy<PK> l;
  if constexpr (PK) l.w = e < end ? __builtin_nontemporal_load(pk + e) : D3_PK_INVALID;
  else l.s = d3_fetch<false>(idx, ush3, e, end, periodic);
  return l;
}
// The row walks below are software-pipelined three deep: while step k is evaluated, the per-atom records of step k+1 are
// already being gathered and the index/shift words of step k+2 are in flight; validity is a predicate, not a branch, so
// no load waits behind a branch on an earlier load (rocprof: the unpipelined walk spent >80 % of its wave cycles waiting).

// exp(x) as 2^(x log2 e) on the hardware v_exp_f32, with the rounding error of the product x*log2(e) compensated (two FMAs):
// ~1-2 ulp like libm's expf but without its range reduction and overflow/underflow selects.  Used where the argument is
// bounded above (counting function: <= k1; C6 interpolation: [-12, 0]); v_exp_f32 flushes to 0 by itself below.
__device__ __forceinline__ float d3_exp(float x) {
#ifdef MI_D3_IEEE
  return expf(x);
#endif
  const float L2E_HI = 1.44269502e+00f, L2E_LO = 1.92596299e-08f, LN2 = 6.93147182e-01f;
  const float t = x * L2E_HI;
  float lo = fmaf(x, L2E_HI, -t);
  lo = fmaf(x, L2E_LO, lo);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, lo * LN2, e);
}
__device__ __forceinline__ float d3_exp_neg(float x) { return d3_exp(x); }

// `_cn_counting` (dftd3.py:608-645)
__device__ __forceinline__ float d3_cn_count(float rinv, float rci, float rcj, float k1, float* dcn) {
  const float rr = (rci + rcj) * rinv;
  const float f = D3_RCP(1.0f + d3_exp(-k1 * (rr - 1.0f)));
  if (dcn) *dcn = -f * (1.0f - f) * k1 * rr * rinv;
  return f;
}

// A stored index is a neighbour iff 0 <= j < N and, in the padded-matrix layout, j < fill_value (dftd3.py:871: `j >= fill_value` is
// padding).  The reference reads out of bounds for anything else (e.g. -1 padding with the default fill_value = N); here such
// entries are padding too.  One unsigned compare against a wave-uniform limit.
template <bool CSR>
__device__ __forceinline__ unsigned d3_index_limit(int N, int fill_value) {
  return CSR ? (unsigned)N : (unsigned)min(N, max(fill_value, 0));
}

template <class T, bool CSR>
__device__ __forceinline__ void d3_row(int i, int M, const int* __restrict__ nptr, long long& beg, long long& end) {
  if (CSR) { beg = nptr[i]; end = nptr[i + 1]; }
  else { beg = (long long)i * M; end = beg + M; }
}

// Lock-step row walks (cn, chain): the D3_LS_WAVES waves of a block own consecutive atoms, whose rows list nearly the same
// neighbours in nearly the same order.  One barrier per trip keeps them within a trip of each other, so the cache lines one
// wave's gather brings into the CU's L1 serve the others (probe on the 40-Bohr list: 0.95 -> 0.80 ms for the cn walk; the
// gather, not the list stream, is what bounds these passes -- DESIGN.md 3.2).  All waves run the block's maximum trip count.
#ifndef D3_LS_WAVES
#ifndef D3_LS_WAVES
#define D3_LS_WAVES 8
#endif
#endif
#ifndef D3_CH_LS_WAVES
#define D3_CH_LS_WAVES 4  // the chain pass's own width (round 6): with ONE gather per neighbour and two trips of records in flight, 4 waves in lock-step
                          // beat 8 (0.601 -> 0.545 ms; 16: 0.82) -- fewer waves wait for the slowest gather of a trip, and L1 still serves 4 rows' shared lines
#endif
template <int W = D3_LS_WAVES>
__device__ __forceinline__ int d3_lockstep_trips(long long beg, long long end) {
  __shared__ int trips_sh[W];
  const int w = threadIdx.x / MI_WAVE;
  if ((threadIdx.x & (MI_WAVE - 1)) == 0) trips_sh[w] = (int)((end - beg + MI_WAVE - 1) / MI_WAVE);
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int k = 0; k < W; ++k) t = max(t, trips_sh[k]);
  return t;
}

// ---- pass 1: coordination numbers ------------------------------------------------------------------
// (round 5, with the 4 B/slot companion as input the pass is all gather: 3 / 2 instead of 2 / 1 is worth 0.487 -> 0.472 ms on the headline
// list, 4 / 2 the same, 8 lock-step waves still beat 4 and 16: profiles/r05_ab_cn_pipeline.log)
#ifndef D3_CN_DS
#define D3_CN_DS 3  // trips of index / shift words in flight ahead of the one being evaluated
#endif
#ifndef D3_CN_DG
#define D3_CN_DG 2  // trips of gathered atom records in flight (<= D3_CN_DS)
#endif
// PKIN (round 5): the rows are read from a packed companion the neighbour search wrote next to the matrix (`mi_nl_neighbors_packed`, 4 B/slot,
// same word format as the copy this pass otherwise writes) instead of the 16 B/slot API arrays.  `gate_flag` / `gate_want`: the launch does
// its work only when (*gate_flag != 0) == gate_want -- the PKIN launch runs when the companion's flag is clear, the plain launch beside
// it when the search raised it (a shift outside {-1, 0, 1}); one of the two exits at once, no host round trip.
// The j-side Gaussian weights of the factorised C6 interpolation (d3_weights_atom, defined with the interpolation below) are a function of
// an atom's own coordination number, so the CN stage writes them the moment it knows that number: one launch less on the chain of small
// kernels in front of the energy pass (round 6; `d3_weights_kernel` before).
struct D3Species;
struct D3Weights { const D3Species* sinfo; const float* fcr; float k3; const float4* apos_f32; float4* aw; float4* aw_s; };
__device__ __forceinline__ void d3_weights_atom(int j, int place, float cnj, const float4* __restrict__ aaux, const D3Weights& W);
#define D3_CN_PARAMS const T* __restrict__ pos, const int* __restrict__ numbers, int N, const int* __restrict__ idx, const int* __restrict__ ush, const int* __restrict__ nptr, int M, int fill_value, const T* __restrict__ cell, const int* __restrict__ batch_idx, D3Dev P, const typename Vec4<T>::type* __restrict__ apos, float4* __restrict__ aaux, float* __restrict__ cn, unsigned* __restrict__ pk_out, int* __restrict__ pk_flag, const int* __restrict__ inv, const typename Vec4<T>::type* __restrict__ acn, float4* __restrict__ aaux_s, int* __restrict__ rmax_bits, const unsigned* __restrict__ pk_in, D3Weights W
#define D3_CN_ARGS pos, numbers, N, idx, ush, nptr, M, fill_value, cell, batch_idx, P, apos, aaux, cn, pk_out, pk_flag, inv, acn, aaux_s, rmax_bits, pk_in, W
template <class T, bool CSR, bool BIG, bool SORT, bool PKIN>  // BIG: the caller's list is far larger than the Infinity Cache -> streamed with non-temporal loads
__device__ __forceinline__ void d3_cn_body(D3_CN_PARAMS, int vblock /* the block of D3_LS_WAVES consecutive rows to work as */) {
  static_assert(!PKIN || !CSR, "a packed companion belongs to a padded matrix");
  const int lane = threadIdx.x & (MI_WAVE - 1);
  const int i0 = __builtin_amdgcn_readfirstlane(vblock * (blockDim.x / MI_WAVE) + threadIdx.x / MI_WAVE);
  const int k0 = i0 < N ? i0 : N - 1;
  // SORT: rows are walked in the spatial order, the gathered record is acn[j] = {position, place of j in the order << 7 | Z_j}
  const int i = SORT ? __builtin_amdgcn_readfirstlane(inv[k0]) : k0;
  __shared__ float rc_lds[SORT ? 128 : 1];
  if (SORT) {
    for (int t = threadIdx.x; t < 128; t += blockDim.x) rc_lds[t] = t < P.nz ? P.rcov[t] : 0.0f;
    __syncthreads();
  }
  const typename Vec4<T>::type* __restrict__ arec = SORT ? acn : apos;
  const int zi = numbers[i];
  const bool live = i0 < N && zi != 0;  // idle waves still take part in the block's lock-step barriers
  const bool periodic = (cell != nullptr) && (ush != nullptr);
  T cm[9];
  if (periodic) { const T* c = cell + 9 * (size_t)(batch_idx ? batch_idx[i] : 0); for (int k = 0; k < 9; ++k) cm[k] = c[k]; }
  const T pix = pos[3 * (size_t)i], piy = pos[3 * (size_t)i + 1], piz = pos[3 * (size_t)i + 2];
  const float rci = P.rcov[zi];
  long long beg, end;
  d3_row<T, CSR>(i, M, nptr, beg, end);
  if (!live) end = beg;
  const int trips = d3_lockstep_trips(beg, end);
  // the reference sums in fp32 sequentially (dftd3.py:911); lanes hold fp64 partials here so the result is the
  // correctly rounded sum whatever the lane/iteration order
  double acc = 0.0;
  float rmx = 0.0f;  // largest pair distance of this row: every 64th row reports it (the list's cutoff, for the grid of the spatial order)
  const Int3* __restrict__ ush3 = reinterpret_cast<const Int3*>(ush);
  long long e = beg + lane;
  const unsigned jlim = d3_index_limit<CSR>(N, fill_value);
  // software pipeline: the index / shift words of D3_CN_DS trips ahead and the gathered records of D3_CN_DG trips ahead are in flight
  // while trip 0 is evaluated (the pass is latency-bound on its L2 gathers under load: registers are plentiful here, 8 waves fit anyway)
  D3Step s[D3_CN_DS + 1];
  bool v[D3_CN_DG + 1];
  typename Vec4<T>::type p[D3_CN_DG + 1];
  auto fetch = [&](long long at) { if constexpr (PKIN) return d3_fetch_pk(pk_in, at, end); else return d3_fetch<BIG>(idx, ush3, at, end, periodic); };
#pragma unroll
  for (int k = 0; k < D3_CN_DS; ++k) s[k] = fetch(e + (long long)k * MI_WAVE);
#pragma unroll
  for (int k = 0; k < D3_CN_DG; ++k) { v[k] = s[k].in && ((unsigned)s[k].j < jlim); p[k] = arec[v[k] ? s[k].j : i]; }
  for (int trip = 0; trip < trips; ++trip) {
    __syncthreads();  // lock-step: the block's waves walk rows of consecutive atoms, i.e. nearly the same neighbours in nearly the same order
    s[D3_CN_DS] = fetch(e + (long long)D3_CN_DS * MI_WAVE);
    v[D3_CN_DG] = s[D3_CN_DG].in && ((unsigned)s[D3_CN_DG].j < jlim);
    p[D3_CN_DG] = arec[v[D3_CN_DG] ? s[D3_CN_DG].j : i];
    const D3Step s0 = s[0];
    const bool v0 = v[0];
    const auto p0 = p[0];
    const int code0 = SORT ? d3_code_of(p0.w) : 0;  // place << 7 | Z
    if (pk_out && s0.in) {  // wave-uniform pointer test; one coalesced 4-byte store per slot of this trip
      const unsigned cx = (unsigned)(s0.sh.a + 1), cy = (unsigned)(s0.sh.b + 1), cz = (unsigned)(s0.sh.c + 1);
      if (v0 && (cx > 2u || cy > 2u || cz > 2u)) *pk_flag = 1;  // benign race: every writer stores 1
      const unsigned jw = SORT ? (unsigned)(code0 >> 7) : (unsigned)s0.j;  // what the later passes gather by
      __builtin_nontemporal_store(v0 ? (jw | ((cx & 3u) << 26) | ((cy & 3u) << 28) | (cz << 30)) : D3_PK_INVALID, pk_out + e);
    }
    if (__any(v0)) {  // a step of pure padding costs nothing (padded matrices are mostly padding)
      bool valid = v0 && (SORT ? (code0 & 127) != 0 : !(p0.w < (T)0));  // padding atom (Z == 0)
      const PairGeom<T> g = d3_geom<T>(p0, pix, piy, piz, s0.sh, cm, periodic);
      valid = valid && g.ok;
      const float f = d3_cn_count(g.rinv, rci, SORT ? rc_lds[code0 & 127] : (float)p0.w, P.k1, nullptr);
      acc += valid ? (double)f : 0.0;
      rmx = fmaxf(rmx, valid ? g.r : 0.0f);
    }
#pragma unroll
    for (int k = 0; k < D3_CN_DS; ++k) s[k] = s[k + 1];
#pragma unroll
    for (int k = 0; k < D3_CN_DG; ++k) { v[k] = v[k + 1]; p[k] = p[k + 1]; }
    e += MI_WAVE;
  }
  acc = wave_sum(acc);
  if (lane == 0 && live) { cn[i] = (float)acc; aaux[i].x = (float)acc; if (SORT) aaux_s[k0].x = (float)acc; }
  if (lane == 0 && i0 < N) d3_weights_atom(i, SORT ? k0 : -1, live ? (float)acc : 0.0f, aaux, W);  // (atoms outside the tables get all-zero weights)
  if (rmax_bits && (i0 & 63) == 0) {  // (wave-uniform) positive floats order like their bit patterns
#pragma unroll
    for (int o = MI_WAVE / 2; o > 0; o >>= 1) rmx = fmaxf(rmx, __shfl_xor(rmx, o, MI_WAVE));
    if (lane == 0 && rmx > 0.0f) atomicMax(rmax_bits, __float_as_int(rmx));
  }
}
template <class T, bool CSR, bool BIG, bool SORT, bool PKIN = false>
__global__ __launch_bounds__(D3_LS_WAVES * MI_WAVE) void d3_cn_kernel(D3_CN_PARAMS, const int* __restrict__ gate_flag, int gate_want) {
  if (gate_flag && ((*gate_flag != 0) != (gate_want != 0))) return;  // block-uniform (see PKIN above)
  d3_cn_body<T, CSR, BIG, SORT, PKIN>(D3_CN_ARGS, (int)blockIdx.x);
}
// The CN stage when the neighbour search summed the coordination numbers itself (mi_nl_neighbors_packed_cn, csrc/nlist.hip).  One launch of
// a small grid decides on the device: if the block's flag is clear and its fingerprint equals the one the pack kernel just computed from
// THIS call's inputs (D3Guard), the numbers are copied into cn / the per-atom records and the walk over the list does not happen; else the
// blocks take the rows in a stride loop and run the ordinary pass -- from the companion, or from the caller's arrays when `comp_flag` says
// the companion is unusable.  Padded matrix, caller's atom order (no spatial order: that pass has to write its place-coded list anyway).
#define D3_CN_PRE_GRID 2048
template <class T>
__global__ __launch_bounds__(D3_LS_WAVES * MI_WAVE) void d3_cn_pre_kernel(D3_CN_PARAMS, const int* __restrict__ comp_flag, const int* __restrict__ cn_hdr,
                                                                          const unsigned long long* __restrict__ got_slots, const float* __restrict__ cn_pre,
                                                                          int n_blocks) {
  __shared__ int adopt_sh;
  if (threadIdx.x < MI_WAVE) {
    const unsigned long long* want_slots = reinterpret_cast<const unsigned long long*>(cn_hdr) + MI_CN_SLOT_OFFSET_U64;
    unsigned long long a = want_slots[threadIdx.x], b = got_slots[threadIdx.x];
#pragma unroll
    for (int o = MI_WAVE / 2; o > 0; o >>= 1) { a += __shfl_xor(a, o, MI_WAVE); b += __shfl_xor(b, o, MI_WAVE); }
    // (the sums were taken over the list the companion describes: a companion that failed its check against the caller's arrays -- *comp_flag --
    // vouches for nothing)
    if (threadIdx.x == 0) adopt_sh = (a == b && cn_hdr[0] == 0 && *comp_flag == 0 && __int_as_float(cn_hdr[3]) == mi_cn_scale(P.k1)) ? 1 : 0;
  }
  __syncthreads();
  if (adopt_sh) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
      const int z = numbers[i];
      if (z == 0) continue;  // outputs of such atoms stay zero, as in the pass itself
      const float c = cn_pre[i];
      cn[i] = c;
      aaux[i].x = c;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) d3_weights_atom(i, -1, numbers[i] == 0 ? 0.0f : cn_pre[i], aaux, W);
    // the list's cutoff, for the grid of the spatial order (what the pass reports as the largest pair distance it met)
    if (rmax_bits && blockIdx.x == 0 && threadIdx.x == 0 && __int_as_float(cn_hdr[2]) > 0.0f) atomicMax(rmax_bits, cn_hdr[2]);
    return;
  }
  const bool from_companion = *comp_flag == 0;
  for (int vb = blockIdx.x; vb < n_blocks; vb += gridDim.x) {
    if (from_companion) d3_cn_body<T, false, true, false, true>(D3_CN_ARGS, vb);
    else d3_cn_body<T, false, true, false, false>(D3_CN_ARGS, vb);
    __syncthreads();  // the lock-step bookkeeping in LDS is reused by the next row block
  }
}
#undef D3_CN_ARGS
#undef D3_CN_PARAMS

// `_s5_switch` (dftd3.py:341-423)
__device__ __forceinline__ void d3_s5(float r, float on, float off, float inv_w, float& sw, float& dsw) {
  if (off <= on) { sw = 1.0f; dsw = 0.0f; return; }  // switch disabled (the default): kernel-uniform, a scalar branch around the polynomial
  if (r <= on) { sw = 1.0f; dsw = 0.0f; return; }
  if (r >= off) { sw = 0.0f; dsw = 0.0f; return; }
  const float t = (r - on) * inv_w, t2 = t * t, t3 = t2 * t, t4 = t3 * t, t5 = t4 * t;
  sw = 1.0f - (10.0f * t3 - 15.0f * t4 + 6.0f * t5);
  dsw = (-30.0f * t2 + 60.0f * t3 - 30.0f * t4) * inv_w;
}

#define D3_SMAX 16  // species held in LDS per wave (16 x 25 float4 = 6.4 KB); more species fall back to the global table

struct D3Species { int S; int factorized; int pad[2]; };
__device__ __forceinline__ int d3_species_count(const D3Species* info) { return info->S; }
#define D3_FROW 44  // factorised block per partner species: 5 c6 rows x 8 floats (b = 0..4 used) + {q, r0^6, r0^8, 0} of the BJ damping

__global__ void d3_mark_species_kernel(const int* __restrict__ numbers, int N, int nz, int* __restrict__ present) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int z = numbers[i];
  if (z > 0 && z < nz) present[z] = 1;  // benign race: every writer stores 1
}
// one block: compact ids of the species present, then the dense [S][S][25] table of {c6, cn_ref_i, cn_ref_j^T}
//
// Factorised form.  In Grimme's tables the reference CN of atom i at reference point (a, b) is a property of (Z_i, a) alone, and
// the set of populated points is a rectangle (a < n_ref(Z_i), b < n_ref(Z_j)).  When the tables of the species present have
// that structure (checked here, on the device, entry by entry and bit by bit) the Gaussian weight splits,
//   exp(k3 [(CN_i - c_i(a))^2 + (CN_j - c_j(b))^2] - max) = u_a(i) v_b(j),
// so a pair needs 5 exponentials instead of 25 and u_a is wave-uniform.  `ftab` [S][S][5][8] holds c6, `fcr` [S][8] holds
// {c(0..4), validity bits}.  Tables without that structure keep the general 25-term path.
__global__ void d3_compact_species_kernel(const int* __restrict__ present, const float* __restrict__ c6ab, const float* __restrict__ cnref, int nz,
                                          int* __restrict__ smap, D3Species* __restrict__ info, float4* __restrict__ ctab,
                                          float* __restrict__ ftab, float* __restrict__ fcr, float k3, const float* __restrict__ r4r2,
                                          float a1, float a2, const float* __restrict__ rcov, float* __restrict__ crc) {
  __shared__ int zlist[D3_SMAX];
  __shared__ int count;
  __shared__ int fact_ok;
  if (threadIdx.x == 0) {
    fact_ok = 1;
    int S = 0;
    for (int z = 0; z < nz; ++z) {
      if (z > 0 && present[z]) { smap[z] = S < D3_SMAX ? S : -1; if (S < D3_SMAX) zlist[S] = z; ++S; }
      else smap[z] = -1;
    }
    count = S;
    info->S = S;
  }
  __syncthreads();
  const int S = count;
  if (S > D3_SMAX) { if (threadIdx.x == 0) info->factorized = 0; return; }
  if (threadIdx.x < 16) crc[threadIdx.x] = (int)threadIdx.x < S ? rcov[zlist[threadIdx.x]] : -1.0f;  // radii by compact id (D3CRec); the padding code's is negative
  for (int k = threadIdx.x; k < S * S * 25; k += blockDim.x) {
    const int pq = k % 25, p = pq / 5, q = pq % 5, sj = (k / 25) % S, si = k / (25 * S);
    const int zi = zlist[si], zj = zlist[sj];
    const size_t a = ((size_t)zi * nz + zj) * 25 + pq, b = (((size_t)zj * nz + zi) * 5 + q) * 5 + p;
    ctab[k] = make_float4(c6ab[a], cnref[a], cnref[b], 0.0f);
    ftab[((size_t)si * S + sj) * D3_FROW + p * 8 + q] = c6ab[a];
    const size_t dii = ((size_t)zi * nz + zi) * 25 + p * 6, djj = ((size_t)zj * nz + zj) * 25 + q * 6;  // (p,p) / (q,q) entries
    const bool valid = c6ab[a] != 0.0f, rect = (c6ab[dii] != 0.0f) && (c6ab[djj] != 0.0f);
    const bool same = __float_as_int(cnref[a]) == __float_as_int(cnref[dii]) && __float_as_int(cnref[b]) == __float_as_int(cnref[djj]);
    if (valid != rect || (valid && !same)) fact_ok = 0;  // benign race: every writer stores 0
  }
  for (int k = threadIdx.x; k < S * S * 5; k += blockDim.x) {  // zero the 3 padding floats of each row
    float* row = ftab + (size_t)(k / 5) * D3_FROW + (k % 5) * 8;
    row[5] = row[6] = row[7] = 0.0f;
  }
  for (int k = threadIdx.x; k < S * S; k += blockDim.x) {  // BJ damping constants of the species pair (`_bj_damping`, dftd3.py:648-687)
    const float q = 3.0f * r4r2[zlist[k / S]] * r4r2[zlist[k % S]];
    const float r0 = a1 * sqrtf(q) + a2, r02 = r0 * r0, r04 = r02 * r02;
    float* d = ftab + (size_t)k * D3_FROW + 40;
    d[0] = q; d[1] = r04 * r02; d[2] = r04 * r04; d[3] = 0.0f;
  }
  for (int si = threadIdx.x; si < S; si += blockDim.x) {
    const int zi = zlist[si];
    int bits = 0;
    for (int p = 0; p < 5; ++p) {
      const size_t dii = ((size_t)zi * nz + zi) * 25 + p * 6;
      const bool has = c6ab[dii] != 0.0f;
      fcr[si * 8 + p] = has ? cnref[dii] : 1.0e30f;  // D3_NOREF: masks the point in d3_c6_fact without bit tests
      bits |= has << p;
    }
    fcr[si * 8 + 5] = __int_as_float(bits);
    fcr[si * 8 + 6] = fcr[si * 8 + 7] = 0.0f;
  }
  __syncthreads();
  if (threadIdx.x == 0) info->factorized = fact_ok && k3 < 0.0f;  // the -inf masking of missing points needs k3 < 0
}
// (Round 6 tried both as ONE launch -- every block marks, takes a ticket, the last one builds the tables -- and measured 58 us instead of
// 4.6 + 9.6: the device-scope fence in front of each ticket makes a block write back its XCD's L2, which at that point holds the tail of the
// 5 GB neighbour list the search has just written.  Cross-block hand-offs inside a kernel are not cheap on an 8-L2 part; a kernel boundary is.)

// `_c6ab_interpolate` (dftd3.py:427-547) on 25 packed terms {c6, cn_ref_i, cn_ref_j}.  Branch-free: the reference's
// `continue`s (c6 == 0, exp_arg - max < -12) become selects, so every lane runs the same 25 + 25 steps with no
// load -> branch -> load dependency; exponent arguments are kept in registers between the max pass and the sum pass.
__device__ __forceinline__ void d3_c6(float cn_i, float cn_j, const float4* __restrict__ t25, float k3, float& c6, float& dci) {
  // pass A: exponent arguments of all 25 reference points (kept in registers) and their maximum.  Empty reference points
  // (c6 == 0, which the reference skips in both of its loops) get -inf: they can neither set the maximum nor survive the
  // exp_arg - max >= -12 test below, so pass B needs no table access at all for terms it drops.
  float a[25];
  float mx = -1e20f, d0 = 0.0f;  // d0: CN_i - cn_ref_i of the dominant term (see D3Half: the d's may be taken relative to any constant)
#pragma unroll
  for (int c = 0; c < 5; ++c) {  // 5 terms per chunk: bounds the registers the scheduler may spend on hoisted table reads
    float4 v[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = t25[5 * c + k];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float di = cn_i - v[k].y, dj = cn_j - v[k].z;
      const float sq = di * di + dj * dj;
      const float at = (v[k].x != 0.0f) ? k3 * sq : -INFINITY;
      a[5 * c + k] = at;
      d0 = at > mx ? di : d0;
      mx = fmaxf(mx, at);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // pass B: Gaussian weights of the surviving terms.  Wave-uniform skip: a term no lane keeps costs a compare and a
  // scalar branch (typically only a few reference points are within e^-12 of the dominant one); lanes that do not keep a
  // term add exact zeros, which is what the reference's `continue` amounts to.
  float w = 0.0f, z = 0.0f, wdi = 0.0f, zdi = 0.0f;
#pragma unroll
  for (int t = 0; t < 25; ++t) {
    const float arg = a[t] - mx;
    const bool keep = !(arg < -12.0f);
    if (__builtin_amdgcn_ballot_w64(keep) == 0) continue;
    const float2 v = *reinterpret_cast<const float2*>(&t25[t]);  // {c6, cn_ref_i}
    const float di = (cn_i - v.y) - d0;
    const float L = keep ? d3_exp_neg(arg) : 0.0f;
    const float cL = v.x * L;
    w += L;
    z += cL;
    wdi += L * di;
    zdi += cL * di;
  }
  if (w > 1e-12f) {
    const float wi = D3_RCP(w);
    c6 = z * wi;
    const float si = zdi - c6 * wdi;
    dci = ((2.0f * k3) * wi) * si;
  } else {
    c6 = 0.0f; dci = 0.0f;
  }
}

// Factorised `_c6ab_interpolate`: same sums, same thresholds (points with c6 == 0 never count; terms more than e^-12 below the
// dominant one are dropped; w <= 1e-12 -> 0), with the exponent argument split as A_a(i) + B_b(j).  `Ap`, `u`, `di` are the
// wave-uniform per-atom halves (A_a - max A, exp of it, CN_i - c_i(a)); rows whose u is 0 for the whole wave are skipped.
// `di` is stored RELATIVE to the dominant row (the one with A'_a = 0): dC6/dCN_i = 2 k3/W sum_a u_a d_a (T_a - C6 S_a), and because
// sum_a u_a (T_a - C6 S_a) = Z - C6 W = 0 any constant may be subtracted from the d_a.  With the dominant row's own d the largest term
// drops out exactly; with the raw d_a = CN_i - c_i(a) the difference Z_d - C6 W_d cancels to ~|d| eps C6, which an approximate
// (biased) reciprocal in C6 = Z/W turns into a coherent error over all pairs of the atom: 3e-4 on forces of 0.6 Ha/Bohr when CN_i
// lies far outside the reference range (|d| ~ 40-80, dense test systems; tests/test_sweep_gpu.py).
struct D3Half { float Ap[5], u[5], di[5], vcut[5]; };  // vcut[a] = exp(-12 - A'_a): v_b survives row a iff v_b >= vcut[a]

__device__ __forceinline__ D3Half d3_half_i(float cn_i, const float* __restrict__ cr, float k3) {
  D3Half h;
  const int bits = __float_as_int(cr[5]);
  float A[5], mx = -INFINITY;
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    h.di[a] = cn_i - cr[a];
    A[a] = ((bits >> a) & 1) ? k3 * (h.di[a] * h.di[a]) : -INFINITY;
    mx = fmaxf(mx, A[a]);
  }
  float d0 = 0.0f;
#pragma unroll
  for (int a = 4; a >= 0; --a) d0 = (A[a] == mx) ? h.di[a] : d0;  // the first dominant row's CN_i - c_i(a)
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    h.Ap[a] = A[a] - mx;
    const bool keep = h.Ap[a] >= -12.0f;  // false for -inf and NaN (no populated point at all)
    h.u[a] = d3_uni(keep ? d3_exp_neg(keep ? h.Ap[a] : 0.0f) : 0.0f);
    h.vcut[a] = d3_uni(keep ? d3_exp_neg(-12.0f - h.Ap[a]) : INFINITY);
    h.Ap[a] = d3_uni(h.Ap[a]);
    h.di[a] = d3_uni(h.di[a] - d0);
  }
  return h;
}

typedef float d3_f2 __attribute__((ext_vector_type(2)));

// The j-side half of the factorised weights is a property of atom j alone: v_b(j) = exp(B_b - max_b B_b) with
// B_b = k3 (CN_j - c_j(b))^2.  It is evaluated ONCE per atom after the CN pass (5 exponentials per atom instead of per
// pair) and travels with the per-atom record the energy pass gathers anyway:
//   aw[2j] = {v_0..v_3},  aw[2j+1] = {v_4, r4r2[Z_j], bits(Z_j << 8 | species id), 0}      (one 32-byte line segment)
// Reference points that do not exist carry c_j(b) = D3_NOREF in `fcr`, so (CN_j - c)^2 overflows to +inf and k3 * inf = -inf
// masks them (the factorised path is only selected for k3 < 0); weights below e^-12 are stored as exact zeros.
#define D3_NOREF 1.0e30f
// fp32 positions: the record is {x, y, z, v_4 | v_0, v_1, v_2, v_3} with the 4-bit species id in the two spare top bits of v_0 and v_1
// (weights are in [0, 1]: sign and top exponent bit are always clear), so the energy pass gathers 32 bytes per neighbour instead
// of 16 (position) + 32 (weights): that pass is bound by its per-neighbour gathers once the exponentials are gone.
__device__ __forceinline__ void d3_weights_atom(int j, int place /* in the spatial order, or -1 */, float cnj, const float4* __restrict__ aaux, const D3Weights& W) {
  if (!W.sinfo->factorized || W.sinfo->S > D3_SMAX) return;
  const float4 a = aaux[j];
  const int code = __float_as_int(a.z), sj = code & 0xff;
  float v[5] = {0, 0, 0, 0, 0};
  if (sj < D3_SMAX) {  // real atom of a present species (padding atoms carry 0xff)
    float B[5], mx = -INFINITY;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      const float dj = cnj - W.fcr[sj * 8 + b];
      B[b] = W.k3 * (dj * dj);
      mx = fmaxf(mx, B[b]);
    }
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      const float Bp = B[b] - mx;
      const bool keep = Bp >= -12.0f;  // false for -inf and NaN (no populated point at all)
      v[b] = keep ? d3_exp_neg(keep ? Bp : 0.0f) : 0.0f;
    }
  }
  float4 r0, r1;
  if (W.apos_f32) {
    const float4 p = W.apos_f32[j];
    const int s4 = sj < D3_SMAX ? sj : 0;  // padding atoms: all weights are zero, any table row will do
    r0 = make_float4(p.x, p.y, p.z, v[4]);
    r1 = make_float4(__int_as_float(__float_as_int(v[0]) | ((s4 & 3) << 30)), __int_as_float(__float_as_int(v[1]) | ((s4 >> 2) << 30)), v[2], v[3]);
  } else {
    r0 = make_float4(v[0], v[1], v[2], v[3]);
    r1 = make_float4(v[4], a.y, a.z, 0.0f);
  }
  W.aw[2 * (size_t)j] = r0; W.aw[2 * (size_t)j + 1] = r1;
  if (place >= 0) { W.aw_s[2 * (size_t)place] = r0; W.aw_s[2 * (size_t)place + 1] = r1; }
}

// Per pair: the (a, b) contraction only.  Same sums and thresholds as `_c6ab_interpolate`: a term survives iff
// A'_a + B'_b >= -12, evaluated in weight space as v_b >= exp(-12 - A'_a) (wave-uniform per row).
__device__ __forceinline__ void d3_c6_fact(const D3Half& h, const float* v, const float* __restrict__ c6rows, float k3, float& c6, float& dci) {
  // plain fp32 FMAs: on gfx950 a v_pk_fma_f32 issues as two passes (the fp32 vector peak IS the unpacked rate), so packing the
  // {w, z} pairs bought nothing and cost ~50 v_mov per 64 pairs to line the operands up in register pairs
  float w = 0.0f, z = 0.0f, wdi = 0.0f, zdi = 0.0f;
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    if (__builtin_amdgcn_readfirstlane(__float_as_int(h.u[a])) == 0) continue;  // wave-uniform: u depends on atom i only
    const float4 r03 = *reinterpret_cast<const float4*>(c6rows + a * 8);
    const float r4 = c6rows[a * 8 + 4];
    const float cr[5] = {r03.x, r03.y, r03.z, r03.w, r4};
    float S = 0.0f, T = 0.0f;  // sum_b L, sum_b c6_ab L over the terms that survive the e^-12 cut
    if (__builtin_amdgcn_readfirstlane(__float_as_int(h.Ap[a])) == 0) {
      // the dominant row (A'_a = 0): the cut is B'_b >= -12, which the stored v_b already carry (smaller weights are stored as 0)
#pragma unroll
      for (int b = 0; b < 5; ++b) { S += v[b]; T = fmaf(v[b], cr[b], T); }
    } else {
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        const float L = (v[b] >= h.vcut[a]) ? v[b] : 0.0f;
        S += L; T = fmaf(L, cr[b], T);
      }
    }
    const float ud = h.u[a] * h.di[a];  // wave-uniform
    w = fmaf(h.u[a], S, w); z = fmaf(h.u[a], T, z);
    wdi = fmaf(ud, S, wdi); zdi = fmaf(ud, T, zdi);
  }
  if (w > 1e-12f) {
    const float wi = D3_RCP(w);
    c6 = z * wi;
    const float si = zdi - c6 * wdi;
    dci = ((2.0f * k3) * wi) * si;
  } else {
    c6 = 0.0f; dci = 0.0f;
  }
}

// float4 of LDS per wave of the energy pass: general form 16 x 25 {c6, cn_ref_i, cn_ref_j} (6.4 KB), factorised form 16 x 44 floats (2.8 KB)
__host__ __device__ constexpr int d3_wave_f4(int mode) { return mode == 1 ? D3_SMAX * 25 : mode == 2 ? D3_SMAX * D3_FROW / 4 : 1; }

// ---- pass 2: energy, direct force, dE/dCN ------------------------------------------------------------
// MODE 0: global [nz,nz,25] table (> 16 species); 1: general 25-term interpolation from the LDS-staged compact table;
// 2: factorised interpolation.  All three are launched; the two that do not match the device-side species info exit at once.
template <class T, bool CSR, int MODE, bool PK>
__device__ __forceinline__ void d3_energy_body(const T* __restrict__ pos, const int* __restrict__ numbers, int N, const int* __restrict__ idx,
                                                        const int* __restrict__ ush, const int* __restrict__ nptr, int M, int fill_value,
                                                        const T* __restrict__ cell, const int* __restrict__ batch_idx, D3Dev P,
                                                        const float* __restrict__ cn, int want_virial, const int* __restrict__ smap,
                                                        const D3Species* __restrict__ sinfo, const float4* __restrict__ ctab,
                                                        const float* __restrict__ ftab, const float* __restrict__ fcr,
                                                        const typename Vec4<T>::type* __restrict__ apos, const float4* __restrict__ aaux,
                                                        const float4* __restrict__ aw, float* __restrict__ dEdCN,
                                                        float* __restrict__ forces, float* __restrict__ e_atom, double* __restrict__ v_atom,
                                                        const unsigned* __restrict__ pk, const int* __restrict__ pk_flag,
                                                        const int* __restrict__ inv, const typename Vec4<T>::type* __restrict__ apos_s,
                                                        const float4* __restrict__ aaux_s, const float4* __restrict__ aw_s, float* __restrict__ dEdCN_s,
                                                        float4* __restrict__ lds_buf /* 4 * d3_wave_f4(MODE) float4 of the kernel's LDS */,
                                                        int vblock = -1 /* block index to work as (the fallback kernel's grid-stride loop); -1: blockIdx.x */) {
  constexpr bool LDS = MODE == 1;
  constexpr bool PACKED = MODE == 2 && sizeof(T) == 4;  // one 32-byte record per neighbour (see d3_weights_atom)
  constexpr int WAVE_F4 = d3_wave_f4(MODE);  // float4 per wave
  const int S = sinfo->S;
  const int want_mode = S > D3_SMAX ? 0 : (sinfo->factorized ? 2 : 1);
  if (want_mode != MODE) return;
  // (folding the fallback into this kernel as in the chain pass costs it 8 % -- 93 VGPRs and a branch per fetch -- so it stays separate)
  if (PK ? *pk_flag != 0 : (pk_flag != nullptr && *pk_flag == 0)) return;
  constexpr bool use_pk = PK;
  const int lane = threadIdx.x & (MI_WAVE - 1);
  const int k0 = __builtin_amdgcn_readfirstlane((vblock < 0 ? (int)blockIdx.x : vblock) * (blockDim.x / MI_WAVE) + threadIdx.x / MI_WAVE);
  if (k0 >= N) return;
  // spatial order (inv != NULL): rows are taken in that order; the packed list then holds PLACES in the order and the packed variants
  // gather from the ordered copies of the records, the plain variants keep the caller's indices and the records in atom order
  const int i = inv ? __builtin_amdgcn_readfirstlane(inv[k0]) : k0;
  if (PK && inv) { apos = apos_s; aaux = aaux_s; aw = aw_s; }
  const int self = (PK && inv) ? k0 : i;  // where this atom's own records are (masked-out lanes gather them)
  const int zi = numbers[i];
  if (zi == 0) return;
  const bool periodic = (cell != nullptr) && (ush != nullptr);
  T cm[9];
  if (periodic) { const T* c = cell + 9 * (size_t)(batch_idx ? batch_idx[i] : 0); for (int k = 0; k < 9; ++k) cm[k] = d3_uni(c[k]); }
  const T pix = d3_uni(pos[3 * (size_t)i]), piy = d3_uni(pos[3 * (size_t)i + 1]), piz = d3_uni(pos[3 * (size_t)i + 2]);
  const float cn_i = d3_uni(cn[i]), r4r2_i = d3_uni(P.r4r2[zi]);
  const float4* __restrict__ tab_i = P.tab + (size_t)zi * P.nz * 25;
  // stage this element's rows of the compact species table in the wave's private LDS slice
  const int code_i = (zi << 8) | (smap[zi] & 0xff);  // own species: a safe table row for masked-out lanes
  float4* my_tab = lds_buf + (MODE != 0 ? ((threadIdx.x / MI_WAVE) & 3) * WAVE_F4 : 0);
  if (LDS) {
    const float4* __restrict__ src = ctab + (size_t)smap[zi] * S * 25;
    for (int k = lane; k < S * 25; k += MI_WAVE) my_tab[k] = src[k];
  }
  float* my_f = reinterpret_cast<float*>(my_tab);  // MODE 2: [S][40] c6 rows of this element
  D3Half hi;
  if (MODE == 2) {
    const float4* __restrict__ src = reinterpret_cast<const float4*>(ftab + (size_t)smap[zi] * S * D3_FROW);
    for (int k = lane; k < S * (D3_FROW / 4); k += MI_WAVE) my_tab[k] = src[k];
    hi = d3_half_i(cn_i, fcr + smap[zi] * 8, P.k3);
  }
  // the table is staged as float4 and read back as float / float2 / float4: keep the compiler from moving those reads above
  // the staging stores (the LDS itself executes a wave's accesses in order)
  if (MODE != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long beg, end;
  d3_row<T, CSR>(i, M, nptr, beg, end);
  double Fx = 0, Fy = 0, Fz = 0, E = 0;
  // virial: f (x) r is symmetric (f is parallel to r) -> six components; fp32 lane partials (~40 terms each), fp64 across lanes
  float V[6] = {0, 0, 0, 0, 0, 0};  // xx xy xz yy yz zz
  double dacc = 0.0;
  const Int3* __restrict__ ush3 = reinterpret_cast<const Int3*>(ush);
  long long e = beg + lane;
  const unsigned jlim = d3_index_limit<CSR>(N, fill_value);
  static_assert(use_pk == PK, "the energy pass picks its list format at compile time");
  D3Lazy<PK> s0 = d3_fetch_lazy<PK>(idx, ush3, pk, e, end, periodic), s1 = d3_fetch_lazy<PK>(idx, ush3, pk, e + MI_WAVE, end, periodic);
  bool v0 = d3_lazy_in(s0) && ((unsigned)d3_lazy_j(s0) < jlim);
  using PosRec = typename Vec4<T>::type;
  auto pos_of = [&](int j) -> PosRec {
    if constexpr (PACKED) return aw[2 * (size_t)j];
    else return apos[j];
  };
  // per-neighbour data besides the position: MODE 0/1 {CN_j, r4r2_j, Z_j << 8 | species id} (aaux); MODE 2 the weight record
  // (fp64 positions: two float4 of aw; fp32 positions: the second half of the packed record)
  auto aux_of = [&](int j, float4& lo, float4& hi4) {
    if (PACKED) { lo = aw[2 * (size_t)j + 1]; hi4 = lo; }
    else if (MODE == 2) { lo = aw[2 * (size_t)j]; hi4 = aw[2 * (size_t)j + 1]; }
    else { lo = aaux[j]; hi4 = lo; }
  };
  PosRec p0 = pos_of(v0 ? d3_lazy_j(s0) : self);
  float4 a0, b0;
  aux_of(v0 ? d3_lazy_j(s0) : self, a0, b0);
  for (long long base = beg; base < end; base += MI_WAVE) {
    e += MI_WAVE;
    const D3Lazy<PK> s2 = d3_fetch_lazy<PK>(idx, ush3, pk, e + MI_WAVE, end, periodic);
    const bool v1 = d3_lazy_in(s1) && ((unsigned)d3_lazy_j(s1) < jlim);
    const PosRec p1 = pos_of(v1 ? d3_lazy_j(s1) : self);
    float4 a1, b1;
    aux_of(v1 ? d3_lazy_j(s1) : self, a1, b1);
    if (__any(v0)) {
      bool valid = v0 && (PACKED || !(p0.w < (T)0));  // padding atom (Z == 0); the packed record marks it by all-zero weights
      const PairGeom<T> g = d3_geom<T>(p0, pix, piy, piz, d3_lazy_sh(s0), cm, periodic);
      valid = valid && g.ok;
      float c6, dci, r4r2_j = 0.0f;
      int code;
      if (MODE == 2) {
        float vj[5];
        int sj;
        if (PACKED) {
          const int w0 = __float_as_int(a0.x), w1 = __float_as_int(a0.y);
          sj = ((w0 >> 30) & 3) | (((w1 >> 30) & 3) << 2);
          vj[0] = __int_as_float(w0 & 0x3fffffff); vj[1] = __int_as_float(w1 & 0x3fffffff); vj[2] = a0.z; vj[3] = a0.w; vj[4] = (float)p0.w;
        } else {
          sj = __float_as_int(b0.z) & 0xff;
          vj[0] = a0.x; vj[1] = a0.y; vj[2] = a0.z; vj[3] = a0.w; vj[4] = b0.x;
        }
        code = valid ? sj : (code_i & 0xff);
        d3_c6_fact(hi, vj, my_f + code * D3_FROW, P.k3, c6, dci);
      } else {
        code = valid ? __float_as_int(a0.z) : code_i;
        r4r2_j = a0.y;
        d3_c6(cn_i, a0.x, LDS ? my_tab + (code & 0xff) * 25 : tab_i + (size_t)(code >> 8) * 25, P.k3, c6, dci);
      }
      valid = valid && !(c6 < 1e-12f);
      // `_bj_damping` (dftd3.py:648-687)
      const float r = valid ? g.r : 1.0f;
      float q, r06, r08;
      if (MODE == 2) {  // species-pair constants staged with the c6 rows
        const float4 bj = *reinterpret_cast<const float4*>(my_f + code * D3_FROW + 40);
        q = bj.x; r06 = bj.y; r08 = bj.z;
      } else {
        q = 3.0f * r4r2_i * r4r2_j;
        const float r0 = P.a1 * D3_SQRT(q) + P.a2;
        const float r02 = r0 * r0, r04 = r02 * r02;
        r06 = r04 * r02; r08 = r04 * r04;
      }
      const float r2 = r * r, r4 = r2 * r2, r6 = r4 * r2, r8 = r4 * r4;
      const float i6 = D3_RCP(r6 + r06), i8 = D3_RCP(r8 + r08);
      const float damp = P.s6 * i6 + P.s8 * q * i8;
      // `_dispersion_energy_force` (dftd3.py:690-731)
      const float eij = -c6 * damp;
      const float r5 = r4 * r, r7 = r6 * r;
      const float d6 = -6.0f * P.s6 * r5 * i6 * i6;
      const float d8 = -8.0f * P.s8 * q * r7 * i8 * i8;
      const float dEdr = -c6 * (d6 + d8);
      float sw, dsw;
      d3_s5(r, P.s5_on, P.s5_off, P.inv_w, sw, dsw);
      const float esw = valid ? eij * sw : 0.0f;
      const float dEsw = valid ? sw * dEdr + eij * dsw : 0.0f;
      const float fx = dEsw * (g.rx * g.rinv), fy = dEsw * (g.ry * g.rinv), fz = dEsw * (g.rz * g.rinv);
      Fx += (double)fx; Fy += (double)fy; Fz += (double)fz;
      E += (double)esw;
      dacc += valid ? (double)(-damp * dci) : 0.0;
      if (want_virial) {
        V[0] += fx * g.rx; V[1] += fx * g.ry; V[2] += fx * g.rz;
        V[3] += fy * g.ry; V[4] += fy * g.rz; V[5] += fz * g.rz;
      }
    }
    s0 = s1; v0 = v1; p0 = p1; a0 = a1; b0 = b1; s1 = s2;
  }
  Fx = wave_sum(Fx); Fy = wave_sum(Fy); Fz = wave_sum(Fz); E = wave_sum(E);
  dacc = wave_sum(dacc);
  double V6[6];
  if (want_virial) {
#pragma unroll
    for (int k = 0; k < 6; ++k) V6[k] = wave_sum((double)V[k]);
  }
  if (lane == 0) {
    forces[3 * (size_t)i] = (float)Fx; forces[3 * (size_t)i + 1] = (float)Fy; forces[3 * (size_t)i + 2] = (float)Fz;
    dEdCN[i] = (float)dacc;
    if (inv) dEdCN_s[k0] = (float)dacc;
    e_atom[i] = 0.5f * (float)E;
    if constexpr (sizeof(T) == 4) {
      if (P.crec) {  // the chain pass's record of this atom (D3CRec): position + dE/dCN with the species id in its spare exponent bits
        const D3CRec f = d3_crec_format(S);
        if (f.on) {
          bool bad;
          P.crec[self] = make_float4(pix, piy, piz, d3_crec_encode((float)dacc, code_i & 0xff, f, &bad));
          if (bad) *P.cflag = 1;  // benign race: every writer stores 1
        }
      }
    }
  }
  if (want_virial && lane < 9) {
    const int r = lane / 3, c = lane - 3 * r, lo = r < c ? r : c, hi2 = r < c ? c : r;
    const int m = lo == 0 ? hi2 : (lo == 1 ? hi2 + 2 : 5);  // row-major (r, c) -> index in {xx, xy, xz, yy, yz, zz}
    double v = V6[0];
#pragma unroll
    for (int q = 1; q < 6; ++q) v = m == q ? V6[q] : v;
    v_atom[9 * (size_t)i + lane] = -0.5 * v;
  }
}


#define D3_ENERGY_PARAMS const T* __restrict__ pos, const int* __restrict__ numbers, int N, const int* __restrict__ idx, const int* __restrict__ ush, const int* __restrict__ nptr, int M, int fill_value, const T* __restrict__ cell, const int* __restrict__ batch_idx, D3Dev P, const float* __restrict__ cn, int want_virial, const int* __restrict__ smap, const D3Species* __restrict__ sinfo, const float4* __restrict__ ctab, const float* __restrict__ ftab, const float* __restrict__ fcr, const typename Vec4<T>::type* __restrict__ apos, const float4* __restrict__ aaux, const float4* __restrict__ aw, float* __restrict__ dEdCN, float* __restrict__ forces, float* __restrict__ e_atom, double* __restrict__ v_atom, const unsigned* __restrict__ pk, const int* __restrict__ pk_flag, const int* __restrict__ inv, const typename Vec4<T>::type* __restrict__ apos_s, const float4* __restrict__ aaux_s, const float4* __restrict__ aw_s, float* __restrict__ dEdCN_s
#define D3_ENERGY_ARGS pos, numbers, N, idx, ush, nptr, M, fill_value, cell, batch_idx, P, cn, want_virial, smap, sinfo, ctab, ftab, fcr, apos, aaux, aw, dEdCN, forces, e_atom, v_atom, pk, pk_flag, inv, apos_s, aaux_s, aw_s, dEdCN_s, lds_buf
template <class T, bool CSR, int MODE, bool PK>
__global__ __launch_bounds__(256) void d3_energy_kernel(D3_ENERGY_PARAMS) {
  __shared__ float4 lds_buf[4 * d3_wave_f4(MODE)];
  d3_energy_body<T, CSR, MODE, PK>(D3_ENERGY_ARGS);
}
// the fp32 factorised variant is latency-bound once its contraction is cheap (gathers from L2): it fits the register budget of
// D3_ENERGY_WAVES waves per SIMD without spilling, the other variants would spill under that cap
#ifndef D3_ENERGY_WAVES
#define D3_ENERGY_WAVES 8  // 63 VGPRs / 78 SGPRs, no scratch (same-box A/B: 5 waves 1.09 ms, 6 waves 0.94, 7 / 8 waves 0.92)
#endif
// (the plain-list variants need a few registers more for their decoded pipeline steps: 7 waves / SIMD without scratch instead of 8 with 5-7 spills)
template <class T, bool CSR, int MODE, bool PK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PK ? D3_ENERGY_WAVES : D3_ENERGY_WAVES - 1, PK ? D3_ENERGY_WAVES : D3_ENERGY_WAVES - 1))) void d3_energy_kernel_w5(D3_ENERGY_PARAMS) {
  __shared__ float4 lds_buf[4 * d3_wave_f4(MODE)];
  d3_energy_body<T, CSR, MODE, PK>(D3_ENERGY_ARGS);
}
// Everything but the common case in ONE launch behind the packed factorised variant: the packed general / global-table variants and the
// three plain-list variants (the bodies select themselves on the device-side species info and the packed-list flag; all but at most one
// return at once).  Round 3: this replaces three dead launches per call (~7 us each on the dependent D3 chain) by one.
// Round 5: a grid of at most D3_FALLBACK_GRID blocks that take the `n_blocks` block indices of the main variant's grid in a stride loop, and
// leave at once when the main variant (packed list, factorised interpolation) has done the work -- the common case, where this launch used
// to cost the dependent chain 14 - 16 us for tens of thousands of blocks that each ran five body prologues.
#define D3_FALLBACK_GRID 2048
template <class T, bool CSR>
__global__ __launch_bounds__(256) void d3_energy_fallback_kernel(D3_ENERGY_PARAMS, const unsigned* __restrict__ pk_packed, int n_blocks) {
  __shared__ float4 lds_buf[4 * d3_wave_f4(1)];  // one buffer for all bodies (the general form's is the largest): at most one of them works
  if (sinfo->S <= D3_SMAX && sinfo->factorized && *pk_flag == 0) return;  // d3_energy_body<T, CSR, 2, true> took this call
  for (int vb = blockIdx.x; vb < n_blocks; vb += gridDim.x) {
    d3_energy_body<T, CSR, 1, true>(pos, numbers, N, idx, ush, nptr, M, fill_value, cell, batch_idx, P, cn, want_virial, smap, sinfo, ctab, ftab, fcr, apos, aaux,
                                    aw, dEdCN, forces, e_atom, v_atom, pk_packed, pk_flag, inv, apos_s, aaux_s, aw_s, dEdCN_s, lds_buf, vb);
    d3_energy_body<T, CSR, 0, true>(pos, numbers, N, idx, ush, nptr, M, fill_value, cell, batch_idx, P, cn, want_virial, smap, sinfo, ctab, ftab, fcr, apos, aaux,
                                    aw, dEdCN, forces, e_atom, v_atom, pk_packed, pk_flag, inv, apos_s, aaux_s, aw_s, dEdCN_s, lds_buf, vb);
    d3_energy_body<T, CSR, 2, false>(D3_ENERGY_ARGS, vb);
    d3_energy_body<T, CSR, 1, false>(D3_ENERGY_ARGS, vb);
    d3_energy_body<T, CSR, 0, false>(D3_ENERGY_ARGS, vb);
  }
}
#undef D3_ENERGY_ARGS
#undef D3_ENERGY_PARAMS

// ---- pass 3: chain-rule force through the coordination numbers ---------------------------------------
#ifndef D3_CH_DS
#define D3_CH_DS 3  // chain pass: trips of list words in flight ahead of the evaluated one.  With two gathers per neighbour 3/1, 3/2, 4/2 were equal or 2 %
                    // slower than 2/1 (profiles/r05_ab_chain_pipeline.log); with the ONE gather of the chain records (round 6) a second trip of records in
                    // flight hides the gather behind the block's lock-step barrier: 2/1 0.674, 3/1 0.675, 3/2 0.596, 4/2 0.600, 4/3 0.619 (70 VGPRs, 7 waves),
                    // 5/3 0.63, 5/4 0.64 ms; records as far ahead as the list words (2/2, 3/3) 0.74 (profiles/r06_ab_chain_pipeline_one_gather.log)
#endif
#ifndef D3_CH_DG
#define D3_CH_DG 2  // ... and of gathered records (< D3_CH_DS)
#endif
template <class T, bool CSR, bool PK, bool CREC>
__device__ __forceinline__ void d3_chain_body(const T* __restrict__ pos, const int* __restrict__ numbers, int N, const int* __restrict__ idx,
                                              const int* __restrict__ ush, const int* __restrict__ nptr, int M, int fill_value,
                                              const T* __restrict__ cell, const int* __restrict__ batch_idx, const D3Dev& P,
                                              const typename Vec4<T>::type* __restrict__ apos, const float* __restrict__ dEdCN,
                                              int want_virial, float* __restrict__ forces, double* __restrict__ v_atom,
                                              const unsigned* __restrict__ pk, const bool use_pk,
                                              const int* __restrict__ inv, const typename Vec4<T>::type* __restrict__ apos_s,
                                              const float* __restrict__ dEdCN_s, const D3CRec& cf, const float* rc_lds) {
  const int lane = threadIdx.x & (MI_WAVE - 1);
  const int i0 = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / MI_WAVE) + threadIdx.x / MI_WAVE);
  const int k0 = i0 < N ? i0 : N - 1;
  const int i = inv ? __builtin_amdgcn_readfirstlane(inv[k0]) : k0;  // rows in the spatial order (see d3_sort_key_kernel)
  const bool ordered = use_pk && inv != nullptr;                      // the packed list holds places in that order: gather the ordered copies
  // CREC: ONE gather per neighbour, the chain record {x, y, z, dE/dCN_j | species id} the energy pass left (indexed like the ordered copies)
  const typename Vec4<T>::type* __restrict__ arec = CREC ? reinterpret_cast<const typename Vec4<T>::type*>(P.crec) : (ordered ? apos_s : apos);
  const float* __restrict__ drec = ordered ? dEdCN_s : dEdCN;
  const int self = (CREC ? inv != nullptr : ordered) ? k0 : i;
  const int zi = numbers[i];
  const bool live = i0 < N && zi != 0;  // idle waves still take part in the block's lock-step barriers
  const bool periodic = (cell != nullptr) && (ush != nullptr);
  T cm[9];
  if (periodic) { const T* c = cell + 9 * (size_t)(batch_idx ? batch_idx[i] : 0); for (int k = 0; k < 9; ++k) cm[k] = c[k]; }
  const T pix = pos[3 * (size_t)i], piy = pos[3 * (size_t)i + 1], piz = pos[3 * (size_t)i + 2];
  const float rci = P.rcov[zi], di = dEdCN[i];
  long long beg, end;
  d3_row<T, CSR>(i, M, nptr, beg, end);
  if (!live) end = beg;
  const int trips = d3_lockstep_trips<D3_CH_LS_WAVES>(beg, end);
  double Fx = 0, Fy = 0, Fz = 0;
  // virial: f (x) r is symmetric (f is parallel to r) -> six components, fp32 lane partials (~40 terms each) summed in fp64 across lanes,
  // exactly as in the energy pass.  Nine fp64 lane partials cost 27 instructions per pair (18 of them at the fp64 rate) against 6 FMAs
  // here, and this pass is VALU-bound (round 3: staging the gathered records in LDS instead changed its time by 4 %, tools/probe/probe_lds.hip)
  float V[6] = {0, 0, 0, 0, 0, 0};  // xx xy xz yy yz zz
  const Int3* __restrict__ ush3 = reinterpret_cast<const Int3*>(ush);
  long long e = beg + lane;
  const unsigned jlim = d3_index_limit<CSR>(N, fill_value);
  // software pipeline as in the CN pass: list words D3_CH_DS trips ahead, gathered records D3_CH_DG trips ahead of the one being evaluated
  D3Step s[D3_CH_DS + 1];
  bool v[D3_CH_DG + 1];
  typename Vec4<T>::type p[D3_CH_DG + 1];
  float d[D3_CH_DG + 1];
#pragma unroll
  for (int k = 0; k < D3_CH_DS; ++k) s[k] = d3_fetch_any<PK>(idx, ush3, pk, use_pk, e + (long long)k * MI_WAVE, end, periodic);
#pragma unroll
  for (int k = 0; k < D3_CH_DG; ++k) {
    v[k] = s[k].in && ((unsigned)s[k].j < jlim); p[k] = arec[v[k] ? s[k].j : self];
    if (!CREC) d[k] = drec[v[k] ? s[k].j : self];
  }
  for (int trip = 0; trip < trips; ++trip) {
    __syncthreads();  // lock-step (see d3_cn_kernel)
    s[D3_CH_DS] = d3_fetch_any<PK>(idx, ush3, pk, use_pk, e + (long long)D3_CH_DS * MI_WAVE, end, periodic);
    v[D3_CH_DG] = s[D3_CH_DG].in && ((unsigned)s[D3_CH_DG].j < jlim);
    p[D3_CH_DG] = arec[v[D3_CH_DG] ? s[D3_CH_DG].j : self];
    if (!CREC) d[D3_CH_DG] = drec[v[D3_CH_DG] ? s[D3_CH_DG].j : self];
    const D3Step s0 = s[0];
    const bool v0 = v[0];
    const auto p0 = p[0];
    if (__any(v0)) {
      float rcj, dsum;
      if constexpr (CREC) {
        rcj = rc_lds[d3_crec_id((float)p0.w, cf)];
        dsum = fmaf(d3_crec_value((float)p0.w, cf), cf.up, di);  // = di + dE/dCN_j, rounded once: the scaling by a power of two is exact
      } else {
        rcj = (float)p0.w;
        dsum = di + d[0];
      }
      bool valid = v0 && !(rcj < 0.0f);  // padding atom (Z == 0)
      const PairGeom<T> g = d3_geom<T>(p0, pix, piy, piz, s0.sh, cm, periodic);
      valid = valid && g.ok;
      float dcn;
      d3_cn_count(g.rinv, rci, rcj, P.k1, &dcn);
      const float dEdr = valid ? dsum * dcn : 0.0f;
      const float fx = dEdr * (g.rx * g.rinv), fy = dEdr * (g.ry * g.rinv), fz = dEdr * (g.rz * g.rinv);
      Fx += (double)fx; Fy += (double)fy; Fz += (double)fz;
      if (want_virial) {
        V[0] = fmaf(fx, g.rx, V[0]); V[1] = fmaf(fx, g.ry, V[1]); V[2] = fmaf(fx, g.rz, V[2]);
        V[3] = fmaf(fy, g.ry, V[3]); V[4] = fmaf(fy, g.rz, V[4]); V[5] = fmaf(fz, g.rz, V[5]);
      }
    }
#pragma unroll
    for (int k = 0; k < D3_CH_DS; ++k) s[k] = s[k + 1];
#pragma unroll
    for (int k = 0; k < D3_CH_DG; ++k) { v[k] = v[k + 1]; p[k] = p[k + 1]; if (!CREC) d[k] = d[k + 1]; }
    e += MI_WAVE;
  }
  Fx = wave_sum(Fx); Fy = wave_sum(Fy); Fz = wave_sum(Fz);
  double V6[6];
  if (want_virial) {
#pragma unroll
    for (int k = 0; k < 6; ++k) V6[k] = wave_sum((double)V[k]);
  }
  if (lane == 0 && live) {
    forces[3 * (size_t)i] = forces[3 * (size_t)i] + (float)Fx;
    forces[3 * (size_t)i + 1] = forces[3 * (size_t)i + 1] + (float)Fy;
    forces[3 * (size_t)i + 2] = forces[3 * (size_t)i + 2] + (float)Fz;
  }
  if (want_virial && lane < 9 && live) {
    const int r = lane / 3, c = lane - 3 * r, lo = r < c ? r : c, hi2 = r < c ? c : r;
    const int m = lo == 0 ? hi2 : (lo == 1 ? hi2 + 2 : 5);  // row-major (r, c) -> index in {xx, xy, xz, yy, yz, zz}
    double v = V6[0];
#pragma unroll
    for (int q = 1; q < 6; ++q) v = m == q ? V6[q] : v;
    v_atom[9 * (size_t)i + lane] += -0.5 * v;
  }
}
template <class T, bool CSR, bool PK>
__global__ __launch_bounds__(D3_CH_LS_WAVES * MI_WAVE) void d3_chain_kernel(const T* __restrict__ pos, const int* __restrict__ numbers, int N, const int* __restrict__ idx,
                                                       const int* __restrict__ ush, const int* __restrict__ nptr, int M, int fill_value,
                                                       const T* __restrict__ cell, const int* __restrict__ batch_idx, D3Dev P,
                                                       const typename Vec4<T>::type* __restrict__ apos, const float* __restrict__ dEdCN,
                                                       int want_virial, float* __restrict__ forces, double* __restrict__ v_atom,
                                                       const unsigned* __restrict__ pk, const int* __restrict__ pk_flag,
                                                       const int* __restrict__ inv, const typename Vec4<T>::type* __restrict__ apos_s,
                                                       const float* __restrict__ dEdCN_s, const D3Species* __restrict__ sinfo) {
  const bool use_pk = PK && *pk_flag == 0;
  D3CRec cf{};
  if constexpr (sizeof(T) == 4) {
    // chain records (D3CRec): block-uniform decision.  They are indexed like the list the energy pass walked: with a packed list that the
    // passes found unusable (the energy pass's fallback launch did the work) the two-gather walk below stays
    __shared__ float rc_lds[16];
    cf = d3_crec_format(sinfo->S);
    if (P.crec != nullptr && cf.on && *P.cflag == 0 && (!PK || use_pk)) {
      if (threadIdx.x < 16) rc_lds[threadIdx.x] = P.crc[threadIdx.x];
      __syncthreads();
      d3_chain_body<T, CSR, PK, true>(pos, numbers, N, idx, ush, nptr, M, fill_value, cell, batch_idx, P, apos, dEdCN, want_virial, forces, v_atom, pk, use_pk, inv,
                                      apos_s, dEdCN_s, cf, rc_lds);
      return;
    }
  }
  d3_chain_body<T, CSR, PK, false>(pos, numbers, N, idx, ush, nptr, M, fill_value, cell, batch_idx, P, apos, dEdCN, want_virial, forces, v_atom, pk, use_pk, inv,
                                   apos_s, dEdCN_s, cf, nullptr);
}

// per-system reduction of per-atom energies / virials.  A few hundred waves each own a contiguous slab of atoms, keep a
// running (system, sum) pair and touch global memory with ONE atomic per system change (batches are contiguous per
// system, so that is ~1 per wave): 10 values x 256 waves instead of one atomic per 64 atoms on the same addresses.
// (round 3: the partial sums of a system are spread over D3_REDUCE_SLOTS slot rows -- 2560 fp64 atomics on ten addresses were 27 us of
// serialisation on the single-system headline box; d3_finish_kernel folds the slots)
#define D3_REDUCE_WAVES 256
#define D3_REDUCE_SLOTS 16
__global__ __launch_bounds__(256) void d3_reduce_kernel(const float* __restrict__ e_atom, const double* __restrict__ v_atom,
                                                        const int* __restrict__ batch_idx, int N, int want_virial,
                                                        double* __restrict__ sums /*[B][D3_REDUCE_SLOTS][10], zeroed*/) {
  const int lane = threadIdx.x & (MI_WAVE - 1);
  const int wave = blockIdx.x * (blockDim.x / MI_WAVE) + threadIdx.x / MI_WAVE;
  const int slot = wave & (D3_REDUCE_SLOTS - 1);
  const int chunks = (N + MI_WAVE - 1) / MI_WAVE;
  const int per = (chunks + D3_REDUCE_WAVES - 1) / D3_REDUCE_WAVES;
  const int c0 = wave * per, c1 = (c0 + per < chunks) ? c0 + per : chunks;
  int cur = -1;
  double acc[10];  // fp64 partial sums (the reference accumulates energies in fp64 and rounds once, dftd3.py:1031)
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = 0.0;
  auto flush = [&]() {
    if (cur < 0) return;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      if (k > 0 && !want_virial) break;
      const double v = wave_sum(acc[k]);
      if (lane == 0) atomicAdd(&sums[10 * ((size_t)cur * D3_REDUCE_SLOTS + slot) + k], v);
      acc[k] = 0.0;
    }
  };
  for (int c = c0; c < c1; ++c) {
    const int i = c * MI_WAVE + lane;
    const bool in = i < N;
    const int s = in ? (batch_idx ? batch_idx[i] : 0) : -1;
    const int s0 = __shfl(s, 0, MI_WAVE);
    if (__all(!in || s == s0)) {
      if (s0 != cur) { flush(); cur = s0; }
      if (in) {
        acc[0] += (double)e_atom[i];
        if (want_virial) for (int k = 0; k < 9; ++k) acc[k + 1] += v_atom[9 * (size_t)i + k];
      }
    } else if (in) {  // a chunk straddling systems: per-lane atomics
      atomicAdd(&sums[10 * ((size_t)s * D3_REDUCE_SLOTS + slot)], (double)e_atom[i]);
      if (want_virial) for (int k = 0; k < 9; ++k) atomicAdd(&sums[10 * ((size_t)s * D3_REDUCE_SLOTS + slot) + 1 + k], v_atom[9 * (size_t)i + k]);
    }
  }
  flush();
}
// (a separate launch: folding the slots in the reduce kernel's last block needs a device-scope fence per block, which costs more than this
// launch does -- see d3_mark_species_kernel)
__global__ void d3_finish_kernel(const double* __restrict__ sums, int B, int want_virial, float* __restrict__ energy, float* __restrict__ virial) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 10 * B) return;
  const int s = t / 10, k = t - 10 * s;
  double v = 0.0;
#pragma unroll
  for (int q = 0; q < D3_REDUCE_SLOTS; ++q) v += sums[10 * ((size_t)s * D3_REDUCE_SLOTS + q) + k];  // fixed order: deterministic given the slot sums
  if (k == 0) energy[s] = (float)v;
  else if (want_virial) virial[9 * (size_t)s + (k - 1)] = (float)v;
}

inline long long d3_sort_cap(int N, int B) { return 4ll * N + 8ll * (B > 0 ? B : 1); }
struct D3Layout { size_t dEdCN, e_atom, v_atom, guard, sums, tab, present, smap, sinfo, ctab, ftab, fcr, apos, aaux, aw, inv, skeys, sgrid, sbins, apos_s, acn, aaux_s, aw_s, dEdCN_s, crec, crc, total; };
D3Layout d3_layout(int N, int nz, int dtype, int B) {
  D3Layout L;
  size_t o = 0;
  auto take = [&](size_t b) { size_t at = o; o += mi_align(b); return at; };
  L.dEdCN = take(sizeof(float) * (size_t)N);
  L.e_atom = take(sizeof(float) * (size_t)N);
  L.v_atom = take(sizeof(double) * 9 * (size_t)N);  // fp64: the direct and the chain-rule part of an atom's virial can cancel (dense systems)
  // `guard`, `sums` and `present` sit next to each other: all start a call as zeros and are cleared by ONE memset (round 5: one launch less
  // on the dependent chain of small D3 kernels).  guard = MI_CN_SLOTS fingerprint words + the "companion unusable" flag (D3Guard)
  L.guard = take(sizeof(unsigned long long) * MI_CN_SLOTS + 256);
  L.sums = take(sizeof(double) * 10 * D3_REDUCE_SLOTS * (size_t)(B > 0 ? B : 1));
  L.present = take(sizeof(int) * ((size_t)nz + 2));  // + 2: the atom-order probe {far-apart consecutive pairs, largest pair distance (bits)}, cleared with the table
  L.tab = take(sizeof(float4) * (size_t)nz * nz * 25);
  L.smap = take(sizeof(int) * (size_t)nz);
  L.sinfo = take(sizeof(D3Species));
  L.ctab = take(sizeof(float4) * D3_SMAX * D3_SMAX * 25);
  L.ftab = take(sizeof(float) * D3_SMAX * D3_SMAX * D3_FROW);
  L.fcr = take(sizeof(float) * D3_SMAX * 8);
  L.apos = take((dtype == MI_F32 ? 16 : 32) * (size_t)N);
  L.aaux = take(sizeof(float4) * (size_t)N);
  L.aw = take(sizeof(float4) * 2 * (size_t)N);
  // spatial order: permutation, bin keys, counting-sort counters (<= N/16 bins), ordered copies of the gathered records
  L.inv = take(sizeof(int) * (size_t)N);
  L.skeys = take(sizeof(int) * (size_t)N);
  L.sgrid = take(sizeof(D3Grid) * (size_t)(B > 0 ? B : 1));
  L.sbins = take(sizeof(int) * bs_scratch_ints(d3_sort_cap(N, B) + 1));  // the search's own bound on its cells: 4 N + 8 B
  L.apos_s = take((dtype == MI_F32 ? 16 : 32) * (size_t)N);
  L.acn = take((dtype == MI_F32 ? 16 : 32) * (size_t)N);
  L.aaux_s = take(sizeof(float4) * (size_t)N);
  L.aw_s = take(sizeof(float4) * 2 * (size_t)N);
  L.dEdCN_s = take(sizeof(float) * (size_t)N);
  L.crec = take(sizeof(float4) * (size_t)N);  // chain records (D3CRec; fp32 positions)
  L.crc = take(sizeof(float) * 16);
  L.total = o;
  return L;
}

// ---- is the caller's atom order spatially coherent?  (host side of the spatial order) ------------------------------------------------
// Ordered input (lattice order, molecules, a previous sort) is best left alone: the records in atom order ARE coherent, and sorting them
// on a grid that is not the neighbour search's own makes the 32-byte gathers of the energy pass worse (0.92 -> 1.18 ms on the headline
// box), whereas a randomly ordered box gains 8.2 -> 6.2 ms per step.  The order of an MD system does not change from step to step, so
// the decision is taken from a measurement of an EARLIER call, without any synchronisation: every D3_ORDER_PROBE_EVERY-th call the key
// kernel counts the consecutive atoms that sit in far-apart bins and the count is copied to pinned host memory asynchronously; a later
// call reads whatever has arrived (first calls: "coherent").  Both paths give bit-identical results, so a switch is invisible.
// This is the ONE piece of state the library keeps between calls (include/nvalchemiops_hip.h says so): per device, a pinned ring of
// D3_ORDER_SLOTS records and the (n_atoms, n_systems) it describes.  A change of (n_atoms, n_systems) starts a new GENERATION with a
// fresh ring slot, so an asynchronous copy still in flight for the previous system lands in ITS slot and can never be taken for a
// measurement of the new one.  Two different systems of equal size alternating on one device share a record: they then share
// one heuristic, never a result.  Under HIP-graph capture nothing is probed or published and the decision taken at capture time is frozen
// into the graph.  The ring (64 bytes of pinned host memory per device) lives until the process exits.
#define D3_ORDER_PROBE_EVERY 64
#define D3_ORDER_SLOTS 8
struct D3OrderState { int* h_count = nullptr; /* pinned [D3_ORDER_SLOTS][2] = {far-apart pairs, largest pair distance bits} */ int n_atoms = -1, n_systems = -1; long long calls = 0; unsigned gen = 0; };
static D3OrderState g_d3_order[16];
static std::mutex g_d3_order_mu;
static bool d3_order_decide(int N, int B, hipStream_t st, bool* probe, float* rc_est, int** publish_to) {
  *probe = false;
  *publish_to = nullptr;
  *rc_est = 0.0f;
  const char* force = getenv("NVALCHEMIOPS_D3_SORT");  // tuning aid: 0 = never, 1 = always
  if (force && atoi(force) == 0) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
  std::lock_guard<std::mutex> lock(g_d3_order_mu);
  D3OrderState& S = g_d3_order[dev];
  if (!S.h_count) {
    if (capturing || hipHostMalloc(reinterpret_cast<void**>(&S.h_count), 2 * D3_ORDER_SLOTS * sizeof(int), hipHostMallocDefault) != hipSuccess) { S.h_count = nullptr; return force != nullptr; }
    for (int k = 0; k < 2 * D3_ORDER_SLOTS; ++k) S.h_count[k] = 0;
  }
  if (S.n_atoms != N || S.n_systems != B) {  // another system: new generation, fresh slot (late copies of the old one keep their slot)
    S.n_atoms = N; S.n_systems = B; S.calls = 0; ++S.gen;
    S.h_count[2 * (S.gen % D3_ORDER_SLOTS)] = 0; S.h_count[2 * (S.gen % D3_ORDER_SLOTS) + 1] = 0;
  }
  volatile int* h = S.h_count + 2 * (S.gen % D3_ORDER_SLOTS);
  *publish_to = S.h_count + 2 * (S.gen % D3_ORDER_SLOTS);
  *probe = !capturing && (S.calls < 4 || (S.calls % D3_ORDER_PROBE_EVERY) == 0);  // the first calls, then now and then
  ++S.calls;
  const int bits = h[1];
  *rc_est = __builtin_bit_cast(float, bits);
  if (force) return true;
  return 4ll * h[0] > (long long)N;  // more than a quarter of the consecutive pairs are far apart
}
static void d3_order_publish(const int* d_count, int* h_slot, hipStream_t st) {
  if (h_slot) (void)hipMemcpyAsync(h_slot, d_count, 2 * sizeof(int), hipMemcpyDeviceToHost, st);
}

template <class T, bool CSR>
int d3_impl(const T* pos, const int* numbers, int N, const int* idx, const int* ush, const int* nptr, int M, int fill_value, const T* cell,
            const int* batch_idx, int B, const mi_d3_params* hp, int want_virial, float* energy, float* forces, float* cn, float* virial,
            char* ws, const D3Layout& L, unsigned* pk, long long n_entries, const unsigned* pre, const void* cn_block, int verify_stride, int verify_phase,
            hipStream_t st) {
  // `pk` (optional, [N*M] words + one flag word in front): packed copy of a periodic padded list, see d3_fetch_pk
  int* pk_flag = nullptr;
  if (pk) { pk_flag = reinterpret_cast<int*>(pk); pk += 64; }
  int* ws_pk_flag = pk_flag;  // cleared below, unless the search's companion makes the workspace copy unnecessary
  // `pre` (optional, same layout, read-only): the companion the neighbour search wrote with the matrix (mi_nl_neighbors_packed).  The CN
  // pass then streams 4 B/slot instead of 16 and -- unless the spatial order is on, whose packed list holds places, not indices -- writes
  // nothing; the energy and chain passes read `pre` directly.
  const int* pre_flag = pre ? reinterpret_cast<const int*>(pre) : nullptr;
  const unsigned* pre_words = pre ? pre + 64 : nullptr;
  // device-side guards of the search's by-products (D3Guard): fingerprint slots + the flag every pass consults instead of the companion's
  // own header flag (raised by the pack kernel when the header says "unusable" or the sampled rows do not match the caller's arrays)
  unsigned long long* gslots = reinterpret_cast<unsigned long long*>(ws + L.guard);
  int* gflag = reinterpret_cast<int*>(ws + L.guard + sizeof(unsigned long long) * MI_CN_SLOTS);
  float* dEdCN = reinterpret_cast<float*>(ws + L.dEdCN);
  float* e_atom = reinterpret_cast<float*>(ws + L.e_atom);
  double* v_atom = reinterpret_cast<double*>(ws + L.v_atom);
  float4* tab = reinterpret_cast<float4*>(ws + L.tab);
  int* present = reinterpret_cast<int*>(ws + L.present);
  int* smap = reinterpret_cast<int*>(ws + L.smap);
  D3Species* sinfo = reinterpret_cast<D3Species*>(ws + L.sinfo);
  float4* ctab = reinterpret_cast<float4*>(ws + L.ctab);
  float* ftab = reinterpret_cast<float*>(ws + L.ftab);
  float* fcr = reinterpret_cast<float*>(ws + L.fcr);
  auto* apos = reinterpret_cast<typename Vec4<T>::type*>(ws + L.apos);
  float4* aaux = reinterpret_cast<float4*>(ws + L.aaux);
  float4* aw = reinterpret_cast<float4*>(ws + L.aw);
  // spatial order (see d3_sort_key_kernel): only with the packed list (periodic, shifts given), enough atoms for it to matter, codes that
  // fit (place < 2^24, Z < 128) -- and only when the caller's atom order is NOT spatially coherent already (D3OrderState)
  const bool sortable = pk != nullptr && cell != nullptr && N >= D3_SORT_MIN_ATOMS && N < (1 << 24) && hp->nz <= 128;
  int* order_probe = present + hp->nz;  // cleared with `present` below
  bool probe = false;
  float rc_est = 0.0f;
  int* order_slot = nullptr;  // pinned ring slot of this system's generation (null: nothing to publish)
  const bool sorted = sortable && d3_order_decide(N, B, st, &probe, &rc_est, &order_slot);
  if (pre && !sorted) {  // the later passes take the search's companion as their packed list (its flag says whether it is usable)
    pk = const_cast<unsigned*>(pre_words);
    pk_flag = gflag;  // read-only from here on: the CN launches below get no output list
  } else if (ws_pk_flag) {
    MI_HIP_CHECK(hipMemsetAsync(ws_pk_flag, 0, sizeof(int), st));
  }
  int* inv = sorted ? reinterpret_cast<int*>(ws + L.inv) : nullptr;
  auto* apos_s = reinterpret_cast<typename Vec4<T>::type*>(ws + L.apos_s);
  auto* acn = reinterpret_cast<typename Vec4<T>::type*>(ws + L.acn);
  float4* aaux_s = reinterpret_cast<float4*>(ws + L.aaux_s);
  float4* aw_s = reinterpret_cast<float4*>(ws + L.aw_s);
  float* dEdCN_s = reinterpret_cast<float*>(ws + L.dEdCN_s);
  D3Dev P;
  P.rcov = hp->rcov; P.r4r2 = hp->r4r2; P.tab = tab; P.nz = hp->nz;
  P.a1 = hp->a1; P.a2 = hp->a2; P.s6 = hp->s6; P.s8 = hp->s8; P.k1 = hp->k1; P.k3 = hp->k3; P.s5_on = hp->s5_on; P.s5_off = hp->s5_off;
  // inv_w in double on the host, then cast (dftd3.py:1983-1986)
  P.inv_w = (hp->s5_off > hp->s5_on) ? (float)(1.0 / ((double)hp->s5_off - (double)hp->s5_on)) : 0.0f;
  const char* crec_env = getenv("NVALCHEMIOPS_D3_CHAIN_RECORDS");  // "0": the two-gather chain walk (A/B, tests); read per call like NVALCHEMIOPS_D3_SORT
  const bool crec_off = crec_env && crec_env[0] == '0';
  P.crec = (sizeof(T) == 4 && !crec_off) ? reinterpret_cast<float4*>(ws + L.crec) : nullptr;
  P.crc = reinterpret_cast<const float*>(ws + L.crc);
  P.cflag = gflag + 4;  // inside the guard block: cleared with it by the memset below
  // outputs are zeroed like the reference wrapper does (dftd3.py:1933-1936; atoms with Z == 0 keep zeros): the per-atom arrays by
  // the pack kernel below, energy / virial are written for every system by the finish kernel
  // (one memset: the per-system reduction slots `sums`, only touched by the reduce kernel at the very end, lie directly in front of `present`)
  // (in whole 16-byte words: a length that is not a multiple of 16 is split into two fill kernels by the runtime, ~5 us more on the dependent
  // chain; the round-up stays inside the 256-byte padding of `present`)
  MI_HIP_CHECK(hipMemsetAsync(ws + L.guard, 0, ((L.present - L.guard) + sizeof(int) * ((size_t)hp->nz + 2) + 15) / 16 * 16, st));
  D3Grid* sgrid = reinterpret_cast<D3Grid*>(ws + L.sgrid);
  if (sorted || (sortable && probe)) {
    d3_sort_setup_kernel<T><<<1, 256, 0, st>>>(cell, B, N, rc_est, sgrid, d3_sort_cap(N, B));
    MI_LAUNCH_CHECK();
  }
  if (sorted) {
    int* skeys = reinterpret_cast<int*>(ws + L.skeys);
    const BsScratch bins = bs_carve(reinterpret_cast<int*>(ws + L.sbins), d3_sort_cap(N, B) + 1);
    MI_HIP_CHECK(bs_clear(bins, st));
    d3_sort_key_kernel<T><<<mi_blocks(N, 256), 256, 0, st>>>(pos, cell, batch_idx, N, sgrid, skeys, bins.count, order_probe);
    MI_LAUNCH_CHECK();
    MI_HIP_CHECK(bs_sort(bins, skeys, N, nullptr, inv, nullptr, st));
  } else if (sortable && probe) {
    d3_sort_key_kernel<T><<<mi_blocks(N, 256), 256, 0, st>>>(pos, cell, batch_idx, N, sgrid, nullptr, nullptr, order_probe);
    MI_LAUNCH_CHECK();
  }
  const bool publish = sortable && probe;  // (after the CN pass, which adds the largest pair distance)
  d3_mark_species_kernel<<<mi_blocks(N, 256), 256, 0, st>>>(numbers, N, hp->nz, present);
  MI_LAUNCH_CHECK();
  d3_compact_species_kernel<<<1, 256, 0, st>>>(present, hp->c6ab, hp->cn_ref, hp->nz, smap, sinfo, ctab, ftab, fcr, hp->k3, hp->r4r2, hp->a1, hp->a2, hp->rcov,
                                               reinterpret_cast<float*>(ws + L.crc));
  MI_LAUNCH_CHECK();
  const long long nt = (long long)hp->nz * hp->nz * 25;
  // the search's coordination numbers are taken only in the caller's atom order (the spatial order's CN pass writes its place-coded list anyway)
  const bool use_cn = !CSR && cn_block != nullptr && pre != nullptr && !sorted;
  D3Guard G{};
  G.atom_blocks = mi_blocks(use_cn && 9 * B > N ? 9 * B : N, 256);
  int verify_blocks = 0;
  if (use_cn) { G.slots = gslots; G.K = mi_cn_scale(hp->k1); G.cell = cell; G.n_cell = 9 * B; G.batch_idx = batch_idx; }
  if (pre) {
    G.nm = idx; G.nsh = ush; G.words = pre_words; G.hdr_flag = pre_flag; G.flag = gflag; G.M = M; G.stride = verify_stride; G.phase = verify_phase;
    // one wave per sampled row; one block at least: it also forwards the header flag
    verify_blocks = verify_stride > 0 ? mi_blocks(((long long)N + verify_stride - 1) / verify_stride, 4) : 1;
  }
  const D3Tables TB{hp->c6ab, hp->cn_ref, hp->nz, sinfo, tab, G.atom_blocks + verify_blocks};  // (its blocks leave at once unless > 16 species are present)
  d3_pack_atoms_kernel<T><<<G.atom_blocks + verify_blocks + mi_blocks(nt, 256), 256, 0, st>>>(pos, numbers, N, hp->rcov, hp->r4r2, smap, hp->nz, apos, aaux, forces, cn,
                                                                                              dEdCN, e_atom, want_virial ? v_atom : nullptr, inv, apos_s, aaux_s, acn, G, TB, P.crec);
  MI_LAUNCH_CHECK();
  const int blocks = mi_blocks(N, 4);
  const D3Weights W{sinfo, fcr, hp->k3, sizeof(T) == 4 ? reinterpret_cast<const float4*>(apos) : nullptr, aw, aw_s};
#define MI_D3_CN(BIG_, SORT_, PKIN_, OUT_, OUTFLAG_, GATE_, WANT_)                                                                                 \
  d3_cn_kernel<T, CSR, BIG_, SORT_, PKIN_><<<mi_blocks(N, D3_LS_WAVES), D3_LS_WAVES * MI_WAVE, 0, st>>>(                                             \
      pos, numbers, N, idx, ush, nptr, M, fill_value, cell, batch_idx, P, apos, aaux, cn, OUT_, OUTFLAG_, inv, acn, aaux_s,                           \
      sortable ? order_probe + 1 : nullptr, pre_words, W, GATE_, WANT_)
  const bool big_list = (double)n_entries * ((cell && ush) ? 16.0 : 4.0) > 2.0e9;  // list bytes (see d3_fetch)
  if constexpr (!CSR) {
    if (use_cn) {
      // the search summed the coordination numbers: ONE small launch adopts them, or -- fingerprint mismatch, overflowed rows -- runs the pass
      const int n_blocks = mi_blocks(N, D3_LS_WAVES);
      const int* hdr = reinterpret_cast<const int*>(cn_block);
      MI_TIMED("d3_cn", st, (d3_cn_pre_kernel<T><<<n_blocks < D3_CN_PRE_GRID ? n_blocks : D3_CN_PRE_GRID, D3_LS_WAVES * MI_WAVE, 0, st>>>(
                                pos, numbers, N, idx, ush, nptr, M, fill_value, cell, batch_idx, P, apos, aaux, cn, nullptr, nullptr, inv, acn, aaux_s,
                                sortable ? order_probe + 1 : nullptr, pre_words, W, gflag, hdr, gslots,
                                reinterpret_cast<const float*>((const char*)cn_block + MI_CN_HEADER_BYTES), n_blocks)));
    } else if (pre) {
      // companion given: the PKIN launch works when it is usable (D3Guard), the plain launch beside it otherwise.  Sorted: both write the
      // place-coded list of the spatial order into the workspace copy; otherwise nothing is written (the later passes read `pre`).
      unsigned* out = sorted ? pk : nullptr;
      int* outflag = sorted ? pk_flag : nullptr;
      if (sorted) { MI_TIMED("d3_cn", st, (MI_D3_CN(true, true, true, out, outflag, gflag, 0))); MI_D3_CN(true, true, false, out, outflag, gflag, 1); }
      else { MI_TIMED("d3_cn", st, (MI_D3_CN(true, false, true, out, outflag, gflag, 0))); MI_D3_CN(true, false, false, out, outflag, gflag, 1); }
    }
  }
  if (!pre) {
    if (big_list) { if (sorted) { MI_TIMED("d3_cn", st, (MI_D3_CN(true, true, false, pk, pk_flag, nullptr, 0))); } else { MI_TIMED("d3_cn", st, (MI_D3_CN(true, false, false, pk, pk_flag, nullptr, 0))); } }
    else { if (sorted) { MI_TIMED("d3_cn", st, (MI_D3_CN(false, true, false, pk, pk_flag, nullptr, 0))); } else { MI_TIMED("d3_cn", st, (MI_D3_CN(false, false, false, pk, pk_flag, nullptr, 0))); } }
  }
#undef MI_D3_CN
  if (publish) d3_order_publish(order_probe, order_slot, st);
  MI_LAUNCH_CHECK();
  // all three variants are launched; two of them exit at once on the device-side species info.  Only the fp32 factorised variant
  // is instantiated with the 5-waves-per-SIMD register cap (the others would spill under it).
  auto launch_energy = [&](auto mode, auto packed) {
    constexpr int MODE_ = decltype(mode)::value;
    constexpr bool PK_ = decltype(packed)::value;
    const unsigned* pk_in = PK_ ? pk : nullptr;
    if constexpr (MODE_ == 2 && sizeof(T) == 4)
      d3_energy_kernel_w5<T, CSR, MODE_, PK_><<<blocks, 256, 0, st>>>(pos, numbers, N, idx, ush, nptr, M, fill_value, cell, batch_idx, P, cn, want_virial,
                                                                     smap, sinfo, ctab, ftab, fcr, apos, aaux, aw, dEdCN, forces, e_atom, v_atom, pk_in, pk_flag, inv, apos_s, aaux_s, aw_s, dEdCN_s);
    else
      d3_energy_kernel<T, CSR, MODE_, PK_><<<blocks, 256, 0, st>>>(pos, numbers, N, idx, ush, nptr, M, fill_value, cell, batch_idx, P, cn, want_virial,
                                                                  smap, sinfo, ctab, ftab, fcr, apos, aaux, aw, dEdCN, forces, e_atom, v_atom, pk_in, pk_flag, inv, apos_s, aaux_s, aw_s, dEdCN_s);
  };
  using Plain = std::integral_constant<bool, false>;
  using Packed = std::integral_constant<bool, true>;
  auto launch_modes = [&](auto packed) {
    launch_energy(std::integral_constant<int, 2>{}, packed);
    launch_energy(std::integral_constant<int, 1>{}, packed);
    launch_energy(std::integral_constant<int, 0>{}, packed);
  };
  // with a packed list the PK variants run; if the CN pass found a shift outside {-1, 0, 1} they exit and the fallback launch (energy) or
  // the in-kernel fallback (chain) walks the caller's arrays
  if (pk) {
    MI_TIMED("d3_energy", st, (launch_energy(std::integral_constant<int, 2>{}, Packed{})));
    MI_LAUNCH_CHECK();
    d3_energy_fallback_kernel<T, CSR><<<blocks < D3_FALLBACK_GRID ? blocks : D3_FALLBACK_GRID, 256, 0, st>>>(
        pos, numbers, N, idx, ush, nptr, M, fill_value, cell, batch_idx, P, cn, want_virial, smap, sinfo, ctab, ftab, fcr, apos, aaux, aw, dEdCN, forces, e_atom,
        v_atom, nullptr, pk_flag, inv, apos_s, aaux_s, aw_s, dEdCN_s, pk, blocks);
  } else { MI_TIMED("d3_energy", st, (launch_modes(Plain{}))); }
  MI_LAUNCH_CHECK();
  auto launch_chain = [&](auto packed) {
    constexpr bool PK_ = decltype(packed)::value;
    d3_chain_kernel<T, CSR, PK_><<<mi_blocks(N, D3_CH_LS_WAVES), D3_CH_LS_WAVES * MI_WAVE, 0, st>>>(pos, numbers, N, idx, ush, nptr, M, fill_value, cell, batch_idx, P, apos,
                                                                                             dEdCN, want_virial, forces, v_atom, PK_ ? pk : nullptr, pk_flag, inv, apos_s, dEdCN_s, sinfo);
  };
  if (pk) { MI_TIMED("d3_chain", st, (launch_chain(Packed{}))); }
  else { MI_TIMED("d3_chain", st, (launch_chain(Plain{}))); }
  MI_LAUNCH_CHECK();
  double* sums = reinterpret_cast<double*>(ws + L.sums);  // zeroed with `present` at the top of the call
  d3_reduce_kernel<<<D3_REDUCE_WAVES / 4, 256, 0, st>>>(e_atom, v_atom, batch_idx, N, want_virial, sums);
  MI_LAUNCH_CHECK();
  d3_finish_kernel<<<mi_blocks(10ll * B, 256), 256, 0, st>>>(sums, B, want_virial, energy, virial);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

}  // namespace

extern "C" {

size_t mi_d3_workspace_bytes(int n_atoms, int n_systems, int nz) {
  if (n_atoms < 0 || nz < 1 || n_systems < 1) return 0;
  return d3_layout(n_atoms, nz, MI_F64, n_systems).total;  // sized for the wider dtype
}

size_t mi_d3_workspace_bytes_entries(int n_atoms, int n_systems, int nz, long long n_entries) {
  const size_t base = mi_d3_workspace_bytes(n_atoms, n_systems, nz);
  if (base == 0 || n_entries <= 0 || n_atoms >= D3_PK_MAX_ATOMS) return base;
  return base + 512 + sizeof(unsigned) * (size_t)n_entries;  // alignment slack + flag line + the words
}

size_t mi_d3_workspace_bytes_packed(int n_atoms, int n_systems, int nz, int max_neighbors) {
  return mi_d3_workspace_bytes_entries(n_atoms, n_systems, nz, max_neighbors > 0 ? (long long)n_atoms * max_neighbors : 0);
}

}  // extern "C" (the helper below has internal linkage)
static int d3_entry(const void* positions, const int32_t* numbers, int n_atoms, int dtype, const int32_t* idx_j, const int32_t* unit_shifts,
          const int32_t* neighbor_ptr, int max_neighbors, long long n_list_entries, int fill_value, const void* cell,
          const int32_t* batch_idx, int n_systems,
          const mi_d3_params* params, int compute_virial, float* energy, float* forces, float* coord_num, float* virial, void* workspace,
          size_t workspace_bytes, void* stream, const void* packed_list, const void* cn_block = nullptr, int verify_stride = 64, int verify_phase = 0) {
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_F64, "dtype");
  MI_REQUIRE(n_atoms >= 0 && n_systems >= 1, "sizes");
  if (n_atoms == 0) return MI_OK;
  MI_REQUIRE(positions && numbers && idx_j && params && energy && forces && coord_num && workspace, "null pointer");
  MI_REQUIRE(params->rcov && params->r4r2 && params->c6ab && params->cn_ref && params->nz >= 2, "D3 parameter tables");
  MI_REQUIRE(!compute_virial || virial, "virial output");
  D3Layout L = d3_layout(n_atoms, params->nz, dtype, n_systems);
  if (workspace_bytes < L.total) { mi_set_error("workspace too small: %zu < %zu", workspace_bytes, L.total); return MI_EWORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  const bool csr = neighbor_ptr != nullptr;
  // fill_value >= n_atoms: every stored index is then a neighbour (the companion marks padding itself; an index limit below n_atoms would
  // turn real entries into padding, which only the CN pass re-checks)
  MI_REQUIRE(!packed_list || (!csr && cell && unit_shifts && max_neighbors > 0 && n_atoms < D3_PK_MAX_ATOMS && fill_value >= n_atoms),
             "packed_list: periodic padded matrix (cell + shifts), fill_value >= n_atoms, n_atoms < 2^26");
  // a periodic padded list is re-read by all three passes: with the larger workspace the CN pass leaves a 4 B/slot copy for the others
  unsigned* pk = nullptr;
  // CSR: `n_list_entries` = length of idx_j (0 = not given: no packed copy); matrix: n_atoms x max_neighbors
  const long long n_entries = csr ? n_list_entries : (long long)n_atoms * max_neighbors;
  if (cell && unit_shifts && n_entries > 0 && n_atoms < D3_PK_MAX_ATOMS &&
      workspace_bytes >= mi_d3_workspace_bytes_entries(n_atoms, n_systems, params->nz, n_entries))
    pk = reinterpret_cast<unsigned*>((char*)workspace + ((mi_d3_workspace_bytes(n_atoms, n_systems, params->nz) + 255) & ~(size_t)255));
#define MI_D3_CALL(T_, CSR_)                                                                                                              \
  return d3_impl<T_, CSR_>((const T_*)positions, numbers, n_atoms, idx_j, unit_shifts, neighbor_ptr, max_neighbors, fill_value, (const T_*)cell, \
                           batch_idx, n_systems, params, compute_virial, energy, forces, coord_num, virial, (char*)workspace, L, pk, n_entries,      \
                           (const unsigned*)packed_list, cn_block, verify_stride, verify_phase, st)
  if (dtype == MI_F32) { if (csr) MI_D3_CALL(float, true); else MI_D3_CALL(float, false); }
  else { if (csr) MI_D3_CALL(double, true); else MI_D3_CALL(double, false); }
#undef MI_D3_CALL
}

extern "C" {
int mi_d3(const void* positions, const int32_t* numbers, int n_atoms, int dtype, const int32_t* idx_j, const int32_t* unit_shifts,
          const int32_t* neighbor_ptr, int max_neighbors, long long n_list_entries, int fill_value, const void* cell,
          const int32_t* batch_idx, int n_systems,
          const mi_d3_params* params, int compute_virial, float* energy, float* forces, float* coord_num, float* virial, void* workspace,
          size_t workspace_bytes, void* stream) {
  return d3_entry(positions, numbers, n_atoms, dtype, idx_j, unit_shifts, neighbor_ptr, max_neighbors, n_list_entries, fill_value, cell, batch_idx,
                  n_systems, params, compute_virial, energy, forces, coord_num, virial, workspace, workspace_bytes, stream, nullptr);
}

int mi_d3_packed(const void* positions, const int32_t* numbers, int n_atoms, int dtype, const int32_t* neighbor_matrix,
                 const int32_t* neighbor_matrix_shifts, int max_neighbors, int fill_value, const void* cell, const int32_t* batch_idx,
                 int n_systems, const mi_d3_params* params, int compute_virial, float* energy, float* forces, float* coord_num, float* virial,
                 void* workspace, size_t workspace_bytes, const void* packed_list, size_t packed_bytes, void* stream) {
  MI_REQUIRE(packed_list != nullptr, "packed_list");
  MI_REQUIRE(max_neighbors > 0 && packed_bytes >= 256 + sizeof(unsigned) * (size_t)n_atoms * (size_t)max_neighbors,
             "packed_bytes: mi_nl_packed_bytes(n_atoms, max_neighbors)");
  return d3_entry(positions, numbers, n_atoms, dtype, neighbor_matrix, neighbor_matrix_shifts, nullptr, max_neighbors, 0, fill_value, cell, batch_idx,
                  n_systems, params, compute_virial, energy, forces, coord_num, virial, workspace, workspace_bytes, stream, packed_list);
}

int mi_d3_packed_cn(const void* positions, const int32_t* numbers, int n_atoms, int dtype, const int32_t* neighbor_matrix,
                    const int32_t* neighbor_matrix_shifts, int max_neighbors, int fill_value, const void* cell, const int32_t* batch_idx,
                    int n_systems, const mi_d3_params* params, int compute_virial, float* energy, float* forces, float* coord_num, float* virial,
                    void* workspace, size_t workspace_bytes, const void* packed_list, size_t packed_bytes, const void* cn_block, size_t cn_bytes,
                    int verify_stride, int verify_phase, void* stream) {
  MI_REQUIRE(packed_list != nullptr, "packed_list");
  MI_REQUIRE(max_neighbors > 0 && packed_bytes >= 256 + sizeof(unsigned) * (size_t)n_atoms * (size_t)max_neighbors,
             "packed_bytes: mi_nl_packed_bytes(n_atoms, max_neighbors)");
  MI_REQUIRE(!cn_block || cn_bytes >= MI_CN_HEADER_BYTES + sizeof(float) * (size_t)n_atoms, "cn_bytes: mi_nl_cn_bytes(n_atoms)");
  MI_REQUIRE(verify_stride >= 0 && verify_phase >= 0, "verify_stride / verify_phase");
  return d3_entry(positions, numbers, n_atoms, dtype, neighbor_matrix, neighbor_matrix_shifts, nullptr, max_neighbors, 0, fill_value, cell, batch_idx,
                  n_systems, params, compute_virial, energy, forces, coord_num, virial, workspace, workspace_bytes, stream, packed_list, cn_block,
                  verify_stride, verify_phase);
}

}  // extern "C"

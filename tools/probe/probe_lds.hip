// Tuning probe (not part of the library): the D3 chain-rule walk with the neighbour records staged in LDS.
//
// Question: the energy / chain passes of mi_d3 stream a 4 B/slot packed list and pay one or two
// 16-byte L2 gathers per neighbour (~21 distinct cache lines per 64-lane gather).  If a block owns one cell of a search grid and stages the
// records of that cell's (2R+1)^3 neighbourhood in LDS, the packed word can be a 16-bit neighbourhood-local slot and the gather a ds_read_b128.
//   translate  = the CN pass as mi_d3 runs it (lock-step wave-per-atom walk of the caller's 16 B/slot list, one 16-byte gather per
//                neighbour, fp64 lane partials) that also derives the slot of every entry and writes the 2 B/slot list
//   chain_lds  = block per cell (16 waves, ~140 KB LDS), rows walked from the 2 B/slot list, records from LDS
//   chain_ref  = plain wave-per-atom walk of the caller's list (correctness reference for chain_lds)
// Orthorhombic single box, all atoms inside the box (probe only).
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WAVE 64
struct Int3 { int a, b, c; };
__device__ __forceinline__ double wsum(double v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64); return v; }

struct Grid { int nx, ny, nz, R; float lx, ly, lz; };

__device__ __forceinline__ float p_exp(float x) {
  const float L2E_HI = 1.44269502e+00f, L2E_LO = 1.92596299e-08f, LN2 = 6.93147182e-01f;
  const float t = x * L2E_HI;
  float lo = fmaf(x, L2E_HI, -t);
  lo = fmaf(x, L2E_LO, lo);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, lo * LN2, e);
}
__device__ __forceinline__ float p_cn_count(float rinv, float rci, float rcj, float k1, float* dcn) {
  const float rr = (rci + rcj) * rinv;
  const float f = __builtin_amdgcn_rcpf(1.0f + p_exp(-k1 * (rr - 1.0f)));
  if (dcn) *dcn = -f * (1.0f - f) * k1 * rr * rinv;
  return f;
}

// meta word of an atom: cell x | y << 7 | z << 14 | rank in cell << 21 | species << 28
#define LS_WAVES 8
#define NB_CELLS_MAX 344

__global__ __launch_bounds__(LS_WAVES* WAVE) void translate_kernel(const float4* __restrict__ apos /* {x,y,z,meta} by atom index */, int N,
                                                                  const int* __restrict__ idx, const int* __restrict__ ush, int M, int Mp,
                                                                  const int* __restrict__ cell_start, Grid G, const float* __restrict__ rcov_tab,
                                                                  float k1, unsigned short* __restrict__ pk, float* __restrict__ cn,
                                                                  int* __restrict__ flags /* [0] out-of-window entries */, int do_pack) {
  __shared__ unsigned short nbpre[LS_WAVES][NB_CELLS_MAX + 8];
  __shared__ int trips_sh[LS_WAVES];
  __shared__ float rc_tab[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x < 16) rc_tab[threadIdx.x] = rcov_tab[threadIdx.x];
  const int i0 = __builtin_amdgcn_readfirstlane(blockIdx.x * LS_WAVES + w);
  const int i = i0 < N ? i0 : N - 1;
  const bool live = i0 < N;
  const float4 pi = apos[i];
  const unsigned mi = __float_as_uint(pi.w);
  const int cix = mi & 127, ciy = (mi >> 7) & 127, ciz = (mi >> 14) & 127;
  const int R = G.R, D = 2 * R + 1, ncell = D * D * D;
  if (do_pack) {  // prefix of the neighbourhood cells' occupancies, in neighbourhood order (x fastest)
    int carry = 0;
    for (int n0 = 0; n0 < ncell; n0 += 64) {
      const int n = n0 + lane;
      int cnt = 0;
      if (n < ncell) {
        const int dx = n % D - R, dy = (n / D) % D - R, dz = n / (D * D) - R;
        int tx = cix + dx, ty = ciy + dy, tz = ciz + dz;
        tx = (tx % G.nx + G.nx) % G.nx; ty = (ty % G.ny + G.ny) % G.ny; tz = (tz % G.nz + G.nz) % G.nz;
        const int id = tx + G.nx * (ty + G.ny * tz);
        cnt = cell_start[id + 1] - cell_start[id];
      }
      int inc = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(inc, o, 64); if (lane >= o) inc += up; }
      if (n < ncell) nbpre[w][n] = (unsigned short)(carry + inc - cnt);
      carry += __shfl(inc, 63, 64);
    }
  }
  const float rci = rc_tab[mi >> 28];
  const long long beg = (long long)i * M, end = live ? beg + M : beg;
  if (lane == 0) trips_sh[w] = (int)((end - beg + 63) / 64);
  __syncthreads();
  int trips = 0;
#pragma unroll
  for (int k = 0; k < LS_WAVES; ++k) trips = max(trips, trips_sh[k]);
  const Int3* __restrict__ ush3 = reinterpret_cast<const Int3*>(ush);
  double acc = 0.0;
  long long e = beg + lane;
  // 3-deep software pipeline as in d3_cn_kernel
  int j0 = 0, j1 = 0; Int3 s0 = {0, 0, 0}, s1 = {0, 0, 0}; bool in0 = e < end, in1 = e + 64 < end;
  if (in0) { j0 = __builtin_nontemporal_load(idx + e); const int* u = (const int*)(ush3 + e); s0 = Int3{__builtin_nontemporal_load(u), __builtin_nontemporal_load(u + 1), __builtin_nontemporal_load(u + 2)}; }
  if (in1) { j1 = __builtin_nontemporal_load(idx + e + 64); const int* u = (const int*)(ush3 + e + 64); s1 = Int3{__builtin_nontemporal_load(u), __builtin_nontemporal_load(u + 1), __builtin_nontemporal_load(u + 2)}; }
  bool v0 = in0 && (unsigned)j0 < (unsigned)N;
  float4 p0 = apos[v0 ? j0 : i];
  for (int trip = 0; trip < trips; ++trip) {
    __syncthreads();
    int j2 = 0; Int3 s2 = {0, 0, 0}; const bool in2 = e + 128 < end;
    if (in2) { j2 = __builtin_nontemporal_load(idx + e + 128); const int* u = (const int*)(ush3 + e + 128); s2 = Int3{__builtin_nontemporal_load(u), __builtin_nontemporal_load(u + 1), __builtin_nontemporal_load(u + 2)}; }
    const bool v1 = in1 && (unsigned)j1 < (unsigned)N;
    const float4 p1 = apos[v1 ? j1 : i];
    const unsigned mj = __float_as_uint(p0.w);
    if (do_pack && in0) {
      const int rx = (int)(mj & 127) + s0.a * G.nx - cix + R, ry = (int)((mj >> 7) & 127) + s0.b * G.ny - ciy + R, rz = (int)((mj >> 14) & 127) + s0.c * G.nz - ciz + R;
      const bool inw = (unsigned)rx < (unsigned)D && (unsigned)ry < (unsigned)D && (unsigned)rz < (unsigned)D;
      unsigned short word = 0xffff;
      if (v0 && inw) word = (unsigned short)(nbpre[w][rx + D * (ry + D * rz)] + ((mj >> 21) & 127));
      if (v0 && !inw) atomicAdd(flags, 1);
      __builtin_nontemporal_store(word, pk + (size_t)i * Mp + (e - beg));
    }
    if (__any(v0)) {
      const float dx = (p0.x - pi.x) + (float)s0.a * G.lx, dy = (p0.y - pi.y) + (float)s0.b * G.ly, dz = (p0.z - pi.z) + (float)s0.c * G.lz;
      const float r2 = dx * dx + dy * dy + dz * dz;
      const bool ok = !(r2 < 1e-24f);
      const float rinv = ok ? __builtin_amdgcn_rsqf(ok ? r2 : 1.0f) : 0.0f;
      const float f = p_cn_count(rinv, rci, rc_tab[mj >> 28], k1, nullptr);
      acc += (v0 && ok) ? (double)f : 0.0;
    }
    j0 = j1; s0 = s1; in0 = in1; v0 = v1; p0 = p1; j1 = j2; s1 = s2; in1 = in2;
    e += 64;
  }
  acc = wsum(acc);
  if (lane == 0 && live) cn[i] = (float)acc;
}

// ---- chain walk, records staged in LDS ---------------------------------------------------------------------------------------
#define LW 16
#define NBMAX 7680
__global__ __launch_bounds__(LW* WAVE) void chain_lds_kernel(const float4* __restrict__ srec /* cell-sorted {x,y,z,dE/dCN} */,
                                                            const unsigned char* __restrict__ sspec, const int* __restrict__ sidx,
                                                            const int* __restrict__ cell_start, Grid G, const unsigned short* __restrict__ pk, int Mp,
                                                            const float* __restrict__ rcov_tab, float k1, int want_virial, float* __restrict__ forces,
                                                            double* __restrict__ v_atom, int* __restrict__ work, int* __restrict__ flags) {
  __shared__ float4 A[NBMAX];
  __shared__ unsigned short Bm[NBMAX];
  __shared__ int nb_start[NB_CELLS_MAX], nb_pre[NB_CELLS_MAX + 1];
  __shared__ short nb_code[NB_CELLS_MAX];
  __shared__ int wave_tot[8];
  __shared__ double part[LW][12];
  __shared__ float rc_tab[16];
  __shared__ int next_cell;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 16) rc_tab[tid] = rcov_tab[tid];
  const int R = G.R, D = 2 * R + 1, ncell = D * D * D;
  const int total_cells = G.nx * G.ny * G.nz;
  const int nch = Mp / 256;
  for (;;) {
    __syncthreads();
    if (tid == 0) next_cell = atomicAdd(work, 1);
    __syncthreads();
    const int c = next_cell;
    if (c >= total_cells) break;
    const int c_beg = cell_start[c], n_c = cell_start[c + 1] - c_beg;
    if (n_c == 0) continue;
    const int cx = c % G.nx, cy = (c / G.nx) % G.ny, cz = c / (G.nx * G.ny);
    // neighbourhood table + exclusive prefix of the occupancies
    int cnt = 0;
    if (tid < ncell) {
      const int dx = tid % D - R, dy = (tid / D) % D - R, dz = tid / (D * D) - R;
      int tx = cx + dx, ty = cy + dy, tz = cz + dz;
      int sx = 0, sy = 0, sz = 0;
      while (tx < 0) { tx += G.nx; --sx; } while (tx >= G.nx) { tx -= G.nx; ++sx; }
      while (ty < 0) { ty += G.ny; --sy; } while (ty >= G.ny) { ty -= G.ny; ++sy; }
      while (tz < 0) { tz += G.nz; --sz; } while (tz >= G.nz) { tz -= G.nz; ++sz; }
      const int id = tx + G.nx * (ty + G.ny * tz);
      const int b = cell_start[id];
      cnt = cell_start[id + 1] - b;
      nb_start[tid] = b;
      nb_code[tid] = (short)((sx + 1) | ((sy + 1) << 2) | ((sz + 1) << 4));
    }
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(inc, o, 64); if (lane >= o) inc += up; }
    if (lane == 63 && wave < 8) wave_tot[wave] = inc;
    __syncthreads();
    if (tid < ncell) {
      int off = 0;
      for (int k = 0; k < wave; ++k) off += wave_tot[k];
      nb_pre[tid] = off + inc - cnt;
      if (tid == ncell - 1) nb_pre[ncell] = off + inc;
    }
    __syncthreads();
    const int total = nb_pre[ncell];
    if (total > NBMAX) { if (tid == 0) atomicAdd(flags + 1, 1); continue; }
    for (int n = wave; n < ncell; n += LW) {
      const int s = nb_start[n], p = nb_pre[n], m = nb_pre[n + 1] - p;
      const int code = nb_code[n];
      for (int k = lane; k < m; k += 64) {
        A[p + k] = srec[s + k];
        Bm[p + k] = (unsigned short)(sspec[s + k] | (code << 4));
      }
    }
    __syncthreads();
    // rows: full rounds of LW atoms, then the left-over atoms split into q segments each
    const int full = (n_c / LW) * LW, left = n_c - full;
    int q = 1;
    if (left) { q = LW / left; if (q > nch) q = nch; }
    const int rounds = n_c / LW + (left ? 1 : 0);
    for (int r = 0; r < rounds; ++r) {
      const bool last = left && r == rounds - 1;
      int a, k0 = 0, k1c = nch;
      bool active = true;
      if (!last) a = r * LW + wave;
      else {
        a = full + wave / q;
        active = wave < left * q;
        const int seg = wave % q;
        k0 = seg * nch / q; k1c = (seg + 1) * nch / q;
        if (!active) { a = full; k0 = k1c = 0; }
      }
      const int is = c_beg + a;
      const float4 ri = srec[is];
      const float rci = rc_tab[sspec[is] & 15], di = ri.w;
      const int io = __builtin_amdgcn_readfirstlane(sidx[is]);
      typedef unsigned p_u2 __attribute__((ext_vector_type(2)));
      const p_u2* __restrict__ row = reinterpret_cast<const p_u2*>(pk + (size_t)io * Mp);
      double Fx = 0, Fy = 0, Fz = 0;
      double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      p_u2 wn = {0xffffffffu, 0xffffffffu};
      if (k0 < k1c) wn = __builtin_nontemporal_load(row + (size_t)k0 * 64 + lane);
      for (int k = k0; k < k1c; ++k) {
        const p_u2 wc = wn;
        if (k + 1 < k1c) wn = __builtin_nontemporal_load(row + (size_t)(k + 1) * 64 + lane);
        const unsigned sl[4] = {wc.x & 0xffffu, wc.x >> 16, wc.y & 0xffffu, wc.y >> 16};
        float4 pj[4]; unsigned mj[4]; bool val[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { val[t] = sl[t] != 0xffffu; const unsigned s = val[t] ? sl[t] : 0u; pj[t] = A[s]; mj[t] = Bm[s]; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (!__any(val[t])) continue;
          const float sx = (float)((int)((mj[t] >> 4) & 3) - 1), sy = (float)((int)((mj[t] >> 6) & 3) - 1), sz = (float)((int)((mj[t] >> 8) & 3) - 1);
          const float dx = (pj[t].x - ri.x) + sx * G.lx, dy = (pj[t].y - ri.y) + sy * G.ly, dz = (pj[t].z - ri.z) + sz * G.lz;
          const float r2 = dx * dx + dy * dy + dz * dz;
          const bool ok = !(r2 < 1e-24f);
          const float rinv = ok ? __builtin_amdgcn_rsqf(ok ? r2 : 1.0f) : 0.0f;
          float dcn;
          p_cn_count(rinv, rci, rc_tab[mj[t] & 15], k1, &dcn);
          const float dEdr = (val[t] && ok) ? (di + pj[t].w) * dcn : 0.0f;
          const float fx = dEdr * (dx * rinv), fy = dEdr * (dy * rinv), fz = dEdr * (dz * rinv);
          Fx += (double)fx; Fy += (double)fy; Fz += (double)fz;
          if (want_virial) {
            V[0] += (double)(fx * dx); V[1] += (double)(fx * dy); V[2] += (double)(fx * dz);
            V[3] += (double)(fy * dx); V[4] += (double)(fy * dy); V[5] += (double)(fy * dz);
            V[6] += (double)(fz * dx); V[7] += (double)(fz * dy); V[8] += (double)(fz * dz);
          }
        }
      }
      Fx = wsum(Fx); Fy = wsum(Fy); Fz = wsum(Fz);
      if (want_virial) {
#pragma unroll
        for (int k = 0; k < 9; ++k) V[k] = wsum(V[k]);
      }
      if (!last || q == 1) {
        if (active && lane == 0) { forces[3 * (size_t)io] = (float)Fx; forces[3 * (size_t)io + 1] = (float)Fy; forces[3 * (size_t)io + 2] = (float)Fz; }
        if (active && want_virial && lane < 9) { double v = V[0];
#pragma unroll
          for (int k = 1; k < 9; ++k) v = lane == k ? V[k] : v;
          v_atom[9 * (size_t)io + lane] = -0.5 * v; }
      } else {
        if (lane == 0) { part[wave][0] = Fx; part[wave][1] = Fy; part[wave][2] = Fz; }
        if (lane < 9) { double v = V[0];
#pragma unroll
          for (int k = 1; k < 9; ++k) v = lane == k ? V[k] : v;
          part[wave][3 + lane] = v; }
        __syncthreads();
        if (active && (wave % q) == 0 && lane < 12) {
          double s = 0.0;
          for (int g = 0; g < q; ++g) s += part[wave + g][lane];
          if (lane < 3) forces[3 * (size_t)io + lane] = (float)s;
          else if (want_virial) v_atom[9 * (size_t)io + lane - 3] = -0.5 * s;
        }
      }
    }
  }
}

// plain reference: wave per atom, caller's list, global gathers
__global__ __launch_bounds__(256) void chain_ref_kernel(const float4* __restrict__ apos, const float* __restrict__ dEdCN, int N, const int* __restrict__ idx,
                                                       const int* __restrict__ ush, int M, Grid G, const float* __restrict__ rcov_tab, float k1,
                                                       int want_virial, float* __restrict__ forces, double* __restrict__ v_atom) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const float4 pi = apos[i];
  const float rci = rcov_tab[__float_as_uint(pi.w) >> 28], di = dEdCN[i];
  const Int3* ush3 = reinterpret_cast<const Int3*>(ush);
  double Fx = 0, Fy = 0, Fz = 0;
  double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (long long e = (long long)i * M + lane; e < (long long)(i + 1) * M; e += 64) {
    const int j = idx[e];
    if ((unsigned)j >= (unsigned)N) continue;
    const Int3 s = ush3[e];
    const float4 pj = apos[j];
    const float dx = (pj.x - pi.x) + (float)s.a * G.lx, dy = (pj.y - pi.y) + (float)s.b * G.ly, dz = (pj.z - pi.z) + (float)s.c * G.lz;
    const float r2 = dx * dx + dy * dy + dz * dz;
    const bool ok = !(r2 < 1e-24f);
    const float rinv = ok ? __builtin_amdgcn_rsqf(ok ? r2 : 1.0f) : 0.0f;
    float dcn;
    p_cn_count(rinv, rci, rcov_tab[__float_as_uint(pj.w) >> 28], k1, &dcn);
    const float dEdr = ok ? (di + dEdCN[j]) * dcn : 0.0f;
    const float fx = dEdr * (dx * rinv), fy = dEdr * (dy * rinv), fz = dEdr * (dz * rinv);
    Fx += (double)fx; Fy += (double)fy; Fz += (double)fz;
    if (want_virial) {
      V[0] += (double)(fx * dx); V[1] += (double)(fx * dy); V[2] += (double)(fx * dz);
      V[3] += (double)(fy * dx); V[4] += (double)(fy * dy); V[5] += (double)(fy * dz);
      V[6] += (double)(fz * dx); V[7] += (double)(fz * dy); V[8] += (double)(fz * dz);
    }
  }
  Fx = wsum(Fx); Fy = wsum(Fy); Fz = wsum(Fz);
  if (lane == 0) { forces[3 * (size_t)i] = (float)Fx; forces[3 * (size_t)i + 1] = (float)Fy; forces[3 * (size_t)i + 2] = (float)Fz; }
  if (want_virial) {
    for (int k = 0; k < 9; ++k) { const double v = wsum(V[k]); if (lane == 0) v_atom[9 * (size_t)i + k] = -0.5 * v; }
  }
}


// ---- v2: pruned neighbourhood (per-row x half-width), 20 B records staged with global_load_lds, 16-bit word = slot | shifted-bits << 13,
// software-pipelined LDS reads, branch-free pair math ------------------------------------------------------------------------------
struct Grid2 { int nx, ny, nz, R; float lx, ly, lz; signed char dxlim[5][5]; };
#define NBMAX2 7680

// per-wave table of the pruned neighbourhood: prefix of cell occupancies, x fastest; pruned cells count 0
__device__ __forceinline__ void nb_prefix_wave(const Grid2& G, int cix, int ciy, int ciz, const int* __restrict__ cell_start, unsigned short* __restrict__ pre, int lane) {
  const int R = G.R, D = 2 * R + 1, ncell = D * D * D;
  int carry = 0;
  for (int n0 = 0; n0 < ncell; n0 += 64) {
    const int n = n0 + lane;
    int cnt = 0;
    if (n < ncell) {
      const int dx = n % D - R, dy = (n / D) % D - R, dz = n / (D * D) - R;
      if ((dx < 0 ? -dx : dx) <= G.dxlim[dz < 0 ? -dz : dz][dy < 0 ? -dy : dy]) {
        int tx = cix + dx, ty = ciy + dy, tz = ciz + dz;
        tx = tx < 0 ? tx + G.nx : (tx >= G.nx ? tx - G.nx : tx); ty = ty < 0 ? ty + G.ny : (ty >= G.ny ? ty - G.ny : ty); tz = tz < 0 ? tz + G.nz : (tz >= G.nz ? tz - G.nz : tz);
        const int id = tx + G.nx * (ty + G.ny * tz);
        cnt = cell_start[id + 1] - cell_start[id];
      }
    }
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(inc, o, 64); if (lane >= o) inc += up; }
    if (n < ncell) pre[n] = (unsigned short)(cnt ? carry + inc - cnt : 0xffff);
    carry += __shfl(inc, 63, 64);
  }
}

__global__ __launch_bounds__(LS_WAVES* WAVE) void translate2_kernel(const float4* __restrict__ apos, int N, const int* __restrict__ idx, const int* __restrict__ ush,
                                                                   int M, int Mp, const int* __restrict__ cell_start, Grid2 G, const float* __restrict__ rcov_tab,
                                                                   float k1, unsigned* __restrict__ pk32, float* __restrict__ cn, int* __restrict__ flags,
                                                                   int* __restrict__ rowflag) {
  __shared__ unsigned short nbpre[LS_WAVES][NB_CELLS_MAX + 8];
  __shared__ int trips_sh[LS_WAVES];
  __shared__ float rc_tab[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x < 16) rc_tab[threadIdx.x] = rcov_tab[threadIdx.x];
  const int i0 = __builtin_amdgcn_readfirstlane(blockIdx.x * LS_WAVES + w);
  const int i = i0 < N ? i0 : N - 1;
  const bool live = i0 < N;
  const float4 pi = apos[i];
  const unsigned mi = __float_as_uint(pi.w);
  const int cix = mi & 127, ciy = (mi >> 7) & 127, ciz = (mi >> 14) & 127;
  const int R = G.R, D = 2 * R + 1;
  nb_prefix_wave(G, cix, ciy, ciz, cell_start, nbpre[w], lane);
  const float rci = rc_tab[mi >> 28];
  const long long beg = (long long)i * M, end = live ? beg + M : beg;
  if (lane == 0) trips_sh[w] = (int)((end - beg + 63) / 64);
  __syncthreads();
  int trips = 0;
#pragma unroll
  for (int k = 0; k < LS_WAVES; ++k) trips = max(trips, trips_sh[k]);
  trips = (trips + 1) & ~1;  // pairs of trips share one 4-byte store per lane
  const Int3* __restrict__ ush3 = reinterpret_cast<const Int3*>(ush);
  double acc = 0.0;
  long long e = beg + lane;
  int j0 = 0, j1 = 0; Int3 s0 = {0, 0, 0}, s1 = {0, 0, 0}; bool in0 = e < end, in1 = e + 64 < end;
  if (in0) { j0 = __builtin_nontemporal_load(idx + e); const int* u = (const int*)(ush3 + e); s0 = Int3{__builtin_nontemporal_load(u), __builtin_nontemporal_load(u + 1), __builtin_nontemporal_load(u + 2)}; }
  if (in1) { j1 = __builtin_nontemporal_load(idx + e + 64); const int* u = (const int*)(ush3 + e + 64); s1 = Int3{__builtin_nontemporal_load(u), __builtin_nontemporal_load(u + 1), __builtin_nontemporal_load(u + 2)}; }
  bool v0 = in0 && (unsigned)j0 < (unsigned)N;
  float4 p0 = apos[v0 ? j0 : i];
  unsigned held = 0xffffu;
  bool bad = false;
  unsigned* __restrict__ row32 = pk32 + (size_t)i * (Mp / 2);
  for (int trip = 0; trip < trips; ++trip) {
    __syncthreads();
    int j2 = 0; Int3 s2 = {0, 0, 0}; const bool in2 = e + 128 < end;
    if (in2) { j2 = __builtin_nontemporal_load(idx + e + 128); const int* u = (const int*)(ush3 + e + 128); s2 = Int3{__builtin_nontemporal_load(u), __builtin_nontemporal_load(u + 1), __builtin_nontemporal_load(u + 2)}; }
    const bool v1 = in1 && (unsigned)j1 < (unsigned)N;
    const float4 p1 = apos[v1 ? j1 : i];
    const unsigned mj = __float_as_uint(p0.w);
    {
      const int rx = (int)(mj & 127) + s0.a * G.nx - cix + R, ry = (int)((mj >> 7) & 127) + s0.b * G.ny - ciy + R, rz = (int)((mj >> 14) & 127) + s0.c * G.nz - ciz + R;
      const bool inw = (unsigned)rx < (unsigned)D && (unsigned)ry < (unsigned)D && (unsigned)rz < (unsigned)D;
      const unsigned pre = nbpre[w][inw ? rx + D * (ry + D * rz) : 0];
      const bool okw = v0 && inw && pre != 0xffffu && ((unsigned)(s0.a + 1) <= 2u) && ((unsigned)(s0.b + 1) <= 2u) && ((unsigned)(s0.c + 1) <= 2u);
      bad = bad || (v0 && !okw);
      const unsigned word = okw ? ((pre + ((mj >> 21) & 127)) | ((s0.a != 0) << 13) | ((s0.b != 0) << 14) | ((s0.c != 0) << 15)) : 0xffffu;
      if (trip & 1) { if (live) __builtin_nontemporal_store(held | (word << 16), row32 + 64 * (trip >> 1) + lane); }
      else held = word;
    }
    if (__any(v0)) {
      const float dx = (p0.x - pi.x) + (float)s0.a * G.lx, dy = (p0.y - pi.y) + (float)s0.b * G.ly, dz = (p0.z - pi.z) + (float)s0.c * G.lz;
      const float r2 = dx * dx + dy * dy + dz * dz;
      const bool ok = !(r2 < 1e-24f);
      const float rinv = ok ? __builtin_amdgcn_rsqf(ok ? r2 : 1.0f) : 0.0f;
      const float f = p_cn_count(rinv, rci, rc_tab[mj >> 28], k1, nullptr);
      acc += (v0 && ok) ? (double)f : 0.0;
    }
    j0 = j1; s0 = s1; in0 = in1; v0 = v1; p0 = p1; j1 = j2; s1 = s2; in1 = in2;
    e += 64;
  }
  acc = wsum(acc);
  if (lane == 0 && live) cn[i] = (float)acc;
  if (__any(bad) && lane == 0 && live) { rowflag[i] = 1; atomicAdd(flags, 1); }
}

__global__ __launch_bounds__(LW* WAVE) void chain_lds2_kernel(const float4* __restrict__ srec, const float* __restrict__ srcov, const int* __restrict__ sidx,
                                                             const int* __restrict__ cell_start, Grid2 G, const unsigned short* __restrict__ pk, int Mp,
                                                             float k1, int want_virial, float* __restrict__ forces, double* __restrict__ v_atom,
                                                             int* __restrict__ work, int* __restrict__ flags) {
  __shared__ float4 A[NBMAX2];
  __shared__ float Rc[NBMAX2];
  __shared__ int seg_src[128], seg_len[128], seg_dst[129];
  __shared__ int wave_tot[4];
  __shared__ double part[LW][12];
  __shared__ int next_cell;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = G.R, D = 2 * R + 1, nseg = 2 * D * D;
  const int total_cells = G.nx * G.ny * G.nz;
  const int nch = Mp / 256;
  for (;;) {
    __syncthreads();
    if (tid == 0) next_cell = atomicAdd(work, 1);
    __syncthreads();
    const int c = next_cell;
    if (c >= total_cells) break;
    const int c_beg = cell_start[c], n_c = cell_start[c + 1] - c_beg;
    if (n_c == 0) continue;
    const int cx = c % G.nx, cy = (c / G.nx) % G.ny, cz = c / (G.nx * G.ny);
    // segment table: row (dy, dz) x {first, second} contiguous run of cells along x
    int len = 0;
    if (tid < nseg) {
      const int row = tid >> 1, half = tid & 1;
      const int dy = row % D - R, dz = row / D - R;
      const int lim = G.dxlim[dz < 0 ? -dz : dz][dy < 0 ? -dy : dy];
      int ty = cy + dy, tz = cz + dz;
      ty = ty < 0 ? ty + G.ny : (ty >= G.ny ? ty - G.ny : ty); tz = tz < 0 ? tz + G.nz : (tz >= G.nz ? tz - G.nz : tz);
      const int rowbase = G.nx * (ty + G.ny * tz);
      int src = 0;
      if (lim >= 0) {
        const int xlo = cx - lim, xhi = cx + lim;  // D <= nx: at most one wrap
        int a, b;  // cell range of this half, in x order of the neighbourhood
        if (xlo < 0) { if (half == 0) { a = xlo + G.nx; b = G.nx - 1; } else { a = 0; b = xhi; } }
        else if (xhi >= G.nx) { if (half == 0) { a = xlo; b = G.nx - 1; } else { a = 0; b = xhi - G.nx; } }
        else { if (half == 0) { a = xlo; b = xhi; } else { a = 0; b = -1; } }
        if (a <= b) { src = cell_start[rowbase + a]; len = cell_start[rowbase + b + 1] - src; }
      }
      seg_src[tid] = src; seg_len[tid] = len;
    }
    int inc = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(inc, o, 64); if (lane >= o) inc += up; }
    if (lane == 63 && wave < 4) wave_tot[wave] = inc;
    __syncthreads();
    if (tid < nseg) {
      int off = 0;
      for (int k = 0; k < wave; ++k) off += wave_tot[k];
      seg_dst[tid] = off + inc - len;
      if (tid == nseg - 1) seg_dst[nseg] = off + inc;
    }
    __syncthreads();
    const int total = seg_dst[nseg];
    if (total > NBMAX2) { if (tid == 0) atomicAdd(flags + 1, 1); continue; }
    // staging: asynchronous global -> LDS copies, 64 records per wave-instruction
    for (int sg = wave; sg < nseg; sg += LW) {
      const int src = __builtin_amdgcn_readfirstlane(seg_src[sg]), ln = __builtin_amdgcn_readfirstlane(seg_len[sg]), dst = __builtin_amdgcn_readfirstlane(seg_dst[sg]);
      for (int o = 0; o < ln; o += 64) {
        if (o + lane < ln) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srec + src + o + lane), (__attribute__((address_space(3))) void*)(A + dst + o), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcov + src + o + lane), (__attribute__((address_space(3))) void*)(Rc + dst + o), 4, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float sgx = (cx - R < 0) ? -G.lx : G.lx, sgy = (cy - R < 0) ? -G.ly : G.ly, sgz = (cz - R < 0) ? -G.lz : G.lz;  // the one image a neighbourhood can reach per axis
    const int full = (n_c / LW) * LW, left = n_c - full;
    int q = 1;
    if (left) { q = LW / left; if (q > nch) q = nch; }
    const int rounds = n_c / LW + (left ? 1 : 0);
    for (int r = 0; r < rounds; ++r) {
      const bool last = left && r == rounds - 1;
      int a, k0 = 0, k1c = nch;
      bool active = true;
      if (!last) a = r * LW + wave;
      else {
        a = full + wave / q;
        active = wave < left * q;
        const int seg = wave % q;
        k0 = seg * nch / q; k1c = (seg + 1) * nch / q;
        if (!active) { a = full; k0 = k1c = 0; }
      }
      const int is = c_beg + a;
      const float4 ri = srec[is];
      const float rci = srcov[is], di = ri.w;
      const int io = __builtin_amdgcn_readfirstlane(sidx[is]);
      typedef unsigned p_u2 __attribute__((ext_vector_type(2)));
      const p_u2* __restrict__ row = reinterpret_cast<const p_u2*>(pk + (size_t)io * Mp);
      double Fx = 0, Fy = 0, Fz = 0;
      double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // xx xy xz yy yz zz in [0..5] (f is parallel to r: the tensor is symmetric), mirrored at the end
      const p_u2 none = {0xffffffffu, 0xffffffffu};
      p_u2 w0 = none, w1 = none, w2 = none;  // words of chunk k, k+1 (prefetched), k+2 (in flight from memory)
      if (k0 < k1c) w0 = __builtin_nontemporal_load(row + (size_t)k0 * 64 + lane);
      if (k0 + 1 < k1c) w1 = __builtin_nontemporal_load(row + (size_t)(k0 + 1) * 64 + lane);
      struct Rec { float4 p; float rc; unsigned w; };
      auto fetch = [&](unsigned word) { Rec r; r.w = word; const unsigned s = word == 0xffffu ? 0u : (word & 0x1fffu); r.p = A[s]; r.rc = Rc[s]; return r; };
      auto eval = [&](const Rec& c) {
        const bool val = c.w != 0xffffu;
        const float sx = (c.w & 0x2000u) ? sgx : 0.0f, sy = (c.w & 0x4000u) ? sgy : 0.0f, sz = (c.w & 0x8000u) ? sgz : 0.0f;
        const float dx = (c.p.x - ri.x) + sx, dy = (c.p.y - ri.y) + sy, dz = (c.p.z - ri.z) + sz;
        const float r2 = dx * dx + dy * dy + dz * dz;
        const bool ok = val && !(r2 < 1e-24f);
        const float rinv = ok ? __builtin_amdgcn_rsqf(ok ? r2 : 1.0f) : 0.0f;
        float dcn;
        p_cn_count(rinv, rci, c.rc, k1, &dcn);
        const float dEdr = ok ? (di + c.p.w) * dcn : 0.0f;
        const float fx = dEdr * (dx * rinv), fy = dEdr * (dy * rinv), fz = dEdr * (dz * rinv);
        Fx += (double)fx; Fy += (double)fy; Fz += (double)fz;
        if (want_virial) {
          V[0] += (double)(fx * dx); V[1] += (double)(fx * dy); V[2] += (double)(fx * dz);
          V[3] += (double)(fy * dy); V[4] += (double)(fy * dz); V[5] += (double)(fz * dz);
        }
      };
      // LDS reads run two trips ahead of the evaluation
      Rec c0 = fetch(w0.x & 0xffffu), c1 = fetch(w0.x >> 16);
      for (int k = k0; k < k1c; ++k) {
        if (k + 2 < k1c) w2 = __builtin_nontemporal_load(row + (size_t)(k + 2) * 64 + lane); else w2 = none;
        const Rec c2 = fetch(w0.y & 0xffffu);
        eval(c0);
        const Rec c3 = fetch(w0.y >> 16);
        eval(c1);
        c0 = fetch(w1.x & 0xffffu);
        eval(c2);
        c1 = fetch(w1.x >> 16);
        eval(c3);
        w0 = w1; w1 = w2;
      }
      V[6] = V[2]; V[7] = V[4]; V[8] = V[5]; V[5] = V[4]; V[4] = V[3]; V[3] = V[1];  // row-major 3x3 from {xx xy xz yy yz zz}
      { const double yz = V[5], zz = V[8]; V[5] = yz; V[7] = yz; V[8] = zz; }
      Fx = wsum(Fx); Fy = wsum(Fy); Fz = wsum(Fz);
      if (want_virial) {
#pragma unroll
        for (int k = 0; k < 9; ++k) V[k] = wsum(V[k]);
      }
      if (!last || q == 1) {
        if (active && lane == 0) { forces[3 * (size_t)io] = (float)Fx; forces[3 * (size_t)io + 1] = (float)Fy; forces[3 * (size_t)io + 2] = (float)Fz; }
        if (active && want_virial && lane < 9) { double v = V[0];
#pragma unroll
          for (int k = 1; k < 9; ++k) v = lane == k ? V[k] : v;
          v_atom[9 * (size_t)io + lane] = -0.5 * v; }
      } else {
        if (lane == 0) { part[wave][0] = Fx; part[wave][1] = Fy; part[wave][2] = Fz; }
        if (lane < 9) { double v = V[0];
#pragma unroll
          for (int k = 1; k < 9; ++k) v = lane == k ? V[k] : v;
          part[wave][3 + lane] = v; }
        __syncthreads();
        if (active && (wave % q) == 0 && lane < 12) {
          double s = 0.0;
          for (int g = 0; g < q; ++g) s += part[wave + g][lane];
          if (lane < 3) forces[3 * (size_t)io + lane] = (float)s;
          else if (want_virial) v_atom[9 * (size_t)io + lane - 3] = -0.5 * s;
        }
      }
    }
  }
}

extern "C" {
int probe_translate(const void* apos, int N, const int* idx, const int* ush, int M, int Mp, const int* cell_start, int nx, int ny, int nz, int R,
                    float lx, float ly, float lz, const float* rcov_tab, float k1, void* pk, float* cn, int* flags, int do_pack, void* stream) {
  Grid G{nx, ny, nz, R, lx, ly, lz};
  translate_kernel<<<(N + LS_WAVES - 1) / LS_WAVES, LS_WAVES * WAVE, 0, (hipStream_t)stream>>>((const float4*)apos, N, idx, ush, M, Mp, cell_start, G, rcov_tab, k1,
                                                                                           (unsigned short*)pk, cn, flags, do_pack);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
int probe_chain_lds(const void* srec, const void* sspec, const int* sidx, const int* cell_start, int nx, int ny, int nz, int R, float lx, float ly, float lz,
                    const void* pk, int Mp, const float* rcov_tab, float k1, int want_virial, float* forces, double* v_atom, int* work, int* flags,
                    int blocks, void* stream) {
  Grid G{nx, ny, nz, R, lx, ly, lz};
  (void)hipMemsetAsync(work, 0, sizeof(int), (hipStream_t)stream);
  chain_lds_kernel<<<blocks, LW * WAVE, 0, (hipStream_t)stream>>>((const float4*)srec, (const unsigned char*)sspec, sidx, cell_start, G, (const unsigned short*)pk, Mp,
                                                                rcov_tab, k1, want_virial, forces, v_atom, work, flags);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
int probe_chain_ref(const void* apos, const float* dEdCN, int N, const int* idx, const int* ush, int M, float lx, float ly, float lz, const float* rcov_tab,
                    float k1, int want_virial, float* forces, double* v_atom, void* stream) {
  Grid G{0, 0, 0, 0, lx, ly, lz};
  chain_ref_kernel<<<(N + 3) / 4, 256, 0, (hipStream_t)stream>>>((const float4*)apos, dEdCN, N, idx, ush, M, G, rcov_tab, k1, want_virial, forces, v_atom);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

static Grid2 make_grid2(int nx, int ny, int nz, int R, float lx, float ly, float lz, const signed char* dxlim) {
  Grid2 G; G.nx = nx; G.ny = ny; G.nz = nz; G.R = R; G.lx = lx; G.ly = ly; G.lz = lz;
  for (int a = 0; a < 5; ++a) for (int b = 0; b < 5; ++b) G.dxlim[a][b] = dxlim[a * 5 + b];
  return G;
}
int probe_translate2(const void* apos, int N, const int* idx, const int* ush, int M, int Mp, const int* cell_start, int nx, int ny, int nz, int R,
                     float lx, float ly, float lz, const signed char* dxlim /*[host] 5x5*/, const float* rcov_tab, float k1, void* pk, float* cn, int* flags,
                     int* rowflag, void* stream) {
  Grid2 G = make_grid2(nx, ny, nz, R, lx, ly, lz, dxlim);
  translate2_kernel<<<(N + LS_WAVES - 1) / LS_WAVES, LS_WAVES * WAVE, 0, (hipStream_t)stream>>>((const float4*)apos, N, idx, ush, M, Mp, cell_start, G, rcov_tab, k1,
                                                                                            (unsigned*)pk, cn, flags, rowflag);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
int probe_chain_lds2(const void* srec, const float* srcov, const int* sidx, const int* cell_start, int nx, int ny, int nz, int R, float lx, float ly, float lz,
                     const signed char* dxlim, const void* pk, int Mp, float k1, int want_virial, float* forces, double* v_atom, int* work, int* flags,
                     int blocks, void* stream) {
  Grid2 G = make_grid2(nx, ny, nz, R, lx, ly, lz, dxlim);
  (void)hipMemsetAsync(work, 0, sizeof(int), (hipStream_t)stream);
  chain_lds2_kernel<<<blocks, LW * WAVE, 0, (hipStream_t)stream>>>((const float4*)srec, srcov, sidx, cell_start, G, (const unsigned short*)pk, Mp, k1, want_virial,
                                                                 forces, v_atom, work, flags);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
}

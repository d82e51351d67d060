// TEST PROGRAM (never part of the product): runs the kernel bodies of csrc/fft_lds.h with SEVERAL host threads per block -- thread t plays
// (tid = t, nthreads = T), MI_FFT_SYNC() is a pthread barrier -- under ThreadSanitizer.  Two things are checked that the single-thread harness
// cannot see: (1) the tid-strided partition of every phase gives bit-identical results for any thread count, (2) no two threads touch the same
// LDS / global element between two barriers (a data race here is a missing __syncthreads on the GPU).  Built and run by
// tests/test_fft_lds_cpu.py::test_barrier_placement_under_thread_sanitizer:  g++ -fsanitize=thread -O1 -pthread.
#include <pthread.h>
#include <stdio.h>
#include <string.h>

#include <thread>
#include <vector>

static thread_local pthread_barrier_t* tl_barrier = nullptr;
static bool g_drop_barriers = false;  // negative control (argv[1] = "drop"): the detector must then report races
static void host_barrier() {
  if (tl_barrier && !g_drop_barriers) pthread_barrier_wait(tl_barrier);
}
#define MI_FFT_HOST_BARRIER host_barrier
#include "fft_lds.h"

using namespace mifft;

template <class F> static void run_block(int nth, F body) {
  if (nth == 1) { tl_barrier = nullptr; body(0, 1); return; }
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, nullptr, nth);
  std::vector<std::thread> th;
  for (int t = 0; t < nth; ++t) th.emplace_back([&, t] { tl_barrier = &bar; body(t, nth); });
  for (auto& x : th) x.join();
  pthread_barrier_destroy(&bar);
}

template <class R> static std::vector<R> solve(int B, int nx, int ny, int nz, int nch, int nth, const std::vector<R>& mesh) {
  const Geom g = make_geom(B, nx, ny, nz);
  std::vector<char> tab(tables_bytes<R>(g));
  run_block(nth, [&](int tid, int n) { tables_body<R>(tab.data(), g, tid, n); });
  const Tables<R> tb = tables_at<R>(tab.data(), g);
  const size_t ncol = (size_t)ny * g.P;
  std::vector<Cx<R>> spec((size_t)B * nx * ncol), conv((size_t)B * nch * nx * ncol);
  // one buffer per kernel, EXACTLY the size its launch requests: the AddressSanitizer build of this file (argv[1] = "asan" skips the threads)
  // then sees any body that walks past its LDS
  std::vector<char> lds_plane(plane_lds_bytes<R>(g)), lds_fwd_cols(fwd_cols_lds_bytes<R>(g)), lds_inv_cols(inv_cols_lds_bytes<R>(g));
  std::vector<R> out((size_t)B * nch * nx * ny * nz);
  std::vector<Cx<R>> nat((size_t)B * nx * ncol);  // the natural-order by-product is written too (scattered stores: one owner per element)
  R recip[18] = {0.61, 0.02, -0.03, 0.05, 0.57, 0.01, -0.02, 0.04, 0.52, 0.66, 0.0, 0.0, 0.0, 0.6, 0.0, 0.0, 0.0, 0.55};
  const R alpha[2] = {R(0.35), R(0.42)}, volume[2] = {R(1100.0), R(900.0)};
  for (int bx = 0; bx < B * nx; ++bx)
    run_block(nth, [&](int tid, int n) {
      fwd_plane_body<R>(mesh.data() + (size_t)bx * ny * nz, spec.data() + (size_t)bx * ncol, (Cx<R>*)lds_plane.data(), g, tb, tid, n);
    });
  const int blocks = (int)((ncol + MI_SOLVE_COLS - 1) / MI_SOLVE_COLS);
  for (int b = 0; b < B; ++b)
    for (int blk = 0; blk < blocks; ++blk)
      run_block(nth, [&](int tid, int n) {
        fwd_cols_body<R>(spec.data() + (size_t)b * nx * ncol, (Cx<R>*)lds_fwd_cols.data(), g, tb, recip + 9 * b, alpha[b], volume[b], 4, blk * MI_SOLVE_COLS, tid, n,
                              nat.data() + (size_t)b * nx * ncol);
      });
  for (int b = 0; b < B; ++b)
    for (int ch = 0; ch < nch; ++ch)
      for (int blk = 0; blk < blocks; ++blk)
        run_block(nth, [&](int tid, int n) {
          inv_cols_body<R>(spec.data() + (size_t)b * nx * ncol, conv.data() + ((size_t)b * nch + ch) * nx * ncol, (Cx<R>*)lds_inv_cols.data(), g, tb, recip + 9 * b, ch,
                                blk * MI_SOLVE_COLS, tid, n);
        });
  for (int p = 0; p < B * nch * nx; ++p)
    run_block(nth, [&](int tid, int n) { inv_plane_body<R>(conv.data() + (size_t)p * ncol, out.data() + (size_t)p * ny * nz, (Cx<R>*)lds_plane.data(), g, tb, tid, n); });
  // the PLAIN column bodies (mi_fft_lds: transforms on their own): natural-order spectrum out of the forward one, and back in through the inverse
  // one; their results join `out`, so the thread-count comparison and the bounds run cover them too
  std::vector<Cx<R>> nat2((size_t)B * nx * ncol), work((size_t)B * nx * ncol);
  std::vector<char> lds_plane_nat(plane_lds_bytes_nat<R>(g));
  for (int bx = 0; bx < B * nx; ++bx)
    run_block(nth, [&](int tid, int n) {
      fwd_plane_body<R, true, true>(mesh.data() + (size_t)bx * ny * nz, spec.data() + (size_t)bx * ncol, (Cx<R>*)lds_plane_nat.data(), g, tb, tid, n);
    });
  for (int b = 0; b < B; ++b)
    for (int blk = 0; blk < blocks; ++blk)
      run_block(nth, [&](int tid, int n) {
        fwd_cols_body<R, true, true>(spec.data() + (size_t)b * nx * ncol, (Cx<R>*)lds_fwd_cols.data(), g, tb, nullptr, R(1), R(1), 1, blk * MI_SOLVE_COLS, tid, n,
                                     nat2.data() + (size_t)b * nx * ncol);
      });
  for (int b = 0; b < B; ++b)
    for (int blk = 0; blk < blocks; ++blk)
      run_block(nth, [&](int tid, int n) {
        inv_cols_body<R, true, true>(nat2.data() + (size_t)b * nx * ncol, work.data() + (size_t)b * nx * ncol, (Cx<R>*)lds_inv_cols.data(), g, tb, nullptr, 0,
                                     blk * MI_SOLVE_COLS, tid, n);
      });
  std::vector<R> back((size_t)B * nx * ny * nz);
  for (int p = 0; p < B * nx; ++p)
    run_block(nth, [&](int tid, int n) { inv_plane_body<R, true, true>(work.data() + (size_t)p * ncol, back.data() + (size_t)p * ny * nz, (Cx<R>*)lds_plane_nat.data(), g, tb, tid, n); });
  out.insert(out.end(), back.begin(), back.end());
  for (const auto& v : nat2) { out.push_back(v.re); out.push_back(v.im); }
  return out;
}

template <class R> static int check(int nx, int ny, int nz, bool asan, const char* what) {
  const int B = 2, nch = 4;
  int bad = 0;
  std::vector<R> mesh((size_t)B * nx * ny * nz);
  unsigned long long z = 88172645463325252ull + nx * 131 + ny * 17 + nz;
  for (auto& v : mesh) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; v = (R)((double)(z % 20001) / 10000.0 - 1.0); }
  const std::vector<R> ref = solve<R>(B, nx, ny, nz, nch, 1, mesh);
  if (asan) { printf("%s mesh %dx%dx%d single thread: in bounds\n", what, nx, ny, nz); return 0; }
  for (int nth : {3, 16}) {
    if (g_drop_barriers && nth != 3) continue;
    if (sizeof(R) == 4 && nth != 16) continue;  // fp32: one thread count is enough for the partition, the layout is what differs
    const std::vector<R> got = solve<R>(B, nx, ny, nz, nch, nth, mesh);
    const bool same = memcmp(ref.data(), got.data(), ref.size() * sizeof(R)) == 0;
    printf("%s mesh %dx%dx%d threads %d: %s\n", what, nx, ny, nz, nth, same ? "bit-identical" : "DIFFERENT");
    bad += !same;
  }
  return bad;
}

int main(int argc, char** argv) {
  g_drop_barriers = argc > 1 && strcmp(argv[1], "drop") == 0;
  const bool asan = argc > 1 && strcmp(argv[1], "asan") == 0;  // bounds run: single-threaded, every buffer exactly sized
  const int shapes[][3] = {{16, 8, 32}, {128, 8, 8}, {8, 32, 16}};
  int bad = 0;
  for (const auto& s : shapes) bad += check<double>(s[0], s[1], s[2], asan, "fp64");
  for (const auto& s : shapes) bad += check<float>(s[0], s[1], s[2], asan, "fp32");
  return bad ? 1 : 0;
}

// Probe (round 5): device buffers whose virtual -> physical map is chosen by the caller at chunk granularity (HIP virtual memory management):
// one virtual range backed by n physical chunks created one after the other and mapped in order, reversed, or in a pseudo-random permutation.
// Question behind it: the 40-Bohr fill is 2x slower into physically CONTIGUOUS buffers than into ordinary hipMalloc ones and
// bimodal across hipMalloc buffers -- is the fast state "physically scattered", and can it be had on purpose?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

extern "C" {

size_t vmm_granularity(int recommended) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  int dev = 0;
  hipGetDevice(&dev);
  prop.location.id = dev;
  size_t g = 0;
  if (hipMemGetAllocationGranularity(&g, &prop, recommended ? hipMemAllocationGranularityRecommended : hipMemAllocationGranularityMinimum) != hipSuccess) return 0;
  return g;
}

// mode 0: chunk k of the virtual range = k-th chunk created; 1: reversed; 2: random permutation (seed); 3: stride permutation (k * 7919 mod n)
void* vmm_alloc(size_t bytes, size_t chunk, int mode, unsigned seed) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  int dev = 0;
  hipGetDevice(&dev);
  prop.location.id = dev;
  const size_t n = (bytes + chunk - 1) / chunk, total = n * chunk;
  void* base = nullptr;
  if (hipMemAddressReserve(&base, total, chunk, nullptr, 0) != hipSuccess) { fprintf(stderr, "reserve failed\n"); return nullptr; }
  std::vector<hipMemGenericAllocationHandle_t> h(n);
  for (size_t k = 0; k < n; ++k)
    if (hipMemCreate(&h[k], chunk, &prop, 0) != hipSuccess) { fprintf(stderr, "create %zu failed\n", k); return nullptr; }
  std::vector<size_t> perm(n);
  for (size_t k = 0; k < n; ++k) perm[k] = k;
  if (mode == 1) for (size_t k = 0; k < n; ++k) perm[k] = n - 1 - k;
  if (mode == 2) { srand(seed); for (size_t k = n - 1; k > 0; --k) { size_t j = (size_t)rand() % (k + 1); size_t t = perm[k]; perm[k] = perm[j]; perm[j] = t; } }
  if (mode == 3) { size_t s = 7919 % n; if (s == 0) s = 1; while (true) { size_t a = s, b = n; while (b) { size_t t = a % b; a = b; b = t; } if (a == 1) break; ++s; } for (size_t k = 0; k < n; ++k) perm[k] = (k * s) % n; }
  for (size_t k = 0; k < n; ++k)
    if (hipMemMap((char*)base + k * chunk, chunk, 0, h[perm[k]], 0) != hipSuccess) { fprintf(stderr, "map %zu failed\n", k); return nullptr; }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if (hipMemSetAccess(base, total, &acc, 1) != hipSuccess) { fprintf(stderr, "set access failed\n"); return nullptr; }
  for (size_t k = 0; k < n; ++k) hipMemRelease(h[k]);  // the mapping keeps the memory alive
  return base;
}

}  // extern "C"

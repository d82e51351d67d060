// capi.cpp -- library-level entry points of libnvalchemiops_hip.so: version, per-thread error string and the optional
// per-kernel HIP-event timing used by bench.py (roofline.achieved is measured live with these events on the launch stream).
#include <hip/hip_runtime_api.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/nvalchemiops_hip.h"

static thread_local char g_err[512] = "";

void mi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {
struct Rec { const char* name; hipEvent_t a, b; };
std::mutex g_mu;
bool g_timing = false;
std::string g_only;  // when not empty: only brackets of this name are recorded (mi_timing_select)
bool g_skipped = false;  // the bracket opened last was filtered out: its mi_timing_end must not close an older record
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t take_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

// called around kernel launches by the MI_TIMED macro (common.h); no-ops unless timing is enabled
void mi_timing_begin(const char* name, void* stream) {
  if (!g_timing) return;
  std::lock_guard<std::mutex> lk(g_mu);
  g_skipped = !g_only.empty() && g_only != name;
  if (g_skipped) return;
  Rec r{name, take_event(), take_event()};
  (void)hipEventRecord(r.a, (hipStream_t)stream);
  g_recs.push_back(r);
}
void mi_timing_end(void* stream) {
  if (!g_timing) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_skipped) { g_skipped = false; return; }
  if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().b, (hipStream_t)stream);
}

extern "C" {
int mi_version(void) { return 1; }
const char* mi_last_error(void) { return g_err; }

int mi_timing_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_timing = on != 0;
  return MI_OK;
}

// Restricts the recording to the brackets called `name` (NULL or "": every bracket again).  An event record is a packet of its own in the
// stream: ~5 us of bubble on either side of a kernel.  A benchmark that needs ONE kernel's duration inside its timed region selects it and
// leaves the other ~25 launches of a step back to back.
int mi_timing_select(const char* name) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_only = name ? name : "";
  return MI_OK;
}

// Writes one line per kernel: "<name> <launches> <total_ms>\n"; waits for the recorded events; clears the records.
int mi_timing_report(char* buf, int cap) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::map<std::string, std::pair<int, double>> acc;
  for (auto& r : g_recs) {
    float ms = 0.0f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      auto& s = acc[r.name];
      s.first += 1;
      s.second += ms;
    }
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  int off = 0;
  if (buf && cap > 0) buf[0] = 0;
  for (auto& kv : acc) {
    int n = snprintf(buf + off, cap > off ? cap - off : 0, "%s %d %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
    if (n < 0 || off + n >= cap) break;
    off += n;
  }
  return MI_OK;
}

// Same records, per-launch statistics: "<name> <launches> <total_ms> <median_ms> <min_ms> <max_ms>\n"; clears the records.
int mi_timing_report_stats(char* buf, int cap) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::map<std::string, std::vector<double>> acc;
  for (auto& r : g_recs) {
    float ms = 0.0f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) acc[r.name].push_back(ms);
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  int off = 0;
  if (buf && cap > 0) buf[0] = 0;
  for (auto& kv : acc) {
    std::vector<double>& v = kv.second;
    std::sort(v.begin(), v.end());
    double tot = 0.0;
    for (double x : v) tot += x;
    const size_t n = v.size();
    const double med = n % 2 ? v[n / 2] : 0.5 * (v[n / 2 - 1] + v[n / 2]);
    int w = snprintf(buf + off, cap > off ? cap - off : 0, "%s %zu %.6f %.6f %.6f %.6f\n", kv.first.c_str(), n, tot, med, v.front(), v.back());
    if (w < 0 || off + w >= cap) break;
    off += w;
  }
  return MI_OK;
}
}

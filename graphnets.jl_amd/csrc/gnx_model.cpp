// gnx_model: a list of GNBlock / GNCore layers run back to back and replayed as ONE hipGraph (include/gnx.h).
// Host code only: every launch goes through gnx_block_forward / gnx_core_forward.
#include <cstdlib>
#include <cstring>
#include <memory>

#include "gnx_internal.h"

struct gnx_model {
  const gnx_graphs* h = nullptr;
  int64_t R = 1;
  struct Layer {
    int kind = 0;
    gnx_block_params block{};
    gnx_core_params core{};
    int in[3] = {0, 0, 0}, out[3] = {0, 0, 0};
    float* y[3] = {nullptr, nullptr, nullptr};  // library-owned outputs of this layer (not for the last layer)
    void* ws = nullptr;
    size_t ws_bytes = 0;
  };
  std::vector<Layer> layers;
  std::vector<void*> owned;  // device allocations
  std::vector<gnx_prepared*> prepared;  // parameters prepared by the model itself (layers whose descriptor carried none)
  // captured graph and the pointers it was captured with
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipStream_t cap_stream = nullptr;
  const void* cap_ptrs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  uint32_t cap_flags = 0;
  bool small = false;  // the captured forward is a handful of kernels: launched one by one (see gnx_model_forward)
  std::mutex mu;
};

namespace {

using namespace gnx;

void drop_graph(gnx_model* m) {
  if (m->exec) { (void)hipGraphExecDestroy(m->exec); m->exec = nullptr; }
  if (m->graph) { (void)hipGraphDestroy(m->graph); m->graph = nullptr; }
}

int32_t run_layers(gnx_model* m, const float* ef, const float* nf, const float* gf, float* ef_out, float* nf_out, float* gf_out, uint32_t flags,
                   hipStream_t s) {
  const float* x[3] = {ef, nf, gf};
  const size_t n = m->layers.size();
  for (size_t i = 0; i < n; ++i) {
    gnx_model::Layer& L = m->layers[i];
    float* y[3];
    for (int t = 0; t < 3; ++t) y[t] = i + 1 == n ? (t == 0 ? ef_out : (t == 1 ? nf_out : gf_out)) : L.y[t];
    const float* xi[3];
    for (int t = 0; t < 3; ++t) xi[t] = L.in[t] > 0 ? x[t] : nullptr;
    int32_t rc;
    if (L.kind == GNX_LAYER_BLOCK)
      rc = gnx_block_forward(m->h, &L.block, xi[0], xi[1], xi[2], m->R, y[0], y[1], y[2], L.ws, L.ws_bytes, flags, s);
    else
      rc = gnx_core_forward(m->h, &L.core, xi[0], xi[1], xi[2], m->R, y[0], y[1], y[2], L.ws, L.ws_bytes, flags, s);
    if (rc) return rc;
    for (int t = 0; t < 3; ++t) x[t] = L.out[t] > 0 ? y[t] : nullptr;
  }
  return GNX_OK;
}

}  // namespace

extern "C" {

int32_t gnx_model_create(const gnx_graphs* h, const gnx_layer* layers, int32_t n_layers, int64_t R, gnx_model** out) {
  if (!h || !layers || !out || n_layers <= 0) return fail(GNX_ERR_INVALID_ARG, "gnx_model_create: NULL argument or no layers");
  if (R <= 0 || (R > 1 && h->G != 1)) return fail(GNX_ERR_INVALID_ARG, "n_replicas > 1 needs a single-graph handle (shared adjacency, batch.jl:66)");
  *out = nullptr;
  std::unique_ptr<gnx_model> m(new gnx_model());
  m->h = h;
  m->R = R;
  m->layers.resize((size_t)n_layers);
  // every error path below frees what the model holds so far (the early returns used to leak the prepared objects)
  struct Cleanup {
    gnx_model* m;
    ~Cleanup() {
      if (!m) return;
      for (void* q : m->owned) (void)hipFree(q);
      for (gnx_prepared* x : m->prepared) gnx_prepared_destroy(x);
      m->owned.clear(); m->prepared.clear();
    }
  } cleanup{m.get()};
  for (int i = 0; i < n_layers; ++i) {
    gnx_model::Layer& L = m->layers[(size_t)i];
    if (!layers[i].params) return fail(GNX_ERR_INVALID_ARG, "gnx_model_create: layer params is NULL");
    L.kind = layers[i].kind;
    if (L.kind == GNX_LAYER_BLOCK) {
      L.block = *static_cast<const gnx_block_params*>(layers[i].params);
    } else if (L.kind == GNX_LAYER_CORE) {
      L.core = *static_cast<const gnx_core_params*>(layers[i].params);
      L.block = L.core.block;
    } else {
      return fail(GNX_ERR_INVALID_ARG, "gnx_model_create: unknown layer kind");
    }
    // parameters prepared once, here, unless the caller brought its own (the weights are device pointers that stay the caller's)
    {
      gnx_prepared* q = nullptr;
      int32_t rcp = GNX_OK;
      if (L.kind == GNX_LAYER_BLOCK && !L.block.prepared) { rcp = gnx_block_prepare(&L.block, nullptr, &q); L.block.prepared = q; }
      else if (L.kind == GNX_LAYER_CORE && !L.core.prepared) { rcp = gnx_core_prepare(&L.core, nullptr, &q); L.core.prepared = q; }
      if (q) m->prepared.push_back(q);
      if (rcp) return rcp;
    }
    const int in[3] = {L.block.de, L.block.dn, L.block.dg}, o[3] = {L.block.oe, L.block.on, L.block.og};
    for (int t = 0; t < 3; ++t) { L.in[t] = in[t]; L.out[t] = o[t]; }
    if (i > 0)
      for (int t = 0; t < 3; ++t)
        if (m->layers[(size_t)i - 1].out[t] != L.in[t]) return fail(GNX_ERR_DIMS, "gnx_model_create: output widths of a layer differ from the next layer's input widths");
  }
  // intermediates + workspaces (this also compiles run-time specialised kernels: gnx_block_workspace_bytes)
  auto alloc = [&](size_t bytes, void** p) -> int32_t {
    *p = nullptr;
    if (bytes == 0) return GNX_OK;
    GNX_HIP(hipMalloc(p, bytes));
    m->owned.push_back(*p);
    return GNX_OK;
  };
  const size_t rows[3] = {(size_t)R * h->E, (size_t)R * h->N, (size_t)R * h->G};
  for (int i = 0; i < n_layers; ++i) {
    gnx_model::Layer& L = m->layers[(size_t)i];
    L.ws_bytes = L.kind == GNX_LAYER_BLOCK ? gnx_block_workspace_bytes(h, &L.block, R) : gnx_core_workspace_bytes(h, &L.core, R);
    if (L.ws_bytes == 0) return fail(GNX_ERR_DIMS, "gnx_model_create: a layer's parameters were rejected (widths)");
    int32_t rc = alloc(L.ws_bytes, &L.ws);
    if (!rc && i + 1 < n_layers)
      for (int t = 0; t < 3 && !rc; ++t) rc = alloc(sizeof(float) * rows[t] * (size_t)L.out[t], reinterpret_cast<void**>(&L.y[t]));
    if (rc) return rc;
  }
  // the planes were made by launches on the NULL stream: they are complete before any forward on ANY stream (a non-blocking one included) reads them
  if (!m->prepared.empty()) GNX_HIP(hipStreamSynchronize(nullptr));
  cleanup.m = nullptr;
  *out = m.release();
  return GNX_OK;
}

int32_t gnx_model_destroy(gnx_model* m) {
  if (!m) return GNX_OK;
  drop_graph(m);
  if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
  for (void* q : m->owned) (void)hipFree(q);
  for (gnx_prepared* q : m->prepared) gnx_prepared_destroy(q);
  delete m;
  return GNX_OK;
}

int32_t gnx_model_refresh_weights(gnx_model* m, void* stream) {
  if (!m) return fail(GNX_ERR_INVALID_ARG, "NULL model");
  std::lock_guard<std::mutex> lk(m->mu);
  for (gnx_prepared* q : m->prepared)
    if (const int32_t rc = gnx_prepared_refresh(q, stream)) return rc;
  return GNX_OK;  // (the captured forward reads the same plane buffers: no re-capture)
}

int32_t gnx_model_out_dims(const gnx_model* m, int32_t dims[3]) {
  if (!m || !dims) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  for (int t = 0; t < 3; ++t) dims[t] = m->layers.back().out[t];
  return GNX_OK;
}

int32_t gnx_model_forward(gnx_model* m, const float* ef, const float* nf, const float* gf, float* ef_out, float* nf_out, float* gf_out, uint32_t flags,
                          void* stream) {
  if (!m) return fail(GNX_ERR_INVALID_ARG, "NULL model");
  hipStream_t s = (hipStream_t)stream;
  std::lock_guard<std::mutex> lk(m->mu);
  bool wide = false;  // (any layer at matrix-core widths: the whole forward — its replayed graph — takes one turn on the device)
  for (const auto& L : m->layers) wide = wide || matrix_core_widths(L.kind == GNX_LAYER_CORE ? L.core.block : L.block);
  DeviceTurn turn(s, wide);
  const uint32_t lflags = flags & ~GNX_FLAG_NO_GRAPH;
  if (flags & GNX_FLAG_NO_GRAPH) return run_layers(m, ef, nf, gf, ef_out, nf_out, gf_out, lflags, s);
  const void* ptrs[6] = {ef, nf, gf, ef_out, nf_out, gf_out};
  if (!m->exec || std::memcmp(ptrs, m->cap_ptrs, sizeof ptrs) != 0 || m->cap_flags != lflags) {
    drop_graph(m);
    // one eager pass first: argument errors surface here (outside a capture), code objects are loaded
    int32_t rc = run_layers(m, ef, nf, gf, ef_out, nf_out, gf_out, lflags, s);
    if (rc) return rc;
    if (!m->cap_stream) GNX_HIP(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
    GNX_HIP(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal));
    rc = run_layers(m, ef, nf, gf, ef_out, nf_out, gf_out, lflags, m->cap_stream);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(m->cap_stream, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
    m->graph = g;
    // A replay costs ~5 us of launch latency per hipGraphLaunch on top of its kernels, a plain launch ~2-3 us of host time that the GPU does
    // not wait for while the host stays ahead: a forward of <= 4 kernels (one narrow GNBlock: 2) is faster launched one by one — BASELINE
    // configs[1] from C: 24.9 us/step eager, 29.9 us/step as a one-forward graph, 25.0 as 200 forwards in ONE graph (tests/c/abi_bench.c)
    size_t n_nodes = 0;
    static const size_t min_nodes = getenv("GNX_MODEL_GRAPH_MIN_NODES") ? (size_t)atoi(getenv("GNX_MODEL_GRAPH_MIN_NODES")) : 5;
    m->small = hipGraphGetNodes(m->graph, nullptr, &n_nodes) == hipSuccess && n_nodes < min_nodes;
    GNX_HIP(hipGraphInstantiate(&m->exec, m->graph, nullptr, nullptr, 0));
    std::memcpy(m->cap_ptrs, ptrs, sizeof ptrs);
    m->cap_flags = lflags;
    return GNX_OK;  // the eager pass above already produced this call's outputs
  }
  if (m->small) return run_layers(m, ef, nf, gf, ef_out, nf_out, gf_out, lflags, s);
  GNX_HIP(hipGraphLaunch(m->exec, s));
  return GNX_OK;
}

}  // extern "C"

// gnx_block_prepare / gnx_core_prepare (include/gnx.h): the weight blocks of a layer in the forms the six-term kernels stage, made once.
//
// The reference moves a model to the device once and then calls it (`model |> device`, examples/sort/sort.jl:29,89; Functors.@functor GNBlock,
// src/gnblock.jl:8).  Until round 4 every forward re-split and re-laid-out its weights — nine *_prep launches, ~45 us, per forward of BASELINE
// configs[3] — "because the weights are the caller's and may change between calls".  A prepared object is the caller saying when they change:
// made after the upload, refreshed (gnx_prepared_refresh) after an optimiser step, destroyed before the weights are freed.
#include <algorithm>
#include <mutex>
#include <vector>

#include "gnx_internal.h"

namespace gnx {
int32_t launch_edge_x6_prep(const float* We, int ldw, void* scratch, hipStream_t s, int n_out, const float* ln_gamma, const float* ln_beta);  // gnx_edge_x6.hip
size_t edge_x6_fold_scratch_bytes();
int32_t launch_proj_x6_prep(const float* Ws, const float* Wd, int ldw, void* scratch, hipStream_t s);  // gnx_edge_x6.hip
size_t proj_x6_scratch_bytes();
int32_t launch_ffn_x6_prep(const float* W1, const float* W2, int d, void* scratch, hipStream_t s, const float* ln_gamma, const float* ln_beta,
                           const float* b1);  // gnx_ffn_x6.hip
size_t ffn_x6_fold_scratch_bytes(int d);
int32_t launch_edge_enc_prep(const float* We, int ldw, void* scratch, hipStream_t s);                  // gnx_edge_x6.hip
int32_t launch_node_x6_prep(const float* Wn, int ldw, void* scratch, hipStream_t s);                   // gnx_edge_x6.hip
size_t node_x6_scratch_bytes();
size_t edge_enc_scratch_bytes();
size_t ffn_x6_scratch_bytes(int d);
}  // namespace gnx

struct gnx_prepared {
  struct Entry {
    gnx::PreparedKind kind;
    const void* w0;  // what a launcher asks with: the weight the planes were made from, and (a second weight | the gamma of a folded LayerNorm)
    const void* w1;
    const void* x0;  // further sources of a folded entry — EDGE: beta; FFN: W2, beta, b1
    const void* x1;
    const void* x2;
    bool folded;     // the LayerNorm in front of the weight block is part of the planes (scaled rows + the constant vector behind them)
    int32_t n;       // EDGE: output width; PROJ: row distance of the weight matrix; FFN: width d
    int32_t ldw;     // EDGE: row distance
    void* planes;
    size_t bytes;
  };
  int device = 0;
  std::vector<Entry> entries;
};

namespace gnx {

static thread_local const gnx_prepared* tl_prepared = nullptr;

PreparedScope::PreparedScope(const gnx_prepared* q) : prev(tl_prepared) { tl_prepared = q; }
PreparedScope::~PreparedScope() { tl_prepared = prev; }

const void* prepared_planes(PreparedKind kind, const void* w0, const void* w1, int32_t n) {
  const gnx_prepared* q = tl_prepared;
  if (!q) return nullptr;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != q->device) { (void)hipGetLastError(); return nullptr; }
  for (const auto& e : q->entries)
    if (e.kind == kind && e.w0 == w0 && e.w1 == w1 && e.n == n) return e.planes;
  return nullptr;
}

namespace {

int32_t run_entry(const gnx_prepared::Entry& e, hipStream_t s) {
  switch (e.kind) {
    case PREP_EDGE:
      return e.folded ? launch_edge_x6_prep(static_cast<const float*>(e.w0), e.ldw, e.planes, s, e.n, static_cast<const float*>(e.w1), static_cast<const float*>(e.x0))
                      : launch_edge_x6_prep(static_cast<const float*>(e.w0), e.ldw, e.planes, s, e.n, nullptr, nullptr);
    case PREP_PROJ: return launch_proj_x6_prep(static_cast<const float*>(e.w0), static_cast<const float*>(e.w1), e.n, e.planes, s);
    case PREP_FFN:
      return e.folded ? launch_ffn_x6_prep(static_cast<const float*>(e.w0), static_cast<const float*>(e.x0), e.n, e.planes, s, static_cast<const float*>(e.w1),
                                           static_cast<const float*>(e.x1), static_cast<const float*>(e.x2))
                      : launch_ffn_x6_prep(static_cast<const float*>(e.w0), static_cast<const float*>(e.w1), e.n, e.planes, s, nullptr, nullptr, nullptr);
    case PREP_ENC: return launch_edge_enc_prep(static_cast<const float*>(e.w0), e.n, e.planes, s);
    case PREP_NODE: return launch_node_x6_prep(static_cast<const float*>(e.w0), e.n, e.planes, s);
  }
  return fail(GNX_ERR_INVALID_ARG, "prepared parameters: unknown entry");
}

int32_t add_entry(gnx_prepared* q, PreparedKind kind, const void* w0, const void* w1, int32_t n, int32_t ldw, size_t bytes) {
  for (const auto& e : q->entries)
    if (e.kind == kind && e.w0 == w0 && e.w1 == w1 && e.n == n) return GNX_OK;  // (two layers sharing a weight block)
  gnx_prepared::Entry e{kind, w0, w1, nullptr, nullptr, nullptr, false, n, ldw, nullptr, bytes};
  GNX_HIP(hipMalloc(&e.planes, bytes));
  q->entries.push_back(e);
  return GNX_OK;
}

// the planes of a weight block with the LayerNorm in front of it folded in (asked for with the weight and the gamma pointer)
int32_t add_folded(gnx_prepared* q, PreparedKind kind, const void* w, const void* gamma, const void* x0, const void* x1, const void* x2, int32_t n, int32_t ldw,
                   size_t bytes) {
  for (const auto& e : q->entries)
    if (e.kind == kind && e.w0 == w && e.w1 == gamma && e.n == n) return GNX_OK;
  gnx_prepared::Entry e{kind, w, gamma, x0, x1, x2, true, n, ldw, nullptr, bytes};
  GNX_HIP(hipMalloc(&e.planes, bytes));
  q->entries.push_back(e);
  return GNX_OK;
}

// the matrix-core forms of a block's edge function: the ef rows' block (128 -> 128 or 128 -> at most 32) and the two node-projection blocks
int32_t add_block(gnx_prepared* q, const gnx_block_params& p) {
  const float* We = p.edgefn.weight;
  if (We && p.de == 10 && p.dn == 5 && p.oe == 128) return add_entry(q, PREP_ENC, We, nullptr, p.oe, p.oe, edge_enc_scratch_bytes());  // the encoder form
  if (!We || p.de != 128 || p.dn <= 0 || p.oe <= 0) return GNX_OK;  // (no six-term form: nothing to prepare, the forward runs as before)
  int32_t rc = GNX_OK;
  if (p.oe == 128 || p.oe <= 32) rc = add_entry(q, PREP_EDGE, We, nullptr, p.oe, p.oe, sizeof(uint16_t) * 3 * 128 * (size_t)((p.oe + 31) / 32 * 32));
  if (rc == GNX_OK && p.dn == 64 && p.oe == 128)
    rc = add_entry(q, PREP_PROJ, We + (size_t)p.de * p.oe, We + (size_t)(p.de + p.dn) * p.oe, p.oe, p.oe, proj_x6_scratch_bytes());
  if (rc == GNX_OK && p.dn == 64 && p.oe == 128 && p.on == 64 && p.nodefn.weight)  // the node update at core widths (k_node_x6)
    rc = add_entry(q, PREP_NODE, p.nodefn.weight, nullptr, p.on, p.on, node_x6_scratch_bytes());
  return rc;
}

int32_t finish(gnx_prepared* q, int32_t rc, hipStream_t s, gnx_prepared** out) {
  for (size_t i = 0; rc == GNX_OK && i < q->entries.size(); ++i) rc = run_entry(q->entries[i], s);
  if (rc) { gnx_prepared_destroy(q); return rc; }
  *out = q;
  return GNX_OK;
}

}  // namespace
}  // namespace gnx

using namespace gnx;

extern "C" {

int32_t gnx_block_prepare(const gnx_block_params* p, void* stream, gnx_prepared** out) {
  if (!p || !out) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  *out = nullptr;
  gnx_prepared* q = new gnx_prepared();
  GNX_HIP(hipGetDevice(&q->device));
  return finish(q, add_block(q, *p), (hipStream_t)stream, out);
}

int32_t gnx_core_prepare(const gnx_core_params* p, void* stream, gnx_prepared** out) {
  if (!p || !out) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  *out = nullptr;
  gnx_prepared* q = new gnx_prepared();
  GNX_HIP(hipGetDevice(&q->device));
  int32_t rc = add_block(q, p->block);
  const int d[3] = {p->block.de, p->block.dn, p->block.dg};
  for (int t = 0; t < 3 && rc == GNX_OK; ++t)  // the FeedForwards whose width has the six-term kernel
    if ((d[t] == 128 || d[t] == 64) && p->ff[t].fc1.weight && p->ff[t].fc2.weight)
      rc = add_entry(q, PREP_FFN, p->ff[t].fc1.weight, p->ff[t].fc2.weight, d[t], 0, ffn_x6_scratch_bytes(d[t]));
  // the one-launch form of a core's edge rows (128-wide edges): both weight blocks with gn1 / gn2 folded in
  const gnx_block_params& b = p->block;
  if (rc == GNX_OK && b.de == 128 && b.oe == 128 && b.dn > 0 && b.edgefn.weight && p->ln1[0].gamma && p->ln1[0].beta && p->ln2[0].gamma && p->ln2[0].beta &&
      p->ff[0].fc1.weight && p->ff[0].fc2.weight) {
    rc = add_folded(q, PREP_EDGE, b.edgefn.weight, p->ln1[0].gamma, p->ln1[0].beta, nullptr, nullptr, 128, b.oe, edge_x6_fold_scratch_bytes());
    if (rc == GNX_OK)
      rc = add_folded(q, PREP_FFN, p->ff[0].fc1.weight, p->ln2[0].gamma, p->ff[0].fc2.weight, p->ln2[0].beta, p->ff[0].fc1.bias, 128, 0,
                      ffn_x6_fold_scratch_bytes(128));
  }
  return finish(q, rc, (hipStream_t)stream, out);
}

int32_t gnx_prepared_refresh(gnx_prepared* q, void* stream) {
  if (!q) return GNX_OK;
  int dev = -1;
  GNX_HIP(hipGetDevice(&dev));
  if (dev != q->device) return fail(GNX_ERR_INVALID_ARG, "prepared parameters live on another device");
  for (const auto& e : q->entries)
    if (const int32_t rc = run_entry(e, (hipStream_t)stream)) return rc;
  return GNX_OK;
}

int32_t gnx_prepared_destroy(gnx_prepared* q) {
  if (!q) return GNX_OK;
  for (auto& e : q->entries) (void)hipFree(e.planes);
  delete q;
  return GNX_OK;
}

int64_t gnx_prepared_bytes(const gnx_prepared* q) {
  int64_t b = 0;
  if (q) for (const auto& e : q->entries) b += (int64_t)e.bytes;
  return b;
}

}  // extern "C"

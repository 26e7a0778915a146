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
int32_t launch_edge_x6_prep(const float* We, int ldw, void* scratch, hipStream_t s, int n_out);      // gnx_edge_x6.hip
int32_t launch_proj_x6_prep(const float* Ws, const float* Wd, int ldw, void* scratch, hipStream_t s);  // gnx_edge_x6.hip
size_t proj_x6_scratch_bytes();
int32_t launch_ffn_x6_prep(const float* W1, const float* W2, int d, void* scratch, hipStream_t s);     // gnx_ffn_x6.hip
int32_t launch_edge_enc_prep(const float* We, int ldw, void* scratch, hipStream_t s);                  // gnx_edge_x6.hip
size_t edge_enc_scratch_bytes();
size_t ffn_x6_scratch_bytes(int d);
}  // namespace gnx

struct gnx_prepared {
  struct Entry {
    gnx::PreparedKind kind;
    const void* w0;  // the weight pointers the planes were made from: what a launcher asks with
    const void* w1;
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
    case PREP_EDGE: return launch_edge_x6_prep(static_cast<const float*>(e.w0), e.ldw, e.planes, s, e.n);
    case PREP_PROJ: return launch_proj_x6_prep(static_cast<const float*>(e.w0), static_cast<const float*>(e.w1), e.n, e.planes, s);
    case PREP_FFN: return launch_ffn_x6_prep(static_cast<const float*>(e.w0), static_cast<const float*>(e.w1), e.n, e.planes, s);
    case PREP_ENC: return launch_edge_enc_prep(static_cast<const float*>(e.w0), e.n, e.planes, s);
  }
  return fail(GNX_ERR_INVALID_ARG, "prepared parameters: unknown entry");
}

int32_t add_entry(gnx_prepared* q, PreparedKind kind, const void* w0, const void* w1, int32_t n, int32_t ldw, size_t bytes) {
  for (const auto& e : q->entries)
    if (e.kind == kind && e.w0 == w0 && e.w1 == w1 && e.n == n) return GNX_OK;  // (two layers sharing a weight block)
  gnx_prepared::Entry e{kind, w0, w1, n, ldw, nullptr, bytes};
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

// C-ABI entry points of the forward path: argument validation (mirroring the reference's assertions), workspace
// carving and kernel-path selection.  No allocation and no synchronisation happens here, so every call is
// asynchronous on the caller's stream and can be captured into a hipGraph.
#include <algorithm>
#include <cstdlib>

#include "gnx_device.h"

namespace gnx {

// implemented in gnx_generic.hip / gnx_narrow.hip / gnx_wide.hip
int32_t launch_block_generic(const BlockArgs& a, int64_t R, int tile_n_cap, hipStream_t s, int phase);
int32_t launch_graph(const BlockArgs& a, int64_t R, hipStream_t s);
int32_t launch_layernorm2(const float* x, size_t rows, int d, const gnx_layernorm& l1, const gnx_layernorm& l2, float eps,
                          int eps_mode, float* y1, float* y2, hipStream_t s);
int32_t launch_ffn_residual(const float* z, const float* x, size_t rows, int d, const gnx_ffn& ff, float* out, hipStream_t s);
int32_t launch_pad(const gnx_graphs* h, int kind, bool pad, const float* src, int d, int64_t R, float* dst, hipStream_t s);
int32_t launch_calibration(int n, hipStream_t s);
int xent_blocks(int64_t cols);
int32_t launch_xent_backward(const float* logits, const float* targets, int d, int64_t cols, const float* upstream, float* dl, hipStream_t s);
int32_t launch_xent(const float* logits, const float* targets, int d, int64_t cols, float* out, float* ws, hipStream_t s);
int32_t launch_collapse(const gnx_graphs* h, const float* ef, int d, int64_t R, float* out, hipStream_t s);
int32_t launch_collapse_padded(const gnx_graphs* h, const float* ef, int d, int64_t R, float* out, hipStream_t s);
int32_t launch_fn_input(const gnx_graphs* h, int kind, const float* ef, int de, const float* nf, int dn, const float* gf, int dg,
                        int64_t R, float* out, hipStream_t s);
// returns 1 when the path does not apply to these dims (caller falls through to the next path)
// phase bit 1: edge + node update (leaves per-tile partial sums in the workspace); bit 2: graph update from them
int32_t launch_block_narrow(const gnx_graphs* h, const BlockArgs& a, int64_t R, hipStream_t s, int phase);
void warm_block_narrow(const gnx_graphs* h, const gnx_block_params* p);
bool block_narrow_ready(const gnx_graphs* h, const BlockArgs& a, hipStream_t s);
bool block_narrow_ffe_applies(const gnx_graphs* h, const BlockArgs& a, int act1, int act2);
bool block_narrow_chain_applies(const gnx_graphs* h, const BlockArgs& a);
int32_t launch_block_narrow_chained(const gnx_graphs* h, const BlockArgs& a, int64_t R, hipStream_t s);
static_assert(GNX_ACT_IDENTITY == 0 && GNX_ACT_RELU == 1 && GNX_ACT_TANH == 2 && GNX_ACT_SIGMOID == 3 && GNX_ACT_GELU == 4,
              "act_apply (gnx_device.h) hard-codes the activation codes");
int32_t launch_block_wide(const gnx_graphs* h, const BlockArgs& a, int64_t R, hipStream_t s, int phase);
size_t wide_workspace_bytes(const gnx_graphs* h, const gnx_block_params* p, int64_t R);
void warm_block_wide(const gnx_graphs* h, const gnx_block_params* p, bool rows_gemm);
// narrow-width GNCore kernels (gnx_core_narrow.hip)
bool core_narrow_width(int d);
int32_t launch_ln1_rows(const float* x, size_t rows, int d, const gnx_layernorm& l1, float eps, int eps_mode, float* y, hipStream_t s);
int32_t launch_ffn_fused(const gnx_graphs* h, int entity, const float* z, int d, const gnx_ffn& ff, const float* add1, const float* add2, float* out,
                         int64_t R, hipStream_t s, const float* ln_stats = nullptr, const gnx_layernorm* ln = nullptr, void* scratch = nullptr, size_t scratch_bytes = 0,
                         bool ln_inline = false, float ln_eps = 0.f, int ln_mode = 0);
bool ffn_x6_applies(const float* z, int d, const gnx_ffn& ff, const float* add1, const float* add2, const float* out, size_t scratch_bytes);  // gnx_ffn_x6.hip
bool block_wide_edge_x6_applies(const gnx_graphs* h, const BlockArgs& a);  // gnx_wide.hip
bool edge_n_enabled();  // gnx_edge_n.hip
bool ffn_fused_applies(const float* z, int d, const gnx_ffn& ff, const float* add1, const float* add2, const float* out);
bool block_wide_ln_applies(const gnx_graphs* h, const BlockArgs& a);
bool ln_stats_applies(const float* x, int d);
int32_t launch_ln_stats(const float* x, size_t rows, int d, float eps, int eps_mode, float* stats, hipStream_t s);
bool core_post3_applies(const size_t rows[3], const int d[3], const gnx_ffn ff[3], bool deferred, hipStream_t s);
int32_t launch_core_post3(const float* const x[3], const size_t rows[3], const int d[3], const gnx_layernorm l2[3], const gnx_ffn ff[3], float eps,
                          int eps_mode, float* const out[3], hipStream_t s, const BlockArgs* blk, int n_rows, bool skip_edges = false);
int32_t launch_core_post(const float* x, size_t rows, int d, const gnx_layernorm& l2, const gnx_ffn& ff, float eps, int eps_mode,
                         float* out, hipStream_t s);
int32_t launch_dense_rows(const gnx_graphs* h, int entity, const float* A, int K, const gnx_dense& d, int OUT, const float* add1,
                          const float* add2, float* out, int64_t R, hipStream_t s, const char* name);

static int32_t check_block(const gnx_graphs* h, const gnx_block_params* p, int64_t R) {
  if (!h || !p) return fail(GNX_ERR_INVALID_ARG, "NULL handle or params");
  if (R <= 0) return fail(GNX_ERR_INVALID_ARG, "n_replicas must be >= 1");
  if (R > 1 && h->G != 1)
    return fail(GNX_ERR_INVALID_ARG, "n_replicas > 1 needs a single-graph handle (shared adjacency, batch.jl:66)");
  if (R > 65535) return fail(GNX_ERR_TOO_LARGE, "n_replicas exceeds the grid limit (65535)");
  const int d[6] = {p->de, p->dn, p->dg, p->oe, p->on, p->og};
  for (int i = 0; i < 6; ++i)
    if (d[i] < 0) return fail(GNX_ERR_DIMS, "negative feature width");
  if (p->de + p->dn + p->dg == 0) return fail(GNX_ERR_DIMS, "all input widths are 0 (gnblock.jl:48, batch.jl:56)");
  if (p->oe + p->on + p->og == 0) return fail(GNX_ERR_DIMS, "all output widths are 0 (gnblock.jl:49)");
  const gnx_dense* fn[3] = {&p->edgefn, &p->nodefn, &p->graphfn};
  const int out[3] = {p->oe, p->on, p->og};
  const int in[3] = {p->de + 2 * p->dn + p->dg, p->oe + p->dn + p->dg, p->oe + p->on + p->dg};
  for (int i = 0; i < 3; ++i) {
    if (out[i] > 0 && in[i] > 0 && !fn[i]->weight) return fail(GNX_ERR_INVALID_ARG, "Dense weight is NULL");
    if (fn[i]->act < GNX_ACT_IDENTITY || fn[i]->act > GNX_ACT_GELU) return fail(GNX_ERR_INVALID_ARG, "unknown activation");
  }
  return GNX_OK;
}

struct BlockWs {
  size_t agg_off, part_off, total;
};

static BlockWs block_ws(const gnx_graphs* h, const gnx_block_params* p, int64_t R) {
  BlockWs w;
  w.agg_off = 0;
  const size_t agg = align_up(sizeof(float) * (size_t)R * h->N * p->oe, 256);
  w.part_off = w.agg_off + agg;
  // partial-sum rows: generic path [n_tiles][C]; fused narrow path [n_wtiles (or workgroups)][4*ceil(C/4)]
  const size_t C = (size_t)(p->oe + p->on);
  const size_t rows_bytes = std::max((size_t)h->n_tiles() * C, (size_t)h->n_wtiles() * ((C + 3) / 4 * 4));
  const size_t wide_part = wide_workspace_bytes(h, p, R);
  const size_t part = align_up(std::max(sizeof(float) * (size_t)R * rows_bytes, wide_part), 256);
  w.total = w.part_off + part + 256;
  return w;
}

// ln1 (optional, 3 entries): LayerNorm applied to the inputs on load — only together with *fused_ln: the caller passes a bool
// that is set when the fused narrow kernel took the call; otherwise nothing was launched and the caller must normalise itself.
static int32_t block_forward_impl(const gnx_graphs* h, const gnx_block_params* p, const float* ef, const float* nf,
                                  const float* gf, int64_t R, float* ef_out, float* nf_out, float* gf_out, void* ws,
                                  size_t ws_bytes, uint32_t flags, hipStream_t s, int phase = 3, const gnx_layernorm* ln1 = nullptr,
                                  float ln_eps = 0.f, int ln_mode = 0, bool* fused_ln = nullptr, const float* const* wide_ln_stats = nullptr,
                                  BlockArgs* args_out = nullptr, const gnx_ffn* ffe = nullptr, const gnx_layernorm* ffe_ln2 = nullptr, bool* ffe_took = nullptr,
                                  const gnx_pending_update* chain_prev = nullptr, bool* chain_took = nullptr, bool* edge_x6_out = nullptr, bool ln_inline_e = false,
                                  void* ffe_scratch = nullptr) {
  FormScope forms(flags);  // the forms this call selected (gnx.h: GNX_FLAG_FFN_FP32 ...) for every dispatch predicate below
  PreparedScope prepared(p ? p->prepared : nullptr);  // the layer's prepared weight planes, if the caller made them (a core passes its own through its block)
  int32_t rc = check_block(h, p, R);
  if (rc) return rc;
  if (phase & 1) {
    // a batch without a single edge has (DE, 0) edge features: its (empty) buffers may be NULL
    if ((p->de > 0 && !ef && h->E > 0) || (p->dn > 0 && !nf) || (p->dg > 0 && !gf))
      return fail(GNX_ERR_INVALID_ARG, "an input with non-zero width is NULL (width 0 <=> nothing)");
    if ((p->oe > 0 && !ef_out && h->E > 0) || (p->on > 0 && !nf_out))
      return fail(GNX_ERR_INVALID_ARG, "an output with non-zero width is NULL");
  }
  if (phase & 2) {
    if ((p->dg > 0 && !gf) || (p->og > 0 && !gf_out)) return fail(GNX_ERR_INVALID_ARG, "gf / gf_out is NULL");
  }
  const BlockWs w = block_ws(h, p, R);
  if (!ws || ws_bytes < w.total) return fail(GNX_ERR_WORKSPACE, "workspace missing or smaller than gnx_block_workspace_bytes()");
  if (((uintptr_t)ws & 15) != 0) return fail(GNX_ERR_WORKSPACE, "workspace must be 16-byte aligned");

  BlockArgs a{};
  a.de = p->de; a.dn = p->dn; a.dg = p->dg;
  a.oe = p->oe; a.on = p->on; a.og = p->og;
  a.We = p->edgefn.weight; a.be = p->edgefn.bias; a.act_e = p->edgefn.act;
  a.Wn = p->nodefn.weight; a.bn = p->nodefn.bias; a.act_n = p->nodefn.act;
  a.Wg = p->graphfn.weight; a.bg = p->graphfn.bias; a.act_g = p->graphfn.act;
  a.ef = p->de ? ef : nullptr; a.nf = p->dn ? nf : nullptr; a.gf = p->dg ? gf : nullptr;
  a.ef_out = ef_out; a.nf_out = nf_out; a.gf_out = gf_out;
  a.agg = reinterpret_cast<float*>(static_cast<char*>(ws) + w.agg_off);
  a.partials = reinterpret_cast<float*>(static_cast<char*>(ws) + w.part_off);
  a.colptr = h->d_colptr; a.rowval = h->d_rowval; a.node_off = h->d_node_off; a.edge_off = h->d_edge_off;
  a.tile_off = h->d_tile_off; a.tiles = h->d_tiles;
  a.wtile_off = h->d_wtile_off; a.wtiles = h->d_wtiles; a.n_wtiles = (int)h->n_wtiles();
  a.N = (int)h->N; a.E = (int)h->E; a.G = (int)h->G; a.n_tiles = (int)h->n_tiles();
  a.packs = h->d_packs; a.n_packs = h->n_packs;

  if (ln1 && wide_ln_stats) {
    // matrix-core path with ef / nf normalised on load from their row statistics (gf arrives normalised).  wide_ln_stats[0] == nullptr:
    // only ASK whether this block takes that form (nothing is launched)
    for (int t = 0; t < 2; ++t) { a.ln_g[t] = ln1[t].gamma; a.ln_b[t] = ln1[t].beta; }
    *fused_ln = !(flags & (GNX_FLAG_FORCE_GENERIC | GNX_FLAG_NO_MFMA)) && block_wide_ln_applies(h, a);
    if (edge_x6_out) *edge_x6_out = *fused_ln && block_wide_edge_x6_applies(h, a);  // (the caller may then leave the edge rows' statistics to that kernel)
    if (!*fused_ln || !wide_ln_stats[0]) return GNX_OK;
    a.ln_stats[0] = wide_ln_stats[0]; a.ln_stats[1] = wide_ln_stats[1];
    if (ln_inline_e) { a.ln_stats[0] = nullptr; a.ln_inline_e = 1; a.ln_eps = ln_eps; a.ln_mode = ln_mode; }  // no table for the edges: k_edge_x6 computes them in registers
    if (ln_inline_e && ffe && ffe_ln2 && ffe_scratch && (phase & 1)) {  // ... and the edge FeedForward + residuals run in that launch too: ef_out receives the CORE's edge output
      a.ffe_w1 = ffe->fc1.weight; a.ffe_b1 = ffe->fc1.bias; a.ffe_w2 = ffe->fc2.weight; a.ffe_b2 = ffe->fc2.bias;
      a.ffe_g2 = ffe_ln2->gamma; a.ffe_be2 = ffe_ln2->beta; a.ffe_act1 = ffe->fc1.act; a.ffe_act2 = ffe->fc2.act; a.ffe_scratch = ffe_scratch;
    }
    return launch_block_wide(h, a, R, s, phase);
  }
  if (ln1) {
    *fused_ln = false;
    if (flags & GNX_FLAG_FORCE_GENERIC) return GNX_OK;
    for (int t = 0; t < 3; ++t) { a.ln_g[t] = ln1[t].gamma; a.ln_b[t] = ln1[t].beta; }
    a.ln_eps = ln_eps; a.ln_mode = ln_mode;
    if (!block_narrow_ready(h, a, s)) return GNX_OK;  // nothing launched: the caller runs gn1 as its own kernels
    *fused_ln = true;
    if (ffe && ffe_ln2 && ffe_took && (phase & 1) && block_narrow_ffe_applies(h, a, ffe->fc1.act, ffe->fc2.act)) {
      // narrow core: the edge FeedForward and both residual terms run in the block kernel's edge lanes — ef_out receives the CORE's output
      a.ffe_w1 = ffe->fc1.weight; a.ffe_b1 = ffe->fc1.bias; a.ffe_w2 = ffe->fc2.weight; a.ffe_b2 = ffe->fc2.bias;
      a.ffe_g2 = ffe_ln2->gamma; a.ffe_be2 = ffe_ln2->beta; a.ffe_act1 = ffe->fc1.act; a.ffe_act2 = ffe->fc2.act;
      *ffe_took = true;
    }
    if (args_out) *args_out = a;
    return launch_block_narrow(h, a, R, s, phase);
  }
  if (chain_took) {  // gnx_block_forward_chained: this call's edge + node update with the previous call's graph update at the front of the launch
    *chain_took = !(flags & (GNX_FLAG_FORCE_GENERIC)) && block_narrow_chain_applies(h, a);
    if (!*chain_took) return GNX_OK;  // nothing launched: the caller runs the plain form
    if (chain_prev && chain_prev->workspace) {
      a.prev_partials = reinterpret_cast<const float*>(static_cast<const char*>(chain_prev->workspace) + w.part_off);
      a.prev_gf = chain_prev->gf; a.prev_gf_out = chain_prev->gf_out;
    }
    return launch_block_narrow_chained(h, a, R, s);
  }
  if (!(flags & GNX_FLAG_FORCE_GENERIC)) {
    rc = launch_block_narrow(h, a, R, s, phase);  // fused wave-per-tile kernel: ahead-of-time width sets, else specialised at run time
    if (rc != 1) return rc;
    if (!(flags & GNX_FLAG_NO_MFMA)) {
      rc = launch_block_wide(h, a, R, s, phase);  // fp32 MFMA gathered-row GEMMs
      if (rc != 1) return rc;
    }
  }
  return launch_block_generic(a, R, h->tile_n_cap, s, phase);
}

}  // namespace gnx

extern "C" int32_t gnx_ensure_collapse(const gnx_graphs* h);

using namespace gnx;

extern "C" {

size_t gnx_block_workspace_bytes(const gnx_graphs* h, const gnx_block_params* p, int64_t R) {
  if (!h || !p || R <= 0) return 0;
  if (check_block(h, p, R) == GNX_OK) {
    warm_block_narrow(h, p);        // run-time specialisation happens here, not in a capture
    warm_block_wide(h, p, false);   // ... and so does the build of the matrix-core tables when these widths take that path
  }
  return block_ws(h, p, R).total;
}

int32_t gnx_block_forward(const gnx_graphs* h, const gnx_block_params* p, const float* ef, const float* nf, const float* gf,
                          int64_t R, float* ef_out, float* nf_out, float* gf_out, void* ws, size_t ws_bytes, uint32_t flags,
                          void* stream) {
  DeviceTurn turn((hipStream_t)stream, p && matrix_core_widths(*p));  // (one matrix-core call at a time per device: gnx_internal.h)
  return block_forward_impl(h, p, ef, nf, gf, R, ef_out, nf_out, gf_out, ws, ws_bytes, flags, (hipStream_t)stream,
                            (flags & GNX_FLAG_DEFER_GRAPH_UPDATE) ? 1 : 3);
}

int32_t gnx_block_forward_chained(const gnx_graphs* h, const gnx_block_params* p, const float* ef, const float* nf, const float* gf, int64_t R, float* ef_out,
                                  float* nf_out, float* gf_out, void* ws, size_t ws_bytes, uint32_t flags, void* stream, const gnx_pending_update* prev,
                                  gnx_pending_update* pending) {
  if (!pending) return fail(GNX_ERR_INVALID_ARG, "pending is NULL");
  DeviceTurn turn((hipStream_t)stream, p && matrix_core_widths(*p));
  if (flags & GNX_FLAG_DEFER_GRAPH_UPDATE) return fail(GNX_ERR_INVALID_ARG, "gnx_block_forward_chained defers the graph update itself");
  if (prev && prev->workspace && (prev->workspace == ws || (p && p->og > 0 && prev->gf_out == gf_out)))
    return fail(GNX_ERR_INVALID_ARG, "the pending call's workspace / gf_out must not be this call's (its graph update has not run yet)");
  bool took = false;
  int32_t rc = block_forward_impl(h, p, ef, nf, gf, R, ef_out, nf_out, gf_out, ws, ws_bytes, flags, (hipStream_t)stream, 1, nullptr, 0.f, 0, nullptr, nullptr, nullptr,
                                  nullptr, nullptr, nullptr, prev, &took);
  if (rc) return rc;
  if (took) {  // prev's graph update rode in this launch; this call's is pending
    pending->workspace = ws; pending->workspace_bytes = ws_bytes; pending->gf = gf; pending->gf_out = gf_out;
    return GNX_OK;
  }
  // not the two-launch narrow form (matrix-core / generic kernels, run-time specialised widths, batches of small graphs whose graph update
  // already runs inside the block kernel): finish the previous call the plain way, run this call whole, nothing stays pending
  if (prev && prev->workspace) {
    rc = gnx_block_graph_update(h, p, prev->gf, R, prev->gf_out, const_cast<void*>(prev->workspace), prev->workspace_bytes, flags, stream);
    if (rc) return rc;
  }
  pending->workspace = nullptr; pending->workspace_bytes = 0; pending->gf = nullptr; pending->gf_out = nullptr;
  return block_forward_impl(h, p, ef, nf, gf, R, ef_out, nf_out, gf_out, ws, ws_bytes, flags, (hipStream_t)stream, 3);
}

int32_t gnx_block_forward_steps(const gnx_graphs* h, const gnx_block_params* p, const gnx_block_step* steps, int64_t n_steps, int64_t R, uint32_t flags,
                                void* stream) {
  if (n_steps < 0 || (n_steps > 0 && !steps)) return fail(GNX_ERR_INVALID_ARG, "gnx_block_forward_steps: steps is NULL / n_steps is negative");
  if (flags & GNX_FLAG_DEFER_GRAPH_UPDATE) return fail(GNX_ERR_INVALID_ARG, "gnx_block_forward_steps finishes every step's graph update itself");
  DeviceTurn turn((hipStream_t)stream, p && matrix_core_widths(*p));  // (one turn for the whole loop; the calls below nest inside it)
  gnx_pending_update pend{};
  auto flush = [&]() -> int32_t {
    if (!pend.workspace) return GNX_OK;
    const int32_t rc = gnx_block_graph_update(h, p, pend.gf, R, pend.gf_out, const_cast<void*>(pend.workspace), pend.workspace_bytes, flags, stream);
    pend = gnx_pending_update{};
    return rc;
  };
  for (int64_t i = 0; i < n_steps; ++i) {
    const gnx_block_step& st = steps[i];
    // a step that shares its predecessor's workspace / gf_out cannot start before that one's graph update has run
    if (pend.workspace && (pend.workspace == st.workspace || (p && p->og > 0 && pend.gf_out == st.gf_out))) {
      if (int32_t rc = flush()) return rc;
    }
    gnx_pending_update next{};
    if (int32_t rc = gnx_block_forward_chained(h, p, st.ef, st.nf, st.gf, R, st.ef_out, st.nf_out, st.gf_out, st.workspace, st.workspace_bytes, flags, stream,
                                               pend.workspace ? &pend : nullptr, &next)) {
      // (an argument error of step i: what is pending belongs to step i - 1, whose arguments were valid — finish it, report the error)
      (void)flush();
      return rc;
    }
    pend = next;
  }
  return flush();
}

int32_t gnx_block_graph_update(const gnx_graphs* h, const gnx_block_params* p, const float* gf, int64_t R, float* gf_out, void* ws,
                               size_t ws_bytes, uint32_t flags, void* stream) {
  // the kernels of this phase read only gf, the graph function's parameters and the workspace
  return block_forward_impl(h, p, nullptr, nullptr, gf, R, nullptr, nullptr, gf_out, ws, ws_bytes, flags, (hipStream_t)stream, 2);
}

// FeedForward width from which the two Dense layers run on the matrix cores (hidden activations staged in HBM)
static bool ffn_on_mfma(int d) { return d >= 32; }

// workspace of a core: LN1 and LN2 outputs for edges, nodes, graphs, the FFN hidden buffer, then the block workspace
static void core_ws(const gnx_graphs* h, const gnx_core_params* p, int64_t R, size_t off[8], size_t* total) {
  const size_t rows[3] = {(size_t)R * h->E, (size_t)R * h->N, (size_t)R * h->G};
  const int d[3] = {p->block.de, p->block.dn, p->block.dg};
  size_t o = 0, hidden = 0;
  for (int t = 0; t < 3; ++t) {
    off[2 * t] = o; o += align_up(sizeof(float) * rows[t] * d[t], 256);
    off[2 * t + 1] = o; o += align_up(sizeof(float) * rows[t] * d[t], 256);
    if (ffn_on_mfma(d[t])) hidden = std::max(hidden, sizeof(float) * rows[t] * 4 * (size_t)d[t]);
  }
  off[6] = o; o += align_up(hidden, 256);
  off[7] = o;
  *total = o + block_ws(h, &p->block, R).total;
}

// side stream + fork / join events of the handle (see gnx_internal.h); failure just leaves the core on one stream
static void ensure_aux(const gnx_graphs* h) {
  std::call_once(h->aux_once, [h]() {
    // the streams and their events belong to the HANDLE's device, whatever device is current in the querying thread
    int prev = -1;
    (void)hipGetDevice(&prev);
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev != h->device ? prev : -1};
    if (prev != h->device && hipSetDevice(h->device) != hipSuccess) { (void)hipGetLastError(); return; }
    for (auto& ax : h->aux) {
      hipStream_t st = nullptr;
      hipEvent_t e1 = nullptr, e2 = nullptr;
      if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess &&
          hipEventCreateWithFlags(&e2, hipEventDisableTiming) == hipSuccess) {
        ax.stream = st; ax.fork = e1; ax.join = e2;
      } else {
        if (e1) (void)hipEventDestroy(e1);
        if (e2) (void)hipEventDestroy(e2);
        if (st) (void)hipStreamDestroy(st);
        (void)hipGetLastError();
        break;
      }
    }
  });
}

size_t gnx_core_workspace_bytes(const gnx_graphs* h, const gnx_core_params* p, int64_t R) {
  if (!h || !p || R <= 0) return 0;
  ensure_aux(h);
  warm_block_wide(h, &p->block, ffn_on_mfma(p->block.de) || ffn_on_mfma(p->block.dn) || ffn_on_mfma(p->block.dg));
  size_t off[8], total;
  core_ws(h, p, R, off, &total);
  {  // run-time specialisation of a narrow core's combined FeedForward launch happens here (as for the block: never in a capture)
    const size_t rows[3] = {(size_t)R * h->E, (size_t)R * h->N, (size_t)R * h->G};
    const int d[3] = {p->block.de, p->block.dn, p->block.dg};
    if (h->E > 0 && core_narrow_width(d[0]) && core_narrow_width(d[1]) && core_narrow_width(d[2])) (void)core_post3_applies(rows, d, p->ff, true, nullptr);
  }
  return total;
}

int32_t gnx_core_forward(const gnx_graphs* h, const gnx_core_params* p, const float* ef, const float* nf, const float* gf,
                         int64_t R, float* ef_out, float* nf_out, float* gf_out, void* ws, size_t ws_bytes, uint32_t flags,
                         void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!h || !p) return fail(GNX_ERR_INVALID_ARG, "NULL handle or params");
  DeviceTurn turn(s, matrix_core_widths(p->block));  // (one matrix-core call at a time per device: gnx_internal.h)
  FormScope forms(flags);  // the forms this call selected (gnx.h: GNX_FLAG_FFN_FP32 ...)
  gnx_block_params b = p->block;
  b.prepared = p->prepared;  // (the core's object holds its block's planes too; block.prepared is ignored)
  PreparedScope prepared(p->prepared);
  // GNFeedForward / GNGraphNorm need all three widths > 0 and graphnetadd needs all three present
  // (gnfeedforward.jl:18, gngraphnorm.jl:10, gncore.jl:61-68); the block maps dims => dims (gncore.jl:49).
  if (b.de <= 0 || b.dn <= 0 || b.dg <= 0) return fail(GNX_ERR_DIMS, "GNCore needs all(dims .> 0) (gnfeedforward.jl:18)");
  if (b.oe != b.de || b.on != b.dn || b.og != b.dg) return fail(GNX_ERR_DIMS, "GNCore's block must map dims => dims (gncore.jl:49)");
  int32_t rc = check_block(h, &b, R);
  if (rc) return rc;
  if (((!ef || !ef_out) && h->E > 0) || !nf || !gf || !nf_out || !gf_out) return fail(GNX_ERR_INVALID_ARG, "GNCore needs ef, nf, gf and all outputs");
  for (int t = 0; t < 3; ++t) {
    if (!p->ln1[t].gamma || !p->ln1[t].beta || !p->ln2[t].gamma || !p->ln2[t].beta) return fail(GNX_ERR_INVALID_ARG, "LayerNorm parameter is NULL");
    if (!p->ff[t].fc1.weight || !p->ff[t].fc2.weight) return fail(GNX_ERR_INVALID_ARG, "FeedForward weight is NULL");
  }
  if (p->eps_mode != 0 && p->eps_mode != 1) return fail(GNX_ERR_INVALID_ARG, "eps_mode must be 0 or 1");
  {
    const int dd[3] = {b.de, b.dn, b.dg};
    for (int t = 0; t < 3; ++t) {
      const bool generic = !ffn_on_mfma(dd[t]) || (flags & (GNX_FLAG_FORCE_GENERIC | GNX_FLAG_NO_MFMA));
      if (generic && (size_t)dd[t] * 5 * 4 * sizeof(float) > 64 * 1024)
        return fail(GNX_ERR_DIMS, "GNCore width too large for the generic FFN kernel (80*d bytes of LDS)");
    }
  }
  size_t off[8], total;
  core_ws(h, p, R, off, &total);
  if (!ws || ws_bytes < total) return fail(GNX_ERR_WORKSPACE, "workspace missing or smaller than gnx_core_workspace_bytes()");
  if (((uintptr_t)ws & 15) != 0) return fail(GNX_ERR_WORKSPACE, "workspace must be 16-byte aligned");
  char* base = static_cast<char*>(ws);
  const size_t rows[3] = {(size_t)R * h->E, (size_t)R * h->N, (size_t)R * h->G};
  const int d[3] = {b.de, b.dn, b.dg};
  const float* x[3] = {ef, nf, gf};
  float* out[3] = {ef_out, nf_out, gf_out};
  float* l1[3];
  float* l2[3];
  for (int t = 0; t < 3; ++t) {
    l1[t] = reinterpret_cast<float*>(base + off[2 * t]);
    l2[t] = reinterpret_cast<float*>(base + off[2 * t + 1]);
  }
  // All three widths narrow: the fused block kernel normalises its inputs as it loads them (gn1 never materialised) when it
  // is available for this width set; gn2 is recomputed inside k_core_post either way.
  bool fused_ln = false, defer_gu = false, edge_ff_done = false;
  BlockArgs blk_args{};
  const bool all_narrow = core_narrow_width(d[0]) && core_narrow_width(d[1]) && core_narrow_width(d[2]) && !(flags & GNX_FLAG_FORCE_GENERIC);
  if (all_narrow) {
    // (the graph level of a NARROW core on the handle's side stream — graph update + the G-row / N-row k_core_post launches behind the
    // edges' k_core_post — was measured: README ex.3 model 298 vs 271 us; two fork/join pairs cost more than the ~20 us they hide)
    // when the three FeedForwards go out as ONE launch (k_core_post3), the block's graph update runs inside it: the block is launched
    // without its k_graph_t
    defer_gu = h->E > 0 && !(flags & GNX_FLAG_DEFER_GRAPH_UPDATE) && core_post3_applies(rows, d, p->ff, true, s);
    // (the FeedForward moves into the block kernel only together with the one-launch post kernel, which then skips the edge rows)
    rc = block_forward_impl(h, &b, ef, nf, gf, R, out[0], out[1], out[2], base + off[7], ws_bytes - off[7], flags, s, defer_gu ? 1 : 3, p->ln1, p->eps,
                            p->eps_mode, &fused_ln, nullptr, &blk_args, defer_gu ? &p->ff[0] : nullptr, &p->ln2[0], &edge_ff_done);
    if (rc) return rc;
    defer_gu = defer_gu && fused_ln;
  }
  // Wide edges and nodes: the matrix-core kernels normalise x as they load it (block: gn1, fused FeedForward: gn2) from one pass of
  // row statistics — neither LayerNorm output of ef / nf exists in HBM.  Taken when the block runs in the projected quad-row form
  // and both FeedForwards are the fused kernel's; gf (G rows) is normalised by the ordinary kernel.
  const bool no_ln_fuse = form(GNX_FLAG_NO_LN_FUSE);
  bool wide_ln = false, edge_x6 = false;
  if (!fused_ln && !all_narrow && !no_ln_fuse && !(flags & (GNX_FLAG_FORCE_GENERIC | GNX_FLAG_NO_MFMA)) && h->E > 0) {
    const float* ask[2] = {nullptr, nullptr};
    bool ok = true;
    for (int t = 0; t < 2; ++t)
      ok = ok && ln_stats_applies(x[t], d[t]) && ffn_fused_applies(x[t], d[t], p->ff[t], out[t], x[t], out[t]) &&
           (((uintptr_t)p->ln2[t].gamma | (uintptr_t)p->ln2[t].beta) & 15) == 0;
    if (ok) {
      rc = block_forward_impl(h, &b, x[0], x[1], l1[2], R, out[0], out[1], out[2], base + off[7], ws_bytes - off[7], flags, s, 3, p->ln1, p->eps, p->eps_mode, &wide_ln, ask,
                              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &edge_x6);
      if (rc) return rc;
    }
  }
  // Both consumers of the edge rows' statistics hold whole rows in registers when they are the six-term kernels (k_edge_x6: gn1, k_ffn_x6: gn2) and
  // compute them there, bit-identical to k_ln_stats_v4 (gnx_x6_stats.h): the statistics pass over ef — 512 MB at 1M edges — is not launched.
  const bool inline_e = wide_ln && edge_x6 && d[0] == 128 && rows[0] >= 4096 && !form(GNX_FLAG_LN_STATS_PASS) &&
                        ffn_x6_applies(x[0], d[0], p->ff[0], out[0], x[0], out[0], sizeof(float) * rows[0] * d[0]);
  // ... and then ONE launch does both (the edge form of k_ffn_x6: ef' stays in the accumulator — never written, never read back; GNX_CORE_EDGE_SPLIT=1: two launches)
  const bool fuse_e = inline_e && !form(GNX_FLAG_CORE_EDGE_SPLIT) && !(edge_n_enabled() && d[1] == 64);  // (the opt-in k_edge_n gathers RAW source rows: its projection tables are not the one-launch form's)
  // Round 6 (profiles/r06_overlap_hazard.log): the GENERAL kernels (k_rows_gemm, k_ffn_fused) came out wrong now and then beside another queue's
  // matrix kernel, but only in launches that normalise on load from a statistics table; the site was found later in the round (their LayerNorm
  // branch consumed an LDS read too early on a shared CU) and is guarded (GNX_LN_GUARD).  This rule predates the guard and stays as a second,
  // independent protection of the path small batches take: a statistics table is consumed by six-term kernels only — an entity whose rows a general
  // kernel would normalise on load gets its LayerNorms MATERIALISED instead (k_layernorm2, one more pass over rows that are few whenever this
  // happens: below 4096, or widths without a six-term kernel).  GNX_FLAG_LN_ON_LOAD restores the statistics-table forms (exact too since the guard).
  const bool ln_on_load = form(GNX_FLAG_LN_ON_LOAD) || form(GNX_FLAG_FP32_MFMA | GNX_FLAG_PROJ_FP32 | GNX_FLAG_LN_STATS_PASS | GNX_FLAG_CORE_EDGE_SPLIT);  // (the diagnostic forms keep their tables)
  if (wide_ln && !ln_on_load && !inline_e) wide_ln = false;  // the edge rows' table would feed k_rows_gemm / k_ffn_fused: everything materialised (the branch below)
  const bool node_x6_forms = d[0] == 128 && d[1] == 64 && h->N >= 4096 && p->block.nodefn.act <= GNX_ACT_RELU;  // k_proj_x6, k_node_x6, k_ffn_x6<64> take the node rows
  const bool node_mat = wide_ln && !ln_on_load && !node_x6_forms;  // edges by the six-term kernels (statistics in registers), node LayerNorms materialised
  if (wide_ln) {
    const float* stats[2] = {l1[0], node_mat ? nullptr : l1[1]};  // the (unused) gn1 buffers hold the statistics: 2 floats per row
    const float* node_in = node_mat ? l1[1] : x[1];              // node_mat: gn1(nf) itself (and l2[1] = gn2(nf) for the FeedForward)
    auto node_ln = [&](hipStream_t st) -> int32_t {
      return node_mat ? launch_layernorm2(x[1], rows[1], d[1], p->ln1[1], p->ln2[1], p->eps, p->eps_mode, l1[1], l2[1], st)
                      : launch_ln_stats(x[1], rows[1], d[1], p->eps, p->eps_mode, l1[1], st);
    };
    auto node_ffn = [&](hipStream_t st) -> int32_t {  // out[1] = nf' + nf + FF(gn2(nf))
      return node_mat ? launch_ffn_fused(h, 1, l2[1], d[1], p->ff[1], out[1], x[1], out[1], R, st)
                      : launch_ffn_fused(h, 1, x[1], d[1], p->ff[1], out[1], x[1], out[1], R, st, l1[1], &p->ln2[1], l2[1], sizeof(float) * rows[1] * d[1]);
    };
    const bool no_fork0 = form(GNX_FLAG_NO_FORK);
    // A side stream and its pair of events from the handle's pool, held while this call enqueues its work (a second host thread in this section —
    // same handle, another stream, other buffers — takes the next set; with every set taken a caller runs everything on its own stream).
    std::unique_lock<std::mutex> aux_lk;
    const gnx_graphs::AuxSet* aux = nullptr;
    if (!no_fork0 && !profile_enabled())
      for (auto& ax : h->aux) {
        if (!ax.stream) break;
        std::unique_lock<std::mutex> lk(ax.mu, std::try_to_lock);
        if (lk.owns_lock()) { aux_lk = std::move(lk); aux = &ax; break; }
      }
    const bool fork0 = aux != nullptr;
    if ((rc = launch_layernorm2(x[2], rows[2], d[2], p->ln1[2], p->ln2[2], p->eps, p->eps_mode, l1[2], l2[2], s))) return rc;
    if (fork0) {
      // side stream: node statistics, gf fold, node projections (latency / matrix-core work) beside the edge statistics pass (HBM-bound)
      hipStream_t ax = aux->stream;
      bool took0 = false;
      GNX_HIP(hipEventRecord(aux->fork, s));
      GNX_HIP(hipStreamWaitEvent(ax, aux->fork, 0));
      rc = node_ln(ax);
      if (rc == GNX_OK) rc = block_forward_impl(h, &b, x[0], node_in, l1[2], R, out[0], out[1], out[2], base + off[7], ws_bytes - off[7], flags, ax, 4, p->ln1, p->eps, p->eps_mode, &took0, stats);
      const hipError_t e1 = hipEventRecord(aux->join, ax);
      const int32_t rc2 = inline_e ? GNX_OK : launch_ln_stats(x[0], rows[0], d[0], p->eps, p->eps_mode, l1[0], s);
      const hipError_t e2 = hipStreamWaitEvent(s, aux->join, 0);
      if (rc) return rc;
      if (rc2) return rc2;
      GNX_HIP(e1);
      GNX_HIP(e2);
    } else {
      if (!inline_e && (rc = launch_ln_stats(x[0], rows[0], d[0], p->eps, p->eps_mode, l1[0], s))) return rc;
      if ((rc = node_ln(s))) return rc;
    }
    // The graph level of the core — the block's graph update (four 5-us launches) and the G-row FeedForward — is independent of the edge /
    // node FeedForwards that follow the block: it runs on the handle's side stream behind them (fork after the node update, join
    // before returning; inside a capture the side stream joins the captured graph).  GNX_NO_FORK=1: everything on the caller's stream.
    const bool no_fork = no_fork0;
    const bool fork = !no_fork && aux != nullptr;
    bool took = false;
    rc = block_forward_impl(h, &b, x[0], node_in, l1[2], R, out[0], out[1], out[2], base + off[7], ws_bytes - off[7], flags, s, (fork ? 1 : 3) | (fork0 ? 8 : 0), p->ln1, p->eps, p->eps_mode, &took, stats,
                            nullptr, fuse_e ? &p->ff[0] : nullptr, fuse_e ? &p->ln2[0] : nullptr, nullptr, nullptr, nullptr, nullptr, inline_e, fuse_e ? l2[0] : nullptr);
    if (rc) return rc;
    if (!took) return fail(GNX_ERR_INVALID_ARG, "gnx_core_forward: the block declined the form it had accepted");
    if (fork) {
      hipStream_t ax = aux->stream;
      GNX_HIP(hipEventRecord(aux->fork, s));
      GNX_HIP(hipStreamWaitEvent(ax, aux->fork, 0));
      rc = block_forward_impl(h, &b, x[0], node_in, l1[2], R, out[0], out[1], out[2], base + off[7], ws_bytes - off[7], flags, ax, 2, p->ln1, p->eps, p->eps_mode, &took, stats);
      if (rc == GNX_OK) {  // the G-row FeedForward: out = gf' + gf + FF(gn2(gf)); the hidden buffer is its alone (the wide FeedForwards are the fused kernel)
        float* hidden2 = reinterpret_cast<float*>(base + off[6]);
        rc = launch_ffn_fused(h, 2, l2[2], d[2], p->ff[2], out[2], x[2], out[2], R, ax);
        if (rc == 1) {
          if (ffn_on_mfma(d[2])) {
            rc = launch_dense_rows(h, 2, l2[2], d[2], p->ff[2].fc1, 4 * d[2], nullptr, nullptr, hidden2, R, ax, "k_rows_gemm_ff1");
            if (rc == GNX_OK) rc = launch_dense_rows(h, 2, hidden2, 4 * d[2], p->ff[2].fc2, d[2], out[2], x[2], out[2], R, ax, "k_rows_gemm_ff2");
          } else if (core_narrow_width(d[2])) {
            rc = launch_core_post(x[2], rows[2], d[2], p->ln2[2], p->ff[2], p->eps, p->eps_mode, out[2], ax);
          } else {
            rc = launch_ffn_residual(l2[2], x[2], rows[2], d[2], p->ff[2], out[2], ax);
          }
        }
      }
      // the node FeedForward rides on the side stream too: its workgroups fill the CUs that the edge FeedForward's last, partly
      // filled round of tiles leaves idle (7813 tiles on 512 slots: 15.26 rounds)
      static const bool node_ffn_main = getenv("GNX_NODE_FFN_MAIN") != nullptr;
      if (rc == GNX_OK && !node_ffn_main) rc = node_ffn(ax);
      // the join is recorded even after a failure: a capture must not end with the side stream un-joined
      const hipError_t e1 = hipEventRecord(aux->join, ax);
      // (the edges' gn2 buffer is unused in this form: room for the split weight planes of k_ffn_x6)
      int32_t rc2 = fuse_e ? GNX_OK  // (the block's edge launch was the edge form of k_ffn_x6: out[0] is final)
                           : launch_ffn_fused(h, 0, x[0], d[0], p->ff[0], out[0], x[0], out[0], R, s, inline_e ? nullptr : l1[0], &p->ln2[0], l2[0], sizeof(float) * rows[0] * d[0], inline_e,
                                              p->eps, p->eps_mode);
      if (rc2 == GNX_OK && node_ffn_main) rc2 = node_ffn(s);
      const hipError_t e2 = hipStreamWaitEvent(s, aux->join, 0);
      if (rc) return rc;
      if (rc2) return rc2;
      GNX_HIP(e1);
      GNX_HIP(e2);
      return GNX_OK;
    }
  } else if (!fused_ln) {
    for (int t = 0; t < 3; ++t) {
      // narrow widths: only gn1(x) is materialised (the block needs it); gn2 is recomputed inside k_core_post
      const bool narrow = core_narrow_width(d[t]) && !(flags & GNX_FLAG_FORCE_GENERIC);
      if (narrow) rc = launch_ln1_rows(x[t], rows[t], d[t], p->ln1[t], p->eps, p->eps_mode, l1[t], s);
      else rc = launch_layernorm2(x[t], rows[t], d[t], p->ln1[t], p->ln2[t], p->eps, p->eps_mode, l1[t], l2[t], s);
      if (rc) return rc;
    }
    rc = block_forward_impl(h, &b, l1[0], l1[1], l1[2], R, out[0], out[1], out[2], base + off[7], ws_bytes - off[7], flags, s);
    if (rc) return rc;
  }
  float* hidden = reinterpret_cast<float*>(base + off[6]);
  if (all_narrow) {  // the three entities' FeedForward + residual in one launch when the width triple has the combined kernel
    const int n_rows = (int)(h->G == 1 ? (h->n_wtiles() + 3) / 4 : h->n_wtiles());  // partial-sum rows of the fused narrow block (gnx_narrow.hip)
    rc = launch_core_post3(x, rows, d, p->ln2, p->ff, p->eps, p->eps_mode, out, s, defer_gu ? &blk_args : nullptr, n_rows, edge_ff_done);
    if (rc != 1) return rc;
    if (edge_ff_done) return fail(GNX_ERR_INVALID_ARG, "internal: the edge FeedForward ran in the block kernel but the one-launch post kernel declined");
  }
  for (int t = 0; t < 3; ++t) {
    if (ffn_on_mfma(d[t]) && !(flags & (GNX_FLAG_FORCE_GENERIC | GNX_FLAG_NO_MFMA))) {
      // out = block(LN1 x) + x + fc2(relu(fc1(LN2 x)))      (gncore.jl:56-68, gnfeedforward.jl:27-31)
      if (wide_ln && t < 2) {
        const bool inl = t == 0 && inline_e;
        if (t == 0 && fuse_e) continue;  // (the block's edge launch was the edge form of k_ffn_x6: out[0] is final)
        if (t == 1 && node_mat) {        // (materialised node LayerNorms: gn2(nf) is in l2[1])
          if ((rc = launch_ffn_fused(h, 1, l2[1], d[1], p->ff[1], out[1], x[1], out[1], R, s))) return rc;
          continue;
        }
        if ((rc = launch_ffn_fused(h, t, x[t], d[t], p->ff[t], out[t], x[t], out[t], R, s, inl ? nullptr : l1[t], &p->ln2[t], l2[t], sizeof(float) * rows[t] * d[t], inl, p->eps, p->eps_mode))) return rc;
        continue;
      }
      // hidden layer never leaves the chip (d = 64, 128); the edges' launch may use the hidden buffer of the two-GEMM form as its scratch
      rc = launch_ffn_fused(h, t, l2[t], d[t], p->ff[t], out[t], x[t], out[t], R, s, nullptr, nullptr, t < 2 ? hidden : nullptr, sizeof(float) * rows[t] * 4 * (size_t)d[t]);
      if (rc == GNX_OK) continue;
      if (rc != 1) return rc;
      if ((rc = launch_dense_rows(h, t, l2[t], d[t], p->ff[t].fc1, 4 * d[t], nullptr, nullptr, hidden, R, s, "k_rows_gemm_ff1"))) return rc;
      if ((rc = launch_dense_rows(h, t, hidden, 4 * d[t], p->ff[t].fc2, d[t], out[t], x[t], out[t], R, s, "k_rows_gemm_ff2"))) return rc;
    } else if (core_narrow_width(d[t]) && !(flags & GNX_FLAG_FORCE_GENERIC)) {
      if ((rc = launch_core_post(x[t], rows[t], d[t], p->ln2[t], p->ff[t], p->eps, p->eps_mode, out[t], s))) return rc;
    } else if ((rc = launch_ffn_residual(l2[t], x[t], rows[t], d[t], p->ff[t], out[t], s))) {
      return rc;
    }
  }
  return GNX_OK;
}

static int32_t pad_impl(const gnx_graphs* h, int32_t kind, bool pad, const float* src, int32_t d, int64_t R, float* dst, void* stream) {
  if (!h) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  if (kind != 0 && kind != 1) return fail(GNX_ERR_INVALID_ARG, "kind must be 0 (edges) or 1 (nodes)");
  // the packed side of a batch without edges has no rows (its buffer may be NULL, as for the forwards); the padded side always exists
  const bool packed_empty = (kind == 0 ? h->E : h->N) == 0;
  if ((pad ? !dst : !src) || ((pad ? !src : !dst) && !packed_empty)) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  if (d <= 0 || R <= 0) return fail(GNX_ERR_INVALID_ARG, "d and n_replicas must be >= 1");
  if (R > 1 && h->G != 1) return fail(GNX_ERR_INVALID_ARG, "n_replicas > 1 needs a single-graph handle");
  return launch_pad(h, kind, pad, src, d, R, dst, (hipStream_t)stream);
}

int32_t gnx_row_stats(const float* x, int64_t rows, int32_t d, float eps, int32_t eps_mode, float* stats, void* stream) {
  if (!x || !stats || rows < 0) return fail(GNX_ERR_INVALID_ARG, "gnx_row_stats: NULL argument / negative row count");
  if (eps_mode != 0 && eps_mode != 1) return fail(GNX_ERR_INVALID_ARG, "eps_mode must be 0 or 1");
  return launch_ln_stats(x, (size_t)rows, d, eps, eps_mode, stats, (hipStream_t)stream);
}

int32_t gnx_pad_features(const gnx_graphs* h, int32_t kind, const float* packed, int32_t d, int64_t R, float* padded, void* stream) {
  return pad_impl(h, kind, true, packed, d, R, padded, stream);
}

size_t gnx_xent_workspace_bytes(int64_t cols) { return cols > 0 ? sizeof(float) * (size_t)xent_blocks(cols) + 16 : 0; }

int32_t gnx_logit_cross_entropy(const float* logits, const float* targets, int32_t d, int64_t cols, float* loss_out, void* ws,
                                size_t ws_bytes, void* stream) {
  if (!logits || !targets || !loss_out) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  if (d <= 0 || cols <= 0) return fail(GNX_ERR_INVALID_ARG, "d and cols must be >= 1");
  if (!ws || ws_bytes < gnx_xent_workspace_bytes(cols)) return fail(GNX_ERR_WORKSPACE, "workspace missing or too small");
  return launch_xent(logits, targets, d, cols, loss_out, static_cast<float*>(ws), (hipStream_t)stream);
}

int32_t gnx_logit_cross_entropy_backward(const float* logits, const float* targets, int32_t d, int64_t cols, const float* upstream,
                                         float* d_logits, void* stream) {
  if (!logits || !targets || !upstream || !d_logits) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  if (d <= 0 || cols <= 0) return fail(GNX_ERR_INVALID_ARG, "d and cols must be >= 1");
  return launch_xent_backward(logits, targets, d, cols, upstream, d_logits, (hipStream_t)stream);
}

int32_t gnx_collapse_edges(const gnx_graphs* h, const float* ef, int32_t d, int64_t R, float* out, void* stream) {
  if (!h) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  if (d <= 0 || R <= 0 || (R > 1 && h->G != 1) || R > 65535) return fail(GNX_ERR_INVALID_ARG, "bad d / n_replicas");
  int32_t rc = gnx_ensure_collapse(h);
  if (rc) return rc;
  if (h->E == 0) return GNX_OK;  // a batch without edges: no collapsed column, nothing to write (ef / out may be NULL)
  if (!ef || !out) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  return launch_collapse(h, ef, d, R, out, (hipStream_t)stream);
}

int32_t gnx_collapse_padded(const gnx_graphs* h, const float* ef, int32_t d, int64_t R, float* out, void* stream) {
  if (!h || !out || (!ef && h->E > 0)) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  if (d <= 0 || R <= 0 || (R > 1 && h->G != 1) || R > 65535 || h->G > 65535) return fail(GNX_ERR_INVALID_ARG, "bad d / n_replicas / more than 65535 graphs");
  if ((size_t)h->PN * (h->PN + 1) / 2 * (size_t)d >= ((size_t)1 << 31) * 256) return fail(GNX_ERR_TOO_LARGE, "padded triangle too large");
  return launch_collapse_padded(h, ef, d, R, out, (hipStream_t)stream);
}

int32_t gnx_fn_input(const gnx_graphs* h, int32_t kind, const float* ef, int32_t de, const float* nf, int32_t dn, const float* gf,
                     int32_t dg, int64_t R, float* out, void* stream) {
  if (!h) return fail(GNX_ERR_INVALID_ARG, "NULL handle or output");
  if (kind < 0 || kind > 2) return fail(GNX_ERR_INVALID_ARG, "kind must be 0 (edge), 1 (node) or 2 (graph)");
  if (kind == 0 && h->E == 0 && de >= 0 && dn >= 0 && dg >= 0) return GNX_OK;  // the edge function input of a batch without edges has no rows (out may be NULL)
  if (!out) return fail(GNX_ERR_INVALID_ARG, "NULL handle or output");
  if (de < 0 || dn < 0 || dg < 0) return fail(GNX_ERR_DIMS, "negative feature width");
  if (!ef && h->E > 0) de = 0;  // (a batch without edges: its (de, 0) edge features have no buffer — the sums over them are rows of de zeros, as in the reference)
  if (!nf) dn = 0;
  if (!gf) dg = 0;
  if (de + dn + dg == 0) return fail(GNX_ERR_ALL_NOTHING, "ef, nf and gf are all nothing");
  if (kind >= 1 && de == 0) return fail(GNX_ERR_INVALID_ARG, "node / graph function inputs need the updated edge features (nodefninput.jl, graphfninput.jl)");
  if (kind == 2 && dn == 0) return fail(GNX_ERR_INVALID_ARG, "the graph function input needs the updated node features (graphfninput.jl:1-13)");
  if (R <= 0 || (R > 1 && h->G != 1) || R > 65535) return fail(GNX_ERR_INVALID_ARG, "bad n_replicas");
  return launch_fn_input(h, kind, ef, de, nf, dn, gf, dg, R, out, (hipStream_t)stream);
}

int32_t gnx_profile_calibrate(int32_t n, void* stream) { return launch_calibration(n, (hipStream_t)stream); }

int32_t gnx_unpad_features(const gnx_graphs* h, int32_t kind, const float* padded, int32_t d, int64_t R, float* packed, void* stream) {
  return pad_impl(h, kind, false, padded, d, R, packed, stream);
}

}  // extern "C"

// GNBlock whose update functions are Flux `Chain`s of `Dense` layers.
//
// The reference's GNBlock is a struct of three arbitrary Chains (src/gnblock.jl:1-6); its constructor builds `Chain(Dense)` for
// each (src/gnblock.jl:55-60) and that one-layer form is what gnx_block_forward fuses.  Users replace them by MLPs
// (`GNBlock(Chain(Dense(20 => 64, relu), Dense(64 => 3)), ...)`): this entry point runs such a block, composed from the pieces the
// library already has — no new kernels:
//   (a LayerNorm as the edge function's first layer: run behind an identity Dense — gnx_internal.h, ChainLnFirst)
//   edge function : its FIRST Dense is the edge update of a one-layer block (gnx_block_forward with node / graph outputs switched
//                   off: the fused / matrix-core edge kernels, the (K_e, E) input never materialised); further layers are row-wise
//                   Dense launches (k_rows_gemm) over [E][width] arrays that ping-pong in the workspace;
//   node function : getnodefninput (gnx_fn_input kind 1: [sum_{e->n} ef' ; nf ; gf_g], N rows) then row-wise Dense layers;
//   graph function: getgraphfninput (kind 2, G rows) then row-wise Dense layers.
// Semantics are exactly (m::GNBlock)(x) of src/gnblock.jl:63-69 with Chain update functions; zero-width outputs -> `nothing`.
#include <algorithm>

#include "gnx_internal.h"

namespace gnx {
int32_t launch_dense_rows(const gnx_graphs* h, int entity, const float* A, int K, const gnx_dense& d, int OUT, const float* add1,
                          const float* add2, float* out, int64_t R, hipStream_t s, const char* name);
int32_t launch_fn_input(const gnx_graphs* h, int kind, const float* ef, int de, const float* nf, int dn, const float* gf, int dg,
                        int64_t R, float* out, hipStream_t s);
}  // namespace gnx

using namespace gnx;

// A `LayerNorm(d)` layer value of a Chain (gnx_dense.kind = GNX_LAYER_LAYERNORM): y = gamma . (x - mean) / (sigma + eps) + beta per row, one
// wavefront per row, any width — the arithmetic of k_layernorm2 (gnx_generic.hip), i.e. of GNGraphNorm's LayerNorms (gngraphnorm.jl:19-26)
__global__ __launch_bounds__(256) void k_chain_layernorm(const float* __restrict__ x, size_t rows, int d, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, int eps_mode, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * d;
  float s = 0.f;
  for (int k = lane; k < d; k += 64) s += xr[k];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mu = s / (float)d;
  float v = 0.f;
  for (int k = lane; k < d; k += 64) { const float c = xr[k] - mu; v = fmaf(c, c, v); }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  v /= (float)d;
  const float inv = eps_mode == 0 ? 1.f / (sqrtf(v) + eps) : 1.f / sqrtf(v + eps);
  for (int k = lane; k < d; k += 64) y[row * d + k] = fmaf(gamma[k], (xr[k] - mu) * inv, beta[k]);
}

__global__ void k_chain_identity(float* __restrict__ w, int k) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)k * k) w[i] = (i / k == i % k) ? 1.f : 0.f;
}

namespace gnx {
int32_t ChainLnFirst::fill(float* ident, hipStream_t s) const {
  if (!on || ke == 0) return GNX_OK;
  GNX_LAUNCH(k_chain_identity, dim3((unsigned)(((size_t)ke * ke + 255) / 256)), dim3(256), 0, s, ident, ke);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}
// one layer of a Chain, row-wise over all rows of the entity: Dense (launch_dense_rows) or LayerNorm; shared with the backward's recompute
int32_t launch_chain_layer(const gnx_graphs* h, int entity, const gnx_dense& layer, const float* x, int k_in, int width, float* out, int64_t R, hipStream_t s,
                           const char* name) {
  if (!chain_layer_is_ln(layer)) return launch_dense_rows(h, entity, x, k_in, layer, width, nullptr, nullptr, out, R, s, name);
  const size_t rows = (size_t)R * (entity == 0 ? (size_t)h->E : (entity == 1 ? (size_t)h->N : (size_t)h->G));
  if (rows == 0 || width == 0) return GNX_OK;
  ProfScope ps("k_chain_layernorm", s);
  GNX_LAUNCH(k_chain_layernorm, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, rows, width, layer.weight, layer.bias, kChainLnEps, chain_layer_ln_mode(layer), out);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}
}  // namespace gnx

namespace {

struct ChainWs {
  size_t block_off, block_bytes;  // workspace of the one-layer edge block
  size_t e_buf[2], n_in, n_buf[2], g_in, g_buf[2];
  size_t ident;  // the identity layer of a LayerNorm-first edge function (ChainLnFirst)
  size_t total;
};

int max_width(const gnx_chain& c) {
  int m = 0;
  for (int i = 0; i < c.n_layers; ++i) m = std::max(m, c.widths[i]);
  return m;
}
int out_width(const gnx_chain& c) { return c.n_layers > 0 ? c.widths[c.n_layers - 1] : 0; }

int32_t check_chain(const gnx_chain& c, const char* what, int k_in) {
  if (c.n_layers < 0 || c.n_layers > kChainMaxLayers) return fail(GNX_ERR_INVALID_ARG, std::string(what) + ": n_layers must be 0..16");
  if (c.n_layers > 0 && (!c.layers || !c.widths)) return fail(GNX_ERR_INVALID_ARG, std::string(what) + ": layers / widths is NULL");
  for (int i = 0; i < c.n_layers; ++i) {
    if (c.widths[i] < 0) return fail(GNX_ERR_DIMS, std::string(what) + ": negative layer width");
    if (i + 1 < c.n_layers && c.widths[i] == 0) return fail(GNX_ERR_DIMS, std::string(what) + ": only the last layer of a Chain may have width 0");
    if (c.layers[i].act < GNX_ACT_IDENTITY || c.layers[i].act > GNX_ACT_GELU) return fail(GNX_ERR_INVALID_ARG, std::string(what) + ": unknown activation");
    const int kind = c.layers[i].kind & 0xff;
    if (kind != GNX_LAYER_DENSE && kind != GNX_LAYER_LAYERNORM) return fail(GNX_ERR_INVALID_ARG, std::string(what) + ": unknown layer kind");
    if (kind == GNX_LAYER_LAYERNORM) {  // a LayerNorm(d) layer value: gamma, beta of its input's width, no activation of its own
      if (c.widths[i] != (i > 0 ? c.widths[i - 1] : k_in)) return fail(GNX_ERR_DIMS, std::string(what) + ": a LayerNorm layer keeps the width of its input");
      if (c.layers[i].act != GNX_ACT_IDENTITY) return fail(GNX_ERR_INVALID_ARG, std::string(what) + ": a LayerNorm layer has no activation");
      if (c.widths[i] > 0 && (!c.layers[i].weight || !c.layers[i].bias)) return fail(GNX_ERR_INVALID_ARG, std::string(what) + ": a LayerNorm layer needs gamma and beta");
    }
  }
  return GNX_OK;
}

gnx_block_params edge_block(const gnx_chain_block_params* p) {  // the one-layer block that performs the edge function's first Dense
  gnx_block_params b{};
  b.de = p->de; b.dn = p->dn; b.dg = p->dg;
  b.oe = p->edgefn.widths[0]; b.on = 0; b.og = 0;
  b.edgefn = p->edgefn.layers[0];
  return b;
}

int32_t check_params(const gnx_graphs* h, const gnx_chain_block_params* p, int64_t R) {
  if (!h || !p) return fail(GNX_ERR_INVALID_ARG, "NULL handle or params");
  if (R <= 0 || (R > 1 && h->G != 1) || R > 65535) return fail(GNX_ERR_INVALID_ARG, "bad n_replicas");
  if (p->de < 0 || p->dn < 0 || p->dg < 0) return fail(GNX_ERR_DIMS, "negative feature width");
  if (p->de + p->dn + p->dg == 0) return fail(GNX_ERR_DIMS, "all input widths are 0 (gnblock.jl:48, batch.jl:56)");
  int32_t rc;
  if ((rc = check_chain(p->edgefn, "edgefn", p->de + 2 * p->dn + p->dg))) return rc;
  const int oe = out_width(p->edgefn);
  if ((rc = check_chain(p->nodefn, "nodefn", oe + p->dn + p->dg))) return rc;
  const int on = out_width(p->nodefn), og = out_width(p->graphfn);
  if ((rc = check_chain(p->graphfn, "graphfn", oe + on + p->dg))) return rc;
  if (oe + on + og == 0) return fail(GNX_ERR_DIMS, "all output widths are 0 (gnblock.jl:49)");
  // A zero-width ef' / nf' is an empty segment of the next function's input, exactly as for one-layer update functions
  // (gnblock.jl:63-69 computes getnodefninput / getgraphfninput over the 0-row h_ef; width 0 <=> nothing in launch_fn_input).
  return GNX_OK;
}

ChainWs layout(const gnx_graphs* h, const gnx_chain_block_params* p, int64_t R, size_t ident_floats) {
  ChainWs w{};
  size_t o = 0;
  auto take = [&](size_t floats) { const size_t at = o; o += align_up(sizeof(float) * floats, 256); return at; };
  const int oe = out_width(p->edgefn), on = out_width(p->nodefn), og = out_width(p->graphfn);
  w.block_off = 0; w.block_bytes = 0;
  if (oe > 0) {
    const gnx_block_params b = edge_block(p);
    w.block_bytes = gnx_block_workspace_bytes(h, &b, R);
    o = align_up(w.block_bytes, 256);
  }
  const size_t rows[3] = {(size_t)R * h->E, (size_t)R * h->N, (size_t)R * h->G};
  for (int i = 0; i < 2; ++i) w.e_buf[i] = take(p->edgefn.n_layers > 1 ? rows[0] * max_width(p->edgefn) : 0);
  w.n_in = take(on > 0 ? rows[1] * (size_t)(oe + p->dn + p->dg) : 0);
  for (int i = 0; i < 2; ++i) w.n_buf[i] = take(p->nodefn.n_layers > 1 ? rows[1] * max_width(p->nodefn) : 0);
  w.g_in = take(og > 0 ? rows[2] * (size_t)(oe + on + p->dg) : 0);
  for (int i = 0; i < 2; ++i) w.g_buf[i] = take(p->graphfn.n_layers > 1 ? rows[2] * max_width(p->graphfn) : 0);
  w.ident = take(ident_floats);
  w.total = o + 256;
  return w;
}

// layers [first, n) of a chain, row-wise: x [rows][k_in] -> ... -> out [rows][out_width]; intermediates ping-pong in buf[0/1]
int32_t run_layers(const gnx_graphs* h, int entity, const gnx_chain& c, int first, const float* x, int k_in, float* const buf[2], float* out,
                   int64_t R, hipStream_t s, const char* name) {
  const float* cur = x;
  int k = k_in;
  for (int i = first; i < c.n_layers; ++i) {
    float* dst = i + 1 == c.n_layers ? out : buf[(i - first) & 1];
    if (c.widths[i] > 0 && k > 0 && !c.layers[i].weight) return fail(GNX_ERR_INVALID_ARG, "Chain: Dense weight is NULL");
    const int32_t rc = launch_chain_layer(h, entity, c.layers[i], cur, k, c.widths[i], dst, R, s, name);
    if (rc) return rc;
    cur = dst;
    k = c.widths[i];
  }
  return GNX_OK;
}

}  // namespace

extern "C" {

size_t gnx_chain_block_workspace_bytes(const gnx_graphs* h, const gnx_chain_block_params* p0, int64_t R) {
  if (check_params(h, p0, R) != GNX_OK) return 0;
  ChainLnFirst lnf;
  const gnx_chain_block_params* p = lnf.init(p0, nullptr);
  // further layers of a chain are row-wise Dense launches on the matrix-core kernel: its tables are built here, outside any capture
  if (p->edgefn.n_layers > 1 || p->nodefn.n_layers > 1 || p->graphfn.n_layers > 1 || out_width(p->edgefn) == 0) (void)gnx_ensure_wide_tables(h);
  return layout(h, p, R, lnf.ident_floats()).total;
}

int32_t gnx_chain_block_forward(const gnx_graphs* h, const gnx_chain_block_params* p0, const float* ef, const float* nf, const float* gf, int64_t R,
                                float* ef_out, float* nf_out, float* gf_out, void* ws, size_t ws_bytes, uint32_t flags, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  gnx::FormScope forms(flags);  // the forms this call selected (gnx.h: GNX_FLAG_EDGE_FP32 ...)
  int32_t rc = check_params(h, p0, R);
  if (rc) return rc;
  ChainLnFirst lnf;
  {  // (the identity layer's weights sit at a fixed place of the workspace: the layout does not depend on the pointer)
    const gnx_chain_block_params* q = lnf.init(p0, nullptr);
    if (lnf.on && ws) lnf.layers[0].weight = reinterpret_cast<const float*>(static_cast<char*>(ws) + layout(h, q, R, lnf.ident_floats()).ident);
  }
  const gnx_chain_block_params* p = lnf.on ? &lnf.p : p0;
  const int de = p->de, dn = p->dn, dg = p->dg;
  const int oe = out_width(p->edgefn), on = out_width(p->nodefn), og = out_width(p->graphfn);
  bool chain_wide = de > 32 || dn > 32 || dg > 32;  // (any layer of any chain at matrix-core widths)
  for (const gnx_chain* c : {&p->edgefn, &p->nodefn, &p->graphfn})
    for (int i = 0; i < c->n_layers; ++i) chain_wide = chain_wide || (c->widths && c->widths[i] > 32);
  gnx::DeviceTurn turn(s, chain_wide);
  if ((de > 0 && !ef && h->E > 0) || (dn > 0 && !nf) || (dg > 0 && !gf)) return fail(GNX_ERR_INVALID_ARG, "an input with non-zero width is NULL (width 0 <=> nothing)");
  if ((oe > 0 && !ef_out && h->E > 0) || (on > 0 && !nf_out) || (og > 0 && !gf_out)) return fail(GNX_ERR_INVALID_ARG, "an output with non-zero width is NULL");
  const ChainWs w = layout(h, p, R, lnf.ident_floats());
  if (!ws || ws_bytes < w.total) return fail(GNX_ERR_WORKSPACE, "workspace missing or smaller than gnx_chain_block_workspace_bytes()");
  if (((uintptr_t)ws & 15) != 0) return fail(GNX_ERR_WORKSPACE, "workspace must be 16-byte aligned");
  char* base = static_cast<char*>(ws);
  auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
  if ((rc = lnf.fill(F(w.ident), s))) return rc;
  // ---- edge function (gnblock.jl:65): first Dense fused with getedgefninput, the rest row-wise ----
  if (oe > 0 && h->E > 0) {
    const gnx_block_params b = edge_block(p);
    float* first_out = p->edgefn.n_layers == 1 ? ef_out : F(w.e_buf[0]);
    if ((rc = gnx_block_forward(h, &b, ef, nf, gf, R, first_out, nullptr, nullptr, base + w.block_off, w.block_bytes, flags, stream))) return rc;
    if (p->edgefn.n_layers > 1) {
      float* const bufs[2] = {F(w.e_buf[1]), F(w.e_buf[0])};  // layer 1 reads e_buf[0], writes e_buf[1], ...
      if ((rc = run_layers(h, 0, p->edgefn, 1, first_out, p->edgefn.widths[0], bufs, ef_out, R, s, "k_rows_gemm_chain_e"))) return rc;
    }
  }
  // ---- node function (gnblock.jl:66): sees the NEW ef', the OLD nf and gf ----
  if (on > 0) {
    if ((rc = launch_fn_input(h, 1, ef_out, oe, nf, dn, gf, dg, R, F(w.n_in), s))) return rc;
    float* const bufs[2] = {F(w.n_buf[0]), F(w.n_buf[1])};
    if ((rc = run_layers(h, 1, p->nodefn, 0, F(w.n_in), oe + dn + dg, bufs, nf_out, R, s, "k_rows_gemm_chain_n"))) return rc;
  }
  // ---- graph function (gnblock.jl:67): NEW ef', NEW nf', OLD gf ----
  if (og > 0) {
    if ((rc = launch_fn_input(h, 2, ef_out, oe, nf_out, on, gf, dg, R, F(w.g_in), s))) return rc;
    float* const bufs[2] = {F(w.g_buf[0]), F(w.g_buf[1])};
    if ((rc = run_layers(h, 2, p->graphfn, 0, F(w.g_in), oe + on + dg, bufs, gf_out, R, s, "k_rows_gemm_chain_g"))) return rc;
  }
  return GNX_OK;
}

}  // extern "C"

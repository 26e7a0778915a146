// Fused GNBlock path for narrow feature widths (README-sized dims): ONE launch does the edge update, the
// edge->node segmented sum, the node update and the per-tile partial sums of the graph update; a second tiny launch
// finishes the graph update.  The kernels themselves live in gnx_wave_kernel.h (self-contained device code).
//
// Why it can be fused: the reference's edge order is CSC order (src/pad.jl:30 — sorted by destination), so the
// in-edges of a node range [n0, n1) are the contiguous edge range [colptr[n0], colptr[n1]).  A wavefront that owns
// a node tile therefore owns every edge that aggregates into it: ef' never has to be re-read from HBM, the
// edge->node sum (nodefninput.jl:3) needs no atomics, and its order is fixed.
//
// Widths are template parameters (fully unrolled FMAs, weights as scalar operands).  launch_block_narrow() first
// looks the width set up in the ahead-of-time list below; any other width set with every width <= 32 (and <= 1024 weights) is specialised
// at run time (gnx_jit.cpp: hiprtc on the same header text) — the analogue of Julia compiling a GNBlock for its own
// dims on first use.  1 ("not applicable") sends the caller on to the MFMA / generic kernels.
#include <cstdlib>

#include "gnx_device.h"
#include "gnx_wave_kernel.h"

namespace gnx {

// Rows of the partial-sum table per replica: one per workgroup (one graph) or one per wave tile (several graphs).
static int partial_rows(const gnx_graphs* h) { return (int)(h->G == 1 ? (h->n_wtiles() + 3) / 4 : h->n_wtiles()); }

// Threads of the graph update: one wavefront per graph while a graph has <= 256 partial rows (the usual heterogeneous batch:
// C3 has ~16 rows per graph, C5 ~3), else 256, and 1024 from 1024 rows on (C2: 2032 rows, two per thread in flight at once).
static int graph_update_threads(const gnx_graphs* h) {
  const int64_t rows = h->G == 1 ? (h->n_wtiles() + 3) / 4 : h->max_wtiles_per_graph;
  return rows <= 256 ? 64 : (rows >= 1024 ? 1024 : 256);
}

template <int DE, int DN, int DG, int OE, int ON, int EPT, bool LN, bool ONEG, bool FFE = false, bool CHAIN = false>
static int32_t launch_wave_g(const gnx_graphs* h, const BlockArgs& a, int64_t R, hipStream_t s, int phase) {
  constexpr int C = OE + ON;
  const int n_rows = partial_rows(h);
  const unsigned grid = (unsigned)((a.n_wtiles + 3) / 4) + (CHAIN ? (unsigned)a.prev_blocks : 0u);
  if (phase & 1) {
    ProfScope ps("k_block_wave", s);
#ifdef GNX_WAVE_STAMPS_BUILD  // diagnostic build: GNX_WAVE_STAMPS_DUMP=<file> writes [wave tile][8] shader-clock stamps of every (eager) launch
    static unsigned long long* d_dbg = nullptr;
    static size_t dbg_cap = 0;
    const char* dump = getenv("GNX_WAVE_STAMPS_DUMP");
    if (dump) {
      if (dbg_cap < (size_t)a.n_wtiles) { if (d_dbg) (void)hipFree(d_dbg); dbg_cap = (size_t)a.n_wtiles; (void)hipMalloc((void**)&d_dbg, dbg_cap * 64); }
      (void)hipMemsetAsync(d_dbg, 0, (size_t)a.n_wtiles * 64, s);
      (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_wave_dbg), &d_dbg, sizeof(d_dbg), 0, hipMemcpyHostToDevice, s);
    }
#endif
    if constexpr (FFE) GNX_LAUNCH((k_block_wave_ffe<DE, DN, DG, OE, ON, EPT, ONEG>), dim3(grid, (unsigned)R), dim3(kThreads), 0, s, a, n_rows);
    else GNX_LAUNCH((k_block_wave<DE, DN, DG, OE, ON, EPT, LN, ONEG, false, false, CHAIN>), dim3(grid, (unsigned)R), dim3(kThreads), 0, s, a, n_rows);
    GNX_HIP(hipGetLastError());
#ifdef GNX_WAVE_STAMPS_BUILD
    if (dump) {
      (void)hipStreamSynchronize(s);
      std::vector<unsigned long long> hs((size_t)a.n_wtiles * 8);
      (void)hipMemcpy(hs.data(), d_dbg, hs.size() * 8, hipMemcpyDeviceToHost);
      if (FILE* f = fopen(dump, "wb")) { fwrite(hs.data(), 8, hs.size(), f); fclose(f); }
    }
#endif
  }
  if ((phase & 2) && a.og > 0) {
    if constexpr (C > 0) {
      const int threads = graph_update_threads(h);
      const size_t lds = sizeof(float) * (size_t)graph_update_lds_floats(C, a.dg, a.og, threads);
      ProfScope ps("k_graph_t", s);
      GNX_LAUNCH((k_graph_t<C, ONEG>), dim3((unsigned)a.G, (unsigned)R), dim3(threads), lds, s, a, n_rows);
      GNX_HIP(hipGetLastError());
    }
  }
  return GNX_OK;
}

// Batches of small graphs (every graph <= 8 wave tiles: the handle has a pack table): ONE launch — 512-thread workgroups that own whole
// graphs run the graph update themselves (k_block_wave<..., PACK>).  Only for the whole block in one call (phase 3): a caller that
// splits off the graph update, or a narrow GNCore that runs it inside its FeedForward launch, reads the partial rows of the two-launch form.
// GNX_FLAG_NO_PACK keeps the two launches.
template <int DE, int DN, int DG, int OE, int ON, int EPT>
static bool launch_wave_pack(const gnx_graphs* h, const BlockArgs& a, int64_t R, hipStream_t s, int phase, int32_t* rc) {
  constexpr int C = OE + ON;
  if constexpr (EPT != 2 || C == 0) return false;
  else {
    if (h->G <= 1 || h->n_packs <= 0 || !a.packs || phase != 3 || a.og <= 0 || form(GNX_FLAG_NO_PACK)) return false;
    ProfScope ps("k_block_wave", s);
    GNX_LAUNCH((k_block_wave<DE, DN, DG, OE, ON, EPT, false, false, true>), dim3((unsigned)h->n_packs, (unsigned)R), dim3(kPackThreads), 0, s, a, 0);
    const hipError_t e = hipGetLastError();
    *rc = e == hipSuccess ? GNX_OK : hip_fail(e, "k_block_wave<PACK>");
    return true;
  }
}

template <int DE, int DN, int DG, int OE, int ON, int EPT, bool LN = false>
static int32_t launch_wave_t(const gnx_graphs* h, const BlockArgs& a, int64_t R, hipStream_t s, int phase) {
  if constexpr (!LN) {
    int32_t rc = GNX_OK;
    if (launch_wave_pack<DE, DN, DG, OE, ON, EPT>(h, a, R, s, phase, &rc)) return rc;
  }
  return h->G == 1 ? launch_wave_g<DE, DN, DG, OE, ON, EPT, LN, true>(h, a, R, s, phase) : launch_wave_g<DE, DN, DG, OE, ON, EPT, LN, false>(h, a, R, s, phase);
}

template <int DE, int DN, int DG, int OE, int ON>
static int32_t launch_fused(const gnx_graphs* h, const BlockArgs& a, int64_t R, hipStream_t s, int phase) {
  // the kernel's EPT must match the handle's wave-tile edge cap (GNX_WTILE_E at handle creation: 64, 128 or 256)
  if (h->wtile_e_cap == 64) return launch_wave_t<DE, DN, DG, OE, ON, 1>(h, a, R, s, phase);
  if (h->wtile_e_cap == 128) return launch_wave_t<DE, DN, DG, OE, ON, 2>(h, a, R, s, phase);
  if (h->wtile_e_cap == 256) {
    if constexpr ((DE + DN) * 4 <= 64) return launch_wave_t<DE, DN, DG, OE, ON, 4>(h, a, R, s, phase);
  }
  return 1;
}

// Instantiated width sets.  (de, dn, dg) => (oe, on); og is free (the graph update is its own small kernel).
#define GNX_NARROW_DIMS(X) \
  X(10, 5, 0, 3, 4)        \
  X(3, 4, 5, 3, 4)         \
  X(10, 5, 3, 3, 4)        \
  X(0, 2, 0, 2, 2)         \
  X(2, 2, 2, 2, 2)         \
  X(4, 3, 2, 3, 4)         \
  X(8, 8, 8, 16, 8)        \
  X(10, 5, 3, 10, 5)       \
  X(10, 5, 0, 10, 5)

bool jit_eligible(const BlockArgs& a, int ept);
int32_t jit_get(const BlockArgs& a, int ept, hipStream_t s, hipFunction_t* block, hipFunction_t* graph);

// Same launch geometry as launch_wave_t, kernels specialised at run time (gnx_jit.cpp) for this width set.
static int32_t launch_wave_jit(const gnx_graphs* h, const BlockArgs& a, int64_t R, hipStream_t s, int phase) {
  const int ept = h->wtile_e_cap / 64;
  if (ept * 64 != h->wtile_e_cap || (ept != 1 && ept != 2 && ept != 4)) return 1;
  hipFunction_t fb = nullptr, fg = nullptr;
  const int32_t rc = jit_get(a, ept, s, &fb, &fg);
  if (rc) return rc;
  const int C = a.oe + a.on;
  BlockArgs aa = a;
  int n_rows = partial_rows(h);
  void* params[] = {&aa, &n_rows};
  if (phase & 1) {
    ProfScope ps("k_block_wave", s);
    GNX_HIP(module_launch(fb, (unsigned)((a.n_wtiles + 3) / 4), (unsigned)R, 1, kThreads, 1, 1, 0, s, params));
  }
  if ((phase & 2) && a.og > 0) {
    const int threads = graph_update_threads(h);
    const size_t lds = sizeof(float) * (size_t)graph_update_lds_floats(C, a.dg, a.og, threads);
    ProfScope ps("k_graph_t", s);
    GNX_HIP(module_launch(fg, (unsigned)a.G, (unsigned)R, 1, threads, 1, 1, (unsigned)lds, s, params));
  }
  return GNX_OK;
}

// compiles + loads the run-time specialised kernels of this width set ahead of the first forward (called from
// gnx_block_workspace_bytes, which every caller runs before a forward and never inside a stream capture)
void warm_block_narrow(const gnx_graphs* h, const gnx_block_params* p) {
  BlockArgs a{};
  a.de = p->de; a.dn = p->dn; a.dg = p->dg; a.oe = p->oe; a.on = p->on; a.og = p->og;
  a.G = (int)h->G;  // selects the one-graph / several-graphs variant of the kernel
#define GNX_CASE(DE, DN, DG, OE, ON) \
  if (a.de == DE && a.dn == DN && a.dg == DG && a.oe == OE && a.on == ON) return;
  static const bool jit_all = getenv("GNX_JIT_ALL") != nullptr;  // (diagnostic: specialise even the ahead-of-time width sets; read once)
  if (!jit_all) { GNX_NARROW_DIMS(GNX_CASE) }
#undef GNX_CASE
  if (h->n_wtiles() == 0 || h->E == 0) return;
  hipFunction_t fb, fg;
  (void)jit_get(a, h->wtile_e_cap / 64, nullptr, &fb, &fg);
}

static bool wants_ln(const BlockArgs& a) { return a.ln_g[0] || a.ln_g[1] || a.ln_g[2]; }

// LayerNorm-on-load variant (GNCore: block(gn1(x)) straight from x): ahead of time for the README ex.3 core widths at the
// default wave-tile size, any other eligible width set through the run-time specialiser.
static bool ln_aot(const gnx_graphs* h, const BlockArgs& a) {
  return a.de == 10 && a.dn == 5 && a.dg == 3 && a.oe == 10 && a.on == 5 && h->wtile_e_cap == 128;
}

// Can the fused kernel run this width set (with LayerNorm on load if a.ln_g is set) right now?  Compiles the run-time
// specialised kernel if needed — except while `s` is being captured.  The GNCore forward asks before it decides to skip the
// separate gn1 kernels.
bool block_narrow_ready(const gnx_graphs* h, const BlockArgs& a, hipStream_t s) {
  if (a.n_wtiles == 0 || a.E == 0) return false;
  if (wants_ln(a)) {
    if (ln_aot(h, a)) return true;
  } else {
#define GNX_CASE(DE, DN, DG, OE, ON) \
    if (a.de == DE && a.dn == DN && a.dg == DG && a.oe == OE && a.on == ON) return h->wtile_e_cap == 64 || h->wtile_e_cap == 128 || h->wtile_e_cap == 256;
    GNX_NARROW_DIMS(GNX_CASE)
#undef GNX_CASE
  }
  const int ept = h->wtile_e_cap / 64;
  if (ept * 64 != h->wtile_e_cap || (ept != 1 && ept != 2 && ept != 4)) return false;
  hipFunction_t fb, fg;
  return jit_get(a, ept, s, &fb, &fg) == GNX_OK;
}

// gnx_block_forward_chained: can the previous call's graph update ride at the front of this call's block kernel?  The two-launch form of
// an ahead-of-time width set at the default wave-tile size, graph function small enough for the kernel's LDS, <= 256 partial rows per
// graph (one wavefront per graph) or one graph.  (Batches that take the pack form run their graph update inside the kernel already.)
static bool pack_form(const gnx_graphs* h, const BlockArgs& a) { return h->G > 1 && h->n_packs > 0 && a.packs && a.og > 0 && !getenv("GNX_NO_PACK"); }
bool block_narrow_chain_applies(const gnx_graphs* h, const BlockArgs& a) {
  if (a.n_wtiles == 0 || a.E == 0 || a.og <= 0 || h->wtile_e_cap != 128 || wants_ln(a) || a.ffe_w1 || pack_form(h, a)) return false;
  const int C = a.oe + a.on;
  if (C <= 0) return false;
  const int wsl = wave_slice_floats(a.oe, 2);
  if (h->G == 1) { if (graph_update_lds_floats(C, a.dg, a.og, kThreads) > 4 * wsl) return false; }
  else if (h->max_wtiles_per_graph > 256 || graph_update_lds_floats(C, a.dg, a.og, 64) > wsl) return false;
#define GNX_CASE(DE, DN, DG, OE, ON) \
  if (a.de == DE && a.dn == DN && a.dg == DG && a.oe == OE && a.on == ON) return true;
  GNX_NARROW_DIMS(GNX_CASE)
#undef GNX_CASE
  return false;
}
// the edge + node update of THIS call with a.prev_* (the previous call's pending graph update) at the front of the same launch
int32_t launch_block_narrow_chained(const gnx_graphs* h, const BlockArgs& a0, int64_t R, hipStream_t s) {
  BlockArgs a = a0;
  a.prev_blocks = a.prev_partials ? (h->G == 1 ? 1 : (int)((h->G + 3) / 4)) : 0;
#define GNX_CASE(DE, DN, DG, OE, ON)                                                                                               \
  if (a.de == DE && a.dn == DN && a.dg == DG && a.oe == OE && a.on == ON) {                                                         \
    if constexpr (OE + ON > 0) {                                                                                                   \
      return h->G == 1 ? launch_wave_g<DE, DN, DG, OE, ON, 2, false, true, false, true>(h, a, R, s, 1)                              \
                       : launch_wave_g<DE, DN, DG, OE, ON, 2, false, false, false, true>(h, a, R, s, 1);                            \
    }                                                                                                                              \
  }
  GNX_NARROW_DIMS(GNX_CASE)
#undef GNX_CASE
  return fail(GNX_ERR_INVALID_ARG, "internal: chained launch for a width set without that kernel");
}

// the edge FeedForward + residual of a narrow GNCore inside the block kernel (k_block_wave<..., FFE>): ahead-of-time widths, identity / relu
// activations.  GNX_FLAG_NO_FFE keeps the FeedForward in k_core_post3.
bool block_narrow_ffe_applies(const gnx_graphs* h, const BlockArgs& a, int act1, int act2) {
  return ln_aot(h, a) && a.ln_g[0] && act1 <= GNX_ACT_RELU && act2 <= GNX_ACT_RELU && a.act_e <= GNX_ACT_RELU && h->max_in_degree <= h->wtile_e_cap &&
         !form(GNX_FLAG_NO_FFE);  // (max_in_degree: every wave tile is ONE chunk of edges — the kernel runs the FeedForward once, at its end)
}

int32_t launch_block_narrow(const gnx_graphs* h, const BlockArgs& a, int64_t R, hipStream_t s, int phase) {
  if (a.n_wtiles == 0 || a.E == 0) return 1;
  if (wants_ln(a)) {  // only reached after block_narrow_ready(): a miss here would silently drop the LayerNorm
    if (a.ffe_w1) {   // (set by gnx_core_forward only after block_narrow_ffe_applies())
      if (!ln_aot(h, a)) return fail(GNX_ERR_INVALID_ARG, "internal: FeedForward-in-the-edge-lanes requested for a width set without that kernel");
      return h->G == 1 ? launch_wave_g<10, 5, 3, 10, 5, 2, true, true, true>(h, a, R, s, phase) : launch_wave_g<10, 5, 3, 10, 5, 2, true, false, true>(h, a, R, s, phase);
    }
    if (ln_aot(h, a)) return launch_wave_t<10, 5, 3, 10, 5, 2, true>(h, a, R, s, phase);
    const int32_t rc = launch_wave_jit(h, a, R, s, phase);
    return rc == 1 ? fail(GNX_ERR_INVALID_ARG, "internal: LayerNorm-on-load requested but the fused kernel is not available") : rc;
  }
  static const bool jit_all = getenv("GNX_JIT_ALL") != nullptr;  // testing: run-time specialise even the listed width sets
  if (jit_all && launch_wave_jit(h, a, R, s, phase) == GNX_OK) return GNX_OK;
  // 16-B vector copies assume fp32-aligned buffers (always true for fp32 arrays); nothing else is required
#define GNX_CASE(DE, DN, DG, OE, ON) \
  if (a.de == DE && a.dn == DN && a.dg == DG && a.oe == OE && a.on == ON) return launch_fused<DE, DN, DG, OE, ON>(h, a, R, s, phase);
  GNX_NARROW_DIMS(GNX_CASE)
#undef GNX_CASE
  return launch_wave_jit(h, a, R, s, phase);  // any other narrow width set: compiled on first use (1 if not eligible)
}

}  // namespace gnx

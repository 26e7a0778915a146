// Internal declarations shared by the translation units of libgnx.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#if defined(__HIPCC__)
#include <hip/hip_ext.h>  // hipExtLaunchKernelGGL (device-compiler only: the host sanitizer build of the .cpp units does not launch)
#endif
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "gnx.h"

#include "gnx_device.h"  // gnx::Tile, gnx::BlockArgs

namespace gnx {
// std::vector whose resize() leaves trivially constructible elements uninitialised: the O(N + E) arrays of a handle are written in full by the
// pass that follows their allocation — value-initialising them first is a second sweep over (and the first touch of) 19 MB for a 1M-edge batch
template <class T>
struct uninit_alloc : std::allocator<T> {
  template <class U> struct rebind { using other = uninit_alloc<U>; };
  uninit_alloc() = default;
  template <class U> uninit_alloc(const uninit_alloc<U>&) {}
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
  }
};
using vec_i64 = std::vector<int64_t, uninit_alloc<int64_t>>;
using vec_i32 = std::vector<int32_t, uninit_alloc<int32_t>>;
// process-wide serial number of a handle: an address can be reused by a later handle (malloc recycles), a serial cannot — caches that bake a
// handle's device tables into a captured launch sequence key on it (gnx_dist.hip)
inline uint64_t next_handle_serial() {
  static std::atomic<uint64_t> counter{0};
  return ++counter;
}
}  // namespace gnx

struct gnx_graphs {
  int64_t G = 0, N = 0, E = 0, PN = 0;
  int64_t max_in_degree = 0;
  int device = 0;
  const uint64_t serial = gnx::next_handle_serial();
  // host copies (int64, 0-based, global)
  std::vector<int64_t> h_node_off, h_edge_off;
  gnx::vec_i64 h_colptr, h_rowval;
  gnx::vec_i32 t_colptr32, t_rowval32;  // transient: the device-format copies a constructor already made (uploaded as they are, then freed)
  void* d_arena = nullptr;  // ONE device allocation behind the arrays every handle has (colptr .. wave tiles below): a dozen hipMalloc / hipFree pairs cost ~3 ms per handle
  size_t arena_bytes = 0;   // its size (a destroyed handle's arena goes to a small process-wide cache: gnx_build_csc.hip)
  // A handle built on the device (gnx_build_csc.hip) has no host copy of colptr / rowval / the tile tables: the counts are kept, the
  // int64 host copies (accessors, collapse / CSR / matrix-core table builders) are downloaded on first use (gnx_ensure_host_csc)
  bool csc_on_device_only = false;
  mutable std::mutex host_csc_mu;
  int64_t n_tiles_ = 0, n_wtiles_ = 0;
  std::vector<gnx::Tile> h_tiles;
  std::vector<int32_t> h_tile_off;  // [G+1] tiles of graph g = [tile_off[g], tile_off[g+1])
  // wave tiles: same idea at wavefront granularity (<= wtile_e_cap edges, <= 64 nodes), one wave64 per tile
  std::vector<gnx::Tile> h_wtiles;
  std::vector<int32_t> h_wtile_off;
  // device copies (int32)
  int32_t* d_colptr = nullptr;    // [N+1]
  int32_t* d_rowval = nullptr;    // [E] global source node id
  int32_t* d_node_off = nullptr;  // [G+1]
  int32_t* d_edge_off = nullptr;  // [G+1]
  int32_t* d_tile_off = nullptr;  // [G+1]
  gnx::Tile* d_tiles = nullptr;   // [n_tiles]
  int32_t* d_wtile_off = nullptr; // [G+1]
  // wide (MFMA) path: destination node of every edge, and 128-row chunks of each graph's edges / nodes.  The COUNTS (what workspace
  // sizes need) are computed with the handle; the tables themselves — O(N + E) host loops and a dozen uploads that a batch of narrow
  // blocks never reads — are built on first use (gnx_ensure_wide_tables: every launcher that reads them calls it; thread-safe).
  int64_t n_etiles = 0, n_ntiles = 0, n_gtiles = 0;
  int64_t agg_rows_bound = 0;     // >= n_agg_rows, known without the tables: non-empty nodes + one row per chunk
  std::vector<int32_t> h_etile_off, h_ntile_off;  // [G+1] (eager: O(G))
  mutable std::mutex wide_mu;                 // serialises the build; a failed build leaves nothing behind and may be retried
  mutable std::atomic<bool> wide_built{false};
  mutable void* d_wide_arena = nullptr;   // device-built wide tables: ONE allocation behind the ten arrays below (then they are not freed one by one)
  mutable int32_t* d_edge_dst = nullptr;  // [E]
  mutable std::vector<gnx::Tile> h_etiles, h_ntiles, h_gtiles;  // h_gtiles: 128-row chunks of the graph rows (n0/n1 = graph ids)
  mutable gnx::Tile* d_etiles = nullptr;
  mutable gnx::Tile* d_ntiles = nullptr;
  mutable gnx::Tile* d_gtiles = nullptr;
  mutable int32_t* d_etile_off = nullptr;
  mutable int32_t* d_ntile_off = nullptr;
  // wide path, edge->node aggregation inside the edge GEMM: an aggregation CHUNK is a 64-row pass of an edge tile (chunk
  // 2t + pass); every chunk writes one partial-sum row per distinct destination it holds, rows [chunk_row0[c], chunk_row0[c+1])
  // of a table of n_agg_rows rows.  A node's sum = its row in its first chunk (+ row 0 of each further chunk it runs into).
  mutable int32_t* d_chunk_row0 = nullptr;       // [2 * n_etiles + 1]
  mutable int32_t* d_node_agg_row = nullptr;     // [N] row of the node's first partial, -1 for a node without in-edges
  mutable int32_t* d_node_agg_parts = nullptr;   // [N] number of chunks the node's in-edges run through
  mutable int32_t* d_node_agg_chunk = nullptr;   // [N] its first chunk
  mutable int64_t n_agg_rows = 0;
  // edge tiles whose destinations span more than gnx::kPdRowsCap consecutive nodes (the edge GEMM stages a tile's destination
  // projections in LDS when they fit: launch_block_wide picks that kernel when nearly every tile qualifies)
  mutable int64_t n_etiles_wide_span = 0;
  gnx::Tile* d_wtiles = nullptr;  // [n_wtiles]
  // graph-aligned packs of wave tiles (batches whose graphs all have <= 8 wave tiles): [n_packs][8] tile ids, -1 = empty slot; a graph's
  // tiles sit in adjacent slots of ONE pack (best-fit decreasing over the graphs).  n_packs = 0: not applicable.
  int32_t* d_packs = nullptr;
  int32_t n_packs = 0;
  int32_t wtile_e_cap = 0;
  int32_t max_wtiles_per_graph = 0;
  int32_t tile_e_cap = 0, tile_n_cap = 0;
  // edge collapsing tables (built on first use; the handle stays logically immutable)
  mutable std::once_flag collapse_once;
  mutable std::vector<int64_t> h_collapse_off;  // [G+1]
  mutable int32_t* d_collapse_edge = nullptr;   // [n_collapsed] edge id of i->j (i >= j)
  mutable int32_t* d_collapse_rev = nullptr;    // [n_collapsed] edge id of j->i, or -1
  mutable int32_t collapse_rc = 0;
  // CSR view for the backward pass (out-edges of every node, edge ids in CSC numbering), built on first use
  mutable std::once_flag csr_once;
  mutable int32_t* d_csr_ptr = nullptr;  // [N+1]
  mutable int32_t* d_csr_eid = nullptr;  // [E]
  mutable int32_t csr_rc = 0;
  // side streams for the small graph-level launches of a wide GNCore (forked behind the edge / node FeedForward and joined before the
  // core returns; created by gnx_core_workspace_bytes, i.e. outside any stream capture).  A POOL: a forward takes a free (stream, fork event,
  // join event) set for the time it ENQUEUES its work (an event may be re-recorded once the waits on it have been enqueued: a wait binds to the
  // record that precedes it), so concurrent forwards on one handle — distinct buffers, distinct streams — each get their side stream; a caller
  // that finds every set taken runs on its own stream alone.  Apart from the lazily built tables (mutex-guarded) a handle is immutable.
  static constexpr int kAuxSets = 4;
  struct AuxSet { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; std::mutex mu; };
  mutable std::once_flag aux_once;
  mutable AuxSet aux[kAuxSets];
  int64_t n_tiles() const { return n_tiles_; }
  int64_t n_wtiles() const { return n_wtiles_; }
};

namespace gnx {

// forms of the forward selected per call (gnx.h: GNX_FLAG_FFN_FP32 ...), defaults from the environment read once per process (gnx_forms.cpp)
uint32_t env_form_flags();
struct FormScope {  // an exported forward opens one: the functions below it read the call's forms with form()
  explicit FormScope(uint32_t call_flags);
  ~FormScope();
  FormScope(const FormScope&) = delete;
  FormScope& operator=(const FormScope&) = delete;
  uint32_t prev;
};
// a Chain's layer entries (include/gnx.h: gnx_dense.kind): Dense or a LayerNorm(d) layer value (gamma, beta in weight, bias; eps = Flux's default)
constexpr float kChainLnEps = 1e-5f;
inline bool chain_layer_is_ln(const gnx_dense& l) { return (l.kind & 0xff) == GNX_LAYER_LAYERNORM; }
inline int chain_layer_ln_mode(const gnx_dense& l) { return (l.kind & GNX_LAYER_LN_SQRT_EPS) ? 1 : 0; }
int32_t launch_chain_layer(const gnx_graphs* h, int entity, const gnx_dense& layer, const float* x, int k_in, int width, float* out, int64_t R, hipStream_t s,
                           const char* name);  // gnx_chain.cpp
// An edge function whose FIRST layer is a LayerNorm, `Chain(LayerNorm(K_e), Dense ...)`: the chain entry points fuse a chain's first layer with
// getedgefninput (the (K_e, E) input is never materialised) and that layer must be a Dense — so such a chain runs as Chain(Dense(I), LayerNorm, ...):
// the identity layer's output IS the function input, bit for bit (x . 1 + 0 in fp32 and in the six-term split: h + m + l = x exactly), and both
// the forward and the pullback (the scatter of the input gradient into d_ef / d_nf[src] / d_nf[dst] / d_gf included) stay those of a Dense-first
// chain.  `ident` = K_e x K_e floats in the call's workspace, written by fill() on the call's stream.  Not copyable: p.edgefn points into it.
constexpr int kChainMaxLayers = 16;
struct ChainLnFirst {
  bool on = false;
  int ke = 0;
  gnx_dense layers[kChainMaxLayers + 1];
  int32_t widths[kChainMaxLayers + 1];
  gnx_dense_grad grads_e[kChainMaxLayers + 1];
  gnx_chain_block_params p;
  gnx_chain_block_grads grads;
  ChainLnFirst() = default;
  ChainLnFirst(const ChainLnFirst&) = delete;
  ChainLnFirst& operator=(const ChainLnFirst&) = delete;
  static bool applies(const gnx_chain_block_params* p0) {
    return p0 && p0->edgefn.n_layers > 0 && p0->edgefn.n_layers <= kChainMaxLayers && p0->edgefn.layers && p0->edgefn.widths && chain_layer_is_ln(p0->edgefn.layers[0]);
  }
  // the parameters the entry point works on: *p0 itself, or the rewritten chain (ident may be NULL for a layout query)
  const gnx_chain_block_params* init(const gnx_chain_block_params* p0, const float* ident) {
    if (!applies(p0)) return p0;
    on = true;
    ke = p0->de + 2 * p0->dn + p0->dg;
    p = *p0;
    layers[0] = gnx_dense{ident, nullptr, GNX_ACT_IDENTITY, GNX_LAYER_DENSE};
    widths[0] = ke;
    for (int i = 0; i < p0->edgefn.n_layers; ++i) { layers[i + 1] = p0->edgefn.layers[i]; widths[i + 1] = p0->edgefn.widths[i]; }
    p.edgefn.layers = layers; p.edgefn.widths = widths; p.edgefn.n_layers = p0->edgefn.n_layers + 1;
    return &p;
  }
  const gnx_chain_block_grads* init_grads(const gnx_chain_block_grads* g0) {  // (after init(): the caller's gradient slots, shifted behind the identity layer's empty one)
    if (!on || !g0) return g0;
    grads = *g0;
    if (g0->edgefn) {
      grads_e[0] = gnx_dense_grad{nullptr, nullptr};
      for (int i = 0; i + 1 < p.edgefn.n_layers; ++i) grads_e[i + 1] = g0->edgefn[i];
      grads.edgefn = grads_e;
    }
    return &grads;
  }
  size_t ident_floats() const { return on ? (size_t)ke * ke : 0; }
  int32_t fill(float* ident, hipStream_t s) const;  // gnx_chain.cpp
};
unsigned lds_pad_bytes();  // experiment switch GNX_LDS_PAD_KB: dynamic LDS added to EVERY launch (0 by default) — a workgroup that owns most of a CU's LDS shares the CU with no other LDS-using kernel
bool form(uint32_t bit);  // is the form selected for the call this thread is in (outside a call: by the environment's defaults)

// Optional turn-taking of matrix-core calls on a DEVICE (gnx_forms.cpp; OFF by default from round 6 on, GNX_TAKE_TURNS=1 switches it on).  Round 5
// saw k_rows_gemm / k_ffn_fused return wrong values — row pairs off by ~1 % — while a dense bf16 matrix kernel ran on ANOTHER stream of the device (the
// library's own k_edge_x6 or a hipBLASLt GEMM), blamed the fp32 matrix instruction and serialised the library's calls at matrix-core widths: a per-device
// lock over the enqueue section, a wait for the previous such call's end when that ran on another stream, an event at the call's end.  Round 6 found the
// site (profiles/r06_overlap_hazard.log): the LayerNorm-on-load branch of those kernels consumed an LDS read right behind the compiler's counted wait,
// and on a CU shared with another kernel's workgroups the last 16 lanes of a wave got the previous values; with the guard there (GNX_LN_GUARD) every
// scenario that failed is exact with calls overlapping (2 640 forward + backward runs and graph replays, the thread probes: 2 040 forwards).
inline bool matrix_core_widths(const gnx_block_params& b) {  // (the fused narrow kernels take widths up to 32: no matrix instruction)
  return b.de > 32 || b.dn > 32 || b.dg > 32 || b.oe > 32 || b.on > 32 || b.og > 32;
}
struct DeviceTurn {
  DeviceTurn(hipStream_t s, bool matrix_core_widths);
  ~DeviceTurn();
  DeviceTurn(const DeviceTurn&) = delete;
  DeviceTurn& operator=(const DeviceTurn&) = delete;
  void* chain = nullptr;  // the device's chain while held
  hipStream_t stream = nullptr;
  bool record = false;
};

// Prepared parameters (gnx.h: gnx_block_prepare / gnx_core_prepare; gnx_prepare.cpp): the weight blocks of a layer in the forms the six-term
// kernels stage — split into bf16 planes, transposed, slot-permuted — made ONCE when the weights are uploaded instead of by a *_prep launch in
// front of every forward.  An exported forward publishes the layer's prepared object to the launchers below it (PreparedScope); a launcher asks
// for the planes of the very weight pointers it was handed (prepared_planes: a miss — other weights, another device, no prepared object —
// means "run the prep launch into the workspace as before").
enum PreparedKind : int32_t { PREP_EDGE = 1, PREP_PROJ = 2, PREP_FFN = 3, PREP_ENC = 4, PREP_NODE = 5 };
struct PreparedScope {
  explicit PreparedScope(const gnx_prepared* q);
  ~PreparedScope();
  PreparedScope(const PreparedScope&) = delete;
  PreparedScope& operator=(const PreparedScope&) = delete;
  const gnx_prepared* prev;
};
const void* prepared_planes(PreparedKind kind, const void* w0, const void* w1, int32_t n);

void set_error(const std::string& msg);
int32_t fail(int32_t code, const std::string& msg);
int32_t hip_fail(hipError_t e, const char* what);

#define GNX_HIP(expr)                                        \
  do {                                                       \
    hipError_t _e = (expr);                                  \
    if (_e != hipSuccess) return gnx::hip_fail(_e, #expr);   \
  } while (0)

// profiling (gnx_profile.cpp): per-kernel DISPATCH timestamps.  A ProfScope names the launches made inside it on this thread; GNX_LAUNCH /
// GNX_MODULE_LAUNCH then attach a start / stop event pair to each kernel's own dispatch packet (hipExtLaunchKernel): the pair's elapsed
// time is the kernel's begin -> end as the command processor stamps it — the figure rocprofv3's kernel trace reports — and not a bracket
// of marker packets around the launch (round 3: such brackets read 8-14 % above rocprofv3 on the same launches; every marker is a
// barrier with its own cache maintenance, and its cost grows with the kernel it waits for).  Off (the default): plain launches.
struct ProfScope {
  ProfScope(const char* name, hipStream_t s);
  ~ProfScope();
  const char* prev;
};
// true (and a fresh event pair, recorded under the innermost ProfScope of this thread) when per-kernel timing is on and a scope is open
bool prof_take_events(hipEvent_t* start, hipEvent_t* stop);

#if defined(__HIPCC__)
#define GNX_LAUNCH(kernel, grid, block, lds, stream, ...)                                                              \
  do {                                                                                                                 \
    hipEvent_t gnx_e0_ = nullptr, gnx_e1_ = nullptr;                                                                   \
    if (gnx::prof_take_events(&gnx_e0_, &gnx_e1_))                                                                     \
      hipExtLaunchKernelGGL(kernel, grid, block, (lds) + gnx::lds_pad_bytes(), stream, gnx_e0_, gnx_e1_, 0, ##__VA_ARGS__); \
    else                                                                                                               \
      hipLaunchKernelGGL(kernel, grid, block, (lds) + gnx::lds_pad_bytes(), stream, ##__VA_ARGS__);                     \
  } while (0)
// run-time compiled kernels (hipFunction_t): grid in BLOCKS like hipModuleLaunchKernel (the Ext form counts work-items)
inline hipError_t module_launch(hipFunction_t f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, unsigned lds, hipStream_t s,
                                void** params) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof_take_events(&e0, &e1)) return hipExtModuleLaunchKernel(f, gx * bx, gy * by, gz * bz, bx, by, bz, lds, s, params, nullptr, e0, e1, 0);
  return hipModuleLaunchKernel(f, gx, gy, gz, bx, by, bz, lds, s, params, nullptr);
}
#endif  // __HIPCC__

bool profile_enabled();  // per-kernel timing is on: callers keep everything on one stream (overlapped kernels would share their time)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// graph-aligned packs of wave tiles from h->h_wtile_off (gnx_graphs.cpp; both builders): fills `packs` ([n_packs][8]) and h->n_packs
void build_packs(gnx_graphs* h, std::vector<int32_t>& packs);
// released handle arenas (gnx_build_csc.hip)
void* arena_take(int dev, size_t bytes, size_t* got);
void arena_give(int dev, void* ptr, size_t bytes);
struct DenseCscOnDevice {  // gnx_build_device.hip: the CSC of a dense batch, left on the device inside two cached blocks (release_dense_csc)
  int32_t* d_colptr = nullptr; int32_t* d_rowval = nullptr; int64_t E = 0; std::vector<int32_t> edge_off;
  int device = 0; void* block = nullptr; size_t block_bytes = 0; void* rv_block = nullptr; size_t rv_bytes = 0;
};
void release_dense_csc(DenseCscOnDevice& k);
int32_t build_wide_tables_on_device(const gnx_graphs* h);  // gnx_build_csc.hip; 1 = not applicable (host builder)
int32_t build_handle_from_csc_on_device(gnx_graphs* h, const void* colptr_cat, const void* rowval_cat, int32_t index_base, int32_t index_bits, int tile_e_cap, int tile_n_cap,
                                        int wtile_e_cap, int64_t tiles_bound, int64_t wtiles_bound, int64_t max_tiles_per_graph_bound);

}  // namespace gnx
// builds the wide-path tables of a handle if they do not exist yet (the workspace queries call it; launchers call it with their stream:
// inside a capture a missing table is an error, not a build).  A failure is not latched.
extern "C" int32_t gnx_ensure_wide_tables(const gnx_graphs* h, void* stream = nullptr);
// int64 host copies of colptr / rowval of a handle that was built on the device (no-op otherwise)
extern "C" int32_t gnx_ensure_host_csc(const gnx_graphs* h);
namespace gnx {


}  // namespace gnx

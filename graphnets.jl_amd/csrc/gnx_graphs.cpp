// Graph-batch handles: the MI355X-side replacement of GNGraphBatch (reference src/gngraphbatch.jl:33-54).
// The reference precomputes seven dense one-hot "broadcaster" tensors over a padded PN^2 edge grid; here the same
// index semantics are held as CSC (colptr/rowval), per-graph offsets and a tile table, all int32 in HBM.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "gnx_internal.h"

namespace gnx {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int32_t fail(int32_t code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
int32_t hip_fail(hipError_t e, const char* what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  (void)hipGetLastError();
  return (int32_t)e > 0 ? (int32_t)e : 1;
}

// GNX_TIME_BUILD=1: phase timings of handle construction on stderr (diagnostic; like every switch of the library, read once per process)
static bool time_build_on() {
  static const bool on = getenv("GNX_TIME_BUILD") != nullptr;
  return on;
}
struct BuildTimer {
  bool on = time_build_on();
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void lap(const char* what) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[gnx build] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

static int env_int(const char* name, int dflt);
// the tile caps of a handle (edges / nodes per workgroup tile, edges per wave tile): environment overrides read ONCE per process
static void tile_caps(int* tile_e, int* tile_n, int* wtile_e) {
  static const int te = env_int("GNX_TILE_E", 512), tn = env_int("GNX_TILE_N", 128), we = env_int("GNX_WTILE_E", 128);
  *tile_e = te; *tile_n = tn; *wtile_e = (we == 64 || we == 128 || we == 256) ? we : 128;
}
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  if (!v || !*v) return dflt;
  int x = atoi(v);
  return x > 0 ? x : dflt;
}

template <typename T>
static inline int adj_value(const void* base, int64_t idx, bool* ok) {
  T v = reinterpret_cast<const T*>(base)[idx];
  if (v == (T)0) return 0;
  if (v == (T)1) return 1;
  *ok = false;
  return 0;
}

static int adj_at(const void* base, int32_t kind, int64_t idx, bool* ok) {
  switch (kind) {
    case GNX_ELEM_U8: return adj_value<uint8_t>(base, idx, ok);
    case GNX_ELEM_I32: return adj_value<int32_t>(base, idx, ok);
    case GNX_ELEM_I64: return adj_value<int64_t>(base, idx, ok);
    case GNX_ELEM_F32: return adj_value<float>(base, idx, ok);
    case GNX_ELEM_F64: return adj_value<double>(base, idx, ok);
    default: *ok = false; return 0;
  }
}

// graph-aligned packs for the in-kernel graph update (gnx_narrow.hip): every graph <= 8 wave tiles, more than one graph.  Best-fit
// decreasing over the graphs' wave-tile counts (h_wtile_off): a graph's tiles adjacent in ONE pack of 8 slots.
void build_packs(gnx_graphs* h, std::vector<int32_t>& packs) {
  packs.clear();
  h->n_packs = 0;
  if (!(h->G > 1 && h->max_wtiles_per_graph >= 1 && h->max_wtiles_per_graph <= 8)) return;
  constexpr int CAP = 8;
  std::vector<int32_t> order((size_t)h->G);
  for (int64_t g = 0; g < h->G; ++g) order[(size_t)g] = (int32_t)g;
  auto cnt_of = [&](int32_t g) { return h->h_wtile_off[(size_t)g + 1] - h->h_wtile_off[(size_t)g]; };
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cnt_of(a) > cnt_of(b); });
  std::vector<int32_t> fill;                  // slots used per pack
  std::vector<std::vector<int32_t>> open(CAP + 1);  // open[r]: packs with r free slots
  for (int32_t g : order) {
    const int c = cnt_of(g);
    if (c <= 0) continue;  // (a graph always has >= 1 node, hence >= 1 wave tile; kept for safety)
    int r = c;
    while (r <= CAP && open[(size_t)r].empty()) ++r;  // best fit: the fullest pack that still takes the graph
    int32_t pk;
    if (r > CAP) { pk = (int32_t)fill.size(); fill.push_back(0); packs.insert(packs.end(), CAP, -1); }
    else { pk = open[(size_t)r].back(); open[(size_t)r].pop_back(); }
    for (int k = 0; k < c; ++k) packs[(size_t)pk * CAP + fill[(size_t)pk] + k] = h->h_wtile_off[(size_t)g] + k;
    fill[(size_t)pk] += c;
    if (fill[(size_t)pk] < CAP) open[(size_t)(CAP - fill[(size_t)pk])].push_back(pk);
  }
  h->n_packs = (int32_t)fill.size();
}

// wide path: the COUNTS of its 128-row tiles and the per-graph tile offsets (O(G)); the tables themselves on first use (ensure_wide_tables)
static void fill_wide_counts(gnx_graphs* h) {
  const int64_t BM = 128;
  h->h_etile_off.assign(h->G + 1, 0);
  h->h_ntile_off.assign(h->G + 1, 0);
  for (int64_t g = 0; g < h->G; ++g) {
    h->h_etile_off[g + 1] = h->h_etile_off[g] + (int32_t)((h->h_edge_off[g + 1] - h->h_edge_off[g] + BM - 1) / BM);
    h->h_ntile_off[g + 1] = h->h_ntile_off[g] + (int32_t)((h->h_node_off[g + 1] - h->h_node_off[g] + BM - 1) / BM);
  }
  h->n_etiles = h->h_etile_off[h->G];
  h->n_ntiles = h->h_ntile_off[h->G];
  h->n_gtiles = (h->G + BM - 1) / BM;
  // rows of the per-destination partial-sum table: one per (chunk, destination in it) <= non-empty nodes + one extra row per chunk a
  // node's edge run can spill into; bounded without walking the graph by min(E, N + 2 * n_etiles)
  h->agg_rows_bound = std::min<int64_t>(h->E, h->N + 2 * h->n_etiles);
}

// Build tiles, upload, finish the handle.  h_colptr / h_rowval (global) / h_node_off / h_edge_off are filled.
static int32_t finalize(gnx_graphs* h) {
  if (h->N >= (int64_t)INT32_MAX || h->E >= (int64_t)INT32_MAX)
    return fail(GNX_ERR_TOO_LARGE, "graph batch exceeds int32 device indices");
  BuildTimer bt;
  GNX_HIP(hipGetDevice(&h->device));
  { int te, tn, we; tile_caps(&te, &tn, &we); h->tile_e_cap = te; h->tile_n_cap = tn; h->wtile_e_cap = we; }
  // greedy tiling inside each graph: add nodes while edges <= cap and nodes <= cap; a node whose in-degree
  // exceeds the cap becomes a single-node tile (kernels loop over its edges).
  auto build_tiles = [&](int e_cap, int n_cap, std::vector<gnx::Tile>& tiles, std::vector<int32_t>& off, int64_t* max_deg) {
    off.assign(h->G + 1, 0);
    for (int64_t g = 0; g < h->G; ++g) {
      off[g] = (int32_t)tiles.size();
      int64_t n = h->h_node_off[g];
      const int64_t nend = h->h_node_off[g + 1];
      while (n < nend) {
        int64_t n1 = n;
        const int64_t e0 = h->h_colptr[n];
        while (n1 < nend && (n1 - n) < n_cap) {
          const int64_t deg = h->h_colptr[n1 + 1] - h->h_colptr[n1];
          *max_deg = std::max(*max_deg, deg);
          if (n1 > n && h->h_colptr[n1 + 1] - e0 > e_cap) break;
          ++n1;
        }
        gnx::Tile t;
        t.n0 = (int32_t)n; t.n1 = (int32_t)n1;
        t.e0 = (int32_t)e0; t.e1 = (int32_t)h->h_colptr[n1];
        t.g = (int32_t)g;
        t.win0 = (int32_t)h->h_node_off[g]; t.win1 = (int32_t)nend;
        t.flags = 0;
        tiles.push_back(t);
        n = n1;
      }
    }
    off[h->G] = (int32_t)tiles.size();
  };
  h->max_in_degree = 0;
  // (the two tile kinds on two host threads were measured: 2.5 instead of 0.9 ms for 4096 graphs — the second thread's start and the
  // allocator cost more than the 0.45 ms it takes over)
  build_tiles(h->tile_e_cap, h->tile_n_cap, h->h_tiles, h->h_tile_off, &h->max_in_degree);
  build_tiles(h->wtile_e_cap, 64, h->h_wtiles, h->h_wtile_off, &h->max_in_degree);
  h->n_tiles_ = (int64_t)h->h_tiles.size(); h->n_wtiles_ = (int64_t)h->h_wtiles.size();
  for (int64_t g = 0; g < h->G; ++g) {  // wave tiles per graph (graph-update launch geometry)
    const int32_t cnt = h->h_wtile_off[g + 1] - h->h_wtile_off[g];
    h->max_wtiles_per_graph = std::max(h->max_wtiles_per_graph, cnt);
    for (int32_t t = h->h_wtile_off[g]; t < h->h_wtile_off[g + 1]; ++t) h->h_wtiles[(size_t)t].flags = cnt;
  }

  bt.lap("tile tables (host)");
  // the eager device arrays live in ONE allocation (256-B aligned slices), filled by one copy each
  struct Slice { void** dst; const void* src; size_t bytes; size_t off; };
  std::vector<int32_t> node_off32(h->h_node_off.begin(), h->h_node_off.end()), edge_off32(h->h_edge_off.begin(), h->h_edge_off.end());
  std::vector<int32_t> colptr32_tmp, rowval32_tmp;
  const int32_t* colptr32 = nullptr;
  const int32_t* rowval32 = nullptr;
  // (a constructor that already produced the device-format arrays — the CSC one, in its validation pass — hands them over)
  if (h->t_colptr32.size() == h->h_colptr.size()) colptr32 = h->t_colptr32.data();
  else { colptr32_tmp.assign(h->h_colptr.begin(), h->h_colptr.end()); colptr32 = colptr32_tmp.data(); }
  if (h->t_rowval32.size() == h->h_rowval.size() && !h->h_rowval.empty()) rowval32 = h->t_rowval32.data();
  else { rowval32_tmp.assign(h->h_rowval.begin(), h->h_rowval.end()); rowval32 = rowval32_tmp.data(); }
  std::vector<int32_t> packs;  // [n_packs][8] (filled below when applicable)
  fill_wide_counts(h);
  build_packs(h, packs);
  Slice sl[] = {
      {(void**)&h->d_colptr, colptr32, h->h_colptr.size() * sizeof(int32_t), 0},
      {(void**)&h->d_rowval, rowval32, h->h_rowval.size() * sizeof(int32_t), 0},
      {(void**)&h->d_node_off, node_off32.data(), node_off32.size() * sizeof(int32_t), 0},
      {(void**)&h->d_edge_off, edge_off32.data(), edge_off32.size() * sizeof(int32_t), 0},
      {(void**)&h->d_tile_off, h->h_tile_off.data(), h->h_tile_off.size() * sizeof(int32_t), 0},
      {(void**)&h->d_tiles, h->h_tiles.data(), h->h_tiles.size() * sizeof(gnx::Tile), 0},
      {(void**)&h->d_wtile_off, h->h_wtile_off.data(), h->h_wtile_off.size() * sizeof(int32_t), 0},
      {(void**)&h->d_wtiles, h->h_wtiles.data(), h->h_wtiles.size() * sizeof(gnx::Tile), 0},
      {(void**)&h->d_packs, packs.data(), packs.size() * sizeof(int32_t), 0},
  };
  size_t total = 0;
  for (Slice& x : sl) { x.off = total; total += (std::max<size_t>(x.bytes, 16) + 255) / 256 * 256; }
  {
    size_t got = 0;
    h->d_arena = arena_take(h->device, total, &got);  // a released handle's block of about this size (the device is synchronised first), else a fresh one
    if (h->d_arena) h->arena_bytes = got;
    else { GNX_HIP(hipMalloc(&h->d_arena, total)); h->arena_bytes = total; }
  }
  for (Slice& x : sl) {
    *x.dst = static_cast<char*>(h->d_arena) + x.off;  // (never NULL, as before: an empty array still has its 256-B slice)
    if (x.bytes) GNX_HIP(hipMemcpy(*x.dst, x.src, x.bytes, hipMemcpyHostToDevice));
  }
  gnx::vec_i32().swap(h->t_colptr32);
  gnx::vec_i32().swap(h->t_rowval32);
  bt.lap("device arrays (one allocation, nine copies)");
  return GNX_OK;
}


// The wide (matrix-core) path's tables, built on first use: 128-row edge / node / graph tiles, the destination of every edge, and the
// aggregation chunks of the edge GEMM's fused edge -> node sums (see gnx_internal.h).  Host loops over N + E and a dozen uploads.
static int32_t build_wide_tables(const gnx_graphs* h) {
  BuildTimer bt;
  int prev = -1;
  (void)hipGetDevice(&prev);
  struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev != h->device ? prev : -1};
  if (prev != h->device) GNX_HIP(hipSetDevice(h->device));
  int32_t rc;
  // wide path tables: 128-row chunks that never cross a graph boundary (so a tile has ONE per-graph bias)
  {
    const int BM = 128;
    for (int64_t g = 0; g < h->G; ++g) {
      if (h->h_etile_off[g] != (int32_t)h->h_etiles.size() || h->h_ntile_off[g] != (int32_t)h->h_ntiles.size())
        return fail(GNX_ERR_INVALID_ARG, "wide tables: tile offsets disagree with the handle's counts");
      for (int64_t e = h->h_edge_off[g]; e < h->h_edge_off[g + 1]; e += BM) {
        gnx::Tile t{};
        t.e0 = (int32_t)e; t.e1 = (int32_t)std::min<int64_t>(e + BM, h->h_edge_off[g + 1]); t.g = (int32_t)g;
        t.n0 = (int32_t)h->h_node_off[g]; t.n1 = (int32_t)h->h_node_off[g + 1];
        h->h_etiles.push_back(t);
      }
      for (int64_t n = h->h_node_off[g]; n < h->h_node_off[g + 1]; n += BM) {
        gnx::Tile t{};
        t.n0 = (int32_t)n; t.n1 = (int32_t)std::min<int64_t>(n + BM, h->h_node_off[g + 1]); t.g = (int32_t)g;
        t.e0 = (int32_t)h->h_colptr[t.n0]; t.e1 = (int32_t)h->h_colptr[t.n1];
        h->h_ntiles.push_back(t);
      }
    }
    if ((int64_t)h->h_etiles.size() != h->n_etiles || (int64_t)h->h_ntiles.size() != h->n_ntiles) return fail(GNX_ERR_INVALID_ARG, "wide tables: tile counts disagree");
    auto up = [&](const void* src, size_t bytes, void** dst) -> int32_t {
      GNX_HIP(hipMalloc(dst, std::max<size_t>(bytes, 16)));
      if (bytes) GNX_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
      return GNX_OK;
    };
    if ((rc = up(h->h_etiles.data(), h->h_etiles.size() * sizeof(gnx::Tile), (void**)&h->d_etiles))) return rc;
    if ((rc = up(h->h_ntiles.data(), h->h_ntiles.size() * sizeof(gnx::Tile), (void**)&h->d_ntiles))) return rc;
    for (int64_t g0 = 0; g0 < h->G; g0 += BM) {
      gnx::Tile t{};
      t.n0 = (int32_t)g0; t.n1 = (int32_t)std::min<int64_t>(g0 + BM, h->G); t.g = (int32_t)g0;
      h->h_gtiles.push_back(t);
    }
    if ((rc = up(h->h_gtiles.data(), h->h_gtiles.size() * sizeof(gnx::Tile), (void**)&h->d_gtiles))) return rc;
    if ((rc = up(h->h_etile_off.data(), h->h_etile_off.size() * sizeof(int32_t), (void**)&h->d_etile_off))) return rc;
    if ((rc = up(h->h_ntile_off.data(), h->h_ntile_off.size() * sizeof(int32_t), (void**)&h->d_ntile_off))) return rc;
    std::vector<int32_t> dst((size_t)h->E);
    for (int64_t n = 0; n < h->N; ++n)
      for (int64_t e = h->h_colptr[n]; e < h->h_colptr[n + 1]; ++e) dst[(size_t)e] = (int32_t)n;
    if ((rc = up(dst.data(), dst.size() * sizeof(int32_t), (void**)&h->d_edge_dst))) return rc;
    bt.lap("wide tiles + edge_dst");
    {
      // aggregation chunks (64-row passes of the 128-edge tiles): rows of the partial-sum table per chunk = distinct
      // destinations in the chunk; edges are dst-sorted, so that is a count of non-empty nodes between its first and last dst
      std::vector<int32_t> nonempty((size_t)h->N + 1, 0);  // nonempty[n] = nodes m < n with in-degree > 0
      for (int64_t n = 0; n < h->N; ++n) nonempty[(size_t)n + 1] = nonempty[(size_t)n] + (h->h_colptr[n + 1] > h->h_colptr[n] ? 1 : 0);
      const size_t n_chunks = 2 * h->h_etiles.size();
      std::vector<int32_t> row0(n_chunks + 1, 0);
      std::vector<int32_t> chunk_of_edge_first((size_t)h->N, -1);
      for (size_t t = 0; t < h->h_etiles.size(); ++t) {
        for (int pass = 0; pass < 2; ++pass) {
          const int64_t c0 = h->h_etiles[t].e0 + 64 * pass, c1 = std::min<int64_t>(c0 + 64, h->h_etiles[t].e1);
          const size_t c = 2 * t + pass;
          row0[c + 1] = row0[c] + (c1 > c0 ? nonempty[(size_t)dst[(size_t)c1 - 1] + 1] - nonempty[(size_t)dst[(size_t)c0]] : 0);
        }
      }
      h->n_agg_rows = row0[n_chunks];
      int64_t wide_span = 0;
      for (const gnx::Tile& t : h->h_etiles)
        if (t.e1 > t.e0 && dst[(size_t)t.e1 - 1] - dst[(size_t)t.e0] + 1 > gnx::kPdRowsCap) ++wide_span;
      h->n_etiles_wide_span = wide_span;
      if (h->n_agg_rows > h->agg_rows_bound) return fail(GNX_ERR_INVALID_ARG, "wide tables: more aggregation rows than the bound workspaces are sized with");
      std::vector<int32_t> agg_row((size_t)h->N, -1), parts((size_t)h->N, 0), first_chunk((size_t)h->N, 0);
      // chunk of an edge: tiles are 128-edge chunks of each graph's edge range, in graph order
      auto chunk_of = [&](int64_t e, int64_t g) { return (int64_t)2 * h->h_etile_off[(size_t)g] + (e - h->h_edge_off[(size_t)g]) / 64; };
      for (int64_t g = 0; g < h->G; ++g)
        for (int64_t n = h->h_node_off[(size_t)g]; n < h->h_node_off[(size_t)g + 1]; ++n) {
          const int64_t e0 = h->h_colptr[n], e1 = h->h_colptr[n + 1];
          if (e1 <= e0) continue;
          const int64_t ca = chunk_of(e0, g), cb = chunk_of(e1 - 1, g);
          const int64_t chunk_e0 = h->h_edge_off[(size_t)g] + ((e0 - h->h_edge_off[(size_t)g]) / 64) * 64;  // first edge of chunk ca
          agg_row[(size_t)n] = row0[(size_t)ca] + (nonempty[(size_t)n] - nonempty[(size_t)dst[(size_t)chunk_e0]]);
          parts[(size_t)n] = (int32_t)(cb - ca + 1);
          first_chunk[(size_t)n] = (int32_t)ca;
        }
      if ((rc = up(row0.data(), row0.size() * sizeof(int32_t), (void**)&h->d_chunk_row0))) return rc;
      if ((rc = up(agg_row.data(), agg_row.size() * sizeof(int32_t), (void**)&h->d_node_agg_row))) return rc;
      if ((rc = up(parts.data(), parts.size() * sizeof(int32_t), (void**)&h->d_node_agg_parts))) return rc;
      if ((rc = up(first_chunk.data(), first_chunk.size() * sizeof(int32_t), (void**)&h->d_node_agg_chunk))) return rc;
      bt.lap("aggregation tables");
    }
  }
  return GNX_OK;
}

}  // namespace gnx

namespace gnx {
int32_t build_csc_on_device(const void* const* adj, const void* packed, int packed_on_device, const int64_t* n_nodes, int64_t G, int32_t elem_kind, int32_t row_major,
                            gnx::vec_i64& h_colptr, gnx::vec_i64& h_rowval, const std::vector<int64_t>& h_node_off, DenseCscOnDevice* keep);
}

using namespace gnx;

extern "C" int32_t gnx_graphs_destroy(gnx_graphs* h);

// Shared body of the two CSC constructors: graph g's arrays through two accessors.  Two passes: sizes (so that every array is allocated
// once), then validation fused with the copy — int64 host copies (accessors, collapse / CSR builders) and the int32 device-format arrays
// in one sweep over the input.
template <class CP, class RV>
static int32_t create_csc_impl(CP colptr_of, RV rowval_of, const int64_t* n_nodes, int64_t n_graphs, int32_t index_base, gnx_graphs** out) {
  BuildTimer bt;
  gnx_graphs* h = new gnx_graphs();
  h->G = n_graphs;
  h->h_node_off.resize((size_t)n_graphs + 1);
  h->h_edge_off.resize((size_t)n_graphs + 1);
  h->h_node_off[0] = 0; h->h_edge_off[0] = 0;
  for (int64_t g = 0; g < n_graphs; ++g) {
    const int64_t n = n_nodes[g];
    const int64_t* cp = colptr_of(g);
    if (n <= 0 || !cp) { delete h; return fail(GNX_ERR_ADJ_SHAPE, "graph must have N >= 1 nodes and a colptr"); }
    if (cp[0] != index_base) { delete h; return fail(GNX_ERR_CSC, "colptr[0] must equal index_base"); }
    const int64_t eg = cp[n] - index_base;
    if (eg < 0 || eg > n * n) { delete h; return fail(GNX_ERR_CSC, "colptr must be non-decreasing with at most N entries per column"); }
    if (eg > 0 && !rowval_of(g)) { delete h; return fail(GNX_ERR_CSC, "rowval is NULL but the graph has edges"); }
    h->PN = std::max(h->PN, n);
    h->h_node_off[(size_t)g + 1] = h->h_node_off[(size_t)g] + n;
    h->h_edge_off[(size_t)g + 1] = h->h_edge_off[(size_t)g] + eg;
  }
  h->N = h->h_node_off.back();
  h->E = h->h_edge_off.back();
  if (h->N >= (int64_t)INT32_MAX || h->E >= (int64_t)INT32_MAX) { delete h; return fail(GNX_ERR_TOO_LARGE, "graph batch exceeds int32 device indices"); }
  h->h_colptr.resize((size_t)h->N + 1);
  h->h_rowval.resize((size_t)h->E);
  h->t_colptr32.resize((size_t)h->N + 1);
  h->t_rowval32.resize((size_t)h->E);
  h->h_colptr[0] = 0; h->t_colptr32[0] = 0;
  int64_t* hc = h->h_colptr.data();
  int64_t* hr = h->h_rowval.data();
  int32_t* c32 = h->t_colptr32.data();
  int32_t* r32 = h->t_rowval32.data();
  // graphs are independent: large batches are validated and copied by a few host threads, each on a contiguous range of graphs with
  // about the same number of nodes + edges (4096 small graphs, 590k nodes + 1M edges: 4.3-5.5 -> 2.0 ms)
  auto run_range = [&](int64_t g_lo, int64_t g_hi) -> const char* {
  const char* bad = nullptr;
  for (int64_t g = g_lo; g < g_hi && !bad; ++g) {
    const int64_t n = n_nodes[g], base = h->h_node_off[(size_t)g], ebase = h->h_edge_off[(size_t)g];
    const int64_t* cp = colptr_of(g);
    const int64_t* rv = rowval_of(g);
    const int64_t eg = h->h_edge_off[(size_t)g + 1] - ebase;  // = cp[n] - index_base: what the arrays were sized with
    int64_t a = 0;  // cp[j] - index_base
    for (int64_t j = 0; j < n; ++j) {
      const int64_t b = cp[j + 1] - index_base;
      // b > eg: a colptr that rises above cp[n] and comes back would be written (and, packed form, read) past the graph's slice
      if (b < a || b - a > n || b > eg) { bad = "colptr must be non-decreasing with at most N entries per column"; break; }
      int64_t prev = -1;
      for (int64_t k = a; k < b; ++k) {
        const int64_t i = rv[k] - index_base;
        if (i <= prev || i >= n) { bad = "rowval out of range or not strictly increasing inside a column"; break; }
        prev = i;
        hr[ebase + k] = base + i;
        r32[ebase + k] = (int32_t)(base + i);
      }
      if (bad) break;
      hc[base + j + 1] = ebase + b;
      c32[base + j + 1] = (int32_t)(ebase + b);
      a = b;
    }
    if (!bad && a != eg) bad = "colptr must be non-decreasing with at most N entries per column";
  }
  return bad;
  };
  const char* bad = nullptr;
  const int64_t work = h->N + h->E;
  const int n_thr = (int)std::min<int64_t>(std::min<int64_t>(8, std::max(1u, std::thread::hardware_concurrency())), std::min<int64_t>(n_graphs, work / 400000));
  if (n_thr <= 1) {
    bad = run_range(0, n_graphs);
  } else {
    std::vector<int64_t> cut((size_t)n_thr + 1, n_graphs);
    cut[0] = 0;
    int t = 1;
    for (int64_t g = 0; g < n_graphs && t < n_thr; ++g)
      if (h->h_node_off[(size_t)g + 1] + h->h_edge_off[(size_t)g + 1] >= work * t / n_thr) cut[(size_t)t++] = g + 1;
    std::vector<const char*> res((size_t)n_thr, nullptr);
    std::vector<std::thread> thr;
    for (int i = 1; i < n_thr; ++i) thr.emplace_back([&, i] { res[(size_t)i] = run_range(cut[(size_t)i], cut[(size_t)i + 1]); });
    res[0] = run_range(cut[0], cut[1]);
    for (auto& th : thr) th.join();
    for (int i = 0; i < n_thr && !bad; ++i) bad = res[(size_t)i];  // the first failing range in graph order: the message a serial pass gives
  }
  if (bad) { delete h; return fail(GNX_ERR_CSC, bad); }
  bt.lap("csc validation + copy");
  int32_t rc = finalize(h);
  if (rc) { gnx_graphs_destroy(h); return rc; }
  *out = h;
  return GNX_OK;
}

// The device builder's front end: the O(G) part of create_csc_impl's first pass (sizes, per-graph checks that need no sweep) on the host,
// everything O(N + E) in kernels.  Same error codes and messages as the host builder; 1 = not applicable (the host builder runs).
template <class CPAT>
static int32_t create_csc_device(const void* colptr_cat, const void* rowval_cat, const int64_t* n_nodes, int64_t n_graphs, int32_t index_base, int32_t index_bits,
                                 CPAT cp_at, gnx_graphs** out) {
  BuildTimer bt;
  std::unique_ptr<gnx_graphs> h(new gnx_graphs());
  h->G = n_graphs;
  h->h_node_off.resize((size_t)n_graphs + 1);
  h->h_edge_off.resize((size_t)n_graphs + 1);
  h->h_node_off[0] = 0; h->h_edge_off[0] = 0;
  { int te, tn, we; tile_caps(&te, &tn, &we); h->tile_e_cap = te; h->tile_n_cap = tn; h->wtile_e_cap = we; }
  int64_t cpo = 0, tb = 0, wb = 0, per_graph = 1;
  for (int64_t g = 0; g < n_graphs; ++g) {
    const int64_t n = n_nodes[g];
    if (cp_at(cpo) != index_base) return fail(GNX_ERR_CSC, "colptr[0] must equal index_base");
    const int64_t eg = cp_at(cpo + n) - index_base;
    if (eg < 0 || eg > n * n) return fail(GNX_ERR_CSC, "colptr must be non-decreasing with at most N entries per column");
    cpo += n + 1;
    h->PN = std::max(h->PN, n);
    h->h_node_off[(size_t)g + 1] = h->h_node_off[(size_t)g] + n;
    h->h_edge_off[(size_t)g + 1] = h->h_edge_off[(size_t)g] + eg;
    // a graph's greedy tiling: at most N_g / n_cap tiles end on the node cap, fewer than 2 E_g / e_cap on the edge cap (a tile that ends on
    // it holds, with its successor, more than e_cap edges), one on the graph's end
    const int64_t bt_g = std::min(n, n / h->tile_n_cap + 2 * eg / h->tile_e_cap + 2), bw_g = std::min(n, n / 64 + 2 * eg / h->wtile_e_cap + 2);
    tb += bt_g; wb += bw_g;
    per_graph = std::max(per_graph, std::max(bt_g, bw_g));
  }
  h->N = h->h_node_off.back();
  h->E = h->h_edge_off.back();
  if (h->N >= (int64_t)INT32_MAX || h->E >= (int64_t)INT32_MAX) return fail(GNX_ERR_TOO_LARGE, "graph batch exceeds int32 device indices");
  GNX_HIP(hipGetDevice(&h->device));
  fill_wide_counts(h.get());
  bt.lap("csc: sizes (host, O(G))");
  const int32_t rc = build_handle_from_csc_on_device(h.get(), colptr_cat, rowval_cat, index_base, index_bits, h->tile_e_cap, h->tile_n_cap, h->wtile_e_cap, tb, wb, per_graph);
  bt.lap("csc: validation + tables (device)");
  if (rc) { gnx_graphs_destroy(h.release()); return rc; }
  *out = h.release();
  return GNX_OK;
}

extern "C" {

int32_t gnx_version(void) { return GNX_VERSION; }
const char* gnx_last_error(void) { return g_last_error.c_str(); }

static int32_t create_dense_impl(const void* const* adj, const void* packed, int packed_on_device, const int64_t* n_nodes, int64_t n_graphs, int32_t elem_kind,
                                 int32_t row_major, gnx_graphs** out);

int32_t gnx_graphs_create_dense(const void* const* adj, const int64_t* n_nodes, int64_t n_graphs, int32_t elem_kind,
                                int32_t row_major, gnx_graphs** out) {
  if (!out) return fail(GNX_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  if (n_graphs <= 0) return fail(GNX_ERR_NO_GRAPHS, "length(adj_mats) must be > 0 (checks.jl:8)");
  if (!adj || !n_nodes) return fail(GNX_ERR_INVALID_ARG, "adj / n_nodes is NULL");
  return create_dense_impl(adj, nullptr, 0, n_nodes, n_graphs, elem_kind, row_major, out);
}

// the same batch from ONE buffer holding the matrices one after the other (graph g: n_g x n_g elements of elem_kind): `adj_bytes` must equal
// sum(n_g^2) * sizeof(element) — nothing is read past it.  on_device = 0: host memory (pinned memory travels as one DMA); 1: device memory of
// the current device (no copy: the scan reads it where it is).  What a data loader holds — and what the bindings call: one pointer, no per-graph work.
int32_t gnx_graphs_create_dense_packed(const void* adj_cat, int64_t adj_bytes, const int64_t* n_nodes, int64_t n_graphs, int32_t elem_kind, int32_t row_major,
                                       int32_t on_device, gnx_graphs** out) {
  if (!out) return fail(GNX_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  if (n_graphs <= 0) return fail(GNX_ERR_NO_GRAPHS, "length(adj_mats) must be > 0 (checks.jl:8)");
  if (!adj_cat || !n_nodes) return fail(GNX_ERR_INVALID_ARG, "adj_cat / n_nodes is NULL");
  if (elem_kind < GNX_ELEM_U8 || elem_kind > GNX_ELEM_F64) return fail(GNX_ERR_INVALID_ARG, "bad elem_kind");
  static const int64_t esz_tab[5] = {1, 4, 8, 4, 8};
  int64_t total = 0;
  for (int64_t g = 0; g < n_graphs; ++g) {
    if (n_nodes[g] <= 0) return fail(GNX_ERR_ADJ_SHAPE, "adjacency matrix must be N x N with N >= 1 (checks.jl:11)");
    // (n^2 and the running total are checked products / sums — n = 2^32 wrapped to 0 and passed the size check below)
    int64_t sq = 0;
    if (n_nodes[g] > (int64_t)1 << 31 || __builtin_mul_overflow(n_nodes[g], n_nodes[g], &sq) || __builtin_add_overflow(total, sq, &total))
      return fail(GNX_ERR_TOO_LARGE, "adjacency matrices too large: sum(n_nodes^2) exceeds the 63-bit element count");
  }
  int64_t want_bytes = 0;
  if (__builtin_mul_overflow(total, esz_tab[elem_kind], &want_bytes) || adj_bytes != want_bytes) return fail(GNX_ERR_INVALID_ARG, "adj_bytes must equal sum(n_nodes^2) elements");
  if (on_device) return create_dense_impl(nullptr, adj_cat, 1, n_nodes, n_graphs, elem_kind, row_major, out);
  std::vector<const void*> ptrs((size_t)n_graphs);  // (the host scan of small batches walks per-graph pointers)
  int64_t at = 0;
  for (int64_t g = 0; g < n_graphs; ++g) { ptrs[(size_t)g] = static_cast<const char*>(adj_cat) + at * esz_tab[elem_kind]; at += n_nodes[g] * n_nodes[g]; }
  return create_dense_impl(ptrs.data(), adj_cat, 0, n_nodes, n_graphs, elem_kind, row_major, out);
}

static int32_t create_dense_impl(const void* const* adj, const void* packed, int packed_on_device, const int64_t* n_nodes, int64_t n_graphs, int32_t elem_kind,
                                 int32_t row_major, gnx_graphs** out) {
  if (elem_kind < GNX_ELEM_U8 || elem_kind > GNX_ELEM_F64) return fail(GNX_ERR_INVALID_ARG, "bad elem_kind");
  BuildTimer bt;
  gnx_graphs* h = new gnx_graphs();
  h->G = n_graphs;
  h->h_node_off.push_back(0);
  int64_t total_entries = 0;
  for (int64_t g = 0; g < n_graphs; ++g) {
    const int64_t n = n_nodes[g];
    if (n <= 0 || (adj && !adj[g])) {
      delete h;
      return fail(GNX_ERR_ADJ_SHAPE, "adjacency matrix must be N x N with N >= 1 (checks.jl:11)");
    }
    h->PN = std::max(h->PN, n);
    h->h_node_off.push_back(h->h_node_off.back() + n);
    total_entries += n * n;
  }
  // large dense batches: scan + compaction on the GPU (gnx_build_device.hip); small ones on the host (no launch cost)
  static const int64_t dev_threshold = getenv("GNX_BUILD_DEVICE_MIN") ? atoll(getenv("GNX_BUILD_DEVICE_MIN")) : (1 << 22);
  bool built = false;
  if (total_entries >= dev_threshold || packed_on_device) {
    // scan + compaction on the GPU; the CSC then STAYS there and the device builder makes the handle's tables from it (no round trip of
    // 4 (N + E) bytes through the host, no host tiling pass).  GNX_BUILD_CSC_DEVICE=0: the CSC comes back and finalize() runs (the validator)
    static const bool dev_tables = !(getenv("GNX_BUILD_CSC_DEVICE") && atoi(getenv("GNX_BUILD_CSC_DEVICE")) == 0);
    DenseCscOnDevice keep;
    const int32_t rc = build_csc_on_device(adj, packed, packed_on_device, n_nodes, n_graphs, elem_kind, row_major, h->h_colptr, h->h_rowval, h->h_node_off,
                                           dev_tables ? &keep : nullptr);
    if (rc != 1 && rc != GNX_OK) { delete h; return rc; }
    if (rc == 1 && packed_on_device) { delete h; return fail(GNX_ERR_TOO_LARGE, "a dense batch given in device memory exceeds the device builder's capacity (4M nodes)"); }
    built = rc == GNX_OK;
    if (built && dev_tables) {
      struct Free { DenseCscOnDevice& k; ~Free() { release_dense_csc(k); } } free_keep{keep};
      { int te, tn, we; tile_caps(&te, &tn, &we); h->tile_e_cap = te; h->tile_n_cap = tn; h->wtile_e_cap = we; }
      h->h_edge_off.assign(keep.edge_off.begin(), keep.edge_off.end());
      h->N = h->h_node_off.back();
      h->E = keep.E;
      int64_t tb = 0, wb = 0, per_graph = 1;
      for (int64_t g = 0; g < n_graphs; ++g) {
        const int64_t n = n_nodes[g], eg = h->h_edge_off[(size_t)g + 1] - h->h_edge_off[(size_t)g];
        const int64_t bt_g = std::min(n, n / h->tile_n_cap + 2 * eg / h->tile_e_cap + 2), bw_g = std::min(n, n / 64 + 2 * eg / h->wtile_e_cap + 2);
        tb += bt_g; wb += bw_g;
        per_graph = std::max(per_graph, std::max(bt_g, bw_g));
      }
      if (h->N >= (int64_t)INT32_MAX || h->E >= (int64_t)INT32_MAX) { delete h; return fail(GNX_ERR_TOO_LARGE, "graph batch exceeds int32 device indices"); }
      hipError_t de = hipGetDevice(&h->device);
      if (de != hipSuccess) { delete h; return hip_fail(de, "hipGetDevice"); }
      fill_wide_counts(h);
      bt.lap("dense -> csc (device, kept there)");
      const int32_t rc2 = build_handle_from_csc_on_device(h, keep.d_colptr, keep.d_rowval, 0, 0 /* device-resident global int32 CSC */, h->tile_e_cap, h->tile_n_cap,
                                                          h->wtile_e_cap, tb, wb, per_graph);
      bt.lap("tables (device)");
      if (rc2 == GNX_OK) { *out = h; return GNX_OK; }
      if (rc2 != 1) { gnx_graphs_destroy(h); return rc2; }
      // not applicable there (scan capacity): bring the CSC to the host after all
      if (h->d_arena) { arena_give(h->device, h->d_arena, h->arena_bytes); h->d_arena = nullptr; }
      gnx::vec_i32 c32((size_t)h->N + 1), r32((size_t)h->E);
      hipError_t e1 = hipMemcpy(c32.data(), keep.d_colptr, c32.size() * sizeof(int32_t), hipMemcpyDeviceToHost);
      if (e1 == hipSuccess && h->E) e1 = hipMemcpy(r32.data(), keep.d_rowval, r32.size() * sizeof(int32_t), hipMemcpyDeviceToHost);
      if (e1 != hipSuccess) { delete h; return hip_fail(e1, "dense batch: CSC to the host"); }
      h->h_colptr.assign(c32.begin(), c32.end());
      h->h_rowval.assign(r32.begin(), r32.end());
      h->h_edge_off.clear();
    }
  }
  if (built) {
    for (int64_t g = 0; g <= n_graphs; ++g) h->h_edge_off.push_back(h->h_colptr[(size_t)h->h_node_off[g]]);
  } else {
    h->h_edge_off.push_back(0);
    h->h_colptr.assign(1, 0);
    h->h_rowval.clear();
    for (int64_t g = 0; g < n_graphs; ++g) {
      const int64_t n = n_nodes[g];
      const int64_t base = h->h_node_off[g];
      bool ok = true;
      for (int64_t j = 0; j < n; ++j) {      // destination (column) — slowest
        for (int64_t i = 0; i < n; ++i) {    // source (row)
          const int64_t idx = row_major ? i * n + j : j * n + i;
          if (adj_at(adj[g], elem_kind, idx, &ok)) h->h_rowval.push_back(base + i);
        }
        h->h_colptr.push_back((int64_t)h->h_rowval.size());
      }
      if (!ok) {
        delete h;
        return fail(GNX_ERR_ADJ_VALUE, "adjacency entries must be exactly 0 or 1 (pad.jl:30, gngraphbatch.jl:207)");
      }
      h->h_edge_off.push_back((int64_t)h->h_rowval.size());
    }
  }
  h->N = h->h_node_off.back();
  h->E = h->h_edge_off.back();
  bt.lap("dense -> csc");
  int32_t rc = finalize(h);
  if (rc) { gnx_graphs_destroy(h); return rc; }
  *out = h;
  return GNX_OK;
}

int32_t gnx_graphs_create_csc(const int64_t* const* colptr, const int64_t* const* rowval, const int64_t* n_nodes,
                              int64_t n_graphs, int32_t index_base, gnx_graphs** out) {
  if (!out) return fail(GNX_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  if (n_graphs <= 0) return fail(GNX_ERR_NO_GRAPHS, "length(adj_mats) must be > 0 (checks.jl:8)");
  if (!colptr || !rowval || !n_nodes) return fail(GNX_ERR_INVALID_ARG, "colptr / rowval / n_nodes is NULL");
  if (index_base != 0 && index_base != 1) return fail(GNX_ERR_INVALID_ARG, "index_base must be 0 or 1");
  return create_csc_impl([&](int64_t g) { return colptr[g]; }, [&](int64_t g) { return rowval[g]; }, n_nodes, n_graphs, index_base, out);
}

int32_t gnx_graphs_create_csc_packed(const int64_t* colptr_cat, const int64_t* rowval_cat, const int64_t* n_nodes, int64_t n_graphs,
                                     int32_t index_base, gnx_graphs** out) {
  if (!out) return fail(GNX_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  if (n_graphs <= 0) return fail(GNX_ERR_NO_GRAPHS, "length(adj_mats) must be > 0 (checks.jl:8)");
  if (!colptr_cat || !n_nodes) return fail(GNX_ERR_INVALID_ARG, "colptr / n_nodes is NULL");
  if (index_base != 0 && index_base != 1) return fail(GNX_ERR_INVALID_ARG, "index_base must be 0 or 1");
  // graph g's colptr starts behind the (n + 1)-entry colptrs of the graphs before it, its rowval behind their edges
  std::vector<int64_t> cpo((size_t)n_graphs + 1, 0), rvo((size_t)n_graphs + 1, 0);
  for (int64_t g = 0; g < n_graphs; ++g) {
    if (n_nodes[g] <= 0) return fail(GNX_ERR_ADJ_SHAPE, "graph must have N >= 1 nodes and a colptr");
    cpo[(size_t)g + 1] = cpo[(size_t)g] + n_nodes[g] + 1;
    const int64_t eg = colptr_cat[cpo[(size_t)g + 1] - 1] - index_base;
    if (eg < 0) return fail(GNX_ERR_CSC, "colptr must be non-decreasing with at most N entries per column");
    rvo[(size_t)g + 1] = rvo[(size_t)g] + eg;
  }
  if (rvo.back() > 0 && !rowval_cat) return fail(GNX_ERR_CSC, "rowval is NULL but the graph has edges");
  return create_csc_impl([&](int64_t g) { return colptr_cat + cpo[(size_t)g]; },
                         [&](int64_t g) { return rowval_cat ? rowval_cat + rvo[(size_t)g] : nullptr; }, n_nodes, n_graphs, index_base, out);
}

// The length-checked, index-width-tagged form of the packed constructor: what a host binding should call (nothing is read past
// colptr_len / rowval_len entries, whatever the arrays contain).
int32_t gnx_graphs_create_csc_cat(const void* colptr_cat, int64_t colptr_len, const void* rowval_cat, int64_t rowval_len, const int64_t* n_nodes,
                                  int64_t n_graphs, int32_t index_base, int32_t index_bits, gnx_graphs** out) {
  if (!out) return fail(GNX_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  if (n_graphs <= 0) return fail(GNX_ERR_NO_GRAPHS, "length(adj_mats) must be > 0 (checks.jl:8)");
  if (!colptr_cat || !n_nodes) return fail(GNX_ERR_INVALID_ARG, "colptr / n_nodes is NULL");
  if (index_base != 0 && index_base != 1) return fail(GNX_ERR_INVALID_ARG, "index_base must be 0 or 1");
  if (index_bits != 32 && index_bits != 64) return fail(GNX_ERR_INVALID_ARG, "index_bits must be 32 or 64");
  if (colptr_len < 0 || rowval_len < 0) return fail(GNX_ERR_INVALID_ARG, "negative array length");
  auto cp_at = [&](int64_t i) -> int64_t { return index_bits == 64 ? static_cast<const int64_t*>(colptr_cat)[i] : (int64_t)static_cast<const int32_t*>(colptr_cat)[i]; };
  int64_t cpo = 0, rvo = 0;
  for (int64_t g = 0; g < n_graphs; ++g) {
    if (n_nodes[g] <= 0) return fail(GNX_ERR_ADJ_SHAPE, "graph must have N >= 1 nodes and a colptr");
    if (n_nodes[g] + 1 > colptr_len - cpo) return fail(GNX_ERR_INVALID_ARG, "colptr_cat is shorter than sum(n_nodes) + n_graphs entries");
    cpo += n_nodes[g] + 1;
    const int64_t eg = cp_at(cpo - 1) - index_base;
    if (eg < 0) return fail(GNX_ERR_CSC, "colptr must be non-decreasing with at most N entries per column");
    if (eg > rowval_len - rvo) return fail(GNX_ERR_INVALID_ARG, "rowval_cat is shorter than the edges the colptr arrays announce");
    rvo += eg;
  }
  if (cpo != colptr_len) return fail(GNX_ERR_INVALID_ARG, "colptr_cat must hold exactly sum(n_nodes) + n_graphs entries");
  if (rvo != rowval_len) return fail(GNX_ERR_INVALID_ARG, "rowval_cat must hold exactly the edges the colptr arrays announce");
  if (rvo > 0 && !rowval_cat) return fail(GNX_ERR_CSC, "rowval is NULL but the graph has edges");
  {
    // large batches: validation, device-format arrays and both tile tables in kernels (gnx_build_csc.hip); 1 = not applicable here
    static const bool dev_build = !(getenv("GNX_BUILD_CSC_DEVICE") && atoi(getenv("GNX_BUILD_CSC_DEVICE")) == 0);
    static const int64_t dev_min = getenv("GNX_BUILD_CSC_DEVICE_MIN") ? atoll(getenv("GNX_BUILD_CSC_DEVICE_MIN")) : 65536;
    if (dev_build && cpo + rvo >= dev_min) {
      const int32_t rc = create_csc_device(colptr_cat, rowval_cat, n_nodes, n_graphs, index_base, index_bits, cp_at, out);
      if (rc != 1) return rc;
    }
  }
  if (index_bits == 64) return gnx_graphs_create_csc_packed(static_cast<const int64_t*>(colptr_cat), static_cast<const int64_t*>(rowval_cat), n_nodes, n_graphs, index_base, out);
  // 32-bit indices on the host path: widened copies (the device-side builder reads them as they are)
  std::vector<int64_t> c64((size_t)colptr_len), r64((size_t)rowval_len);
  for (int64_t i = 0; i < colptr_len; ++i) c64[(size_t)i] = static_cast<const int32_t*>(colptr_cat)[i];
  for (int64_t i = 0; i < rowval_len; ++i) r64[(size_t)i] = static_cast<const int32_t*>(rowval_cat)[i];
  return gnx_graphs_create_csc_packed(c64.data(), r64.empty() ? nullptr : r64.data(), n_nodes, n_graphs, index_base, out);
}

int32_t gnx_graphs_destroy(gnx_graphs* h) {
  if (!h) return GNX_OK;
  arena_give(h->device, h->d_arena, h->arena_bytes);  // colptr, rowval, node / edge / tile offsets, tiles, wave tiles, packs: kept for the next handle of this size
  if (h->d_wide_arena) {
    (void)hipFree(h->d_wide_arena);
    h->d_edge_dst = nullptr; h->d_chunk_row0 = nullptr; h->d_node_agg_row = nullptr; h->d_node_agg_parts = nullptr; h->d_node_agg_chunk = nullptr;
    h->d_etiles = nullptr; h->d_ntiles = nullptr; h->d_gtiles = nullptr; h->d_etile_off = nullptr; h->d_ntile_off = nullptr;
  }
  (void)hipFree(h->d_edge_dst);
  (void)hipFree(h->d_chunk_row0);
  (void)hipFree(h->d_node_agg_row);
  (void)hipFree(h->d_node_agg_parts);
  (void)hipFree(h->d_node_agg_chunk);
  (void)hipFree(h->d_etiles);
  (void)hipFree(h->d_ntiles);
  (void)hipFree(h->d_gtiles);
  (void)hipFree(h->d_etile_off);
  (void)hipFree(h->d_ntile_off);
  (void)hipFree(h->d_collapse_edge);
  (void)hipFree(h->d_collapse_rev);
  (void)hipFree(h->d_csr_ptr);
  (void)hipFree(h->d_csr_eid);
  for (auto& ax : h->aux) {
    if (ax.fork) (void)hipEventDestroy(ax.fork);
    if (ax.join) (void)hipEventDestroy(ax.join);
    if (ax.stream) (void)hipStreamDestroy(ax.stream);
  }
  delete h;
  return GNX_OK;
}

// lower-triangle edge list + reverse-edge lookup (gngraphbatch.jl:56-81 as index tables instead of a PN^2 x PN(PN+1)/2 matrix)
static int32_t build_collapse(const gnx_graphs* h) {
  std::vector<int32_t> edge, rev;
  h->h_collapse_off.assign(h->G + 1, 0);
  for (int64_t g = 0; g < h->G; ++g) {
    for (int64_t j = h->h_node_off[g]; j < h->h_node_off[g + 1]; ++j) {   // destination (column), edge order
      for (int64_t e = h->h_colptr[j]; e < h->h_colptr[j + 1]; ++e) {
        const int64_t i = h->h_rowval[e];                                  // source (row), global id
        if (i < j) continue;                                               // keep the lower triangle i >= j
        int64_t r = -1;                                                    // reverse edge j -> i: row j in column i
        const int64_t* b = h->h_rowval.data() + h->h_colptr[i];
        const int64_t* en = h->h_rowval.data() + h->h_colptr[i + 1];
        const int64_t* it = std::lower_bound(b, en, j);
        if (it != en && *it == j) r = it - h->h_rowval.data();
        edge.push_back((int32_t)e);
        rev.push_back((int32_t)r);
      }
    }
    h->h_collapse_off[g + 1] = (int64_t)edge.size();
  }
  GNX_HIP(hipMalloc((void**)&h->d_collapse_edge, std::max<size_t>(edge.size(), 1) * sizeof(int32_t)));
  GNX_HIP(hipMalloc((void**)&h->d_collapse_rev, std::max<size_t>(edge.size(), 1) * sizeof(int32_t)));
  if (!edge.empty()) {
    GNX_HIP(hipMemcpy(h->d_collapse_edge, edge.data(), edge.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    GNX_HIP(hipMemcpy(h->d_collapse_rev, rev.data(), rev.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  return GNX_OK;
}

static int32_t build_csr(const gnx_graphs* h) {
  std::vector<int32_t> ptr((size_t)h->N + 1, 0), eid((size_t)h->E);
  for (int64_t e = 0; e < h->E; ++e) ptr[(size_t)h->h_rowval[e] + 1]++;
  for (int64_t n = 0; n < h->N; ++n) ptr[n + 1] += ptr[n];
  std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
  for (int64_t e = 0; e < h->E; ++e) eid[(size_t)fill[(size_t)h->h_rowval[e]]++] = (int32_t)e;  // ascending e inside a source: fixed order
  GNX_HIP(hipMalloc((void**)&h->d_csr_ptr, ptr.size() * sizeof(int32_t)));
  GNX_HIP(hipMalloc((void**)&h->d_csr_eid, std::max<size_t>(eid.size(), 1) * sizeof(int32_t)));
  GNX_HIP(hipMemcpy(h->d_csr_ptr, ptr.data(), ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  if (!eid.empty()) GNX_HIP(hipMemcpy(h->d_csr_eid, eid.data(), eid.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  return GNX_OK;
}

// frees whatever a failed build_wide_tables left behind, so that a later call starts from nothing
static void drop_wide_tables(const gnx_graphs* h) {
  if (h->d_wide_arena) {  // device-built: the ten arrays are slices of one allocation
    (void)hipFree(h->d_wide_arena);
    h->d_wide_arena = nullptr;
    h->d_edge_dst = nullptr; h->d_chunk_row0 = nullptr; h->d_node_agg_row = nullptr; h->d_node_agg_parts = nullptr; h->d_node_agg_chunk = nullptr;
    h->d_etiles = nullptr; h->d_ntiles = nullptr; h->d_gtiles = nullptr; h->d_etile_off = nullptr; h->d_ntile_off = nullptr;
  }
  void** dev[] = {(void**)&h->d_edge_dst, (void**)&h->d_chunk_row0, (void**)&h->d_node_agg_row, (void**)&h->d_node_agg_parts, (void**)&h->d_node_agg_chunk,
                  (void**)&h->d_etiles, (void**)&h->d_ntiles, (void**)&h->d_gtiles, (void**)&h->d_etile_off, (void**)&h->d_ntile_off};
  for (void** d : dev) { if (*d) (void)hipFree(*d); *d = nullptr; }
  h->h_etiles.clear(); h->h_ntiles.clear(); h->h_gtiles.clear();
  h->n_agg_rows = 0; h->n_etiles_wide_span = 0;
}

// Built by the workspace queries of every path that reads them (callers run those outside a capture); a launcher that still finds them
// missing builds them here — unless `stream` is being captured (hipMalloc + synchronous copies are not capturable): that is an error
// the caller can act on, and it is NOT remembered: the next call outside the capture builds the tables.  A failed build frees its
// partial allocations, keeps its own message in gnx_last_error() and may be retried.
int32_t gnx_ensure_wide_tables(const gnx_graphs* h, void* stream) {
  if (!h) return fail(GNX_ERR_INVALID_ARG, "NULL handle");
  if (h->wide_built.load(std::memory_order_acquire)) return GNX_OK;
  if (stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &st) != hipSuccess) { (void)hipGetLastError(); st = hipStreamCaptureStatusActive; }
    if (st != hipStreamCaptureStatusNone)
      return fail(GNX_ERR_INVALID_ARG, "the matrix-core tables of this handle are not built yet and the stream is being captured: call the "
                                       "gnx_*_workspace_bytes query of this layer (or one eager forward) before the capture");
  }
  std::lock_guard<std::mutex> lk(h->wide_mu);
  if (h->wide_built.load(std::memory_order_relaxed)) return GNX_OK;
  int32_t rc;
  {
    int prev = -1;
    (void)hipGetDevice(&prev);
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev != h->device ? prev : -1};
    if (prev != h->device && hipSetDevice(h->device) != hipSuccess) { (void)hipGetLastError(); return fail(GNX_ERR_INVALID_ARG, "cannot select the handle's device"); }
    rc = build_wide_tables_on_device(h);  // kernels over the handle's device arrays; 1 = not applicable
  }
  if (rc == 1) {
    drop_wide_tables(h);
    rc = gnx_ensure_host_csc(h);
    if (!rc) rc = build_wide_tables(h);
  }
  if (rc) { drop_wide_tables(h); return rc; }  // (the message of the failing step stays in gnx_last_error())
  h->wide_built.store(true, std::memory_order_release);
  return GNX_OK;
}

int32_t gnx_ensure_host_csc(const gnx_graphs* h) {
  if (!h) return fail(GNX_ERR_INVALID_ARG, "NULL handle");
  if (!h->csc_on_device_only) return GNX_OK;
  std::lock_guard<std::mutex> lk(h->host_csc_mu);
  if (h->h_colptr.size() == (size_t)h->N + 1) return GNX_OK;
  int prev = -1;
  (void)hipGetDevice(&prev);
  struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev != h->device ? prev : -1};
  if (prev != h->device) GNX_HIP(hipSetDevice(h->device));
  gnx::vec_i32 c32((size_t)h->N + 1), r32((size_t)h->E);
  GNX_HIP(hipMemcpy(c32.data(), h->d_colptr, c32.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (h->E) GNX_HIP(hipMemcpy(r32.data(), h->d_rowval, r32.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  gnx_graphs* m = const_cast<gnx_graphs*>(h);  // (logically const: a cache of what the device arrays hold)
  m->h_rowval.assign(r32.begin(), r32.end());
  m->h_colptr.assign(c32.begin(), c32.end());
  return GNX_OK;
}

int32_t gnx_ensure_csr(const gnx_graphs* h) {
  std::call_once(h->csr_once, [&] { h->csr_rc = gnx_ensure_host_csc(h); if (!h->csr_rc) h->csr_rc = build_csr(h); });
  return h->csr_rc;
}

int32_t gnx_ensure_collapse(const gnx_graphs* h) {
  std::call_once(h->collapse_once, [&] { h->collapse_rc = gnx_ensure_host_csc(h); if (!h->collapse_rc) h->collapse_rc = build_collapse(h); });
  return h->collapse_rc;
}

int32_t gnx_collapse_offsets(const gnx_graphs* h, int64_t* off) {
  if (!h || !off) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  int32_t rc = gnx_ensure_collapse(h);
  if (rc) return rc;
  memcpy(off, h->h_collapse_off.data(), h->h_collapse_off.size() * sizeof(int64_t));
  return GNX_OK;
}

int32_t gnx_graphs_get_info(const gnx_graphs* h, gnx_graphs_info* out) {
  if (!h || !out) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  out->n_graphs = h->G;
  out->n_nodes = h->N;
  out->n_edges = h->E;
  out->node_block_size = h->PN;
  out->edge_block_size = h->PN * h->PN;
  out->n_tiles = h->n_tiles();
  out->max_in_degree = h->max_in_degree;
  out->device = h->device;
  out->reserved = 0;
  return GNX_OK;
}

int32_t gnx_graphs_get_offsets(const gnx_graphs* h, int64_t* node_off, int64_t* edge_off) {
  if (!h) return fail(GNX_ERR_INVALID_ARG, "NULL handle");
  if (node_off) memcpy(node_off, h->h_node_off.data(), h->h_node_off.size() * sizeof(int64_t));
  if (edge_off) memcpy(edge_off, h->h_edge_off.data(), h->h_edge_off.size() * sizeof(int64_t));
  return GNX_OK;
}

// diagnostic: the handle's DEVICE tables copied to the host as they are (tests compare the device builder's with the host builder's)
int32_t gnx_graphs_get_table(const gnx_graphs* h, int32_t which, void* out, int64_t capacity_bytes, int64_t* bytes) {
  if (!h) return fail(GNX_ERR_INVALID_ARG, "NULL handle");
  const void* src[9] = {h->d_colptr, h->d_rowval, h->d_node_off, h->d_edge_off, h->d_tile_off, h->d_tiles, h->d_wtile_off, h->d_wtiles, h->d_packs};
  const size_t sz[9] = {(size_t)(h->N + 1) * 4, (size_t)h->E * 4, (size_t)(h->G + 1) * 4, (size_t)(h->G + 1) * 4, (size_t)(h->G + 1) * 4, (size_t)h->n_tiles() * sizeof(gnx::Tile),
                        (size_t)(h->G + 1) * 4, (size_t)h->n_wtiles() * sizeof(gnx::Tile), (size_t)h->n_packs * 8 * 4};
  if (which >= 9 && which <= 19) {  // the matrix-core path's tables (built now if they are not yet)
    if (int32_t rcw = gnx_ensure_wide_tables(h)) return rcw;
    const int64_t info[5] = {h->n_agg_rows, h->n_etiles_wide_span, h->n_etiles, h->n_ntiles, h->n_gtiles};
    const void* wsrc[10] = {h->d_etiles, h->d_ntiles, h->d_gtiles, h->d_etile_off, h->d_ntile_off, h->d_edge_dst, h->d_chunk_row0, h->d_node_agg_row, h->d_node_agg_parts,
                            h->d_node_agg_chunk};
    const size_t wsz[10] = {(size_t)h->n_etiles * sizeof(gnx::Tile), (size_t)h->n_ntiles * sizeof(gnx::Tile), (size_t)h->n_gtiles * sizeof(gnx::Tile), (size_t)(h->G + 1) * 4,
                            (size_t)(h->G + 1) * 4, (size_t)h->E * 4, (size_t)(2 * h->n_etiles + 1) * 4, (size_t)h->N * 4, (size_t)h->N * 4, (size_t)h->N * 4};
    const size_t need = which == 19 ? sizeof info : wsz[which - 9];
    if (bytes) *bytes = (int64_t)need;
    if (!out) return GNX_OK;
    if (capacity_bytes < (int64_t)need) return fail(GNX_ERR_INVALID_ARG, "buffer smaller than the table");
    if (which == 19) { memcpy(out, info, sizeof info); return GNX_OK; }
    int prevd = -1;
    (void)hipGetDevice(&prevd);
    struct RestoreW { int d; ~RestoreW() { if (d >= 0) (void)hipSetDevice(d); } } restorew{prevd != h->device ? prevd : -1};
    if (prevd != h->device) GNX_HIP(hipSetDevice(h->device));
    if (need) GNX_HIP(hipMemcpy(out, wsrc[which - 9], need, hipMemcpyDeviceToHost));
    return GNX_OK;
  }
  if (which < 0 || which > 8) return fail(GNX_ERR_INVALID_ARG, "which must be 0..19 (colptr, rowval, node_off, edge_off, tile_off, tiles, wtile_off, wtiles, packs; 9..19: the matrix-core tables)");
  if (bytes) *bytes = (int64_t)sz[which];
  if (!out) return GNX_OK;
  if (capacity_bytes < (int64_t)sz[which]) return fail(GNX_ERR_INVALID_ARG, "buffer smaller than the table");
  int prev = -1;
  (void)hipGetDevice(&prev);
  struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev != h->device ? prev : -1};
  if (prev != h->device) GNX_HIP(hipSetDevice(h->device));
  if (sz[which]) GNX_HIP(hipMemcpy(out, src[which], sz[which], hipMemcpyDeviceToHost));
  return GNX_OK;
}

int32_t gnx_graphs_get_csc(const gnx_graphs* h, int64_t* colptr, int64_t* rowval) {
  if (!h) return fail(GNX_ERR_INVALID_ARG, "NULL handle");
  if (int32_t rc = gnx_ensure_host_csc(h)) return rc;
  if (colptr) memcpy(colptr, h->h_colptr.data(), h->h_colptr.size() * sizeof(int64_t));
  if (rowval && !h->h_rowval.empty()) memcpy(rowval, h->h_rowval.data(), h->h_rowval.size() * sizeof(int64_t));
  return GNX_OK;
}

}  // extern "C"

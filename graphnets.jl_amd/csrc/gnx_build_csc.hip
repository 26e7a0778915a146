// Device-side GNGraphBatch construction from CSC (VERDICT r3 "next" 2).  The reference builds its batch with Julia loops
// over every PN^2 slot and calls `batch` in every training iteration (examples/sort/sort.jl:31-46, 122-126); here the packed colptr / rowval
// arrays of the graphs are uploaded as they are and everything O(N + E) happens in kernels:
//   k_csc_columns  one thread per destination column: colptr monotone / bounded, rowval in range and strictly increasing inside the column
//                  (what create_csc_impl checks on the host, same first-error order), and in the same sweep the device-format arrays —
//                  global int32 colptr (N + 1) and rowval (global source ids) — plus the largest in-degree;
//   k_tile_next    per node, for both tile kinds (workgroup tiles <= tile_n_cap nodes / tile_e_cap edges, wave tiles <= 64 / wtile_e_cap):
//                  where the greedy tile that STARTS at this node ends (binary search in colptr) — the host builder's inner loop;
//   k_tile_jump / k_tile_flag   the greedy tiling of a graph is the orbit of its first node under `next`: base-4 pointer doubling marks
//                  it for every graph at once (levels = digits of the largest possible tile count of a graph, known from sizes);
//   k_scan2_*      exclusive scan of the two flag arrays -> tile ids, tile counts per graph;
//   k_tile_write   the 32-byte tile records, per-graph tile offsets, wave tiles per graph.
// One small readback (status, counts) ends the build.  The tables are BIT-IDENTICAL to the host builder's (gnx_graphs.cpp::finalize; test:
// tests/test_gpu_build.py); the host builder stays for dense input, small batches and as the validator (GNX_BUILD_CSC_DEVICE=0).
#include <algorithm>
#include <climits>
#include <cstring>
#include <mutex>

#include "gnx_internal.h"

namespace gnx {

struct CscBuildStats {
  int first_bad;   // min over failures of 2 * column + kind (0: colptr, 1: rowval); INT_MAX: none — the host pass's first error
  int max_deg;
  int n_tiles, n_wtiles;
  int max_wtiles_per_graph;
  int pad[3];
};

struct CscBuildArgs {
  const int* node_off;  // [G+1]
  const int* edge_off;  // [G+1]
  int G, N, E, base;
  int tile_n_cap, tile_e_cap, wtile_n_cap, wtile_e_cap;
};

__device__ __forceinline__ int graph_of_node(const int* node_off, int G, int j) {
  int lo = 0, hi = G;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (node_off[mid] <= j) lo = mid; else hi = mid; }
  return lo;
}

template <typename IDX>
__global__ __launch_bounds__(256) void k_csc_columns(const IDX* __restrict__ cp_raw, const IDX* __restrict__ rv_raw, CscBuildArgs a, int* __restrict__ colptr,
                                                     int* __restrict__ rowval, int* __restrict__ node_graph, CscBuildStats* st) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  int deg = 0;
  if (j < a.N) {
    const int g = graph_of_node(a.node_off, a.G, j);
    node_graph[j] = g;
    const int n0 = a.node_off[g], n = a.node_off[g + 1] - n0, e0 = a.edge_off[g];
    const long long eg = a.edge_off[g + 1] - e0;
    const long long cpo = (long long)n0 + g;  // graph g's colptr starts behind the (n + 1)-entry colptrs of the graphs before it
    const long long lo = (long long)cp_raw[cpo + (j - n0)] - a.base, hi = (long long)cp_raw[cpo + (j - n0) + 1] - a.base;
    if (j == 0) colptr[0] = 0;
    if (hi < lo || hi - lo > n || hi > eg) {
      atomicMin(&st->first_bad, 2 * j);
      colptr[j + 1] = e0;  // (any in-range value: the build is abandoned)
    } else {
      colptr[j + 1] = e0 + (int)hi;
      if (lo >= 0) {  // (lo < 0: an earlier column of this graph is already flagged; nothing of this column is read)
        deg = (int)(hi - lo);
        long long prev = -1;
        bool bad = false;
        for (long long k = lo; k < hi; ++k) {
          const long long i = (long long)rv_raw[e0 + k] - a.base;
          bad |= (i <= prev) | (i >= n);
          prev = i;
          rowval[e0 + k] = n0 + (int)i;
        }
        if (bad) atomicMin(&st->first_bad, 2 * j + 1);
      }
    }
  }
  // largest in-degree: one atomic per wavefront
  for (int off = 32; off > 0; off >>= 1) deg = max(deg, __shfl_xor(deg, off));
  if ((threadIdx.x & 63) == 0 && deg > 0) atomicMax(&st->max_deg, deg);
}

// a CSC that is on the device already (global int32 colptr / rowval: the dense-adjacency builder's output): graph of every node, largest in-degree
__global__ __launch_bounds__(256) void k_adopt_csc(const int* __restrict__ colptr, CscBuildArgs a, int* __restrict__ node_graph, CscBuildStats* st) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  int deg = 0;
  if (j < a.N) {
    node_graph[j] = graph_of_node(a.node_off, a.G, j);
    deg = colptr[j + 1] - colptr[j];
  }
  for (int off = 32; off > 0; off >>= 1) deg = max(deg, __shfl_xor(deg, off));
  if ((threadIdx.x & 63) == 0 && deg > 0) atomicMax(&st->max_deg, deg);
}

// end of the greedy tile that starts at node j (gnx_graphs.cpp::finalize, build_tiles): nodes are added while the tile has < n_cap nodes
// and — except for its first node — its edges stay <= e_cap
__device__ __forceinline__ int tile_end(const int* __restrict__ colptr, int j, int nend, int n_cap, int e_cap) {
  const int hi = min(j + n_cap, nend);
  const int limit = colptr[j] + e_cap;
  int lo = j + 1, up = hi;  // largest x in [j + 1, hi] with colptr[x] <= limit (colptr is non-decreasing), or j + 1
  if (colptr[up] <= limit) return up;
  while (up - lo > 1) { const int mid = (lo + up) >> 1; if (colptr[mid] <= limit) lo = mid; else up = mid; }
  return lo;
}

// next[kind][j] for j < N, next[kind][N] = N; flag[kind][j] = 1 for the first node of every graph (the roots of the orbits)
// (A colptr that k_csc_columns flagged is not monotone: the greedy tiling of such an array has no bound, so nothing of the tiling runs on it —
// every thread of k_tile_next then writes the trivial orbit "no tile" and k_tile_write returns at once: same stream, so the flag is final.)
__global__ __launch_bounds__(256) void k_tile_next(const int* __restrict__ colptr, const int* __restrict__ node_graph, CscBuildArgs a, int* __restrict__ next,
                                                   int* __restrict__ flag, const CscBuildStats* __restrict__ st) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int S = a.N + 1;
  if (j > a.N) return;
  if (st->first_bad != INT_MAX) { next[j] = a.N; next[S + j] = a.N; flag[j] = 0; flag[S + j] = 0; return; }
  if (j == a.N) { next[j] = a.N; next[S + j] = a.N; flag[j] = 0; flag[S + j] = 0; return; }
  const int g = node_graph[j];
  const int nend = a.node_off[g + 1];
  next[j] = tile_end(colptr, j, nend, a.tile_n_cap, a.tile_e_cap);
  next[S + j] = tile_end(colptr, j, nend, a.wtile_n_cap, a.wtile_e_cap);
  const int root = a.node_off[g] == j;
  flag[j] = root; flag[S + j] = root;
}

// J_{k+1}[x] = J_k applied four times (both kinds: grid.y)
__global__ __launch_bounds__(256) void k_tile_jump(const int* __restrict__ jin, int* __restrict__ jout, int S) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= S) return;
  const int* in = jin + (size_t)blockIdx.y * S;
  int x = j;
  x = in[x]; x = in[x]; x = in[x]; x = in[x];
  jout[(size_t)blockIdx.y * S + j] = x;
}

// a flagged node flags J[x], J^2[x], J^3[x].  Levels are processed from the highest down: after level k every orbit position whose base-4
// digits below k are zero is flagged.  (A node flagged DURING this pass may propagate in it too: it only flags further true orbit members.)
__global__ __launch_bounds__(256) void k_tile_flag(const int* __restrict__ jl, int* __restrict__ flag, int S) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= S - 1) return;
  const int* J = jl + (size_t)blockIdx.y * S;
  int* F = flag + (size_t)blockIdx.y * S;
  if (!F[j]) return;
  int x = j;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    x = J[x];
    if (x < S - 1) F[x] = 1;
  }
}

// exclusive scan of flag[kind][0..S) (grid.y = kind), two levels, 2048 elements per block
constexpr int SCAN2_B = 2048;
__global__ __launch_bounds__(256) void k_scan2_blocks(const int* __restrict__ in, int S, int* __restrict__ out, int* __restrict__ block_sums, int nb) {
  __shared__ int s[256];
  const int b0 = blockIdx.x * SCAN2_B, t = threadIdx.x;
  const int* src = in + (size_t)blockIdx.y * S;
  int* dst = out + (size_t)blockIdx.y * S;
  int v[8], sum = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) { const int i = b0 + t * 8 + u; v[u] = i < S ? src[i] : 0; sum += v[u]; }
  s[t] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int x = t >= off ? s[t - off] : 0;
    __syncthreads();
    s[t] += x;
    __syncthreads();
  }
  int run = s[t] - sum;
#pragma unroll
  for (int u = 0; u < 8; ++u) { const int i = b0 + t * 8 + u; if (i < S) dst[i] = run; run += v[u]; }
  if (t == 255) block_sums[(size_t)blockIdx.y * nb + blockIdx.x] = s[255];
}
// one workgroup per kind scans the block sums in place (nb <= 2048)
__global__ __launch_bounds__(256) void k_scan2_sums(int* __restrict__ block_sums, int nb) {
  __shared__ int s[256];
  int* p = block_sums + (size_t)blockIdx.x * nb;
  const int t = threadIdx.x;
  int v[8], sum = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) { const int i = t * 8 + u; v[u] = i < nb ? p[i] : 0; sum += v[u]; }
  s[t] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int x = t >= off ? s[t - off] : 0;
    __syncthreads();
    s[t] += x;
    __syncthreads();
  }
  int run = s[t] - sum;
#pragma unroll
  for (int u = 0; u < 8; ++u) { const int i = t * 8 + u; if (i < nb) p[i] = run; run += v[u]; }
}

// tile records, per-graph tile offsets, wave tiles per graph; pos = scanned flags WITHOUT the block prefix (added here)
__global__ __launch_bounds__(256) void k_tile_write(const int* __restrict__ colptr, const int* __restrict__ node_graph, CscBuildArgs a, const int* __restrict__ next,
                                                    const int* __restrict__ flag, const int* __restrict__ pos, const int* __restrict__ block_prefix, int nb,
                                                    Tile* __restrict__ tiles, int* __restrict__ tile_off, Tile* __restrict__ wtiles, int* __restrict__ wtile_off,
                                                    int tiles_bound, int wtiles_bound, CscBuildStats* st) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int S = a.N + 1;
  if (j > a.N || st->first_bad != INT_MAX) return;  // (malformed input: the build is abandoned, no table is written)
  auto at = [&](int kind, int x) { return pos[(size_t)kind * S + x] + block_prefix[(size_t)kind * nb + x / SCAN2_B]; };
  if (j == a.N) {
    const int nt = at(0, j), nw = at(1, j);
    tile_off[a.G] = nt; wtile_off[a.G] = nw;
    st->n_tiles = nt; st->n_wtiles = nw;
    return;
  }
  const int g = node_graph[j];
  const int n0 = a.node_off[g], nend = a.node_off[g + 1];
  int wcount = 0;
  if (j == n0 || flag[S + j]) wcount = at(1, nend) - at(1, n0);  // wave tiles of graph g
  if (j == n0) {
    tile_off[g] = at(0, j); wtile_off[g] = at(1, j);
    atomicMax(&st->max_wtiles_per_graph, wcount);
  }
  if (flag[j]) {
    const int n1 = next[j];
    Tile t;
    t.n0 = j; t.n1 = n1; t.e0 = colptr[j]; t.e1 = colptr[n1]; t.g = g; t.win0 = n0; t.win1 = nend; t.flags = 0;
    const int slot = at(0, j);
    if (slot < tiles_bound) tiles[slot] = t;  // (the bound holds for every monotone colptr; the count is checked on the host)
  }
  if (flag[S + j]) {
    const int n1 = next[S + j];
    Tile t;
    t.n0 = j; t.n1 = n1; t.e0 = colptr[j]; t.e1 = colptr[n1]; t.g = g; t.win0 = n0; t.win1 = nend; t.flags = wcount;
    const int slot = at(1, j);
    if (slot < wtiles_bound) wtiles[slot] = t;
  }
}

// ---- process-wide caches: the build stream and scratch buffer of a device, and released handle arenas (hipMalloc / hipFree cost ~0.1-2 ms
// each and synchronise the device; a training loop builds a batch of the same size every iteration) ----
namespace {
// One build at a time per device: the scratch buffer and the build stream are shared by every handle built on that device, so a build holds
// `mu` from dev_scratch() until its last synchronisation (two host threads batching on one device serialise here; their kernels would
// serialise on the one stream anyway).  The entries are heap nodes: a pointer to one stays valid while another device's entry is added.
struct DevScratch { int dev; hipStream_t stream; void* buf; size_t cap; std::mutex mu; };
std::mutex g_cache_mu;
std::vector<std::unique_ptr<DevScratch>> g_scratch;
struct Arena { int dev; void* ptr; size_t bytes; };
std::vector<Arena> g_arenas;
size_t g_arena_bytes = 0;

size_t arena_cache_limit() {
  static const size_t lim = [] { const char* v = getenv("GNX_ARENA_CACHE_MB"); return (size_t)(v ? std::max(0, atoi(v)) : 1024) << 20; }();
  return lim;
}
}  // namespace

// a released arena that fits `bytes` (<= 2x, same device), or nullptr.  The previous owner's kernels may still be in flight on streams this
// library does not know: the device is synchronised before the block is written again (what hipFree did for the old owner, later)
void* arena_take(int dev, size_t bytes, size_t* got) {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  int best = -1;
  for (int i = 0; i < (int)g_arenas.size(); ++i)
    if (g_arenas[i].dev == dev && g_arenas[i].bytes >= bytes && g_arenas[i].bytes <= 2 * bytes + (1 << 20) && (best < 0 || g_arenas[i].bytes < g_arenas[best].bytes)) best = i;
  if (best < 0) return nullptr;
  void* p = g_arenas[best].ptr;
  *got = g_arenas[best].bytes;
  g_arena_bytes -= g_arenas[best].bytes;
  g_arenas.erase(g_arenas.begin() + best);
  (void)hipDeviceSynchronize();
  return p;
}

void arena_give(int dev, void* ptr, size_t bytes) {
  if (!ptr) return;
  if (bytes == 0 || bytes > arena_cache_limit() / 2) { (void)hipFree(ptr); return; }
  std::lock_guard<std::mutex> lk(g_cache_mu);
  g_arenas.push_back({dev, ptr, bytes});
  g_arena_bytes += bytes;
  while (g_arena_bytes > arena_cache_limit() || g_arenas.size() > 8) {  // oldest out
    (void)hipFree(g_arenas.front().ptr);
    g_arena_bytes -= g_arenas.front().bytes;
    g_arenas.erase(g_arenas.begin());
  }
}

static int32_t dev_scratch(int dev, size_t bytes, hipStream_t* stream, void** buf, std::unique_lock<std::mutex>* hold) {
  DevScratch* s = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (auto& x : g_scratch) if (x->dev == dev) s = x.get();
    if (!s) {
      g_scratch.emplace_back(new DevScratch{dev, nullptr, nullptr, 0, {}});
      s = g_scratch.back().get();
    }
  }
  *hold = std::unique_lock<std::mutex>(s->mu);  // (released by the caller's return: every exit path of a build)
  if (!s->stream) GNX_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  if (s->cap < bytes) {
    if (s->buf) { GNX_HIP(hipStreamSynchronize(s->stream)); GNX_HIP(hipFree(s->buf)); s->buf = nullptr; s->cap = 0; }
    const size_t cap = bytes + bytes / 4;
    GNX_HIP(hipMalloc(&s->buf, cap));
    s->cap = cap;
  }
  *stream = s->stream; *buf = s->buf;
  return GNX_OK;
}

// returns 1 when the device path does not apply (the caller runs the host builder), GNX_OK with a finished handle, or an error.
// Preconditions (checked by the caller's O(G) pass): h->G, N, E, PN, h_node_off, h_edge_off filled; arrays hold exactly N + G / E indices.
int32_t build_handle_from_csc_on_device(gnx_graphs* h, const void* colptr_cat, const void* rowval_cat, int32_t index_base, int32_t index_bits,
                                        int tile_e_cap, int tile_n_cap, int wtile_e_cap, int64_t tiles_bound, int64_t wtiles_bound, int64_t max_tiles_per_graph_bound) {
  const int64_t N = h->N, E = h->E, G = h->G;
  const int64_t S = N + 1;
  if (S > (int64_t)SCAN2_B * SCAN2_B) return 1;  // two-level scan capacity (4 M nodes)
  int levels = 1;
  for (int64_t c = 4; c < max_tiles_per_graph_bound; c *= 4) ++levels;  // 4^levels >= the largest possible number of tiles of a graph
  const size_t w = index_bits / 8;  // 0: colptr_cat / rowval_cat are DEVICE pointers to the global int32 colptr [N + 1] / rowval [E] (nothing to validate or convert)
  // device scratch: raw arrays, node -> graph, next / jump levels, flags, scanned flags, block sums, stats
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) / 256 * 256; return at; };
  const size_t o_cp = take((size_t)(N + G) * w), o_rv = take((size_t)std::max<int64_t>(E, 1) * w), o_ng = take((size_t)N * 4);
  const size_t o_jump = take((size_t)levels * 2 * S * 4), o_flag = take((size_t)2 * S * 4), o_pos = take((size_t)2 * S * 4);
  const int nb = (int)((S + SCAN2_B - 1) / SCAN2_B);
  const size_t o_bs = take((size_t)2 * nb * 4), o_st = take(sizeof(CscBuildStats));
  if (off > ((size_t)1 << 30)) return 1;  // (one graph of tens of millions of nodes: the host builder)
  hipStream_t s = nullptr;
  void* scratch = nullptr;
  std::unique_lock<std::mutex> build_lock;
  int32_t rc = dev_scratch(h->device, off, &s, &scratch, &build_lock);
  if (rc) return rc;
  char* sb = static_cast<char*>(scratch);
  // the handle's arena: colptr, rowval, node / edge / tile offsets, tiles, wave tiles, packs (capacity from the bounds)
  struct Slice { void** dst; size_t bytes; size_t at; };
  const size_t packs_bytes = G > 1 ? (size_t)G * 8 * sizeof(int32_t) : 0;
  Slice sl[] = {{(void**)&h->d_colptr, (size_t)S * 4, 0}, {(void**)&h->d_rowval, (size_t)E * 4, 0}, {(void**)&h->d_node_off, (size_t)(G + 1) * 4, 0},
                {(void**)&h->d_edge_off, (size_t)(G + 1) * 4, 0}, {(void**)&h->d_tile_off, (size_t)(G + 1) * 4, 0}, {(void**)&h->d_tiles, (size_t)tiles_bound * sizeof(Tile), 0},
                {(void**)&h->d_wtile_off, (size_t)(G + 1) * 4, 0}, {(void**)&h->d_wtiles, (size_t)wtiles_bound * sizeof(Tile), 0}, {(void**)&h->d_packs, packs_bytes, 0}};
  size_t total = 0;
  for (Slice& x : sl) { x.at = total; total += (std::max<size_t>(x.bytes, 16) + 255) / 256 * 256; }
  size_t got = 0;
  h->d_arena = arena_take(h->device, total, &got);
  if (h->d_arena) h->arena_bytes = got;
  else { GNX_HIP(hipMalloc(&h->d_arena, total)); h->arena_bytes = total; }
  for (Slice& x : sl) *x.dst = static_cast<char*>(h->d_arena) + x.at;
  // uploads: the raw index arrays as they are, the O(G) offsets
  std::vector<int32_t> off32((size_t)2 * (G + 1));
  for (int64_t g = 0; g <= G; ++g) { off32[(size_t)g] = (int32_t)h->h_node_off[(size_t)g]; off32[(size_t)(G + 1 + g)] = (int32_t)h->h_edge_off[(size_t)g]; }
  GNX_HIP(hipMemcpyAsync(h->d_node_off, off32.data(), (size_t)(G + 1) * 4, hipMemcpyHostToDevice, s));
  GNX_HIP(hipMemcpyAsync(h->d_edge_off, off32.data() + (G + 1), (size_t)(G + 1) * 4, hipMemcpyHostToDevice, s));
  if (index_bits == 0) {
    GNX_HIP(hipMemcpyAsync(h->d_colptr, colptr_cat, (size_t)S * 4, hipMemcpyDeviceToDevice, s));
    if (E) GNX_HIP(hipMemcpyAsync(h->d_rowval, rowval_cat, (size_t)E * 4, hipMemcpyDeviceToDevice, s));
  } else {
    GNX_HIP(hipMemcpyAsync(sb + o_cp, colptr_cat, (size_t)(N + G) * w, hipMemcpyHostToDevice, s));
    if (E) GNX_HIP(hipMemcpyAsync(sb + o_rv, rowval_cat, (size_t)E * w, hipMemcpyHostToDevice, s));
  }
  CscBuildStats init{};
  init.first_bad = INT_MAX;
  GNX_HIP(hipMemcpyAsync(sb + o_st, &init, sizeof init, hipMemcpyHostToDevice, s));
  CscBuildArgs a{h->d_node_off, h->d_edge_off, (int)G, (int)N, (int)E, index_base, tile_n_cap, tile_e_cap, 64, wtile_e_cap};
  int* node_graph = reinterpret_cast<int*>(sb + o_ng);
  int* jump = reinterpret_cast<int*>(sb + o_jump);
  int* flag = reinterpret_cast<int*>(sb + o_flag);
  int* pos = reinterpret_cast<int*>(sb + o_pos);
  int* bsum = reinterpret_cast<int*>(sb + o_bs);
  CscBuildStats* st = reinterpret_cast<CscBuildStats*>(sb + o_st);
  const unsigned gN = (unsigned)((N + 255) / 256), gS = (unsigned)((S + 255) / 256);
  if (index_bits == 0)
    GNX_LAUNCH(k_adopt_csc, dim3(gN), dim3(256), 0, s, h->d_colptr, a, node_graph, st);
  else if (index_bits == 64)
    GNX_LAUNCH((k_csc_columns<long long>), dim3(gN), dim3(256), 0, s, reinterpret_cast<const long long*>(sb + o_cp), reinterpret_cast<const long long*>(sb + o_rv), a, h->d_colptr,
               h->d_rowval, node_graph, st);
  else
    GNX_LAUNCH((k_csc_columns<int>), dim3(gN), dim3(256), 0, s, reinterpret_cast<const int*>(sb + o_cp), reinterpret_cast<const int*>(sb + o_rv), a, h->d_colptr, h->d_rowval,
               node_graph, st);
  GNX_LAUNCH(k_tile_next, dim3(gS), dim3(256), 0, s, h->d_colptr, node_graph, a, jump, flag, st);
  for (int l = 1; l < levels; ++l)
    GNX_LAUNCH(k_tile_jump, dim3(gS, 2), dim3(256), 0, s, jump + (size_t)(l - 1) * 2 * S, jump + (size_t)l * 2 * S, (int)S);
  for (int l = levels - 1; l >= 0; --l) GNX_LAUNCH(k_tile_flag, dim3(gS, 2), dim3(256), 0, s, jump + (size_t)l * 2 * S, flag, (int)S);
  GNX_LAUNCH(k_scan2_blocks, dim3((unsigned)nb, 2), dim3(256), 0, s, flag, (int)S, pos, bsum, nb);
  GNX_LAUNCH(k_scan2_sums, dim3(2), dim3(256), 0, s, bsum, nb);
  GNX_LAUNCH(k_tile_write, dim3(gS), dim3(256), 0, s, h->d_colptr, node_graph, a, jump, flag, pos, bsum, nb, h->d_tiles, h->d_tile_off, h->d_wtiles, h->d_wtile_off,
             (int)std::min<int64_t>(tiles_bound, INT_MAX), (int)std::min<int64_t>(wtiles_bound, INT_MAX), st);
  GNX_HIP(hipGetLastError());
  CscBuildStats out{};
  GNX_HIP(hipMemcpyAsync(&out, st, sizeof out, hipMemcpyDeviceToHost, s));
  GNX_HIP(hipStreamSynchronize(s));
  if (out.first_bad != INT_MAX)
    return fail(GNX_ERR_CSC, (out.first_bad & 1) ? "rowval out of range or not strictly increasing inside a column" : "colptr must be non-decreasing with at most N entries per column");
  if (out.n_tiles > tiles_bound || out.n_wtiles > wtiles_bound) return fail(GNX_ERR_INVALID_ARG, "device batch construction: more tiles than the bound the tables were sized with");
  h->max_in_degree = out.max_deg;
  h->n_tiles_ = out.n_tiles; h->n_wtiles_ = out.n_wtiles;
  h->max_wtiles_per_graph = out.max_wtiles_per_graph;
  h->csc_on_device_only = true;
  if (G > 1 && h->max_wtiles_per_graph >= 1 && h->max_wtiles_per_graph <= 8) {  // graph-aligned packs: a sequential best-fit over the graphs, on the host from G + 1 ints
    h->h_wtile_off.resize((size_t)G + 1);
    GNX_HIP(hipMemcpyAsync(h->h_wtile_off.data(), h->d_wtile_off, (size_t)(G + 1) * 4, hipMemcpyDeviceToHost, s));
    GNX_HIP(hipStreamSynchronize(s));
    std::vector<int32_t> packs;
    build_packs(h, packs);
    if ((size_t)h->n_packs * 8 * sizeof(int32_t) > packs_bytes) return fail(GNX_ERR_INVALID_ARG, "device batch construction: pack table larger than its bound");
    if (!packs.empty()) {
      GNX_HIP(hipMemcpyAsync(h->d_packs, packs.data(), packs.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
      GNX_HIP(hipStreamSynchronize(s));
    }
  }
  return GNX_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The matrix-core path's tables (gnx_internal.h: 128-row edge / node / graph tiles, the destination of every edge, the aggregation
// chunks of the edge GEMM's fused edge -> node sums) built by kernels from the handle's DEVICE arrays — every handle has those, whatever
// built it.  The host builder (gnx_graphs.cpp::build_wide_tables: loops over N + E on the lazily downloaded CSC, a dozen allocations
// and uploads, ~3-6 ms at 1M edges) stays as the validator (GNX_BUILD_WIDE_DEVICE=0) and for batches beyond the scan's capacity.
// Tables bit-identical (tests/test_gpu_build.py).  A training loop that rebuilds its batch every iteration pays this per iteration.
// ---------------------------------------------------------------------------------------------------------------------------------
struct WideArgs2 {
  const int *colptr, *node_off, *edge_off, *etile_off, *ntile_off;
  int G, N, E, n_etiles, n_ntiles, n_gtiles;
};
__device__ __forceinline__ int seg_of(const int* off, int n, int x) {  // i in [0, n) with off[i] <= x < off[i + 1] (off non-decreasing; empty segments skipped)
  int lo = 0, hi = n;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= x) lo = mid; else hi = mid; }
  return lo;
}
__global__ __launch_bounds__(256) void k_wide_tiles(WideArgs2 a, Tile* __restrict__ etiles, Tile* __restrict__ ntiles, Tile* __restrict__ gtiles) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n_etiles) {
    const int g = seg_of(a.etile_off, a.G, i);
    Tile t{};
    t.e0 = a.edge_off[g] + (i - a.etile_off[g]) * 128; t.e1 = min(t.e0 + 128, a.edge_off[g + 1]); t.g = g;
    t.n0 = a.node_off[g]; t.n1 = a.node_off[g + 1];
    etiles[i] = t;
  }
  if (i < a.n_ntiles) {
    const int g = seg_of(a.ntile_off, a.G, i);
    Tile t{};
    t.n0 = a.node_off[g] + (i - a.ntile_off[g]) * 128; t.n1 = min(t.n0 + 128, a.node_off[g + 1]); t.g = g;
    t.e0 = a.colptr[t.n0]; t.e1 = a.colptr[t.n1];
    ntiles[i] = t;
  }
  if (i < a.n_gtiles) {
    Tile t{};
    t.n0 = i * 128; t.n1 = min(i * 128 + 128, a.G); t.g = i * 128;
    gtiles[i] = t;
  }
}
// dst[e] = the node whose column holds edge e; nz[n] = 1 if node n has in-edges (nz[N] = 0)
__global__ __launch_bounds__(256) void k_wide_dst(const int* __restrict__ colptr, int N, int* __restrict__ dst, int* __restrict__ nz) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n > N) return;
  if (n == N) { nz[n] = 0; return; }
  const int e0 = colptr[n], e1 = colptr[n + 1];
  nz[n] = e1 > e0;
  for (int e = e0; e < e1; ++e) dst[e] = n;
}
__global__ __launch_bounds__(256) void k_scan_finish(int* __restrict__ out, int S, const int* __restrict__ block_prefix) {  // out[i] += prefix of its 2048-block
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < S) out[i] += block_prefix[i / SCAN2_B];
}
// rows of the partial-sum table per aggregation chunk (a 64-edge pass of a 128-edge tile) = distinct destinations in it; tiles whose
// destinations span more than kPdRowsCap consecutive nodes are counted (the edge GEMM's LDS form needs none)
__global__ __launch_bounds__(256) void k_wide_chunks(const Tile* __restrict__ etiles, int n_etiles, const int* __restrict__ dst, const int* __restrict__ nonempty,
                                                     int* __restrict__ cnt, int* __restrict__ stats) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > 2 * n_etiles) return;
  if (c == 2 * n_etiles) { cnt[c] = 0; return; }
  const Tile t = etiles[c >> 1];
  const int c0 = t.e0 + 64 * (c & 1), c1 = min(c0 + 64, t.e1);
  cnt[c] = c1 > c0 ? nonempty[dst[c1 - 1] + 1] - nonempty[dst[c0]] : 0;
  if ((c & 1) == 0 && t.e1 > t.e0 && dst[t.e1 - 1] - dst[t.e0] + 1 > kPdRowsCap) atomicAdd(&stats[1], 1);
}
__global__ __launch_bounds__(256) void k_wide_nodes(WideArgs2 a, const int* __restrict__ dst, const int* __restrict__ nonempty, const int* __restrict__ row0,
                                                    int* __restrict__ agg_row, int* __restrict__ parts, int* __restrict__ first_chunk, int* __restrict__ stats) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n == 0) stats[0] = row0[2 * a.n_etiles];  // n_agg_rows
  if (n >= a.N) return;
  const int e0 = a.colptr[n], e1 = a.colptr[n + 1];
  if (e1 <= e0) { agg_row[n] = -1; parts[n] = 0; first_chunk[n] = 0; return; }
  const int g = seg_of(a.node_off, a.G, n);
  const int eb = a.edge_off[g];
  const int ca = 2 * a.etile_off[g] + (e0 - eb) / 64, cb = 2 * a.etile_off[g] + (e1 - 1 - eb) / 64;
  const int chunk_e0 = eb + ((e0 - eb) / 64) * 64;  // first edge of chunk ca
  agg_row[n] = row0[ca] + (nonempty[n] - nonempty[dst[chunk_e0]]);
  parts[n] = cb - ca + 1;
  first_chunk[n] = ca;
}

// 1 = not applicable (the host builder runs); GNX_OK = the handle's wide tables are built (one device allocation: h->d_wide_arena)
int32_t build_wide_tables_on_device(const gnx_graphs* h) {
  static const bool on = !(getenv("GNX_BUILD_WIDE_DEVICE") && atoi(getenv("GNX_BUILD_WIDE_DEVICE")) == 0);
  if (!on) return 1;
  const int64_t N = h->N, E = h->E, G = h->G, net = h->n_etiles, nnt = h->n_ntiles, ngt = h->n_gtiles;
  const int64_t S1 = N + 1, S2 = 2 * net + 1;
  if (S1 > (int64_t)SCAN2_B * SCAN2_B || S2 > (int64_t)SCAN2_B * SCAN2_B) return 1;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += (std::max<size_t>(bytes, 16) + 255) / 256 * 256; return at; };
  const size_t o_et = take((size_t)net * sizeof(Tile)), o_nt = take((size_t)nnt * sizeof(Tile)), o_gt = take((size_t)ngt * sizeof(Tile));
  const size_t o_eo = take((size_t)(G + 1) * 4), o_no = take((size_t)(G + 1) * 4), o_dst = take((size_t)E * 4), o_row0 = take((size_t)S2 * 4);
  const size_t o_ar = take((size_t)N * 4), o_pa = take((size_t)N * 4), o_fc = take((size_t)N * 4);
  const size_t persistent = off;
  // scratch behind the persistent part of the same allocation (freed with it: a few MB)
  const int nb1 = (int)((S1 + SCAN2_B - 1) / SCAN2_B), nb2 = (int)((S2 + SCAN2_B - 1) / SCAN2_B);
  const size_t o_nz = take((size_t)S1 * 4), o_ne = take((size_t)S1 * 4), o_cnt = take((size_t)S2 * 4), o_bs = take((size_t)std::max(nb1, nb2) * 4), o_st = take(16);
  (void)persistent;
  hipStream_t s = nullptr;
  void* scratch = nullptr;
  std::unique_lock<std::mutex> build_lock;
  int32_t rc = dev_scratch(h->device, 256, &s, &scratch, &build_lock);  // (the device's build stream)
  if (rc) return rc;
  char* base = nullptr;
  GNX_HIP(hipMalloc((void**)&base, off));
  h->d_wide_arena = base;
  h->d_etiles = reinterpret_cast<Tile*>(base + o_et); h->d_ntiles = reinterpret_cast<Tile*>(base + o_nt); h->d_gtiles = reinterpret_cast<Tile*>(base + o_gt);
  h->d_etile_off = reinterpret_cast<int*>(base + o_eo); h->d_ntile_off = reinterpret_cast<int*>(base + o_no); h->d_edge_dst = reinterpret_cast<int*>(base + o_dst);
  h->d_chunk_row0 = reinterpret_cast<int*>(base + o_row0); h->d_node_agg_row = reinterpret_cast<int*>(base + o_ar);
  h->d_node_agg_parts = reinterpret_cast<int*>(base + o_pa); h->d_node_agg_chunk = reinterpret_cast<int*>(base + o_fc);
  int* nz = reinterpret_cast<int*>(base + o_nz);
  int* nonempty = reinterpret_cast<int*>(base + o_ne);
  int* cnt = reinterpret_cast<int*>(base + o_cnt);
  int* bs = reinterpret_cast<int*>(base + o_bs);
  int* stats = reinterpret_cast<int*>(base + o_st);
  GNX_HIP(hipMemcpyAsync(h->d_etile_off, h->h_etile_off.data(), (size_t)(G + 1) * 4, hipMemcpyHostToDevice, s));
  GNX_HIP(hipMemcpyAsync(h->d_ntile_off, h->h_ntile_off.data(), (size_t)(G + 1) * 4, hipMemcpyHostToDevice, s));
  GNX_HIP(hipMemsetAsync(stats, 0, 16, s));
  WideArgs2 a{h->d_colptr, h->d_node_off, h->d_edge_off, h->d_etile_off, h->d_ntile_off, (int)G, (int)N, (int)E, (int)net, (int)nnt, (int)ngt};
  const int64_t mt = std::max(std::max(net, nnt), std::max<int64_t>(ngt, 1));
  GNX_LAUNCH(k_wide_tiles, dim3((unsigned)((mt + 255) / 256)), dim3(256), 0, s, a, h->d_etiles, h->d_ntiles, h->d_gtiles);
  GNX_LAUNCH(k_wide_dst, dim3((unsigned)((S1 + 255) / 256)), dim3(256), 0, s, h->d_colptr, (int)N, h->d_edge_dst, nz);
  GNX_LAUNCH(k_scan2_blocks, dim3((unsigned)nb1, 1), dim3(256), 0, s, nz, (int)S1, nonempty, bs, nb1);
  GNX_LAUNCH(k_scan2_sums, dim3(1), dim3(256), 0, s, bs, nb1);
  GNX_LAUNCH(k_scan_finish, dim3((unsigned)((S1 + 255) / 256)), dim3(256), 0, s, nonempty, (int)S1, bs);
  GNX_LAUNCH(k_wide_chunks, dim3((unsigned)((S2 + 255) / 256)), dim3(256), 0, s, h->d_etiles, (int)net, h->d_edge_dst, nonempty, cnt, stats);
  GNX_LAUNCH(k_scan2_blocks, dim3((unsigned)nb2, 1), dim3(256), 0, s, cnt, (int)S2, h->d_chunk_row0, bs, nb2);
  GNX_LAUNCH(k_scan2_sums, dim3(1), dim3(256), 0, s, bs, nb2);
  GNX_LAUNCH(k_scan_finish, dim3((unsigned)((S2 + 255) / 256)), dim3(256), 0, s, h->d_chunk_row0, (int)S2, bs);
  GNX_LAUNCH(k_wide_nodes, dim3((unsigned)((std::max<int64_t>(N, 1) + 255) / 256)), dim3(256), 0, s, a, h->d_edge_dst, nonempty, h->d_chunk_row0, h->d_node_agg_row,
             h->d_node_agg_parts, h->d_node_agg_chunk, stats);
  GNX_HIP(hipGetLastError());
  int out[4] = {0, 0, 0, 0};
  GNX_HIP(hipMemcpyAsync(out, stats, 16, hipMemcpyDeviceToHost, s));
  GNX_HIP(hipStreamSynchronize(s));
  h->n_agg_rows = out[0];
  h->n_etiles_wide_span = out[1];
  if (h->n_agg_rows > h->agg_rows_bound) return fail(GNX_ERR_INVALID_ARG, "wide tables: more aggregation rows than the bound workspaces are sized with");
  return GNX_OK;
}

}  // namespace gnx

// Device-side GNGraphBatch construction: dense 0/1 adjacency matrices -> CSC on the GPU.
// The reference builds its batch with single-threaded Julia loops over every PN^2 slot (src/gngraphbatch.jl:33-54,
// src/pad.jl:26-64); the host path of gnx_graphs_create_dense does the same O(sum N_g^2) scan in C++.  For large dense
// batches (BASELINE config 5: 4096 graphs, ~85M adjacency entries) the scan/compaction runs here instead:
//   k_adj_count : one wavefront per (graph, destination column): lanes stride over the rows, ballot + popcount
//   scan        : two-level exclusive prefix sum of the per-column counts -> colptr
//   k_adj_fill  : same traversal, ballot-prefix compaction of the source ids into rowval (edge order = CSC order)
#include <algorithm>
#include <cstring>
#include <mutex>
#include <thread>

#include "gnx_internal.h"

namespace gnx {

struct AdjMeta {
  const int64_t* adj_off;   // [G] element offset of graph g's matrix in the packed buffer
  const int32_t* n;         // [G]
  const int32_t* node_off;  // [G+1]
  int G;
  int elem_kind, row_major;
};

__device__ __forceinline__ int adj_elem(const void* base, int kind, int64_t idx, int* bad) {
  double v;
  switch (kind) {
    case GNX_ELEM_U8: v = reinterpret_cast<const uint8_t*>(base)[idx]; break;
    case GNX_ELEM_I32: v = reinterpret_cast<const int32_t*>(base)[idx]; break;
    case GNX_ELEM_I64: v = (double)reinterpret_cast<const int64_t*>(base)[idx]; break;
    case GNX_ELEM_F32: v = reinterpret_cast<const float*>(base)[idx]; break;
    default: v = reinterpret_cast<const double*>(base)[idx]; break;
  }
  if (v == 0.0) return 0;
  if (v == 1.0) return 1;
  *bad = 1;
  return 0;
}

// FILL = false: counts[c] = #ones of column c;  FILL = true: rowval[colptr[c] + k] = global source id of the k-th one
template <bool FILL>
__global__ __launch_bounds__(256) void k_adj_columns(const void* adj, AdjMeta m, int N, int* counts_or_colptr, int* rowval, int* bad_flag) {
  const int lane = threadIdx.x & 63;
  const int c = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);  // global column = destination node
  if (c >= N) return;
  int lo = 0, hi = m.G;  // graph of column c
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (m.node_off[mid] <= c) lo = mid; else hi = mid; }
  const int g = lo, n = m.n[g], j = c - m.node_off[g];
  const int64_t base = m.adj_off[g];
  int bad = 0, total = 0;
  const int out0 = FILL ? counts_or_colptr[c] : 0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    int one = 0;
    if (i < n) one = adj_elem(adj, m.elem_kind, base + (m.row_major ? (int64_t)i * n + j : (int64_t)j * n + i), &bad);
    const unsigned long long mask = __ballot(one);
    if (FILL && one) rowval[out0 + total + __popcll(mask & ((1ull << lane) - 1ull))] = m.node_off[g] + i;
    total += __popcll(mask);
  }
  if (!FILL) {
    if (lane == 0) counts_or_colptr[c] = total;
    if (bad) atomicOr(bad_flag, 1);
  }
}

// exclusive scan, two levels: 2048 elements per block
constexpr int SCAN_B = 2048;
__global__ __launch_bounds__(256) void k_scan_blocks(const int* in, int n, int* out, int* block_sums) {
  __shared__ int s[256];
  const int b0 = blockIdx.x * SCAN_B, t = threadIdx.x;
  int v[8], sum = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) { const int i = b0 + t * 8 + u; v[u] = i < n ? in[i] : 0; sum += v[u]; }
  s[t] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int x = t >= off ? s[t - off] : 0;
    __syncthreads();
    s[t] += x;
    __syncthreads();
  }
  int run = s[t] - sum;  // exclusive prefix of this thread inside the block
#pragma unroll
  for (int u = 0; u < 8; ++u) { const int i = b0 + t * 8 + u; if (i < n) out[i] = run; run += v[u]; }
  if (t == 255 && block_sums) block_sums[blockIdx.x] = s[255];
}
__global__ void k_scan_add(int* out, int n, const int* block_prefix) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += block_prefix[i / SCAN_B];
}

// per-graph edge offsets from the global colptr: edge_off[g] = colptr[node_off[g]]
__global__ void k_gather_offsets(const int* __restrict__ cp, const int* __restrict__ node_off, int G, int* __restrict__ edge_off) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g <= G) edge_off[g] = cp[node_off[g]];
}

// {bad flag, number of edges} in one small readback
__global__ void k_adj_summary(const int* __restrict__ cp, int N, const int* __restrict__ bad, int* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = *bad; out[1] = cp[N]; }
}

// Returns 1 when the device path does not apply (caller falls back to the host scan).
// keep != nullptr: the CSC stays on the device (keep->d_colptr [N+1], keep->d_rowval [E], inside blocks the caller hands back to the arena cache
// with release_dense_csc) and only the G + 1 edge offsets come back — the handle's tables are then built by the device builder
// (gnx_build_csc.hip) without a round trip through the host.
// The matrices: adj[g] (G host pointers), or — packed != nullptr — ONE buffer holding them one after the other, on the host (packed_on_device
// = 0; pinned memory travels as one DMA, pageable memory through the pinned staging pair) or already on the device (1: no copy at all).
// Device memory comes from the process-wide cache of released blocks (arena_take / arena_give): a loop that batches every iteration
// allocates nothing after its first build.
int32_t build_csc_on_device(const void* const* adj, const void* packed, int packed_on_device, const int64_t* n_nodes, int64_t G, int32_t elem_kind, int32_t row_major,
                            gnx::vec_i64& h_colptr, gnx::vec_i64& h_rowval, const std::vector<int64_t>& h_node_off, DenseCscOnDevice* keep) {
  const int64_t N = h_node_off.back();
  if (N <= 0 || N + 1 > (int64_t)SCAN_B * SCAN_B) return 1;  // two-level scan capacity
  static const size_t esz_tab[5] = {1, 4, 8, 4, 8};
  const size_t esz = esz_tab[elem_kind];
  int dev = 0;
  { const hipError_t e = hipGetDevice(&dev); if (e != hipSuccess) return hip_fail(e, "hipGetDevice"); }
  // host-side metadata in ONE array: adj_off [G] int64 | n [G] int32 | node_off [G+1] int32
  const size_t meta_bytes = align_up((size_t)G * 8, 16) + align_up((size_t)G * 4, 16) + align_up((size_t)(G + 1) * 4, 16);
  std::vector<char> meta(meta_bytes);
  int64_t* adj_off = reinterpret_cast<int64_t*>(meta.data());
  int32_t* n32 = reinterpret_cast<int32_t*>(meta.data() + align_up((size_t)G * 8, 16));
  int32_t* node_off32 = reinterpret_cast<int32_t*>(meta.data() + align_up((size_t)G * 8, 16) + align_up((size_t)G * 4, 16));
  int64_t total = 0;
  for (int64_t g = 0; g < G; ++g) { adj_off[g] = total; total += n_nodes[g] * n_nodes[g]; n32[g] = (int32_t)n_nodes[g]; node_off32[g] = (int32_t)h_node_off[g]; }
  node_off32[G] = (int32_t)N;
  const size_t bytes_total = (size_t)total * esz;
  // one device block for everything temporary: [adjacency (unless it is on the device already) | meta | counts | block sums x 2 | bad | summary | colptr]
  size_t off = 0;
  auto take = [&](size_t b) { const size_t at = off; off += align_up(std::max<size_t>(b, 16), 256); return at; };
  const size_t o_adj = take(packed_on_device ? 16 : bytes_total), o_meta = take(meta_bytes), o_cnt = take((size_t)(N + 1) * 4), o_bs = take(SCAN_B * 4), o_bad = take(16);
  const size_t o_zero_end = off;  // [o_cnt, o_zero_end) starts as zeros
  const size_t o_bp = take(SCAN_B * 4), o_sum = take(16), o_cp = take((size_t)(N + 1) * 4), o_eoff = take((size_t)(G + 1) * 4);
  size_t got = 0;
  char* blk = static_cast<char*>(arena_take(dev, off, &got));
  if (!blk) { const hipError_t e = hipMalloc((void**)&blk, off); if (e != hipSuccess) return hip_fail(e, "dense batch: device block"); got = off; }
  void* rv_blk = nullptr;
  size_t rv_got = 0;
  auto cleanup = [&]() { arena_give(dev, blk, got); if (rv_blk) arena_give(dev, rv_blk, rv_got); };
#define GNX_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return hip_fail(_e, #expr); } } while (0)
  const void* d_adj = packed_on_device ? packed : blk + o_adj;
  if (!packed_on_device) {
    hipPointerAttribute_t at{};
    const bool pinned = packed && hipPointerGetAttributes(&at, packed) == hipSuccess && at.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    if (pinned) {
      GNX_TRY(hipMemcpyAsync(blk + o_adj, packed, bytes_total, hipMemcpyHostToDevice, nullptr));  // one DMA at the link's rate
    } else {
      // Pageable host memory (G separate arrays, or one packed buffer).  One hipMemcpy per graph is a driver round trip per graph (4096
      // graphs: ~200 ms); instead the bytes are packed into two pinned staging buffers that alternate — while one travels (asynchronous copy,
      // full PCIe rate) the host fills the other — so the upload costs one host memcpy of the bytes plus a handful of DMA transfers.
      // The staging buffers are process-wide and PORTABLE pinned memory (usable from every device: a one-process-many-devices host —
      // gnx_dist_*, a Julia session — builds handles on devices 1..n-1 too); the two events belong to the CURRENT device and live for
      // this call only (an event of device 0 cannot be recorded on a stream of device r).  A failure to set the staging up is not an
      // error of the batch: the caller falls back to the host scan (return 1).
      static std::mutex stage_mu;
      static char* stage[2] = {nullptr, nullptr};
      constexpr size_t STAGE = (size_t)32 << 20;
      std::lock_guard<std::mutex> lk(stage_mu);
      hipEvent_t stage_ev[2] = {nullptr, nullptr};
      struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int b = 0; b < 2; ++b) if (e[b]) (void)hipEventDestroy(e[b]); } } ev_guard{stage_ev};
      for (int b = 0; b < 2; ++b) {
        if (!stage[b] && hipHostMalloc((void**)&stage[b], STAGE, hipHostMallocPortable) != hipSuccess) { stage[b] = nullptr; (void)hipGetLastError(); cleanup(); return 1; }
        if (hipEventCreateWithFlags(&stage_ev[b], hipEventDisableTiming) != hipSuccess) { stage_ev[b] = nullptr; (void)hipGetLastError(); cleanup(); return 1; }
      }
      size_t done = 0;      // bytes of the packed adjacency stream already handed to a copy
      int64_t g = 0;        // current graph
      size_t g_done = 0;    // bytes of graph g already staged
      int buf = 0;
      bool in_flight[2] = {false, false};
      const unsigned n_thr_max = std::min<unsigned>(4, std::max(1u, std::thread::hardware_concurrency()));
      while (done < bytes_total) {
        if (in_flight[buf]) { GNX_TRY(hipEventSynchronize(stage_ev[buf])); in_flight[buf] = false; }
        size_t fill = 0;
        struct Seg { char* dst; const char* src; size_t bytes; };
        std::vector<Seg> segs;
        if (packed) {  // one contiguous source: equal pieces for the copy threads
          fill = std::min(STAGE, bytes_total - done);
          const size_t piece = align_up((fill + n_thr_max - 1) / n_thr_max, 4096);
          for (size_t o = 0; o < fill; o += piece) segs.push_back({stage[buf] + o, static_cast<const char*>(packed) + done + o, std::min(piece, fill - o)});
        } else {
          while (fill < STAGE && g < G) {
            const size_t gbytes = (size_t)(n_nodes[g] * n_nodes[g]) * esz;
            const size_t tk = std::min(gbytes - g_done, STAGE - fill);
            segs.push_back({stage[buf] + fill, static_cast<const char*>(adj[g]) + g_done, tk});
            fill += tk; g_done += tk;
            if (g_done == gbytes) { ++g; g_done = 0; }
          }
        }
        {
          // the host copy into the staging buffer is what this upload costs (one thread: ~10 GB/s against the link's ~50): a few threads
          // take contiguous runs of the segments (about equal bytes each)
          const int n_thr = fill >= ((size_t)4 << 20) ? (int)n_thr_max : 1;
          auto copy_range = [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) memcpy(segs[i].dst, segs[i].src, segs[i].bytes); };
          if (n_thr <= 1) copy_range(0, segs.size());
          else {
            std::vector<size_t> cut((size_t)n_thr + 1, segs.size());
            cut[0] = 0;
            size_t acc = 0; int t = 1;
            for (size_t i = 0; i < segs.size() && t < n_thr; ++i) { acc += segs[i].bytes; if (acc >= fill * (size_t)t / n_thr) cut[(size_t)t++] = i + 1; }
            std::vector<std::thread> thr;
            for (int i = 1; i < n_thr; ++i) thr.emplace_back(copy_range, cut[(size_t)i], cut[(size_t)i + 1]);
            copy_range(cut[0], cut[1]);
            for (auto& th : thr) th.join();
          }
        }
        GNX_TRY(hipMemcpyAsync(blk + o_adj + done, stage[buf], fill, hipMemcpyHostToDevice, nullptr));
        GNX_TRY(hipEventRecord(stage_ev[buf], nullptr));
        in_flight[buf] = true;
        done += fill;
        buf ^= 1;
      }
      for (int b = 0; b < 2; ++b)
        if (in_flight[b]) GNX_TRY(hipEventSynchronize(stage_ev[b]));
    }
  }
  GNX_TRY(hipMemcpyAsync(blk + o_meta, meta.data(), meta_bytes, hipMemcpyHostToDevice, nullptr));
  GNX_TRY(hipMemsetAsync(blk + o_cnt, 0, o_zero_end - o_cnt, nullptr));
  const int64_t* d_off = reinterpret_cast<const int64_t*>(blk + o_meta);
  const int32_t* d_n = reinterpret_cast<const int32_t*>(blk + o_meta + align_up((size_t)G * 8, 16));
  const int32_t* d_noff = reinterpret_cast<const int32_t*>(blk + o_meta + align_up((size_t)G * 8, 16) + align_up((size_t)G * 4, 16));
  int32_t* d_cnt = reinterpret_cast<int32_t*>(blk + o_cnt);
  int32_t* d_bs = reinterpret_cast<int32_t*>(blk + o_bs);
  int32_t* d_bad = reinterpret_cast<int32_t*>(blk + o_bad);
  int32_t* d_bp = reinterpret_cast<int32_t*>(blk + o_bp);
  int32_t* d_sum = reinterpret_cast<int32_t*>(blk + o_sum);
  int32_t* d_cp = reinterpret_cast<int32_t*>(blk + o_cp);
  const int nb = (int)((N + 1 + SCAN_B - 1) / SCAN_B);
  AdjMeta m{d_off, d_n, d_noff, (int)G, elem_kind, row_major};
  const unsigned grid = (unsigned)((N * 64 + 255) / 256);
  GNX_LAUNCH(k_adj_columns<false>, dim3(grid), dim3(256), 0, 0, d_adj, m, (int)N, d_cnt, (int*)nullptr, d_bad);
  GNX_LAUNCH(k_scan_blocks, dim3(nb), dim3(256), 0, 0, d_cnt, (int)(N + 1), d_cp, d_bs);
  GNX_LAUNCH(k_scan_blocks, dim3(1), dim3(256), 0, 0, d_bs, SCAN_B, d_bp, (int*)nullptr);
  GNX_LAUNCH(k_scan_add, dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, 0, d_cp, (int)(N + 1), d_bp);
  GNX_LAUNCH(k_adj_summary, dim3(1), dim3(64), 0, 0, d_cp, (int)N, d_bad, d_sum);
  GNX_TRY(hipGetLastError());
  int32_t summary[2] = {0, 0};
  GNX_TRY(hipMemcpy(summary, d_sum, sizeof summary, hipMemcpyDeviceToHost));
  if (summary[0]) { cleanup(); return fail(GNX_ERR_ADJ_VALUE, "adjacency entries must be exactly 0 or 1 (pad.jl:30, gngraphbatch.jl:207)"); }
  const int32_t E = summary[1];
  const size_t rv_bytes = std::max<size_t>((size_t)E, 4) * sizeof(int32_t);
  rv_blk = arena_take(dev, rv_bytes, &rv_got);
  if (!rv_blk) { GNX_TRY(hipMalloc(&rv_blk, rv_bytes)); rv_got = rv_bytes; }
  int32_t* d_rv = static_cast<int32_t*>(rv_blk);
  GNX_LAUNCH(k_adj_columns<true>, dim3(grid), dim3(256), 0, 0, d_adj, m, (int)N, d_cp, d_rv, d_bad);
  GNX_TRY(hipGetLastError());
  if (keep) {
    int32_t* d_eoff = reinterpret_cast<int32_t*>(blk + o_eoff);
    GNX_LAUNCH(k_gather_offsets, dim3((unsigned)((G + 1 + 255) / 256)), dim3(256), 0, 0, d_cp, d_noff, (int)G, d_eoff);
    keep->edge_off.resize((size_t)G + 1);
    GNX_TRY(hipMemcpy(keep->edge_off.data(), d_eoff, (G + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
    keep->d_colptr = d_cp; keep->d_rowval = d_rv; keep->E = E;
    keep->device = dev; keep->block = blk; keep->block_bytes = got; keep->rv_block = rv_blk; keep->rv_bytes = rv_got;  // (handed over: release_dense_csc)
    return GNX_OK;
  }
  std::vector<int32_t> cp32(N + 1), rv32((size_t)E);
  GNX_TRY(hipMemcpy(cp32.data(), d_cp, (N + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (E) GNX_TRY(hipMemcpy(rv32.data(), d_rv, (size_t)E * sizeof(int32_t), hipMemcpyDeviceToHost));
  cleanup();
#undef GNX_TRY
  h_colptr.assign(cp32.begin(), cp32.end());
  h_rowval.assign(rv32.begin(), rv32.end());
  return GNX_OK;
}

// the blocks behind a DenseCscOnDevice go back to the cache of released device blocks (the caller's kernels on them have been enqueued on the
// NULL stream or synchronised: arena_take synchronises the device before a block is written again)
void release_dense_csc(DenseCscOnDevice& k) {
  if (k.block) arena_give(k.device, k.block, k.block_bytes);
  if (k.rv_block) arena_give(k.device, k.rv_block, k.rv_bytes);
  k.block = k.rv_block = nullptr; k.d_colptr = nullptr; k.d_rowval = nullptr;
}

}  // namespace gnx

// Host-only part of the multi-GPU split (no device, no RCCL): the partitioner and the gather plan — what decides which rank owns which
// graph and where a graph's gf' row lands in the gathered table.  Plain C++ so that the CPU sanitiser build (tests/c/host_asan_driver.cpp)
// and CPU tests can drive it; gnx_dist.hip builds its communicator from these.
#include <algorithm>
#include <numeric>
#include <vector>

#include "gnx_internal.h"

using namespace gnx;

extern "C" {

int32_t gnx_dist_partition(const int64_t* edge_counts, int64_t n_graphs, int32_t n_ranks, int64_t* shard_off, int64_t* shard_graphs) {
  if (!edge_counts || !shard_off || !shard_graphs) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  if (n_graphs <= 0) return fail(GNX_ERR_NO_GRAPHS, "n_graphs must be > 0");
  if (n_ranks <= 0) return fail(GNX_ERR_INVALID_ARG, "n_ranks must be >= 1");
  // graphs by edge count, descending (stable: ties keep ascending ids), dealt in snake order: rank 0..R-1, R-1..0, ...
  std::vector<int64_t> order((size_t)n_graphs);
  std::iota(order.begin(), order.end(), (int64_t)0);
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return edge_counts[a] > edge_counts[b]; });
  std::vector<std::vector<int64_t>> shards((size_t)n_ranks);
  for (int64_t i = 0; i < n_graphs; ++i) {
    const int64_t rnd = i / n_ranks, pos = i % n_ranks;
    shards[(size_t)(rnd % 2 == 0 ? pos : n_ranks - 1 - pos)].push_back(order[(size_t)i]);
  }
  int64_t o = 0;
  for (int32_t r = 0; r < n_ranks; ++r) {
    std::sort(shards[(size_t)r].begin(), shards[(size_t)r].end());  // a rank keeps its graphs in original order
    shard_off[r] = o;
    for (int64_t gidx : shards[(size_t)r]) shard_graphs[o++] = gidx;
  }
  shard_off[n_ranks] = o;
  return GNX_OK;
}

// Validation of a partition and the gather plan of its all-gather (host only, no device): every rank contributes max_count rows
// (zero padded), so original graph g = shard_graphs[shard_off[r] + k] is row r * max_count + k of the gathered table.
int32_t gnx_dist_gather_plan(const int64_t* shard_off, const int64_t* shard_graphs, int32_t n_ranks, int64_t n_graphs, int32_t* src_row,
                             int64_t* max_count_out) {
  if (!shard_off || !shard_graphs) return fail(GNX_ERR_INVALID_ARG, "NULL argument");
  if (n_ranks <= 0 || n_graphs <= 0) return fail(GNX_ERR_INVALID_ARG, "n_ranks and n_graphs must be >= 1");
  if (shard_off[0] != 0 || shard_off[n_ranks] != n_graphs) return fail(GNX_ERR_INVALID_ARG, "shard_off must run from 0 to n_graphs");
  int64_t max_count = 0;
  for (int r = 0; r < n_ranks; ++r) {
    if (shard_off[r + 1] < shard_off[r]) return fail(GNX_ERR_INVALID_ARG, "shard_off must be non-decreasing");
    max_count = std::max(max_count, shard_off[r + 1] - shard_off[r]);
  }
  if ((uint64_t)max_count * (uint64_t)n_ranks >= (1ull << 31)) return fail(GNX_ERR_TOO_LARGE, "gathered table exceeds int32 row indices");
  std::vector<char> seen((size_t)n_graphs, 0);
  for (int r = 0; r < n_ranks; ++r) {
    for (int64_t i = shard_off[r]; i < shard_off[r + 1]; ++i) {
      const int64_t gidx = shard_graphs[i];
      if (gidx < 0 || gidx >= n_graphs || seen[(size_t)gidx]) return fail(GNX_ERR_INVALID_ARG, "shard_graphs must be a permutation of 0..n_graphs-1");
      seen[(size_t)gidx] = 1;
      if (src_row) src_row[gidx] = (int32_t)(r * max_count + (i - shard_off[r]));
    }
  }
  if (max_count_out) *max_count_out = max_count;
  return GNX_OK;
}

}  // extern "C"

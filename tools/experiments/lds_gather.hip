// Experiment (not part of the library): north_star asks for "LDS staging of node tiles for the gather"; k_block_wave
// gathers nf[src] straight from the L2.  For heterogeneous batches every graph's node table is <= 256 rows x 20 B = 5 KB: does staging it in
// LDS (coalesced copy, barrier, gather from LDS) beat the L2 gather?  Same bytes, same work items — a workgroup per (graph, chunk of <= 512
// edges): 40-B edge row in, source index, 20-B source row (A: from global memory = L2 hits for graph-local sources; B: from the staged
// table), 12-B row out — on the C3 law (512 graphs) and the C5 law (4096 graphs), 1M edges, hipGraph of launches over rotating buffer sets
// (cache-cold streams).   hipcc -O3 --offload-arch=gfx950 tools/experiments/lds_gather.hip -o /tmp/lds_gather && /tmp/lds_gather
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) F3u { float x, y, z; };
struct __attribute__((packed, aligned(4))) F2u { float x, y; };
struct Item { int g, e0, e1, n0, n1; };

template <bool LDS>
__global__ __launch_bounds__(256) void k_gather(const Item* __restrict__ items, const float* __restrict__ ef, const int* __restrict__ rowval, const float* __restrict__ nf,
                                                float* __restrict__ out) {
  __shared__ float s_tab[256 * 5];
  const Item it = items[blockIdx.x];
  if (LDS) {
    const int nfl = (it.n1 - it.n0) * 5;
    for (int i = threadIdx.x; i < nfl; i += 256) s_tab[i] = nf[(size_t)it.n0 * 5 + i];  // coalesced copy of the graph's node rows
  }
  // the edge rows and indices are requested before the barrier: the staging copy runs under them
  float acc[2][3];
  int src[2];
  F4u a[2], b[2]; F2u c[2];
  bool ok[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = it.e0 + threadIdx.x + 256 * u;
    ok[u] = e < it.e1;
    const int ec = ok[u] ? e : it.e1 - 1;
    const float* p = ef + (size_t)ec * 10;
    a[u] = *(const F4u*)p; b[u] = *(const F4u*)(p + 4); c[u] = *(const F2u*)(p + 8);
    src[u] = rowval[ec];
  }
  if (LDS) __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    float g0, g1, g2, g3, g4;
    if (LDS) { const float* q = s_tab + (src[u] - it.n0) * 5; g0 = q[0]; g1 = q[1]; g2 = q[2]; g3 = q[3]; g4 = q[4]; }
    else { const float* q = nf + (size_t)src[u] * 5; const F4u g = *(const F4u*)q; g0 = g.x; g1 = g.y; g2 = g.z; g3 = g.w; g4 = q[4]; }
    acc[u][0] = a[u].x + b[u].x + c[u].x + g0; acc[u][1] = a[u].y + b[u].y + c[u].y + g1 + g4; acc[u][2] = a[u].z + b[u].z + a[u].w + b[u].w + g2 + g3;
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (ok[u]) { F3u o; o.x = acc[u][0]; o.y = acc[u][1]; o.z = acc[u][2]; *(F3u*)(out + (size_t)(it.e0 + threadIdx.x + 256 * u) * 3) = o; }
}

int main() {
  for (int G : {512, 4096}) {
    std::mt19937 rng(G);
    std::vector<int> n(G);
    double sq = 0;
    for (auto& v : n) { v = 32 + rng() % 225; sq += (double)v * v; }
    const int E = 1000000;
    std::vector<int> eg(G), noff(G + 1, 0), eoff(G + 1, 0);
    int tot = 0;
    for (int g = 0; g < G; ++g) { eg[g] = (int)(E / sq * n[g] * n[g]); tot += eg[g]; }
    for (int g = 0; tot < E; g = (g + 1) % G) { ++eg[g]; ++tot; }
    for (int g = 0; g < G; ++g) { noff[g + 1] = noff[g] + n[g]; eoff[g + 1] = eoff[g] + eg[g]; }
    const int N = noff[G];
    std::vector<int> rowval(E);
    std::vector<Item> items;
    for (int g = 0; g < G; ++g) {
      for (int e = eoff[g]; e < eoff[g + 1]; ++e) rowval[e] = noff[g] + rng() % n[g];
      for (int c = eoff[g]; c < eoff[g + 1]; c += 512) items.push_back({g, c, std::min(c + 512, eoff[g + 1]), noff[g], noff[g + 1]});
    }
    constexpr int NS = 8, REP = 40;
    float *ef[NS], *nf[NS], *out[NS];
    int* d_rv; Item* d_items;
    CK(hipMalloc(&d_rv, E * 4)); CK(hipMemcpy(d_rv, rowval.data(), E * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_items, items.size() * sizeof(Item))); CK(hipMemcpy(d_items, items.data(), items.size() * sizeof(Item), hipMemcpyHostToDevice));
    for (int s = 0; s < NS; ++s) {
      CK(hipMalloc(&ef[s], (size_t)E * 40)); CK(hipMalloc(&nf[s], (size_t)N * 20)); CK(hipMalloc(&out[s], (size_t)E * 12));
      CK(hipMemset(ef[s], 0, (size_t)E * 40)); CK(hipMemset(nf[s], 0, (size_t)N * 20));
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int lds = 0; lds < 2; ++lds) {
      hipGraph_t graph; hipGraphExec_t exec;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int r = 0; r < REP; ++r) {
        const int s = r % NS;
        if (lds) hipLaunchKernelGGL(k_gather<true>, dim3((unsigned)items.size()), dim3(256), 0, st, d_items, ef[s], d_rv, nf[s], out[s]);
        else hipLaunchKernelGGL(k_gather<false>, dim3((unsigned)items.size()), dim3(256), 0, st, d_items, ef[s], d_rv, nf[s], out[s]);
      }
      CK(hipStreamEndCapture(st, &graph)); CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int w = 0; w < 60; ++w) CK(hipGraphLaunch(exec, st));  // ~100 ms: settled clocks
      CK(hipStreamSynchronize(st));
      float best = 1e9f;
      for (int t = 0; t < 5; ++t) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(exec, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
      }
      printf("G = %4d graphs (%d nodes, %d edges, %zu workgroups): %-34s %6.2f us per launch\n", G, N, E, items.size(),
             lds ? "B: node table staged in LDS" : "A: sources gathered from the L2", best * 1e3 / REP);
    }
  }
  return 0;
}

// AddressSanitizer build of libgnx's HOST shim (gnx_graphs.cpp, gnx_model.cpp, gnx_jit.cpp, gnx_profile.cpp), driven on the CPU
// ("build -fsanitize=address host shim").  The kernels live in .hip translation units that a host sanitizer cannot
// see; the handful of device-side entry points the shim calls are stubbed here to answer "no device" — what the real ones
// answer on a box without a GPU.  What runs under ASan: adjacency / CSC validation and conversion, tile-table construction,
// every error path that must free a half-built handle, model-descriptor validation, the run-time kernel specialiser's
// source handling (hiprtc compiles gfx950 code without a GPU) and the profiling registry.
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "gnx_internal.h"

extern "C" const char gnx_jit_source[] = R"(
namespace gnx { struct BlockArgs { int x; };
template <int DE, int DN, int DG, int OE, int ON, int EPT, bool LN, bool ONEG> __global__ void k_block_wave(BlockArgs a, int n) {}
template <int C, bool ONEG> __global__ void k_graph_t(BlockArgs a, int n) {} }
)";

namespace gnx {
void release_dense_csc(DenseCscOnDevice&) {}
int32_t build_csc_on_device(const void* const*, const void*, int, const int64_t*, int64_t, int32_t, int32_t, gnx::vec_i64&, gnx::vec_i64&,
                            const std::vector<int64_t>&, DenseCscOnDevice*) {
  return fail(100, "stub: no device");
}
void* arena_take(int, size_t, size_t*) { return nullptr; }
void arena_give(int, void* p, size_t) { if (p) (void)hipFree(p); }
int32_t build_wide_tables_on_device(const gnx_graphs*) { return 1; }
int32_t build_handle_from_csc_on_device(gnx_graphs*, const void*, const void*, int32_t, int32_t, int, int, int, int64_t, int64_t, int64_t) { return 1; }  // the host builder runs
}  // namespace gnx
extern "C" {
size_t gnx_block_workspace_bytes(const gnx_graphs*, const gnx_block_params*, int64_t) { return 256; }
size_t gnx_core_workspace_bytes(const gnx_graphs*, const gnx_core_params*, int64_t) { return 256; }
int32_t gnx_block_forward(const gnx_graphs*, const gnx_block_params*, const float*, const float*, const float*, int64_t, float*, float*, float*, void*,
                          size_t, uint32_t, void*) { return 100; }
int32_t gnx_core_forward(const gnx_graphs*, const gnx_core_params*, const float*, const float*, const float*, int64_t, float*, float*, float*, void*,
                         size_t, uint32_t, void*) { return 100; }
// prepared parameters (gnx_prepare.cpp launches kernels): the model asks for them at create — "nothing to prepare" here
int32_t gnx_block_prepare(const gnx_block_params*, void*, gnx_prepared** out) { *out = nullptr; return GNX_OK; }
int32_t gnx_core_prepare(const gnx_core_params*, void*, gnx_prepared** out) { *out = nullptr; return GNX_OK; }
int32_t gnx_prepared_refresh(gnx_prepared*, void*) { return GNX_OK; }
int32_t gnx_prepared_destroy(gnx_prepared*) { return GNX_OK; }
}

static int failures = 0;
#define EXPECT(cond)                                                      \
  do {                                                                    \
    if (!(cond)) { ++failures; fprintf(stderr, "FAILED %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #cond, gnx_last_error()); } \
  } while (0)

int main() {
  std::mt19937 rng(7);
  // ---- dense adjacency: validation, every element kind, row/column-major; creation then fails at the device (no GPU) and
  //      the half-built handle must be released without a leak ----
  for (int trial = 0; trial < 50; ++trial) {
    const int G = 1 + (int)(rng() % 6);
    std::vector<std::vector<double>> mats64(G);
    std::vector<std::vector<int32_t>> mats32(G);
    std::vector<const void*> p64(G), p32(G);
    std::vector<int64_t> nn(G);
    for (int g = 0; g < G; ++g) {
      const int n = 1 + (int)(rng() % 40);
      nn[g] = n;
      mats64[g].resize((size_t)n * n);
      mats32[g].resize((size_t)n * n);
      for (size_t i = 0; i < mats64[g].size(); ++i) { const int v = rng() % 4 == 0; mats64[g][i] = v; mats32[g][i] = v; }
      p64[g] = mats64[g].data(); p32[g] = mats32[g].data();
    }
    gnx_graphs* h = nullptr;
    int32_t rc = gnx_graphs_create_dense(p64.data(), nn.data(), G, GNX_ELEM_F64, trial & 1, &h);
    EXPECT(rc != GNX_OK || h != nullptr);  // no GPU here: a HIP error code (> 0) and no handle, or (with a GPU) a handle
    if (h) gnx_graphs_destroy(h);
    h = nullptr;
    rc = gnx_graphs_create_dense(p32.data(), nn.data(), G, GNX_ELEM_I32, trial & 1, &h);
    if (h) gnx_graphs_destroy(h);
    mats32[0][0] = 7;  // not 0/1
    h = nullptr;
    EXPECT(gnx_graphs_create_dense(p32.data(), nn.data(), G, GNX_ELEM_I32, 0, &h) == GNX_ERR_ADJ_VALUE && h == nullptr);
    // the packed form (one buffer, its byte length checked): the same scan over per-graph windows of the buffer, never past its end
    mats32[0][0] = 0;
    std::vector<int32_t> cat;
    for (int g = 0; g < G; ++g) cat.insert(cat.end(), mats32[g].begin(), mats32[g].end());
    std::vector<int32_t> exact(cat);  // (an exactly-sized heap block: a read past it is an ASan report)
    h = nullptr;
    rc = gnx_graphs_create_dense_packed(exact.data(), (int64_t)(exact.size() * 4), nn.data(), G, GNX_ELEM_I32, trial & 1, 0, &h);
    EXPECT(rc != GNX_OK || h != nullptr);
    if (h) gnx_graphs_destroy(h);
    h = nullptr;
    EXPECT(gnx_graphs_create_dense_packed(exact.data(), (int64_t)(exact.size() * 4) - 4, nn.data(), G, GNX_ELEM_I32, 0, 0, &h) == GNX_ERR_INVALID_ARG && h == nullptr);
    exact[exact.size() - 1] = 9;  // not 0/1, in the last entry of the last graph
    EXPECT(gnx_graphs_create_dense_packed(exact.data(), (int64_t)(exact.size() * 4), nn.data(), G, GNX_ELEM_I32, 0, 0, &h) == GNX_ERR_ADJ_VALUE && h == nullptr);
  }
  // ---- CSC input: well-formed (then the device step fails or succeeds), and every malformation ----
  for (int trial = 0; trial < 50; ++trial) {
    const int G = 1 + (int)(rng() % 5), base = trial & 1;
    std::vector<std::vector<int64_t>> cps(G), rvs(G);
    std::vector<const int64_t*> cpp(G), rvp(G);
    std::vector<int64_t> nn(G);
    for (int g = 0; g < G; ++g) {
      const int n = 1 + (int)(rng() % 300);
      nn[g] = n;
      cps[g].push_back(base);
      for (int j = 0; j < n; ++j) {
        int64_t deg = 0;
        for (int i = 0; i < n; ++i)
          if (rng() % 16 == 0) { rvs[g].push_back(i + base); ++deg; }
        cps[g].push_back(cps[g].back() + deg);
      }
      if (rvs[g].empty()) rvs[g].push_back(base);  // keeps .data() non-NULL; not referenced when there are no edges
      cpp[g] = cps[g].data(); rvp[g] = rvs[g].data();
    }
    gnx_graphs* h = nullptr;
    int32_t rc = gnx_graphs_create_csc(cpp.data(), rvp.data(), nn.data(), G, base, &h);
    EXPECT((rc == GNX_OK) == (h != nullptr));
    if (h) {
      gnx_graphs_info info;
      EXPECT(gnx_graphs_get_info(h, &info) == GNX_OK && info.n_graphs == G);
      std::vector<int64_t> no(G + 1), eo(G + 1);
      EXPECT(gnx_graphs_get_offsets(h, no.data(), eo.data()) == GNX_OK);
      gnx_graphs_destroy(h);
    }
    {  // the packed form (two arrays instead of 2 G pointers) answers exactly what the pointer form answers, malformed input included
      std::vector<int64_t> cpc, rvc;
      for (int g = 0; g < G; ++g) {  // (an edgeless graph's rvs holds one dummy entry: only the real edges are concatenated)
        cpc.insert(cpc.end(), cps[g].begin(), cps[g].end());
        rvc.insert(rvc.end(), rvs[g].begin(), rvs[g].begin() + (cps[g].back() - base));
      }
      gnx_graphs* hp = nullptr;
      const int32_t rcp = gnx_graphs_create_csc_packed(cpc.data(), rvc.empty() ? nullptr : rvc.data(), nn.data(), G, base, &hp);
      EXPECT(rcp == rc);
      if (rcp == GNX_OK) gnx_graphs_destroy(hp);
      if (!rvc.empty()) {
        rvc[0] = (int64_t)1 << 40;  // (out of range for whichever graph owns the first edge)
        hp = nullptr;
        EXPECT(gnx_graphs_create_csc_packed(cpc.data(), rvc.data(), nn.data(), G, base, &hp) == GNX_ERR_CSC && hp == nullptr);
      }
      EXPECT(gnx_graphs_create_csc_packed(nullptr, nullptr, nn.data(), G, base, &hp) == GNX_ERR_INVALID_ARG);
    }
    // malformed: colptr decreasing
    if (nn[0] >= 2) {
      auto bad = cps[0];
      bad[1] = bad[0] - 1;
      cpp[0] = bad.data();
      h = nullptr;
      EXPECT(gnx_graphs_create_csc(cpp.data(), rvp.data(), nn.data(), G, base, &h) == GNX_ERR_CSC && h == nullptr);
      cpp[0] = cps[0].data();
    }
    // malformed: source index out of range
    if (cps[0].back() > base) {
      auto bad = rvs[0];
      bad[0] = nn[0] + base + 5;
      rvp[0] = bad.data();
      h = nullptr;
      EXPECT(gnx_graphs_create_csc(cpp.data(), rvp.data(), nn.data(), G, base, &h) == GNX_ERR_CSC && h == nullptr);
      rvp[0] = rvs[0].data();
    }
  }
  // ---- malformed: a colptr that rises above colptr[n] and comes back (arrays are sized from colptr[n]; the old per-column
  //      check let the columns above it be written before the decrease was seen).  As the LAST graph, so the stray writes would fall
  //      behind the end of the allocation, with rowval sized generously so that only the library's own write can overflow ----
  for (int base = 0; base < 2; ++base) {
    std::vector<int64_t> cp0 = {0, 1, 2, 3}, rv0 = {0, 1, 2};            // a well-formed 3-node graph first
    std::vector<int64_t> cp1 = {0, 4, 8, 4, 4}, rv1(64);                  // n = 4: rises to 8, returns to cp[n] = 4
    for (size_t i = 0; i < rv1.size(); ++i) rv1[i] = (int64_t)(i % 4);
    for (auto* v : {&cp0, &rv0, &cp1, &rv1}) for (auto& x : *v) x += base;
    const int64_t nn[2] = {3, 4};
    const int64_t* cpp[2] = {cp0.data(), cp1.data()};
    const int64_t* rvp[2] = {rv0.data(), rv1.data()};
    gnx_graphs* h = nullptr;
    EXPECT(gnx_graphs_create_csc(cpp, rvp, nn, 2, base, &h) == GNX_ERR_CSC && h == nullptr);
    std::vector<int64_t> cpc(cp0), rvc(rv0);
    cpc.insert(cpc.end(), cp1.begin(), cp1.end());
    rvc.insert(rvc.end(), rv1.begin(), rv1.begin() + 4);                  // the packed form holds exactly cp[n] entries for the graph
    h = nullptr;
    EXPECT(gnx_graphs_create_csc_packed(cpc.data(), rvc.data(), nn, 2, base, &h) == GNX_ERR_CSC && h == nullptr);
    EXPECT(gnx_graphs_create_csc_cat(cpc.data(), (int64_t)cpc.size(), rvc.data(), (int64_t)rvc.size(), nn, 2, base, 64, &h) == GNX_ERR_CSC && h == nullptr);
    // the length-checked packed form refuses arrays shorter or longer than the graphs need, and takes 32-bit indices
    std::vector<int64_t> okc(cp0), okr(rv0);
    EXPECT(gnx_graphs_create_csc_cat(okc.data(), (int64_t)okc.size() - 1, okr.data(), (int64_t)okr.size(), nn, 1, base, 64, &h) == GNX_ERR_INVALID_ARG && h == nullptr);
    EXPECT(gnx_graphs_create_csc_cat(okc.data(), (int64_t)okc.size(), okr.data(), (int64_t)okr.size() - 1, nn, 1, base, 64, &h) == GNX_ERR_INVALID_ARG && h == nullptr);
    EXPECT(gnx_graphs_create_csc_cat(okc.data(), (int64_t)okc.size(), okr.data(), (int64_t)okr.size(), nn, 1, base, 16, &h) == GNX_ERR_INVALID_ARG && h == nullptr);
    {
      std::vector<int32_t> c32(cpc.begin(), cpc.end()), r32(rvc.begin(), rvc.end());  // the malformed pair again, as 32-bit indices
      EXPECT(gnx_graphs_create_csc_cat(c32.data(), (int64_t)c32.size(), r32.data(), (int64_t)r32.size(), nn, 2, base, 32, &h) == GNX_ERR_CSC && h == nullptr);
      std::vector<int32_t> oc32(okc.begin(), okc.end()), or32(okr.begin(), okr.end());
      const int32_t rc32 = gnx_graphs_create_csc_cat(oc32.data(), (int64_t)oc32.size(), or32.data(), (int64_t)or32.size(), nn, 1, base, 32, &h);
      EXPECT((rc32 == GNX_OK) == (h != nullptr) && rc32 >= 0);  // well-formed: a handle, or (no GPU) a HIP error code
      if (h) gnx_graphs_destroy(h);
      h = nullptr;
    }
  }
  {
    gnx_graphs* h = nullptr;
    EXPECT(gnx_graphs_create_csc(nullptr, nullptr, nullptr, 1, 0, &h) == GNX_ERR_INVALID_ARG);
    EXPECT(gnx_graphs_create_csc(nullptr, nullptr, nullptr, 0, 0, &h) == GNX_ERR_NO_GRAPHS);
    EXPECT(gnx_graphs_destroy(nullptr) == GNX_OK);
    gnx_graphs_info info;
    EXPECT(gnx_graphs_get_info(nullptr, &info) != GNX_OK);
  }
  // ---- multi-GPU split, host side: partition of 4096 graphs over 8 ranks, gather plans of EQUAL and UNEQUAL shards (the first real
  //      8-GPU run must not be the first execution of this index arithmetic), rejection of broken partitions ----
  {
    const int64_t G = 4096; const int R = 8;
    std::vector<int64_t> ecount(G), off(R + 1), ids(G);
    for (auto& e : ecount) e = 1 + (int64_t)(rng() % 5000);
    EXPECT(gnx_dist_partition(ecount.data(), G, R, off.data(), ids.data()) == GNX_OK);
    EXPECT(off[0] == 0 && off[R] == G);
    std::vector<int32_t> src(G);
    int64_t mc = 0;
    EXPECT(gnx_dist_gather_plan(off.data(), ids.data(), R, G, src.data(), &mc) == GNX_OK && mc == G / R);
    for (int r = 0; r < R; ++r)
      for (int64_t i = off[r]; i < off[r + 1]; ++i) EXPECT(src[(size_t)ids[i]] == (int32_t)(r * mc + (i - off[r])));
    // 8 unequal shards (sizes 1 .. 2000, one of them EMPTY) of a shuffled permutation
    const int64_t sizes[8] = {1, 700, 0, 2000, 33, 512, 849, 1};
    std::vector<int64_t> uoff(9, 0), perm(G);
    for (int r = 0; r < 8; ++r) uoff[r + 1] = uoff[r] + sizes[r];
    EXPECT(uoff[8] == G);
    for (int64_t i = 0; i < G; ++i) perm[i] = i;
    for (int64_t i = G - 1; i > 0; --i) std::swap(perm[i], perm[(size_t)(rng() % (uint64_t)(i + 1))]);
    EXPECT(gnx_dist_gather_plan(uoff.data(), perm.data(), 8, G, src.data(), &mc) == GNX_OK && mc == 2000);
    std::vector<char> hit((size_t)(8 * mc), 0);
    for (int r = 0; r < 8; ++r)
      for (int64_t i = uoff[r]; i < uoff[r + 1]; ++i) {
        const int32_t row = src[(size_t)perm[i]];
        EXPECT(row == (int32_t)(r * mc + (i - uoff[r])) && !hit[(size_t)row]);
        hit[(size_t)row] = 1;
      }
    EXPECT(gnx_dist_gather_plan(uoff.data(), perm.data(), 8, G, nullptr, nullptr) == GNX_OK);  // validation only
    perm[5] = perm[6];                                                                         // not a permutation any more
    EXPECT(gnx_dist_gather_plan(uoff.data(), perm.data(), 8, G, src.data(), &mc) == GNX_ERR_INVALID_ARG);
    uoff[3] = uoff[2] - 1;                                                                     // decreasing offsets
    EXPECT(gnx_dist_gather_plan(uoff.data(), ids.data(), 8, G, src.data(), &mc) == GNX_ERR_INVALID_ARG);
    EXPECT(gnx_dist_gather_plan(nullptr, ids.data(), 8, G, src.data(), &mc) == GNX_ERR_INVALID_ARG);
    EXPECT(gnx_dist_partition(ecount.data(), G, 0, off.data(), ids.data()) == GNX_ERR_INVALID_ARG);
  }
  // ---- model descriptors ----
  {
    gnx_model* m = nullptr;
    gnx_layer l{};
    EXPECT(gnx_model_create(nullptr, &l, 1, 1, &m) != GNX_OK && m == nullptr);
    EXPECT(gnx_model_destroy(nullptr) == GNX_OK);
    int32_t dims[3];
    EXPECT(gnx_model_out_dims(nullptr, dims) != GNX_OK);
  }
  // ---- run-time specialiser: compile-only entry (no GPU needed); the stub source above stands in for the kernel text ----
  {
    gnx_block_params p{};
    p.de = 7; p.dn = 3; p.dg = 2; p.oe = 5; p.on = 6; p.og = 1;
    size_t bytes = 0;
    const int32_t rc = gnx_jit_precompile(&p, 128, &bytes);
    EXPECT(rc == GNX_OK ? bytes > 0 : gnx_last_error()[0] != 0);  // hiprtc present: a code object; absent: a message
    EXPECT(gnx_jit_precompile(&p, 100, &bytes) == GNX_ERR_INVALID_ARG);
    p.oe = 500;
    EXPECT(gnx_jit_precompile(&p, 128, &bytes) == GNX_ERR_DIMS);
    int64_t st[4];
    EXPECT(gnx_jit_stats(st) == GNX_OK);
  }
  // ---- profiling registry ----
  {
    EXPECT(gnx_profile_enable(1) == GNX_OK);
    EXPECT(gnx_profile_reset() == GNX_OK);
    gnx_profile_entry e[4];
    int32_t n = -1;
    EXPECT(gnx_profile_read(e, 4, &n) == GNX_OK && n == 0);
    EXPECT(gnx_profile_enable(0) == GNX_OK);
  }
  if (failures) { fprintf(stderr, "%d expectation(s) failed\n", failures); return 1; }
  printf("host_asan_driver: ok\n");
  return 0;
}

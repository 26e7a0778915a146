/* abi_smoke.c — README example 1 of GraphNets.jl through include/gnx.h from plain C, exactly as a Julia `ccall` (or any C
 * host) would drive libgnx.so: struct layout by the C compiler (not ctypes), host adjacency in, device feature buffers,
 * one gnx_block_forward, results checked against a double-precision evaluation of the same formulas written inline
 * (/root/reference/README.md:28-60: adj = [1 0 1; 1 1 0; 0 0 1], batch_size 2, (10,5,0) => (3,4,5)).
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/c/abi_smoke.c -o abi_smoke \
 *       -L graphnets.jl_amd -lgnx -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,...
 *   ./abi_smoke            full run (needs a GPU); exit 0 = pass
 *   ./abi_smoke --symbols  no GPU work: the library loads, gnx_version() answers, argument validation answers
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gnx.h"

#define CHECK_GNX(expr)                                                                  \
  do {                                                                                   \
    int32_t rc_ = (expr);                                                                \
    if (rc_ != GNX_OK) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, gnx_last_error()); return 1; } \
  } while (0)
#define CHECK_HIP(expr)                                                                  \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #expr, hipGetErrorString(e_)); return 1; } \
  } while (0)

enum { N = 3, E = 5, B = 2, DE = 10, DN = 5, OE = 3, ON = 4, OG = 5, KE = DE + 2 * DN, KN = OE + DN, KG = OE + ON };

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (float)(*s >> 8) / 16777216.0f - 0.5f; }

static float* to_device(const float* h, size_t n) {
  float* d = NULL;
  if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return NULL;
  if (hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

int main(int argc, char** argv) {
  /* layout facts a binding relies on */
  if (sizeof(gnx_dense) != 24 || sizeof(gnx_block_params) != 104 || sizeof(gnx_core_params) != 360 || sizeof(gnx_graphs_info) != 64 || sizeof(gnx_profile_entry) != 72) {
    fprintf(stderr, "struct layout differs from the documented one\n");
    return 1;
  }
  if (gnx_version() != GNX_VERSION) { fprintf(stderr, "version mismatch\n"); return 1; }
  /* validation happens before any GPU work: bad element (2) in the adjacency -> GNX_ERR_ADJ_VALUE */
  {
    const int64_t bad[4] = {1, 2, 0, 1}, n2 = 2;
    const void* ptrs[1] = {bad};
    gnx_graphs* h = NULL;
    if (gnx_graphs_create_dense(ptrs, &n2, 1, GNX_ELEM_I64, 0, &h) != GNX_ERR_ADJ_VALUE || h != NULL) {
      fprintf(stderr, "expected GNX_ERR_ADJ_VALUE\n");
      return 1;
    }
    if (gnx_graphs_create_dense(ptrs, &n2, 0, GNX_ELEM_I64, 0, &h) != GNX_ERR_NO_GRAPHS) { fprintf(stderr, "expected GNX_ERR_NO_GRAPHS\n"); return 1; }
  }
  if (argc > 1 && strcmp(argv[1], "--symbols") == 0) { printf("abi_smoke: symbols ok (version %d)\n", gnx_version()); return 0; }

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fprintf(stderr, "no GPU visible\n"); return 2; }

  /* Julia column-major adj_mat = [1 0 1; 1 1 0; 0 0 1]: A[i,j] = 1 <=> edge i -> j; column-major storage, row_major = 0 */
  const float adj[N * N] = {1, 1, 0, /* column 1 */ 0, 1, 0, /* column 2 */ 1, 0, 1 /* column 3 */};
  const void* adj_ptrs[1] = {adj};
  const int64_t nn = N;
  gnx_graphs* h = NULL;
  CHECK_GNX(gnx_graphs_create_dense(adj_ptrs, &nn, 1, GNX_ELEM_F32, 0, &h));
  gnx_graphs_info info;
  CHECK_GNX(gnx_graphs_get_info(h, &info));
  if (info.n_graphs != 1 || info.n_nodes != N || info.n_edges != E || info.node_block_size != N || info.edge_block_size != N * N) {
    fprintf(stderr, "unexpected handle info\n");
    return 1;
  }
  int64_t colptr[N + 1], rowval[E];
  CHECK_GNX(gnx_graphs_get_csc(h, colptr, rowval));
  /* edge order = ones of vec(A) column-major (src/pad.jl:30): (1->1), (2->1), (2->2), (1->3), (3->3), 0-based below */
  const int64_t want_src[E] = {0, 1, 1, 0, 2}, want_cp[N + 1] = {0, 2, 3, 5};
  if (memcmp(rowval, want_src, sizeof want_src) || memcmp(colptr, want_cp, sizeof want_cp)) { fprintf(stderr, "edge order differs from the reference's\n"); return 1; }

  /* features: Julia (D, T, B) column-major = packed [B][T][D]; weights (out x in) column-major */
  unsigned seed = 12345u;
  float ef[B * E * DE], nf[B * N * DN], We[KE * OE], be[OE], Wn[KN * ON], bn[ON], Wg[KG * OG], bg[OG];
  for (size_t i = 0; i < sizeof ef / sizeof *ef; ++i) ef[i] = frand(&seed) + 0.5f;
  for (size_t i = 0; i < sizeof nf / sizeof *nf; ++i) nf[i] = frand(&seed) + 0.5f;
  for (size_t i = 0; i < sizeof We / sizeof *We; ++i) We[i] = frand(&seed);
  for (size_t i = 0; i < sizeof Wn / sizeof *Wn; ++i) Wn[i] = frand(&seed);
  for (size_t i = 0; i < sizeof Wg / sizeof *Wg; ++i) Wg[i] = frand(&seed);
  for (int j = 0; j < OE; ++j) be[j] = 0.2f * frand(&seed);
  for (int j = 0; j < ON; ++j) bn[j] = 0.2f * frand(&seed);
  for (int j = 0; j < OG; ++j) bg[j] = 0.2f * frand(&seed);

  float *d_ef = to_device(ef, sizeof ef / 4), *d_nf = to_device(nf, sizeof nf / 4), *d_We = to_device(We, sizeof We / 4), *d_be = to_device(be, OE),
        *d_Wn = to_device(Wn, sizeof Wn / 4), *d_bn = to_device(bn, ON), *d_Wg = to_device(Wg, sizeof Wg / 4), *d_bg = to_device(bg, OG);
  float *d_eo = NULL, *d_no = NULL, *d_go = NULL;
  void* ws = NULL;
  if (!d_ef || !d_nf || !d_We || !d_be || !d_Wn || !d_bn || !d_Wg || !d_bg) { fprintf(stderr, "device allocation failed\n"); return 1; }
  CHECK_HIP(hipMalloc((void**)&d_eo, sizeof(float) * B * E * OE));
  CHECK_HIP(hipMalloc((void**)&d_no, sizeof(float) * B * N * ON));
  CHECK_HIP(hipMalloc((void**)&d_go, sizeof(float) * B * 1 * OG));

  gnx_block_params p;
  memset(&p, 0, sizeof p);
  p.de = DE; p.dn = DN; p.dg = 0; p.oe = OE; p.on = ON; p.og = OG;
  p.edgefn.weight = d_We; p.edgefn.bias = d_be; p.edgefn.act = GNX_ACT_IDENTITY;
  p.nodefn.weight = d_Wn; p.nodefn.bias = d_bn; p.nodefn.act = GNX_ACT_RELU;
  p.graphfn.weight = d_Wg; p.graphfn.bias = d_bg; p.graphfn.act = GNX_ACT_TANH;
  const size_t ws_bytes = gnx_block_workspace_bytes(h, &p, B);
  if (ws_bytes == 0) { fprintf(stderr, "workspace size 0: %s\n", gnx_last_error()); return 1; }
  CHECK_HIP(hipMalloc(&ws, ws_bytes));
  /* gf = nothing (NULL, width 0); batch_size = n_replicas = 2; default stream */
  CHECK_GNX(gnx_block_forward(h, &p, d_ef, d_nf, NULL, B, d_eo, d_no, d_go, ws, ws_bytes, 0, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  float eo[B * E * OE], no[B * N * ON], go[B * OG];
  CHECK_HIP(hipMemcpy(eo, d_eo, sizeof eo, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(no, d_no, sizeof no, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(go, d_go, sizeof go, hipMemcpyDeviceToHost));

  /* the same block in double precision: edgefninput.jl:1-8, nodefninput.jl:1-7, graphfninput.jl:1-7, gnblock.jl:63-69 */
  const int dst_of[E] = {0, 0, 1, 2, 2};
  double worst = 0.0;
  for (int b = 0; b < B; ++b) {
    double he[E][OE], agg[N][OE], hn[N][ON], se[OE] = {0}, sn[ON] = {0};
    memset(agg, 0, sizeof agg);
    for (int e = 0; e < E; ++e) {
      double x[KE];
      for (int k = 0; k < DE; ++k) x[k] = ef[(b * E + e) * DE + k];
      for (int k = 0; k < DN; ++k) x[DE + k] = nf[(b * N + want_src[e]) * DN + k];
      for (int k = 0; k < DN; ++k) x[DE + DN + k] = nf[(b * N + dst_of[e]) * DN + k];
      for (int j = 0; j < OE; ++j) {
        double y = be[j];
        for (int k = 0; k < KE; ++k) y += (double)We[k * OE + j] * x[k];
        he[e][j] = y;
        agg[dst_of[e]][j] += y;
        se[j] += y;
        worst = fmax(worst, fabs(y - eo[(b * E + e) * OE + j]));
      }
    }
    for (int n = 0; n < N; ++n)
      for (int j = 0; j < ON; ++j) {
        double y = bn[j];
        for (int k = 0; k < OE; ++k) y += (double)Wn[k * ON + j] * agg[n][k];
        for (int k = 0; k < DN; ++k) y += (double)Wn[(OE + k) * ON + j] * nf[(b * N + n) * DN + k];
        hn[n][j] = y > 0 ? y : 0;
        sn[j] += hn[n][j];
        worst = fmax(worst, fabs(hn[n][j] - no[(b * N + n) * ON + j]));
      }
    for (int j = 0; j < OG; ++j) {
      double y = bg[j];
      for (int k = 0; k < OE; ++k) y += (double)Wg[k * OG + j] * se[k];
      for (int k = 0; k < ON; ++k) y += (double)Wg[(OE + k) * OG + j] * sn[k];
      worst = fmax(worst, fabs(tanh(y) - go[b * OG + j]));
    }
    (void)he;
  }
  printf("abi_smoke: README example 1 through the C ABI, max |hip - double| = %.3e\n", worst);
  if (!(worst <= 2e-5)) { fprintf(stderr, "mismatch\n"); return 1; }

  CHECK_GNX(gnx_graphs_destroy(h));
  hipFree(ws); hipFree(d_eo); hipFree(d_no); hipFree(d_go); hipFree(d_ef); hipFree(d_nf);
  hipFree(d_We); hipFree(d_be); hipFree(d_Wn); hipFree(d_bn); hipFree(d_Wg); hipFree(d_bg);
  return 0;
}

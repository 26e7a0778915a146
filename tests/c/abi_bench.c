/* abi_bench.c — throughput of the hot path through include/gnx.h from plain C: no Python, no PyTorch.  What a Julia `ccall` host (or any
 * C host) gets from libgnx.so when it drives the boundary the way bench.py drives it through torch (VERDICT r3 #3 / "next" 1b).
 *
 *   abi_bench --mode block   BASELINE configs[1] (C2: one Erdos-Renyi graph, 100k nodes / 1M edges) at README widths (10,5,0)=>(3,4,5):
 *                            (a) K gnx_block_forward calls captured by THIS program into one hipGraph on its own stream, rotating over
 *                                8 disjoint buffer sets (cache-cold) — bench.py's headline procedure, line for line, without torch;
 *                            (b) the library-owned replay: a one-layer gnx_model per buffer set, one gnx_model_forward (= one
 *                                hipGraphLaunch) per step.
 *   abi_bench --mode c4      BASELINE configs[3]: encoder (10,5,0)=>(128,64,32), 2 x GNCore(128,64,32), decoder =>(3,4,5) as ONE gnx_model
 *                            (library-owned intermediates and hipGraph), K replays.  --core-dims 10,5,3 = README ex.3's own widths.
 *   --csc FILE               the graph as int64 {N, E, colptr[N+1], rowval[E]} (0-based; bench.py writes the exact C2 graph of its own
 *                            line here); without it the program draws its own graph of the same law (E distinct directed pairs of an
 *                            N-node graph, uniformly, reference edge order).
 *   --steps K --warmup W --nodes N --edges E
 * Prints ONE JSON line.  Timing: after W warm-up steps and 150 ms of the same load (clock settling), wall clock around the K steps between two
 * hipStreamSynchronize (median of 3 regions), and HIP events
 * on the same stream.  Features U[0,1), weights glorot-uniform, biases 0, LayerNorm 1 / 0 (Flux's initialisation): as bench.py.
 */
#define _POSIX_C_SOURCE 200809L
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gnx.h"

#define CHECK_GNX(expr)                                                                                          \
  do {                                                                                                           \
    int32_t rc_ = (expr);                                                                                        \
    if (rc_ != GNX_OK) { fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #expr, rc_, gnx_last_error()); exit(1); } \
  } while (0)
#define CHECK_HIP(expr)                                                                                          \
  do {                                                                                                           \
    hipError_t e_ = (expr);                                                                                      \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); exit(1); } \
  } while (0)

enum { NSETS = 8 };

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static uint64_t rnd64(void) { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17; return g_rng; }
static float rnd01(void) { return (float)(rnd64() >> 40) / 16777216.0f; }
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static int cmp_u64(const void* a, const void* b) { const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }
static int cmp_dbl(const void* a, const void* b) { const double x = *(const double*)a, y = *(const double*)b; return x < y ? -1 : x > y; }

static float* dev_uniform(size_t n, float lo, float hi) { /* device array of n floats, U[lo, hi) */
  float* h = (float*)malloc((n ? n : 1) * sizeof(float));
  float* d = NULL;
  for (size_t i = 0; i < n; ++i) h[i] = lo + (hi - lo) * rnd01();
  CHECK_HIP(hipMalloc((void**)&d, (n ? n : 1) * sizeof(float)));
  if (n) CHECK_HIP(hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice));
  free(h);
  return d;
}
static float* dev_const(size_t n, float v) { return dev_uniform(n, v, v); }
static gnx_dense dense(int out, int in, int act) { /* Flux Dense(in => out): glorot-uniform weight, zero bias */
  gnx_dense d;
  const float s = sqrtf(6.0f / (float)((in + out) > 0 ? in + out : 1));
  d.weight = dev_uniform((size_t)out * in, -s, s);
  d.bias = dev_const((size_t)out, 0.f);
  d.act = act;
  d.kind = 0;
  return d;
}
static gnx_block_params block(const int in[3], const int out[3]) { /* GNBlock(in => out), gnblock.jl:47-61 */
  gnx_block_params p;
  memset(&p, 0, sizeof p);
  p.de = in[0]; p.dn = in[1]; p.dg = in[2]; p.oe = out[0]; p.on = out[1]; p.og = out[2];
  p.edgefn = dense(out[0], in[0] + 2 * in[1] + in[2], GNX_ACT_IDENTITY);
  p.nodefn = dense(out[1], out[0] + in[1] + in[2], GNX_ACT_IDENTITY);
  p.graphfn = dense(out[2], out[0] + out[1] + in[2], GNX_ACT_IDENTITY);
  return p;
}
static gnx_core_params core(const int d[3]) { /* GNCore(dims), gncore.jl:46-54 */
  gnx_core_params c;
  memset(&c, 0, sizeof c);
  c.block = block(d, d);
  for (int t = 0; t < 3; ++t) {
    c.ln1[t].gamma = dev_const((size_t)d[t], 1.f); c.ln1[t].beta = dev_const((size_t)d[t], 0.f);
    c.ln2[t].gamma = dev_const((size_t)d[t], 1.f); c.ln2[t].beta = dev_const((size_t)d[t], 0.f);
    c.ff[t].fc1 = dense(4 * d[t], d[t], GNX_ACT_RELU);
    c.ff[t].fc2 = dense(d[t], 4 * d[t], GNX_ACT_IDENTITY);
  }
  c.eps = 1e-5f;
  c.eps_mode = 0;
  return c;
}

/* E distinct directed pairs of an N-node graph, uniformly; reference edge order = sorted by (dst, src) = sorted key dst * N + src */
static void draw_graph(int64_t N, int64_t E, int64_t** colptr_out, int64_t** rowval_out) {
  const size_t cap = (size_t)E + (size_t)E / 8 + 1024;
  uint64_t* k = (uint64_t*)malloc(cap * sizeof(uint64_t));
  size_t n = 0;
  while (n < (size_t)E) {
    for (size_t i = n; i < cap; ++i) k[i] = rnd64() % ((uint64_t)N * (uint64_t)N);
    qsort(k, cap, sizeof(uint64_t), cmp_u64);
    n = 0;
    for (size_t i = 0; i < cap; ++i)
      if (i == 0 || k[i] != k[i - 1]) k[n++] = k[i];
  }
  for (size_t i = n - 1; i > 0; --i) { const size_t j = (size_t)(rnd64() % (i + 1)); const uint64_t t = k[i]; k[i] = k[j]; k[j] = t; } /* keep a uniform subset of E */
  qsort(k, (size_t)E, sizeof(uint64_t), cmp_u64);
  int64_t* cp = (int64_t*)calloc((size_t)N + 1, sizeof(int64_t));
  int64_t* rv = (int64_t*)malloc((size_t)E * sizeof(int64_t));
  for (int64_t e = 0; e < E; ++e) { cp[k[e] / (uint64_t)N + 1]++; rv[e] = (int64_t)(k[e] % (uint64_t)N); }
  for (int64_t j = 0; j < N; ++j) cp[j + 1] += cp[j];
  free(k);
  *colptr_out = cp; *rowval_out = rv;
}

static void read_graph(const char* path, int64_t* N, int64_t* E, int64_t** colptr_out, int64_t** rowval_out) {
  FILE* f = fopen(path, "rb");
  int64_t hdr[2];
  if (!f || fread(hdr, sizeof(int64_t), 2, f) != 2 || hdr[0] <= 0 || hdr[1] < 0) { fprintf(stderr, "cannot read %s\n", path); exit(1); }
  *N = hdr[0]; *E = hdr[1];
  int64_t* cp = (int64_t*)malloc(((size_t)*N + 1) * sizeof(int64_t));
  int64_t* rv = (int64_t*)malloc(((size_t)*E + 1) * sizeof(int64_t));
  if (fread(cp, sizeof(int64_t), (size_t)*N + 1, f) != (size_t)*N + 1 || fread(rv, sizeof(int64_t), (size_t)*E, f) != (size_t)*E) { fprintf(stderr, "%s is truncated\n", path); exit(1); }
  fclose(f);
  *colptr_out = cp; *rowval_out = rv;
}

typedef struct { double wall_us, event_us, reps_us[3]; } timing;

/* W warm-up calls, then 3 regions of K calls of `step(i, ctx)` bracketed by stream synchronisation: median wall time per step, and the
 * HIP-event time of the median region */
static timing time_steps(void (*step)(int, void*), void* ctx, int K, int W, hipStream_t s) {
  timing t;
  hipEvent_t e0, e1;
  double ev[3];
  CHECK_HIP(hipEventCreate(&e0)); CHECK_HIP(hipEventCreate(&e1));
  for (int i = 0; i < W; ++i) step(i, ctx);
  CHECK_HIP(hipStreamSynchronize(s));
  /* untimed: the same load for 150 ms, so that the timed regions run at settled clocks (after an idle period the MI355X's power management
   * needs tens of milliseconds of continuous load: bench.py::spin_up has the trace) */
  for (const double t_w = now_s(); now_s() - t_w < 0.150;) {
    for (int i = 0; i < (K < 64 ? K : 64); ++i) step(i, ctx);
    CHECK_HIP(hipStreamSynchronize(s));
  }
  for (int r = 0; r < 3; ++r) {
    CHECK_HIP(hipStreamSynchronize(s));
    const double t0 = now_s();
    CHECK_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < K; ++i) step(i, ctx);
    CHECK_HIP(hipEventRecord(e1, s));
    CHECK_HIP(hipStreamSynchronize(s));
    t.reps_us[r] = (now_s() - t0) * 1e6 / K;
    float ms = 0.f;
    CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    ev[r] = (double)ms * 1e3 / K;
  }
  double w[3] = {t.reps_us[0], t.reps_us[1], t.reps_us[2]};
  qsort(w, 3, sizeof(double), cmp_dbl);
  qsort(ev, 3, sizeof(double), cmp_dbl);
  t.wall_us = w[1]; t.event_us = ev[1];
  CHECK_HIP(hipEventDestroy(e0)); CHECK_HIP(hipEventDestroy(e1));
  return t;
}

typedef struct { hipGraphExec_t exec; hipStream_t s; } replay_ctx;
static void step_replay(int i, void* c) { (void)i; replay_ctx* r = (replay_ctx*)c; CHECK_HIP(hipGraphLaunch(r->exec, r->s)); }

typedef struct { const gnx_graphs* h; const gnx_block_params* p; const float *ef[NSETS], *nf[NSETS]; float *eo[NSETS], *no[NSETS], *go[NSETS]; void* ws[NSETS]; size_t ws_bytes; hipStream_t s; } eager_ctx;
static void step_eager(int i, void* c) {
  eager_ctx* x = (eager_ctx*)c;
  const int b = i % NSETS;
  CHECK_GNX(gnx_block_forward(x->h, x->p, x->ef[b], x->nf[b], NULL, 1, x->eo[b], x->no[b], x->go[b], x->ws[b], x->ws_bytes, 0, x->s));
}

typedef struct { gnx_model* m[NSETS]; int nsets; const float *ef[NSETS], *nf[NSETS]; float *eo[NSETS], *no[NSETS], *go[NSETS]; hipStream_t s; } model_ctx;
static void step_model(int i, void* c) {
  model_ctx* x = (model_ctx*)c;
  const int b = i % x->nsets;
  CHECK_GNX(gnx_model_forward(x->m[b], x->ef[b], x->nf[b], NULL, x->eo[b], x->no[b], x->go[b], 0, x->s));
}

int main(int argc, char** argv) {
  const char *mode = "block", *csc = NULL;
  int K = 200, W = 20;
  int64_t N = 100000, E = 1000000;
  int cd[3] = {128, 64, 32};
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--mode") && i + 1 < argc) mode = argv[++i];
    else if (!strcmp(argv[i], "--csc") && i + 1 < argc) csc = argv[++i];
    else if (!strcmp(argv[i], "--steps") && i + 1 < argc) K = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) W = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--nodes") && i + 1 < argc) N = atoll(argv[++i]);
    else if (!strcmp(argv[i], "--edges") && i + 1 < argc) E = atoll(argv[++i]);
    else if (!strcmp(argv[i], "--core-dims") && i + 1 < argc) { if (sscanf(argv[++i], "%d,%d,%d", &cd[0], &cd[1], &cd[2]) != 3) { fprintf(stderr, "--core-dims a,b,c\n"); return 2; } }
    else { fprintf(stderr, "usage: abi_bench [--mode block|c4] [--csc FILE] [--steps K] [--warmup W] [--nodes N] [--edges E] [--core-dims a,b,c]\n"); return 2; }
  }
  if (K < 1 || W < 0 || N < 1 || E < 0 || E > N * N) { fprintf(stderr, "bad sizes\n"); return 2; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fprintf(stderr, "no GPU visible\n"); return 3; }
  CHECK_HIP(hipSetDevice(0));

  int64_t *colptr = NULL, *rowval = NULL;
  if (csc) read_graph(csc, &N, &E, &colptr, &rowval); else draw_graph(N, E, &colptr, &rowval);
  gnx_graphs* h = NULL;
  double t_batch[3];
  for (int r = 0; r < 3; ++r) { /* GNGraphBatch construction through the boundary: best of 3 */
    if (h) CHECK_GNX(gnx_graphs_destroy(h));
    const double t0 = now_s();
    CHECK_GNX(gnx_graphs_create_csc_cat(colptr, N + 1, rowval, E, &N, 1, 0, 64, &h));
    t_batch[r] = (now_s() - t0) * 1e3;
  }
  qsort(t_batch, 3, sizeof(double), cmp_dbl);
  hipStream_t s;
  CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int in0[3] = {10, 5, 0}, out0[3] = {3, 4, 5};

  if (!strcmp(mode, "block")) {
    gnx_block_params p = block(in0, out0);
    const size_t ws_bytes = gnx_block_workspace_bytes(h, &p, 1);
    if (!ws_bytes) { fprintf(stderr, "workspace query failed: %s\n", gnx_last_error()); return 1; }
    const float *ef[NSETS], *nf[NSETS];
    float *eo[NSETS], *no[NSETS], *go[NSETS];
    void* ws[NSETS];
    for (int b = 0; b < NSETS; ++b) {
      ef[b] = dev_uniform((size_t)E * 10, 0.f, 1.f); nf[b] = dev_uniform((size_t)N * 5, 0.f, 1.f);
      CHECK_HIP(hipMalloc((void**)&eo[b], sizeof(float) * (size_t)E * 3)); CHECK_HIP(hipMalloc((void**)&no[b], sizeof(float) * (size_t)N * 4));
      CHECK_HIP(hipMalloc((void**)&go[b], sizeof(float) * 5)); CHECK_HIP(hipMalloc(&ws[b], ws_bytes));
    }
    for (int i = 0; i < 2; ++i) /* eager steps before any capture: code objects loaded, argument errors surface here */
      for (int b = 0; b < NSETS; ++b) CHECK_GNX(gnx_block_forward(h, &p, ef[b], nf[b], NULL, 1, eo[b], no[b], go[b], ws[b], ws_bytes, 0, s));
    CHECK_HIP(hipStreamSynchronize(s));
    /* (a) the caller's own capture: K forwards over the rotating sets as ONE hipGraph; a "step" of the timing loop is one replay = K forwards */
    hipGraph_t graph;
    replay_ctx rc;
    rc.s = s;
    CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < K; ++i) { const int b = i % NSETS; CHECK_GNX(gnx_block_forward(h, &p, ef[b], nf[b], NULL, 1, eo[b], no[b], go[b], ws[b], ws_bytes, 0, s)); }
    CHECK_HIP(hipStreamEndCapture(s, &graph));
    CHECK_HIP(hipGraphInstantiate(&rc.exec, graph, NULL, NULL, 0));
    const timing ta = time_steps(step_replay, &rc, 1, 1, s);
    /* (a2) the same K steps as ONE gnx_block_forward_steps call (round 6: the library's loop over batches — it chains the steps itself: one
     * launch per step + one flush inside the call), captured the same way: what bench.py's headline times through the Python mirror */
    hipGraph_t graph2;
    replay_ctx rc2;
    rc2.s = s;
    gnx_block_step* st = (gnx_block_step*)calloc((size_t)K, sizeof *st);
    for (int i = 0; i < K; ++i) {
      const int b = i % NSETS;
      st[i].ef = ef[b]; st[i].nf = nf[b]; st[i].gf = NULL; st[i].ef_out = eo[b]; st[i].nf_out = no[b]; st[i].gf_out = go[b]; st[i].workspace = ws[b]; st[i].workspace_bytes = ws_bytes;
    }
    CHECK_GNX(gnx_block_forward_steps(h, &p, st, K, 1, 0, s));  /* (eager once: argument errors surface here) */
    CHECK_HIP(hipStreamSynchronize(s));
    CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    CHECK_GNX(gnx_block_forward_steps(h, &p, st, K, 1, 0, s));
    CHECK_HIP(hipStreamEndCapture(s, &graph2));
    CHECK_HIP(hipGraphInstantiate(&rc2.exec, graph2, NULL, NULL, 0));
    const timing ta2 = time_steps(step_replay, &rc2, 1, 1, s);
    free(st);
    /* (b) library-owned replay: one one-layer model per buffer set, one gnx_model_forward per step */
    model_ctx mc;
    mc.nsets = NSETS; mc.s = s;
    const gnx_layer layer = {GNX_LAYER_BLOCK, 0, &p};
    for (int b = 0; b < NSETS; ++b) {
      CHECK_GNX(gnx_model_create(h, &layer, 1, 1, &mc.m[b]));
      mc.ef[b] = ef[b]; mc.nf[b] = nf[b]; mc.eo[b] = eo[b]; mc.no[b] = no[b]; mc.go[b] = go[b];
    }
    const timing tb = time_steps(step_model, &mc, K, W > NSETS ? W : NSETS, s);
    /* (c) no graph at all: one gnx_block_forward (two kernel launches) per step, issued by this thread as fast as it can */
    eager_ctx ec;
    ec.h = h; ec.p = &p; ec.ws_bytes = ws_bytes; ec.s = s;
    for (int b = 0; b < NSETS; ++b) { ec.ef[b] = ef[b]; ec.nf[b] = nf[b]; ec.eo[b] = eo[b]; ec.no[b] = no[b]; ec.go[b] = go[b]; ec.ws[b] = ws[b]; }
    const timing tc = time_steps(step_eager, &ec, K, W, s);
    float g5[5];
    CHECK_HIP(hipMemcpy(g5, go[0], sizeof g5, hipMemcpyDeviceToHost));
    printf("{\"bench\": \"abi_bench\", \"mode\": \"block\", \"workload\": \"C2%s: %lld nodes / %lld edges, (10,5,0)=>(3,4,5), through include/gnx.h from C (no Python, no torch)\", "
           "\"steps\": %d, \"captured_us_per_step\": %.4f, \"captured_event_us_per_step\": %.4f, \"captured_reps_us\": [%.4f, %.4f, %.4f], "
           "\"captured_what\": \"%d gnx_block_forward calls captured by the C program into one hipGraph, %d rotating buffer sets, median of 3 replays\", "
           "\"steps_us_per_step\": %.4f, \"steps_reps_us\": [%.4f, %.4f, %.4f], \"steps_what\": \"ONE gnx_block_forward_steps call over the same steps, captured into one hipGraph by the C program\", "
           "\"model_us_per_step\": %.4f, \"model_event_us_per_step\": %.4f, \"model_reps_us\": [%.4f, %.4f, %.4f], "
           "\"model_what\": \"gnx_model_forward per step (library-owned hipGraph of one forward, one hipGraphLaunch per step), %d models over %d buffer sets\", "
           "\"eager_us_per_step\": %.4f, \"eager_reps_us\": [%.4f, %.4f, %.4f], \"eager_what\": \"one gnx_block_forward per step on the stream, no hipGraph\", "
           "\"batch_ms\": %.3f, \"gf_out0\": %.6g}\n",
           csc ? " (bench.py's graph)" : " law (own draw)", (long long)N, (long long)E, K, ta.wall_us / K, ta.event_us / K, ta.reps_us[0] / K, ta.reps_us[1] / K,
           ta.reps_us[2] / K, K, NSETS, ta2.wall_us / K, ta2.reps_us[0] / K, ta2.reps_us[1] / K, ta2.reps_us[2] / K, tb.wall_us, tb.event_us, tb.reps_us[0], tb.reps_us[1], tb.reps_us[2], NSETS, NSETS, tc.wall_us, tc.reps_us[0], tc.reps_us[1],
           tc.reps_us[2], t_batch[0], (double)g5[0]);
    for (int b = 0; b < NSETS; ++b) CHECK_GNX(gnx_model_destroy(mc.m[b]));
    CHECK_HIP(hipGraphExecDestroy(rc.exec)); CHECK_HIP(hipGraphDestroy(graph));
    CHECK_HIP(hipGraphExecDestroy(rc2.exec)); CHECK_HIP(hipGraphDestroy(graph2));
  } else if (!strcmp(mode, "c4")) {
    gnx_block_params enc = block(in0, cd), dec = block(cd, out0);
    gnx_core_params c1 = core(cd), c2 = core(cd);
    const gnx_layer layers[4] = {{GNX_LAYER_BLOCK, 0, &enc}, {GNX_LAYER_CORE, 0, &c1}, {GNX_LAYER_CORE, 0, &c2}, {GNX_LAYER_BLOCK, 0, &dec}};
    model_ctx mc;
    mc.nsets = 1; mc.s = s;
    CHECK_GNX(gnx_model_create(h, layers, 4, 1, &mc.m[0]));
    mc.ef[0] = dev_uniform((size_t)E * 10, 0.f, 1.f); mc.nf[0] = dev_uniform((size_t)N * 5, 0.f, 1.f);
    CHECK_HIP(hipMalloc((void**)&mc.eo[0], sizeof(float) * (size_t)E * 3)); CHECK_HIP(hipMalloc((void**)&mc.no[0], sizeof(float) * (size_t)N * 4));
    CHECK_HIP(hipMalloc((void**)&mc.go[0], sizeof(float) * 5));
    const timing t = time_steps(step_model, &mc, K, W > 2 ? W : 2, s);
    float g5[5];
    CHECK_HIP(hipMemcpy(g5, mc.go[0], sizeof g5, hipMemcpyDeviceToHost));
    printf("{\"bench\": \"abi_bench\", \"mode\": \"c4\", \"workload\": \"C4: encoder -> 2 x GNCore(%d,%d,%d) -> decoder on the C2%s graph (%lld nodes / %lld edges) as ONE gnx_model, "
           "through include/gnx.h from C (no Python, no torch)\", \"steps\": %d, \"model_us_per_step\": %.3f, \"model_event_us_per_step\": %.3f, \"model_reps_us\": [%.3f, %.3f, %.3f], "
           "\"model_what\": \"gnx_model_forward per step: library-owned intermediates, one hipGraphLaunch of the captured 4-layer forward\", \"batch_ms\": %.3f, \"gf_out0\": %.6g}\n",
           cd[0], cd[1], cd[2], csc ? " (bench.py's)" : "-law", (long long)N, (long long)E, K, t.wall_us, t.event_us, t.reps_us[0], t.reps_us[1], t.reps_us[2], t_batch[0], (double)g5[0]);
    CHECK_GNX(gnx_model_destroy(mc.m[0]));
  } else {
    fprintf(stderr, "unknown --mode %s\n", mode);
    return 2;
  }
  CHECK_GNX(gnx_graphs_destroy(h));
  free(colptr); free(rowval);
  return 0; /* (device buffers are released with the process) */
}

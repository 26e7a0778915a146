// Hardware ceilings for a ~60 MB, ~10 us streaming problem on MI355X, measured the same way bench.py measures
// (hipGraph of N launches, rotating buffer sets > Infinity Cache).  Build: hipcc -O3 --offload-arch=gfx950 -o mb microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

// stream: read nr float4, write nw float4 (nw <= nr), grid-stride
__global__ void k_stream(const float4* __restrict__ in, float4* __restrict__ out, size_t nr, size_t nw) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float4 acc = make_float4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += stride) {
    float4 v = in[i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    if (i < nw) out[i] = v;
  }
  if (acc.x == 12345.678f) out[0] = acc;
}

// each thread: one 40-B row in (dword-aligned vector loads), one 12-B row out
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) F3u { float x, y, z; };
struct __attribute__((packed, aligned(4))) F2u { float x, y; };
__global__ void k_rows(const float* __restrict__ ef, float* __restrict__ out, int E) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float* p = ef + (size_t)e * 10;
  F4u a = *(const F4u*)p, b = *(const F4u*)(p + 4);
  F2u c = *(const F2u*)(p + 8);
  F3u o; o.x = a.x + b.x + c.x; o.y = a.y + b.y + c.y; o.z = a.z + b.z + a.w + b.w;
  *(F3u*)(out + (size_t)e * 3) = o;
}
// + rowval + random 20-B gather
__global__ void k_rows_gather(const float* __restrict__ ef, const int* __restrict__ rowval, const float* __restrict__ nf, float* __restrict__ out, int E) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float* p = ef + (size_t)e * 10;
  F4u a = *(const F4u*)p, b = *(const F4u*)(p + 4);
  F2u c = *(const F2u*)(p + 8);
  const int s = rowval[e];
  const float* q = nf + (size_t)s * 5;
  F4u g = *(const F4u*)q; float g4 = q[4];
  F3u o; o.x = a.x + b.x + c.x + g.x; o.y = a.y + b.y + c.y + g.y + g4; o.z = a.z + b.z + a.w + b.w + g.z + g.w;
  *(F3u*)(out + (size_t)e * 3) = o;
}
__global__ void k_gather_only(const int* __restrict__ rowval, const float* __restrict__ nf, float* __restrict__ out, int E) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int s = rowval[e];
  const float* q = nf + (size_t)s * 5;
  F4u g = *(const F4u*)q; float g4 = q[4];
  out[e] = g.x + g.y + g.z + g.w + g4;
}


template <int ROWF, int NLOAD>  // row stride in floats, number of floats loaded (4 -> dwordx4, 3 -> dwordx3, 5 -> x4+x1, 8 -> 2 x4)
__global__ void k_gather_var(const int* __restrict__ rowval, const float* __restrict__ nf, float* __restrict__ out, int E) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int s = rowval[e];
  const float* q = nf + (size_t)s * ROWF;
  float r = 0.f;
  if constexpr (NLOAD == 3) { F3u g = *(const F3u*)q; r = g.x + g.y + g.z; }
  else if constexpr (NLOAD == 4) { F4u g = *(const F4u*)q; r = g.x + g.y + g.z + g.w; }
  else if constexpr (NLOAD == 5) { F4u g = *(const F4u*)q; r = g.x + g.y + g.z + g.w + q[4]; }
  else if constexpr (NLOAD == 8) { F4u g = *(const F4u*)q; F4u h = *(const F4u*)(q + 4); r = g.x + g.y + g.z + g.w + h.x + h.y + h.z + h.w; }
  else if constexpr (NLOAD == 1) { r = q[0]; }
  out[e] = r;
}


// coalesced variant of rows+gather: a 256-thread WG moves the same bytes as 256 edges (640 float4 in, rowval, gather,
// 192 float4 out) but with fully coalesced 16-B accesses for the streams
__global__ void k_flat_gather(const float4* __restrict__ ef4, const int* __restrict__ rowval, const float* __restrict__ nf,
                              float4* __restrict__ out4, int E, int do_gather) {
  const int t = threadIdx.x;
  const size_t b = blockIdx.x;
  const int e = (int)(b * 256 + t);
  if (e >= E) return;
  float4 a0 = ef4[b * 640 + t], a1 = ef4[b * 640 + 256 + t], a2 = make_float4(0, 0, 0, 0);
  if (t < 128) a2 = ef4[b * 640 + 512 + t];
  float r = a0.x + a0.y + a0.z + a0.w + a1.x + a1.y + a1.z + a1.w + a2.x + a2.y;
  if (do_gather) {
    const int s = rowval[e];
    const float* q = nf + (size_t)s * 5;
    F4u g = *(const F4u*)q;
    r += g.x + g.y + g.z + g.w + q[4];
  }
  if (t < 192) out4[b * 192 + t] = make_float4(r, r, r, r);
}


typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v3f __attribute__((ext_vector_type(3)));
typedef v4f v4f_u __attribute__((aligned(4)));
typedef v2f v2f_u __attribute__((aligned(4)));
// rows + gather with NON-TEMPORAL streaming loads/stores (so the streams do not evict the gathered table from L2)
__global__ void k_rows_gather_nt(const float* __restrict__ ef, const int* __restrict__ rowval, const float* __restrict__ nf, float* __restrict__ out, int E, int mode) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float* p = ef + (size_t)e * 10;
  v4f a = __builtin_nontemporal_load((const v4f_u*)p), b = __builtin_nontemporal_load((const v4f_u*)(p + 4));
  v2f c = __builtin_nontemporal_load((const v2f_u*)(p + 8));
  const int s = (mode & 1) ? __builtin_nontemporal_load(rowval + e) : rowval[e];
  const float* q = nf + (size_t)s * 5;
  F4u g = *(const F4u*)q; float g4 = q[4];
  float ox = a.x + b.x + c.x + g.x, oy = a.y + b.y + c.y + g.y + g4, oz = a.z + b.z + a.w + b.w + g.z + g.w;
  float* o = out + (size_t)e * 3;
  if (mode & 2) { __builtin_nontemporal_store(ox, o); __builtin_nontemporal_store(oy, o + 1); __builtin_nontemporal_store(oz, o + 2); }
  else { F3u v; v.x = ox; v.y = oy; v.z = oz; *(F3u*)o = v; }
}


// rows + gather with the DEPENDENT chain first: rowval -> gather issued before the bulk row loads
__global__ void k_gather_first(const float* __restrict__ ef, const int* __restrict__ rowval, const float* __restrict__ nf, float* __restrict__ out, int E) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int s = rowval[e];
  const float* q = nf + (size_t)s * 5;
  F4u g = *(const F4u*)q; float g4 = q[4];
  asm volatile("" ::: "memory");
  const float* p = ef + (size_t)e * 10;
  F4u a = *(const F4u*)p, b = *(const F4u*)(p + 4);
  F2u c = *(const F2u*)(p + 8);
  F3u o; o.x = a.x + b.x + c.x + g.x; o.y = a.y + b.y + c.y + g.y + g4; o.z = a.z + b.z + a.w + b.w + g.z + g.w;
  *(F3u*)(out + (size_t)e * 3) = o;
}

template <class F>
static double time_graph(hipStream_t s, int n, F&& body) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < n; ++i) body(i);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  double best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms / n < best) best = ms / n;
  }
  return best * 1e3;  // us per launch
}

int main() {
  const int E = 1000000, N = 100000, NS = 8, NL = 200;
  hipStream_t s; CK(hipStreamCreate(&s));
  std::vector<float*> ef(NS), out(NS), nf(NS); std::vector<int*> rv(NS);
  std::vector<int> h_rv(E);
  srand(1); for (int i = 0; i < E; ++i) h_rv[i] = (int)(((unsigned)rand() * 2654435761u) % N);
  for (int k = 0; k < NS; ++k) {
    CK(hipMalloc(&ef[k], (size_t)E * 12 * 4)); CK(hipMalloc(&out[k], (size_t)E * 4 * 4)); CK(hipMalloc(&nf[k], (size_t)N * 5 * 4 + 64));
    CK(hipMalloc(&rv[k], (size_t)E * 4));
    CK(hipMemset(ef[k], 0, (size_t)E * 12 * 4)); CK(hipMemset(nf[k], 0, (size_t)N * 5 * 4 + 64));
    CK(hipMemcpy(rv[k], h_rv.data(), (size_t)E * 4, hipMemcpyHostToDevice));
  }
  printf("empty 2048 WGs x256      : %7.2f us\n", time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, s, (int*)nullptr); }));
  printf("empty 8192 WGs x256      : %7.2f us\n", time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_empty, dim3(8192), dim3(256), 0, s, (int*)nullptr); }));
  printf("empty 1 WG               : %7.2f us\n", time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, s, (int*)nullptr); }));
  const size_t nr = (size_t)E * 10 / 4 + (size_t)E / 4, nw = (size_t)E * 3 / 4 + N;  // ~44 MB read, ~13.6 MB write
  for (int grid : {1024, 2048, 4096, 8192}) {
    double t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, s, (const float4*)ef[i % NS], (float4*)out[i % NS], nr, nw); });
    printf("stream 44MB rd + 13.6MB wr, grid %5d (cold): %7.2f us  -> %.2f TB/s\n", grid, t, (nr + nw) * 16 / t / 1e6);
  }
  {
    double t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s, (const float4*)ef[0], (float4*)out[0], nr, nw); });
    printf("stream same, one buffer set (warm)         : %7.2f us  -> %.2f TB/s\n", t, (nr + nw) * 16 / t / 1e6);
  }
  {
    double t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_rows, dim3((E + 255) / 256), dim3(256), 0, s, ef[i % NS], out[i % NS], E); });
    printf("rows: 40-B row in / 12-B row out per thread (cold): %7.2f us -> %.2f TB/s\n", t, 52.0 * E / t / 1e6);
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_rows_gather, dim3((E + 255) / 256), dim3(256), 0, s, ef[i % NS], rv[i % NS], nf[i % NS], out[i % NS], E); });
    printf("rows + rowval + 20-B random gather (cold)        : %7.2f us -> %.2f TB/s alg(58 B/edge)\n", t, 58.0 * E / t / 1e6);
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_gather_only, dim3((E + 255) / 256), dim3(256), 0, s, rv[i % NS], nf[i % NS], out[i % NS], E); });
    printf("rowval + 20-B random gather only (cold)          : %7.2f us\n", t);
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_rows_gather, dim3((E + 255) / 256), dim3(256), 0, s, ef[0], rv[0], nf[0], out[0], E); });
    printf("rows + rowval + gather, one buffer set (warm)    : %7.2f us\n", t);
  }

  {
    float* nf8; CK(hipMalloc(&nf8, (size_t)N * 8 * 4 + 64)); CK(hipMemset(nf8, 0, (size_t)N * 8 * 4 + 64));
    std::vector<int> h_seq(E); for (int i = 0; i < E; ++i) h_seq[i] = (int)(((long long)i * N) / E);
    int* rv_seq; CK(hipMalloc(&rv_seq, (size_t)E * 4)); CK(hipMemcpy(rv_seq, h_seq.data(), (size_t)E * 4, hipMemcpyHostToDevice));
    std::vector<int> h_small(E); for (int i = 0; i < E; ++i) h_small[i] = h_rv[i] % 4096;
    int* rv_small; CK(hipMalloc(&rv_small, (size_t)E * 4)); CK(hipMemcpy(rv_small, h_small.data(), (size_t)E * 4, hipMemcpyHostToDevice));
    const dim3 gr((E + 255) / 256), bl(256);
    double t;
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_empty, gr, bl, 0, s, (int*)nullptr); });
    printf("empty 3907 WGs                          : %7.2f us\n", t);
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL((k_gather_var<5, 5>), gr, bl, 0, s, rv[i % NS], nf[i % NS], out[i % NS], E); });
    printf("gather 20B rows x4+x1 random   (cold)   : %7.2f us\n", t);
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL((k_gather_var<5, 5>), gr, bl, 0, s, rv[0], nf[0], out[0], E); });
    printf("gather 20B rows x4+x1 random   (warm)   : %7.2f us\n", t);
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL((k_gather_var<5, 4>), gr, bl, 0, s, rv[0], nf[0], out[0], E); });
    printf("gather 20B stride, x4 only     (warm)   : %7.2f us\n", t);
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL((k_gather_var<5, 1>), gr, bl, 0, s, rv[0], nf[0], out[0], E); });
    printf("gather 20B stride, x1 only     (warm)   : %7.2f us\n", t);
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL((k_gather_var<3, 3>), gr, bl, 0, s, rv[0], nf[0], out[0], E); });
    printf("gather 12B rows x3             (warm)   : %7.2f us\n", t);
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL((k_gather_var<4, 4>), gr, bl, 0, s, rv[0], nf[0], out[0], E); });
    printf("gather 16B rows x4 (aligned)   (warm)   : %7.2f us\n", t);
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL((k_gather_var<8, 8>), gr, bl, 0, s, rv[0], nf8, out[0], E); });
    printf("gather 32B rows 2 x x4         (warm)   : %7.2f us\n", t);
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL((k_gather_var<5, 5>), gr, bl, 0, s, rv_seq, nf[0], out[0], E); });
    printf("gather 20B rows sequential src (warm)   : %7.2f us\n", t);
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL((k_gather_var<5, 5>), gr, bl, 0, s, rv_small, nf[0], out[0], E); });
    printf("gather 20B rows, 4096-row table (warm)  : %7.2f us\n", t);
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL((k_gather_var<5, 1>), gr, bl, 0, s, rv_seq, nf[0], out[0], E); });
    printf("rowval + 1 coalesced-ish dword  (warm)  : %7.2f us\n", t);
  }

  {
    const dim3 gr(E / 256), bl(256);
    double t;
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_flat_gather, gr, bl, 0, s, (const float4*)ef[i % NS], rv[i % NS], nf[i % NS], (float4*)out[i % NS], E, 0); });
    printf("flat coalesced 40B/edge in, 12B/edge out, no gather (cold): %7.2f us\n", t);
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_flat_gather, gr, bl, 0, s, (const float4*)ef[i % NS], rv[i % NS], nf[i % NS], (float4*)out[i % NS], E, 1); });
    printf("flat coalesced + rowval + random 20B gather        (cold): %7.2f us\n", t);
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_flat_gather, gr, bl, 0, s, (const float4*)ef[0], rv[0], nf[0], (float4*)out[0], E, 1); });
    printf("flat coalesced + rowval + random 20B gather        (warm): %7.2f us\n", t);
  }

  {
    const dim3 gr((E + 255) / 256), bl(256);
    for (int mode = 0; mode < 4; ++mode) {
      double t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_rows_gather_nt, gr, bl, 0, s, ef[i % NS], rv[i % NS], nf[i % NS], out[i % NS], E, mode); });
      printf("rows(nt loads) + gather, mode %d (1: nt rowval, 2: nt stores)  (cold): %7.2f us\n", mode, t);
    }
    double t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_rows_gather_nt, gr, bl, 0, s, ef[0], rv[0], nf[0], out[0], E, 1); });
    printf("rows(nt loads) + gather, mode 1 (warm): %7.2f us\n", t);
  }

  {
    const dim3 gr((E + 255) / 256), bl(256);
    double t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_gather_first, gr, bl, 0, s, ef[i % NS], rv[i % NS], nf[i % NS], out[i % NS], E); });
    printf("rowval->gather FIRST, then rows (cold): %7.2f us\n", t);
    t = time_graph(s, NL, [&](int i) { hipLaunchKernelGGL(k_gather_first, gr, bl, 0, s, ef[0], rv[0], nf[0], out[0], E); });
    printf("rowval->gather FIRST, then rows (warm): %7.2f us\n", t);
  }
  return 0;
}

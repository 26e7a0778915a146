// Probe (not part of the library): v_mfma_f32_4x4x1_16b_f32 with the A matrix of ONE block broadcast to all sixteen (cbsz = 4, abid = t) as
// the FMA stream of a "one row per lane" kernel:   d[i] = fma(a[lane 4t + i], b[own lane], c[i])   for i = 0..3, every lane —
// four outputs of the lane's own row per instruction, the four weights taken from lanes 4t..4t+3 of the A register (sixteen weight quads per
// register, picked by the instruction's abid field): no scalar loads, no LDS, no data movement.
//   1. semantics + bits: against fmaf on the host, every abid
//   2. cycles per instruction on one wave per SIMD, alone and with independent v_fma_f32 / v_pk_fma_f32 between them
//   3. two / four waves per SIMD: a wave that issues only these MFMAs beside a wave that issues only packed FMAs — do the two pipes overlap?
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/mfma_4x4_rowbcast_probe.hip -o /tmp/mfma4 && /tmp/mfma4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int T>
__device__ __forceinline__ v4f mm(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, T, 0); }

__global__ void k_sem(const float* a, const float* b, const float* c, float* d) {  // one wave; d[t][lane][4]
  const int l = threadIdx.x;
  const float av = a[l], bv = b[l];
  v4f cv = {c[4 * l], c[4 * l + 1], c[4 * l + 2], c[4 * l + 3]};
  v4f r[16];
  r[0] = mm<0>(av, bv, cv); r[1] = mm<1>(av, bv, cv); r[2] = mm<2>(av, bv, cv); r[3] = mm<3>(av, bv, cv);
  r[4] = mm<4>(av, bv, cv); r[5] = mm<5>(av, bv, cv); r[6] = mm<6>(av, bv, cv); r[7] = mm<7>(av, bv, cv);
  r[8] = mm<8>(av, bv, cv); r[9] = mm<9>(av, bv, cv); r[10] = mm<10>(av, bv, cv); r[11] = mm<11>(av, bv, cv);
  r[12] = mm<12>(av, bv, cv); r[13] = mm<13>(av, bv, cv); r[14] = mm<14>(av, bv, cv); r[15] = mm<15>(av, bv, cv);
  for (int t = 0; t < 16; ++t)
    for (int i = 0; i < 4; ++i) d[(t * 64 + l) * 4 + i] = r[t][i];
}

// MODE 0: NM MFMAs (10 independent accumulators) per step; NV independent v_fma_f32 (NV > 0) or v_pk_fma_f32 (NV < 0: -NV of them) after each MFMA
// ROLE split (SPLIT = true, 512-thread workgroups): waves 0-3 issue only the MFMAs, waves 4-7 only 2*10 packed FMAs per step
template <int NV, bool SPLIT, int WHICH>
__global__ __launch_bounds__(SPLIT ? 512 : 256) void k_rate(float* out, unsigned long long* cyc, int iters, float seed) {
  const int wv = threadIdx.x >> 6;
  v4f acc[10];
  v2f pa[20];
  float fa[8];
#pragma unroll
  for (int j = 0; j < 10; ++j) acc[j] = v4f{(float)j, 1.f, 2.f, 3.f};
#pragma unroll
  for (int j = 0; j < 20; ++j) pa[j] = v2f{(float)j, -(float)j};
#pragma unroll
  for (int j = 0; j < 8; ++j) fa[j] = (float)j;
  const float av = seed + threadIdx.x * 1e-6f, bv = 1.f - threadIdx.x * 1e-6f;
  v2f x = {av, bv}, w = {bv, av};
  const bool do_m = !SPLIT || (WHICH != 2 && wv < 4), do_v = SPLIT && (WHICH != 1 && wv >= 4);
  const unsigned long long t0 = clock64();
  if (do_m) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        acc[j] = j & 1 ? mm<3>(av, bv, acc[j]) : mm<11>(av, bv, acc[j]);
        if constexpr (NV > 0) {
#pragma unroll
          for (int u = 0; u < NV; ++u) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(fa[(j * NV + u) & 7]) : "v"(av), "v"(bv));
        } else if constexpr (NV < 0) {
#pragma unroll
          for (int u = 0; u < -NV; ++u) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[(j * -NV + u) % 20]) : "v"(w), "v"(x));
        }
      }
    }
  }
  if (do_v) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 20; ++j) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pa[j]) : "v"(w), "v"(x));
    }
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 10; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
#pragma unroll
  for (int j = 0; j < 20; ++j) s += pa[j].x + pa[j].y;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += fa[j];
  if (s == 12345.678f) out[0] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wv] = t1 - t0;
}

template <int NV, bool SPLIT, int WHICH>
static int run(const char* what, int wg_per_cu, float* out, unsigned long long* cyc) {
  const int iters = 2000, grid = 256 * wg_per_cu, threads = SPLIT ? 512 : 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int warm = 0; warm < 40; ++warm) hipLaunchKernelGGL((k_rate<NV, SPLIT, WHICH>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
  CK(hipEventRecord(e0));
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k_rate<NV, SPLIT, WHICH>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[8];
  CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
  const double steps = (double)iters;
  printf("%-78s %d wg/CU: %8.3f ms per launch; wave 0: %6.1f cycles per step", what, wg_per_cu, ms / 10, h[0] / steps);
  if (SPLIT) printf(", wave 4: %6.1f", h[4] / steps);
  printf("\n");
  return 0;
}

int main() {
  // ---- 1. semantics ----
  std::vector<float> a(64), b(64), c(256), d(16 * 64 * 4);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) / 16777216.f - 0.5f) * 4.f; };
  for (auto& v : a) v = rnd();
  for (auto& v : b) v = rnd();
  for (auto& v : c) v = rnd();
  a[5] = 1e-39f; c[7] = -0.f;  // a subnormal weight, a negative zero
  float *da, *db, *dc, *dd, *out;
  unsigned long long* cyc;
  CK(hipMalloc(&da, 256)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&dc, 1024)); CK(hipMalloc(&dd, d.size() * 4)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 64));
  CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
  CK(hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int t = 0; t < 16; ++t)
    for (int l = 0; l < 64; ++l)
      for (int i = 0; i < 4; ++i) {
        const float want = fmaf(a[4 * t + i], b[l], c[4 * l + i]), got = d[(t * 64 + l) * 4 + i];
        if (memcmp(&want, &got, 4) != 0 && bad++ < 8) printf("  mismatch t=%d lane=%d i=%d: want %.9g got %.9g\n", t, l, i, want, got);
      }
  printf("semantics d[i] = fma(a[lane 4*abid + i], b[lane], c[i]) over 16 abid x 64 lanes x 4: %s (%d of 4096 differ from fmaf bitwise)\n", bad ? "DIFFERENT" : "bit-identical to fmaf", bad);

  // ---- 2. one wave per SIMD ----
  if (run<0, false, 0>("10 MFMA 4x4x1 per step", 1, out, cyc)) return 1;
  if (run<1, false, 0>("10 x (MFMA + 1 v_fma_f32)", 1, out, cyc)) return 1;
  if (run<2, false, 0>("10 x (MFMA + 2 v_fma_f32)", 1, out, cyc)) return 1;
  if (run<-1, false, 0>("10 x (MFMA + 1 v_pk_fma_f32)", 1, out, cyc)) return 1;
  if (run<-2, false, 0>("10 x (MFMA + 2 v_pk_fma_f32)", 1, out, cyc)) return 1;
  if (run<0, false, 0>("10 MFMA 4x4x1 per step", 2, out, cyc)) return 1;
  if (run<0, false, 0>("10 MFMA 4x4x1 per step", 4, out, cyc)) return 1;
  // ---- 3. role split: waves 0-3 MFMA only, waves 4-7 packed FMA only ----
  if (run<0, true, 1>("split, only the MFMA waves work (10 MFMA per step)", 1, out, cyc)) return 1;
  if (run<0, true, 2>("split, only the VALU waves work (20 v_pk_fma_f32 per step)", 1, out, cyc)) return 1;
  if (run<0, true, 0>("split, both (10 MFMA | 20 v_pk_fma_f32 per step)", 1, out, cyc)) return 1;
  if (run<0, true, 0>("split, both (10 MFMA | 20 v_pk_fma_f32 per step)", 2, out, cyc)) return 1;
  return 0;
}

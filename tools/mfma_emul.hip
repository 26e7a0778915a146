// fp32 products on the bf16 matrix cores — a feasibility measurement for DESIGN.md section 8 (NOT part of libgnx):
//   x = hi + mid + lo (three bf16 parts: 24 mantissa bits, fp32's exponent range),  a*b ~ hh + hm + mh + hl + lh + mm, fp32 accumulation.
// (1) accuracy of C = A*B (32 x K times K x 32, one wavefront) against float64, in units of sum|a||b|, for: the fp32 MFMA
//     (v_mfma_f32_32x32x2_f32), six-term and three-term bf16 emulation (v_mfma_f32_32x32x16_bf16);
// (2) matrix-pipe rate, registers only: 8 fp32 MFMAs vs 6 (3) bf16 MFMAs per 32 x 32 x 16 block, and the same with the operand split
//     (6 VALU conversions + 4 subtractions per element pair) done inside the loop.
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mfma_emul.hip -o /tmp/mfma_emul && /tmp/mfma_emul
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

// A [32][K] row-major, B [K][32] row-major, C [3][32][32]: variant 0 fp32 MFMA, 1 six-term emulation, 2 three-term emulation
__global__ void k_acc(const float* A, const float* B, int K, float* C) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 c0 = {0}, c1 = {0}, c2 = {0};
  for (int k = 0; k < K; k += 2) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], c0, 0, 0, 0);
  for (int k = 0; k < K; k += 16) {
    bf16x8 ah, am, al, bh, bm, bl;
    for (int j = 0; j < 8; ++j) {  // lane map of the 32x32x16 bf16 MFMA: A[row r][k = 8h + j], B[k = 8h + j][col r]
      __bf16 x, y, z;
      split3(A[r * K + k + 8 * h + j], x, y, z); ah[j] = x; am[j] = y; al[j] = z;
      split3(B[(k + 8 * h + j) * 32 + r], x, y, z); bh[j] = x; bm[j] = y; bl[j] = z;
    }
    // small terms first
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c2, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c2, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c2, 0, 0, 0);
  }
  for (int q = 0; q < 16; ++q) {  // C/D layout: col = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
    const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
    C[0 * 1024 + row * 32 + r] = c0[q];
    C[1 * 1024 + row * 32 + r] = c1[q];
    C[2 * 1024 + row * 32 + r] = c2[q];
  }
}

// mode 0: 8 fp32 MFMAs per step; 1: 6 bf16 MFMAs; 2: 3 bf16 MFMAs; 3: 6 bf16 MFMAs + both operand fragments split from fp32 registers
// inside the loop; 4: as 3 with only ONE operand split per step (the other one comes pre-split, as constant weights would)
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(int iters, float seed, float* out) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  float fa = seed + threadIdx.x, fb = seed - threadIdx.x;
  bf16x8 ah, am, al, bh, bm, bl;
  float xa[8], xb[8];
  for (int j = 0; j < 8; ++j) { xa[j] = seed * (j + 1) + threadIdx.x; xb[j] = seed * (j + 3) - threadIdx.x; }
  for (int j = 0; j < 8; ++j) { __bf16 x, y, z; split3(xa[j], x, y, z); ah[j] = x; am[j] = y; al[j] = z; split3(xb[j], x, y, z); bh[j] = x; bm[j] = y; bl[j] = z; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MODE == 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
      } else {
        if (MODE >= 3) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            __bf16 x, y, z;
            split3(xa[j] + acc[i][j], x, y, z); ah[j] = x; am[j] = y; al[j] = z;  // (data-dependent: the split cannot be hoisted)
            if (MODE == 3) { split3(xb[j] + acc[i][j + 8], x, y, z); bh[j] = x; bm[j] = y; bl[j] = z; }
          }
        }
        if (MODE != 2) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i], 0, 0, 0);
        }
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
  if (s == 12345.678f) out[0] = s;  // keep the work
}

template <int MODE>
static double rate(const char* what, int blocks_per_cu) {
  float* d; CK(hipMalloc((void**)&d, 4));
  const int iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_rate<MODE>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, 10, 1e-30f, d);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_rate<MODE>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, iters, 1e-30f, d);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  // every (wave, i, iteration) is one 32 x 32 x 16 block = 32768 fp32-equivalent flops
  const double flops = 256.0 * blocks_per_cu * 4 /*waves*/ * 4 /*i*/ * (double)iters * 32768.0;
  printf("  %-78s %7.3f ms  %7.1f TFLOP/s fp32-equivalent\n", what, ms, flops / ms / 1e9);
  CK(hipFree(d));
  return flops / ms / 1e9;
}

int main() {
  const int K = 512;
  std::vector<float> A(32 * K), B(K * 32);
  srand(7);
  for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f + 3.f;  // a mean away from zero, like un-normalised features
  for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
  float *dA, *dB, *dC;
  CK(hipMalloc((void**)&dA, A.size() * 4)); CK(hipMalloc((void**)&dB, B.size() * 4)); CK(hipMalloc((void**)&dC, 3 * 1024 * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_acc, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
  std::vector<float> C(3 * 1024);
  CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  const char* names[3] = {"fp32 MFMA (32x32x2)", "bf16 x 6 terms (32x32x16)", "bf16 x 3 terms (hh + hm + mh)"};
  printf("accuracy of a 32 x %d x 32 product against float64, max over the tile of |err| / sum|a||b|:\n", K);
  for (int v = 0; v < 3; ++v) {
    double worst = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double ref = 0, scale = 0;
      for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 32 + j]; scale += std::fabs((double)A[i * K + k] * B[k * 32 + j]); }
      worst = std::max(worst, std::fabs((double)C[v * 1024 + i * 32 + j] - ref) / scale);
    }
    printf("  %-32s %.3e\n", names[v], worst);
  }
  printf("matrix-pipe rate, registers only, 4 waves per SIMD (256 CUs x 4 workgroups of 256 threads):\n");
  rate<0>("8 x v_mfma_f32_32x32x2_f32 per 32x32x16 block", 4);
  rate<1>("6 x v_mfma_f32_32x32x16_bf16 (operands pre-split)", 4);
  rate<2>("3 x v_mfma_f32_32x32x16_bf16 (operands pre-split)", 4);
  rate<4>("6 x bf16 MFMA + ONE operand fragment split per block (8 elements: the other pre-split, e.g. weights)", 4);
  rate<3>("6 x bf16 MFMA + BOTH operand fragments split per block (16 elements)", 4);
  return 0;
}

// What the fp32 matrix cores deliver to a dense stream of v_mfma_f32_32x32x2_f32 on MI355X (registers only: no LDS, no memory), by waves per SIMD.
// The nominal peak (256 CUs x 4 SIMDs x 4096 FLOP / 64 cycles x 2.4 GHz = 157 TF/s) assumes the boost clock; a dense MFMA stream runs at the clock the power
// limit leaves.  Build: hipcc -O3 --offload-arch=gfx950 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  // operands: 8 + 8 per-lane pseudo-random values in registers (constant or zero operands toggle few wires and read high: the clock a dense
  // stream gets depends on the power its data draw); a0 = 0 selects the constant-operand form
  float av[8], bv[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    unsigned x = (threadIdx.x * 8 + u + blockIdx.x * 2048) * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    unsigned y = x * 3266489917u; y ^= y >> 16;
    av[u] = a0 == 0.f ? 1.f : ((x & 0xffffff) / 16777216.f - 0.5f) * a0;
    bv[u] = a0 == 0.f ? 0.5f : ((y & 0xffffff) / 16777216.f - 0.5f) * b0;
  }
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(u + i) & 7], bv[(u + 3 * i) & 7], acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) s += acc[i][q];
  if (s == 12345.678f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<unsigned long long*>(out)[1] = t1 - t0;
}

int main() {
  float* out; CK(hipMalloc(&out, 64));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000;  // x 16 x NACC MFMAs per wave
  for (int rnd = 0; rnd < 2; ++rnd)
  for (int wps = 1; wps <= 4; wps *= 2) {           // waves per SIMD = workgroups (256 threads = one wave per SIMD) per CU
    printf("%s operands, ", rnd ? "pseudo-random" : "constant");
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((k_mfma<4>), dim3(cus * wps), dim3(256), 0, 0, out, iters / wps, rnd ? 2.f : 0.f, 2.f);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long clk[2]; CK(hipMemcpy(clk, out, 16, hipMemcpyDeviceToHost));
      const double mfmas = (double)cus * wps * 4 * (iters / wps) * 16 * 4;      // workgroups x waves x iterations x 16 x NACC
      const double flops = mfmas * 4096.0;
      const double cyc_per_mfma = (double)clk[1] / ((double)(iters / wps) * 16 * 4);  // shader clocks per MFMA of wave 0 (its SIMD runs wps such waves)
      printf("waves/SIMD %d: %.1f us, %.1f TF/s, %.1f shader-clock ticks per MFMA issued by one wave, implied matrix-core clock %.2f GHz (64 cycles per MFMA)\n",
             wps, ms * 1e3, flops / (ms * 1e-3) * 1e-12, cyc_per_mfma, flops / (ms * 1e-3) / (cus * 4.0 * 64.0) * 1e-9);
    }
  }
  return 0;
}

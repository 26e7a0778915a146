// A dense matrix-instruction load as a tiny shared library (no gnx code), for tools/experiments/thread_race_probe.py PROBE_OTHER=agg_<kind>:
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/libmfma_agg.so tools/experiments/mfma_agg.hip
// kinds: bf16 / f32 (operands in registers), bf16lds (operands re-read from a 64 KB LDS image with ds_read_b128 before every instruction, as a GEMM does)
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void k_agg(int iters, float* sink) {
  __shared__ __attribute__((aligned(16))) unsigned s_img[KIND == 2 ? 16384 : 4];
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  bf16x8 a[8], b[8];
  unsigned h = 0x9e3779b9u * (threadIdx.x + 257u * blockIdx.x + 1u);
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    unsigned w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; w[j] = (h & 0x807f807fu) | 0x3f003f00u; }
    a[f] = __builtin_bit_cast(bf16x8, u32x4{w[0], w[1], w[2], w[3]});
    b[f] = __builtin_bit_cast(bf16x8, u32x4{w[4], w[5], w[6], w[7]});
  }
  if (KIND == 2) {
    for (int i = threadIdx.x; i < 16384; i += 256) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; s_img[i] = (h & 0x807f807fu) | 0x3f003f00u; }
    __syncthreads();
  }
  const float fa = __uint_as_float((h & 0x807fffffu) | 0x3f000000u), fb = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      if (KIND == 2) {
        const bf16x8 al = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4*>(s_img)[(threadIdx.x + 67 * u + 131 * it) & 4095]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b[(u + 3) & 7], acc, 0, 0, 0);
      } else if (KIND == 0 || KIND == 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u & 7], b[(u + 3) & 7], acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] *= 0.001f;
    if (KIND == 3) {  // a 240-register wave that keeps writing its HIGH registers (and runs matrix instructions on them)
      asm volatile(
          "v_mov_b32 v200, %0\n\tv_mov_b32 v201, %0\n\tv_mov_b32 v202, %0\n\tv_mov_b32 v203, %0\n\tv_mov_b32 v204, %0\n\tv_mov_b32 v205, %0\n\tv_mov_b32 v206, %0\n\tv_mov_b32 v207, %0\n\t"
          "v_mov_b32 v208, 0\n\tv_mov_b32 v209, 0\n\tv_mov_b32 v210, 0\n\tv_mov_b32 v211, 0\n\tv_mov_b32 v212, 0\n\tv_mov_b32 v213, 0\n\tv_mov_b32 v214, 0\n\tv_mov_b32 v215, 0\n\t"
          "v_mov_b32 v216, 0\n\tv_mov_b32 v217, 0\n\tv_mov_b32 v218, 0\n\tv_mov_b32 v219, 0\n\tv_mov_b32 v220, 0\n\tv_mov_b32 v221, 0\n\tv_mov_b32 v222, 0\n\tv_mov_b32 v223, 0\n\t"
          "s_nop 4\n\t"
          "v_mfma_f32_32x32x16_bf16 v[208:223], v[200:203], v[204:207], v[208:223]\n\t"
          "v_mfma_f32_32x32x16_bf16 v[208:223], v[204:207], v[200:203], v[208:223]\n\t"
          "v_mfma_f32_32x32x16_bf16 v[208:223], v[200:203], v[204:207], v[208:223]\n\t"
          "v_mfma_f32_32x32x16_bf16 v[208:223], v[204:207], v[200:203], v[208:223]\n\t"
          "s_nop 15\n\ts_nop 15\n\t"
          "v_mov_b32 v236, v208\n\tv_mov_b32 v237, v209\n\tv_mov_b32 v238, v210\n\tv_mov_b32 v239, v211"
          :: "v"(0x3f803f80u + (unsigned)it)
          : "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219",
            "v220", "v221", "v222", "v223", "v236", "v237", "v238", "v239");
    }
  }
  if (acc[0] == 12345.678f) sink[0] = acc[1];
}

extern "C" int agg_launch(int kind, int iters, int grid, void* stream) {
  static float* sink = nullptr;
  if (!sink && hipMalloc(&sink, 64) != hipSuccess) return 1;
  hipStream_t s = (hipStream_t)stream;
  if (kind == 0) hipLaunchKernelGGL(k_agg<0>, dim3(grid), dim3(256), 0, s, iters, sink);
  else if (kind == 1) hipLaunchKernelGGL(k_agg<1>, dim3(grid), dim3(256), 0, s, iters, sink);
  else if (kind == 3) hipLaunchKernelGGL(k_agg<3>, dim3(grid), dim3(256), 0, s, iters, sink);
  else hipLaunchKernelGGL(k_agg<2>, dim3(grid), dim3(256), 0, s, iters, sink);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

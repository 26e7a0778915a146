// Rate probe (not part of the library): how fast does gfx950 issue v_pk_fma_f32 when src0 is an SGPR pair (the form PkStream / CorePostStream
// use: one weight per instruction, op_sel picks its half for both rows) against the all-VGPR form and against plain v_fma_f32?  Registers
// only, 1 / 2 / 4 waves per SIMD, every CU.
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/pk_fma_rate.hip -o /tmp/pk_rate && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096, UNROLL = 20;  // 20 independent accumulators (a FeedForward group)

template <int MODE>
__global__ __launch_bounds__(256) void rate(const float* w, float* out) {
  typedef const float __attribute__((address_space(4))) * cfloatp;
  v2f ws;
  asm volatile("s_load_dwordx2 %0, %1, 0x0\n s_waitcnt lgkmcnt(0)" : "=s"(ws) : "s"(reinterpret_cast<cfloatp>(reinterpret_cast<size_t>(w))));
  v2f acc[UNROLL], x = {1.0f + threadIdx.x * 1e-6f, 1.0f - threadIdx.x * 1e-6f}, wv = {w[0], w[1]};
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) acc[j] = v2f{(float)j, (float)-j};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      if constexpr (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[j]) : "s"(ws), "v"(x));
      else if constexpr (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[j]) : "s"(ws), "v"(x));
      else if constexpr (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(wv), "v"(x));
      else { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].x) : "s"(ws.x), "v"(x.x)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].y) : "s"(ws.x), "v"(x.y)); }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) s += acc[j].x + acc[j].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, const float* w, float* out, int wg_per_cu) {
  const int grid = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int warm = 0; warm < 30; ++warm) hipLaunchKernelGGL(rate<MODE>, dim3(grid), dim3(256), 0, 0, w, out);  // ~100 ms: settled clocks
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(rate<MODE>, dim3(grid), dim3(256), 0, 0, w, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 10.0 * grid * 256.0 * ITERS * UNROLL * 4.0;  // 2 FMAs = 4 flop per lane and instruction (pair)
  printf("%-44s %d waves/SIMD: %7.1f TFLOP/s\n", name, wg_per_cu, flop / (ms * 1e-3) / 1e12);
}

int main() {
  float hw[2] = {1.0000001f, 0.9999999f};
  float *w, *out;
  hipMalloc(&w, sizeof hw); hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
  hipMemcpy(w, hw, sizeof hw, hipMemcpyHostToDevice);
  for (int wg = 1; wg <= 4; wg *= 2) {
    run<0>("v_pk_fma_f32, SGPR pair src0 (lo broadcast)", w, out, wg);
    run<1>("v_pk_fma_f32, SGPR pair src0 (hi broadcast)", w, out, wg);
    run<2>("v_pk_fma_f32, all VGPR", w, out, wg);
    run<3>("2 x v_fma_f32, SGPR src0", w, out, wg);
  }
  return 0;
}

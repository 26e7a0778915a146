// Probe (not part of the library): what v_pk_fma_f32 reads when src0 is an SGPR pair, per op_sel / op_sel_hi setting, on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/pk_fma_sgpr_probe.hip -o /tmp/pk_probe && /tmp/pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const float __attribute__((address_space(4))) * cfloatp;
__global__ void probe(const float* w, const float* x, float* out) {
  v2f ws;
  asm volatile("s_load_dwordx2 %0, %1, 0x0\n s_waitcnt lgkmcnt(0)" : "=s"(ws) : "s"(reinterpret_cast<cfloatp>(reinterpret_cast<size_t>(w))));
  v2f xv = {x[2 * threadIdx.x], x[2 * threadIdx.x + 1]};
  v2f a = {0.f, 0.f}, b = {0.f, 0.f}, c = {0.f, 0.f};
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "s"(ws), "v"(xv));                                   // default selects
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(b) : "s"(ws), "v"(xv));                 // lo weight on both rows
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(c) : "s"(ws), "v"(xv));  // hi weight on both rows
  float* o = out + 6 * threadIdx.x;
  o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; o[4] = c.x; o[5] = c.y;
}
int main() {
  float hw[2] = {3.f, 5.f}, hx[4] = {1.f, 10.f, 2.f, 20.f}, ho[12];
  float *w, *x, *o;
  hipMalloc(&w, sizeof hw); hipMalloc(&x, sizeof hx); hipMalloc(&o, sizeof ho);
  hipMemcpy(w, hw, sizeof hw, hipMemcpyHostToDevice); hipMemcpy(x, hx, sizeof hx, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(2), 0, 0, w, x, o);
  hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
  for (int t = 0; t < 2; ++t)
    printf("lane %d  x = (%g, %g)  w = (3, 5):  default (%g, %g)   lo-broadcast (%g, %g)   hi-broadcast (%g, %g)\n", t, hx[2 * t], hx[2 * t + 1],
           ho[6 * t], ho[6 * t + 1], ho[6 * t + 2], ho[6 * t + 3], ho[6 * t + 4], ho[6 * t + 5]);
  printf("expected: default (3x0, 5x1)   lo-broadcast (3x0, 3x1)   hi-broadcast (5x0, 5x1)\n");
  return 0;
}

// Stand-alone cut of the site of profiles/r06_overlap_hazard.log: a workgroup stages rows through the LayerNorm-on-load branch of k_ffn_fused / k_rows_gemm —
// row statistics (mean, 1 / sigma) in LDS as float2 s_ln[rows], read back eight lanes per address; gamma / beta as float4 from LDS; the first vector
// instruction consumes the statistics right behind the compiler's counted wait — with everything else of those kernels removed (no matrix instruction, no
// global loads in the loop, no epilogue).  Every workgroup re-stages NEW statistics per round (round r: mean = row + r, inv = 1 + r / 64) and checks what its
// lanes read against what was written; mismatching (round, lane) pairs are counted per lane.  GUARD = the two-part guard of the library.
//
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o /tmp/lds_hazard_repro.so tools/experiments/lds_hazard_repro.hip
//   python tools/experiments/lds_hazard_repro.py          (launches it beside a torch bf16 GEMM loop on another stream)
#include <hip/hip_runtime.h>

namespace {
constexpr int ROWS = 128, KQ = 8;  // rows of the tile, quads per row chunk (a thread stages quad tid % 8 of rows tid / 8 + 64 i: 512 threads, i < 2)

template <bool GUARD>
__global__ __launch_bounds__(512) void k_victim(const float* __restrict__ x, int rounds, unsigned long long* __restrict__ bad_by_lane, float* __restrict__ sink) {
  __shared__ float2 s_ln[ROWS];
  __shared__ __attribute__((aligned(16))) float4 s_lng[KQ];
  __shared__ __attribute__((aligned(16))) float4 s_lnb[KQ];
  __shared__ __attribute__((aligned(16))) float sA[ROWS * 33];
  const int tid = threadIdx.x, a_r = tid >> 3, a_c4 = tid & 7;
  float4 v0 = reinterpret_cast<const float4*>(x)[(blockIdx.x * 512 + tid) & 4095];  // (some data to normalise: the same every round)
  float acc = 0.f;
  unsigned long long bad = 0;
  for (int r = 0; r < rounds; ++r) {
    __syncthreads();  // readers of the previous round are done
    if (tid < ROWS) s_ln[tid] = make_float2((float)(tid + r), 1.f + (float)(r & 63) * (1.f / 64.f));
    if (tid >= ROWS && tid < ROWS + KQ) {
      const int q = tid - ROWS;
      s_lng[q] = make_float4(1.f + q, 2.f + q, 3.f + q, 4.f + q);
      s_lnb[q] = make_float4(0.5f * r, 0.25f * r, 0.125f * r, (float)q);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = a_r + 64 * i;
      float4 v = v0;
      const float2 st = s_ln[row];
      const float4 g = s_lng[a_c4], b = s_lnb[a_c4];
      float sx = st.x, sy = st.y;
      if constexpr (GUARD) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "+v"(sx), "+v"(sy)::"memory");
      v.x = fmaf(g.x, (v.x - sx) * sy, b.x); v.y = fmaf(g.y, (v.y - sx) * sy, b.y);
      v.z = fmaf(g.z, (v.z - sx) * sy, b.z); v.w = fmaf(g.w, (v.w - sx) * sy, b.w);
      // what the row SHOULD have been normalised with
      const float ex = (float)(row + r), ey = 1.f + (float)(r & 63) * (1.f / 64.f);
      const float wx = fmaf(g.x, (v0.x - ex) * ey, b.x);
      bad += (v.x != wx) ? 1ull : 0ull;
      float* d = sA + row * 33 + 4 * a_c4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      acc += v.y + v.w;
    }
  }
  if (bad) atomicAdd(bad_by_lane + (tid & 63), bad);
  if (acc == 123.456f) sink[0] = acc + sA[tid];  // (keeps the staging alive)
}
}  // namespace

extern "C" int lds_hazard_victim(const float* x, int rounds, int blocks, int guard, unsigned long long* bad_by_lane, float* sink, void* stream) {
  if (guard) hipLaunchKernelGGL(k_victim<true>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, x, rounds, bad_by_lane, sink);
  else hipLaunchKernelGGL(k_victim<false>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, x, rounds, bad_by_lane, sink);
  return (int)hipGetLastError();
}

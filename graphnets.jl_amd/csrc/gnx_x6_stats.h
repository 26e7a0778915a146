// Row statistics of a 128-wide row held as the B-fragment windows of the six-term kernels (k_edge_x6, k_ffn_x6): lane (n, hi) of a wave holds
// quads 4 s + 2 hi + e (s < 8, e < 2) of row n — computed in registers with EXACTLY the arithmetic and the association of k_ln_stats_v4<2>
// (gnx_generic.hip: lane `sub` of 16 adds quads sub and sub + 16, each as (x + y) + (z + w); then the 16-lane butterfly xor 1, xor 2, half
// mirror, mirror), so that gn1 / gn2 applied from them are bit-identical to the materialised LayerNorm and to the statistics kernel:
//     p[4 s + 2 hi + e] = Q(s, e) + Q(s + 4, e)            (s < 4: both quads sit in this lane)
//     t[s] = p[4 s + 2 hi] + p[4 s + 2 hi + 1]             (xor 1)
//     T[s] = t[s] + t'[s]                                  (xor 2: t' from the row's other lane, n + 32)
//     total = (T[0] + T[1]) + (T[2] + T[3])                (half mirror, mirror)
// The statistics pass over ef (512 MB at 1M edges: 87-129 us per core) is then not launched at all.
#pragma once

namespace gnx {

typedef float f32x4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float x6_tree(const float (&t)[4]) {
  float T[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) T[s] = t[s] + __shfl_xor(t[s], 32);
  return (T[0] + T[1]) + (T[2] + T[3]);
}

// raw[s][e]: quad 4 s + 2 hi + e of the lane's row.  -> (mean, 1 / (sigma + eps) or 1 / sqrt(var + eps))
__device__ __forceinline__ void x6_row_stats(const f32x4s (&raw)[8][2], float eps, int eps_mode, float& mu_o, float& inv_o) {
  float t[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float p[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float acc = 0.f;
      acc += (raw[s][e].x + raw[s][e].y) + (raw[s][e].z + raw[s][e].w);
      acc += (raw[s + 4][e].x + raw[s + 4][e].y) + (raw[s + 4][e].z + raw[s + 4][e].w);
      p[e] = acc;
    }
    t[s] = p[0] + p[1];
  }
  const float mu = x6_tree(t) * (1.f / 128.f);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float p[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float var = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4s q = raw[s + 4 * h][e];
        const float cx = q.x - mu, cy = q.y - mu, cz = q.z - mu, cw = q.w - mu;
        var = fmaf(cx, cx, var); var = fmaf(cy, cy, var); var = fmaf(cz, cz, var); var = fmaf(cw, cw, var);
      }
      p[e] = var;
    }
    t[s] = p[0] + p[1];
  }
  const float var = x6_tree(t) * (1.f / 128.f);
  mu_o = mu;
  inv_o = eps_mode == 0 ? 1.f / (sqrtf(var) + eps) : 1.f / sqrtf(var + eps);
}

// The same statistics for the 16 x 16 x 32 form of the kernels (round 6): lane (c, q), q = lane >> 4, holds floats 32 s + 8 q + 4 e .. + 3 of its row
// (s < 4, e < 2), i.e. quads i = 8 s + 2 q + e — four lanes (q = 0..3: lane ^ 16, lane ^ 32) per row.  The association of k_ln_stats_v4<2> again:
//     p[i] = Q(i) + Q(i + 16)                 (i < 16: s < 2; both quads sit in this lane: s and s + 2)
//     t[4 s + q] = p[8 s + 2 q] + p[8 s + 2 q + 1]              (xor 1: e = 0, 1, in the lane)
//     T[2 s + (q >> 1)] = t[4 s + q] + t[4 s + (q ^ 1)]          (xor 2: the lane 16 away)
//     total = (T[0] + T[1]) + (T[2] + T[3])                      (T[2 s], T[2 s + 1]: the lane 32 away; then s = 0, 1 in the lane)
// raw[s][e]: quad 8 s + 2 q + e of the lane's row.
__device__ __forceinline__ float x6_tree16(const float (&t)[2]) {
  float U[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float T = t[s] + __shfl_xor(t[s], 16);
    U[s] = T + __shfl_xor(T, 32);
  }
  return U[0] + U[1];
}

__device__ __forceinline__ void x6_row_stats16(const f32x4s (&raw)[4][2], float eps, int eps_mode, float& mu_o, float& inv_o) {
  float t[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float p[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float acc = 0.f;
      acc += (raw[s][e].x + raw[s][e].y) + (raw[s][e].z + raw[s][e].w);
      acc += (raw[s + 2][e].x + raw[s + 2][e].y) + (raw[s + 2][e].z + raw[s + 2][e].w);
      p[e] = acc;
    }
    t[s] = p[0] + p[1];
  }
  const float mu = x6_tree16(t) * (1.f / 128.f);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float p[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float var = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4s q = raw[s + 2 * h][e];
        const float cx = q.x - mu, cy = q.y - mu, cz = q.z - mu, cw = q.w - mu;
        var = fmaf(cx, cx, var); var = fmaf(cy, cy, var); var = fmaf(cz, cz, var); var = fmaf(cw, cw, var);
      }
      p[e] = var;
    }
    t[s] = p[0] + p[1];
  }
  const float var = x6_tree16(t) * (1.f / 128.f);
  mu_o = mu;
  inv_o = eps_mode == 0 ? 1.f / (sqrtf(var) + eps) : 1.f / sqrtf(var + eps);
}

}  // namespace gnx

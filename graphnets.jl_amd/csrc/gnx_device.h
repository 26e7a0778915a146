// Device-side argument block and helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "gnx_internal.h"

namespace gnx {

// Everything a block-forward kernel needs, passed by value (lives in SGPRs / kernarg segment).
struct BlockArgs {
  int de, dn, dg;  // effective input widths (0 <=> `nothing`)
  int oe, on, og;
  const float *We, *be, *Wn, *bn, *Wg, *bg;
  int act_e, act_n, act_g;
  const float *ef, *nf, *gf;  // replica 0
  float *ef_out, *nf_out, *gf_out;
  float* agg;       // workspace [R][N][oe]       (generic path)
  float* partials;  // workspace [R][n_tiles][oe+on]
  const int* colptr;
  const int* rowval;
  const int* node_off;
  const int* edge_off;
  const int* tile_off;
  const Tile* tiles;
  const int* wtile_off;
  const Tile* wtiles;
  int N, E, G, n_tiles, n_wtiles;
};

__device__ __forceinline__ float act_apply(float x, int act) {
  switch (act) {
    case GNX_ACT_RELU: return fmaxf(x, 0.f);
    case GNX_ACT_TANH: return tanhf(x);
    case GNX_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
    case GNX_ACT_GELU: return 0.5f * x * (1.f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
    default: return x;
  }
}

// index i in [0, n) with cp[i] <= e < cp[i+1] (cp non-decreasing, cp[0] <= e < cp[n]); skips empty segments.
__device__ __forceinline__ int segment_of(const int* cp, int n, int e) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (cp[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

}  // namespace gnx

// Device-side argument block and helpers shared by the kernel translation units.
// This header and gnx_wave_kernel.h are ALSO the source text of the run-time specialised kernels (gnx_jit.cpp compiles
// them with hiprtc for width sets outside the ahead-of-time list), so they depend on nothing but the HIP device
// runtime: no host headers, no fixed-width typedefs.
#pragma once
#ifndef GNX_JIT
#include <hip/hip_runtime.h>
#endif

namespace gnx {

// A work tile: a contiguous node range [n0, n1) of ONE graph and its (contiguous, dst-sorted) in-edges
// [e0, e1) = [colptr[n0], colptr[n1]).  Because the reference's edge order is CSC order
// (src/pad.jl:30), every edge->node sum is complete inside its tile: no atomics, fixed summation order.
struct Tile {
  int n0, n1;
  int e0, e1;
  int g;        // graph id
  int win0;     // [win0, win1): the graph's node range — every source of the tile's edges lies inside it
  int win1;
  int flags;    // wave tiles: number of wave tiles of graph g; else 0.  Keeps the record 32 B = one s_load_dwordx8.
};

// Everything a block-forward kernel needs, passed by value (lives in SGPRs / kernarg segment).
// destination rows of the projected edge update staged in LDS per 128-edge tile (k_rows_gemm<..., NL = 3>): 3 workgroups x (38.7 + 28 x 0.5) KB fit a CU's 160 KB
constexpr int kPdRowsCap = 28;

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
  // optional LayerNorm applied to the inputs as they are loaded (GNCore: block(gn1(x)) without materialising gn1(x)); fused
  // narrow path only.  ln_g[t] == nullptr <=> off.  t = 0 edges, 1 nodes, 2 graphs.
  const float* ln_g[3];
  const float* ln_b[3];
  float ln_eps;
  int ln_mode;
  // matrix-core path: row statistics [R][rows][2] (mean, inv) of ef / nf when THEY are to be normalised on load (then ln_g / ln_b
  // hold gamma / beta; gf arrives normalised); nullptr <=> the input is used as it is
  const float* ln_stats[3];
  int ln_inline_e;  // wide path: gn1 of the edge rows with NO statistics table (ln_stats[0] == nullptr) — k_edge_x6 computes them in registers (ln_eps / ln_mode below)
  // fused narrow path, batches of SMALL graphs: workgroups that own whole graphs (k_block_wave<..., PACK>).  packs[p][8] = the wave tiles
  // of pack p (-1: empty slot), the tiles of a graph adjacent — the graph update then runs inside the block kernel, from LDS
  const int* packs;
  int n_packs;
  // narrow GNCore, edges: the FeedForward and both residual terms in the block kernel's edge lanes (k_block_wave<..., FFE>):
  // ef_out receives y = (x + ef') + W2 act1(W1 gn2(x) + b1) + b2 instead of ef' (gncore.jl:56-59, gnfeedforward.jl:27-31); gn2 shares
  // x-hat with gn1 (ln_eps / ln_mode above).  ffe_w1 == nullptr <=> off.
  // Wide GNCore at 128-wide edges (ln_inline_e): the same fields ask for the edge form of k_ffn_x6 — edge update and edge FeedForward in one launch,
  // ef' never written; ffe_scratch: ffn_x6_scratch_bytes(128) for the prepared FeedForward weights.
  const float *ffe_w1, *ffe_b1, *ffe_w2, *ffe_b2, *ffe_g2, *ffe_be2;
  int ffe_act1, ffe_act2;
  void* ffe_scratch;
  // chained calls (gnx_block_forward_chained; k_block_wave<..., CHAIN>): the first prev_blocks workgroups of the launch finish the graph
  // update of the PREVIOUS call on this handle — its partial rows, its gf, its gf_out — with this call's graph function
  const float* prev_partials;
  const float* prev_gf;
  float* prev_gf_out;
  int prev_blocks;
};

// relu as ONE v_max_f32.  fmaxf(x, 0.f) compiles to two under the default IEEE mode: a canonicalising v_max_f32 x, x, x in front of the
// maximum whenever the compiler cannot see that x is the result of an arithmetic instruction (the packed FMAs of the streamed products are
// inline asm) — 125 of the 3 700 vector instructions of the narrow core kernel.  (NaN: v_max_f32 returns the other operand, as fmaxf does.)
__device__ __forceinline__ float relu_f(float x) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}

// activation codes = GNX_ACT_* of include/gnx.h (static_assert'ed in gnx_forward.hip)
__device__ __forceinline__ float act_apply(float x, int act) {
  switch (act) {
    case 1: return relu_f(x);
    case 2: return tanhf(x);
    case 3: return 1.f / (1.f + expf(-x));
    case 4: return 0.5f * x * (1.f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));  // NNlib.gelu (tanh form)
    default: return x;
  }
}

// derivative of the activation expressed through its OUTPUT y (what the backward pass has): relu, tanh, sigmoid, identity
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
  switch (act) {
    case 1: return y > 0.f ? 1.f : 0.f;
    case 2: return 1.f - y * y;
    case 3: return y * (1.f - y);
    default: return 1.f;
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

// One row's LayerNorm statistics (mean, 1 / sigma) from the table the statistics kernel wrote (round 6 tried it as a system-scope load: no
// difference to the hazard it chased — the table is read correctly; profiles/r06_overlap_hazard.log)
__device__ __forceinline__ float2 ld_stats(const float2* p) { return *p; }

}  // namespace gnx

#ifndef GNX_JIT
#include "gnx_internal.h"
#endif

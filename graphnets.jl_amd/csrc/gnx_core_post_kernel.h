// GNCore at narrow widths: FeedForward + residual kernels (see gnx_core_narrow.hip for the launchers).  Self-contained device code like
// gnx_wave_kernel.h: included by gnx_core_narrow.hip for the ahead-of-time instantiations AND part of the source text gnx_jit.cpp compiles
// at run time (k_core_post3 for width triples outside the ahead-of-time list), so it depends on gnx_device.h / gnx_wave_kernel.h only.
#pragma once
#include "gnx_device.h"
#include "gnx_wave_kernel.h"

#ifdef GNX_JIT
struct gnx_dense { const float* weight; const float* bias; int act; int reserved; };  // include/gnx.h (not visible to the run-time compiler)
#endif

namespace gnx {

// The FeedForward's 8 D^2 weights (800 at D = 10) do not fit the ~100 scalar registers: as SGPR operands they are streamed in
// groups, and hipcc's scheduler hoists the scalar loads until it spills (k_core_post<10> with scalar weights: 237 spilled SGPRs,
// i.e. a v_readlane in front of most FMAs, 126 VGPRs).  Here the workgroup stages the weights in LDS once and every lane reads
// them with UNIFORM addresses (ds_read_b128 of one address is a broadcast: no bank conflict) into a few VGPRs; M = 2 rows per
// thread share each read, which keeps the LDS at ~half its rate (D + 4*ceil(D/4)/... reads per 8 D M FMAs).
template <int D, int M>
__device__ __forceinline__ void core_post_lds_body(const float* __restrict__ x, size_t rows, const float* gamma2, const float* beta2, gnx_dense fc1,
                                                   gnx_dense fc2, float eps, int eps_mode, float* __restrict__ out, unsigned blk, unsigned nblk) {
  constexpr int H = 4 * D;
  constexpr int DP = (D + 3) / 4 * 4;  // padded row of W2 in LDS (16-B reads)
  __shared__ __attribute__((aligned(16))) float s_w1[D * H];   // W1 (4D x D column-major): element (j, k) at k*H + j
  __shared__ __attribute__((aligned(16))) float s_w2[H * DP];  // W2 (D x 4D column-major): element (k, j) at j*D + k -> row j padded to DP
  __shared__ __attribute__((aligned(16))) float s_b1[H];
  __shared__ float s_v[3 * D];                                 // b2 | gamma2 | beta2
  for (int i = threadIdx.x; i < D * H; i += 256) s_w1[i] = fc1.weight[i];
  for (int i = threadIdx.x; i < H * DP; i += 256) { const int jrow = i / DP, k = i % DP; s_w2[i] = k < D ? fc2.weight[jrow * D + k] : 0.f; }
  for (int i = threadIdx.x; i < H; i += 256) s_b1[i] = fc1.bias ? fc1.bias[i] : 0.f;
  for (int i = threadIdx.x; i < D; i += 256) { s_v[i] = fc2.bias ? fc2.bias[i] : 0.f; s_v[D + i] = gamma2[i]; s_v[2 * D + i] = beta2[i]; }
  __syncthreads();

  const size_t stride = (size_t)nblk * 256;
  const size_t row0 = (size_t)blk * 256 + threadIdx.x;
  if (row0 >= rows) return;
  float rs[M][D], z[M][D], acc[M][DP];
  size_t row[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const size_t rm = row0 + m * stride;
    row[m] = rm < rows ? rm : row0;  // clamped: a lane without an m-th row recomputes its first one (its store is skipped)
    float blk[D];
    load_row<D>(x + row[m] * D, z[m]);
    load_row<D>(out + row[m] * D, blk);  // block(gn1(x)) written by the block forward
#pragma unroll
    for (int k = 0; k < D; ++k) rs[m][k] = z[m][k] + blk[k];  // the two residual terms (gncore.jl:56-59)
    normalise<D>(z[m], eps, eps_mode);
#pragma unroll
    for (int k = 0; k < D; ++k) z[m][k] = fmaf(s_v[D + k], z[m][k], s_v[2 * D + k]);
#pragma unroll
    for (int k = 0; k < DP; ++k) acc[m][k] = k < D ? s_v[k] : 0.f;
  }
  // four hidden units at a time, produced and consumed in registers
#pragma unroll 1
  for (int j0 = 0; j0 < H; j0 += 4) {
    float h[M][4];
    const float4 bb = *reinterpret_cast<const float4*>(s_b1 + j0);
#pragma unroll
    for (int m = 0; m < M; ++m) { h[m][0] = bb.x; h[m][1] = bb.y; h[m][2] = bb.z; h[m][3] = bb.w; }
#pragma unroll
    for (int k = 0; k < D; ++k) {
      // (left alone the scheduler hoists every weight read of the iteration: 22 float4 = 88 VGPRs; other waves cover the LDS latency)
      if (k % 4 == 0) __builtin_amdgcn_sched_barrier(0);
      const float4 w = *reinterpret_cast<const float4*>(s_w1 + k * H + j0);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        h[m][0] = fmaf(w.x, z[m][k], h[m][0]); h[m][1] = fmaf(w.y, z[m][k], h[m][1]);
        h[m][2] = fmaf(w.z, z[m][k], h[m][2]); h[m][3] = fmaf(w.w, z[m][k], h[m][3]);
      }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) act_row<4>(h[m], fc1.act);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < DP / 4; ++q) {
        const float4 w = *reinterpret_cast<const float4*>(s_w2 + (j0 + jj) * DP + 4 * q);
#pragma unroll
        for (int m = 0; m < M; ++m) {
          acc[m][4 * q] = fmaf(w.x, h[m][jj], acc[m][4 * q]); acc[m][4 * q + 1] = fmaf(w.y, h[m][jj], acc[m][4 * q + 1]);
          acc[m][4 * q + 2] = fmaf(w.z, h[m][jj], acc[m][4 * q + 2]); acc[m][4 * q + 3] = fmaf(w.w, h[m][jj], acc[m][4 * q + 3]);
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < M; ++m) {
    float o[D];
#pragma unroll
    for (int k = 0; k < D; ++k) o[k] = acc[m][k];
    act_row<D>(o, fc2.act);
#pragma unroll
    for (int k = 0; k < D; ++k) o[k] = rs[m][k] + o[k];
    if (m == 0 || row0 + m * stride < rows) store_row<D>(out + row[m] * D, o);
  }
}


template <int D, int M>
__global__ __launch_bounds__(256) void k_core_post(const float* __restrict__ x, size_t rows, const float* gamma2, const float* beta2,
                                                   gnx_dense fc1, gnx_dense fc2, float eps, int eps_mode, float* __restrict__ out) {
  core_post_lds_body<D, M>(x, rows, gamma2, beta2, fc1, fc2, eps, eps_mode, out, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weights as SCALAR operands, streamed by hand.  Left to the compiler, scalar weight loads are hoisted until the SGPR file spills;
// staged in LDS (k_core_post above), every broadcast ds_read_b128 still moves 64 x 16 B and the LDS runs exactly as long as the
// VALU (22 reads per 176 FMAs at M = 2: 704 LDS clocks per 704 VALU clocks over four SIMDs).  Here a GROUP of N consecutive
// weights is fetched by s_load_dwordx{16,8,4,2} issued from inline asm — volatile asm statements keep their order, so the
// compiler can neither hoist nor merge them — into one of two register sets: the next group is in flight while the FMAs of the
// current one run (scalar loads return out of order, so the only wait is lgkmcnt(0), placed BEFORE the next issue), and the FMAs
// take the weight as their SGPR operand.  No LDS, no workgroup barrier, 4*D live weight registers.
// ---------------------------------------------------------------------------------------------------------------------------------
// (CorePostStream — the streamed FeedForward of two rows held as register pairs — lives in gnx_wave_kernel.h: k_block_wave<..., FFE> runs it in its edge lanes too)
template <int D, bool TRANS>
__device__ __forceinline__ void core_post_s_body(const float* __restrict__ x, size_t rows, const float* gamma2, const float* beta2, gnx_dense fc1,
                                                 gnx_dense fc2, float eps, int eps_mode, float* __restrict__ out, unsigned blk, unsigned nblk) {
  // A thread walks GNX_CORE_POST_UNITS units of two rows (rows row0 + m*stride, m = 2u, 2u+1).  With more than one unit, while a unit's FMAs
  // run the lines of the thread's NEXT unit are pulled into the L2 by one dword load per row and array whose destination is never read
  // (kept reserved until the next unit's own loads have returned: loads complete in order), so that the next unit starts from the L2
  // instead of from HBM.  Measured on the README ex.3 model: 39.6 us per k_core_post3 launch with two units against 38.1 us with one
  // (half as many, twice as long waves) — not taken, the default is one unit.
  constexpr int M = 2, U = GNX_CORE_POST_UNITS;
  const size_t stride = (size_t)nblk * 256;
  const size_t row0 = (size_t)blk * 256 + threadIdx.x;
  if (row0 >= rows) return;
  const cfloatp b2 = as_const(fc2.bias ? fc2.bias : k_zero_bias), g2 = as_const(gamma2), be2 = as_const(beta2);
  P2 z[D], acc[D];
  CorePostStream<D, TRANS> st{as_const(fc1.weight), as_const(fc2.weight), as_const(fc1.bias ? fc1.bias : k_zero_bias), fc1.act, z, acc};
  float touched = 0.f;
#pragma unroll 1
  for (int u = 0; u < U; ++u) {
    const size_t rbase = row0 + (size_t)(M * u) * stride;
    if (rbase >= rows) break;
    st.G0.issue(st.template group_ptr<0>());
    float rs[M][D];
    size_t row[M];
    float zr[M][D], blkr[M][D];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const size_t rm = rbase + m * stride;
      row[m] = rm < rows ? rm : rbase;  // clamped: a lane without a second row recomputes its first one (its store is skipped)
      load_row<D>(x + row[m] * D, zr[m]);
      load_row<D>(out + row[m] * D, blkr[m]);  // block(gn1(x)) written by the block forward
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
#pragma unroll
      for (int k = 0; k < D; ++k) rs[m][k] = zr[m][k] + blkr[m][k];  // the two residual terms (gncore.jl:56-59)
      normalise<D>(zr[m], eps, eps_mode);
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const float v = fmaf(g2[k], zr[m][k], be2[k]);
        if (m == 0) { z[k].x = v; acc[k].x = b2[k]; } else { z[k].y = v; acc[k].y = b2[k]; }
      }
    }
    pin_pairs<D>(z);
    pin_pairs<D>(acc);
    // (behind the pins: this unit's rows have arrived, so the touch loads are the only requests in flight under the FMAs and the
    // waits the compiler placed for the unit's own loads did not have to cover them)
    if (U > 1 && u + 1 < U) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const size_t rn = rbase + (size_t)(M + m) * stride;
        if (rn < rows) {
          asm volatile("global_load_dword %0, %1, off" : "+v"(touched) : "v"(x + rn * D));
          asm volatile("global_load_dword %0, %1, off" : "+v"(touched) : "v"(out + rn * D));
        }
      }
    }
    st.template run<0>();
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float o[D];
#pragma unroll
      for (int k = 0; k < D; ++k) o[k] = m == 0 ? acc[k].x : acc[k].y;
      if constexpr (TRANS) act_row<D>(o, fc2.act);
      else if (fc2.act == 1) {
#pragma unroll
        for (int k = 0; k < D; ++k) o[k] = relu_f(o[k]);
      }
#pragma unroll
      for (int k = 0; k < D; ++k) o[k] = rs[m][k] + o[k];
      if (m == 0 || rbase + m * stride < rows) store_row<D>(out + row[m] * D, o);
    }
  }
  if constexpr (U > 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(touched));  // the touch loads' destination stays reserved until they have landed
}
template <int D, bool TRANS>
__global__ __launch_bounds__(256) void k_core_post_s(const float* __restrict__ x, size_t rows, const float* gamma2, const float* beta2,
                                                     gnx_dense fc1, gnx_dense fc2, float eps, int eps_mode, float* __restrict__ out) {
  core_post_s_body<D, TRANS>(x, rows, gamma2, beta2, fc1, fc2, eps, eps_mode, out, blockIdx.x, gridDim.x);
}

// The three entities of a core in ONE launch (workgroup ranges: edges | nodes | graphs): the node and graph rows ride in the shadow of
// the edge rows instead of paying two more kernel boundaries (7.3 + 4.7 us of the README ex.3 core on the 1M-edge graph).  Edges and
// nodes: the streamed two-rows-per-thread body; graphs (few rows): the LDS body, one row per thread.
struct PostJob {
  const float* x; size_t rows; const float* gamma; const float* beta; gnx_dense fc1, fc2; float* out; unsigned blocks;
};
// GU: the block's graph update (graph_update_rows over the partial-sum rows k_block_wave left) runs HERE, in the graph job's workgroups
// (one per graph), followed by that graph's FeedForward + residual — the block is launched without its k_graph_t, whose 6.5 us then
// hide behind the edge rows.
template <int D0, int D1, int D2, bool GU>
__global__ __launch_bounds__(256) void k_core_post3(PostJob e, PostJob n, PostJob g, float eps, int eps_mode, BlockArgs a, int n_rows) {
  // workgroup ranges: graphs | edges | nodes — the graph job (a serial chain of a few microseconds) is dispatched FIRST, so that it runs
  // beside the edge rows instead of behind them
  const unsigned b = blockIdx.x;
  if (b >= g.blocks && b < g.blocks + e.blocks) core_post_s_body<D0, false>(e.x, e.rows, e.gamma, e.beta, e.fc1, e.fc2, eps, eps_mode, e.out, b - g.blocks, e.blocks);
  else if (b >= g.blocks) core_post_s_body<D1, false>(n.x, n.rows, n.gamma, n.beta, n.fc1, n.fc2, eps, eps_mode, n.out, b - g.blocks - e.blocks, n.blocks);
  else if constexpr (!GU) core_post_lds_body<D2, 1>(g.x, g.rows, g.gamma, g.beta, g.fc1, g.fc2, eps, eps_mode, g.out, b, g.blocks);
  else {
    constexpr int C = D0 + D1, CP = (C + 3) / 4 * 4;
    __shared__ float s_g[graph_update_lds_floats(C, D2, D2, 256)];
    const unsigned gb = b;
    const int gi = (int)(gb % (unsigned)a.G);
    const size_t r = gb / (unsigned)a.G;
    const bool oneg = a.G == 1;  // one row per workgroup of k_block_wave, else one per wave tile (gnx_narrow.hip: partial_rows)
    const int t0 = oneg ? 0 : a.wtile_off[gi], t1 = oneg ? (a.n_wtiles + 3) / 4 : a.wtile_off[gi + 1];
    graph_update_rows<C, false, 16>(a, a.partials + r * (size_t)n_rows * CP, gi, r, t0, t1, (int)threadIdx.x, 256, s_g);
    __syncthreads();  // gf' of this graph is in memory (written by this workgroup): the FeedForward below reads it as the block's output
    const size_t row = r * (size_t)a.G + gi;
    core_post_lds_body<D2, 1>(g.x + row * D2, 1, g.gamma, g.beta, g.fc1, g.fc2, eps, eps_mode, g.out + row * D2, 0, 1);
  }
}

}  // namespace gnx

// Dimension-generic kernels (any DE/DN/DG/DE'/DN'/DG', any activation): the always-available HIP path and the
// cross-check for the specialised kernels.  Lanes run along the feature dimension so that global reads/writes of
// feature rows are coalesced; every reduction has a fixed order (no atomics) so results are bitwise reproducible.
//
// Reference semantics: src/edgefninput.jl:1-47, src/nodefninput.jl:1-24, src/graphfninput.jl:1-13,
// src/gnblock.jl:63-69, src/gngraphnorm.jl:19-26, src/gnfeedforward.jl:27-40, src/gncore.jl:56-68.
#include <algorithm>

#include "gnx_device.h"

namespace gnx {

// ---------------------------------------------------------------------------------------------------------
// edge update: ef'[e] = act(We * [ef_e ; nf_src ; nf_dst ; gf_g] + be)          one workgroup per (tile, replica)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_edge_generic(BlockArgs a) {
  extern __shared__ int s_cp[];
  const Tile t = a.tiles[blockIdx.x];
  const size_t r = blockIdx.y;
  const int nn = t.n1 - t.n0;
  for (int i = threadIdx.x; i <= nn; i += blockDim.x) s_cp[i] = a.colptr[t.n0 + i];
  __syncthreads();
  const float* ef = a.ef ? a.ef + r * (size_t)a.E * a.de : nullptr;
  const float* nf = a.nf ? a.nf + r * (size_t)a.N * a.dn : nullptr;
  const float* gf = a.gf ? a.gf + (r * (size_t)a.G + t.g) * a.dg : nullptr;
  float* out = a.ef_out + r * (size_t)a.E * a.oe;
  const int oe = a.oe;
  const int total = (t.e1 - t.e0) * oe;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int el = idx / oe, j = idx - el * oe;
    const int e = t.e0 + el;
    float acc = a.be ? a.be[j] : 0.f;
    const float* w = a.We + j;
    if (a.de) {
      const float* x = ef + (size_t)e * a.de;
      for (int k = 0; k < a.de; ++k, w += oe) acc = fmaf(*w, x[k], acc);
    }
    if (a.dn) {
      const float* xs = nf + (size_t)a.rowval[e] * a.dn;
      for (int k = 0; k < a.dn; ++k, w += oe) acc = fmaf(*w, xs[k], acc);
      const float* xd = nf + (size_t)(t.n0 + segment_of(s_cp, nn, e)) * a.dn;
      for (int k = 0; k < a.dn; ++k, w += oe) acc = fmaf(*w, xd[k], acc);
    }
    for (int k = 0; k < a.dg; ++k, w += oe) acc = fmaf(*w, gf[k], acc);
    out[(size_t)e * oe + j] = act_apply(acc, a.act_e);
  }
}

// ---------------------------------------------------------------------------------------------------------
// node update: agg[n] = sum_{e->n} ef'[e];  nf'[n] = act(Wn * [agg_n ; nf_n ; gf_g] + bn);  per-tile partial sums
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_node_generic(BlockArgs a) {
  const Tile t = a.tiles[blockIdx.x];
  const size_t r = blockIdx.y;
  const int nn = t.n1 - t.n0;
  const int oe = a.oe, on = a.on;
  const float* efo = a.ef_out + r * (size_t)a.E * oe;
  float* agg = a.agg + (r * (size_t)a.N + t.n0) * oe;
  // phase 1: contiguous segmented sums (edges are dst-sorted, src/pad.jl:30)
  for (int idx = threadIdx.x; idx < nn * oe; idx += blockDim.x) {
    const int n = idx / oe, k = idx - n * oe;
    float s = 0.f;
    const int e1 = a.colptr[t.n0 + n + 1];
    for (int e = a.colptr[t.n0 + n]; e < e1; ++e) s += efo[(size_t)e * oe + k];
    agg[(size_t)n * oe + k] = s;
  }
  __syncthreads();
  const float* nf = a.nf ? a.nf + (r * (size_t)a.N + t.n0) * a.dn : nullptr;
  const float* gf = a.gf ? a.gf + (r * (size_t)a.G + t.g) * a.dg : nullptr;
  float* nfo = a.nf_out + (r * (size_t)a.N + t.n0) * on;
  for (int idx = threadIdx.x; idx < nn * on; idx += blockDim.x) {
    const int n = idx / on, j = idx - n * on;
    float acc = a.bn ? a.bn[j] : 0.f;
    const float* w = a.Wn + j;
    const float* x = agg + (size_t)n * oe;
    for (int k = 0; k < oe; ++k, w += on) acc = fmaf(*w, x[k], acc);
    if (a.dn) {
      const float* xn = nf + (size_t)n * a.dn;
      for (int k = 0; k < a.dn; ++k, w += on) acc = fmaf(*w, xn[k], acc);
    }
    for (int k = 0; k < a.dg; ++k, w += on) acc = fmaf(*w, gf[k], acc);
    nfo[(size_t)n * on + j] = act_apply(acc, a.act_n);
  }
  if (a.og == 0) return;
  __syncthreads();
  // phase 3: this tile's contribution to the graph-level sums, fixed order over the tile's nodes
  float* part = a.partials + (r * (size_t)a.n_tiles + blockIdx.x) * (oe + on);
  for (int c = threadIdx.x; c < oe + on; c += blockDim.x) {
    float s = 0.f;
    if (c < oe) {
      for (int n = 0; n < nn; ++n) s += agg[(size_t)n * oe + c];
    } else {
      for (int n = 0; n < nn; ++n) s += nfo[(size_t)n * on + (c - oe)];
    }
    part[c] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------
// graph update: gf'[g] = act(Wg * [sum_e ef' ; sum_n nf' ; gf_g] + bg)         one workgroup per (graph, replica)
// ---------------------------------------------------------------------------------------------------------
constexpr int kGraphThreads = 1024;
__global__ __launch_bounds__(kGraphThreads) void k_graph(BlockArgs a) {
  extern __shared__ float s_f[];
  const int g = blockIdx.x;
  const size_t r = blockIdx.y;
  const int C = a.oe + a.on;
  const int t0 = a.tile_off[g], t1 = a.tile_off[g + 1];
  const int cpp = C < kGraphThreads ? C : kGraphThreads;  // columns per pass
  const int nsl = cpp > 0 ? kGraphThreads / cpp : 1;      // tile slices summed in parallel
  float* s_part = s_f;                         // [nsl][C]
  float* s_x = s_f + (size_t)nsl * C;          // [C + dg]
  const float* part = a.partials + r * (size_t)a.n_tiles * C;
  if (cpp > 0) {
    const int sl = threadIdx.x / cpp, cc = threadIdx.x - sl * cpp;
    if (sl < nsl) {
      for (int c = cc; c < C; c += cpp) {
        // 16 independent chains (the loads are what costs: keep 16 in flight), combined in a fixed order
        float s16[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) s16[u] = 0.f;
        // every pass issues 16 predicated loads at once (no serial tail: a tail loop would pay one L2/HBM round trip
        // per row, which is what made this kernel 8 us for 2k tiles)
        for (int ti = t0 + sl; ti < t1; ti += 16 * nsl) {
          float v16[16];  // unconditional (clamped) loads: all 16 in flight; duplicates discarded by the select
#pragma unroll
          for (int u = 0; u < 16; ++u) v16[u] = part[(size_t)min(ti + u * nsl, t1 - 1) * C + c];
#pragma unroll
          for (int u = 0; u < 16; ++u) s16[u] += ti + u * nsl < t1 ? v16[u] : 0.f;
        }
#pragma unroll
        for (int w = 8; w > 0; w >>= 1)
#pragma unroll
          for (int u = 0; u < w; ++u) s16[u] += s16[u + w];
        const float s0 = s16[0], s1 = 0.f, s2 = 0.f, s3 = 0.f;
        s_part[(size_t)sl * C + c] = (s0 + s1) + (s2 + s3);
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int sl = 0; sl < nsl; ++sl) s += s_part[(size_t)sl * C + c];
    s_x[c] = s;
  }
  const float* gf = a.gf ? a.gf + (r * (size_t)a.G + g) * a.dg : nullptr;
  for (int k = threadIdx.x; k < a.dg; k += blockDim.x) s_x[C + k] = gf[k];
  __syncthreads();
  const int K = C + a.dg, og = a.og;
  float* out = a.gf_out + (r * (size_t)a.G + g) * og;
  for (int j = threadIdx.x; j < og; j += blockDim.x) {
    float acc = a.bg ? a.bg[j] : 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(a.Wg[(size_t)k * og + j], s_x[k], acc);
    out[j] = act_apply(acc, a.act_g);
  }
}

size_t graph_kernel_lds_bytes(int oe, int on, int dg) {
  const int C = oe + on;
  const int cpp = C < kGraphThreads ? C : kGraphThreads;
  const int nsl = cpp > 0 ? kGraphThreads / cpp : 1;
  return sizeof(float) * ((size_t)nsl * C + C + dg + 1);
}

int32_t launch_graph(const BlockArgs& a, int64_t R, hipStream_t s) {
  if (a.og == 0) return GNX_OK;
  ProfScope ps("k_graph", s);
  GNX_LAUNCH(k_graph, dim3((unsigned)a.G, (unsigned)R), dim3(kGraphThreads), graph_kernel_lds_bytes(a.oe, a.on, a.dg), s, a);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

int32_t launch_block_generic(const BlockArgs& a, int64_t R, int tile_n_cap, hipStream_t s, int phase) {
  const dim3 grid((unsigned)a.n_tiles, (unsigned)R);
  if ((phase & 1) && a.oe > 0 && a.E > 0) {
    ProfScope ps("k_edge_generic", s);
    GNX_LAUNCH(k_edge_generic, grid, dim3(256), sizeof(int) * (size_t)(tile_n_cap + 1), s, a);
    GNX_HIP(hipGetLastError());
  }
  if (phase & 1) {
    ProfScope ps("k_node_generic", s);
    GNX_LAUNCH(k_node_generic, grid, dim3(256), 0, s, a);
    GNX_HIP(hipGetLastError());
  }
  return (phase & 2) ? launch_graph(a, R, s) : GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// GNCore pieces (generic): two LayerNorms sharing statistics, and FFN + residual
// ---------------------------------------------------------------------------------------------------------
// One wave per row.  y1 = g1*xhat+b1, y2 = g2*xhat+b2 (gn1 and gn2 normalise the same x, gncore.jl:56-59).
__global__ __launch_bounds__(256) void k_layernorm2(const float* __restrict__ x, size_t rows, int d, const float* g1,
                                                    const float* b1, const float* g2, const float* b2, float eps,
                                                    int eps_mode, float* __restrict__ y1, float* __restrict__ y2) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * d;
  float s = 0.f;
  for (int k = lane; k < d; k += 64) s += xr[k];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mu = s / (float)d;
  float v = 0.f;
  for (int k = lane; k < d; k += 64) { const float c = xr[k] - mu; v = fmaf(c, c, v); }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  v /= (float)d;
  const float inv = eps_mode == 0 ? 1.f / (sqrtf(v) + eps) : 1.f / sqrtf(v + eps);
  for (int k = lane; k < d; k += 64) {
    const float xh = (xr[k] - mu) * inv;
    y1[row * d + k] = fmaf(g1[k], xh, b1[k]);
    y2[row * d + k] = fmaf(g2[k], xh, b2[k]);
  }
}

// sum over the 16 lanes of a DPP row (result in every lane of the row), fixed order
__device__ __forceinline__ float row16_sum_g(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// Widths that are multiples of 64 (GNCore at 128/64, the sort example's 384): 16 lanes per row, every lane keeps its
// Q = d/64 float4 of the row in registers (x is read ONCE, 16-B accesses), statistics by DPP row reductions (a DPP row
// is exactly the 16 lanes of one LayerNorm row), both outputs written from registers.  4 rows per wave, 16 per block.
template <int Q>
__global__ __launch_bounds__(256) void k_layernorm2_v4(const float* __restrict__ x, size_t rows, const float* __restrict__ g1,
                                                       const float* __restrict__ b1, const float* __restrict__ g2,
                                                       const float* __restrict__ b2, float eps, int eps_mode,
                                                       float* __restrict__ y1, float* __restrict__ y2) {
  constexpr int D = 64 * Q;
  const int sub = threadIdx.x & 15;
  size_t row = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = row < rows;
  row = live ? row : rows - 1;  // clamped: every lane takes part in the DPP reductions
  const float4* xr = reinterpret_cast<const float4*>(x + row * D);
  float4 v[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) v[q] = xr[sub + 16 * q];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < Q; ++q) s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
  const float mu = row16_sum_g(s) * (1.f / (float)D);
  float var = 0.f;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    v[q].x -= mu; v[q].y -= mu; v[q].z -= mu; v[q].w -= mu;
    var = fmaf(v[q].x, v[q].x, var); var = fmaf(v[q].y, v[q].y, var); var = fmaf(v[q].z, v[q].z, var); var = fmaf(v[q].w, v[q].w, var);
  }
  var = row16_sum_g(var) * (1.f / (float)D);
  const float inv = eps_mode == 0 ? 1.f / (sqrtf(var) + eps) : 1.f / sqrtf(var + eps);
  if (!live) return;
  float4* o1 = reinterpret_cast<float4*>(y1 + row * D);
  float4* o2 = reinterpret_cast<float4*>(y2 + row * D);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int c = sub + 16 * q;
    const float4 ga = reinterpret_cast<const float4*>(g1)[c], ba = reinterpret_cast<const float4*>(b1)[c];
    const float4 gb = reinterpret_cast<const float4*>(g2)[c], bb = reinterpret_cast<const float4*>(b2)[c];
    const float4 xh = make_float4(v[q].x * inv, v[q].y * inv, v[q].z * inv, v[q].w * inv);
    o1[c] = make_float4(fmaf(ga.x, xh.x, ba.x), fmaf(ga.y, xh.y, ba.y), fmaf(ga.z, xh.z, ba.z), fmaf(ga.w, xh.w, ba.w));
    o2[c] = make_float4(fmaf(gb.x, xh.x, bb.x), fmaf(gb.y, xh.y, bb.y), fmaf(gb.z, xh.z, bb.z), fmaf(gb.w, xh.w, bb.w));
  }
}

// Row statistics only — stats[row] = (mean, 1/(sigma + eps) or 1/sqrt(var + eps)) with exactly the arithmetic of k_layernorm2_v4: the
// matrix-core kernels of a wide GNCore normalise x as they LOAD it ((x - mean) * inv, then fma(gamma, ., beta): bit-identical to the
// materialised gn1(x) / gn2(x)), so neither LayerNorm output is written to or read back from HBM (gncore.jl:56-59).
template <int Q>
__global__ __launch_bounds__(256) void k_ln_stats_v4(const float* __restrict__ x, size_t rows, float eps, int eps_mode, float2* __restrict__ stats) {
  constexpr int D = 64 * Q;
  const int sub = threadIdx.x & 15;
  size_t row = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = row < rows;
  row = live ? row : rows - 1;  // clamped: every lane takes part in the DPP reductions
  const float4* xr = reinterpret_cast<const float4*>(x + row * D);
  float4 v[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) v[q] = xr[sub + 16 * q];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < Q; ++q) s += (v[q].x + v[q].y) + (v[q].z + v[q].w);
  const float mu = row16_sum_g(s) * (1.f / (float)D);
  float var = 0.f;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    v[q].x -= mu; v[q].y -= mu; v[q].z -= mu; v[q].w -= mu;
    var = fmaf(v[q].x, v[q].x, var); var = fmaf(v[q].y, v[q].y, var); var = fmaf(v[q].z, v[q].z, var); var = fmaf(v[q].w, v[q].w, var);
  }
  var = row16_sum_g(var) * (1.f / (float)D);
  const float inv = eps_mode == 0 ? 1.f / (sqrtf(var) + eps) : 1.f / sqrtf(var + eps);
  if (live && sub == 0) stats[row] = make_float2(mu, inv);  // (written whole, through lanes 0..3 or behind a fence: no difference to what round 6 chased — profiles/r06_overlap_hazard.log)
}

// 1: this width / alignment is not covered (the caller materialises the LayerNorm outputs instead)
bool ln_stats_applies(const float* x, int d) { return ((uintptr_t)x & 15) == 0 && d % 64 == 0 && d <= 512; }

int32_t launch_ln_stats(const float* x, size_t rows, int d, float eps, int eps_mode, float* stats, hipStream_t s) {
  if (rows == 0) return GNX_OK;
  if (!ln_stats_applies(x, d) || ((uintptr_t)stats & 7)) return fail(GNX_ERR_INVALID_ARG, "launch_ln_stats: width / alignment not covered");
  ProfScope ps("k_ln_stats", s);
  const dim3 grid((unsigned)((rows + 15) / 16));
  switch (d / 64) {
#define GNX_LN_CASE(Q) case Q: GNX_LAUNCH((k_ln_stats_v4<Q>), grid, dim3(256), 0, s, x, rows, eps, eps_mode, reinterpret_cast<float2*>(stats)); break;
    GNX_LN_CASE(1) GNX_LN_CASE(2) GNX_LN_CASE(3) GNX_LN_CASE(4) GNX_LN_CASE(5) GNX_LN_CASE(6) GNX_LN_CASE(7) GNX_LN_CASE(8)
#undef GNX_LN_CASE
  }
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

// One wave per row: out[row] += x[row] + W2*relu(W1*z[row]+b1)+b2  (z = LN2(x); out already holds block(LN1 x)).
__global__ __launch_bounds__(256) void k_ffn_residual(const float* __restrict__ z, const float* __restrict__ x,
                                                      size_t rows, int d, gnx_dense fc1, gnx_dense fc2,
                                                      float* __restrict__ out) {
  extern __shared__ float s_f[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t row = (size_t)blockIdx.x * 4 + wv;
  if (row >= rows) return;
  const int h = 4 * d;
  float* s_z = s_f + (size_t)wv * (d + h);
  float* s_h = s_z + d;
  for (int k = lane; k < d; k += 64) s_z[k] = z[row * d + k];
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < h; j += 64) {
    float acc = fc1.bias ? fc1.bias[j] : 0.f;
    for (int k = 0; k < d; ++k) acc = fmaf(fc1.weight[(size_t)k * h + j], s_z[k], acc);
    s_h[j] = act_apply(acc, fc1.act);
  }
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < d; j += 64) {
    float acc = fc2.bias ? fc2.bias[j] : 0.f;
    for (int k = 0; k < h; ++k) acc = fmaf(fc2.weight[(size_t)k * d + j], s_h[k], acc);
    out[row * d + j] += x[row * d + j] + act_apply(acc, fc2.act);
  }
}

int32_t launch_layernorm2(const float* x, size_t rows, int d, const gnx_layernorm& l1, const gnx_layernorm& l2, float eps,
                          int eps_mode, float* y1, float* y2, hipStream_t s) {
  if (rows == 0 || d == 0) return GNX_OK;
  ProfScope ps("k_layernorm2", s);
  const bool al16 = (((uintptr_t)x | (uintptr_t)y1 | (uintptr_t)y2 | (uintptr_t)l1.gamma | (uintptr_t)l1.beta | (uintptr_t)l2.gamma | (uintptr_t)l2.beta) & 15) == 0;
  if (al16 && d % 64 == 0 && d <= 512) {
    const dim3 grid((unsigned)((rows + 15) / 16));
    switch (d / 64) {
#define GNX_LN_CASE(Q) case Q: GNX_LAUNCH((k_layernorm2_v4<Q>), grid, dim3(256), 0, s, x, rows, l1.gamma, l1.beta, l2.gamma, l2.beta, eps, eps_mode, y1, y2); break;
      GNX_LN_CASE(1) GNX_LN_CASE(2) GNX_LN_CASE(3) GNX_LN_CASE(4) GNX_LN_CASE(5) GNX_LN_CASE(6) GNX_LN_CASE(7) GNX_LN_CASE(8)
#undef GNX_LN_CASE
    }
    GNX_HIP(hipGetLastError());
    return GNX_OK;
  }
  GNX_LAUNCH(k_layernorm2, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, rows, d, l1.gamma, l1.beta, l2.gamma,
                     l2.beta, eps, eps_mode, y1, y2);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

int32_t launch_ffn_residual(const float* z, const float* x, size_t rows, int d, const gnx_ffn& ff, float* out, hipStream_t s) {
  if (rows == 0 || d == 0) return GNX_OK;
  ProfScope ps("k_ffn_residual", s);
  GNX_LAUNCH(k_ffn_residual, dim3((unsigned)((rows + 3) / 4)), dim3(256), sizeof(float) * 4 * (size_t)(5 * d), s, z, x, rows,
                     d, ff.fc1, ff.fc2, out);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// reference-layout bridges (src/pad.jl:12-64, src/unpad.jl:1-17)
// ---------------------------------------------------------------------------------------------------------
// edges: slot = src_local + PN * dst_local inside graph g's PN^2 grid (column-major, adjacency padded to PN).
template <bool PAD>
__global__ void k_pad_edges(const int* colptr, const int* rowval, const int* node_off, int N, int E, int G, int PN, int d,
                            int64_t R, const float* src, float* dst) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)R * E * d;
  if (idx >= total) return;
  const int k = (int)(idx % d);
  const size_t re = idx / d;
  const int e = (int)(re % E);
  const size_t r = re / E;
  const int n = segment_of(colptr, N, e);
  const int g = segment_of(node_off, G, n);
  const int base = node_off[g];
  const size_t slot = (size_t)(rowval[e] - base) + (size_t)PN * (n - base);
  const size_t b = G == 1 ? r : (size_t)g;
  const size_t pidx = (b * PN * PN + slot) * d + k;
  if (PAD) dst[pidx] = src[idx]; else dst[idx] = src[pidx];
}

template <bool PAD>
__global__ void k_pad_nodes(const int* node_off, int N, int G, int PN, int d, int64_t R, const float* src, float* dst) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)R * N * d;
  if (idx >= total) return;
  const int k = (int)(idx % d);
  const size_t rn = idx / d;
  const int n = (int)(rn % N);
  const size_t r = rn / N;
  const int g = segment_of(node_off, G, n);
  const size_t b = G == 1 ? r : (size_t)g;
  const size_t pidx = (b * PN + (n - node_off[g])) * d + k;
  if (PAD) dst[pidx] = src[idx]; else dst[idx] = src[pidx];
}

// ---------------------------------------------------------------------------------------------------------
// materialised update-function inputs (the reference's exported building blocks)
// ---------------------------------------------------------------------------------------------------------
struct FnInArgs {
  const float *ef, *nf, *gf;
  int de, dn, dg, N, E, G;
  const int *colptr, *rowval, *edge_dst, *node_off, *edge_off;
  float* out;
};

// one thread per output element; lanes run along the concatenated feature dim (coalesced row writes)
__global__ void k_fn_input_edge(FnInArgs a) {
  const int K = a.de + 2 * a.dn + a.dg;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t r = blockIdx.y;
  if (idx >= (size_t)a.E * K) return;
  const int e = (int)(idx / K), k = (int)(idx % K);
  float v;
  if (k < a.de) v = a.ef[(r * a.E + e) * a.de + k];
  else if (k < a.de + a.dn) v = a.nf[(r * a.N + a.rowval[e]) * a.dn + (k - a.de)];
  else if (k < a.de + 2 * a.dn) v = a.nf[(r * a.N + a.edge_dst[e]) * a.dn + (k - a.de - a.dn)];
  else v = a.gf[(r * a.G + segment_of(a.edge_off, a.G, e)) * a.dg + (k - a.de - 2 * a.dn)];
  a.out[r * (size_t)a.E * K + idx] = v;
}

__global__ void k_fn_input_node(FnInArgs a) {
  const int K = a.de + a.dn + a.dg;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t r = blockIdx.y;
  if (idx >= (size_t)a.N * K) return;
  const int n = (int)(idx / K), k = (int)(idx % K);
  float v = 0.f;
  if (k < a.de) {
    for (int e = a.colptr[n]; e < a.colptr[n + 1]; ++e) v += a.ef[(r * a.E + e) * a.de + k];  // CSC order
  } else if (k < a.de + a.dn) {
    v = a.nf[(r * a.N + n) * a.dn + (k - a.de)];
  } else {
    v = a.gf[(r * a.G + segment_of(a.node_off, a.G, n)) * a.dg + (k - a.de - a.dn)];
  }
  a.out[r * (size_t)a.N * K + idx] = v;
}

// one workgroup per (graph, replica); every column is summed by the whole workgroup in a fixed order
__global__ __launch_bounds__(256) void k_fn_input_graph(FnInArgs a) {
  __shared__ float s_red[256];
  const int K = a.de + a.dn + a.dg, g = blockIdx.x, tid = threadIdx.x;
  const size_t r = blockIdx.y;
  float* out = a.out + (r * a.G + g) * (size_t)K;
  for (int k = 0; k < a.de + a.dn; ++k) {
    const bool edge = k < a.de;
    const int t0 = edge ? a.edge_off[g] : a.node_off[g], t1 = edge ? a.edge_off[g + 1] : a.node_off[g + 1];
    const float* base = edge ? a.ef + r * (size_t)a.E * a.de + k : a.nf + r * (size_t)a.N * a.dn + (k - a.de);
    const int stride = edge ? a.de : a.dn;
    float s = 0.f;
    for (int t = t0 + tid; t < t1; t += 256) s += base[(size_t)t * stride];
    s_red[tid] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (tid < w) s_red[tid] += s_red[tid + w];
      __syncthreads();
    }
    if (tid == 0) out[k] = s_red[0];
    __syncthreads();
  }
  for (int k = tid; k < a.dg; k += 256) out[a.de + a.dn + k] = a.gf[(r * a.G + g) * a.dg + k];
}

int32_t launch_fn_input(const gnx_graphs* h, int kind, const float* ef, int de, const float* nf, int dn, const float* gf, int dg,
                        int64_t R, float* out, hipStream_t s) {
  if (kind == 0) { if (int32_t rcw = gnx_ensure_wide_tables(h, s)) return rcw; }  // (the edge form reads the destination of every edge)
  FnInArgs a{ef, nf, gf, de, dn, dg, (int)h->N, (int)h->E, (int)h->G, h->d_colptr, h->d_rowval, h->d_edge_dst, h->d_node_off, h->d_edge_off, out};
  if (kind == 0) {
    const size_t total = (size_t)h->E * (de + 2 * dn + dg);
    if (total) GNX_LAUNCH(k_fn_input_edge, dim3((unsigned)((total + 255) / 256), (unsigned)R), dim3(256), 0, s, a);
  } else if (kind == 1) {
    const size_t total = (size_t)h->N * (de + dn + dg);
    if (total) GNX_LAUNCH(k_fn_input_node, dim3((unsigned)((total + 255) / 256), (unsigned)R), dim3(256), 0, s, a);
  } else {
    GNX_LAUNCH(k_fn_input_graph, dim3((unsigned)h->G, (unsigned)R), dim3(256), 0, s, a);
  }
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

// out[r][c][k] = (ef[r][edge[c]][k] + ef[r][rev[c]][k]) / 2   (collapsef, gngraphbatch.jl:83-85, on real edges only)
__global__ void k_collapse(const int* __restrict__ edge, const int* __restrict__ rev, int n, int d, int E, const float* __restrict__ ef,
                           float* __restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t r = blockIdx.y;
  if (idx >= (size_t)n * d) return;
  const int c = (int)(idx / d), k = (int)(idx % d);
  const float* base = ef + r * (size_t)E * d;
  const int rv = rev[c];
  const float a = base[(size_t)edge[c] * d + k];
  const float b = rv >= 0 ? base[(size_t)rv * d + k] : 0.f;
  out[r * (size_t)n * d + idx] = (a + b) / 2.f;
}

int32_t launch_collapse(const gnx_graphs* h, const float* ef, int d, int64_t R, float* out, hipStream_t s) {
  const size_t n = (size_t)h->h_collapse_off.back();
  if (n == 0) return GNX_OK;
  GNX_LAUNCH(k_collapse, dim3((unsigned)((n * d + 255) / 256), (unsigned)R), dim3(256), 0, s, h->d_collapse_edge, h->d_collapse_rev,
                     (int)n, d, (int)h->E, ef, out);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

// collapsef, the padded array form (gngraphbatch.jl:83-85 with the edge_collapser of :67-82): for every coordinate (i, j), i >= j, of
// the PN x PN lower triangle, in column-major order (l = j*PN - j(j-1)/2 + (i - j)), out[b][l][:] = (P[i->j] + P[j->i]) / 2 where P is
// the zero-padded edge grid (a self loop gives P[i->i]).  Edge i->j of a graph is found by binary search in column j of its CSC.
__device__ __forceinline__ int find_edge(const int* __restrict__ colptr, const int* __restrict__ rowval, int src, int dst) {
  int lo = colptr[dst], hi = colptr[dst + 1];
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (rowval[mid] < src) lo = mid + 1; else hi = mid;
  }
  return lo < colptr[dst + 1] && rowval[lo] == src ? lo : -1;
}
__global__ void k_collapse_padded(const int* __restrict__ colptr, const int* __restrict__ rowval, const int* __restrict__ node_off, int G, int PN,
                                  int shared, int d, int E, const float* __restrict__ ef, float* __restrict__ out) {
  const size_t L = (size_t)PN * (PN + 1) / 2;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t b = blockIdx.y;  // graph of the batch, or replica of the shared graph
  if (idx >= L * d) return;
  const size_t l = idx / d;
  const int k = (int)(idx % d);
  // column j of the triangle: the largest j with j*PN - j(j-1)/2 <= l
  int j = (int)(((2.0 * PN + 1.0) - sqrt((2.0 * PN + 1.0) * (2.0 * PN + 1.0) - 8.0 * (double)l)) / 2.0);
  j = j < 0 ? 0 : (j > PN - 1 ? PN - 1 : j);
  while (j > 0 && (size_t)j * PN - (size_t)j * (j - 1) / 2 > l) --j;
  while (j + 1 < PN && (size_t)(j + 1) * PN - (size_t)(j + 1) * j / 2 <= l) ++j;
  const int i = j + (int)(l - ((size_t)j * PN - (size_t)j * (j - 1) / 2));
  const int g = shared ? 0 : (int)b;
  const int n0 = node_off[g], nn = node_off[g + 1] - n0;
  const float* base = ef + (shared ? b * (size_t)E * d : 0);
  float v = 0.f;
  if (i < nn && j < nn) {
    const int e1 = find_edge(colptr, rowval, n0 + i, n0 + j);
    if (i == j) {
      if (e1 >= 0) v = base[(size_t)e1 * d + k];
    } else {
      const int e2 = find_edge(colptr, rowval, n0 + j, n0 + i);
      v = ((e1 >= 0 ? base[(size_t)e1 * d + k] : 0.f) + (e2 >= 0 ? base[(size_t)e2 * d + k] : 0.f)) / 2.f;
    }
  }
  out[b * L * d + idx] = v;
}

int32_t launch_collapse_padded(const gnx_graphs* h, const float* ef, int d, int64_t R, float* out, hipStream_t s) {
  const size_t L = (size_t)h->PN * (h->PN + 1) / 2;
  const bool shared = R > 1;
  const size_t B = shared ? (size_t)R : (size_t)h->G;
  if (L == 0 || B == 0) return GNX_OK;
  GNX_LAUNCH(k_collapse_padded, dim3((unsigned)((L * d + 255) / 256), (unsigned)B), dim3(256), 0, s, h->d_colptr, h->d_rowval, h->d_node_off,
                     (int)h->G, (int)h->PN, shared ? 1 : 0, d, (int)h->E, ef, out);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// readout: logitcrossentropy over packed columns (examples/sort/sort.jl:69-81)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_xent_partial(const float* __restrict__ logits, const float* __restrict__ targets, int d,
                                                      size_t cols, float* __restrict__ partial) {
  __shared__ float s_red[256];
  float acc = 0.f;
  for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < cols; c += (size_t)gridDim.x * 256) {
    const float* x = logits + c * d;
    const float* y = targets + c * d;
    float mx = x[0];
    for (int k = 1; k < d; ++k) mx = fmaxf(mx, x[k]);
    float se = 0.f, dot = 0.f, ysum = 0.f;
    for (int k = 0; k < d; ++k) { se += expf(x[k] - mx); dot = fmaf(y[k], x[k], dot); ysum += y[k]; }
    acc += ysum * (mx + logf(se)) - dot;  // -sum_k y_k (x_k - logsumexp(x))
  }
  s_red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) s_red[threadIdx.x] += s_red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = s_red[0];
}
__global__ __launch_bounds__(256) void k_xent_final(const float* __restrict__ partial, int n, size_t cols, float* __restrict__ out) {
  __shared__ float s_red[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
  s_red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) s_red[threadIdx.x] += s_red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = s_red[0] / (float)cols;
}

__global__ void k_xent_backward(const float* __restrict__ logits, const float* __restrict__ targets, int d, size_t cols,
                                const float* __restrict__ upstream, float* __restrict__ dl) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  const float* x = logits + c * d;
  const float* y = targets + c * d;
  float mx = x[0];
  for (int k = 1; k < d; ++k) mx = fmaxf(mx, x[k]);
  float se = 0.f, ysum = 0.f;
  for (int k = 0; k < d; ++k) { se += expf(x[k] - mx); ysum += y[k]; }
  const float g = upstream[0] / (float)cols;
  for (int k = 0; k < d; ++k) dl[c * d + k] = g * (ysum * expf(x[k] - mx) / se - y[k]);
}

int32_t launch_xent_backward(const float* logits, const float* targets, int d, int64_t cols, const float* upstream, float* dl, hipStream_t s) {
  GNX_LAUNCH(k_xent_backward, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s, logits, targets, d, (size_t)cols, upstream, dl);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

int xent_blocks(int64_t cols) { return (int)std::min<int64_t>(std::max<int64_t>((cols + 255) / 256, 1), 1024); }

int32_t launch_xent(const float* logits, const float* targets, int d, int64_t cols, float* out, float* ws, hipStream_t s) {
  const int nb = xent_blocks(cols);
  GNX_LAUNCH(k_xent_partial, dim3(nb), dim3(256), 0, s, logits, targets, d, (size_t)cols, ws);
  GNX_LAUNCH(k_xent_final, dim3(1), dim3(256), 0, s, ws, nb, (size_t)cols, out);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

__global__ void k_null() {}

// n empty launches through the same event bracket as the real kernels: what a bracket costs by itself (bench.py
// subtracts it, so that the event-based kernel time can be compared with rocprofv3's kernel-trace duration)
int32_t launch_calibration(int n, hipStream_t s) {
  for (int i = 0; i < n; ++i) {
    ProfScope ps("__empty_bracket__", s);
    GNX_LAUNCH(k_null, dim3(1), dim3(64), 0, s);
  }
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

int32_t launch_pad(const gnx_graphs* h, int kind, bool pad, const float* src, int d, int64_t R, float* dst, hipStream_t s) {
  const size_t B = h->G == 1 ? (size_t)R : (size_t)h->G;
  const size_t T = kind == 0 ? (size_t)h->E : (size_t)h->N;
  const size_t PT = kind == 0 ? (size_t)h->PN * h->PN : (size_t)h->PN;
  if (pad) GNX_HIP(hipMemsetAsync(dst, 0, B * PT * d * sizeof(float), s));
  const size_t total = (size_t)R * T * d;
  if (total == 0) return GNX_OK;
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (kind == 0) {
    if (pad) GNX_LAUNCH(k_pad_edges<true>, dim3(grid), dim3(256), 0, s, h->d_colptr, h->d_rowval, h->d_node_off, (int)h->N, (int)h->E, (int)h->G, (int)h->PN, d, R, src, dst);
    else GNX_LAUNCH(k_pad_edges<false>, dim3(grid), dim3(256), 0, s, h->d_colptr, h->d_rowval, h->d_node_off, (int)h->N, (int)h->E, (int)h->G, (int)h->PN, d, R, src, dst);
  } else {
    if (pad) GNX_LAUNCH(k_pad_nodes<true>, dim3(grid), dim3(256), 0, s, h->d_node_off, (int)h->N, (int)h->G, (int)h->PN, d, R, src, dst);
    else GNX_LAUNCH(k_pad_nodes<false>, dim3(grid), dim3(256), 0, s, h->d_node_off, (int)h->N, (int)h->G, (int)h->PN, d, R, src, dst);
  }
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

}  // namespace gnx

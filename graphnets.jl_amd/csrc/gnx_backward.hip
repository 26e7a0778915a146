// Backward pass of the GNBlock forward — dimension-generic, deterministic, correctness first.
//
// Forward (src/gnblock.jl:63-69):  ef' = se(We Xe + be), Xe = [ef ; nf[src] ; nf[dst] ; gf[g]]
//                                  nf' = sn(Wn Xn + bn), Xn = [sum_{e->n} ef' ; nf ; gf[g]]
//                                  gf' = sg(Wg Xg + bg), Xg = [sum_e ef' ; sum_n nf' ; gf]
// Backward, with upstream gradients G_ef', G_nf', G_gf':
//   graph:  dg = G_gf' * sg'            dXg = Wg^T dg                       dWg = dg Xg^T
//   node :  dn = (G_nf' + dXg[nodes]) * sn'      dXn = Wn^T dn              dWn = sum_n dn Xn^T
//   edge :  de = (G_ef' + dXg[edges] + dXn[dst][agg]) * se'   dXe = We^T de   dWe = sum_e de Xe^T
//   d_ef = dXe[ef] ; d_nf[n] = dXn[n][nf] + sum_{e: dst=n} dXe[e][dst-seg] + sum_{e: src=n} dXe[e][src-seg] ;
//   d_gf[g] = dXg[g][gf] + sum_{n in g} dXn[n][gf] + sum_{e in g} dXe[e][gf]
// The three function inputs are materialised by the forward's building-block kernels (k_fn_input_*); the gather<->scatter
// duality of nf[src] is resolved with a CSR view (out-edges per node) instead of atomics; every sum has a fixed order.
#include <algorithm>

#include "gnx_device.h"

extern "C" int32_t gnx_ensure_csr(const gnx_graphs* h);
extern "C" size_t gnx_chain_block_workspace_bytes(const gnx_graphs* h, const gnx_chain_block_params* p, int64_t R);

namespace gnx {

// matrix-core primitives (gnx_backward_wide.hip)
bool bw_use_mfma(size_t rows, int J, int K);     // dX
bool bw_use_mfma_dw(size_t rows, int J, int K);  // dW
size_t dw_mfma_partial_floats(size_t rows, int J, int K);
int32_t dw_mfma(const float* delta, const float* X, size_t rows, int J, int K, float* dW, float* partial, hipStream_t s);
int32_t dx_mfma(const gnx_graphs* h, int entity, const float* delta, const float* W, int J, int K, int ka, int kb, float* out, int64_t R,
                float* WT, bool fill, hipStream_t s, const char* name, const float* gmul = nullptr, int gmul_act = 0,
                float* tile_colsum = nullptr, int* n_tiles_out = nullptr);
int32_t transpose_w(const float* W, int K, int J, float* WT, hipStream_t s);
int32_t rows_times_wt(const gnx_graphs* h, int entity, const float* A, int J, const float* WT, int K, int ka, int kb, float* out, const float* add1,
                      int64_t R, hipStream_t s, const char* name);
int32_t segsum_rows(const float* src, const int* ptr, const int* idx, int N, int E, int D, int64_t R, float* out, hipStream_t s, const char* name);
int32_t add_cols(const float* in, int ld, int off, size_t rows, int d, float* out, int accumulate, hipStream_t s);
int32_t launch_dense_rows(const gnx_graphs* h, int entity, const float* A, int K, const gnx_dense& d, int OUT, const float* add1,
                          const float* add2, float* out, int64_t R, hipStream_t s, const char* name);

// csrc/gnx_dropout.hip
bool dropout_active(const gnx_dropout* d);
int32_t check_dropout(const gnx_dropout* d);
int32_t launch_dropout(const gnx_dropout& d, int entity, size_t n, const float* in, float* out, int mode, hipStream_t s);

int32_t launch_fn_input(const gnx_graphs* h, int kind, const float* ef, int de, const float* nf, int dn, const float* gf, int dg,
                        int64_t R, float* out, hipStream_t s);

// delta[m][j] = (G[m][j] + extra1[i1(m)][o1 + j] + extra2[i2(m)][o2 + j]) * act'(out[m][j]);  one thread per element.
// kind 0: rows = graphs (no extras); 1: rows = nodes (extra1 = dXg rows by graph); 2: rows = edges (extra1 = dXg by graph,
// extra2 = dXn by destination node).
// d/dz of NNlib.gelu's tanh form 0.5 z (1 + tanh(c (z + 0.044715 z^3))): unlike relu / tanh / sigmoid it is not a function of the
// OUTPUT (gelu is not monotonic), so for a gelu level the pre-activation z = W x + b is recomputed into the delta buffer first
// (k_fw_dense) and DeltaArgs::out points at it: act code 4 here means "out holds z".
__device__ __forceinline__ float gelu_grad_pre(float z) {
  const float c = 0.7978845608028654f, k = 0.044715f;
  const float t = tanhf(c * (z + k * z * z * z));
  return 0.5f * (1.f + t) + 0.5f * z * (1.f - t * t) * c * (1.f + 3.f * k * z * z);
}
__device__ __forceinline__ float act_grad_bw(float v, int act) { return act == 4 ? gelu_grad_pre(v) : act_grad_from_out(v, act); }

struct DeltaArgs {
  const float* G; const float* out; float* delta;
  const float* ex1; int ex1_stride, ex1_off;
  const float* ex2; int ex2_stride, ex2_off;
  const int* seg_off;   // node_off / edge_off (graph of a row)
  const int* edge_dst;
  int J, rows, n_seg, act, kind;
};
__global__ void k_bw_delta(DeltaArgs a, size_t ex1_rep, size_t ex2_rep) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t r = blockIdx.y;
  if (idx >= (size_t)a.rows * a.J) return;
  const int m = (int)(idx / a.J), j = (int)(idx % a.J);
  const size_t o = r * (size_t)a.rows * a.J + idx;
  float g = a.G ? a.G[o] : 0.f;
  if (a.kind >= 1 && a.ex1) g += a.ex1[r * ex1_rep + (size_t)segment_of(a.seg_off, a.n_seg, m) * a.ex1_stride + a.ex1_off + j];
  if (a.kind == 2 && a.ex2) g += a.ex2[r * ex2_rep + (size_t)a.edge_dst[m] * a.ex2_stride + a.ex2_off + j];
  a.delta[o] = g * act_grad_bw(a.out[o], a.act);
}
// the same, four columns per thread with 16-B accesses (J % 4 == 0, every base / stride / offset a multiple of 4 floats)
__global__ void k_bw_delta_v4(DeltaArgs a, size_t ex1_rep, size_t ex2_rep) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t r = blockIdx.y;
  const int J4 = a.J >> 2;
  if (idx >= (size_t)a.rows * J4) return;
  const int m = (int)(idx / J4), j = 4 * (int)(idx % J4);
  const size_t o = r * (size_t)a.rows * a.J + (size_t)m * a.J + j;
  float4 g = a.G ? *reinterpret_cast<const float4*>(a.G + o) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 y = *reinterpret_cast<const float4*>(a.out + o);
  if (a.kind >= 1 && a.ex1) {
    const float4 u = *reinterpret_cast<const float4*>(a.ex1 + r * ex1_rep + (size_t)segment_of(a.seg_off, a.n_seg, m) * a.ex1_stride + a.ex1_off + j);
    g.x += u.x; g.y += u.y; g.z += u.z; g.w += u.w;
  }
  if (a.kind == 2 && a.ex2) {
    const float4 u = *reinterpret_cast<const float4*>(a.ex2 + r * ex2_rep + (size_t)a.edge_dst[m] * a.ex2_stride + a.ex2_off + j);
    g.x += u.x; g.y += u.y; g.z += u.z; g.w += u.w;
  }
  g.x *= act_grad_bw(y.x, a.act); g.y *= act_grad_bw(y.y, a.act); g.z *= act_grad_bw(y.z, a.act); g.w *= act_grad_bw(y.w, a.act);
  *reinterpret_cast<float4*>(a.delta + o) = g;
}
static void launch_delta(const DeltaArgs& a, size_t ex1_rep, size_t ex2_rep, unsigned Ru, hipStream_t s) {
  const bool al = (((uintptr_t)a.G | (uintptr_t)a.out | (uintptr_t)a.delta | (uintptr_t)a.ex1 | (uintptr_t)a.ex2) & 15) == 0;
  const bool v4 = al && a.J % 4 == 0 && a.ex1_stride % 4 == 0 && a.ex1_off % 4 == 0 && a.ex2_stride % 4 == 0 && a.ex2_off % 4 == 0 && ex1_rep % 4 == 0 &&
                  ex2_rep % 4 == 0;
  if (v4) GNX_LAUNCH(k_bw_delta_v4, dim3((unsigned)(((size_t)a.rows * (a.J / 4) + 255) / 256), Ru), dim3(256), 0, s, a, ex1_rep, ex2_rep);
  else GNX_LAUNCH(k_bw_delta, dim3((unsigned)(((size_t)a.rows * a.J + 255) / 256), Ru), dim3(256), 0, s, a, ex1_rep, ex2_rep);
}

// dX[m][k] = sum_j W[k*J + j] * delta[m][j]   (W is (J x K) column-major); one thread per (m, k)
__global__ void k_bw_dx(const float* __restrict__ delta, const float* __restrict__ W, int rows, int J, int K, float* __restrict__ dX,
                        int k0, int k1, float* __restrict__ direct, int direct_w) {
  // columns [k0, k1) additionally go to `direct` (row stride direct_w) — e.g. the ef segment of dXe is d_ef itself
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t r = blockIdx.y;
  if (idx >= (size_t)rows * K) return;
  const int m = (int)(idx / K), k = (int)(idx % K);
  const float* d = delta + (r * rows + m) * (size_t)J;
  float acc = 0.f;
  for (int j = 0; j < J; ++j) acc = fmaf(W[(size_t)k * J + j], d[j], acc);
  dX[r * (size_t)rows * K + idx] = acc;
  if (direct && k >= k0 && k < k1) direct[(r * rows + m) * (size_t)direct_w + (k - k0)] = acc;
}

// d_nf[n][k] = dXn[n][a_off+k] + sum_{in-edges} dXe[e][dst_off+k] + sum_{out-edges} dXe[e][src_off+k]
__global__ void k_bw_dnf(const float* dXn, int Kn, int n_off, const float* dXe, int Ke, int src_off, int dst_off, const int* colptr,
                         const int* csr_ptr, const int* csr_eid, int N, int E, int dn, float* d_nf) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t r = blockIdx.y;
  if (idx >= (size_t)N * dn) return;
  const int n = (int)(idx / dn), k = (int)(idx % dn);
  float acc = dXn ? dXn[(r * N + n) * (size_t)Kn + n_off + k] : 0.f;
  if (dXe) {
    const float* base = dXe + r * (size_t)E * Ke;
    for (int e = colptr[n]; e < colptr[n + 1]; ++e) acc += base[(size_t)e * Ke + dst_off + k];
    for (int i = csr_ptr[n]; i < csr_ptr[n + 1]; ++i) acc += base[(size_t)csr_eid[i] * Ke + src_off + k];
  }
  d_nf[r * (size_t)N * dn + idx] = acc;
}

// per-graph column sums of a row tensor (sum_e ef', sum_n nf' for Xg), two stages, fixed order:
// stage 1: workgroup (slice s of graph g): thread (grp, c) strides over the slice's rows for column c -> partial[g][s][c]
__global__ __launch_bounds__(256) void k_bw_colsum1(const float* __restrict__ in, int d, int rows_total, const int* __restrict__ off, int S,
                                                    int G, float* __restrict__ partial, int ld, int coff) {  // ld: row length of `in`, coff: first column
  __shared__ float s_red[256];
  const int sl = blockIdx.x % S, g = blockIdx.x / S, tid = threadIdx.x;  // (slice, graph) flattened into grid.x: grid.y/z stop at 65535
  const size_t r = blockIdx.z;
  const int t0 = off[g], t1 = off[g + 1];
  const int per = (t1 - t0 + S - 1) / S;
  const int a0 = t0 + sl * per, a1 = min(a0 + per, t1);
  const float* base = in + r * (size_t)rows_total * ld + coff;
  for (int c0 = 0; c0 < d; c0 += 256) {
    const int dc = min(d - c0, 256);         // columns handled in this pass
    const int groups = 256 / dc;             // row groups working in parallel
    const int c = tid % dc, grp = tid / dc;
    float acc = 0.f;
    if (grp < groups && a1 > a0) {
      for (int m = a0 + grp; m < a1; m += 8 * groups) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[(size_t)min(m + u * groups, a1 - 1) * ld + c0 + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += m + u * groups < a1 ? v[u] : 0.f;
      }
    }
    s_red[tid] = grp < groups ? acc : 0.f;
    __syncthreads();
    if (tid < dc) {
      float t = 0.f;
      for (int q = 0; q < groups; ++q) t += s_red[q * dc + tid];
      partial[((r * G + g) * S + sl) * (size_t)d + c0 + tid] = t;
    }
    __syncthreads();
  }
}
// stage 2: out[(r*G+g)*out_stride + out_off + c] = sum_s partial[g][s][c]
__global__ void k_bw_colsum2(const float* __restrict__ partial, int d, int S, int G, float* __restrict__ out, int out_stride, int out_off,
                             int accumulate) {
  const int g = blockIdx.x;
  const size_t r = blockIdx.y;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float acc = 0.f;
    for (int sl = 0; sl < S; ++sl) acc += partial[((r * G + g) * S + sl) * (size_t)d + c];
    float* o = out + (r * G + g) * (size_t)out_stride + out_off + c;
    *o = accumulate ? *o + acc : acc;
  }
}
// d_gf[g][k] = dXg[g][g_off + k] (or 0): the graph function's own share; the node / edge shares are added by column sums
__global__ void k_bw_dgf_init(const float* __restrict__ dXg, int Kg, int g_off, int GR, int dg, float* __restrict__ d_gf) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= GR * dg) return;
  d_gf[idx] = dXg ? dXg[(size_t)(idx / dg) * Kg + g_off + idx % dg] : 0.f;
}
__global__ void k_bw_copy_gf(const float* __restrict__ gf, int dg, int GR, float* __restrict__ out, int out_stride, int out_off) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= GR * dg) return;
  out[(size_t)(idx / dg) * out_stride + out_off + idx % dg] = gf[idx];
}

// weight / bias gradients: dW[k*J + j] = sum_m delta[m][j] * X[m][k], db[j] = sum_m delta[m][j]  over ALL rows (and replicas).
// stage 1: one workgroup per chunk of CH rows: threads over the (j, k) pairs, rows looped in order -> partial[chunk][J*(K+1)]
constexpr int BW_CH = 2048;
// stage 1: one workgroup per chunk of BW_CH rows; thread = (pair p, row slice): every slice walks its rows in order, the slices
// are then added in order through LDS.  (All threads of a workgroup read the same few rows: L1 broadcasts.)
__global__ __launch_bounds__(256) void k_bw_dw_partial(const float* __restrict__ delta, const float* __restrict__ X, size_t rows, int J, int K,
                                                       float* __restrict__ partial) {
  __shared__ float s_red[256];
  const size_t m0 = (size_t)blockIdx.x * BW_CH;
  const size_t m1 = m0 + BW_CH < rows ? m0 + BW_CH : rows;
  const int P = J * (K + 1);  // pair index p: k = p / J (k == K -> bias), j = p % J
  const int tid = threadIdx.x;
  for (int p0 = 0; p0 < P; p0 += 256) {
    const int pc = min(P - p0, 256);
    const int nsl = 256 / pc;
    const int pl = tid % pc, sl = tid / pc;
    float acc = 0.f;
    if (sl < nsl) {
      const int p = p0 + pl, k = p / J, j = p % J;
      // 8 rows in flight per thread (clamped, unconditional loads: a load inside the guarded loop body costs one memory
      // round trip per row)
      for (size_t m = m0 + sl; m < m1; m += 8 * (size_t)nsl) {
        float dv[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const size_t mm = m + (size_t)u * nsl < m1 ? m + (size_t)u * nsl : m1 - 1;
          dv[u] = delta[mm * J + j];
          xv[u] = k < K ? X[mm * K + k] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = m + (size_t)u * nsl < m1 ? fmaf(dv[u], xv[u], acc) : acc;
      }
    }
    s_red[tid] = sl < nsl ? acc : 0.f;
    __syncthreads();
    if (tid < pc) {
      float t = 0.f;
      for (int q = 0; q < nsl; ++q) t += s_red[q * pc + tid];
      partial[(size_t)blockIdx.x * P + p0 + tid] = t;
    }
    __syncthreads();
  }
}
// stage 2: one workgroup per pair, threads over the chunks, fixed-order LDS tree
__global__ __launch_bounds__(256) void k_bw_dw_final(const float* __restrict__ partial, int nchunks, int J, int K, float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float s_red[256];
  const int P = J * (K + 1), p = blockIdx.x, tid = threadIdx.x;
  float acc = 0.f;
  for (int c = tid; c < nchunks; c += 256) acc += partial[(size_t)c * P + p];
  s_red[tid] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) s_red[tid] += s_red[tid + w];
    __syncthreads();
  }
  if (tid == 0) {
    const int k = p / J, j = p % J;
    if (k < K) { if (dW) dW[(size_t)k * J + j] = s_red[0]; }
    else if (db) db[j] = s_red[0];
  }
}

static int32_t colsum_all(const float* in, size_t rows, int d, float* out, float* part, int* d_off2, hipStream_t s);
__global__ void k_set_off2(int* off2, int rows);

// Few rows, many weights (the reference's sort example: 4 graphs of <= 10 nodes at width 384): one THREAD per (k, j) pair
// walks all rows in order — the chunked kernels above would put every pair on one workgroup.  k == K is the bias.
__global__ void k_bw_dw_small(const float* __restrict__ delta, const float* __restrict__ X, int rows, int J, int K, float* __restrict__ dW,
                              float* __restrict__ db) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (size_t)J * (K + 1)) return;
  const int k = (int)(p / J), j = (int)(p % J);
  float acc = 0.f;
  if (k < K) {
    for (int m = 0; m < rows; ++m) acc = fmaf(delta[(size_t)m * J + j], X[(size_t)m * K + k], acc);
    if (dW) dW[p] = acc;
  } else {
    for (int m = 0; m < rows; ++m) acc += delta[(size_t)m * J + j];
    if (db) db[j] = acc;
  }
}

// generic dX launch with a profiling scope
static void launch_bw_dx(dim3 grid, hipStream_t s, const float* delta, const float* W, int rows, int J, int K, float* dX, int k0, int k1, float* direct,
                         int direct_w) {
  if (grid.x == 0 || grid.y == 0) return;  // a function without inputs (K = 0: e.g. a node function when oe = dn = dg = 0) has no dX
  ProfScope ps("bw_dx_generic", s);
  GNX_LAUNCH(k_bw_dx, grid, dim3(256), 0, s, delta, W, rows, J, K, dX, k0, k1, direct, direct_w);
}

struct BwLayout {
  size_t Xe, Xn, Xg, de_, dn_, dg_, dXe, dXn, dXg, part, wt, off2, ssrc, sdst, sg, tnf, total;
};
static BwLayout bw_layout(const gnx_graphs* h, const gnx_block_params* p, int64_t R) {
  const size_t E = h->E, N = h->N, G = h->G;
  const size_t Ke = p->de + 2 * p->dn + p->dg, Kn = p->oe + p->dn + p->dg, Kg = p->oe + p->on + p->dg;
  BwLayout L{};
  size_t o = 0;
  auto take = [&](size_t floats) { const size_t at = o; o += align_up(floats * sizeof(float), 256); return at; };
  L.Xe = take(R * E * Ke); L.Xn = take(R * N * Kn); L.Xg = take(R * G * Kg);
  L.de_ = take(R * E * p->oe); L.dn_ = take(R * N * p->on); L.dg_ = take(R * G * p->og);
  L.dXe = take(R * E * Ke); L.dXn = take(R * N * Kn); L.dXg = take(R * G * Kg);
  const size_t ch_e = (R * E + BW_CH - 1) / BW_CH, ch_n = (R * N + BW_CH - 1) / BW_CH, ch_g = (R * G + BW_CH - 1) / BW_CH;
  const size_t pmax = std::max({ch_e * p->oe * (Ke + 1), ch_n * p->on * (Kn + 1), ch_g * p->og * (Kg + 1)});
  const size_t cs = std::max((size_t)R * G * 256, (size_t)2048) * std::max({p->oe, p->on, 1});  // column-sum slices
  const size_t pm = std::max(dw_mfma_partial_floats(R * E, p->oe, (int)Ke), dw_mfma_partial_floats(R * N, p->on, (int)Kn));
  L.part = take(std::max({pmax, cs, pm}));
  L.wt = take(std::max({(size_t)p->oe * Ke, (size_t)p->on * Kn, (size_t)p->og * Kg}));
  L.off2 = take(16);
  L.ssrc = take(R * N * p->oe); L.sdst = take(R * N * p->oe); L.sg = take(R * G * p->oe); L.tnf = take(R * N * p->dn);
  L.total = o + 256;
  return L;
}

static int32_t dw_reduce(const float* delta, const float* X, size_t rows, int J, int K, const gnx_dense_grad& g, float* partial, hipStream_t s) {
  if (J == 0 || rows == 0 || (!g.weight && !g.bias)) {
    return GNX_OK;
  }
  const int nchunks = (int)((rows + BW_CH - 1) / BW_CH);
  const int P = J * (K + 1);
  ProfScope ps("bw_dw_generic", s);
  if (nchunks <= 2 && P >= 4096) {
    GNX_LAUNCH(k_bw_dw_small, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, delta, X, (int)rows, J, K, g.weight, g.bias);
    GNX_HIP(hipGetLastError());
    return GNX_OK;
  }
  GNX_LAUNCH(k_bw_dw_partial, dim3(nchunks), dim3(256), 0, s, delta, X, rows, J, K, partial);
  GNX_LAUNCH(k_bw_dw_final, dim3(P), dim3(256), 0, s, partial, nchunks, J, K, g.weight, g.bias);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

// weight + bias gradient of one Dense: matrix cores for real matrices (bias by column sums), the generic reduction otherwise
static int32_t dw_auto(const float* delta, const float* X, size_t rows, int J, int K, const gnx_dense_grad& g, float* partial, int* off2, hipStream_t s) {
  if (!bw_use_mfma_dw(rows, J, K)) return dw_reduce(delta, X, rows, J, K, g, partial, s);
  int32_t rc = dw_mfma(delta, X, rows, J, K, g.weight, partial, s);
  if (rc) return rc;
  GNX_LAUNCH(k_set_off2, dim3(1), dim3(1), 0, s, off2, (int)rows);
  return colsum_all(delta, rows, J, g.bias, partial, off2, s);
}

// ---------------------------------------------------------------------------------------------------------
// GNCore backward pieces
// ---------------------------------------------------------------------------------------------------------
// y[m][j] = act(sum_k W[k*J + j] * x[m][k] + b[j]);  one thread per (m, j)   (recomputation of the FeedForward hidden layer)
__global__ void k_fw_dense(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b, size_t rows, int K, int J,
                           int act, float* __restrict__ y) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * J) return;
  const size_t m = idx / J;
  const int j = (int)(idx % J);
  float acc = b ? b[j] : 0.f;
  const float* xr = x + m * K;
  for (int k = 0; k < K; k += 8) {
    float w[8], xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int kk = k + u < K ? k + u : K - 1; w[u] = W[(size_t)kk * J + j]; xv[u] = xr[kk]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = k + u < K ? fmaf(w[u], xv[u], acc) : acc;
  }
  y[idx] = act_apply(acc, act);
}

// LayerNorm pullback for BOTH norms of a GNCore (they normalise the same x): one wave per row.
//   xhat = (x - mu) / s,  s = sigma + eps (mode 0) or sqrt(sigma^2 + eps) (mode 1);  y_i = gamma_i xhat + beta_i
//   dxhat = dy1*gamma1 + dy2*gamma2;  dx = (dxhat - mean(dxhat)) / s - c * sum(dxhat*c) / (D * q),  c = x - mu,
//   q = sigma*s^2 (mode 0) or s^3 (mode 1);  dx_out = resid + dx.   t1 = dy1*xhat, t2 = dy2*xhat feed the gamma gradients.
__global__ __launch_bounds__(256) void k_ln_backward(const float* __restrict__ x, size_t rows, int d, const float* g1, const float* g2,
                                                     const float* __restrict__ dy1, const float* __restrict__ dy2,
                                                     const float* __restrict__ resid, float eps, int eps_mode, float* __restrict__ dx,
                                                     float* __restrict__ t1, float* __restrict__ t2) {
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
  const float sigma = sqrtf(v);
  const float sden = eps_mode == 0 ? sigma + eps : sqrtf(v + eps);
  const float q = eps_mode == 0 ? sigma * sden * sden : sden * sden * sden;
  float sum_dxh = 0.f, sum_dxh_c = 0.f;
  for (int k = lane; k < d; k += 64) {
    const float c = xr[k] - mu;
    const float a1 = dy1 ? dy1[row * d + k] : 0.f, a2 = dy2 ? dy2[row * d + k] : 0.f;
    const float dxh = a1 * g1[k] + a2 * g2[k];
    sum_dxh += dxh;
    sum_dxh_c = fmaf(dxh, c, sum_dxh_c);
    const float xh = c / sden;
    if (t1) t1[row * d + k] = a1 * xh;
    if (t2) t2[row * d + k] = a2 * xh;
  }
  for (int o = 32; o > 0; o >>= 1) { sum_dxh += __shfl_xor(sum_dxh, o); sum_dxh_c += __shfl_xor(sum_dxh_c, o); }
  const float mean_dxh = sum_dxh / (float)d;
  const float coef = q > 0.f ? sum_dxh_c / ((float)d * q) : 0.f;
  if (dx) {
    for (int k = lane; k < d; k += 64) {
      const float c = xr[k] - mu;
      const float a1 = dy1 ? dy1[row * d + k] : 0.f, a2 = dy2 ? dy2[row * d + k] : 0.f;
      const float dxh = a1 * g1[k] + a2 * g2[k];
      dx[row * d + k] = (resid ? resid[row * d + k] : 0.f) + (dxh - mean_dxh) / sden - c * coef;
    }
  }
}

// LayerNorm pullback for widths that are multiples of 64 (same mathematics as k_ln_backward): 16 lanes per row (= one DPP
// row), the row's x / dy1 / dy2 live in registers, statistics by DPP; a workgroup walks 8 x 16 rows and keeps the column
// partial sums of the four parameter gradients (dgamma1 = sum dy1*xhat, dbeta1 = sum dy1, same for norm 2) in registers,
// reduced over its 16 row slots through LDS in a fixed order -> cpart[block][4][D]; no t1 / t2 tensors, no extra passes.
__device__ __forceinline__ float row16_sum_b(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
  return v;
}
constexpr int LNB_ROWS = 128;  // rows per workgroup
template <int Q>
__global__ __launch_bounds__(256) void k_ln_backward_v4(const float* __restrict__ x, size_t rows, const float* __restrict__ g1, const float* __restrict__ g2,
                                                        const float* __restrict__ dy1, const float* __restrict__ dy2, const float* __restrict__ resid,
                                                        float eps, int eps_mode, float* __restrict__ dx, float* __restrict__ cpart) {
  constexpr int D = 64 * Q;
  __shared__ __attribute__((aligned(16))) float s_red[16 * D];
  const int tid = threadIdx.x, sub = tid & 15, slot = tid >> 4;
  float4 ga[Q], gb[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) { ga[q] = reinterpret_cast<const float4*>(g1)[sub + 16 * q]; gb[q] = reinterpret_cast<const float4*>(g2)[sub + 16 * q]; }
  float4 cg1[Q], cb1[Q], cg2[Q], cb2[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) cg1[q] = cb1[q] = cg2[q] = cb2[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t row_base = (size_t)blockIdx.x * LNB_ROWS;
  for (int it = 0; it < LNB_ROWS / 16; ++it) {
    size_t row = row_base + it * 16 + slot;
    const bool live = row < rows;
    row = live ? row : rows - 1;  // clamped: all lanes take part in the DPP reductions
    float4 c[Q], a1[Q], a2[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      c[q] = reinterpret_cast<const float4*>(x + row * D)[sub + 16 * q];
      a1[q] = reinterpret_cast<const float4*>(dy1 + row * D)[sub + 16 * q];
      a2[q] = reinterpret_cast<const float4*>(dy2 + row * D)[sub + 16 * q];
    }
    float sm = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) sm += (c[q].x + c[q].y) + (c[q].z + c[q].w);
    const float mu = row16_sum_b(sm) * (1.f / (float)D);
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      c[q].x -= mu; c[q].y -= mu; c[q].z -= mu; c[q].w -= mu;
      var = fmaf(c[q].x, c[q].x, var); var = fmaf(c[q].y, c[q].y, var); var = fmaf(c[q].z, c[q].z, var); var = fmaf(c[q].w, c[q].w, var);
    }
    var = row16_sum_b(var) * (1.f / (float)D);
    const float sigma = sqrtf(var);
    const float sden = eps_mode == 0 ? sigma + eps : sqrtf(var + eps);
    const float qd = eps_mode == 0 ? sigma * sden * sden : sden * sden * sden;
    const float inv = 1.f / sden;
    float4 dxh[Q];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      dxh[q] = make_float4(a1[q].x * ga[q].x + a2[q].x * gb[q].x, a1[q].y * ga[q].y + a2[q].y * gb[q].y,
                           a1[q].z * ga[q].z + a2[q].z * gb[q].z, a1[q].w * ga[q].w + a2[q].w * gb[q].w);
      s1 += (dxh[q].x + dxh[q].y) + (dxh[q].z + dxh[q].w);
      s2 = fmaf(dxh[q].x, c[q].x, s2); s2 = fmaf(dxh[q].y, c[q].y, s2); s2 = fmaf(dxh[q].z, c[q].z, s2); s2 = fmaf(dxh[q].w, c[q].w, s2);
    }
    const float mean_dxh = row16_sum_b(s1) * (1.f / (float)D);
    const float tot = row16_sum_b(s2);
    const float coef = qd > 0.f ? tot / ((float)D * qd) : 0.f;
    if (live) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const float4 xh = make_float4(c[q].x * inv, c[q].y * inv, c[q].z * inv, c[q].w * inv);
        cg1[q].x = fmaf(a1[q].x, xh.x, cg1[q].x); cg1[q].y = fmaf(a1[q].y, xh.y, cg1[q].y); cg1[q].z = fmaf(a1[q].z, xh.z, cg1[q].z); cg1[q].w = fmaf(a1[q].w, xh.w, cg1[q].w);
        cg2[q].x = fmaf(a2[q].x, xh.x, cg2[q].x); cg2[q].y = fmaf(a2[q].y, xh.y, cg2[q].y); cg2[q].z = fmaf(a2[q].z, xh.z, cg2[q].z); cg2[q].w = fmaf(a2[q].w, xh.w, cg2[q].w);
        cb1[q].x += a1[q].x; cb1[q].y += a1[q].y; cb1[q].z += a1[q].z; cb1[q].w += a1[q].w;
        cb2[q].x += a2[q].x; cb2[q].y += a2[q].y; cb2[q].z += a2[q].z; cb2[q].w += a2[q].w;
        if (dx) {
          float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
          if (resid) rr = reinterpret_cast<const float4*>(resid + row * D)[sub + 16 * q];
          float4 o;
          o.x = rr.x + (dxh[q].x - mean_dxh) * inv - c[q].x * coef;
          o.y = rr.y + (dxh[q].y - mean_dxh) * inv - c[q].y * coef;
          o.z = rr.z + (dxh[q].z - mean_dxh) * inv - c[q].z * coef;
          o.w = rr.w + (dxh[q].w - mean_dxh) * inv - c[q].w * coef;
          reinterpret_cast<float4*>(dx + row * D)[sub + 16 * q] = o;
        }
      }
    }
  }
  // column partial sums of the workgroup: one quantity at a time through LDS [16 slots][D], summed over the slots in order
  float* outp = cpart + (size_t)blockIdx.x * 4 * D;
#pragma unroll
  for (int which = 0; which < 4; ++which) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float4 v = which == 0 ? cg1[q] : (which == 1 ? cb1[q] : (which == 2 ? cg2[q] : cb2[q]));
      reinterpret_cast<float4*>(s_red + slot * D)[sub + 16 * q] = v;
    }
    __syncthreads();
    for (int cidx = tid; cidx < D; cidx += 256) {
      float acc = 0.f;
#pragma unroll
      for (int sl = 0; sl < 16; ++sl) acc += s_red[sl * D + cidx];
      outp[which * D + cidx] = acc;
    }
  }
}

int32_t launch_layernorm2(const float* x, size_t rows, int d, const gnx_layernorm& l1, const gnx_layernorm& l2, float eps, int eps_mode,
                          float* y1, float* y2, hipStream_t s);

// stage 2 for many slices: out[c] = sum_s partial[s][c]; 64 columns per workgroup, 4 slice groups, fixed order
__global__ __launch_bounds__(256) void k_bw_colsum_final(const float* __restrict__ partial, int d, int S, float* __restrict__ out, int ld = 0) {
  ld = ld ? ld : d;  // distance between slices
  __shared__ float s_red[4][64];
  const int tid = threadIdx.x, cl = tid & 63, sg = tid >> 6;
  const int c = blockIdx.x * 64 + cl;
  float acc = 0.f;
  if (c < d) {
    for (int sl = sg; sl < S; sl += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)min(sl + 4 * u, S - 1) * ld + c];  // clamped: value unused past S
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += sl + 4 * u < S ? v[u] : 0.f;
    }
  }
  s_red[sg][cl] = acc;
  __syncthreads();
  if (tid < 64 && c < d) out[c] = (s_red[0][cl] + s_red[1][cl]) + (s_red[2][cl] + s_red[3][cl]);
}

// column sums over ALL rows of a [rows][d] tensor -> out[d] (one "graph" spanning everything): up to 2048 row slices in
// stage 1 (k_bw_colsum1: enough workgroups to pull HBM bandwidth), parallel fixed-order stage 2
static int32_t colsum_all(const float* in, size_t rows, int d, float* out, float* part, int* d_off2, hipStream_t s) {
  if (!out || d == 0) return GNX_OK;
  if (rows == 0) { GNX_HIP(hipMemsetAsync(out, 0, sizeof(float) * d, s)); return GNX_OK; }
  ProfScope ps("bw_colsum_all", s);
  const int S = (int)std::min<size_t>(std::max<size_t>(rows / 512, 1), 2048);
  GNX_LAUNCH(k_bw_colsum1, dim3((unsigned)S, 1, 1), dim3(256), 0, s, in, d, (int)rows, d_off2, S, 1, part, d, 0);
  GNX_LAUNCH(k_bw_colsum_final, dim3((unsigned)((d + 63) / 64)), dim3(256), 0, s, part, d, S, out, 0);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

__global__ void k_set_off2(int* off2, int rows) { off2[0] = 0; off2[1] = rows; }

// x <- act(x) in place (a gelu FeedForward keeps its pre-activation until delta1 is formed, then becomes the hidden layer)
__global__ void k_act_inplace(float* __restrict__ x, size_t n, int act) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) x[idx] = act_apply(x[idx], act);
}

}  // namespace gnx

using namespace gnx;

extern "C" {

size_t gnx_block_backward_workspace_bytes(const gnx_graphs* h, const gnx_block_params* p, int64_t R) {
  if (!h || !p || R <= 0) return 0;
  (void)gnx_ensure_wide_tables(h);  // what the backward reads is built here, outside any capture (a failure resurfaces in the backward)
  (void)gnx_ensure_csr(h);
  return bw_layout(h, p, R).total;
}

int32_t gnx_block_backward(const gnx_graphs* h, const gnx_block_params* p, const float* ef, const float* nf, const float* gf,
                           const float* ef_out, const float* nf_out, const float* gf_out, const float* g_ef_out, const float* g_nf_out,
                           const float* g_gf_out, int64_t R, float* d_ef, float* d_nf, float* d_gf, const gnx_block_grads* grads,
                           void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!h || !p) return fail(GNX_ERR_INVALID_ARG, "NULL handle or params");
  DeviceTurn turn(s, matrix_core_widths(*p));  // (one matrix-core call at a time per device: gnx_internal.h)
  if (R <= 0 || (R > 1 && h->G != 1) || R > 65535) return fail(GNX_ERR_INVALID_ARG, "bad n_replicas");
  const int de = p->de, dn = p->dn, dg = p->dg, oe = p->oe, on = p->on, og = p->og;
  if (de < 0 || dn < 0 || dg < 0 || oe < 0 || on < 0 || og < 0 || de + dn + dg == 0 || oe + on + og == 0) return fail(GNX_ERR_DIMS, "bad widths");
  if ((de && !ef && h->E > 0) || (dn && !nf) || (dg && !gf)) return fail(GNX_ERR_INVALID_ARG, "a forward input with non-zero width is NULL");
  if ((oe && !ef_out && h->E > 0) || (on && !nf_out) || (og && !gf_out)) return fail(GNX_ERR_INVALID_ARG, "a forward output with non-zero width is NULL");
  const int acts[3] = {p->edgefn.act, p->nodefn.act, p->graphfn.act};
  for (int a : acts)
    if (a < 0 || a > GNX_ACT_GELU) return fail(GNX_ERR_INVALID_ARG, "unknown activation code");
  const BwLayout L = bw_layout(h, p, R);
  if (!ws || ws_bytes < L.total) return fail(GNX_ERR_WORKSPACE, "workspace missing or smaller than gnx_block_backward_workspace_bytes()");
  int32_t rc = gnx_ensure_wide_tables(h, stream);  // (the delta kernels read the destination of every edge; the matrix-core pullbacks the 128-row tiles)
  if (rc) return rc;
  rc = gnx_ensure_csr(h);
  if (rc) return rc;
  char* base = static_cast<char*>(ws);
  auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
  float *Xe = F(L.Xe), *Xn = F(L.Xn), *Xg = F(L.Xg), *dlt_e = F(L.de_), *dlt_n = F(L.dn_), *dlt_g = F(L.dg_);
  float *dXe = F(L.dXe), *dXn = F(L.dXn), *dXg = F(L.dXg), *part = F(L.part), *wt = F(L.wt);
  int* off2 = reinterpret_cast<int*>(base + L.off2);
  int dxe_stride = 0, dxe_col0 = 0;  // row length / first column of what dXe holds (set below)
  const int E = (int)h->E, N = (int)h->N, G = (int)h->G;
  const int Ke = de + 2 * dn + dg, Kn = oe + dn + dg, Kg = oe + on + dg;
  dxe_stride = Ke;
  const gnx_block_grads none{};
  const gnx_block_grads& gr = grads ? *grads : none;
  auto blocks = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
  const unsigned Ru = (unsigned)R;

  // per-graph column sums of a column block of a row tensor, two stages, fixed order
  int64_t me = 1, mn = 1;
  for (int64_t g = 0; g < h->G; ++g) { me = std::max(me, h->h_edge_off[g + 1] - h->h_edge_off[g]); mn = std::max(mn, h->h_node_off[g + 1] - h->h_node_off[g]); }
  auto colsum = [&](const float* in, int d, int ld, int coff, int rows_total, const int* off, int64_t max_rows, float* out, int out_stride, int out_off,
                    int accumulate) {
    if (d == 0) return;
    const int S = (int)std::min<int64_t>(std::max<int64_t>(max_rows / 2048, 1), 256);
    GNX_LAUNCH(k_bw_colsum1, dim3((unsigned)S * (unsigned)G, 1, Ru), dim3(256), 0, s, in, d, rows_total, off, S, G, part, ld, coff);
    GNX_LAUNCH(k_bw_colsum2, dim3((unsigned)G, Ru), dim3(64), 0, s, part, d, S, G, out, out_stride, out_off, accumulate);
  };
  // Edge level on the matrix cores: regrouped (see below) — the edge function's input Xe is never materialised.
  // (a gelu edge function needs its pre-activation, hence the materialised Xe of the generic form)
  const bool mfma_e = oe > 0 && E > 0 && acts[0] != GNX_ACT_GELU && bw_use_mfma((size_t)R * E, oe, Ke);
  // gelu: the level's pre-activation z = W x + b, recomputed into its delta buffer; the delta kernel then reads it in place
  auto preact = [&](int level, const float* X, const gnx_dense& d, size_t rows, int K, int J, float* z) {
    if (acts[level] != GNX_ACT_GELU || rows == 0 || J == 0) return;
    ProfScope ps("bw_gelu_preact", s);
    GNX_LAUNCH(k_fw_dense, blocks(rows * J), dim3(256), 0, s, X, d.weight, d.bias, rows, K, J, GNX_ACT_IDENTITY, z);
  };
  // function inputs, exactly as the forward's building blocks define them
  { ProfScope ps("bw_fn_inputs", s);
  if (oe && E && !mfma_e && (rc = launch_fn_input(h, 0, ef, de, nf, dn, gf, dg, R, Xe, s))) return rc;
  if (on && (rc = launch_fn_input(h, 1, ef_out, oe, nf, dn, gf, dg, R, Xn, s))) return rc; }
  if (og) {  // Xg = [sum_e ef' ; sum_n nf' ; gf] with parallel two-stage column sums (one workgroup per graph would walk 1M rows)
    colsum(ef_out, oe, oe, 0, E, h->d_edge_off, me, Xg, Kg, 0, 0);
    colsum(nf_out, on, on, 0, N, h->d_node_off, mn, Xg, Kg, oe, 0);
    if (dg) GNX_LAUNCH(k_bw_copy_gf, blocks((size_t)R * G * dg), dim3(256), 0, s, gf, dg, (int)(R * G), Xg, Kg, oe + on);
    GNX_HIP(hipGetLastError());
  }

  // graph level
  const bool have_g = og > 0;
  if (have_g) {
    preact(2, Xg, p->graphfn, (size_t)R * G, Kg, og, dlt_g);
    DeltaArgs a{g_gf_out, acts[2] == GNX_ACT_GELU ? dlt_g : gf_out, dlt_g, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, og, G, G, acts[2], 0};
    GNX_LAUNCH(k_bw_delta, dim3(blocks((size_t)G * og).x, Ru), dim3(256), 0, s, a, (size_t)0, (size_t)0);
    launch_bw_dx(dim3(blocks((size_t)G * Kg).x, Ru), s, dlt_g, p->graphfn.weight, G, og, Kg, dXg, 0, 0, (float*)nullptr, 0);
    if ((rc = dw_reduce(dlt_g, Xg, (size_t)R * G, og, Kg, gr.graphfn, part, s))) return rc;
  }
  // node level
  const bool have_n = on > 0;
  if (have_n) {
    preact(1, Xn, p->nodefn, (size_t)R * N, Kn, on, dlt_n);
    DeltaArgs a{g_nf_out, acts[1] == GNX_ACT_GELU ? dlt_n : nf_out, dlt_n, have_g ? dXg : nullptr, Kg, oe, nullptr, 0, 0, h->d_node_off, nullptr, on, N, G, acts[1], 1};
    launch_delta(a, (size_t)G * Kg, (size_t)0, Ru, s);
    if (bw_use_mfma((size_t)R * N, on, Kn)) {  // matrix cores: dXn = dn Wn^T, dWn = Xn^T dn, dbn = column sums
      if ((rc = dx_mfma(h, 1, dlt_n, p->nodefn.weight, on, Kn, 0, Kn, dXn, R, wt, true, s, "bw_dx_node"))) return rc;
      if ((rc = dw_auto(dlt_n, Xn, (size_t)R * N, on, Kn, gr.nodefn, part, off2, s))) return rc;
    } else {
      launch_bw_dx(dim3(blocks((size_t)N * Kn).x, Ru), s, dlt_n, p->nodefn.weight, N, on, Kn, dXn, 0, 0, (float*)nullptr, 0);
      if ((rc = dw_reduce(dlt_n, Xn, (size_t)R * N, on, Kn, gr.nodefn, part, s))) return rc;
    }
  }
  // edge level
  const bool have_e = oe > 0 && E > 0;
  if (have_e) {
    preact(0, Xe, p->edgefn, (size_t)R * E, Ke, oe, dlt_e);
    DeltaArgs a{g_ef_out, acts[0] == GNX_ACT_GELU ? dlt_e : ef_out, dlt_e, have_g ? dXg : nullptr, Kg, 0, have_n ? dXn : nullptr, Kn, 0, h->d_edge_off, h->d_edge_dst, oe, E, G, acts[0], 2};
    { ProfScope ps("bw_delta_edge", s);
    launch_delta(a, (size_t)G * Kg, (size_t)N * Kn, Ru, s); }
    if (mfma_e) {
      // Matrix cores, regrouped so that only the ef block is a per-edge product.  With de = delta of the edges and
      //   S_dst[n] = sum_{e: dst = n} de[e]  (contiguous CSC segments)   S_src[n] = sum_{e: src = n} de[e]  (CSR view)   S_g = sum_{n in g} S_dst[n]
      //   d_ef = de We_ef^T                                 dWe[ef rows]  = ef^T de
      //   d_nf += S_src We_src^T + S_dst We_dst^T           dWe[src rows] = nf^T S_src     dWe[dst rows] = nf^T S_dst
      //   d_gf += S_g We_gf^T                               dWe[gf rows]  = gf^T S_g
      // (sum_e nf[src(e)]^T de[e] = sum_n nf[n]^T S_src[n]: the gathered operand never exists), every sum in a fixed order.
      float *S_src = F(L.ssrc), *S_dst = F(L.sdst), *S_g = F(L.sg), *T_nf = F(L.tnf);
      if ((rc = transpose_w(p->edgefn.weight, Ke, oe, wt, s))) return rc;
      if (d_ef && de && (rc = rows_times_wt(h, 0, dlt_e, oe, wt, Ke, 0, de, d_ef, nullptr, R, s, "bw_dx_edge_ef"))) return rc;
      const bool need_seg = (dn > 0 && (d_nf || gr.edgefn.weight)) || (dg > 0 && (d_gf || gr.edgefn.weight));
      if (need_seg) {
        if ((rc = segsum_rows(dlt_e, h->d_colptr, nullptr, N, E, oe, R, S_dst, s, "bw_segsum_dst"))) return rc;
        if (dn && (rc = segsum_rows(dlt_e, h->d_csr_ptr, h->d_csr_eid, N, E, oe, R, S_src, s, "bw_segsum_src"))) return rc;
        if (dg) colsum(S_dst, oe, oe, 0, N, h->d_node_off, mn, S_g, oe, 0, 0);
      }
      if (d_nf && dn) {  // the node function's own share (dXn nf columns) is added below
        if ((rc = rows_times_wt(h, 1, S_src, oe, wt, Ke, de, de + dn, T_nf, nullptr, R, s, "bw_dx_nf_src"))) return rc;
        if ((rc = rows_times_wt(h, 1, S_dst, oe, wt, Ke, de + dn, de + 2 * dn, d_nf, T_nf, R, s, "bw_dx_nf_dst"))) return rc;
        if (have_n && (rc = add_cols(dXn, Kn, oe, (size_t)R * N, dn, d_nf, 1, s))) return rc;
      }
      if (dg) {  // dXe_g[g][:] = S_g[g] We^T (G rows: generic), its gf columns are the edges' share of d_gf
        launch_bw_dx(dim3(blocks((size_t)G * Ke).x, Ru), s, S_g, p->edgefn.weight, G, oe, Ke, dXe, 0, 0, (float*)nullptr, 0);
      }
      if (gr.edgefn.weight) {
        float* dW = gr.edgefn.weight;
        auto dw_any = [&](const float* delta, const float* X, size_t rows, int K, float* out) -> int32_t {
          if (K == 0) return GNX_OK;
          if (bw_use_mfma_dw(rows, oe, K)) return dw_mfma(delta, X, rows, oe, K, out, part, s);
          return dw_reduce(delta, X, rows, oe, K, gnx_dense_grad{out, nullptr}, part, s);
        };
        if ((rc = dw_any(dlt_e, ef, (size_t)R * E, de, dW))) return rc;
        if ((rc = dw_any(S_src, nf, (size_t)R * N, dn, dW + (size_t)de * oe))) return rc;
        if ((rc = dw_any(S_dst, nf, (size_t)R * N, dn, dW + (size_t)(de + dn) * oe))) return rc;
        if ((rc = dw_any(S_g, gf, (size_t)R * G, dg, dW + (size_t)(de + 2 * dn) * oe))) return rc;
      }
      GNX_LAUNCH(k_set_off2, dim3(1), dim3(1), 0, s, off2, (int)(R * E));
      if ((rc = colsum_all(dlt_e, (size_t)R * E, oe, gr.edgefn.bias, part, off2, s))) return rc;
    } else {
      launch_bw_dx(dim3(blocks((size_t)E * Ke).x, Ru), s, dlt_e, p->edgefn.weight, E, oe, Ke, dXe, 0, de, d_ef, de);
      if ((rc = dw_reduce(dlt_e, Xe, (size_t)R * E, oe, Ke, gr.edgefn, part, s))) return rc;
    }
  } else {
    if (d_ef && de && E) GNX_HIP(hipMemsetAsync(d_ef, 0, sizeof(float) * (size_t)R * E * de, s));
    if (gr.edgefn.weight && oe) GNX_HIP(hipMemsetAsync(gr.edgefn.weight, 0, sizeof(float) * (size_t)oe * Ke, s));
    if (gr.edgefn.bias && oe) GNX_HIP(hipMemsetAsync(gr.edgefn.bias, 0, sizeof(float) * (size_t)oe, s));
  }
  if (!have_n) {
    if (gr.nodefn.weight && on) GNX_HIP(hipMemsetAsync(gr.nodefn.weight, 0, sizeof(float) * (size_t)on * Kn, s));
    if (gr.nodefn.bias && on) GNX_HIP(hipMemsetAsync(gr.nodefn.bias, 0, sizeof(float) * (size_t)on, s));
  }
  // input gradients that need sums
  if (d_nf && dn && !mfma_e) {
    ProfScope ps("bw_dnf", s);
    GNX_LAUNCH(k_bw_dnf, dim3(blocks((size_t)N * dn).x, Ru), dim3(256), 0, s, have_n ? dXn : nullptr, Kn, oe, have_e ? dXe : nullptr, dxe_stride,
                       de - dxe_col0, de + dn - dxe_col0, h->d_colptr, h->d_csr_ptr, h->d_csr_eid, N, E, dn, d_nf);
  }
  if (d_gf && dg) {  // d_gf[g] = dXg[g][gf cols] + sum_{n in g} dXn[n][gf cols] + sum_{e in g} dXe[e][gf cols]
    ProfScope ps("bw_dgf", s);
    GNX_LAUNCH(k_bw_dgf_init, blocks((size_t)R * G * dg), dim3(256), 0, s, have_g ? dXg : nullptr, Kg, oe + on, (int)(R * G), dg, d_gf);
    if (have_n) colsum(dXn, dg, Kn, oe + dn, N, h->d_node_off, mn, d_gf, dg, 0, 1);
    if (have_e && mfma_e) { if ((rc = add_cols(dXe, Ke, de + 2 * dn, (size_t)R * G, dg, d_gf, 1, s))) return rc; }  // dXe holds S_g We^T here
    else if (have_e) colsum(dXe, dg, dxe_stride, de + 2 * dn - dxe_col0, E, h->d_edge_off, me, d_gf, dg, 0, 1);
  }
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}


// ---- GNCore backward ----
namespace {
struct CoreBwLayout {
  size_t l1[3], l2[3], bout[3], dl1[3], dz2[3], h, dh, t1, t2, off2, blk_fw, blk_bw, part, wt, tcs, lnpart, total;
};
CoreBwLayout core_bw_layout(const gnx_graphs* h, const gnx_core_params* p, int64_t R) {
  const size_t rows[3] = {(size_t)R * h->E, (size_t)R * h->N, (size_t)R * h->G};
  const int d[3] = {p->block.de, p->block.dn, p->block.dg};
  CoreBwLayout L{};
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes, 256); return at; };
  size_t hmax = 0, tmax = 0;
  for (int t = 0; t < 3; ++t) {
    const size_t b = sizeof(float) * rows[t] * d[t];
    L.l1[t] = take(b); L.l2[t] = take(b); L.bout[t] = take(b); L.dl1[t] = take(b); L.dz2[t] = take(b);
    hmax = std::max(hmax, sizeof(float) * rows[t] * 4 * (size_t)d[t]);
    tmax = std::max(tmax, b);
  }
  L.h = take(hmax); L.dh = take(hmax); L.t1 = take(tmax); L.t2 = take(tmax); L.off2 = take(64);
  L.blk_fw = take(gnx_block_workspace_bytes(h, &p->block, R));
  L.blk_bw = take(gnx_block_backward_workspace_bytes(h, &p->block, R));
  size_t pmax = sizeof(float) * 2048 * 4 * (size_t)std::max(d[0], std::max(d[1], d[2]));
  for (int t = 0; t < 3; ++t) {
    const size_t ch = (rows[t] + BW_CH - 1) / BW_CH;
    pmax = std::max(pmax, sizeof(float) * ch * (size_t)(4 * d[t]) * (d[t] + 1));
    pmax = std::max(pmax, sizeof(float) * ch * (size_t)d[t] * (4 * d[t] + 1));
    pmax = std::max(pmax, sizeof(float) * dw_mfma_partial_floats(rows[t], 4 * d[t], d[t]));
  }
  L.part = take(pmax);
  L.wt = take(sizeof(float) * 4 * (size_t)std::max(d[0], std::max(d[1], d[2])) * std::max(d[0], std::max(d[1], d[2])));
  size_t tcs = 0;  // per-tile column sums of delta1 (tiles of the matrix-core path: 128-row chunks per graph)
  const size_t nt[3] = {(size_t)h->n_etiles, (size_t)h->n_ntiles, (size_t)h->n_gtiles};
  for (int t = 0; t < 3; ++t) tcs = std::max(tcs, sizeof(float) * (size_t)R * nt[t] * 4 * d[t]);
  L.tcs = take(tcs);
  size_t lnp = 0;  // k_ln_backward_v4: [blocks][4][D] column partial sums
  for (int t = 0; t < 3; ++t) lnp = std::max(lnp, sizeof(float) * ((rows[t] + 127) / 128) * 4 * (size_t)d[t]);
  L.lnpart = take(lnp);
  L.total = o + 256;
  return L;
}
}  // namespace

size_t gnx_core_backward_workspace_bytes(const gnx_graphs* h, const gnx_core_params* p, int64_t R) {
  if (!h || !p || R <= 0) return 0;
  (void)gnx_ensure_wide_tables(h);
  (void)gnx_ensure_csr(h);
  return core_bw_layout(h, p, R).total;
}

}  // extern "C"

// `dr` (gnx_core_backward_train): the FeedForwards' outputs were multiplied by the Dropout masks of *dr in the forward
// (gnx_core_forward_train, csrc/gnx_dropout.hip), so the upstream gradient of every FeedForward branch is g_out .* mask — the mask
// regenerated from (seed, entity, element); the block branch and the residual see g_out as it is.
static int32_t core_backward_impl(const gnx_graphs* h, const gnx_core_params* p, const gnx_dropout* dr, const float* ef, const float* nf, const float* gf,
                                  const float* g_ef_out, const float* g_nf_out, const float* g_gf_out, int64_t R, float* d_ef, float* d_nf,
                                  float* d_gf, const gnx_core_grads* grads, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!h || !p) return fail(GNX_ERR_INVALID_ARG, "NULL handle or params");
  const gnx_block_params& b = p->block;
  DeviceTurn turn(s, matrix_core_widths(b));
  if (b.de <= 0 || b.dn <= 0 || b.dg <= 0 || b.oe != b.de || b.on != b.dn || b.og != b.dg) return fail(GNX_ERR_DIMS, "GNCore needs dims => dims with all(dims .> 0)");
  if ((!ef && h->E > 0) || !nf || !gf) return fail(GNX_ERR_INVALID_ARG, "GNCore needs ef, nf and gf");
  if (R <= 0 || (R > 1 && h->G != 1) || R > 65535) return fail(GNX_ERR_INVALID_ARG, "bad n_replicas");
  for (int t = 0; t < 3; ++t) {
    if (p->ff[t].fc2.act != GNX_ACT_IDENTITY) return fail(GNX_ERR_INVALID_ARG, "core backward: fc2 must be identity");
  }
  const CoreBwLayout L = core_bw_layout(h, p, R);
  if (!ws || ws_bytes < L.total) return fail(GNX_ERR_WORKSPACE, "workspace missing or smaller than gnx_core_backward_workspace_bytes()");
  if (int32_t rcw = gnx_ensure_wide_tables(h, stream)) return rcw;
  char* base = static_cast<char*>(ws);
  auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
  const size_t rows[3] = {(size_t)R * h->E, (size_t)R * h->N, (size_t)R * h->G};
  const int d[3] = {b.de, b.dn, b.dg};
  const float* x[3] = {ef, nf, gf};
  const float* gout[3] = {g_ef_out, g_nf_out, g_gf_out};
  float* dxo[3] = {d_ef, d_nf, d_gf};
  const gnx_core_grads none{};
  const gnx_core_grads& gr = grads ? *grads : none;
  int32_t rc;
  auto blocks = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
  int* off2 = reinterpret_cast<int*>(base + L.off2);
  float* part = F(L.part);

  // 1. recompute gn1(x), gn2(x) and the block's outputs
  for (int t = 0; t < 3; ++t)
    if ((rc = launch_layernorm2(x[t], rows[t], d[t], p->ln1[t], p->ln2[t], p->eps, p->eps_mode, F(L.l1[t]), F(L.l2[t]), s))) return rc;
  if ((rc = gnx_block_forward(h, &b, F(L.l1[0]), F(L.l1[1]), F(L.l1[2]), R, F(L.bout[0]), F(L.bout[1]), F(L.bout[2]), base + L.blk_fw,
                              gnx_block_workspace_bytes(h, &b, R), 0, stream))) return rc;
  // 2. FeedForward pullback per entity: f = W2 h + b2, h = act1(W1 z + b1), z = gn2(x); upstream of f is g_out
  for (int t = 0; t < 3; ++t) {
    float* dz2 = F(L.dz2[t]);
    const int D = d[t], H = 4 * d[t];
    if (rows[t] == 0 || !gout[t]) {  // an entity without rows (a batch without edges), or no upstream gradient on it: the FeedForward branch contributes nothing
      if (rows[t]) GNX_HIP(hipMemsetAsync(dz2, 0, sizeof(float) * rows[t] * D, s));
      if (gr.ff[t].fc1.weight) GNX_HIP(hipMemsetAsync(gr.ff[t].fc1.weight, 0, sizeof(float) * (size_t)H * D, s));
      if (gr.ff[t].fc1.bias) GNX_HIP(hipMemsetAsync(gr.ff[t].fc1.bias, 0, sizeof(float) * H, s));
      if (gr.ff[t].fc2.weight) GNX_HIP(hipMemsetAsync(gr.ff[t].fc2.weight, 0, sizeof(float) * (size_t)H * D, s));
      if (gr.ff[t].fc2.bias) GNX_HIP(hipMemsetAsync(gr.ff[t].fc2.bias, 0, sizeof(float) * D, s));
      continue;
    }
    float* hbuf = F(L.h); float* dh = F(L.dh);
    const float* gff = gout[t];  // upstream of the FeedForward branch
    if (dropout_active(dr)) {
      if ((rc = launch_dropout(*dr, t, rows[t] * (size_t)D, gout[t], F(L.t1), 1, s))) return rc;  // (t1 is free until step 4)
      gff = F(L.t1);
    }
    // gelu is not a function of its output: hbuf first holds the PRE-activation z1 = W1 z + b1, delta1 = dh .* gelu'(z1) is formed from
    // it (act code 4 of the delta kernel = "out holds z"), then hbuf becomes h = gelu(z1) in place for dW2
    const int act1 = p->ff[t].fc1.act;
    const bool gelu1 = act1 == GNX_ACT_GELU;
    gnx_dense fc1_re = p->ff[t].fc1;
    if (gelu1) fc1_re.act = GNX_ACT_IDENTITY;
    auto gelu_delta_then_hidden = [&]() {
      DeltaArgs a{dh, hbuf, dh, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, H, (int)rows[t], 1, GNX_ACT_GELU, 0};
      { ProfScope ps("bw_delta", s); launch_delta(a, 0, 0, 1, s); }
      ProfScope ps("bw_gelu_hidden", s);
      GNX_LAUNCH(k_act_inplace, blocks(rows[t] * H), dim3(256), 0, s, hbuf, rows[t] * (size_t)H, GNX_ACT_GELU);
    };
    if (bw_use_mfma(rows[t], D, H)) {  // matrix cores (gnx_backward_wide.hip); rows[t] = R * (rows of entity t)
      float* wt = F(L.wt);
      if ((rc = launch_dense_rows(h, t, F(L.l2[t]), D, fc1_re, H, nullptr, nullptr, hbuf, R, s, "bw_ff1_recompute"))) return rc;
      if (!gelu1 && (rc = dw_auto(gff, hbuf, rows[t], D, H, gr.ff[t].fc2, part, off2, s))) return rc;           // dW2 = h^T g, db2
      // delta1 = (g W2^T) .* act1'(h) in the GEMM epilogue, with per-tile column sums of delta1 for db1
      int n_tiles = 0;
      const bool dw1_mfma = bw_use_mfma_dw(rows[t], H, D);
      float* tcs = gr.ff[t].fc1.bias && dw1_mfma && !gelu1 ? F(L.tcs) : nullptr;
      if ((rc = dx_mfma(h, t, gff, p->ff[t].fc2.weight, D, H, 0, H, dh, R, wt, true, s, "bw_dx_ff2", gelu1 ? nullptr : hbuf, act1, tcs, &n_tiles))) return rc;
      if (gelu1) {
        gelu_delta_then_hidden();
        if ((rc = dw_auto(gff, hbuf, rows[t], D, H, gr.ff[t].fc2, part, off2, s))) return rc;
      }
      if (dw1_mfma) rc = dw_mfma(dh, F(L.l2[t]), rows[t], H, D, gr.ff[t].fc1.weight, part, s);                   // dW1 = z^T delta1
      else rc = dw_reduce(dh, F(L.l2[t]), rows[t], H, D, gr.ff[t].fc1, part, s);                                   // (+ db1)
      if (rc) return rc;
      if (tcs) {
        ProfScope ps("bw_colsum_all", s);
        GNX_LAUNCH(k_bw_colsum_final, dim3((unsigned)((H + 63) / 64)), dim3(256), 0, s, tcs, H, (int)(R * n_tiles), gr.ff[t].fc1.bias, 0);
      } else if (gelu1 && dw1_mfma && gr.ff[t].fc1.bias) {
        GNX_LAUNCH(k_set_off2, dim3(1), dim3(1), 0, s, off2, (int)rows[t]);
        if ((rc = colsum_all(dh, rows[t], H, gr.ff[t].fc1.bias, part, off2, s))) return rc;
      }
      if ((rc = dx_mfma(h, t, dh, p->ff[t].fc1.weight, H, D, 0, D, dz2, R, wt, true, s, "bw_dx_ff1"))) return rc;        // dz2 = delta1 W1^T
      GNX_HIP(hipGetLastError());
      continue;
    }
    { ProfScope ps("bw_fw_dense_generic", s);
    GNX_LAUNCH(k_fw_dense, blocks(rows[t] * H), dim3(256), 0, s, F(L.l2[t]), p->ff[t].fc1.weight, p->ff[t].fc1.bias, rows[t], D, H, fc1_re.act, hbuf); }
    if (!gelu1 && (rc = dw_reduce(gff, hbuf, rows[t], D, H, gr.ff[t].fc2, part, s))) return rc;             // dW2 = g^T h
    launch_bw_dx(dim3(blocks(rows[t] * H).x, 1), s, gff, p->ff[t].fc2.weight, (int)rows[t], D, H, dh, 0, 0, (float*)nullptr, 0);
    if (gelu1) {
      gelu_delta_then_hidden();
      if ((rc = dw_reduce(gff, hbuf, rows[t], D, H, gr.ff[t].fc2, part, s))) return rc;
    } else {
      DeltaArgs a{dh, hbuf, dh, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, H, (int)rows[t], 1, act1, 0};  // delta1 = dh * act1'(h), in place
      GNX_LAUNCH(k_bw_delta, dim3(blocks(rows[t] * H).x, 1), dim3(256), 0, s, a, (size_t)0, (size_t)0);
    }
    if ((rc = dw_reduce(dh, F(L.l2[t]), rows[t], H, D, gr.ff[t].fc1, part, s))) return rc;                      // dW1 = delta1^T z
    launch_bw_dx(dim3(blocks(rows[t] * D).x, 1), s, dh, p->ff[t].fc1.weight, (int)rows[t], H, D, dz2, 0, 0, (float*)nullptr, 0);
    GNX_HIP(hipGetLastError());
  }
  // 3. block pullback: inputs gn1(x), outputs recomputed above, upstream g_out -> gradients w.r.t. gn1(x)
  if ((rc = gnx_block_backward(h, &b, F(L.l1[0]), F(L.l1[1]), F(L.l1[2]), F(L.bout[0]), F(L.bout[1]), F(L.bout[2]), g_ef_out, g_nf_out, g_gf_out, R,
                               F(L.dl1[0]), F(L.dl1[1]), F(L.dl1[2]), &gr.block, base + L.blk_bw, gnx_block_backward_workspace_bytes(h, &b, R), stream))) return rc;
  // 4. LayerNorm pullbacks (both norms at once) + residual; gamma/beta gradients as column sums over all rows
  for (int t = 0; t < 3; ++t) {
    if (rows[t] == 0) {  // (sums over nothing: zeros, not what the buffers held)
      for (float* o : {gr.ln1[t].gamma, gr.ln1[t].beta, gr.ln2[t].gamma, gr.ln2[t].beta})
        if (o && d[t] > 0) GNX_HIP(hipMemsetAsync(o, 0, sizeof(float) * (size_t)d[t], s));
      continue;
    }
    float* t1 = F(L.t1); float* t2 = F(L.t2);
    const bool al16 = (((uintptr_t)x[t] | (uintptr_t)F(L.dl1[t]) | (uintptr_t)F(L.dz2[t]) | (uintptr_t)gout[t] | (uintptr_t)dxo[t] | (uintptr_t)p->ln1[t].gamma |
                        (uintptr_t)p->ln2[t].gamma) & 15) == 0;
    if (al16 && d[t] % 64 == 0 && d[t] <= 512 && rows[t] >= 1024) {
      const unsigned nb = (unsigned)((rows[t] + LNB_ROWS - 1) / LNB_ROWS);
      float* cpart = F(L.lnpart);
      {
        ProfScope ps("bw_layernorm", s);
        switch (d[t] / 64) {
#define GNX_LNB_CASE(Q) case Q: GNX_LAUNCH((k_ln_backward_v4<Q>), dim3(nb), dim3(256), 0, s, x[t], rows[t], p->ln1[t].gamma, p->ln2[t].gamma, \
                                                   F(L.dl1[t]), F(L.dz2[t]), gout[t], p->eps, p->eps_mode, dxo[t], cpart); break;
          GNX_LNB_CASE(1) GNX_LNB_CASE(2) GNX_LNB_CASE(3) GNX_LNB_CASE(4) GNX_LNB_CASE(5) GNX_LNB_CASE(6) GNX_LNB_CASE(7) GNX_LNB_CASE(8)
#undef GNX_LNB_CASE
        }
      }
      ProfScope ps("bw_colsum_all", s);
      float* outs[4] = {gr.ln1[t].gamma, gr.ln1[t].beta, gr.ln2[t].gamma, gr.ln2[t].beta};
      for (int w = 0; w < 4; ++w)
        if (outs[w]) GNX_LAUNCH(k_bw_colsum_final, dim3((unsigned)((d[t] + 63) / 64)), dim3(256), 0, s, cpart + (size_t)w * d[t], d[t], (int)nb, outs[w], 4 * d[t]);
      GNX_HIP(hipGetLastError());
      continue;
    }
    { ProfScope ps("bw_layernorm", s);
    GNX_LAUNCH(k_ln_backward, dim3((unsigned)((rows[t] + 3) / 4)), dim3(256), 0, s, x[t], rows[t], d[t], p->ln1[t].gamma, p->ln2[t].gamma, F(L.dl1[t]),
                       F(L.dz2[t]), gout[t], p->eps, p->eps_mode, dxo[t], t1, t2); }
    GNX_LAUNCH(k_set_off2, dim3(1), dim3(1), 0, s, off2, (int)rows[t]);
    if ((rc = colsum_all(t1, rows[t], d[t], gr.ln1[t].gamma, part, off2, s))) return rc;
    if ((rc = colsum_all(F(L.dl1[t]), rows[t], d[t], gr.ln1[t].beta, part, off2, s))) return rc;
    if ((rc = colsum_all(t2, rows[t], d[t], gr.ln2[t].gamma, part, off2, s))) return rc;
    if ((rc = colsum_all(F(L.dz2[t]), rows[t], d[t], gr.ln2[t].beta, part, off2, s))) return rc;
  }
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

extern "C" {
int32_t gnx_core_backward(const gnx_graphs* h, const gnx_core_params* p, const float* ef, const float* nf, const float* gf,
                          const float* g_ef_out, const float* g_nf_out, const float* g_gf_out, int64_t R, float* d_ef, float* d_nf,
                          float* d_gf, const gnx_core_grads* grads, void* ws, size_t ws_bytes, void* stream) {
  return core_backward_impl(h, p, nullptr, ef, nf, gf, g_ef_out, g_nf_out, g_gf_out, R, d_ef, d_nf, d_gf, grads, ws, ws_bytes, stream);
}
int32_t gnx_core_backward_train(const gnx_graphs* h, const gnx_core_params* p, const gnx_dropout* dropout, const float* ef, const float* nf,
                                const float* gf, const float* g_ef_out, const float* g_nf_out, const float* g_gf_out, int64_t R, float* d_ef,
                                float* d_nf, float* d_gf, const gnx_core_grads* grads, void* ws, size_t ws_bytes, void* stream) {
  if (int32_t rc = check_dropout(dropout)) return rc;
  return core_backward_impl(h, p, dropout, ef, nf, gf, g_ef_out, g_nf_out, g_gf_out, R, d_ef, d_nf, d_gf, grads, ws, ws_bytes, stream);
}
}  // extern "C"

// ---- GNBlock with Chain update functions: backward ----
namespace {
int chain_out(const gnx_chain& c) { return c.n_layers > 0 ? c.widths[c.n_layers - 1] : 0; }
int chain_max(const gnx_chain& c) {
  int m = 0;
  for (int i = 0; i < c.n_layers; ++i) m = std::max(m, c.widths[i]);
  return m;
}
gnx_block_params chain_edge_block(const gnx_chain_block_params* p) {  // the one-layer block that is the edge chain's first Dense
  gnx_block_params b{};
  b.de = p->de; b.dn = p->dn; b.dg = p->dg;
  b.oe = p->edgefn.widths[0]; b.on = 0; b.og = 0;
  b.edgefn = p->edgefn.layers[0];
  return b;
}
struct ChainBwLayout {
  size_t act[3][kChainMaxLayers + 1];  // stored layer outputs per chain
  size_t Xn, Xg, dXn, dXg, gbuf[3], blk_fw, blk_fw_bytes, blk_bw, blk_bw_bytes, part, wt, off2, ident, total;
};
ChainBwLayout chain_bw_layout(const gnx_graphs* h, const gnx_chain_block_params* p, int64_t R, size_t ident_floats) {
  ChainBwLayout L{};
  size_t o = 0;
  auto take = [&](size_t floats) { const size_t at = o; o += align_up(floats * sizeof(float), 256); return at; };
  const gnx_chain* ch[3] = {&p->edgefn, &p->nodefn, &p->graphfn};
  const size_t rows[3] = {(size_t)R * h->E, (size_t)R * h->N, (size_t)R * h->G};
  const int oe = chain_out(p->edgefn), on = chain_out(p->nodefn), og = chain_out(p->graphfn);
  const int Kn = oe + p->dn + p->dg, Kg = oe + on + p->dg;
  const int k0[3] = {p->de + 2 * p->dn + p->dg, Kn, Kg};
  size_t gmax = 1, pmax = 2048 * 4, wmax = 1;
  for (int t = 0; t < 3; ++t) {
    for (int i = 0; i < ch[t]->n_layers; ++i) {
      L.act[t][i] = take(rows[t] * (size_t)ch[t]->widths[i]);
      const int J = ch[t]->widths[i], K = i > 0 ? ch[t]->widths[i - 1] : k0[t];
      const size_t chunks = (rows[t] + BW_CH - 1) / BW_CH;
      pmax = std::max({pmax, chunks * (size_t)J * (K + 1), (size_t)2048 * J, dw_mfma_partial_floats(rows[t], J, K)});
      wmax = std::max(wmax, (size_t)J * K);
    }
    gmax = std::max(gmax, rows[t] * (size_t)chain_max(*ch[t]));
  }
  pmax = std::max(pmax, std::max((size_t)R * h->G * 256, (size_t)2048) * (size_t)std::max({p->dg, oe, 1}));  // column-sum slices
  L.Xn = take(on > 0 ? rows[1] * Kn : 0); L.dXn = take(on > 0 ? rows[1] * Kn : 0);
  L.Xg = take(og > 0 ? rows[2] * Kg : 0); L.dXg = take(og > 0 ? rows[2] * Kg : 0);
  for (int i = 0; i < 3; ++i) L.gbuf[i] = take(gmax);
  const gnx_block_params b = chain_edge_block(p);
  L.blk_fw_bytes = gnx_block_workspace_bytes(h, &b, R);
  L.blk_fw = o; o += align_up(L.blk_fw_bytes, 256);
  L.blk_bw_bytes = gnx_block_backward_workspace_bytes(h, &b, R);
  L.blk_bw = o; o += align_up(L.blk_bw_bytes, 256);
  L.part = take(pmax); L.wt = take(wmax); L.off2 = take(16);
  L.ident = take(ident_floats);  // the identity layer of a LayerNorm-first edge function (gnx_internal.h: ChainLnFirst)
  L.total = o + 256;
  return L;
}
}  // namespace

extern "C" {

size_t gnx_chain_block_backward_workspace_bytes(const gnx_graphs* h, const gnx_chain_block_params* p0, int64_t R) {
  if (!h || !p0 || R <= 0 || gnx_chain_block_workspace_bytes(h, p0, R) == 0) return 0;  // (the forward's query validates the chains)
  ChainLnFirst lnf;
  const gnx_chain_block_params* p = lnf.init(p0, nullptr);
  return chain_bw_layout(h, p, R, lnf.ident_floats()).total;
}

int32_t gnx_chain_block_backward(const gnx_graphs* h, const gnx_chain_block_params* p0, const float* ef, const float* nf, const float* gf,
                                 const float* g_ef_out, const float* g_nf_out, const float* g_gf_out, int64_t R, float* d_ef, float* d_nf,
                                 float* d_gf, const gnx_chain_block_grads* grads0, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!h || !p0) return fail(GNX_ERR_INVALID_ARG, "NULL handle or params");
  if (gnx_chain_block_workspace_bytes(h, p0, R) == 0) return GNX_ERR_DIMS;  // gnx_last_error() holds the reason
  ChainLnFirst lnf;  // a LayerNorm as the edge function's first layer: behind an identity Dense (its gradient slot is empty)
  const gnx_chain_block_params* p = lnf.init(p0, nullptr);
  const gnx_chain_block_grads* grads = lnf.init_grads(grads0);
  bool chain_wide = p->de > 32 || p->dn > 32 || p->dg > 32;
  for (const gnx_chain* c : {&p->edgefn, &p->nodefn, &p->graphfn})
    for (int i = 0; i < c->n_layers; ++i) chain_wide = chain_wide || (c->widths && c->widths[i] > 32);
  DeviceTurn turn(s, chain_wide);
  const gnx_chain* ch[3] = {&p->edgefn, &p->nodefn, &p->graphfn};
  const int de = p->de, dn = p->dn, dg = p->dg;
  const int oe = chain_out(p->edgefn), on = chain_out(p->nodefn), og = chain_out(p->graphfn);
  if ((de > 0 && !ef && h->E > 0) || (dn > 0 && !nf) || (dg > 0 && !gf)) return fail(GNX_ERR_INVALID_ARG, "an input with non-zero width is NULL");
  if (oe == 0) return fail(GNX_ERR_DIMS, "chain backward: not implemented for an edge function without output (the forward takes it; train such a block with one-layer update functions)");
  const ChainBwLayout L = chain_bw_layout(h, p, R, lnf.ident_floats());
  if (!ws || ws_bytes < L.total) return fail(GNX_ERR_WORKSPACE, "workspace missing or smaller than gnx_chain_block_backward_workspace_bytes()");
  if (int32_t rcw = gnx_ensure_wide_tables(h, stream)) return rcw;
  if (((uintptr_t)ws & 15) != 0) return fail(GNX_ERR_WORKSPACE, "workspace must be 16-byte aligned");
  char* base = static_cast<char*>(ws);
  auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
  if (lnf.on) {
    lnf.layers[0].weight = F(L.ident);
    if (int32_t rci = lnf.fill(F(L.ident), s)) return rci;
  }
  const int E = (int)h->E, N = (int)h->N, G = (int)h->G;
  const size_t rows[3] = {(size_t)R * h->E, (size_t)R * h->N, (size_t)R * h->G};
  const int trow[3] = {E, N, G};
  const int Kn = oe + dn + dg, Kg = oe + on + dg;
  const unsigned Ru = (unsigned)R;
  auto blocks = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
  float *Xn = F(L.Xn), *Xg = F(L.Xg), *dXn = F(L.dXn), *dXg = F(L.dXg), *part = F(L.part), *wt = F(L.wt);
  int* off2 = reinterpret_cast<int*>(base + L.off2);
  const gnx_dense_grad no_grad{nullptr, nullptr};
  auto grad_of = [&](int t, int i) -> const gnx_dense_grad& {
    const gnx_dense_grad* a = !grads ? nullptr : (t == 0 ? grads->edgefn : t == 1 ? grads->nodefn : grads->graphfn);
    return a ? a[i] : no_grad;
  };
  auto A = [&](int t, int i) { return F(L.act[t][i]); };
  const gnx_block_params b1 = chain_edge_block(p);
  int32_t rc;

  // ---- forward again, every layer's output kept (gnx_chain_block_forward's sequence) ----
  if (E > 0) {
    if ((rc = gnx_block_forward(h, &b1, ef, nf, gf, R, A(0, 0), nullptr, nullptr, base + L.blk_fw, L.blk_fw_bytes, 0, stream))) return rc;
    for (int i = 1; i < ch[0]->n_layers; ++i)
      if ((rc = launch_chain_layer(h, 0, ch[0]->layers[i], A(0, i - 1), ch[0]->widths[i - 1], ch[0]->widths[i], A(0, i), R, s, "bw_chain_fw_e"))) return rc;
  }
  const float* ef_out = A(0, ch[0]->n_layers - 1);
  const float* nf_out = on > 0 ? A(1, ch[1]->n_layers - 1) : nullptr;
  if (on > 0) {
    if ((rc = launch_fn_input(h, 1, ef_out, oe, nf, dn, gf, dg, R, Xn, s))) return rc;
    for (int i = 0; i < ch[1]->n_layers; ++i)
      if ((rc = launch_chain_layer(h, 1, ch[1]->layers[i], i ? A(1, i - 1) : Xn, i ? ch[1]->widths[i - 1] : Kn, ch[1]->widths[i], A(1, i), R, s, "bw_chain_fw_n"))) return rc;
  }
  if (og > 0) {
    if ((rc = launch_fn_input(h, 2, ef_out, oe, nf_out, on, gf, dg, R, Xg, s))) return rc;
    for (int i = 0; i < ch[2]->n_layers; ++i)
      if ((rc = launch_chain_layer(h, 2, ch[2]->layers[i], i ? A(2, i - 1) : Xg, i ? ch[2]->widths[i - 1] : Kg, ch[2]->widths[i], A(2, i), R, s, "bw_chain_fw_g"))) return rc;
  }

  // ---- row-wise pullback of layers [first, n) of chain t.  `cur` holds delta of the LAST layer on entry; on exit `*g_first` (rows x K_first)
  //      holds the gradient w.r.t. the input of layer `first` ----
  float* gb[3] = {F(L.gbuf[0]), F(L.gbuf[1]), F(L.gbuf[2])};
  auto delta_rows = [&](int t, int i, float* buf, const float* Ain, int K) {  // buf <- buf .* act'(layer i), all R*T rows, no extras
    const gnx_dense& d = ch[t]->layers[i];
    const int J = ch[t]->widths[i];
    const float* outp = A(t, i);
    if (d.act == GNX_ACT_GELU) {
      ProfScope ps("bw_gelu_preact", s);
      GNX_LAUNCH(k_fw_dense, blocks(rows[t] * J), dim3(256), 0, s, Ain, d.weight, d.bias, rows[t], K, J, GNX_ACT_IDENTITY, gb[2]);
      outp = gb[2];
    }
    DeltaArgs a{buf, outp, buf, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, J, (int)rows[t], 1, d.act, 0};
    ProfScope ps("bw_delta", s);
    launch_delta(a, 0, 0, 1, s);
  };
  // an entity without rows (a batch without edges): its layers' parameter gradients are sums over nothing — zeros, not what the buffers held
  auto zero_layer_grads = [&](int t, int i, int K) -> int32_t {
    const gnx_dense_grad& gz = grad_of(t, i);
    const int J = ch[t]->widths[i];
    if (gz.weight && J > 0) GNX_HIP(hipMemsetAsync(gz.weight, 0, sizeof(float) * (chain_layer_is_ln(ch[t]->layers[i]) ? (size_t)J : (size_t)J * K), s));
    if (gz.bias && J > 0) GNX_HIP(hipMemsetAsync(gz.bias, 0, sizeof(float) * (size_t)J, s));
    return GNX_OK;
  };
  auto pull_layers = [&](int t, int first, float* cur, float* other, const float* X0, int K0, float* g_first, float** result) -> int32_t {
    for (int i = ch[t]->n_layers - 1; i >= first; --i) {
      const int J = ch[t]->widths[i], K = i > 0 ? ch[t]->widths[i - 1] : K0;
      const float* Ain = i > 0 ? A(t, i - 1) : X0;
      float* gin = i == first && g_first ? g_first : other;
      if (i == first && result) *result = gin;
      const gnx_dense& d = ch[t]->layers[i];
      if (rows[t] == 0 && J > 0) { if (int32_t rz = zero_layer_grads(t, i, K)) return rz; }
      if (rows[t] == 0 || J == 0) continue;
      int32_t r2;
      if (chain_layer_is_ln(d)) {
        // a LayerNorm layer value: dx through the normalisation (k_ln_backward with one norm), dgamma = column sums of dy . xhat, dbeta = of dy
        float* t1 = gb[2];  // (free here: the gelu pre-activation buffer of delta_rows / last_delta is consumed inside those calls)
        { ProfScope ps("bw_layernorm", s);
          GNX_LAUNCH(k_ln_backward, dim3((unsigned)((rows[t] + 3) / 4)), dim3(256), 0, s, Ain, rows[t], J, d.weight, d.weight, cur, (const float*)nullptr, (const float*)nullptr,
                     kChainLnEps, chain_layer_ln_mode(d), gin, t1, (float*)nullptr); }
        GNX_LAUNCH(k_set_off2, dim3(1), dim3(1), 0, s, off2, (int)rows[t]);
        if ((r2 = colsum_all(t1, rows[t], J, grad_of(t, i).weight, part, off2, s))) return r2;
        if ((r2 = colsum_all(cur, rows[t], J, grad_of(t, i).bias, part, off2, s))) return r2;
      } else
      if (bw_use_mfma(rows[t], J, K)) {
        if (gin && K > 0 && (r2 = dx_mfma(h, t, cur, d.weight, J, K, 0, K, gin, R, wt, true, s, "bw_dx_chain"))) return r2;
        if ((r2 = dw_auto(cur, Ain, rows[t], J, K, grad_of(t, i), part, off2, s))) return r2;
      } else {
        if (gin && K > 0) launch_bw_dx(dim3(blocks(rows[t] * K).x, 1), s, cur, d.weight, (int)rows[t], J, K, gin, 0, 0, (float*)nullptr, 0);
        if ((r2 = dw_reduce(cur, Ain, rows[t], J, K, grad_of(t, i), part, s))) return r2;
      }
      if (i > first) {  // delta of layer i-1 from the gradient w.r.t. its output
        delta_rows(t, i - 1, gin, i - 1 > 0 ? A(t, i - 2) : X0, i - 1 > 0 ? ch[t]->widths[i - 2] : K0);
        std::swap(cur, other);
      }
    }
    return GNX_OK;
  };
  // delta of a chain's LAST layer: (upstream + the shares of the levels above) .* act'   [kind 0 graphs / 1 nodes / 2 edges]
  auto last_delta = [&](int t, const float* upstream, float* out, const float* X0, int K0, int act_override) {
    const int li = ch[t]->n_layers - 1, J = ch[t]->widths[li];
    const gnx_dense& d = ch[t]->layers[li];
    const int act = act_override >= 0 ? act_override : d.act;
    const float* outp = A(t, li);
    if (act == GNX_ACT_GELU) {
      const float* Ain = li > 0 ? A(t, li - 1) : X0;
      const int K = li > 0 ? ch[t]->widths[li - 1] : K0;
      ProfScope ps("bw_gelu_preact", s);
      GNX_LAUNCH(k_fw_dense, blocks(rows[t] * J), dim3(256), 0, s, Ain, d.weight, d.bias, rows[t], K, J, GNX_ACT_IDENTITY, gb[2]);
      outp = gb[2];
    }
    DeltaArgs a{upstream, outp, out, t >= 1 && og > 0 ? dXg : nullptr, Kg, t == 1 ? oe : 0, t == 0 && on > 0 ? dXn : nullptr, Kn, 0,
                t == 1 ? h->d_node_off : h->d_edge_off, h->d_edge_dst, J, trow[t], G, act, t == 2 ? 0 : (t == 1 ? 1 : 2)};
    if (t == 0) { a.ex1 = og > 0 ? dXg : nullptr; a.ex1_off = 0; }
    ProfScope ps("bw_delta", s);
    launch_delta(a, (size_t)G * Kg, (size_t)N * Kn, Ru, s);
  };

  // ---- graph chain ----
  if (og > 0) {
    last_delta(2, g_gf_out, gb[0], Xg, Kg, -1);
    if ((rc = pull_layers(2, 0, gb[0], gb[1], Xg, Kg, dXg, nullptr))) return rc;
  }
  // ---- node chain: upstream + dXg[graph of the node][oe : oe+on] ----
  if (on > 0) {
    last_delta(1, g_nf_out, gb[0], Xn, Kn, -1);
    if ((rc = pull_layers(1, 0, gb[0], gb[1], Xn, Kn, dXn, nullptr))) return rc;
  }
  // ---- edge chain: upstream + dXg[graph][0:oe] + dXn[dst][0:oe]; tail layers row-wise, the first layer through the block pullback ----
  float* g_first = gb[0];  // gradient w.r.t. the FIRST layer's output
  if (E > 0) {
    if (ch[0]->n_layers == 1) {
      last_delta(0, g_ef_out, gb[0], nullptr, 0, GNX_ACT_IDENTITY);  // the sum only: the block pullback applies the layer's own act'
    } else {
      last_delta(0, g_ef_out, gb[0], nullptr, 0, -1);
      if ((rc = pull_layers(0, 1, gb[0], gb[1], nullptr, 0, nullptr, &g_first))) return rc;  // g_first: where the last dX went
    }
  } else {
    for (int i = 1; i < ch[0]->n_layers; ++i)
      if ((rc = zero_layer_grads(0, i, ch[0]->widths[i - 1]))) return rc;  // (layer 0: the block pullback below zeroes its own)
  }
  gnx_block_grads g1{};
  g1.edgefn = grad_of(0, 0);
  if ((rc = gnx_block_backward(h, &b1, ef, nf, gf, A(0, 0), nullptr, nullptr, E > 0 ? g_first : nullptr, nullptr, nullptr, R, d_ef, d_nf, d_gf, &g1,
                               base + L.blk_bw, L.blk_bw_bytes, stream))) return rc;
  // ---- the node / graph functions' own shares of d_nf, d_gf ----
  if (d_nf && dn > 0 && on > 0 && (rc = add_cols(dXn, Kn, oe, rows[1], dn, d_nf, 1, s))) return rc;
  if (d_gf && dg > 0) {
    if (on > 0) {
      int64_t mn = 1;
      for (int64_t g = 0; g < h->G; ++g) mn = std::max(mn, h->h_node_off[g + 1] - h->h_node_off[g]);
      const int S = (int)std::min<int64_t>(std::max<int64_t>(mn / 2048, 1), 256);
      GNX_LAUNCH(k_bw_colsum1, dim3((unsigned)S * (unsigned)G, 1, Ru), dim3(256), 0, s, dXn, dg, N, h->d_node_off, S, G, part, Kn, oe + dn);
      GNX_LAUNCH(k_bw_colsum2, dim3((unsigned)G, Ru), dim3(64), 0, s, part, dg, S, G, d_gf, dg, 0, 1);
    }
    if (og > 0 && (rc = add_cols(dXg, Kg, oe + on, rows[2], dg, d_gf, 1, s))) return rc;
  }
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

}  // extern "C"

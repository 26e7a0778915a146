// GNCore at narrow widths (d <= 16, e.g. README ex.3's core_dims (10,5,3)): the parts around the block.
//   k_ln1_rows<D>  : gn1(x) — LayerNorm over the feature dim of every row (gngraphnorm.jl:19-26); one THREAD per row
//   k_core_post<D> : out = x + block_out + FeedForward(gn2(x))  (gncore.jl:56-68, gnfeedforward.jl:27-40) in one pass:
//                    gn2 is recomputed in registers (same mean/std as gn1), the 4D hidden units are produced and consumed
//                    one at a time (never stored), weights are wave-uniform scalar operands.
// At d = 10 a wave-per-row kernel keeps 54 of 64 lanes idle and the hidden tile bounced through LDS; here a row is
// D registers of one lane and the kernel is a pure stream: x in, (block_out in,) out.
#include "gnx_device.h"

namespace gnx {

namespace {

struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) F3u { float x, y, z; };
struct __attribute__((packed, aligned(4))) F2u { float x, y; };
typedef const float __attribute__((address_space(4))) * cfloatp;
__device__ __forceinline__ cfloatp as_const(const float* p) { return reinterpret_cast<cfloatp>(reinterpret_cast<uintptr_t>(p)); }

template <int D>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&x)[D]) {
  constexpr int Q = D / 4, R = D % 4;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const F4u v = *reinterpret_cast<const F4u*>(p + 4 * q);
    x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
  }
  if constexpr (R == 3) { const F3u v = *reinterpret_cast<const F3u*>(p + 4 * Q); x[4 * Q] = v.x; x[4 * Q + 1] = v.y; x[4 * Q + 2] = v.z; }
  else if constexpr (R == 2) { const F2u v = *reinterpret_cast<const F2u*>(p + 4 * Q); x[4 * Q] = v.x; x[4 * Q + 1] = v.y; }
  else if constexpr (R == 1) { x[4 * Q] = p[4 * Q]; }
}
template <int D>
__device__ __forceinline__ void store_row(float* __restrict__ p, const float (&x)[D]) {
  constexpr int Q = D / 4, R = D % 4;
#pragma unroll
  for (int q = 0; q < Q; ++q) { F4u v; v.x = x[4 * q]; v.y = x[4 * q + 1]; v.z = x[4 * q + 2]; v.w = x[4 * q + 3]; *reinterpret_cast<F4u*>(p + 4 * q) = v; }
  if constexpr (R == 3) { F3u v; v.x = x[4 * Q]; v.y = x[4 * Q + 1]; v.z = x[4 * Q + 2]; *reinterpret_cast<F3u*>(p + 4 * Q) = v; }
  else if constexpr (R == 2) { F2u v; v.x = x[4 * Q]; v.y = x[4 * Q + 1]; *reinterpret_cast<F2u*>(p + 4 * Q) = v; }
  else if constexpr (R == 1) { p[4 * Q] = x[4 * Q]; }
}

// xhat = (x - mu) * rstd over the D registers of a row; eps_mode 0: 1/(sigma+eps) (Flux 0.14 normalise), 1: 1/sqrt(var+eps)
template <int D>
__device__ __forceinline__ void normalise(float (&x)[D], float eps, int eps_mode) {
  float mu = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) mu += x[k];
  mu /= (float)D;
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) { x[k] -= mu; var = fmaf(x[k], x[k], var); }
  var /= (float)D;
  const float rstd = eps_mode == 0 ? 1.f / (sqrtf(var) + eps) : 1.f / sqrtf(var + eps);
#pragma unroll
  for (int k = 0; k < D; ++k) x[k] *= rstd;
}

}  // namespace

template <int D>
__global__ __launch_bounds__(256) void k_ln1_rows(const float* __restrict__ x, size_t rows, const float* gamma, const float* beta,
                                                  float eps, int eps_mode, float* __restrict__ y) {
  const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const cfloatp g = as_const(gamma), b = as_const(beta);
  float v[D];
  load_row<D>(x + row * D, v);
  normalise<D>(v, eps, eps_mode);
#pragma unroll
  for (int k = 0; k < D; ++k) v[k] = fmaf(g[k], v[k], b[k]);
  store_row<D>(y + row * D, v);
}

template <int D>
__global__ __launch_bounds__(256) void k_core_post(const float* __restrict__ x, size_t rows, const float* gamma2, const float* beta2,
                                                   gnx_dense fc1, gnx_dense fc2, float eps, int eps_mode, float* __restrict__ out) {
  const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  constexpr int H = 4 * D;
  const cfloatp g = as_const(gamma2), b = as_const(beta2);
  const cfloatp W1 = as_const(fc1.weight), b1 = as_const(fc1.bias), W2 = as_const(fc2.weight), b2 = as_const(fc2.bias);
  float xr[D], z[D], acc[D], blk[D];
  load_row<D>(x + row * D, xr);
  load_row<D>(out + row * D, blk);  // block(gn1(x)) written by the block forward
#pragma unroll
  for (int k = 0; k < D; ++k) z[k] = xr[k];
  normalise<D>(z, eps, eps_mode);
#pragma unroll
  for (int k = 0; k < D; ++k) { z[k] = fmaf(g[k], z[k], b[k]); acc[k] = fc2.bias ? b2[k] : 0.f; }
#pragma unroll
  for (int j = 0; j < H; ++j) {  // hidden unit j: produced and consumed in registers
    float h = fc1.bias ? b1[j] : 0.f;
#pragma unroll
    for (int k = 0; k < D; ++k) h = fmaf(W1[k * H + j], z[k], h);
    h = act_apply(h, fc1.act);
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] = fmaf(W2[j * D + i], h, acc[i]);
  }
#pragma unroll
  for (int k = 0; k < D; ++k) acc[k] = xr[k] + blk[k] + act_apply(acc[k], fc2.act);
  store_row<D>(out + row * D, acc);
}

#define GNX_CORE_WIDTHS(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)

bool core_narrow_width(int d) { return d >= 1 && d <= 16; }

int32_t launch_ln1_rows(const float* x, size_t rows, int d, const gnx_layernorm& l1, float eps, int eps_mode, float* y, hipStream_t s) {
  if (rows == 0) return GNX_OK;
  ProfScope ps("k_ln1_rows", s);
  const dim3 grid((unsigned)((rows + 255) / 256));
  switch (d) {
#define GNX_CASE(D) case D: hipLaunchKernelGGL((k_ln1_rows<D>), grid, dim3(256), 0, s, x, rows, l1.gamma, l1.beta, eps, eps_mode, y); break;
    GNX_CORE_WIDTHS(GNX_CASE)
#undef GNX_CASE
    default: return fail(GNX_ERR_DIMS, "launch_ln1_rows: width not instantiated");
  }
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

int32_t launch_core_post(const float* x, size_t rows, int d, const gnx_layernorm& l2, const gnx_ffn& ff, float eps, int eps_mode,
                         float* out, hipStream_t s) {
  if (rows == 0) return GNX_OK;
  ProfScope ps("k_core_post", s);
  const dim3 grid((unsigned)((rows + 255) / 256));
  switch (d) {
#define GNX_CASE(D) case D: hipLaunchKernelGGL((k_core_post<D>), grid, dim3(256), 0, s, x, rows, l2.gamma, l2.beta, ff.fc1, ff.fc2, eps, eps_mode, out); break;
    GNX_CORE_WIDTHS(GNX_CASE)
#undef GNX_CASE
    default: return fail(GNX_ERR_DIMS, "launch_core_post: width not instantiated");
  }
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

}  // namespace gnx

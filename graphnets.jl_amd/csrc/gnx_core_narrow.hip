// GNCore at narrow widths (d <= 16, e.g. README ex.3's core_dims (10,5,3)): the parts around the block.
//   k_ln1_rows<D>  : gn1(x) — LayerNorm over the feature dim of every row (gngraphnorm.jl:19-26); one THREAD per row
//   k_core_post<D> : out = x + block_out + FeedForward(gn2(x))  (gncore.jl:56-68, gnfeedforward.jl:27-40) in one pass:
//                    gn2 is recomputed in registers (same mean/std as gn1), the 4D hidden units are produced and consumed
//                    one at a time (never stored), weights are wave-uniform scalar operands.
// At d = 10 a wave-per-row kernel keeps 54 of 64 lanes idle and the hidden tile bounced through LDS; here a row is
// D registers of one lane and the kernel is a pure stream: x in, (block_out in,) out.
#include "gnx_device.h"
#include "gnx_wave_kernel.h"  // load_row / store_row / fma_rows / act_row

namespace gnx {

namespace {

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
  constexpr int HB = D % 2 == 0 ? 8 : 4;  // hidden units per pass (divides 4D)
  const cfloatp g = as_const(gamma2), b = as_const(beta2);
  const cfloatp W1 = as_const(fc1.weight), W2 = as_const(fc2.weight);
  const cfloatp b1 = as_const(fc1.bias ? fc1.bias : k_zero_bias), b2 = as_const(fc2.bias ? fc2.bias : k_zero_bias);
  float xr[D], z[1][D], acc[1][D], blk[D];
  load_row<D>(x + row * D, xr);
  load_row<D>(out + row * D, blk);  // block(gn1(x)) written by the block forward
#pragma unroll
  for (int k = 0; k < D; ++k) z[0][k] = xr[k];
  normalise<D>(z[0], eps, eps_mode);
#pragma unroll
  for (int k = 0; k < D; ++k) { z[0][k] = fmaf(g[k], z[0][k], b[k]); acc[0][k] = b2[k]; }
  // HB hidden units at a time, produced and consumed in registers: h = act(W1[:, j0:j0+HB]' z + b1) with W1's rows read
  // j-contiguous (W1 is (4D x D) column-major: element (j, k) at k*4D + j), then acc += W2[:, j0:j0+HB] h
#pragma unroll
  for (int j0 = 0; j0 < H; j0 += HB) {
    float h[1][HB];
#pragma unroll
    for (int jj = 0; jj < HB; ++jj) h[0][jj] = b1[j0 + jj];
    fma_rows<D, HB, 1, D, H>(W1 + j0, z, h);
    act_row<HB>(h[0], fc1.act);
    fma_rows<HB, D, 1, HB>(W2 + j0 * D, h, acc);
  }
  act_row<D>(acc[0], fc2.act);
#pragma unroll
  for (int k = 0; k < D; ++k) acc[0][k] = xr[k] + blk[k] + acc[0][k];
  store_row<D>(out + row * D, acc[0]);
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

// GNCore at narrow widths (d <= 16, e.g. README ex.3's core_dims (10,5,3)): the parts around the block.
//   k_ln1_rows<D>  : gn1(x) — LayerNorm over the feature dim of every row (gngraphnorm.jl:19-26); one THREAD per row
//   k_core_post<D> / k_core_post_s<D> : out = x + block_out + FeedForward(gn2(x))  (gncore.jl:56-68, gnfeedforward.jl:27-40) in one
//                    pass: gn2 is recomputed in registers (same mean/std as gn1), the 4D hidden units are produced and consumed
//                    in registers (never stored).  Weights: LDS broadcast (k_core_post; one row per thread, small batches) or
//                    hand-streamed scalar operands of packed FMAs (k_core_post_s; two rows per thread).
// At d = 10 a wave-per-row kernel keeps 54 of 64 lanes idle and the hidden tile bounced through LDS; here a row is
// D registers of one lane and the kernel is a pure stream: x in, (block_out in,) out.
#include "gnx_device.h"
#include "gnx_wave_kernel.h"  // load_row / store_row / fma_rows / act_row
#include "gnx_core_post_kernel.h"  // k_core_post / k_core_post_s / k_core_post3 (also compiled at run time for other width triples)

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

#ifdef GNX_CORE_FEW
#define GNX_CORE_WIDTHS(X) X(3) X(5) X(10)
#else
#define GNX_CORE_WIDTHS(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
#endif

bool core_narrow_width(int d) { return d >= 1 && d <= 16; }

int32_t launch_ln1_rows(const float* x, size_t rows, int d, const gnx_layernorm& l1, float eps, int eps_mode, float* y, hipStream_t s) {
  if (rows == 0) return GNX_OK;
  ProfScope ps("k_ln1_rows", s);
  const dim3 grid((unsigned)((rows + 255) / 256));
  switch (d) {
#define GNX_CASE(D) case D: GNX_LAUNCH((k_ln1_rows<D>), grid, dim3(256), 0, s, x, rows, l1.gamma, l1.beta, eps, eps_mode, y); break;
    GNX_CORE_WIDTHS(GNX_CASE)
#undef GNX_CASE
    default: return fail(GNX_ERR_DIMS, "launch_ln1_rows: width not instantiated");
  }
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

// rows per workgroup of the streamed two-rows-per-thread body
static size_t core_post_s_rows_per_block() { return (size_t)512 * GNX_CORE_POST_UNITS; }

int32_t launch_core_post(const float* x, size_t rows, int d, const gnx_layernorm& l2, const gnx_ffn& ff, float eps, int eps_mode,
                         float* out, hipStream_t s) {
  if (rows == 0) return GNX_OK;
  ProfScope ps("k_core_post", s);
  static const int rows_env = getenv("GNX_CORE_POST_ROWS") ? atoi(getenv("GNX_CORE_POST_ROWS")) : 2;  // (A/B switch, read once)
  const int M = rows_env == 2 && rows >= 65536 ? 2 : 1;  // two rows per thread once there are rows to spare
  static const bool lds_weights = getenv("GNX_CORE_POST_LDS") != nullptr;  // A/B: the LDS-broadcast form
  const bool trans = ff.fc1.act > GNX_ACT_RELU || ff.fc2.act > GNX_ACT_RELU;
  const bool streamed = !lds_weights && M == 2;
  const size_t per_block = streamed ? core_post_s_rows_per_block() : 256 * (size_t)M;
  const dim3 grid((unsigned)((rows + per_block - 1) / per_block));
  switch (d) {
#define GNX_CASE(D)                                                                                                                                  \
  case D:                                                                                                                                            \
    if (!lds_weights && M == 2 && trans) GNX_LAUNCH((k_core_post_s<D, true>), grid, dim3(256), 0, s, x, rows, l2.gamma, l2.beta, ff.fc1, ff.fc2, eps, eps_mode, out); \
    else if (!lds_weights && M == 2) GNX_LAUNCH((k_core_post_s<D, false>), grid, dim3(256), 0, s, x, rows, l2.gamma, l2.beta, ff.fc1, ff.fc2, eps, eps_mode, out); \
    else if (M == 2) GNX_LAUNCH((k_core_post<D, 2>), grid, dim3(256), 0, s, x, rows, l2.gamma, l2.beta, ff.fc1, ff.fc2, eps, eps_mode, out); \
    else GNX_LAUNCH((k_core_post<D, 1>), grid, dim3(256), 0, s, x, rows, l2.gamma, l2.beta, ff.fc1, ff.fc2, eps, eps_mode, out);             \
    break;
    GNX_CORE_WIDTHS(GNX_CASE)
#undef GNX_CASE
    default: return fail(GNX_ERR_DIMS, "launch_core_post: width not instantiated");
  }
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

int32_t jit_get_core_post3(int d0, int d1, int d2, hipStream_t s, hipFunction_t* fn);  // gnx_jit.cpp

// One launch for the three entities of a core: the edge and node levels have rows to spare (two rows per thread), the activations are
// identity / relu and the combined kernel exists for the width triple — ahead of time for README ex.3's (10,5,3), specialised at run
// time (hiprtc, like the fused block kernel; never inside a stream capture) for any other triple of narrow widths.  `deferred`: the
// caller wants to launch the block WITHOUT its graph update (it then runs inside this launch); the run-time specialised kernel
// exists in that form only.
bool core_post3_applies(const size_t rows[3], const int d[3], const gnx_ffn ff[3], bool deferred, hipStream_t s) {
  static const bool off = getenv("GNX_CORE_POST_SPLIT") != nullptr || getenv("GNX_CORE_POST_LDS") != nullptr;
  if (off || rows[0] < 65536 || rows[1] < 65536 || rows[2] == 0 || rows[2] >= 65536) return false;
  for (int t = 0; t < 3; ++t)
    if (ff[t].fc1.act > GNX_ACT_RELU || ff[t].fc2.act > GNX_ACT_RELU || !core_narrow_width(d[t])) return false;
  if (d[0] == 10 && d[1] == 5 && d[2] == 3) return true;
  if (!deferred) return false;
  hipFunction_t fn;
  return jit_get_core_post3(d[0], d[1], d[2], s, &fn) == GNX_OK;
}
// blk != nullptr: the block was launched without its graph update — it runs inside this launch (n_rows = partial-sum rows per replica).
// 1 = not applicable (three launches).
// skip_edges: the edge rows are final already (k_block_wave<..., FFE> ran their FeedForward and residual): the edge job gets no workgroups
int32_t launch_core_post3(const float* const x[3], const size_t rows[3], const int d[3], const gnx_layernorm l2[3], const gnx_ffn ff[3], float eps,
                          int eps_mode, float* const out[3], hipStream_t s, const BlockArgs* blk, int n_rows, bool skip_edges) {
  if (!core_post3_applies(rows, d, ff, blk != nullptr, s)) return blk ? fail(GNX_ERR_INVALID_ARG, "internal: deferred graph update without the combined kernel") : 1;
  PostJob j[3];
  for (int t = 0; t < 3; ++t) {
    const size_t per = t < 2 ? core_post_s_rows_per_block() : 256;
    j[t] = PostJob{x[t], rows[t], l2[t].gamma, l2[t].beta, ff[t].fc1, ff[t].fc2, out[t], (unsigned)((rows[t] + per - 1) / per)};
  }
  if (blk) j[2].blocks = (unsigned)rows[2];  // one workgroup per graph (and replica)
  if (skip_edges) j[0].blocks = 0;
  const dim3 grid(j[0].blocks + j[1].blocks + j[2].blocks);
  ProfScope ps("k_core_post", s);
  if (d[0] == 10 && d[1] == 5 && d[2] == 3) {
    if (blk) GNX_LAUNCH((k_core_post3<10, 5, 3, true>), grid, dim3(256), 0, s, j[0], j[1], j[2], eps, eps_mode, *blk, n_rows);
    else GNX_LAUNCH((k_core_post3<10, 5, 3, false>), grid, dim3(256), 0, s, j[0], j[1], j[2], eps, eps_mode, BlockArgs{}, 0);
  } else {
    hipFunction_t fn = nullptr;
    if (jit_get_core_post3(d[0], d[1], d[2], s, &fn) != GNX_OK) return fail(GNX_ERR_INVALID_ARG, "internal: the combined kernel of this width triple is not loaded");
    BlockArgs a = *blk;
    void* params[] = {&j[0], &j[1], &j[2], &eps, &eps_mode, &a, &n_rows};
    GNX_HIP(module_launch(fn, grid.x, 1, 1, 256, 1, 1, 0, s, params));
  }
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

}  // namespace gnx

// Matrix-core versions of the two heavy primitives of the backward pass (gnx_backward.hip) for wide feature widths:
//   dX = delta * W^T        rows x J times J x K       -> k_rows_gemm (gnx_wide.hip) on a transposed copy of W
//   dW = X^T * delta        K x J, reduced over ALL rows -> k_dw_gemm below: split over row chunks, fixed-order final sum
// Both are exact fp32 (v_mfma_f32_32x32x2_f32) with a fixed summation order, like the forward.
#include <algorithm>

#include "gnx_device.h"
#include "gnx_x6_mma.h"

namespace gnx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

int32_t launch_rows_matmul(const gnx_graphs* h, int entity, const float* A, int K, const float* B, int ldw, int OUT, float* out, int64_t R,
                           hipStream_t s, const char* name, const float* gmul, int gmul_act, float* tile_colsum, int* n_tiles_out, const float* add1);

// WT[j*K + k] = W[k*J + j]   (W = [K][J] row-major, i.e. the (J x K) column-major Dense weight)
__global__ void k_transpose_w(const float* __restrict__ W, int K, int J, float* __restrict__ WT) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= K * J) return;
  const int j = idx / K, k = idx % K;
  WT[idx] = W[(size_t)k * J + j];
}

// partial[chunk][k][j] = sum_{m in chunk} X[m][k] * D[m][j] for one 128 x 128 tile of (k, j).
// Workgroup = 4 waves, each a 64 x 64 quadrant (2 x 2 MFMA blocks); rows stream through LDS 32 at a time as TWO row-major
// panels (X[32][128], D[32][128], coalesced 16-B loads, next panel prefetched into registers during the MFMAs).  The MFMA
// reduction index is the ROW: A fragment = X[row 2s+hi][k-col l31], B fragment = D[row 2s+hi][j-col l31] — both are
// 32 consecutive floats of one LDS row per half-wave: conflict-free without padding.
template <bool VEC4, bool X6>  // X6 (the default form): six bf16 matrix-core terms per product, fragments split on the fly (gnx_x6_mma.h)
__global__ __launch_bounds__(256) void k_dw_gemm(const float* __restrict__ X, int K, const float* __restrict__ D, int J, size_t rows, int CH,
                                                 float* __restrict__ partial) {
  constexpr int RC = 32;
  __shared__ __attribute__((aligned(16))) float sX[RC * 128];
  __shared__ __attribute__((aligned(16))) float sD[RC * 128];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  const int k0 = blockIdx.y * 128, j0 = blockIdx.z * 128;
  const size_t m0 = (size_t)blockIdx.x * CH;
  const size_t m1 = m0 + CH < rows ? m0 + CH : rows;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  float4 rx[4], rd[4];
  auto load4 = [](const float* p, int avail) {  // up to 4 floats starting at p, `avail` of them inside the matrix
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (avail >= 4 && VEC4) return *reinterpret_cast<const float4*>(p);
    if (avail > 0) v.x = p[0];
    if (avail > 1) v.y = p[1];
    if (avail > 2) v.z = p[2];
    if (avail > 3) v.w = p[3];
    return v;
  };
  auto load = [&](size_t mb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 256 * i, row = q >> 5, c4 = q & 31;
      const size_t m = mb + row;
      const bool in = m < m1;
      rx[i] = load4(X + m * K + k0 + 4 * c4, in ? K - (k0 + 4 * c4) : 0);
      rd[i] = load4(D + m * J + j0 + 4 * c4, in ? J - (j0 + 4 * c4) : 0);
    }
  };
  if (m0 < m1) load(m0);
  for (size_t mb = m0; mb < m1; mb += RC) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 256 * i;
      *reinterpret_cast<float4*>(sX + 4 * q) = rx[i];
      *reinterpret_cast<float4*>(sD + 4 * q) = rd[i];
    }
    __syncthreads();
    if (mb + RC < m1) load(mb + RC);
    if constexpr (X6) {
#pragma unroll
      for (int s16 = 0; s16 < RC / 16; ++s16) {  // lane (l31, hi): rows 16 s16 + 8 hi .. + 7 of the panel, for both operands
        X6Frag fa6[2], fb6[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa6[i] = x6_frag(sX + (16 * s16 + 8 * hi) * 128 + (wm * 2 + i) * 32 + l31, 128);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb6[j] = x6_frag(sD + (16 * s16 + 8 * hi) * 128 + (wn * 2 + j) * 32 + l31, 128);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = x6_mma(fa6[i], fb6[j], acc[i][j]);
      }
    } else
#pragma unroll
    for (int sp = 0; sp < RC / 2; ++sp) {
      float fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = sX[(2 * sp + hi) * 128 + (wm * 2 + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = sD[(2 * sp + hi) * 128 + (wn * 2 + j) * 32 + l31];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }
  float* out = partial + (size_t)blockIdx.x * K * J;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int jj = j0 + (wn * 2 + j) * 32 + l31;
#pragma unroll
      for (int q = 0; q < 16; ++q) {  // C/D layout of the 32x32 MFMA: row = (q&3) + 8*(q>>2) + 4*hi, column = l31
        const int k = k0 + (wm * 2 + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * hi;
        if (k < K && jj < J) out[(size_t)k * J + jj] = acc[i][j][q];
      }
    }
}

// dW[p] = sum over chunks (in order) of partial[c][p]
__global__ void k_dw_final2(const float* __restrict__ partial, int nchunks, size_t KJ, float* __restrict__ dW) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= KJ) return;
  float acc = 0.f;
  for (int c = 0; c < nchunks; c += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)min(c + u, nchunks - 1) * KJ + p];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += c + u < nchunks ? v[u] : 0.f;
  }
  dW[p] = acc;
}

// dX = delta W^T on the matrix cores: also for narrow layers (from J*K = 64) and small batches of wide layers — the generic
// kernel puts one thread on each (row, k) with strided weight reads, the GEMM tiles read the weights coalesced and waste
// matrix-core time that nobody else wants (GNCore(10,5,3) FeedForward dX: 1.19 ms -> 0.30 ms on 1M edges).
bool bw_use_mfma(size_t rows, int J, int K) {
  static const bool off = getenv("GNX_BW_GENERIC") != nullptr;
  static const int min_w = getenv("GNX_BW_MFMA_MIN") ? atoi(getenv("GNX_BW_MFMA_MIN")) : 64;
  return !off && rows >= 64 && (size_t)J * K >= (size_t)min_w && J >= 2 && K >= 2;
}
// dW = X^T delta on the matrix cores only for real matrices: a 128 x 128 tile per row chunk is wasted on a 10 x 40 gradient
// (GNCore(10,5,3): 0.70 ms + partial traffic against 0.54 ms for the row-parallel generic reduction)
bool bw_use_mfma_dw(size_t rows, int J, int K) {
  static const bool off = getenv("GNX_BW_GENERIC") != nullptr;
  return !off && rows >= 64 && (size_t)J * K >= 1024 && J >= 8 && K >= 8;
}

// rows per workgroup of k_dw_gemm: at most 1024 chunks; small batches get 128-row chunks so that several workgroups share the rows
int dw_mfma_chunk_rows(size_t rows) { return (int)((std::max<size_t>(128, (rows + 1023) / 1024) + 31) / 32 * 32); }
size_t dw_mfma_partial_floats(size_t rows, int J, int K) {
  const size_t ch = dw_mfma_chunk_rows(rows);
  return (rows + ch - 1) / ch * (size_t)J * K;
}

// dW[k*J + j] = sum_m X[m][k] * delta[m][j] over `rows` rows (all replicas).  partial: dw_mfma_partial_floats() floats.
int32_t dw_mfma(const float* delta, const float* X, size_t rows, int J, int K, float* dW, float* partial, hipStream_t s) {
  if (!dW || rows == 0 || J == 0 || K == 0) return GNX_OK;
  const int CH = dw_mfma_chunk_rows(rows);
  const unsigned nch = (unsigned)((rows + CH - 1) / CH);
  const dim3 grid(nch, (unsigned)((K + 127) / 128), (unsigned)((J + 127) / 128));
  const bool v4 = (((uintptr_t)delta | (uintptr_t)X) & 15) == 0 && J % 4 == 0 && K % 4 == 0;
  {
    ProfScope ps("k_dw_gemm", s);
    if (form(GNX_FLAG_FP32_MFMA)) {  // (the backward's entry points carry no flags: the process-wide defaults decide; either bit selects the fp32 instruction)
      if (v4) GNX_LAUNCH((k_dw_gemm<true, false>), grid, dim3(256), 0, s, X, K, delta, J, rows, CH, partial);
      else GNX_LAUNCH((k_dw_gemm<false, false>), grid, dim3(256), 0, s, X, K, delta, J, rows, CH, partial);
    } else {
      if (v4) GNX_LAUNCH((k_dw_gemm<true, true>), grid, dim3(256), 0, s, X, K, delta, J, rows, CH, partial);
      else GNX_LAUNCH((k_dw_gemm<false, true>), grid, dim3(256), 0, s, X, K, delta, J, rows, CH, partial);
    }
  }
  const size_t KJ = (size_t)K * J;
  {
    ProfScope ps("k_dw_final2", s);
    GNX_LAUNCH(k_dw_final2, dim3((unsigned)((KJ + 255) / 256)), dim3(256), 0, s, partial, (int)nch, KJ, dW);
  }
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

// dX[:, ka:kb) = delta * W^T[:, ka:kb) on the matrix cores.  WT: K*J floats of workspace, filled here when `fill` is set.
// Optional: out *= act'(gmul) in the epilogue, and per-tile column sums of the result (tile_colsum, *n_tiles_out tiles).
int32_t dx_mfma(const gnx_graphs* h, int entity, const float* delta, const float* W, int J, int K, int ka, int kb, float* out, int64_t R,
                float* WT, bool fill, hipStream_t s, const char* name, const float* gmul, int gmul_act, float* tile_colsum, int* n_tiles_out) {
  if (kb <= ka || J == 0) return GNX_OK;
  if (fill) {
    GNX_LAUNCH(k_transpose_w, dim3((unsigned)((K * J + 255) / 256)), dim3(256), 0, s, W, K, J, WT);
    GNX_HIP(hipGetLastError());
  }
  return launch_rows_matmul(h, entity, delta, J, WT + ka, K, kb - ka, out, R, s, name, gmul, gmul_act, tile_colsum, n_tiles_out, nullptr);
}

int32_t transpose_w(const float* W, int K, int J, float* WT, hipStream_t s) {
  if (K * J == 0) return GNX_OK;
  GNX_LAUNCH(k_transpose_w, dim3((unsigned)((K * J + 255) / 256)), dim3(256), 0, s, W, K, J, WT);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

// out[rows, kb-ka) = A[rows, J] * WT[:, ka:kb) (+ add1) over the rows of one entity type; WT = transposed weights ([J][K])
int32_t rows_times_wt(const gnx_graphs* h, int entity, const float* A, int J, const float* WT, int K, int ka, int kb, float* out, const float* add1,
                      int64_t R, hipStream_t s, const char* name) {
  if (kb <= ka || J == 0) return GNX_OK;
  return launch_rows_matmul(h, entity, A, J, WT + ka, K, kb - ka, out, R, s, name, nullptr, 0, nullptr, nullptr, add1);
}

// out[n][:] = sum over t in [ptr[n], ptr[n+1]) of src[idx ? idx[t] : t][:], in order, 4 rows in flight (clamped loads);
// thread = (node, 4-column chunk).  idx == nullptr: the in-edges of a node (CSC, contiguous); idx = csr_eid: its out-edges.
template <bool VEC4>
__global__ void k_segsum_rows(const float* __restrict__ src, const int* __restrict__ ptr, const int* __restrict__ idx, int N, int E, int D,
                              float* __restrict__ out) {
  const int D4 = (D + 3) / 4;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)N * D4) return;
  const int n = (int)(gid / D4), c = 4 * (int)(gid % D4);
  const size_t r = blockIdx.y;
  const float* base = src + r * (size_t)E * D + c;
  const int t0 = ptr[n], t1 = ptr[n + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t = t0; t < t1; t += 4) {
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tt = min(t + j, t1 - 1);
      const float* p = base + (size_t)(idx ? idx[tt] : tt) * D;
      if (VEC4) {
        v[j] = *reinterpret_cast<const float4*>(p);
      } else {
        v[j].x = p[0];
        v[j].y = c + 1 < D ? p[1] : 0.f;
        v[j].z = c + 2 < D ? p[2] : 0.f;
        v[j].w = c + 3 < D ? p[3] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (t + j < t1) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
  }
  float* o = out + (r * N + n) * (size_t)D + c;
  if (VEC4) {
    *reinterpret_cast<float4*>(o) = acc;
  } else {
    o[0] = acc.x;
    if (c + 1 < D) o[1] = acc.y;
    if (c + 2 < D) o[2] = acc.z;
    if (c + 3 < D) o[3] = acc.w;
  }
}

int32_t segsum_rows(const float* src, const int* ptr, const int* idx, int N, int E, int D, int64_t R, float* out, hipStream_t s, const char* name) {
  if (N == 0 || D == 0) return GNX_OK;
  ProfScope ps(name, s);
  const size_t total = (size_t)N * ((D + 3) / 4);
  const dim3 grid((unsigned)((total + 255) / 256), (unsigned)R);
  const bool v4 = D % 4 == 0 && (((uintptr_t)src | (uintptr_t)out) & 15) == 0;
  if (v4) GNX_LAUNCH((k_segsum_rows<true>), grid, dim3(256), 0, s, src, ptr, idx, N, E, D, out);
  else GNX_LAUNCH((k_segsum_rows<false>), grid, dim3(256), 0, s, src, ptr, idx, N, E, D, out);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

// out[m][k] (+)= in[m*ld + off + k]   (k < d)
__global__ void k_add_cols(const float* __restrict__ in, int ld, int off, size_t rows, int d, float* __restrict__ out, int accumulate) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * d) return;
  const float v = in[(idx / d) * ld + off + idx % d];
  out[idx] = accumulate ? out[idx] + v : v;
}
int32_t add_cols(const float* in, int ld, int off, size_t rows, int d, float* out, int accumulate, hipStream_t s) {
  if (rows * d == 0) return GNX_OK;
  GNX_LAUNCH(k_add_cols, dim3((unsigned)((rows * d + 255) / 256)), dim3(256), 0, s, in, ld, off, rows, d, out, accumulate);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

}  // namespace gnx

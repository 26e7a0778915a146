// Fused position-wise FeedForward of a GNCore (gnfeedforward.jl:27-40, gncore.jl:56-68) with its fp32 products carried by the bf16 matrix cores:
//
//     x = hi + mid + lo      three bf16 parts hold fp32's 24 mantissa bits exactly (round-to-nearest remainders; same exponent range)
//     a*b ~ hh + hm + mh + hl + lh + mm     accumulated in fp32 by v_mfma_f32_32x32x16_bf16; the dropped terms are <= 2^-23 |a||b|
//
// gfx950 has no xf32 and its fp32 MFMA runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD); a bf16 MFMA delivers 16x that, so six of them per
// fp32 product are 2.7x less matrix-pipe time at the accuracy of the fp32 instruction (tools/mfma_emul.hip on the MI355X: worst |err| / sum|a||b|
// 8.8e-8 against 1.06e-7; tests/test_gpu_core.py compares both kernels with float64).  k_ffn_fused (fp32 MFMA) stays: GNX_FFN_FP32=1 selects it.
//
// Round 2's first kernel on this scheme read every operand fragment from LDS and was LDS-bound (32 B/clk per wave).  Here the
// operands that are reused live in REGISTERS:
//   * transposed domain: H^T = W1^T z^T, out^T = W2^T H^T — weights are the A operand, batch rows sit on the lanes (N = 32 rows per wave);
//   * a wave keeps the B fragments of its 32 z rows (all K = D, three parts: 96 registers at D = 128) for the whole tile;
//   * a 32 x 32 accumulator block of H^T (32 hidden units x the wave's rows) is, up to a permutation of k that the prepared W2 planes absorb,
//     the B-operand layout of the second product: the hidden slice is activated, split and consumed in registers and never touches LDS;
//   * the out^T accumulator (D x 32 per wave: 64 registers) lives across all 4D/32 hidden slices;
//   * only the WEIGHT fragments come from LDS — one 1-KB ds_read_b128 fragment per two MFMAs, 64 B/clk per CU of the 256 it delivers — and they
//     get there by LDS-DMA from a copy prepared once per call in fragment order (k_ffn_x6_prep: split, transposed, slot-permuted), double-buffered
//     per 32-unit hidden slice: one workgroup barrier per 96 MFMAs of every wave.
// 256 threads = 4 waves x 32 rows = 128 rows per workgroup, <= 256 registers (two waves per SIMD), 74 KB of LDS: TWO workgroups per CU, so that one's
// prologue (strided z loads, 64 splits per lane) and epilogue (residual loads, stores) run beside the other's matrix instructions — as one 8-wave
// workgroup per CU they were 50 k of a tile's 166 k clocks with an idle matrix pipe (diagnostic build -DGNX_X6_STAMPS_BUILD).
#include <algorithm>
#include <cstdio>
#include <type_traits>
#include <vector>

#include "gnx_device.h"
#include "gnx_x6_stats.h"

namespace gnx {

typedef float f32x16x __attribute__((ext_vector_type(16)));
typedef float f32x4x __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8x __attribute__((ext_vector_type(8)));

// Timing experiment only (tools/build_variant.sh x6p gnx_ffn_x6.hip -DGNX_X6_SHAPE_PROBE; results are garbage): every 32 x 32 x 16 matrix instruction
// replaced by two 16 x 16 x 32 ones on the same operand registers — the same matrix-pipe cycles, LDS reads and vector work as a 16 x 16 x 32 port of
// these kernels would have, to see what clock the chip holds on that shape (MI355X_MICROARCH.md, DVFS give-back (7)) before writing the port.
#ifdef GNX_X6_SHAPE_PROBE
__device__ __forceinline__ f32x16x x6_mfma_probe(bf16x8x a, bf16x8x b, f32x16x c) {
  f32x4x c0 = {c[0], c[1], c[2], c[3]}, c1 = {c[4], c[5], c[6], c[7]};
  c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
  c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w; c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
  return c;
}
#define GNX_X6_MFMA(a, b, c, x, y, z) x6_mfma_probe(a, b, c)
#else
#define GNX_X6_MFMA(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z)
#endif

// Round 6: the matrix instructions are v_mfma_f32_16x16x32_bf16 (GNX_X6_S16 = 1, the default) instead of v_mfma_f32_32x32x16_bf16 — the same
// matrix-pipe cycles per product (two 16-cycle instructions per 32-cycle one), LDS bytes and vector work, but the MI355X holds a higher clock on
// this shape under load (MI355X_MICROARCH.md, DVFS give-back (7)): the timing probe above measured one GNCore forward 1.59 -> 1.35 ms and config 4
// 3.56 -> 3.13 ms on the same box before the port was written (profiles/r06_ab_shape_probe.log).  Layouts of this form:
//   * lane (c, q) = (lane & 15, lane >> 4); a wave's 32 rows are two column blocks nb = 0, 1 of the transposed products: row 16 nb + c;
//   * B fragments (the rows): for k32-step s the lane holds k = 32 s + 8 q + j (j < 8) of its row 16 nb + c — zh / zm / zl [nb * KS2 + s];
//   * A fragments (weights, 1 KB = 64 lanes x 8 bf16 as before): W1 of slice hs, group g = 2 s + mb: lane (m, q), j: W1[32 s + 8 q + j][32 hs + 16 mb + m];
//     W2, group g = output block of 16: lane (m, q), j: W2[32 hs + unit(q, j)][16 g + m], unit(q, j) = 16 (j >> 2) + 4 q + (j & 3) — the C/D layout of
//     the first product (lane (c, q), block mb, register i: hidden unit 16 mb + 4 q + i of row c) IS the B operand of the second with slot j = 4 mb + i;
//   * the edge form reads k_edge_x6_prep's planes (32 x 32 x 16 fragment order) through a lane-dependent address: nothing is re-prepared.
// -DGNX_X6_S16=0 builds the 32 x 32 x 16 form of rounds 4-5 (tools/build_variant.sh) for same-box A/Bs.
#ifndef GNX_X6_S16
#define GNX_X6_S16 1
#endif
#define GNX_X6_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

#ifndef GNX_X6_ADEPTH
#define GNX_X6_ADEPTH 2
#endif
namespace {
constexpr int XR = 32;         // rows per wave
constexpr int XW = 4;          // waves per workgroup
constexpr int XBM = XR * XW;   // rows per workgroup
constexpr int XHS = 32;        // hidden units per slice

__device__ __forceinline__ void split3x(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;  // exact
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);    // exact: the second remainder has at most 8 significant bits
}
// position kk (0..31) of a slice's hidden units in the k order of the second product  <->  hidden unit of the slice: kk = 16 t + 8 h + j is
// the unit in accumulator register q = 8 t + j of lane half h, i.e. row (q & 3) + 8 (q >> 2) + 4 h of the 32 x 32 C/D layout
__host__ __device__ inline int x6_hidden_of_slot(int kk) {
  const int t = kk >> 4, h = (kk >> 3) & 1, j = kk & 7, q = 8 * t + j;
  return (q & 3) + 8 * (q >> 2) + 4 * h;
}
// S16: a 32 x 32 block of a transposed product lives in one f32x16 as four 16 x 16 blocks b = 2 mb + nb (mb: 16 outputs, nb: 16 rows), four registers each
__device__ __forceinline__ f32x4x x6_sub4(const f32x16x& v, int b) { return f32x4x{v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]}; }
__device__ __forceinline__ void x6_set4(f32x16x& v, int b, const f32x4x& x) { v[4 * b] = x.x; v[4 * b + 1] = x.y; v[4 * b + 2] = x.z; v[4 * b + 3] = x.w; }
// six-term product of one 16 x 16 block over one k32-step: small terms first (the order of the 32 x 32 x 16 form)
__device__ __forceinline__ f32x4x x6_mma16(const bf16x8x (&A)[3], const bf16x8x& bh, const bf16x8x& bm, const bf16x8x& bl, f32x4x acc) {
  acc = GNX_X6_MFMA16(A[1], bm, acc);
  acc = GNX_X6_MFMA16(A[2], bh, acc);
  acc = GNX_X6_MFMA16(A[0], bl, acc);
  acc = GNX_X6_MFMA16(A[1], bh, acc);
  acc = GNX_X6_MFMA16(A[0], bm, acc);
  acc = GNX_X6_MFMA16(A[0], bh, acc);
  return acc;
}
}  // namespace

// (row statistics in registers exist for 128-wide rows: eight k16-steps)
template <int KSX>
__device__ __forceinline__ void x6_row_stats_if(const f32x4x (&raw)[KSX][2], float eps, int eps_mode, float& mu, float& inv) {
  if constexpr (KSX == 8) x6_row_stats(raw, eps, eps_mode, mu, inv);
}
// ... in the 16 x 16 x 32 form: four k32-steps, four lanes per row
template <int KS2X>
__device__ __forceinline__ void x6_row_stats16_if(const f32x4x (&raw)[KS2X][2], float eps, int eps_mode, float& mu, float& inv) {
  if constexpr (KS2X == 4) x6_row_stats16(raw, eps, eps_mode, mu, inv);
}

// Prepared weights, per hidden slice hs (32 units) one contiguous block of 2 * NF fragments (NF = 3 D / 16) of 1 KB = 64 lanes x 8 bf16:
//   fragment 3 s + p            (s < D/16: k16-step of the first product, p: part)  lane (m, h), j: part_p( W1[16 s + 8 h + j][32 hs + m] )
//   fragment NF + 3 (2 ob + t) + p   (ob < D/32: output block, t < 2: k16-step)     lane (m, h), j: part_p( W2[32 hs + unit(16 t + 8 h + j)][32 ob + m] )
// W1 = fc1.weight ((4D x D) column-major == [D][4D] row-major), W2 = fc2.weight ((D x 4D) column-major == [4D][D] row-major).
// gamma != nullptr: W1's rows are scaled by gamma[k] — the LayerNorm in front of the FeedForward folded into its first layer (the constant
// W1^T beta joins b1: k_fold_beta)
__global__ void k_ffn_x6_prep(const float* __restrict__ W1, const float* __restrict__ W2, int D, __bf16* __restrict__ Wp, const float* __restrict__ gamma) {
  const int H = 4 * D, NF = 3 * D / 16;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (hs, g, lane, j): g < D/16 groups of three fragments per product
  if (idx >= D * H) return;
  const int j = idx & 7, lane = (idx >> 3) & 63, g = (idx >> 9) % (D / 16), hs = (idx >> 9) / (D / 16);
  __bf16* slice = Wp + (size_t)hs * 2 * NF * 512;
  __bf16 a, b, c;
#if GNX_X6_S16
  {
    const int m16 = lane & 15, q = lane >> 4;
    {  // W1, group g = 2 s + mb
      const int s32 = g >> 1, mb = g & 1, k = 32 * s32 + 8 * q + j;
      split3x((gamma ? gamma[k] : 1.f) * W1[(size_t)k * H + 32 * hs + 16 * mb + m16], a, b, c);
      __bf16* f = slice + (size_t)(3 * g) * 512 + lane * 8 + j;
      f[0] = a; f[512] = b; f[1024] = c;
    }
    {  // W2, group g = output block of 16; slot (q, j) of the k32-step = hidden unit 16 (j >> 2) + 4 q + (j & 3) of the slice
      const int n = 32 * hs + 16 * (j >> 2) + 4 * q + (j & 3);
      split3x(W2[(size_t)n * D + 16 * g + m16], a, b, c);
      __bf16* f = slice + (size_t)(NF + 3 * g) * 512 + lane * 8 + j;
      f[0] = a; f[512] = b; f[1024] = c;
    }
    return;
  }
#endif
  const int m = lane & 31, h = lane >> 5;
  {
    const int k = 16 * g + 8 * h + j;
    split3x((gamma ? gamma[k] : 1.f) * W1[(size_t)k * H + 32 * hs + m], a, b, c);
    __bf16* f = slice + (size_t)(3 * g) * 512 + lane * 8 + j;
    f[0] = a; f[512] = b; f[1024] = c;
  }
  {
    const int ob = g >> 1, t = g & 1;
    const int n = 32 * hs + x6_hidden_of_slot(16 * t + 8 * h + j);
    split3x(W2[(size_t)n * D + 32 * ob + m], a, b, c);
    __bf16* f = slice + (size_t)(NF + 3 * g) * 512 + lane * 8 + j;
    f[0] = a; f[512] = b; f[1024] = c;
  }
}

// EDGE form (GNCore, D = 128): the projected edge update of the same rows — k_edge_x6's phase (gnx_edge_x6.hip) — runs behind the FeedForward, whose
// result waits in the out^T accumulator: ef' is added there, slice by slice, and never reaches memory.
// Both LayerNorms of the core's edge rows are FOLDED into the weight planes (round 5): gn1(x) = g1 . xhat + b1' and gn2(x) = g2 . xhat + b2' share
// xhat = (x - mean) / (sigma + eps), so We^T gn1(x) = (g1 . We)^T xhat + We^T b1' and W1^T gn2(x) + bias1 = (g2 . W1)^T xhat + (W1^T b2' + bias1): the
// preparation kernels scale the weight rows (k_edge_x6_prep / k_ffn_x6_prep with gamma) and k_fold_beta makes the two constant vectors; the kernel
// normalises and splits the rows ONCE — the same fragments feed the FeedForward's first product and the edge update; round 4 reloaded, re-normalised and
// re-split the rows between the two (11 k of a tile's 180 k clocks and a second pass over x)
struct FfnX6Edge {
  const Tile* tiles;       // the handle's edge tiles (<= 128 rows): one workgroup each
  const float* c1;         // [128] We_e^T beta1: what gn1's shift leaves behind once its scale is folded into the weight planes
  const __bf16* Wpe;       // k_edge_x6_prep's fragments of (gamma1 . We)
  const float* psrc;       // [R][N][128]
  const float* pdst;       // [R][N][128] (bias and gf fold included)
  size_t N;
  const int* src;          // rowval [E]
  const int* dst;          // edge_dst [E]
  int act;
  float* colsum;           // [R][n_tiles][128] or nullptr
  size_t n_tiles;
  float* agg_out;          // [R][n_agg_rows][128] or nullptr
  size_t n_agg_rows;
  const int* chunk_row0;   // [2 n_tiles + 1]
};

struct FfnX6Args {
  const float* z;          // [R][rows][D]: gn2(x), or x itself with ln_stats (normalised on load)
  const __bf16* Wp;        // prepared weights (k_ffn_x6_prep)
  const float* b1;         // [4D] or nullptr
  const float* b2;         // [D] or nullptr
  const float* add1;       // [R][rows][D] or nullptr
  const float* add2;
  float* out;
  size_t rows;             // rows per replica
  int act1;
  const float* ln_stats;   // [R][rows][2] (mean, 1/sigma) from k_ln_stats_v4, or nullptr
  const float* ln_g;
  const float* ln_b;
  int ln_inline;           // D = 128: no ln_stats — the row statistics are computed here, in registers (gnx_x6_stats.h), with ln_eps / ln_mode
  float ln_eps;
  int ln_mode;
  FfnX6Edge e;             // EDGE instantiation only
};

// Diagnostic build only (tools/build_variant.sh x6st gnx_ffn_x6.hip -DGNX_X6_STAMPS_BUILD; run with GNX_X6_STAMPS=1, GNX_X6_STAMP_WAVE=<0..3>):
// shader-clock stamps of ONE wave of every workgroup, written straight to a buffer of their own (32 per workgroup; no output is computed from them,
// no register array kept for them); the launchers print the average distance between consecutive stamps.  In the real build no stamp executes.
#ifdef GNX_X6_STAMPS_BUILD
constexpr int XNST = 32;
static __device__ unsigned long long* g_x6_dbg = nullptr;  // [workgroup][XNST]
static __device__ int g_x6_wave = 0;
#define GNX_XSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (xst_p) xst_p[i] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)

// stamp buffer of a launch of n_wg workgroups (nullptr: stamps off)
static unsigned long long* x6_stamps_begin(size_t n_wg, hipStream_t s) {
  static unsigned long long* d_dbg = nullptr;
  static size_t dbg_cap = 0;
  if (!getenv("GNX_X6_STAMPS")) return nullptr;
  if (dbg_cap < n_wg) { if (d_dbg) (void)hipFree(d_dbg); dbg_cap = n_wg; (void)hipMalloc((void**)&d_dbg, dbg_cap * XNST * 8); }
  (void)hipMemsetAsync(d_dbg, 0, n_wg * XNST * 8, s);
  const int wave = getenv("GNX_X6_STAMP_WAVE") ? atoi(getenv("GNX_X6_STAMP_WAVE")) : 0;
  (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_x6_dbg), &d_dbg, sizeof(d_dbg), 0, hipMemcpyHostToDevice, s);
  (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_x6_wave), &wave, sizeof(wave), 0, hipMemcpyHostToDevice, s);
  return d_dbg;
}

static void x6_stamps_end(const unsigned long long* d_dbg, size_t n_wg, const char* what, const char* const* names, hipStream_t s) {
  if (!d_dbg) return;
  (void)hipStreamSynchronize(s);
  std::vector<unsigned long long> hs(n_wg * XNST);
  (void)hipMemcpy(hs.data(), d_dbg, hs.size() * 8, hipMemcpyDeviceToHost);
  double sum[XNST] = {0};
  size_t cnt[XNST] = {0};
  unsigned long long t_min = ~0ull, t_max = 0;
  for (size_t w = 0; w < n_wg; ++w) {
    unsigned long long prev = 0;
    for (int i = 0; i < XNST; ++i) {
      const unsigned long long t = hs[w * XNST + i];
      if (!t) continue;
      if (prev) { sum[i] += (double)(t - prev); ++cnt[i]; }
      prev = t; t_min = std::min(t_min, t); t_max = std::max(t_max, t);
    }
  }
  fprintf(stderr, "[gnx x6 stamps, %s] %zu workgroups, clocks from the previous stamp (average):", what, n_wg);
  double tot = 0;
  for (int i = 1; i < XNST; ++i)
    if (cnt[i]) { fprintf(stderr, "  %s %.0f", names && names[i] ? names[i] : "?", sum[i] / cnt[i]); tot += sum[i] / cnt[i]; }
  fprintf(stderr, "  | sum %.0f;  first stamp -> last stamp %llu\n", tot, t_max - t_min);
}
#else
#define GNX_XSTAMP(i) do { } while (0)
#endif

// TRANS: fc1's activation — in the EDGE form: or the edge function's — is tanh / sigmoid / gelu (the run-time switch of act_apply); else both are identity / relu
// EDGE: see FfnX6Edge
template <int D, bool TRANS, bool EDGE>
__global__ __launch_bounds__(64 * XW) __attribute__((amdgpu_waves_per_eu(2, D == 128 ? 2 : 3))) void k_ffn_x6(FfnX6Args a) {
  static_assert(!EDGE || D == 128, "the edge update in front of the FeedForward is the 128 -> 128 form");
  constexpr int H = 4 * D;
  constexpr int KS = D / 16;          // k16-steps of the first product
  constexpr int NOB = D / 32;         // 32-output blocks of the second product
  constexpr int NF = 3 * KS;          // fragments per product and slice
  constexpr int SLB = 2 * NF * 1024;  // bytes per slice
  constexpr int NSL = H / XHS;
  constexpr bool S16 = GNX_X6_S16 != 0;  // the 16 x 16 x 32 form (see the top of the file)
  constexpr int KS2 = D / 32;         // k32-steps of the first product in that form
  static_assert(NF % XW == 0, "fragments of half a slice divide over the waves");
  // W1 and W2 fragments of a slice travel separately: while a wave multiplies slice hs + 1's W1 fragments (first product) it splits slice hs's
  // hidden block, then multiplies it (second product) — see the loop.  ONE W1 buffer (restaged behind a barrier in the middle of the step) and two
  // W2 buffers: 72 KB.  Three OBJECTS, and a loop unrolled by two: the compiler orders an LDS read behind every LDS-DMA that may alias it —
  // through one array it waits for the NEXT pieces in front of this step's first fragment.
  __shared__ __attribute__((aligned(16))) unsigned char s_w1[SLB / 2];
  __shared__ __attribute__((aligned(16))) unsigned char s_w2a[SLB / 2];
  __shared__ __attribute__((aligned(16))) unsigned char s_w2b[SLB / 2];
  __shared__ float s_b1[H];
  __shared__ int s_src[EDGE ? XBM : 1], s_dst[EDGE ? XBM : 1];
  __shared__ int s_seg[2][EDGE ? 68 : 1];  // per 64-row pass: first row of every destination run; [n_seg] = valid rows of the pass; [65] = n_seg; [66] = the pass's first row of the partial-sum table
  __shared__ __attribute__((aligned(16))) float s_bc[EDGE ? 2 * D : 4];  // EDGE: b2 | We^T beta1 — the slices' constant addends (a global load behind each slice's wait was a round trip of its own)

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef GNX_X6_STAMPS_BUILD
  unsigned long long* xst_p = g_x6_dbg && lane == 0 && wv == g_x6_wave && blockIdx.y == 0 ? g_x6_dbg + (size_t)blockIdx.x * XNST : nullptr;
#endif
  GNX_XSTAMP(0);
  const int hi = lane >> 5, n = lane & 31;
  const int c16 = lane & 15, q16 = lane >> 4;  // S16: column (row of the batch) within a 16-block, k / output quarter
  const size_t r = blockIdx.y;
  const size_t rows = a.rows;  // per replica
  size_t wg_row0 = (size_t)blockIdx.x * XBM, rend = rows;
  if constexpr (EDGE) {
    const Tile t = a.e.tiles[blockIdx.x];
    if (t.e1 <= t.e0) return;  // (whole workgroup)
    wg_row0 = (size_t)t.e0; rend = (size_t)t.e1;
  }
  const size_t row0 = wg_row0 + (size_t)wv * XR;
  // (waves beyond the last row keep working on the clamped last row — they share the barriers — and store nothing)
  const size_t rown = row0 + n < rend ? row0 + n : rend - 1;
  const float* __restrict__ zrow = a.z + (r * rows + rown) * D;
  // S16: the lane's two rows (column blocks nb = 0, 1 of the wave's 32 rows)
  const size_t rown16[2] = {row0 + c16 < rend ? row0 + c16 : rend - 1, row0 + 16 + c16 < rend ? row0 + 16 + c16 : rend - 1};

  // fragment f of a slice is LDS-DMA piece f (lane l writes bytes [16 l, 16 l + 16) of the piece); which = 0: the W1 half of the slice, 1: the W2 half
  auto stage = [&](int hs, int which, unsigned char* dst) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(a.Wp) + (size_t)hs * SLB + (size_t)which * (SLB / 2);
#pragma unroll
    for (int i = 0; i < NF / XW; ++i) {
      const int pc = wv + XW * i;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)pc * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
    }
  };
  // ONE piece (i < NF / XW) of such a half slice — the FeedForward's loop issues its pieces one per group of six matrix instructions: as a burst of six
  // behind a barrier they were ~600 clocks of issue in front of an idle matrix pipe, twice per hidden slice (stamps, one workgroup per CU: 70 k clocks
  // for a FeedForward phase whose 1536 matrix instructions take 49 k)
  auto stage_piece = [&](int hs, int which, unsigned char* dst, int i) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(a.Wp) + (size_t)hs * SLB + (size_t)which * (SLB / 2) + (size_t)(wv + XW * i) * 1024;  // (wave-uniform)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (unsigned)(lane * 16)),
                                     (__attribute__((address_space(3))) void*)(dst + (wv + XW * i) * 1024), 16, 0, 0);
  };
  // the edge update's weight block of a 32-output slice: a W1-sized set of fragments (K = D)
  auto stage_e = [&](int ob, unsigned char* dst) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(a.e.Wpe) + (size_t)ob * (SLB / 2);
#pragma unroll
    for (int i = 0; i < NF / XW; ++i) {
      const int pc = wv + XW * i;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)pc * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
    }
  };
  stage(0, 0, s_w1);
  stage(0, 1, s_w2a);
  if constexpr (EDGE) {
    if (tid < XBM) {
      const int trows = (int)(rend - wg_row0), rc = tid < trows ? tid : trows - 1;
      s_src[tid] = a.e.src[wg_row0 + rc];
      s_dst[tid] = a.e.dst[wg_row0 + rc];
    }
  }
  for (int i = tid; i < H; i += 64 * XW) s_b1[i] = a.b1 ? a.b1[i] : 0.f;
  if constexpr (EDGE) {
    if (tid < D) { s_bc[tid] = a.b2 ? a.b2[tid] : 0.f; s_bc[D + tid] = a.e.c1[tid]; }
  }

  // ---- the wave's z rows as B fragments: lane (n, hi) holds k = 16 s + 8 hi + j (j < 8) of row n for every k16-step s, in three parts ----
  bf16x8x zh[KS], zm[KS], zl[KS];  // (S16: index nb * KS2 + s)
  float mu = 0.f, inv = 1.f;
  // load the wave's rows; stats: their statistics from the registers (gnx_x6_stats.h); ln: normalise (and, outside the EDGE form, scale and shift with (g, b)); split
  auto load_z = [&](const float* __restrict__ g, const float* __restrict__ b, bool ln, bool stats) {
    if constexpr (S16) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float* __restrict__ zr = a.z + (r * rows + rown16[nb]) * D;
        f32x4x raw[KS2][2];
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
          raw[s][0] = *reinterpret_cast<const f32x4x*>(zr + 32 * s + 8 * q16);
          raw[s][1] = *reinterpret_cast<const f32x4x*>(zr + 32 * s + 8 * q16 + 4);
        }
        float mu_ = 0.f, inv_ = 1.f;
        if (a.ln_stats != nullptr && !stats) {
          const float2 st = reinterpret_cast<const float2*>(a.ln_stats)[r * rows + rown16[nb]];
          mu_ = st.x; inv_ = st.y;
        }
        if (stats) x6_row_stats16_if(raw, a.ln_eps, a.ln_mode, mu_, inv_);
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
          float v[8] = {raw[s][0].x, raw[s][0].y, raw[s][0].z, raw[s][0].w, raw[s][1].x, raw[s][1].y, raw[s][1].z, raw[s][1].w};
          if (ln) {
            if constexpr (EDGE) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = (v[j] - mu_) * inv_;
            } else {
              const f32x4x g0 = *reinterpret_cast<const f32x4x*>(g + 32 * s + 8 * q16), g1 = *reinterpret_cast<const f32x4x*>(g + 32 * s + 8 * q16 + 4);
              const f32x4x b0 = *reinterpret_cast<const f32x4x*>(b + 32 * s + 8 * q16), b1 = *reinterpret_cast<const f32x4x*>(b + 32 * s + 8 * q16 + 4);
              const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = fmaf(gg[j], (v[j] - mu_) * inv_, bb[j]);
            }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            __bf16 x, y, w;
            split3x(v[j], x, y, w);
            zh[nb * KS2 + s][j] = x; zm[nb * KS2 + s][j] = y; zl[nb * KS2 + s][j] = w;
          }
        }
      }
      return;
    }
    f32x4x raw[KS][2];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      raw[s][0] = *reinterpret_cast<const f32x4x*>(zrow + 16 * s + 8 * hi);
      raw[s][1] = *reinterpret_cast<const f32x4x*>(zrow + 16 * s + 8 * hi + 4);
    }
    if (stats) x6_row_stats_if(raw, a.ln_eps, a.ln_mode, mu, inv);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      float v[8] = {raw[s][0].x, raw[s][0].y, raw[s][0].z, raw[s][0].w, raw[s][1].x, raw[s][1].y, raw[s][1].z, raw[s][1].w};
      if (ln) {  // (x - mean) * inv, then fma(gamma, ., beta): the arithmetic of k_layernorm2_v4 / k_ffn_fused
        if constexpr (EDGE) {  // scale and shift live in the prepared weight planes: xhat itself is what both products read
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (v[j] - mu) * inv;
        } else {
          const f32x4x g0 = *reinterpret_cast<const f32x4x*>(g + 16 * s + 8 * hi), g1 = *reinterpret_cast<const f32x4x*>(g + 16 * s + 8 * hi + 4);
          const f32x4x b0 = *reinterpret_cast<const f32x4x*>(b + 16 * s + 8 * hi), b1 = *reinterpret_cast<const f32x4x*>(b + 16 * s + 8 * hi + 4);
          const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaf(gg[j], (v[j] - mu) * inv, bb[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __bf16 x, y, w;
        split3x(v[j], x, y, w);
        zh[s][j] = x; zm[s][j] = y; zl[s][j] = w;
      }
    }
  };
  if constexpr (EDGE) {
    load_z(nullptr, nullptr, true, true);  // xhat = (x - mean) / (sigma + eps): the operand of the FeedForward's first product AND of the edge update
  } else {
    const bool ln_in = D == 128 && a.ln_inline != 0;
    if (a.ln_stats != nullptr) {
      const float2 st = reinterpret_cast<const float2*>(a.ln_stats)[r * rows + rown];
      mu = st.x; inv = st.y;
    }
    load_z(a.ln_g, a.ln_b, a.ln_stats != nullptr || ln_in, ln_in);
  }

  f32x16x accO[NOB];
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int q = 0; q < 16; ++q) accO[ob][q] = 0.f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of W1(0), W2(0) are in LDS
  __syncthreads();                                  // ... and everybody else's; s_b1 too

  // Activation + split of a finished H^T block, one PAIR of accumulator registers (q = 2 u, 2 u + 1; q = 8 t + j: element j of the B fragment of
  // k16-step t, slot 16 t + 8 hi + j) at a time, as bit operations on dwords (14 vector instructions per pair):
  //   X(u): relu as an INTEGER maximum of the float's bits with 0 (identity: with INT_MIN) — one instruction, no canonicalisation; hi = one
  //         v_cvt_pk_bf16_f32 of the pair = dword u of the hi fragment; its two floats by shift / mask; first remainders
  //   Y(u): mid = cvt of the remainders; second remainders; lo = their cvt
  // and as a two-stage pipeline over the k16-steps of the first product: step s runs X(s) beside Y(s - 1), written line by line alternately — two
  // independent dependency chains, so that consecutive vector instructions between the matrix instructions do not wait for each other.
  const int relu_floor = a.act1 == 1 ? 0 : (int)0x80000000;
  unsigned hhw[8], hmw[8], hlw[8];
  auto cvt2 = [](float x0, float x1) -> unsigned {
    typedef float f32x2x __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2x __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2x{x0, x1}, bf16x2x));
  };
  auto relu_bits = [&](float v) -> float {
    if constexpr (TRANS) return act_apply(v, a.act1);
    else return __int_as_float(max(__float_as_int(v), relu_floor));
  };
  // one k16-step's share: X(ux) if ux >= 0, Y(uy) if uy >= 0; rx / ry: the first remainders of the pair in flight
  // S16: pair U = 4 nb + w is slots j = 2 w, 2 w + 1 of column block nb's B fragment: registers 2 (w & 1), + 1 of block 2 (w >> 1) + nb
  auto pair_reg = [](int u) -> int { return S16 ? 4 * (2 * ((u & 3) >> 1) + (u >> 2)) + 2 * (u & 1) : 2 * u; };
  auto split_xy = [&](int ux, int uy, const f32x16x& accH, float (&rx)[2], const float (&ry)[2]) {
    float m0 = 0.f, m1 = 0.f, q0, q1;
    unsigned hb = 0, mb = 0;
    if (ux >= 0) { m0 = relu_bits(accH[pair_reg(ux)]); }
    if (uy >= 0) { mb = cvt2(ry[0], ry[1]); hmw[uy] = mb; }
    if (ux >= 0) { m1 = relu_bits(accH[pair_reg(ux) + 1]); }
    if (uy >= 0) { q0 = ry[0] - __uint_as_float(mb << 16); }
    if (ux >= 0) { hb = cvt2(m0, m1); hhw[ux] = hb; }
    if (uy >= 0) { q1 = ry[1] - __uint_as_float(mb & 0xffff0000u); }
    if (ux >= 0) { rx[0] = m0 - __uint_as_float(hb << 16); }
    if (uy >= 0) { hlw[uy] = cvt2(q0, q1); }
    if (ux >= 0) { rx[1] = m1 - __uint_as_float(hb & 0xffff0000u); }
  };
  typedef unsigned u32x4x __attribute__((ext_vector_type(4)));
  auto frag = [](const unsigned (&w)[8], int t) -> bf16x8x { return __builtin_bit_cast(bf16x8x, u32x4x{w[4 * t], w[4 * t + 1], w[4 * t + 2], w[4 * t + 3]}); };
  // H^T block of slice hs (its 32 hidden units x the wave's 32 rows) = b1 + W1^T z^T into accN — and, between its matrix instructions, activation
  // and split of the FINISHED block accC (SPLIT): two independent streams in one schedule.  Per k16-step: the next step's three weight fragments
  // requested, six MFMAs, 13 vector instructions of the split: every matrix instruction is followed by two of them — issued while the pipe works.
  // (Left to the compiler the split is one block of ~170 vector instructions between the two products, in front of an idle matrix pipe.)
  auto gemm1 = [&](auto split_c, int hs, const unsigned char* w1, f32x16x& accN, const f32x16x& accC, auto&& piece) {
    constexpr bool SPLIT = decltype(split_c)::value;
    const unsigned char* wb = w1 + lane * 16;
#pragma unroll
    for (int q = 0; q < 16; ++q) accN[q] = S16 ? s_b1[hs * XHS + 16 * (q >> 3) + 4 * q16 + (q & 3)]  // block 2 mb + nb = q >> 2: unit 16 mb + 4 q16 + i
                                               : s_b1[hs * XHS + (q & 3) + 8 * (q >> 2) + 4 * hi];
    constexpr int AD = GNX_X6_ADEPTH;  // weight fragments in flight: AD - 1 k16-steps ahead of their MFMAs
    bf16x8x A[AD][3];
#pragma unroll
    for (int d = 0; d < AD - 1; ++d)
#pragma unroll
      for (int p3 = 0; p3 < 3; ++p3) A[d][p3] = *reinterpret_cast<const bf16x8x*>(wb + (3 * d + p3) * 1024);
    constexpr int PPS = 8 / KS;  // register pairs of the finished block split per k16-step (D = 128: one, D = 64: two)
    float rr[2][PPS][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < PPS; ++j) rr[i][j][0] = rr[i][j][1] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c = s % AD, cr = s & 1;
      if (s + AD - 1 < KS) {
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) A[(s + AD - 1) % AD][p3] = *reinterpret_cast<const bf16x8x*>(wb + (3 * (s + AD - 1) + p3) * 1024);
      }
      constexpr bool PIECE = !std::is_same<std::decay_t<decltype(piece)>, std::nullptr_t>::value;
      if constexpr (PIECE) { if (s < NF / XW) piece(s); }
      if constexpr (S16) {  // group s = 2 s32 + mb: the three fragments of (k32-step s32, output block mb) against both column blocks
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int b = 2 * (s & 1) + nb, zi = nb * KS2 + (s >> 1);
          x6_set4(accN, b, x6_mma16(A[c], zh[zi], zm[zi], zl[zi], x6_sub4(accN, b)));
        }
      } else {
      accN = GNX_X6_MFMA(A[c][1], zm[s], accN, 0, 0, 0);  // small terms first
      accN = GNX_X6_MFMA(A[c][2], zh[s], accN, 0, 0, 0);
      accN = GNX_X6_MFMA(A[c][0], zl[s], accN, 0, 0, 0);
      accN = GNX_X6_MFMA(A[c][1], zh[s], accN, 0, 0, 0);
      accN = GNX_X6_MFMA(A[c][0], zm[s], accN, 0, 0, 0);
      accN = GNX_X6_MFMA(A[c][0], zh[s], accN, 0, 0, 0);
      }
      if constexpr (SPLIT) {
        // (instruction selection places pure vector instructions wherever their operands are ready — Y(s - 1) right behind X(s - 1), in the
        // previous step's region, one dependent chain again; the empty volatile statement pins the remainders to THIS region)
#pragma unroll
        for (int j = 0; j < PPS; ++j) {
          if (s > 0) asm volatile("" : "+v"(rr[cr ^ 1][j][0]), "+v"(rr[cr ^ 1][j][1]));
          split_xy(PPS * s + j, s > 0 ? PPS * (s - 1) + j : -1, accC, rr[cr][j], rr[cr ^ 1][j]);
        }
      }
      if (s + AD - 1 < KS) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      if constexpr (S16) {  // twelve 16-cycle matrix instructions per group: the same 13 (26) vector instructions of the split spread one (two) per gap
#pragma unroll
        for (int i = 0; i < 11; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if constexpr (PIECE) { if (i == 0 && s < NF / XW) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }
          if constexpr (SPLIT) { if constexpr (PPS == 1) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); else __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (SPLIT) { if constexpr (PPS == 1) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); else __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); }
      } else {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (PIECE) { if (i == 0 && s < NF / XW) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }  // the step's LDS-DMA piece behind its first matrix instruction
        if constexpr (SPLIT) { if constexpr (PPS == 1) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); else __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); }
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if constexpr (SPLIT) { if constexpr (PPS == 1) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); else __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (SPLIT) {  // Y of the last pair(s) (beside the first matrix instructions of the second product)
#pragma unroll
      for (int j = 0; j < PPS; ++j) split_xy(-1, PPS * (KS - 1) + j, accC, rr[0][j], rr[(KS - 1) & 1][j]);
    }
  };
  // out^T (D outputs x the wave's rows) += W2^T[:, the slice's slots] H^T
  auto gemm2 = [&](const unsigned char* w2, auto&& piece) {
    constexpr bool PIECE = !std::is_same<std::decay_t<decltype(piece)>, std::nullptr_t>::value;
    const unsigned char* wb = w2 + lane * 16;
    const bf16x8x hh[2] = {frag(hhw, 0), frag(hhw, 1)}, hm[2] = {frag(hmw, 0), frag(hmw, 1)}, hl[2] = {frag(hlw, 0), frag(hlw, 1)};
    bf16x8x A[2][3];  // the fragments of group g + 1 are requested in front of the six MFMAs of group g (g = 2 ob + t)
#pragma unroll
    for (int p3 = 0; p3 < 3; ++p3) A[0][p3] = *reinterpret_cast<const bf16x8x*>(wb + p3 * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 2 * NOB; ++g) {
      const int c = g & 1, ob = g >> 1, t = g & 1;
      if (g + 1 < 2 * NOB) {
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) A[c ^ 1][p3] = *reinterpret_cast<const bf16x8x*>(wb + (3 * (g + 1) + p3) * 1024);
      }
      if constexpr (PIECE) { if (g < NF / XW) piece(g); }
      constexpr int NMM = S16 ? 12 : 6;  // matrix instructions of the group
      if constexpr (S16) {  // group g = output block of 16 (block t = g & 1 of accO[ob]); the slice's 32 hidden units are ONE k32-step: hh / hm / hl [nb]
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) x6_set4(accO[ob], 2 * t + nb, x6_mma16(A[c], hh[nb], hm[nb], hl[nb], x6_sub4(accO[ob], 2 * t + nb)));
      } else {
      accO[ob] = GNX_X6_MFMA(A[c][1], hm[t], accO[ob], 0, 0, 0);
      accO[ob] = GNX_X6_MFMA(A[c][2], hh[t], accO[ob], 0, 0, 0);
      accO[ob] = GNX_X6_MFMA(A[c][0], hl[t], accO[ob], 0, 0, 0);
      accO[ob] = GNX_X6_MFMA(A[c][1], hh[t], accO[ob], 0, 0, 0);
      accO[ob] = GNX_X6_MFMA(A[c][0], hm[t], accO[ob], 0, 0, 0);
      accO[ob] = GNX_X6_MFMA(A[c][0], hh[t], accO[ob], 0, 0, 0);
      }
      if (g + 1 < 2 * NOB) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      if constexpr (PIECE) {
        if (g < NF / XW) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, NMM - 1, 0); }
        else __builtin_amdgcn_sched_group_barrier(0x008, NMM, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, NMM, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  GNX_XSTAMP(1);  // prologue done: z rows split, first weight pieces in LDS
  static_assert(KS == 8 || KS == 4, "one or two pairs of the 16 hidden registers are split per k16-step of the first product");
  f32x16x accA, accB;  // H^T blocks: the one being produced and the one being consumed, alternating
  gemm1(std::false_type{}, 0, s_w1, accA, accA, nullptr);
  __syncthreads();  // every wave is done with W1(0)
  stage(1, 0, s_w1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // step hs:  W2(hs + 1) requested  |  first product of slice hs + 1 (W1 buffer) with the split of slice hs between its MFMAs  |  barrier: W1 buffer
  //           free -> W1(hs + 2) requested  |  second product of slice hs (W2 from w2c)  |  the requests have landed, barrier
  auto step = [&](int hs, const unsigned char* w2c, unsigned char* w2s, f32x16x& accC, f32x16x& accN) {
    gemm1(std::true_type{}, hs + 1, s_w1, accN, accC, [&](int i) { stage_piece(hs + 1, 1, w2s, i); });
    __syncthreads();
    // (the last step has no W1(hs + 2): it requests W1(NSL - 1) again — the bytes the buffer holds already, read by nobody — so that the pieces sit
    //  in straight-line code between the matrix instructions)
    const int hs2 = hs + 2 < NSL ? hs + 2 : NSL - 1;
    gemm2(w2c, [&](int i) { stage_piece(hs2, 0, s_w1, i); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  static_assert(NSL % 2 == 0, "slice loop unrolled by two");
  for (int hs = 0; hs + 2 < NSL; hs += 2) {
    step(hs, s_w2a, s_w2b, accA, accB);
    step(hs + 1, s_w2b, s_w2a, accB, accA);
  }
  step(NSL - 2, s_w2a, s_w2b, accA, accB);
  {  // the last slice: nothing left to produce
    float rr[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int u = 0; u <= 8; ++u) split_xy(u < 8 ? u : -1, u - 1, accB, rr[u & 1], rr[(u & 1) ^ 1]);
    gemm2(s_w2b, nullptr);
  }

  GNX_XSTAMP(2);  // all slices done
  if constexpr (EDGE) {
    // ---- the projected edge update of the tile BEHIND its FeedForward: ef' = act(We^T gn1(x) + Ps[src] + Pd[dst]) per 32-output slice (k_edge_x6's
    //      scheme: weight slices double-buffered in the two W2 buffers, the staged block / the column-sum partials in the W1 buffer), its per-destination
    //      sums and column sums written as k_edge_x6 writes them — and ef' itself NOT written: each slice's epilogue adds it to the FeedForward block of
    //      the same outputs, in the two-launch form's order ((W2^T H + b2) + ef') + x — every output bit-identical to k_edge_x6 + k_ffn_x6. ----
    stage_e(0, s_w2a);                         // (free since the barrier that ended the last step)
    // (the row fragments of the prologue are still in their registers: xhat feeds the edge update as it fed the FeedForward)
    const int trows = (int)(rend - wg_row0);
    const bool wave_full = trows >= (wv + 1) * XR;
    const int tile_id = blockIdx.x;
    if (a.e.agg_out && wv < 2) {  // destination runs of the two 64-row passes (rows are dst-sorted), by wave 0 and wave 1
      const int pass = wv;
      const int nvalid = min(max(trows - 64 * pass, 0), 64);
      const int d = lane < nvalid ? s_dst[64 * pass + lane] : -1;
      const int dprev = lane > 0 && lane < nvalid ? s_dst[64 * pass + lane - 1] : -2;
      const bool head = lane < nvalid && d != dprev;
      const unsigned long long mask = __ballot(head);
      const int rank = __popcll(mask & ((1ull << lane) - 1ull));
      if (head) s_seg[pass][rank] = lane;
      if (lane == 0) { const int ns = __popcll(mask); s_seg[pass][ns] = nvalid; s_seg[pass][65] = ns; s_seg[pass][66] = a.e.chunk_row0[2 * tile_id + pass]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // slice 0's pieces of this wave have landed
    __syncthreads();  // ... everybody's; every wave is done with the FeedForward's buffers: W1 = staging area, W2b = the next slice's fragments
    GNX_XSTAMP(3);
    // this thread's first run of the per-destination sums (group grp = tid / 8 takes run grp % 16 of pass grp / 16, then every 16th) and its row of the
    // partial-sum table: the same for all four slices — read from LDS here, once, not in front of every slice's sums (three dependent LDS round trips
    // per slice, under the other workgroup's fragment reads: stamps, ~4 k clocks per slice in the sums)
    int seg_n = 0, seg_r0 = 0, seg_r1 = 0, seg_row0 = 0;
    if (a.e.agg_out) {
      const int pass = tid >> 7, g16 = (tid >> 3) & 15;
      seg_n = s_seg[pass][65];
      seg_row0 = s_seg[pass][66];
      if (g16 < seg_n) { seg_r0 = s_seg[pass][g16]; seg_r1 = s_seg[pass][g16 + 1]; }
    }
    constexpr int ELDE = 36;
    float* s_e = reinterpret_cast<float*>(s_w1);   // [128][36]
    float* s_cs = s_e + XBM * ELDE;                // [32][32]
    float* s_zero = s_cs + 32 * 32;                // [32]: a row of zeros — what the sums read for the rows beyond a run's end (no predicate on the data)
    static_assert((XBM * ELDE + 32 * 32 + 32) * 4 <= SLB / 2, "staging area + column-sum partials + the zero row fit the W1 buffer");
    if (tid < 32) s_zero[tid] = 0.f;               // (visible to every wave behind slice 0's first barrier)
    float* sE = s_e + wv * (XR * ELDE);
    const int er0 = lane >> 3, eq0 = lane & 7;  // (row er + 8 i of the wave's 32, 16-byte quad eq of the 32-column block)
    const float* __restrict__ ps = a.e.psrc + r * a.e.N * D;
    const float* __restrict__ pd = a.e.pdst + r * a.e.N * D;
    // residual rows and output rows of the TILE: a wave-uniform base and 32-bit offsets inside the tile (four 64-bit row addresses per array were what
    // the register allocator spilled across the slices — 15 registers reloaded from scratch in front of every slice's stores)
    const float* __restrict__ xres = a.add1 + (r * rows + wg_row0) * D;
    float* __restrict__ outp = a.out + (r * rows + wg_row0) * D;
    const f32x4x zero4 = {0.f, 0.f, 0.f, 0.f};
    const int edge_floor = a.e.act == 1 ? 0 : (int)0x80000000;  // relu as an integer maximum of the float's bits with 0 (identity: with INT_MIN)
    // the gathered addends of a slice: 8 rows x 128 contiguous bytes per instruction
    auto gather = [&](int ob, f32x4x (&us)[4], f32x4x (&ud)[4]) {
      int er = er0, eq = eq0;
      asm volatile("" : "+v"(er), "+v"(eq));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gs = s_src[wv * XR + er + 8 * i], gd = s_dst[wv * XR + er + 8 * i];
        us[i] = *reinterpret_cast<const f32x4x*>(ps + (size_t)gs * D + 32 * ob + 4 * eq);
        ud[i] = *reinterpret_cast<const f32x4x*>(pd + (size_t)gd * D + 32 * ob + 4 * eq);
      }
    };
    auto slice = [&](int ob, const unsigned char* cur, unsigned char* nxt, const f32x16x& accF, f32x4x (&us)[4], f32x4x (&ud)[4], f32x4x (&usn)[4], f32x4x (&udn)[4]) {
      // (the lane's coordinates made opaque per slice: everything derived from them — a dozen offsets and addresses — is then recomputed here, a few
      //  vector instructions, instead of living across all four slices: the register allocator spilled them, and reloaded them in the middle of the
      //  matrix instructions behind an s_waitcnt vmcnt(0) each)
      int er = er0, eq = eq0;
      asm volatile("" : "+v"(er), "+v"(eq));
      if (ob + 1 < NOB) stage_e(ob + 1, nxt);
      // S16: k_edge_x6_prep's planes keep their 32 x 32 x 16 fragment order (fragment 3 s + p, lane (m, h): W[16 s + 8 h + j][32 ob + m]); lane (c16, q16)
      // of group (s32, mb) needs W[32 s32 + 8 q16 + j][32 ob + 16 mb + c16] = fragment 3 (2 s32 + (q16 >> 1)) + p, lane 16 mb + c16 + 32 (q16 & 1)
      const unsigned char* wb = S16 ? cur + (3 * (q16 >> 1)) * 1024 + (c16 + 32 * (q16 & 1)) * 16 : cur + lane * 16;
      auto eoff = [](int g, int p3) -> int { return S16 ? (6 * (g >> 1) + p3) * 1024 + (g & 1) * 256 : (3 * g + p3) * 1024; };
      f32x16x acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      bf16x8x A[2][3];
#pragma unroll
      for (int p3 = 0; p3 < 3; ++p3) A[0][p3] = *reinterpret_cast<const bf16x8x*>(wb + eoff(0, p3));
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int c = s & 1;
        if (s + 1 < KS) {
#pragma unroll
          for (int p3 = 0; p3 < 3; ++p3) A[c ^ 1][p3] = *reinterpret_cast<const bf16x8x*>(wb + eoff(s + 1, p3));
        }
        if constexpr (S16) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const int b = 2 * (s & 1) + nb, zi = nb * KS2 + (s >> 1);
            x6_set4(acc, b, x6_mma16(A[c], zh[zi], zm[zi], zl[zi], x6_sub4(acc, b)));
          }
        } else {
        acc = GNX_X6_MFMA(A[c][1], zm[s], acc, 0, 0, 0);  // small terms first
        acc = GNX_X6_MFMA(A[c][2], zh[s], acc, 0, 0, 0);
        acc = GNX_X6_MFMA(A[c][0], zl[s], acc, 0, 0, 0);
        acc = GNX_X6_MFMA(A[c][1], zh[s], acc, 0, 0, 0);
        acc = GNX_X6_MFMA(A[c][0], zm[s], acc, 0, 0, 0);
        acc = GNX_X6_MFMA(A[c][0], zh[s], acc, 0, 0, 0);
        }
      }
      // the gathered addends of the slice (8 rows x 128 contiguous bytes per instruction) and the residual quads (x: the cache has the rows) are
      // requested here — across the matrix instructions they would not fit the register file beside out^T —, and the FeedForward block of these
      // outputs goes through the wave's slice of the staging area into (row, quad) form while they travel (the other workgroup of the CU has the
      // matrix pipe meanwhile)
      GNX_XSTAMP(4 + 6 * ob);  // matrix instructions issued
      // slice 0 requests its gathered addends here (beside out^T's 64 registers they do not fit across the matrix instructions); the later slices'
      // were requested a slice AHEAD — in front of the previous slice's sums, when its accumulator and weight fragments are dead: they have arrived
      if (ob == 0) gather(0, us, ud);
      f32x4x u1[4], vf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned lrow = (unsigned)min(wv * XR + er + 8 * i, trows - 1);
        u1[i] = *reinterpret_cast<const f32x4x*>(xres + (lrow * D + 32 * ob + 4 * eq));
      }
      // (S16: block b = 2 mb + nb of the container holds outputs 16 mb + 4 q16 + (0..3) of row 16 nb + c16)
      auto stage_block = [&](const f32x16x& blk) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float* d = S16 ? sE + (16 * (g & 1) + c16) * ELDE + 16 * (g >> 1) + 4 * q16 : sE + n * ELDE + 8 * g + 4 * hi;
          *reinterpret_cast<f32x4x*>(d) = f32x4x{blk[4 * g], blk[4 * g + 1], blk[4 * g + 2], blk[4 * g + 3]};
        }
      };
      stage_block(accF);
#pragma unroll
      for (int i = 0; i < 4; ++i) vf[i] = *reinterpret_cast<const f32x4x*>(sE + (er + 8 * i) * ELDE + 4 * eq);
      // (LDS operations of one wave execute in order: the edge block may follow into the same slice)
      stage_block(acc);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the addends — and with them (in-order counter) the next slice's fragments
      asm volatile("" : "+v"(us[0]), "+v"(us[1]), "+v"(us[2]), "+v"(us[3]), "+v"(ud[0]), "+v"(ud[1]), "+v"(ud[2]), "+v"(ud[3]));
      asm volatile("" : "+v"(u1[0]), "+v"(u1[1]), "+v"(u1[2]), "+v"(u1[3]));
      GNX_XSTAMP(5 + 6 * ob);  // addends (and the next slice's fragments) have arrived
      const f32x4x bq = *reinterpret_cast<const f32x4x*>(s_bc + 32 * ob + 4 * eq);
      const f32x4x c1q = *reinterpret_cast<const f32x4x*>(s_bc + D + 32 * ob + 4 * eq);  // We^T beta1 (gn1's shift, folded)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int lr = er + 8 * i;
        f32x4x v = *reinterpret_cast<const f32x4x*>(sE + lr * ELDE + 4 * eq);
        v += c1q;
        v += us[i];
        v += ud[i];
        float vv[4] = {v.x, v.y, v.z, v.w};
        // (TRANS covers the edge function's activation too: the run-time switch over the transcendental forms, inlined at the 16 places of the four
        //  slices, was 3/4 of this kernel's 100 KB of code — more than the instruction cache holds — for an activation the default layers do not have)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (TRANS) vv[e] = act_apply(vv[e], a.e.act);
          else vv[e] = __int_as_float(max(__float_as_int(vv[e]), edge_floor));
        }
        v = f32x4x{vv[0], vv[1], vv[2], vv[3]};
        const bool ok = wave_full || wv * XR + lr < trows;
        if (!ok) v = zero4;  // (rows beyond the tile: zero for the sums below)
        *reinterpret_cast<f32x4x*>(sE + lr * ELDE + 4 * eq) = v;
        f32x4x o = vf[i];  // the two-launch form's order (gnx_core_forward: add1 = the block's output, add2 = x): ((W2^T H + b2) + ef') + x
        o += bq;
        o += v;
        o += u1[i];
        // (wave_full is wave-uniform: the common case stores without a per-lane predicate — no exec-mask branch around the instruction)
        const unsigned ooff = (unsigned)(wv * XR + lr) * D + 32 * ob + 4 * eq;
        if (wave_full) *reinterpret_cast<f32x4x*>(outp + ooff) = o;
        else if (ok) *reinterpret_cast<f32x4x*>(outp + ooff) = o;
      }
      if (ob + 1 < NOB) { gather(ob + 1, usn, udn); __builtin_amdgcn_sched_barrier(0); }
      GNX_XSTAMP(6 + 6 * ob);  // epilogue done
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (LDS traffic only) the finished block of all four waves is staged
      GNX_XSTAMP(7 + 6 * ob);  // barrier 1 passed
      const int q4 = tid & 7, grp = tid >> 3;  // 8 quads x 32 row groups
      f32x4x c4 = {0.f, 0.f, 0.f, 0.f};        // this thread's share of the tile's column sums
      if (a.e.agg_out) {
        // per-destination sums (fixed order; k_edge_x6): groups 0-15 take the runs of pass 0, groups 16-31 those of pass 1
        const int pass = grp >> 4, g16 = grp & 15;
        const int n_seg = seg_n;
        float* agg = a.e.agg_out + (r * a.e.n_agg_rows + (size_t)seg_row0) * D + 32 * ob + 4 * q4;
        const float* base = s_e + 64 * pass * ELDE + 4 * q4;
        for (int sgm = g16; sgm < n_seg; sgm += 16) {
          int r0 = seg_r0, r1 = seg_r1;
          if (sgm != g16) { r0 = s_seg[pass][sgm]; r1 = s_seg[pass][sgm + 1]; }  // (more than 16 runs in a 64-row pass: the later rounds from LDS)
          f32x4x t4 = {0.f, 0.f, 0.f, 0.f};
          for (int rr = r0; rr < r1; rr += 4) {
            f32x4x u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] = *reinterpret_cast<const f32x4x*>(rr + j < r1 ? base + (rr + j) * ELDE : s_zero + 4 * q4);  // (x + 0 is x)
#pragma unroll
            for (int j = 0; j < 4; ++j) t4 += u[j];
          }
          *reinterpret_cast<f32x4x*>(agg + (size_t)sgm * D) = t4;
          c4 += t4;
        }
      } else if (a.e.colsum) {  // (no fused aggregation: the rows themselves, grp, grp + 32, .. ascending)
#pragma unroll
        for (int i = 0; i < 4; ++i) c4 += *reinterpret_cast<const f32x4x*>(s_e + (grp + 32 * i) * ELDE + 4 * q4);
      }
      if (a.e.colsum) *reinterpret_cast<f32x4x*>(s_cs + grp * 32 + 4 * q4) = c4;
      GNX_XSTAMP(8 + 6 * ob);  // per-destination sums done
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the staging area may be overwritten; the column-sum partials are complete
      GNX_XSTAMP(9 + 6 * ob);  // barrier 2 passed
      if (a.e.colsum && tid < 32) {  // fixed order: the 32 groups ascending
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 32; ++w) sum += s_cs[w * 32 + tid];
        a.e.colsum[(r * a.e.n_tiles + (size_t)tile_id) * D + 32 * ob + tid] = sum;
      }
    };
    static_assert(NOB == 4, "four slices, placed by hand");
    f32x4x usA[4], udA[4], usB[4], udB[4];
    slice(0, s_w2a, s_w2b, accO[0], usA, udA, usB, udB);
    slice(1, s_w2b, s_w2a, accO[1], usB, udB, usA, udA);
    slice(2, s_w2a, s_w2b, accO[2], usA, udA, usB, udB);
    slice(3, s_w2b, s_w2a, accO[3], usB, udB, usA, udA);
#ifdef GNX_X6_STAMPS_BUILD
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GNX_XSTAMP(28);
#endif
    return;
  }

  // ---- epilogue.  In the C/D layout lane (n, hi) holds outputs 32 ob + 8 g + 4 hi + (0..3) of ITS row in registers 4 g .. 4 g + 3: accessed from
  //      there, every instruction touches 32 rows with 32 bytes each.  Instead each 32-output block takes a round trip through a wave-private
  //      4.5-KB slice of the (now idle) W1 buffer — LDS operations of one wave execute in order, no barrier — and comes back as (row, 16-byte
  //      quad) = (lane / 8 + 8 i, lane % 8): an instruction then covers 8 rows x 128 contiguous bytes, whole cache lines, for the two residual
  //      loads and the store alike (same instruction count). ----
  constexpr int ELD = 36;  // floats per staged row: 32 + 4 (conflict-free 16-byte writes, one 2-way conflict per read phase)
  constexpr bool E_IN_W1 = XW * XR * ELD * 4 <= SLB / 2;  // (D = 64: the W1 buffer is 12 KB — the staging gets its own 18 KB)
  __shared__ __attribute__((aligned(16))) float s_e[E_IN_W1 ? 4 : XW * XR * ELD];
  float* sE = (E_IN_W1 ? reinterpret_cast<float*>(s_w1) : s_e) + wv * (XR * ELD);
  const int er = lane >> 3, eq = lane & 7;
  const bool wave_full_p = row0 + XR <= rend;  // (wave-uniform: the common case stores without a per-lane predicate)
  const float* __restrict__ r1 = a.add1 ? a.add1 + r * rows * D : nullptr;
  const float* __restrict__ r2 = a.add2 ? a.add2 + r * rows * D : nullptr;
  float* __restrict__ ob_out = a.out + r * rows * D;
  // every residual quad of the tile is requested HERE, in one go (the z fragments' 96 registers are free): one memory round trip for the
  // epilogue instead of one per 32-output block
  f32x4x u1[NOB][4], u2[NOB][4];
  const f32x4x zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t grow = row0 + er + 8 * i < rend ? row0 + er + 8 * i : rend - 1;
      const size_t off = grow * D + 32 * ob + 4 * eq;
      u1[ob][i] = r1 ? *reinterpret_cast<const f32x4x*>(r1 + off) : zero;
      u2[ob][i] = r2 ? *reinterpret_cast<const f32x4x*>(r2 + off) : zero;
    }
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
    const f32x4x bq = a.b2 ? *reinterpret_cast<const f32x4x*>(a.b2 + 32 * ob + 4 * eq) : zero;
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // (S16: block g = 2 mb + nb: outputs 16 mb + 4 q16 + (0..3) of row 16 nb + c16)
      float* d = S16 ? sE + (16 * (g & 1) + c16) * ELD + 16 * (g >> 1) + 4 * q16 : sE + n * ELD + 8 * g + 4 * hi;
      *reinterpret_cast<f32x4x*>(d) = f32x4x{accO[ob][4 * g], accO[ob][4 * g + 1], accO[ob][4 * g + 2], accO[ob][4 * g + 3]};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4x v = *reinterpret_cast<const f32x4x*>(sE + (er + 8 * i) * ELD + 4 * eq);
      v += bq;
      v += u1[ob][i];
      v += u2[ob][i];
      if (wave_full_p) *reinterpret_cast<f32x4x*>(ob_out + (row0 + er + 8 * i) * D + 32 * ob + 4 * eq) = v;
      else if (row0 + er + 8 * i < rend) *reinterpret_cast<f32x4x*>(ob_out + (row0 + er + 8 * i) * D + 32 * ob + 4 * eq) = v;
    }
  }
#ifdef GNX_X6_STAMPS_BUILD
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GNX_XSTAMP(3);
#endif
}

size_t ffn_x6_scratch_bytes(int d) { return (size_t)3 * d * 4 * d * sizeof(__bf16) * 2; }

int32_t launch_fold_beta(const float* W, int ldw, int K, int n_out, const float* beta, const float* bias, float* out, hipStream_t s);  // gnx_edge_x6.hip

// the folded form's scratch: the planes with (gamma . W1), then the 4 d floats of b1 + W1^T beta
size_t ffn_x6_fold_scratch_bytes(int d) { return ffn_x6_scratch_bytes(d) + (size_t)4 * d * sizeof(float); }

// fc1 / fc2 weights of a FeedForward at width d (64 or 128) -> the fragments k_ffn_x6 stages (scratch: ffn_x6_scratch_bytes(d), 16-byte aligned).
// ln_gamma / ln_beta != nullptr: the LayerNorm in front of it folded in — planes of (gamma . W1) and, behind them, b1 + W1^T beta (ffn_x6_fold_scratch_bytes)
int32_t launch_ffn_x6_prep(const float* W1, const float* W2, int d, void* scratch, hipStream_t s, const float* ln_gamma, const float* ln_beta, const float* b1) {
  ProfScope ps("k_ffn_x6_prep", s);
  GNX_LAUNCH(k_ffn_x6_prep, dim3((unsigned)((d * 4 * d + 255) / 256)), dim3(256), 0, s, W1, W2, d, static_cast<__bf16*>(scratch), ln_gamma);
  GNX_HIP(hipGetLastError());
  if (ln_gamma) {
    if (!ln_beta) return fail(GNX_ERR_INVALID_ARG, "k_ffn_x6_prep: the folded form needs gamma and beta");
    return launch_fold_beta(W1, 4 * d, d, 4 * d, ln_beta, b1, reinterpret_cast<float*>(static_cast<char*>(scratch) + ffn_x6_scratch_bytes(d)), s);
  }
  return GNX_OK;
}

bool ffn_x6_applies(const float* z, int d, const gnx_ffn& ff, const float* add1, const float* add2, const float* out, size_t scratch_bytes) {
  if (form(GNX_FLAG_FFN_FP32)) return false;  // (the call asked for the fp32 matrix instruction)
  if ((d != 128 && d != 64) || ff.fc2.act != GNX_ACT_IDENTITY || scratch_bytes < ffn_x6_scratch_bytes(d)) return false;
  const uintptr_t al = (uintptr_t)z | (uintptr_t)ff.fc2.bias | (uintptr_t)add1 | (uintptr_t)add2 | (uintptr_t)out;
  return (al & 15) == 0;
}

// out = add1 + add2 + fc2(act1(fc1(z))) over `nrows` rows per replica; `scratch`: ffn_x6_scratch_bytes(d), 16-byte aligned, free until the launch has run
int32_t launch_ffn_x6(const float* z, size_t nrows, int d, const gnx_ffn& ff, const float* add1, const float* add2, float* out, int64_t R, hipStream_t s,
                      const float* ln_stats, const gnx_layernorm* ln, void* scratch, bool ln_inline, float ln_eps, int ln_mode) {
  if (nrows == 0) return GNX_OK;
  if (!scratch || ((uintptr_t)scratch & 15)) return fail(GNX_ERR_INVALID_ARG, "k_ffn_x6: scratch missing or misaligned");
  if (!z || !ff.fc1.weight || !ff.fc2.weight || !out) return fail(GNX_ERR_INVALID_ARG, "k_ffn_x6: NULL operand");
  if ((ln_stats || ln_inline) && (!ln || !ln->gamma || !ln->beta || ((uintptr_t)ln->gamma & 15) || ((uintptr_t)ln->beta & 15) || ((uintptr_t)ln_stats & 7)))
    return fail(GNX_ERR_INVALID_ARG, "k_ffn_x6: LayerNorm parameters missing or misaligned");
  if (ln_inline && (ln_stats || d != 128)) return fail(GNX_ERR_INVALID_ARG, "k_ffn_x6: statistics in the kernel are for width 128 and exclude a statistics table");
  const __bf16* Wp = static_cast<const __bf16*>(prepared_planes(PREP_FFN, ff.fc1.weight, ff.fc2.weight, d));  // made once with the layer (gnx_core_prepare) ...
  if (!Wp) {                                                                                                      // ... or by a launch in front of this forward
    if (const int32_t rc = launch_ffn_x6_prep(ff.fc1.weight, ff.fc2.weight, d, scratch, s, nullptr, nullptr, nullptr)) return rc;
    Wp = static_cast<const __bf16*>(scratch);
  }
  FfnX6Args a{};
  a.z = z; a.Wp = Wp; a.b1 = ff.fc1.bias; a.b2 = ff.fc2.bias; a.add1 = add1; a.add2 = add2; a.out = out; a.rows = nrows; a.act1 = ff.fc1.act;
  if (ln_stats) { a.ln_stats = ln_stats; a.ln_g = ln->gamma; a.ln_b = ln->beta; }
  if (ln_inline) { a.ln_inline = 1; a.ln_eps = ln_eps; a.ln_mode = ln_mode; a.ln_g = ln->gamma; a.ln_b = ln->beta; }
#ifdef GNX_X6_STAMPS_BUILD
  const size_t n_wg = (nrows + XBM - 1) / XBM;
  const unsigned long long* d_dbg = x6_stamps_begin(n_wg, s);
#endif
  ProfScope ps("k_ffn_x6", s);
  const dim3 grid((unsigned)((nrows + XBM - 1) / XBM), (unsigned)R);
  const bool trans = ff.fc1.act > GNX_ACT_RELU;
  if (d == 128) { if (trans) GNX_LAUNCH((k_ffn_x6<128, true, false>), grid, dim3(64 * XW), 0, s, a); else GNX_LAUNCH((k_ffn_x6<128, false, false>), grid, dim3(64 * XW), 0, s, a); }
  else { if (trans) GNX_LAUNCH((k_ffn_x6<64, true, false>), grid, dim3(64 * XW), 0, s, a); else GNX_LAUNCH((k_ffn_x6<64, false, false>), grid, dim3(64 * XW), 0, s, a); }
  GNX_HIP(hipGetLastError());
#ifdef GNX_X6_STAMPS_BUILD
  static const char* const names[XNST] = {nullptr, "prologue", "slices", "epilogue"};
  x6_stamps_end(d_dbg, n_wg, d == 128 ? "k_ffn_x6<128>" : "k_ffn_x6<64>", names, s);
#endif
  return GNX_OK;
}

int32_t launch_edge_x6_prep(const float* We, int ldw, void* scratch, hipStream_t s, int n_out, const float* ln_gamma, const float* ln_beta);  // gnx_edge_x6.hip
size_t edge_x6_scratch_bytes();

// A GNCore's edge rows in ONE launch (EDGE form of k_ffn_x6): out = x + ef' + FF(gn2(x)), ef' = act(We^T gn1(x) + Ps[src] + Pd[dst]) with its per-destination
// sums (agg_out) and column sums (colsum) as k_edge_x6 writes them; ef' itself is never written.  Row statistics of x in the kernel.
// Both LayerNorms are folded into the weight planes (see FfnX6Edge): prepared with the layer (gnx_core_prepare), or here, per call.
// scratch_e: edge_x6_fold_scratch_bytes(), scratch_f: ffn_x6_fold_scratch_bytes(128); both 16-byte aligned and free until the launch has run.
int32_t launch_core_edge_x6(const Tile* tiles, size_t n_tiles, const float* x, size_t E, const gnx_layernorm* ln1, float ln_eps, int ln_mode, const float* We, int ldw,
                            const float* psrc, const float* pdst, size_t N, const int* src, const int* dst, int act, float* colsum, float* agg_out, size_t n_agg_rows,
                            const int* chunk_row0, const gnx_ffn& ff, const gnx_layernorm* ln2, float* out, int64_t R, void* scratch_e, void* scratch_f, hipStream_t s) {
  if (n_tiles == 0) return GNX_OK;
  if (!scratch_e || !scratch_f || (((uintptr_t)scratch_e | (uintptr_t)scratch_f) & 15)) return fail(GNX_ERR_INVALID_ARG, "k_ffn_x6 (edge form): scratch missing or misaligned");
  if (!tiles || !x || !We || !psrc || !pdst || !src || !dst || !ff.fc1.weight || !ff.fc2.weight || !out) return fail(GNX_ERR_INVALID_ARG, "k_ffn_x6 (edge form): NULL operand");
  if (!ln1 || !ln2 || !ln1->gamma || !ln1->beta || !ln2->gamma || !ln2->beta ||
      (((uintptr_t)ln1->gamma | (uintptr_t)ln1->beta | (uintptr_t)ln2->gamma | (uintptr_t)ln2->beta | (uintptr_t)x | (uintptr_t)out | (uintptr_t)psrc | (uintptr_t)pdst | (uintptr_t)ff.fc2.bias |
        (uintptr_t)agg_out) & 15))
    return fail(GNX_ERR_INVALID_ARG, "k_ffn_x6 (edge form): LayerNorm parameters missing, or an operand not 16-byte aligned");
  if (ff.fc2.act != GNX_ACT_IDENTITY) return fail(GNX_ERR_INVALID_ARG, "k_ffn_x6 (edge form): fc2 with an activation");
  // both weight blocks with their LayerNorm folded in (planes, then the constant vector): made once with the layer (gnx_core_prepare: looked up
  // by the weight and the gamma pointer), or by launches in front of this forward
  int32_t rc = GNX_OK;
  const char* Wpe = static_cast<const char*>(prepared_planes(PREP_EDGE, We, ln1->gamma, 128));
  if (!Wpe) {
    if ((rc = launch_edge_x6_prep(We, ldw, scratch_e, s, 128, ln1->gamma, ln1->beta))) return rc;
    Wpe = static_cast<const char*>(scratch_e);
  }
  const char* Wp = static_cast<const char*>(prepared_planes(PREP_FFN, ff.fc1.weight, ln2->gamma, 128));
  if (!Wp) {
    if ((rc = launch_ffn_x6_prep(ff.fc1.weight, ff.fc2.weight, 128, scratch_f, s, ln2->gamma, ln2->beta, ff.fc1.bias))) return rc;
    Wp = static_cast<const char*>(scratch_f);
  }
  FfnX6Args a{};
  a.z = x; a.Wp = reinterpret_cast<const __bf16*>(Wp); a.b1 = reinterpret_cast<const float*>(Wp + ffn_x6_scratch_bytes(128));  // b1 + W1^T beta2
  a.b2 = ff.fc2.bias; a.add1 = x; a.add2 = nullptr; a.out = out; a.rows = E; a.act1 = ff.fc1.act;
  a.ln_g = ln2->gamma; a.ln_b = ln2->beta; a.ln_eps = ln_eps; a.ln_mode = ln_mode;
  a.e.tiles = tiles; a.e.c1 = reinterpret_cast<const float*>(Wpe + edge_x6_scratch_bytes()); a.e.Wpe = reinterpret_cast<const __bf16*>(Wpe); a.e.psrc = psrc; a.e.pdst = pdst; a.e.N = N; a.e.src = src; a.e.dst = dst;
  a.e.act = act; a.e.colsum = colsum; a.e.n_tiles = n_tiles; a.e.agg_out = agg_out; a.e.n_agg_rows = n_agg_rows; a.e.chunk_row0 = chunk_row0;
  ProfScope ps("k_core_edge_x6", s);
  const dim3 grid((unsigned)n_tiles, (unsigned)R);
#ifdef GNX_X6_STAMPS_BUILD
  const unsigned long long* d_dbg = x6_stamps_begin(n_tiles, s);
  const unsigned dyn_lds = getenv("GNX_X6_ONE_WG") ? 8192u : 0u;  // (diagnostic: 8 KB of unused dynamic LDS leave room for ONE workgroup per CU — a wave's phases without a partner)
#else
  constexpr unsigned dyn_lds = 0;
#endif
  if (ff.fc1.act > GNX_ACT_RELU || act > GNX_ACT_RELU) GNX_LAUNCH((k_ffn_x6<128, true, true>), grid, dim3(64 * XW), dyn_lds, s, a);
  else GNX_LAUNCH((k_ffn_x6<128, false, true>), grid, dim3(64 * XW), dyn_lds, s, a);
  GNX_HIP(hipGetLastError());
#ifdef GNX_X6_STAMPS_BUILD
  static const char* const names[XNST] = {nullptr, "prologue", "FeedForward", "transition",
    "s0:mfma", "s0:addends", "s0:epilogue", "s0:barrier1", "s0:sums", "s0:barrier2", "s1:mfma", "s1:addends", "s1:epilogue", "s1:barrier1", "s1:sums", "s1:barrier2",
    "s2:mfma", "s2:addends", "s2:epilogue", "s2:barrier1", "s2:sums", "s2:barrier2", "s3:mfma", "s3:addends", "s3:epilogue", "s3:barrier1", "s3:sums", "s3:barrier2", "drain"};
  x6_stamps_end(d_dbg, n_tiles, "k_core_edge_x6", names, s);
#endif
  return GNX_OK;
}

}  // namespace gnx

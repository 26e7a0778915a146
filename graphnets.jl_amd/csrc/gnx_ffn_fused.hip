// Fused position-wise FeedForward of a GNCore on the fp32 matrix cores (gnfeedforward.jl:27-40, gncore.jl:56-68):
//
//     out[m, :] = add1[m, :] + add2[m, :] + b2 + W2^T act1(W1^T z[m, :] + b1)          z = gn2(x), add1 = block output, add2 = x
//
// The two-GEMM form writes the (rows x 4D) hidden activations to HBM and reads them back (2 x 2 GB per core at C4's edge
// width) and its epilogues — 35-45 % of a tile's lifetime by the shader-clock stamps — are bursts of exactly that traffic.
// Here a workgroup keeps a 128-row tile for the whole FeedForward: the hidden layer is produced 64 units at a time on the
// matrix cores, activated, parked in LDS in A-operand layout and immediately consumed by the second GEMM, whose 128 x D
// accumulator lives in registers across all 4D/64 slices.  HBM sees z once (re-streamed per slice from L2), the residuals
// and the output once, and nothing of the hidden layer.
//   per slice: GEMM1 (K = D, in 32-wide chunks: z chunk + W1 chunk through LDS) -> bias, activation -> sH[128][65]
//              GEMM2 (K = 64, two chunks: A = sH, W2 chunk through LDS) -> accO
// 512 threads = 8 waves as 4 (row blocks) x 2 (column halves); v_mfma_f32_32x32x2_f32 (exact fp32); 58 KB of LDS -> 2 workgroups
// (16 waves) per CU.
#include <algorithm>
#include <cstdio>
#include <type_traits>
#include <vector>

#include "gnx_device.h"
#include "gnx_x6_mma.h"

#ifndef GNX_LN_GUARD  // see the LayerNorm branch of store_step (and csrc/gnx_wide.hip)
#define GNX_LN_GUARD 1
#endif

namespace gnx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // staging registers: a first-class vector (HIP's float4 struct copies as memcpy
                                                            // and the arrays of it ended up in LDS / scratch here)

namespace {
constexpr int FBM = 128;   // rows per workgroup
constexpr int FHS = 64;    // hidden units per slice
constexpr int FKC = 32;    // K chunk
// compile-time loop: the step index must be a constant inside the body (register arrays stay registers, branches fold)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
__device__ __forceinline__ void lds_barrier_f() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
}  // namespace

struct FfnArgs {
  const Tile* tiles;
  int row_kind;            // 0: rows = edges of the tile, 1: rows = nodes / graphs (n0, n1)
  const float* z;          // [R][rows][D]  gn2(x)
  const float* W1;         // (4D x D) column-major == [D][4D] row-major
  const float* b1;         // [4D] or nullptr
  const float* W2;         // (D x 4D) column-major == [4D][D] row-major
  const float* b2;         // [D] or nullptr
  const float* add1;       // [R][rows][D] or nullptr
  const float* add2;
  float* out;              // [R][rows][D]
  size_t rep_stride;       // rows_total * D
  int act1;
  // z = gn2(x) computed on load: z points at x, ln_stats at [R][rows_total][2] (mean, inv) from k_ln_stats_v4, ln_g / ln_b at gamma /
  // beta of gn2 — (x - mean) * inv, then fma(gamma, ., beta), the arithmetic of k_layernorm2_v4 (nullptr: z is used as it is)
  const float* ln_stats;
  const float* ln_g;
  const float* ln_b;
};

// (4 waves per SIMD = two workgroups per CU: without the attribute the compiler takes 130+ registers and the second workgroup is gone)
#ifdef GNX_FFN_STAMPS_BUILD  // diagnostic build only (tools/build_variant.sh ffnst gnx_ffn_fused.hip -DGNX_FFN_STAMPS_BUILD): shader-clock stamps of wave 0
static __device__ unsigned long long* g_ffn_dbg = nullptr;  // [tile][8]
#define GNX_FSTAMP(acc, t0) do { const unsigned long long t1_ = clock64(); acc += t1_ - t0; t0 = t1_; } while (0)
#else
#define GNX_FSTAMP(acc, t0) do { } while (0)
#endif

template <int D, bool X6>  // X6 (the default form): every product as six bf16 matrix-core terms, fragments split on the fly (gnx_x6_mma.h); else v_mfma_f32_32x32x2f32
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void k_ffn_fused(FfnArgs a) {
  constexpr int H = 4 * D;
  constexpr int LDA = FKC + 1;                 // z chunk row stride (odd: conflict-free A fragments)
  constexpr int LDH = FHS + 1;                 // hidden slice row stride
  constexpr int TNO = D / 64;                  // 32-column output blocks per wave (wave = 64 rows x D/2 columns)
  constexpr int NC1 = D / FKC;                 // chunks of GEMM1
  constexpr int NT = 512;
  constexpr int NB1 = (FKC * FHS / 4) / NT;    // float4 of a W1 chunk [32][64] per thread (= 1)
  constexpr int NB2 = (FKC * D / 4) / NT;      // float4 of a W2 chunk [32][D] per thread (= D/64)
  constexpr int POOLF = FBM * LDA + FKC * FHS; // sA [128][33] + W1 chunk [32][64]; the W2 chunk [32][D] of GEMM2 reuses the sA region
  static_assert(FKC * D <= FBM * LDA, "the W2 chunk must fit in the z-chunk region");
  __shared__ __attribute__((aligned(16))) float s_pool[POOLF + FBM * LDH];  // 58 KB -> two workgroups per CU
  float* sA = s_pool;                 // [128][33]   z chunk (GEMM1)
  float* sB1 = s_pool + FBM * LDA;    // [32][64]    W1 chunk (GEMM1)
  float* sB2 = s_pool;                // [32][D]     W2 chunk (GEMM2; sA is idle then)
  float* sH = s_pool + POOLF;         // [128][65]   activated hidden slice
  float* sC = s_pool;                 // epilogue staging [64][D + 4]
  static_assert(64 * (D + 4) <= POOLF + FBM * LDH, "epilogue staging must fit");
  __shared__ float s_b1[H];                                           // fc1 bias: read once per hidden slice — as a global load the
                                                                      // read sits in front of its use and its wait (vmcnt is in order)
                                                                      // also waits for the next step's prefetch
  __shared__ float2 s_ln[FBM];                                        // (mean, inv) of the tile's rows
  __shared__ __attribute__((aligned(16))) f32x4 s_lng[D / 4], s_lnb[D / 4];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;  // wm 0..3: rows 32*wm.., wn: column half
  const int hi = lane >> 5, l31 = lane & 31;
  const Tile t = a.tiles[blockIdx.x];
  const size_t r = blockIdx.y;
  const int row0 = a.row_kind == 0 ? t.e0 : t.n0;
  const int rows = (a.row_kind == 0 ? t.e1 : t.n1) - row0;
  const float* __restrict__ zb = a.z + r * a.rep_stride + (size_t)row0 * D;

  for (int i = tid; i < H; i += NT) s_b1[i] = a.b1 ? a.b1[i] : 0.f;  // (visible after the first barrier of the step loop)
  const bool ln = a.ln_stats != nullptr;
  if (ln) {
    if (tid < FBM) s_ln[tid] = ld_stats(reinterpret_cast<const float2*>(a.ln_stats + r * (a.rep_stride / D) * 2) + row0 + (tid < rows ? tid : rows - 1));
    if (tid >= FBM && tid < FBM + D / 4) { s_lng[tid - FBM] = reinterpret_cast<const f32x4*>(a.ln_g)[tid - FBM]; s_lnb[tid - FBM] = reinterpret_cast<const f32x4*>(a.ln_b)[tid - FBM]; }
  }
  // (s_ln / s_lng / s_lnb / s_b1 are visible after the first barrier of the step loop, which precedes the first store_step)

  f32x16 accO[TNO];
#pragma unroll
  for (int j = 0; j < TNO; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) accO[j][q] = 0.f;

  // register staging of the next chunk (global -> registers during the MFMAs of the current one -> LDS)
  f32x4 ra[2];        // z chunk: 128 rows x 8 float4 = 1024 float4 -> 2 per thread
  f32x4 rb1[NB1], rb2[NB2];  // W1 chunk / W2 chunk (separate arrays: a shared one is demoted to LDS by the compiler)
  const int a_c4 = tid & 7, a_r = tid >> 3;  // z chunk: thread -> (row a_r + 64 i, float4 a_c4)

  // chunk schedule of one slice: 0 .. NC1-1 = GEMM1 chunks, NC1, NC1+1 = GEMM2 chunks
  constexpr int NSTEP = NC1 + 2;
  auto load_step = [&](int hs, int st) {
    if (st < NC1) {
      const int kc = st * FKC;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = a_r + 64 * i;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        ra[i] = row < rows ? *reinterpret_cast<const f32x4*>(zb + (size_t)row * D + kc + 4 * a_c4) : zero;
      }
#pragma unroll
      for (int i = 0; i < NB1; ++i) {  // W1[(kc + kk) * H + hs*64 + 4*c4], chunk [32][64]: 16 float4 per row
        const int q = tid + NT * i, kk = q >> 4, c4 = q & 15;
        rb1[i] = *reinterpret_cast<const f32x4*>(a.W1 + (size_t)(kc + kk) * H + hs * FHS + 4 * c4);
      }
    } else {
      const int k0 = hs * FHS + (st - NC1) * FKC;  // rows of W2
#pragma unroll
      for (int i = 0; i < NB2; ++i) {  // chunk [32][D]: D/4 float4 per row
        const int q = tid + NT * i, kk = q / (D / 4), c4 = q % (D / 4);
        rb2[i] = *reinterpret_cast<const f32x4*>(a.W2 + (size_t)(k0 + kk) * D + 4 * c4);
      }
    }
  };
  auto store_step = [&](int st) {
    if (st < NC1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float* d = sA + (a_r + 64 * i) * LDA + 4 * a_c4;
        f32x4 v = ra[i];
        if (ln && a_r + 64 * i < rows) {  // (rows beyond the tile stay zero)
          const float2 st2 = s_ln[a_r + 64 * i];
          const f32x4 g = s_lng[st * (FKC / 4) + a_c4], b = s_lnb[st * (FKC / 4) + a_c4];
          // (the guard of k_rows_gemm's LayerNorm branch, csrc/gnx_wide.hip: the LDS reads retired and sixteen idle issue slots in front of the first use)
          float sx_ = st2.x, sy_ = st2.y;
#if GNX_LN_GUARD
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "+v"(sx_), "+v"(sy_)::"memory");
#endif
          v.x = fmaf(g.x, (v.x - sx_) * sy_, b.x); v.y = fmaf(g.y, (v.y - sx_) * sy_, b.y);
          v.z = fmaf(g.z, (v.z - sx_) * sy_, b.z); v.w = fmaf(g.w, (v.w - sx_) * sy_, b.w);
        }
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
#pragma unroll
      for (int i = 0; i < NB1; ++i) *reinterpret_cast<f32x4*>(sB1 + 4 * (tid + NT * i)) = rb1[i];  // [kk][64]
    } else {
#pragma unroll
      for (int i = 0; i < NB2; ++i) *reinterpret_cast<f32x4*>(sB2 + 4 * (tid + NT * i)) = rb2[i];  // [kk][D]
    }
  };

#ifdef GNX_FFN_STAMPS_BUILD
  unsigned long long fs_sync = 0, fs_issue = 0, fs_mma1 = 0, fs_hid = 0, fs_mma2 = 0, fs_t = clock64();
  const unsigned long long fs_start = fs_t;
#endif
  load_step(0, 0);
  for (int hs = 0; hs < H / FHS; ++hs) {
    f32x16 accH;
#pragma unroll
    for (int q = 0; q < 16; ++q) accH[q] = 0.f;
    for (int st = 0; st < NSTEP; ++st) {
      __syncthreads();  // readers of the previous chunk are done (and, at st == NC1, sH has been written by everyone)
      store_step(st);
      // prefetch the next step (possibly the first chunk of the next slice) as soon as the staging registers are free — in front of the
      // barrier, whose wait is then part of the loads' cover (the barrier orders LDS traffic only; the loads are waited for at the next store_step)
      if (st + 1 < NSTEP) load_step(hs, st + 1);
      else if (hs + 1 < H / FHS) load_step(hs + 1, 0);
      GNX_FSTAMP(fs_issue, fs_t);
      __syncthreads();
      GNX_FSTAMP(fs_sync, fs_t);
      if (st < NC1) {
        // GEMM1: accH[32 x 32 per wave] += z chunk * W1 chunk     (wave rows 32*wm.., hidden columns 32*wn..)
        // fragments of k-step kk + 1 requested from LDS before the MFMA of step kk (pinned: left alone the compiler reads, waits, multiplies —
        // C4 6.38 -> 6.27 ms; s_setprio(1) around the matrix-core sections on top of it: 6.28 vs 6.25 ms, not kept)
        if constexpr (X6) {
#pragma unroll
          for (int s16 = 0; s16 < FKC / 16; ++s16)
            accH = x6_mma(x6_frag(sA + (wm * 32 + l31) * LDA + 16 * s16 + 8 * hi, 1), x6_frag(sB1 + (16 * s16 + 8 * hi) * FHS + wn * 32 + l31, FHS), accH);
        } else {
        float fa1[2], fb1[2];
        fb1[0] = sB1[hi * FHS + wn * 32 + l31];
        fa1[0] = sA[(wm * 32 + l31) * LDA + hi];
#pragma unroll
        for (int kk = 0; kk < FKC / 2; ++kk) {
          const int c = kk & 1, n = c ^ 1;
          if (kk + 1 < FKC / 2) {
            fb1[n] = sB1[(2 * (kk + 1) + hi) * FHS + wn * 32 + l31];
            fa1[n] = sA[(wm * 32 + l31) * LDA + 2 * (kk + 1) + hi];
          }
          __builtin_amdgcn_sched_barrier(0);
          accH = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[c], fb1[c], accH, 0, 0, 0);
        }
        }
        GNX_FSTAMP(fs_mma1, fs_t);
        if (st == NC1 - 1) {
          // hidden slice: bias + activation, parked in LDS as the A operand of GEMM2 (C/D layout: row = (q&3)+8(q>>2)+4hi)
          const int hcol = wn * 32 + l31;
          const float b = s_b1[hs * FHS + hcol];
          float hv[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) hv[q] = accH[q] + b;
          switch (a.act1) {  // ONE wave-uniform switch, the loop inside each case (a switch per element is 16 branch chains)
            case 0: break;
            case 1:
#pragma unroll
              for (int q = 0; q < 16; ++q) hv[q] = fmaxf(hv[q], 0.f);
              break;
            default:
#pragma unroll
              for (int q = 0; q < 16; ++q) hv[q] = act_apply(hv[q], a.act1);
              break;
          }
#pragma unroll
          for (int q = 0; q < 16; ++q) sH[(wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * hi) * LDH + hcol] = hv[q];
          GNX_FSTAMP(fs_hid, fs_t);
        }
      } else {
        // GEMM2: accO[64 x D/2 per wave] += sH[:, 32-wide k range] * W2 chunk
        const int kb = (st - NC1) * FKC;
        if constexpr (X6) {
#pragma unroll
          for (int s16 = 0; s16 < FKC / 16; ++s16) {
            const X6Frag fa6 = x6_frag(sH + (wm * 32 + l31) * LDH + kb + 16 * s16 + 8 * hi, 1);
#pragma unroll
            for (int j = 0; j < TNO; ++j) accO[j] = x6_mma(fa6, x6_frag(sB2 + (16 * s16 + 8 * hi) * D + (wn * TNO + j) * 32 + l31, D), accO[j]);
          }
        } else {
        float fa2[2], fb2[2][TNO];
        fa2[0] = sH[(wm * 32 + l31) * LDH + kb + hi];
#pragma unroll
        for (int j = 0; j < TNO; ++j) fb2[0][j] = sB2[hi * D + (wn * TNO + j) * 32 + l31];
#pragma unroll
        for (int kk = 0; kk < FKC / 2; ++kk) {
          const int c = kk & 1, n = c ^ 1;
          if (kk + 1 < FKC / 2) {
            fa2[n] = sH[(wm * 32 + l31) * LDH + kb + 2 * (kk + 1) + hi];
#pragma unroll
            for (int j = 0; j < TNO; ++j) fb2[n][j] = sB2[(2 * (kk + 1) + hi) * D + (wn * TNO + j) * 32 + l31];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < TNO; ++j) accO[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa2[c], fb2[c][j], accO[j], 0, 0, 0);
        }
        }
        GNX_FSTAMP(fs_mma2, fs_t);
      }
    }
  }
#ifdef GNX_FFN_STAMPS_BUILD
  const unsigned long long fs_loop_end = clock64();
#endif

  // ---- epilogue: two 64-row passes through LDS -> full-row 16-B stores with bias and the two residuals.  The residual rows of a
  //      pass are requested in one go (unconditional loads of clamped rows) before the pass's LDS traffic, pass 1's before pass 0's
  //      stores: one load -> add -> store per quad is a memory round trip per quad (vmcnt counts the stores too) ----
  constexpr int LDC = D + 4;
  constexpr int NC4 = (64 * D / 4) / NT;
  float* out_tile = a.out + r * a.rep_stride + (size_t)row0 * D;
  const float* add1_tile = a.add1 ? a.add1 + r * a.rep_stride + (size_t)row0 * D : nullptr;
  const float* add2_tile = a.add2 ? a.add2 + r * a.rep_stride + (size_t)row0 * D : nullptr;
  const int q4 = tid % (D / 4), lr0 = tid / (D / 4);
  constexpr int NGR = NT / (D / 4);  // rows between the quads of a thread
  f32x4 u1[2][NC4], u2[2][NC4];
  auto issue_residuals = [&](int pass) {
#pragma unroll
    for (int i = 0; i < NC4; ++i) {
      const unsigned off = (unsigned)min(64 * pass + lr0 + NGR * i, rows - 1) * D + 4u * q4;
      if (add1_tile) u1[pass][i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(add1_tile) + (off << 2));
      if (add2_tile) u2[pass][i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(add2_tile) + (off << 2));
    }
  };
  issue_residuals(0);
  f32x4 b2v = {0.f, 0.f, 0.f, 0.f};
  if (a.b2) b2v = *reinterpret_cast<const f32x4*>(a.b2 + 4 * q4);
  static_for<0, 2>([&](auto pass_c) {
    constexpr int pass = decltype(pass_c)::value;
    lds_barrier_f();
    if ((wm >> 1) == pass) {
#pragma unroll
      for (int j = 0; j < TNO; ++j) {
        const int col = (wn * TNO + j) * 32 + l31;
#pragma unroll
        for (int q = 0; q < 16; ++q) sC[((wm & 1) * 32 + (q & 3) + 8 * (q >> 2) + 4 * hi) * LDC + col] = accO[j][q];
      }
    }
    lds_barrier_f();
    f32x4 v[NC4];
#pragma unroll
    for (int i = 0; i < NC4; ++i) v[i] = *reinterpret_cast<const f32x4*>(sC + (lr0 + NGR * i) * LDC + 4 * q4) + b2v;
    if (add1_tile) {
#pragma unroll
      for (int i = 0; i < NC4; ++i) v[i] += u1[pass][i];
    }
    if (add2_tile) {
#pragma unroll
      for (int i = 0; i < NC4; ++i) v[i] += u2[pass][i];
    }
    if (pass == 0) issue_residuals(1);  // BEFORE pass 0's stores
#pragma unroll
    for (int i = 0; i < NC4; ++i) {
      const int row = 64 * pass + lr0 + NGR * i;
      if (row < rows) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(out_tile) + (((unsigned)row * D + 4u * q4) << 2)) = v[i];
    }
  });
#ifdef GNX_FFN_STAMPS_BUILD
  if (g_ffn_dbg && tid == 0 && blockIdx.y == 0) {
    unsigned long long* o = g_ffn_dbg + (size_t)blockIdx.x * 8;
    const unsigned long long te = clock64();
    o[0] = fs_sync; o[1] = fs_issue; o[2] = fs_mma1; o[3] = fs_hid; o[4] = fs_mma2; o[5] = te - fs_loop_end; o[6] = te - fs_start;
  }
#endif
}

// (Round 2 built the variant this structure suggests — z tile RESIDENT in LDS for the whole tile, weights through a two-deep LDS
// ring, ONE barrier per 32-MFMA step instead of two per 16-MFMA step, fragment reads batched 8 k-steps ahead; 131 KB of LDS, so
// one 8-wave workgroup per CU — and measured it on C4: 3.15 vs 2.42 ms per edge FeedForward, matrix-core busy 55 % vs 72 %
// (profiles/r02_ab_ffn_zres.log, profiles/r02_c4_pmc_ffn_fused_vs_zres.json).  At two waves per SIMD nothing covers a wave that
// waits at the barrier; the two independent 8-wave workgroups of this kernel fill each other's gaps.  Not kept.)

// out = add1 + add2 + fc2(act1(fc1(z))) over all rows of one entity type.  1 = not applicable (the caller runs the two GEMMs).
// does launch_ffn_fused take this FeedForward (else 1 = "not applicable")
bool ffn_fused_applies(const float* z, int d, const gnx_ffn& ff, const float* add1, const float* add2, const float* out) {
  static const bool off = getenv("GNX_NO_FFN_FUSED") != nullptr;
  if (off || (d != 64 && d != 128) || ff.fc2.act != GNX_ACT_IDENTITY) return false;
  const uintptr_t al = (uintptr_t)z | (uintptr_t)ff.fc1.weight | (uintptr_t)ff.fc2.weight | (uintptr_t)ff.fc2.bias | (uintptr_t)add1 | (uintptr_t)add2 | (uintptr_t)out;
  return (al & 15) == 0;
}

bool ffn_x6_applies(const float* z, int d, const gnx_ffn& ff, const float* add1, const float* add2, const float* out, size_t scratch_bytes);  // gnx_ffn_x6.hip
int32_t launch_ffn_x6(const float* z, size_t nrows, int d, const gnx_ffn& ff, const float* add1, const float* add2, float* out, int64_t R, hipStream_t s,
                      const float* ln_stats, const gnx_layernorm* ln, void* scratch, bool ln_inline, float ln_eps, int ln_mode);

// scratch (optional; scratch_bytes of device memory the caller does not need until this launch has run): lets the edge FeedForward at d = 128 run
// as k_ffn_x6 — the fp32 products carried by six bf16 matrix-core terms (gnx_ffn_x6.hip) — instead of the fp32-MFMA kernel below
int32_t launch_ffn_fused(const gnx_graphs* h, int entity, const float* z, int d, const gnx_ffn& ff, const float* add1, const float* add2, float* out,
                         int64_t R, hipStream_t s, const float* ln_stats, const gnx_layernorm* ln, void* scratch, size_t scratch_bytes, bool ln_inline, float ln_eps,
                         int ln_mode) {
  if (!ffn_fused_applies(z, d, ff, add1, add2, out)) return (ln_stats || ln_inline) ? fail(GNX_ERR_INVALID_ARG, "k_ffn_fused: LayerNorm on load asked for a FeedForward it does not take") : 1;
  const size_t nrows = entity == 0 ? (size_t)h->E : (entity == 1 ? (size_t)h->N : (size_t)h->G);
  if (nrows == 0) return GNX_OK;
  if (scratch && nrows >= 4096 && ffn_x6_applies(z, d, ff, add1, add2, out, scratch_bytes))
    return launch_ffn_x6(z, nrows, d, ff, add1, add2, out, R, s, ln_stats, ln, scratch, ln_inline, ln_eps, ln_mode);
  if (ln_inline) return fail(GNX_ERR_INVALID_ARG, "internal: statistics in the kernel asked of the fp32 FeedForward");
  if (ln_stats && (!ln || !ln->gamma || !ln->beta || ((uintptr_t)ln->gamma & 15) || ((uintptr_t)ln->beta & 15) || ((uintptr_t)ln_stats & 7)))
    return fail(GNX_ERR_INVALID_ARG, "k_ffn_fused: LayerNorm parameters missing or misaligned");
  if (int32_t rcw = gnx_ensure_wide_tables(h, s)) return rcw;
  FfnArgs a{};
  a.tiles = entity == 0 ? h->d_etiles : (entity == 1 ? h->d_ntiles : h->d_gtiles);
  a.row_kind = entity == 0 ? 0 : 1;
  a.z = z; a.W1 = ff.fc1.weight; a.b1 = ff.fc1.bias; a.W2 = ff.fc2.weight; a.b2 = ff.fc2.bias;
  a.add1 = add1; a.add2 = add2; a.out = out; a.rep_stride = nrows * (size_t)d; a.act1 = ff.fc1.act;
  if (ln_stats) { a.ln_stats = ln_stats; a.ln_g = ln->gamma; a.ln_b = ln->beta; }
  const unsigned n_tiles = (unsigned)(entity == 0 ? h->n_etiles : (entity == 1 ? h->n_ntiles : h->n_gtiles));
  if (!a.tiles || !z || !ff.fc1.weight || !ff.fc2.weight || !out) return fail(GNX_ERR_INVALID_ARG, "k_ffn_fused: NULL operand");
  ProfScope ps("k_ffn_fused", s);
#ifdef GNX_FFN_STAMPS_BUILD
  static unsigned long long* d_dbg = nullptr;
  static size_t dbg_cap = 0;
  const bool stamps = getenv("GNX_FFN_STAMPS") != nullptr;
  if (stamps) {
    if (dbg_cap < n_tiles) { if (d_dbg) (void)hipFree(d_dbg); dbg_cap = n_tiles; (void)hipMalloc((void**)&d_dbg, dbg_cap * 64); }
    (void)hipMemsetAsync(d_dbg, 0, (size_t)n_tiles * 64, s);
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_ffn_dbg), &d_dbg, sizeof(d_dbg), 0, hipMemcpyHostToDevice, s);
  }
#endif
  // (the fp32 matrix instruction only where the call asks for it: profiles/r05_mfma_mix_hazard.log)
  if (form(GNX_FLAG_FFN_FP32)) {
    if (d == 128) GNX_LAUNCH((k_ffn_fused<128, false>), dim3(n_tiles, (unsigned)R), dim3(512), 0, s, a);
    else GNX_LAUNCH((k_ffn_fused<64, false>), dim3(n_tiles, (unsigned)R), dim3(512), 0, s, a);
  } else {
    if (d == 128) GNX_LAUNCH((k_ffn_fused<128, true>), dim3(n_tiles, (unsigned)R), dim3(512), 0, s, a);
    else GNX_LAUNCH((k_ffn_fused<64, true>), dim3(n_tiles, (unsigned)R), dim3(512), 0, s, a);
  }
  GNX_HIP(hipGetLastError());
#ifdef GNX_FFN_STAMPS_BUILD
  if (stamps) {
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> hs((size_t)n_tiles * 8);
    (void)hipMemcpy(hs.data(), d_dbg, hs.size() * 8, hipMemcpyDeviceToHost);
    double m[7] = {0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < n_tiles; ++i) for (int j = 0; j < 7; ++j) m[j] += (double)hs[i * 8 + j];
    fprintf(stderr, "[gnx ffn stamps] D=%d tiles=%u: per tile (shader clocks, wave 0): barriers+LDS stores %.0f  load issue %.0f  GEMM1 MFMA sections %.0f  hidden slice -> LDS %.0f  GEMM2 MFMA sections %.0f  epilogue %.0f  total %.0f\n",
            d, n_tiles, m[0] / n_tiles, m[1] / n_tiles, m[2] / n_tiles, m[3] / n_tiles, m[4] / n_tiles, m[5] / n_tiles, m[6] / n_tiles);
  }
#endif
  return GNX_OK;
}

}  // namespace gnx


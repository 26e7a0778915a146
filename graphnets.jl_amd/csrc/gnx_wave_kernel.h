// The fused narrow-width kernels (see gnx_narrow.hip for the design notes).  Self-contained device code: included by
// gnx_narrow.hip for the ahead-of-time width sets AND compiled at run time by gnx_jit.cpp (hiprtc) for any other
// narrow width set, which is why it only uses gnx_device.h and compiler builtins.
#pragma once
#include "gnx_device.h"

namespace gnx {

namespace {

constexpr int kThreads = 256;

// dword-aligned multi-dword accesses: gfx950 global loads/stores of 8/12/16 B only need 4-B alignment, so a feature
// row of D floats moves in ceil(D/4) instructions whatever D is.
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) F3u { float x, y, z; };
struct __attribute__((packed, aligned(4))) F2u { float x, y; };

// Weights / biases are read-only for the whole launch and their indices are wave-uniform.  Reading them through the
// constant address space makes hipcc emit scalar loads (s_load_*, SGPR operands of v_fma) instead of per-lane vector
// loads — with plain global pointers the stores of this kernel make them "clobberable" and they land in VGPRs.
typedef const float __attribute__((address_space(4))) * cfloatp;
__device__ __forceinline__ cfloatp as_const(const float* p) { return reinterpret_cast<cfloatp>(reinterpret_cast<size_t>(p)); }

template <int D>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&x)[D > 0 ? D : 1]) {
  constexpr int Q = D / 4, R = D % 4;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const F4u v = *reinterpret_cast<const F4u*>(p + 4 * q);
    x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
  }
  if constexpr (R == 3) {
    const F3u v = *reinterpret_cast<const F3u*>(p + 4 * Q);
    x[4 * Q] = v.x; x[4 * Q + 1] = v.y; x[4 * Q + 2] = v.z;
  } else if constexpr (R == 2) {
    const F2u v = *reinterpret_cast<const F2u*>(p + 4 * Q);
    x[4 * Q] = v.x; x[4 * Q + 1] = v.y;
  } else if constexpr (R == 1) {
    x[4 * Q] = p[4 * Q];
  }
}

template <int D>
__device__ __forceinline__ void store_row(float* __restrict__ p, const float (&x)[D > 0 ? D : 1]) {
  constexpr int Q = D / 4, R = D % 4;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    F4u v; v.x = x[4 * q]; v.y = x[4 * q + 1]; v.z = x[4 * q + 2]; v.w = x[4 * q + 3];
    *reinterpret_cast<F4u*>(p + 4 * q) = v;
  }
  if constexpr (R == 3) {
    F3u v; v.x = x[4 * Q]; v.y = x[4 * Q + 1]; v.z = x[4 * Q + 2];
    *reinterpret_cast<F3u*>(p + 4 * Q) = v;
  } else if constexpr (R == 2) {
    F2u v; v.x = x[4 * Q]; v.y = x[4 * Q + 1];
    *reinterpret_cast<F2u*>(p + 4 * Q) = v;
  } else if constexpr (R == 1) {
    p[4 * Q] = x[4 * Q];
  }
}

// activation of a whole register row: ONE (wave-uniform) switch, the loop inside each case
template <int N>
__device__ __forceinline__ void act_row(float (&v)[N], int act) {
  switch (act) {
    case 0: break;
    case 1:
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = relu_f(v[j]);
      break;
    default:
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = act_apply(v[j], act);
      break;
  }
}

// stands in for a NULL bias (Dense(...; bias=false)) so that the bias read needs no branch
__constant__ float k_zero_bias[64] = {0.f};

// acc[m][j] += sum_{k<K} W[k*LD + j] * x[m][k] for M register rows at once; W = first of K j-contiguous weight rows,
// read through the scalar cache.  With many weights the rows are streamed in GROUPS (<= 32 scalars) separated by
// scheduling fences and an opaque pointer, so that only one group of SGPR weights is live at a time: unfenced, the
// compiler hoists every s_load of the fully unrolled product to the top and spills SGPRs into VGPR lanes (one
// v_readlane per use) once K*OUT exceeds the ~100 available scalars.
#ifndef GNX_FENCE_T
#define GNX_FENCE_T 48  // products with at most this many weights are left to the compiler
#endif
#ifndef GNX_FENCE_G
#define GNX_FENCE_G 32  // scalars per fenced group
#endif
template <int K, int OUT, int M, int KX, int LD = OUT>  // LD: distance between weight rows (>= OUT)
__device__ __forceinline__ void fma_rows(cfloatp W, const float (&x)[M][KX], float (&acc)[M][OUT > 0 ? OUT : 1]) {
  if constexpr (K > 0 && OUT > 0) {
    if constexpr (K * OUT <= GNX_FENCE_T) {
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
          for (int j = 0; j < OUT; ++j) acc[m][j] = fmaf(W[k * LD + j], x[m][k], acc[m][j]);
    } else {
      constexpr int KG = GNX_FENCE_G / OUT > 0 ? GNX_FENCE_G / OUT : 1;
#pragma unroll
      for (int k0 = 0; k0 < K; k0 += KG) {
        cfloatp Wk = W + k0 * LD;
        // the group's weight pointer becomes available only once the previous group's last FMA has been issued: the
        // loads are invariant (no memory chain), a data dependency is the one fence that instruction selection honours
        asm volatile("" : "+s"(Wk) : "v"(acc[M - 1][OUT - 1]));
#pragma unroll
        for (int kk = 0; kk < KG; ++kk) {
          if (k0 + kk < K) {
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
              for (int j = 0; j < OUT; ++j) acc[m][j] = fmaf(Wk[kk * LD + j], x[m][k0 + kk], acc[m][j]);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Hand-streamed scalar weights + packed FMAs (used by k_core_post_s and by k_block_wave's large edge products).
// A GROUP of N consecutive weights is fetched by s_load_dwordx{16,8,4,2} issued from inline asm — volatile asm statements keep their
// order, so the compiler can neither hoist nor merge them — into one of two register sets: the next group is in flight while the FMAs
// of the current one run (scalar loads return out of order, so the only wait is lgkmcnt(0), placed BEFORE the next issue).  Two rows
// (edges) of a lane are held as a register PAIR and every FMA is one v_pk_fma_f32 whose src0 is the SGPR pair holding the weight:
// op_sel picks its low or high half for both rows (tools/experiments/pk_fma_sgpr_probe.hip) — half the VALU issue of scalar FMAs, no LDS.
// ---------------------------------------------------------------------------------------------------------------------------------
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v8f_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));

template <int N>
struct SGroup {  // N (even, <= 32) consecutive floats held in SGPRs
  static_assert(N % 2 == 0 && N >= 2 && N <= 32, "group size");
  static constexpr int N16 = N / 16, N8 = (N % 16) / 8, N4 = (N % 8) / 4, N2 = (N % 4) / 2;
  v16f_t a, b; v8f_t c; v4f_t d; v2f_t e;
  __device__ __forceinline__ void issue(cfloatp p) {
    if constexpr (N16 >= 1) asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(a) : "s"(p));
    if constexpr (N16 >= 2) asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=s"(b) : "s"(p));
    if constexpr (N8 == 1) asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(c) : "s"(p), "n"(64 * N16));
    if constexpr (N4 == 1) asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(d) : "s"(p), "n"(64 * N16 + 32 * N8));
    if constexpr (N2 == 1) asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(e) : "s"(p), "n"(64 * N16 + 32 * N8 + 16 * N4));
  }
  // every use of the group is ordered behind this: ONE s_waitcnt, the other pieces are tied to the volatile order by empty statements
  __device__ __forceinline__ void wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (N16 >= 1) asm volatile("" : "+s"(a));
    if constexpr (N16 >= 2) asm volatile("" : "+s"(b));
    if constexpr (N8 == 1) asm volatile("" : "+s"(c));
    if constexpr (N4 == 1) asm volatile("" : "+s"(d));
    if constexpr (N2 == 1) asm volatile("" : "+s"(e));
  }
  __device__ __forceinline__ v2f_t pair(int q) const {  // floats 2q, 2q+1 = an aligned SGPR pair of the group's registers; q constant
    v2f_t r;
    r.x = get(2 * q); r.y = get(2 * q + 1);
    return r;
  }
  __device__ __forceinline__ float get(int i) const {  // i is a constant once the loops are unrolled
    if (i < 16 * N16) return i < 16 ? a[i & 15] : b[i & 15];
    i -= 16 * N16;
    if (i < 8 * N8) return c[i & 7];
    i -= 8 * N8;
    if (i < 4 * N4) return d[i & 3];
    i -= 4 * N4;
    return e[i & 1];
  }
};

typedef v2f_t P2;  // (row 0, row 1) of a lane
template <bool HI>
__device__ __forceinline__ void pk_fma_sw(P2& acc, v2f_t wpair, P2 x) {  // acc += w * x, w = low / high half of the SGPR pair
  if constexpr (!HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(wpair), "v"(x));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(wpair), "v"(x));
}
// ties N register pairs into the order of the volatile statements (no instruction)
template <int N, int O = 0>
__device__ __forceinline__ void pin_pairs(P2 (&v)[N]) {
  if constexpr (O + 8 <= N) {
    asm volatile("" : "+v"(v[O]), "+v"(v[O + 1]), "+v"(v[O + 2]), "+v"(v[O + 3]), "+v"(v[O + 4]), "+v"(v[O + 5]), "+v"(v[O + 6]), "+v"(v[O + 7]));
    pin_pairs<N, O + 8>(v);
  } else if constexpr (O + 4 <= N) {
    asm volatile("" : "+v"(v[O]), "+v"(v[O + 1]), "+v"(v[O + 2]), "+v"(v[O + 3]));
    pin_pairs<N, O + 4>(v);
  } else if constexpr (O + 2 <= N) {
    asm volatile("" : "+v"(v[O]), "+v"(v[O + 1]));
    pin_pairs<N, O + 2>(v);
  } else if constexpr (O + 1 <= N) {
    asm volatile("" : "+v"(v[O]));
  }
}


// acc[e % OUT] += W[e] * x[e / OUT] for e in [0, NEL): W = NEL consecutive weights (input k's OUT weights, then input k+1's, ...), x and
// acc hold BOTH rows of the lane as pairs.  Groups of GS weights, the last one of TAIL; walked by a compile-time recursion.
#ifndef GNX_PK_GS
#define GNX_PK_GS 24
#endif
template <int NEL, int OUT, int KX, int GS>
struct PkStream {
  static_assert(NEL % 2 == 0 && GS % 2 == 0 && NEL >= 2, "even element counts");
  static constexpr int NGRP = (NEL + GS - 1) / GS, TAIL = NEL - GS * (NGRP - 1);
  cfloatp W;
  P2 (&x)[KX];
  P2 (&acc)[OUT];
  SGroup<GS> G0, G1;
  SGroup<TAIL> GT;  // the last group (its own register set: the full sets are dead by the time it is consumed)

  template <int GI>
  __device__ __forceinline__ void issue() {
    if constexpr (GI == NGRP - 1) GT.issue(W + GI * GS);
    else if constexpr (GI % 2 == 0) G0.issue(W + GI * GS);
    else G1.issue(W + GI * GS);
  }
  template <int GI, class GRP>
  __device__ __forceinline__ void consume(const GRP& cur) {
    constexpr int n = GI == NGRP - 1 ? TAIL : GS;
#pragma unroll
    for (int q = 0; q < n / 2; ++q) {
      const int e0 = GI * GS + 2 * q, e1 = e0 + 1;
      const v2f_t w = cur.pair(q);
      pk_fma_sw<false>(acc[e0 % OUT], w, x[e0 / OUT]);
      pk_fma_sw<true>(acc[e1 % OUT], w, x[e1 / OUT]);
    }
    pin_pairs<OUT>(acc);
  }
  template <int GI>
  __device__ __forceinline__ void run() {
    if constexpr (GI < NGRP) {
      if constexpr (GI == NGRP - 1) GT.wait();
      else if constexpr (GI % 2 == 0) G0.wait();
      else G1.wait();
      if constexpr (GI + 1 < NGRP) issue<GI + 1>();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (GI == NGRP - 1) consume<GI>(GT);
      else if constexpr (GI % 2 == 0) consume<GI>(G0);
      else consume<GI>(G1);
      __builtin_amdgcn_sched_barrier(0);
      run<GI + 1>();
    }
  }
};

// out = x + block_out + W2 act(W1 gn2(x) + b1) + b2, TWO rows per thread held as register pairs: every FMA is one v_pk_fma_f32
// whose src0 is the SGPR pair holding the weight (op_sel picks its low or high half for both rows) — half the VALU issue of scalar FMAs.
// The hidden layer is produced and consumed in two halves of HB = 2D units; the weight stream of a half is [b1 half | D rows of W1
// (2D consecutive hidden units of input k) | D row PAIRS of W2 (2D consecutive floats: hidden units 2g, 2g+1)], every group 2D floats
// = 2D packed FMAs.  The 2(2D+1) groups are walked by a compile-time recursion (GI = global group index: half, position and
// register set are constants of each step).  The FMAs are (non-volatile) asm statements too: left as C, the vectoriser packs them
// itself — with the weight copied into a VGPR pair first — and collects them behind the loads of ALL groups (800 spilled SGPRs).
#ifndef GNX_CORE_POST_UNITS
#define GNX_CORE_POST_UNITS 1  // units of two rows per thread of the streamed FeedForward body (2 + the L2 touch loads: 39.6 vs 38.1 us, see below)
#endif
// TRANS = false: both activations are identity / relu (the reference's FeedForward) — the tanh / sigmoid / gelu expansions of a run-time
// activation switch cost ~60 registers on every path, relu's included.
template <int D, bool TRANS>
struct CorePostStream {
  static constexpr int H = 4 * D, HB = 2 * D, NG = 2 * D + 1, TOTAL = 2 * NG;
  cfloatp W1, W2, b1;
  int act1;
  P2 (&z)[D];
  P2 (&acc)[D];
  P2 h[HB];
  SGroup<HB> G0, G1;

  template <int GI>
  __device__ __forceinline__ cfloatp group_ptr() const {
    constexpr int p = GI / NG, i = GI % NG;
    if constexpr (i == 0) return b1 + p * HB;
    else if constexpr (i <= D) return W1 + (i - 1) * H + p * HB;
    else return W2 + (p * HB + 2 * (i - D - 1)) * D;
  }
  template <int GI>
  __device__ __forceinline__ void consume(const SGroup<HB>& cur) {
    constexpr int i = GI % NG;
    if constexpr (i == 0) {
      // h[j] = (b[j], b[j]): ONE v_pk_mov_b32 per hidden unit — both halves of the destination from the low (op_sel 0,0) or the high
      // (op_sel 1,1) half of the SGPR pair that holds b[2q], b[2q+1] (one unique scalar operand: within the constant-bus limit)
#pragma unroll
      for (int q = 0; q < HB / 2; ++q) {
        const v2f_t w = cur.pair(q);
        asm("v_pk_mov_b32 %0, %1, %1" : "=v"(h[2 * q]) : "s"(w));
        asm("v_pk_mov_b32 %0, %1, %1 op_sel:[1,1]" : "=v"(h[2 * q + 1]) : "s"(w));
      }
      pin_pairs<HB>(h);
    } else if constexpr (i <= D) {
      constexpr int k = i - 1;
#pragma unroll
      for (int q = 0; q < HB / 2; ++q) {
        const v2f_t w = cur.pair(q);
        pk_fma_sw<false>(h[2 * q], w, z[k]);
        pk_fma_sw<true>(h[2 * q + 1], w, z[k]);
      }
      if constexpr (i == D) {
        if constexpr (TRANS) {
          float t[HB];
#pragma unroll
          for (int j = 0; j < HB; ++j) t[j] = h[j].x;
          act_row<HB>(t, act1);
#pragma unroll
          for (int j = 0; j < HB; ++j) { h[j].x = t[j]; t[j] = h[j].y; }
          act_row<HB>(t, act1);
#pragma unroll
          for (int j = 0; j < HB; ++j) h[j].y = t[j];
        } else if (act1 == 1) {
#pragma unroll
          for (int j = 0; j < HB; ++j) { h[j].x = relu_f(h[j].x); h[j].y = relu_f(h[j].y); }
        }
      }
      pin_pairs<HB>(h);
    } else {
      constexpr int j = 2 * (i - D - 1);
#pragma unroll
      for (int q = 0; q < D; ++q) {  // pair q = weights 2q, 2q+1 of [W2 row j | W2 row j+1]
        const v2f_t w = cur.pair(q);
        pk_fma_sw<false>(acc[(2 * q) % D], w, h[j + (2 * q) / D]);
        pk_fma_sw<true>(acc[(2 * q + 1) % D], w, h[j + (2 * q + 1) / D]);
      }
      pin_pairs<D>(acc);
    }
  }
  template <int GI>
  __device__ __forceinline__ void run() {
    if constexpr (GI < TOTAL) {
      if constexpr (GI % 2 == 0) {
        G0.wait();
        if constexpr (GI + 1 < TOTAL) G1.issue(group_ptr<GI + 1>());
        __builtin_amdgcn_sched_barrier(0);
        consume<GI>(G0);
      } else {
        G1.wait();
        if constexpr (GI + 1 < TOTAL) G0.issue(group_ptr<GI + 1>());
        __builtin_amdgcn_sched_barrier(0);
        consume<GI>(G1);
      }
      __builtin_amdgcn_sched_barrier(0);
      run<GI + 1>();
    }
  }
};

// xhat = (x - mu) * rstd over the D registers of a row; eps_mode 0: 1/(sigma+eps) (Flux 0.14 normalise), 1: 1/sqrt(var+eps)
// (mu_o, rstd_o: the row's statistics for a caller that needs x-hat of the SAME row again — (x - mu) * rstd repeats the two operations below, hence the bits)
template <int D>
__device__ __forceinline__ void normalise_s(float (&x)[D], float eps, int eps_mode, float& mu_o, float& rstd_o) {
  // (instruction count matters: these kernels are VALU-bound at core widths and an IEEE division is ~10 instructions.  The run-time eps
  // convention selects the ARGUMENT of one square root and one division — written as `mode ? 1/(sqrt(v)+eps) : 1/sqrt(v+eps)` the compiler
  // evaluates both sides and selects: two square roots and two divisions per row.  The two means stay divisions by D: as products with
  // 1/D they save 20 instructions per row but are not the reference's arithmetic — at D = 3 the extra rounding of the mean showed in a
  // row with a small sigma.)
  float mu = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) mu += x[k];
  mu /= (float)D;
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) { x[k] -= mu; var = fmaf(x[k], x[k], var); }
  var /= (float)D;
  float sd = sqrtf(eps_mode == 0 ? var : var + eps);
  sd = eps_mode == 0 ? sd + eps : sd;
  const float rstd = 1.f / sd;
#pragma unroll
  for (int k = 0; k < D; ++k) x[k] *= rstd;
  mu_o = mu; rstd_o = rstd;
}
template <int D>
__device__ __forceinline__ void normalise(float (&x)[D], float eps, int eps_mode) {
  float mu, rstd;
  normalise_s<D>(x, eps, eps_mode, mu, rstd);
}
// LayerNorm of a register row with scalar-operand affine parameters (gngraphnorm.jl:19-26)
template <int D>
__device__ __forceinline__ void ln_row(float (&x)[D > 0 ? D : 1], const float* g, const float* b, float eps, int eps_mode, float* stats = nullptr) {
  if constexpr (D > 0) {
    float mu, rstd;
    normalise_s<D>(x, eps, eps_mode, mu, rstd);
    if (stats) { stats[0] = mu; stats[1] = rstd; }
    const cfloatp gc = as_const(g), bc = as_const(b);
#pragma unroll
    for (int k = 0; k < D; ++k) x[k] = fmaf(gc[k], x[k], bc[k]);
  }
}

// XCD-aware block -> tile map: blocks b and b+8 share an XCD (and its L2), so give every XCD one contiguous
// chunk of tiles; tiles of one graph (which share the graph's node rows) then meet in one L2.  Bijective for any nt.
__device__ __forceinline__ int xcd_tile(int b, int nt) {
  const int per = nt >> 3, rem = nt & 7;
  const int x = b & 7, i = b >> 3;
  return x * per + (x < rem ? x : rem) + i;
}

// sum over the 16 lanes of a DPP row, result in every lane of the row; pure VALU (no LDS), order-symmetric
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror
  return v;
}

}  // namespace

__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// sum over the 64 lanes of the wave, same bits in every lane, fixed association
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}

// =========================================================================================================
// Graph update from the per-tile partial sums, rows of CP = 4*ceil(C/4) floats [sum_e ef' ; sum_n nf' ; pad]:
//   gf'[g] = act(Wg * [sum_e ef' ; sum_n nf' ; gf_g] + bg)                       (graphfninput.jl:1-13, gnblock.jl:67)
// Latency is everything here (a few KB of work): the thread's rows (<= 16 float4 in flight), its slice of Wg / bg / gf
// are all issued before the first wait; the sums are reduced with DPP + one LDS hop in a fixed order which depends only
// on (t0, t1, nthr): bitwise reproducible.
// WAVE: executed by ONE wavefront (nthr = 64) — LDS operations of one wave execute in order, no workgroup barrier.
// s_g: (nthr/16)*C + (C+dg+4) + (C+dg+1)*og floats of LDS.
//
// (Round 2 also built the graph update INTO k_block_wave — write-through rows, drained, sharded line-spaced arrival tickets,
// the last arriver reads the rows with sc1 loads — bitwise equal to this form and slower or equal at every size: C2 27.4 vs
// 26.7 us/step, 4096 graphs 25.5 vs 24.6, 512 graphs 20.6 vs 20.7, 5k edges 8.6 vs 8.3: on MI355X the in-launch hand-off costs
// what a kernel boundary plus this kernel costs.  profiles/r02_ab_single_launch_*.log; the code is in the history.)
// =========================================================================================================
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));

// One partial-sum row: lane c < C holds column c in `mine`; lane q < CP/4 stores columns 4q..4q+3 as ONE 16-byte store.
template <int C>
__device__ __forceinline__ void store_partial_row(float mine, float* row, int lane) {
  constexpr int Q = (C + 3) / 4;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    float4 v;
    v.x = readlane_f(mine, 4 * q);
    v.y = 4 * q + 1 < C ? readlane_f(mine, 4 * q + 1) : 0.f;
    v.z = 4 * q + 2 < C ? readlane_f(mine, 4 * q + 2) : 0.f;
    v.w = 4 * q + 3 < C ? readlane_f(mine, 4 * q + 3) : 0.f;
    if (lane == q) *reinterpret_cast<float4*>(row + 4 * q) = v;
  }
}

template <int C, bool WAVE, int F4_IN_FLIGHT>
__device__ __forceinline__ void graph_update_rows(const BlockArgs& a, const float* __restrict__ base, int g, size_t r, int t0, int t1, int tid, int nthr, float* s_g) {
  constexpr int Q = (C + 3) / 4, CP = 4 * Q;
  constexpr int RIF = Q >= F4_IN_FLIGHT ? 1 : F4_IN_FLIGHT / Q;  // rows in flight per thread; the per-thread accumulation order (rows ascending) does not depend on it
  const int K = C + a.dg, og = a.og;
  const int nrow16 = nthr >> 4;
  float* s_x = s_g + (size_t)nrow16 * C;  // [K]   graph-function input
  float* s_w = s_x + K + 4;               // [K*og] weights, [og] bias

  // prefetch weights / bias / gf (tiny, L2) — independent of the partial sums
  const int nw = K * og;
  float w_reg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + i * nthr;
    w_reg[i] = idx < nw ? a.Wg[idx] : (idx < nw + og ? (a.bg ? a.bg[idx - nw] : 0.f) : 0.f);
  }
  float gf_reg = 0.f;
  if (tid < a.dg) gf_reg = a.gf[(r * (size_t)a.G + g) * a.dg + tid];

  float acc[Q][4];
#pragma unroll
  for (int q = 0; q < Q; ++q) acc[q][0] = acc[q][1] = acc[q][2] = acc[q][3] = 0.f;
  for (int row0 = t0 + tid; row0 < t1; row0 += RIF * nthr) {
    float4 val[RIF][Q];
#pragma unroll
    for (int u = 0; u < RIF; ++u) {
      const int row = row0 + u * nthr;
      if (row < t1) {
#pragma unroll
        for (int q = 0; q < Q; ++q) val[u][q] = *reinterpret_cast<const float4*>(base + (size_t)row * CP + 4 * q);
      }
    }
#pragma unroll
    for (int u = 0; u < RIF; ++u) {
      if (row0 + u * nthr < t1) {
#pragma unroll
        for (int q = 0; q < Q; ++q) { acc[q][0] += val[u][q].x; acc[q][1] += val[u][q].y; acc[q][2] += val[u][q].z; acc[q][3] += val[u][q].w; }
      }
    }
  }
  // weights to LDS (loads have long since landed)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + i * nthr;
    if (idx < nw + og) s_w[idx] = w_reg[i];
  }
  for (int idx = tid + 4 * nthr; idx < nw + og; idx += nthr) s_w[idx] = idx < nw ? a.Wg[idx] : (a.bg ? a.bg[idx - nw] : 0.f);
  if (a.ln_g[2] && a.dg > 0) {  // GNCore: the graph function sees gn1(gf); a few values, every thread computes the statistics
    const float* gp = a.gf + (r * (size_t)a.G + g) * a.dg;
    float mu = 0.f;
    for (int k = 0; k < a.dg; ++k) mu += gp[k];
    mu /= (float)a.dg;
    float var = 0.f;
    for (int k = 0; k < a.dg; ++k) { const float c = gp[k] - mu; var = fmaf(c, c, var); }
    var /= (float)a.dg;
    const float rstd = a.ln_mode == 0 ? 1.f / (sqrtf(var) + a.ln_eps) : 1.f / sqrtf(var + a.ln_eps);
    for (int k = tid; k < a.dg; k += nthr) s_x[C + k] = fmaf(a.ln_g[2][k], (gp[k] - mu) * rstd, a.ln_b[2][k]);
  } else {
    if (tid < a.dg) s_x[C + tid] = gf_reg;
    for (int k = tid + nthr; k < a.dg; k += nthr) s_x[C + k] = a.gf[(r * (size_t)a.G + g) * a.dg + k];
  }
  const int lane = tid & 63, row16 = tid >> 4;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float x = row16_sum(acc[c >> 2][c & 3]);
    if ((lane & 15) == 0) s_g[row16 * C + c] = x;
  }
  if constexpr (WAVE) __builtin_amdgcn_wave_barrier(); else __syncthreads();
  // second stage: wave 0 sums the <= 64 row sums of every column (fixed order: DPP tree + 4 readlanes)
  if (tid < 64) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float x = wave_sum(tid < nrow16 ? s_g[tid * C + c] : 0.f);
      if (tid == 0) s_x[c] = x;
    }
  }
  if constexpr (WAVE) __builtin_amdgcn_wave_barrier(); else __syncthreads();
  float* out = a.gf_out + (r * (size_t)a.G + g) * og;
  for (int j = tid; j < og; j += nthr) {
    float y = s_w[nw + j];
    for (int k = 0; k < K; ++k) y = fmaf(s_w[k * og + j], s_x[k], y);
    out[j] = act_apply(y, a.act_g);
  }
}

// The graph update of ONE graph by ONE wavefront when the column sums are already in a register (lane c < C holds column c): the tail
// of graph_update_rows — same input vector [sums ; gf (normalised if asked)], same weights, same order of the FMAs, hence the same bits.
// Two halves so that the weight loads can be in flight under the workgroup barrier in front of the sums: graph_wave_prefetch (lane
// j < og: bias + its first KW weights, clamped unconditional loads), then graph_update_wave.  s_x: >= C + dg floats of wave-private LDS.
constexpr int kGraphWavePrefetch = 8;  // (24 took the README-dims kernel from 58 to 93 registers: 5 instead of 8 waves per SIMD, 23.1 -> 28.8 us/step)
struct GraphWaveRegs { float bias; float w[kGraphWavePrefetch]; };
template <int C>
__device__ __forceinline__ void graph_wave_prefetch(const BlockArgs& a, int lane, GraphWaveRegs& gr) {
  const int K = C + a.dg, og = a.og;
  const int j = lane < og ? lane : og - 1;
  gr.bias = a.bg ? a.bg[j] : 0.f;
#pragma unroll
  for (int u = 0; u < kGraphWavePrefetch; ++u) gr.w[u] = a.Wg[(size_t)(u < K ? u : K - 1) * og + j];
}
template <int C>
__device__ __forceinline__ void graph_update_wave(const BlockArgs& a, float xsum, int g, size_t r, int lane, float* s_x, const GraphWaveRegs& gr) {
  const int K = C + a.dg, og = a.og;
  if (lane < C) s_x[lane] = xsum;
  const float* gp = a.dg > 0 ? a.gf + (r * (size_t)a.G + g) * a.dg : nullptr;
  if (a.ln_g[2] && a.dg > 0) {
    float mu = 0.f;
    for (int k = 0; k < a.dg; ++k) mu += gp[k];
    mu /= (float)a.dg;
    float var = 0.f;
    for (int k = 0; k < a.dg; ++k) { const float c = gp[k] - mu; var = fmaf(c, c, var); }
    var /= (float)a.dg;
    const float rstd = a.ln_mode == 0 ? 1.f / (sqrtf(var) + a.ln_eps) : 1.f / sqrtf(var + a.ln_eps);
    for (int k = lane; k < a.dg; k += 64) s_x[C + k] = fmaf(a.ln_g[2][k], (gp[k] - mu) * rstd, a.ln_b[2][k]);
  } else {
    for (int k = lane; k < a.dg; k += 64) s_x[C + k] = gp[k];
  }
  __builtin_amdgcn_wave_barrier();
  float* out = a.gf_out + (r * (size_t)a.G + g) * og;
  if (lane < og) {  // FMAs in the order k = 0, 1, 2, ... (the order of graph_update_rows)
    float y = gr.bias;
#pragma unroll
    for (int u = 0; u < kGraphWavePrefetch; ++u)
      if (u < K) y = fmaf(gr.w[u], s_x[u], y);
    for (int k = kGraphWavePrefetch; k < K; ++k) y = fmaf(a.Wg[(size_t)k * og + lane], s_x[k], y);
    out[lane] = act_apply(y, a.act_g);
  }
  for (int j = lane + 64; j < og; j += 64) {  // (graph functions wider than a wavefront: plain loop)
    float y = a.bg ? a.bg[j] : 0.f;
    for (int k = 0; k < K; ++k) y = fmaf(a.Wg[(size_t)k * og + j], s_x[k], y);
    out[j] = act_apply(y, a.act_g);
  }
}

// LDS floats graph_update_rows needs with nthr threads
__host__ __device__ constexpr int graph_update_lds_floats(int C, int dg, int og, int nthr) {
  return (nthr / 16) * C + (C + dg + 4) + (C + dg + 1) * og + 8;
}
// per-wave LDS slice of k_block_wave, in floats: [ef' of the tile | per-node projections | tile-local dst of each edge]
__host__ __device__ constexpr int wave_slice_floats(int OE, int EPT) {
  return (64 * EPT * OE + 4) + (64 * OE + 4) + (64 * EPT + 4 + 15) / 16 * 4;
}

// =========================================================================================================
// k_block_wave — the whole GNBlock edge + node update, ONE WAVEFRONT PER TILE.
//
// A wave tile is a node range of one graph with <= 64 nodes and <= 64*EPT in-edges (contiguous, CSC order).  The
// wave owns every edge that aggregates into its nodes, so the edge->node sum needs no atomics and no workgroup
// barrier: lanes exchange data through a wave-private LDS slice (LDS operations of one wave execute in order), and
// reductions are DPP + readlane.  A workgroup is just four independent waves; 28-32 of them are resident per CU at
// different points of their (load -> gather -> compute -> store) chain, which is what overlaps HBM with compute.
//   lanes as EDGES : EPT edges per lane — ef row (dword-aligned 16-B loads), rowval, nf[src] row gather, W*x, store
//   lanes as NODES : lane n < nn — colptr, own nf row, pd[n] = b + We[:,dst]*nf[n] (+ gf fold), segmented sum of
//                    ef' from LDS, node update, store
// =========================================================================================================
// LN: LayerNorm the inputs on load (BlockArgs::ln_*).  ONEG: the batch is ONE graph (see below).
// SGPR budget: a CU admits floor(800 / (ceil(sgpr/16)*16 + 16)) 256-thread workgroups (MI355X_MICROARCH, residency): 8 up to 80
// SGPRs, 7 up to 96, 6 beyond.  C2 is 2032 workgroups = ONE round at 8 per CU (2048 slots) but 1.13 rounds at 7.
#ifndef GNX_WAVE_PK
#define GNX_WAVE_PK 1  // large edge products of a two-edges-per-lane tile through PkStream (0: fma_rows)
#endif
#ifndef GNX_WAVE_SGPRS
#define GNX_WAVE_SGPRS 80
#endif
#ifdef GNX_WAVE_STAMPS_BUILD  // diagnostic build only (tools/build_variant.sh ... -DGNX_WAVE_STAMPS_BUILD): per-wave shader-clock stamps
static __device__ unsigned long long* g_wave_dbg = nullptr;  // [n_wtiles][8], set by the launcher
#define GNX_WSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); wst_[i] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GNX_WSTAMP(i) do { } while (0)
#endif
// PACK: batches of SMALL graphs (every graph <= 8 wave tiles; BlockArgs::packs).  A 512-thread workgroup = 8 wave tiles that hold WHOLE
// graphs: after the tiles one barrier, then the wave of a graph's first tile adds the graph's partial rows from LDS — in the association
// of graph_update_rows for <= 8 rows, ((a0+a1)+(a2+a3))+((a4+a5)+(a6+a7)) — and runs the graph function: no partial rows in HBM, no
// second launch, bit-identical gf'.
constexpr int kPackThreads = 512;
// FFE (narrow GNCore; needs LN, DE == OE, two edges per lane on the streamed-weights path, identity / relu activations, and a batch without
// a node of more than 64 * EPT in-edges: the launcher checks): the edge lanes also run the core's edge FeedForward and add both residual
// terms — ef_out receives y = (x + ef') + FF(gn2(x)) (gncore.jl:56-59) instead of ef', so block_out and the second read of x never touch HBM
// for edges (the two-kernel form moved 120 MB per core for them on the 1M-edge graph).  Register budget (126, four waves per SIMD as
// before): the FeedForward runs LAST, when nothing of the node phase is live (as a phase of the edge loop: 187 registers); the raw rows
// wait in the lane's own, still unused ef' slots of the wave's LDS slice while the edge product runs (kept in registers: 134) and ef' is
// read back from those slots at the end.  gn2 shares x-hat with gn1 (normalise() of the same row: the same bits); the association is
// k_core_post_s's — (x + ef') + (b2 + W2 h) — so the result is BIT-IDENTICAL to the two-kernel form (tests/test_gpu_core.py).
// CHAIN (gnx_block_forward_chained): the first a.prev_blocks workgroups of the launch — dispatched first, so they run beside the tiles — finish
// the graph update of the PREVIOUS call (graph_update_rows over ITS partial rows: one workgroup for a one-graph batch, one wavefront per
// graph otherwise), the rest are this call's tiles.  In a loop over batches the second launch of every step disappears (an empty launch is
// ~4 us of a ~25-us step).  A template parameter: the plain kernel keeps its registers (58 at README widths) and has no barrier.
// (The body is a device function so that the kernel can exist under two resource limits: k_block_wave with the 80-SGPR cap that buys the eighth
// workgroup per CU, k_block_wave_ffe without it — at its 120+ vector registers a CU holds four workgroups whatever the scalar count, and under
// the cap that form kept ~70 scalars in lanes of a vector register: a v_readlane_b32, often with an s_nop behind it, per use.)
// The kernel's arguments for one PHASE of the body.  FRESH: a pointer into the kernel-argument segment that the compiler cannot connect to the
// argument loads of another phase (kernel arguments are lowered to scalar loads at the top of the kernel and then LIVE to their last use: in the
// FFE form ~60 of them waited in lanes of a vector register, a v_writelane_b32 and one v_readlane_b32 per use each — vector instructions in a
// kernel bound by their issue; a second s_load_dword is free there).  Otherwise the kernel's own copy.
typedef const BlockArgs __attribute__((address_space(4))) * cargsp;
template <bool FRESH>
__device__ __forceinline__ auto phase_args(const BlockArgs& a) {
  if constexpr (FRESH) {
    cargsp p = (cargsp)__builtin_amdgcn_kernarg_segment_ptr();  // (BlockArgs is the kernel's first argument)
    asm volatile("" : "+s"(p));
    return p;
  } else {
    return &a;
  }
}

template <int DE, int DN, int DG, int OE, int ON, int EPT, bool LN, bool ONEG, bool PACK, bool FFE, bool CHAIN>
__device__ __forceinline__ void block_wave_body(BlockArgs& a, int n_rows) {
  static_assert(!(PACK && ONEG), "packs are for batches of several graphs");
  static_assert(!CHAIN || (!PACK && !FFE && !LN && OE + ON > 0), "CHAIN: the two-launch form of a plain block");
  static_assert(!FFE || (LN && DE == OE && DE > 0 && EPT == 2 && (DE + DN) * OE > 96 && ((DE + DN) * OE) % 2 == 0 && GNX_WAVE_PK && !PACK),
                "FFE: a core's block (dims => dims) with LayerNorm on load, two edges per lane, streamed weights");
  constexpr int OE1 = OE > 0 ? OE : 1, ON1 = ON > 0 ? ON : 1, DE1 = DE > 0 ? DE : 1, DN1 = DN > 0 ? DN : 1, DG1 = DG > 0 ? DG : 1;
  constexpr int TEW = 64 * EPT;
  constexpr int C = OE + ON, C1 = C > 0 ? C : 1;
  constexpr int WAVES = (PACK ? kPackThreads : kThreads) / 64;
  constexpr int WSL = wave_slice_floats(OE, EPT);
  __shared__ __attribute__((aligned(16))) float s_mem[WAVES * WSL];  // one slice per wave

  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  typedef const int __attribute__((address_space(4))) * cintp;
  int bx = blockIdx.x, gx = gridDim.x;  // this call's tiles are the blocks behind the chained ones
  if constexpr (CHAIN) {
    if ((int)blockIdx.x < a.prev_blocks) {
      if (a.og > 0) {
        constexpr int CPc = (C + 3) / 4 * 4;
        a.gf = a.prev_gf; a.gf_out = a.prev_gf_out;  // (this workgroup returns below: the kernel's own copy of the arguments is its to change)
        const float* pbase = a.prev_partials + blockIdx.y * (size_t)n_rows * CPc;
        if constexpr (ONEG) {
          graph_update_rows<C, false, 4>(a, pbase, 0, blockIdx.y, 0, (a.n_wtiles + 3) / 4, (int)threadIdx.x, kThreads, s_mem);
        } else {
          const int g = (int)blockIdx.x * WAVES + wv;  // one wavefront per graph (the launcher chains only while a graph has <= 256 rows)
          if (g < a.G) graph_update_rows<C, true, 4>(a, pbase, g, blockIdx.y, a.wtile_off[g], a.wtile_off[g + 1], lane, 64, s_mem + wv * WSL);
        }
      }
      return;
    }
    bx -= a.prev_blocks; gx -= a.prev_blocks;
  }
  int wt;
  if constexpr (PACK) {  // the wave's tile from the pack table (-1: an empty slot of the pack)
    const cintp pk = reinterpret_cast<cintp>(reinterpret_cast<size_t>(a.packs)) + (size_t)xcd_tile(bx, gx) * WAVES;
    wt = pk[wv];
  } else {
    wt = __builtin_amdgcn_readfirstlane(xcd_tile(bx, gx) * WAVES + wv);
  }
  // ONEG (one graph): the four waves of a workgroup all belong to it, so their graph-update partial sums are added in the
  // workgroup (one barrier at the very end) and the graph update reads a quarter of the rows.  Several graphs: a workgroup may
  // straddle two graphs, every wave stores its own row and the kernel has no workgroup barrier at all.  (A template
  // parameter, not a run-time branch: the mere presence of the barrier path cost the multi-graph case 4 %.)
  const bool active = PACK ? wt >= 0 : wt < a.n_wtiles;  // wave-uniform
  int tile_g = -1, tile_cnt = 0;  // PACK: the tile's graph and that graph's number of wave tiles
  bool owner = false;             // PACK: this wave's tile is the first of its graph (the pack keeps a graph's tiles adjacent): it runs the graph update
#ifdef GNX_WAVE_STAMPS_BUILD
  unsigned long long wst_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  GNX_WSTAMP(0);
  if constexpr (!ONEG && !PACK) { if (!active) return; }
  float mine = 0.f;  // lane c < C: this wave's total of graph-update column c
  // FFE: the raw edge rows and ef' of the lane's two edges, kept until the end of the kernel (the FeedForward runs LAST, when nothing of
  // the node phase is live any more: 100 of its ~125 registers are its own)
  float xr[FFE ? EPT : 1][FFE ? DE1 : 1];
  float xst[2 * EPT];  // FFE: (mean, 1/sigma) of the lane's edge rows from gn1 — gn2 normalises the same rows (parked in the wave's slice beside the rows)
#pragma unroll
  for (int i = 0; i < 2 * EPT; ++i) xst[i] = 0.f;
  bool ffv[EPT];
  size_t ff_row0 = 0;
#pragma unroll
  for (int i = 0; i < EPT; ++i) ffv[i] = false;
  do {
  if constexpr (ONEG || PACK) { if (!active) break; }
  const auto A0 = phase_args<FFE>(a);  // tile record, loads, LayerNorm of the node-side rows
  float* s_out = s_mem + wv * WSL;                                        // ef' of the wave's tile
  float* s_pd = s_out + (TEW * OE + 4);                                   // per node: bias' + We[:, dst-seg]*nf[n]
  unsigned char* s_dst = reinterpret_cast<unsigned char*>(s_pd + (64 * OE + 4));  // tile-local destination of each edge

  const cintp tw = reinterpret_cast<cintp>(reinterpret_cast<size_t>(A0->wtiles)) + (size_t)wt * (sizeof(Tile) / sizeof(int));
  const int n0 = tw[0], n1 = tw[1], e0 = tw[2], e1 = tw[3], g = tw[4];  // s_load_dwordx8
  if constexpr (PACK) {
    tile_g = g; tile_cnt = tw[7];
    const cintp pk = reinterpret_cast<cintp>(reinterpret_cast<size_t>(A0->packs)) + (size_t)xcd_tile(bx, gx) * WAVES;
    const int wprev = wv > 0 ? pk[wv > 0 ? wv - 1 : 0] : -1;  // (a graph's first tile never follows an empty slot: slots fill from the left)
    owner = wprev < 0 || (reinterpret_cast<cintp>(reinterpret_cast<size_t>(A0->wtiles)) + (size_t)wprev * (sizeof(Tile) / sizeof(int)))[4] != g;
  }
  const int nn = n1 - n0, ne = e1 - e0;
  GNX_WSTAMP(1);  // (the stamp's own s_waitcnt lgkmcnt(0) makes this "tile record arrived")

  const size_t r = blockIdx.y;
  const float* __restrict__ ef = DE > 0 ? A0->ef + r * (size_t)A0->E * DE : nullptr;
  const float* __restrict__ nf = DN > 0 ? A0->nf + r * (size_t)A0->N * DN : nullptr;
  const cfloatp gf = DG > 0 ? as_const(A0->gf + (r * (size_t)A0->G + g) * DG) : nullptr;

  // ---- issue every load up front, branch-free (indices clamped into the tile; results of clamped lanes unused) ----
  const bool is_node = lane < nn;
  const int nl = lane < nn ? lane : nn - 1;
  const int cp0 = A0->colptr[n0 + nl], cp1 = A0->colptr[n0 + nl + 1];
  float xn[1][DN1];
  if constexpr (DN > 0) load_row<DN>(nf + (size_t)(n0 + nl) * DN, xn[0]);
  const int cn0 = ne < TEW ? ne : TEW;
  float x[EPT][DE1];
  float xs[EPT][DN1];
  int src[EPT];
  // (Order of the requests.  The per-wave stamps of the diagnostic build — tools/wave_stamps.py — show a median wave of C2 spending
  // ~25 k of its ~43 k shader clocks between "tile record arrived" and "gathers issued": it waits for its source indices, and those
  // return only when the HBM queues reach them behind the edge rows EVERY wave requested in the first microsecond (46 MB at
  // ~4.5 TB/s).  Requesting the indices ahead of the edge rows, so that in-order vmcnt lets the gathers go out while the rows still
  // stream, changes nothing measurable (25.4 vs 26.4 us/step): the indices of a late-dispatched wave still queue behind the rows
  // of the earlier ones.  Round 2 also HELD the edge rows back until the indices had arrived (indices -> wait -> gathers + rows in one
  // burst, so that the gather's round trip runs under the row stream): slower everywhere — C2 25.9 vs 25.3 us/step, 512 graphs 21.6 vs
  // 20.8, 4096 graphs 25.2 vs 24.9 (profiles/r02_ab_hold_rows.log): the extra dependent round trip at the start costs more than the
  // overlap returns.)
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    int ec = lane + 64 * i;
    ec = ec < cn0 ? ec : cn0 - 1;
    int e = e0 + (ec > 0 ? ec : 0);
    e = e < A0->E ? e : A0->E - 1;
    if constexpr (DE > 0) load_row<DE>(ef + (size_t)e * DE, x[i]);
    if constexpr (DN > 0) src[i] = A0->rowval[e];
  }
  if constexpr (DN > 0) {
#pragma unroll
    for (int i = 0; i < EPT; ++i) load_row<DN>(nf + (size_t)src[i] * DN, xs[i]);
  }
  float gfr[1][DG1];
#pragma unroll
  for (int k = 0; k < DG; ++k) gfr[0][k] = gf[k];
  if constexpr (LN) {  // gn1(x) applied in registers: rows are normalised as they arrive (gathered rows once per edge)
    ln_row<DN>(xn[0], A0->ln_g[1], A0->ln_b[1], A0->ln_eps, A0->ln_mode);
    ln_row<DG>(gfr[0], A0->ln_g[2], A0->ln_b[2], A0->ln_eps, A0->ln_mode);
  }  // (the edge rows and the gathered rows are normalised at the start of the edge phase: their loads are still in flight)
  GNX_WSTAMP(2);  // every load issued
  const auto A1 = phase_args<FFE>(a);  // node-side preparation + edge phase
  const cfloatp We = as_const(A1->We);
  const cfloatp be = as_const(A1->be ? A1->be : k_zero_bias);
  // ---- lanes as nodes: destination index of each in-edge, per-node part of the edge update ----
  //   pd[n] = be + We[:, gf-seg] * gf[g] + We[:, dst-seg] * nf[n]      (edgefninput.jl:5-6 hoisted out of the edge loop)
  if (is_node) {
    if (nn > 1)
      for (int e = cp0 - e0; e < cp1 - e0; ++e) s_dst[e] = (unsigned char)lane;
    if constexpr (OE > 0) {  // weight rows are read j-contiguous (one s_load_dwordx{4,8,16} per row)
      float b[1][OE1];
#pragma unroll
      for (int j = 0; j < OE; ++j) b[0][j] = be[j];
      fma_rows<DG, OE, 1, DG1>(We + (DE + 2 * DN) * OE, gfr, b);
      fma_rows<DN, OE, 1, DN1>(We + (DE + DN) * OE, xn, b);
#pragma unroll
      for (int j = 0; j < OE; ++j) s_pd[lane * OE + j] = b[0][j];
    }
  }
  __builtin_amdgcn_wave_barrier();
  GNX_WSTAMP(3);  // node-side preparation done (needed colptr, own nf row)

  // ---- lanes as edges ----
  float psum[OE1];  // single-node tiles only: this lane's share of the node's edge sum
#pragma unroll
  for (int j = 0; j < OE1; ++j) psum[j] = 0.f;
  for (int c0 = 0; c0 < ne; c0 += TEW) {
    const int cn = (ne - c0) < TEW ? (ne - c0) : TEW;
    if (c0 > 0) {  // further chunks of a single-node tile with a huge in-degree: loaded in place
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        int ec = lane + 64 * i;
        ec = ec < cn ? ec : cn - 1;
        const int e = e0 + c0 + ec;
        if constexpr (DE > 0) load_row<DE>(ef + (size_t)e * DE, x[i]);
        if constexpr (DN > 0) {
          src[i] = A1->rowval[e];
          load_row<DN>(nf + (size_t)src[i] * DN, xs[i]);
        }
        if constexpr (FFE) {
#pragma unroll
          for (int k = 0; k < DE; ++k) xr[i][k] = x[i][k];
        }
        if constexpr (LN) {
          ln_row<DE>(x[i], A1->ln_g[0], A1->ln_b[0], A1->ln_eps, A1->ln_mode);
          ln_row<DN>(xs[i], A1->ln_g[1], A1->ln_b[1], A1->ln_eps, A1->ln_mode);
        }
      }
    }
    if constexpr (LN) {
      if (c0 == 0) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
          if constexpr (FFE) {
#pragma unroll
            for (int k = 0; k < DE; ++k) xr[i][k] = x[i][k];
            ln_row<DE>(x[i], A1->ln_g[0], A1->ln_b[0], A1->ln_eps, A1->ln_mode, &xst[2 * i]);
          } else {
            ln_row<DE>(x[i], A1->ln_g[0], A1->ln_b[0], A1->ln_eps, A1->ln_mode);
          }
          ln_row<DN>(xs[i], A1->ln_g[1], A1->ln_b[1], A1->ln_eps, A1->ln_mode);
        }
      }
    }
    constexpr int KE = DE + DN;  // weight rows applied per edge (the dst / gf rows were hoisted into pd)
    if constexpr (OE > 0 && KE * OE > 96) {
      // Many weights: all EPT edges of the lane advance together over groups of weight rows
      float acc[EPT][OE1];
      bool valid[EPT];
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const int el = lane + 64 * i;
        valid[i] = el < cn;
        const int elc = valid[i] ? el : cn - 1;
        const int dl = nn > 1 ? (int)s_dst[elc] : 0;
#pragma unroll
        for (int j = 0; j < OE; ++j) acc[i][j] = s_pd[dl * OE + j];
      }
      if constexpr (EPT == 2 && (KE * OE) % 2 == 0 && GNX_WAVE_PK) {
        // the two edges of the lane as register pairs, the (DE + DN) x OE weights (ef rows then src rows: consecutive in We) streamed
        // through SGPR groups into packed FMAs (PkStream above)
        P2 accp[OE1], xp[KE];
        PkStream<KE * OE, OE, KE, GNX_PK_GS> st{We, xp, accp};
        st.template issue<0>();
        if constexpr (FFE) {  // the raw rows wait in the (still unused) ef' slots of the lane's own two edges while the product runs
#pragma unroll
          for (int m = 0; m < EPT; ++m)
#pragma unroll
            for (int k = 0; k < DE; ++k) s_out[(lane + 64 * m) * OE + k] = xr[m][k];
          // (the per-node rows s_pd were read into acc[] above — LDS operations of one wave execute in order — and are dead from here on)
          *reinterpret_cast<v4f_t*>(s_pd + 4 * lane) = v4f_t{xst[0], xst[1], xst[2], xst[3]};
          asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int j = 0; j < OE; ++j) { accp[j].x = acc[0][j]; accp[j].y = acc[1][j]; }
#pragma unroll
        for (int k = 0; k < DE; ++k) { xp[k].x = x[0][k]; xp[k].y = x[1][k]; }
#pragma unroll
        for (int k = 0; k < DN; ++k) { xp[DE + k].x = xs[0][k]; xp[DE + k].y = xs[1][k]; }
        pin_pairs<OE1>(accp);
        pin_pairs<KE>(xp);
        st.template run<0>();
#pragma unroll
        for (int j = 0; j < OE; ++j) { acc[0][j] = accp[j].x; acc[1][j] = accp[j].y; }
        if constexpr (FFE) {
          asm volatile("" ::: "memory");
#pragma unroll
          for (int m = 0; m < EPT; ++m)
#pragma unroll
            for (int k = 0; k < DE; ++k) xr[m][k] = s_out[(lane + 64 * m) * OE + k];
          asm volatile("" ::: "memory");
        }
      } else {
        fma_rows<DE, OE, EPT, DE1>(We, x, acc);
        fma_rows<DN, OE, EPT, DN1>(We + DE * OE, xs, acc);
      }
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const int el = lane + 64 * i;
        if constexpr (FFE) {  // (identity / relu only: the transcendental expansions of the run-time switch cost registers on every path)
          if (A1->act_e == 1) {
#pragma unroll
            for (int j = 0; j < OE; ++j) acc[i][j] = relu_f(acc[i][j]);
          }
        } else {
          act_row<OE1>(acc[i], A1->act_e);
        }
        if (valid[i]) {
          if constexpr (!FFE) store_row<OE>(A1->ef_out + (r * (size_t)A1->E + e0 + c0 + el) * OE, acc[i]);
          if (nn > 1 || FFE) {
#pragma unroll
            for (int j = 0; j < OE; ++j) s_out[el * OE + j] = acc[i][j];
          }
          if (nn == 1) {
#pragma unroll
            for (int j = 0; j < OE; ++j) psum[j] += acc[i][j];
          }
        }
      }
      if constexpr (FFE) {  // (the host launches this form only for batches without a node of more than 64 * EPT in-edges: one chunk per tile)
#pragma unroll
        for (int m = 0; m < EPT; ++m) ffv[m] = valid[m];
        ff_row0 = r * (size_t)A1->E + e0 + c0 + lane;
      }
    } else
    if constexpr (OE > 0) {
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const int el = lane + 64 * i;
        if (el < cn) {
          const int dl = nn > 1 ? (int)s_dst[el] : 0;
          float acc[OE1];
#pragma unroll
          for (int j = 0; j < OE; ++j) acc[j] = s_pd[dl * OE + j];
#pragma unroll
          for (int k = 0; k < DE; ++k)
#pragma unroll
            for (int j = 0; j < OE; ++j) acc[j] = fmaf(We[k * OE + j], x[i][k], acc[j]);
#pragma unroll
          for (int k = 0; k < DN; ++k)
#pragma unroll
            for (int j = 0; j < OE; ++j) acc[j] = fmaf(We[(DE + k) * OE + j], xs[i][k], acc[j]);
          act_row<OE1>(acc, A1->act_e);
          store_row<OE>(A1->ef_out + (r * (size_t)A1->E + e0 + c0 + el) * OE, acc);
          if (nn > 1) {
#pragma unroll
            for (int j = 0; j < OE; ++j) s_out[el * OE + j] = acc[j];
          } else {
#pragma unroll
            for (int j = 0; j < OE; ++j) psum[j] += acc[j];
          }
        }
      }
    }
    if constexpr (FFE) break;  // (one chunk per tile: the launcher's condition — the reload path and what it keeps alive are not compiled)
  }
  __builtin_amdgcn_wave_barrier();
  GNX_WSTAMP(4);  // edge phase done (needed ef rows, rowval, gathered rows; ef' stores issued)
  const auto A2 = phase_args<FFE>(a);  // node phase
  const cfloatp Wn = as_const(A2->Wn);
  const cfloatp bn = as_const(A2->bn ? A2->bn : k_zero_bias);

  // ---- lanes as nodes: edge->node sum (nodefninput.jl:3), node update ----
  float v[C1];  // per-lane contribution to the tile's graph-level partial sums: [agg ; nf']
#pragma unroll
  for (int c = 0; c < C1; ++c) v[c] = 0.f;
  if constexpr (OE > 0) {
    if (nn == 1) {
#pragma unroll
      for (int j = 0; j < OE; ++j) {
        const float tot = wave_sum(psum[j]);
        v[j] = lane == 0 ? tot : 0.f;
      }
    } else if (is_node) {  // contiguous segmented sum (edges are dst-sorted, src/pad.jl:30), fixed order
      for (int e = cp0 - e0; e < cp1 - e0; ++e) {
#pragma unroll
        for (int j = 0; j < OE; ++j) v[j] += s_out[e * OE + j];
      }
    }
  }
  if constexpr (ON > 0) {
    if (is_node) {
      float acc1[1][ON1];
      float (&acc)[ON1] = acc1[0];
#pragma unroll
      for (int j = 0; j < ON; ++j) acc[j] = bn[j];
      fma_rows<DG, ON, 1, DG1>(Wn + (OE + DN) * ON, gfr, acc1);
      float vin[1][OE1];
#pragma unroll
      for (int k = 0; k < OE; ++k) vin[0][k] = v[k];
      fma_rows<OE, ON, 1, OE1>(Wn, vin, acc1);
      fma_rows<DN, ON, 1, DN1>(Wn + OE * ON, xn, acc1);
      act_row<ON1>(acc, A2->act_n);
#pragma unroll
      for (int j = 0; j < ON; ++j) v[OE + j] = acc[j];
      store_row<ON>(A2->nf_out + (r * (size_t)A2->N + n0 + lane) * ON, acc);
    }
  }

  // ---- per-tile partial sums for the graph update (graphfninput.jl:3-4): sum_e ef' = sum_n agg[n], sum_n nf'.
  //      Stored transposed [c][tile] so the graph update reads them with 16-B loads. ----
  if (A2->og > 0) {
    if constexpr (C > 0) {
      float sel[(C + 15) / 16];
#pragma unroll
      for (int g = 0; g < (C + 15) / 16; ++g) sel[g] = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float s16 = row16_sum(v[c]);  // (unconditionally: a DPP reduction inside the select's branch would run with a partial EXEC mask)
        sel[c >> 4] = (lane & 15) == (c & 15) ? s16 : sel[c >> 4];
      }
      // Lane l of every DPP row now holds its ROW's sum of columns l%16 (+16, +32, +48); the four rows are added as (r0 + r1) + (r2 + r3)
      // — the association wave_sum() uses, so the bits are the ones the per-column form (4 readlanes + 3 adds + a select per column)
      // produced — by two lane exchanges per 16 columns instead.
      float tot = 0.f;
#pragma unroll
      for (int g = 0; g < (C + 15) / 16; ++g) {
        float w = sel[g];
        w += __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, __float_as_int(w)));
        w += __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(w)));
        tot = (lane >> 4) == g ? w : tot;
      }
      mine = lane < C ? tot : mine;
    }
  }
  GNX_WSTAMP(5);  // node phase done
  } while (0);
#ifdef GNX_WAVE_STAMPS_BUILD
  if (active && g_wave_dbg && lane == 0 && blockIdx.y == 0) {
    wst_[6] = clock64();
    unsigned long long* o = g_wave_dbg + (size_t)wt * 8;
    for (int i = 0; i < 7; ++i) o[i] = wst_[i];
    o[7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);  // XCC_ID, HW_ID
  }
#endif
  const auto A3 = phase_args<FFE>(a);
  if (A3->og > 0) {
    if constexpr (C > 0) {
      const size_t r = blockIdx.y;
      constexpr int CP = (C + 3) / 4 * 4;
      float* __restrict__ pbase = A3->partials + r * (size_t)n_rows * CP;
      if constexpr (PACK) {
        __shared__ float s_rows[WAVES][C1];
        if (lane < C) s_rows[wv][lane] = active ? mine : 0.f;
        GraphWaveRegs gr;
        if (active && owner) graph_wave_prefetch<C>(a, lane, gr);  // in flight under the barrier
        __syncthreads();
        if (active && owner) {
          float rr[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) rr[k] = (k < tile_cnt && wv + k < WAVES && lane < C) ? 0.f + s_rows[(wv + k) < WAVES ? wv + k : 0][lane < C ? lane : 0] : 0.f;
          const float xs = ((rr[0] + rr[1]) + (rr[2] + rr[3])) + ((rr[4] + rr[5]) + (rr[6] + rr[7]));
          graph_update_wave<C>(a, xs, tile_g, r, lane, s_mem + wv * WSL, gr);
        }
      } else if constexpr (ONEG) {
        __shared__ float s_blk[WAVES][C1];
        if (lane < C) s_blk[wv][lane] = mine;
        __syncthreads();
        if (wv == 0) {
          float tot = 0.f;
          if (lane < C) tot = (s_blk[0][lane] + s_blk[1][lane]) + (s_blk[2][lane] + s_blk[3][lane]);
          store_partial_row<C>(tot, pbase + (size_t)xcd_tile(bx, gx) * CP, lane);
        }
      } else {
        store_partial_row<C>(mine, pbase + (size_t)wt * CP, lane);
      }
    }
  }
  if constexpr (FFE) {
    if (active) {
      // y = (x + ef') + (b2 + W2 act1(W1 gn2(x) + b1)) for the lane's two edges: k_core_post_s's body, fed from registers
      // (its arguments are read from the kernel-argument segment HERE, through a pointer the compiler cannot connect to the loads at the top
      // of the kernel: held from there, these ~20 scalars sit in lanes of a vector register across the whole block part)
      const auto ka = phase_args<true>(a);
      const float* const ffe_b1 = ka->ffe_b1;
      const float* const ffe_b2 = ka->ffe_b2;
      const int ffe_act2 = ka->ffe_act2;
      float* const ef_out_t = ka->ef_out;
      P2 z[DE1], acc2[DE1];
      CorePostStream<DE1, false> fs{as_const(ka->ffe_w1), as_const(ka->ffe_w2), as_const(ffe_b1 ? ffe_b1 : k_zero_bias), ka->ffe_act1, z, acc2};
      fs.G0.issue(fs.template group_ptr<0>());
      const cfloatp b2 = as_const(ffe_b2 ? ffe_b2 : k_zero_bias), g2 = as_const(ka->ffe_g2), be2 = as_const(ka->ffe_be2);
      float rs[EPT][DE1];
      const float* s_ef = s_mem + wv * WSL;  // ef' of the lane's two edges, still in the wave's slice
      const v4f_t st4 = *reinterpret_cast<const v4f_t*>(s_ef + (TEW * OE + 4) + 4 * lane);  // (mean, 1/sigma) of the two rows, parked beside them
#pragma unroll
      for (int m = 0; m < EPT; ++m) {
#pragma unroll
        for (int k = 0; k < DE; ++k) rs[m][k] = xr[m][k] + s_ef[(lane + 64 * m) * OE + k];  // the two residual terms (gncore.jl:56-59)
#pragma unroll
        for (int k = 0; k < DE; ++k) { xr[m][k] -= st4[2 * m]; xr[m][k] *= st4[2 * m + 1]; }  // x-hat: normalise()'s two operations on the same operands
#pragma unroll
        for (int k = 0; k < DE; ++k) {
          const float v = fmaf(g2[k], xr[m][k], be2[k]);
          if (m == 0) { z[k].x = v; acc2[k].x = b2[k]; } else { z[k].y = v; acc2[k].y = b2[k]; }
        }
      }
      pin_pairs<DE1>(z);
      pin_pairs<DE1>(acc2);
      fs.template run<0>();
#pragma unroll
      for (int m = 0; m < EPT; ++m) {
        float o[DE1];
#pragma unroll
        for (int k = 0; k < DE; ++k) o[k] = m == 0 ? acc2[k].x : acc2[k].y;
        if (ffe_act2 == 1) {
#pragma unroll
          for (int k = 0; k < DE; ++k) o[k] = relu_f(o[k]);
        }
#pragma unroll
        for (int k = 0; k < DE; ++k) o[k] = rs[m][k] + o[k];
        if (ffv[m]) store_row<OE>(ef_out_t + (ff_row0 + 64 * m) * OE, o);
      }
    }
  }
}

template <int DE, int DN, int DG, int OE, int ON, int EPT, bool LN = false, bool ONEG = false, bool PACK = false, bool FFE = false, bool CHAIN = false>
__global__ __launch_bounds__(PACK ? kPackThreads : kThreads) __attribute__((amdgpu_num_sgpr(GNX_WAVE_SGPRS))) void k_block_wave(BlockArgs a, int n_rows) {
  static_assert(!FFE, "the FFE form is k_block_wave_ffe");
  block_wave_body<DE, DN, DG, OE, ON, EPT, LN, ONEG, PACK, false, CHAIN>(a, n_rows);
}
template <int DE, int DN, int DG, int OE, int ON, int EPT, bool ONEG>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4))) void k_block_wave_ffe(BlockArgs a, int n_rows) {
  block_wave_body<DE, DN, DG, OE, ON, EPT, true, ONEG, false, true, false>(a, n_rows);
}

// Graph update for the wave path: one workgroup per graph (one wavefront when the graph has <= 256 partial rows).
template <int C, bool ONEG = false>
__global__ void k_graph_t(BlockArgs a, int n_rows) {
  extern __shared__ float s_g[];
  constexpr int CP = (C + 3) / 4 * 4;
  const int g = blockIdx.x;
  // one graph: k_block_wave stored one row per WORKGROUP (4 wave tiles); several graphs: one row per wave tile
  const int t0 = ONEG ? 0 : a.wtile_off[g], t1 = ONEG ? (a.n_wtiles + 3) / 4 : a.wtile_off[g + 1];
  const float* base = a.partials + blockIdx.y * (size_t)n_rows * CP;
  if (blockDim.x == 64) graph_update_rows<C, true, 16>(a, base, g, blockIdx.y, t0, t1, (int)threadIdx.x, 64, s_g);
  else graph_update_rows<C, false, 16>(a, base, g, blockIdx.y, t0, t1, (int)threadIdx.x, (int)blockDim.x, s_g);
}

}  // namespace gnx

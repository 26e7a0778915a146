// The projected edge update of a wide block / GNCore at 128 -> 128 (edgefninput.jl:2-7 regrouped, gnblock.jl:57-60)
//
//     ef'[e] = act( We_e^T gn1(ef[e]) + Ps[src(e)] + Pd[dst(e)] )        Ps = We_s^T nf, Pd = We_d^T nf + b' (+ gf fold): the node projections
//
// with its fp32 products carried by the bf16 matrix cores (six terms of an exact three-way split of both operands, fp32 accumulation: see
// gnx_ffn_x6.hip) — and the outputs the rest of the wide block reads: the per-destination sums of ef' (one row per destination and 64-row
// chunk of the tile: the node update's first segment, format of k_rows_gemm's fused aggregation) and the tile's column sums (graph update).
//
// Structure of k_ffn_x6 with ONE product: transposed domain, a wave's 32 edge rows on the lanes with their three bf16 parts resident in
// registers (96), the weight block as prepared fragments per 32-output slice through LDS (LDS-DMA, double-buffered), 48 MFMAs per slice and wave.
// What k_rows_gemm spends between its matrix instructions on fp32 — 209 us of matrix time at peak for 1M edges — is 80 us here; the kernel is
// left with its traffic (ef in, ef' out, 1M gathered 512-byte projection rows).  Per slice: the gathered addends of the slice are requested
// BEFORE its matrix instructions; the 32 x 32 block goes through a wave-private LDS slice into (row, 16-byte quad) form — 8 rows x 128
// contiguous bytes per instruction for the gathers and the store alike —, the finished values stay there for the per-destination sums
// (edges are dst-sorted: a destination is a contiguous run) and the column sums, both in a fixed order.
// 256 threads = 4 waves = one 128-edge tile of the handle's edge-tile table; 72 KB of LDS: two workgroups per CU.
#include <cstdio>

#include "gnx_device.h"
#include "gnx_x6_stats.h"

namespace gnx {

typedef float f32x16e __attribute__((ext_vector_type(16)));
typedef float f32x4e __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8e __attribute__((ext_vector_type(8)));

namespace {
constexpr int ER = 32, EW = 4, EBM = ER * EW;  // rows per wave, waves, rows per workgroup (= the edge tiles' row cap)
constexpr int EK = 128, EOUT = 128;            // K = de, outputs = oe
constexpr int EKS = EK / 16, ENOB = EOUT / 32, ENF = 3 * EKS;
constexpr int ESLB = ENF * 1024;               // bytes of one 32-output slice of prepared weight fragments
constexpr int ELDE = 36;                       // floats per staged row (32 + 4)

__device__ __forceinline__ unsigned ecvt2(float x0, float x1) {
  typedef float f2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 b2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f2_{x0, x1}, b2_));
}
__device__ __forceinline__ void esplit2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = ecvt2(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = ecvt2(r0, r1);
  l = ecvt2(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}
}  // namespace

// W ([K = 128][ldw] row-major, the first 128 columns) -> per 32-output slice ob one block of ENF fragments of 1 KB = 64 lanes x 8 bf16:
//   fragment 3 s + p (s: k16-step, p: part), lane (m, h), j:  part_p( W[16 s + 8 h + j][32 ob + m] )
// (n_out < 32 * n_ob: the columns beyond n_out are zero — the narrow form's single, padded slice)
// gamma != nullptr: the rows are scaled by gamma[k] first — a LayerNorm's scale folded into the weights it feeds (W^T (gamma . xhat + beta) =
// (gamma . W)^T xhat + W^T beta; the constant W^T beta comes from k_fold_beta)
__global__ void k_edge_x6_prep(const float* __restrict__ W, int ldw, int n_out, int n_ob, __bf16* __restrict__ Wp, const float* __restrict__ gamma) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (ob, s, lane, j pair)
  if (idx >= n_ob * EKS * 64 * 4) return;
  const int jp = idx & 3, lane = (idx >> 2) & 63, s = (idx >> 8) % EKS, ob = (idx >> 8) / EKS;
  const int m = lane & 31, h = lane >> 5, k = 16 * s + 8 * h + 2 * jp;
  unsigned hh, mm, ll;
  const bool in = 32 * ob + m < n_out;
  const float g0 = gamma ? gamma[k] : 1.f, g1 = gamma ? gamma[k + 1] : 1.f;
  esplit2(in ? g0 * W[(size_t)k * ldw + 32 * ob + m] : 0.f, in ? g1 * W[(size_t)(k + 1) * ldw + 32 * ob + m] : 0.f, hh, mm, ll);
  unsigned* o = reinterpret_cast<unsigned*>(Wp) + ((size_t)ob * ENF + 3 * s) * 256 + lane * 4 + jp;
  o[0] = hh; o[256] = mm; o[512] = ll;
}

struct EdgeX6Args {
  const Tile* tiles;
  const float* ef;         // [R][E][128]
  size_t E;
  const float* ln_stats;   // [R][E][2] (mean, 1/sigma) or nullptr
  const float* ln_g;
  const float* ln_b;
  const __bf16* Wp;
  const float* psrc;       // [R][N][128]
  const float* pdst;       // [R][N][128] (bias and gf fold included)
  size_t N;
  const int* src;          // rowval [E]
  const int* dst;          // edge_dst [E]
  int act;
  float* out;              // [R][E][128]
  float* colsum;           // [R][n_tiles][128] or nullptr
  size_t n_tiles;
  float* agg_out;          // [R][n_agg_rows][128] or nullptr
  size_t n_agg_rows;
  const int* chunk_row0;   // [2 n_tiles + 1]
  int oe;                  // NARROW form: output width (1..32); else 128
  int ln_inline;           // no ln_stats: the row statistics of gn1 are computed here, in registers (gnx_x6_stats.h), with ln_eps / ln_mode
  float ln_eps;
  int ln_mode;
  // ENCODER form (KSX = 2): the UNPROJECTED edge update of a block whose inputs are narrow — (10, 5, .) => 128: README ex.3's / config 4's encoder —
  // W^T [ef ; nf[src] ; nf[dst]] + b with all 20 inputs of a row assembled in the lane's registers (one zero-padded K = 32)
  const float* nf;         // [R][N][5]
  const float* bias;       // [128] or nullptr
  const float* bias_g;     // [R][G][128] (bias + gf fold) or nullptr
  int G;
};

// NARROW: a block whose edge output is at most 32 wide (config 4's decoder: 128 -> 3) — ONE slice of zero-padded weight fragments, the addends and
// the outputs as single floats (rows of a.oe), no per-destination sums (the node update of such a block adds up the ef' rows itself: 12 bytes each)
// KSX: k16-steps of the contraction — 8: the projected form (K = 128 edge inputs, two gathered projection rows as addends); 2: the ENCODER form
// ((10, 5, .) => 128 unprojected: ef, nf[src], nf[dst] assembled per lane into one zero-padded K = 32, the bias as the only addend)
// TRANS: the activation is tanh / sigmoid / gelu (the run-time switch of act_apply, inlined per slice: three quarters of the kernel's code); else
// identity / relu
template <bool NARROW, int KSX, bool TRANS>
__global__ __launch_bounds__(64 * EW) __attribute__((amdgpu_waves_per_eu(2, NARROW ? 3 : (KSX == 2 ? 4 : 2)))) void k_edge_x6(EdgeX6Args a) {
  constexpr bool ENC = KSX == 2;
  static_assert(KSX == EKS || (KSX == 2 && !NARROW), "K = 128, or the encoder's K = 32");
  constexpr int NOBK = NARROW ? 1 : ENOB;
  constexpr int NFX = 3 * KSX, SLBX = NFX * 1024;  // fragments / bytes of one 32-output slice of prepared weights
  const int OUTW = NARROW ? a.oe : EOUT;
  __shared__ __attribute__((aligned(16))) unsigned char s_wa[SLBX];
  __shared__ __attribute__((aligned(16))) unsigned char s_wb[SLBX];
  __shared__ __attribute__((aligned(16))) float s_e[EBM * ELDE];  // the finished 32-column block of the tile, [row][36]
  __shared__ __attribute__((aligned(16))) float s_cs[32 * 32];    // column-sum partials [row group][column]
  __shared__ int s_src[EBM], s_dst[EBM];
  __shared__ int s_seg[2][66];  // per 64-row pass: first row of every destination run; [n_seg] = valid rows of the pass; [65] = n_seg

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, n = lane & 31;
  const int tile_id = blockIdx.x;
  const size_t r = blockIdx.y;
  const Tile t = a.tiles[tile_id];
  const int row0 = t.e0, rows = t.e1 - t.e0;
  if (rows <= 0) return;  // (whole workgroup)
  int agg_row0[2] = {0, 0};
  if (a.agg_out) { agg_row0[0] = a.chunk_row0[2 * tile_id]; agg_row0[1] = a.chunk_row0[2 * tile_id + 1]; }

  auto stage = [&](int ob, unsigned char* dst) {
    const unsigned char* srcp = reinterpret_cast<const unsigned char*>(a.Wp) + (size_t)ob * SLBX;
#pragma unroll
    for (int i = 0; i < (NFX + EW - 1) / EW; ++i) {
      const int pc = wv + EW * i;
      if (NFX % EW == 0 || pc < NFX)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcp + (size_t)pc * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
    }
  };
  stage(0, s_wa);
  if (tid < EBM) {
    const int rc = tid < rows ? tid : rows - 1;
    s_src[tid] = a.src[row0 + rc];
    s_dst[tid] = a.dst[row0 + rc];
  }

  // ---- the wave's rows as B fragments (gn1 on load), three bf16 parts ----
  const int lrow = wv * ER + n;
  const int lrc = lrow < rows ? lrow : rows - 1;
  const float* __restrict__ zrow = a.ef + (r * a.E + (size_t)row0 + lrc) * EK;
  bf16x8e zh[KSX], zm[KSX], zl[KSX];
  if constexpr (ENC) {
    // The row's 20 inputs, lane-local (no LDS hop, no division): k = 16 s + 8 hi + j of the zero-padded K = 32 is
    //   lane half 0: ef[0..7] (step 0), nf[dst][1..4] + four zeros (step 1);   half 1: ef[8], ef[9], nf[src][0..4], nf[dst][0] (step 0), zeros (step 1)
    // (k_edge_enc_prep lays the weight rows out in that order).  Rows are 40 / 20 bytes: dword-aligned vector loads.
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
    typedef unsigned u32x4e __attribute__((ext_vector_type(4)));
    const float* __restrict__ efp = a.ef + (r * a.E + (size_t)row0 + lrc) * 10;
    const float* __restrict__ nfs = a.nf + (r * a.N + (size_t)a.src[row0 + lrc]) * 5;
    const float* __restrict__ nfd = a.nf + (r * a.N + (size_t)a.dst[row0 + lrc]) * 5;
    float v0[8], v1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (hi == 0) {
      const f32x4u e0 = *reinterpret_cast<const f32x4u*>(efp), e1 = *reinterpret_cast<const f32x4u*>(efp + 4), d1 = *reinterpret_cast<const f32x4u*>(nfd + 1);
      v0[0] = e0.x; v0[1] = e0.y; v0[2] = e0.z; v0[3] = e0.w; v0[4] = e1.x; v0[5] = e1.y; v0[6] = e1.z; v0[7] = e1.w;
      v1[0] = d1.x; v1[1] = d1.y; v1[2] = d1.z; v1[3] = d1.w;
    } else {
      const f32x2u e2 = *reinterpret_cast<const f32x2u*>(efp + 8);
      const f32x4u s0 = *reinterpret_cast<const f32x4u*>(nfs);
      v0[0] = e2.x; v0[1] = e2.y; v0[2] = s0.x; v0[3] = s0.y; v0[4] = s0.z; v0[5] = s0.w; v0[6] = nfs[4]; v0[7] = nfd[0];
    }
    unsigned ph[4], pm[4], pl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) esplit2(v0[2 * j], v0[2 * j + 1], ph[j], pm[j], pl[j]);
    zh[0] = __builtin_bit_cast(bf16x8e, u32x4e{ph[0], ph[1], ph[2], ph[3]});
    zm[0] = __builtin_bit_cast(bf16x8e, u32x4e{pm[0], pm[1], pm[2], pm[3]});
    zl[0] = __builtin_bit_cast(bf16x8e, u32x4e{pl[0], pl[1], pl[2], pl[3]});
#pragma unroll
    for (int j = 0; j < 4; ++j) esplit2(v1[2 * j], v1[2 * j + 1], ph[j], pm[j], pl[j]);
    zh[KSX - 1] = __builtin_bit_cast(bf16x8e, u32x4e{ph[0], ph[1], ph[2], ph[3]});
    zm[KSX - 1] = __builtin_bit_cast(bf16x8e, u32x4e{pm[0], pm[1], pm[2], pm[3]});
    zl[KSX - 1] = __builtin_bit_cast(bf16x8e, u32x4e{pl[0], pl[1], pl[2], pl[3]});
  } else {
    float mu = 0.f, inv = 1.f;
    const bool ln = a.ln_stats != nullptr || a.ln_inline != 0;
    if (a.ln_stats != nullptr) {
      const float2 st = reinterpret_cast<const float2*>(a.ln_stats)[r * a.E + (size_t)row0 + lrc];
      mu = st.x; inv = st.y;
    }
    f32x4e raw[EKS][2];
#pragma unroll
    for (int s = 0; s < EKS; ++s) {
      raw[s][0] = *reinterpret_cast<const f32x4e*>(zrow + 16 * s + 8 * hi);
      raw[s][1] = *reinterpret_cast<const f32x4e*>(zrow + 16 * s + 8 * hi + 4);
    }
    if (a.ln_inline != 0) x6_row_stats(raw, a.ln_eps, a.ln_mode, mu, inv);
    typedef unsigned u32x4e __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int s = 0; s < KSX && s < EKS; ++s) {
      float v[8] = {raw[s][0].x, raw[s][0].y, raw[s][0].z, raw[s][0].w, raw[s][1].x, raw[s][1].y, raw[s][1].z, raw[s][1].w};
      if (ln) {
        const f32x4e g0 = *reinterpret_cast<const f32x4e*>(a.ln_g + 16 * s + 8 * hi), g1 = *reinterpret_cast<const f32x4e*>(a.ln_g + 16 * s + 8 * hi + 4);
        const f32x4e b0 = *reinterpret_cast<const f32x4e*>(a.ln_b + 16 * s + 8 * hi), b1 = *reinterpret_cast<const f32x4e*>(a.ln_b + 16 * s + 8 * hi + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(gg[j], (v[j] - mu) * inv, bb[j]);
      }
      unsigned ph[4], pm[4], pl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) esplit2(v[2 * j], v[2 * j + 1], ph[j], pm[j], pl[j]);
      zh[s] = __builtin_bit_cast(bf16x8e, u32x4e{ph[0], ph[1], ph[2], ph[3]});
      zm[s] = __builtin_bit_cast(bf16x8e, u32x4e{pm[0], pm[1], pm[2], pm[3]});
      zl[s] = __builtin_bit_cast(bf16x8e, u32x4e{pl[0], pl[1], pl[2], pl[3]});
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // slice 0's pieces of this wave have landed
  __syncthreads();                                  // ... everybody's; s_src / s_dst too
  // destination runs of the two 64-row passes (rows are dst-sorted), by wave 0 and wave 1
  if (a.agg_out && wv < 2) {
    const int pass = wv;
    const int nvalid = min(max(rows - 64 * pass, 0), 64);
    const int d = lane < nvalid ? s_dst[64 * pass + lane] : -1;
    const int dprev = lane > 0 && lane < nvalid ? s_dst[64 * pass + lane - 1] : -2;
    const bool head = lane < nvalid && d != dprev;
    const unsigned long long mask = __ballot(head);
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    if (head) s_seg[pass][rank] = lane;
    if (lane == 0) { const int ns = __popcll(mask); s_seg[pass][ns] = nvalid; s_seg[pass][65] = ns; }
  }

  const int er = lane >> 3, eq = lane & 7;  // (row er + 8 i of the wave's 32, 16-byte quad eq of the 32-column block)
  const int act_floor = a.act == 1 ? 0 : (int)0x80000000;  // relu as an integer maximum of the float's bits with 0 (identity: with INT_MIN)
  const bool wave_full = rows >= (wv + 1) * ER;
  float* sE = s_e + wv * (ER * ELDE);
  const float* __restrict__ ps = ENC ? nullptr : a.psrc + r * a.N * OUTW;
  const float* __restrict__ pd = ENC ? nullptr : a.pdst + r * a.N * OUTW;
  const float* __restrict__ benc = !ENC ? nullptr : (a.bias_g ? a.bias_g + (r * a.G + (size_t)t.g) * EOUT : a.bias);  // (an edge tile lies inside one graph)
  float* __restrict__ outp = a.out + (r * a.E + (size_t)row0) * OUTW;
  int gs[4], gd[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { gs[i] = s_src[wv * ER + er + 8 * i]; gd[i] = s_dst[wv * ER + er + 8 * i]; }

  // the gathered addends of a slice: 8 rows x 128 contiguous bytes per instruction, from both projection tables
  auto gather = [&](int ob, f32x4e (&us)[4], f32x4e (&ud)[4]) {
    if constexpr (ENC) return;  // (no gathered addends: the slice adds its bias quad itself)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (NARROW) {
        float s4[4], d4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool in = 4 * eq + j < OUTW;
          s4[j] = in ? ps[(size_t)gs[i] * OUTW + 4 * eq + j] : 0.f;
          d4[j] = in ? pd[(size_t)gd[i] * OUTW + 4 * eq + j] : 0.f;
        }
        us[i] = f32x4e{s4[0], s4[1], s4[2], s4[3]};
        ud[i] = f32x4e{d4[0], d4[1], d4[2], d4[3]};
      } else {
        us[i] = *reinterpret_cast<const f32x4e*>(ps + (size_t)gs[i] * EOUT + 32 * ob + 4 * eq);
        ud[i] = *reinterpret_cast<const f32x4e*>(pd + (size_t)gd[i] * EOUT + 32 * ob + 4 * eq);
      }
    }
  };
  // slice ob: its addends (us, ud) were requested a slice ahead — behind the previous slice's stores, in front of its sums — and the next slice's are requested here
  auto slice = [&](int ob, const unsigned char* cur, unsigned char* nxt, f32x4e (&us)[4], f32x4e (&ud)[4], f32x4e (&usn)[4], f32x4e (&udn)[4]) {
    if (ob + 1 < NOBK) stage(ob + 1, nxt);
    const unsigned char* wb = cur + lane * 16;
    f32x16e acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    bf16x8e A[2][3];
#pragma unroll
    for (int p3 = 0; p3 < 3; ++p3) A[0][p3] = *reinterpret_cast<const bf16x8e*>(wb + p3 * 1024);
#pragma unroll
    for (int s = 0; s < KSX; ++s) {
      const int c = s & 1;
      if (s + 1 < KSX) {
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) A[c ^ 1][p3] = *reinterpret_cast<const bf16x8e*>(wb + (3 * (s + 1) + p3) * 1024);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][1], zm[s], acc, 0, 0, 0);  // small terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][2], zh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][0], zl[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][1], zh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][0], zm[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][0], zh[s], acc, 0, 0, 0);
    }
    // the gathered addends are needed now — and with them (in-order counter) the next slice's fragments have landed; the stores below are never waited for
    f32x4e bq = {0.f, 0.f, 0.f, 0.f};  // ENCODER form: the slice's bias quad is the only addend (the same for every row; 36 registers instead of the four addend arrays)
    if constexpr (ENC) { if (benc) bq = *reinterpret_cast<const f32x4e*>(benc + 32 * ob + 4 * eq); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (!ENC)
      asm volatile("" : "+v"(us[0]), "+v"(us[1]), "+v"(us[2]), "+v"(us[3]), "+v"(ud[0]), "+v"(ud[1]), "+v"(ud[2]), "+v"(ud[3]));  // (their loads were issued a slice ago: the wait above is theirs)
    // C/D layout (lane (n, hi): outputs 8 g + 4 hi + (0..3) of row n in registers 4 g ..) -> the wave's slice of s_e -> (row, quad)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<f32x4e*>(sE + n * ELDE + 8 * g + 4 * hi) = f32x4e{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lr = er + 8 * i;
      f32x4e v = *reinterpret_cast<const f32x4e*>(sE + lr * ELDE + 4 * eq);
      if constexpr (ENC) v += bq;
      else { v += us[i]; v += ud[i]; }
      float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (TRANS) vv[e] = act_apply(vv[e], a.act);
        else vv[e] = __int_as_float(max(__float_as_int(vv[e]), act_floor));
      }
      v = f32x4e{vv[0], vv[1], vv[2], vv[3]};
      // (wave_full — every row of the wave inside the tile — is wave-uniform: the common case stores without a per-lane predicate.  A predicated
      //  store is an exec-mask branch around each instruction; round 5 measured what sixteen of them per slice cost k_edge_n: 100 us of 473)
      const bool ok = wave_full || wv * ER + lr < rows;
      if (!ok) v = f32x4e{0.f, 0.f, 0.f, 0.f};  // (rows beyond the tile: zero for the sums below)
      *reinterpret_cast<f32x4e*>(sE + lr * ELDE + 4 * eq) = v;
      if constexpr (NARROW) {
        const float o4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (ok && 4 * eq + j < OUTW) outp[(size_t)(wv * ER + lr) * OUTW + 4 * eq + j] = o4[j];
      } else {
        if (wave_full) *reinterpret_cast<f32x4e*>(outp + (size_t)(wv * ER + lr) * EOUT + 32 * ob + 4 * eq) = v;
        else if (ok) *reinterpret_cast<f32x4e*>(outp + (size_t)(wv * ER + lr) * EOUT + 32 * ob + 4 * eq) = v;
      }
    }
    if (ob + 1 < NOBK) gather(ob + 1, usn, udn);  // under the sums below and the next slice's matrix instructions
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (LDS traffic only) the finished block of all four waves is in s_e; the next slice's fragments are complete
    const int q4 = tid & 7, grp = tid >> 3;           // 8 quads x 32 row groups
    f32x4e c4 = {0.f, 0.f, 0.f, 0.f};  // this thread's share of the tile's column sums
    if (!NARROW && a.agg_out) {
      // per-destination sums: groups 0-15 take the runs of pass 0, groups 16-31 those of pass 1 (16 runs per sweep; the 1M-edge graph has ~7 per
      // pass), four rows of a run requested at a time.  Every valid row lies in exactly one run: the column sums are the sums of the run sums.
      const int pass = grp >> 4, g16 = grp & 15;
      const int n_seg = s_seg[pass][65];
      float* agg = a.agg_out + (r * a.n_agg_rows + (size_t)agg_row0[pass]) * EOUT + 32 * ob + 4 * q4;
      const float* base = s_e + 64 * pass * ELDE + 4 * q4;
      for (int sgm = g16; sgm < n_seg; sgm += 16) {
        const int r0 = s_seg[pass][sgm], r1 = s_seg[pass][sgm + 1];
        f32x4e t4 = {0.f, 0.f, 0.f, 0.f};
        for (int rr = r0; rr < r1; rr += 4) {
          f32x4e u[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) u[j] = *reinterpret_cast<const f32x4e*>(base + min(rr + j, r1 - 1) * ELDE);
#pragma unroll
          for (int j = 0; j < 4; ++j) { if (rr + j < r1) t4 += u[j]; }
        }
        *reinterpret_cast<f32x4e*>(agg + (size_t)sgm * EOUT) = t4;
        c4 += t4;
      }
    } else if (a.colsum) {  // (no fused aggregation: the rows themselves, grp, grp + 32, .. ascending)
#pragma unroll
      for (int i = 0; i < 4; ++i) c4 += *reinterpret_cast<const f32x4e*>(s_e + (grp + 32 * i) * ELDE + 4 * q4);
    }
    if (a.colsum) *reinterpret_cast<f32x4e*>(s_cs + grp * 32 + 4 * q4) = c4;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // s_e may be overwritten by the next slice; the column-sum partials are complete
    if (a.colsum && tid < 32) {  // fixed order: the 32 groups ascending
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 32; ++w) sum += s_cs[w * 32 + tid];
      if (!NARROW || tid < OUTW) a.colsum[(r * a.n_tiles + (size_t)tile_id) * OUTW + 32 * ob + tid] = sum;
    }
  };
  static_assert(ENOB % 2 == 0, "slice loop unrolled by two");
  f32x4e ua[4], da[4], ub[4], db[4];
  gather(0, ua, da);
  if constexpr (NARROW) {
    slice(0, s_wa, s_wb, ua, da, ub, db);
  } else {
    for (int ob = 0; ob < ENOB; ob += 2) {
      slice(ob, s_wa, s_wb, ua, da, ub, db);
      slice(ob + 1, s_wb, s_wa, ub, db, ua, da);
    }
  }
}

// ---- the node projections that feed it: Ps = Ws^T gn1(nf), Pd = Wd^T gn1(nf) + b (+ gf fold per graph), nf 64 wide, 128 outputs each — both tables in ONE
//      launch on the same scheme (k_rows_gemm's two-weight-block launch on the fp32 matrix instruction takes 59 us for C2's 100k nodes; 3.3 GFLOP are 21 us
//      of fp32 matrix time and 8 us as six bf16 terms; the tables are 102 MB written).  Eight 32-output slices (four per table) of 12 prepared fragments,
//      double-buffered by LDS-DMA; a wave's 32 node rows on the lanes (48 registers); a slice's block through the wave's staging slice into (row, quad)
//      form, bias quad added on the destination table, 8 rows x 128 contiguous bytes per store.  No sums: one barrier per slice (the weight buffers).
namespace {
constexpr int PK = 64, PKS = PK / 16, PNF = 3 * PKS, PSLB = PNF * 1024, PNOB = 8;
}

// Ws, Wd ([64][ldw] row-major, the first 128 columns of each) -> per slice ob (table ob / 4, outputs 32 (ob % 4) + m) one block of PNF fragments (k_edge_x6_prep's format)
__global__ void k_proj_x6_prep(const float* __restrict__ Ws, const float* __restrict__ Wd, int ldw, __bf16* __restrict__ Wp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (ob, s, lane, j pair)
  if (idx >= PNOB * PKS * 64 * 4) return;
  const int jp = idx & 3, lane = (idx >> 2) & 63, s = (idx >> 8) % PKS, ob = (idx >> 8) / PKS;
  const int m = lane & 31, h = lane >> 5, k = 16 * s + 8 * h + 2 * jp;
  const float* __restrict__ W = ob < 4 ? Ws : Wd;
  const int col = 32 * (ob & 3) + m;
  unsigned hh, mm, ll;
  esplit2(W[(size_t)k * ldw + col], W[(size_t)(k + 1) * ldw + col], hh, mm, ll);
  unsigned* o = reinterpret_cast<unsigned*>(Wp) + ((size_t)ob * PNF + 3 * s) * 256 + lane * 4 + jp;
  o[0] = hh; o[256] = mm; o[512] = ll;
}

struct ProjX6Args {
  const Tile* tiles;       // node tiles: rows [n0, n1) of graph g
  const float* nf;         // [R][N][64]
  size_t N;
  const float* ln_stats;   // [R][N][2] (mean, 1/sigma) or nullptr
  const float* ln_g;
  const float* ln_b;
  const __bf16* Wp;
  const float* bias;       // [128] or nullptr: the destination table's bias
  const float* bias_g;     // [R][G][128] (bias + gf fold) or nullptr
  int G;
  float* out_s;            // [R][N][128]
  float* out_d;            // [R][N][128]
  // k_edge_n's form (gnx_edge_n.hip): the source side is multiplied per EDGE from the raw 64-wide row, so only the destination table is produced
  // (only_d: slices 4..7) and the normalised rows gn1(nf) — the very values this kernel splits — are written out as the table the edges gather
  // (zn_out [R][N][64]; nullptr without a LayerNorm: the edges gather nf itself)
  float* zn_out;
  int only_d;
};

__global__ __launch_bounds__(64 * EW) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_proj_x6(ProjX6Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char s_wa[PSLB];
  __shared__ __attribute__((aligned(16))) unsigned char s_wb[PSLB];
  // the staging area holds HALF a wave's block (16 rows): with all 32 rows the workgroup needs 43 008 bytes of LDS = three workgroups per CU = 768 for the
  // chip — C2's 100k nodes are 782 tiles, 1.02 rounds: the last 14 tiles ran alone behind the rest.  33 792 bytes = four per CU, one round up to 131k nodes.
  __shared__ __attribute__((aligned(16))) float s_e[EBM / 2 * ELDE];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, n = lane & 31;
  const size_t r = blockIdx.y;
  const Tile t = a.tiles[blockIdx.x];
  const int row0 = t.n0, rows = t.n1 - t.n0;
  if (rows <= 0) return;  // (whole workgroup)
  auto stage = [&](int ob, unsigned char* dst) {
    const unsigned char* srcp = reinterpret_cast<const unsigned char*>(a.Wp) + (size_t)ob * PSLB;
#pragma unroll
    for (int i = 0; i < PNF / EW; ++i) {
      const int pc = wv + EW * i;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcp + (size_t)pc * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
    }
  };
  const int ob0 = a.only_d ? 4 : 0;
  stage(ob0, s_wa);
  // ---- the wave's rows as B fragments (gn1 on load), three bf16 parts ----
  const int lrow = wv * ER + n;
  const int lrc = lrow < rows ? lrow : rows - 1;
  const float* __restrict__ zrow = a.nf + (r * a.N + (size_t)row0 + lrc) * PK;
  bf16x8e zh[PKS], zm[PKS], zl[PKS];
  {
    float mu = 0.f, inv = 1.f;
    const bool ln = a.ln_stats != nullptr;
    if (ln) {
      const float2 st = reinterpret_cast<const float2*>(a.ln_stats)[r * a.N + (size_t)row0 + lrc];
      mu = st.x; inv = st.y;
    }
    typedef unsigned u32x4e __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int s = 0; s < PKS; ++s) {
      const f32x4e r0 = *reinterpret_cast<const f32x4e*>(zrow + 16 * s + 8 * hi), r1 = *reinterpret_cast<const f32x4e*>(zrow + 16 * s + 8 * hi + 4);
      float v[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
      if (ln) {
        const f32x4e g0 = *reinterpret_cast<const f32x4e*>(a.ln_g + 16 * s + 8 * hi), g1 = *reinterpret_cast<const f32x4e*>(a.ln_g + 16 * s + 8 * hi + 4);
        const f32x4e b0 = *reinterpret_cast<const f32x4e*>(a.ln_b + 16 * s + 8 * hi), b1 = *reinterpret_cast<const f32x4e*>(a.ln_b + 16 * s + 8 * hi + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(gg[j], (v[j] - mu) * inv, bb[j]);
      }
      if (a.zn_out && lrow < rows) {
        float* __restrict__ zo = a.zn_out + (r * a.N + (size_t)row0 + lrow) * PK + 16 * s + 8 * hi;
        *reinterpret_cast<f32x4e*>(zo) = f32x4e{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4e*>(zo + 4) = f32x4e{v[4], v[5], v[6], v[7]};
      }
      unsigned ph[4], pm[4], pl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) esplit2(v[2 * j], v[2 * j + 1], ph[j], pm[j], pl[j]);
      zh[s] = __builtin_bit_cast(bf16x8e, u32x4e{ph[0], ph[1], ph[2], ph[3]});
      zm[s] = __builtin_bit_cast(bf16x8e, u32x4e{pm[0], pm[1], pm[2], pm[3]});
      zl[s] = __builtin_bit_cast(bf16x8e, u32x4e{pl[0], pl[1], pl[2], pl[3]});
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // slice 0's pieces of this wave have landed
  __syncthreads();                                  // ... everybody's
  const int er = lane >> 3, eq = lane & 7;
  const bool wave_full = rows >= (wv + 1) * ER;  // (wave-uniform: the common case stores without a per-lane predicate)
  float* sE = s_e + wv * (ER / 2 * ELDE);
  const float* __restrict__ bp = a.bias_g ? a.bias_g + (r * a.G + (size_t)t.g) * EOUT : a.bias;
  auto slice = [&](int ob, const unsigned char* cur, unsigned char* nxt) {
    if (ob + 1 < PNOB) stage(ob + 1, nxt);
    const unsigned char* wb = cur + lane * 16;
    f32x16e acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    bf16x8e A[2][3];
#pragma unroll
    for (int p3 = 0; p3 < 3; ++p3) A[0][p3] = *reinterpret_cast<const bf16x8e*>(wb + p3 * 1024);
#pragma unroll
    for (int s = 0; s < PKS; ++s) {
      const int c = s & 1;
      if (s + 1 < PKS) {
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) A[c ^ 1][p3] = *reinterpret_cast<const bf16x8e*>(wb + (3 * (s + 1) + p3) * 1024);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][1], zm[s], acc, 0, 0, 0);  // small terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][2], zh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][0], zl[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][1], zh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][0], zm[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][0], zh[s], acc, 0, 0, 0);
    }
    const int tab = ob >> 2, c0 = 32 * (ob & 3) + 4 * eq;
    f32x4e b4 = {0.f, 0.f, 0.f, 0.f};
    if (tab == 1 && bp) b4 = *reinterpret_cast<const f32x4e*>(bp + c0);
    float* __restrict__ outp = (tab ? a.out_d : a.out_s) + (r * a.N + (size_t)row0) * EOUT + c0;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {  // rows 0..15 of the wave's block, then 16..31, through the same 16 staged rows (one wave's LDS operations execute in order)
      if ((n >> 4) == hf) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4e*>(sE + (n & 15) * ELDE + 8 * g + 4 * hi) = f32x4e{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int lr = er + 8 * i;
        f32x4e v = *reinterpret_cast<const f32x4e*>(sE + lr * ELDE + 4 * eq);
        v += b4;
        const int row = wv * ER + 16 * hf + lr;
        if (wave_full) *reinterpret_cast<f32x4e*>(outp + (size_t)row * EOUT) = v;
        else if (row < rows) *reinterpret_cast<f32x4e*>(outp + (size_t)row * EOUT) = v;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next slice's pieces of this wave (the stores too: 2 KB per wave)
    __syncthreads();                                  // everybody's pieces; everybody is done with `cur`
  };
  for (int ob = ob0; ob < PNOB; ob += 2) {
    slice(ob, s_wa, s_wb);
    slice(ob + 1, s_wb, s_wa);
  }
}

size_t proj_x6_scratch_bytes() { return (size_t)PNOB * PSLB; }


bool proj_x6_applies(int dn, int oe, const float* nf, const float* W, const float* out, size_t N) {
  if (form(GNX_FLAG_EDGE_FP32) || form(GNX_FLAG_PROJ_FP32)) return false;  // (the call asked for the fp32 matrix instruction throughout / for the projections alone)
  return dn == PK && oe == EOUT && N >= 4096 && (((uintptr_t)nf | (uintptr_t)W | (uintptr_t)out) & 15) == 0;
}

// Ws, Wd ([64][ldw], the first 128 columns of each) -> the fragments k_proj_x6 stages (scratch: proj_x6_scratch_bytes(), 16-byte aligned)
int32_t launch_proj_x6_prep(const float* Ws, const float* Wd, int ldw, void* scratch, hipStream_t s) {
  ProfScope ps("k_proj_x6_prep", s);
  GNX_LAUNCH(k_proj_x6_prep, dim3((unsigned)((PNOB * PKS * 64 * 4 + 255) / 256)), dim3(256), 0, s, Ws, Wd, ldw, static_cast<__bf16*>(scratch));
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

// Ps = Ws^T z, Pd = Wd^T z + bias (per graph with bias_g), z = nf or gn1(nf) from ln_stats; scratch: proj_x6_scratch_bytes(), 16-byte aligned
int32_t launch_proj_x6(const Tile* tiles, size_t n_tiles, const float* nf, size_t N, const float* ln_stats, const float* ln_g, const float* ln_b, const float* Ws, const float* Wd,
                       int ldw, const float* bias, const float* bias_g, int G, float* out_s, float* out_d, int64_t R, void* scratch, hipStream_t s, bool only_d, float* zn_out) {
  if (n_tiles == 0) return GNX_OK;
  if (!tiles || !nf || !Ws || !Wd || (!out_s && !only_d) || !out_d || !scratch || ((uintptr_t)scratch & 15)) return fail(GNX_ERR_INVALID_ARG, "k_proj_x6: NULL operand or misaligned scratch");
  if ((uintptr_t)zn_out & 15) return fail(GNX_ERR_INVALID_ARG, "k_proj_x6: the table of normalised rows is not 16-byte aligned");
  if (ln_stats && (!ln_g || !ln_b || (((uintptr_t)ln_g | (uintptr_t)ln_b) & 15) || ((uintptr_t)ln_stats & 7))) return fail(GNX_ERR_INVALID_ARG, "k_proj_x6: LayerNorm parameters missing or misaligned");
  if ((((uintptr_t)bias | (uintptr_t)bias_g | (uintptr_t)out_s | (uintptr_t)out_d) & 15)) return fail(GNX_ERR_INVALID_ARG, "k_proj_x6: operand not 16-byte aligned");
  const __bf16* Wp = static_cast<const __bf16*>(prepared_planes(PREP_PROJ, Ws, Wd, ldw));  // made once with the layer (gnx_*_prepare) ...
  if (!Wp) {                                                                                    // ... or by a launch in front of this forward
    if (const int32_t rc = launch_proj_x6_prep(Ws, Wd, ldw, scratch, s)) return rc;
    Wp = static_cast<const __bf16*>(scratch);
  }
  ProjX6Args a{};
  a.tiles = tiles; a.nf = nf; a.N = N; a.ln_stats = ln_stats; a.ln_g = ln_g; a.ln_b = ln_b; a.Wp = Wp; a.bias = bias; a.bias_g = bias_g; a.G = G; a.out_s = out_s; a.out_d = out_d;
  a.zn_out = zn_out; a.only_d = only_d ? 1 : 0;
  ProfScope ps("k_rows_gemm_proj", s);  // (the name the projections have in every profile and bench line)
  GNX_LAUNCH(k_proj_x6, dim3((unsigned)n_tiles, (unsigned)R), dim3(64 * EW), 0, s, a);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

// ---- the node update of a block at core widths on the same scheme: nf' = act(Wn^T [ sum of the in-edges' ef' (128) | gn1(nf) (64) ] + b (+ gf fold per graph)),
//      nodefninput.jl:1-7 + gnblock.jl:66.  k_rows_gemm ran it on the fp32 matrix instruction: 45-49 us for C2's 100k nodes against ~20 us of traffic —
//      and of the forward's kernels at core widths it was the one a foreign bf16 GEMM on another stream was seen to disturb (profiles/r05_mfma_mix_hazard.log).
//      K = 128 + 64 = 192 (12 k16-steps), 64 outputs = two 32-output slices of 36 prepared fragments; BOTH slices are requested up front (72 KB of LDS, two
//      workgroups per CU), the finished block goes through the first slice's buffer — idle by then — into (row, quad) form; a wave's 32 node rows on the lanes.
//      The summed in-edge rows come from the edge kernel's per-destination partial sums: a node's first partial row, plus the first row of every further
//      64-row chunk its in-edges run through (one node in ~6 on the ER graph has a second part), added in chunk order.
namespace {
constexpr int NK = 192, NKS = NK / 16, NNF = 3 * NKS, NSLB = NNF * 1024, NOUT = 64, NNOB = NOUT / 32;
}

// Wn ([192 (+ dg)][ldw] row-major: the 128 rows of the summed edges, then the 64 node rows; the first 64 columns) -> per slice ob NNF fragments (k_edge_x6_prep's format)
__global__ void k_node_x6_prep(const float* __restrict__ W, int ldw, __bf16* __restrict__ Wp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (ob, s, lane, j pair)
  if (idx >= NNOB * NKS * 64 * 4) return;
  const int jp = idx & 3, lane = (idx >> 2) & 63, s = (idx >> 8) % NKS, ob = (idx >> 8) / NKS;
  const int m = lane & 31, h = lane >> 5, k = 16 * s + 8 * h + 2 * jp;
  const int col = 32 * ob + m;
  unsigned hh, mm, ll;
  esplit2(W[(size_t)k * ldw + col], W[(size_t)(k + 1) * ldw + col], hh, mm, ll);
  unsigned* o = reinterpret_cast<unsigned*>(Wp) + ((size_t)ob * NNF + 3 * s) * 256 + lane * 4 + jp;
  o[0] = hh; o[256] = mm; o[512] = ll;
}

struct NodeX6Args {
  const Tile* tiles;       // node tiles: rows [n0, n1) of graph g
  const float* nf;         // [R][N][64]
  size_t N;
  const float* ln_stats;   // [R][N][2] (mean, 1/sigma) or nullptr
  const float* ln_g;
  const float* ln_b;
  const float* agg;        // [R][n_agg_rows][128]: the edge kernel's per-destination partial sums
  size_t n_agg_rows;
  const int* agg_row;      // [N] row of the node's first partial, -1 without in-edges
  const int* agg_parts;    // [N] chunks the node's in-edges run through
  const int* agg_chunk;    // [N] its first chunk
  const int* chunk_row0;   // [2 n_etiles + 1]
  const __bf16* Wp;
  const float* bias;       // [64] or nullptr
  const float* bias_g;     // [R][G][64] (bias + gf fold) or nullptr
  int G;
  int act;                 // identity / relu
  float* out;              // [R][N][64]
  float* colsum;           // [R][n_tiles][64] or nullptr: the tile's column sums (graph update)
  size_t n_tiles;
};

__global__ __launch_bounds__(64 * EW) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_node_x6(NodeX6Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char s_w0[NSLB];  // slice 0's fragments; then the staging area [128][36] + column-sum partials [32][32]
  __shared__ __attribute__((aligned(16))) unsigned char s_w1[NSLB];
  static_assert((EBM * ELDE + 32 * 32) * 4 <= NSLB, "staging area + column-sum partials fit the first weight buffer");
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, n = lane & 31;
  const size_t r = blockIdx.y;
  const Tile t = a.tiles[blockIdx.x];
  const int row0 = t.n0, rows = t.n1 - t.n0;
  if (rows <= 0) return;  // (whole workgroup)
  auto stage = [&](int ob, unsigned char* dst) {
    const unsigned char* srcp = reinterpret_cast<const unsigned char*>(a.Wp) + (size_t)ob * NSLB;
#pragma unroll
    for (int i = 0; i < NNF / EW; ++i) {
      const int pc = wv + EW * i;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcp + (size_t)pc * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
    }
  };
  stage(0, s_w0);
  stage(1, s_w1);
  // ---- the wave's rows as B fragments, three bf16 parts: k16-steps 0..7 = the summed in-edge rows, 8..11 = gn1(nf) ----
  typedef unsigned u32x4e __attribute__((ext_vector_type(4)));
  const int lrow = wv * ER + n;
  const int lrc = lrow < rows ? lrow : rows - 1;
  const size_t node = (size_t)row0 + lrc;
  bf16x8e zh[NKS], zm[NKS], zl[NKS];
  {
    // (the three index tables are read together, and the second part's row — one node in ~6 has one: nearly every wave — is looked up beside the first
    //  part's data: index -> {first rows, second row's index} -> second rows, three round trips instead of five)
    const int arow = a.agg_row[node], parts_raw = a.agg_parts[node], chunk = a.agg_chunk[node];
    const int parts = arow >= 0 ? parts_raw : 0;
    const int prow1 = parts > 1 ? a.chunk_row0[chunk + 1] : 0;
    const float* __restrict__ aggr = a.agg + r * a.n_agg_rows * EOUT;
    f32x4e acc8[EKS][2];
    const f32x4e zero4 = {0.f, 0.f, 0.f, 0.f};
    {
      const float* __restrict__ p0 = aggr + (size_t)(arow >= 0 ? arow : 0) * EOUT + 8 * hi;
#pragma unroll
      for (int s = 0; s < EKS; ++s) {
        acc8[s][0] = *reinterpret_cast<const f32x4e*>(p0 + 16 * s);
        acc8[s][1] = *reinterpret_cast<const f32x4e*>(p0 + 16 * s + 4);
      }
      if (arow < 0) {
#pragma unroll
        for (int s = 0; s < EKS; ++s) { acc8[s][0] = zero4; acc8[s][1] = zero4; }
      }
    }
    // further parts (in chunk order): the loop runs while ANY row of the wave has one left
    if (__any(parts > 1)) {
      for (int p = 1; __any(p < parts); ++p) {
        const bool has = p < parts;
        const int prow = p == 1 ? prow1 : (has ? a.chunk_row0[chunk + p] : 0);
        const float* __restrict__ pp = aggr + (size_t)prow * EOUT + 8 * hi;
#pragma unroll
        for (int s = 0; s < EKS; ++s) {
          const f32x4e u0 = *reinterpret_cast<const f32x4e*>(pp + 16 * s), u1 = *reinterpret_cast<const f32x4e*>(pp + 16 * s + 4);
          if (has) { acc8[s][0] += u0; acc8[s][1] += u1; }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < EKS; ++s) {
      const float v[8] = {acc8[s][0].x, acc8[s][0].y, acc8[s][0].z, acc8[s][0].w, acc8[s][1].x, acc8[s][1].y, acc8[s][1].z, acc8[s][1].w};
      unsigned ph[4], pm[4], pl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) esplit2(v[2 * j], v[2 * j + 1], ph[j], pm[j], pl[j]);
      zh[s] = __builtin_bit_cast(bf16x8e, u32x4e{ph[0], ph[1], ph[2], ph[3]});
      zm[s] = __builtin_bit_cast(bf16x8e, u32x4e{pm[0], pm[1], pm[2], pm[3]});
      zl[s] = __builtin_bit_cast(bf16x8e, u32x4e{pl[0], pl[1], pl[2], pl[3]});
    }
    // gn1(nf) (or nf itself)
    const float* __restrict__ zrow = a.nf + (r * a.N + node) * PK;
    float mu = 0.f, inv = 1.f;
    const bool ln = a.ln_stats != nullptr;
    if (ln) {
      const float2 st = reinterpret_cast<const float2*>(a.ln_stats)[r * a.N + node];
      mu = st.x; inv = st.y;
    }
#pragma unroll
    for (int s = 0; s < PKS; ++s) {
      const f32x4e r0 = *reinterpret_cast<const f32x4e*>(zrow + 16 * s + 8 * hi), r1 = *reinterpret_cast<const f32x4e*>(zrow + 16 * s + 8 * hi + 4);
      float v[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
      if (ln) {
        const f32x4e g0 = *reinterpret_cast<const f32x4e*>(a.ln_g + 16 * s + 8 * hi), g1 = *reinterpret_cast<const f32x4e*>(a.ln_g + 16 * s + 8 * hi + 4);
        const f32x4e b0 = *reinterpret_cast<const f32x4e*>(a.ln_b + 16 * s + 8 * hi), b1 = *reinterpret_cast<const f32x4e*>(a.ln_b + 16 * s + 8 * hi + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(gg[j], (v[j] - mu) * inv, bb[j]);
      }
      unsigned ph[4], pm[4], pl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) esplit2(v[2 * j], v[2 * j + 1], ph[j], pm[j], pl[j]);
      zh[EKS + s] = __builtin_bit_cast(bf16x8e, u32x4e{ph[0], ph[1], ph[2], ph[3]});
      zm[EKS + s] = __builtin_bit_cast(bf16x8e, u32x4e{pm[0], pm[1], pm[2], pm[3]});
      zl[EKS + s] = __builtin_bit_cast(bf16x8e, u32x4e{pl[0], pl[1], pl[2], pl[3]});
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // both slices' pieces of this wave have landed
  __syncthreads();                                  // ... everybody's
  const int er = lane >> 3, eq = lane & 7;
  const bool wave_full = rows >= (wv + 1) * ER;
  float* s_e = reinterpret_cast<float*>(s_w0);      // [128][36] once slice 0's fragments have been read by everyone
  float* s_cs = s_e + EBM * ELDE;                   // [32][32]
  float* sE = s_e + wv * (ER * ELDE);
  const float* __restrict__ bp = a.bias_g ? a.bias_g + (r * a.G + (size_t)t.g) * NOUT : a.bias;
  const int act_floor = a.act == 1 ? 0 : (int)0x80000000;
  float* __restrict__ outp = a.out + (r * a.N + (size_t)row0) * NOUT;
#pragma unroll
  for (int ob = 0; ob < NNOB; ++ob) {
    const unsigned char* wb = (ob == 0 ? s_w0 : s_w1) + lane * 16;
    f32x16e acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    bf16x8e A[2][3];
#pragma unroll
    for (int p3 = 0; p3 < 3; ++p3) A[0][p3] = *reinterpret_cast<const bf16x8e*>(wb + p3 * 1024);
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const int c = s & 1;
      if (s + 1 < NKS) {
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) A[c ^ 1][p3] = *reinterpret_cast<const bf16x8e*>(wb + (3 * (s + 1) + p3) * 1024);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][1], zm[s], acc, 0, 0, 0);  // small terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][2], zh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][0], zl[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][1], zh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][0], zm[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c][0], zh[s], acc, 0, 0, 0);
    }
    if (ob == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave has read slice 0's fragments: their buffer is the staging area now
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<f32x4e*>(sE + n * ELDE + 8 * g + 4 * hi) = f32x4e{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    f32x4e b4 = {0.f, 0.f, 0.f, 0.f};
    if (bp) b4 = *reinterpret_cast<const f32x4e*>(bp + 32 * ob + 4 * eq);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lr = er + 8 * i;
      f32x4e v = *reinterpret_cast<const f32x4e*>(sE + lr * ELDE + 4 * eq);
      v += b4;
      float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) vv[e] = __int_as_float(max(__float_as_int(vv[e]), act_floor));
      v = f32x4e{vv[0], vv[1], vv[2], vv[3]};
      const bool ok = wave_full || wv * ER + lr < rows;
      if (!ok) v = f32x4e{0.f, 0.f, 0.f, 0.f};  // (rows beyond the tile: zero for the column sums)
      if (a.colsum) *reinterpret_cast<f32x4e*>(sE + lr * ELDE + 4 * eq) = v;
      if (wave_full) *reinterpret_cast<f32x4e*>(outp + (size_t)(wv * ER + lr) * NOUT + 32 * ob + 4 * eq) = v;
      else if (ok) *reinterpret_cast<f32x4e*>(outp + (size_t)(wv * ER + lr) * NOUT + 32 * ob + 4 * eq) = v;
    }
    if (a.colsum) {  // the tile's column sums of this slice, fixed order: 32 row groups x 4 rows, then the groups ascending
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const int q4 = tid & 7, grp = tid >> 3;
      f32x4e c4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) c4 += *reinterpret_cast<const f32x4e*>(s_e + (grp + 32 * i) * ELDE + 4 * q4);
      *reinterpret_cast<f32x4e*>(s_cs + grp * 32 + 4 * q4) = c4;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (tid < 32) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 32; ++w) sum += s_cs[w * 32 + tid];
        a.colsum[(r * a.n_tiles + (size_t)blockIdx.x) * NOUT + 32 * ob + tid] = sum;
      }
      if (ob + 1 < NNOB) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the staging area and the partials may be overwritten
    }
  }
}

size_t node_x6_scratch_bytes() { return (size_t)NNOB * NSLB; }

// does the node update of these widths run as k_node_x6?  (the arithmetic flags as for the projections: GNX_FLAG_EDGE_FP32 = the fp32 instruction
// throughout, GNX_FLAG_PROJ_FP32 = the node-side kernels alone)
bool node_x6_applies(int oe, int dn, int on, int act, const float* nf, const float* Wn, const float* out, size_t N) {
  if (form(GNX_FLAG_EDGE_FP32) || form(GNX_FLAG_PROJ_FP32)) return false;
  return oe == EOUT && dn == PK && on == NOUT && (act == GNX_ACT_IDENTITY || act == GNX_ACT_RELU) && N >= 4096 && (((uintptr_t)nf | (uintptr_t)Wn | (uintptr_t)out) & 15) == 0;
}

int32_t launch_node_x6_prep(const float* Wn, int ldw, void* scratch, hipStream_t s) {
  ProfScope ps("k_node_x6_prep", s);
  GNX_LAUNCH(k_node_x6_prep, dim3((unsigned)((NNOB * NKS * 64 * 4 + 255) / 256)), dim3(256), 0, s, Wn, ldw, static_cast<__bf16*>(scratch));
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

// nf' = act(Wn^T [sum of the in-edges' ef' | z] + bias (per graph with bias_g)), z = nf or gn1(nf) from ln_stats; scratch: node_x6_scratch_bytes(), 16-byte aligned
int32_t launch_node_x6(const Tile* tiles, size_t n_tiles, const float* nf, size_t N, const float* ln_stats, const float* ln_g, const float* ln_b, const float* agg,
                       size_t n_agg_rows, const int* agg_row, const int* agg_parts, const int* agg_chunk, const int* chunk_row0, const float* Wn, int ldw, const float* bias,
                       const float* bias_g, int G, int act, float* out, float* colsum, int64_t R, void* scratch, hipStream_t s) {
  if (n_tiles == 0) return GNX_OK;
  if (!tiles || !nf || !agg || !agg_row || !agg_parts || !agg_chunk || !chunk_row0 || !Wn || !out || !scratch || ((uintptr_t)scratch & 15))
    return fail(GNX_ERR_INVALID_ARG, "k_node_x6: NULL operand or misaligned scratch");
  if (ln_stats && (!ln_g || !ln_b || (((uintptr_t)ln_g | (uintptr_t)ln_b) & 15) || ((uintptr_t)ln_stats & 7))) return fail(GNX_ERR_INVALID_ARG, "k_node_x6: LayerNorm parameters missing or misaligned");
  if ((((uintptr_t)bias | (uintptr_t)bias_g | (uintptr_t)agg | (uintptr_t)out) & 15)) return fail(GNX_ERR_INVALID_ARG, "k_node_x6: operand not 16-byte aligned");
  if (ldw != NOUT) return fail(GNX_ERR_INVALID_ARG, "k_node_x6: the node function's weight rows are 64 wide");
  const __bf16* Wp = static_cast<const __bf16*>(prepared_planes(PREP_NODE, Wn, nullptr, ldw));  // made once with the layer (gnx_*_prepare) ...
  if (!Wp) {                                                                                      // ... or by a launch in front of this forward
    if (const int32_t rc = launch_node_x6_prep(Wn, ldw, scratch, s)) return rc;
    Wp = static_cast<const __bf16*>(scratch);
  }
  NodeX6Args a{};
  a.tiles = tiles; a.nf = nf; a.N = N; a.ln_stats = ln_stats; a.ln_g = ln_g; a.ln_b = ln_b; a.agg = agg; a.n_agg_rows = n_agg_rows; a.agg_row = agg_row; a.agg_parts = agg_parts;
  a.agg_chunk = agg_chunk; a.chunk_row0 = chunk_row0; a.Wp = Wp; a.bias = bias; a.bias_g = bias_g; a.G = G; a.act = act; a.out = out; a.colsum = colsum; a.n_tiles = n_tiles;
  ProfScope ps("k_rows_gemm_node", s);  // (the name the node update has in every profile and bench line)
  GNX_LAUNCH(k_node_x6, dim3((unsigned)n_tiles, (unsigned)R), dim3(64 * EW), 0, s, a);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

size_t edge_x6_scratch_bytes() { return sizeof(__bf16) * 3 * (size_t)EK * EOUT; }

// ---- the ENCODER form: (10, 5, .) => 128 unprojected ----
// W ([10 + 5 + 5 (+ dg)][ldw] row-major: the ef rows, the source rows, the destination rows) -> per 32-output slice 6 fragments in k_edge_x6_prep's format over
// the zero-padded K = 32 of the kernel's lane assembly: k 0..9 = ef rows, 10..14 = source rows, 15 = destination row 0, 16..19 = destination rows 1..4, 20..31 = 0
__global__ void k_edge_enc_prep(const float* __restrict__ W, int ldw, __bf16* __restrict__ Wp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (ob, s, lane, j pair)
  if (idx >= ENOB * 2 * 64 * 4) return;
  const int jp = idx & 3, lane = (idx >> 2) & 63, s = (idx >> 8) & 1, ob = idx >> 9;
  const int m = lane & 31, h = lane >> 5;
  auto wrow = [](int k) { return k < 15 ? k : (k == 15 ? 15 : (k < 20 ? k : -1)); };  // (rows 0..9 ef, 10..14 source, 15..19 destination: the weight's own order)
  float w[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int k = 16 * s + 8 * h + 2 * jp + e, row = wrow(k);
    w[e] = row >= 0 ? W[(size_t)row * ldw + 32 * ob + m] : 0.f;
  }
  unsigned hh, mm, ll;
  esplit2(w[0], w[1], hh, mm, ll);
  unsigned* o = reinterpret_cast<unsigned*>(Wp) + ((size_t)ob * 6 + 3 * s) * 256 + lane * 4 + jp;
  o[0] = hh; o[256] = mm; o[512] = ll;
}

size_t edge_enc_scratch_bytes() { return (size_t)ENOB * 6 * 1024; }

int32_t launch_edge_enc_prep(const float* We, int ldw, void* scratch, hipStream_t s) {
  ProfScope ps("k_edge_x6_prep", s);
  GNX_LAUNCH(k_edge_enc_prep, dim3((unsigned)((ENOB * 2 * 64 * 4 + 255) / 256)), dim3(256), 0, s, We, ldw, static_cast<__bf16*>(scratch));
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

// ef' = act(We^T [ef ; nf[src] ; nf[dst]] + bias) at (10, 5, .) => 128 with the per-destination sums and column sums of k_edge_x6; bias_g: per-graph bias (gf fold)
int32_t launch_edge_enc(const Tile* tiles, size_t n_tiles, const float* ef, size_t E, const float* nf, size_t N, const float* We, int ldw, const float* bias, const float* bias_g,
                        int G, const int* src, const int* dst, int act, float* out, float* colsum, float* agg_out, size_t n_agg_rows, const int* chunk_row0, int64_t R,
                        void* scratch, hipStream_t s) {
  if (n_tiles == 0) return GNX_OK;
  if (!tiles || !ef || !nf || !We || !src || !dst || !out || !scratch || (((uintptr_t)scratch | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)bias_g | (uintptr_t)agg_out) & 15))
    return fail(GNX_ERR_INVALID_ARG, "k_edge_x6 (encoder form): NULL or misaligned operand");
  if (agg_out && !chunk_row0) return fail(GNX_ERR_INVALID_ARG, "k_edge_x6 (encoder form): per-destination sums need the chunk table");
  const __bf16* Wp = static_cast<const __bf16*>(prepared_planes(PREP_ENC, We, nullptr, ldw));
  if (!Wp) {
    if (const int32_t rc = launch_edge_enc_prep(We, ldw, scratch, s)) return rc;
    Wp = static_cast<const __bf16*>(scratch);
  }
  EdgeX6Args a{};
  a.tiles = tiles; a.ef = ef; a.E = E; a.Wp = Wp; a.N = N; a.src = src; a.dst = dst; a.act = act; a.out = out; a.colsum = colsum; a.n_tiles = n_tiles;
  a.agg_out = agg_out; a.n_agg_rows = n_agg_rows; a.chunk_row0 = chunk_row0; a.oe = EOUT; a.nf = nf; a.bias = bias; a.bias_g = bias_g; a.G = G;
  ProfScope ps("k_rows_gemm_edge", s);  // (the name the edge update has in every profile and bench line)
  if (act > GNX_ACT_RELU) GNX_LAUNCH((k_edge_x6<false, 2, true>), dim3((unsigned)n_tiles, (unsigned)R), dim3(64 * EW), 0, s, a);
  else GNX_LAUNCH((k_edge_x6<false, 2, false>), dim3((unsigned)n_tiles, (unsigned)R), dim3(64 * EW), 0, s, a);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

// We ([128][ldw], its first 128 columns) -> the fragments k_edge_x6 (and the edge form of k_ffn_x6) stage per 32-output slice
// out[o] = (bias ? bias[o] : 0) + sum_k beta[k] W[k][o], k ascending (K x n_out weights with row distance ldw): the constant a folded LayerNorm leaves behind
__global__ void k_fold_beta(const float* __restrict__ W, int ldw, int K, int n_out, const float* __restrict__ beta, const float* __restrict__ bias, float* __restrict__ out) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n_out) return;
  float acc = bias ? bias[o] : 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(beta[k], W[(size_t)k * ldw + o], acc);
  out[o] = acc;
}
int32_t launch_fold_beta(const float* W, int ldw, int K, int n_out, const float* beta, const float* bias, float* out, hipStream_t s) {
  GNX_LAUNCH(k_fold_beta, dim3((unsigned)((n_out + 127) / 128)), dim3(128), 0, s, W, ldw, K, n_out, beta, bias, out);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

// the folded form's scratch: the planes of (gamma . We), then the 128 floats of We^T beta
size_t edge_x6_fold_scratch_bytes() { return edge_x6_scratch_bytes() + 128 * sizeof(float); }

// ln_gamma / ln_beta != nullptr (128 -> 128 only): the LayerNorm folded in — planes of (gamma . We) and, behind them (edge_x6_scratch_bytes()), We^T beta
int32_t launch_edge_x6_prep(const float* We, int ldw, void* scratch, hipStream_t s, int n_out, const float* ln_gamma, const float* ln_beta) {
  ProfScope ps("k_edge_x6_prep", s);
  const int n_ob = (n_out + 31) / 32;
  GNX_LAUNCH(k_edge_x6_prep, dim3((unsigned)((n_ob * EKS * 64 * 4 + 255) / 256)), dim3(256), 0, s, We, ldw, n_out, n_ob, static_cast<__bf16*>(scratch), ln_gamma);
  GNX_HIP(hipGetLastError());
  if (ln_gamma) {
    if (n_out != EOUT || !ln_beta) return fail(GNX_ERR_INVALID_ARG, "k_edge_x6_prep: the folded form is 128 -> 128 with gamma and beta");
    return launch_fold_beta(We, ldw, EK, EOUT, ln_beta, nullptr, reinterpret_cast<float*>(static_cast<char*>(scratch) + edge_x6_scratch_bytes()), s);
  }
  return GNX_OK;
}

int32_t launch_edge_x6(const Tile* tiles, size_t n_tiles, const float* ef, size_t E, const float* ln_stats, const float* ln_g, const float* ln_b, const float* We, int ldw,
                       const float* psrc, const float* pdst, size_t N, const int* src, const int* dst, int act, float* out, float* colsum, float* agg_out,
                       size_t n_agg_rows, const int* chunk_row0, int64_t R, void* scratch, hipStream_t s, bool ln_inline, float ln_eps, int ln_mode, int oe) {
  if (n_tiles == 0) return GNX_OK;
  if (oe != EOUT && (oe < 1 || oe > 32 || agg_out)) return fail(GNX_ERR_INVALID_ARG, "k_edge_x6: output width 128, or 1..32 without per-destination sums");
  const __bf16* Wp = static_cast<const __bf16*>(prepared_planes(PREP_EDGE, We, nullptr, oe));  // made once with the layer (gnx_*_prepare) ...
  if (!Wp) {                                                                                        // ... or by a launch in front of this forward
    if (const int32_t rc = launch_edge_x6_prep(We, ldw, scratch, s, oe, nullptr, nullptr)) return rc;
    Wp = static_cast<const __bf16*>(scratch);
  }
  EdgeX6Args a{};
  a.tiles = tiles; a.ef = ef; a.E = E; a.ln_stats = ln_stats; a.ln_g = ln_g; a.ln_b = ln_b; a.Wp = Wp; a.psrc = psrc; a.pdst = pdst; a.N = N;
  a.src = src; a.dst = dst; a.act = act; a.out = out; a.colsum = colsum; a.n_tiles = n_tiles; a.agg_out = agg_out; a.n_agg_rows = n_agg_rows; a.chunk_row0 = chunk_row0;
  if (ln_inline) { if (ln_stats || !ln_g || !ln_b) return fail(GNX_ERR_INVALID_ARG, "k_edge_x6: statistics in the kernel exclude a statistics table and need gamma / beta"); a.ln_inline = 1; a.ln_eps = ln_eps; a.ln_mode = ln_mode; }
  a.oe = oe;
  ProfScope ps("k_rows_gemm_edge", s);  // (the name the edge update has in every profile and bench line)
  const bool trans = act > GNX_ACT_RELU;
  if (oe == EOUT) { if (trans) GNX_LAUNCH((k_edge_x6<false, EKS, true>), dim3((unsigned)n_tiles, (unsigned)R), dim3(64 * EW), 0, s, a); else GNX_LAUNCH((k_edge_x6<false, EKS, false>), dim3((unsigned)n_tiles, (unsigned)R), dim3(64 * EW), 0, s, a); }
  else { if (trans) GNX_LAUNCH((k_edge_x6<true, EKS, true>), dim3((unsigned)n_tiles, (unsigned)R), dim3(64 * EW), 0, s, a); else GNX_LAUNCH((k_edge_x6<true, EKS, false>), dim3((unsigned)n_tiles, (unsigned)R), dim3(64 * EW), 0, s, a); }
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

}  // namespace gnx

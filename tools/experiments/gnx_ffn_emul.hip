// NOT PART OF libgnx — kept as the source of a measured experiment (round 2).  Wired into launch_ffn_fused behind an
// environment switch it ran config 4's edge FeedForward in 3.27 ms against 2.41 ms for k_ffn_fused (two cores: 6.54 vs 4.82 ms) and had an
// unresolved parity defect on 0.06 % of the outputs at 1M edges.  Why it is slower although it issues 2.7x fewer matrix-pipe cycles: a
// v_mfma_f32_32x32x16_bf16 consumes 2 KB of operand fragments per 32 cycles, and with one 32 x 32 block per wave and every fragment read
// from LDS the six-term product needs 6 x 1 KB of LDS reads per 192 pipe cycles = 32 B/clk per wave; sixteen waves per CU ask for
// 512 B/clk of an LDS that delivers 256 — plus 32 spilled registers at D = 128 and the split's VALU work between the barriers.  The
// emulation needs its operands in REGISTERS (z fragments resident per tile: 96 VGPRs, two waves per SIMD), not this kernel's structure.
//
// Fused FeedForward of a GNCore with the fp32 products EMULATED on the bf16 matrix cores:
//
//     x = hi + mid + lo  (three bf16 parts: 24 mantissa bits, fp32's exponent range)
//     a*b ~ hh + hm + mh + hl + lh + mm   accumulated in fp32 by v_mfma_f32_32x32x16_bf16
//
// Six 32-cycle MFMAs cover a 32 x 32 x 16 block that takes eight 64-cycle v_mfma_f32_32x32x2_f32: 2.7x less matrix-pipe time, and
// the products are as exact as the fp32 MFMA's (tools/mfma_emul.hip on the MI355X: worst |err| / sum|a||b| 8.8e-8 vs 1.06e-7).
//
// Same mathematics as k_ffn_fused (gnfeedforward.jl:27-40, gncore.jl:56-68), computed in the TRANSPOSED domain so that the hidden
// layer never leaves the registers:  H^T = W1^T z^T  (A = W1^T rows n, B = z rows m),  out^T = W2^T H^T.
// A 32 x 32 accumulator block of H^T has the batch row m on the lane and 16 hidden indices in its registers — which is the B-operand
// layout of the next MFMA up to a permutation of k that the pre-split W2 planes absorb (k_ffn_emul_prep).  512 threads = 8 waves =
// 4 row blocks (mb) x 2 halves of the 64-unit hidden slice (nb): wave (nb, mb) owns the H^T block (32 of the slice's hidden units x its
// 32 rows) and a PARTIAL out^T (all D outputs x its 32 rows, summed over ITS hidden units); the two partials of a row block are
// added once per tile in the epilogue.
#include <algorithm>
#include <cstdio>
#include <type_traits>

#include "gnx_device.h"

namespace gnx {

typedef float f32x16e __attribute__((ext_vector_type(16)));
typedef float f32x4e __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int EBM = 128;   // rows per workgroup
constexpr int EHS = 64;    // hidden units per slice (32 per wave half)
constexpr int EKC = 32;    // K chunk of GEMM1
constexpr int ELD = 40;    // bf16 per LDS row of a 32-k plane: 80 B = 5 x 16 B, conflict-free 16-B reads over 16 rows
template <int I, int N, class F>
__device__ __forceinline__ void static_for_e(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_e<I + 1, N>(f);
  }
}
__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}
// position kk (0..31) of a wave's 32 hidden units in the k order of GEMM2  <->  local hidden index: kk = 16 t + 8 h + j holds the unit
// that sits in accumulator register q = 8 t + j of lane half h, i.e. row (q & 3) + 8 (q >> 2) + 4 h of the 32 x 32 C/D layout
__host__ __device__ inline int hidden_of_slot(int kk) {
  const int t = kk >> 4, h = (kk >> 3) & 1, j = kk & 7, q = 8 * t + j;
  return (q & 3) + 8 * (q >> 2) + 4 * h;
}
}  // namespace

// W1 ([D][H] row-major) -> W1p[p][n][k] (k-contiguous per hidden unit), W2 ([H][D] row-major) -> W2p[p][o][kk] (kk = slot order of
// hidden_of_slot inside every 32-block), p = 0 hi, 1 mid, 2 lo
__global__ void k_ffn_emul_prep(const float* __restrict__ W1, const float* __restrict__ W2, int D, __bf16* __restrict__ W1p, __bf16* __restrict__ W2p) {
  const int H = 4 * D;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D * H) return;
  __bf16 a, b, c;
  {
    const int n = idx / D, k = idx % D;
    split3(W1[(size_t)k * H + n], a, b, c);
    W1p[idx] = a; W1p[(size_t)D * H + idx] = b; W1p[2 * (size_t)D * H + idx] = c;
  }
  {
    const int o = idx / H, kk = idx % H;
    const int n = (kk & ~31) + hidden_of_slot(kk & 31);
    split3(W2[(size_t)n * D + o], a, b, c);
    W2p[idx] = a; W2p[(size_t)D * H + idx] = b; W2p[2 * (size_t)D * H + idx] = c;
  }
}

struct FfnEmulArgs {
  const Tile* tiles;
  int row_kind;
  const float* z;          // [R][rows][D]
  const __bf16* W1p;       // [3][H][D]
  const float* b1;         // [H] or nullptr
  const __bf16* W2p;       // [3][D][H]
  const float* b2;
  const float* add1;
  const float* add2;
  float* out;
  size_t rep_stride;       // rows_total * D
  int act1;
  const float* ln_stats;   // z = gn2(x) on load (see k_ffn_fused), or nullptr
  const float* ln_g;
  const float* ln_b;
};

template <int D>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void k_ffn_emul(FfnEmulArgs a) {
  constexpr int H = 4 * D;
  constexpr int NOB = D / 32;                   // 32-output blocks (all of them per wave: partial sums over the wave's hidden units)
  constexpr int NC1 = D / EKC;                  // K chunks of GEMM1
  constexpr int NSTEP = NC1 + 2;                // + the two k16-steps of GEMM2
  constexpr int NT = 512;
  constexpr int ZP = EBM * ELD;                 // bf16 of one z plane
  constexpr int W1P = EHS * ELD;                // bf16 of one W1 chunk plane
  constexpr int W2P = D * ELD;                  // bf16 of one W2 step plane
  constexpr int REG1 = 3 * (ZP + W1P), REG2 = 3 * W2P;
  constexpr int REGB = 2 * (REG1 > REG2 ? REG1 : REG2);                               // bytes of the operand region
  constexpr int EPIB = 8 * 32 * 33 * 4;                                                // bytes of the epilogue staging
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[REGB > EPIB ? REGB : EPIB];
  __bf16* zp = reinterpret_cast<__bf16*>(s_raw);                 // [3][128][ELD]
  __bf16* w1p = zp + 3 * ZP;                                     // [3][64][ELD]
  __bf16* w2p = reinterpret_cast<__bf16*>(s_raw);                // [3][D][ELD]   (GEMM2 steps; the GEMM1 planes are dead then)
  float* sE = reinterpret_cast<float*>(s_raw);                   // [8][32][33]   (epilogue)
  __shared__ float s_b1[H];
  __shared__ float2 s_ln[EBM];
  __shared__ __attribute__((aligned(16))) f32x4e s_lng[D / 4], s_lnb[D / 4];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nb = wv & 1, mb = wv >> 1;
  const int hi = lane >> 5, l31 = lane & 31;
  const Tile t = a.tiles[blockIdx.x];
  const size_t r = blockIdx.y;
  const int row0 = a.row_kind == 0 ? t.e0 : t.n0;
  const int rows = (a.row_kind == 0 ? t.e1 : t.n1) - row0;
  const float* __restrict__ zb = a.z + r * a.rep_stride + (size_t)row0 * D;

  for (int i = tid; i < H; i += NT) s_b1[i] = a.b1 ? a.b1[i] : 0.f;
  const bool ln = a.ln_stats != nullptr;
  if (ln) {
    if (tid < EBM) s_ln[tid] = reinterpret_cast<const float2*>(a.ln_stats + r * (a.rep_stride / D) * 2)[row0 + (tid < rows ? tid : rows - 1)];
    if (tid >= EBM && tid < EBM + D / 4) { s_lng[tid - EBM] = reinterpret_cast<const f32x4e*>(a.ln_g)[tid - EBM]; s_lnb[tid - EBM] = reinterpret_cast<const f32x4e*>(a.ln_b)[tid - EBM]; }
  }

  f32x16e accO[NOB];
#pragma unroll
  for (int j = 0; j < NOB; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) accO[j][q] = 0.f;

  // staging registers of the next step (global -> registers during this step's MFMAs -> LDS)
  f32x4e ra[2];   // z chunk, fp32: 128 rows x 8 quads
  f32x4e rb[3];   // weight planes, raw 16-B quads of 8 bf16: W1 chunk 768 quads (2 per thread, the second for tid < 256), W2 step 1536 (3)
  const int a_c4 = tid & 7, a_r = tid >> 3;

  auto load_step = [&](int hs, int st) {
    if (st < NC1) {
      const int kc = st * EKC;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = a_r + 64 * i;
        const f32x4e zero = {0.f, 0.f, 0.f, 0.f};
        ra[i] = row < rows ? *reinterpret_cast<const f32x4e*>(zb + (size_t)row * D + kc + 4 * a_c4) : zero;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = tid + NT * i;
        if (q < 768) {  // plane p, hidden unit nn of the slice, quad c of its 32-k chunk row
          const int p = q >> 8, rem = q & 255, nn = rem >> 2, c = rem & 3;
          rb[i] = *reinterpret_cast<const f32x4e*>(a.W1p + (size_t)p * D * H + (size_t)(hs * EHS + nn) * D + kc + 8 * c);
        }
      }
    } else {
      const int tt = st - NC1;  // k16-step of GEMM2: slots [16 tt, 16 tt + 16) of both halves
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int q = tid + NT * i;            // 3 planes x D outputs x (2 halves x 2 quads)
        const int p = q / (4 * D), rem = q % (4 * D), o = rem >> 2, c = rem & 3;
        if (p < 3) rb[i] = *reinterpret_cast<const f32x4e*>(a.W2p + (size_t)p * D * H + (size_t)o * H + hs * EHS + 32 * (c >> 1) + 16 * tt + 8 * (c & 1));
      }
    }
  };
  auto store_step = [&](int st) {
    if (st < NC1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4e v = ra[i];
        if (ln && a_r + 64 * i < rows) {  // (rows beyond the tile stay zero)
          const float2 st2 = s_ln[a_r + 64 * i];
          const f32x4e g = s_lng[st * (EKC / 4) + a_c4], b = s_lnb[st * (EKC / 4) + a_c4];
          v.x = fmaf(g.x, (v.x - st2.x) * st2.y, b.x); v.y = fmaf(g.y, (v.y - st2.x) * st2.y, b.y);
          v.z = fmaf(g.z, (v.z - st2.x) * st2.y, b.z); v.w = fmaf(g.w, (v.w - st2.x) * st2.y, b.w);
        }
        bf16x4 ph, pm, pl;
        { __bf16 x, y, w; split3(v.x, x, y, w); ph[0] = x; pm[0] = y; pl[0] = w; }
        { __bf16 x, y, w; split3(v.y, x, y, w); ph[1] = x; pm[1] = y; pl[1] = w; }
        { __bf16 x, y, w; split3(v.z, x, y, w); ph[2] = x; pm[2] = y; pl[2] = w; }
        { __bf16 x, y, w; split3(v.w, x, y, w); ph[3] = x; pm[3] = y; pl[3] = w; }
        __bf16* d = zp + (a_r + 64 * i) * ELD + 4 * a_c4;
        *reinterpret_cast<bf16x4*>(d) = ph;
        *reinterpret_cast<bf16x4*>(d + ZP) = pm;
        *reinterpret_cast<bf16x4*>(d + 2 * ZP) = pl;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = tid + NT * i;
        if (q < 768) {
          const int p = q >> 8, rem = q & 255, nn = rem >> 2, c = rem & 3;
          *reinterpret_cast<f32x4e*>(w1p + p * W1P + nn * ELD + 8 * c) = rb[i];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int q = tid + NT * i;
        const int p = q / (4 * D), rem = q % (4 * D), o = rem >> 2, c = rem & 3;
        if (p < 3) *reinterpret_cast<f32x4e*>(w2p + p * W2P + o * ELD + 8 * c) = rb[i];  // row o: [half 0: 16 slots | half 1: 16 slots]
      }
    }
  };

  bf16x8 hh[2], hm[2], hl[2];  // the wave's activated hidden block, split, as the B fragments of GEMM2's two k16-steps
  load_step(0, 0);
  for (int hs = 0; hs < H / EHS; ++hs) {
    f32x16e accH;
#pragma unroll
    for (int q = 0; q < 16; ++q) accH[q] = 0.f;
    for (int st = 0; st < NSTEP; ++st) {
      __syncthreads();
      store_step(st);
      __syncthreads();
      if (st + 1 < NSTEP) load_step(hs, st + 1);
      else if (hs + 1 < H / EHS) load_step(hs + 1, 0);
      if (st < NC1) {
        // GEMM1: H^T block (32 hidden units of this half x 32 rows) += W1^T chunk * z^T chunk
#pragma unroll
        for (int s = 0; s < EKC / 16; ++s) {
          const int ao = (32 * nb + l31) * ELD + 16 * s + 8 * hi, bo = (32 * mb + l31) * ELD + 16 * s + 8 * hi;
          const bf16x8 Ah = *reinterpret_cast<const bf16x8*>(w1p + ao), Am = *reinterpret_cast<const bf16x8*>(w1p + W1P + ao),
                       Al = *reinterpret_cast<const bf16x8*>(w1p + 2 * W1P + ao);
          const bf16x8 Bh = *reinterpret_cast<const bf16x8*>(zp + bo), Bm = *reinterpret_cast<const bf16x8*>(zp + ZP + bo),
                       Bl = *reinterpret_cast<const bf16x8*>(zp + 2 * ZP + bo);
          accH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, accH, 0, 0, 0);  // small terms first
          accH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, accH, 0, 0, 0);
          accH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, accH, 0, 0, 0);
          accH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, accH, 0, 0, 0);
          accH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, accH, 0, 0, 0);
          accH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, accH, 0, 0, 0);
        }
        if (st == NC1 - 1) {
          // bias + activation + split, straight into the B fragments of GEMM2 (register q = 8 t + j -> slot j of step t)
          float hv[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) hv[q] = accH[q] + s_b1[hs * EHS + 32 * nb + (q & 3) + 8 * (q >> 2) + 4 * hi];
          switch (a.act1) {
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
          for (int q = 0; q < 16; ++q) {
            __bf16 x, y, w;
            split3(hv[q], x, y, w);
            hh[q >> 3][q & 7] = x; hm[q >> 3][q & 7] = y; hl[q >> 3][q & 7] = w;
          }
        }
      } else {
        // GEMM2, k16-step tt: partial out^T (all outputs x this wave's rows) += W2^T[:, the wave's 16 slots] * H^T[those slots]
        const int tt = st - NC1;
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
          const int ao = (32 * ob + l31) * ELD + 16 * nb + 8 * hi;
          const bf16x8 Ah = *reinterpret_cast<const bf16x8*>(w2p + ao), Am = *reinterpret_cast<const bf16x8*>(w2p + W2P + ao),
                       Al = *reinterpret_cast<const bf16x8*>(w2p + 2 * W2P + ao);
          accO[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, hm[tt], accO[ob], 0, 0, 0);
          accO[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, hh[tt], accO[ob], 0, 0, 0);
          accO[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, hl[tt], accO[ob], 0, 0, 0);
          accO[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, hh[tt], accO[ob], 0, 0, 0);
          accO[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, hm[tt], accO[ob], 0, 0, 0);
          accO[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, hh[tt], accO[ob], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: one pass per 32-output block.  Every wave parks its partial block [o][m] in LDS; a thread (row m, 8 outputs) adds the
  //      two halves' partials, the bias and the residuals and writes two quads of the row ----
  float* out_tile = a.out + r * a.rep_stride + (size_t)row0 * D;
  const float* add1_tile = a.add1 ? a.add1 + r * a.rep_stride + (size_t)row0 * D : nullptr;
  const float* add2_tile = a.add2 ? a.add2 + r * a.rep_stride + (size_t)row0 * D : nullptr;
  const int em = tid >> 2, ec = tid & 3;  // row of the tile, 8-output group of the block
  const int emc = em < rows ? em : rows - 1;
  static_for_e<0, NOB>([&](auto ob_c) {
    constexpr int ob = decltype(ob_c)::value;
    const unsigned off = (unsigned)emc * D + 32u * ob + 8u * ec;
    f32x4e u1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, u2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (add1_tile) { u1[0] = *reinterpret_cast<const f32x4e*>(add1_tile + off); u1[1] = *reinterpret_cast<const f32x4e*>(add1_tile + off + 4); }
    if (add2_tile) { u2[0] = *reinterpret_cast<const f32x4e*>(add2_tile + off); u2[1] = *reinterpret_cast<const f32x4e*>(add2_tile + off + 4); }
    __syncthreads();  // the operand region / the previous pass's staging is free
#pragma unroll
    for (int q = 0; q < 16; ++q) sE[(wv * 32 + (q & 3) + 8 * (q >> 2) + 4 * hi) * 33 + l31] = accO[ob][q];
    __syncthreads();
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ol = 8 * ec + e, w0 = 2 * (em >> 5);
      v[e] = sE[((w0 + 0) * 32 + ol) * 33 + (em & 31)] + sE[((w0 + 1) * 32 + ol) * 33 + (em & 31)];
      if (a.b2) v[e] += a.b2[32 * ob + ol];
    }
    f32x4e o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
    o0 += u1[0]; o1 += u1[1];
    o0 += u2[0]; o1 += u2[1];
    if (em < rows) {
      *reinterpret_cast<f32x4e*>(out_tile + off) = o0;
      *reinterpret_cast<f32x4e*>(out_tile + off + 4) = o1;
    }
  });
}

size_t ffn_emul_scratch_bytes(int d) { return 2 * 3 * (size_t)d * 4 * d * sizeof(__bf16); }

// experimental path of launch_ffn_fused (GNX_FFN_EMUL=1): `scratch` holds the pre-split weight planes (ffn_emul_scratch_bytes, 16-B aligned)
int32_t launch_ffn_emul(const gnx_graphs* h, int entity, const float* z, int d, const gnx_ffn& ff, const float* add1, const float* add2, float* out,
                        int64_t R, hipStream_t s, const float* ln_stats, const gnx_layernorm* ln, void* scratch) {
  const size_t nrows = entity == 0 ? (size_t)h->E : (entity == 1 ? (size_t)h->N : (size_t)h->G);
  if (nrows == 0) return GNX_OK;
  if (!scratch || ((uintptr_t)scratch & 15)) return fail(GNX_ERR_INVALID_ARG, "k_ffn_emul: scratch missing or misaligned");
  __bf16* W1p = static_cast<__bf16*>(scratch);
  __bf16* W2p = W1p + 3 * (size_t)d * 4 * d;
  {
    ProfScope ps("k_ffn_emul_prep", s);
    hipLaunchKernelGGL(k_ffn_emul_prep, dim3((unsigned)((d * 4 * d + 255) / 256)), dim3(256), 0, s, ff.fc1.weight, ff.fc2.weight, d, W1p, W2p);
  }
  FfnEmulArgs a{};
  a.tiles = entity == 0 ? h->d_etiles : (entity == 1 ? h->d_ntiles : h->d_gtiles);
  a.row_kind = entity == 0 ? 0 : 1;
  a.z = z; a.W1p = W1p; a.b1 = ff.fc1.bias; a.W2p = W2p; a.b2 = ff.fc2.bias;
  a.add1 = add1; a.add2 = add2; a.out = out; a.rep_stride = nrows * (size_t)d; a.act1 = ff.fc1.act;
  if (ln_stats) { a.ln_stats = ln_stats; a.ln_g = ln->gamma; a.ln_b = ln->beta; }
  const unsigned n_tiles = (unsigned)(entity == 0 ? h->h_etiles.size() : (entity == 1 ? h->h_ntiles.size() : h->h_gtiles.size()));
  ProfScope ps("k_ffn_emul", s);
  if (d == 128) hipLaunchKernelGGL((k_ffn_emul<128>), dim3(n_tiles, (unsigned)R), dim3(512), 0, s, a);
  else if (d == 64) hipLaunchKernelGGL((k_ffn_emul<64>), dim3(n_tiles, (unsigned)R), dim3(512), 0, s, a);
  else return fail(GNX_ERR_DIMS, "k_ffn_emul: width not instantiated");
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

}  // namespace gnx

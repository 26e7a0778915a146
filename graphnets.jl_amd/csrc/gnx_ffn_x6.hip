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
// 512 threads = 8 waves x 32 rows = 256 rows per workgroup, two waves per SIMD (<= 256 registers), 96 KB of LDS: one workgroup per CU.
#include <cstdio>

#include "gnx_device.h"

namespace gnx {

typedef float f32x16x __attribute__((ext_vector_type(16)));
typedef float f32x4x __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8x __attribute__((ext_vector_type(8)));

namespace {
constexpr int XR = 32;         // rows per wave
constexpr int XW = 8;          // waves per workgroup
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
}  // namespace

// Prepared weights, per hidden slice hs (32 units) one contiguous block of 2 * NF fragments (NF = 3 D / 16) of 1 KB = 64 lanes x 8 bf16:
//   fragment 3 s + p            (s < D/16: k16-step of the first product, p: part)  lane (m, h), j: part_p( W1[16 s + 8 h + j][32 hs + m] )
//   fragment NF + 3 (2 ob + t) + p   (ob < D/32: output block, t < 2: k16-step)     lane (m, h), j: part_p( W2[32 hs + unit(16 t + 8 h + j)][32 ob + m] )
// W1 = fc1.weight ((4D x D) column-major == [D][4D] row-major), W2 = fc2.weight ((D x 4D) column-major == [4D][D] row-major).
__global__ void k_ffn_x6_prep(const float* __restrict__ W1, const float* __restrict__ W2, int D, __bf16* __restrict__ Wp) {
  const int H = 4 * D, NF = 3 * D / 16;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (hs, g, lane, j): g < D/16 groups of three fragments per product
  if (idx >= D * H) return;
  const int j = idx & 7, lane = (idx >> 3) & 63, g = (idx >> 9) % (D / 16), hs = (idx >> 9) / (D / 16);
  const int m = lane & 31, h = lane >> 5;
  __bf16* slice = Wp + (size_t)hs * 2 * NF * 512;
  __bf16 a, b, c;
  {
    const int k = 16 * g + 8 * h + j;
    split3x(W1[(size_t)k * H + 32 * hs + m], a, b, c);
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
};

template <int D>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ffn_x6(FfnX6Args a) {
  constexpr int H = 4 * D;
  constexpr int KS = D / 16;          // k16-steps of the first product
  constexpr int NOB = D / 32;         // 32-output blocks of the second product
  constexpr int NF = 3 * KS;          // fragments per product and slice
  constexpr int SLB = 2 * NF * 1024;  // bytes per slice
  constexpr int NSL = H / XHS;
  static_assert((2 * NF) % XW == 0, "fragments of a slice divide over the waves");
  // (two OBJECTS, and a slice loop unrolled by two: the compiler orders an LDS read behind every LDS-DMA that may alias it — with one array of two
  // buffers it waits for the NEXT slice's pieces in front of this slice's first fragment read)
  __shared__ __attribute__((aligned(16))) unsigned char s_w0[SLB];
  __shared__ __attribute__((aligned(16))) unsigned char s_w1[SLB];
  __shared__ float s_b1[H];

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, n = lane & 31;
  const size_t r = blockIdx.y;
  const size_t row0 = (size_t)blockIdx.x * XBM + (size_t)wv * XR;
  const size_t rows = a.rows;
  // (waves beyond the last row keep working on the clamped last row — they share the barriers — and store nothing)
  const size_t rown = row0 + n < rows ? row0 + n : rows - 1;
  const bool row_ok = row0 + n < rows;
  const float* __restrict__ zrow = a.z + (r * rows + rown) * D;

  // slice 0 on its way to LDS: fragment f of the slice is LDS-DMA piece f (lane l writes bytes [16 l, 16 l + 16) of the piece)
  auto stage = [&](int hs, unsigned char* dst) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(a.Wp) + (size_t)hs * SLB;
#pragma unroll
    for (int i = 0; i < 2 * NF / XW; ++i) {
      const int pc = wv + XW * i;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)pc * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
    }
  };
  stage(0, s_w0);
  for (int i = tid; i < H; i += 512) s_b1[i] = a.b1 ? a.b1[i] : 0.f;

  // ---- the wave's z rows as B fragments: lane (n, hi) holds k = 16 s + 8 hi + j (j < 8) of row n for every k16-step s, in three parts ----
  bf16x8x zh[KS], zm[KS], zl[KS];
  {
    float mu = 0.f, inv = 1.f;
    const bool ln = a.ln_stats != nullptr;
    if (ln) {
      const float2 st = reinterpret_cast<const float2*>(a.ln_stats)[r * rows + rown];
      mu = st.x; inv = st.y;
    }
    f32x4x raw[KS][2];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      raw[s][0] = *reinterpret_cast<const f32x4x*>(zrow + 16 * s + 8 * hi);
      raw[s][1] = *reinterpret_cast<const f32x4x*>(zrow + 16 * s + 8 * hi + 4);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      float v[8] = {raw[s][0].x, raw[s][0].y, raw[s][0].z, raw[s][0].w, raw[s][1].x, raw[s][1].y, raw[s][1].z, raw[s][1].w};
      if (ln) {  // (x - mean) * inv, then fma(gamma, ., beta): the arithmetic of k_layernorm2_v4 / k_ffn_fused
        const f32x4x g0 = *reinterpret_cast<const f32x4x*>(a.ln_g + 16 * s + 8 * hi), g1 = *reinterpret_cast<const f32x4x*>(a.ln_g + 16 * s + 8 * hi + 4);
        const f32x4x b0 = *reinterpret_cast<const f32x4x*>(a.ln_b + 16 * s + 8 * hi), b1 = *reinterpret_cast<const f32x4x*>(a.ln_b + 16 * s + 8 * hi + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(gg[j], (v[j] - mu) * inv, bb[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __bf16 x, y, w;
        split3x(v[j], x, y, w);
        zh[s][j] = x; zm[s][j] = y; zl[s][j] = w;
      }
    }
  }

  f32x16x accO[NOB];
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int q = 0; q < 16; ++q) accO[ob][q] = 0.f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of slice 0 are in LDS
  __syncthreads();                                  // ... and everybody else's; s_b1 too

  auto slice = [&](int hs, const unsigned char* cur, unsigned char* nxt) {
    if (hs + 1 < NSL) stage(hs + 1, nxt);  // into the buffer slice hs - 1 was read from (every wave is past the barrier that ended that slice)
    const unsigned char* wb = cur + lane * 16;
    // ---- H^T block (the slice's 32 hidden units x the wave's 32 rows) = b1 + W1^T z^T ----
    f32x16x accH;
#pragma unroll
    for (int q = 0; q < 16; ++q) accH[q] = s_b1[hs * XHS + (q & 3) + 8 * (q >> 2) + 4 * hi];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bf16x8x Ah = *reinterpret_cast<const bf16x8x*>(wb + (3 * s + 0) * 1024), Am = *reinterpret_cast<const bf16x8x*>(wb + (3 * s + 1) * 1024),
                    Al = *reinterpret_cast<const bf16x8x*>(wb + (3 * s + 2) * 1024);
      accH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, zm[s], accH, 0, 0, 0);  // small terms first
      accH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, zh[s], accH, 0, 0, 0);
      accH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, zl[s], accH, 0, 0, 0);
      accH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, zh[s], accH, 0, 0, 0);
      accH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, zm[s], accH, 0, 0, 0);
      accH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, zh[s], accH, 0, 0, 0);
    }
    // ---- activation, split: register q = 8 t + j is element j of the B fragment of k16-step t (slot 16 t + 8 hi + j) ----
    float hv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) hv[q] = accH[q];
    switch (a.act1) {
      case 0: break;
      case 1:
#pragma unroll
        for (int q = 0; q < 16; ++q) hv[q] = relu_f(hv[q]);
        break;
      default:
#pragma unroll
        for (int q = 0; q < 16; ++q) hv[q] = act_apply(hv[q], a.act1);
        break;
    }
    bf16x8x hh[2], hm[2], hl[2];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      __bf16 x, y, w;
      split3x(hv[q], x, y, w);
      hh[q >> 3][q & 7] = x; hm[q >> 3][q & 7] = y; hl[q >> 3][q & 7] = w;
    }
    // ---- out^T (D outputs x the wave's rows) += W2^T[:, the slice's slots] H^T ----
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int f = NF + 3 * (2 * ob + t);
        const bf16x8x Ah = *reinterpret_cast<const bf16x8x*>(wb + (f + 0) * 1024), Am = *reinterpret_cast<const bf16x8x*>(wb + (f + 1) * 1024),
                      Al = *reinterpret_cast<const bf16x8x*>(wb + (f + 2) * 1024);
        accO[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, hm[t], accO[ob], 0, 0, 0);
        accO[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, hh[t], accO[ob], 0, 0, 0);
        accO[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, hl[t], accO[ob], 0, 0, 0);
        accO[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, hh[t], accO[ob], 0, 0, 0);
        accO[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, hm[t], accO[ob], 0, 0, 0);
        accO[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, hh[t], accO[ob], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of slice hs + 1 have landed
    __syncthreads();                                  // every wave is done with slice hs; slice hs + 1 is complete
  };
  static_assert(NSL % 2 == 0, "slice loop unrolled by two");
  for (int hs = 0; hs < NSL; hs += 2) {
    slice(hs, s_w0, s_w1);
    slice(hs + 1, s_w1, s_w0);
  }

  // ---- epilogue from the C/D layout: lane (n, hi) holds outputs 32 ob + 8 g + 4 hi + (0..3) of its row in registers 4 g .. 4 g + 3 —
  //      one 16-byte access per (ob, g), the two lane halves of a row adjacent ----
  float* __restrict__ orow = a.out + (r * rows + rown) * D;
  const float* __restrict__ r1 = a.add1 ? a.add1 + (r * rows + rown) * D : nullptr;
  const float* __restrict__ r2 = a.add2 ? a.add2 + (r * rows + rown) * D : nullptr;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
    f32x4x u1[4], u2[4], bq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * ob + 8 * g + 4 * hi;
      const f32x4x zero = {0.f, 0.f, 0.f, 0.f};
      u1[g] = r1 ? *reinterpret_cast<const f32x4x*>(r1 + c) : zero;
      u2[g] = r2 ? *reinterpret_cast<const f32x4x*>(r2 + c) : zero;
      bq[g] = a.b2 ? *reinterpret_cast<const f32x4x*>(a.b2 + c) : zero;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * ob + 8 * g + 4 * hi;
      f32x4x v = {accO[ob][4 * g], accO[ob][4 * g + 1], accO[ob][4 * g + 2], accO[ob][4 * g + 3]};
      v += bq[g];
      v += u1[g];
      v += u2[g];
      if (row_ok) *reinterpret_cast<f32x4x*>(orow + c) = v;
    }
  }
}

size_t ffn_x6_scratch_bytes(int d) { return (size_t)3 * d * 4 * d * sizeof(__bf16) * 2; }

bool ffn_x6_applies(const float* z, int d, const gnx_ffn& ff, const float* add1, const float* add2, const float* out, size_t scratch_bytes) {
  if (getenv("GNX_FFN_FP32") != nullptr) return false;  // (read per call: tests compare the two kernels in one process)
  if (d != 128 || ff.fc2.act != GNX_ACT_IDENTITY || scratch_bytes < ffn_x6_scratch_bytes(d)) return false;
  const uintptr_t al = (uintptr_t)z | (uintptr_t)ff.fc2.bias | (uintptr_t)add1 | (uintptr_t)add2 | (uintptr_t)out;
  return (al & 15) == 0;
}

// out = add1 + add2 + fc2(act1(fc1(z))) over `nrows` rows per replica; `scratch`: ffn_x6_scratch_bytes(d), 16-byte aligned, free until the launch has run
int32_t launch_ffn_x6(const float* z, size_t nrows, int d, const gnx_ffn& ff, const float* add1, const float* add2, float* out, int64_t R, hipStream_t s,
                      const float* ln_stats, const gnx_layernorm* ln, void* scratch) {
  if (nrows == 0) return GNX_OK;
  if (!scratch || ((uintptr_t)scratch & 15)) return fail(GNX_ERR_INVALID_ARG, "k_ffn_x6: scratch missing or misaligned");
  if (!z || !ff.fc1.weight || !ff.fc2.weight || !out) return fail(GNX_ERR_INVALID_ARG, "k_ffn_x6: NULL operand");
  if (ln_stats && (!ln || !ln->gamma || !ln->beta || ((uintptr_t)ln->gamma & 15) || ((uintptr_t)ln->beta & 15) || ((uintptr_t)ln_stats & 7)))
    return fail(GNX_ERR_INVALID_ARG, "k_ffn_x6: LayerNorm parameters missing or misaligned");
  __bf16* Wp = static_cast<__bf16*>(scratch);
  {
    ProfScope ps("k_ffn_x6_prep", s);
    GNX_LAUNCH(k_ffn_x6_prep, dim3((unsigned)((d * 4 * d + 255) / 256)), dim3(256), 0, s, ff.fc1.weight, ff.fc2.weight, d, Wp);
    GNX_HIP(hipGetLastError());
  }
  FfnX6Args a{};
  a.z = z; a.Wp = Wp; a.b1 = ff.fc1.bias; a.b2 = ff.fc2.bias; a.add1 = add1; a.add2 = add2; a.out = out; a.rows = nrows; a.act1 = ff.fc1.act;
  if (ln_stats) { a.ln_stats = ln_stats; a.ln_g = ln->gamma; a.ln_b = ln->beta; }
  ProfScope ps("k_ffn_x6", s);
  GNX_LAUNCH((k_ffn_x6<128>), dim3((unsigned)((nrows + XBM - 1) / XBM), (unsigned)R), dim3(512), 0, s, a);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

}  // namespace gnx

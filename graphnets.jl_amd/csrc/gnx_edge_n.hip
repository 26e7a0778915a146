// k_edge_n — the projected edge update of a wide block at (128, 64, .) -> 128 (edgefninput.jl:2-7 regrouped, gnblock.jl:57-60):
//
//     ef'[e] = act( We_e^T gn1(ef[e]) + We_s^T gn1(nf)[src(e)] + Pd[dst(e)] )          Pd = We_d^T gn1(nf) + b' (+ gf fold): one projection per NODE
//
// Successor of k_edge_x6<false> (gnx_edge_x6.hip; round 4: 338 us at 1M edges, memory-bound at 4.9 TB/s of 1.66 GB with the matrix pipe 0.29 busy —
// 0.51 GB of that the 1M gathers of PROJECTED source rows, 512 bytes each).  Two changes, both trading idle matrix time for bytes and barriers:
//   * the SOURCE side is gathered raw — the 256-byte row of gn1(nf) — and multiplied by We_s here: the contraction is K = 128 + 64 instead of 128
//     (+50 % matrix instructions on a pipe that was 0.29 busy), the gather is half as long, and the projection launch writes one table, not two;
//   * the product runs in the NORMAL domain (rows x outputs: the row fragments are the A operand), so that the finished 32 x 32 block has its
//     output on the lane and its rows in the registers: addend, activation, store, per-destination sums and column sums are register arithmetic
//     (gnx_edge_n.h) — no LDS staging, one workgroup barrier per weight hand-over instead of two per 32-output slice.
// The fp32 products are carried by the bf16 matrix cores as in k_ffn_x6: six terms of an exact three-way split of both operands, fp32 accumulation.
// One 128-edge tile per 256-thread workgroup, a wave's 32 rows on its lanes; the weight fragments of a 32-output slice travel by LDS-DMA through
// three 24-KB buffers (ef part: four slices; source part: the four half-size slices at once); 73 KB of LDS: two workgroups per CU.
#include <cstdio>

#include "gnx_edge_n.h"
#include "gnx_x6_stats.h"

namespace gnx {

typedef float f32x16n __attribute__((ext_vector_type(16)));
typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef int i32x4n __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8n __attribute__((ext_vector_type(8)));
typedef unsigned u32x4n __attribute__((ext_vector_type(4)));
typedef bf16x8n bf16x8r_alias;  // (bf16x8r of gnx_edge_n.h is the same type)

namespace {
constexpr int NR = 32, NW = 4, NBM = NR * NW;  // rows per wave, waves, rows per workgroup (= the edge tiles' row cap)
constexpr int NOUT = 128, NOB = NOUT / 32;     // outputs, 32-output slices
constexpr int NKE = 8, NKS = 4;                // k16-steps of the ef part (K = 128) and of the source part (K = 64)
constexpr int NSLE = NKE * 3 * 1024;           // bytes of a slice's ef-part fragments (24 KB)
constexpr int NSLS = NKS * 3 * 1024;           // ... source-part fragments (12 KB)
constexpr int NSL = NSLE + NSLS;               // a 32-output slice of the prepared weights: ef part, then source part (36 KB)

__device__ __forceinline__ unsigned ncvt2(float x0, float x1) {
  typedef float f2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 b2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f2_{x0, x1}, b2_));
}
__device__ __forceinline__ void nsplit2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = ncvt2(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = ncvt2(r0, r1);
  l = ncvt2(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
}
// eight values of a k16-step window -> the three bf16 fragments
__device__ __forceinline__ void nsplit8(const float (&v)[8], bf16x8n& zh, bf16x8n& zm, bf16x8n& zl) {
  unsigned ph[4], pm[4], pl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) nsplit2(v[2 * j], v[2 * j + 1], ph[j], pm[j], pl[j]);
  zh = __builtin_bit_cast(bf16x8n, u32x4n{ph[0], ph[1], ph[2], ph[3]});
  zm = __builtin_bit_cast(bf16x8n, u32x4n{pm[0], pm[1], pm[2], pm[3]});
  zl = __builtin_bit_cast(bf16x8n, u32x4n{pl[0], pl[1], pl[2], pl[3]});
}
}  // namespace

// W ([16 ks][ldw] row-major, its first 128 columns) -> per 32-output slice ob one block of 3 ks fragments of 1 KB = 64 lanes x 8 bf16:
//   fragment 3 s + p (s: k16-step, p: part), lane (m, h), j:  part_p( W[16 s + 8 h + j][32 ob + m] )         (k_edge_x6_prep's format, any K)
// Wp: the slice's block starts at byte ob * NSL + part_off
__global__ void k_edge_n_prep(const float* __restrict__ W, int ldw, int ks, __bf16* __restrict__ Wp, int part_off) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (ob, s, lane, j pair)
  if (idx >= NOB * ks * 64 * 4) return;
  const int jp = idx & 3, lane = (idx >> 2) & 63, s = (idx >> 8) % ks, ob = (idx >> 8) / ks;
  const int m = lane & 31, h = lane >> 5, k = 16 * s + 8 * h + 2 * jp;
  unsigned hh, mm, ll;
  nsplit2(W[(size_t)k * ldw + 32 * ob + m], W[(size_t)(k + 1) * ldw + 32 * ob + m], hh, mm, ll);
  unsigned* o = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(Wp) + (size_t)ob * NSL + part_off) + (size_t)(3 * s) * 256 + lane * 4 + jp;
  o[0] = hh; o[256] = mm; o[512] = ll;
}

struct EdgeNArgs {
  const Tile* tiles;
  const float* ef;         // [R][E][128]
  size_t E;
  const float* ln_stats;   // [R][E][2] (mean, 1/sigma) or nullptr
  const float* ln_g;
  const float* ln_b;
  int ln_inline;           // no ln_stats: the row statistics of gn1 are computed here, in registers (gnx_x6_stats.h), with ln_eps / ln_mode
  float ln_eps;
  int ln_mode;
  const __bf16* Wp;        // [4][36 KB]: per 32-output slice the ef-part fragments (24 KB), then the source-part fragments (12 KB)
  const float* zsrc;       // [R][N][64]: gn1(nf), or nf itself
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
};

// acc += rows x W over KS k16-steps: six terms per step, small terms first; the weight fragments of step s + 1 are requested in front of step s
template <int KS>
__device__ __forceinline__ void edge_n_mma(f32x16n& acc, const unsigned char* wb, const bf16x8n (&zh)[KS], const bf16x8n (&zm)[KS], const bf16x8n (&zl)[KS]) {
  bf16x8n W[2][3];
#pragma unroll
  for (int p3 = 0; p3 < 3; ++p3) W[0][p3] = *reinterpret_cast<const bf16x8n*>(wb + p3 * 1024);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int c = s & 1;
    if (s + 1 < KS) {
#pragma unroll
      for (int p3 = 0; p3 < 3; ++p3) W[c ^ 1][p3] = *reinterpret_cast<const bf16x8n*>(wb + (3 * (s + 1) + p3) * 1024);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zm[s], W[c][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], W[c][2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zl[s], W[c][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], W[c][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zm[s], W[c][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], W[c][0], acc, 0, 0, 0);
  }
}

// TRANS: the edge function's activation is tanh / sigmoid / gelu (the run-time switch of act_apply); else identity / relu
template <bool TRANS>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_edge_n(EdgeNArgs a) {
  // two OBJECTS: the compiler orders an LDS read behind every LDS-DMA that may alias it
  __shared__ __attribute__((aligned(16))) unsigned char s_wa[NSL];
  __shared__ __attribute__((aligned(16))) unsigned char s_wb[NSL];
  __shared__ __attribute__((aligned(16))) int s_dst[NBM];
  __shared__ float s_pin[2][NOB][32];   // per chunk: the second wave's share of the run that straddles the two waves
  __shared__ float s_pout[2][NOB][32];  // ... the first wave's
  __shared__ float s_tot[NW][NOB][32];  // per wave: column sums of its rows

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, n = lane & 31;
  const int tile_id = blockIdx.x;
  const size_t r = blockIdx.y;
  const Tile t = a.tiles[tile_id];
  const int row0 = t.e0, rows = t.e1 - t.e0;
  if (rows <= 0) return;  // (whole workgroup)

  auto stage = [&](int ob, unsigned char* dst) {
    const unsigned char* srcp = reinterpret_cast<const unsigned char*>(a.Wp) + (size_t)ob * NSL;
#pragma unroll
    for (int i = 0; i < NSL / 1024 / NW; ++i) {
      const int pc = wv + NW * i;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcp + (size_t)pc * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
    }
  };
  static_assert(NSL / 1024 % NW == 0, "a slice's fragments divide over the waves");
  stage(0, s_wa);
  stage(1, s_wb);
  if (tid < NBM) s_dst[tid] = a.dst[row0 + (tid < rows ? tid : rows - 1)];

  // ---- the wave's rows as A fragments, three bf16 parts: lane (n, hi) holds k = 16 s + 8 hi + j of row n — the ef row (gn1 on load) and,
  //      behind the lane's source index, the gathered 64-wide row of gn1(nf) ----
  const int lrow = wv * NR + n;
  const int lrc = lrow < rows ? lrow : rows - 1;
#ifdef GNX_EN_NO_GATHER  // (timing-only ablation builds: tools/build_variant.sh)
  const int gsrc = (row0 + lrc) % (int)a.N;
#else
  const int gsrc = a.src[row0 + lrc];
#endif
  const float* __restrict__ zrow = a.ef + (r * a.E + (size_t)row0 + lrc) * 128;
  bf16x8n zh[NKE], zm[NKE], zl[NKE], sh[NKS], sm[NKS], sl[NKS];
  {
    float mu = 0.f, inv = 1.f;
    const bool ln = a.ln_stats != nullptr || a.ln_inline != 0;
    if (a.ln_stats != nullptr) {
      const float2 st = reinterpret_cast<const float2*>(a.ln_stats)[r * a.E + (size_t)row0 + lrc];
      mu = st.x; inv = st.y;
    }
    f32x4s raw[NKE][2];
#pragma unroll
    for (int s = 0; s < NKE; ++s) {
      raw[s][0] = *reinterpret_cast<const f32x4s*>(zrow + 16 * s + 8 * hi);
      raw[s][1] = *reinterpret_cast<const f32x4s*>(zrow + 16 * s + 8 * hi + 4);
    }
    const float* __restrict__ srow = a.zsrc + (r * a.N + (size_t)gsrc) * 64;
    f32x4n rawS[NKS][2];
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      rawS[s][0] = *reinterpret_cast<const f32x4n*>(srow + 16 * s + 8 * hi);
      rawS[s][1] = *reinterpret_cast<const f32x4n*>(srow + 16 * s + 8 * hi + 4);
    }
    if (a.ln_inline != 0) x6_row_stats(raw, a.ln_eps, a.ln_mode, mu, inv);
#pragma unroll
    for (int s = 0; s < NKE; ++s) {
      float v[8] = {raw[s][0].x, raw[s][0].y, raw[s][0].z, raw[s][0].w, raw[s][1].x, raw[s][1].y, raw[s][1].z, raw[s][1].w};
      if (ln) {
        const f32x4n g0 = *reinterpret_cast<const f32x4n*>(a.ln_g + 16 * s + 8 * hi), g1 = *reinterpret_cast<const f32x4n*>(a.ln_g + 16 * s + 8 * hi + 4);
        const f32x4n b0 = *reinterpret_cast<const f32x4n*>(a.ln_b + 16 * s + 8 * hi), b1 = *reinterpret_cast<const f32x4n*>(a.ln_b + 16 * s + 8 * hi + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(gg[j], (v[j] - mu) * inv, bb[j]);
      }
      nsplit8(v, zh[s], zm[s], zl[s]);
    }
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const float v[8] = {rawS[s][0].x, rawS[s][0].y, rawS[s][0].z, rawS[s][0].w, rawS[s][1].x, rawS[s][1].y, rawS[s][1].z, rawS[s][1].w};
      nsplit8(v, sh[s], sm[s], sl[s]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of slices 0 and 1 have landed
  __syncthreads();                                  // ... everybody's; s_dst too

#ifdef GNX_EN_NO_SUMS
  const bool want_sums = false;
#else
  const bool want_sums = a.agg_out != nullptr || a.colsum != nullptr;
#endif
  const EdgeNRuns rn = edge_n_runs(s_dst, rows, wv, lane);

  // ---- per 32-output slice: K = 128 + 64 in one accumulator, then the register epilogue: lane (o, hi) holds rows 8 g + 4 hi + j of output 32 ob + o ----
  const int nvalid = min(max(rows - wv * NR, 0), NR);  // rows of this wave inside the tile
  const float* __restrict__ pd = a.pdst + r * a.N * NOUT;
  float* __restrict__ outp = a.out + (r * a.E + (size_t)row0 + wv * NR) * NOUT;
  const int pass = wv >> 1;
  float* agg_base = nullptr;
  if (a.agg_out) agg_base = a.agg_out + (r * a.n_agg_rows + (size_t)a.chunk_row0[2 * tile_id + pass]) * NOUT + n;
  // the destination addends of a slice — one dword per row, 128 contiguous bytes per row over the lanes — are what its accumulator STARTS from:
  // requested a slice ahead (under the previous slice's epilogue), they cost no register beside the accumulator
  auto addends = [&](int ob, f32x16n& acc) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const i32x4n d4 = *reinterpret_cast<const i32x4n*>(&s_dst[wv * NR + 8 * g + 4 * hi]);  // destinations of the lane's rows 8 g + 4 hi + (0..3)
      const int dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#ifdef GNX_EN_NO_PD
        acc[4 * g + j] = (float)dd[j];
#else
        acc[4 * g + j] = pd[(size_t)dd[j] * NOUT + 32 * ob + n];
#endif
      }
    }
  };
  // mid: between a slice's matrix instructions and its epilogue — the hand-over of the weight buffers (a workgroup barrier, the next request)
  auto slice = [&](int ob, const unsigned char* wbuf, f32x16n& acc, f32x16n& acc_next, auto mid) {
    __builtin_amdgcn_sched_barrier(0);
    edge_n_mma<NKE>(acc, wbuf + lane * 16, zh, zm, zl);
    edge_n_mma<NKS>(acc, wbuf + NSLE + lane * 16, sh, sm, sl);
    mid();
    if (ob + 1 < NOB) addends(ob + 1, acc_next);
    float part = 0.f;  // column sums: this half's rows in register order, then the two halves
    f32x16r S;
#pragma unroll
    for (int q = 0; q < 16; ++q) S[q] = 0.f;
    float* __restrict__ orow = outp + (size_t)(4 * hi) * NOUT + 32 * ob + n;  // the lane's row 4 hi, output 32 ob + n
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int q = 8 * t2 + j;
        float x = acc[q];
        if constexpr (TRANS) x = act_apply(x, a.act);
        else if (a.act == 1) x = relu_f(x);
        v[j] = x;
      }
      if (nvalid == NR) {  // (wave-uniform) every row of the wave lies inside the tile: sixteen plain dword stores, 128 contiguous bytes per row
#ifndef GNX_EN_NO_STORE
#pragma unroll
        for (int j = 0; j < 8; ++j) orow[(size_t)(8 * (2 * t2 + (j >> 2)) + (j & 3)) * NOUT] = v[j];
#endif
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int lr = 8 * (2 * t2 + (j >> 2)) + 4 * hi + (j & 3);
          if (lr < nvalid) orow[(size_t)(lr - 4 * hi) * NOUT] = v[j];
          else v[j] = 0.f;  // (rows beyond the tile: zero for the sums)
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) part += v[j];
      if (want_sums && a.agg_out) {
        bf16x8r vh, vm, vl;
        nsplit8(v, vh, vm, vl);
        edge_n_piece_sums_step(S, rn, t2, vh, vm, vl);
      }
    }
    if (want_sums) {  // (what the once-per-tile step below needs goes to LDS at once: nothing stays in registers across the slices)
      const float tt = part + __shfl_xor(part, 32);
      if (hi == 0) s_tot[wv][ob][n] = tt;
      if (a.agg_out) {
        edge_n_store_pieces(S, rn, agg_base + 32 * ob, NOUT, hi);
        if (wv & 1) { if (hi == 0) s_pin[pass][ob][n] = rn.first_in ? S[0] : 0.f; }  // piece 0 (lanes hi = 0)
        else if (hi == 1) s_pout[pass][ob][n] = S[15];                                // piece 31 (lanes hi = 1)
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  f32x16n accA, accB;
  addends(0, accA);
  slice(0, s_wa, accA, accB, [&] { __syncthreads(); stage(2, s_wa); });  // (every wave is done with buffer a)
  slice(1, s_wb, accB, accA, [&] { __syncthreads(); stage(3, s_wb); });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // slices 2 and 3 (this wave's pieces)
  __syncthreads();                                  // ... everybody's
  slice(2, s_wa, accA, accB, [] {});
  slice(3, s_wb, accB, accA, [] {});
  if (!want_sums) return;  // (uniform)
  // ---- once per tile: the run that straddles the two waves of a chunk, and the column sums over the four waves ----
  __syncthreads();
  if (a.agg_out && (wv & 1) == 0 && rn.last_out && rn.chunk_live && hi == 1) {
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) agg_base[(size_t)rn.idx_straddle * NOUT + 32 * ob] = s_pout[pass][ob][n] + s_pin[pass][ob][n];
  }
  if (a.colsum && tid < NOUT) {  // fixed order: the four waves ascending
    const int ob = tid >> 5, o = tid & 31;
    a.colsum[(r * a.n_tiles + (size_t)tile_id) * NOUT + tid] = ((s_tot[0][ob][o] + s_tot[1][ob][o]) + s_tot[2][ob][o]) + s_tot[3][ob][o];
  }
}

size_t edge_n_scratch_bytes() { return (size_t)NOB * NSL; }

bool edge_n_enabled() {
  // OPT-IN (GNX_FLAG_EDGE_N / env GNX_EDGE_N=1).  Measured on the MI355X at 1M edges (profiles/r05_edge_n_ab.log): 375 us against k_edge_x6's 324 — 243 us of it the rows, the split and
  // the 288 matrix instructions per wave (k_edge_x6: 192), the rest the dword-granular epilogue; its lower traffic floor (1.4 GB against 1.66) is not reached.
  return form(GNX_FLAG_EDGE_N);
}

// We ([128 + 64 ..][ldw]: the ef rows, then the source rows) -> scratch: the ef-part fragments (4 x 24 KB), then the source-part fragments (4 x 12 KB)
int32_t launch_edge_n_prep(const float* We, int ldw, void* scratch, hipStream_t s) {
  ProfScope ps("k_edge_x6_prep", s);  // (the name the edge update's weight preparation has in every profile)
  __bf16* Wp = static_cast<__bf16*>(scratch);
  GNX_LAUNCH(k_edge_n_prep, dim3((unsigned)((NOB * NKE * 64 * 4 + 255) / 256)), dim3(256), 0, s, We, ldw, NKE, Wp, 0);
  GNX_LAUNCH(k_edge_n_prep, dim3((unsigned)((NOB * NKS * 64 * 4 + 255) / 256)), dim3(256), 0, s, We + (size_t)128 * ldw, ldw, NKS, Wp, NSLE);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

int32_t launch_edge_n(const Tile* tiles, size_t n_tiles, const float* ef, size_t E, const float* ln_stats, const float* ln_g, const float* ln_b, const float* We, int ldw,
                      const float* zsrc, const float* pdst, size_t N, const int* src, const int* dst, int act, float* out, float* colsum, float* agg_out, size_t n_agg_rows,
                      const int* chunk_row0, int64_t R, void* scratch, hipStream_t s, bool ln_inline, float ln_eps, int ln_mode) {
  if (n_tiles == 0) return GNX_OK;
  if (!tiles || !ef || !We || !zsrc || !pdst || !src || !dst || !out || !scratch) return fail(GNX_ERR_INVALID_ARG, "k_edge_n: NULL operand");
  if ((((uintptr_t)ef | (uintptr_t)zsrc | (uintptr_t)scratch | (uintptr_t)ln_g | (uintptr_t)ln_b) & 15) || ((uintptr_t)ln_stats & 7))
    return fail(GNX_ERR_INVALID_ARG, "k_edge_n: operand not 16-byte aligned");
  if (agg_out && !chunk_row0) return fail(GNX_ERR_INVALID_ARG, "k_edge_n: per-destination sums need the chunk table");
  if (const int32_t rc = launch_edge_n_prep(We, ldw, scratch, s)) return rc;
  EdgeNArgs a{};
  a.tiles = tiles; a.ef = ef; a.E = E; a.ln_stats = ln_stats; a.ln_g = ln_g; a.ln_b = ln_b;
  if (ln_inline) {
    if (ln_stats || !ln_g || !ln_b) return fail(GNX_ERR_INVALID_ARG, "k_edge_n: statistics in the kernel exclude a statistics table and need gamma / beta");
    a.ln_inline = 1; a.ln_eps = ln_eps; a.ln_mode = ln_mode;
  } else if (ln_stats && (!ln_g || !ln_b)) return fail(GNX_ERR_INVALID_ARG, "k_edge_n: LayerNorm parameters missing");
  a.Wp = static_cast<const __bf16*>(scratch);
  a.zsrc = zsrc; a.pdst = pdst; a.N = N; a.src = src; a.dst = dst; a.act = act; a.out = out; a.colsum = colsum; a.n_tiles = n_tiles;
  a.agg_out = agg_out; a.n_agg_rows = n_agg_rows; a.chunk_row0 = chunk_row0;
  ProfScope ps("k_rows_gemm_edge", s);  // (the name the edge update has in every profile and bench line)
  if (act > GNX_ACT_RELU) GNX_LAUNCH(k_edge_n<true>, dim3((unsigned)n_tiles, (unsigned)R), dim3(64 * NW), 0, s, a);
  else GNX_LAUNCH(k_edge_n<false>, dim3((unsigned)n_tiles, (unsigned)R), dim3(64 * NW), 0, s, a);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

}  // namespace gnx

// EXPERIMENT (round 4, measured and rejected; not part of the library): the encoder's DIRECT edge update on the vector units.  Built into the library
// (csrc/ + a branch in launch_block_wide in front of the k_rows_gemm edge launch; 195 GPU tests green, five dedicated cases against the oracle and against
// k_rows_gemm) it took 205-209 us for config 4's encoder on C2 against 194 us for k_rows_gemm<128,true,32,0,false,2>; block 267-271 vs 256-257 us/step.
// Its 640 packed FMAs per wave and tile are 37 us of vector time over the launch; the rest is the serial chain of a tile (indices -> gathers -> LDS ->
// barrier -> registers -> four slices of [reads, FMAs, stores, barrier, sums, barrier]) with three workgroups per CU to hide it — removing the staging
// loops' divisions, one thread per gathered node row and weight reads four steps ahead changed nothing (208 us).  What it would need: a persistent
// workgroup that requests tile i + 1's rows (indices two tiles ahead) while it computes tile i.
// The DIRECT edge update of a wide block with NARROW inputs (edgefninput.jl:2-7, gnblock.jl:57-60) — config 4's encoder, (10, 5, 0) => (128, 64, 32):
//
//     ef'[e] = act( We^T [ ef[e] | nf[src(e)] | nf[dst(e)] ] + b (+ gf fold per graph) )          K = de + 2 dn <= 24 inputs, oe = 32 .. 128 outputs
//
// 2 K oe FLOP per edge are a few percent of what the vector units deliver while the kernel streams its 4 oe output bytes per edge: no matrix
// instruction is needed — and none of k_rows_gemm's machinery around one (K chunks through LDS, operand streams, a 128 x 128 accumulator tile): that kernel
// takes 207 us for the 552 MB of config 4's encoder.  Here a 128-edge tile of the handle's edge-tile table is assembled ONCE in LDS ([row][K]: the ef rows
// are one contiguous piece, the node rows L2 hits), every thread keeps the inputs of ITS four rows in registers (thread (er, eq) of wave wv: rows
// 32 wv + er + 8 i, the 16-byte quad eq of each 32-output slice — k_edge_x6's epilogue mapping, so an instruction stores 8 rows x 128 contiguous bytes),
// and a slice is K broadcast reads of a weight quad with 4 x 4 FMAs each, in fp32, in the order k = 0 .. K - 1.  The finished slice is staged for the
// per-destination sums (edges are dst-sorted: one partial row per destination run and 64-row chunk, the node update's first segment) and the column
// sums (graph update), both in a fixed order — the code of k_edge_x6.
#include <cstdio>

#include "gnx_device.h"

namespace gnx {

typedef float f32x4v __attribute__((ext_vector_type(4)));

typedef float f32x2v __attribute__((ext_vector_type(2)));

namespace {
// acc += w * (one element of xpair, broadcast to both halves): ONE v_pk_fma_f32 — two outputs of a row per instruction.  (Written as fmaf the compiler
// also picks the packed FMA, but materialises {x, x} pairs for it: twice the registers for the rows' inputs, 120 spilled.)
__device__ __forceinline__ void pk_fma_bcast(bool hi, f32x2v& acc, f32x2v w, f32x2v xpair) {  // (hi: a constant after unrolling)
  if (!hi) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(xpair));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(xpair));
}
constexpr int VR = 32, VW = 4, VBM = VR * VW;  // rows per wave, waves, rows per workgroup (= the edge tiles' row cap)
constexpr int VLDE = 36;                       // floats per staged row (32 + 4)
constexpr int VOMAX = 128;                     // widest output (46 KB of LDS at K = 20: three workgroups per CU)
}  // namespace

struct EdgeValuArgs {
  const Tile* tiles;
  const float* ef;         // [R][E][de]
  size_t E;
  int de;
  const float* nf;         // [R][N][dn]
  size_t N;
  int dn;
  const float* W;          // [de + 2 dn][oe] row-major (the Dense weight, (oe x K) column-major)
  int oe;
  const float* bias;       // [oe] or nullptr
  const float* bias_g;     // [R][G][oe] (bias + gf fold) or nullptr
  int G;
  const int* src;          // rowval [E]
  const int* dst;          // edge_dst [E]
  int act;
  float* out;              // [R][E][oe]
  float* colsum;           // [R][n_tiles][oe] or nullptr
  size_t n_tiles;
  float* agg_out;          // [R][n_agg_rows][oe] or nullptr
  size_t n_agg_rows;
  const int* chunk_row0;   // [2 n_tiles + 1]
};

// KP: K rounded up to a multiple of 4 (the padding columns of the tile are zero)
template <int KP>
__global__ __launch_bounds__(64 * VW) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_edge_valu(EdgeValuArgs a) {  // (<= 168 registers: three workgroups per CU, as the LDS allows)
  __shared__ __attribute__((aligned(16))) float s_x[VBM * KP];      // the tile's input rows
  __shared__ __attribute__((aligned(16))) float s_w[KP * VOMAX];    // the weight block, [k][oe]
  __shared__ __attribute__((aligned(16))) float s_e[VBM * VLDE];    // the finished 32-column slice of the tile, [row][36]
  __shared__ __attribute__((aligned(16))) float s_cs[32 * 32];      // column-sum partials [row group][column]
  __shared__ int s_dst[VBM];
  __shared__ int s_seg[2][66];  // per 64-row pass: first row of every destination run; [n_seg] = valid rows of the pass; [65] = n_seg

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_id = blockIdx.x;
  const size_t r = blockIdx.y;
  const Tile t = a.tiles[tile_id];
  const int row0 = t.e0, rows = t.e1 - t.e0;
  if (rows <= 0) return;  // (whole workgroup)
  const int de = a.de, dn = a.dn, K = de + 2 * dn, oe = a.oe;
  int agg_row0[2] = {0, 0};
  if (a.agg_out) { agg_row0[0] = a.chunk_row0[2 * tile_id]; agg_row0[1] = a.chunk_row0[2 * tile_id + 1]; }

  // ---- the tile's rows [ef | nf[src] | nf[dst] | 0 ..] and the weight block into LDS ----
  {
    const float* __restrict__ efp = a.ef + (r * a.E + (size_t)row0) * de;  // rows * de contiguous floats
    {  // (row, k) of element tid, then + 256 elements per round without a division
      int row = tid / de, k = tid - row * de;
      const int drow = (64 * VW) / de, dk = 64 * VW - drow * de;
      for (int i = tid; i < rows * de; i += 64 * VW) {
        s_x[row * KP + k] = efp[i];
        row += drow; k += dk;
        if (k >= de) { k -= de; ++row; }
      }
    }
    // the node rows: thread (row = tid / 2, side = tid % 2) copies nf[src] or nf[dst] of its row — the index load first, every element behind it
    const float* __restrict__ nfp = a.nf + r * a.N * dn;
    {
      const int row = tid >> 1, side = tid & 1;
      if (row < rows) {
        const int node = side ? a.dst[row0 + row] : a.src[row0 + row];
        const float* __restrict__ np = nfp + (size_t)node * dn;
        float* xp = s_x + row * KP + de + side * dn;
        for (int k = 0; k < dn; ++k) xp[k] = np[k];
      }
    }
    for (int i = tid; i < rows * (KP - K); i += 64 * VW) {
      const int row = i / (KP - K), k = i - row * (KP - K);
      s_x[row * KP + K + k] = 0.f;
    }
    for (int i = tid; i < (VBM - rows) * KP; i += 64 * VW) s_x[rows * KP + i] = 0.f;  // (rows beyond the tile)
    for (int i = tid; i < KP * oe / 4; i += 64 * VW)  // (rows K .. KP - 1: zero, as the padding columns of the tile)
      reinterpret_cast<f32x4v*>(s_w)[i] = i < K * oe / 4 ? reinterpret_cast<const f32x4v*>(a.W)[i] : f32x4v{0.f, 0.f, 0.f, 0.f};
    if (tid < VBM) s_dst[tid] = a.dst[row0 + (tid < rows ? tid : rows - 1)];
  }
  __syncthreads();
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

  const int er = lane >> 3, eq = lane & 7;  // (row er + 8 i of the wave's 32, 16-byte quad eq of a 32-column slice)
  f32x2v x[4][KP / 2];                      // the inputs of this thread's four rows, as register pairs (k, k + 1)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k4 = 0; k4 < KP / 4; ++k4) {
      const f32x4v q = *reinterpret_cast<const f32x4v*>(s_x + (wv * VR + er + 8 * i) * KP + 4 * k4);
      x[i][2 * k4] = f32x2v{q.x, q.y}; x[i][2 * k4 + 1] = f32x2v{q.z, q.w};
    }
  float* sE = s_e + wv * (VR * VLDE);
  float* __restrict__ outp = a.out + (r * a.E + (size_t)row0) * oe;
  const float* __restrict__ bp = a.bias_g ? a.bias_g + (r * a.G + (size_t)t.g) * oe : a.bias;

  for (int ob = 0; ob < oe / 32; ++ob) {
    f32x2v acc[4][2];  // [row][output pair]
    const f32x4v b4 = bp ? *reinterpret_cast<const f32x4v*>(bp + 32 * ob + 4 * eq) : f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i][0] = f32x2v{b4.x, b4.y}; acc[i][1] = f32x2v{b4.z, b4.w}; }
    const float* wq = s_w + 32 * ob + 4 * eq;
    // Weight quads four steps ahead of their FMAs, and no further: `tok` is 0, opaque to the compiler and made to depend on each step's last FMA, and
    // the address of the read four steps later hangs on it (left alone the compiler requests all K quads of a slice at once — 80 registers beside
    // the 80 of x — and spills; neither a scheduling barrier nor a memory clobber holds LDS reads of a non-escaping array in place).
    int tok = 0;
    constexpr int WD = 4;  // weight quads in flight
    f32x4v wb[WD + 1];
#pragma unroll
    for (int k = 0; k < WD; ++k) wb[k] = *reinterpret_cast<const f32x4v*>(wq + k * oe);  // (one address for the 8 lanes of a quad column: a broadcast read)
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      if (k + WD < KP) wb[(k + WD) % (WD + 1)] = *reinterpret_cast<const f32x4v*>(wq + tok + (k + WD) * oe);
      const f32x4v w = wb[k % (WD + 1)];
      const f32x2v w01 = {w.x, w.y}, w23 = {w.z, w.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        pk_fma_bcast(k % 2 != 0, acc[i][0], w01, x[i][k / 2]);
        pk_fma_bcast(k % 2 != 0, acc[i][1], w23, x[i][k / 2]);
      }
      asm volatile("" : "+v"(tok) : "v"(acc[3][1].y));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int lr = er + 8 * i;
      float vv[4] = {acc[i][0].x, acc[i][0].y, acc[i][1].x, acc[i][1].y};
      if (a.act == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = relu_f(vv[e]);
      } else if (a.act > 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = act_apply(vv[e], a.act);
      }
      const bool ok = wv * VR + lr < rows;
      f32x4v v = {vv[0], vv[1], vv[2], vv[3]};
      if (!ok) v = f32x4v{0.f, 0.f, 0.f, 0.f};  // (rows beyond the tile: zero for the sums below)
      *reinterpret_cast<f32x4v*>(sE + lr * VLDE + 4 * eq) = v;
      if (ok) *reinterpret_cast<f32x4v*>(outp + (size_t)(wv * VR + lr) * oe + 32 * ob + 4 * eq) = v;
    }
    if (a.agg_out == nullptr && a.colsum == nullptr) continue;  // (uniform) nothing reads the staged slice
    __syncthreads();  // the finished slice of all four waves is in s_e
    const int q4 = tid & 7, grp = tid >> 3;  // 8 quads x 32 row groups
    f32x4v c4 = {0.f, 0.f, 0.f, 0.f};        // this thread's share of the tile's column sums
    if (a.agg_out) {
      // per-destination sums: groups 0-15 take the runs of pass 0, groups 16-31 those of pass 1 (16 runs per sweep), four rows of a run requested at a
      // time.  Every valid row lies in exactly one run: the column sums are the sums of the run sums.
      const int pass = grp >> 4, g16 = grp & 15;
      const int n_seg = s_seg[pass][65];
      float* agg = a.agg_out + (r * a.n_agg_rows + (size_t)agg_row0[pass]) * oe + 32 * ob + 4 * q4;
      const float* base = s_e + 64 * pass * VLDE + 4 * q4;
      for (int sgm = g16; sgm < n_seg; sgm += 16) {
        const int r0 = s_seg[pass][sgm], r1 = s_seg[pass][sgm + 1];
        f32x4v t4 = {0.f, 0.f, 0.f, 0.f};
        for (int rr = r0; rr < r1; rr += 4) {
          f32x4v u[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) u[j] = *reinterpret_cast<const f32x4v*>(base + min(rr + j, r1 - 1) * VLDE);
#pragma unroll
          for (int j = 0; j < 4; ++j) { if (rr + j < r1) t4 += u[j]; }
        }
        *reinterpret_cast<f32x4v*>(agg + (size_t)sgm * oe) = t4;
        c4 += t4;
      }
    } else {  // (no fused aggregation: the rows themselves, grp, grp + 32, .. ascending)
#pragma unroll
      for (int i = 0; i < 4; ++i) c4 += *reinterpret_cast<const f32x4v*>(s_e + (grp + 32 * i) * VLDE + 4 * q4);
    }
    if (a.colsum) *reinterpret_cast<f32x4v*>(s_cs + grp * 32 + 4 * q4) = c4;
    __syncthreads();  // s_e may be overwritten by the next slice; the column-sum partials are complete
    if (a.colsum && tid < 32) {  // fixed order: the 32 groups ascending
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 32; ++w) sum += s_cs[w * 32 + tid];
      a.colsum[(r * a.n_tiles + (size_t)tile_id) * oe + 32 * ob + tid] = sum;
    }
    // (the next slice writes s_cs only behind its first barrier: the 32 readers above are past it by then)
  }
}

bool edge_valu_applies(int de, int dn, int oe, const float* ef, const float* W, const float* out, const float* bias, size_t E) {
  const int K = de + 2 * dn;
  if (getenv("GNX_EDGE_DIRECT_MFMA") != nullptr) return false;  // (read per call: k_rows_gemm instead)
  if (de < 1 || dn < 1 || K > 24 || oe < 32 || oe > VOMAX || oe % 32 != 0 || E < 4096) return false;
  return (((uintptr_t)W | (uintptr_t)out | (uintptr_t)bias) & 15) == 0 && ((uintptr_t)ef & 3) == 0;
}

int32_t launch_edge_valu(const Tile* tiles, size_t n_tiles, const float* ef, size_t E, int de, const float* nf, size_t N, int dn, const float* W, int oe, const float* bias,
                         const float* bias_g, int G, const int* src, const int* dst, int act, float* out, float* colsum, float* agg_out, size_t n_agg_rows,
                         const int* chunk_row0, int64_t R, hipStream_t s) {
  if (n_tiles == 0) return GNX_OK;
  if (!tiles || !ef || !nf || !W || !src || !dst || !out) return fail(GNX_ERR_INVALID_ARG, "k_edge_valu: NULL operand");
  if (!edge_valu_applies(de, dn, oe, ef, W, out, bias, E) || (((uintptr_t)bias_g | (uintptr_t)agg_out) & 15)) return fail(GNX_ERR_INVALID_ARG, "k_edge_valu: widths or alignment outside the kernel's form");
  EdgeValuArgs a{};
  a.tiles = tiles; a.ef = ef; a.E = E; a.de = de; a.nf = nf; a.N = N; a.dn = dn; a.W = W; a.oe = oe; a.bias = bias; a.bias_g = bias_g; a.G = G; a.src = src; a.dst = dst;
  a.act = act; a.out = out; a.colsum = colsum; a.n_tiles = n_tiles; a.agg_out = agg_out; a.n_agg_rows = n_agg_rows; a.chunk_row0 = chunk_row0;
  ProfScope ps("k_rows_gemm_edge", s);  // (the name the edge update has in every profile and bench line)
  const dim3 grid((unsigned)n_tiles, (unsigned)R), block(64 * VW);
  const int kp = (de + 2 * dn + 3) / 4 * 4;
  switch (kp) {
    case 4: GNX_LAUNCH(k_edge_valu<4>, grid, block, 0, s, a); break;
    case 8: GNX_LAUNCH(k_edge_valu<8>, grid, block, 0, s, a); break;
    case 12: GNX_LAUNCH(k_edge_valu<12>, grid, block, 0, s, a); break;
    case 16: GNX_LAUNCH(k_edge_valu<16>, grid, block, 0, s, a); break;
    case 20: GNX_LAUNCH(k_edge_valu<20>, grid, block, 0, s, a); break;
    default: GNX_LAUNCH(k_edge_valu<24>, grid, block, 0, s, a); break;
  }
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

}  // namespace gnx

// Wide-feature path: the per-edge and per-node Dense updates as gathered-row GEMMs on the fp32 matrix cores.
//
//   out[m, :] = act( sum over segments  A_seg[row_seg(m), :] * W[seg rows, :]  +  bias'[g] )
//
// edge update (edgefninput.jl:1-8 + gnblock.jl:65):  segments  ef[m] | nf[src(m)] | nf[dst(m)]          (gf folded)
// node update (nodefninput.jl:1-7 + gnblock.jl:66):  segments  sum_{e->m} ef'[e] | nf[m]                (gf folded)
// The one-hot batched matmuls of the reference become row indices; the `vcat` never exists — every segment is just a
// range of K-chunks of the same GEMM; gf[g] (constant per graph, and a tile never crosses a graph) is folded into the
// tile's bias: bias'[g] = b + W[gf rows]^T gf[g].
//
// MI355X mapping: v_mfma_f32_32x32x2_f32 (exact fp32 — the 1e-5 bar rules out bf16/fp16, and gfx950 has no xf32),
// 256-thread workgroup = 4 waves, tile 128 rows x BN cols, K streamed in 32-wide chunks through LDS (A padded to a
// 33-float row stride: conflict-free ds_read_b32 for the 32x32x2 A fragment; B = W rows as stored by Flux), next chunk
// prefetched into registers while the current one is on the matrix cores.  Epilogue: bias' + activation, 128-B row
// segments to HBM, and fixed-order column sums of the tile (the graph update's partial sums, graphfninput.jl:3-4).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "gnx_device.h"
#include "gnx_x6_mma.h"

#ifndef GNX_LN_GUARD  // see store_chunk's LayerNorm branch
#define GNX_LN_GUARD 1
#endif

namespace gnx {

int32_t launch_edge_x6(const Tile* tiles, size_t n_tiles, const float* ef, size_t E, const float* ln_stats, const float* ln_g, const float* ln_b, const float* We, int ldw,
                       const float* psrc, const float* pdst, size_t N, const int* src, const int* dst, int act, float* out, float* colsum, float* agg_out,
                       size_t n_agg_rows, const int* chunk_row0, int64_t R, void* scratch, hipStream_t s, bool ln_inline = false, float ln_eps = 0.f,
                       int ln_mode = 0, int oe = 128);  // gnx_edge_x6.hip
size_t proj_x6_scratch_bytes();  // gnx_edge_x6.hip
bool proj_x6_applies(int dn, int oe, const float* nf, const float* W, const float* out, size_t N);
int32_t launch_proj_x6(const Tile* tiles, size_t n_tiles, const float* nf, size_t N, const float* ln_stats, const float* ln_g, const float* ln_b, const float* Ws, const float* Wd,
                       int ldw, const float* bias, const float* bias_g, int G, float* out_s, float* out_d, int64_t R, void* scratch, hipStream_t s, bool only_d = false,
                       float* zn_out = nullptr);
// the encoder form of k_edge_x6 ((10, 5, .) => 128 unprojected; gnx_edge_x6.hip)
size_t edge_enc_scratch_bytes();
int32_t launch_edge_enc(const Tile* tiles, size_t n_tiles, const float* ef, size_t E, const float* nf, size_t N, const float* We, int ldw, const float* bias, const float* bias_g,
                        int G, const int* src, const int* dst, int act, float* out, float* colsum, float* agg_out, size_t n_agg_rows, const int* chunk_row0, int64_t R,
                        void* scratch, hipStream_t s);
// gnx_edge_n.hip: the edge update with the source side gathered raw (K = 128 + 64) and a register epilogue
size_t edge_n_scratch_bytes();
size_t edge_x6_fold_scratch_bytes();  // gnx_edge_x6.hip
bool node_x6_applies(int oe, int dn, int on, int act, const float* nf, const float* Wn, const float* out, size_t N);  // gnx_edge_x6.hip
size_t node_x6_scratch_bytes();
int32_t launch_node_x6(const Tile* tiles, size_t n_tiles, const float* nf, size_t N, const float* ln_stats, const float* ln_g, const float* ln_b, const float* agg,
                       size_t n_agg_rows, const int* agg_row, const int* agg_parts, const int* agg_chunk, const int* chunk_row0, const float* Wn, int ldw, const float* bias,
                       const float* bias_g, int G, int act, float* out, float* colsum, int64_t R, void* scratch, hipStream_t s);
bool edge_n_enabled();
int32_t launch_edge_n(const Tile* tiles, size_t n_tiles, const float* ef, size_t E, const float* ln_stats, const float* ln_g, const float* ln_b, const float* We, int ldw,
                      const float* zsrc, const float* pdst, size_t N, const int* src, const int* dst, int act, float* out, float* colsum, float* agg_out, size_t n_agg_rows,
                      const int* chunk_row0, int64_t R, void* scratch, hipStream_t s, bool ln_inline, float ln_eps, int ln_mode);
int32_t launch_core_edge_x6(const Tile* tiles, size_t n_tiles, const float* x, size_t E, const gnx_layernorm* ln1, float ln_eps, int ln_mode, const float* We, int ldw,
                            const float* psrc, const float* pdst, size_t N, const int* src, const int* dst, int act, float* colsum, float* agg_out, size_t n_agg_rows,
                            const int* chunk_row0, const gnx_ffn& ff, const gnx_layernorm* ln2, float* out, int64_t R, void* scratch_e, void* scratch_f, hipStream_t s);  // gnx_ffn_x6.hip

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef GNX_GEMM_WPE  // 3 waves per SIMD = 3 workgroups per CU with a 168-register budget: NO spilled register at BN = 128.  A 4th workgroup
// (128 registers, ~20 spilled) is slower once the loads are pipelined: a scratch reload is a vector-memory operation, so the wait
// in front of its first use also waits for every global load issued before it (edge GEMM 433 vs 462 us).
#define GNX_GEMM_WPE 3
#endif
#ifndef GNX_GEMM_GRP  // row quads per epilogue operand group (two groups in flight)
#define GNX_GEMM_GRP 2
#endif
constexpr int BM = 128;   // rows (edges or nodes) per workgroup tile

struct WSeg {
  const float* base;  // replica 0
  size_t rep_stride;  // floats between replicas
  int width;          // K extent of this segment
  int mode;           // 0: row m itself, 1: row idx_a[m], 2: row idx_b[m], 3: sum of rows [cp[m], cp[m+1]) of base,
                      // 4: the node's in-edge sum from the partial-sum table the edge GEMM wrote (agg_* below),
                      // 5: the narrow segments WideArgs::pk side by side in ONE K range (set by launch_gemm)
  int w_row0;         // first row of W (= first input feature index) of this segment
  int vec;            // rows can be read as 16-B quads (width % 4 == 0, 16-B aligned base): set by launch_gemm
  int ln;             // LayerNorm on load (WideArgs::ln_*): mode 0, quad rows, width <= 256
};

struct WideArgs {
  const Tile* tiles;
  int row_kind;  // 0: rows = edges [e0, e1) of the tile, 1: rows = nodes [n0, n1)
  WSeg seg[3];
  int nseg;
  const int* idx_a;  // rowval  (global source node of every edge)
  const int* idx_b;  // edge_dst (global destination node of every edge)
  const int* cp;     // colptr
  const float* W;    // (OUT x K) column-major == [K][OUT] row-major
  int ldw;           // distance between rows of W (0: OUT) — lets a launch multiply by a column block of a wider matrix
  const float* bias;
  int OUT, act;
  const float* bias_g;  // [R][G][OUT] per-graph bias with gf folded in (k_fold_bias), or nullptr: use `bias`
  int n_graphs;
  float* out;
  size_t out_rep_stride;
  float* colsum;     // [R][n_tiles][OUT] or nullptr
  size_t colsum_rep_stride;
  // node-projection form of the edge update: out = act(acc + bias' + gadd_a[idx_a[m]] + gadd_b[idx_b[m]]), tables [R][rows][OUT]
  const float* gadd_a;
  const float* gadd_b;
  size_t gadd_rep_stride;
  const float* add1;  // optional residual inputs with the layout of `out` (may alias `out`): v = act(..) + add1 + add2
  const float* add2;
  const float* gmul;  // optional, layout of `out`: v *= act'(gmul) (derivative through the stored output, act = gmul_act) BEFORE the
  int gmul_act;       // column sums — the backward's  delta = (g W^T) .* act'(h)  in the GEMM epilogue
  // edge -> node aggregation fused into the edge GEMM (nodefninput.jl:3 without re-reading ef'): every 64-row pass of an edge tile
  // ("chunk" 2*tile + pass) adds up the rows of each destination it holds (edges are dst-sorted: contiguous runs) and writes one
  // row per destination to agg_out[chunk_row0[chunk] + k].  The node GEMM's mode-4 loader reads node m's sum as row
  // node_agg_row[m] (+ the first row of each further chunk its in-edges run into; -1: no in-edges).
  float* agg_out;              // edge launch: [R][n_agg_rows][OUT], or nullptr
  size_t agg_rep_stride;
  const int* chunk_row0;       // [2 * n_etiles + 1]
  const int* node_agg_row;     // mode 4: [N]
  const int* node_agg_parts;   //         [N]
  const int* node_agg_chunk;   //         [N]
  // a SECOND weight block / bias / output of the same shape over the same rows (the src and dst node projections): its column tiles
  // run next to the first one's on the same XCD, the A tile is fetched once
  const float* W2;
  const float* bias2;
  const float* bias_g2;
  float* out2;
  int n_rtiles, n_ctiles;      // row tiles / column tiles of this launch (set by launch_gemm; W2: both halves)
  int epi;                     // EPI_* (set by launch_gemm)
  // LayerNorm applied to the rows of ONE mode-0 segment as they are loaded (GNCore's gn1 / gn2 never materialised, gncore.jl:56-59):
  // (x - mean) * inv, then fma(gamma, ., beta) — the arithmetic of k_layernorm2_v4, statistics from k_ln_stats_v4
  const float* ln_stats;       // [R][rows of the entity][2] (mean, inv), or nullptr
  size_t ln_rep_stride;        // floats between replicas of ln_stats
  const float* ln_g;           // [width of the segment]
  const float* ln_b;
  int ln_width;                // (set by launch_gemm)
  WSeg pk[3];                  // mode 5: the packed segments (modes 0-2), their W rows consecutive from seg[0].w_row0
  int npk;
  int pd_lds;                  // gathered addends: every tile's destination rows fit the LDS table of the NL = 3 kernel (set by launch_block_wide from the handle's tile statistics)
  int stagger;                 // start delay per residency slot (units of 64*127 clocks), 0 = none (set by launch_gemm)
  int fp32;                    // 1: the products on v_mfma_f32_32x32x2f32 — only where the CALL's flags ask (GNX_FLAG_EDGE_FP32 / _PROJ_FP32 / _EDGE_NARROW_FP32 /
                               // _FFN_FP32: set by the callers from form()); 0, the default: six bf16 matrix-core terms per product (gnx_x6_mma.h)
  unsigned long long* stamps;  // diagnostic builds only (GNX_WIDE_STAMPS): [tile][8] shader-clock stamps of wave 0
};

// what the VEC4 epilogue reads from global memory besides the accumulators (WideArgs::epi, set by launch_gemm from the pointers)
enum { EPI_NONE = 0, EPI_GADD = 1, EPI_ADD1 = 2, EPI_ADD12 = 3, EPI_GMUL = 4, EPI_GMUL_ADD = 5 };

// activation of N registers: ONE wave-uniform switch with the loop inside each case (a switch per element puts a branch — and a
// wait for everything in flight — between the elements)
template <int N>
__device__ __forceinline__ void act_apply_n(float* v, int act) {
  switch (act) {
    case 0: break;
    case 1:
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = relu_f(v[j]);
      break;
    case 2:
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = act_apply(v[j], 2);
      break;
    case 3:
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = act_apply(v[j], 3);
      break;
    default:
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = act_apply(v[j], 4);
      break;
  }
}

// 16-B access at a wave-uniform base plus a 32-bit element offset: one VGPR of address instead of a 64-bit pair per access
// (global_load_dwordx4 v, v_off, s[base]).  The hosts keep every table addressed this way below 4 GB (launch_gemm).
// a pointer the compiler may keep in scalar registers (wave-uniform by construction, e.g. selected by the block index)
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const float*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float4 ld4(const float* ubase, unsigned off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(ubase) + (off << 2));
}
__device__ __forceinline__ float ld1(const float* ubase, unsigned off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ubase) + (off << 2));
}
// The same load kept out of the compiler's s_waitcnt bookkeeping (pass 1's source rows of the NL = 3 epilogue): with loads and
// stores pending on one counter the compiler waits for ALL of them at the load's first use — every row store of pass 0 acknowledged
// by memory — where the hardware retires the counter in issue order and a counted wait (all but the stores issued since) is enough.
// The destination counts as written at the statement, so it must not be read, copied or spilled before asm_wait_rows (checked in
// the .s: no v_mov / scratch access to these registers in between).  s_nop 4: SALU-written base -> VMEM read of it.
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ld4_untracked(v4f& dst, const float* ubase, unsigned off) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off << 2), "s"(ubase) : "memory");
}
__device__ __forceinline__ void st4(float* ubase, unsigned off, float4 v) {
  *reinterpret_cast<float4*>(reinterpret_cast<char*>(ubase) + (off << 2)) = v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits until every global
// STORE of the wave has been acknowledged — in the epilogue that is a full HBM write round trip per barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifndef GNX_GEMM_W8  // 1: the 128-column tile is computed by 8 waves (32 x 64 per wave, two 8-wave workgroups per CU) instead of 4 (64 x 64, three)
#define GNX_GEMM_W8 0
#endif
template <int BN>
struct WaveLayout {
  static constexpr int WAVES = (BN == 128 && GNX_GEMM_W8) ? 8 : 4;
  static constexpr int WM = (BN == 32 || WAVES == 8) ? 4 : 2;
  static constexpr int WN = WAVES / WM;
  static constexpr int TM = BM / (WM * 32);
  static constexpr int TN = BN / (WN * 32);
  static constexpr int WT = 64 * WAVES;         // threads of the workgroup
#ifndef GNX_GEMM_WPE64  // 64-column tiles (the node update at C4's widths): 4 workgroups per CU — 782 node tiles are ONE round on 1024 slots but
#define GNX_GEMM_WPE64 4  // 1.02 rounds on 768; the ~25 registers the row-sum loader then spills cost less than the second round (66 -> 60 us)
#endif
  static constexpr int WPE = WAVES == 8 ? 4 : (BN == 64 ? GNX_GEMM_WPE64 : GNX_GEMM_WPE);  // waves per SIMD asked of the register allocator
};

// NL = number of epilogue operand streams read from global memory (EPI_* below): 0, 1 or 2 float4 per output quad; 3: the gathered
// addends of the node-projection form with the DESTINATION rows staged in LDS (edges are dst-sorted: a tile's destinations are a
// short run of consecutive rows of the table, ~14 on the 1M-edge graph, fetched ONCE per tile by LDS-DMA in the prologue instead of
// once per edge) and the SOURCE rows of a whole 64-row pass requested at once — pass 0's behind the last chunk's loads into the
// chunk staging registers, pass 1's into the accumulator registers pass 0 has just handed to LDS — so the epilogue waits for one
// covered round trip instead of four serial ones (two streams, two groups of two row quads in flight: by the stamps 19.5 k of a
// tile's 82 k clocks).  The VEC4
// epilogue requests them a GROUP of row quads ahead (straight-line code: the operands of group g+1 are in flight while group g
// is finished and stored).  Written as one load -> use -> store per quad, every quad paid a full memory round trip (vmcnt
// counts stores too: the wait for the quad's operands also waited for the previous quad's store) — 16 serial round trips per
// tile, 55 of a tile's 63 us on the 1M-edge edge GEMM.
// TRANS: the activation may be tanh / sigmoid / gelu.  Their expansions need ~10 temporaries per element: with them in the code the
// register allocator parks the operands in flight in scratch memory (load, wait, scratch store) on EVERY path, relu's included —
// so identity / relu launches get an instantiation without them.
// VEC4: W, the output and the epilogue operands are accessed as 16-B quads (OUT % 4 == 0, aligned); else element by element.
// LD: what the loader knows — 0: quad rows of modes 0-2; 1: also sums of rows (modes 3 / 4, the node update); 2: also segments whose
// rows are not quads (width % 4 != 0: element loads) and the packed form of narrow segments (mode 5).  Their loops and index tables
// cost the plain / gathered quad loader registers it does not have (spilled LDS addresses are reloaded from scratch memory in front of every access,
// and a scratch reload waits on vmcnt, i.e. for the global loads just issued): launches whose segments are all quad rows of
// modes 0-2 (the core's edge update, the FeedForward layers, the projections) get a kernel without them.
#ifndef GNX_GEMM_WPE_PLAIN  // waves per SIMD of the plain quad GEMM (no epilogue operand, lean loader: the node projections, FeedForward layers, dX of the backward)
#define GNX_GEMM_WPE_PLAIN 3
#endif
#ifndef GNX_GEMM_WPE_X6  // waves per SIMD of the 128-column kernels in the six-term form (three bf16 parts of each fragment: ~25 more live registers than the fp32 form)
#define GNX_GEMM_WPE_X6 3
#endif
template <int BN, int NL, int LD, bool X6 = false>
constexpr int gemm_wpe() { return (X6 && BN == 128) ? GNX_GEMM_WPE_X6 : ((BN == 128 && NL == 0 && LD == 0) ? GNX_GEMM_WPE_PLAIN : WaveLayout<BN>::WPE); }
template <int BN, bool VEC4, int KC, int NL, bool TRANS, int LD, bool X6>
__global__ __launch_bounds__(WaveLayout<BN>::WT) __attribute__((amdgpu_waves_per_eu(gemm_wpe<BN, NL, LD, X6>()))) void k_rows_gemm(WideArgs a) {
  using L = WaveLayout<BN>;
  constexpr int WT = L::WT;
  constexpr bool ONESEG = NL == 3;  // the projected edge update: ONE segment, the tile's own rows (mode 0) — its record stays in scalar registers,
                                    // no per-row index is read, the chunk loop tests one width (as generic code every chunk paid four serial scalar-load round trips)
  constexpr bool FULL = LD >= 1;   // loader with the row-sum modes 3 / 4
  constexpr bool ELEM = LD >= 2;   // ... and element-wise / packed segments
  constexpr int LDA = KC + 1;               // A row stride: odd => conflict-free ds_read_b32 of the A fragment
  constexpr int C4R = KC / 4;               // float4 per A row chunk
  constexpr int RPP = WT / C4R;             // A rows loaded per pass of the workgroup
  constexpr int NA4 = BM / RPP;             // float4 of the A chunk per thread
  constexpr int NB4 = (KC * BN / 4) / WT;  // float4 of the B chunk per thread
  constexpr int LDC = BN + 4;                       // epilogue staging: [64 rows][BN + 4]
  constexpr int POOL = (BM * LDA + KC * BN) > 64 * LDC ? (BM * LDA + KC * BN) : 64 * LDC;
  __shared__ __attribute__((aligned(16))) float s_pool[POOL];
  float* sA = s_pool;                               // [BM][LDA]
  float* sB = s_pool + BM * LDA;                    // [KC][BN]   (BM*LDA*4 is a multiple of 16)
  float* sC = s_pool;                               // reused after the K loop
  __shared__ int s_ia[BM], s_ib[BM];  // gather indices, or colptr range for the segment-sum mode
  __shared__ int s_ic[FULL ? BM : 1];  // mode 4: row of the node's SECOND partial sum (-1: none)
  __shared__ __attribute__((aligned(16))) float s_bias[BN];
  constexpr int PD_RPI = 256 / BN;  // destination rows per LDS-DMA piece (one wave instruction writes 64 x 16 B = 1 KiB of LDS: lane-linear)
  constexpr int PD_ROWS = (kPdRowsCap + PD_RPI - 1) / PD_RPI * PD_RPI;
  __shared__ __attribute__((aligned(16))) float s_pd[NL == 3 ? PD_ROWS * BN : 4];  // [row - first destination of the tile][BN]
  constexpr bool LNOK = VEC4 && LD <= 1;  // LayerNorm on load: the quad loaders only
  __shared__ float2 s_ln[LNOK ? BM : 1];                               // (mean, inv) of the tile's rows
  __shared__ __attribute__((aligned(16))) float4 s_lng[LNOK ? 64 : 1];  // gamma, beta of the normalised segment (width <= 256)
  __shared__ __attribute__((aligned(16))) float4 s_lnb[LNOK ? 64 : 1];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv / L::WN, wn = wv % L::WN;
  // block -> (row tile, column tile).  With several column tiles the SAME A rows are needed by all of them: blocks b and
  // b+8 land on the same XCD (round-robin dispatch), so the column tiles of a row tile are placed 8 blocks apart — they
  // run close in time on one XCD and the A tile is fetched from HBM once and re-read from that XCD's L2.
  int tile_id, ctile;
  if (a.n_ctiles > 1) {
    const int b = blockIdx.x;
    ctile = (b >> 3) % a.n_ctiles;
    tile_id = (b / (8 * a.n_ctiles)) * 8 + (b & 7);
    if (tile_id >= a.n_rtiles) return;  // grid padded to a multiple of 8 row tiles (whole block leaves: no barrier yet)
  } else {
    tile_id = blockIdx.x;
    ctile = 0;
  }
  if (a.stagger > 0 && blockIdx.x < 1024) {
    const int slot = blockIdx.x >> 8;
    for (int i = 0; i < slot * a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  const float* Wsel = a.W;
  const float* bias_sel = a.bias;
  const float* bias_g_sel = a.bias_g;
  float* out_sel = a.out;
  if (a.W2) {
    const int nct1 = a.n_ctiles >> 1;
    if (ctile >= nct1) { ctile -= nct1; Wsel = a.W2; bias_sel = a.bias2; bias_g_sel = a.bias_g2; out_sel = a.out2; }
  }
  tile_id = __builtin_amdgcn_readfirstlane(tile_id);  // wave-uniform by construction: lets the tile / chunk table reads be scalar loads
  ctile = __builtin_amdgcn_readfirstlane(ctile);
  Wsel = uniform_ptr(Wsel); bias_sel = uniform_ptr(bias_sel); bias_g_sel = uniform_ptr(bias_g_sel);  // (the weight chunk's loads: scalar base + one offset register)
  const Tile t = a.tiles[tile_id];
  int agg_row0[2] = {0, 0};  // first partial-sum row of the tile's two 64-row passes (read here: in the epilogue the load would wait for every store in flight)
  if (VEC4 && a.agg_out) { agg_row0[0] = a.chunk_row0[2 * tile_id]; agg_row0[1] = a.chunk_row0[2 * tile_id + 1]; }
  const int n0 = ctile * BN;
  const size_t r = blockIdx.z;
  const int row0 = a.row_kind == 0 ? t.e0 : t.n0;
  const int rows = (a.row_kind == 0 ? t.e1 : t.n1) - row0;

#ifdef GNX_WIDE_STAMPS_BUILD  // diagnostic build only (hipcc -DGNX_WIDE_STAMPS_BUILD): the shipped kernel executes no stamp
  unsigned long long st[6];
  st[0] = clock64();
#endif
  // ONE register array serves the K loop's chunk staging (A quads, then B quads) and, from the last chunk on, the epilogue's
  // first operand group: declared separately, the compiler keeps both sets alive through the loop and spills
  constexpr int NC4_ = (64 * BN / 4) / WT, GRP_ = NC4_ > GNX_GEMM_GRP ? GNX_GEMM_GRP : NC4_;
  constexpr int NSTG = NL == 3 ? (NA4 + NB4 > NC4_ ? NA4 + NB4 : NC4_) : (NA4 + NB4 > 4 * GRP_ ? NA4 + NB4 : 4 * GRP_);
  float4 stg[NSTG];  // K loop: A quads, B quads; epilogue: two operand buffers of 2 GRP quads each
  float4 rs[FULL ? NA4 : 1];  // mode 4: second partial-sum rows of the chunk
#define ra(i) stg[i]
#define rb(i) stg[NA4 + (i)]
  const int a_c4_0 = tid % C4R, a_r_0 = tid / C4R;
  // The row-sum loaders (LD >= 1) run with spilled registers at four workgroups per CU, and what the allocator parks in scratch memory are the
  // per-row LDS addresses derived from these two — reloaded inside the chunk loader, between its global loads, and a scratch reload waits for
  // every load in flight (by the .s: four serial memory round trips per chunk of the node update).  Derived afresh from a thread id the compiler
  // cannot trace back, in each place that needs them, they are a few VALU operations instead of registers held across the K loop.
#define GNX_LOADER_COORDS                                        \
  int a_c4 = a_c4_0, a_r = a_r_0;                                \
  if (FULL || gemm_wpe<BN, NL, LD>() > WaveLayout<BN>::WPE) { int tl_ = tid; asm volatile("" : "+v"(tl_)); a_c4 = tl_ % C4R; a_r = tl_ / C4R; }

  // The mode is tested ONCE per chunk, outside the row loop, and every mode is straight-line code with unconditional loads of
  // clamped addresses whose validity is recorded in bit masks and applied when the chunk goes to LDS: with a test in front of each
  // load every load sat in its own branch and the compiler's counter merge at each join put an `s_waitcnt vmcnt(0)` between
  // them (four serial memory round trips per chunk — by the stamp build 34 of the K loop's 57 k clocks on the 1M-edge edge GEMM
  // went into issuing loads); a select on the loaded value at this point waits for the load just the same.
  unsigned okmask = 0;  // quads: bit i: ra(i) holds data (else zero), bit NA4 + i: rb(i), bit NA4 + NB4 + i: rs[i]
  int pend_ln = -1;     // the chunk in the staging registers is columns [pend_ln, pend_ln + KC) of the normalised segment (-1: plain)
  unsigned moremask = 0;  // mode 4: bit i: row i of the staged chunk has partial sums beyond the first two (fetched by store_chunk)
  const float* more_base = nullptr;
  int more_w = 0, more_k = 0;
  unsigned emask = 0;   // elements of element-wise loaded quads (FULL / !VEC4): bit 4 i + e: ra(i)[e], bit 16 + 4 i + e: rb(i)[e]
  auto load_chunk = [&](int si, int kc) {
    GNX_LOADER_COORDS
    okmask = 0;
    emask = 0;
    moremask = 0;
    const WSeg sg = a.seg[ONESEG ? 0 : si];
    pend_ln = (LNOK && sg.ln) ? kc : -1;
    const float* base = sg.base + r * sg.rep_stride;
    const int k = kc + 4 * a_c4;
    if (!FULL || (sg.mode <= 2 && (!ELEM || sg.vec))) {
      const bool kok = k < sg.width;
      const int kcl = kok ? k : 0;
      const float* ub = (ONESEG || sg.mode == 0) ? base + (size_t)row0 * sg.width : base;  // uniform: the tile's first row, or the gathered table
#pragma unroll
      for (int i = 0; i < NA4; ++i) {
        const int row = a_r + RPP * i;
        const int rc = min(row, rows - 1);
        int grow = rc;
        if (!ONESEG) { const int ia = s_ia[rc], ib = s_ib[rc]; grow = sg.mode == 0 ? rc : (sg.mode == 1 ? ia : ib); }
        ra(i) = ld4(ub, (unsigned)grow * (unsigned)sg.width + (unsigned)kcl);
        const bool ok = kok && row < rows;
        okmask |= ok ? (1u << i) : 0u;
        if (FULL) emask |= ok ? (15u << (4 * i)) : 0u;
      }
    } else if (ELEM && sg.mode <= 2) {
      // rows that are not quads (width % 4 != 0): element loads
      const float* ub = sg.mode == 0 ? base + (size_t)row0 * sg.width : base;
#pragma unroll
      for (int i = 0; i < NA4; ++i) {
        const int row = a_r + RPP * i;
        const int rc = min(row, rows - 1);
        const int ia = s_ia[rc], ib = s_ib[rc];
        const unsigned ro = (unsigned)(sg.mode == 0 ? rc : (sg.mode == 1 ? ia : ib)) * (unsigned)sg.width;
        float4 v;
        v.x = ld1(ub, ro + (unsigned)min(k, sg.width - 1));
        v.y = ld1(ub, ro + (unsigned)min(k + 1, sg.width - 1));
        v.z = ld1(ub, ro + (unsigned)min(k + 2, sg.width - 1));
        v.w = ld1(ub, ro + (unsigned)min(k + 3, sg.width - 1));
        ra(i) = v;
        if (row < rows) emask |= ((k < sg.width ? 1u : 0u) | (k + 1 < sg.width ? 2u : 0u) | (k + 2 < sg.width ? 4u : 0u) | (k + 3 < sg.width ? 8u : 0u)) << (4 * i);
      }
    } else if (ELEM && sg.mode == 5) {
      // narrow segments side by side in one K range (e.g. the encoder's [ef(10) | nf_src(5) | nf_dst(5)]: one chunk instead of
      // three): element kk of the range belongs to segment (kk >= p1) + (kk >= p2)
      const int p1 = a.pk[0].width, p2 = a.npk > 1 ? p1 + a.pk[1].width : 0x7fffffff;
      const float* b0 = a.pk[0].base + r * a.pk[0].rep_stride;
      const float* b1 = a.npk > 1 ? a.pk[1].base + r * a.pk[1].rep_stride : b0;
      const float* b2 = a.npk > 2 ? a.pk[2].base + r * a.pk[2].rep_stride : b0;
      const int w0 = a.pk[0].width, w1 = a.npk > 1 ? a.pk[1].width : 1, w2 = a.npk > 2 ? a.pk[2].width : 1;
      const int m0 = a.pk[0].mode, m1 = a.npk > 1 ? a.pk[1].mode : 0, m2 = a.npk > 2 ? a.pk[2].mode : 0;
#pragma unroll
      for (int i = 0; i < NA4; ++i) {
        const int row = a_r + RPP * i;
        const int rc = min(row, rows - 1);
        const int ia = s_ia[rc], ib = s_ib[rc], own = row0 + rc;
        const int r0 = m0 == 0 ? own : (m0 == 1 ? ia : ib), r1 = m1 == 0 ? own : (m1 == 1 ? ia : ib), r2 = m2 == 0 ? own : (m2 == 1 ? ia : ib);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kk = min(k + e, sg.width - 1);
          const bool s1 = kk >= p1, s2 = kk >= p2;
          const float* b = s2 ? b2 : (s1 ? b1 : b0);
          const unsigned off = s2 ? (unsigned)r2 * (unsigned)w2 + (unsigned)(kk - p2) : (s1 ? (unsigned)r1 * (unsigned)w1 + (unsigned)(kk - p1) : (unsigned)r0 * (unsigned)w0 + (unsigned)kk);
          v[e] = b[off];
          emask |= (row < rows && k + e < sg.width) ? (1u << (4 * i + e)) : 0u;
        }
        ra(i) = make_float4(v[0], v[1], v[2], v[3]);
      }
    } else if (FULL && sg.mode == 3) {
      // segment sum over the node's in-edges (CSC range), 4 rows in flight: unconditional clamped loads (a load inside the
      // guarded loop body cannot be hoisted, which costs one memory round trip per edge); quad / element form chosen outside
      auto segsum_rows = [&](auto vec_c) {
        constexpr bool VEC = decltype(vec_c)::value;
#pragma unroll
        for (int i = 0; i < NA4; ++i) {
          const int row = a_r + RPP * i;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (row < rows && k < sg.width) {
            const int e0 = s_ia[row], e1 = s_ib[row];
            for (int e = e0; e < e1; e += 4) {
              float4 u[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float* p = base + (size_t)min(e + j, e1 - 1) * sg.width + k;
                if (VEC) {
                  u[j] = *reinterpret_cast<const float4*>(p);
                } else {
                  u[j].x = p[0];
                  u[j].y = k + 1 < sg.width ? p[1] : 0.f;
                  u[j].z = k + 2 < sg.width ? p[2] : 0.f;
                  u[j].w = k + 3 < sg.width ? p[3] : 0.f;
                }
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if (e + j < e1) { v.x += u[j].x; v.y += u[j].y; v.z += u[j].z; v.w += u[j].w; }
              }
            }
          }
          ra(i) = v;
        }
      };
      if (!ELEM || sg.vec) segsum_rows(std::true_type{}); else segsum_rows(std::false_type{});
      emask |= 0xffffu;
    } else if (FULL) {
      // mode 4 — the node's in-edge sum: the first and the second partial row of every tile row in one go (clamped,
      // unconditional loads; added when the chunk goes to LDS), the rare further parts (in-degree beyond two chunks) after them
      const bool kok = k < sg.width;
      const int kcl = kok ? k : 0;
#pragma unroll
      for (int i = 0; i < NA4; ++i) {
        const int rc = min(a_r + RPP * i, rows - 1);
        const int pr = s_ia[rc], p2 = s_ic[FULL ? rc : 0];
        const bool ok = kok && pr >= 0 && a_r + RPP * i < rows;
        ra(i) = ld4(base, (unsigned)max(pr, 0) * (unsigned)sg.width + (unsigned)kcl);
        rs[FULL ? i : 0] = ld4(base, (unsigned)max(p2, 0) * (unsigned)sg.width + (unsigned)kcl);
        emask |= ok ? (15u << (4 * i)) : 0u;
        okmask |= (ok && p2 >= 0) ? (1u << (NA4 + NB4 + i)) : 0u;
        moremask |= (ok && s_ib[rc] > 2) ? (1u << i) : 0u;  // in-degree beyond two chunks: the further parts are added when the chunk goes to LDS
      }
      more_base = base; more_w = sg.width; more_k = k;
      // (the further parts used to be fetched HERE, inside an `if (any row has more)`: loads in a branch between this chunk's requests and
      //  the matrix-core work make the compiler drain vmcnt at the join — every chunk of the node update waited for the loads it had
      //  just issued, 36 k of a tile's 102 k clocks by the stamps, whether or not any node of the tile had such a degree)
    }
    const int ldw = a.ldw ? a.ldw : a.OUT;
    const float* wb = Wsel + (size_t)(sg.w_row0 + kc) * ldw + n0;  // uniform: first row of the chunk, first column of the tile
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      const int q = tid + WT * i;
      const int kk = q / (BN / 4), c4 = q % (BN / 4);
      const int n = n0 + 4 * c4;
      const bool rok = kc + kk < sg.width;
      const unsigned ro = rok ? (unsigned)kk * (unsigned)ldw : 0u;
      if (VEC4) {
        const bool ok = rok && n < a.OUT;
        rb(i) = ld4(wb, ok ? ro + 4u * c4 : 0u);
        okmask |= ok ? (1u << (NA4 + i)) : 0u;
      } else {
        const int nl = a.OUT - 1 - n0;  // last valid column of the tile (>= 0)
        float4 v;
        v.x = ld1(wb, ro + (unsigned)min(4 * c4, nl));
        v.y = ld1(wb, ro + (unsigned)min(4 * c4 + 1, nl));
        v.z = ld1(wb, ro + (unsigned)min(4 * c4 + 2, nl));
        v.w = ld1(wb, ro + (unsigned)min(4 * c4 + 3, nl));
        rb(i) = v;
        if (rok) emask |= ((n < a.OUT ? 1u : 0u) | (n + 1 < a.OUT ? 2u : 0u) | (n + 2 < a.OUT ? 4u : 0u) | (n + 3 < a.OUT ? 8u : 0u)) << (16 + 4 * i);
      }
    }
  };
  // The first chunk does not wait for the index arrays when its segment is the tile's own rows (mode 0: the loader's index reads are
  // then unused): its loads go out here, one memory round trip ahead of the index loads' round trip instead of behind it.
  int si = 0, kc = 0;
  const int seg0_width = a.seg[0].width;
  while (si < a.nseg && a.seg[si].width == 0) ++si;
  const bool early_first = si < a.nseg && a.seg[si].mode == 0;
  if (early_first) load_chunk(si, kc);
  // per-row indices
  bool need_cp = false, need_idx = false, need_agg = false;
  for (int s = 0; s < a.nseg; ++s) {
    need_cp |= a.seg[s].mode == 3;
    need_idx |= a.seg[s].mode == 1 || a.seg[s].mode == 2;
    need_agg |= a.seg[s].mode == 4;
  }
  for (int s = 0; s < a.npk; ++s) need_idx |= a.pk[s].mode != 0;
  need_idx |= a.gadd_a != nullptr || a.agg_out != nullptr;
  if (tid < BM) {
    const int m = tid < rows ? tid : rows - 1;
    if (need_cp) {
      s_ia[tid] = a.cp[row0 + m];
      s_ib[tid] = tid < rows ? a.cp[row0 + m + 1] : s_ia[tid];
    } else if (need_agg) {  // (row of the first partial, number of further chunks << 24 | first chunk is resolved in the loader)
      const int parts = tid < rows ? a.node_agg_parts[row0 + m] : 0;
      s_ia[tid] = a.node_agg_row[row0 + m];
      s_ib[tid] = parts;
      // a node whose in-edges run into a second 64-row chunk (one in ~6 on the ER graph): resolve that row ONCE here — in the chunk
      // loader the lookup is two dependent loads in front of the row load, per chunk
      if (FULL) s_ic[tid] = parts > 1 ? a.chunk_row0[a.node_agg_chunk[row0 + m] + 1] : -1;
    } else if (need_idx) {
      s_ia[tid] = a.idx_a[row0 + m];
      s_ib[tid] = a.idx_b[row0 + m];
    }
  }
  if (LNOK && a.ln_stats) {
    if (tid < BM) s_ln[tid] = ld_stats(reinterpret_cast<const float2*>(a.ln_stats + r * a.ln_rep_stride) + row0 + (tid < rows ? tid : rows - 1));
    if (tid >= BM && tid < BM + 64) {
      const int q = tid - BM;
      const bool in = 4 * q < a.ln_width;
      s_lng[q] = in ? reinterpret_cast<const float4*>(a.ln_g)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      s_lnb[q] = in ? reinterpret_cast<const float4*>(a.ln_b)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // tile bias: b + W[gf rows]^T gf[g], folded once per graph by k_fold_bias (a per-tile fold is a runtime-length chain
  // of dependent global loads: 32 round trips, more than the tile's whole MFMA time)
  if (tid < BN) {
    const int n = n0 + tid;
    float b = 0.f;
    if (n < a.OUT) b = bias_g_sel ? bias_g_sel[(r * a.n_graphs + t.g) * (size_t)a.OUT + n] : (bias_sel ? bias_sel[n] : 0.f);
    s_bias[tid] = b;
  }
  __syncthreads();
  int pd_first = 0;
  if (NL == 3) {
    // the tile's destination rows, first to last (the host launches this form only when every edge tile's run fits the table):
    // no register, no wait — the pieces are in LDS once this wave's vmcnt has drained and a barrier has been passed; the K loop's
    // first __syncthreads() does both (a launch without K chunks waits explicitly below)
    pd_first = __builtin_amdgcn_readfirstlane(s_ib[0]);
    const int pd_last = min(__builtin_amdgcn_readfirstlane(s_ib[rows - 1]), pd_first + kPdRowsCap - 1);
    const float* pdt = a.gadd_b + r * a.gadd_rep_stride;
    const int span = pd_last - pd_first + 1;
    const int prow = lane / (BN / 4), pq = lane % (BN / 4);
    const unsigned pcol = (unsigned)min(n0 + 4 * pq, a.OUT - 4);
    for (int p = wv; p * PD_RPI < span; p += L::WAVES) {
      const int row = min(pd_first + p * PD_RPI + prow, pd_last);
      const float* g = pdt + ((unsigned)row * (unsigned)a.OUT + pcol);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(s_pd + p * 256), 16, 0, 0);
    }
  }

  f32x16 acc[L::TM][L::TN];
#pragma unroll
  for (int i = 0; i < L::TM; ++i)
#pragma unroll
    for (int j = 0; j < L::TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  auto store_chunk = [&]() {
    GNX_LOADER_COORDS
    if (FULL && moremask) {  // rare (a node whose in-edges run through more than two 64-row chunks): synchronous loads, behind the wait for the chunk itself
#pragma unroll
      for (int i = 0; i < NA4; ++i) {
        if ((moremask >> i) & 1u) {
          const int rc = min(a_r + RPP * i, rows - 1);
          const int parts = s_ib[rc];
          const int c0 = a.node_agg_chunk[row0 + rc];
          for (int j = 2; j < parts; ++j) {
            const float4 u = ld4(more_base, (unsigned)a.chunk_row0[c0 + j] * (unsigned)more_w + (unsigned)more_k);
            float4& t4 = rs[FULL ? i : 0];
            t4.x += u.x; t4.y += u.y; t4.z += u.z; t4.w += u.w;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NA4; ++i) {
      float* d = sA + (a_r + RPP * i) * LDA + 4 * a_c4;
      float4 v = ra(i);
      if (LNOK && pend_ln >= 0) {  // (columns beyond the segment read gamma / beta of valid LDS slots and are zeroed by the masks below)
        const float2 st = s_ln[a_r + RPP * i];
        const float4 g = s_lng[((pend_ln >> 2) + a_c4) & 63], b = s_lnb[((pend_ln >> 2) + a_c4) & 63];
        // Round 6 (profiles/r06_overlap_hazard.log): with another kernel's workgroups on the CU (a hipBLASLt GEMM through another queue) the LAST 16
        // LANES of a wave used wrong statistics here now and then — the rows they stage came out normalised with other numbers, ~1 % off — when
        // the first vector instruction consumed the (8-lanes-per-address) LDS read right behind the compiler's counted wait; every LDS read of the
        // branch retired plus sixteen idle issue slots in front of the first use: 0 of 360 runs wrong where 54-63 of 120 were.  The cause on the
        // hardware side is not established; the guard costs a few clocks per staged quad.  (-DGNX_LN_GUARD=0: the unguarded form, A/B runs.)
        float sx_ = st.x, sy_ = st.y;
#if GNX_LN_GUARD
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "+v"(sx_), "+v"(sy_)::"memory");
#endif
        v.x = fmaf(g.x, (v.x - sx_) * sy_, b.x); v.y = fmaf(g.y, (v.y - sx_) * sy_, b.y);
        v.z = fmaf(g.z, (v.z - sx_) * sy_, b.z); v.w = fmaf(g.w, (v.w - sx_) * sy_, b.w);
      }
      if (FULL) {
        const unsigned m = emask >> (4 * i);
        v.x = (m & 1u) ? v.x : 0.f; v.y = (m & 2u) ? v.y : 0.f; v.z = (m & 4u) ? v.z : 0.f; v.w = (m & 8u) ? v.w : 0.f;
        if ((okmask >> (NA4 + NB4 + i)) & 1u) { const float4 u = rs[FULL ? i : 0]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
      } else if (!((okmask >> i) & 1u)) {
        v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      const int q = tid + WT * i;
      float4 v = rb(i);
      if (VEC4) {
        if (!((okmask >> (NA4 + i)) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const unsigned m = emask >> (16 + 4 * i);
        v.x = (m & 1u) ? v.x : 0.f; v.y = (m & 2u) ? v.y : 0.f; v.z = (m & 4u) ? v.z : 0.f; v.w = (m & 8u) ? v.w : 0.f;
      }
      *reinterpret_cast<float4*>(sB + 4 * q) = v;  // q = kk*(BN/4) + c4  ->  sB[kk][4*c4]
    }
  };

  const int hi = lane >> 5, l31 = lane & 31;
  // ---- epilogue operands (declared here: the first group is requested during the LAST chunk's matrix-core work) ----
  float* out = out_sel + r * a.out_rep_stride;
  constexpr int NC4 = (64 * BN / 4) / WT;  // float4 per thread and pass
  constexpr int NG = WT / (BN / 4);        // row groups that share a column quad
  float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // VEC4 epilogue: thread = (column quad q4, row lr0 + NG*i of the pass), GRP row quads per step
  constexpr int GRP = NC4 > GNX_GEMM_GRP ? GNX_GEMM_GRP : NC4;
  constexpr int NGRP = NC4 / GRP;
  constexpr int NTG = 2 * NGRP;            // operand groups of the tile, in order (pass, group): group t lives in buffer t & 1
  const int q4 = tid % (BN / 4), lr0 = tid / (BN / 4);
  const int ncol = n0 + 4 * q4;
  const bool col_ok = ncol < a.OUT;
  float* out_tile = out + (size_t)row0 * a.OUT;
  static_assert(GRP == GRP_, "operand group size");
  // operand streams X / Y, double-buffered over the groups, both buffers in the chunk staging registers
#define ex(b, u) stg[(b) * 2 * GRP + (u)]
#define ey(b, u) stg[(b) * 2 * GRP + GRP + (u)]
  const float* xb = nullptr;
  const float* yb = nullptr;
  int xk = 0, yk = 0;  // row of the operand: 0 the output row itself, 1 idx_a[row], 2 idx_b[row]
  if (NL == 1 || NL == 2) {
    const size_t own = r * a.out_rep_stride + (size_t)row0 * a.OUT;  // operands with the layout of `out`: the tile's first row
    switch (a.epi) {
      case EPI_GADD: xb = a.gadd_a + r * a.gadd_rep_stride; yb = a.gadd_b + r * a.gadd_rep_stride; xk = 1; yk = 2; break;
      case EPI_ADD1: xb = a.add1 + own; break;
      case EPI_ADD12: xb = a.add1 + own; yb = a.add2 + own; break;
      case EPI_GMUL: xb = a.gmul + own; break;
      default: xb = a.gmul + own; yb = a.add1 + own; break;  // EPI_GMUL_ADD
    }
  }
  // one output quad's worth of an operand row: a 16-B load, or (rows that are not quads) four element loads of clamped columns
  auto ldq = [&](const float* ub, unsigned rowoff) -> float4 {
    if (VEC4) return ld4(ub, rowoff + (unsigned)(col_ok ? ncol : 0));
    const int last = a.OUT - 1;
    return make_float4(ld1(ub, rowoff + (unsigned)min(ncol, last)), ld1(ub, rowoff + (unsigned)min(ncol + 1, last)),
                       ld1(ub, rowoff + (unsigned)min(ncol + 2, last)), ld1(ub, rowoff + (unsigned)min(ncol + 3, last)));
  };
  // NL = 3: the source rows of a whole pass, into `buf` (pass 0: the staging registers; pass 1: e1)
  v4f e1[NL == 3 ? NC4 : 1];
  auto issue_src_rows0 = [&]() {
    const float* pst = a.gadd_a + r * a.gadd_rep_stride;
#pragma unroll
    for (int u = 0; u < NC4; ++u) {
      const int row = min(lr0 + NG * u, rows - 1);
      stg[u] = ldq(pst, (unsigned)s_ia[row] * (unsigned)a.OUT);
    }
  };
  auto issue_src_rows1 = [&]() {
    const float* pst = a.gadd_a + r * a.gadd_rep_stride;
#pragma unroll
    for (int u = 0; u < NC4; ++u) {
      const int row = min(64 + lr0 + NG * u, rows - 1);
      ld4_untracked(e1[u], pst, (unsigned)s_ia[row] * (unsigned)a.OUT + (unsigned)(col_ok ? ncol : 0));
    }
  };
  auto issue_operands = [&](int pass, int g, int buf) {  // unconditional loads of clamped rows / columns: nothing to branch around
#pragma unroll
    for (int u = 0; u < GRP; ++u) {
      const int row = min(64 * pass + lr0 + NG * (g * GRP + u), rows - 1);
      const int ia = s_ia[row], ib = s_ib[row];  // (read unconditionally: a select, not a branch around an LDS read)
      const float4 xv = ldq(xb, (unsigned)(xk == 0 ? row : ia) * (unsigned)a.OUT);
      ex(buf, u) = xv;
      if (NL == 2) {
        const float4 yv = ldq(yb, (unsigned)(yk == 0 ? row : ib) * (unsigned)a.OUT);
        ey(buf, u) = yv;
      }
    }
  };
#ifdef GNX_WIDE_STAMPS_BUILD
  st[1] = clock64();
  unsigned long long t_sync = 0, t_mfma = 0, t_issue = 0, t_estage = 0, t_egroups = 0, t_eagg = 0;
#endif
  // the matrix-core work of the chunk in LDS.  Fragments of k-step kk+1 are requested from LDS before the MFMAs of step kk are
  // issued (explicit two-deep register pipeline: left to itself the compiler places each ds_read right in front of its first use)
  auto mma_chunk = [&]() {
    if constexpr (X6) {
      // the DEFAULT form (gnx_x6_mma.h): the chunk's two 16-steps as six bf16 matrix-core terms each, the fp32 fragments split on the fly — the
      // same LDS reads as the fp32 form below (eight per 16-step, row and fragment), 6 x 8-pass instead of 8 x 16-pass matrix instructions
#pragma unroll
      for (int s16 = 0; s16 < KC / 16; ++s16) {
        // (one A fragment alive at a time: with all of a step's fragments split up front the 128-column kernels kept 25-40 registers in scratch memory)
        X6Frag fb6[L::TN];
#pragma unroll
        for (int j = 0; j < L::TN; ++j) fb6[j] = x6_frag(sB + (16 * s16 + 8 * hi) * BN + (wn * L::TN + j) * 32 + l31, BN);
#pragma unroll
        for (int i = 0; i < L::TM; ++i) {
          const X6Frag fa6 = x6_frag(sA + ((i * L::WM + wm) * 32 + l31) * LDA + 16 * s16 + 8 * hi, 1);
#pragma unroll
          for (int j = 0; j < L::TN; ++j) acc[i][j] = x6_mma(fa6, fb6[j], acc[i][j]);
        }
      }
      return;
    }
    float fa[2][L::TM], fb[2][L::TN];
#pragma unroll
    for (int i = 0; i < L::TM; ++i) fa[0][i] = sA[((i * L::WM + wm) * 32 + l31) * LDA + hi];
#pragma unroll
    for (int j = 0; j < L::TN; ++j) fb[0][j] = sB[hi * BN + (wn * L::TN + j) * 32 + l31];
#pragma unroll
    for (int kk = 0; kk < KC / 2; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < KC / 2) {
#pragma unroll
        for (int i = 0; i < L::TM; ++i) fa[nxt][i] = sA[((i * L::WM + wm) * 32 + l31) * LDA + 2 * (kk + 1) + hi];
#pragma unroll
        for (int j = 0; j < L::TN; ++j) fb[nxt][j] = sB[(2 * (kk + 1) + hi) * BN + (wn * L::TN + j) * 32 + l31];
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the reads above in front of this step's MFMAs
#pragma unroll
      for (int i = 0; i < L::TM; ++i)
#pragma unroll
        for (int j = 0; j < L::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
    }
  };
  if (si < a.nseg) { if (!early_first) load_chunk(si, kc); }
  else if (NL == 1 || NL == 2) {  // (no K at all: bias / operands only)
    issue_operands(0, 0, 0);
    if (NTG > 1) issue_operands(NGRP > 1 ? 0 : 1, NGRP > 1 ? 1 : 0, 1);
  } else if (NL == 3) {
    __syncthreads();  // the destination rows' LDS-DMA (no K loop barrier drains it; a wait the compiler can see — behind an inline-asm wait it
                      // still counts the DMA as pending and puts vmcnt(0) in front of every read of s_pd, which waits for the operand loads too)
    issue_src_rows0();
  }
  while (si < a.nseg) {
#ifdef GNX_WIDE_STAMPS_BUILD
    const unsigned long long tA = clock64();
#endif
    __syncthreads();  // everyone is done reading the previous chunk
    store_chunk();
    __syncthreads();
#ifdef GNX_WIDE_STAMPS_BUILD
    const unsigned long long tB = clock64();
    t_sync += tB - tA;
#endif
    // advance and prefetch the next chunk while the matrix cores work on this one
    kc += KC;
    if (ONESEG) {
      if (kc >= seg0_width) si = a.nseg;
    } else if (kc >= a.seg[si].width) {
      kc = 0;
      ++si;
      while (si < a.nseg && a.seg[si].width == 0) ++si;
    }
    if (si < a.nseg) load_chunk(si, kc);
    else if (NL == 1 || NL == 2) {  // last chunk: the staging registers are free — the epilogue's first two operand groups take them
      issue_operands(0, 0, 0);
      if (NTG > 1) issue_operands(NGRP > 1 ? 0 : 1, NGRP > 1 ? 1 : 0, 1);
    } else if (NL == 3) {
      issue_src_rows0();
    }
#ifdef GNX_WIDE_STAMPS_BUILD
    t_issue += clock64() - tB;
#endif
    mma_chunk();
#ifdef GNX_WIDE_STAMPS_BUILD
    t_mfma += clock64() - tB;
#endif
  }
#ifdef GNX_WIDE_STAMPS_BUILD
  st[2] = clock64();
#endif

  // ---- epilogue: the tile goes through LDS (two 64-row passes) so that HBM sees full-row 16-B vector stores (the direct
  //      C/D-layout store is 64 dword stores per lane — measured: ff1 3.43 vs 2.84 ms, ff2 3.08 vs 2.55 ms on C4); bias', the gathered node projections, the activation, the column
  //      sums for the graph update and the residual adds are applied on the vectorised side ----
  __shared__ int s_seg[66];  // per-destination runs of a pass: s_seg[k] = first row of run k, s_seg[n_seg] = valid rows; s_seg[65] = n_seg
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#ifdef GNX_WIDE_STAMPS_BUILD
    const unsigned long long te0 = clock64();
#endif
    lds_barrier();  // K-loop readers (pass 0) / previous pass readers are done with the pool
    if (VEC4 && a.agg_out && wv == 0) {  // rows are dst-sorted: a destination is a contiguous run; found from the indices alone
      const int nvalid = min(max(rows - 64 * pass, 0), 64);
      const int d = lane < nvalid ? s_ib[64 * pass + lane] : -1;
      const int dprev = lane > 0 && lane < nvalid ? s_ib[64 * pass + lane - 1] : -2;
      const bool head = lane < nvalid && d != dprev;
      const unsigned long long mask = __ballot(head);
      const int rank = __popcll(mask & ((1ull << lane) - 1ull));
      if (head) s_seg[rank] = lane;
      if (lane == 0) { const int ns = __popcll(mask); s_seg[ns] = nvalid; s_seg[65] = ns; }
    }
#pragma unroll
    for (int i = 0; i < L::TM; ++i) {
      const int rbase = (i * L::WM + wm) * 32;  // 32-row block of this wave (interleaved: every wave hands half of its accumulators to pass 0)
      if (L::TM == 2 ? i == pass : rbase / 64 == pass) {  // (TM = 2: block i of every wave belongs to pass i — a compile-time test, so pass 0 frees half of the accumulators)
#pragma unroll
        for (int j = 0; j < L::TN; ++j) {
          const int col = (wn * L::TN + j) * 32 + l31;
#pragma unroll
          for (int q = 0; q < 16; ++q)  // C/D layout of the 32x32 MFMA: row = (q&3) + 8*(q>>2) + 4*hi
            sC[(rbase - 64 * pass + (q & 3) + 8 * (q >> 2) + 4 * hi) * LDC + col] = acc[i][j][q];
        }
      }
    }
    lds_barrier();
    if (NL == 3 && pass == 0) {
      // pass 0's source rows (requested in front of the last chunk's matrix-core work) are USED here, in front of pass 1's requests:
      // the wait the compiler places is then vmcnt(0) with nothing younger in flight; placed at their first real use, behind the
      // requests below, it is vmcnt(0) too (their registers were written on both sides of a branch inside the K loop), i.e. a wait
      // for the rows just requested
#pragma unroll
      for (int u = 0; u < NC4; ++u) asm volatile("" : "+v"(stg[u].x), "+v"(stg[u].y), "+v"(stg[u].z), "+v"(stg[u].w));
      issue_src_rows1();  // into the registers of the accumulators just handed to LDS
    }
#ifdef GNX_WIDE_STAMPS_BUILD
    const unsigned long long te1 = clock64();
    t_estage += te1 - te0;
#endif
    {
#pragma unroll
      for (int g = 0; g < NGRP; ++g) {
        const int tg = pass * NGRP + g, cur = tg & 1;
        float4 v[GRP];
        const float4 b = *reinterpret_cast<const float4*>(s_bias + 4 * q4);
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
          v[u] = *reinterpret_cast<const float4*>(sC + (lr0 + NG * (g * GRP + u)) * LDC + 4 * q4);
          v[u].x += b.x; v[u].y += b.y; v[u].z += b.z; v[u].w += b.w;
        }
        if (NL == 2 && a.epi == EPI_GADD) {
#pragma unroll
          for (int u = 0; u < GRP; ++u) {
            const float4 p = ex(cur, u), d = ey(cur, u);
            v[u].x += p.x + d.x; v[u].y += p.y + d.y; v[u].z += p.z + d.z; v[u].w += p.w + d.w;
          }
        }
        if (NL == 3) {  // (same association as the two-stream form: (acc + b) + (p + d) — bit-identical outputs)
          float4 d[GRP];
#pragma unroll
          for (int u = 0; u < GRP; ++u) {
            const int row = min(64 * pass + lr0 + NG * (g * GRP + u), rows - 1);
            d[u] = *reinterpret_cast<const float4*>(s_pd + min(s_ib[row] - pd_first, PD_ROWS - 1) * BN + 4 * q4);  // (the clamp never acts on a launch the host allows)
          }
#pragma unroll
          for (int u = 0; u < GRP; ++u) {
            float4 p = stg[g * GRP + u];
            if (pass == 1) { const v4f q = e1[g * GRP + u]; p = make_float4(q.x, q.y, q.z, q.w); }
            v[u].x += p.x + d[u].x; v[u].y += p.y + d[u].y; v[u].z += p.z + d[u].z; v[u].w += p.w + d[u].w;
          }
        }
        if (TRANS) {
          act_apply_n<4 * GRP>(reinterpret_cast<float*>(v), a.act);
        } else if (a.act == 1) {
#pragma unroll
          for (int u = 0; u < GRP; ++u) { v[u].x = fmaxf(v[u].x, 0.f); v[u].y = fmaxf(v[u].y, 0.f); v[u].z = fmaxf(v[u].z, 0.f); v[u].w = fmaxf(v[u].w, 0.f); }
        }
        if ((NL == 1 || NL == 2) && (a.epi == EPI_GMUL || a.epi == EPI_GMUL_ADD)) {
#pragma unroll
          for (int u = 0; u < GRP; ++u) {
            const float4 x = ex(cur, u);
            v[u].x *= act_grad_from_out(x.x, a.gmul_act); v[u].y *= act_grad_from_out(x.y, a.gmul_act);
            v[u].z *= act_grad_from_out(x.z, a.gmul_act); v[u].w *= act_grad_from_out(x.w, a.gmul_act);
          }
        }
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
          const int lr = lr0 + NG * (g * GRP + u);
          const bool ok = 64 * pass + lr < rows && col_ok;
          if (ok) { cs4.x += v[u].x; cs4.y += v[u].y; cs4.z += v[u].z; cs4.w += v[u].w; }  // column sums BEFORE the residual adds
          if (VEC4 && a.agg_out) *reinterpret_cast<float4*>(sC + lr * LDC + 4 * q4) = v[u];       // the finished value, for the per-destination sums below
        }
        if ((NL == 1 || NL == 2) && (a.epi == EPI_ADD1 || a.epi == EPI_ADD12)) {
#pragma unroll
          for (int u = 0; u < GRP; ++u) { const float4 x = ex(cur, u); v[u].x += x.x; v[u].y += x.y; v[u].z += x.z; v[u].w += x.w; }
        }
        if (NL == 2 && (a.epi == EPI_ADD12 || a.epi == EPI_GMUL_ADD)) {
#pragma unroll
          for (int u = 0; u < GRP; ++u) { const float4 y = ey(cur, u); v[u].x += y.x; v[u].y += y.y; v[u].z += y.z; v[u].w += y.w; }
        }
        if ((NL == 1 || NL == 2) && tg + 2 < NTG) issue_operands((tg + 2) / NGRP, (tg + 2) % NGRP, cur);  // this buffer is consumed: refill it, BEFORE the stores
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
          const int row = 64 * pass + lr0 + NG * (g * GRP + u);
          if (row < rows && col_ok) {
            const unsigned o = (unsigned)row * (unsigned)a.OUT + (unsigned)ncol;
            if (VEC4) {
              st4(out_tile, o, v[u]);
            } else {  // rows that are not quads: element stores of the valid columns
              out_tile[o] = v[u].x;
              if (ncol + 1 < a.OUT) out_tile[o + 1] = v[u].y;
              if (ncol + 2 < a.OUT) out_tile[o + 2] = v[u].z;
              if (ncol + 3 < a.OUT) out_tile[o + 3] = v[u].w;
            }
          }
        }
      }
    }
    if (NL == 3 && pass == 0) {
      // pass 1's source rows have had this pass's group work to arrive: wait for all but the row stores issued since — on a full
      // tile every wave has issued exactly NC4 of them (a wave of a partial tile may have skipped some: it waits for everything)
      // ONE statement naming every destination (two statements on the sides of a branch make the compiler copy the registers into
      // each statement's operands in front of the branch, i.e. read them before the wait); the branch is inside the string
      static_assert(NC4 == 8 || NC4 == 4 || NC4 == 2, "source-row registers named in one statement");
      const int full = __builtin_amdgcn_readfirstlane((rows == BM && n0 + BN <= a.OUT) ? 1 : 0);  // wave-uniform
#define GNX_WAIT_ROWS "s_cmp_eq_u32 %[f], 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(%[n])\n\ts_branch 2f\n1:\n\ts_waitcnt vmcnt(0)\n2:"
      if constexpr (NC4 == 8)
        asm volatile(GNX_WAIT_ROWS : "+v"(e1[0]), "+v"(e1[1]), "+v"(e1[2]), "+v"(e1[3]), "+v"(e1[4]), "+v"(e1[5]), "+v"(e1[6]), "+v"(e1[7]) : [f] "s"(full), [n] "i"(NC4) : "memory", "scc");
      else if constexpr (NC4 == 4)
        asm volatile(GNX_WAIT_ROWS : "+v"(e1[0]), "+v"(e1[1]), "+v"(e1[2]), "+v"(e1[3]) : [f] "s"(full), [n] "i"(NC4) : "memory", "scc");
      else
        asm volatile(GNX_WAIT_ROWS : "+v"(e1[0]), "+v"(e1[1]) : [f] "s"(full), [n] "i"(NC4) : "memory", "scc");
#undef GNX_WAIT_ROWS
    }
#ifdef GNX_WIDE_STAMPS_BUILD
    const unsigned long long te2 = clock64();
    t_egroups += te2 - te1;
#endif
    if (VEC4 && a.agg_out) {
      // ---- per-destination sums of this 64-row pass (rows are dst-sorted: a destination is a contiguous run) ----
      lds_barrier();             // every finished value is back in sC (s_seg was filled before this pass's second barrier)
      const int n_seg = s_seg[65];
      const int q4 = tid % (BN / 4), grp = tid / (BN / 4);
      const int n = n0 + 4 * q4;
      if (n < a.OUT) {
        float* agg = a.agg_out + r * a.agg_rep_stride + (size_t)agg_row0[pass] * a.OUT + n;
        for (int sgm = grp; sgm < n_seg; sgm += NG) {
          const int r0 = s_seg[sgm], r1 = s_seg[sgm + 1];
          float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int rr = r0; rr < r1; ++rr) {  // (four clamped rows per step, requested together: measured no faster)
            const float4 u = *reinterpret_cast<const float4*>(sC + rr * LDC + 4 * q4);
            t4.x += u.x; t4.y += u.y; t4.z += u.z; t4.w += u.w;
          }
          *reinterpret_cast<float4*>(agg + (size_t)sgm * a.OUT) = t4;
        }
      }
    }
#ifdef GNX_WIDE_STAMPS_BUILD
    t_eagg += clock64() - te2;  // barrier + per-destination sums of this pass (the column-sum tail is the epilogue's remainder)
#endif
  }
  if (a.colsum) {  // fixed-order reduction over the NG row groups that share a column quad
    lds_barrier();
    float* s_cs = sC;  // [NG][BN]
    {
      const int c4 = tid % (BN / 4), grp = tid / (BN / 4);
      *reinterpret_cast<float4*>(s_cs + grp * BN + 4 * c4) = cs4;
    }
    lds_barrier();
    int tc = tid;  // (re-derived: an LDS address kept since the prologue is parked in scratch memory, and its reload here waits for every store of the tile)
    asm volatile("" : "+v"(tc));
    if (tc < BN && n0 + tc < a.OUT) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < NG; ++w) sum += s_cs[w * BN + tc];
      a.colsum[r * a.colsum_rep_stride + (size_t)tile_id * a.OUT + n0 + tc] = sum;
    }
  }
#ifdef GNX_WIDE_STAMPS_BUILD
  if (a.stamps && tid == 0 && ctile == 0 && blockIdx.z == 0) {
    unsigned long long* o = a.stamps + (size_t)tile_id * 8;
    o[0] = st[1] - st[0]; o[1] = t_sync; o[2] = t_mfma; o[3] = clock64() - st[2]; o[4] = clock64() - st[0];
    o[5] = t_estage | (t_egroups << 32); o[6] = t_issue | (t_eagg << 32);  // epilogue: staging (barrier, acc -> LDS, barrier) | operand groups; loads issue time (part of o[2])
    o[7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);  // XCC_ID, HW_ID
  }
#endif
}

#undef GNX_LOADER_COORDS
#undef ra
#undef rb
#undef ex
#undef ey

// bias'[r][g][n] = b[n] + sum_k W[(w_row0 + k)*OUT + n] * gf[r][g][k]      (the gf segment of edgefninput.jl:6 /
// nodefninput.jl:5: constant per graph, so it is a rank-1 fold instead of K more columns of every row's GEMM)
__global__ void k_fold_bias(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ gf, int dg,
                            int w_row0, int OUT, int G, float* __restrict__ out) {
  const int g = blockIdx.x;
  const size_t r = blockIdx.y;
  const float* gfg = gf + (r * G + g) * (size_t)dg;
  for (int n = threadIdx.x; n < OUT; n += blockDim.x) {
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    for (int k = 0; k < dg; k += 8) {
      float w[8], x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {  // clamped, unconditional loads: 16 in flight
        const int kk = min(k + u, dg - 1);
        w[u] = W[(size_t)(w_row0 + kk) * OUT + n];
        x[u] = gfg[kk];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] = fmaf(w[u], k + u < dg ? x[u] : 0.f, acc[u]);
    }
    const float b = bias ? bias[n] : 0.f;
    out[(r * G + g) * (size_t)OUT + n] = b + (((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7])));
  }
}

// ---- graph update for the wide path: two-stage fixed-order reduction of the per-tile column sums ----
// stage 1: out2[r][g][s][c] = sum over the s-th slice of graph g's tile rows of in[r][tile][c]
struct ColsumJob { const float* in; size_t in_rep_stride; const int* tile_off; int C; float* out2; };
// (the edges' and the nodes' sums in ONE launch: blockIdx.x = 2 * graph + job; every launch of this size is ~5 us of latency)
__global__ void k_colsum_slices(ColsumJob j0, ColsumJob j1, int S, int G) {
  const ColsumJob j = (blockIdx.x & 1) ? j1 : j0;  // (the job rides in grid.x: grid.z is the replica, up to 65535)
  if (!j.in || j.C == 0) return;
  const int g = blockIdx.x >> 1, s = blockIdx.y;
  const size_t r = blockIdx.z;
  const int C = j.C;
  const int t0 = j.tile_off[g], t1 = j.tile_off[g + 1];
  const int per = (t1 - t0 + S - 1) / S;
  const int a0 = t0 + s * per, a1 = min(a0 + per, t1);
  const float* base = j.in + r * j.in_rep_stride;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] = 0.f;
    // loads are made unconditional by clamping the row (a predicated load cannot be hoisted over its branch, which
    // serialises one memory round trip per row); the clamped duplicates are discarded by the select
    for (int ti = a0; ti < a1; ti += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = base[(size_t)min(ti + u, a1 - 1) * C + c];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u] += ti + u < a1 ? v[u] : 0.f;
    }
#pragma unroll
    for (int w = 8; w > 0; w >>= 1)
#pragma unroll
      for (int u = 0; u < w; ++u) acc[u] += acc[u + w];
    j.out2[((r * G + g) * S + s) * (size_t)C + c] = acc[0];
  }
}

// stage 2: gf'[g] = act(Wg * [sum_e ef' ; sum_n nf' ; gf_g] + bg).  Latency-bound, so the work is laid out for ONE
// round trip per phase: thread k owns input feature k — it reads 32 weights W[k][j0..j0+32) (32 independent loads) and
// contributes W[k][j]*x[k]; the 256 contributions per output are then added in a fixed order from LDS.
__global__ __launch_bounds__(256) void k_graph_final(const float* pe2, const float* pn2, int S, BlockArgs a) {
  extern __shared__ float s_gf[];
  const int g = blockIdx.x, tid = threadIdx.x;
  const size_t r = blockIdx.y;
  const int oe = a.oe, on = a.on, C = oe + on, K = C + a.dg, og = a.og;
  float* s_x = s_gf;                 // [K]
  float* s_p = s_gf + K + 4;         // [256][33]
  for (int c = tid; c < C; c += 256) {
    const float* p = c < oe ? pe2 + ((r * a.G + g) * S) * (size_t)oe + c : pn2 + ((r * a.G + g) * S) * (size_t)on + (c - oe);
    const int stride = c < oe ? oe : on;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    for (int i = 0; i < S; i += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)min(i + u, S - 1) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += i + u < S ? v[u] : 0.f;
    }
    s_x[c] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  }
  const float* gf = a.gf ? a.gf + (r * (size_t)a.G + g) * a.dg : nullptr;
  for (int k = tid; k < a.dg; k += 256) s_x[C + k] = gf[k];
  __syncthreads();
  float* out = a.gf_out + (r * (size_t)a.G + g) * og;
  for (int j0 = 0; j0 < og; j0 += 32) {
    float part[32];
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) part[jj] = 0.f;
    for (int k = tid; k < K; k += 256) {
      float w[32];
#pragma unroll
      for (int jj = 0; jj < 32; ++jj) w[jj] = j0 + jj < og ? a.Wg[(size_t)k * og + j0 + jj] : 0.f;
      const float xk = s_x[k];
#pragma unroll
      for (int jj = 0; jj < 32; ++jj) part[jj] = fmaf(w[jj], xk, part[jj]);
    }
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) s_p[tid * 33 + jj] = part[jj];
    __syncthreads();
    if (tid < 32 && j0 + tid < og) {
      float y = a.bg ? a.bg[j0 + tid] : 0.f;
      for (int t = 0; t < 256; ++t) y += s_p[t * 33 + tid];
      out[j0 + tid] = act_apply(y, a.act_g);
    }
    __syncthreads();
  }
}

// ---- few rows, wide layers (graph-level Dense of a small batch: M = R*G <= 8 rows, K and N in the hundreds) ----
// y[m][n] = act(b[n] + sum_k x[m][k] * W[(w_row0 + k) * ldw + n]).  The work is a handful of GEMVs: what matters is the
// number of dependent memory round trips.  1024 threads = 64 output lanes x 16 K-slices, every thread keeps 8 weight loads
// in flight (coalesced 256-B rows), the 16 slices are added in a fixed order from LDS.
struct SkinnyJob { const float* x; int ldx, M, K; const float* W; int w_row0, ldw; const float* bias; int N, act; float* y; int ldy; };
// (two jobs per launch — blockIdx.y: the gf folds of the edge and the node function are one launch)
__global__ __launch_bounds__(1024) void k_skinny_dense(SkinnyJob j0, SkinnyJob j1) {
  const SkinnyJob j = blockIdx.y ? j1 : j0;
  if (!j.y || (int)blockIdx.x * 64 >= j.N) return;  // (whole workgroup: before any barrier)
  const float* __restrict__ x = j.x;
  const float* __restrict__ W = j.W;
  const float* __restrict__ bias = j.bias;
  float* __restrict__ y = j.y;
  const int ldx = j.ldx, M = j.M, K = j.K, w_row0 = j.w_row0, ldw = j.ldw, N = j.N, act = j.act, ldy = j.ldy;
  extern __shared__ float s_sk[];  // [M][K] inputs, reused as [16][8][64] partial sums
  const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
  const int n = blockIdx.x * 64 + lane;
  for (int i = tid; i < M * K; i += 1024) s_sk[i] = x[(size_t)(i / K) * ldx + i % K];
  __syncthreads();
  const int per = (K + 15) / 16, k0 = ks * per, k1 = min(k0 + per, K);
  float acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = 0.f;
  const int nc = n < N ? n : N - 1;
  for (int k = k0; k < k1; k += 8) {
    float w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) w[u] = W[(size_t)(w_row0 + min(k + u, k1 - 1)) * ldw + nc];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (k + u < k1) {
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = fmaf(w[u], m < M ? s_sk[m * K + k + u] : 0.f, acc[m]);
      }
    }
  }
  __syncthreads();  // everyone is done reading the inputs: the buffer becomes the partial-sum table
#pragma unroll
  for (int m = 0; m < 8; ++m) s_sk[(ks * 8 + m) * 64 + lane] = acc[m];
  __syncthreads();
  if (tid < 64 * 8) {
    const int m = tid >> 6, l = tid & 63, nn = blockIdx.x * 64 + l;
    if (m < M && nn < N) {
      float v = bias ? bias[nn] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) v += s_sk[(q * 8 + m) * 64 + l];
      y[(size_t)m * ldy + nn] = act_apply(v, act);
    }
  }
}
static bool skinny_ok(int64_t M, int K, int N) { return M >= 1 && M <= 8 && (size_t)K * N >= 4096 && K >= 16; }
static int32_t launch_skinny2(const SkinnyJob& j0, const SkinnyJob& j1, hipStream_t s) {
  const bool two = j1.y != nullptr;
  const size_t lds = sizeof(float) * std::max<size_t>(std::max((size_t)j0.M * j0.K, two ? (size_t)j1.M * j1.K : 0), 16 * 8 * 64);
  const unsigned gx = (unsigned)((std::max(j0.N, two ? j1.N : 0) + 63) / 64);
  GNX_LAUNCH(k_skinny_dense, dim3(gx, two ? 2u : 1u), dim3(1024), lds, s, j0, j1);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}
static int32_t launch_skinny(const float* x, int ldx, int M, int K, const float* W, int w_row0, int ldw, const float* bias, int N, int act, float* y,
                             int ldy, hipStream_t s) {
  return launch_skinny2(SkinnyJob{x, ldx, M, K, W, w_row0, ldw, bias, N, act, y, ldy}, SkinnyJob{}, s);
}

// Xg[r*G + g][:] = [sum_e ef' ; sum_n nf' ; gf_g] from the stage-1 slices (the input of the graph function)
__global__ __launch_bounds__(256) void k_graph_x(const float* pe2, const float* pn2, int S, BlockArgs a, float* __restrict__ xg) {
  const int g = blockIdx.x, tid = threadIdx.x;
  const size_t r = blockIdx.y;
  const int oe = a.oe, on = a.on, C = oe + on, K = C + a.dg;
  float* out = xg + (r * a.G + g) * (size_t)K;
  for (int c = tid; c < C; c += 256) {
    const float* p = c < oe ? pe2 + ((r * a.G + g) * S) * (size_t)oe + c : pn2 + ((r * a.G + g) * S) * (size_t)on + (c - oe);
    const int stride = c < oe ? oe : on;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    for (int i = 0; i < S; i += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)min(i + u, S - 1) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += i + u < S ? v[u] : 0.f;
    }
    out[c] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  }
  const float* gf = a.gf ? a.gf + (r * (size_t)a.G + g) * a.dg : nullptr;
  for (int k = tid; k < a.dg; k += 256) out[C + k] = gf[k];
}

static int wide_slices(const gnx_graphs* h) {
  int64_t mx = 1;
  for (int64_t g = 0; g < h->G; ++g) mx = std::max<int64_t>(mx, h->h_etile_off[g + 1] - h->h_etile_off[g]);
  int S = (int)((mx + 63) / 64);
  return S < 1 ? 1 : (S > 64 ? 64 : S);
}

// the edge update's prepared weight block: three bf16 planes in 32-output slices (k_edge_x6_prep); k_edge_n's adds the source rows' block (K = 128 + 64)
static size_t x6_tab_bytes(int de, int oe) {
  const size_t x6 = sizeof(__bf16) * 3 * (size_t)de * (size_t)((oe + 31) / 32 * 32);
  if (de == 10 && oe == 128) return std::max(x6, edge_enc_scratch_bytes());  // (the encoder form's zero-padded K = 32)
  return (de == 128 && oe == 128) ? std::max(edge_x6_fold_scratch_bytes(), edge_n_scratch_bytes()) : x6;  // (the core's one-launch form: + We^T beta)
}

size_t wide_workspace_bytes(const gnx_graphs* h, const gnx_block_params* p, int64_t R) {
  const size_t per_tile = sizeof(float) * (size_t)R * ((size_t)h->n_etiles * p->oe + (size_t)h->n_ntiles * p->on);
  const size_t stage2 = sizeof(float) * (size_t)R * h->G * wide_slices(h) * (size_t)(p->oe + p->on);
  const size_t bias_g = sizeof(float) * (size_t)R * h->G * (size_t)(p->oe + p->on);
  const size_t proj = sizeof(float) * 2 * (size_t)R * h->N * (size_t)p->oe;  // node projections Ps, Pd
  const size_t xg = sizeof(float) * (size_t)R * h->G * (size_t)(p->oe + p->on + p->dg);  // graph-function input (small batches)
  const size_t agg = sizeof(float) * (size_t)R * (size_t)h->agg_rows_bound * (size_t)p->oe;  // per-destination partial sums of the edge GEMM (rows: an upper bound known without the tables)
  const size_t x6 = x6_tab_bytes(p->de, p->oe);
  const size_t x6p = proj_x6_scratch_bytes();  // the node projections' two weight blocks likewise (k_proj_x6_prep)
  return align_up(per_tile, 256) + align_up(stage2, 256) + align_up(bias_g, 256) + align_up(proj, 256) + align_up(xg, 256) + align_up(agg, 256) + align_up(x6, 256) + align_up(x6p, 256) + 512;
}

static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// W, the output and the epilogue operands can be accessed as 16-B quads
static bool out_vec(const WideArgs& w) {
  return w.OUT % 4 == 0 && (w.ldw == 0 || w.ldw % 4 == 0) && al16(w.W) && al16(w.out) && w.out_rep_stride % 4 == 0 && al16(w.gadd_a) && al16(w.gadd_b) &&
         w.gadd_rep_stride % 4 == 0 && al16(w.add1) && al16(w.add2) && al16(w.gmul) && al16(w.W2) && al16(w.out2);
}

template <int BN>
static int32_t launch_gemm(const WideArgs& w, unsigned n_tiles, int64_t R, hipStream_t s, const char* name) {
  if (n_tiles == 0 || w.OUT == 0) return GNX_OK;
  // host-side operand check before any launch: a kernel fault can take the whole node down
  if (!w.tiles || !w.W || !w.out) return fail(GNX_ERR_INVALID_ARG, "k_rows_gemm: tiles / W / out is NULL");
  const bool vec4 = out_vec(w);
  WideArgs wa = w;
  int ld = 0;  // loader class (template parameter LD)
  bool has_ln = false;
  for (int i = 0; i < w.nseg; ++i) {
    WSeg& g = wa.seg[i];
    if (g.width > 0 && !g.base) return fail(GNX_ERR_INVALID_ARG, "k_rows_gemm: segment base is NULL");
    if ((g.mode == 1 && !w.idx_a) || (g.mode == 2 && !w.idx_b) || (g.mode == 3 && !w.cp) || (w.gadd_a && (!w.gadd_b || !w.idx_a || !w.idx_b)) ||
        (g.mode == 4 && (!w.node_agg_row || !w.node_agg_parts || !w.node_agg_chunk || !w.chunk_row0)) || (w.agg_out && (!w.idx_b || !w.chunk_row0)))
      return fail(GNX_ERR_INVALID_ARG, "k_rows_gemm: index array required by a segment mode is NULL");
    if (g.mode < 0 || g.mode > 4) return fail(GNX_ERR_INVALID_ARG, "k_rows_gemm: segment mode");
    g.vec = g.width % 4 == 0 && al16(g.base) && g.rep_stride % 4 == 0;
    if (g.mode == 4 && !g.vec) return fail(GNX_ERR_INVALID_ARG, "k_rows_gemm: the partial-sum table must be readable as quads");
    if (g.ln && (!w.ln_stats || !w.ln_g || !w.ln_b || g.mode != 0 || !g.vec || g.width > 256 || !al16(w.ln_g) || !al16(w.ln_b) || ((uintptr_t)w.ln_stats & 7)))
      return fail(GNX_ERR_INVALID_ARG, "k_rows_gemm: LayerNorm on load needs a quad-row mode-0 segment of width <= 256");
    if (g.ln) { has_ln = true; wa.ln_width = g.width; }
    if (g.width > 0) ld = std::max(ld, !g.vec ? 2 : (g.mode >= 3 ? 1 : 0));
  }
  if (w.agg_out && (!vec4 || !al16(w.agg_out) || w.agg_rep_stride % 4 != 0)) return fail(GNX_ERR_INVALID_ARG, "k_rows_gemm: fused aggregation needs quad outputs");
  // narrow segments (modes 0-2, consecutive rows of W) that need fewer K chunks side by side than one after the other
  // (the encoder's [ef(10) | nf_src(5) | nf_dst(5)]: one chunk of 32 instead of three): packed into ONE range, element loads
  wa.npk = 0;
  {
    int live[3], nl_ = 0, ktot = 0, chunks = 0;
    bool can = true;
    for (int i = 0; i < w.nseg; ++i) {
      if (w.seg[i].width == 0) continue;
      can &= w.seg[i].mode <= 2 && (nl_ == 0 || w.seg[i].w_row0 == w.seg[live[nl_ - 1]].w_row0 + w.seg[live[nl_ - 1]].width);
      live[nl_++] = i;
      ktot += w.seg[i].width;
      chunks += (w.seg[i].width + 31) / 32;
    }
    if (!has_ln && can && nl_ >= 2 && (ktot + 31) / 32 < chunks) {
      for (int i = 0; i < nl_; ++i) wa.pk[i] = wa.seg[live[i]];
      wa.npk = nl_;
      wa.seg[0] = WSeg{wa.pk[0].base, 0, ktot, 5, wa.pk[0].w_row0, 0};
      wa.nseg = 1;
      ld = 2;
    }
  }
  if (has_ln && (!vec4 || ld > 1 || wa.npk)) return fail(GNX_ERR_INVALID_ARG, "k_rows_gemm: LayerNorm on load needs quad outputs and quad-row segments");
  if (!has_ln) wa.ln_stats = nullptr;
  ProfScope ps(name, s);
  wa.n_rtiles = (int)n_tiles;
  wa.n_ctiles = (w.OUT + BN - 1) / BN * (w.W2 ? 2 : 1);
  if (w.W2 && (!w.out2 || w.colsum || w.agg_out)) return fail(GNX_ERR_INVALID_ARG, "k_rows_gemm: second weight block needs its own output and no column sums");
  const unsigned gx = wa.n_ctiles > 1 ? (n_tiles + 7) / 8 * 8 * (unsigned)wa.n_ctiles : n_tiles;
  const dim3 grid(gx, 1, (unsigned)R);
  // K chunk 32: measured against 64 (fewer barriers but 2 instead of 3 waves/SIMD): 466 vs 616 us on the edge GEMM
#ifdef GNX_WIDE_STAMPS_BUILD  // diagnostic build (tools/build_stamps.sh) + GNX_WIDE_STAMPS=1: per-phase shader clocks of every launch
  static unsigned long long* d_stamps = nullptr;
  static size_t stamps_cap = 0;
  static const bool want_stamps = getenv("GNX_WIDE_STAMPS") != nullptr;
  if (want_stamps && stamps_cap < n_tiles) {
    if (d_stamps) (void)hipFree(d_stamps);
    stamps_cap = (size_t)n_tiles * 2;
    (void)hipMalloc((void**)&d_stamps, stamps_cap * 8 * sizeof(unsigned long long));
  }
  if (want_stamps) { (void)hipMemsetAsync(d_stamps, 0, (size_t)n_tiles * 8 * sizeof(unsigned long long), s); wa.stamps = d_stamps; }
#endif
  const float* a1 = w.add1 ? w.add1 : w.add2;  // a lone add2 is an add1
  const float* a2 = w.add1 ? w.add2 : nullptr;
  wa.add1 = a1; wa.add2 = a2;
  if (w.gadd_a && (a1 || w.gmul)) return fail(GNX_ERR_INVALID_ARG, "k_rows_gemm: gathered addends cannot be combined with residual / gmul operands");
  if (w.gmul && a2) return fail(GNX_ERR_INVALID_ARG, "k_rows_gemm: gmul takes at most one residual operand");
  wa.epi = w.gadd_a ? EPI_GADD : (w.gmul ? (a1 ? EPI_GMUL_ADD : EPI_GMUL) : (a2 ? EPI_ADD12 : (a1 ? EPI_ADD1 : EPI_NONE)));
  int nl = wa.epi == EPI_NONE ? 0 : ((wa.epi == EPI_ADD1 || wa.epi == EPI_GMUL) ? 1 : 2);
  // gathered addends with quad outputs and the lean loader, on a batch whose edge tiles' destinations are short runs of rows: the
  // destination rows go through LDS (NL = 3).  GNX_GEMM_PD_LDS=0 keeps both tables as operand streams (NL = 2) for A/B runs.
  static const bool pd_lds_env = !(getenv("GNX_GEMM_PD_LDS") && atoi(getenv("GNX_GEMM_PD_LDS")) == 0);
  if (wa.epi == EPI_GADD && vec4 && ld == 0 && w.pd_lds && pd_lds_env && w.OUT >= 4 && wa.nseg <= 1 && (wa.nseg == 0 || wa.seg[0].mode == 0)) nl = 3; else wa.pd_lds = 0;
  static const int stagger_env = getenv("GNX_GEMM_STAGGER") ? atoi(getenv("GNX_GEMM_STAGGER")) : 0;
  wa.stagger = (n_tiles >= 2048 && wa.n_ctiles == 1) ? stagger_env : 0;
  const bool trans = w.act > 1;
  // instantiations: quad outputs with the lean loader (every NL), quad outputs with the full loader and no operands (node update,
  // encoder), element outputs with the full loader (every NL; also takes the rare quad-output + full-loader + operands launches)
#define GNX_GEMM_LAUNCH(V, N, T, F) do { if (wa.fp32) GNX_LAUNCH((k_rows_gemm<BN, V, 32, N, T, F, false>), grid, dim3(WaveLayout<BN>::WT), 0, s, wa); \
                                         else GNX_LAUNCH((k_rows_gemm<BN, V, 32, N, T, F, true>), grid, dim3(WaveLayout<BN>::WT), 0, s, wa); } while (0)
#define GNX_GEMM_LAUNCH_N(V, T, F) do { if (nl == 0) GNX_GEMM_LAUNCH(V, 0, T, F); else if (nl == 1) GNX_GEMM_LAUNCH(V, 1, T, F); else GNX_GEMM_LAUNCH(V, 2, T, F); } while (0)
  if (vec4 && ld == 0 && nl == 3) { if (trans) GNX_GEMM_LAUNCH(true, 3, true, 0); else GNX_GEMM_LAUNCH(true, 3, false, 0); }
  else if (vec4 && ld == 0) { if (trans) GNX_GEMM_LAUNCH_N(true, true, 0); else GNX_GEMM_LAUNCH_N(true, false, 0); }
  else if (vec4 && nl == 0 && ld == 1) { if (trans) GNX_GEMM_LAUNCH(true, 0, true, 1); else GNX_GEMM_LAUNCH(true, 0, false, 1); }
  else if (vec4 && nl == 0) { if (trans) GNX_GEMM_LAUNCH(true, 0, true, 2); else GNX_GEMM_LAUNCH(true, 0, false, 2); }
  else {
    if (w.agg_out) return fail(GNX_ERR_INVALID_ARG, "k_rows_gemm: fused aggregation with epilogue operands needs the lean loader");
    if (trans) GNX_GEMM_LAUNCH_N(false, true, 2); else GNX_GEMM_LAUNCH_N(false, false, 2);
  }
#undef GNX_GEMM_LAUNCH_N
#undef GNX_GEMM_LAUNCH
#ifdef GNX_WIDE_STAMPS_BUILD
  if (want_stamps) {
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> hs((size_t)n_tiles * 8);
    (void)hipMemcpy(hs.data(), d_stamps, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (const char* dump = getenv("GNX_WIDE_STAMPS_DUMP")) {  // raw [tile][8] table, for offline timelines
      char path[512];
      snprintf(path, sizeof path, "%s_%s.bin", dump, name);  // (the last launch of each name wins)
      if (FILE* f = fopen(path, "wb")) { fwrite(hs.data(), 8, hs.size(), f); fclose(f); }
    }
    double m[7] = {0, 0, 0, 0, 0, 0, 0}, e_stage = 0, e_groups = 0, e_agg = 0;
    for (size_t i = 0; i < n_tiles; ++i) {
      for (int j = 0; j < 6; ++j) m[j] += (double)hs[i * 8 + j];
      m[6] += (double)(hs[i * 8 + 6] & 0xffffffffull); e_agg += (double)(hs[i * 8 + 6] >> 32);
      e_stage += (double)(hs[i * 8 + 5] & 0xffffffffull); e_groups += (double)(hs[i * 8 + 5] >> 32);
    }
    fprintf(stderr, "[gnx stamps] %s BN=%d tiles=%u ctiles=%d: per tile (shader clocks, wave 0 of column tile 0): prologue %.0f  sync+store %.0f  mfma-loop %.0f (of which issuing the next chunk's loads %.0f)  epilogue %.0f (staging %.0f, operand groups + stores %.0f, barrier + per-destination sums %.0f, rest: column sums)  total %.0f\n",
            name, BN, n_tiles, wa.n_ctiles, m[0] / n_tiles, m[1] / n_tiles, m[2] / n_tiles, m[6] / n_tiles, m[3] / n_tiles, e_stage / n_tiles, e_groups / n_tiles, e_agg / n_tiles, m[4] / n_tiles);
  }
#endif
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

static int32_t launch_gemm_any(const WideArgs& w, unsigned n_tiles, int64_t R, hipStream_t s, const char* name) {
  int bn = w.OUT > 64 ? 128 : (w.OUT > 32 ? 64 : 32);  // 128-wide tiles beat 64-wide ones for OUT = 128 on a full GPU
  // A launch that cannot fill the 256 CUs (small batches: the reference's sort example has 4 graphs) is pure latency: every
  // workgroup walks its K chunks alone, paying a 128-wide tile's MFMA time per chunk for a handful of rows.  Narrower column
  // tiles give more workgroups and 2-4x less matrix-core time per chunk.
  while (bn > 32 && (size_t)n_tiles * ((w.OUT + bn - 1) / bn) * (size_t)R < 256) bn >>= 1;
  static const int bn_env = getenv("GNX_GEMM_BN") ? atoi(getenv("GNX_GEMM_BN")) : 0;  // experiment switch: cap the column tile (64 / 32)
  if (bn_env == 64 || bn_env == 32) bn = std::min(bn, bn_env);
  if (bn == 128) return launch_gemm<128>(w, n_tiles, R, s, name);
  if (bn == 64) return launch_gemm<64>(w, n_tiles, R, s, name);
  return launch_gemm<32>(w, n_tiles, R, s, name);
}

// y[rows, OUT] = act(A[rows, K] * W + b) (+ add1 + add2) over ALL rows of one entity type (0 edges, 1 nodes, 2 graphs):
// the position-wise Dense layers of GNFeedForward (gnfeedforward.jl:27-40) on the matrix cores.
int32_t launch_dense_rows(const gnx_graphs* h, int entity, const float* A, int K, const gnx_dense& d, int OUT, const float* add1,
                          const float* add2, float* out, int64_t R, hipStream_t s, const char* name) {
  const size_t nrows = entity == 0 ? (size_t)h->E : (entity == 1 ? (size_t)h->N : (size_t)h->G);
  if (nrows == 0 || OUT == 0) return GNX_OK;
  if (int32_t rcw = gnx_ensure_wide_tables(h, s)) return rcw;
  WideArgs w{};
  w.tiles = entity == 0 ? h->d_etiles : (entity == 1 ? h->d_ntiles : h->d_gtiles);
  w.row_kind = entity == 0 ? 0 : 1;
  w.seg[0] = WSeg{A, nrows * (size_t)K, K, 0, 0};
  w.nseg = 1;
  w.W = d.weight; w.bias = d.bias; w.OUT = OUT; w.act = d.act;
  w.bias_g = nullptr; w.n_graphs = (int)h->G;
  w.out = out; w.out_rep_stride = nrows * (size_t)OUT;
  w.colsum = nullptr;
  w.add1 = add1; w.add2 = add2;
  w.fp32 = form(GNX_FLAG_FFN_FP32);
  const unsigned n_tiles = (unsigned)(entity == 0 ? h->n_etiles : (entity == 1 ? h->n_ntiles : h->n_gtiles));
  return launch_gemm_any(w, n_tiles, R, s, name);
}

// out[rows, OUT] = A[rows, K] * B over all rows of one entity type, B = [K][OUT] block of a matrix with row distance ldw
// (no bias, no activation): the dX = delta * W^T products of the backward pass (gnx_backward.hip), B = transposed weights.
// Optional epilogue: out *= act'(gmul) (gmul: stored forward output with the layout of out), and per-tile column sums of the
// result into tile_colsum [R][n_tiles][OUT] (n_tiles returned through *n_tiles_out) for a bias gradient.
int32_t launch_rows_matmul(const gnx_graphs* h, int entity, const float* A, int K, const float* B, int ldw, int OUT, float* out, int64_t R,
                           hipStream_t s, const char* name, const float* gmul, int gmul_act, float* tile_colsum, int* n_tiles_out, const float* add1) {
  const size_t nrows = entity == 0 ? (size_t)h->E : (entity == 1 ? (size_t)h->N : (size_t)h->G);
  if (nrows == 0 || OUT == 0 || K == 0) return GNX_OK;
  if (int32_t rcw = gnx_ensure_wide_tables(h, s)) return rcw;
  WideArgs w{};
  w.tiles = entity == 0 ? h->d_etiles : (entity == 1 ? h->d_ntiles : h->d_gtiles);
  w.row_kind = entity == 0 ? 0 : 1;
  w.seg[0] = WSeg{A, nrows * (size_t)K, K, 0, 0};
  w.nseg = 1;
  w.W = B; w.ldw = ldw; w.bias = nullptr; w.OUT = OUT; w.act = GNX_ACT_IDENTITY;
  w.n_graphs = (int)h->G;
  w.out = out; w.out_rep_stride = nrows * (size_t)OUT;
  const unsigned n_tiles = (unsigned)(entity == 0 ? h->n_etiles : (entity == 1 ? h->n_ntiles : h->n_gtiles));
  w.gmul = gmul; w.gmul_act = gmul_act;
  w.add1 = add1;  // optional residual with the layout of out (may alias it)
  w.colsum = tile_colsum; w.colsum_rep_stride = (size_t)n_tiles * OUT;
  w.fp32 = form(GNX_FLAG_FP32_MFMA);  // (any of the two bits; the backward's entry points carry no flags: the process-wide defaults decide)
  if (n_tiles_out) *n_tiles_out = (int)n_tiles;
  return launch_gemm_any(w, n_tiles, R, s, name);
}

// does launch_block_wide take this block (else 1 = "not applicable": the caller's next path), and in which form
static bool wide_applies(const gnx_graphs* h, const BlockArgs& a, bool* project_out) {
  static const bool off = getenv("GNX_NO_WIDE") != nullptr;
  if (off) return false;
  const int ke = a.de + 2 * a.dn + a.dg, kn = a.oe + a.dn + a.dg;
  // the matrix-core path pays when the update is a real GEMM; tiny widths stay on the other paths
  if (std::max(std::max(ke, a.oe), std::max(kn, a.on)) < 32) return false;
  if (a.E == 0 && a.oe > 0) return false;
  // gathered tables (node features, node projections, partial sums) are addressed with 32-bit element offsets
  if ((size_t)h->N * (size_t)std::max(std::max(a.oe, a.dn), 1) * sizeof(float) >= (1ull << 32)) return false;
  // Node-projection form (edgefninput.jl:2-7 regrouped): W*[ef; nf_s; nf_d; gf] = We_e*ef + (We_s*nf)[src] + (We_d*nf + b')[dst].
  // The 2*dn columns of nf are multiplied once per NODE (two small GEMMs) instead of once per EDGE; the edge GEMM keeps
  // K = de and gathers two projected rows in its epilogue.  Same mathematics, different (still fixed) summation order.
  static const bool no_project = getenv("GNX_NO_PROJECT") != nullptr;
  *project_out = !no_project && a.oe > 0 && a.dn >= 16 && a.E >= 2 * (int64_t)a.N;
  return true;
}

// Workspace queries (outside any capture) build the handle's matrix-core tables when a forward with these widths can read them: the wide
// block itself, or a row-wise Dense on the matrix cores (a core's FeedForward from width 32, a Chain's further layers: `rows_gemm`).
void warm_block_wide(const gnx_graphs* h, const gnx_block_params* p, bool rows_gemm) {
  bool need = rows_gemm;
  if (!need && p) {
    BlockArgs a{};
    a.de = p->de; a.dn = p->dn; a.dg = p->dg; a.oe = p->oe; a.on = p->on; a.og = p->og;
    a.N = (int)h->N; a.E = (int)h->E; a.G = (int)h->G;
    bool project = false;
    need = wide_applies(h, a, &project);
  }
  if (need) (void)gnx_ensure_wide_tables(h);  // a failure is not latched: the forward reports it
}

// The wide path can normalise ef and nf as it loads them (BlockArgs::ln_stats): every launch that reads them is then a quad-row,
// quad-output GEMM with the features as mode-0 segments — the projected form with widths that are multiples of 4.
bool block_wide_ln_applies(const gnx_graphs* h, const BlockArgs& a) {
  bool project = false;
  if (!wide_applies(h, a, &project) || !project) return false;
  return a.de % 4 == 0 && a.dn % 4 == 0 && a.oe % 4 == 0 && a.on % 4 == 0 && a.de <= 256 && a.dn <= 256 && a.on > 0 && al16(a.ef) && al16(a.nf) &&
         al16(a.We) && al16(a.Wn) && al16(a.ef_out) && al16(a.nf_out) && al16(a.ln_g[0]) && al16(a.ln_b[0]) && al16(a.ln_g[1]) && al16(a.ln_b[1]);
}

// does the block's edge update run as k_edge_x6 (gnx_edge_x6.hip)?  One predicate for the launcher and for callers that then leave the edge rows' statistics to that kernel
bool block_wide_edge_x6_applies(const gnx_graphs* h, const BlockArgs& a) {
  bool project = false;
  if (!wide_applies(h, a, &project) || !project) return false;
  static const bool no_agg_fuse = getenv("GNX_NO_AGG_FUSE") != nullptr;
  const bool edge_out_vec = a.oe % 4 == 0 && al16(a.We) && al16(a.ef_out) && ((size_t)a.E * a.oe) % 4 == 0 && ((size_t)a.N * a.oe) % 4 == 0;
  const bool ef_vec = a.de % 4 == 0 && al16(a.ef) && ((size_t)a.E * a.de) % 4 == 0;
  const bool agg_fuse = !no_agg_fuse && a.oe > 0 && a.on > 0 && edge_out_vec && ef_vec && h->agg_rows_bound > 0 && (size_t)h->agg_rows_bound * a.oe * sizeof(float) < (1ull << 32);
  return a.de == 128 && a.oe == 128 && a.dn > 0 && edge_out_vec && ef_vec && al16(a.ln_g[0]) && al16(a.ln_b[0]) && (agg_fuse || a.on == 0) && !form(GNX_FLAG_EDGE_FP32) &&
         (size_t)h->E >= 4096;
}

// ... and as k_edge_n (gnx_edge_n.hip: source rows gathered raw, register epilogue)?  Needs the six-term projection launch (64-wide nodes, from 4096
// nodes on) — launch_block_wide checks that part; a core whose edge FeedForward rides in the edge launch keeps round 4's form until that kernel is ported
bool block_wide_edge_n_applies(const gnx_graphs* h, const BlockArgs& a) {
  return edge_n_enabled() && a.dn == 64 && block_wide_edge_x6_applies(h, a) && al16(a.nf);
}

int32_t launch_block_wide(const gnx_graphs* h, const BlockArgs& a, int64_t R, hipStream_t s, int phase) {
  bool project = false;
  if (!wide_applies(h, a, &project)) return 1;
  if ((a.ln_stats[0] || a.ln_stats[1]) && !block_wide_ln_applies(h, a)) return fail(GNX_ERR_INVALID_ARG, "launch_block_wide: LayerNorm on load is not applicable to this block");
  if (int32_t rcw = gnx_ensure_wide_tables(h, s)) return rcw;
  const size_t n_et = (size_t)h->n_etiles, n_nt = (size_t)h->n_ntiles;
  // workspace layout inside a.partials (sized by gnx_block_workspace_bytes >= wide_workspace_bytes)
  float* pe = a.partials;
  float* pn = pe + (size_t)R * n_et * a.oe;
  const int S = wide_slices(h);
  float* stage2 = reinterpret_cast<float*>(reinterpret_cast<char*>(a.partials) +
                                           align_up(sizeof(float) * (size_t)R * (n_et * a.oe + n_nt * a.on), 256));
  float* pe2 = stage2;
  float* pn2 = pe2 + (size_t)R * h->G * S * a.oe;
  float* bias_e = reinterpret_cast<float*>(reinterpret_cast<char*>(stage2) + align_up(sizeof(float) * (size_t)R * h->G * S * (a.oe + a.on), 256));
  float* bias_n = bias_e + (size_t)R * h->G * a.oe;
  float* proj_s = reinterpret_cast<float*>(reinterpret_cast<char*>(bias_e) + align_up(sizeof(float) * (size_t)R * h->G * (a.oe + a.on), 256));
  float* proj_d = proj_s + (size_t)R * h->N * a.oe;
  float* agg_tab = reinterpret_cast<float*>(reinterpret_cast<char*>(proj_s) + align_up(sizeof(float) * 2 * (size_t)R * h->N * a.oe, 256) +
                                            align_up(sizeof(float) * (size_t)R * h->G * (size_t)(a.oe + a.on + a.dg), 256));
  void* x6_tab = reinterpret_cast<char*>(agg_tab) + align_up(sizeof(float) * (size_t)R * (size_t)h->agg_rows_bound * (size_t)a.oe, 256);
  void* x6p_tab = reinterpret_cast<char*>(x6_tab) + align_up(x6_tab_bytes(a.de, a.oe), 256);
  // edge -> node sums inside the edge GEMM's epilogue (the node GEMM then reads ~N rows instead of all E rows of ef')
  static const bool no_agg_fuse = getenv("GNX_NO_AGG_FUSE") != nullptr;
  // (needs quad outputs, and — with the projections' epilogue operands — an ef whose rows are quads: see launch_gemm's instantiations)
  const bool edge_out_vec = a.oe % 4 == 0 && al16(a.We) && al16(a.ef_out) && ((size_t)a.E * a.oe) % 4 == 0 && ((size_t)a.N * a.oe) % 4 == 0;
  const bool ef_vec = a.de % 4 == 0 && al16(a.ef) && ((size_t)a.E * a.de) % 4 == 0;
  const bool agg_fuse = !no_agg_fuse && (phase & 5) && a.oe > 0 && a.on > 0 && edge_out_vec && (!project || ef_vec) && h->n_agg_rows > 0 &&
                        (size_t)h->n_agg_rows * a.oe * sizeof(float) < (1ull << 32);
  int32_t rc = GNX_OK;
  const bool prep = (phase & 4) || ((phase & 1) && !(phase & 8));  // gf fold + node projections
  if (prep && a.dg > 0) {  // fold gf into per-graph biases (one tiny launch per update function)
    ProfScope ps("k_fold_bias", s);
    if (skinny_ok(R * a.G, a.dg, std::max(a.oe, a.on))) {  // a few graphs, wide layers: one round-trip-lean GEMV kernel per function
      const SkinnyJob je{a.gf, a.dg, (int)(R * a.G), a.dg, a.We, a.de + 2 * a.dn, a.oe, a.be, a.oe, GNX_ACT_IDENTITY, bias_e, a.oe};
      const SkinnyJob jn{a.gf, a.dg, (int)(R * a.G), a.dg, a.Wn, a.oe + a.dn, a.on, a.bn, a.on, GNX_ACT_IDENTITY, bias_n, a.on};
      if (a.oe > 0 && a.on > 0) { if ((rc = launch_skinny2(je, jn, s))) return rc; }  // both gf folds in one launch
      else if (a.oe > 0) { if ((rc = launch_skinny2(je, SkinnyJob{}, s))) return rc; }
      else if (a.on > 0) { if ((rc = launch_skinny2(jn, SkinnyJob{}, s))) return rc; }
    } else {
    if (a.oe > 0) GNX_LAUNCH(k_fold_bias, dim3((unsigned)a.G, (unsigned)R), dim3(128), 0, s, a.We, a.be, a.gf, a.dg, a.de + 2 * a.dn, a.oe, a.G, bias_e);
    if (a.on > 0) GNX_LAUNCH(k_fold_bias, dim3((unsigned)a.G, (unsigned)R), dim3(128), 0, s, a.Wn, a.bn, a.gf, a.dg, a.oe + a.dn, a.on, a.G, bias_n);
    }
    GNX_HIP(hipGetLastError());
  }
  const bool proj6 = project && proj_x6_applies(a.dn, a.oe, a.nf, a.We, proj_s, (size_t)a.N) && (!a.ln_stats[1] || (al16(a.ln_g[1]) && al16(a.ln_b[1]))) && al16(a.be);
  // k_edge_n's form (gnx_edge_n.hip): the source side multiplied per edge from the raw 64-wide row — the projection launch then produces the
  // destination table and, under a LayerNorm, the normalised rows the edges gather (in the source table's place).  One predicate for both launches
  // (they may run in different calls: phase 4 on the side stream, phase 1 on the caller's): it depends on the block, never on the phase.
  const bool edge_n = proj6 && block_wide_edge_n_applies(h, a);
  if (prep && proj6) {
    // 64 -> 2 x 128 from 4096 nodes on: both tables in one launch of k_proj_x6 (six bf16 matrix-core terms per fp32 product; gnx_edge_x6.hip)
    if ((rc = launch_proj_x6(h->d_ntiles, n_nt, a.nf, (size_t)a.N, a.ln_stats[1], a.ln_g[1], a.ln_b[1], a.We + (size_t)a.de * a.oe, a.We + (size_t)(a.de + a.dn) * a.oe, a.oe, a.be,
                             a.dg > 0 ? bias_e : nullptr, a.G, proj_s, proj_d, R, x6p_tab, s, edge_n, edge_n && a.ln_stats[1] ? proj_s : nullptr)))
      return rc;
  } else
  if (prep && project) {  // both projections in ONE launch (the second weight block of k_rows_gemm): nf is read once
    WideArgs w{};
    w.tiles = h->d_ntiles; w.row_kind = 1;
    w.seg[0] = WSeg{a.nf, (size_t)a.N * a.dn, a.dn, 0, 0};
    w.nseg = 1;
    if (a.ln_stats[1]) { w.seg[0].ln = 1; w.ln_stats = a.ln_stats[1]; w.ln_rep_stride = 2 * (size_t)a.N; w.ln_g = a.ln_g[1]; w.ln_b = a.ln_b[1]; }
    w.W = a.We + (size_t)a.de * a.oe;              // rows of the src segment
    w.W2 = a.We + (size_t)(a.de + a.dn) * a.oe;    // rows of the dst segment: bias (+ gf fold) rides on the dst projection
    w.bias2 = a.be; w.bias_g2 = a.dg > 0 ? bias_e : nullptr; w.n_graphs = a.G;
    w.OUT = a.oe; w.act = GNX_ACT_IDENTITY;
    w.out = proj_s; w.out2 = proj_d; w.out_rep_stride = (size_t)a.N * a.oe;
    w.fp32 = form(GNX_FLAG_EDGE_FP32 | GNX_FLAG_PROJ_FP32);
    if ((rc = launch_gemm_any(w, (unsigned)n_nt, R, s, "k_rows_gemm_proj"))) return rc;
  }
  // the projected edge update at 128 -> 128 as six bf16 matrix-core terms per fp32 product (gnx_edge_x6.hip)
  // (GNX_FLAG_EDGE_FP32: k_rows_gemm on the fp32 matrix instruction instead)
  const bool edge_x6 = (phase & 1) && block_wide_edge_x6_applies(h, a);
  if (a.ln_inline_e && (phase & 1) && !edge_x6) return fail(GNX_ERR_INVALID_ARG, "internal: edge statistics in the kernel asked of a block that does not run k_edge_x6");
  if (a.ffe_w1 && (phase & 1) && !(edge_x6 && a.ln_inline_e)) return fail(GNX_ERR_INVALID_ARG, "internal: the edge FeedForward inside the edge update asked of a block that does not run k_edge_x6 with its own statistics");
  // ... and at 128 -> at most 32 outputs without fused per-destination sums (config 4's decoder: 128 -> 3) its narrow form: one zero-padded slice
  const bool edge_x6n = (phase & 1) && !edge_x6 && project && a.de == 128 && a.oe >= 1 && a.oe <= 32 && a.dn > 0 && ef_vec && !agg_fuse &&
                        (!a.ln_stats[0] || (al16(a.ln_g[0]) && al16(a.ln_b[0]))) && !form(GNX_FLAG_EDGE_FP32) && !form(GNX_FLAG_EDGE_NARROW_FP32) &&
                        (size_t)h->E >= 4096;  // (GNX_FLAG_EDGE_NARROW_FP32: this form alone back on k_rows_gemm)
  // the ENCODER form: (10, 5, .) => 128 unprojected (README ex.3's / config 4's encoder) on the six-term scheme with the row's 20 inputs assembled in registers
  // (k_rows_gemm's packed element loader: 209 us at 1M edges against ~110 us of traffic; GNX_FLAG_EDGE_FP32 keeps it)
  const bool edge_enc = (phase & 1) && !project && a.de == 10 && a.dn == 5 && a.oe == 128 && !a.ln_stats[0] && !a.ln_inline_e && !a.ffe_w1 && edge_out_vec && (agg_fuse || a.on == 0) &&
                        al16(a.be) && !form(GNX_FLAG_EDGE_FP32) && (size_t)h->E >= 4096 && a.ef && a.nf;
  if (edge_x6n) {
    if ((rc = launch_edge_x6(h->d_etiles, n_et, a.ef, (size_t)a.E, a.ln_stats[0], a.ln_g[0], a.ln_b[0], a.We, a.oe, proj_s, proj_d, (size_t)a.N, a.rowval, h->d_edge_dst, a.act_e,
                             a.ef_out, a.og > 0 ? pe : nullptr, nullptr, 0, nullptr, R, x6_tab, s, false, 0.f, 0, a.oe)))
      return rc;
  } else
  if (edge_enc) {
    if ((rc = launch_edge_enc(h->d_etiles, n_et, a.ef, (size_t)a.E, a.nf, (size_t)a.N, a.We, a.oe, a.be, a.dg > 0 ? bias_e : nullptr, a.G, a.rowval, h->d_edge_dst, a.act_e, a.ef_out,
                              a.og > 0 ? pe : nullptr, agg_fuse ? agg_tab : nullptr, (size_t)h->n_agg_rows, h->d_chunk_row0, R, x6_tab, s)))
      return rc;
  } else
  if (edge_x6 && a.ffe_w1 && edge_n) {
    return fail(GNX_ERR_INVALID_ARG, "internal: the one-launch core form was asked of a block whose projections are k_edge_n's");
  } else
  if (edge_x6 && a.ffe_w1) {  // GNCore: edge update + edge FeedForward + residuals in one launch (edge form of k_ffn_x6); ef_out receives the CORE's edge output
    gnx_ffn ff{};
    ff.fc1.weight = a.ffe_w1; ff.fc1.bias = a.ffe_b1; ff.fc1.act = a.ffe_act1; ff.fc2.weight = a.ffe_w2; ff.fc2.bias = a.ffe_b2; ff.fc2.act = a.ffe_act2;
    const gnx_layernorm ln1{a.ln_g[0], a.ln_b[0]}, ln2{a.ffe_g2, a.ffe_be2};
    if ((rc = launch_core_edge_x6(h->d_etiles, n_et, a.ef, (size_t)a.E, &ln1, a.ln_eps, a.ln_mode, a.We, a.oe, proj_s, proj_d, (size_t)a.N, a.rowval, h->d_edge_dst, a.act_e,
                                  a.og > 0 ? pe : nullptr, agg_fuse ? agg_tab : nullptr, (size_t)h->n_agg_rows, h->d_chunk_row0, ff, &ln2, a.ef_out, R, x6_tab, a.ffe_scratch, s)))
      return rc;
  } else
  if (edge_x6 && edge_n) {
    if ((rc = launch_edge_n(h->d_etiles, n_et, a.ef, (size_t)a.E, a.ln_stats[0], a.ln_g[0], a.ln_b[0], a.We, a.oe, a.ln_stats[1] ? proj_s : a.nf, proj_d, (size_t)a.N, a.rowval,
                            h->d_edge_dst, a.act_e, a.ef_out, a.og > 0 ? pe : nullptr, agg_fuse ? agg_tab : nullptr, (size_t)h->n_agg_rows, h->d_chunk_row0, R, x6_tab, s,
                            a.ln_inline_e != 0, a.ln_eps, a.ln_mode)))
      return rc;
  } else
  if (edge_x6) {
    if ((rc = launch_edge_x6(h->d_etiles, n_et, a.ef, (size_t)a.E, a.ln_stats[0], a.ln_g[0], a.ln_b[0], a.We, a.oe, proj_s, proj_d, (size_t)a.N, a.rowval, h->d_edge_dst, a.act_e,
                             a.ef_out, a.og > 0 ? pe : nullptr, agg_fuse ? agg_tab : nullptr, (size_t)h->n_agg_rows, h->d_chunk_row0, R, x6_tab, s, a.ln_inline_e != 0, a.ln_eps,
                             a.ln_mode)))
      return rc;
  } else
  if ((phase & 1) && a.oe > 0) {
    WideArgs w{};
    w.tiles = h->d_etiles; w.row_kind = 0;
    int ns = 0;
    if (a.de) {
      w.seg[ns++] = WSeg{a.ef, (size_t)a.E * a.de, a.de, 0, 0};
      if (a.ln_stats[0]) { w.seg[ns - 1].ln = 1; w.ln_stats = a.ln_stats[0]; w.ln_rep_stride = 2 * (size_t)a.E; w.ln_g = a.ln_g[0]; w.ln_b = a.ln_b[0]; }
    }
    if (a.dn && !project) {
      w.seg[ns++] = WSeg{a.nf, (size_t)a.N * a.dn, a.dn, 1, a.de};
      w.seg[ns++] = WSeg{a.nf, (size_t)a.N * a.dn, a.dn, 2, a.de + a.dn};
    }
    w.nseg = ns;
    if (project) {
      w.gadd_a = proj_s; w.gadd_b = proj_d; w.gadd_rep_stride = (size_t)a.N * a.oe;
      w.pd_lds = h->n_etiles_wide_span == 0;  // every edge tile's destinations fit the kernel's LDS table
    }
    w.idx_a = a.rowval; w.idx_b = h->d_edge_dst; w.cp = a.colptr;
    w.W = a.We; w.bias = project ? nullptr : a.be; w.OUT = a.oe; w.act = a.act_e;
    w.bias_g = (!project && a.dg > 0) ? bias_e : nullptr; w.n_graphs = a.G;
    w.out = a.ef_out; w.out_rep_stride = (size_t)a.E * a.oe;
    w.colsum = a.og > 0 ? pe : nullptr; w.colsum_rep_stride = n_et * (size_t)a.oe;
    if (agg_fuse) { w.agg_out = agg_tab; w.agg_rep_stride = (size_t)h->n_agg_rows * a.oe; w.chunk_row0 = h->d_chunk_row0; }
    w.fp32 = form(GNX_FLAG_EDGE_FP32) || (a.oe <= 32 && form(GNX_FLAG_EDGE_NARROW_FP32));
    if ((rc = launch_gemm_any(w, (unsigned)n_et, R, s, "k_rows_gemm_edge"))) return rc;
  }
  // the node update at core widths on the six-term scheme (k_node_x6: the summed in-edge rows from the edge kernel's per-destination partial sums);
  // its weight planes: the layer's prepared ones, or made here in the projections' scratch (their planes were consumed by the projection launch)
  const bool node_x6 = (phase & 1) && a.on > 0 && agg_fuse && node_x6_applies(a.oe, a.dn, a.on, a.act_n, a.nf, a.Wn, a.nf_out, (size_t)a.N) &&
                       (!a.ln_stats[1] || (al16(a.ln_g[1]) && al16(a.ln_b[1]))) && al16(a.bn) && node_x6_scratch_bytes() <= proj_x6_scratch_bytes();
  if (node_x6) {
    if ((rc = launch_node_x6(h->d_ntiles, n_nt, a.nf, (size_t)a.N, a.ln_stats[1], a.ln_g[1], a.ln_b[1], agg_tab, (size_t)h->n_agg_rows, h->d_node_agg_row, h->d_node_agg_parts,
                             h->d_node_agg_chunk, h->d_chunk_row0, a.Wn, a.on, a.bn, a.dg > 0 ? bias_n : nullptr, a.G, a.act_n, a.nf_out, a.og > 0 ? pn : nullptr, R, x6p_tab, s)))
      return rc;
  } else
  if ((phase & 1) && a.on > 0) {
    WideArgs w{};
    w.tiles = h->d_ntiles; w.row_kind = 1;
    int ns = 0;
    if (a.oe && agg_fuse) {
      w.seg[ns++] = WSeg{agg_tab, (size_t)h->n_agg_rows * a.oe, a.oe, 4, 0};
      w.node_agg_row = h->d_node_agg_row; w.node_agg_parts = h->d_node_agg_parts; w.node_agg_chunk = h->d_node_agg_chunk; w.chunk_row0 = h->d_chunk_row0;
    } else if (a.oe) {
      w.seg[ns++] = WSeg{a.ef_out, (size_t)a.E * a.oe, a.oe, 3, 0};
    }
    if (a.dn) {
      w.seg[ns++] = WSeg{a.nf, (size_t)a.N * a.dn, a.dn, 0, a.oe};
      if (a.ln_stats[1]) { w.seg[ns - 1].ln = 1; w.ln_stats = a.ln_stats[1]; w.ln_rep_stride = 2 * (size_t)a.N; w.ln_g = a.ln_g[1]; w.ln_b = a.ln_b[1]; }
    }
    w.nseg = ns;
    w.idx_a = nullptr; w.idx_b = nullptr; w.cp = a.colptr;
    w.W = a.Wn; w.bias = a.bn; w.OUT = a.on; w.act = a.act_n;
    w.bias_g = a.dg > 0 ? bias_n : nullptr; w.n_graphs = a.G;
    w.out = a.nf_out; w.out_rep_stride = (size_t)a.N * a.on;
    w.colsum = a.og > 0 ? pn : nullptr; w.colsum_rep_stride = n_nt * (size_t)a.on;
    w.fp32 = form(GNX_FLAG_EDGE_FP32 | GNX_FLAG_PROJ_FP32);
    if ((rc = launch_gemm_any(w, (unsigned)n_nt, R, s, "k_rows_gemm_node"))) return rc;
  }
  if ((phase & 2) && a.og > 0) {
    ProfScope ps("k_graph_wide", s);
    if (a.oe > 0 || a.on > 0) {
      const ColsumJob je{a.oe > 0 ? pe : nullptr, n_et * (size_t)a.oe, h->d_etile_off, a.oe, pe2};
      const ColsumJob jn{a.on > 0 ? pn : nullptr, n_nt * (size_t)a.on, h->d_ntile_off, a.on, pn2};
      GNX_LAUNCH(k_colsum_slices, dim3(2u * (unsigned)a.G, (unsigned)S, (unsigned)R), dim3(128), 0, s, je, jn, S, a.G);
    }
    const int Kg = a.oe + a.on + a.dg;
    if (skinny_ok(R * a.G, Kg, a.og)) {  // small batch, wide layers: assemble Xg, then the round-trip-lean GEMV kernel
      float* xg = reinterpret_cast<float*>(reinterpret_cast<char*>(proj_s) + align_up(sizeof(float) * 2 * (size_t)R * h->N * a.oe, 256));
      GNX_LAUNCH(k_graph_x, dim3((unsigned)a.G, (unsigned)R), dim3(256), 0, s, pe2, pn2, S, a, xg);
      if ((rc = launch_skinny(xg, Kg, (int)(R * a.G), Kg, a.Wg, 0, a.og, a.bg, a.og, a.act_g, a.gf_out, a.og, s))) return rc;
    } else {
      const size_t lds = sizeof(float) * ((size_t)(a.oe + a.on + a.dg + 4) + 256 * 33 + 4);
      GNX_LAUNCH(k_graph_final, dim3((unsigned)a.G, (unsigned)R), dim3(256), lds, s, pe2, pn2, S, a);
    }
    GNX_HIP(hipGetLastError());
  }
  return GNX_OK;
}

}  // namespace gnx

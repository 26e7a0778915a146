// Register epilogue of the projected edge update in the NORMAL domain (rows x outputs) — shared by k_edge_n (gnx_edge_n.hip: a wide block's edge
// update) and the edge form of k_ffn_x6 (gnx_ffn_x6.hip: a GNCore's edge rows in one launch).
//
// The six-term kernels keep a wave's 32 rows on the lanes as operand fragments (lane (n, hi): 8 consecutive k of row n).  Used as the A operand
// — the prepared weight fragments (lane = output column) as B — v_mfma_f32_32x32x16_bf16 leaves the 32 x 32 block in the C/D layout with the
// OUTPUT on the lane and the ROWS in the registers:
//
//     lane (o, hi), register 4 g + j   <->   row 8 g + 4 hi + j of the wave, output o of the 32-output slice
//
// so everything the edge update owes the rest of the block is register arithmetic under wave-uniform control (edgefninput.jl:2-7, nodefninput.jl:
// 2-6, graphfninput.jl:2-6): the destination addend Pd[dst(row)][o] and the store of ef'[row][o] are dword accesses of 128 contiguous bytes per
// row, the per-destination sums (edges are dst-sorted: a destination is a contiguous run of rows) are a sequential pass over the 32 rows whose
// run boundaries are a scalar bit mask, and the tile's column sums are the sums of the run sums.  No LDS staging, no barrier per slice
// (round 4's form: two LDS passes and two workgroup barriers per 32-output slice — 39 % of a core tile's clocks for 11 % of its matrix work).
#pragma once
#include "gnx_device.h"

namespace gnx {

typedef float f32x16r __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8r __attribute__((ext_vector_type(8)));
typedef unsigned u32x4r __attribute__((ext_vector_type(4)));

// What a wave knows about the destination runs of its 64-row chunk (pass) of the tile: computed once per tile from the staged destinations.
// The wave's rows fall into PIECES in row order: piece 0 = the rows in front of the first run start when the wave's row 0 continues a run of the
// chunk's first wave (first_in), else the first run; every further run start opens the next piece; at most 32 pieces.  When the last piece
// continues into the chunk's second wave (last_out) it is numbered 31 instead — a fixed place for the deferred fix-up.
struct EdgeNRuns {
  bf16x8r m[2];      // the 0/1 piece-membership matrix as the A fragments of the sums' product: lane (piece, h), k16-step t, element j <-> row rho(t, h, j)
  int idx0;          // row of the chunk's partial-sum table of piece 0
  int p_lo, p_hi;    // pieces [p_lo, p_hi) are complete inside the wave: stored by the wave itself
  bool first_in;     // the wave's row 0 continues a run of the chunk's first wave (or lies beyond the tile): piece 0 goes to the fix-up
  bool last_out;     // chunk's first wave: its last piece continues in the second wave (or the second wave holds no row): piece 31 goes to the fix-up
  int idx_straddle;  // table row of the run the fix-up completes
  bool chunk_live;   // the chunk holds at least one row
};

// row of the wave that slot (k16-step t, lane half h, element j) of a B fragment taken from the C/D registers 8 t + j stands for
// (C/D layout: register 4 g + jj of lane half h = row 8 g + 4 h + jj; the fragment's elements j = 0..7 are registers 8 t + j: g = 2 t + (j >> 2))
__device__ __forceinline__ int edge_n_row_of_slot(int t, int h, int j) { return 16 * t + 8 * (j >> 2) + 4 * h + (j & 3); }

// s_dst: the tile's destinations (128 entries, clamped beyond the tile); rows: rows of the tile; wv: wave
__device__ __forceinline__ EdgeNRuns edge_n_runs(const int* s_dst, int rows, int wv, int lane) {
  const int pass = wv >> 1, half = wv & 1;
  const int crow = 64 * pass + lane;
  const bool valid = crow < rows;
  const int d = s_dst[crow], dp = lane ? s_dst[crow - 1] : -1;
  const unsigned long long hm64 = __ballot(valid && (lane == 0 || d != dp));
  EdgeNRuns r;
  const unsigned hm = (unsigned)(hm64 >> (32 * half));
  r.first_in = (hm & 1u) == 0u;
  r.last_out = half == 0 && ((hm64 >> 32) & 1ull) == 0ull;
  const int np = __popc(hm) + (r.first_in ? 1 : 0);  // pieces of the wave
  r.idx0 = __popcll(hm64 & ((1ull << (32 * half)) - 1ull)) - (r.first_in ? 1 : 0);  // run starts in front of the wave (- 1: piece 0 continues the last of them)
  r.p_lo = r.first_in ? 1 : 0;
  r.p_hi = np - (r.last_out ? 1 : 0);
  r.idx_straddle = __popcll(hm64 & 0xffffffffull) - 1;
  r.chunk_live = rows > 64 * pass;
  // membership: lane (piece m, h)
  const int m = lane & 31, h = lane >> 5;
  unsigned w[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      unsigned dw = 0;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int row = edge_n_row_of_slot(t, h, 2 * u + e);
        int p = __popc(hm & ((2u << row) - 1u)) - (r.first_in ? 0 : 1);  // piece of the row (run starts at rows <= row)
        if (r.last_out && p == np - 1) p = 31;
        if (p == m) dw |= 0x3f80u << (16 * e);  // bf16 1.0
      }
      w[t][u] = dw;
    }
  r.m[0] = __builtin_bit_cast(bf16x8r, u32x4r{w[0][0], w[0][1], w[0][2], w[0][3]});
  r.m[1] = __builtin_bit_cast(bf16x8r, u32x4r{w[1][0], w[1][1], w[1][2], w[1][3]});
  return r;
}

// S[piece][o] += sum over the rows of k16-step t of the wave (registers 8 t + j of the block) — on the matrix cores: the membership matrix times
// the block, the block's values as the three bf16 parts that hold their 24 mantissa bits exactly (a product with 1.0 is exact; fp32 accumulation
// in the instruction's fixed order).  vh / vm / vl: the parts of registers 8 t .. 8 t + 7 as one B fragment.  Result in the C/D layout: register
// 4 g + j of lane (o, hi) = piece 8 g + 4 hi + j, output o.
__device__ __forceinline__ void edge_n_piece_sums_step(f32x16r& S, const EdgeNRuns& rn, int t, bf16x8r vh, bf16x8r vm, bf16x8r vl) {
  S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rn.m[t], vl, S, 0, 0, 0);  // small parts first
  S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rn.m[t], vm, S, 0, 0, 0);
  S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rn.m[t], vh, S, 0, 0, 0);
}

// stores the pieces that are complete inside the wave to their rows of the chunk's partial-sum table (agg: the table row of the chunk's run 0 at
// this lane's output column)
__device__ __forceinline__ void edge_n_store_pieces(const f32x16r& S, const EdgeNRuns& rn, float* __restrict__ agg, int row_stride, int hi) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (8 * g < rn.p_hi) {  // (wave-uniform)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int p = 8 * g + 4 * hi + j;
        if (p >= rn.p_lo && p < rn.p_hi) agg[(size_t)(rn.idx0 + p) * row_stride] = S[4 * g + j];
      }
    }
  }
}

}  // namespace gnx

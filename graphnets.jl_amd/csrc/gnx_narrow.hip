// Fused GNBlock kernel for narrow feature widths (README-sized dims): ONE launch does the edge update, the
// edge->node segmented sum, the node update and the per-tile partial sums of the graph update.
//
// Why it can be fused: the reference's edge order is CSC order (src/pad.jl:30 — sorted by destination), so the
// in-edges of a node range [n0, n1) are the contiguous edge range [colptr[n0], colptr[n1]).  A workgroup that owns
// a node tile therefore owns every edge that aggregates into it: ef' never has to be re-read from HBM, the
// edge->node sum (nodefninput.jl:3) needs no atomics, and its order is fixed.
//
// Data movement per tile (256 threads, wave64):
//   HBM -> LDS   ef rows of the tile as ONE flat contiguous range, 16-B loads (coalesced along the feature dim);
//                node rows: the tile's own nodes, or the whole graph's node window when it is small (then the
//                nf[src] gather of edgefninput.jl:4 is served from LDS; otherwise it is a global/L2 row gather)
//   compute      weights are wave-uniform -> scalar loads / SGPR operands; gf[g] is folded into a per-tile bias
//   LDS -> HBM   ef' and nf' of the tile as flat contiguous ranges, 16-B stores
// Dims are template parameters (fully unrolled FMAs); launch_block_narrow() dispatches over the instantiated
// set and reports "not applicable" (1) otherwise, in which case the generic kernels run.
#include <cstdlib>

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
__device__ __forceinline__ cfloatp as_const(const float* p) { return reinterpret_cast<cfloatp>(reinterpret_cast<uintptr_t>(p)); }

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

// Deterministic block sum of C per-thread values: DPP row sums -> 16 row leaders write LDS -> fixed-order add.
// After the call, s_red[r*C + c] (r = 0..15) are visible to every thread (one barrier inside).
template <int C>
__device__ __forceinline__ void block_rows_to_lds(const float (&v)[C > 0 ? C : 1], float* s_red) {
  const int lane = threadIdx.x & 63, row = threadIdx.x >> 4;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float x = row16_sum(v[c]);
    if ((lane & 15) == 0) s_red[row * C + c] = x;
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ float sum16(const float* s_red, int C, int c) {
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += s_red[r * C + c];
  return s;
}

}  // namespace

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a release/acquire fence that makes hipcc
// drain vmcnt as well, which would wait for the next tile's prefetch loads (and this tile's output stores) at every
// barrier; the data exchanged between the waves of this kernel goes through LDS only.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Per-thread registers of one tile in flight.
template <int DE, int DN, int EPT, int NV_ND>
struct Stage {
  Tile t;                                // wave-uniform
  int cp0, cp1;                          // tile-local in-edge range of this thread's node
  float xn[DN > 0 ? DN : 1];             // this thread's node row
  float x[EPT][DE > 0 ? DE : 1];         // this thread's edge rows (first chunk)
  int src[EPT];                          // their global source node ids
  float xs[EPT][DN > 0 ? DN : 1];        // gathered source rows
  float4 v_nd[NV_ND];                    // WINDOW: this thread's slice of the graph's node rows, on its way to LDS
};

// TE / TN: compile-time tile capacities (edges per chunk, nodes per tile; TN <= 256 = one node per thread and a dst
// index fits a byte).  WINDOW: every graph has <= WIN nodes, so a graph's node rows are staged in LDS once per graph
// and nf[src] (edgefninput.jl:4) is gathered from LDS; otherwise nf[src] is a row gather from L2/HBM.
//
// Persistent + software pipelined: a workgroup walks `tiles_per_wg` consecutive tiles; while tile i is computed, the
// loads of tile i+1 (ef rows, rowval, colptr, node rows) are already in flight and its nf[src] gather is issued in
// the middle of tile i — one memory round trip per tile is exposed instead of three.
template <int DE, int DN, int DG, int OE, int ON, int TE, int TN, bool WINDOW>
__global__ __launch_bounds__(kThreads) void k_block_fused(BlockArgs a, int tiles_per_wg) {
  constexpr int OE1 = OE > 0 ? OE : 1, ON1 = ON > 0 ? ON : 1;
  constexpr int WIN = 256;
  constexpr int EPT = TE / kThreads;                              // edges per thread and chunk
  constexpr int NV_ND = WINDOW ? (WIN * DN / 4) / kThreads + 1 : 1;
  constexpr int C = OE + ON, C1 = C > 0 ? C : 1;
  static_assert(TE % kThreads == 0 && TN <= 256, "tile shape");
  using St = Stage<DE, DN, EPT, NV_ND>;
  __shared__ __attribute__((aligned(16))) float s_out[TE * OE + 4];            // ef' of the tile (for the node sums)
  __shared__ __attribute__((aligned(16))) float s_pd[TN * OE + 4];             // per node: bias' + We[:, dst-seg] * nf[n]
  __shared__ __attribute__((aligned(16))) float s_nd[WINDOW ? WIN * DN + 8 : 4];  // the current graph's node rows
  __shared__ float s_red[16 * C1 + 4];
  __shared__ unsigned char s_dst[TE + 4];                                      // tile-local destination of each edge

  const size_t r = blockIdx.y;
  const float* __restrict__ ef = DE > 0 ? a.ef + r * (size_t)a.E * DE : nullptr;
  const float* __restrict__ nf = DN > 0 ? a.nf + r * (size_t)a.N * DN : nullptr;
  const cfloatp We = as_const(a.We);
  const cfloatp Wn = as_const(a.Wn);
  const cfloatp be = as_const(a.be);
  const cfloatp bn = as_const(a.bn);
  const int tid = threadIdx.x;
  const int wg = xcd_tile(blockIdx.x, gridDim.x);
  const int ti0 = wg * tiles_per_wg;
  const int ti1 = ti0 + tiles_per_wg < a.n_tiles ? ti0 + tiles_per_wg : a.n_tiles;
  typedef const int __attribute__((address_space(4))) * cintp;
  const cintp tile_words = reinterpret_cast<cintp>(reinterpret_cast<uintptr_t>(a.tiles));  // scalar loads (s_load_dwordx8)
  auto load_tile = [&](int i) {
    Tile t;
    const cintp w = tile_words + (size_t)i * (sizeof(Tile) / sizeof(int));
    t.n0 = w[0]; t.n1 = w[1]; t.e0 = w[2]; t.e1 = w[3]; t.g = w[4]; t.win0 = w[5]; t.win1 = w[6]; t.flags = w[7];
    return t;
  };

  // stage 1: every load whose address depends on the tile only
  auto stage1 = [&](St& st, int staged_g) {
    const Tile& t = st.t;
    const int nn = t.n1 - t.n0, ne = t.e1 - t.e0;
    const int cn = ne < TE ? ne : TE;
    st.cp0 = st.cp1 = 0;
    if (tid < nn) {
      st.cp0 = a.colptr[t.n0 + tid] - t.e0;
      st.cp1 = a.colptr[t.n0 + tid + 1] - t.e0;
      if constexpr (DN > 0) load_row<DN>(nf + (size_t)(t.n0 + tid) * DN, st.xn);
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int el = tid + i * kThreads;
      st.src[i] = t.n0;
      if (el < cn) {
        if constexpr (DE > 0) {
          if (!(a.ablate & 1)) load_row<DE>(ef + (size_t)(t.e0 + el) * DE, st.x[i]);
        }
        if constexpr (DN > 0) {
          if (!(a.ablate & 16)) st.src[i] = a.rowval[t.e0 + el];
        }
      }
    }
    if constexpr (WINDOW && DN > 0) {
      if (t.g != staged_g) {  // node rows of the graph, flat and 16-B coalesced (tail/head handled at write time)
        const float* gnd = nf + (size_t)t.win0 * DN;
        const int n = (t.win1 - t.win0) * DN;
        const int head = (int)(((16u - (unsigned)((uintptr_t)gnd & 15u)) & 15u) >> 2);
        const int h = head < n ? head : n;
        const int n4 = (n - h) >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(gnd + h);
#pragma unroll
        for (int i = 0; i < NV_ND; ++i) {
          const int q = tid + i * kThreads;
          if (q < n4) st.v_nd[i] = g4[q];
        }
      }
    }
  };
  // stage 2: the source-row gather (address depends on rowval)
  auto stage2 = [&](St& st) {
    if constexpr (DN > 0 && !WINDOW) {
      const int ne = st.t.e1 - st.t.e0;
      const int cn = ne < TE ? ne : TE;
#pragma unroll
      for (int i = 0; i < EPT; ++i)
        if (tid + i * kThreads < cn && !(a.ablate & 2)) load_row<DN>(nf + (size_t)st.src[i] * DN, st.xs[i]);
    }
  };

  if (ti0 >= ti1) return;
  St cur, nxt;
  cur.t = load_tile(ti0);
  int staged_g = -1;  // graph whose node rows are in s_nd
  stage1(cur, staged_g);
  if (ti0 + 1 < ti1) nxt.t = load_tile(ti0 + 1);
  stage2(cur);

  for (int ti = ti0; ti < ti1; ++ti) {
    const bool has_next = ti + 1 < ti1;
    const Tile t = cur.t;
    const int nn = t.n1 - t.n0, ne = t.e1 - t.e0;
    const bool is_node = tid < nn;
    const cfloatp gf = DG > 0 ? as_const(a.gf + (r * (size_t)a.G + t.g) * DG) : nullptr;
    if (has_next) stage1(nxt, t.g);  // prefetch: in flight during everything below
    Tile t2 = t;
    if (ti + 2 < ti1) t2 = load_tile(ti + 2);

    // ---- node prep: destination index of every in-edge, per-node part of the edge update, node window ----
    //   pd[n] = be + We[:, gf-seg] * gf[g] + We[:, dst-seg] * nf[n]     (edgefninput.jl:5-6 hoisted out of the edge loop)
    if (is_node) {
      if (nn > 1)
        for (int e = cur.cp0; e < cur.cp1; ++e) s_dst[e] = (unsigned char)tid;
      if constexpr (OE > 0) {
#pragma unroll
        for (int j = 0; j < OE; ++j) {
          float b = a.be ? be[j] : 0.f;
#pragma unroll
          for (int k = 0; k < DG; ++k) b = fmaf(We[(DE + 2 * DN + k) * OE + j], gf[k], b);
#pragma unroll
          for (int k = 0; k < DN; ++k) b = fmaf(We[(DE + DN + k) * OE + j], cur.xn[k], b);
          s_pd[tid * OE + j] = b;
        }
      }
    }
    if constexpr (WINDOW && DN > 0) {
      if (t.g != staged_g) {
        const float* gnd = nf + (size_t)t.win0 * DN;
        const int n = (t.win1 - t.win0) * DN;
        const int head = (int)(((16u - (unsigned)((uintptr_t)gnd & 15u)) & 15u) >> 2);
        const int h = head < n ? head : n;
        const int n4 = (n - h) >> 2;
        if (tid < h) s_nd[tid] = gnd[tid];
        const int done = h + 4 * n4;
        if (tid < n - done) s_nd[done + tid] = gnd[done + tid];
#pragma unroll
        for (int i = 0; i < NV_ND; ++i) {
          const int q = tid + i * kThreads;
          if (q < n4) {
            float* d = s_nd + h + 4 * q;
            d[0] = cur.v_nd[i].x; d[1] = cur.v_nd[i].y; d[2] = cur.v_nd[i].z; d[3] = cur.v_nd[i].w;
          }
        }
        staged_g = t.g;
      }
    }
    lds_barrier();

    // ---- edge update: first chunk from the prefetched registers, further chunks (single-node tile with a huge
    //      in-degree only) loaded in place ----
    float psum[OE1];
#pragma unroll
    for (int j = 0; j < OE1; ++j) psum[j] = 0.f;
    for (int c0 = 0; c0 < ne; c0 += TE) {
      const int cn = (ne - c0) < TE ? (ne - c0) : TE;
      if (c0 > 0) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
          const int el = tid + i * kThreads;
          if (el < cn) {
            if constexpr (DE > 0) load_row<DE>(ef + (size_t)(t.e0 + c0 + el) * DE, cur.x[i]);
            if constexpr (DN > 0) {
              cur.src[i] = a.rowval[t.e0 + c0 + el];
              if constexpr (!WINDOW) load_row<DN>(nf + (size_t)cur.src[i] * DN, cur.xs[i]);
            }
          }
        }
      }
      if constexpr (OE > 0) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
          const int el = tid + i * kThreads;
          if (el < cn) {
            const int dl = nn > 1 ? (int)s_dst[el] : 0;
            float acc[OE1];
#pragma unroll
            for (int j = 0; j < OE; ++j) acc[j] = s_pd[dl * OE + j];
            if constexpr (DE > 0) {
#pragma unroll
              for (int k = 0; k < DE; ++k)
#pragma unroll
                for (int j = 0; j < OE; ++j) acc[j] = fmaf(We[k * OE + j], cur.x[i][k], acc[j]);
            }
            if constexpr (DN > 0) {
              if constexpr (WINDOW) {
#pragma unroll
                for (int k = 0; k < DN; ++k) cur.xs[i][k] = s_nd[(cur.src[i] - t.win0) * DN + k];
              }
#pragma unroll
              for (int k = 0; k < DN; ++k)
#pragma unroll
                for (int j = 0; j < OE; ++j) acc[j] = fmaf(We[(DE + k) * OE + j], cur.xs[i][k], acc[j]);
            }
#pragma unroll
            for (int j = 0; j < OE; ++j) acc[j] = act_apply(acc[j], a.act_e);
            if (!(a.ablate & 4)) store_row<OE>(a.ef_out + (r * (size_t)a.E + t.e0 + c0 + el) * OE, acc);
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
    }
    if (has_next) stage2(nxt);  // next tile's gather flies during the node phase below

    // ---- node update ----
    float v[C1];  // per-thread contribution to the tile's graph-level partial sums: [agg ; nf']
#pragma unroll
    for (int c = 0; c < C1; ++c) v[c] = 0.f;
    if (nn == 1) {
      // every edge of the tile aggregates into its only node: agg = block-wide sum of the per-thread shares
      if constexpr (OE > 0) {
        block_rows_to_lds<OE>(psum, s_red);
        if (tid == 0) {
#pragma unroll
          for (int j = 0; j < OE; ++j) v[j] = sum16(s_red, OE, j);
        }
        lds_barrier();  // s_red is reused below
      }
    } else {
      lds_barrier();  // s_out complete
      if constexpr (OE > 0) {
        if (is_node) {  // contiguous segmented sum (edges are dst-sorted, src/pad.jl:30), fixed order
          for (int e = cur.cp0; e < cur.cp1; ++e) {
#pragma unroll
            for (int j = 0; j < OE; ++j) v[j] += s_out[e * OE + j];
          }
        }
      }
    }
    if constexpr (ON > 0) {
      if (is_node) {
        float acc[ON1];
#pragma unroll
        for (int j = 0; j < ON; ++j) {
          float b = a.bn ? bn[j] : 0.f;
#pragma unroll
          for (int k = 0; k < DG; ++k) b = fmaf(Wn[(OE + DN + k) * ON + j], gf[k], b);
          acc[j] = b;
        }
#pragma unroll
        for (int k = 0; k < OE; ++k)
#pragma unroll
          for (int j = 0; j < ON; ++j) acc[j] = fmaf(Wn[k * ON + j], v[k], acc[j]);
#pragma unroll
        for (int k = 0; k < DN; ++k)
#pragma unroll
          for (int j = 0; j < ON; ++j) acc[j] = fmaf(Wn[(OE + k) * ON + j], cur.xn[k], acc[j]);
#pragma unroll
        for (int j = 0; j < ON; ++j) {
          acc[j] = act_apply(acc[j], a.act_n);
          v[OE + j] = acc[j];
        }
        if (!(a.ablate & 8)) store_row<ON>(a.nf_out + (r * (size_t)a.N + t.n0 + tid) * ON, acc);
      }
    }

    // ---- per-tile partial sums for the graph update (graphfninput.jl:3-4): sum_e ef' = sum_n agg[n], sum_n nf' ----
    if (a.og > 0) {
      if constexpr (C > 0) {
        block_rows_to_lds<C>(v, s_red);
        if (tid < C) a.partials[(r * (size_t)a.n_tiles + ti) * C + tid] = sum16(s_red, C, tid);
      }
    } else {
      lds_barrier();  // s_out / s_pd / s_dst are rewritten by the next tile
    }
    cur = nxt;
    nxt.t = t2;
  }
}

int32_t launch_graph(const BlockArgs& a, int64_t R, hipStream_t s);

template <int DE, int DN, int DG, int OE, int ON, int TE, int TN>
static int32_t launch_fused_t(const gnx_graphs* h, const BlockArgs& a, int64_t R, hipStream_t s) {
  // LDS node window only when EVERY graph fits it (then every source of an edge is inside the staged rows)
  static const bool no_window = getenv("GNX_NO_WINDOW") != nullptr;
  static const int wg_per_cu = getenv("GNX_WG_PER_CU") ? atoi(getenv("GNX_WG_PER_CU")) : 4;
  static const int fixed_k = getenv("GNX_TILES_PER_WG") ? atoi(getenv("GNX_TILES_PER_WG")) : 0;
  const bool window = DN > 0 && h->PN <= 256 && !no_window;
  // persistent grid: about wg_per_cu workgroups per CU (256 CUs), each walking K consecutive tiles
  int K = fixed_k > 0 ? fixed_k : (int)((a.n_tiles + 256 * wg_per_cu - 1) / (256 * wg_per_cu));
  if (K < 1) K = 1;
  const unsigned grid = (unsigned)((a.n_tiles + K - 1) / K);
  {
    ProfScope ps("k_block_fused", s);
    if (window)
      hipLaunchKernelGGL((k_block_fused<DE, DN, DG, OE, ON, TE, TN, true>), dim3(grid, (unsigned)R), dim3(kThreads), 0, s, a, K);
    else
      hipLaunchKernelGGL((k_block_fused<DE, DN, DG, OE, ON, TE, TN, false>), dim3(grid, (unsigned)R), dim3(kThreads), 0, s, a, K);
    GNX_HIP(hipGetLastError());
  }
  return launch_graph(a, R, s);
}

template <int DE, int DN, int DG, int OE, int ON, int TE, int TN>
constexpr bool fused_fits() { return (size_t)TE * (4 * OE + 1) + 256 * 4 * (size_t)DN + (size_t)TN * 4 * OE + 16 * 4 * (OE + ON) + 512 <= 60 * 1024; }

template <int DE, int DN, int DG, int OE, int ON>
static int32_t launch_fused(const gnx_graphs* h, const BlockArgs& a, int64_t R, hipStream_t s) {
  // the kernel's tile shape must cover the handle's tile caps (GNX_TILE_E / GNX_TILE_N at handle creation)
  if (h->tile_e_cap <= 256 && h->tile_n_cap <= 64) {
    if constexpr (fused_fits<DE, DN, DG, OE, ON, 256, 64>()) return launch_fused_t<DE, DN, DG, OE, ON, 256, 64>(h, a, R, s);
  } else if (h->tile_e_cap <= 512 && h->tile_n_cap <= 128) {
    if constexpr (fused_fits<DE, DN, DG, OE, ON, 512, 128>()) return launch_fused_t<DE, DN, DG, OE, ON, 512, 128>(h, a, R, s);
  } else if (h->tile_e_cap <= 1024 && h->tile_n_cap <= 256) {
    if constexpr (fused_fits<DE, DN, DG, OE, ON, 1024, 256>()) return launch_fused_t<DE, DN, DG, OE, ON, 1024, 256>(h, a, R, s);
  }
  return 1;
}

// Instantiated width sets.  (de, dn, dg) => (oe, on); og is free (the graph update is its own small kernel).
#define GNX_NARROW_DIMS(X) \
  X(10, 5, 0, 3, 4)        \
  X(3, 4, 5, 3, 4)         \
  X(10, 5, 3, 3, 4)        \
  X(0, 2, 0, 2, 2)         \
  X(2, 2, 2, 2, 2)         \
  X(4, 3, 2, 3, 4)         \
  X(8, 8, 8, 16, 8)        \
  X(10, 5, 3, 10, 5)       \
  X(10, 5, 0, 10, 5)

int32_t launch_block_narrow(const gnx_graphs* h, const BlockArgs& a, int64_t R, hipStream_t s) {
  if (h->tile_n_cap > 256 || a.n_tiles == 0) return 1;
  // 16-B vector copies assume fp32-aligned buffers (always true for fp32 arrays); nothing else is required
#define GNX_CASE(DE, DN, DG, OE, ON) \
  if (a.de == DE && a.dn == DN && a.dg == DG && a.oe == OE && a.on == ON) return launch_fused<DE, DN, DG, OE, ON>(h, a, R, s);
  GNX_NARROW_DIMS(GNX_CASE)
#undef GNX_CASE
  return 1;
}

}  // namespace gnx

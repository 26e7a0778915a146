// EXPERIMENT (not product): the projected edge update  out[e] = relu(ef[e] * W + Ps[src(e)] + Pd[dst(e)])  (K = OUT = 128, 1M edges)
// as ONE persistent 9-wave workgroup per CU with fixed roles — the loader / consumer form DESIGN §8 derives from round 3's measurements:
//   waves 0-3  consumers: 64 x 64 of the 128 x 128 tile each on the fp32 matrix cores, W (64 KB) RESIDENT in LDS for the workgroup's life,
//              ef chunks from a 3-slot LDS ring; hand their accumulators to a 64-row staging buffer at the end of a tile
//   waves 4-7  epilogue: take a staged pass into registers, add the prefetched source rows and the tile's destination rows (LDS), activation,
//              full-row 16-B stores — while the consumers are already in the next tile's K loop
//   wave  8    loader: LDS-DMA (global_load_lds_dwordx4, inline asm: kept out of the compiler's books) of the ef chunks two chunk steps ahead,
//              of the tile's index rows and of its destination-projection rows
// Workgroup barriers only: three role loops with the same trip count and one barrier per trip (every wave executes the same number of barriers).
// Build: hipcc -O3 --offload-arch=gfx950 -o edge_ws edge_ws.hip ; run: ./edge_ws [E]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// ablation switches (timing-only builds: results are wrong): -DNO_MFMA, -DNO_STORE, -DNO_GATHER, -DNO_ACHUNK
#ifdef NO_STORE
#define STORE_COND(v) ((v).x == 12345.678f)
#else
#define STORE_COND(v) true
#endif
#ifdef NO_GATHER
#define GATHER_OR_ZERO(x) (v4f{(float)idx_[row_], 0.f, 0.f, 0.f})
#else
#define GATHER_OR_ZERO(x) (x)
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, KD = 128, KC = 32, NCH = KD / KC, RING = 3;
constexpr int LDC = BN + 4;
constexpr int PD_ROWS = 20;
// LDS map (bytes); every LDS-DMA target below 64 KB
constexpr int OFF_A = 0;                                  // [RING][BM * KC] floats, swizzled 1-KiB pieces
constexpr int OFF_PD = OFF_A + RING * BM * KC * 4;        // [PD_ROWS][BN]
constexpr int OFF_IDX = OFF_PD + PD_ROWS * BN * 4;        // [2][256] ints: src rows | dst rows of a tile
constexpr int OFF_W = OFF_IDX + 2 * 256 * 4;              // [KD][BN]
constexpr int OFF_C = OFF_W + KD * BN * 4;                // [64][LDC]
constexpr int LDS_BYTES = OFF_C + 64 * LDC * 4;
static_assert(OFF_W <= 65536, "DMA targets below 64 KB");
static_assert(LDS_BYTES <= 163840, "LDS");

struct Args {
  const float* A;      // [E][KD]
  const float* W;      // [KD][BN]
  const float* Ps;     // [N][BN]
  const float* Pd;     // [N][BN]
  const int* src;      // [E]
  const int* dst;      // [E] non-decreasing
  float* out;          // [E][BN]
  int n_tiles;         // E / 128 (full tiles only in this experiment)
  int N;
};

// one LDS-DMA piece: 64 lanes x 16 B, lane-linear at LDS byte address lds_dst (wave-uniform)
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// position of ef element (row, k) inside a ring slot (floats): pieces of 8 rows x 32 floats; the quad index is XORed so that the 32 rows of an
// A fragment read (fixed k) spread over 16 banks x 2 (2-way conflict) instead of 2 banks x 16
__device__ __forceinline__ int a_pos(int row, int k) {
  const int p = row >> 3, r = row & 7;
  const int f = (r >> 1) | ((p & 1) << 2);
  return p * 256 + r * 32 + (((k >> 2) ^ f) << 2) + (k & 3);
}

__global__ __launch_bounds__(576) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_edge_ws(Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* sA = reinterpret_cast<float*>(lds + OFF_A);
  float* sPd = reinterpret_cast<float*>(lds + OFF_PD);
  int* sIdx = reinterpret_cast<int*>(lds + OFF_IDX);
  float* sW = reinterpret_cast<float*>(lds + OFF_W);
  float* sC = reinterpret_cast<float*>(lds + OFF_C);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x, wg = blockIdx.x;
  const int T = (a.n_tiles - wg + nwg - 1) / nwg;  // tiles of this workgroup: wg, wg + nwg, ...
  if (T <= 0) return;                               // (whole workgroup: no barrier executed yet)

  // ---- prologue: W resident (plain loads + ds_write, once), chunks 0 and 1 of the first tile by DMA ----
  // W resident as [k / 4][column][k % 4]: the B fragments of FOUR k-steps are one ds_read_b128 (conflict-free: 32 lanes = 32 consecutive columns)
  for (int i = tid; i < KD * BN; i += 576) { const int k = i / BN, c = i % BN; sW[((k >> 2) * BN + c) * 4 + (k & 3)] = a.W[i]; }
  auto dma_chunk = [&](int g) {  // global chunk counter g of this workgroup: tile wg + (g / 4) * nwg, K range (g % 4) * 32
    const int j = g / NCH, kc = (g % NCH) * KC;
    if (j >= T) return;
#ifdef NO_ACHUNK
    if (g >= 2) return;
#endif
    const size_t row0 = (size_t)(wg + j * nwg) * BM;
    const unsigned slot = OFF_A + (unsigned)(g % RING) * (BM * KC * 4);
    const int r = lane >> 3, q = lane & 7;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int f = (r >> 1) | ((p & 1) << 2);
      const float* gp = a.A + (row0 + 8 * p + r) * KD + kc + ((q ^ f) << 2);
      dma16(gp, slot + p * 1024);
    }
  };
  if (wv == 8) { dma_chunk(0); dma_chunk(1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wg_barrier();

  // Three role loops with the SAME trip count and one barrier per trip each (separate loops: one loop over all roles would keep the
  // consumers' accumulators and the epilogue waves' four operand buffers alive together in every wave's register file).
  const int nsteps = 7 * T + 2;  // per tile: 4 chunk steps, hand-over pass 0, read pass 0, hand-over pass 1; two drain steps at the end
  if (wv < 4) {
    // ================= consumers =================
    const int hi = lane >> 5, l31 = lane & 31;
    const int wm = (wv >> 1) & 1, wn = wv & 1;
    f32x16 acc[2][2];
    for (int s = 0; s < nsteps; ++s) {
      const int j = s / 7, u = s - 7 * j;
      if (j < T) {
        if (u < NCH) {
          if (u == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][jj][q] = 0.f;
          }
          // k-step order permuted so that a lane's fragments of four consecutive steps are ONE quad: instruction (m, t) multiplies k = 8 m + t
          // (lanes 0-31) and k = 8 m + 4 + t (lanes 32-63) of the chunk — the same k for A and B, every k once.  The whole chunk's fragments
          // (16 ds_read_b128, 64 registers) are requested up front; the matrix cores then run back to back behind counted waits.
          const float* slot = sA + ((NCH * j + u) % RING) * (BM * KC);
          const float* wq = sW + (u * (KC / 4)) * BN * 4;
          v4f fa[4][2], fb[4][2];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[m][i] = *reinterpret_cast<const v4f*>(slot + a_pos((i * 2 + wm) * 32 + l31, 4 * (2 * m + hi)));
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) fb[m][jj] = *reinterpret_cast<const v4f*>(wq + ((2 * m + hi) * BN + (wn * 2 + jj) * 32 + l31) * 4);
          }
          __builtin_amdgcn_sched_barrier(0);  // (left to itself the compiler sinks each group's reads to two MFMAs before their use)
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#ifndef NO_MFMA
#pragma unroll
              for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[m][i][t], fb[m][jj][t], acc[i][jj], 0, 0, 0);
#else
              acc[0][0][4 * m + t] += fa[m][0][t] + fb[m][0][t] + fa[m][1][t] + fb[m][1][t];
#endif
            }
#ifndef NO_HAND
        } else if (u == 4) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int q = 0; q < 16; ++q) sC[(wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * hi) * LDC + (wn * 2 + jj) * 32 + l31] = acc[0][jj][q];
        } else if (u == 6) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int q = 0; q < 16; ++q) sC[(wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * hi) * LDC + (wn * 2 + jj) * 32 + l31] = acc[1][jj][q];
#else
        } else if (u == 6 && acc[0][0][0] == 12345.f) { sC[lane] = acc[1][1][3] + acc[0][1][2] + acc[1][0][1];
#endif
        }
      }
      wg_barrier();
    }
  } else if (wv < 8) {
    // ================= epilogue waves (256 threads) =================
    // the tile's seven steps written out (one barrier each): the compiler then sees exactly which buffer lives across which barrier
    const int et = tid - 256, q4 = et & 31, lr0 = et >> 5;
    v4f cp0[8], cp1[8], ps0[8], ps1[8];  // (first-class vectors, used in place: arrays of HIP's float4 struct behind pointers end up in scratch memory)
#define GNX_FINISH(JT, PASS, CP, PS) do {                                                                            \
      const int* idx_ = sIdx + ((JT) & 1) * 256;                                                                     \
      const int first_ = idx_[128];                                                                                  \
      float* ob_ = a.out + (size_t)(wg + (JT) * nwg) * BM * BN;  /* wave-uniform base, 32-bit offsets */             \
      _Pragma("unroll") for (int uu = 0; uu < 8; ++uu) {                                                             \
        const int row_ = 64 * (PASS) + lr0 + 8 * uu;                                                                 \
        const int d_ = min(idx_[128 + row_] - first_, PD_ROWS - 1);                                                  \
        const v4f pd_ = *reinterpret_cast<const v4f*>(sPd + d_ * BN + 4 * q4);                                       \
        v4f v_ = CP[uu] + (PS[uu] + pd_);                                                                            \
        v_.x = fmaxf(v_.x, 0.f); v_.y = fmaxf(v_.y, 0.f); v_.z = fmaxf(v_.z, 0.f); v_.w = fmaxf(v_.w, 0.f);          \
        if (STORE_COND(v_)) *reinterpret_cast<v4f*>(reinterpret_cast<char*>(ob_) + ((unsigned)(row_ * BN + 4 * q4) << 2)) = v_;          \
      }                                                                                                              \
    } while (0)
#define GNX_GATHER(JT, PASS, PS) do {                                                                                \
      const int* idx_ = sIdx + ((JT) & 1) * 256;                                                                     \
      _Pragma("unroll") for (int uu = 0; uu < 8; ++uu) {                                                             \
        const int row_ = 64 * (PASS) + lr0 + 8 * uu;                                                                 \
        PS[uu] = GATHER_OR_ZERO(*reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(a.Ps) + (((unsigned)idx_[row_] * BN + 4u * q4) << 2))); \
      }                                                                                                              \
    } while (0)
#define GNX_TAKE(CP) do { _Pragma("unroll") for (int uu = 0; uu < 8; ++uu) CP[uu] = *reinterpret_cast<const v4f*>(sC + (lr0 + 8 * uu) * LDC + 4 * q4); } while (0)
#ifdef NO_EPI
    for (int s2 = 0; s2 < nsteps; ++s2) wg_barrier();
    if (false)
#endif
    {
    for (int j = 0; j < T; ++j) {
      if (j > 0) GNX_FINISH(j - 1, 0, cp0, ps0);
      wg_barrier();                                   // step 0
      if (j > 0) { GNX_TAKE(cp1); GNX_FINISH(j - 1, 1, cp1, ps1); }  // pass 1 of tile j - 1 (the consumers write sC again at step 4)
      wg_barrier();                                   // step 1
      GNX_GATHER(j, 0, ps0);
      wg_barrier();                                   // step 2
      GNX_GATHER(j, 1, ps1);
      wg_barrier();                                   // step 3
      wg_barrier();                                   // step 4: the consumers hand over pass 0
      GNX_TAKE(cp0);
      wg_barrier();                                   // step 5
      wg_barrier();                                   // step 6: the consumers hand over pass 1
    }
    GNX_FINISH(T - 1, 0, cp0, ps0);
    wg_barrier();                                     // drain step 0
    GNX_TAKE(cp1); GNX_FINISH(T - 1, 1, cp1, ps1);
    wg_barrier();                                     // drain step 1
    }
#undef GNX_FINISH
#undef GNX_GATHER
#undef GNX_TAKE
  } else {
    // ================= loader =================
    for (int s = 0; s < nsteps; ++s) {
      const int j = s / 7, u = s - 7 * j;
      if (j < T && u < NCH) {
        if (u == 0) {  // index rows of tile j: lanes 0-31 the 128 source rows, lanes 32-63 the 128 destination rows
          const size_t row0 = (size_t)(wg + j * nwg) * BM;
          const int* gp = (lane < 32 ? a.src : a.dst) + row0 + 4 * (lane & 31);
          dma16(gp, OFF_IDX + (j & 1) * 1024);
        }
        if (u == 2) {  // destination-projection rows of tile j: first .. last destination, 2 rows per piece
          const int* idx = sIdx + (j & 1) * 256;
          const int first = __builtin_amdgcn_readfirstlane(idx[128]);
          const int last = min(__builtin_amdgcn_readfirstlane(idx[255]), first + PD_ROWS - 1);
          const int prow = lane >> 5, pq = lane & 31;
          for (int p = 0; 2 * p <= last - first; ++p) {
            const int row = min(first + 2 * p + prow, last);
            dma16(a.Pd + (size_t)row * BN + 4 * pq, OFF_PD + p * 1024);
          }
        }
        dma_chunk(NCH * j + u + 2);
        // everything but the chunk just requested (16 pieces) has landed when the step ends; the last chunk steps request nothing: wait for all
#ifdef NO_ACHUNK
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        if (NCH * j + u + 2 < NCH * T) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      wg_barrier();
    }
  }
}

// naive check of sampled rows
__global__ void k_ref(Args a, const int* rows, int nrows, float* ref) {
  const int i = blockIdx.x, c = threadIdx.x;
  if (i >= nrows) return;
  const size_t e = (size_t)rows[i];
  double acc = 0.0;
  for (int k = 0; k < KD; ++k) acc += (double)a.A[e * KD + k] * (double)a.W[k * BN + c];
  acc += (double)a.Ps[(size_t)a.src[e] * BN + c] + (double)a.Pd[(size_t)a.dst[e] * BN + c];
  ref[(size_t)i * BN + c] = (float)fmax(acc, 0.0);
}
__global__ void k_fill(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = ((x & 0xffffff) / 16777216.f - 0.5f) * scale;
  }
}

int main(int argc, char** argv) {
  const int n_tiles = argc > 1 ? atoi(argv[1]) : 7813;
  const size_t E = (size_t)n_tiles * BM;
  const int N = (int)(E / 10) + 1;
  float *A, *W, *Ps, *Pd, *out, *ref; int *src, *dst, *rows;
  CK(hipMalloc(&A, E * KD * 4)); CK(hipMalloc(&W, KD * BN * 4)); CK(hipMalloc(&Ps, (size_t)N * BN * 4)); CK(hipMalloc(&Pd, (size_t)N * BN * 4));
  CK(hipMalloc(&out, E * BN * 4)); CK(hipMalloc(&src, E * 4)); CK(hipMalloc(&dst, E * 4));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, E * KD, 1u, 2.f);
  hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, 0, W, (size_t)KD * BN, 2u, 0.2f);
  hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, Ps, (size_t)N * BN, 3u, 2.f);
  hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, Pd, (size_t)N * BN, 4u, 2.f);
  std::vector<int> hs(E), hd(E);
  unsigned x = 12345;
  for (size_t e = 0; e < E; ++e) { x = x * 1664525u + 1013904223u; hs[e] = (int)((x >> 8) % (unsigned)N); hd[e] = (int)(e / 10); }
  CK(hipMemcpy(src, hs.data(), E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dst, hd.data(), E * 4, hipMemcpyHostToDevice));
  CK(hipMemset(out, 0xff, E * BN * 4));
  Args a{A, W, Ps, Pd, src, dst, out, n_tiles, N};
  CK(hipFuncSetAttribute((const void*)k_edge_ws, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  int dev = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  const int nwg = argc > 2 ? atoi(argv[2]) : prop.multiProcessorCount;
  printf("tiles %d, E %zu, workgroups %d, LDS %d B\n", n_tiles, E, nwg, LDS_BYTES);
  hipLaunchKernelGGL(k_edge_ws, dim3(nwg), dim3(576), LDS_BYTES, 0, a);
  CK(hipGetLastError()); CK(hipDeviceSynchronize());
  // check: 2048 sampled rows (incl. the first and last tiles)
  const int nr = 2048; std::vector<int> hr(nr);
  for (int i = 0; i < nr; ++i) { x = x * 1664525u + 1013904223u; hr[i] = i < 128 ? i : (i < 256 ? (int)(E - 256 + i) : (int)((x >> 4) % E)); }
  CK(hipMalloc(&rows, nr * 4)); CK(hipMalloc(&ref, (size_t)nr * BN * 4)); CK(hipMemcpy(rows, hr.data(), nr * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_ref, dim3(nr), dim3(BN), 0, 0, a, rows, nr, ref); CK(hipDeviceSynchronize());
  std::vector<float> href((size_t)nr * BN), hout(BN);
  CK(hipMemcpy(href.data(), ref, href.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0; int bad = 0;
  for (int i = 0; i < nr; ++i) {
    CK(hipMemcpy(hout.data(), out + (size_t)hr[i] * BN, BN * 4, hipMemcpyDeviceToHost));
    for (int c = 0; c < BN; ++c) { const double d = fabs((double)hout[c] - href[(size_t)i * BN + c]); if (!(d <= 1e-3)) { if (bad < 5) printf("row %d col %d got %g want %g\n", hr[i], c, hout[c], href[(size_t)i * BN + c]); ++bad; } if (d > worst) worst = d; }
  }
  printf("check: %d sampled rows, worst |diff| %.3g, %d bad\n", nr, worst, bad);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_edge_ws, dim3(nwg), dim3(576), LDS_BYTES, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_edge_ws: %.1f us per launch (%.1f TF/s executed, %.2f TB/s algorithmic)\n", ms * 100.f, 2.0 * E * KD * BN / (ms * 1e-4) * 1e-12, (E * (KD + BN) * 4.0) / (ms * 1e-4) * 1e-12);
  }
  return bad ? 1 : 0;
}

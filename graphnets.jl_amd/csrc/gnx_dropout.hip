// Training-mode Dropout of a GNCore's FeedForwards.
//
// The reference's FeedForward is Chain(Dense(d => 4d, relu), Dense(4d => d), Dropout(p)) (src/gnfeedforward.jl:27-31): in training mode
// (inside a gradient call) Flux multiplies the FeedForward's OUTPUT element-wise by a fresh mask, m = rand > p ? 1 / (1 - p) : 0; in test
// mode the layer is the identity, which is what gnx_core_forward computes.  (GNBlock keeps a Dropout(p) field that its forward never applies:
// src/gnblock.jl:63-69.)  A core in training mode is therefore
//     y = x + block(gn1(x)) + m .* ffwd(gn2(x))                                                              (src/gncore.jl:56-59)
// The masks are not stored: element i of entity t's mask is a pure function of (seed, t, i) — Philox-4x32-10 keyed by the seed, counter
// (i / 4, t) — so the backward pass regenerates the forward's mask from the same gnx_dropout value, and a host can ask for it
// (gnx_dropout_mask) to check either pass against its own arithmetic.  Flux's mask comes from the array's own random-number generator; no
// implementation on another device reproduces those bits, so parity here is the FORM (independent per element, keep probability 1 - p, scale
// 1 / (1 - p), the same mask in both passes), tested against float64 with the library's mask handed to the oracle.
//
// gnx_core_forward_train = gnx_core_forward (every fused form as it is) followed by a correction y += (m - 1) .* f with f = ffwd(gn2(x))
// recomputed unfused into the workspace (LayerNorm launch + two row-wise Dense launches per entity) — the training step's backward recomputes
// the same intermediates anyway, and the inference kernels stay free of a mask operand.
#include <algorithm>

#include "gnx_device.h"

namespace gnx {
int32_t launch_dense_rows(const gnx_graphs* h, int entity, const float* A, int K, const gnx_dense& d, int OUT, const float* add1,
                          const float* add2, float* out, int64_t R, hipStream_t s, const char* name);
int32_t launch_layernorm2(const float* x, size_t rows, int d, const gnx_layernorm& l1, const gnx_layernorm& l2, float eps, int eps_mode,
                          float* y1, float* y2, hipStream_t s);

namespace {
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
  const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
  const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
  c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}
// Philox-4x32-10 (Salmon et al., SC'11): counter (quad index, entity), key = the seed
__device__ __forceinline__ void philox4(uint64_t quad, uint32_t entity, uint64_t seed, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)quad, (uint32_t)(quad >> 32), entity, 0x676e78u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = c[i];
}
// Flux._dropout_kernel(y, p, q) = y > p ? 1 / q : 0 with y uniform on [0, 1): 24 random bits -> a float on the grid k / 2^24
__device__ __forceinline__ float mask_of(uint32_t bits, float p, float scale) { return (float)(bits >> 8) * 0x1p-24f > p ? scale : 0.f; }

// MODE 0: out = m;  1: out = in .* m;  2: out += (m - 1) .* in.  One thread per four consecutive elements (one Philox block).
template <int MODE>
__global__ __launch_bounds__(256) void k_dropout(const float* __restrict__ in, float* __restrict__ out, size_t n, float p, float scale, uint64_t seed,
                                                 uint32_t entity, int vec) {
  const size_t quad = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t i0 = quad * 4;
  if (i0 >= n) return;
  uint32_t r[4];
  philox4(quad, entity, seed, r);
  float m[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) m[j] = mask_of(r[j], p, scale);
  if (vec && i0 + 4 <= n) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), o = a;
    if (MODE != 0) a = *reinterpret_cast<const float4*>(in + i0);
    if (MODE == 2) o = *reinterpret_cast<const float4*>(out + i0);
    if (MODE == 0) o = make_float4(m[0], m[1], m[2], m[3]);
    else if (MODE == 1) o = make_float4(a.x * m[0], a.y * m[1], a.z * m[2], a.w * m[3]);
    else o = make_float4(fmaf(m[0] - 1.f, a.x, o.x), fmaf(m[1] - 1.f, a.y, o.y), fmaf(m[2] - 1.f, a.z, o.z), fmaf(m[3] - 1.f, a.w, o.w));
    *reinterpret_cast<float4*>(out + i0) = o;
    return;
  }
  for (int j = 0; j < 4 && i0 + j < n; ++j) {
    if (MODE == 0) out[i0 + j] = m[j];
    else if (MODE == 1) out[i0 + j] = in[i0 + j] * m[j];
    else out[i0 + j] = fmaf(m[j] - 1.f, in[i0 + j], out[i0 + j]);
  }
}
}  // namespace

bool dropout_active(const gnx_dropout* d) { return d && d->p > 0.f; }

int32_t check_dropout(const gnx_dropout* d) {
  if (d && !(d->p >= 0.f && d->p <= 1.f)) return fail(GNX_ERR_INVALID_ARG, "Dropout: p must lie in [0, 1] (Flux.Dropout asserts 0 <= p <= 1)");
  return GNX_OK;
}

// mode as k_dropout's MODE; `in` unused for mode 0
int32_t launch_dropout(const gnx_dropout& d, int entity, size_t n, const float* in, float* out, int mode, hipStream_t s) {
  if (n == 0) return GNX_OK;
  const float scale = d.p < 1.f ? 1.f / (1.f - d.p) : 0.f;
  const int vec = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
  const dim3 grid((unsigned)(((n + 3) / 4 + 255) / 256));
  ProfScope ps("k_dropout", s);
  if (mode == 0) GNX_LAUNCH(k_dropout<0>, grid, dim3(256), 0, s, in, out, n, d.p, scale, d.seed, (uint32_t)entity, vec);
  else if (mode == 1) GNX_LAUNCH(k_dropout<1>, grid, dim3(256), 0, s, in, out, n, d.p, scale, d.seed, (uint32_t)entity, vec);
  else GNX_LAUNCH(k_dropout<2>, grid, dim3(256), 0, s, in, out, n, d.p, scale, d.seed, (uint32_t)entity, vec);
  GNX_HIP(hipGetLastError());
  return GNX_OK;
}

namespace {
struct TrainLayout { size_t core, l1, l2, hid, f, total; };
TrainLayout train_layout(const gnx_graphs* h, const gnx_core_params* p, int64_t R, size_t core_bytes) {
  const size_t rows[3] = {(size_t)R * h->E, (size_t)R * h->N, (size_t)R * h->G};
  const int d[3] = {p->block.de, p->block.dn, p->block.dg};
  size_t bmax = 0;
  for (int t = 0; t < 3; ++t) bmax = std::max(bmax, sizeof(float) * rows[t] * (size_t)std::max(d[t], 0));
  TrainLayout L{};
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += align_up(bytes, 256); return at; };
  L.core = take(core_bytes);
  L.l1 = take(bmax); L.l2 = take(bmax); L.hid = take(4 * bmax); L.f = take(bmax);
  L.total = o + 256;
  return L;
}
}  // namespace
}  // namespace gnx

using namespace gnx;

extern "C" {

int32_t gnx_dropout_mask(const gnx_dropout* d, int32_t entity, int64_t n, float* out, void* stream) {
  if (!d || (!out && n > 0) || n < 0 || entity < 0 || entity > 2) return fail(GNX_ERR_INVALID_ARG, "gnx_dropout_mask: NULL argument, negative length or entity outside 0..2");
  if (int32_t rc = check_dropout(d)) return rc;
  if (n == 0) return GNX_OK;  // (the mask of an entity without rows: a batch without edges)
  return launch_dropout(*d, entity, (size_t)n, nullptr, out, 0, (hipStream_t)stream);
}

size_t gnx_core_train_workspace_bytes(const gnx_graphs* h, const gnx_core_params* p, int64_t R) {
  if (!h || !p || R <= 0) return 0;
  const size_t core = gnx_core_workspace_bytes(h, p, R);
  if (core == 0) return 0;
  (void)gnx_ensure_wide_tables(h);  // the unfused FeedForward runs on the row-wise Dense launcher
  return train_layout(h, p, R, core).total;
}

int32_t gnx_core_forward_train(const gnx_graphs* h, const gnx_core_params* p, const gnx_dropout* dr, const float* ef, const float* nf, const float* gf,
                               int64_t R, float* ef_out, float* nf_out, float* gf_out, void* ws, size_t ws_bytes, uint32_t flags, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!h || !p) return fail(GNX_ERR_INVALID_ARG, "NULL handle or params");
  if (int32_t rc = check_dropout(dr)) return rc;
  const size_t core = gnx_core_workspace_bytes(h, p, R);
  if (core == 0) return fail(GNX_ERR_DIMS, "gnx_core_forward_train: gnx_core_workspace_bytes rejects these parameters (GNCore needs dims => dims, all(dims .> 0), n_replicas >= 1)");
  const TrainLayout L = train_layout(h, p, R, core);
  if (!ws || ws_bytes < (dropout_active(dr) ? L.total : core)) return fail(GNX_ERR_WORKSPACE, "workspace missing or smaller than gnx_core_train_workspace_bytes()");
  if (((uintptr_t)ws & 15) != 0) return fail(GNX_ERR_WORKSPACE, "workspace must be 16-byte aligned");
  char* base = static_cast<char*>(ws);
  // (everything that can refuse the call is checked BEFORE the forward writes an output — a refused call leaves the outputs alone)
  if (dropout_active(dr))
    for (int t = 0; t < 3; ++t)
      if (p->ff[t].fc2.act != GNX_ACT_IDENTITY) return fail(GNX_ERR_INVALID_ARG, "core training forward: fc2 must be identity");
  DeviceTurn turn(s, matrix_core_widths(p->block));
  int32_t rc = gnx_core_forward(h, p, ef, nf, gf, R, ef_out, nf_out, gf_out, base + L.core, core, flags, stream);
  if (rc || !dropout_active(dr)) return rc;
  FormScope forms(flags);
  PreparedScope prepared(p->prepared);  // (the recompute below runs the row-wise Dense launcher: no planes are looked up today, and none is missed if that changes)
  auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
  const size_t rows[3] = {(size_t)R * h->E, (size_t)R * h->N, (size_t)R * h->G};
  const int d[3] = {p->block.de, p->block.dn, p->block.dg};
  const float* x[3] = {ef, nf, gf};
  float* y[3] = {ef_out, nf_out, gf_out};
  for (int t = 0; t < 3; ++t) {
    if (rows[t] == 0) continue;
    if ((rc = launch_layernorm2(x[t], rows[t], d[t], p->ln1[t], p->ln2[t], p->eps, p->eps_mode, F(L.l1), F(L.l2), s))) return rc;
    if ((rc = launch_dense_rows(h, t, F(L.l2), d[t], p->ff[t].fc1, 4 * d[t], nullptr, nullptr, F(L.hid), R, s, "train_ff1"))) return rc;
    if ((rc = launch_dense_rows(h, t, F(L.hid), 4 * d[t], p->ff[t].fc2, d[t], nullptr, nullptr, F(L.f), R, s, "train_ff2"))) return rc;
    if ((rc = launch_dropout(*dr, t, rows[t] * (size_t)d[t], F(L.f), y[t], 2, s))) return rc;
  }
  return GNX_OK;
}

}  // extern "C"

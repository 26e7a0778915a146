/*
 * gn_oracle_c — plain-C fp32 restatement of GraphNets.jl's GNBlock / GNCore forward on packed CSC data.
 *
 * TEST INFRASTRUCTURE ONLY: the checker for tests/ and the `cpu_baseline` ("port") leg of bench.py.  Never
 * linked into, loaded by, or called from the product (libgnx.so / graphnets.jl_amd).
 * PARITY UNPINNED for absolute numerics (see oracle/gn_oracle.py header): the reference is Julia and cannot run
 * here; this file is validated against the float64 numpy restatement, which is pinned to the reference's
 * known-answer index fixtures (test/runtests.jl:487-508, :659-680).
 *
 * Semantics followed (reference file:line):
 *   edge input  [ef ; nf[src] ; nf[dst] ; gf[g]]            src/edgefninput.jl:1-47
 *   node input  [sum_{e->n} ef' ; nf ; gf[g]]               src/nodefninput.jl:1-24
 *   graph input [sum_e ef' ; sum_n nf' ; gf]                src/graphfninput.jl:1-13
 *   block order edge -> node (new ef', old nf, gf) -> graph src/gnblock.jl:63-69
 *   core        x + block(LN1 x) + FF(LN2 x)                src/gncore.jl:56-68, gnfeedforward.jl:27-40,
 *                                                           gngraphnorm.jl:9-26
 * Layout: features [R][T][D] row-major (== Julia (D,T,R) column-major); weights (out x in) column-major
 * (== Flux Dense.weight): W[k*out + j].  Indices int64, 0-based, rowval = global source node id.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { ACT_IDENTITY = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_SIGMOID = 3, ACT_GELU = 4 };

static inline float act_apply(float x, int act) {
  switch (act) {
    case ACT_RELU: return x > 0.f ? x : 0.f;
    case ACT_TANH: return tanhf(x);
    case ACT_SIGMOID: return 1.f / (1.f + expf(-x));
    case ACT_GELU: return 0.5f * x * (1.f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
    default: return x;
  }
}

/* y[j] = act(b[j] + sum_k W[k*out+j] * x[k]) over up to four concatenated segments */
static inline void dense_cat(const float* W, const float* b, int act, int out, const float* const seg[4],
                             const int len[4], float* y) {
  for (int j = 0; j < out; ++j) y[j] = b ? b[j] : 0.f;
  int k0 = 0;
  for (int s = 0; s < 4; ++s) {
    const float* x = seg[s];
    for (int k = 0; k < len[s]; ++k) {
      const float xv = x[k];
      const float* w = W + (size_t)(k0 + k) * out;
      for (int j = 0; j < out; ++j) y[j] += w[j] * xv;
    }
    k0 += len[s];
  }
  for (int j = 0; j < out; ++j) y[j] = act_apply(y[j], act);
}

typedef struct {
  int de, dn, dg, oe, on, og;
  const float *We, *be, *Wn, *bn, *Wg, *bg;
  int act_e, act_n, act_g;
} gn_block_params;

int gn_oracle_block_forward_f32(int64_t N, int64_t E, int64_t G, const int64_t* colptr, const int64_t* rowval,
                                const int64_t* node_off, const int64_t* edge_off, const gn_block_params* p,
                                const float* ef, const float* nf, const float* gf, int64_t R, float* ef_out,
                                float* nf_out, float* gf_out, int nthreads) {
  const int de = ef ? p->de : 0, dn = nf ? p->dn : 0, dg = gf ? p->dg : 0;
  const int oe = p->oe, on = p->on, og = p->og;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  /* ef'/nf' are needed as intermediates even when the caller does not want them */
  float* ef_tmp = ef_out ? ef_out : (float*)malloc(sizeof(float) * (size_t)(R * E * oe + 1));
  float* nf_tmp = nf_out ? nf_out : (float*)malloc(sizeof(float) * (size_t)(R * N * on + 1));
  for (int64_t r = 0; r < R; ++r) {
    const float* efr = ef ? ef + (size_t)r * E * de : NULL;
    const float* nfr = nf ? nf + (size_t)r * N * dn : NULL;
    const float* gfr = gf ? gf + (size_t)r * G * dg : NULL;
    float* efo = ef_tmp + (size_t)r * E * oe;
    float* nfo = nf_tmp + (size_t)r * N * on;
    /* edge + node update, parallel over destination nodes (their in-edges are contiguous in CSC) */
#pragma omp parallel
    {
      float* agg = (float*)malloc(sizeof(float) * (size_t)(oe + 1));
#pragma omp for schedule(dynamic, 256)
      for (int64_t n = 0; n < N; ++n) {
        /* graph of node n: binary search in node_off */
        int64_t lo = 0, hi = G;
        while (hi - lo > 1) { int64_t mid = (lo + hi) >> 1; if (node_off[mid] <= n) lo = mid; else hi = mid; }
        const float* gfg = gfr ? gfr + (size_t)lo * dg : NULL;
        for (int j = 0; j < oe; ++j) agg[j] = 0.f;
        for (int64_t e = colptr[n]; e < colptr[n + 1]; ++e) {
          const float* seg[4] = {efr ? efr + (size_t)e * de : NULL, nfr ? nfr + (size_t)rowval[e] * dn : NULL,
                                 nfr ? nfr + (size_t)n * dn : NULL, gfg};
          const int len[4] = {de, dn, dn, dg};
          float* y = efo + (size_t)e * oe;
          dense_cat(p->We, p->be, p->act_e, oe, seg, len, y);
          for (int j = 0; j < oe; ++j) agg[j] += y[j];
        }
        const float* seg[4] = {agg, nfr ? nfr + (size_t)n * dn : NULL, gfg, NULL};
        const int len[4] = {oe, dn, dg, 0};
        dense_cat(p->Wn, p->bn, p->act_n, on, seg, len, nfo + (size_t)n * on);
      }
      free(agg);
    }
    if (og > 0 && gf_out) {
#pragma omp parallel for schedule(dynamic, 1)
      for (int64_t g = 0; g < G; ++g) {
        float* se = (float*)calloc((size_t)(oe + on + 2), sizeof(float));
        float* sn = se + oe;
        for (int64_t e = edge_off[g]; e < edge_off[g + 1]; ++e)
          for (int j = 0; j < oe; ++j) se[j] += efo[(size_t)e * oe + j];
        for (int64_t n = node_off[g]; n < node_off[g + 1]; ++n)
          for (int j = 0; j < on; ++j) sn[j] += nfo[(size_t)n * on + j];
        const float* seg[4] = {se, sn, gfr ? gfr + (size_t)g * dg : NULL, NULL};
        const int len[4] = {oe, on, dg, 0};
        dense_cat(p->Wg, p->bg, p->act_g, og, seg, len, gf_out + ((size_t)r * G + g) * og);
        free(se);
      }
    }
  }
  if (!ef_out) free(ef_tmp);
  if (!nf_out) free(nf_tmp);
  return 0;
}

/* LayerNorm over the feature dim of `rows` rows of width d: eps_mode 0 = (x-mu)/(sigma+eps) (Flux 0.14
 * normalise), 1 = (x-mu)/sqrt(var+eps).  Writes gamma1*xhat+beta1 and gamma2*xhat+beta2 (gn1 and gn2 share
 * the statistics of x; gncore.jl:56-59). */
static void layernorm2(const float* x, int64_t rows, int d, const float* g1, const float* b1, const float* g2,
                       const float* b2, float eps, int eps_mode, float* y1, float* y2) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < rows; ++i) {
    const float* xi = x + (size_t)i * d;
    float mu = 0.f;
    for (int k = 0; k < d; ++k) mu += xi[k];
    mu /= (float)d;
    float var = 0.f;
    for (int k = 0; k < d; ++k) var += (xi[k] - mu) * (xi[k] - mu);
    var /= (float)d;
    const float inv = eps_mode == 0 ? 1.f / (sqrtf(var) + eps) : 1.f / sqrtf(var + eps);
    for (int k = 0; k < d; ++k) {
      const float xh = (xi[k] - mu) * inv;
      y1[(size_t)i * d + k] = g1[k] * xh + b1[k];
      y2[(size_t)i * d + k] = g2[k] * xh + b2[k];
    }
  }
}

typedef struct {
  gn_block_params block; /* dims => dims */
  const float *ln1_gamma[3], *ln1_beta[3], *ln2_gamma[3], *ln2_beta[3];
  const float *W1[3], *b1[3], *W2[3], *b2[3];
  float eps;
  int eps_mode;
} gn_core_params;

int gn_oracle_core_forward_f32(int64_t N, int64_t E, int64_t G, const int64_t* colptr, const int64_t* rowval,
                               const int64_t* node_off, const int64_t* edge_off, const gn_core_params* p,
                               const float* ef, const float* nf, const float* gf, int64_t R, float* ef_out,
                               float* nf_out, float* gf_out, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  const int d[3] = {p->block.de, p->block.dn, p->block.dg};
  const int64_t rows[3] = {R * E, R * N, R * G};
  const float* x[3] = {ef, nf, gf};
  float* out[3] = {ef_out, nf_out, gf_out};
  float *l1[3], *l2[3];
  for (int t = 0; t < 3; ++t) {
    l1[t] = (float*)malloc(sizeof(float) * (size_t)(rows[t] * d[t] + 1));
    l2[t] = (float*)malloc(sizeof(float) * (size_t)(rows[t] * d[t] + 1));
    layernorm2(x[t], rows[t], d[t], p->ln1_gamma[t], p->ln1_beta[t], p->ln2_gamma[t], p->ln2_beta[t], p->eps,
               p->eps_mode, l1[t], l2[t]);
  }
  gn_oracle_block_forward_f32(N, E, G, colptr, rowval, node_off, edge_off, &p->block, l1[0], l1[1], l1[2], R,
                              out[0], out[1], out[2], nthreads);
  for (int t = 0; t < 3; ++t) {
    const int dd = d[t], hh = 4 * dd;
#pragma omp parallel
    {
      float* h = (float*)malloc(sizeof(float) * (size_t)(hh + dd + 1));
      float* y = h + hh;
#pragma omp for schedule(static)
      for (int64_t i = 0; i < rows[t]; ++i) {
        const float* seg1[4] = {l2[t] + (size_t)i * dd, NULL, NULL, NULL};
        const int len1[4] = {dd, 0, 0, 0};
        dense_cat(p->W1[t], p->b1[t], ACT_RELU, hh, seg1, len1, h);
        const float* seg2[4] = {h, NULL, NULL, NULL};
        const int len2[4] = {hh, 0, 0, 0};
        dense_cat(p->W2[t], p->b2[t], ACT_IDENTITY, dd, seg2, len2, y);
        for (int k = 0; k < dd; ++k) out[t][(size_t)i * dd + k] += x[t][(size_t)i * dd + k] + y[k];
      }
      free(h);
    }
    free(l1[t]);
    free(l2[t]);
  }
  return 0;
}

int gn_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

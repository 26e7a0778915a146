/*
 * gnx.h — C ABI of libgnx.so: MI355X (gfx950) native GNBlock / GNCore forward for GraphNets.jl.
 *
 * The reference has no FFI; its boundary for this path is the Julia callable API exported at
 * src/GraphNets.jl:12-50.  Each entry point below names the reference interface it replaces; the Julia-side
 * `ccall` binding a maintainer would add is shown in INTEGRATION.md (and shipped in julia/GraphNetsHIP.jl).
 *
 * Conventions
 *   - Every function returns int32 status: 0 = GNX_OK; < 0 = invalid argument (mirrors an `@assert` of the
 *     reference, src/checks.jl, src/batch.jl:54-56, src/gnblock.jl:48-49); > 0 = a hipError_t value.  A
 *     message is available from gnx_last_error() (thread-local).  Nothing throws across the ABI.
 *   - Feature buffers are DEVICE pointers, fp32, packed rows: ef [R][E][DE], nf [R][N][DN], gf [R][G][DG]
 *     (row-major) — byte-identical to Julia's column-major (D, T, R) arrays and, for vector-of-graphs
 *     batches (R = 1), to flatunpaddedef / flatunpaddednf (src/views.jl:80-98).  R = number of replicas of
 *     the graph structure: the data batch size B of a shared-adjacency batch (src/batch.jl:66), 1 otherwise.
 *     E, N, G are totals over the graphs of the handle.  NULL <=> `nothing`; the (DE, 0) edge features of a batch WITHOUT
 *     edges have no bytes and may be NULL too (their width still counts: sums over them are rows of DE zeros).
 *   - Edge order inside a graph = order of the ones of vec(A) column-major (src/pad.jl:30): sorted by
 *     destination j then source i, A[i,j] = 1 meaning i -> j (src/gngraphbatch.jl:194-211).
 *   - Dense weights are (out x in) column-major = Flux `Dense.weight` bytes: W[k*out + j]; device pointers.
 *   - The caller owns every buffer; the library owns only gnx_graphs handles (and, for gnx_model, the model's intermediate
 *     tensors).  `stream` is a hipStream_t (NULL = default stream); calls are asynchronous on it and hipGraph-capturable (no
 *     allocation, no sync) with four documented exceptions that happen once, outside any capture: the first use of a width
 *     set that needs a run-time specialised kernel (see gnx_jit_*), the first backward / edge-collapsing call on a handle
 *     (builds the CSR / collapse tables of the handle), the handle's matrix-core tables (widths from 32: built by the
 *     gnx_*_workspace_bytes query of the layer — every caller runs that query before a forward, outside a capture; a forward,
 *     backward or gnx_fn_input(kind 0) that still finds them missing builds them itself, or, when its stream is being captured,
 *     fails with GNX_ERR_INVALID_ARG and a message saying so — not remembered: the next call outside the capture succeeds), and
 *     gnx_model_forward (which manages its own hipGraph).
 *   - The library uses the calling thread's current HIP device; a handle lives on the device it was created on.
 *   - Output buffers must not overlap input buffers or the workspace (the kernels read inputs while they write outputs; the training forward
 *     re-reads x after the outputs exist).  The reference's layers are out-of-place as well (every update allocates its result).
 */
#ifndef GNX_H
#define GNX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNX_VERSION 130 /* 0.1.3: gnx_block_forward_steps; no default path on the fp32 matrix instruction; 120: training-mode Dropout (gnx_dropout, gnx_core_forward_train / _backward_train); 110: gnx_profile_entry is 72 bytes, per-call arithmetic flags, prepared parameters, GNX_FLAG_DIST_NO_GATHER */

#if defined(__GNUC__)
#define GNX_API __attribute__((visibility("default")))
#else
#define GNX_API
#endif

/* status codes (negative = argument errors, named after the reference assertion they mirror) */
#define GNX_OK 0
#define GNX_ERR_INVALID_ARG (-1)   /* NULL where a pointer is required, negative size, bad enum            */
#define GNX_ERR_NO_GRAPHS (-2)     /* checks.jl:8    @assert length(adj_mats) > 0                           */
#define GNX_ERR_ADJ_SHAPE (-3)     /* checks.jl:11   adjacency must be square (N x N), N >= 1               */
#define GNX_ERR_ADJ_VALUE (-4)     /* gngraphbatch.jl:207 / pad.jl:30  entries must be 0 or 1               */
#define GNX_ERR_ALL_NOTHING (-5)   /* batch.jl:56    ef, nf, gf all `nothing`                               */
#define GNX_ERR_DIMS (-6)          /* gnblock.jl:48-49, gnfeedforward.jl:18, gngraphnorm.jl:10              */
#define GNX_ERR_CSC (-7)           /* malformed CSC: colptr not monotone, rowval out of range / unsorted    */
#define GNX_ERR_WORKSPACE (-8)     /* workspace NULL or smaller than gnx_*_workspace_bytes()                */
#define GNX_ERR_TOO_LARGE (-9)     /* N or E does not fit the int32 device indices                          */
#define GNX_ERR_COUNT_MISMATCH (-10) /* checks.jl:41-46 size(ef,2)==num_edges, size(nf,2)==num_nodes — raised by the host
                                      * shims (the ABI sees pointers only); reserved here so the codes stay in one place  */

/* activations (Flux/NNlib): identity is the GNBlock default (gnblock.jl:55-60); relu is used by FeedForward */
#define GNX_ACT_IDENTITY 0
#define GNX_ACT_RELU 1
#define GNX_ACT_TANH 2
#define GNX_ACT_SIGMOID 3
#define GNX_ACT_GELU 4 /* NNlib.gelu (tanh form) */

/* element kinds of dense adjacency input */
#define GNX_ELEM_U8 0
#define GNX_ELEM_I32 1
#define GNX_ELEM_I64 2
#define GNX_ELEM_F32 3
#define GNX_ELEM_F64 4

/* forward flags */
#define GNX_FLAG_FORCE_GENERIC 0x1u /* use the dimension-generic kernels even when a specialised path exists */
#define GNX_FLAG_NO_MFMA 0x2u       /* never pick the MFMA path                                               */
#define GNX_FLAG_DEFER_GRAPH_UPDATE 0x4u /* gnx_block_forward stops after the edge+node update and leaves the per-tile
                                          * partial sums in the workspace; gnx_block_graph_update finishes gf' later   */

/* FORMS of the forward, selected PER CALL (every gnx_*_forward / gnx_model_forward / gnx_dist_* takes them in `flags`; the reference's layers
 * are stateless values, src/gnblock.jl:63-69 — which arithmetic a call runs is an argument of the call, not a property of the process).
 * Each has an environment variable of the same name (GNX_FFN_FP32=1 ...) that is read ONCE per process, at the first library call, and OR-ed
 * into every call's flags as the process-wide default (gnx_default_flags() returns that mask); the library never calls getenv() on a forward. */
#define GNX_FLAG_FFN_FP32 0x20u   /* wide GNCore FeedForwards on the fp32 matrix instruction (k_ffn_fused) instead of six bf16 terms (k_ffn_x6) */
#define GNX_FLAG_EDGE_FP32 0x40u  /* projected edge update and node projections on the fp32 matrix instruction (k_rows_gemm) instead of six bf16 terms */
#define GNX_FLAG_FP32_MFMA (GNX_FLAG_FFN_FP32 | GNX_FLAG_EDGE_FP32) /* every matrix product of the call on the fp32 matrix instruction */
#define GNX_FLAG_PROJ_FP32 0x80u          /* the node-side kernels alone (node projections, node update) on the fp32 instruction            */
#define GNX_FLAG_EDGE_NARROW_FP32 0x100u  /* the 128 -> (<= 32) edge update alone on the fp32 instruction                                  */
/* diagnostic forms — same results (bit-identical where the header says so), kept for A/B runs and for the tests that compare two forms */
#define GNX_FLAG_NO_LN_FUSE 0x200u       /* wide GNCore: materialise gn1 / gn2 (k_layernorm2) instead of normalising on load               */
#define GNX_FLAG_LN_STATS_PASS 0x400u    /* wide GNCore: row statistics of ef by the statistics pass instead of in the six-term kernels    */
#define GNX_FLAG_CORE_EDGE_SPLIT 0x800u  /* wide GNCore: edge update and edge FeedForward as two launches                                  */
#define GNX_FLAG_NO_FORK 0x1000u         /* wide GNCore: everything on the caller's stream (no side stream for the graph level)            */
#define GNX_FLAG_NO_PACK 0x2000u         /* narrow block on small graphs: graph update as its own launch (k_graph_t)                       */
#define GNX_FLAG_NO_FFE 0x4000u          /* narrow GNCore: edge FeedForward in k_core_post3 instead of the block kernel's edge lanes       */
#define GNX_FLAG_NO_JIT 0x8000u          /* never specialise a kernel at run time (generic kernels instead); env GNX_JIT=0 / GNX_NO_JIT=1   */
#define GNX_FLAG_EDGE_N 0x10000u         /* opt-in: k_edge_n (source rows gathered raw, K = 128 + 64; csrc/gnx_edge_n.hip) for the edge update */
#define GNX_FLAG_LN_ON_LOAD 0x20000u     /* wide GNCore: normalise on load from a row-statistics table also where a GENERAL kernel (k_rows_gemm, k_ffn_fused)
                                          * consumes it — rounds 2-5's default; round 6 materialises those LayerNorms instead (csrc/gnx_forward.hip,
                                          * profiles/r06_overlap_hazard.log); the same formula, exact too since that branch is guarded      */
#define GNX_FLAG_FORMS_MASK 0x3ffe0u
GNX_API uint32_t gnx_default_flags(void); /* the forms the environment switched on for this process */

typedef struct gnx_graphs gnx_graphs; /* opaque; replaces GNGraphBatch (src/gngraphbatch.jl:1-54) */

typedef struct gnx_graphs_info {
  int64_t n_graphs;        /* G = length(adj_mats)                                   */
  int64_t n_nodes;         /* sum of N_g                                             */
  int64_t n_edges;         /* sum of E_g                                             */
  int64_t node_block_size; /* PN = max N_g           (gngraphbatch.jl:35)            */
  int64_t edge_block_size; /* PN^2                   (gngraphbatch.jl:36)            */
  int64_t n_tiles;         /* work tiles (node ranges) the kernels iterate over      */
  int64_t max_in_degree;
  int32_t device;
  int32_t reserved;
} gnx_graphs_info;

/* One Flux `Dense(in => out, act)`: y = act.(W*x .+ b).  bias may be NULL (= zeros).
 * `kind` (0 everywhere but inside a gnx_chain): GNX_LAYER_LAYERNORM makes the entry a Flux `LayerNorm(d)` layer value of a Chain — weight =
 * gamma [d], bias = beta [d], act = identity, its width = the width of its input, eps = 1e-5 (Flux's default), (x - mean) / (sigma + eps)
 * (Flux 0.14 `normalise`; `| GNX_LAYER_LN_SQRT_EPS`: / sqrt(sigma^2 + eps)).  gnx_block_params' three functions must be Dense (kind 0). */
#define GNX_LAYER_DENSE 0
#define GNX_LAYER_LAYERNORM 1
#define GNX_LAYER_LN_SQRT_EPS 0x100
typedef struct gnx_dense {
  const float* weight; /* (out x in) column-major, device; LayerNorm entry: gamma [d] */
  const float* bias;   /* (out), device, or NULL;          LayerNorm entry: beta [d]  */
  int32_t act;         /* GNX_ACT_*                       */
  int32_t kind;        /* GNX_LAYER_* (was `reserved`: 0 = Dense) */
} gnx_dense;

/* Prepared parameters (opaque): a layer's weight blocks in the forms the matrix-core kernels stage (bf16 planes of the exact three-way
 * split, transposed, slot-permuted) — made ONCE per upload of the weights by gnx_block_prepare / gnx_core_prepare below instead of by
 * preparation launches in front of every forward (`model |> device` happens once: examples/sort/sort.jl:29,89). */
typedef struct gnx_prepared gnx_prepared;

/* GNBlock((de,dn,dg) => (oe,on,og)) (src/gnblock.jl:47-61): edgefn in = de+2dn+dg, nodefn in = oe+dn+dg
 * (order agg, nf, gf: nodefninput.jl:2-6), graphfn in = oe+on+dg (order edges, nodes, gf: graphfninput.jl:2-6). */
typedef struct gnx_block_params {
  int32_t de, dn, dg; /* input widths; 0 <=> that input is `nothing` */
  int32_t oe, on, og; /* output widths; 0 <=> that output is `nothing` (gnblock.jl:71-78) */
  gnx_dense edgefn, nodefn, graphfn;
  const gnx_prepared* prepared; /* gnx_block_prepare's result for THESE weights, or NULL (the forward then prepares them per call) */
} gnx_block_params;

typedef struct gnx_layernorm { /* Flux LayerNorm(d): gamma .* xhat .+ beta */
  const float* gamma;
  const float* beta;
} gnx_layernorm;

typedef struct gnx_ffn { /* FeedForward (gnfeedforward.jl:27-31): Dense(d=>4d, relu), Dense(4d=>d); Dropout = identity */
  gnx_dense fc1, fc2;
} gnx_ffn;

/* GNCore(dims) (src/gncore.jl:46-59): y = x + block(gn1(x)) + ffwd(gn2(x)); index 0/1/2 = edge/node/graph. */
typedef struct gnx_core_params {
  gnx_block_params block; /* dims => dims */
  gnx_layernorm ln1[3], ln2[3];
  gnx_ffn ff[3];
  float eps;        /* 1e-5 */
  int32_t eps_mode; /* 0: (x-mu)/(sigma+eps) (Flux 0.14 normalise);  1: (x-mu)/sqrt(sigma^2+eps) */
  const gnx_prepared* prepared; /* gnx_core_prepare's result for THESE weights (block and FeedForwards), or NULL; block.prepared is ignored */
} gnx_core_params;

typedef struct gnx_profile_entry {
  char name[48];
  int64_t launches; /* times the named scope was entered (one per step for most scopes)                  */
  double total_ms;  /* sum of the dispatch-timestamp durations of every kernel launched inside the scope */
  int64_t kernels;  /* kernels launched inside the scope (the graph level of a wide block: three per entry) */
} gnx_profile_entry;

/* ---- library ---- */
GNX_API int32_t gnx_version(void);
GNX_API const char* gnx_last_error(void);

/* ---- graph handles: replace batchgraphs / GNGraphBatch(adj_mats) (src/batch.jl:66-67, src/gngraphbatch.jl:33-54) ---- */

/* adj[g] points at an n_nodes[g] x n_nodes[g] matrix of `elem_kind`, HOST memory; `row_major` = 0 for Julia
 * (column-major) input, 1 for C/numpy.  Entries must be exactly 0 or 1. */
GNX_API int32_t gnx_graphs_create_dense(const void* const* adj, const int64_t* n_nodes, int64_t n_graphs, int32_t elem_kind,
                                int32_t row_major, gnx_graphs** out);

/* The same batch from ONE buffer: the matrices one after the other (graph g: n_g x n_g elements of elem_kind), adj_bytes = sum(n_g^2) *
 * sizeof(element) — checked; nothing is read past it.  on_device = 0: HOST memory (a pinned buffer travels as one DMA, a pageable one through
 * the library's pinned staging pair); on_device = 1: DEVICE memory of the current device (no copy; the scan runs on the NULL
 * stream: the buffer must be COMPLETE when the call is made — a producer on a non-blocking stream is synchronised by the caller first, as the
 * Python mirror does).  What a data loader holds and what the
 * bindings call (GNGraphBatch.from_dense_packed): one pointer instead of G — building G pointers costs a Python / Julia host ~1.5 us each. */
GNX_API int32_t gnx_graphs_create_dense_packed(const void* adj_cat, int64_t adj_bytes, const int64_t* n_nodes, int64_t n_graphs, int32_t elem_kind,
                                       int32_t row_major, int32_t on_device, gnx_graphs** out);

/* Per-graph CSC (Julia SparseMatrixCSC colptr / rowval, HOST memory): colptr[g] has n_nodes[g]+1 entries,
 * rowval[g] the local source index of every edge, sorted strictly increasing inside a column; index_base is 1
 * for Julia arrays, 0 for C.  The nz order of CSC *is* the reference edge order.  (API extension: the reference
 * only accepts dense matrices, which cannot hold BASELINE configs 2-5.) */
GNX_API int32_t gnx_graphs_create_csc(const int64_t* const* colptr, const int64_t* const* rowval, const int64_t* n_nodes,
                              int64_t n_graphs, int32_t index_base, gnx_graphs** out);
/* the same batch from TWO arrays: colptr_cat = the graphs' colptr arrays one after the other (n_g + 1 entries each, every one starting at
 * index_base), rowval_cat = their rowval arrays one after the other.  One call and two pointers instead of 2 G pointers: what a host
 * pays per graph to build pointer arrays (8 ms for 4096 graphs through ctypes) is the larger part of batch() on many small graphs. */
GNX_API int32_t gnx_graphs_create_csc_packed(const int64_t* colptr_cat, const int64_t* rowval_cat, const int64_t* n_nodes, int64_t n_graphs,
                                     int32_t index_base, gnx_graphs** out);

/* the packed form WITH its array lengths and index width — what a host binding should call: colptr_cat / rowval_cat hold colptr_len /
 * rowval_len indices of index_bits (32 or 64) bits each; the call fails with GNX_ERR_INVALID_ARG unless colptr_len = sum(n_nodes) + n_graphs
 * and rowval_len = the edges the colptr arrays announce, and never reads past either length whatever the arrays contain. */
GNX_API int32_t gnx_graphs_create_csc_cat(const void* colptr_cat, int64_t colptr_len, const void* rowval_cat, int64_t rowval_len,
                                  const int64_t* n_nodes, int64_t n_graphs, int32_t index_base, int32_t index_bits, gnx_graphs** out);

GNX_API int32_t gnx_graphs_destroy(gnx_graphs* h);
GNX_API int32_t gnx_graphs_get_info(const gnx_graphs* h, gnx_graphs_info* out);
/* host copies, 0-based: node_off[G+1], edge_off[G+1] (what unpadnf/unpadef/efview/nfview index with,
 * src/unpad.jl:1-25, src/views.jl:6-98); any pointer may be NULL */
GNX_API int32_t gnx_graphs_get_offsets(const gnx_graphs* h, int64_t* node_off, int64_t* edge_off);
/* host copies, 0-based global CSC: colptr[N+1], rowval[E] (global source node id) */
GNX_API int32_t gnx_graphs_get_csc(const gnx_graphs* h, int64_t* colptr, int64_t* rowval);

/* diagnostic: one of the handle's DEVICE tables copied to the host as it is — which = 0 colptr [N+1] int32, 1 rowval [E] int32 (global source
 * ids), 2 node_off, 3 edge_off, 4 tile_off [G+1] int32, 5 workgroup tiles (32-byte records {n0, n1, e0, e1, g, win0, win1, flags}), 6 wtile_off,
 * 7 wave tiles, 8 packs [n_packs][8] int32; 9..18 the matrix-core path's tables (built on the spot if they are not yet: 9 edge tiles, 10 node
 * tiles, 11 graph tiles, 12 / 13 their per-graph offsets, 14 destination of every edge, 15 first partial-sum row of every aggregation chunk,
 * 16 / 17 / 18 per node: its partial-sum row, the chunks its in-edges run through, its first chunk), 19 five int64 {partial-sum rows, edge tiles
 * with a wide destination span, edge / node / graph tile counts}.  *bytes = the table's size (out may be NULL to ask for it).  Large batches given as CSC are
 * validated and tiled by kernels (env GNX_BUILD_CSC_DEVICE=0: on the host); the two builders' tables are bit-identical (tests/test_gpu_build.py). */
GNX_API int32_t gnx_graphs_get_table(const gnx_graphs* h, int32_t which, void* out, int64_t capacity_bytes, int64_t* bytes);

/* ---- prepared parameters ----
 * gnx_block_prepare / gnx_core_prepare read the DEVICE weights the descriptor points at (on `stream`, asynchronously) and return an object
 * to put into the descriptor's `prepared` field.  A forward whose descriptor carries it launches no preparation kernel (config 4: nine
 * launches, ~45 us, per forward otherwise); outputs are bit-identical either way.  The planes are looked up by the weight POINTERS the
 * forward is handed: a prepared object made from other weights, or on another device, is simply not used.  Contract: the parameters' VALUES
 * must not change while a prepared object made from them is in use — after an optimiser step call gnx_prepared_refresh (same stream
 * order as the update), and destroy the object before the parameters are freed.  "Parameters" is the weights and, for a core with 128-wide
 * edges, gn1 / gn2's gamma and beta of the edges and the edge FeedForward's first bias: the one-launch form of the core's edge rows has
 * them folded into its planes ((gamma . W)^T xhat + W^T beta; the rows are normalised and split once).  Widths without a six-term kernel prepare nothing (an empty
 * object).  gnx_model_create prepares the layers whose descriptors carry none (see gnx_model_refresh_weights). */
GNX_API int32_t gnx_block_prepare(const gnx_block_params* p, void* stream, gnx_prepared** out);
GNX_API int32_t gnx_core_prepare(const gnx_core_params* p, void* stream, gnx_prepared** out);
GNX_API int32_t gnx_prepared_refresh(gnx_prepared* q, void* stream);
GNX_API int32_t gnx_prepared_destroy(gnx_prepared* q);
GNX_API int64_t gnx_prepared_bytes(const gnx_prepared* q); /* device memory the object holds */

/* ---- forward: replaces (m::GNBlock)(x) (src/gnblock.jl:63-69) ---- */
GNX_API size_t gnx_block_workspace_bytes(const gnx_graphs* h, const gnx_block_params* p, int64_t n_replicas);
GNX_API int32_t gnx_block_forward(const gnx_graphs* h, const gnx_block_params* p, const float* ef, const float* nf,
                          const float* gf, int64_t n_replicas, float* ef_out, float* nf_out, float* gf_out,
                          void* workspace, size_t workspace_bytes, uint32_t flags, void* stream);

/* Second phase of a deferred block forward: gf'[g] = graphfn([sum_e ef' ; sum_n nf' ; gf_g]) (src/gnblock.jl:67,
 * src/graphfninput.jl:1-13) from the partial sums a gnx_block_forward(..., GNX_FLAG_DEFER_GRAPH_UPDATE, ...) call left
 * in `workspace` (same handle, params, n_replicas, flags and workspace; the workspace must not be reused in between).
 * It is a few KB of work that only the caller of gf' waits for: launched on a second stream (after an event recorded
 * behind the first phase) it overlaps the next batch's edge/node update instead of sitting on the critical path. */
GNX_API int32_t gnx_block_graph_update(const gnx_graphs* h, const gnx_block_params* p, const float* gf, int64_t n_replicas,
                               float* gf_out, void* workspace, size_t workspace_bytes, uint32_t flags, void* stream);

/* ---- the graph update of call i inside call i + 1: loops over many batches of the same graphs (serving, training) -------------------------
 * A block forward at README widths is one ~19-us kernel plus a second launch that finishes gf' from a few KB of partial sums (~4 us, what an
 * EMPTY launch costs, + the boundary).  gnx_block_forward_chained runs this call's edge + node update and — in workgroups at the FRONT of
 * the same launch — the graph update that the previous chained call left pending (`prev`: that call's workspace, gf and gf_out, as the
 * library recorded them; NULL or zeroed: nothing pending).  On return *pending describes THIS call's pending graph update: its gf_out is
 * valid only after the next chained call on this stream or after gnx_block_graph_update(h, p, pending->gf, R, pending->gf_out,
 * pending->workspace, pending->workspace_bytes, flags, stream) (the flush; pending->workspace == NULL: nothing to flush).  Same handle,
 * params, n_replicas and flags in consecutive calls; the pending call's workspace and gf_out must differ from this call's (two alternating
 * buffer sets).  Where the two-launch form is not what runs (matrix-core / generic kernels, run-time specialised widths, batches of small
 * graphs whose graph update already runs inside the block kernel) the call finishes `prev` and this call the plain way and leaves nothing
 * pending.  Results are bit-identical to gnx_block_forward's (the same graph_update_rows over the same partial rows). */
typedef struct gnx_pending_update {
  const void* workspace;
  size_t workspace_bytes;
  const float* gf;
  float* gf_out;
} gnx_pending_update;
GNX_API int32_t gnx_block_forward_chained(const gnx_graphs* h, const gnx_block_params* p, const float* ef, const float* nf, const float* gf,
                                  int64_t n_replicas, float* ef_out, float* nf_out, float* gf_out, void* workspace, size_t workspace_bytes,
                                  uint32_t flags, void* stream, const gnx_pending_update* prev, gnx_pending_update* pending);

/* ---- a LOOP over batches as ONE call (round 6): what `for x in batches; y = block(x); end` is in a serving / evaluation loop over resident
 * batches of the same graphs (examples/sort/sort.jl:99-108 walks its batches this way).  Exactly n_steps gnx_block_forward calls in order —
 * the same kernels over the same rows, outputs bit-identical — but the library KNOWS the next step exists, so where the two-launch narrow
 * form runs it uses the chained form above by itself: step i's graph update rides at the front of step i + 1's launch, the last step's is
 * flushed before the call returns its work to the stream (ONE launch per step + one flush instead of two launches per step: 22.2 vs 25.1
 * us/step on BASELINE configs[1]).  Every output of every step is complete once the work this call enqueued is complete.  Consecutive
 * steps must not share a workspace or a gf_out (step i + 1 starts while step i's graph update is pending): a step that does share them with
 * its predecessor is simply run unchained.  Other paths (matrix-core / generic kernels, batches of small graphs): n_steps plain forwards.
 * Capture-safe like gnx_block_forward (bench.py captures K steps into one hipGraph). */
typedef struct gnx_block_step {
  const float* ef;
  const float* nf;
  const float* gf;
  float* ef_out;
  float* nf_out;
  float* gf_out;
  void* workspace;
  size_t workspace_bytes; /* >= gnx_block_workspace_bytes */
} gnx_block_step;
GNX_API int32_t gnx_block_forward_steps(const gnx_graphs* h, const gnx_block_params* p, const gnx_block_step* steps, int64_t n_steps,
                                int64_t n_replicas, uint32_t flags, void* stream);

/* ---- GNBlock with Flux `Chain`s of Dense layers as update functions (src/gnblock.jl:1-6: edgefn / nodefn / graphfn are
 * arbitrary Chains; the constructor's default is Chain(Dense), which is what gnx_block_forward fuses).  widths[i] = output
 * width of layer i; the input width of layer 0 is fixed by the block (de+2dn+dg / oe+dn+dg / oe+on+dg with oe, on = the LAST
 * widths of the edge / node chains); a chain with n_layers = 0 or a last width of 0 <=> that output is `nothing`.
 * The edge function's first layer runs fused with getedgefninput (the fast block kernels); every further layer is a row-wise
 * Dense on the matrix-core GEMM kernel.  A layer entry may also be a `LayerNorm(d)` layer value (gnx_dense.kind = GNX_LAYER_LAYERNORM,
 * `Chain(Dense(a => d, relu), LayerNorm(d), Dense(d => b))`: normalised rows of the layer in front, its width = that layer's) — anywhere in a chain;
 * as the edge function's FIRST layer it runs behind an identity Dense that the library puts in front (the fused launch then writes the
 * function input itself: K_e x K_e floats more of workspace).  The gradient entries of a LayerNorm layer are (gamma, beta).
 * Backward: gnx_chain_block_backward below. */
typedef struct gnx_chain {
  const gnx_dense* layers; /* [n_layers] host array of layer descriptors (device weight pointers inside) */
  const int32_t* widths;   /* [n_layers] host array */
  int32_t n_layers;
  int32_t reserved;
} gnx_chain;
typedef struct gnx_chain_block_params {
  int32_t de, dn, dg; /* input widths; 0 <=> nothing */
  int32_t reserved;
  gnx_chain edgefn, nodefn, graphfn;
} gnx_chain_block_params;
GNX_API size_t gnx_chain_block_workspace_bytes(const gnx_graphs* h, const gnx_chain_block_params* p, int64_t n_replicas);
GNX_API int32_t gnx_chain_block_forward(const gnx_graphs* h, const gnx_chain_block_params* p, const float* ef, const float* nf, const float* gf,
                                int64_t n_replicas, float* ef_out, float* nf_out, float* gf_out, void* workspace, size_t workspace_bytes,
                                uint32_t flags, void* stream);

/* ---- backward of the block: what a Zygote `rrule` / torch autograd function for (m::GNBlock)(x) needs.
 * Inputs: the forward's inputs (ef, nf, gf), its outputs (ef_out, nf_out, gf_out) and the upstream gradients with the
 * outputs' shapes (g_*; NULL = zero).  Outputs (all optional, NULL = not wanted): gradients w.r.t. the inputs (d_ef, d_nf,
 * d_gf, input shapes) and w.r.t. the parameters (grads->*.weight in the (out x in) column-major layout of the weights,
 * grads->*.bias), OVERWRITTEN.  Deterministic: segmented sums in CSC order, the nf[src] gradient is gathered through a
 * CSR view of the same graph (no atomics), weight gradients are two-stage fixed-order reductions.
 * Activations identity / relu / tanh / sigmoid differentiate from the stored outputs; gelu (not a function of its output) from the
 * pre-activation, recomputed per level. */
typedef struct gnx_dense_grad {
  float* weight; /* (out x in) column-major, device, or NULL */
  float* bias;   /* (out), device, or NULL                   */
} gnx_dense_grad;
typedef struct gnx_block_grads {
  gnx_dense_grad edgefn, nodefn, graphfn;
} gnx_block_grads;
GNX_API size_t gnx_block_backward_workspace_bytes(const gnx_graphs* h, const gnx_block_params* p, int64_t n_replicas);
GNX_API int32_t gnx_block_backward(const gnx_graphs* h, const gnx_block_params* p, const float* ef, const float* nf, const float* gf,
                           const float* ef_out, const float* nf_out, const float* gf_out, const float* g_ef_out,
                           const float* g_nf_out, const float* g_gf_out, int64_t n_replicas, float* d_ef, float* d_nf,
                           float* d_gf, const gnx_block_grads* grads, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the Chain block: takes the forward's INPUTS and the upstream gradients (NULL = zero); every layer's output is recomputed
 * into the workspace.  Gradients w.r.t. the inputs (optional) and, per chain, one gnx_dense_grad per layer (host arrays of n_layers
 * entries, or NULL; entries' pointers optional), all OVERWRITTEN.  The tail layers and the node / graph chains are row-wise Dense
 * pullbacks (matrix cores for real matrices), the edge chain's first layer goes through gnx_block_backward; fixed summation orders. */
typedef struct gnx_chain_block_grads {
  const gnx_dense_grad* edgefn;  /* [edgefn.n_layers]  */
  const gnx_dense_grad* nodefn;  /* [nodefn.n_layers]  */
  const gnx_dense_grad* graphfn; /* [graphfn.n_layers] */
} gnx_chain_block_grads;
GNX_API size_t gnx_chain_block_backward_workspace_bytes(const gnx_graphs* h, const gnx_chain_block_params* p, int64_t n_replicas);
GNX_API int32_t gnx_chain_block_backward(const gnx_graphs* h, const gnx_chain_block_params* p, const float* ef, const float* nf, const float* gf,
                                 const float* g_ef_out, const float* g_nf_out, const float* g_gf_out, int64_t n_replicas, float* d_ef,
                                 float* d_nf, float* d_gf, const gnx_chain_block_grads* grads, void* workspace, size_t workspace_bytes,
                                 void* stream);

/* Backward of (m::GNCore)(x) = x + block(gn1(x)) + ffwd(gn2(x)) (src/gncore.jl:56-68).  Takes the forward's INPUTS and the
 * upstream gradients; the intermediates (both LayerNorms, the block's outputs, the FeedForward hidden activations) are
 * recomputed into the workspace.  Gradient buffers are optional (NULL = not wanted) and overwritten.  FeedForward: fc1 with any
 * activation code (gelu differentiates from the recomputed pre-activation), fc2 with identity (the reference's
 * Chain(Dense(d,4d,relu), Dense(4d,d))). */
typedef struct gnx_layernorm_grad {
  float* gamma;
  float* beta;
} gnx_layernorm_grad;
typedef struct gnx_ffn_grad {
  gnx_dense_grad fc1, fc2;
} gnx_ffn_grad;
typedef struct gnx_core_grads {
  gnx_block_grads block;
  gnx_layernorm_grad ln1[3], ln2[3];
  gnx_ffn_grad ff[3];
} gnx_core_grads;
GNX_API size_t gnx_core_backward_workspace_bytes(const gnx_graphs* h, const gnx_core_params* p, int64_t n_replicas);
GNX_API int32_t gnx_core_backward(const gnx_graphs* h, const gnx_core_params* p, const float* ef, const float* nf, const float* gf,
                          const float* g_ef_out, const float* g_nf_out, const float* g_gf_out, int64_t n_replicas, float* d_ef,
                          float* d_nf, float* d_gf, const gnx_core_grads* grads, void* workspace, size_t workspace_bytes, void* stream);

/* ---- forward: replaces (m::GNCore)(x) (src/gncore.jl:56-68); GNCoreList = caller-side fold (gncorelist.jl:43-45) ----
 * Wide cores (block in the matrix cores' projected form, FeedForward widths 64 / 128): gn1 / gn2 of ef and nf are applied by the
 * kernels as they load x (one pass of row statistics; GNX_FLAG_NO_LN_FUSE materialises the LayerNorms: bit-identical for the node rows and
 * for the two-launch form of the edge rows (GNX_FLAG_CORE_EDGE_SPLIT); the one-launch form of the edge rows carries the LayerNorms' scale
 * in its weight planes and their shift in a constant vector — the same formula in another association, within a few fp32 roundings), and
 * the graph level of the core runs on a side stream (joined before gnx_core_forward returns; part of the capture when `stream` is being
 * captured; GNX_FLAG_NO_FORK: one stream).  The side streams are a small pool of the HANDLE that gnx_core_workspace_bytes creates — call
 * it outside a capture, as every workspace query — and a call holds one only while it enqueues its work.
 * THREADING: a handle is immutable after creation apart from tables built once behind a mutex, so concurrent forwards on ONE handle from
 * several host threads are allowed — each with its own stream, workspace and output buffers (tests/test_gpu_core.py::
 * test_two_host_threads_run_core_forwards_on_one_handle_concurrently: bit-identical to the serial run).  Calls at matrix-core widths (any
 * width above 32) TAKE TURNS ON THE DEVICE: every exported forward / backward holds a per-device lock while it enqueues, waits for the
 * previous such call's end when that ran on another stream, and records an event at its own end (one stream: a lock and an event record
 * per call; a stream under capture is left alone).  Reason (profiles/r05_mfma_mix_hazard.log): on the MI355X boxes this was built on, a
 * kernel on the fp32 matrix instruction returns wrong values now and then — one pass of one instruction: 2 rows x 32 columns — while a
 * dense bf16 matrix kernel runs on another stream of the device, the library's own six-term kernels or anybody else's (a bf16 GEMM);
 * nothing is shared between the two.  GNX_ALLOW_OVERLAP=1 (read once) removes the guard.  Matrix kernels of OTHER libraries that the
 * host runs on other streams beside these calls are the host's to keep apart (an event between the streams).
 * ARITHMETIC of the wide FeedForwards (edges / nodes at width 128 or 64, >= 4096 rows), of the projected edge update at 128 -> 128 or
 * 128 -> at most 32 outputs (gnx_block_forward too, >= 4096 edges) and of its node projections at 64-wide nodes (>= 4096 nodes):
 * fp32 in, fp32 out, fp32 accumulation; every fp32 product is evaluated on the bf16 matrix cores as six terms of an EXACT three-way split
 * of both operands (hi + mid + lo bf16 parts = the 24 mantissa bits; the dropped terms are <= 2^-23 |a||b|) — as accurate as the fp32
 * matrix instruction against float64 by test (tests/test_gpu_x6_stress.py: magnitudes over twelve decades, cancellation, a column scaled by
 * 1e20), 2x its speed; inputs that are not finite (or within 0.4 % of the largest finite float) produce NaN where the fp32 instruction may
 * produce an infinity, and operands below ~1e-33 in magnitude may keep only 16 of their 24 mantissa bits (their low parts are bf16
 * subnormals).  The CALL chooses: GNX_FLAG_FFN_FP32 (FeedForwards) / GNX_FLAG_EDGE_FP32 (edge update and projections) / GNX_FLAG_FP32_MFMA
 * (both) run the kernels on the fp32 matrix instruction instead (csrc/gnx_ffn_fused.hip, csrc/gnx_wide.hip); the environment variables
 * GNX_FFN_FP32 / GNX_EDGE_FP32 set the process-wide default (read once).
 * At 128-wide edges a core's edge update runs inside its edge FeedForward's launch, ef' kept in registers (same bits as two launches:
 * GNX_FLAG_CORE_EDGE_SPLIT), and the row statistics of the edge rows are computed in those kernels (same bits as the statistics pass:
 * GNX_FLAG_LN_STATS_PASS). */
GNX_API size_t gnx_core_workspace_bytes(const gnx_graphs* h, const gnx_core_params* p, int64_t n_replicas);
GNX_API int32_t gnx_core_forward(const gnx_graphs* h, const gnx_core_params* p, const float* ef, const float* nf,
                         const float* gf, int64_t n_replicas, float* ef_out, float* nf_out, float* gf_out,
                         void* workspace, size_t workspace_bytes, uint32_t flags, void* stream);

/* ---- GNCore in TRAINING mode: the Dropout(p) that ends each FeedForward chain (src/gnfeedforward.jl:27-31), applied by Flux inside a
 * gradient call and skipped in test mode (= gnx_core_forward).  y = x + block(gn1(x)) + m .* ffwd(gn2(x)) with m = u > p ? 1/(1-p) : 0,
 * u uniform on [0,1), independent per element (Flux._dropout_kernel).  (GNBlock's own `dropout` field is never applied by its forward:
 * src/gnblock.jl:63-69.)  The masks are never stored: element i of entity t (0 edges, 1 nodes, 2 graphs; i counts the packed
 * [R][rows][width] floats) is a pure function of (seed, t, i) — Philox-4x32-10 — so the backward regenerates the forward's masks from the
 * same gnx_dropout value, and gnx_dropout_mask writes them out for a host that wants to check either pass (tests/test_gpu_dropout.py).
 * (A dropped element therefore holds x + block + (f_fused - f_unfused) — two fp32-accurate evaluations of the same FeedForward, ~1e-6 of its
 * scale apart — not x + block exactly as in Flux; the backward treats its FeedForward gradient as exactly zero.)
 * The host draws a fresh `seed` per forward call (one per core of a GNCoreList).  dropout = NULL or p = 0: exactly gnx_core_forward /
 * gnx_core_backward.  The forward is gnx_core_forward (any flags) followed by the correction y += (m - 1) .* f, f recomputed unfused in the
 * workspace (csrc/gnx_dropout.hip). */
typedef struct gnx_dropout {
  float p;          /* drop probability, 0 <= p <= 1 */
  uint32_t reserved;
  uint64_t seed;
} gnx_dropout;
GNX_API int32_t gnx_dropout_mask(const gnx_dropout* dropout, int32_t entity, int64_t n_elements, float* out, void* stream);
GNX_API size_t gnx_core_train_workspace_bytes(const gnx_graphs* h, const gnx_core_params* p, int64_t n_replicas);
GNX_API int32_t gnx_core_forward_train(const gnx_graphs* h, const gnx_core_params* p, const gnx_dropout* dropout, const float* ef, const float* nf,
                               const float* gf, int64_t n_replicas, float* ef_out, float* nf_out, float* gf_out, void* workspace,
                               size_t workspace_bytes, uint32_t flags, void* stream);
/* workspace: gnx_core_backward_workspace_bytes */
GNX_API int32_t gnx_core_backward_train(const gnx_graphs* h, const gnx_core_params* p, const gnx_dropout* dropout, const float* ef, const float* nf,
                                const float* gf, const float* g_ef_out, const float* g_nf_out, const float* g_gf_out, int64_t n_replicas,
                                float* d_ef, float* d_nf, float* d_gf, const gnx_core_grads* grads, void* workspace, size_t workspace_bytes,
                                void* stream);

/* ---- row statistics of a packed [rows][d] tensor: stats[row] = (mean, 1 / (sigma + eps)) (eps_mode 0, Flux 0.14 `normalise`) or
 * (mean, 1 / sqrt(sigma^2 + eps)) (eps_mode 1), uncorrected sigma — the one pass over x from which the wide kernels apply GNGraphNorm's
 * LayerNorms (src/gngraphnorm.jl:19-26) as they load their rows; exported as the building block it is (and so that it can be exercised
 * on its own: tests/overlap_probe.py).  d must be a multiple of 64 up to 512, x 16-byte aligned, stats [rows][2] 8-byte aligned. */
GNX_API int32_t gnx_row_stats(const float* x, int64_t rows, int32_t d, float eps, int32_t eps_mode, float* stats, void* stream);

/* ---- materialised update-function inputs: the reference's exported building blocks getedgefninput /
 * getnodefninput / getgraphfninput (src/edgefninput.jl:1-47, src/nodefninput.jl:1-24, src/graphfninput.jl:1-13).
 * The forward never materialises them; these exist for callers that use the building blocks directly.
 *   kind 0 (edge):  out [R][E][de+2dn+dg] = [ef ; nf[src] ; nf[dst] ; gf[g]]
 *   kind 1 (node):  out [R][N][de+dn+dg]  = [sum_{e->n} ef ; nf ; gf[g]]        (ef = the UPDATED edge features)
 *   kind 2 (graph): out [R][G][de+dn+dg]  = [sum_e ef ; sum_n nf ; gf]
 * A width of 0 / a NULL pointer drops that segment exactly like the `nothing` methods of the reference. */
GNX_API int32_t gnx_fn_input(const gnx_graphs* h, int32_t kind, const float* ef, int32_t de, const float* nf, int32_t dn,
                     const float* gf, int32_t dg, int64_t n_replicas, float* out, void* stream);

/* ---- edge collapsing: unpaddedcollapsedef / flatunpaddedcollapsedef (src/gngraphbatch.jl:56-111, exported at
 * src/GraphNets.jl:50; reference tests test/runtests.jl:4-59).  For every real edge i->j with i >= j (the lower
 * triangle of the adjacency matrix, in edge order) the symmetric average (ef[i->j] + ef[j->i]) / 2; a self loop gives
 * ef[i->i].  If the reverse edge does not exist it contributes 0 (the reference reads the padded slot there, which is
 * 0 for batched inputs and junk for block outputs).
 * gnx_collapse_offsets: off[G+1] = per-graph offsets into the collapsed rows (off[G] = total).
 * gnx_collapse_edges:   out [R][total][d] from ef [R][E][d]. */
GNX_API int32_t gnx_collapse_offsets(const gnx_graphs* h, int64_t* off);
GNX_API int32_t gnx_collapse_edges(const gnx_graphs* h, const float* ef, int32_t d, int64_t n_replicas, float* out, void* stream);
/* collapsef itself (src/gngraphbatch.jl:83-85: batched_mul(ef, edge_collapser) / 2 with the collapser of :67-82), the PADDED array
 * form: out [B][L][d], B = graphs of the batch (or n_replicas of a shared graph), L = PN(PN+1)/2 coordinates (i, j), i >= j, of the
 * padded PN x PN grid in column-major order; out[b][l] = (P[i->j] + P[j->i]) / 2 over the zero-padded edge grid P (P[i->i] on the
 * diagonal).  = Julia (d, L, B).  The reference reads whatever its padded array holds in non-edge slots; here they are 0. */
GNX_API int32_t gnx_collapse_padded(const gnx_graphs* h, const float* ef, int32_t d, int64_t n_replicas, float* out, void* stream);

/* ---- readout loss on packed outputs: Flux.logitcrossentropy(yhat, y) over the columns of
 * flatunpaddednf / flatunpaddedef, as used by the reference's only end-to-end workload (examples/sort/sort.jl:69-81):
 *   loss = mean over columns c of  -sum_k y[k,c] * logsoftmax(yhat[:,c])[k]
 * logits / targets: device [cols][d] rows (= Julia (d, cols) column-major); loss_out: ONE device float.
 * workspace: gnx_xent_workspace_bytes(cols) bytes.  Deterministic two-stage reduction. */
GNX_API size_t gnx_xent_workspace_bytes(int64_t cols);
GNX_API int32_t gnx_logit_cross_entropy(const float* logits, const float* targets, int32_t d, int64_t cols, float* loss_out,
                                void* workspace, size_t workspace_bytes, void* stream);

/* pullback of gnx_logit_cross_entropy w.r.t. the logits: d_logits[c][k] = g * (sum_k' y[k',c] * softmax(yhat[:,c])[k] - y[k,c]) / cols,
 * g = *upstream (ONE device float, the gradient of the scalar loss) */
GNX_API int32_t gnx_logit_cross_entropy_backward(const float* logits, const float* targets, int32_t d, int64_t cols,
                                         const float* upstream, float* d_logits, void* stream);

/* ---- reference-layout bridges: padef/padnf and unpadef/unpadnf (src/pad.jl:12-64, src/unpad.jl:1-17) ----
 * kind 0 = edges: packed [R][E][d] <-> padded [B][PN^2][d];  kind 1 = nodes: packed [R][N][d] <-> padded [B][PN][d],
 * where B = R (one graph in the handle) or G (R must be 1).  Pads are written as zeros.  `packed` may be NULL for a batch
 * without edges (E = 0: padef gives an array of zeros, unpadef writes nothing). */
GNX_API int32_t gnx_pad_features(const gnx_graphs* h, int32_t kind, const float* packed, int32_t d, int64_t n_replicas,
                         float* padded, void* stream);
GNX_API int32_t gnx_unpad_features(const gnx_graphs* h, int32_t kind, const float* padded, int32_t d, int64_t n_replicas,
                           float* packed, void* stream);

/* ---- a list of layers as ONE hipGraph: replaces a host-side chain such as the reference's
 *      `decoder(core(encoder(x)))` (examples/sort/sort.jl:68-75; GNCoreList is a foldl, src/gncorelist.jl:43-45) ------------
 * At README-sized widths a layer is ~25 us of GPU work, less than the host spends launching it: a multi-layer model is
 * launch-bound.  gnx_model_create copies the layer descriptors (NOT the weights: the device pointers inside stay the
 * caller's and must stay valid), sizes every intermediate tensor and workspace once (library-owned device memory, freed by
 * gnx_model_destroy) and compiles any run-time specialised kernel.  gnx_model_forward runs the layers back to back on
 * `stream`; the first call with a given set of input/output pointers captures them into a hipGraph (on an internal stream),
 * later calls with the same pointers replay it with one hipGraphLaunch (new pointers: re-capture) — unless the captured forward is fewer
 * than five kernels (one narrow GNBlock: two), which is launched kernel by kernel: faster than a replay's launch latency (24.9 vs 29.9
 * us/step on BASELINE configs[1]; env GNX_MODEL_GRAPH_MIN_NODES).  GNX_FLAG_NO_GRAPH runs
 * eagerly.  Layer i's output widths must equal layer i+1's input widths (GNX_ERR_DIMS).  One forward at a time per model. */
typedef struct gnx_model gnx_model;
#define GNX_LAYER_BLOCK 0
#define GNX_LAYER_CORE 1
typedef struct gnx_layer {
  int32_t kind;       /* GNX_LAYER_BLOCK: params -> gnx_block_params; GNX_LAYER_CORE: params -> gnx_core_params */
  int32_t reserved;
  const void* params;
} gnx_layer;
#define GNX_FLAG_NO_GRAPH 0x8u
GNX_API int32_t gnx_model_create(const gnx_graphs* h, const gnx_layer* layers, int32_t n_layers, int64_t n_replicas, gnx_model** out);
GNX_API int32_t gnx_model_destroy(gnx_model* m);
/* widths of the model's output (DE', DN', DG'); 0 <=> nothing */
GNX_API int32_t gnx_model_out_dims(const gnx_model* m, int32_t dims[3]);
GNX_API int32_t gnx_model_forward(gnx_model* m, const float* ef, const float* nf, const float* gf, float* ef_out, float* nf_out,
                          float* gf_out, uint32_t flags, void* stream);
/* gnx_model_create prepares the parameters of every layer whose descriptor carries no `prepared` object (the model owns those objects).
 * After the weights' VALUES changed (an optimiser step; same pointers) call this before the next gnx_model_forward. */
GNX_API int32_t gnx_model_refresh_weights(gnx_model* m, void* stream);

/* ---- multi-GPU: whole graphs sharded over the devices of ONE host process, gf' all-gathered ------------
 * The reference has no multi-device code (nothing to cite in /root/reference/src); the contract is BASELINE.json's north_star:
 * "heterogeneous-graph batches shard by graph across the 8 GPUs of one node with RCCL all-gather of graph-level features".
 * Every term of a graph's edge, node and graph update depends on that graph only, so a rank needs nothing from another rank:
 * it builds a gnx_graphs handle of ITS graphs (gnx_graphs_create_*) and keeps its ef / nf / gf rows resident; the only
 * collective is one all-gather of gf'.  One process drives all devices (ncclCommInitAll, ncclGroupStart/End), which is how a
 * Julia session uses a multi-GPU node.  RCCL is loaded at run time (librccl.so.1).
 *
 * gnx_dist_partition (host only): equal graph counts per rank (+-1), balanced by edge count — graphs sorted by E_g descending
 * (ties: ascending id), dealt to ranks in snake order 0..R-1, R-1..0, ...; rank r owns the original graph ids
 * shard_graphs[shard_off[r] .. shard_off[r+1]), ascending.  shard_off has n_ranks+1 entries, shard_graphs n_graphs.
 * gnx_dist_create: communicator over device_ids[0..n) plus the gather plan of a partition (any permutation of the graphs:
 * rank r's local handle holds its graphs in the order they appear in shard_graphs) for gf' rows of `og` floats.
 * gnx_dist_allgather_gf: gf_local[r] = device-r rows [count_r][og] of rank r's graphs; afterwards gf_all[r] = device-r table
 * [n_graphs][og] in ORIGINAL graph order on every rank.  streams[r] (NULL array / entry = default stream): the stream of
 * device r that produced gf_local[r]; the collective runs on internal streams behind it and gf_all[r] is ready on it.
 * gnx_dist_block_forward: per rank gnx_block_forward(h[r], p[r], ...; n_replicas = 1) on device r (parameters replicated by
 * the caller: p[r] points at device-r copies), then the all-gather of gf'.  Arrays have one entry per rank. */
typedef struct gnx_dist gnx_dist;
GNX_API int32_t gnx_dist_partition(const int64_t* edge_counts, int64_t n_graphs, int32_t n_ranks, int64_t* shard_off, int64_t* shard_graphs);
GNX_API int32_t gnx_dist_create(const int32_t* device_ids, int32_t n_devices, const int64_t* shard_off, const int64_t* shard_graphs,
                        int64_t n_graphs, int32_t og, gnx_dist** out);
GNX_API int32_t gnx_dist_destroy(gnx_dist* d);
/* the gather plan of a partition, on the host (no device, no RCCL): validates shard_off / shard_graphs (a permutation of the graphs)
 * and writes src_row[g] = row of original graph g in the gathered [n_ranks * max_count][og] table (every rank contributes max_count rows,
 * zero padded).  gnx_dist_create builds its plan with this function.  src_row / max_count may be NULL. */
GNX_API int32_t gnx_dist_gather_plan(const int64_t* shard_off, const int64_t* shard_graphs, int32_t n_ranks, int64_t n_graphs, int32_t* src_row,
                             int64_t* max_count);
/* out[g][:] = gathered[src_row[g]][:] on the device (all three device pointers): the permutation back to ORIGINAL graph order that
 * follows the all-gather, for hosts that run the collective themselves (one process per GPU over torch.distributed / MPI) */
GNX_API int32_t gnx_dist_permute_rows(const float* gathered, const int32_t* src_row, int64_t n_graphs, int32_t og, float* out, void* stream);
GNX_API int32_t gnx_dist_allgather_gf(gnx_dist* d, const float* const* gf_local, float* const* gf_all, void* const* streams);
GNX_API int32_t gnx_dist_block_forward(gnx_dist* d, const gnx_graphs* const* h, const gnx_block_params* const* p, const float* const* ef,
                               const float* const* nf, const float* const* gf, float* const* ef_out, float* const* nf_out,
                               float* const* gf_out_local, float* const* gf_all, void* const* workspace, const size_t* workspace_bytes,
                               uint32_t flags, void* const* streams);

/* The replay form of the sharded forward, for loops over many batches of the same graphs (training / serving): n_steps block forwards
 * per rank — independent batches: step s of rank r reads ef / nf / gf [s * n_ranks + r] and writes ef_out / nf_out [s * n_ranks + r] with the
 * workspace [s * n_ranks + r] of workspace_bytes[r] bytes — whose gf' rows go straight into the communicator's stacked send buffer; ONE grouped
 * all-gather moves the n_steps tables of every rank and gf_all[r] (device r) receives [n_steps][n_graphs][og] in ORIGINAL graph order.
 * h / p / workspace_bytes / gf_all / streams have one entry per rank.  The launch sequence of a rank is captured into ONE hipGraph per
 * device the first time a set of arguments is seen (that call runs eagerly and captures; up to 32 argument sets are kept per communicator)
 * and replayed with one hipGraphLaunch per device afterwards: the host issues n_ranks graph launches + one grouped collective + n_ranks
 * permute kernels per call, whatever n_steps is.  GNX_FLAG_NO_GRAPH: always eager.  One call at a time per communicator.
 * An argument set is identified by what the captured launches contain: every pointer, the handles' identities (a serial number, not
 * the address) and the CONTENTS of *p[r] — a descriptor rewritten in place, or a handle destroyed and recreated at the same address, is
 * a new set.  GNX_FLAG_DIST_NO_GATHER: the forwards only (no collective, gf_all untouched) — what the all-gather costs is the difference. */
#define GNX_FLAG_DIST_NO_GATHER 0x10u
GNX_API int32_t gnx_dist_block_forward_steps(gnx_dist* d, int32_t n_steps, const gnx_graphs* const* h, const gnx_block_params* const* p,
                                     const float* const* ef, const float* const* nf, const float* const* gf, float* const* ef_out,
                                     float* const* nf_out, float* const* gf_all, void* const* workspace, const size_t* workspace_bytes,
                                     uint32_t flags, void* const* streams);

/* ---- run-time specialisation (the analogue of Julia compiling a GNBlock for its own widths on first use) -----------
 * The fused one-launch kernel is compiled ahead of time for the README / benchmark width sets; for any other width set
 * with every width <= 32 (and at most 1024 weights in the edge and node functions) it is compiled at run time with hiprtc (gfx950), once per process and device, the first time
 * gnx_block_workspace_bytes / gnx_block_forward sees the width set (never while the stream is being captured: such a
 * call runs the generic kernels).  GNX_JIT=0 disables it; GNX_JIT_CACHE=<dir> keeps the code objects on disk.
 * If hiprtc is unavailable the generic HIP kernels run instead.
 * gnx_jit_precompile: compile only (no GPU needed) — build-time / CI check; *code_bytes = size of the code object.
 * gnx_jit_stats: out = {compiled, disk-cache hits, failures, first uses inside a capture}. */
GNX_API int32_t gnx_jit_precompile(const gnx_block_params* p, int32_t wtile_e_cap, size_t* code_bytes);
/* the same check for the one-launch FeedForward + residual kernel of a narrow GNCore (widths 1..16), which is specialised at run time for
 * width triples other than README ex.3's (10,5,3) */
GNX_API int32_t gnx_jit_precompile_core_post(int32_t de, int32_t dn, int32_t dg, size_t* code_bytes);
GNX_API int32_t gnx_jit_stats(int64_t out[4]);

/* ---- per-kernel timing from dispatch timestamps (bench / roofline evidence): while enabled, every kernel the library launches carries a
 * start / stop event pair on its own dispatch packet (hipExtLaunchKernel) — the kernel's begin -> end as rocprofv3's kernel trace reports
 * it, plus a constant ~3.9 us that gnx_profile_calibrate measures on an empty kernel ---- */
GNX_API int32_t gnx_profile_enable(int32_t on);
GNX_API int32_t gnx_profile_reset(void);
/* n launches of an empty kernel timed the same way (entry "__empty_bracket__"): the constant of the method */
GNX_API int32_t gnx_profile_calibrate(int32_t n, void* stream);
/* synchronises the recorded events; writes up to `max` entries, returns how many exist in *n */
GNX_API int32_t gnx_profile_read(gnx_profile_entry* out, int32_t max, int32_t* n);

#ifdef __cplusplus
}
#endif
#endif /* GNX_H */

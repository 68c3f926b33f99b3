/*
 * gn2v.h -- C ABI of libgn2v.so, the MI355X (gfx950) Node2Vec / SkipGram / CBOW engine.
 *
 * This is the drop-in boundary for the one hot call of the reference,
 *     node_embeddings = self._model.fit_transform(graph)
 *         (embiggen/embedders/ensmallen_embedders/node2vec.py:99, model constructed at :65-69)
 * and for the batch generator
 *     graph.node2vec(batch_size, walk_length, window_size, iterations, ...)
 *         (embiggen/sequences/tensorflow_sequences/node2vec_sequence.py:190-201).
 * In the reference both are PyO3 calls into the `ensmallen` wheel; here they are plain C entry
 * points a ctypes / cffi / cgo stub can bind.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; gn2v_last_error() returns the
 *     message of the last failure on the calling thread.
 *   - no C++ types, exceptions or torch types cross the boundary: pointers + sizes only.
 *   - pointers named d_* are DEVICE pointers (HBM of the graph's device); the caller owns them.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); launches are
 *     asynchronous on it unless documented otherwise.
 *   - a handle may be shared by threads (its bookkeeping is locked); launches from different
 *     threads on the same stream are ordered by the stream as usual.
 */
#ifndef GN2V_H
#define GN2V_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GN2V_VERSION 320 /* 0.3.2: gn2v_graph_walk_accel; resident cells for rows of up to 512 floats */

#define GN2V_SENTINEL 0xFFFFFFFFu /* walk positions after a trap node */

/* gn2v_graph_create flags */
#define GN2V_GRAPH_DEVICE_PTRS 1u /* arrays are device pointers, borrowed (not copied) */
#define GN2V_GRAPH_SYMMETRIC 2u   /* undirected: every edge is stored in both directions (lets the
                                     sampler test adjacency from the shorter of two rows)       */

/* gn2v_train_params.flags */
#define GN2V_TRAIN_SCALE_FREE 1u     /* use_scale_free_distribution (node2vec_skipgram.py:101) */
#define GN2V_TRAIN_DOWNSAMPLE 2u     /* stochastic_downsample_by_degree (:97-98)                */
#define GN2V_TRAIN_NORM_LR 4u        /* normalize_learning_rate_by_degree (:99-100)             */
#define GN2V_TRAIN_DETERMINISTIC 8u  /* one wavefront, strict walk order: oracle-exact, slow    */
/* How row updates reach memory.  With none of the three bits set the engine picks: the
 * walk-ordered kernels use atomics on small graphs (where thousands of concurrent wavefronts
 * collide on the same rows all the time: measured CBOW link AUROC 0.80 vs 0.99 at 1 k nodes) and
 * Hogwild write-through stores on larger ones (the CPU reference is racy by design as well; 2x
 * the speed of atomics, DESIGN.md "Update modes") -- SkipGram from 2^16 nodes, CBOW once a table
 * holds GN2V_CBOW_STORES_MIN_ELEMENTS floats (nodes x row stride; BA graphs, link AUROC of stores
 * minus atomics': 8 k nodes -0.015 / -0.006 / -0.002 at d = 32 / 64 / 128, 32 k -0.004 / -0.001 /
 * -0.0001, 64 k -0.0025 / -0.0004 / 0: the limit follows the -0.0004 line, 32 k nodes at d = 128,
 * 64 k at d = 64; graphs without hubs lose nothing from 8 k nodes on); the block path uses plain
 * stores on contextual rows that are exclusive to one XCD and atomics per run of equal centre on
 * the central rows at every size. */
#define GN2V_CBOW_STORES_MIN_ELEMENTS (1u << 22)
#define GN2V_TRAIN_ATOMIC 16u        /* hardware f32 atomics on every element: no lost update   */
#define GN2V_TRAIN_WRITE_BACK 32u    /* read-modify-write, plain L2 write-back stores            */
#define GN2V_TRAIN_WRITE_THROUGH 64u /* read-modify-write, 16 B write-through (sc1) stores       */
/* Walk-ordered SkipGram in the store modes keeps the window's contextual rows in LDS (one HBM
 * read + one write-back per walk position instead of one per pair); rows of high-degree nodes are
 * left in HBM.  Switches: */
#define GN2V_TRAIN_NO_CTX_CACHE 128u  /* always use the plain kernel                             */
#define GN2V_TRAIN_CTX_CACHE_ALL 256u /* cache every row regardless of degree (tests)            */
#define GN2V_TRAIN_CTX_CACHE_NONE 512u /* treat every row as a high-degree row: it stays in HBM
                                          (CBOW: its window slot holds the pending step; tests)  */
/* gn2v_train picks its schedule: SkipGram on graphs of >= GN2V_BLOCK_PATH_MIN_NODES nodes in the
 * default update mode runs the block path (gn2v_train_blocks; one part of 8 XCD slices up to 2^18
 * nodes), everything else the walk-ordered kernels.  The limit is where the link quality of the
 * block path meets that of atomics on every row (BA graphs, 30 epochs: cosine AUROC 0.9959 vs
 * 0.9966 at 2 708 nodes, 0.989 vs 0.994 at 2 048, 0.88 vs 0.96 at 512) -- at 22 x the speed:
 * thousands of wavefronts adding to the same few rows serialise in the atomic units.  Overrides: */
#define GN2V_BLOCK_PATH_MIN_NODES 2560u
#define GN2V_TRAIN_WALK_ORDERED 1024u /* never the block path                                    */
#define GN2V_TRAIN_BLOCK_PATH 2048u   /* always the block path (SkipGram; with
                                         GN2V_TRAIN_DETERMINISTIC: its sequential schedule)      */

/* Block path: the gradient of a run of equal centre is added to the central row with hardware
 * f32 atomics (lane-contiguous; the row is shared by the XCDs, which reach it through their own
 * cells).  This bit: a run that is its centre's only one in the cell writes row + gradient with
 * one write-through store instead -- 1-2 % faster, but an update is lost whenever another XCD
 * holds the same centre at that moment (counted: tests/test_gpu_default_vs_oracle.py). */
#define GN2V_TRAIN_CENTRAL_STORE 4096u

#define GN2V_MODEL_SKIPGRAM 0u
#define GN2V_MODEL_CBOW 1u

/* Walk parameters; names follow Node2VecSkipGramEnsmallen.__init__ (node2vec_skipgram.py:9-36). */
typedef struct {
    uint32_t walk_length;    /* nodes per walk (walk_length-1 steps)              */
    uint32_t iterations;     /* walks per source node per epoch                   */
    float return_weight;     /* 1/p                                                */
    float explore_weight;    /* 1/q                                                */
    uint32_t max_neighbours; /* node2vec_skipgram.py:22,78-81.  0 (the classes' None): exact
                                walks.  Else a step out of a node of higher degree chooses among
                                a per-visit sub-sample of this many of its edges (one per bucket
                                of the row; csrc/walk_kernels.h RowView)          */
    uint32_t flags;          /* reserved                                           */
    /* node2vec_skipgram.py:72-77, node2vec_sequence.py:57-66.  0 = unset = 1.0.  "Only applies
     * to colored graphs / multigraphs, otherwise it has no impact": they act only when the
     * matching type array was attached with gn2v_graph_set_types. */
    float change_node_type_weight; /* x weight of an edge to a node of another type than the current */
    float change_edge_type_weight; /* x weight of an edge of another type than the previous edge   */
} gn2v_walk_params;

typedef struct {
    uint32_t model;   /* GN2V_MODEL_*                                            */
    uint32_t d;       /* embedding_size                                          */
    uint32_t ld;      /* row stride of both tables in floats, multiple of 4, >= d */
    uint32_t epochs;
    uint32_t k;       /* number_of_negative_samples                              */
    uint32_t window;  /* window_size                                             */
    float lr;         /* learning_rate                                           */
    float lr_decay;   /* learning_rate_decay (per epoch)                         */
    float clip;       /* clipping_value                                          */
    uint32_t flags;   /* GN2V_TRAIN_*                                            */
    float init_scale; /* tables start uniform(-init_scale, init_scale)           */
    uint32_t min_dist; /* contexts at walk distance [min_dist, window]; 0 = 1 (Walklets scale s:
                          window = min_dist = s, embedders/ensmallen_embedders/walklets.py)    */
} gn2v_train_params;

/* Filled by gn2v_train / gn2v_stats_read: the handle's counters since the last gn2v_stats_reset
 * (they accumulate over calls).  Times are HIP-event milliseconds measured on the
 * stream the kernels were launched on. */
typedef struct {
    uint64_t pairs;        /* (centre, context) training pairs processed        */
    uint64_t walk_steps;   /* sampled walk transitions                          */
    uint64_t centres;      /* centres processed (CBOW unit)                     */
    double train_ms;       /* sum of training-kernel durations                  */
    double walk_ms;        /* sum of walk-kernel durations                      */
    uint32_t train_launches;
    uint32_t walk_launches;
    uint32_t block_parts;  /* plan of the block path when gn2v_train took it, else 0    */
    uint32_t block_slices;
    uint32_t block_stripes; /* centre stripes of that fit (gn2v_train_blocks), else 0   */
    uint32_t block_group_parts; /* parts whose pairs were extracted + sorted at a time, else 0 */
    uint64_t block_round_walks; /* walks per round of that fit, else 0                   */
    uint32_t resident_launches; /* training launches that ran sgns_resident_kernel (cells in LDS) */
    uint32_t resident_record;   /* pairs a wave of that kernel took per record (last launch), else 0 */
} gn2v_stats;

typedef struct gn2v_graph gn2v_graph;

int gn2v_version(void);
const char *gn2v_last_error(void);
/* number of visible HIP devices; 0 when there is none (never fails) */
int gn2v_device_count(void);

/* CSR graph: row_ptr u64[n_nodes+1], col_idx u32[n_edges] (neighbours ascending per row, no
 * duplicates), cumw f32[n_edges] = per-row inclusive prefix sums of positive edge weights or NULL,
 * sources u32[n_sources] = nodes with out-degree > 0 or NULL when every node is a source.
 * CSR convention as exported at embedders/pecanpy_embedders/node2vec.py:139-163.
 * Host arrays are uploaded (and may be freed on return); with GN2V_GRAPH_DEVICE_PTRS the device
 * arrays are borrowed and must outlive the handle. */
int gn2v_graph_create(const uint64_t *row_ptr, const uint32_t *col_idx, const float *cumw,
                      const uint32_t *sources, uint64_t n_nodes, uint64_t n_edges,
                      uint64_t n_sources, uint32_t flags, int device, gn2v_graph **out);
int gn2v_graph_destroy(gn2v_graph *g);

/* Attach node types u32[n_nodes] and / or edge types u32[n_edges] (aligned with col_idx; a row of
 * a multigraph lists the same neighbour once per edge type).  Equal ids = same type; multi-label
 * nodes get one id per distinct label set; unknown = 0xFFFFFFFF.  NULL leaves that kind untyped
 * (detaches it).  Pointer kind (host: copied / device: borrowed) follows the flags given to
 * gn2v_graph_create.  The ensmallen.Graph argument carries these (has_node_types /
 * has_edge_types, abstract_model.py:329-350); used by change_*_type_weight only. */
int gn2v_graph_set_types(gn2v_graph *g, const uint32_t *node_types, const uint32_t *edge_types);

/* Seeded Barabasi-Albert edge list on the device: (n_nodes-1)*m edges (src > dst). */
int gn2v_ba_edges(uint64_t n_nodes, uint32_t m, uint64_t seed, uint32_t *d_src, uint32_t *d_dst,
                  void *stream);

/* Walks [first_walk, first_walk+n_walks) of (seed, epoch) -> d_out u32[n_walks][walk_length].
 * walk_id = iteration * n_sources + source_index.  Replaces Graph.complete_walks / the walk half
 * of Graph.node2vec (node2vec_sequence.py:190-201). */
int gn2v_walks(gn2v_graph *g, const gn2v_walk_params *wp, uint64_t seed, uint64_t epoch,
               uint64_t first_walk, uint64_t n_walks, uint32_t *d_out, void *stream);

/* The same with ids in groups: walk b of the launch has id first_walk + (b / group) * stride +
 * b % group -- the walks of a whole Node2VecSequence batch, `group` = batch_size source nodes x
 * iterations (stride between the iterations' ids), in ONE launch (node2vec_sequence.py:190-201:
 * one Graph.node2vec call per batch).  Same walks as the calls of gn2v_walks it stands for. */
int gn2v_walks_strided(gn2v_graph *g, const gn2v_walk_params *wp, uint64_t seed, uint64_t epoch,
                       uint64_t first_walk, uint64_t n_walks, uint32_t group, uint64_t stride,
                       uint32_t *d_out, void *stream);

/* Node2VecSequence batch (node2vec_sequence.py:115-128): every walk position with a full window
 * -> d_words i32[n], d_contexts i32[n][2w], n = n_walks * (walk_length - 2w). */
int gn2v_window_batch(const uint32_t *d_walks, uint64_t n_walks, uint32_t walk_length,
                      uint32_t window, int32_t *d_contexts, int32_t *d_words, void *stream);

/* Every (centre, context) pair of the walks, window trimmed at the borders, contexts at distance
 * [min_dist, window]: d_pairs u32[n_walks][walk_length][2*window][2], unused slots hold
 * (GN2V_SENTINEL, GN2V_SENTINEL).  The flat pair list of a batch of walks (tests, tools); the
 * block-partitioned trainer extracts its pairs with gn2v_block_count / gn2v_block_extract. */
int gn2v_walk_pairs(const uint32_t *d_walks, uint64_t n_walks, uint32_t walk_length,
                    uint32_t window, uint32_t min_dist, uint32_t *d_pairs, void *stream);
/* table[r][c] = uniform(-scale, scale) from a hash of (seed, table_id, r, c); padding = 0 */
int gn2v_init_table(float *d_table, uint64_t n_rows, uint32_t d, uint32_t ld, uint64_t seed,
                    uint32_t table_id, float scale, void *stream);

/* One batch of explicit walks through the fused gather->dot->sigmoid->scatter-add kernel
 * (tables updated in place).  Walk b has id first_walk + b in (seed, epoch); d_neg_override
 * (optional) supplies explicit negatives u32[n_walks][walk_length][2w][k] (SkipGram) or
 * [n_walks][walk_length][k] (CBOW) instead of sampled ones.  tp->model is ignored. */
int gn2v_sgns_step(gn2v_graph *g, const gn2v_train_params *tp, const uint32_t *d_walks,
                   uint64_t n_walks, uint32_t walk_length, uint64_t seed, uint64_t epoch,
                   uint64_t first_walk, float lr, float *d_central, float *d_contextual,
                   const uint32_t *d_neg_override, void *stream);
int gn2v_cbow_step(gn2v_graph *g, const gn2v_train_params *tp, const uint32_t *d_walks,
                   uint64_t n_walks, uint32_t walk_length, uint64_t seed, uint64_t epoch,
                   uint64_t first_walk, float lr, float *d_central, float *d_contextual,
                   const uint32_t *d_neg_override, void *stream);

/* The whole of `models.SkipGram/CBOW(...).fit_transform(graph)` (node2vec.py:99): initialise both
 * caller-allocated tables f32[n_nodes][ld], then per epoch generate every walk and train on it.
 * max_walks_per_epoch = 0 trains on all iterations*n_sources walks; otherwise only that many
 * (benchmark budget).  Synchronises `stream` before returning and fills *stats. */
int gn2v_train(gn2v_graph *g, const gn2v_walk_params *wp, const gn2v_train_params *tp,
               uint64_t seed, uint64_t max_walks_per_epoch, float *d_central,
               float *d_contextual, gn2v_stats *stats, void *stream);

/* ---- Block-partitioned training: what gn2v_train runs for SkipGram on
 * graphs of >= GN2V_BLOCK_PATH_MIN_NODES nodes.  The entry points below are the boundary; the
 * steps they are made of (plan, alias tables, placement, extraction, one training step, one
 * round) are declared in gn2v_internal.h for the multi-process trainer and the parity tests. */

/* The whole fit (same contract as gn2v_train: caller-allocated tables f32[n_nodes][ld], filled on
 * return) through the block path on one GPU: automatic plan, alias tables, rounds of
 * stripes x round_walks walks, per round walk generation and, for each of the `stripes` centre
 * stripes in turn (stripe j: the centres c with c % stripes == j -- what `stripes` ranks do side
 * by side) and each group of parts: extraction + sort of the group's pairs from all the round's
 * walks and one gn2v_block_step per part on the stripe's rows of the central table
 * (gn2v_block_io.central_ld).  round_walks = the walks one pass extracts from (0 = automatic:
 * gn2v_block_round_plan of the free HBM); stripes 0 = 1 (none).  With stripes the pairs of a
 * centre meet in runs `stripes` times as long (faster: DESIGN.md 7.4), but the stripes of a round
 * are trained one after the other, not side by side as ranks would be, which costs link quality
 * when a fit has few rounds: an option, not the default.  Both tables are trained in the caller's
 * buffers (the contextual one stored part by part during the fit and put back in node order at
 * the end through one scratch copy; GN2V_BLOCK_LAYOUT=natural in the environment, or too little
 * free memory for that copy, trains it in node order with strided parts instead).  Returns 2
 * when device memory ran out before anything was trained.  gn2v_train calls this for SkipGram on
 * graphs of >= GN2V_BLOCK_PATH_MIN_NODES nodes and falls back to the walk-ordered schedule on
 * that 2. */
int gn2v_train_blocks(gn2v_graph *g, const gn2v_walk_params *wp, const gn2v_train_params *tp,
                      uint64_t seed, uint64_t max_walks_per_epoch, uint64_t round_walks,
                      uint32_t stripes, float *d_central, float *d_contextual, gn2v_stats *stats,
                      void *stream);

/* ---- Several GPUs from C.  The same fit spread over `world` ranks, one process (or thread) per
 * GPU, each with its own graph handle on its own device: the central table striped over the ranks
 * (centre c: rank c % world), the contextual table in parts = P x world parts that travel round
 * the ranks; per round every rank generates its share of the walks, the walks are all-gathered,
 * every rank extracts and trains the pairs whose centre it owns, part by part, while the part it
 * finished last goes to rank - 1 and the next one arrives from rank + 1 (the
 * schedule of embiggen_amd/distributed.py, which this entry point runs without Python).  What
 * moves between ranks goes through a communicator the HOST fills -- RCCL (ncclAllGather,
 * ncclSend / ncclRecv in a group, ncclBroadcast on a stream of its own), MPI, torch.distributed
 * behind callbacks, or plain copies for one rank.  All pointers are device pointers of the
 * calling rank; every function returns 0 on success; every rank makes the same calls in the
 * same order.  `stream`: the stream the library works on -- a transfer starts after what that
 * stream holds at the call, and what the library enqueues after the call (after sendrecv_wait
 * for an exchange) must see its result. */
typedef struct {
    void *ctx;   /* handed back to every function                                             */
    uint32_t rank, world;
    /* recv[r * bytes ...) = `send` of rank r, r = 0 .. world - 1                               */
    int (*all_gather)(void *ctx, const void *send, void *recv, uint64_t bytes, void *stream);
    /* start: `send` (send_bytes) to rank dst, `recv` (recv_bytes) from rank src; *handle names
     * the exchange for sendrecv_wait, which orders `stream` after both halves                  */
    int (*sendrecv_start)(void *ctx, const void *send, uint64_t send_bytes, uint32_t dst,
                          void *recv, uint64_t recv_bytes, uint32_t src, void *stream,
                          void **handle);
    int (*sendrecv_wait)(void *ctx, void *handle, void *stream);
    /* `buf` of rank root into `buf` of every rank                                              */
    int (*broadcast)(void *ctx, void *buf, uint64_t bytes, uint32_t root, void *stream);
} gn2v_comm;

/* The multi-GPU form of gn2v_train for SkipGram (node2vec.py:99 on `world` GPUs): every rank
 * calls it with its own handle, the same parameters and seed, and caller-allocated tables
 * f32[n_nodes][ld] that EVERY rank receives filled, in node order.  round_walks = walks per rank
 * and round (0 = from the free memory, agreed between the ranks).  world = 1 with a copying
 * communicator runs the same code on one GPU. */
int gn2v_train_world(gn2v_graph *g, const gn2v_walk_params *wp, const gn2v_train_params *tp,
                     uint64_t seed, uint64_t max_walks_per_epoch, uint64_t round_walks,
                     const gn2v_comm *comm, float *d_central, float *d_contextual,
                     gn2v_stats *stats, void *stream);

/* ---- GloVe: the third model of the reference's walk-based table (embedders/ensmallen_embedders/
 * node2vec.py:16-26 "Node2Vec GloVe" / "DeepWalk GloVe": models.GloVe; wrapper kwargs
 * node2vec_glove.py:8-30).  The fit = gn2v_walks -> gn2v_cooc_slots -> (sort + sum by key) ->
 * gn2v_glove_step over the non-zero entries for `epochs` epochs. */

/* Co-occurrence slots of the walks: d_keys / d_weights u64[n_walks][walk_length][2*window]; a used
 * slot holds centre << 32 | context and the fixed-point weight round(2^20 / distance), an unused
 * one (INT64_MAX, 0).  Contexts at walk distance [min_dist, window], window trimmed at the
 * borders.  Summing the weights of equal keys gives X_ij * 2^20 exactly, in any order. */
int gn2v_cooc_slots(const uint32_t *d_walks, uint64_t n_walks, uint32_t walk_length,
                    uint32_t window, uint32_t min_dist, uint64_t *d_keys, uint64_t *d_weights,
                    void *stream);

typedef struct {
    const uint32_t *d_rows; /* entry e couples central row d_rows[e] ...                     */
    const uint32_t *d_cols; /* ... with contextual row d_cols[e]                            */
    const float *d_logx;    /* log X_ij (X normalised by its largest entry)                  */
    const float *d_fx;      /* f(X_ij) = X_ij ^ alpha                                        */
    float *d_central;       /* f32[n_nodes][ld]                                              */
    float *d_contextual;    /* f32[n_nodes][ld]                                              */
    float *d_bias_central;  /* f32[n_nodes]                                                  */
    float *d_bias_contextual;
} gn2v_glove_io;

/* One pass of SGD over n_entries non-zero co-occurrence entries: g = f (u.v + b_i + b~_j - log X),
 * u -= lr g v, v -= lr g u, b_i -= lr g, b~_j -= lr g.  flags: GN2V_TRAIN_DETERMINISTIC (entry
 * order, one wavefront: equals the oracle) or one of the update-mode bits (default as for
 * gn2v_sgns_step).  `g` provides the device and the launch-time bookkeeping (gn2v_stats_read). */
int gn2v_glove_step(gn2v_graph *g, const gn2v_glove_io *io, uint64_t n_entries, uint32_t d,
                    uint32_t ld, float lr, uint32_t flags, void *stream);

/* Edge embeddings fused with the row gather: out[e] = op(src_table[src_ids[e]], dst_table[dst_ids[e]]).
 * Device form of the operators of embiggen/embedding_transformers/edge_transformer.py:12-343;
 * method ids follow the reference's method table (:348-361):
 * 0 Hadamard, 1 Sum, 2 Average, 3 L1, 4 AbsoluteL1, 5 SquaredL2, 6 L2, 7 Concatenate, 8 Min,
 * 9 Max, 10 L2Distance, 11 CosineSimilarity.  out is f32[n_edges][out_ld] with d columns written
 * (2d for Concatenate, 1 for L2Distance / CosineSimilarity). */
int gn2v_edge_embedding(const float *d_src_table, const float *d_dst_table, uint32_t d, uint32_t ld,
                        const uint32_t *d_src_ids, const uint32_t *d_dst_ids, uint64_t n_edges,
                        uint32_t method, float *d_out, uint32_t out_ld, void *stream);

/* A block fit (gn2v_train on graphs of >= GN2V_BLOCK_PATH_MIN_NODES nodes, gn2v_train_blocks)
 * hands its round buffers -- walks, pair words, sort storage: tens of GB on large graphs -- back
 * to the graph handle, which keeps up to a third of the device's memory for the handle's next
 * fit (a second fit otherwise waits 1.5-2 s for the driver to clear the same 65 GB again).  This
 * frees them now; gn2v_graph_destroy does too.  The Python classes call it after every fit
 * unless told to keep the buffers. */
int gn2v_graph_release_buffers(gn2v_graph *g);

/* Which accelerators of the walk sampler the handle holds right now (they are built on the first
 * walk that can use them and when memory allows; the walks are the same with or without them):
 * bit 0 the hashed edge set, bit 1 the filter in front of it, bit 2 the edge records, bit 3 their
 * typed form (csrc/walk_kernels.h).  Negative on a NULL handle. */
#define GN2V_WALK_ACCEL_EDGE_SET 1
#define GN2V_WALK_ACCEL_FILTER 2
#define GN2V_WALK_ACCEL_RECORDS 4
#define GN2V_WALK_ACCEL_TYPED_RECORDS 8 /* the 32 B form read by walks with type factors */
int gn2v_graph_walk_accel(gn2v_graph *g);

/* counters accumulated on the handle by the step / walk entry points since the last reset */
int gn2v_stats_reset(gn2v_graph *g, void *stream);
int gn2v_stats_read(gn2v_graph *g, gn2v_stats *stats, void *stream); /* synchronises */

#ifdef __cplusplus
}
#endif
#endif /* GN2V_H */

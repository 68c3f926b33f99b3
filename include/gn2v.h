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

/* ---- Block-partitioned training: how the one call of the reference (node2vec.py:99) is spread
 * over the GPUs of a node, one process per GPU.  The reference has no
 * counterpart (ensmallen trains inside one process).  Nodes are striped over `world` ranks
 * (centre c: rank c % world, row c / world of that rank's central partition) and over `parts`
 * context parts (context x: part x % parts, row x / parts); the parts travel round the ranks, a
 * pair (c, x) is trained on the owner of c while part x % parts is resident there, with
 * negatives drawn inside the cell of x.  `slices` stripes the rows of a part once more
 * (slice = row % slices): with one slice per XCD (8 on an MI355X; any multiple of the XCD count)
 * every XCD owns the rows it updates and they are updated with plain write-back stores; with any
 * other slice count several XCDs share a slice and the updates are write-through stores.
 * cell = part * slices + slice.  No row is ever held by two ranks.
 *
 * A (centre, context) pair is ONE 64-bit word,
 *     cell << (row_bits + ctx_bits) | centre row << ctx_bits | hot << (ctx_bits - 1) | context row
 * with the context row counted inside its cell ((x / parts) / slices): 8 bytes per pair whatever
 * the size of the graph. */
typedef struct {
    uint32_t world;       /* ranks = central partitions                                      */
    uint32_t rank;
    uint32_t parts;       /* context parts (a multiple of world when they travel between ranks) */
    uint32_t slices;      /* 1 .. 16 (XCD cells), up to GN2V_BLOCK_MAX_SLICES with resident cells */
    uint32_t walk_length;
    uint32_t window;      /* window_size                                                      */
    uint32_t min_dist;    /* 0 = 1 (Walklets: = window)                                       */
    uint32_t record;      /* consecutive sorted pairs a wavefront takes at a time; 0 = 32     */
    uint32_t row_bits;    /* out (gn2v_block_plan_check): bits of the centre row in a pair word */
    uint32_t flags;       /* GN2V_TRAIN_DOWNSAMPLE: centres thinned while pairs are extracted */
    /* Hot rows: the `hot_rows` rows of every cell with the highest in-degrees (<=
     * GN2V_BLOCK_HOT_MAX; 0 = none) are flagged by gn2v_block_alias.  They are the targets of the
     * degree-proportional negatives (node2vec_skipgram.py:101-102): so many wavefronts
     * read-modify-write them at once that plain stores keep a fraction of a percent of their
     * updates.  gn2v_block_step accumulates the updates of a flagged row in the workgroup's LDS
     * (exact) and hands the sums to the row with f32 atomics, on average every `hot_flush`
     * updates of the row and workgroup (a power of two; 0 = 16): no update is lost, at the speed
     * of the stores.  A launch whose LDS holds fewer rows takes the hottest. */
    uint32_t hot_rows;
    uint32_t hot_flush;
    uint32_t key_bits;    /* out (gn2v_block_plan_check): bits of a pair word in use (<= 64)   */
    uint32_t ctx_bits;    /* out: low bits of a pair word = context row inside its cell + hot flag */
} gn2v_block_plan;

/* validates the plan against the graph, fills row_bits, ctx_bits, key_bits and the defaults */
int gn2v_block_plan_check(gn2v_graph *g, gn2v_block_plan *plan);

/* rows first_row, first_row + row_stride, ... of the table gn2v_init_table would produce: a
 * rank initialises its partitions without ever holding the whole table */
int gn2v_init_table_rows(float *d_table, uint64_t n_rows, uint32_t d, uint32_t ld, uint64_t seed,
                         uint32_t table_id, float scale, uint64_t first_row, uint64_t row_stride,
                         void *stream);

/* Degree-proportional negatives inside a cell (use_scale_free_distribution,
 * node2vec_skipgram.py:101-102): one Walker alias table per cell, d_alias u64[n_nodes] = threshold
 * (2^32 scale) | alias row << 32, cells in order, rows of a cell in order; d_cell_rows
 * u64[cells + 1] = first entry of every cell.  Weights are the in-degrees (a uniform random
 * edge's endpoint); integer arithmetic throughout.
 * Hot rows (plan->hot_rows per cell: highest in-degree first, ties by row; in-degree >= 1) are
 * flagged: bit 0 of an entry (the row itself; the threshold keeps 31 bits), bit 63 (its alias
 * row), d_hub_bits u32[(n_nodes + 31) / 32] = one bit per node id (gn2v_block_extract copies it
 * into the pair words), d_hot_list u32[cells][GN2V_BLOCK_HOT_MAX] = row inside its cell of hot
 * slot s (GN2V_SENTINEL beyond the cell's count) and d_hot_slot u8[n_nodes] = slot of entry
 * d_cell_rows[cell] + row (0xFF: not hot). */
#define GN2V_BLOCK_HOT_MAX 192u
#define GN2V_BLOCK_HOT_DEFAULT 192u /* what gn2v_train_blocks and the Python trainer flag */
int gn2v_block_alias_temp_bytes(uint64_t n_nodes, uint64_t *bytes);
/* d_inv (or NULL): the round's placement (gn2v_block_placement) -- row i of cell (part, slice) is
 * then node d_inv[(slice + slices * i) * parts + part].  A plan without hot rows may pass NULL
 * for d_hub_bits, d_hot_list and d_hot_slot. */
int gn2v_block_alias(gn2v_graph *g, const gn2v_block_plan *plan, uint64_t *d_alias,
                     uint64_t *d_cell_rows, uint32_t *d_hub_bits, uint32_t *d_hot_list,
                     uint8_t *d_hot_slot, const uint32_t *d_inv, void *d_temp,
                     uint64_t temp_bytes, void *stream);

/* Placement of a round.  The negatives of a pair are drawn -- proportionally to the in-degree,
 * node2vec_skipgram.py:101-102 -- among the nodes of its context's CELL; with resident cells
 * (~200 rows) a fixed cell would confine a context's negatives to the same 200 nodes for a whole
 * fit.  So WHERE a node's contextual row is trained changes every round: a seeded permutation of
 * the node ids (seed, round_id; splitmix64-keyed, sorted with ties in node order: the oracle's
 * o_block_placement is bit-equal) inside their residue classes modulo `classes` -- `parts` when
 * the parts travel between ranks (a row never leaves its part: x' % parts == x % parts), 1 on one
 * GPU (the whole graph is shuffled).  d_place u32[n_nodes]: node -> placed id; d_inv: placed id
 * -> node.  Everything downstream works on placed ids (part = x' % parts, row = x' / parts,
 * slice = row % slices: the pair words, the alias tables); the tables stay where they are and
 * the resident kernel reaches a row through d_inv (gn2v_block_io).  Plans of XCD cells (<= 16
 * slices) keep their fixed cells of >= 32 k rows.
 * gn2v_block_place_walks: d_out[i] = d_place[d_walks[i]] (GN2V_SENTINEL kept): the walks as the
 * extraction reads them for the context side (gn2v_block_count / _extract: d_placed_walks). */
int gn2v_block_placement_temp_bytes(uint64_t n_nodes, uint64_t *bytes);
int gn2v_block_placement(gn2v_graph *g, uint32_t classes, uint64_t seed, uint64_t round_id,
                         uint32_t *d_place, uint32_t *d_inv, void *d_temp, uint64_t temp_bytes,
                         void *stream);
int gn2v_block_place_walks(const uint32_t *d_place, const uint32_t *d_walks, uint64_t n_entries,
                           uint32_t *d_out, void *stream);

/* Pairs of a round.  d_walks holds the walks of ALL ranks for the round (ids first_walk,
 * first_walk + 1, ...; all-gathered); this rank keeps the pairs whose centre it owns and whose
 * context lies in the group of parts part_lo, part_lo + 1, ... (part_n of them, cyclic modulo
 * `parts`; part_lo = part_n = 0: every part).  A round is extracted, sorted and trained a group at
 * a time, so the pair buffers hold one group, not the round.
 * gn2v_block_count: pass 1, fills d_work u64[GN2V_BLOCK_WORK_WORDS] (private to the two calls)
 * and d_cell_offsets u64[cells + 1] = where each cell starts in the sorted pair words (cells
 * outside the group are empty); the last entry is the number of pairs (the caller reads it to
 * size the buffers).
 * gn2v_block_extract: pass 2 + one stable radix sort on the bits above ctx_bits: d_pairs
 * u64[n_pairs] = the pair words (hot flag set when d_hub_bits, optional, flags the context node)
 * grouped by cell and centre row, ties in walk / position / slot order (independent of the launch
 * geometry).  d_temp: gn2v_block_extract_temp_bytes(n_pairs) bytes (the unsorted words + the
 * sort's own storage). */
#define GN2V_BLOCK_WORK_WORDS 532480 /* 8192 extraction waves + 524288 cells */
/* d_placed_walks (or NULL): the same walks with placed node ids (gn2v_block_place_walks): the
 * cell and row of a CONTEXT follow from its placed id, the centres keep their node ids. */
int gn2v_block_count(gn2v_graph *g, const gn2v_block_plan *plan, const uint32_t *d_walks,
                     const uint32_t *d_placed_walks, uint64_t n_walks, uint64_t seed,
                     uint64_t epoch, uint64_t first_walk, uint32_t part_lo, uint32_t part_n,
                     uint64_t *d_work, uint64_t *d_cell_offsets, void *stream);
int gn2v_block_extract_temp_bytes(uint64_t n_pairs, uint64_t *bytes);
int gn2v_block_extract(gn2v_graph *g, const gn2v_block_plan *plan, const uint32_t *d_walks,
                       const uint32_t *d_placed_walks, uint64_t n_walks, uint64_t seed,
                       uint64_t epoch, uint64_t first_walk,
                       uint32_t part_lo, uint32_t part_n, const uint64_t *d_work,
                       const uint32_t *d_hub_bits, uint64_t n_pairs, uint64_t *d_pairs,
                       void *d_temp, uint64_t temp_bytes, void *stream);

/* The cell offsets of a group's SORTED pair words (d_cell_offsets[c] = first word whose cell is
 * >= c, c = 0 .. parts x slices): what gn2v_block_count wrote for a group whose cells it counted,
 * and the only source for a wide group (GN2V_BLOCK_MAX_WIDE_GROUP_CELLS).  Does nothing for a
 * group that was counted (part_n x slices within the LDS counters). */
int gn2v_block_cell_offsets(gn2v_graph *g, const gn2v_block_plan *plan, uint32_t part_n,
                            const uint64_t *d_pairs, uint64_t n_pairs, uint64_t *d_cell_offsets,
                            void *stream);

typedef struct {
    const uint64_t *d_pairs;         /* sorted pair words of the group (gn2v_block_extract)    */
    const uint64_t *d_cell_offsets;  /* [cells + 1] (gn2v_block_count)                         */
    const uint64_t *d_alias;         /* gn2v_block_alias; unused without GN2V_TRAIN_SCALE_FREE */
    const uint64_t *d_cell_rows;     /* gn2v_block_alias: needed for d_alias and for the hot rows */
    const uint32_t *d_hot_list;      /* gn2v_block_alias; both NULL: flagged rows are ordinary rows */
    const uint8_t *d_hot_slot;
    float *d_central;                /* this rank's central partition f32[rows][central_ld]    */
    float *d_context;                /* context part `part`, resident here, f32[rows][context_ld] */
    uint64_t block_id;               /* RNG stream of the negatives: unique per (round, rank)  */
    uint32_t part;
    uint64_t central_ld;             /* floats between consecutive rows of the central partition;
                                        0 = ld.  One GPU that trains the `world` centre stripes
                                        of a plan one after the other ("virtual ranks": the runs
                                        of equal centre grow `world`-fold at the same memory)
                                        keeps the whole table f32[n_nodes][ld] and passes
                                        d_central = table + rank * ld, central_ld = world * ld   */
    uint64_t context_ld;             /* the same for the context part; 0 = ld.  A part trained in
                                        place inside the whole contextual table f32[n_nodes][ld]:
                                        d_context = table + part * ld, context_ld = parts * ld   */
    const uint32_t *d_inv;           /* the round's placement (gn2v_block_placement), or NULL.
                                        Resident cells only: row r of cell (part, slice) is the
                                        contextual row of node x = d_inv[(slice + slices * r) *
                                        parts + part], found at d_context_table + x * ld, or --
                                        d_context_table NULL, placements with classes = parts --
                                        at d_context + (x / parts) * context_ld                   */
    float *d_context_table;          /* the whole contextual table f32[n_nodes][ld] in node order */
} gn2v_block_io;

/* The fused gather -> dot -> sigmoid -> scatter-add step over the pairs of one part: a wavefront
 * takes `record` consecutive sorted pairs at a time (records visited in a golden-ratio stride
 * order), keeps the centre row in registers while the centre does not change and applies
 * [context, k negatives] per pair.  tp: d, ld, k, clip, flags (update mode as gn2v_sgns_step;
 * parts with one slice per XCD use plain stores for the contextual rows: they are exclusive to
 * one XCD; any other slicing keeps the write-through stores). */
int gn2v_block_step(gn2v_graph *g, const gn2v_train_params *tp, const gn2v_block_plan *plan,
                    const gn2v_block_io *io, uint64_t seed, uint64_t epoch, float lr,
                    void *stream);

/* Leave `cus_per_xcd` compute units of every XCD to other work: from now on the training kernels
 * of the multi-GPU path (gn2v_block_step; the walk-ordered gn2v_sgns_step / gn2v_cbow_step keep
 * the caller's stream: they never run beside a transfer) run on a stream of the library's own
 * created with hipExtStreamCreateWithCUMask, ordered after the caller's stream at entry and before
 * it at exit (two events per call).  For multi-GPU jobs: the training kernel's workgroups stay
 * resident for a whole launch, and RCCL's transfer kernels must find a CU.  The
 * mask is verified by a probe launch: active_per_xcd (u32[16], optional) receives the CUs each
 * XCD really runs workgroups on.  0 = back to the caller's stream.  Also set at graph creation
 * from the environment variable GN2V_RESERVE_CUS. */
int gn2v_graph_reserve_cus(gn2v_graph *g, uint32_t cus_per_xcd, uint32_t *active_per_xcd);

/* XCDs (accelerator complexes, one L2 each) the workgroups of this graph's device are spread
 * over, found by a probe launch in gn2v_graph_create: 8 on an MI355X; 0 = unknown. */
int gn2v_graph_xcds(gn2v_graph *g);

/* (parts, slices) of the contextual table for a graph of n_nodes on `world` ranks.
 * Row stride ld <= 512 floats (ld = 0: unknown, this rule is skipped), k negatives, a graph of
 * GN2V_RESIDENT_MIN_NODES up to GN2V_RESIDENT_MAX_NODES nodes: RESIDENT CELLS -- cells of at
 * most the rows that fit one workgroup's LDS beside its staging (220 at d = 128; rows wider than 128
 * floats run workgroups of eight waves instead of sixteen: 134 at 256, 66 at 512), up to
 * GN2V_BLOCK_MAX_SLICES slices per part (one workgroup per cell), as many parts as needed (169 k
 * nodes: 4 x 256 cells of 166 rows; 1 M: 18 x 256; 10 M: 178 x 256 -- 256 slices per part on
 * one GPU; several ranks: two parts per rank with as many slices as hold the rows, 10 M nodes on
 * 8 GPUs: 16 x 2 841, while a part keeps 64 cells).
 * gn2v_block_step then reads and updates every contextual row in the LDS of the one workgroup
 * that owns it: no other CU races for it; gn2v_block_round launches a whole group of parts at
 * once, so that no CU waits for a part's heaviest cell.
 * Otherwise XCD CELLS: slices = 8 (one per XCD; 1 on graphs too small to keep 8 192 rows in a
 * cell) and as many parts (any count; a multiple of world, at least two per rank) as keep
 * >= 32 768 rows in a cell -- the size from which the link quality of racing stores is at or above
 * the walk-ordered schedule's: 10 M nodes -> 38 x 8, 100 M -> 381 x 8. */
#define GN2V_BLOCK_MAX_SLICES 8192u
#define GN2V_BLOCK_MAX_CELLS 524288u      /* parts x slices of a plan                           */
#define GN2V_BLOCK_MAX_GROUP_CELLS 16384u /* cells of an extraction group that are COUNTED in LDS
                                            * (less when the walk staging leaves less than 64 KB) */
/* A group may hold more cells than that ("wide": several ranks, where every scan of a group reads
 * the walks of all ranks and fewer, larger groups pay): gn2v_block_count then counts the pairs
 * only (d_cell_offsets[cells] = their number, the other entries are not written) and the caller
 * asks for the offsets after the sort -- gn2v_block_cell_offsets. */
#define GN2V_BLOCK_MAX_WIDE_GROUP_CELLS 65536u
/* Resident plans (more than 16 slices) sort a group's pair words by (cell, the HIGHEST
 * GN2V_RESIDENT_CENTRE_SORT_BITS bits of the centre row), ties in extraction order: three radix
 * passes instead of five.  What their kernel needs of the order inside a cell is that the pairs of
 * one walk position -- up to 2 w pairs of ONE context, which four groups of a wave would otherwise
 * update side by side -- are scattered, and that is what any bits of the centre do; pairs per
 * (cell, centre) are one or two there, so nothing is gained from equal centres being adjacent.
 * XCD plans keep the full (cell, centre) order: their kernel trains runs of equal centres. */
#define GN2V_RESIDENT_CENTRE_SORT_BITS 8u
#define GN2V_RESIDENT_MIN_NODES 100000u
#define GN2V_RESIDENT_MAX_NODES 115000000u /* 523 776 cells of 220 rows (d = 128, k = 10)       */
int gn2v_block_auto_plan(uint64_t n_nodes, uint32_t world, uint32_t ld, uint32_t k,
                         uint32_t *parts, uint32_t *slices);
/* The same rule for a graph at hand (what gn2v_train_blocks and the Python trainer use): resident
 * cells only while the graph's largest in-degree (computed once per handle) stays below
 * n_edges / CUs -- a launch of resident cells cannot end before its heaviest cell, and the cell
 * of a context that frequent would hold its launch up for a round's worth; such graphs keep the
 * XCD cells,
 * whose records are handed out by tickets to all workgroups of a slice. */
int gn2v_block_auto_plan_graph(gn2v_graph *g, uint32_t world, uint32_t ld, uint32_t k,
                               uint32_t *parts, uint32_t *slices, void *stream);

/* Walks per rank and round, and parts per extraction group, for `free_bytes` of HBM (what is free
 * once tables and graph are resident).  The longer a round, the more pairs of a centre meet in a
 * cell (its row is read once per such run: kernel 0.82 / 0.92 / 0.96 of the roofline at 2^20 /
 * 2^22 / 2^23 walks on the bench graph): the power of two in [2^20, 2^23] that gives 64 pairs per
 * (cell, centre), less when memory is short (>= 2^14).  group_parts, in equal groups: XCD plans
 * at least four groups a round; resident plans on one GPU at least six, where a group is one launch
 * and holds at least 4 096 cells when the plan has them (the round is shortened down to 2^20 walks
 * before such a group is cut; groups above that floor are cut to a third of free_bytes, what a
 * handle keeps between fits); resident plans on several ranks at least two (every scan of a group
 * reads the walks of ALL ranks).  Resident groups may be wide (GN2V_BLOCK_MAX_WIDE_GROUP_CELLS).
 * More, smaller groups when three quarters of free_bytes do not hold the walks plus, per group,
 * its pair words once sorted (twice with `overlap`: the next group is prepared while this one
 * trains) and once unsorted, 8 B per pair.  Pure host function; every rank of a job must use the
 * same values (take the minimum).  gn2v_train_blocks cuts an epoch into EQUAL rounds of at most
 * round_walks. */
int gn2v_block_round_plan(uint64_t free_bytes, uint64_t n_nodes, uint32_t walk_length,
                          uint32_t window, uint32_t world, uint32_t parts, uint32_t slices,
                          uint32_t overlap, uint64_t *round_walks, uint32_t *group_parts);

/* One round of the block schedule on ONE GPU (world = 1) -- the round driver shared by
 * gn2v_train_blocks and by embiggen_amd.distributed.BlockPartitionedTrainer, so that one host
 * loop orders the launches of a fit (the training half of `self._model.fit_transform(graph)`,
 * embedders/ensmallen_embedders/node2vec.py:99).  d_walks = the round's n_walks walks (ids
 * first_walk ...).  For each of the `stripes` centre stripes in turn (plans[j]: the plan with
 * world = stripes, rank = j; stripes = 1: plans[0] with world = 1) and each group of
 * `group_parts` consecutive parts: gn2v_block_count, ONE host read (the group's pair count),
 * gn2v_block_extract, then one gn2v_block_step per part of the group on the stripe's rows of
 * the whole central table (d_central + j * ld, stride stripes * ld), negatives' stream
 * block_id = round_id * stripes + j.
 * Units (stripe, group) are numbered stripe * groups + group; the call starts at io->next_unit
 * and leaves there the first unit it has not trained.  Returns 0 when the round is over
 * (next_unit = stripes * groups), GN2V_ROUND_GROW when the next unit's pairs exceed
 * io->pairs_capacity or its temporary storage io->temp_bytes: io->needed_pairs says how many
 * pairs it has, nothing of that unit was trained, the caller provides larger buffers
 * (gn2v_block_extract_temp_bytes) and calls again with the io otherwise unchanged. */
#define GN2V_ROUND_GROW 3
typedef struct {
    const uint32_t *d_walks;        /* u32[n_walks][walk_length]                                */
    const uint32_t *d_placed_walks; /* a round under a placement: the walks with placed ids,    */
    const uint32_t *d_inv;          /* the placement's inverse (both or neither) and, one table */
    float *d_context_table;         /* in node order, the contextual table (else NULL: parts)   */
    const uint64_t *d_alias;        /* the five tables of gn2v_block_alias (NULL where          */
    const uint64_t *d_cell_rows;    /* gn2v_block_io / gn2v_block_extract allow it)             */
    const uint32_t *d_hub_bits;
    const uint32_t *d_hot_list;
    const uint8_t *d_hot_slot;
    float *d_central;               /* the whole central table f32[n_nodes][ld]                 */
    float *const *context_parts;    /* HOST array [parts] of device pointers: part p's rows     */
    uint64_t context_ld;            /* floats between consecutive rows of a part (0 = ld)       */
    uint64_t *d_work;               /* u64[GN2V_BLOCK_WORK_WORDS]                               */
    uint64_t *d_cell_offsets;       /* u64[cells + 1]                                           */
    uint64_t *d_pairs;              /* u64[pairs_capacity]: the sorted pair words of a group    */
    uint64_t pairs_capacity;
    void *d_temp;                   /* gn2v_block_extract_temp_bytes(pairs of a group) bytes    */
    uint64_t temp_bytes;
    uint32_t group_parts;           /* parts per extraction group (0 = all parts at once)       */
    uint32_t next_unit;             /* in / out                                                 */
    uint64_t needed_pairs;          /* out, with GN2V_ROUND_GROW                                */
    uint64_t pairs_trained;         /* out: pairs of the units this call trained are ADDED      */
    /* a second set (all three or none; d_pairs2 of pairs_capacity words): the next unit is
     * counted, extracted and sorted on a stream of the library's own while this one trains --
     * the resident kernel is bound by the L2 atomic units, the preparation by HBM            */
    uint64_t *d_pairs2;
    uint64_t *d_cell_offsets2;
    uint64_t *d_work2;
} gn2v_block_round_io;
int gn2v_block_round(gn2v_graph *g, const gn2v_train_params *tp, const gn2v_block_plan *plans,
                     uint32_t stripes, gn2v_block_round_io *io, uint64_t n_walks, uint64_t seed,
                     uint64_t epoch, uint64_t first_walk, float lr, uint64_t round_id,
                     void *stream);

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

/* Traffic-calibration utility: table[ids[i]][:] += 1 for i < n with the access shape and store
 * flavour (flags: GN2V_TRAIN_ATOMIC / _WRITE_BACK / _WRITE_THROUGH, default write-through) of the
 * training kernels.  With distinct ids the HBM bytes of the launch are exactly n * ld * 8 + n * 4,
 * which calibrates the rocprofv3 FETCH_SIZE / WRITE_SIZE counters (profiles/README.md). */
int gn2v_touch_rows(float *d_table, uint32_t ld, const uint32_t *d_ids, uint64_t n, uint32_t flags,
                    void *stream);

/* A block fit (gn2v_train on graphs of >= GN2V_BLOCK_PATH_MIN_NODES nodes, gn2v_train_blocks)
 * hands its round buffers -- walks, pair words, sort storage: tens of GB on large graphs -- back
 * to the graph handle, which keeps up to a third of the device's memory for the handle's next
 * fit (a second fit otherwise waits 1.5-2 s for the driver to clear the same 65 GB again).  This
 * frees them now; gn2v_graph_destroy does too; GN2V_KEEP_BUFFERS=0 in the environment never
 * keeps any.  The Python classes call it after every fit unless told to keep the buffers. */
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

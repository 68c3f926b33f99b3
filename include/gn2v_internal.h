/* gn2v_internal.h -- the steps gn2v_train_blocks (gn2v.h) is made of, one entry point each:
 * plan, alias tables, a round's placement, pair extraction, one training step of one part, one
 * round.  NOT part of the drop-in boundary: a binding of the reference's call
 * (embedders/ensmallen_embedders/node2vec.py:99) needs gn2v.h only.  They are exported because
 * the multi-process trainer (embiggen_amd/distributed.py) drives the rounds of several ranks
 * through them and because the parity tests compare every step with the oracle's restatement
 * (oracle/gn2v_oracle.c, "block-partitioned SkipGram") one at a time.  Same conventions as
 * gn2v.h: int status (0 = OK, message in gn2v_last_error), plain pointers and sizes, the
 * caller's stream as void*. */
#ifndef GN2V_INTERNAL_H
#define GN2V_INTERNAL_H

#include "gn2v.h"

#ifdef __cplusplus
extern "C" {
#endif

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
 * at least four groups a round; resident plans on one GPU as few as fit, where a group is one launch
 * and holds at least 4 096 cells when the plan has them (the round is shortened down to 2^20 walks
 * before such a group is cut; groups above that floor are cut to a third of free_bytes, what a
 * handle keeps between fits); resident plans on several ranks ONE group when memory allows (every
 * scan of a group reads the walks of ALL ranks).  Resident groups may be wide
 * (GN2V_BLOCK_MAX_WIDE_GROUP_CELLS).  *round_walks on entry = the caller's cap on the round
 * (0 = none: resident cells cut an epoch into 16-64 rounds, gn2v_train_blocks) -- the groups are
 * sized for the round that will be trained.
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
    /* optional: a hipEvent_t the TRAINING launches of this call wait for (its preparation does
     * not) -- the caller that prepares round t + 1 on another stream while round t trains
     * (gn2v_train_blocks: two lanes of round buffers) hands over round t's "trained" event      */
    void *train_after;
} gn2v_block_round_io;
int gn2v_block_round(gn2v_graph *g, const gn2v_train_params *tp, const gn2v_block_plan *plans,
                     uint32_t stripes, gn2v_block_round_io *io, uint64_t n_walks, uint64_t seed,
                     uint64_t epoch, uint64_t first_walk, float lr, uint64_t round_id,
                     void *stream);

/* Traffic-calibration utility: table[ids[i]][:] += 1 for i < n with the access shape and store
 * flavour (flags: GN2V_TRAIN_ATOMIC / _WRITE_BACK / _WRITE_THROUGH, default write-through) of the
 * training kernels.  With distinct ids the HBM bytes of the launch are exactly n * ld * 8 + n * 4,
 * which calibrates the rocprofv3 FETCH_SIZE / WRITE_SIZE counters (profiles/README.md). */
int gn2v_touch_rows(float *d_table, uint32_t ld, const uint32_t *d_ids, uint64_t n, uint32_t flags,
                    void *stream);

#ifdef __cplusplus
}
#endif

#endif /* GN2V_INTERNAL_H */

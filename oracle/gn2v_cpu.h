/* gn2v_cpu.h -- the CPU twins of the boundary's compute entry points ("CPU twins
 * gn2v_cpu_* with the same signatures (oracle + baseline)").
 *
 * TEST INFRASTRUCTURE, like everything under oracle/: the twins live in libgn2v_oracle.so, never
 * in libgn2v.so (the product has no CPU path and fails without its HIP library), and only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load them.  PARITY UNPINNED, as the
 * oracle they wrap (gn2v_oracle.c's header: the reference's arithmetic lives in the un-vendored
 * ensmallen wheel behind embedders/ensmallen_embedders/node2vec.py:99).
 *
 * Each twin takes the arguments of its include/gn2v.h namesake -- same structs, same order, same
 * meaning -- with HOST pointers wherever the original takes device pointers, so that a parity
 * test reads:  gn2v_walks(g, &wp, seed, 0, 0, n, d_out, stream)  /  gn2v_cpu_walks(cg, &wp, seed,
 * 0, 0, n, out, NULL)  and compares.  What differs, by construction:
 *   - gn2v_cpu_graph_create's `threads` stands where gn2v_graph_create has `device`: <= 1 = the
 *     strictly sequential schedule (walk order; THE CHECKER), > 1 = OpenMP Hogwild over walks
 *     (the timed CPU baseline).  The arrays are borrowed (they must outlive the handle);
 *   - `stream` is accepted and ignored;
 *   - gn2v_cpu_train always runs the walk-ordered schedule with graph-wide negatives -- the
 *     reference's documented semantics (node2vec_skipgram.py:37-119) -- i.e. what gn2v_train runs
 *     with GN2V_TRAIN_WALK_ORDERED | GN2V_TRAIN_DETERMINISTIC; the block path's schedule has its
 *     own restatement (o_block_step and friends in gn2v_oracle.c).
 * All return 0 on success, 1 on a bad argument (message: gn2v_cpu_last_error()). */
#ifndef GN2V_CPU_H
#define GN2V_CPU_H

#include "../include/gn2v.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gn2v_cpu_graph gn2v_cpu_graph;

const char *gn2v_cpu_last_error(void);

/* twin of gn2v_graph_create (include/gn2v.h); flags must be 0 */
int gn2v_cpu_graph_create(const uint64_t *row_ptr, const uint32_t *col_idx, const float *cumw,
                          const uint32_t *sources, uint64_t n_nodes, uint64_t n_edges,
                          uint64_t n_sources, uint32_t flags, int threads, gn2v_cpu_graph **out);
int gn2v_cpu_graph_destroy(gn2v_cpu_graph *g);
/* twin of gn2v_graph_set_types */
int gn2v_cpu_graph_set_types(gn2v_cpu_graph *g, const uint32_t *node_types,
                             const uint32_t *edge_types);

/* twin of gn2v_walks: Graph.complete_walks / the walk half of Graph.node2vec
 * (sequences/tensorflow_sequences/node2vec_sequence.py:190-201) */
int gn2v_cpu_walks(gn2v_cpu_graph *g, const gn2v_walk_params *wp, uint64_t seed, uint64_t epoch,
                   uint64_t first_walk, uint64_t n_walks, uint32_t *out, void *stream);
/* twin of gn2v_walks_strided: walk b has id first_walk + (b / group) * stride + b % group */
int gn2v_cpu_walks_strided(gn2v_cpu_graph *g, const gn2v_walk_params *wp, uint64_t seed,
                           uint64_t epoch, uint64_t first_walk, uint64_t n_walks, uint32_t group,
                           uint64_t stride, uint32_t *out, void *stream);
/* twin of gn2v_window_batch: the Node2VecSequence batch (node2vec_sequence.py:115-128) */
int gn2v_cpu_window_batch(const uint32_t *walks, uint64_t n_walks, uint32_t walk_length,
                          uint32_t window, int32_t *contexts, int32_t *words, void *stream);
/* twin of gn2v_init_table */
int gn2v_cpu_init_table(float *table, uint64_t n_rows, uint32_t d, uint32_t ld, uint64_t seed,
                        uint32_t table_id, float scale, void *stream);
/* twins of gn2v_sgns_step / gn2v_cbow_step: one batch of explicit walks, tables updated in place */
int gn2v_cpu_sgns_step(gn2v_cpu_graph *g, const gn2v_train_params *tp, const uint32_t *walks,
                       uint64_t n_walks, uint32_t walk_length, uint64_t seed, uint64_t epoch,
                       uint64_t first_walk, float lr, float *central, float *contextual,
                       const uint32_t *neg_override, void *stream);
int gn2v_cpu_cbow_step(gn2v_cpu_graph *g, const gn2v_train_params *tp, const uint32_t *walks,
                       uint64_t n_walks, uint32_t walk_length, uint64_t seed, uint64_t epoch,
                       uint64_t first_walk, float lr, float *central, float *contextual,
                       const uint32_t *neg_override, void *stream);
/* twin of gn2v_train: the whole of models.SkipGram/CBOW(...).fit_transform(graph)
 * (node2vec.py:99); fills stats->pairs, ->centres, ->walk_steps */
int gn2v_cpu_train(gn2v_cpu_graph *g, const gn2v_walk_params *wp, const gn2v_train_params *tp,
                   uint64_t seed, uint64_t max_walks_per_epoch, float *central,
                   float *contextual, gn2v_stats *stats, void *stream);

#ifdef __cplusplus
}
#endif
#endif

/* gn2v_experimental.h -- entry points of measured-and-rejected designs, kept for the scripts and
 * tests that document why they were rejected.  NOT part of the drop-in boundary (include/gn2v.h):
 * nothing a binder of the reference's call (embedders/ensmallen_embedders/node2vec.py:99) needs,
 * no compatibility promise.  libgn2v.so exports them. */
#ifndef GN2V_EXPERIMENTAL_H
#define GN2V_EXPERIMENTAL_H

#include "gn2v.h"

#ifdef __cplusplus
extern "C" {
#endif

/* General form of one training step (the row-cache experiments of DESIGN.md section 7.1 run
 * through it): the walk nodes may live in compact row caches (d_walk_rows gives the row of every
 * walk position in d_central / d_contextual) while negatives are drawn from a caller-supplied pool
 * of rows of a third table (the local shard).  With every optional field NULL / 0 this is exactly
 * gn2v_sgns_step / gn2v_cbow_step. */
typedef struct {
    const uint32_t *d_walks;     /* global node ids u32[n_walks][walk_length]                    */
    const uint32_t *d_walk_rows; /* optional u32[n_walks][walk_length]: row of each walk node    */
    float *d_central;
    float *d_contextual;
    float *d_negative;           /* optional: table of the negative rows (default: d_contextual
                                    for SkipGram, d_central for CBOW)                            */
    const uint32_t *d_neg_pool;  /* optional: negatives = d_neg_pool[uniform draw]               */
    uint64_t neg_pool_size;
    uint32_t neg_id_mul;         /* global id of negative row r = r * mul + add (0, 0 = identity), */
    uint32_t neg_id_add;         /*   used to skip negatives equal to the centre / context       */
    const uint32_t *d_neg_override;
    float *d_context_delta;      /* optional, CBOW: the input-side gradient of every centre is
                                    ADDED (f32 atomics) to this table f32[rows][ld] instead of
                                    being applied to d_contextual, which is then only read during
                                    the launch: the caller applies the sum later (the batch form a
                                    CBOW spread over several GPUs needs: DESIGN.md 8)             */
} gn2v_step_io;

int gn2v_step(gn2v_graph *g, const gn2v_train_params *tp, const gn2v_step_io *io,
              uint64_t n_walks, uint32_t walk_length, uint64_t seed, uint64_t epoch,
              uint64_t first_walk, float lr, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GN2V_EXPERIMENTAL_H */

/* gn2v_cpu.c -- the CPU twins declared in gn2v_cpu.h: the boundary's compute entry points
 * (include/gn2v.h) with host pointers, each a few lines over the oracle (gn2v_oracle.c).
 *
 * TEST INFRASTRUCTURE ONLY (part of libgn2v_oracle.so; see gn2v_cpu.h).  PARITY UNPINNED, as the
 * oracle it wraps.  The parameter structs of the oracle (o_walk_params, o_train_params) have the
 * layout of gn2v_walk_params / gn2v_train_params field for field -- asserted below. */
#include "gn2v_cpu.h"

#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* the oracle's side (gn2v_oracle.c); sanitize_check.c includes both sources in one unit */
#ifndef GN2V_ORACLE_INLINE
typedef struct {
    uint64_t n_nodes;
    uint64_t n_edges;
    const uint64_t *row_ptr;
    const uint32_t *col_idx;
    const float *cumw;
    const uint32_t *node_types;
    const uint32_t *edge_types;
} o_graph;
void o_walks(const o_graph *g, const void *wp, const uint32_t *sources, uint64_t n_sources,
             uint64_t seed, uint64_t epoch, uint64_t first_walk, uint64_t n_walks, uint32_t *out);
void o_init_table(float *t, uint64_t n_rows, uint32_t d, uint32_t ld, uint64_t seed,
                  uint32_t table_id, float scale);
void o_train_walks(const o_graph *g, const void *tp, const uint32_t *walks, uint64_t n_walks,
                   uint32_t L, uint64_t seed, uint64_t epoch, uint64_t first_walk, float lr,
                   float *central, float *contextual, const uint32_t *neg_override, int threads);
void o_count_units(const o_graph *g, const void *tp, const uint32_t *walks, uint64_t n_walks,
                   uint32_t L, uint64_t seed, uint64_t epoch, uint64_t first_walk,
                   uint64_t out[3]);
uint64_t o_window_batch(const uint32_t *walks, uint64_t n_walks, uint32_t L, uint32_t w,
                        int32_t *contexts, int32_t *words);
#endif

_Static_assert(sizeof(gn2v_walk_params) == 32 && offsetof(gn2v_walk_params, max_neighbours) == 16 &&
                   offsetof(gn2v_walk_params, change_edge_type_weight) == 28,
               "gn2v_walk_params no longer has o_walk_params' layout");
_Static_assert(sizeof(gn2v_train_params) == 48 && offsetof(gn2v_train_params, flags) == 36 &&
                   offsetof(gn2v_train_params, min_dist) == 44,
               "gn2v_train_params no longer has o_train_params' layout");

struct gn2v_cpu_graph {
    o_graph g;
    const uint32_t *sources;
    uint64_t n_sources;
    int threads;
};

static _Thread_local char last_error[256];

static int fail(const char *msg) {
    snprintf(last_error, sizeof(last_error), "%s", msg);
    return 1;
}

const char *gn2v_cpu_last_error(void) { return last_error; }

int gn2v_cpu_graph_create(const uint64_t *row_ptr, const uint32_t *col_idx, const float *cumw,
                          const uint32_t *sources, uint64_t n_nodes, uint64_t n_edges,
                          uint64_t n_sources, uint32_t flags, int threads, gn2v_cpu_graph **out) {
    if (!row_ptr || (!col_idx && n_edges) || !out) return fail("NULL pointer");
    if (flags) return fail("the CPU twin takes host arrays only (flags must be 0)");
    if (n_nodes == 0 || n_nodes > 0xFFFFFFFEull) return fail("n_nodes out of range");
    if (row_ptr[n_nodes] != n_edges) return fail("row_ptr[n_nodes] != n_edges");
    gn2v_cpu_graph *g = (gn2v_cpu_graph *)calloc(1, sizeof(*g));
    if (!g) return fail("out of memory");
    g->g.n_nodes = n_nodes;
    g->g.n_edges = n_edges;
    g->g.row_ptr = row_ptr;
    g->g.col_idx = col_idx;
    g->g.cumw = cumw;
    g->sources = sources;
    g->n_sources = sources ? n_sources : n_nodes;
    g->threads = threads;
    *out = g;
    return 0;
}

int gn2v_cpu_graph_destroy(gn2v_cpu_graph *g) {
    free(g);
    return 0;
}

int gn2v_cpu_graph_set_types(gn2v_cpu_graph *g, const uint32_t *node_types,
                             const uint32_t *edge_types) {
    if (!g) return fail("NULL handle");
    g->g.node_types = node_types;
    g->g.edge_types = edge_types;
    return 0;
}

int gn2v_cpu_walks(gn2v_cpu_graph *g, const gn2v_walk_params *wp, uint64_t seed, uint64_t epoch,
                   uint64_t first_walk, uint64_t n_walks, uint32_t *out, void *stream) {
    (void)stream;
    if (!g || !wp || (!out && n_walks)) return fail("NULL handle / params / output");
    if (wp->walk_length < 2) return fail("walk_length must be at least 2");
    if (g->n_sources == 0) return fail("the graph has no source node");
    o_walks(&g->g, (const void *)wp, g->sources, g->n_sources, seed, epoch, first_walk, n_walks, out);
    return 0;
}

int gn2v_cpu_walks_strided(gn2v_cpu_graph *g, const gn2v_walk_params *wp, uint64_t seed,
                           uint64_t epoch, uint64_t first_walk, uint64_t n_walks, uint32_t group,
                           uint64_t stride, uint32_t *out, void *stream) {
    if (group == 0) return fail("group must be at least 1");
    for (uint64_t b = 0, q = 0; b < n_walks; b += group, ++q) {
        const uint64_t nb = n_walks - b < group ? n_walks - b : group;
        if (gn2v_cpu_walks(g, wp, seed, epoch, first_walk + q * stride, nb,
                           out ? out + b * wp->walk_length : NULL, stream))
            return 1;
    }
    return 0;
}

int gn2v_cpu_window_batch(const uint32_t *walks, uint64_t n_walks, uint32_t walk_length,
                          uint32_t window, int32_t *contexts, int32_t *words, void *stream) {
    (void)stream;
    if (!walks || !contexts || !words) return fail("NULL pointer");
    if (window < 1 || walk_length <= 2 * window) return fail("walk_length must exceed 2 * window");
    o_window_batch(walks, n_walks, walk_length, window, contexts, words);
    return 0;
}

int gn2v_cpu_init_table(float *table, uint64_t n_rows, uint32_t d, uint32_t ld, uint64_t seed,
                        uint32_t table_id, float scale, void *stream) {
    (void)stream;
    if (!table) return fail("NULL table");
    if (d < 1 || ld < d) return fail("need 1 <= d <= ld");
    o_init_table(table, n_rows, d, ld, seed, table_id, scale);
    return 0;
}

static int step(gn2v_cpu_graph *g, const gn2v_train_params *tp, uint32_t model,
                const uint32_t *walks, uint64_t n_walks, uint32_t walk_length, uint64_t seed,
                uint64_t epoch, uint64_t first_walk, float lr, float *central, float *contextual,
                const uint32_t *neg_override) {
    if (!g || !tp || !central || !contextual || (!walks && n_walks))
        return fail("NULL handle / params / tables / walks");
    if (tp->d < 1 || tp->ld < tp->d || tp->window < 1 || walk_length < 2)
        return fail("need 1 <= d <= ld, window >= 1, walk_length >= 2");
    gn2v_train_params t = *tp; /* tp->model is ignored, as by gn2v_sgns_step / gn2v_cbow_step */
    t.model = model;
    t.flags &= 7u; /* scale-free negatives, downsampling, lr by degree: the bits with a meaning here */
    o_train_walks(&g->g, (const void *)&t, walks, n_walks, walk_length, seed, epoch, first_walk, lr, central,
                  contextual, neg_override, g->threads);
    return 0;
}

int gn2v_cpu_sgns_step(gn2v_cpu_graph *g, const gn2v_train_params *tp, const uint32_t *walks,
                       uint64_t n_walks, uint32_t walk_length, uint64_t seed, uint64_t epoch,
                       uint64_t first_walk, float lr, float *central, float *contextual,
                       const uint32_t *neg_override, void *stream) {
    (void)stream;
    return step(g, tp, GN2V_MODEL_SKIPGRAM, walks, n_walks, walk_length, seed, epoch, first_walk,
                lr, central, contextual, neg_override);
}

int gn2v_cpu_cbow_step(gn2v_cpu_graph *g, const gn2v_train_params *tp, const uint32_t *walks,
                       uint64_t n_walks, uint32_t walk_length, uint64_t seed, uint64_t epoch,
                       uint64_t first_walk, float lr, float *central, float *contextual,
                       const uint32_t *neg_override, void *stream) {
    (void)stream;
    return step(g, tp, GN2V_MODEL_CBOW, walks, n_walks, walk_length, seed, epoch, first_walk, lr,
                central, contextual, neg_override);
}

int gn2v_cpu_train(gn2v_cpu_graph *g, const gn2v_walk_params *wp, const gn2v_train_params *tp,
                   uint64_t seed, uint64_t max_walks_per_epoch, float *central,
                   float *contextual, gn2v_stats *stats, void *stream) {
    (void)stream;
    if (!g || !wp || !tp || !central || !contextual) return fail("NULL handle / params / tables");
    if (tp->model > GN2V_MODEL_CBOW) return fail("unknown model");
    if (tp->d < 1 || tp->ld < tp->d || tp->window < 1 || wp->walk_length < 2)
        return fail("need 1 <= d <= ld, window >= 1, walk_length >= 2");
    if (g->n_sources == 0) return fail("the graph has no source node");
    const uint32_t L = wp->walk_length;
    uint64_t n_walks = g->n_sources * (uint64_t)wp->iterations;
    if (max_walks_per_epoch && max_walks_per_epoch < n_walks) n_walks = max_walks_per_epoch;
    gn2v_train_params t = *tp;
    t.flags &= 7u;
    o_init_table(central, g->g.n_nodes, t.d, t.ld, seed, 0, t.init_scale);
    o_init_table(contextual, g->g.n_nodes, t.d, t.ld, seed, 1, t.init_scale);
    /* a launch's worth of walks at a time, like gn2v_train: the memory of 2^16 walks, not an epoch's */
    const uint64_t batch = 1ull << 16;
    uint32_t *walks = (uint32_t *)malloc(sizeof(uint32_t) * (n_walks < batch ? n_walks : batch) * L + 4);
    if (!walks) return fail("out of memory");
    uint64_t units[3] = {0, 0, 0};
    float lr = t.lr;
    for (uint32_t e = 0; e < t.epochs; ++e) {
        for (uint64_t first = 0; first < n_walks; first += batch) {
            const uint64_t nb = n_walks - first < batch ? n_walks - first : batch;
            o_walks(&g->g, (const void *)wp, g->sources, g->n_sources, seed, e, first, nb, walks);
            o_train_walks(&g->g, (const void *)&t, walks, nb, L, seed, e, first, lr, central, contextual, NULL,
                          g->threads);
            o_count_units(&g->g, (const void *)&t, walks, nb, L, seed, e, first, units);
        }
        lr *= t.lr_decay;
    }
    free(walks);
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->pairs = units[0];
        stats->walk_steps = units[1];
        stats->centres = units[2];
    }
    return 0;
}

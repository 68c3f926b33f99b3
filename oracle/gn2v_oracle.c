/*
 * gn2v_oracle.c -- CPU restatement of the Node2Vec / SkipGram / CBOW hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call anything in oracle/.
 *
 * PARITY UNPINNED.  The reference (monarch-initiative/embiggen 0.11.96) performs this path in one
 * opaque call, `self._model.fit_transform(graph)`
 * (embiggen/embedders/ensmallen_embedders/node2vec.py:99), into the third-party Rust wheel
 * `ensmallen` (setup.py:76, pinned ">=0.8.94"), whose source is not in /root/reference and which
 * cannot be installed here.  The reference's own tests hold no golden vectors for walks, losses
 * or embeddings (tests/test_node_embedding_pipelines.py:17-42 is does-not-raise only).  This
 * file therefore restates the *published* algorithm (node2vec second-order walks + word2vec
 * negative-sampling SGD) under the semantics the reference documents at its call sites:
 *
 *   - kwargs & their meaning ............ node2vec_skipgram.py:37-119, node2vec_cbow.py:37-119
 *       return_weight = 1/p, explore_weight = 1/q (:58-71); window trimmed at walk borders
 *       (:55-57); dot product clipped at +-clipping_value (:45-47); negatives drawn
 *       proportionally to degree when use_scale_free_distribution (:101-102); learning rate
 *       multiplied by learning_rate_decay each epoch (:84-85); iterations = walks per source
 *       node (:53-54).
 *   - two N x d f32 tables [central, contextual] ... node2vec.py:99-112 and the in-tree model
 *       statement tensorflow_embedders/skipgram.py:28-61 (centre embedding . output weights,
 *       1 positive + k sampled negatives) and cbow.py:26-60 (mean of context embeddings vs
 *       centre + k sampled negatives).
 *   - batch form of walk + window ........ sequences/tensorflow_sequences/node2vec_sequence.py
 *       :115-128,:190-203  (contexts[n,2w], words[n]; n = walks*(walk_length-2w)).
 *   - CSR convention ..................... pecanpy_embedders/node2vec.py:139-163.
 *   - max_neighbours ..................... node2vec_skipgram.py:22,78-81 ("approximated walks
 *       ... for graphs containing nodes with high degrees"): rows longer than it are walked over
 *       a per-visit sub-sample, one edge per bucket of the row -- the sorted unique sub-sampling
 *       its authors published for it (GRAPE, Cappelletti et al. 2023); see row_view below.
 *
 * Everything random is counter based (splitmix64 finaliser keyed by seed/epoch/walk/draw) so the
 * HIP implementation can reproduce walks bit-exactly and training to float tolerance.
 *
 * Plain C11; build: see oracle/Makefile.  Two builds of this one source: libgn2v_oracle.so (strict:
 * float sums in the written order; THE CHECKER of every parity test) and libgn2v_oracle_fast.so
 * (-ffast-math -DO_FAST: reassociated sums + prefetch hints; only bench.py's cpu_baseline times it,
 * so that the CPU figure beside the GPU's is a tuned Hogwild trainer, not a serial-latency one).
 * gn2v_cpu.c (same libraries) wraps the walk sampler, the trainers and the whole fit in the argument
 * lists of include/gn2v.h's entry points (gn2v_cpu.h: the boundary's "CPU twins").
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define O_GOLDEN 0x9E3779B97F4A7C15ULL
#define O_TAG_EPOCH 0x6E32764B45590A01ULL
#define O_TAG_NEG 0xA5A5F00DC0FFEE11ULL
#define O_TAG_DOWN 0x5BD1E995D00D1E55ULL
#define O_TAG_BA 0xBA5EBA11BA5EBA11ULL
#define O_TAG_INIT 0x1417AB1E00000000ULL
#define O_SENTINEL 0xFFFFFFFFu

#define O_FLAG_SCALE_FREE 1u
#define O_FLAG_DOWNSAMPLE 2u
#define O_FLAG_NORM_LR 4u
/* Variants of the two readings DESIGN.md 1.1 decides the other way, in the oracle's SkipGram
 * trainer only (tests/test_ensmallen_hook.py prints what each changes; the device has neither):
 * SKIP_CLIPPED: "clip" read as "skip the update when |dot| > clipping_value" (the [uncited] hint
 * about ensmallen) instead of clamping the dot product; SHARED_NEGATIVES: the k negatives of a
 * centre drawn once and reused for all its contexts (the docstring's "per batch" wording,
 * node2vec_skipgram.py:48-50) instead of k fresh ones per pair. */
#define O_FLAG_SKIP_CLIPPED 256u
#define O_FLAG_SHARED_NEGATIVES 512u

typedef struct {
    uint32_t walk_length;
    uint32_t iterations;
    float return_weight;
    float explore_weight;
    uint32_t max_neighbours; /* 0 = exact; else rows longer than this are walked over a sub-sample
                              * of this many edges per visit (row_view below) */
    uint32_t flags;
    float change_node_type_weight; /* 0 = unset = 1.0; only acts on graphs with node types */
    float change_edge_type_weight; /* 0 = unset = 1.0; only acts on graphs with edge types */
} o_walk_params;

typedef struct {
    uint32_t model; /* 0 = SkipGram, 1 = CBOW */
    uint32_t d;
    uint32_t ld; /* row stride in floats (>= d) */
    uint32_t epochs;
    uint32_t k;
    uint32_t window;
    float lr;
    float lr_decay;
    float clip;
    uint32_t flags;
    float init_scale;
    uint32_t min_dist; /* contexts at distance [min_dist, window] from the centre; 0 = 1 */
} o_train_params;

typedef struct {
    uint64_t n_nodes;
    uint64_t n_edges; /* directed */
    const uint64_t *row_ptr;
    const uint32_t *col_idx;
    const float *cumw; /* per-row inclusive prefix sums of weights, or NULL */
    const uint32_t *node_types; /* one id per node (0xFFFFFFFF = unknown), or NULL */
    const uint32_t *edge_types; /* one id per directed edge, aligned with col_idx, or NULL */
} o_graph;

/* General step I/O (mirrors gn2v_step_io): walk nodes may live in row caches, negatives in a
 * third table sampled through a pool (row-sharded multi-GPU training). */
typedef struct {
    const uint32_t *walks;
    const uint32_t *walk_rows; /* optional */
    float *central;
    float *contextual;
    float *negative;           /* optional */
    const uint32_t *neg_pool;  /* optional */
    uint64_t neg_pool_size;
    uint32_t neg_id_mul, neg_id_add;
    const uint32_t *neg_override;
} o_step_io;

/* ---------------------------------------------------------------- RNG */

uint64_t o_mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

uint64_t o_draw(uint64_t key, uint64_t t) { return o_mix64(key + (t + 1) * O_GOLDEN); }

static inline uint64_t mulhi64(uint64_t a, uint64_t b) {
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
}

uint64_t o_epoch_key(uint64_t seed, uint64_t epoch) {
    return o_draw(o_mix64(seed ^ O_TAG_EPOCH), epoch);
}

uint64_t o_walk_key(uint64_t seed, uint64_t epoch, uint64_t walk_id) {
    return o_draw(o_epoch_key(seed, epoch), walk_id);
}

/* ---------------------------------------------------------------- synthetic graph */

/* Parallel-friendly Barabasi-Albert (pointer-chasing form of Batagelj-Brandes): edge e belongs
 * to node v = e/m + 1 and attaches to the endpoint stored at a uniformly random earlier position
 * of the virtual endpoint array; odd positions are resolved by re-deriving that edge's own draw. */
static uint32_t ba_target(uint64_t bkey, uint64_t e, uint32_t m) {
    for (;;) {
        uint64_t v = e / m + 1;
        uint64_t limit = 2 * (v - 1) * m + 1;
        uint64_t x = mulhi64(o_draw(bkey, e), limit);
        if (x == 0) return 0;
        uint64_t p = x - 1;
        uint64_t e2 = p >> 1;
        if ((p & 1) == 0) return (uint32_t)(e2 / m + 1);
        e = e2;
    }
}

void o_ba_edges(uint64_t n_nodes, uint32_t m, uint64_t seed, uint32_t *src, uint32_t *dst) {
    uint64_t bkey = o_mix64(seed ^ O_TAG_BA);
    uint64_t n_e = (n_nodes - 1) * (uint64_t)m;
    for (uint64_t e = 0; e < n_e; ++e) {
        src[e] = (uint32_t)(e / m + 1);
        dst[e] = ba_target(bkey, e, m);
    }
}

/* ---------------------------------------------------------------- walks */

static inline int adj_contains(const uint32_t *col, uint64_t lo, uint64_t hi, uint32_t x) {
    uint64_t end = hi;
    while (lo < hi) {
        uint64_t mid = lo + ((hi - lo) >> 1);
        if (col[mid] < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo < end && col[lo] == x;
}

/* Integer acceptance thresholds on a 2^32 scale.  A candidate edge cur -> x carries the product
 * of three factors, each normalised by its own maximum:
 *   second order  {return_weight (x == prev), 1 (x adjacent to prev), explore_weight (else)}
 *   node type     change_node_type_weight when type(x) != type(cur)          (typed nodes only)
 *   edge type     change_edge_type_weight when type(cur->x) != type(prev->cur) (typed edges only)
 * Semantics of the two type weights: node2vec_sequence.py:57-66 ("weight on the probability of
 * visiting a neighbor node / edge of a different type than the previous node / edge; only applies
 * to colored graphs / multigraphs, otherwise it has no impact"). */
typedef struct {
    int second, node_bias, edge_bias;
    uint64_t t_ret, t_common, t_explore;
    uint64_t fn_same, fn_diff, fe_same, fe_diff;
    uint32_t max_trials; /* rejections before the exact scan */
    /* "Return apart": when return_weight exceeds every other weight (p < 1 and p < q: walks that
     * like to step back), one envelope for all candidates rejects the typical neighbour with
     * probability 1 - explore_weight / return_weight although only ONE neighbour, the previous
     * node, carries the large weight.  The previous node is then proposed on its own: with
     * probability R / (R + deg M) the candidate is prev (accepted when cur -> prev is an edge),
     * else a uniform neighbour, rejected when it is prev and accepted against the envelope M =
     * max(1, explore_weight) of the OTHER neighbours otherwise.  Same law, 2 x (return 2 /
     * explore 0.5) to 4 x (4 / 0.25) fewer trials.  Unweighted, untyped graphs. */
    int apart;
    uint64_t rq, mq;               /* R and M on a 2^20 scale */
    uint64_t s_common, s_explore;  /* 1 / M and explore_weight / M on a 2^32 scale */
} walk_consts;

static inline uint64_t min_u64(uint64_t a, uint64_t b) { return a < b ? a : b; }

/* Trial budget: with a_min the smallest acceptance probability any candidate can have, 37 / a_min
 * trials leave a chance below e^-37 of reaching the (possibly very long) exact scan; clamped to
 * [32, 1024] so that pathological weights still terminate through the scan. */
static uint32_t trial_budget(const walk_consts *c) {
    double a = 1.0, s = 4294967296.0;
    if (c->second) a *= (double)min_u64(c->t_ret, min_u64(c->t_common, c->t_explore)) / s;
    if (c->node_bias) a *= (double)min_u64(c->fn_same, c->fn_diff) / s;
    if (c->edge_bias) a *= (double)min_u64(c->fe_same, c->fe_diff) / s;
    if (!(a > 37.0 / 1024.0)) return 1024;
    double n = ceil(37.0 / a);
    return n < 32.0 ? 32u : (uint32_t)n;
}

static void type_factors(float weight, uint64_t *same, uint64_t *diff) {
    double w = weight == 0.0f ? 1.0 : (double)weight;
    double mx = w > 1.0 ? w : 1.0;
    double s = 4294967296.0;
    *same = (uint64_t)floor(1.0 / mx * s);
    *diff = (uint64_t)floor(w / mx * s);
}

static walk_consts make_walk_consts(const o_graph *g, const o_walk_params *wp) {
    walk_consts c;
    double rw = wp->return_weight, ew = wp->explore_weight;
    double mx = rw > ew ? rw : ew;
    if (mx < 1.0) mx = 1.0;
    double s = 4294967296.0;
    c.second = !(wp->return_weight == 1.0f && wp->explore_weight == 1.0f);
    c.t_ret = (uint64_t)floor(rw / mx * s);
    c.t_common = (uint64_t)floor(1.0 / mx * s);
    c.t_explore = (uint64_t)floor(ew / mx * s);
    type_factors(wp->change_node_type_weight, &c.fn_same, &c.fn_diff);
    type_factors(wp->change_edge_type_weight, &c.fe_same, &c.fe_diff);
    c.node_bias = g->node_types != NULL && c.fn_same != c.fn_diff;
    c.edge_bias = g->edge_types != NULL && c.fe_same != c.fe_diff;
    c.max_trials = trial_budget(&c);
    double m = ew > 1.0 ? ew : 1.0;
    c.apart = c.second && !c.node_bias && !c.edge_bias && g->cumw == NULL && rw > m &&
              rw <= 1024.0 && m <= 1024.0;
    c.rq = c.mq = c.s_common = c.s_explore = 0;
    if (c.apart) {
        c.rq = (uint64_t)floor(rw * 1048576.0);
        c.mq = (uint64_t)floor(m * 1048576.0);
        c.s_common = (uint64_t)floor(1.0 / m * s);
        c.s_explore = (uint64_t)floor(ew / m * s);
        double a = (double)min_u64(c.s_common, c.s_explore) / s;
        if (!(a > 37.0 / 1024.0)) {
            c.max_trials = 1024;
        } else {
            double n = ceil(37.0 / a);
            c.max_trials = n < 32.0 ? 32u : (uint32_t)n;
        }
    }
    return c;
}

/* t * f / 2^32 with t <= 2^32 and f <= 2^32 (f == 2^32 is the identity) */
static inline uint64_t scale32(uint64_t t, uint64_t f) {
    return f >= (1ULL << 32) ? t : (t * f) >> 32;
}

/* acceptance threshold of candidate edge `e` = cur -> x */
static inline uint64_t accept_threshold(const o_graph *g, const walk_consts *c, uint32_t cur,
                                        uint32_t x, uint64_t e, uint32_t prev, uint64_t pstart,
                                        uint64_t pend, uint32_t ptype) {
    uint64_t t = 1ULL << 32;
    if (c->second && prev != O_SENTINEL)
        t = (x == prev) ? c->t_ret
            : adj_contains(g->col_idx, pstart, pend, x) ? c->t_common
                                                        : c->t_explore;
    if (c->node_bias)
        t = scale32(t, g->node_types[cur] != g->node_types[x] ? c->fn_diff : c->fn_same);
    if (c->edge_bias && prev != O_SENTINEL)
        t = scale32(t, g->edge_types[e] != ptype ? c->fe_diff : c->fe_same);
    return t;
}

/* candidate index within the row of `cur` (uniform, or weight proportional via cumw) */
static inline uint64_t pick_index(const o_graph *g, uint64_t start, uint64_t deg, uint64_t r) {
    if (g->cumw == NULL) return ((r >> 32) * deg) >> 32;
    float total = g->cumw[start + deg - 1];
    float f = (float)(r >> 40) * (1.0f / 16777216.0f) * total;
    uint64_t lo = 0, hi = deg;
    while (lo < hi) { /* first idx with cumw > f */
        uint64_t mid = lo + ((hi - lo) >> 1);
        if (g->cumw[start + mid] > f)
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo < deg ? lo : deg - 1;
}

/* max_neighbours (node2vec_skipgram.py:78-81: "Number of maximum neighbours to consider when using
 * approximated walks ... mainly useful for graphs containing nodes with high degrees"; default 100
 * at :22, None = exact): a step that leaves a node of degree > max_neighbours is taken over a
 * SUB-SAMPLE of max_neighbours of its edges, drawn afresh at every visit.  The reference's
 * implementation lives in the absent ensmallen wheel; what is restated here is the algorithm its
 * authors published for it (GRAPE, Cappelletti et al. 2023, "sorted unique sub-sampling"): the
 * row's edge range is cut into max_neighbours buckets of (almost) equal length and ONE edge is
 * drawn uniformly in every bucket -- sorted, distinct, O(1) per element.  Here bucket j of a row
 * of `deg` edges holds step + (j < rem) edges (step = deg / M, rem = deg % M: every edge lies in
 * exactly one bucket) and its element is lo_j + (hash32(vkey, j) * size_j >> 32), vkey = the FIRST
 * draw of the step from the walk's stream -- counter based, so a candidate costs O(1) and the
 * device draws the same sub-sample.  The step then follows the exact node2vec law ON the sub-sample
 * (rejection on the same thresholds, the exact scan over its M elements as fall-back).
 * [unpinned: a fourth reading, DESIGN.md 1.1] */
typedef struct {
    uint64_t start; /* first edge of the row */
    uint64_t n;     /* elements of the view: the degree, or max_neighbours */
    uint64_t step, rem, vkey;
    int on; /* the view is a sub-sample */
} row_view;

/* the draw of bucket j: murmur3's 32-bit finaliser of (upper half of vkey) + (j + 1) golden32 --
 * a sub-sample needs one per candidate, and a 64-bit splitmix each would be the dearest part of
 * the device's step */
static inline uint32_t view_hash(uint64_t vkey, uint64_t j) {
    uint32_t h = (uint32_t)(vkey >> 32) + ((uint32_t)j + 1u) * 0x9E3779B1u;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

/* element j of the view -> edge id (buckets shorter than 2^32 edges) */
static inline uint64_t view_edge(const row_view *v, uint64_t j) {
    if (!v->on) return v->start + j;
    uint64_t lo = j * v->step + (j < v->rem ? j : v->rem);
    return v->start + lo + (((uint64_t)view_hash(v->vkey, j) * (v->step + (j < v->rem ? 1 : 0))) >> 32);
}

/* the bucket whose range holds the row's edge index i */
static inline uint64_t view_bucket(const row_view *v, uint64_t i) {
    uint64_t big = v->rem * (v->step + 1);
    return i < big ? i / (v->step + 1) : v->rem + (i - big) / v->step;
}

static inline double edge_weight(const o_graph *g, uint64_t start, uint64_t e) {
    return (double)g->cumw[e] - (e > start ? (double)g->cumw[e - 1] : 0.0);
}

/* exact fallback after max_trials rejections: integer-weighted scan over the view of the row
 * (unweighted graphs); weighted graphs scale each threshold by the edge weight in double.
 * Returns the chosen edge. */
static uint64_t exact_scan(const o_graph *g, const walk_consts *c, uint64_t r, uint32_t cur,
                           const row_view *v, uint32_t prev, uint64_t pstart, uint64_t pend,
                           uint32_t ptype) {
    uint64_t n = v->n;
    if (g->cumw == NULL) {
        uint64_t total = 0;
        for (uint64_t j = 0; j < n; ++j) {
            uint64_t e = view_edge(v, j);
            total += accept_threshold(g, c, cur, g->col_idx[e], e, prev, pstart, pend, ptype);
        }
        if (total == 0) return view_edge(v, ((r >> 32) * n) >> 32);
        uint64_t target = mulhi64(r, total), acc = 0;
        for (uint64_t j = 0; j < n; ++j) {
            uint64_t e = view_edge(v, j);
            acc += accept_threshold(g, c, cur, g->col_idx[e], e, prev, pstart, pend, ptype);
            if (acc > target) return e;
        }
        return view_edge(v, n - 1);
    }
    double total = 0.0;
    for (uint64_t j = 0; j < n; ++j) {
        uint64_t e = view_edge(v, j);
        uint64_t thr = accept_threshold(g, c, cur, g->col_idx[e], e, prev, pstart, pend, ptype);
        total += edge_weight(g, v->start, e) * (double)thr;
    }
    double target = (double)(r >> 11) * (1.0 / 9007199254740992.0) * total;
    double acc = 0.0;
    for (uint64_t j = 0; j < n; ++j) {
        uint64_t e = view_edge(v, j);
        uint64_t thr = accept_threshold(g, c, cur, g->col_idx[e], e, prev, pstart, pend, ptype);
        acc += edge_weight(g, v->start, e) * (double)thr;
        if (acc > target) return e;
    }
    return view_edge(v, n - 1);
}

void o_walk_one(const o_graph *g, const o_walk_params *wp, uint64_t wkey, uint32_t start_node,
                uint32_t *out) {
    uint32_t L = wp->walk_length;
    walk_consts c = make_walk_consts(g, wp);
    uint64_t ctr = 0;
    uint32_t cur = start_node, prev = O_SENTINEL, ptype = 0;
    uint64_t pstart = 0, pend = 0;
    out[0] = cur;
    uint32_t t = 1;
    for (; t < L; ++t) {
        uint64_t start = g->row_ptr[cur], end = g->row_ptr[cur + 1];
        uint64_t deg = end - start;
        if (deg == 0) break;
        row_view v = {start, deg, 0, 0, 0, 0};
        if (wp->max_neighbours && deg > wp->max_neighbours) { /* this visit's sub-sample */
            v.on = 1;
            v.n = wp->max_neighbours;
            v.step = deg / v.n;
            v.rem = deg % v.n;
            v.vkey = o_draw(wkey, ctr++);
        }
        uint64_t edge;
        int biased = c.node_bias || (prev != O_SENTINEL && (c.second || c.edge_bias));
        if (g->cumw != NULL && v.on) {
            /* weights on a sub-sample: no cumulative sums to search -- the scan over its
             * elements (thresholds all equal when the step is not biased) */
            uint64_t r = o_draw(wkey, ctr++);
            edge = exact_scan(g, &c, r, cur, &v, biased ? prev : O_SENTINEL, pstart, pend, ptype);
        } else if (!biased || deg == 1) {
            uint64_t r = o_draw(wkey, ctr++);
            edge = v.on ? view_edge(&v, ((r >> 32) * v.n) >> 32)
                        : start + pick_index(g, start, deg, r);
        } else if (c.apart) {
            int accepted = 0;
            edge = start;
            for (uint32_t trial = 0; trial < c.max_trials; ++trial) {
                uint64_t r1 = o_draw(wkey, ctr++);
                if (mulhi64(r1, c.rq + v.n * c.mq) < c.rq) { /* the previous node, if an edge */
                    uint64_t lo = start, hi = end;
                    while (lo < hi) {
                        uint64_t mid = lo + ((hi - lo) >> 1);
                        if (g->col_idx[mid] < prev)
                            lo = mid + 1;
                        else
                            hi = mid;
                    }
                    /* ... of the view: a sub-sample holds it when its bucket drew it */
                    if (lo < end && g->col_idx[lo] == prev &&
                        (!v.on || view_edge(&v, view_bucket(&v, lo - start)) == lo)) {
                        edge = lo;
                        accepted = 1;
                        break;
                    }
                    continue;
                }
                uint64_t r2 = o_draw(wkey, ctr++);
                uint64_t e = view_edge(&v, ((r2 >> 32) * v.n) >> 32);
                uint32_t x = g->col_idx[e];
                if (x == prev) continue;
                uint64_t thr = adj_contains(g->col_idx, pstart, pend, x) ? c.s_common : c.s_explore;
                if ((r2 & 0xFFFFFFFFULL) < thr) {
                    edge = e;
                    accepted = 1;
                    break;
                }
            }
            if (!accepted) {
                uint64_t r = o_draw(wkey, ctr++);
                edge = exact_scan(g, &c, r, cur, &v, prev, pstart, pend, ptype);
            }
        } else {
            int accepted = 0;
            edge = start;
            for (uint32_t trial = 0; trial < c.max_trials; ++trial) {
                uint64_t r = o_draw(wkey, ctr++);
                uint64_t e = v.on ? view_edge(&v, ((r >> 32) * v.n) >> 32)
                                  : start + pick_index(g, start, deg, r);
                uint64_t thr = accept_threshold(g, &c, cur, g->col_idx[e], e, prev, pstart, pend,
                                                ptype);
                if ((r & 0xFFFFFFFFULL) < thr) {
                    edge = e;
                    accepted = 1;
                    break;
                }
            }
            if (!accepted) {
                uint64_t r = o_draw(wkey, ctr++);
                edge = exact_scan(g, &c, r, cur, &v, prev, pstart, pend, ptype);
            }
        }
        uint32_t nxt = g->col_idx[edge];
        if (c.edge_bias) ptype = g->edge_types[edge];
        out[t] = nxt;
        prev = cur;
        pstart = start;
        pend = end;
        cur = nxt;
    }
    for (; t < L; ++t) out[t] = O_SENTINEL;
}

/* walks [first_walk, first_walk + n_walks) of the given epoch.  walk_id = iteration * n_sources +
 * source_index; start node = sources[source_index] (or source_index when sources == NULL). */
void o_walks(const o_graph *g, const o_walk_params *wp, const uint32_t *sources,
             uint64_t n_sources, uint64_t seed, uint64_t epoch, uint64_t first_walk,
             uint64_t n_walks, uint32_t *out) {
    uint64_t ekey = o_epoch_key(seed, epoch);
#pragma omp parallel for schedule(dynamic, 64)
    for (uint64_t b = 0; b < n_walks; ++b) {
        uint64_t wid = first_walk + b;
        uint64_t si = wid % n_sources;
        uint32_t s = sources ? sources[si] : (uint32_t)si;
        o_walk_one(g, wp, o_draw(ekey, wid), s, out + b * wp->walk_length);
    }
}

/* ---------------------------------------------------------------- tables */

/* value(row, col) = (2 * u24 - 1) * scale, u24 from a hash of (seed, table, row * d + col) */
void o_init_table(float *t, uint64_t n_rows, uint32_t d, uint32_t ld, uint64_t seed,
                  uint32_t table_id, float scale) {
    uint64_t key = o_mix64(seed ^ (O_TAG_INIT + table_id));
    for (uint64_t r = 0; r < n_rows; ++r) {
        for (uint32_t c = 0; c < ld; ++c) {
            if (c >= d) {
                t[r * ld + c] = 0.0f;
                continue;
            }
            uint64_t h = o_draw(key, r * d + c);
            float u = (float)(h >> 40) * (1.0f / 16777216.0f);
            t[r * ld + c] = (2.0f * u - 1.0f) * scale;
        }
    }
}

/* ---------------------------------------------------------------- training */

static inline uint32_t draw_negative(const o_graph *g, const o_train_params *tp,
                                     const o_step_io *io, uint64_t nkey, uint64_t q) {
    uint64_t r = o_draw(nkey, q);
    if (io->neg_pool) return io->neg_pool[mulhi64(r, io->neg_pool_size)];
    if (tp->flags & O_FLAG_SCALE_FREE) return g->col_idx[mulhi64(r, g->n_edges)];
    return (uint32_t)mulhi64(r, g->n_nodes);
}

static inline uint32_t neg_global_id(const o_step_io *io, uint32_t row) {
    if (io->neg_id_mul == 0 && io->neg_id_add == 0) return row;
    return row * io->neg_id_mul + io->neg_id_add;
}

static inline float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

static inline uint32_t min_dist_of(const o_train_params *tp) {
    return tp->min_dist ? tp->min_dist : 1;
}

static inline int is_context(uint32_t i, uint32_t j, uint32_t md) {
    return (j > i ? j - i : i - j) >= md;
}

/* number of context positions of centre i in a walk of effective length Le */
static inline uint32_t context_count(uint32_t i, uint32_t Le, uint32_t w, uint32_t md) {
    uint32_t lo = i > w ? i - w : 0, hi = i + w < Le - 1 ? i + w : Le - 1, n = 0;
    for (uint32_t j = lo; j <= hi; ++j) n += is_context(i, j, md);
    return n;
}

static inline uint32_t effective_len(const uint32_t *w, uint32_t L) {
    uint32_t n = 0;
    while (n < L && w[n] != O_SENTINEL) ++n;
    return n;
}

/* stochastic_downsample_by_degree: keep a centre with probability min(1, mean_degree/degree) */
static inline int keep_centre(const o_graph *g, const o_train_params *tp, uint64_t wkey,
                              uint32_t i, uint32_t c) {
    if (!(tp->flags & O_FLAG_DOWNSAMPLE)) return 1;
    uint64_t deg = g->row_ptr[c + 1] - g->row_ptr[c];
    if (deg == 0) return 1;
    /* keep iff r32 * deg < mean_deg * 2^32  (integer form) */
    uint64_t r32 = o_draw(wkey ^ O_TAG_DOWN, i) >> 32;
    unsigned __int128 lhs = (unsigned __int128)r32 * deg * g->n_nodes;
    unsigned __int128 rhs = (unsigned __int128)g->n_edges << 32;
    return lhs < rhs;
}

static inline float centre_lr(const o_graph *g, const o_train_params *tp, float lr, uint32_t c) {
    if (!(tp->flags & O_FLAG_NORM_LR)) return lr;
    uint64_t deg = g->row_ptr[c + 1] - g->row_ptr[c];
    return deg ? lr / (float)deg : lr;
}

/* One walk, SkipGram with negative sampling, strictly sequential ("Semantic S"):
 * per centre the central row is copied to u, every (context, negatives...) sample updates the
 * contextual table immediately, the accumulated gradient is added to the central row at the end
 * of the centre.  neg_override (optional) = explicit negatives [L][2w][k] for this walk. */
void o_sgns_walk(const o_graph *g, const o_train_params *tp, const o_step_io *io,
                 const uint32_t *walk, const uint32_t *wrow, uint32_t L, uint64_t wkey, float lr,
                 const uint32_t *neg_override, float *u, float *gacc) {
    float *central = io->central, *contextual = io->contextual;
    float *negative = io->negative ? io->negative : contextual;
    uint32_t d = tp->d, ld = tp->ld, w = tp->window, k = tp->k, md = min_dist_of(tp);
    uint32_t Le = effective_len(walk, L);
    uint64_t nkey = wkey ^ O_TAG_NEG;
    for (uint32_t i = 0; i < Le; ++i) {
        uint32_t c = walk[i];
        if (!keep_centre(g, tp, wkey, i, c)) continue;
        if (context_count(i, Le, w, md) == 0) continue;
        float lrc = centre_lr(g, tp, lr, c);
        float *crow = central + (uint64_t)wrow[i] * ld;
        memcpy(u, crow, d * sizeof(float));
        memset(gacc, 0, d * sizeof(float));
        uint32_t lo = i > w ? i - w : 0;
        uint32_t hi = i + w < Le - 1 ? i + w : Le - 1;
        for (uint32_t j = lo; j <= hi; ++j) {
            if (!is_context(i, j, md)) continue;
            uint32_t slot = j < i ? (j + w - i) : (j + w - i - 1); /* 0 .. 2w-1 */
            uint32_t ctx = walk[j];
#ifdef O_FAST
            /* Tuned build only (the timed CPU baseline, see the Makefile): hints, results
             * unchanged -- the rows of the NEXT context slot are requested while this slot is
             * processed, so that a dozen DRAM accesses are in flight instead of one. */
            {
                uint32_t jn = j + 1 == i ? j + 2 : j + 1;
                if (jn <= hi && !neg_override) {
                    uint32_t slot_n = jn < i ? (jn + w - i) : (jn + w - i - 1);
                    const char *pv = (const char *)(contextual + (uint64_t)wrow[jn] * ld);
                    for (uint32_t b = 0; b < d * sizeof(float); b += 64) __builtin_prefetch(pv + b, 1, 1);
                    for (uint32_t s = 1; s <= k; ++s) {
                        uint64_t q = ((uint64_t)i * 2 * w + slot_n) * k + (s - 1);
                        uint32_t row = draw_negative(g, tp, io, nkey, q);
                        pv = (const char *)(negative + (uint64_t)row * ld);
                        for (uint32_t b = 0; b < d * sizeof(float); b += 64) __builtin_prefetch(pv + b, 1, 1);
                    }
                }
            }
#endif
            for (uint32_t s = 0; s <= k; ++s) {
                float *v;
                float label;
                if (s == 0) {
                    v = contextual + (uint64_t)wrow[j] * ld;
                    label = 1.0f;
                } else {
                    uint32_t qslot = (tp->flags & O_FLAG_SHARED_NEGATIVES) ? 0 : slot;
                    uint64_t q = ((uint64_t)i * 2 * w + qslot) * k + (s - 1);
                    uint32_t row = neg_override ? neg_override[q] : draw_negative(g, tp, io, nkey, q);
                    uint32_t gid = neg_global_id(io, row);
                    label = 0.0f;
                    if (gid == c || gid == ctx) continue;
                    v = negative + (uint64_t)row * ld;
                }
                float dot = 0.0f;
                for (uint32_t x = 0; x < d; ++x) dot += u[x] * v[x];
                if ((tp->flags & O_FLAG_SKIP_CLIPPED) && (dot > tp->clip || dot < -tp->clip))
                    continue;
                if (dot > tp->clip) dot = tp->clip;
                if (dot < -tp->clip) dot = -tp->clip;
                float var = (label - sigmoidf(dot)) * lrc;
                for (uint32_t x = 0; x < d; ++x) {
                    gacc[x] += var * v[x];
                    v[x] += var * u[x];
                }
            }
        }
        for (uint32_t x = 0; x < d; ++x) crow[x] += gacc[x];
    }
}

/* One walk, CBOW with negative sampling.  h = mean of the *contextual* rows of the window (the
 * input side, cbow.py:37-42); the centre and k negatives are scored against the *central* table
 * (the output side); the input gradient / C is added to every context row at the end. */
void o_cbow_walk(const o_graph *g, const o_train_params *tp, const o_step_io *io,
                 const uint32_t *walk, const uint32_t *wrow, uint32_t L, uint64_t wkey, float lr,
                 const uint32_t *neg_override, float *h, float *gacc) {
    float *central = io->central, *contextual = io->contextual;
    float *negative = io->negative ? io->negative : central;
    uint32_t d = tp->d, ld = tp->ld, w = tp->window, k = tp->k, md = min_dist_of(tp);
    uint32_t Le = effective_len(walk, L);
    uint64_t nkey = wkey ^ O_TAG_NEG;
    for (uint32_t i = 0; i < Le; ++i) {
        uint32_t c = walk[i];
        if (!keep_centre(g, tp, wkey, i, c)) continue;
        float lrc = centre_lr(g, tp, lr, c);
        uint32_t lo = i > w ? i - w : 0;
        uint32_t hi = i + w < Le - 1 ? i + w : Le - 1;
        uint32_t C = context_count(i, Le, w, md);
        if (C == 0) continue;
        memset(h, 0, d * sizeof(float));
        memset(gacc, 0, d * sizeof(float));
        for (uint32_t j = lo; j <= hi; ++j) {
            if (!is_context(i, j, md)) continue;
            const float *row = contextual + (uint64_t)wrow[j] * ld;
            for (uint32_t x = 0; x < d; ++x) h[x] += row[x];
        }
        float invC = 1.0f / (float)C;
        for (uint32_t x = 0; x < d; ++x) h[x] *= invC;
        for (uint32_t s = 0; s <= k; ++s) {
            float *v;
            float label;
            if (s == 0) {
                v = central + (uint64_t)wrow[i] * ld;
                label = 1.0f;
            } else {
                uint64_t q = (uint64_t)i * k + (s - 1);
                uint32_t row = neg_override ? neg_override[q] : draw_negative(g, tp, io, nkey, q);
                label = 0.0f;
                if (neg_global_id(io, row) == c) continue;
                v = negative + (uint64_t)row * ld;
            }
            float dot = 0.0f;
            for (uint32_t x = 0; x < d; ++x) dot += h[x] * v[x];
            if (dot > tp->clip) dot = tp->clip;
            if (dot < -tp->clip) dot = -tp->clip;
            float var = (label - sigmoidf(dot)) * lrc;
            for (uint32_t x = 0; x < d; ++x) {
                gacc[x] += var * v[x];
                v[x] += var * h[x];
            }
        }
        for (uint32_t j = lo; j <= hi; ++j) {
            if (!is_context(i, j, md)) continue;
            float *row = contextual + (uint64_t)wrow[j] * ld;
            for (uint32_t x = 0; x < d; ++x) row[x] += gacc[x] * invC;
        }
    }
}

/* Train on explicit walks [n_walks][L]; walk b has id first_walk + b in (seed, epoch).
 * threads <= 1: strictly sequential in walk order (the parity oracle).
 * threads  > 1: OpenMP over walks, unsynchronised updates (Hogwild) -- the CPU baseline. */
void o_train_walks_ex(const o_graph *g, const o_train_params *tp, const o_step_io *io,
                      uint64_t n_walks, uint32_t L, uint64_t seed, uint64_t epoch,
                      uint64_t first_walk, float lr, int threads) {
    uint64_t ekey = o_epoch_key(seed, epoch);
    uint64_t per_walk_neg =
        tp->model == 0 ? (uint64_t)L * 2 * tp->window * tp->k : (uint64_t)L * tp->k;
#pragma omp parallel num_threads(threads > 1 ? threads : 1)
    {
        float *u = (float *)malloc(sizeof(float) * tp->d);
        float *gacc = (float *)malloc(sizeof(float) * tp->d);
#pragma omp for schedule(dynamic, 16)
        for (uint64_t b = 0; b < n_walks; ++b) {
            uint64_t wkey = o_draw(ekey, first_walk + b);
            const uint32_t *ov = io->neg_override ? io->neg_override + b * per_walk_neg : NULL;
            const uint32_t *walk = io->walks + b * L;
            const uint32_t *wrow = io->walk_rows ? io->walk_rows + b * L : walk;
            if (tp->model == 0)
                o_sgns_walk(g, tp, io, walk, wrow, L, wkey, lr, ov, u, gacc);
            else
                o_cbow_walk(g, tp, io, walk, wrow, L, wkey, lr, ov, u, gacc);
        }
        free(u);
        free(gacc);
    }
}

void o_train_walks(const o_graph *g, const o_train_params *tp, const uint32_t *walks,
                   uint64_t n_walks, uint32_t L, uint64_t seed, uint64_t epoch,
                   uint64_t first_walk, float lr, float *central, float *contextual,
                   const uint32_t *neg_override, int threads) {
    o_step_io io;
    memset(&io, 0, sizeof(io));
    io.walks = walks;
    io.central = central;
    io.contextual = contextual;
    io.neg_override = neg_override;
    o_train_walks_ex(g, tp, &io, n_walks, L, seed, epoch, first_walk, lr, threads);
}

/* The units a batch of walks amounts to (what the device's counters report: gn2v_stats):
 * out[0] += (centre, context) pairs of the centres that are trained, out[1] += walk steps
 * (transitions made), out[2] += trained centres (those with at least one context). */
void o_count_units(const o_graph *g, const o_train_params *tp, const uint32_t *walks,
                   uint64_t n_walks, uint32_t L, uint64_t seed, uint64_t epoch,
                   uint64_t first_walk, uint64_t out[3]) {
    uint64_t ekey = o_epoch_key(seed, epoch);
    for (uint64_t b = 0; b < n_walks; ++b) {
        uint32_t Le = effective_len(walks + b * L, L);
        uint64_t wkey = o_draw(ekey, first_walk + b);
        if (Le) out[1] += Le - 1;
        for (uint32_t i = 0; i < Le; ++i) {
            if (!keep_centre(g, tp, wkey, i, walks[b * L + i])) continue;
            uint32_t n = context_count(i, Le, tp->window, min_dist_of(tp));
            out[0] += n;
            out[2] += n != 0;
        }
    }
}

/* Full fit: init both tables, then per epoch generate all walks and train on them in order.
 * Returns the number of (centre, context) training pairs processed. */
uint64_t o_fit(const o_graph *g, const o_walk_params *wp, const o_train_params *tp,
               const uint32_t *sources, uint64_t n_sources, uint64_t seed, float *central,
               float *contextual, int threads) {
    o_init_table(central, g->n_nodes, tp->d, tp->ld, seed, 0, tp->init_scale);
    o_init_table(contextual, g->n_nodes, tp->d, tp->ld, seed, 1, tp->init_scale);
    uint64_t n_walks = n_sources * wp->iterations;
    uint32_t L = wp->walk_length;
    uint32_t *walks = (uint32_t *)malloc(sizeof(uint32_t) * n_walks * L);
    float lr = tp->lr;
    uint64_t pairs = 0;
    for (uint32_t e = 0; e < tp->epochs; ++e) {
        o_walks(g, wp, sources, n_sources, seed, e, 0, n_walks, walks);
        o_train_walks(g, tp, walks, n_walks, L, seed, e, 0, lr, central, contextual, NULL,
                      threads);
        uint64_t units[3] = {0, 0, 0}; /* pairs of the centres that were trained */
        o_count_units(g, tp, walks, n_walks, L, seed, e, 0, units);
        pairs += units[0];
        lr *= tp->lr_decay;
    }
    free(walks);
    return pairs;
}

/* Node2VecSequence batch form (node2vec_sequence.py:115-128,190-203): for every walk position
 * with a full window emit words[n] = centre and contexts[n][2w] = the 2w surrounding nodes. */
uint64_t o_window_batch(const uint32_t *walks, uint64_t n_walks, uint32_t L, uint32_t w,
                        int32_t *contexts, int32_t *words) {
    uint64_t n = 0;
    for (uint64_t b = 0; b < n_walks; ++b) {
        const uint32_t *wk = walks + b * L;
        for (uint32_t i = w; i + w < L; ++i) {
            words[n] = (int32_t)wk[i];
            uint32_t s = 0;
            for (uint32_t j = i - w; j <= i + w; ++j)
                if (j != i) contexts[n * 2 * w + s++] = (int32_t)wk[j];
            ++n;
        }
    }
    return n;
}

/* All (centre, context) pairs of the walks in walk / position / slot order (window trimmed at the
 * borders, contexts at distance [min_dist, w]); returns the number of pairs written. */
uint64_t o_walk_pairs(const uint32_t *walks, uint64_t n_walks, uint32_t L, uint32_t w,
                      uint32_t min_dist, uint32_t *pairs) {
    uint64_t n = 0;
    uint32_t md = min_dist ? min_dist : 1;
    for (uint64_t b = 0; b < n_walks; ++b) {
        const uint32_t *wk = walks + b * L;
        uint32_t Le = effective_len(wk, L);
        for (uint32_t i = 0; i < Le; ++i)
            for (uint32_t slot = 0; slot < 2 * w; ++slot) {
                int64_t j = slot < w ? (int64_t)i - w + slot : (int64_t)i + 1 + (slot - w);
                if (j < 0 || j >= (int64_t)Le) continue;
                if (!is_context(i, (uint32_t)j, md)) continue;
                pairs[2 * n] = wk[i];
                pairs[2 * n + 1] = wk[j];
                ++n;
            }
    }
    return n;
}

/* ---------------------------------------------------------------- GloVe
 * The third model of the reference's walk-based table (embedders/ensmallen_embedders/node2vec.py
 * :16-26 "Node2Vec GloVe": models.GloVe; wrapper kwargs node2vec_glove.py:8-30: alpha = 0.75,
 * epochs = 100, walk_length = 512, iterations = 1, window_size = 5, learning_rate = 0.05,
 * learning_rate_decay = 0.9).  PARITY UNPINNED like the rest of this file: the arithmetic lives in
 * the ensmallen wheel, so this restates the published algorithm (Pennington, Socher, Manning,
 * "GloVe: Global Vectors for Word Representation", EMNLP 2014) under the wrapper's kwargs:
 *   X_ij   = sum over co-occurrences of centre i and context j inside the window of 1 / distance
 *            (symmetric window, trimmed at the walk borders), normalised by the largest entry
 *            (the wrapper has no x_max: the weighting function saturates at the maximum);
 *   loss   = sum_ij f(X_ij) (w_i . w~_j + b_i + b~_j - log X_ij)^2 / 2,  f(x) = x^alpha;
 *   update = plain SGD (learning_rate, multiplied by learning_rate_decay per epoch) over the
 *            non-zero entries in a fixed shuffled order (see o_glove_entries); biases are
 *            internal (two tables out).
 * 1 / distance is accumulated in fixed point (2^20 / distance, rounded) so the sums are exact and
 * independent of the order in which co-occurrences are counted. */

#define O_TAG_GLOVE 0x610FE00000C00C01ULL
#define O_COOC_ONE (1u << 20)
#define O_COOC_UNUSED 0x7FFFFFFFFFFFFFFFULL

static inline uint64_t cooc_weight(uint32_t dist) { return (O_COOC_ONE + dist / 2) / dist; }

/* per slot [walk][position][2w]: key = centre << 32 | context and the fixed-point weight, or
 * (O_COOC_UNUSED, 0) */
void o_cooc_slots(const uint32_t *walks, uint64_t n_walks, uint32_t L, uint32_t w,
                  uint32_t min_dist, uint64_t *keys, uint64_t *weights) {
    uint32_t md = min_dist ? min_dist : 1;
    for (uint64_t b = 0; b < n_walks; ++b) {
        const uint32_t *wk = walks + b * L;
        uint32_t Le = effective_len(wk, L);
        for (uint32_t i = 0; i < L; ++i)
            for (uint32_t slot = 0; slot < 2 * w; ++slot) {
                uint64_t t = (b * L + i) * 2 * w + slot;
                int64_t j = slot < w ? (int64_t)i - w + slot : (int64_t)i + 1 + (slot - w);
                keys[t] = O_COOC_UNUSED;
                weights[t] = 0;
                if (i >= Le || j < 0 || j >= (int64_t)Le) continue;
                if (!is_context(i, (uint32_t)j, md)) continue;
                uint32_t dist = (uint32_t)(j > (int64_t)i ? j - i : i - j);
                keys[t] = ((uint64_t)wk[i] << 32) | wk[j];
                weights[t] = cooc_weight(dist);
            }
    }
}

typedef struct {
    uint64_t key, val;
} kv64;

static int cmp_kv_key(const void *a, const void *b) {
    uint64_t x = ((const kv64 *)a)->key, y = ((const kv64 *)b)->key;
    return x < y ? -1 : x > y;
}

/* sum the weights of equal keys: keys ascending (unsigned), unused slots dropped; returns the
 * number of distinct keys written to the front of keys / weights */
uint64_t o_cooc_reduce(uint64_t *keys, uint64_t *weights, uint64_t n_slots) {
    kv64 *kv = malloc(sizeof(kv64) * (n_slots ? n_slots : 1));
    uint64_t m = 0;
    for (uint64_t t = 0; t < n_slots; ++t)
        if (keys[t] != O_COOC_UNUSED) {
            kv[m].key = keys[t];
            kv[m++].val = weights[t];
        }
    qsort(kv, m, sizeof(kv64), cmp_kv_key);
    uint64_t n = 0;
    for (uint64_t t = 0; t < m; ++t) {
        if (n && keys[n - 1] == kv[t].key) {
            weights[n - 1] += kv[t].val;
        } else {
            keys[n] = kv[t].key;
            weights[n++] = kv[t].val;
        }
    }
    free(kv);
    return n;
}

/* Training order of the non-zero entries.  Entries are grouped by their central row, every row's
 * entries (in ascending mix64(key ^ salt)) are cut into records of O_GLOVE_RECORD, and the records
 * are shuffled (ascending draw(mix64(row ^ salt), record index in the row)).  A record occupies
 * O_GLOVE_RECORD consecutive slots of the output arrays; the unused slots of a row's last record
 * hold col = O_SENTINEL (skipped by the step).  The engine trains a record per wavefront with the
 * central row in registers; the arithmetic per entry does not depend on this layout. */
#define O_GLOVE_RECORD 16u

typedef struct {
    uint64_t row, h, src;
} glove_ent;

static int cmp_ent(const void *a, const void *b) {
    const glove_ent *x = a, *y = b;
    if (x->row != y->row) return x->row < y->row ? -1 : 1;
    if (x->h != y->h) return x->h < y->h ? -1 : 1;
    return x->src < y->src ? -1 : x->src > y->src;
}

/* number of records the entries need (keys ascending, i.e. grouped by row) */
uint64_t o_glove_record_count(const uint64_t *keys, uint64_t n) {
    uint64_t records = 0, run = 0;
    for (uint64_t t = 0; t < n; ++t) {
        ++run;
        if (t + 1 == n || (keys[t + 1] >> 32) != (keys[t] >> 32)) {
            records += (run + O_GLOVE_RECORD - 1) / O_GLOVE_RECORD;
            run = 0;
        }
    }
    return records;
}

/* rows / cols / log X / f(X) of every slot, X = count / max count; arrays of
 * o_glove_record_count(keys, n) * O_GLOVE_RECORD slots */
void o_glove_entries(const uint64_t *keys, const uint64_t *counts, uint64_t n, uint64_t seed,
                     float alpha, uint32_t *rows, uint32_t *cols, float *logx, float *fx) {
    uint64_t salt = o_mix64(seed ^ O_TAG_GLOVE), mx = 1;
    glove_ent *ent = malloc(sizeof(glove_ent) * (n ? n : 1));
    for (uint64_t t = 0; t < n; ++t) {
        ent[t].row = keys[t] >> 32;
        ent[t].h = o_mix64(keys[t] ^ salt);
        ent[t].src = t;
        if (counts[t] > mx) mx = counts[t];
    }
    qsort(ent, n, sizeof(glove_ent), cmp_ent);
    /* records: (shuffle hash, row, index in row, first entry, length) */
    uint64_t n_rec = o_glove_record_count(keys, n);
    glove_ent *rec = malloc(sizeof(glove_ent) * (n_rec ? n_rec : 1));
    uint64_t *first = malloc(sizeof(uint64_t) * (n_rec ? n_rec : 1));
    uint64_t r = 0;
    for (uint64_t t = 0; t < n;) {
        uint64_t e = t;
        while (e < n && ent[e].row == ent[t].row) ++e;
        uint64_t rkey = o_mix64(ent[t].row ^ salt);
        for (uint64_t q = 0; t + q * O_GLOVE_RECORD < e; ++q) {
            rec[r].h = o_draw(rkey, q);
            rec[r].row = ent[t].row;
            rec[r].src = r; /* = (row, q) order: the tie break */
            first[r] = t + q * O_GLOVE_RECORD;
            ++r;
        }
        t = e;
    }
    /* cmp_ent orders by (row, h, src); the shuffle wants (h, row, q): swap the roles */
    for (uint64_t i = 0; i < n_rec; ++i) {
        uint64_t row = rec[i].row;
        rec[i].row = rec[i].h;
        rec[i].h = row;
    }
    qsort(rec, n_rec, sizeof(glove_ent), cmp_ent);
    for (uint64_t i = 0; i < n_rec; ++i) {
        uint64_t t0 = first[rec[i].src];
        for (uint32_t q = 0; q < O_GLOVE_RECORD; ++q) {
            uint64_t slot = i * O_GLOVE_RECORD + q, t = t0 + q;
            rows[slot] = (uint32_t)rec[i].h;
            if (t < n && ent[t].row == rec[i].h) {
                uint64_t src = ent[t].src;
                float x = (float)((double)counts[src] / (double)mx);
                cols[slot] = (uint32_t)(keys[src] & 0xFFFFFFFFu);
                logx[slot] = (float)log((double)x);
                fx[slot] = (float)pow((double)x, (double)alpha);
            } else {
                cols[slot] = O_SENTINEL;
                logx[slot] = 0.0f;
                fx[slot] = 0.0f;
            }
        }
    }
    free(ent);
    free(rec);
    free(first);
}

/* sequential SGD over entries [0, n): for each entry the gradient scale g = f * (u.v + b_i + b~_j
 * - log X); u -= lr g v, v -= lr g u(old), both biases -= lr g; non-finite g skips the entry */
void o_glove_step(const uint32_t *rows, const uint32_t *cols, const float *logx, const float *fx,
                  uint64_t n, float *central, float *contextual, float *bias_c, float *bias_x,
                  uint32_t d, uint32_t ld, float lr) {
    for (uint64_t e = 0; e < n; ++e) {
        if (cols[e] == O_SENTINEL) continue; /* padding slot of a record */
        float *u = central + (uint64_t)rows[e] * ld, *v = contextual + (uint64_t)cols[e] * ld;
        float dot = 0.0f;
        for (uint32_t c = 0; c < d; ++c) dot += u[c] * v[c];
        float diff = dot + bias_c[rows[e]] + bias_x[cols[e]] - logx[e];
        float g = fx[e] * diff;
        if (!isfinite(g)) continue;
        float s = -lr * g;
        for (uint32_t c = 0; c < d; ++c) {
            float uo = u[c];
            u[c] = uo + s * v[c];
            v[c] = v[c] + s * uo;
        }
        bias_c[rows[e]] += s;
        bias_x[cols[e]] += s;
    }
}

/* The engine's record schedule, restated for the tests of its parallel kernel: the slots come in
 * records of O_GLOVE_RECORD with a common row; the four entries of a round (slots 4r .. 4r+3) all
 * see the row and its bias as they were at the start of the round, then their summed
 * contributions are applied (o_glove_step is the limit of one entry per round). */
void o_glove_step_rounds(const uint32_t *rows, const uint32_t *cols, const float *logx,
                         const float *fx, uint64_t n, float *central, float *contextual,
                         float *bias_c, float *bias_x, uint32_t d, uint32_t ld, float lr) {
    float *du = (float *)malloc(sizeof(float) * (d ? d : 1));
    for (uint64_t base = 0; base < n; base += 4) {
        uint32_t i = rows[base];
        float *u = central + (uint64_t)i * ld;
        float db = 0.0f;
        memset(du, 0, sizeof(float) * d);
        for (uint64_t e = base; e < base + 4 && e < n; ++e) {
            if (cols[e] == O_SENTINEL) continue;
            float *v = contextual + (uint64_t)cols[e] * ld;
            float dot = 0.0f;
            for (uint32_t c = 0; c < d; ++c) dot += u[c] * v[c];
            float g = fx[e] * (((dot + bias_c[i]) + bias_x[cols[e]]) - logx[e]);
            if (!isfinite(g)) continue;
            float s = -lr * g;
            for (uint32_t c = 0; c < d; ++c) {
                du[c] += s * v[c];
                v[c] += s * u[c];
            }
            bias_x[cols[e]] += s;
            db += s;
        }
        for (uint32_t c = 0; c < d; ++c) u[c] += du[c];
        bias_c[i] += db;
    }
    free(du);
}

/* GloVe loss of the entries (double accumulation; tests) */
double o_glove_loss(const uint32_t *rows, const uint32_t *cols, const float *logx, const float *fx,
                    uint64_t n, const float *central, const float *contextual,
                    const float *bias_c, const float *bias_x, uint32_t d, uint32_t ld) {
    double loss = 0.0;
    for (uint64_t e = 0; e < n; ++e) {
        if (cols[e] == O_SENTINEL) continue;
        const float *u = central + (uint64_t)rows[e] * ld, *v = contextual + (uint64_t)cols[e] * ld;
        double dot = 0.0;
        for (uint32_t c = 0; c < d; ++c) dot += (double)u[c] * v[c];
        double diff = dot + bias_c[rows[e]] + bias_x[cols[e]] - logx[e];
        loss += 0.5 * fx[e] * diff * diff;
    }
    return loss;
}

/* ---------------------------------------------------------------- block-partitioned SkipGram
 * Sequential restatement of the engine's multi-GPU schedule (embiggen_amd/csrc/block_kernels.h;
 * the reference has no counterpart: its call, node2vec.py:99, runs inside one process).  Nodes
 * are striped over `world` ranks (centre c: rank c % world, row c / world) and over `parts`
 * context parts (x: part x % parts, row x / parts), the rows of a part once more over `slices`
 * (row % slices); cell = part * slices + slice.  A rank extracts, from the walks of all ranks of
 * a round, the pairs whose centre it owns (and whose context lies in a given group of parts) as
 * ONE 64-bit word each,
 *     cell << (row_bits + ctx_bits) | centre row << ctx_bits | hot << (ctx_bits - 1) | context row
 * (context row counted inside its cell: (x / parts) / slices) in walk / position / slot order,
 * sorts them stably on the bits above ctx_bits, and trains one part at a time:
 * per cell, implicit records of `record` consecutive sorted pairs visited in the stride order
 * rec(t) = t * A mod R (A ~ R / golden ratio, coprime with R); inside a record every run of equal
 * centre (at most O_MAX_RUN pairs) is one "centre" of Semantic S (copy of the central row, samples [context, k negatives]
 * applied one after the other, gradient added at the end of the run).  Negative n of the pair at
 * position p of its cell: a row of the cell, degree-proportional through the cell's alias table
 * (or uniform), picked by draw(cell_key, p * k + n), cell_key = draw(draw(mix64(epoch_key ^ TAG_BLOCK), block_id), cell);
 * skipped when it is the context or the centre itself. */

#define O_TAG_BLOCK 0xB10C5EED0B10C5EDULL
#define O_NEG_ATTEMPTS 8u /* draws of a cell-local negative before it is given up */
#define O_MAX_RUN 16u       /* pairs trained against one copy of the central row, at most */

typedef struct {
    uint32_t world, rank, parts, slices;
    uint32_t walk_length, window, min_dist, record;
    uint32_t row_bits;
    uint32_t flags;
    uint32_t hot_rows;  /* rows a cell flags as "hot", 0 = none (see o_block_alias) */
    uint32_t hot_flush; /* device only: how often a hot row's pending sum reaches the row */
    uint32_t key_bits;       /* bits of a pair word in use */
    uint32_t ctx_bits;       /* low bits: context row inside its cell + the hot flag on top */
} o_block_plan;

static inline uint64_t stripe_count(uint64_t n, uint64_t first, uint64_t stride) {
    return n > first ? (n - first + stride - 1) / stride : 0;
}

static uint32_t bits_for(uint64_t n) { /* smallest b with 2^b >= n */
    uint32_t b = 0;
    while ((1ULL << b) < n) ++b;
    return b;
}

uint32_t o_block_row_bits(uint64_t n_nodes, uint32_t world) {
    return bits_for((n_nodes + world - 1) / world);
}

/* rows of the largest cell (part 0, slice 0), plus one bit for the hot flag */
uint32_t o_block_ctx_bits(uint64_t n_nodes, uint32_t parts, uint32_t slices) {
    return bits_for(stripe_count(stripe_count(n_nodes, 0, parts), 0, slices)) + 1;
}

uint32_t o_block_cell_bits(uint32_t parts, uint32_t slices) {
    return bits_for((uint64_t)parts * slices);
}

/* Placement of a round: WHERE a node's contextual row is trained changes from round to round, so
 * that over a fit the negatives a context meets -- drawn inside its cell -- range over the graph
 * (node2vec_skipgram.py:101-102: "proportionally to their degree", over the graph), not over
 * one fixed set of cell-mates.  A seeded permutation of the node ids that keeps every node in
 * its residue class modulo `classes` (classes = parts when the parts travel between ranks: a
 * row never changes its part; 1 on one GPU: the whole graph is shuffled): the nodes of class c
 * in the order of (hash(round key, x), x) receive the placed ids c, c + classes, c + 2 classes,
 * ...; place[x] = x' and inv[x'] = x.  Everything downstream (part = x' % parts, row = x' /
 * parts, slice = row % slices, the pair words, the alias tables) works on placed ids; the
 * tables stay where they are and are addressed through inv. */
#define O_TAG_PLACE 0x91ACE5EED5EED5EDULL

typedef struct {
    uint64_t key;
    uint32_t x;
} place_kv;

static int cmp_place_kv(const void *a, const void *b) {
    const place_kv *p = (const place_kv *)a, *q = (const place_kv *)b;
    if (p->key != q->key) return p->key < q->key ? -1 : 1;
    return p->x < q->x ? -1 : (p->x > q->x ? 1 : 0);
}

void o_block_placement(uint64_t n_nodes, uint32_t classes, uint64_t seed, uint64_t round_id,
                       uint32_t *place, uint32_t *inv) {
    uint64_t pkey = o_draw(o_mix64(seed ^ O_TAG_PLACE), round_id);
    place_kv *kv = (place_kv *)malloc(sizeof(place_kv) * (n_nodes ? n_nodes : 1));
    if (classes < 1) classes = 1;
    for (uint64_t x = 0; x < n_nodes; ++x) {
        kv[x].key = ((uint64_t)(x % classes) << 40) | (o_draw(pkey, x) >> 24);
        kv[x].x = (uint32_t)x;
    }
    qsort(kv, n_nodes, sizeof(place_kv), cmp_place_kv);
    /* class c starts at sorted position c * (n / classes) + min(c, n % classes) */
    for (uint64_t j = 0; j < n_nodes; ++j) {
        uint64_t c = kv[j].x % classes;
        uint64_t start = c * (n_nodes / classes) + (c < n_nodes % classes ? c : n_nodes % classes);
        uint32_t xp = (uint32_t)(c + (uint64_t)classes * (j - start));
        place[kv[j].x] = xp;
        inv[xp] = kv[j].x;
    }
    free(kv);
}

/* pair words of this rank in walk / position / slot order, contexts in the parts part_lo,
 * part_lo + 1, ... (part_n of them, cyclic; 0, 0 = every part); words may be NULL (count only);
 * place (or NULL: identity): the round's placement of the CONTEXT nodes (the centres keep their
 * ids: the central table is not placed) */
uint64_t o_block_extract(const o_graph *g, const o_block_plan *p, const uint32_t *walks,
                         uint64_t n_walks, uint64_t seed, uint64_t epoch, uint64_t first_walk,
                         uint32_t part_lo, uint32_t part_n, const uint32_t *hub_bits,
                         uint64_t *words, const uint32_t *place) {
    uint64_t n = 0, ekey = o_epoch_key(seed, epoch);
    uint32_t L = p->walk_length, w = p->window, md = p->min_dist ? p->min_dist : 1;
    o_train_params tp;
    memset(&tp, 0, sizeof(tp));
    tp.flags = p->flags & O_FLAG_DOWNSAMPLE;
    if (part_lo == 0 && part_n == 0) part_n = p->parts;
    for (uint64_t b = 0; b < n_walks; ++b) {
        const uint32_t *wk = walks + b * L;
        uint32_t Le = effective_len(wk, L);
        uint64_t wkey = o_draw(ekey, first_walk + b);
        for (uint32_t i = 0; i < Le; ++i) {
            uint32_t c = wk[i];
            if (c % p->world != p->rank) continue;
            if (!keep_centre(g, &tp, wkey, i, c)) continue;
            for (uint32_t slot = 0; slot < 2 * w; ++slot) {
                int64_t j = slot < w ? (int64_t)i - w + slot : (int64_t)i + 1 + (slot - w);
                if (j < 0 || j >= (int64_t)Le) continue;
                if (!is_context(i, (uint32_t)j, md)) continue;
                uint32_t x = wk[j];
                uint32_t xp = place ? place[x] : x;
                uint32_t row = xp / p->parts, part = xp % p->parts;
                if ((part + p->parts - part_lo) % p->parts >= part_n) continue;
                uint32_t cell = part * p->slices + row % p->slices;
                if (words) {
                    /* hot context rows carry the flag (updated with atomics on the device) */
                    uint64_t hot = (hub_bits && ((hub_bits[x >> 5] >> (x & 31)) & 1u)) ? 1u : 0u;
                    words[n] = (((((uint64_t)cell << p->row_bits) | (c / p->world))) << p->ctx_bits) |
                               (hot << (p->ctx_bits - 1)) | (row / p->slices);
                }
                ++n;
            }
        }
    }
    return n;
}

typedef struct {
    uint64_t word;
    uint64_t idx;
} block_kv;

static uint32_t g_sort_shift; /* qsort has no context argument */

static int cmp_block_kv(const void *a, const void *b) {
    const block_kv *x = (const block_kv *)a, *y = (const block_kv *)b;
    uint64_t kx = x->word >> g_sort_shift, ky = y->word >> g_sort_shift;
    if (kx != ky) return kx < ky ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* stable sort on the bits above ctx_bits (cell, centre row) */
void o_block_sort(uint64_t *words, uint64_t n, uint32_t ctx_bits) {
    block_kv *kv = (block_kv *)malloc(sizeof(block_kv) * (n ? n : 1));
    for (uint64_t i = 0; i < n; ++i) {
        kv[i].word = words[i];
        kv[i].idx = i;
    }
    g_sort_shift = ctx_bits;
    qsort(kv, n, sizeof(block_kv), cmp_block_kv);
    for (uint64_t i = 0; i < n; ++i) words[i] = kv[i].word;
    free(kv);
}

/* cell_offsets[c] = first sorted position whose cell is >= c, c = 0 .. cells */
void o_block_cell_offsets(const uint64_t *words, uint64_t n, uint32_t cell_shift, uint32_t cells,
                          uint64_t *offsets) {
    uint64_t p = 0;
    for (uint32_t c = 0; c <= cells; ++c) {
        while (p < n && (words[p] >> cell_shift) < c) ++p;
        offsets[c] = c == cells ? n : p;
    }
}

/* floor(w * 2^32 / D) for w < D < 2^48 */
static inline uint64_t scaled_threshold(uint64_t w, uint64_t D) {
    uint64_t q1 = (w << 16) / D, r1 = (w << 16) % D;
    return (q1 << 16) | ((r1 << 16) / D);
}

/* Degree-proportional negatives inside a cell: one Walker alias table per cell (Vose's
 * construction in integers; weights = in-degrees = how often a node is the endpoint of a uniform
 * random directed edge, node2vec_skipgram.py:101-102).  table[cell_rows[c] + i] = threshold on a
 * 2^32 scale | alias row << 32 for row i of cell c.  A draw r picks i = mulhi(r, n) and keeps it
 * when (u32) r < threshold, else takes the alias. */
/* Hot rows of a cell: its `hot_rows` rows of highest in-degree (>= 1; ties: the lower row first;
 * at most O_HOT_MAX).  They are flagged: bit 0 of an entry = the row itself (the threshold keeps
 * its upper 31 bits), bit 63 = its alias row, hub_bits = one bit per node id; hot_list[cell][s] =
 * row inside the cell of slot s in decreasing order of in-degree (0xFFFFFFFF beyond the cell's
 * count), hot_slot[cell_rows[cell] + row] = slot (0xFF: not hot).  The flags steer how the device
 * accumulates the updates of such rows (LDS sums handed over with atomics); this file's
 * arithmetic ignores them. */
#define O_HOT_MAX 192u
void o_block_alias(const o_graph *g, uint32_t parts, uint32_t slices, uint32_t hot_rows,
                   uint64_t *table, uint64_t *cell_rows, uint32_t *hub_bits, uint32_t *hot_list,
                   uint8_t *hot_slot, const uint32_t *inv) {
    uint32_t *indeg = (uint32_t *)calloc(g->n_nodes, sizeof(uint32_t));
    for (uint64_t e = 0; e < g->n_edges; ++e) indeg[g->col_idx[e]]++;
    memset(hub_bits, 0, sizeof(uint32_t) * ((g->n_nodes + 31) / 32));
    memset(hot_slot, 0xFF, g->n_nodes);
    if (hot_rows > O_HOT_MAX) hot_rows = O_HOT_MAX;
    uint64_t run = 0;
    for (uint32_t p = 0; p < parts; ++p) {
        uint64_t part_rows = stripe_count(g->n_nodes, p, parts);
        for (uint32_t sl = 0; sl < slices; ++sl) {
            cell_rows[p * slices + sl] = run;
            run += stripe_count(part_rows, sl, slices);
        }
    }
    cell_rows[parts * slices] = run;
    uint64_t *w = (uint64_t *)malloc(sizeof(uint64_t) * (g->n_nodes ? g->n_nodes : 1));
    uint32_t *st = (uint32_t *)malloc(sizeof(uint32_t) * (g->n_nodes ? g->n_nodes : 1));
    for (uint32_t cell = 0; cell < parts * slices; ++cell) {
        uint32_t part = cell / slices, slice = cell % slices;
        uint64_t lo = cell_rows[cell], n = cell_rows[cell + 1] - lo, D = 0;
        uint64_t *t = table + lo;
        uint32_t *hl = hot_list + (size_t)cell * O_HOT_MAX, n_hot = 0;
        for (uint32_t s = 0; s < O_HOT_MAX; ++s) hl[s] = 0xFFFFFFFFu;
        if (n == 0) continue;
/* the node whose placed id is row i of this cell (inv: the round's placement, or identity) */
#define O_PLACED(i) ((slice + (uint64_t)slices * (i)) * parts + part)
#define O_NODE_OF(i) (inv ? (uint64_t)inv[O_PLACED(i)] : O_PLACED(i))
        for (uint64_t i = 0; i < n; ++i) {
            uint32_t d = indeg[O_NODE_OF(i)];
            D += d;
            /* the hot_rows highest in-degrees by insertion: a later row only displaces a
             * strictly smaller in-degree */
            if (d != 0 && hot_rows != 0 &&
                (n_hot < hot_rows || d > indeg[O_NODE_OF(hl[n_hot - 1])])) {
                uint32_t pos = n_hot < hot_rows ? n_hot++ : n_hot - 1;
                while (pos > 0 && indeg[O_NODE_OF(hl[pos - 1])] < d) {
                    hl[pos] = hl[pos - 1];
                    --pos;
                }
                hl[pos] = (uint32_t)i;
            }
        }
        for (uint32_t s = 0; s < n_hot; ++s) {
            hot_slot[lo + hl[s]] = (uint8_t)s;
            hub_bits[O_NODE_OF(hl[s]) >> 5] |= 1u << (O_NODE_OF(hl[s]) & 31);
        }
        uint64_t n_small = 0, n_large = 0; /* small stack from st[0], large from st[n - 1] */
#define O_HOT(i) ((uint64_t)(hot_slot[lo + (i)] != 0xFF))
        for (uint64_t i = 0; i < n; ++i) {
            uint64_t p = (uint64_t)indeg[O_NODE_OF(i)] * n;
            w[i] = p;
            if (D == 0 || p >= D)
                st[n - 1 - n_large++] = (uint32_t)i;
            else
                st[n_small++] = (uint32_t)i;
        }
        while (n_small && n_large) {
            uint32_t sidx = st[--n_small], lidx = st[n - n_large];
            t[sidx] = (O_HOT(lidx) << 63) | ((uint64_t)lidx << 32) |
                      (scaled_threshold(w[sidx], D) & ~1ull) | O_HOT(sidx);
            uint64_t pl = w[lidx] + w[sidx] - D;
            w[lidx] = pl;
            if (pl < D) {
                --n_large;
                st[n_small++] = lidx;
            }
        }
        while (n_large) {
            uint32_t i = st[n - n_large--];
            t[i] = (O_HOT(i) << 63) | ((uint64_t)i << 32) | 0xFFFFFFFEull | O_HOT(i);
        }
        while (n_small) {
            uint32_t i = st[--n_small];
            t[i] = (O_HOT(i) << 63) | ((uint64_t)i << 32) | 0xFFFFFFFEull | O_HOT(i);
        }
#undef O_NODE_OF
#undef O_PLACED
#undef O_HOT
    }
    free(indeg);
    free(w);
    free(st);
}

static uint64_t gcd_u64(uint64_t a, uint64_t b) {
    while (b) {
        uint64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

uint64_t o_block_record_stride(uint64_t R) {
    if (R < 3) return 1;
    uint64_t s = (uint64_t)((double)R * 0.6180339887498949);
    if (s < 1) s = 1;
    while (gcd_u64(s, R) != 1) ++s;
    return s % R;
}

/* Negative t of a cell's stream: a row inside the part (slice + slices * local), or 0xFFFFFFFF when
 * every attempt fell on the pair's context (row xrow) or centre (node cgid).
 * A cell-local negative that falls on the context or the centre is DRAWN AGAIN (the reference
 * draws over the whole graph, node2vec_skipgram.py:101-102: there the event has probability
 * ~ degree / edges and every pair trains k negatives; in a cell of ~220 rows a hub context would
 * lose its own share of the cell's in-degree): attempt j + 1 = mix64(attempt j + golden), at most
 * O_NEG_ATTEMPTS draws, then the sample is given up (cells of one or two rows). */
static uint32_t block_negative(const o_block_plan *p, uint64_t ckey, uint64_t t, uint64_t cell_n,
                               const uint64_t *cell_alias, uint32_t slice, uint32_t part,
                               uint32_t xrow, uint64_t cgid, const uint32_t *inv) {
    uint64_t r = o_draw(ckey, t);
    for (uint32_t att = 0; att < O_NEG_ATTEMPTS; ++att, r = o_mix64(r + O_GOLDEN)) {
        uint32_t local = (uint32_t)mulhi64(r, cell_n);
        if (cell_alias) {
            uint64_t e = cell_alias[local];
            if ((uint32_t)r >= ((uint32_t)e & ~1u)) local = (uint32_t)(e >> 32) & 0x7FFFFFFFu;
        }
        uint32_t row = slice + p->slices * local;
        uint64_t xp = (uint64_t)row * p->parts + part, xn = inv ? inv[xp] : xp;
        if (row != xrow && xn != cgid) return row;
    }
    return 0xFFFFFFFFu;
}

/* test hook: the negatives o_block_step trains for the pairs of `part`, as node ids
 * (0xFFFFFFFF: given up), out[(position in the sorted words - cell_offsets[part * slices]) * k + n] */
uint64_t o_block_negatives(const o_graph *g, const o_train_params *tp, const o_block_plan *p,
                           const uint64_t *words, const uint64_t *cell_offsets,
                           const uint64_t *alias, const uint64_t *cell_rows, uint64_t block_id,
                           uint32_t part, uint64_t seed, uint64_t epoch, const uint32_t *inv,
                           uint32_t *out) {
    uint32_t k = tp->k;
    uint64_t rowmask = (1ull << p->row_bits) - 1ull;
    uint64_t ekey = o_epoch_key(seed, epoch), n = 0;
    uint64_t part_rows = stripe_count(g->n_nodes, part, p->parts);
    uint64_t base = cell_offsets[part * p->slices];
    for (uint32_t slice = 0; slice < p->slices; ++slice) {
        uint32_t cell = part * p->slices + slice;
        uint64_t lo = cell_offsets[cell], hi = cell_offsets[cell + 1];
        uint64_t ckey = o_draw(o_draw(o_mix64(ekey ^ O_TAG_BLOCK), block_id), cell);
        int use_alias = (tp->flags & O_FLAG_SCALE_FREE) && alias;
        uint64_t cell_n = stripe_count(part_rows, slice, p->slices);
        if (cell_n == 0) continue;
        for (uint64_t q = lo; q < hi; ++q, ++n) {
            uint64_t cgid = ((words[q] >> p->ctx_bits) & rowmask) * p->world + p->rank;
            uint32_t xrow = slice + p->slices * (uint32_t)(words[q] & ((1ULL << (p->ctx_bits - 1)) - 1));
            for (uint32_t s = 0; s < k; ++s) {
                uint32_t row = block_negative(p, ckey, (q - lo) * k + s, cell_n,
                                              use_alias ? alias + cell_rows[cell] : NULL, slice,
                                              part, xrow, cgid, inv);
                uint64_t xp = (uint64_t)row * p->parts + part;
                out[(q - base) * k + s] =
                    row == 0xFFFFFFFFu ? 0xFFFFFFFFu : (uint32_t)(inv ? inv[xp] : xp);
            }
        }
    }
    return n;
}

/* one part of one round, strictly sequential; returns the pairs trained */
/* inv (or NULL: identity): the round's placement; row `local` of cell (part, slice) is then the
 * contextual row of node x = inv[(slice + slices * local) * parts + part], found at row x of
 * `context` when `natural` (context = the whole table in node order), else at row x / parts
 * (context = the rows of part x % parts == part: placements that keep the classes mod parts) */
uint64_t o_block_step(const o_graph *g, const o_train_params *tp, const o_block_plan *p,
                      const uint64_t *words, const uint64_t *cell_offsets,
                      const uint64_t *alias, const uint64_t *cell_rows, float *central,
                      float *context, uint64_t block_id, uint32_t part, uint64_t seed,
                      uint64_t epoch, float lr, const uint32_t *inv, uint32_t natural) {
    uint32_t d = tp->d, ld = tp->ld, k = tp->k, C = p->record ? p->record : 16;
    uint64_t rowmask = (1ull << p->row_bits) - 1ull;
    uint64_t ekey = o_epoch_key(seed, epoch), trained = 0;
    uint64_t part_rows = stripe_count(g->n_nodes, part, p->parts);
    float *u = (float *)malloc(sizeof(float) * d), *gacc = (float *)malloc(sizeof(float) * d);
    for (uint32_t slice = 0; slice < p->slices; ++slice) {
        uint32_t cell = part * p->slices + slice;
        uint64_t lo = cell_offsets[cell], hi = cell_offsets[cell + 1];
        if (hi == lo) continue;
        uint64_t R = (hi - lo + C - 1) / C, A = o_block_record_stride(R);
        uint64_t ckey = o_draw(o_draw(o_mix64(ekey ^ O_TAG_BLOCK), block_id), cell);
        int use_alias = (tp->flags & O_FLAG_SCALE_FREE) && alias;
        uint64_t alias_lo = use_alias ? cell_rows[cell] : 0;
        uint64_t cell_n = stripe_count(part_rows, slice, p->slices);
        if (cell_n == 0) continue;
        for (uint64_t t = 0; t < R; ++t) {
            uint64_t rec = (t * A) % R, p0 = lo + rec * C;
            uint32_t n = (uint32_t)(hi - p0 < C ? hi - p0 : C);
            uint32_t r0 = 0;
            while (r0 < n) {
                uint32_t crow = (uint32_t)((words[p0 + r0] >> p->ctx_bits) & rowmask), r1 = r0 + 1;
                /* a run is at most O_MAX_RUN pairs: a longer stretch of one centre is several runs */
                while (r1 < n && r1 - r0 < O_MAX_RUN &&
                       (uint32_t)((words[p0 + r1] >> p->ctx_bits) & rowmask) == crow)
                    ++r1;
                uint64_t cgid = (uint64_t)crow * p->world + p->rank;
                float lrc = centre_lr(g, tp, lr, (uint32_t)cgid);
                float *cptr = central + (uint64_t)crow * ld;
                memcpy(u, cptr, d * sizeof(float));
                memset(gacc, 0, d * sizeof(float));
                for (uint32_t pr = r0; pr < r1; ++pr) {
                    /* row inside the part; the flag bit above the cell-local row is ignored */
                    uint32_t xrow = slice + p->slices * (uint32_t)(words[p0 + pr] &
                                                                   ((1ULL << (p->ctx_bits - 1)) - 1));
                    for (uint32_t s = 0; s <= k; ++s) {
                        uint32_t row = xrow;
                        float label = 1.0f;
                        uint64_t xp, xn;
                        if (s) {
                            row = block_negative(p, ckey, (p0 - lo + pr) * k + (s - 1), cell_n,
                                                 use_alias ? alias + alias_lo : NULL, slice, part,
                                                 xrow, cgid, inv);
                            label = 0.0f;
                            if (row == 0xFFFFFFFFu) continue;
                        }
                        /* the node behind the row, and where its contextual row lies */
                        xp = (uint64_t)row * p->parts + part;
                        xn = inv ? inv[xp] : xp;
                        float *v = context + (natural ? xn : xn / p->parts) * ld;
                        float dot = 0.0f;
                        for (uint32_t x = 0; x < d; ++x) dot += u[x] * v[x];
                        if (dot > tp->clip) dot = tp->clip;
                        if (dot < -tp->clip) dot = -tp->clip;
                        float var = (label - sigmoidf(dot)) * lrc;
                        for (uint32_t x = 0; x < d; ++x) {
                            gacc[x] += var * v[x];
                            v[x] += var * u[x];
                        }
                    }
                }
                for (uint32_t x = 0; x < d; ++x) cptr[x] += gacc[x];
                trained += r1 - r0;
                r0 = r1;
            }
        }
    }
    free(u);
    free(gacc);
    return trained;
}

/* rows first_row, first_row + stride, ... of the table o_init_table would produce */
void o_init_table_rows(float *t, uint64_t n_rows, uint32_t d, uint32_t ld, uint64_t seed,
                       uint32_t table_id, float scale, uint64_t first_row, uint64_t row_stride) {
    uint64_t key = o_mix64(seed ^ (O_TAG_INIT + table_id));
    for (uint64_t r = 0; r < n_rows; ++r) {
        uint64_t gr = first_row + r * row_stride;
        for (uint32_t c = 0; c < ld; ++c) {
            float v = 0.0f;
            if (c < d) {
                uint64_t h = o_draw(key, gr * d + c);
                float uu = (float)(h >> 40) * (1.0f / 16777216.0f);
                v = (2.0f * uu - 1.0f) * scale;
            }
            t[r * ld + c] = v;
        }
    }
}

// Random-walk sampler: one walker per wavefront lane over a CSR graph.
// Replaces the walk generation inside ensmallen's `fit_transform` / `Graph.node2vec`
// (reference call sites: embedders/ensmallen_embedders/node2vec.py:99,
//  sequences/tensorflow_sequences/node2vec_sequence.py:190-201).
//
// Second-order (return_weight = 1/p, explore_weight = 1/q; node2vec_skipgram.py:58-71) bias is
// sampled exactly by rejection: candidate ~ uniform (or weight-proportional) over N(cur), accepted
// against an integer threshold = product of the normalised factors of the candidate edge
//   {return, 1, explore} by class {prev, common neighbour, other}      (second order)
//   change_node_type_weight when type(candidate) != type(cur)          (graphs with node types)
//   change_edge_type_weight when type(edge) != type(previous edge)     (graphs with edge types)
// (node2vec_sequence.py:57-66); after max_trials rejections the lane falls back to an exact
// integer-weighted scan of the row.  All arithmetic that decides a transition is integer (or
// single rounded f32/f64 ops), so the walks are bit-identical to oracle/gn2v_oracle.c.
#pragma once
#include "rng.h"

namespace gn2v {


struct GraphView {
    const uint64_t *row_ptr;
    const uint32_t *col_idx;
    const float *cumw;        // nullptr for unweighted graphs
    const uint32_t *sources;  // nullptr when every node is a source
    const uint32_t *node_types;  // one id per node, or nullptr
    const uint32_t *edge_types;  // one id per directed edge (aligned with col_idx), or nullptr
    uint64_t n_nodes;
    uint64_t n_edges;
    uint64_t n_sources;
    uint32_t symmetric;  // every edge is stored in both directions (undirected graph)
    // Set of the directed edges (open addressing, key = source << 32 | destination, load <= 1/2;
    // nullptr: not built): the second-order sampler's "is x adjacent to prev" in about one memory
    // round trip instead of a binary search of log2(degree) dependent ones.  An accelerator only:
    // the answer, hence every walk, is the same.
    const unsigned long long *edge_set;
    uint64_t edge_mask;  // slots - 1 (a power of two)
    // A filter in front of it: one 64-bit word per ~8 edges with three bits set per edge (8-16
    // bits per edge: 256 MB for the 10 M / 100 M bench graph -- the size of the Infinity Cache).
    // Most candidates are NOT adjacent to the previous node, and a word that lacks one of the
    // three bits says so for certain (5 % pass by chance and are looked up in the set): the
    // adjacency test of the typical candidate becomes a cache hit instead of a random DRAM
    // access -- the sampler runs at the DRAM's random-access ceiling (5.4e10 sectors per second)
    // once the binary searches are gone.
    const unsigned long long *edge_filter;
    uint64_t filter_mask;  // words - 1 (a power of two)
    // Edge records (walk_rec_kernel): per directed edge, the destination with its row and the
    // signature of its neighbourhood; per node, that signature.  nullptr: not built.
    const uint4 *edge_rec;
    const uint32_t *node_sig;
    const uint4 *edge_rec_typed;  // two uint4 per edge: the record, then (node type, edge type)
};

constexpr unsigned long long kNoEdge = ~0ULL;

__device__ __forceinline__ uint64_t edge_slot(unsigned long long key, uint64_t mask) {
    return mix64(key) & mask;
}

constexpr unsigned long long kFilterSalt = 0xF117E2ED6E5E7ULL;

// (word, three bits) of an edge in the filter
__device__ __forceinline__ unsigned long long filter_bits(unsigned long long key, uint64_t mask,
                                                          uint64_t *word) {
    const uint64_t h = mix64(key ^ kFilterSalt);
    *word = h & mask;
    return (1ULL << (h >> 58)) | (1ULL << ((h >> 52) & 63)) | (1ULL << ((h >> 46) & 63));
}

static __global__ void edge_set_kernel(const uint64_t *__restrict__ row_ptr,
                                       const uint32_t *__restrict__ col, uint64_t n_nodes,
                                       unsigned long long *__restrict__ table, uint64_t mask,
                                       unsigned long long *__restrict__ filter,
                                       uint64_t filter_mask) {
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_nodes;
         u += (uint64_t)gridDim.x * blockDim.x) {
        for (uint64_t e = row_ptr[u]; e < row_ptr[u + 1]; ++e) {
            const unsigned long long key = (u << 32) | col[e];
            if (filter) {
                uint64_t word;
                const unsigned long long bits = filter_bits(key, filter_mask, &word);
                atomicOr(&filter[word], bits);
            }
            uint64_t slot = edge_slot(key, mask);
            for (;;) {
                const unsigned long long old = atomicCAS(&table[slot], kNoEdge, key);
                if (old == kNoEdge || old == key) break;
                slot = (slot + 1) & mask;
            }
        }
    }
}

__device__ __forceinline__ bool edge_set_contains(const GraphView &g, uint32_t u, uint32_t v) {
    const unsigned long long key = ((unsigned long long)u << 32) | v;
    if (g.edge_filter) {
        uint64_t word;
        const unsigned long long bits = filter_bits(key, g.filter_mask, &word);
        if ((g.edge_filter[word] & bits) != bits) return false;
    }
    uint64_t slot = edge_slot(key, g.edge_mask);
    for (;;) {
        const unsigned long long k = g.edge_set[slot];
        if (k == key) return true;
        if (k == kNoEdge) return false;
        slot = (slot + 1) & g.edge_mask;
    }
}

struct WalkConsts {
    uint32_t walk_length;
    uint32_t second_order;
    uint32_t node_bias, edge_bias;        // type factors active (types present and weight != 1)
    uint32_t max_trials;                  // rejections before the exact scan (host: trial_budget)
    uint64_t t_ret, t_common, t_explore;  // acceptance thresholds on a 2^32 scale
    uint64_t t_min, t_max;                // min / max of (t_common, t_explore)
    uint64_t fn_same, fn_diff, fe_same, fe_diff;
    // "Return apart" (return_weight above every other weight; unweighted untyped graphs): the
    // previous node is proposed on its own -- with probability R / (R + deg M) -- and the other
    // neighbours are accepted against their own envelope M = max(1, explore_weight): 2-4 x fewer
    // trials than one envelope for all (oracle: walk_consts.apart)
    uint32_t apart;
    uint64_t rq, mq;                      // R and M on a 2^20 scale
    uint64_t s_common, s_explore;         // 1 / M and explore_weight / M on a 2^32 scale
    uint64_t s_min, s_max;
    // max_neighbours (0: exact walks): a step out of a node of higher degree is taken over a
    // sub-sample of that many of its edges (RowView)
    uint32_t max_neighbours;
};

// gn2v_walks_strided: walk b of a launch has id first_walk + (b / id_group) * id_stride +
// b % id_group (a Node2VecSequence batch: id_group sources x iterations); id_group 0:
// first_walk + b.  (Kernel arguments of their own, not members of WalkConsts: the constants
// live in the scratch frame of the out-of-line exact scan.)
__device__ __forceinline__ uint64_t walk_id_of(uint32_t id_group, uint64_t id_stride,
                                               uint64_t first_walk, uint64_t b) {
    if (id_group == 0) return first_walk + b;
    const uint32_t b32 = (uint32_t)b, q = b32 / id_group;  // a launch holds < 2^31 walks
    return first_walk + (uint64_t)q * id_stride + (b32 - q * id_group);
}

// The row a step chooses from: all `deg` edges of the node, or -- degree > max_neighbours -- this
// visit's SUB-SAMPLE of max_neighbours of them (node2vec_skipgram.py:78-81; the oracle's row_view
// states the algorithm and its source): the row is cut into n buckets of step + (j < rem) edges,
// element j = bucket j's first edge + (hash32(vkey, j) x its size >> 32), vkey = the first draw of
// the step.  Counter based: a candidate costs one hash, nothing is enumerated.
struct RowView {
    uint64_t start, n, vkey;
    uint32_t step, rem;  // step == 0: the whole row (n = degree)
};

__device__ __forceinline__ RowView whole_row(uint64_t start, uint64_t deg) {
    return RowView{start, deg, 0, 0, 0};
}

// this visit's view of a row of `deg` > max_nb edges
__device__ __forceinline__ RowView sub_sampled_row(uint64_t start, uint64_t deg, uint32_t max_nb,
                                                   uint64_t vkey) {
    const uint32_t step = (uint32_t)(deg / max_nb);
    return RowView{start, max_nb, vkey, step, (uint32_t)(deg - (uint64_t)step * max_nb)};
}

// the draw of bucket j (32 bits are plenty for an offset inside a bucket, and a 64-bit splitmix
// per candidate was the dearest part of the step: 16 % of the sampler's rate on the bench graph)
__device__ __forceinline__ uint32_t view_hash(uint64_t vkey, uint64_t j) {
    uint32_t h = (uint32_t)(vkey >> 32) + ((uint32_t)j + 1u) * 0x9E3779B1u;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

// element j of the view -> edge id
__device__ __forceinline__ uint64_t view_edge(const RowView &v, uint64_t j) {
    if (v.step == 0) return v.start + j;
    const uint64_t lo = j * v.step + (j < v.rem ? j : v.rem);
    return v.start + lo +
           (((uint64_t)view_hash(v.vkey, j) * ((uint64_t)v.step + (j < v.rem ? 1u : 0u))) >> 32);
}

// is edge index i of the row (counted from its start) an element of the sub-sample?
__device__ __forceinline__ bool view_holds(const RowView &v, uint64_t i) {
    if (v.step == 0) return true;
    const uint64_t big = (uint64_t)v.rem * (v.step + 1u);
    const uint64_t b = i < big ? i / (v.step + 1u) : v.rem + (i - big) / v.step;
    return view_edge(v, b) == v.start + i;
}

__device__ __forceinline__ bool adj_contains(const uint32_t *__restrict__ col, uint64_t lo,
                                             uint64_t hi, uint32_t x) {
    const uint64_t end = hi;
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (col[mid] < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo < end && col[lo] == x;
}

// Is x adjacent to prev?  On a symmetric graph the question can be asked from either side, so a
// long adjacency list of prev (a hub: up to 17 dependent loads) is replaced by the list of x when
// that one is shorter -- in lock step the deepest search of the wave sets the pace.
constexpr uint64_t kLongRow = 64;

__device__ __forceinline__ bool is_common_neighbour(const GraphView &g, uint32_t x, uint32_t prev,
                                                    uint64_t pstart, uint64_t pend) {
    if (g.edge_set) return edge_set_contains(g, prev, x);
    if (g.symmetric && pend - pstart > kLongRow) {
        const uint64_t xs = g.row_ptr[x], xe = g.row_ptr[x + 1];
        if (xe - xs < pend - pstart) return adj_contains(g.col_idx, xs, xe, prev);
    }
    return adj_contains(g.col_idx, pstart, pend, x);
}

__device__ __forceinline__ uint64_t pick_index(const GraphView &g, uint64_t start, uint64_t deg,
                                               uint64_t r) {
    if (g.cumw == nullptr) return ((r >> 32) * deg) >> 32;
    const float total = g.cumw[start + deg - 1];
    const float f = __fmul_rn(__fmul_rn((float)(r >> 40), 1.0f / 16777216.0f), total);
    uint64_t lo = 0, hi = deg;
    while (lo < hi) {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (g.cumw[start + mid] > f)
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo < deg ? lo : deg - 1;
}

// t * f / 2^32 with t, f <= 2^32 (f == 2^32 is the identity)
__device__ __forceinline__ uint64_t scale32(uint64_t t, uint64_t f) {
    return f >= (1ULL << 32) ? t : (t * f) >> 32;
}

// acceptance threshold of candidate edge e = cur -> x (TYPED = false compiles the type factors
// out: the untyped instantiation of the walk kernel needs a third fewer registers)
template <bool TYPED>
__device__ __forceinline__ uint64_t accept_threshold(const GraphView &g, const WalkConsts &c,
                                                     uint32_t cur, uint32_t x, uint64_t e,
                                                     uint32_t prev, uint64_t pstart,
                                                     uint64_t pend, uint32_t ptype) {
    uint64_t t = 1ULL << 32;
    if (c.second_order && prev != kSentinel)
        t = (x == prev) ? c.t_ret
            : is_common_neighbour(g, x, prev, pstart, pend) ? c.t_common
                                                            : c.t_explore;
    if constexpr (TYPED) {
        if (c.node_bias)
            t = scale32(t, g.node_types[cur] != g.node_types[x] ? c.fn_diff : c.fn_same);
        if (c.edge_bias && prev != kSentinel)
            t = scale32(t, g.edge_types[e] != ptype ? c.fe_diff : c.fe_same);
    }
    return t;
}

// Bounds of the acceptance threshold that need no adjacency search: the class of a candidate other
// than `prev` is either "common neighbour" or "other", so its threshold lies in [lo, hi]; a draw
// below lo is accepted and a draw at or above hi rejected whatever the class turns out to be
// (lo == hi when the class does not matter or is known).  Same decision as accept_threshold.
struct ThresholdBounds {
    uint64_t lo, hi;
};

template <bool TYPED>
__device__ __forceinline__ ThresholdBounds threshold_bounds(const GraphView &g,
                                                            const WalkConsts &c, uint32_t cur,
                                                            uint32_t x, uint64_t e, uint32_t prev,
                                                            uint32_t ptype) {
    ThresholdBounds b{1ULL << 32, 1ULL << 32};
    if (c.second_order && prev != kSentinel) {
        if (x == prev) {
            b.lo = b.hi = c.t_ret;
        } else {
            b.lo = c.t_min;
            b.hi = c.t_max;
        }
    }
    if constexpr (TYPED) {
        if (c.node_bias) {
            const uint64_t f = g.node_types[cur] != g.node_types[x] ? c.fn_diff : c.fn_same;
            b.lo = scale32(b.lo, f);
            b.hi = scale32(b.hi, f);
        }
        if (c.edge_bias && prev != kSentinel) {
            const uint64_t f = g.edge_types[e] != ptype ? c.fe_diff : c.fe_same;
            b.lo = scale32(b.lo, f);
            b.hi = scale32(b.hi, f);
        }
    }
    return b;
}

// exact fallback after max_trials rejections (rare: only for extreme weights on low-weight rows;
// the one step of a weighted row under a sub-sample); returns the chosen edge
template <bool TYPED>
// (the view BY VALUE: a reference would take the address of the caller's view and park it -- a
// loop-carried variable of every trial -- in scratch memory: 21 % of the sampler's rate)
__device__ __noinline__ uint64_t exact_scan(const GraphView &g, const WalkConsts &c, uint64_t r,
                                            uint32_t cur, const RowView v, uint32_t prev,
                                            uint64_t pstart, uint64_t pend, uint32_t ptype) {
    const uint64_t n = v.n;
    if (g.cumw == nullptr) {
        uint64_t total = 0;
        for (uint64_t j = 0; j < n; ++j) {
            const uint64_t e = view_edge(v, j);
            total += accept_threshold<TYPED>(g, c, cur, g.col_idx[e], e, prev, pstart, pend, ptype);
        }
        if (total == 0) return view_edge(v, ((r >> 32) * n) >> 32);
        const uint64_t target = mulhi64(r, total);
        uint64_t acc = 0;
        for (uint64_t j = 0; j < n; ++j) {
            const uint64_t e = view_edge(v, j);
            acc += accept_threshold<TYPED>(g, c, cur, g.col_idx[e], e, prev, pstart, pend, ptype);
            if (acc > target) return e;
        }
        return view_edge(v, n - 1);
    }
    double total = 0.0;
    for (uint64_t j = 0; j < n; ++j) {
        const uint64_t e = view_edge(v, j);
        const double w = __dsub_rn((double)g.cumw[e], e > v.start ? (double)g.cumw[e - 1] : 0.0);
        const uint64_t thr =
            accept_threshold<TYPED>(g, c, cur, g.col_idx[e], e, prev, pstart, pend, ptype);
        total = __dadd_rn(total, __dmul_rn(w, (double)thr));
    }
    const double target =
        __dmul_rn(__dmul_rn((double)(r >> 11), 1.0 / 9007199254740992.0), total);
    double acc = 0.0;
    for (uint64_t j = 0; j < n; ++j) {
        const uint64_t e = view_edge(v, j);
        const double w = __dsub_rn((double)g.cumw[e], e > v.start ? (double)g.cumw[e - 1] : 0.0);
        const uint64_t thr =
            accept_threshold<TYPED>(g, c, cur, g.col_idx[e], e, prev, pstart, pend, ptype);
        acc = __dadd_rn(acc, __dmul_rn(w, (double)thr));
        if (acc > target) return e;
    }
    return view_edge(v, n - 1);
}

// One lane = one walker.  Output row-major u32[n_walks][walk_length]; every 16 steps a wave
// flushes a [64 walks][16 steps] LDS tile so each walk row is written as whole 64-byte segments.
constexpr int kWalkBlock = 256;
constexpr int kTileSteps = 16;

// SUB: some row of the graph is longer than max_neighbours (the host knows the largest degree):
// only then is the sub-sampling code compiled in -- exact walks keep their instruction count.
template <bool TYPED, bool SUB>
__global__ __launch_bounds__(kWalkBlock) void walk_kernel(GraphView g, WalkConsts c, uint64_t ekey,
                                                          uint64_t first_walk, uint64_t n_walks,
                                                          uint32_t *__restrict__ out,
                                                          unsigned long long *__restrict__ counters,
                                                          uint32_t id_group, uint64_t id_stride) {
    __shared__ uint32_t tile[kWalkBlock / 64][64][kTileSteps + 1];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint64_t wave_base = ((uint64_t)blockIdx.x * kWalkBlock + (uint64_t)wave * 64);
    if (wave_base >= n_walks) return;
    const uint64_t b = wave_base + lane;
    const bool live = b < n_walks;
    const uint32_t L = c.walk_length;

    uint64_t wkey = 0, ctr = 0;
    uint32_t cur = kSentinel, prev = kSentinel, ptype = 0;
    uint64_t pstart = 0, pend = 0;
    if (live) {
        const uint64_t wid = walk_id_of(id_group, id_stride, first_walk, b);
        const uint64_t si = wid % g.n_sources;
        cur = g.sources ? g.sources[si] : (uint32_t)si;
        wkey = draw(ekey, wid);
    }
    bool dead = !live;
    uint32_t steps = 0;

    for (uint32_t t0 = 0; t0 < L; t0 += kTileSteps) {
        const uint32_t tn = min((uint32_t)kTileSteps, L - t0);
        for (uint32_t tt = 0; tt < tn; ++tt) {
            const uint32_t t = t0 + tt;
            uint32_t val = kSentinel;
            if (t == 0) {
                val = cur;
            } else if (!dead) {
                const uint64_t start = g.row_ptr[cur];
                const uint64_t end = g.row_ptr[cur + 1];
                const uint64_t deg = end - start;
                if (deg == 0) {
                    dead = true;
                } else {
                    uint64_t edge;  // the edge the step takes
                    const bool biased =
                        (TYPED && c.node_bias) ||
                        (prev != kSentinel && (c.second_order || (TYPED && c.edge_bias)));
                    // degree > max_neighbours: this visit's sub-sample (RowView)
                    RowView v = whole_row(start, deg);
                    if constexpr (SUB)
                        if (deg > c.max_neighbours)
                            v = sub_sampled_row(start, deg, c.max_neighbours, draw(wkey, ctr++));
                    // candidate j of the view: uniform, or weight proportional on a whole row
                    auto pick = [&](uint64_t r) -> uint64_t {
                        return SUB && v.step ? view_edge(v, ((r >> 32) * v.n) >> 32)
                                             : start + pick_index(g, start, deg, r);
                    };
                    if (SUB && g.cumw != nullptr && v.step) {
                        // weights on a sub-sample: the scan over its elements (oracle: the same)
                        const uint64_t r = draw(wkey, ctr++);
                        edge = exact_scan<TYPED>(g, c, r, cur, v, biased ? prev : kSentinel, pstart,
                                                 pend, ptype);
                    } else if (!biased || deg == 1) {
                        edge = pick(draw(wkey, ctr++));
                    } else {
                        if (!TYPED && c.apart) {
                            // The previous node proposed on its own (WalkConsts.apart).  Trial j
                            // draws r1 (prev or a uniform neighbour?) and, for a neighbour, r2
                            // (which one, and the acceptance draw): counter based, so the
                            // candidate of trial j + 1 is fetched while trial j waits for its own
                            // candidate and, when its draw lies between the class thresholds,
                            // for the adjacency test -- one memory round trip per trial instead
                            // of two; the look-ahead of an accepted trial is simply not consumed.
                            bool accepted = false;
                            edge = start;
                            uint32_t trial = 0;
                            const uint64_t z = c.rq + v.n * c.mq;
                            // the trial in hand: direct (prev) or a neighbour e -> x with draw r32
                            bool direct = mulhi64(draw(wkey, ctr), z) < c.rq;
                            uint64_t r2 = direct ? 0 : draw(wkey, ctr + 1);
                            uint64_t e = direct ? start : view_edge(v, ((r2 >> 32) * v.n) >> 32);
                            uint32_t x = direct ? 0u : g.col_idx[e];
                            while (trial < c.max_trials) {
                                const uint64_t used = direct ? 1 : 2;
                                // look ahead: the next trial's candidate
                                const bool n_direct = mulhi64(draw(wkey, ctr + used), z) < c.rq;
                                const uint64_t n_r2 = n_direct ? 0 : draw(wkey, ctr + used + 1);
                                const bool n_fetch = !n_direct && trial + 1 < c.max_trials;
                                const uint64_t n_e =
                                    n_fetch ? view_edge(v, ((n_r2 >> 32) * v.n) >> 32) : start;
                                const uint32_t n_x = n_fetch ? g.col_idx[n_e] : 0u;
                                ++trial;
                                ctr += used;
                                if (direct) {
                                    uint64_t lo = start, hi = end;
                                    while (lo < hi) {
                                        const uint64_t mid = lo + ((hi - lo) >> 1);
                                        if (g.col_idx[mid] < prev)
                                            lo = mid + 1;
                                        else
                                            hi = mid;
                                    }
                                    // (of a sub-sample only when its bucket drew it)
                                    if (lo < end && g.col_idx[lo] == prev && view_holds(v, lo - start)) {
                                        edge = lo;
                                        accepted = true;
                                    }
                                } else if (x != prev) {
                                    const uint64_t r32 = r2 & 0xFFFFFFFFULL;
                                    if (r32 < c.s_min)
                                        accepted = true;
                                    else if (r32 < c.s_max)
                                        accepted = r32 < (is_common_neighbour(g, x, prev, pstart, pend)
                                                              ? c.s_common
                                                              : c.s_explore);
                                    if (accepted) edge = e;
                                }
                                if (accepted) break;
                                direct = n_direct;
                                r2 = n_r2;
                                e = n_e;
                                x = n_x;
                            }
                            if (!accepted) {
                                const uint64_t r = draw(wkey, ctr++);
                                edge = exact_scan<TYPED>(g, c, r, cur, v, prev, pstart, pend, ptype);
                            }
                        } else {
                        // Trials in two phases so that the wave pays for an adjacency search
                        // only when some lane's draw falls between the class thresholds: phase A
                        // runs trials until one is decided without the search (accept) or needs
                        // it; phase B searches for the lanes that need it, in lock step.
                        bool accepted = false;
                        edge = start;
                        uint32_t trial = 0;
                        while (trial < c.max_trials) {
                            uint64_t r = 0, e = start;
                            bool pending = false;
                            while (trial < c.max_trials) {
                                r = draw(wkey, ctr++);
                                e = pick(r);
                                ++trial;
                                const ThresholdBounds b = threshold_bounds<TYPED>(
                                    g, c, cur, g.col_idx[e], e, prev, ptype);
                                const uint64_t r32 = r & 0xFFFFFFFFULL;
                                if (r32 < b.lo) {
                                    accepted = true;
                                    break;
                                }
                                if (r32 < b.hi) {
                                    pending = true;
                                    break;
                                }
                            }
                            if (pending) {
                                const uint64_t thr = accept_threshold<TYPED>(
                                    g, c, cur, g.col_idx[e], e, prev, pstart, pend, ptype);
                                accepted = (r & 0xFFFFFFFFULL) < thr;
                            }
                            if (accepted) {
                                edge = e;
                                break;
                            }
                            if (!pending) break;  // trials exhausted
                        }
                        if (!accepted) {
                            const uint64_t r = draw(wkey, ctr++);
                            edge = exact_scan<TYPED>(g, c, r, cur, v, prev, pstart, pend, ptype);
                        }
                        }
                    }
                    const uint32_t nxt = g.col_idx[edge];
                    if constexpr (TYPED)
                        if (c.edge_bias) ptype = g.edge_types[edge];
                    val = nxt;
                    prev = cur;
                    pstart = start;
                    pend = end;
                    cur = nxt;
                    ++steps;
                }
            }
            tile[wave][lane][tt] = val;
        }
        // flush: 64 walks x tn steps; lane -> (walk = lane/4 + 16*pass, 4-step quarter = lane%4)
        __builtin_amdgcn_wave_barrier();
        for (int pass = 0; pass < 4; ++pass) {
            const int wrow = (lane >> 2) + 16 * pass;
            const uint64_t wb = wave_base + wrow;
            const int q = lane & 3;
            if (wb < n_walks) {
                uint32_t *dst = out + wb * L + t0 + q * 4;
                for (int e = 0; e < 4; ++e) {
                    const uint32_t tt = q * 4 + e;
                    if (tt < tn) dst[e] = tile[wave][wrow][tt];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (counters) {
        // one atomic per wave
        uint32_t s = steps;
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) atomicAdd(&counters[1], (unsigned long long)s);
    }
}

// ---- edge records -------------------------------------------------------------------------------
// The CSR walk pays three dependent memory round trips per step: row_ptr[cur] (one 64 B sector for
// 16 B), col_idx[start + i] (one sector for 4 B) and the adjacency test of the candidate against
// the previous node (filter word, sometimes the set).  An edge record carries, beside the
// destination x of edge e, everything the NEXT step needs to know about x:
//   .x = x   .y = signature of N(x)   .z/.w = row start of x (40 bits) | degree of x (24 bits)
// 16 B, four to a sector, so a step of an unweighted untyped walk reads ONE sector for the
// candidate and already holds the row of the node it moves to.  The signature is a 32-bit Bloom
// word of the node's out-neighbours (bit sig_slot(y) for every y in N(x)): "x adjacent to prev"
// is answered "no" for certain from registers when prev's signature lacks x's bit -- the typical
// case on a sparse graph -- and goes to the filter / set / binary search only otherwise.  An
// accelerator only: every decision, hence every walk, is the same as on the CSR arrays.
__device__ __forceinline__ uint32_t sig_slot(uint32_t x) { return (x * 0x9E3779B1u) >> 27; }

constexpr uint64_t kRecMaxDegree = (1ULL << 24) - 1;
constexpr uint64_t kRecMaxEdges = 1ULL << 40;

// flag[0] |= 1 when a row is too long for a record
static __global__ void node_sig_kernel(const uint64_t *__restrict__ row_ptr,
                                       const uint32_t *__restrict__ col, uint64_t n_nodes,
                                       uint32_t *__restrict__ sig, uint32_t *__restrict__ flag) {
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n_nodes;
         u += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t lo = row_ptr[u], hi = row_ptr[u + 1];
        if (hi - lo > kRecMaxDegree) atomicOr(flag, 1u);
        uint32_t s = 0;
        for (uint64_t e = lo; e < hi && s != 0xFFFFFFFFu; ++e) s |= 1u << sig_slot(col[e]);
        sig[u] = s;
    }
}

static __global__ void edge_rec_kernel(const uint64_t *__restrict__ row_ptr,
                                       const uint32_t *__restrict__ col,
                                       const uint32_t *__restrict__ sig, uint64_t n_edges,
                                       uint4 *__restrict__ rec) {
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges;
         e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t x = col[e];
        const uint64_t lo = row_ptr[x], deg = row_ptr[x + 1] - lo;
        rec[e] = make_uint4(x, sig[x], (uint32_t)lo, (uint32_t)(lo >> 32) | (uint32_t)(deg << 8));
    }
}

// what a walker knows about a node it stands on (or came from)
struct NodeRow {
    uint32_t id;
    uint32_t sig;
    uint64_t start;
    uint32_t deg;
};

__device__ __forceinline__ NodeRow row_of_record(const uint4 r) {
    return NodeRow{r.x, r.y, (uint64_t)r.z | ((uint64_t)(r.w & 0xFFu) << 32), r.w >> 8};
}

// is_common_neighbour with prev's signature in front
__device__ __forceinline__ bool rec_maybe_common(const NodeRow &prev, uint32_t x) {
    return (prev.sig >> sig_slot(x)) & 1u;
}

// The typed form of a record, 32 B (two to a sector): the 16 B above, then the node type of x and
// the type of edge e -- the two reads a type factor costs per candidate on the CSR arrays.
static __global__ void edge_rec_typed_kernel(const uint64_t *__restrict__ row_ptr,
                                             const uint32_t *__restrict__ col,
                                             const uint32_t *__restrict__ sig,
                                             const uint32_t *__restrict__ node_types,
                                             const uint32_t *__restrict__ edge_types,
                                             uint64_t n_edges, uint4 *__restrict__ rec) {
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges;
         e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t x = col[e];
        const uint64_t lo = row_ptr[x], deg = row_ptr[x + 1] - lo;
        rec[2 * e] = make_uint4(x, sig[x], (uint32_t)lo, (uint32_t)(lo >> 32) | (uint32_t)(deg << 8));
        rec[2 * e + 1] =
            make_uint4(node_types ? node_types[x] : 0u, edge_types ? edge_types[e] : 0u, 0u, 0u);
    }
}

// a candidate: the record of edge cur -> x (TYPED: with the types of x and of the edge)
template <bool TYPED>
struct Candidate {
    uint4 a;
    uint32_t ntype, etype;
};

template <bool TYPED>
__device__ __forceinline__ Candidate<TYPED> fetch_candidate(const uint4 *__restrict__ rec,
                                                            uint64_t e) {
    Candidate<TYPED> k;
    if constexpr (TYPED) {
        k.a = rec[2 * e];
        const uint2 ty = *reinterpret_cast<const uint2 *>(rec + 2 * e + 1);
        k.ntype = ty.x;
        k.etype = ty.y;
    } else {
        k.a = rec[e];
        k.ntype = k.etype = 0;
    }
    return k;
}

// The walk of walk_kernel<TYPED> read from the edge records (weighted graphs too: the candidate's
// index comes from the row's cumulative weights, its record from here): per walk the
// same draws in the same order and the same decisions, hence the same walks -- with the lanes of a
// wave out of lock step.  In walk_kernel a step lasts as long as the slowest lane's trials (the
// maximum of 64 geometric variables: 7 round trips at acceptance 1/2 where the mean is 2).  Here a
// lane runs ONE trial per pass of the wave and moves on to its next step as soon as a trial is
// accepted; the lanes meet again only where a tile of 16 steps is flushed.  The adjacency test of a
// candidate that needs one is a pass of its own ("pending": the filter word is loaded beside the
// other lanes' candidates, one wait for both).  Draws are counter based per walk, so the order in
// which lanes run changes no walk.
template <bool TYPED, bool SUB>
__global__ __launch_bounds__(kWalkBlock, 3) void walk_rec_kernel(GraphView g, WalkConsts c,
                                                                  uint64_t ekey, uint64_t first_walk,
                                                                  uint64_t n_walks,
                                                                  uint32_t *__restrict__ out,
                                                                  unsigned long long *__restrict__ counters,
                                                                  uint32_t id_group,
                                                                  uint64_t id_stride) {
    __shared__ uint32_t tile[kWalkBlock / 64][64][kTileSteps + 1];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint64_t wave_base = ((uint64_t)blockIdx.x * kWalkBlock + (uint64_t)wave * 64);
    if (wave_base >= n_walks) return;
    const uint64_t b = wave_base + lane;
    const bool live = b < n_walks;
    const uint32_t L = c.walk_length;
    const uint4 *__restrict__ rec = TYPED ? g.edge_rec_typed : g.edge_rec;
    // the envelope in use: "return apart" proposes prev on its own and never through a neighbour
    const bool apart = !TYPED && c.apart;
    const uint64_t k_ret = apart ? 0 : c.t_ret;
    const uint64_t k_common = apart ? c.s_common : c.t_common;
    const uint64_t k_explore = apart ? c.s_explore : c.t_explore;
    const uint64_t k_min = apart ? c.s_min : c.t_min, k_max = apart ? c.s_max : c.t_max;

    uint64_t wkey = 0, ctr = 0;
    NodeRow cur{kSentinel, 0, 0, 0}, prev{kSentinel, 0, 0, 0};
    uint32_t cur_ntype = 0, ptype = 0;
    if (live) {
        const uint64_t wid = walk_id_of(id_group, id_stride, first_walk, b);
        const uint64_t si = wid % g.n_sources;
        cur.id = g.sources ? g.sources[si] : (uint32_t)si;
        cur.start = g.row_ptr[cur.id];
        cur.deg = (uint32_t)(g.row_ptr[cur.id + 1] - cur.start);
        cur.sig = g.node_sig[cur.id];
        if constexpr (TYPED)
            if (g.node_types) cur_ntype = g.node_types[cur.id];
        wkey = draw(ekey, wid);
    }
    bool dead = !live;
    uint32_t steps = 0;
    // the step in hand: trials so far; a candidate waiting for its adjacency test; the visit's
    // sub-sample when the row is longer than max_neighbours (set at the step's first trial)
    uint32_t trial = 0;
    bool pend = false;
    RowView v = whole_row(0, 0);
    Candidate<TYPED> held{};
    uint32_t held_r32 = 0;
    bool ntype_differs = false, etype_differs = false;  // of the held candidate (TYPED)

    for (uint32_t t0 = 0; t0 < L; t0 += kTileSteps) {
        const uint32_t tend = min(t0 + (uint32_t)kTileSteps, L);
        uint32_t t = t0;
        if (dead) {
            for (; t < tend; ++t) tile[wave][lane][t - t0] = kSentinel;
        } else if (t == 0) {
            tile[wave][lane][0] = cur.id;
            t = 1;
        }
        while (__any(t < tend)) {
            const bool active = t < tend;
            const uint64_t deg = cur.deg;
            const bool walked = prev.id != kSentinel;
            const bool second = c.second_order && walked;
            const bool biased =
                ((TYPED && c.node_bias) || second || (TYPED && c.edge_bias && walked)) && deg != 1;
            // ---- what this pass loads: a candidate's record, or a held candidate's filter word
            bool do_trial = active && !pend && deg != 0, direct = false;
            const bool do_pend = active && pend;
            uint64_t r32 = 0;
            Candidate<TYPED> x{};
            bool scanned = false;  // the candidate came out of an exact scan: accepted as it is
            if (do_trial) {
                if constexpr (SUB) {
                    if (trial == 0) {  // the row this step chooses from (RowView)
                        v = whole_row(cur.start, deg);
                        if (deg > c.max_neighbours)
                            v = sub_sampled_row(cur.start, deg, c.max_neighbours,
                                                draw(wkey, ctr++));
                    }
                } else {
                    v = whole_row(cur.start, deg);
                }
                uint64_t r = draw(wkey, ctr++);
                if (SUB && g.cumw != nullptr && v.step) {
                    // weights on a sub-sample: the scan over its elements (walk_kernel, oracle)
                    x = fetch_candidate<TYPED>(
                        rec, exact_scan<TYPED>(g, c, r, cur.id, v, biased ? prev.id : kSentinel,
                                               prev.start, prev.start + prev.deg, ptype));
                    scanned = true;
                } else {
                    if (apart && biased) {
                        direct = mulhi64(r, c.rq + v.n * c.mq) < c.rq;
                        if (!direct) r = draw(wkey, ctr++);
                    }
                    r32 = r & 0xFFFFFFFFULL;
                    // (weighted graphs: the candidate is found in the row's cumulative weights, as
                    // in walk_kernel; its record then saves the row_ptr read and most adjacency
                    // tests)
                    if (!direct)
                        x = fetch_candidate<TYPED>(
                            rec, SUB && v.step ? view_edge(v, ((r >> 32) * v.n) >> 32)
                                               : cur.start + pick_index(g, cur.start, deg, r));
                }
                ++trial;
            }
            unsigned long long fword = ~0ULL, fbits = 0;
            if (do_pend && g.edge_filter) {
                uint64_t word;
                fbits = filter_bits(((unsigned long long)prev.id << 32) | held.a.x, g.filter_mask,
                                    &word);
                fword = g.edge_filter[word];
            }
            // ---- decisions
            bool accepted = false, back = false;
            if (active && !pend && deg == 0) {
                dead = true;
                for (; t < tend; ++t) tile[wave][lane][t - t0] = kSentinel;
            }
            if (do_trial) {
                if (!biased || scanned) {
                    accepted = true;
                } else if (direct) {
                    // prev is a neighbour of cur on a symmetric graph; otherwise look for the
                    // edge cur -> prev in the row.  Under a sub-sample the edge must be found in
                    // any case: it counts only when its bucket drew it.
                    if (SUB && v.step) {
                        uint64_t lo = cur.start, hi = cur.start + deg;
                        while (lo < hi) {
                            const uint64_t mid = lo + ((hi - lo) >> 1);
                            if (g.col_idx[mid] < prev.id)
                                lo = mid + 1;
                            else
                                hi = mid;
                        }
                        accepted = back = lo < cur.start + deg && g.col_idx[lo] == prev.id &&
                                          view_holds(v, lo - cur.start);
                    } else {
                        accepted = back = g.symmetric || adj_contains(g.col_idx, cur.start,
                                                                      cur.start + deg, prev.id);
                    }
                } else {
                    uint64_t lo = 1ULL << 32, hi = 1ULL << 32;
                    if (second) {
                        if (x.a.x == prev.id) {
                            lo = hi = k_ret;
                        } else if (!rec_maybe_common(prev, x.a.x)) {
                            lo = hi = k_explore;
                        } else {
                            lo = k_min;
                            hi = k_max;
                        }
                    }
                    if constexpr (TYPED) {
                        if (c.node_bias) {
                            ntype_differs = cur_ntype != x.ntype;
                            const uint64_t f = ntype_differs ? c.fn_diff : c.fn_same;
                            lo = scale32(lo, f);
                            hi = scale32(hi, f);
                        }
                        if (c.edge_bias && walked) {
                            etype_differs = x.etype != ptype;
                            const uint64_t f = etype_differs ? c.fe_diff : c.fe_same;
                            lo = scale32(lo, f);
                            hi = scale32(hi, f);
                        }
                    }
                    if (r32 < lo) {
                        accepted = true;
                    } else if (r32 < hi) {
                        pend = true;
                        held = x;
                        held_r32 = (uint32_t)r32;
                    }
                }
            } else if (do_pend) {
                pend = false;
                x = held;
                r32 = held_r32;
                bool common = (fword & fbits) == fbits;  // no filter: always "maybe"
                if (common) {
                    if (g.edge_set) {
                        const unsigned long long key = ((unsigned long long)prev.id << 32) | x.a.x;
                        uint64_t slot = edge_slot(key, g.edge_mask);
                        for (;;) {
                            const unsigned long long k = g.edge_set[slot];
                            common = k == key;
                            if (k == key || k == kNoEdge) break;
                            slot = (slot + 1) & g.edge_mask;
                        }
                    } else {
                        common = is_common_neighbour(g, x.a.x, prev.id, prev.start,
                                                     prev.start + prev.deg);
                    }
                }
                uint64_t thr = common ? k_common : k_explore;
                if constexpr (TYPED) {
                    if (c.node_bias) thr = scale32(thr, ntype_differs ? c.fn_diff : c.fn_same);
                    if (c.edge_bias && walked)
                        thr = scale32(thr, etype_differs ? c.fe_diff : c.fe_same);
                }
                accepted = r32 < thr;
            }
            if ((do_trial || do_pend) && !accepted && !pend && trial >= c.max_trials) {
                // rare: the exact scan of the row
                const uint64_t r = draw(wkey, ctr++);
                x = fetch_candidate<TYPED>(
                    rec, exact_scan<TYPED>(g, c, r, cur.id, v, prev.id, prev.start,
                                           prev.start + prev.deg, ptype));
                accepted = true;
            }
            if (accepted) {
                const NodeRow nxt = back ? prev : row_of_record(x.a);
                if constexpr (TYPED) {
                    cur_ntype = x.ntype;
                    ptype = x.etype;
                }
                tile[wave][lane][t - t0] = nxt.id;
                prev = cur;
                cur = nxt;
                ++steps;
                ++t;
                trial = 0;
            }
        }
        // flush: 64 walks x tn steps; lane -> (walk = lane/4 + 16*pass, 4-step quarter = lane%4)
        const uint32_t tn = tend - t0;
        __builtin_amdgcn_wave_barrier();
        for (int pass = 0; pass < 4; ++pass) {
            const int wrow = (lane >> 2) + 16 * pass;
            const uint64_t wb = wave_base + wrow;
            const int q = lane & 3;
            if (wb < n_walks) {
                uint32_t *dst = out + wb * L + t0 + q * 4;
                for (int e = 0; e < 4; ++e) {
                    const uint32_t tt = q * 4 + e;
                    if (tt < tn) dst[e] = tile[wave][wrow][tt];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (counters) {
        uint32_t s = steps;
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) atomicAdd(&counters[1], (unsigned long long)s);
    }
}

// Node2VecSequence batch form: words / contexts of every full-window position.
static __global__ void window_kernel(const uint32_t *__restrict__ walks, uint64_t n_walks, uint32_t L,
                              uint32_t w, int32_t *__restrict__ contexts,
                              int32_t *__restrict__ words) {
    const uint32_t per = L - 2 * w;
    const uint64_t n = n_walks * per;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = i / per;
        const uint32_t pos = (uint32_t)(i % per) + w;
        const uint32_t *wk = walks + b * L;
        words[i] = (int32_t)wk[pos];
        int32_t *dst = contexts + i * 2 * w;
        for (uint32_t s = 0; s < w; ++s) {
            dst[s] = (int32_t)wk[pos - w + s];
            dst[w + s] = (int32_t)wk[pos + 1 + s];
        }
    }
}

// (centre, context) pairs of every walk position, window trimmed at the walk borders: slot
// [walk][position][2w] holds the pair or (sentinel, sentinel).
static __global__ void pairs_kernel(const uint32_t *__restrict__ walks, uint64_t n_walks,
                                    uint32_t L, uint32_t w, uint32_t min_dist,
                                    uint32_t *__restrict__ pairs) {
    const uint64_t n = n_walks * L * 2 * w;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t slot = (uint32_t)(t % (2 * w));
        const uint64_t pos = t / (2 * w);
        const uint32_t i = (uint32_t)(pos % L);
        const uint64_t b = pos / L;
        const int64_t j = slot < w ? (int64_t)i - w + slot : (int64_t)i + 1 + (slot - w);
        uint32_t c = kSentinel, x = kSentinel;
        if (j >= 0 && j < (int64_t)L) {
            const uint32_t dist = (uint32_t)(j > (int64_t)i ? j - i : i - j);
            const uint32_t ci = walks[b * L + i], xj = walks[b * L + j];
            // a sentinel at j means the walk ended before j (sentinels are a suffix)
            if (dist >= min_dist && ci != kSentinel && xj != kSentinel) {
                c = ci;
                x = xj;
            }
        }
        pairs[2 * t] = c;
        pairs[2 * t + 1] = x;
    }
}

}  // namespace gn2v

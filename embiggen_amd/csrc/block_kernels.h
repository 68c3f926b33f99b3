// Block-partitioned SkipGram: the fused negative-sampling step over pairs grouped by (context
// cell, centre) -- the multi-GPU trainer and, with one rank, the default for large graphs.
//
// The reference has no counterpart (ensmallen trains in one process with rayon threads; the call
// being replaced is embedders/ensmallen_embedders/node2vec.py:99).  Scheme:
// nodes are striped over `world` ranks (centre c lives on rank c % world, row c / world) and over
// `parts` context parts (context x lives in part x % parts, row x / parts; parts is a multiple of
// world and the parts travel round the ranks).  Inside a part the rows are striped once more over
// `slices` (slice = row % slices).  When the number of slices is a multiple of the number of XCDs
// (8 slices on an MI355X: one per XCD) every contextual row is only ever touched by the workgroups
// of ONE XCD during a launch -- its L2 is then coherent for that row, plain write-back stores are
// safe and hub rows (degree-proportional negatives) stay L2 resident.  With any other slice count
// several XCDs share a slice and the rows are updated with write-through stores, like unsliced
// parts (gn2v_block_step decides).  cell = part * slices + slice.
//
// Per round and group of parts: every (centre, context) pair of the round's walks whose centre
// this rank owns and whose context lies in the group is emitted as ONE 64-bit word,
//     cell << (row_bits + ctx_bits) | centre row << ctx_bits | hot << (ctx_bits - 1) | context row
// (the context row counted inside its cell: (x / parts) / slices), by two passes over the walks
// (count, then write at scanned offsets -- the output is in walk / position / slot order, whatever
// the launch geometry); one stable radix sort on the bits above ctx_bits groups the words by cell
// and centre, and the training kernel walks a cell in implicit records of `record` consecutive
// pairs: a wave keeps the centre row in registers while the centre does not change, records are
// visited in a golden-ratio stride order (a hub centre owns thousands of consecutive records;
// concurrent waves must not all accumulate into the same row).  Nothing is packed, padded or
// moved after the sort.  8 B per pair whatever the graph's size (100 M nodes on one GPU:
// 12 + 27 + 17 bits); a round's pairs are extracted and sorted a group of parts at a time, so the
// buffers hold 1 / groups of the round.
#pragma once
#include "../../include/gn2v_internal.h"
#include "train_kernels.h"

namespace gn2v {

constexpr uint64_t kTagBlock = 0xB10C5EED0B10C5EDULL;
constexpr int kPrepBlock = 256;
constexpr uint32_t kPrepWaves = 8192;  // fixed: the extraction order does not depend on it
constexpr uint32_t kMaxCells = GN2V_BLOCK_MAX_CELLS;   // parts x slices (resident cells at 100 M nodes: 1 925 x 256)
constexpr uint32_t kMaxGroupCells = GN2V_BLOCK_MAX_GROUP_CELLS;  // one extraction group (LDS histogram)
// negatives' stream of a cell: draw(draw(key, block_id), cell) -- one stream per (block, cell),
// whatever the number of cells (cell_stream_key; the oracle restates it)
constexpr uint32_t kMaxRecord = 32;
// A run -- consecutive pairs of one centre inside a record, trained against ONE copy of the central
// row, their gradients summed -- is at most this long; a longer stretch of equal centres is cut
// into several runs, each starting from the row the previous one left.  (A run is a mini-batch on
// the centre's row: 32 pairs x (1 + k) samples summed at one stale value overshoot when a small
// graph fills every record with one centre -- link AUROC 0.985 -> 0.93 with 8 centre stripes on
// BA 200 k.)  The oracle restates it (O_MAX_RUN).
constexpr uint32_t kMaxRun = 16;
// draws of a cell-local negative before it is given up (the context or the centre every time)
constexpr uint32_t kNegAttempts = 8;
// records whose runs are at least this many per 100 pairs are trained pair per group
constexpr uint32_t kPpgMinPct = 75;
constexpr uint32_t kCursorStep = 64;  // u64 words between the ticket cursors of two slices
constexpr uint32_t kTicket = 4;       // records a wave takes per ticket (one returning atomic)
// "Hot" contextual rows: the rows of a cell with the highest in-degrees -- the targets of the
// degree-proportional negatives, read-modify-written by so many waves at once that plain stores
// keep a fraction of a percent of their updates.  A cell flags up to kHotMax of them
// (gn2v_block_alias: one bit in the alias-table entries, one in the pair words, a slot number per
// row); a workgroup accumulates the updates of a flagged row in LDS (ds_add_f32: exact inside the
// workgroup) and hands the sum to the row with f32 atomics every few updates: no add is lost,
// at the speed of the stores (SURVEY section 7 "hard parts": LDS combine before the HBM atomic).
// In the staged sample lists a hot row is kHubBit | slot.
constexpr uint32_t kHubBit = 0x80000000u;
constexpr uint32_t kHotMax = GN2V_BLOCK_HOT_MAX;

// Division by a number that is the same for a whole launch (parts, slices, world): a 32-bit
// integer division costs ~35 vector instructions on gfx950, and the extraction does three per
// walk position and pass.  Granlund-Montgomery: q = (mulhi(x, m) + x) >> s for every 32-bit x
// (the sum in 33 bits: t + ((x - t) >> 1) >> (s - 1)); set up on the host (FastDiv::of).
struct FastDiv {
    uint32_t m, s, d;
    static FastDiv of(uint32_t d) {
        FastDiv f{0, 0, d};
        if (d <= 1) return f;  // s == 0: q = x (d == 1)
        uint32_t l = 0;
        while ((1ull << l) < d) ++l;
        f.s = l;
        f.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
        return f;
    }
    __device__ __forceinline__ uint32_t div(uint32_t x) const {
        if (s == 0) return x;
        const uint32_t t = __umulhi(x, m);
        return (t + ((x - t) >> 1)) >> (s - 1);
    }
};

struct BlockPlan {
    uint32_t world, rank, parts, slices;
    FastDiv dparts, dslices, dworld;
    uint32_t L, window, min_dist, record;
    uint32_t row_bits;  // bits of the centre row in a pair word
    uint32_t ctx_bits;  // bits below it: the context row inside its cell + the hot flag on top
    uint32_t flags;  // kFlagDownsample: centres are thinned at extraction (needs the graph)
    uint32_t hot_rows;  // rows a cell flags as hot (0: none)
};

// --------------------------------------------------------------------------------------------
// Negative sampling inside a cell, proportional to the degree (use_scale_free_distribution,
// node2vec_skipgram.py:101-102: a uniform random edge's endpoint): one Walker alias table per cell,
// 8 B per row (threshold on a 2^32 scale | alias row << 32), built once per graph and plan.  A draw
// costs one 8 B read from a table that stays L2 resident (39 k rows = 312 KB per cell on the bench
// graph) -- a pool of edge endpoints, the first form of this sampler, was 4 B per directed edge:
// 3 MB per cell of uniformly hit lines competing with the embedding rows for the 4 MB of L2.
// All arithmetic is integer (weights = in-degrees), so the oracle builds identical tables.
// --------------------------------------------------------------------------------------------
__device__ __host__ __forceinline__ uint64_t stripe_count(uint64_t n, uint64_t first,
                                                          uint64_t stride) {
    return n > first ? (n - first + stride - 1) / stride : 0;
}

// floor(w * 2^32 / D) for w < D < 2^48, in two 16-bit steps (no 128-bit division on the device)
__device__ __host__ __forceinline__ unsigned long long scaled_threshold(unsigned long long w,
                                                                        unsigned long long D) {
    const unsigned long long q1 = (w << 16) / D, r1 = (w << 16) % D;
    return (q1 << 16) | ((r1 << 16) / D);
}

static __global__ void indegree_kernel(const uint32_t *__restrict__ col, uint64_t n_edges,
                                       uint32_t *__restrict__ indeg) {
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges;
         e += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&indeg[col[e]], 1u);
}

// cell_rows[c] = first table entry of cell c (cells in order, rows of a cell in order), in closed
// form: with n = q parts + r the parts p < r hold q + 1 rows, and a part of R = q2 slices + r2
// rows gives its slices s < r2 one row more -- the sums of stripe_count the serial loop of rounds
// 2-5 ran (5 ms a round on the bench graph, 52 ms at 100 M nodes: one thread, 454 k divisions).
static __global__ void cell_rows_kernel(uint64_t n_nodes, uint32_t parts, uint32_t slices,
                                        unsigned long long *__restrict__ cell_rows) {
    const uint64_t cells = (uint64_t)parts * slices;
    const uint64_t q = n_nodes / parts, r = n_nodes % parts;
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c <= cells;
         c += (uint64_t)gridDim.x * blockDim.x) {
        if (c == cells) {
            cell_rows[c] = n_nodes;
            continue;
        }
        const uint64_t p = c / slices, sl = c - p * slices;
        const uint64_t part_rows = q + (p < r ? 1 : 0);
        const uint64_t q2 = part_rows / slices, r2 = part_rows % slices;
        cell_rows[c] = p * q + (p < r ? p : r) + sl * q2 + (sl < r2 ? sl : r2);
    }
}

// Vose's construction, one thread per cell, integer arithmetic: row i of the cell has weight
// p_i = indeg_i * n against the cell total D; "small" rows (p < D) keep threshold p * 2^32 / D and
// borrow the rest from a "large" row.  work: u64 weight[n_nodes] | u32 stack[n_nodes].
// Entry of row i: bit 0 = row i is hot, bits 1..31 = threshold (its lowest bit dropped), bits
// 32..62 = alias row, bit 63 = the alias row is hot.
// Hot rows of a cell: its `hot_rows` rows of highest in-degree (>= 1; ties: the lower row first).
// hot_list[cell][s] = row inside the cell of slot s (kSentinel beyond the cell's count), slots in
// decreasing order of in-degree, so a launch whose LDS holds fewer slots takes a prefix;
// hot_slot[cell_rows[cell] + row] = slot (0xFF: not hot; preset by the host); hub_bits: one bit
// per node id (for the extraction).
static __global__ void alias_kernel(const uint32_t *__restrict__ indeg, uint64_t n_nodes,
                                    uint32_t parts, uint32_t slices,
                                    const unsigned long long *__restrict__ cell_rows,
                                    unsigned long long *__restrict__ table,
                                    unsigned long long *__restrict__ weight,
                                    uint32_t *__restrict__ stack,
                                    uint32_t *__restrict__ hub_bits, uint32_t hot_rows,
                                    uint32_t *__restrict__ hot_list,
                                    uint8_t *__restrict__ hot_slot,
                                    const uint32_t *__restrict__ inv) {
    const uint32_t cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= parts * slices) return;
    const uint32_t part = cell / slices, slice = cell - part * slices;
    const uint64_t lo = cell_rows[cell], n = cell_rows[cell + 1] - lo;
    // (the hot tables are optional for plans without hot rows: resident cells)
    uint32_t *hl = hot_list ? hot_list + (size_t)cell * kHotMax : nullptr;
    if (hl)
        for (uint32_t s = 0; s < kHotMax; ++s) hl[s] = kSentinel;
    if (!hl || !hot_slot || !hub_bits) hot_rows = 0;
    if (n == 0) return;
    unsigned long long *w = weight + lo, *t = table + lo;
    uint32_t *st = stack + lo;
    // the node whose placed id is row i of this cell (inv: the round's placement, or identity)
    auto node_of = [&](uint64_t i) -> uint64_t {
        const uint64_t xp = (slice + (uint64_t)slices * i) * parts + part;
        return inv ? (uint64_t)inv[xp] : xp;
    };
    unsigned long long D = 0;
    uint32_t n_hot = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t d = indeg[node_of(i)];
        D += d;
        if (d != 0 && hot_rows != 0 &&
            (n_hot < hot_rows || d > indeg[node_of(hl[n_hot - 1])])) {  // insertion, stable
            uint32_t pos = n_hot < hot_rows ? n_hot++ : n_hot - 1;
            while (pos > 0 && indeg[node_of(hl[pos - 1])] < d) {
                hl[pos] = hl[pos - 1];
                --pos;
            }
            hl[pos] = (uint32_t)i;
        }
    }
    for (uint32_t s = 0; s < n_hot; ++s) {
        hot_slot[lo + hl[s]] = (uint8_t)s;
        const uint64_t x = node_of(hl[s]);
        atomicOr(&hub_bits[x >> 5], 1u << (x & 31));
    }
    uint64_t n_small = 0, n_large = 0;  // small stack grows from st[0], large from st[n - 1]
    auto hot = [&](uint64_t i) -> unsigned long long {
        return hot_rows != 0 && hot_slot[lo + i] != kNoSlot;
    };
    for (uint64_t i = 0; i < n; ++i) {
        const unsigned long long p = (unsigned long long)indeg[node_of(i)] * n;
        w[i] = p;
        if (D == 0 || p >= D)
            st[n - 1 - n_large++] = (uint32_t)i;
        else
            st[n_small++] = (uint32_t)i;
    }
    while (n_small && n_large) {
        const uint32_t sidx = st[--n_small];
        const uint32_t lidx = st[n - n_large];  // top of the large stack
        t[sidx] = (hot(lidx) << 63) | ((unsigned long long)lidx << 32) |
                  (scaled_threshold(w[sidx], D) & ~1ull) | hot(sidx);
        const unsigned long long pl = w[lidx] + w[sidx] - D;
        w[lidx] = pl;
        if (pl < D) {  // the large row became small: move it over
            --n_large;
            st[n_small++] = lidx;
        }
    }
    while (n_large) {
        const uint32_t i = st[n - n_large--];
        t[i] = (hot(i) << 63) | ((unsigned long long)i << 32) | 0xFFFFFFFEull | hot(i);
    }
    while (n_small) {  // only through rounding
        const uint32_t i = st[--n_small];
        t[i] = (hot(i) << 63) | ((unsigned long long)i << 32) | 0xFFFFFFFEull | hot(i);
    }
}

// --------------------------------------------------------------------------------------------
// Placement of a round: WHERE a node's contextual row is trained changes from
// round to round, so that over a fit the negatives a context meets -- drawn inside its cell --
// range over the graph (node2vec_skipgram.py:101-102) instead of over one fixed set of ~200
// cell-mates.  A seeded permutation of the node ids inside their residue classes modulo
// `classes` (= parts when the parts travel between ranks: a row never changes its part; 1 on
// one GPU): keys (class << 40 | hash(round key, x) >> 24) are sorted (stable radix sort: ties
// in node order), and the j-th node of class c receives the placed id c + classes * j.
// place[x] = x', inv[x'] = x.  Everything downstream works on placed ids (part = x' % parts,
// row = x' / parts, slice = row % slices); the tables stay where they are and the resident
// kernel reaches a row through inv.  The oracle restates it (o_block_placement): bit-equal.
// --------------------------------------------------------------------------------------------
constexpr uint64_t kTagPlace = 0x91ACE5EED5EED5EDULL;

// LPT order of a resident launch: key = ~pairs of the cell (ascending sort = heaviest first)
static __global__ void cell_order_keys_kernel(const unsigned long long *__restrict__ cell_offsets,
                                              uint32_t first_cell, uint32_t n, uint32_t *keys,
                                              uint32_t *vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long sz = cell_offsets[first_cell + i + 1] - cell_offsets[first_cell + i];
    keys[i] = ~(uint32_t)(sz > 0xFFFFFFFFull ? 0xFFFFFFFFull : sz);
    vals[i] = i;
}

static __global__ void place_keys_kernel(uint64_t n_nodes, uint32_t classes, uint64_t pkey,
                                         unsigned long long *__restrict__ keys,
                                         uint32_t *__restrict__ vals) {
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n_nodes;
         x += (uint64_t)gridDim.x * blockDim.x) {
        keys[x] = ((unsigned long long)(x % classes) << 40) | (draw(pkey, x) >> 24);
        vals[x] = (uint32_t)x;
    }
}

static __global__ void place_scatter_kernel(uint64_t n_nodes, uint32_t classes,
                                            const uint32_t *__restrict__ sorted,
                                            uint32_t *__restrict__ place,
                                            uint32_t *__restrict__ inv) {
    const uint64_t per = n_nodes / classes, extra = n_nodes % classes;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_nodes;
         j += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t x = sorted[j];
        const uint64_t c = x % classes;
        const uint64_t start = c * per + (c < extra ? c : extra);
        const uint32_t xp = (uint32_t)(c + (uint64_t)classes * (j - start));
        place[x] = xp;
        inv[xp] = x;
    }
}

// out[i] = place[walks[i]] (the walks with their nodes' placed ids: what the extraction reads
// for the CONTEXT side -- one gather per walk position and round instead of one per pass)
static __global__ void place_walks_kernel(const uint32_t *__restrict__ place,
                                          const uint32_t *__restrict__ walks, uint64_t n,
                                          uint32_t *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t x = walks[i];
        out[i] = x == kSentinel ? kSentinel : place[x];
    }
}

// --------------------------------------------------------------------------------------------
// Pair extraction.  Wave w owns the contiguous chunk of walks [w * chunk, (w + 1) * chunk) and
// emits their pairs in walk / position / slot order; WRITE = false counts (per wave and per cell),
// WRITE = true writes at the wave's scanned offset.
// --------------------------------------------------------------------------------------------
struct ExtractArgs {
    GraphView g;
    BlockPlan p;
    const uint32_t *walks;
    uint64_t n_walks;
    uint64_t ekey;        // for the centre down-sampling draws (same as the walk-ordered kernels)
    uint64_t first_walk;  // id of walks[0]
    unsigned long long *wave_counts;  // [kPrepWaves]: counts out (count pass), offsets in (write)
    unsigned long long *cell_counts;  // [cells] (count pass); nullptr: the group has more cells than
                                      // LDS counters -- only the waves' totals are counted, and
                                      // the cell offsets are read off the sorted words afterwards
                                      // (cell_offsets_from_sorted_kernel)
    const uint32_t *placed;  // the walks with placed node ids (place_walks_kernel), or nullptr:
                             // a context's cell follows from its node id itself
    const uint32_t *hub_bits;         // one bit per node (gn2v_block_alias), or nullptr
    unsigned long long *pairs;        // [n_pairs] pair words (write pass)
    uint32_t part_lo, part_n;         // only contexts in the parts part_lo, part_lo + 1, ...
                                      // (part_n of them, cyclic) are emitted
};

__device__ __forceinline__ bool keep_centre_at(const GraphView &g, uint64_t wkey, uint32_t i,
                                               uint32_t c) {
    const uint64_t deg = g.row_ptr[c + 1] - g.row_ptr[c];
    if (deg == 0) return true;
    const uint64_t r32 = draw(wkey ^ kTagDown, i) >> 32;
    const uint64_t x = r32 * deg;
    const uint64_t lhs_lo = x * g.n_nodes, lhs_hi = mulhi64(x, g.n_nodes);
    const uint64_t rhs_lo = g.n_edges << 32, rhs_hi = g.n_edges >> 32;
    return lhs_hi < rhs_hi || (lhs_hi == rhs_hi && lhs_lo < rhs_lo);
}

template <bool WRITE>
__global__ __launch_bounds__(kPrepBlock) void block_extract_kernel(ExtractArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t waves_per_block = kPrepBlock / 64;
    // cells of the group of parts being extracted, numbered (part - part_lo) mod parts, slice
    const uint32_t cells = a.part_n * a.p.slices;
    const uint32_t L = a.p.L, w = a.p.window, w2 = 2 * a.p.window;
    // per wave: the walk, per position its context cell (kSentinel: not in this group of parts)
    // and its context row inside the cell (| hot flag), the compacted own positions
    uint32_t *s_walk = smem + wave * 4 * L;
    uint32_t *s_cell = s_walk + L;
    uint32_t *s_loc = s_cell + L;
    uint32_t *s_own = s_loc + L;
    unsigned int *s_hist = smem + waves_per_block * 4 * L;  // [cells], shared by the block
    const uint64_t gw = (uint64_t)blockIdx.x * waves_per_block + wave;
    const uint64_t chunk = (a.n_walks + kPrepWaves - 1) / kPrepWaves;
    const uint64_t b0 = gw * chunk;
    const uint64_t b1 = b0 + chunk < a.n_walks ? b0 + chunk : a.n_walks;
    if constexpr (!WRITE) {
        if (a.cell_counts)
            for (uint32_t c = threadIdx.x; c < cells; c += kPrepBlock) s_hist[c] = 0;
        __syncthreads();
    }
    unsigned long long base = 0;
    if constexpr (WRITE) base = a.wave_counts[gw];
    unsigned long long total = 0;
    const uint64_t lt_mask = (1ULL << lane) - 1;
    const uint32_t hub_shift = a.p.ctx_bits - 1;
    for (uint64_t b = b0; b < b1; ++b) {
        wave_sync();
        uint32_t Le = L;
        for (uint32_t t = lane; t < L; t += 64) {
            const uint32_t x = a.walks[b * L + t];
            s_walk[t] = x;
            uint32_t cell = kSentinel, loc = 0;
            if (x == kSentinel) {
                Le = min(Le, t);
            } else {
                const uint32_t xp = a.placed ? a.placed[b * L + t] : x;
                const uint32_t row = a.p.dparts.div(xp), part = xp - row * a.p.parts;
                const uint32_t rel = part >= a.part_lo ? part - a.part_lo
                                                       : part + a.p.parts - a.part_lo;
                if (rel < a.part_n) {
                    loc = a.p.dslices.div(row);
                    cell = part * a.p.slices + (row - loc * a.p.slices);
                    if (WRITE && a.hub_bits && ((a.hub_bits[x >> 5] >> (x & 31)) & 1u))
                        loc |= 1u << hub_shift;
                }
            }
            s_cell[t] = cell;
            s_loc[t] = loc;
        }
        for (int off = 32; off > 0; off >>= 1) Le = min(Le, (uint32_t)__shfl_xor(Le, off));
        wave_sync();
        const uint64_t wkey = draw(a.ekey, a.first_walk + b);
        // positions whose centre this rank owns (1 / world of them), compacted in walk order
        uint32_t n_own = 0;
        for (uint32_t i0 = 0; i0 < Le; i0 += 64) {
            const uint32_t i = i0 + lane;
            bool own = false;
            if (i < Le) {
                const uint32_t c = s_walk[i];
                own = c - a.p.dworld.div(c) * a.p.world == a.p.rank &&
                      (!(a.p.flags & kFlagDownsample) || keep_centre_at(a.g, wkey, i, c));
            }
            const uint64_t m = __ballot(own);
            if (own) s_own[n_own + __popcll(m & lt_mask)] = i;
            n_own += __popcll(m);
        }
        wave_sync();
        const uint32_t n_slots = n_own * w2;
        for (uint32_t t0 = 0; t0 < n_slots; t0 += 64) {
            const uint32_t t = t0 + lane;
            bool valid = false;
            unsigned long long word = 0;
            uint32_t cell = 0;
            if (t < n_slots) {
                const uint32_t idx = t / w2, slot = t - idx * w2;
                const uint32_t i = s_own[idx];
                const int64_t j = slot < w ? (int64_t)i - w + slot : (int64_t)i + 1 + (slot - w);
                if (j >= 0 && j < (int64_t)Le) {
                    const uint32_t dist = (uint32_t)(j > (int64_t)i ? j - i : i - j);
                    cell = s_cell[j];
                    if (dist >= a.p.min_dist && cell != kSentinel) {
                        if constexpr (WRITE)
                            word = ((((unsigned long long)cell << a.p.row_bits) |
                                     a.p.dworld.div(s_walk[i]))
                                    << a.p.ctx_bits) |
                                   s_loc[j];
                        valid = true;
                    }
                }
            }
            const uint64_t mask = __ballot(valid);
            if constexpr (WRITE) {
                if (valid) a.pairs[base + __popcll(mask & lt_mask)] = word;
                base += __popcll(mask);
            } else {
                if (valid && a.cell_counts) {
                    const uint32_t part = a.p.dslices.div(cell), sl = cell - part * a.p.slices;
                    const uint32_t rel = part >= a.part_lo ? part - a.part_lo
                                                           : part + a.p.parts - a.part_lo;
                    atomicAdd(&s_hist[rel * a.p.slices + sl], 1u);
                }
                total += __popcll(mask);
            }
        }
    }
    if constexpr (!WRITE) {
        if (lane == 0) a.wave_counts[gw] = total;
        __syncthreads();
        for (uint32_t c = threadIdx.x; a.cell_counts && c < cells; c += kPrepBlock)
            if (s_hist[c]) {
                const uint32_t rel = c / a.p.slices, sl = c - rel * a.p.slices;
                uint32_t part = a.part_lo + rel;
                if (part >= a.p.parts) part -= a.p.parts;
                atomicAdd(&a.cell_counts[part * a.p.slices + sl], (unsigned long long)s_hist[c]);
            }
    }
}

// The same extraction for walks of at most 128 nodes and windows of at most 31 (the reference's
// defaults: 128 and 5), without visiting every window slot of every centre.  The walk's
// positions whose context cell lies in the group, and its own centres, are two 128-bit masks; a
// centre's pairs are the set bits of (window of i) & in-group, so
//   count(i)   = popc(window(i) & in_group)                      own centres only
//   rank(i, j) = sum_{i' < i} count(i') + popc(window(i) & in_group & below(j))
// is where the pair (i, j) stands among the walk's pairs in (centre position, context position)
// order -- the order block_extract_kernel emits them in.  The count pass adds popc(window(j) &
// own) to the cell of every in-group position j (one LDS atomic per position, not per pair); the
// write pass runs over (in-group position, window offset) and stores every pair at its rank.
// Work follows the pairs a group EMITS: a group is 1 / 6 of the parts on the bench graph, 1 / 61
// at 100 M nodes, and the slot-by-slot kernel visited all 1 280 slots of a walk for each.
struct Mask128 {
    uint64_t lo, hi;
};
__device__ __forceinline__ Mask128 window_mask(uint32_t p, uint32_t w, uint32_t md) {
    Mask128 m{0, 0};
    if (md > w) return m;
    const uint64_t side = (1ull << (w - md + 1)) - 1;
    const uint64_t pat = side | (side << (w + md));  // bit k: offset k - w, md <= |offset| <= w
    if (p >= w) {
        const uint32_t sh = p - w;
        if (sh == 0) {
            m.lo = pat;
        } else if (sh < 64) {
            m.lo = pat << sh;
            m.hi = pat >> (64 - sh);
        } else {
            m.hi = pat << (sh - 64);
        }
    } else {
        m.lo = pat >> (w - p);
    }
    return m;
}
__device__ __forceinline__ Mask128 below_mask(uint32_t j) {
    Mask128 m;
    m.lo = j < 64 ? (1ull << j) - 1 : ~0ull;
    m.hi = j <= 64 ? 0 : (j >= 128 ? ~0ull : (1ull << (j - 64)) - 1);
    return m;
}
__device__ __forceinline__ uint32_t popc_and(const Mask128 &a, const Mask128 &b) {
    return (uint32_t)(__popcll(a.lo & b.lo) + __popcll(a.hi & b.hi));
}

template <bool WRITE>
__global__ __launch_bounds__(kPrepBlock) void block_extract_fast_kernel(ExtractArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t waves_per_block = kPrepBlock / 64;
    const uint32_t cells = a.part_n * a.p.slices;
    const uint32_t L = a.p.L, w = a.p.window, w2 = 2 * a.p.window, md = a.p.min_dist;
    uint32_t *s_walk = smem + wave * 5 * L;
    uint32_t *s_cell = s_walk + L;
    uint32_t *s_loc = s_cell + L;
    uint32_t *s_pre = s_loc + L;   // pairs of the walk before centre position p
    uint32_t *s_j = s_pre + L;     // the in-group positions, in walk order
    unsigned int *s_hist = smem + waves_per_block * 5 * L;
    const uint64_t gw = (uint64_t)blockIdx.x * waves_per_block + wave;
    const uint64_t chunk = (a.n_walks + kPrepWaves - 1) / kPrepWaves;
    const uint64_t b0 = gw * chunk;
    const uint64_t b1 = b0 + chunk < a.n_walks ? b0 + chunk : a.n_walks;
    if constexpr (!WRITE) {
        if (a.cell_counts)
            for (uint32_t c = threadIdx.x; c < cells; c += kPrepBlock) s_hist[c] = 0;
        __syncthreads();
    }
    unsigned long long base = 0;
    if constexpr (WRITE) base = a.wave_counts[gw];
    unsigned long long total = 0;
    const uint64_t lt_mask = (1ULL << lane) - 1;
    const uint32_t hub_shift = a.p.ctx_bits - 1;
    for (uint64_t b = b0; b < b1; ++b) {
        wave_sync();
        uint32_t Le = L;
        for (uint32_t t = lane; t < L; t += 64) {
            const uint32_t x = a.walks[b * L + t];
            s_walk[t] = x;
            uint32_t cell = kSentinel, loc = 0;
            if (x == kSentinel) {
                Le = min(Le, t);
            } else {
                const uint32_t xp = a.placed ? a.placed[b * L + t] : x;
                const uint32_t row = a.p.dparts.div(xp), part = xp - row * a.p.parts;
                const uint32_t rel = part >= a.part_lo ? part - a.part_lo
                                                       : part + a.p.parts - a.part_lo;
                if (rel < a.part_n) {
                    loc = a.p.dslices.div(row);
                    cell = part * a.p.slices + (row - loc * a.p.slices);
                    if (WRITE && a.hub_bits && ((a.hub_bits[x >> 5] >> (x & 31)) & 1u))
                        loc |= 1u << hub_shift;
                }
            }
            s_cell[t] = cell;
            s_loc[t] = loc;
        }
        for (int off = 32; off > 0; off >>= 1) Le = min(Le, (uint32_t)__shfl_xor(Le, off));
        wave_sync();
        const uint64_t wkey = draw(a.ekey, a.first_walk + b);
        // the two masks: lane l speaks for the positions l and l + 64
        Mask128 in_group, own;
        bool g0, g1, o0 = false, o1 = false;
        {
            const uint32_t p0 = lane, p1 = lane + 64;
            g0 = p0 < Le && s_cell[p0] != kSentinel;
            g1 = p1 < Le && s_cell[p1] != kSentinel;
            if (p0 < Le) {
                const uint32_t c = s_walk[p0];
                o0 = c - a.p.dworld.div(c) * a.p.world == a.p.rank &&
                     (!(a.p.flags & kFlagDownsample) || keep_centre_at(a.g, wkey, p0, c));
            }
            if (p1 < Le) {
                const uint32_t c = s_walk[p1];
                o1 = c - a.p.dworld.div(c) * a.p.world == a.p.rank &&
                     (!(a.p.flags & kFlagDownsample) || keep_centre_at(a.g, wkey, p1, c));
            }
            in_group.lo = __ballot(g0);
            in_group.hi = __ballot(g1);
            own.lo = __ballot(o0);
            own.hi = __ballot(o1);
        }
        if ((in_group.lo | in_group.hi) == 0 || (own.lo | own.hi) == 0) continue;
        if constexpr (!WRITE) {
            // per in-group position: the own centres that have it in their window
            uint32_t n0 = 0, n1 = 0;
            if (g0) n0 = popc_and(window_mask(lane, w, md), own);
            if (g1) n1 = popc_and(window_mask(lane + 64, w, md), own);
            if ((n0 | n1) && a.cell_counts) {
                const uint32_t slices = a.p.slices;
                auto gidx = [&](uint32_t cell) {
                    const uint32_t part = a.p.dslices.div(cell), sl = cell - part * slices;
                    const uint32_t rel = part >= a.part_lo ? part - a.part_lo
                                                           : part + a.p.parts - a.part_lo;
                    return rel * slices + sl;
                };
                if (n0) atomicAdd(&s_hist[gidx(s_cell[lane])], n0);
                if (n1) atomicAdd(&s_hist[gidx(s_cell[lane + 64])], n1);
            }
            uint32_t n = n0 + n1;
            for (int off = 32; off > 0; off >>= 1) n += (uint32_t)__shfl_xor(n, off);
            total += n;
        } else {
            // pairs before each centre position: an exclusive scan of count(p) over the walk
            uint32_t c0 = o0 ? popc_and(window_mask(lane, w, md), in_group) : 0;
            uint32_t c1 = o1 ? popc_and(window_mask(lane + 64, w, md), in_group) : 0;
            uint32_t x0 = c0, x1 = c1;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t y0 = (uint32_t)__shfl_up(x0, off), y1 = (uint32_t)__shfl_up(x1, off);
                if (lane >= off) {
                    x0 += y0;
                    x1 += y1;
                }
            }
            const uint32_t t0 = (uint32_t)__shfl(x0, 63), t1 = (uint32_t)__shfl(x1, 63);
            if ((uint32_t)lane < L) s_pre[lane] = x0 - c0;
            if ((uint32_t)lane + 64 < L) s_pre[lane + 64] = t0 + x1 - c1;
            // One lane per (position, window slot).  The positions are the in-group contexts, or
            // -- when the rank owns fewer centres than the group holds contexts (several ranks: a
            // rank of 8 owns an eighth of the positions and a group holds half of them) -- the
            // own centres: the word's place follows from (centre, context) alone, so either
            // enumeration writes the same words to the same places, and the shorter one leaves
            // fewer lanes without a pair (write pass of a rank of 8: 9.8 -> 5.3 ms a group of 8 parts).
            const bool by_centre = (uint64_t)a.p.world * a.part_n > a.p.parts;
            const Mask128 &listed = by_centre ? own : in_group;
            const uint32_t nj0 = (uint32_t)__popcll(listed.lo);
            if (by_centre ? o0 : g0) s_j[__popcll(listed.lo & lt_mask)] = lane;
            if (by_centre ? o1 : g1) s_j[nj0 + __popcll(listed.hi & lt_mask)] = lane + 64;
            const uint32_t nj = nj0 + (uint32_t)__popcll(listed.hi);
            wave_sync();
            const uint32_t n_slots = nj * w2;
            for (uint32_t s0 = 0; s0 < n_slots; s0 += 64) {
                const uint32_t t = s0 + lane;
                if (t < n_slots) {
                    const uint32_t jx = t / w2, slot = t - jx * w2;
                    const uint32_t q = s_j[jx];
                    // the other position at offset -w .. -1, 1 .. w of q
                    const int64_t o = slot < w ? (int64_t)q - w + slot : (int64_t)q + 1 + (slot - w);
                    const uint32_t dist = slot < w ? w - slot : slot - w + 1;
                    if (o >= 0 && o < (int64_t)Le && dist >= md) {
                        const uint32_t ou = (uint32_t)o;
                        const Mask128 &wanted = by_centre ? in_group : own;
                        const bool hit = ou < 64 ? (wanted.lo >> ou) & 1 : (wanted.hi >> (ou - 64)) & 1;
                        if (hit) {
                            const uint32_t iu = by_centre ? q : ou, j = by_centre ? ou : q;
                            Mask128 wi = window_mask(iu, w, md);
                            const Mask128 bj = below_mask(j);
                            wi.lo &= in_group.lo & bj.lo;
                            wi.hi &= in_group.hi & bj.hi;
                            const uint32_t rank =
                                s_pre[iu] + (uint32_t)(__popcll(wi.lo) + __popcll(wi.hi));
                            a.pairs[base + rank] =
                                ((((unsigned long long)s_cell[j] << a.p.row_bits) |
                                  a.p.dworld.div(s_walk[iu]))
                                 << a.p.ctx_bits) |
                                s_loc[j];
                        }
                    }
                }
            }
            base += t0 + t1;
        }
    }
    if constexpr (!WRITE) {
        if (lane == 0) a.wave_counts[gw] = total;
        __syncthreads();
        for (uint32_t c = threadIdx.x; a.cell_counts && c < cells; c += kPrepBlock)
            if (s_hist[c]) {
                const uint32_t rel = c / a.p.slices, sl = c - rel * a.p.slices;
                uint32_t part = a.part_lo + rel;
                if (part >= a.p.parts) part -= a.p.parts;
                atomicAdd(&a.cell_counts[part * a.p.slices + sl], (unsigned long long)s_hist[c]);
            }
    }
}

// wave_counts -> exclusive offsets in place, total -> total_out[0]; cell_counts -> cell_offsets
__global__ __launch_bounds__(1024) void block_scan_kernel(unsigned long long *wave_counts,
                                                          const unsigned long long *cell_counts,
                                                          uint32_t cells,
                                                          unsigned long long *cell_offsets) {
    __shared__ unsigned long long part[1024];
    constexpr uint32_t per = kPrepWaves / 1024;
    unsigned long long loc[per];
    unsigned long long sum = 0;
    for (uint32_t e = 0; e < per; ++e) {
        loc[e] = sum;
        sum += wave_counts[threadIdx.x * per + e];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    // the cells: thread t sums the cells [t * chunk, (t + 1) * chunk)
    __shared__ unsigned long long cpart[1024];
    const uint32_t chunk = (cells + 1023) / 1024;
    const uint32_t c0 = min(cells, threadIdx.x * chunk), c1 = min(cells, c0 + chunk);
    unsigned long long csum = 0;
    for (uint32_t c = c0; cell_counts && c < c1; ++c) csum += cell_counts[c];
    cpart[threadIdx.x] = csum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0, off = 0;
        for (uint32_t i = 0; i < 1024; ++i) {
            const unsigned long long v = part[i], cv = cpart[i];
            part[i] = run;
            cpart[i] = off;
            run += v;
            off += cv;
        }
        // (no cell counters: the total of the waves; the other offsets follow the sort)
        cell_offsets[cells] = cell_counts ? off : run;
    }
    __syncthreads();
    for (uint32_t e = 0; e < per; ++e) wave_counts[threadIdx.x * per + e] = part[threadIdx.x] + loc[e];
    unsigned long long off = cpart[threadIdx.x];
    for (uint32_t c = c0; cell_counts && c < c1; ++c) {
        cell_offsets[c] = off;
        off += cell_counts[c];
    }
}

// cell_offsets[c] = first sorted pair word whose cell is >= c, c = 0 .. cells (the oracle's
// o_block_cell_offsets): what the counters of the counting pass give for groups that have them
static __global__ void cell_offsets_from_sorted_kernel(const unsigned long long *__restrict__ words,
                                                       unsigned long long n, uint32_t cell_shift,
                                                       uint32_t cells,
                                                       unsigned long long *__restrict__ cell_offsets) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > cells) return;
    unsigned long long lo = 0, hi = n;
    while (lo < hi) {
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        if ((words[mid] >> cell_shift) < c)
            lo = mid + 1;
        else
            hi = mid;
    }
    cell_offsets[c] = lo;
}

// --------------------------------------------------------------------------------------------
// Training kernel over the sorted pairs of one part.
// --------------------------------------------------------------------------------------------
struct BlockArgs {
    GraphView g;
    BlockPlan p;
    const unsigned long long *pairs;  // sorted pair words (see the head of this file)
    float *const *part_ptrs;  // resident cells, a group of parts per launch: the rows of part p
                              // (blockIdx.y counts from `part`); nullptr: `context` is the part's
    const unsigned long long *cell_offsets;  // [cells + 1] into pairs
    const unsigned long long *alias;      // alias tables (threshold | alias << 32), or nullptr:
                                          // negatives uniform over the rows of the cell
    const unsigned long long *cell_rows;  // [cells + 1]: first table entry of every cell
    const uint32_t *hot_list;  // [cells][kHotMax]: row inside its cell of hot slot s (or nullptr)
    const uint8_t *hot_slot;   // [n_nodes], by cell_rows[cell] + row inside the cell: slot | 0xFF
    float *central;    // this rank's central partition  [rows][cld]
    uint64_t cld;      // floats between its rows (ld, or world * ld inside the whole table)
    float *context;    // the resident context part      [rows][xld]
    uint64_t xld;      // floats between its rows (ld, or parts * ld inside the whole table)
    // resident cells under a round's placement: row r of cell (part, slice) is
    // the contextual row of node x = inv[(slice + slices r) parts + part] -- row x of ctx_table
    // (the whole table in node order: one GPU) or, ctx_table == nullptr, row x / parts of
    // `context` (placements that keep the classes modulo parts: x % parts == part)
    const uint32_t *inv;
    float *ctx_table;
    // resident cells: the launch's workgroups take its cells heaviest first -- workgroup i (in
    // dispatch order) trains cell order[i] of the launch (part offset * slices + slice); nullptr:
    // in index order.  The last workgroups of a launch then finish light cells, not a hub's.
    const uint32_t *order;
    unsigned long long *cursors;  // record tickets of the part's cells, one per slice, kCursorStep
                                  // words apart (zeroed per launch)
    unsigned long long *counters;
    uint64_t n_nodes;
    uint64_t ekey;
    uint64_t block_id;
    uint32_t part;
    uint32_t sweep;  // 1: second launch -- every workgroup serves every cell's leftover records
    uint32_t xcds;   // XCDs the workgroups are spread over (0 = unknown)
    uint32_t central_store;  // 1: a centre's only run in a cell stores row + gradient instead of
                             // adding the gradient with atomics (GN2V_TRAIN_CENTRAL_STORE)
    uint32_t hot_n;      // hot slots the LDS of this launch holds (0: hot rows are ordinary rows)
    uint32_t hot_mask;   // a slot's pending sum goes to its row after ~hot_mask + 1 updates (2^j - 1)
    uint32_t reread;     // 1: a row is read a second time right before its stores (small graphs)
    uint32_t k, ld, flags;
    float lr, clip;
};

__device__ __forceinline__ float *sample_base(const BlockArgs &a, float *table, uint32_t row) {
    return table + (uint64_t)row * a.xld;
}

__device__ __forceinline__ uint64_t cell_stream_key(uint64_t ekey, uint64_t block_id,
                                                    uint32_t cell) {
    return draw(draw(mix64(ekey ^ kTagBlock), block_id), cell);
}

__device__ __forceinline__ uint64_t gcd64(uint64_t a, uint64_t b) {
    while (b) {
        const uint64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

// stride of the record visiting order: close to R / golden ratio, coprime with R
__device__ __forceinline__ uint64_t record_stride(uint64_t R) {
    if (R < 3) return 1;
    uint64_t s = (uint64_t)((double)R * 0.6180339887498949);
    if (s < 1) s = 1;
    while (gcd64(s, R) != 1) ++s;
    return s % R;
}

// ---- hot rows: the workgroup's LDS copies -------------------------------------------------------
// For each of the cell's hot rows a workgroup holds, in LDS, base[s][ld] -- the row as the
// workgroup last saw it in memory -- and delta[s][ld] -- what its waves have added to it since
// they last handed their sum over.  A sample on a hot row never touches memory: it reads base +
// delta, scores, and adds var * u to delta[s] with ds_add_f32 (atomic inside the CU's LDS: the
// waves of the workgroup lose nothing to one another, and each sees the others' updates at once).
// Every ~hot_mask + 1 updates of a slot -- decided by mantissa bits of the update's own
// coefficient: the same in the 16 lanes of a group, no counter -- the group HANDS OVER, alone in
// the workgroup for that slot (an LDS try-lock; a second group that draws the slot meanwhile
// moves on): it takes delta out of the LDS (ds_wrxchg with 0; added to base at once, so the
// workgroup's view is continuous) and adds it to the row in memory with RETURNING f32 atomics --
// the only place where workgroups meet, so every update arrives exactly once --; what they
// return, the row as it was when the sum arrived (the other workgroups' sums and this one's
// earlier ones included), plus the sum is the new base: one round trip.  What is pending when
// the workgroup leaves the cell is handed over then.
// Why the copies: a first form kept only delta and read the row from memory for every sample.
// The atomics drop the row's lines from the L2, so the hottest row of a cell -- 4 % of its
// samples -- became a load on one memory channel for every wave: -14 % on the bench with a single
// hot row.  Why the lock and the row coming back with the hand-over: what a workgroup does not
// see (the other workgroups' pending sums, and their handed-over sums until its own next
// hand-over) acts like a mini-batch on the row -- the restoring force of the loss is computed on
// a stale value -- and a hub row DIVERGES when too many updates are in limbo: measured on BA 1 M
// at learning rate 0.01, ~ 400 unseen updates are fine, ~ 770 blow the tables up to 1e11 (the
// bound is on unseen updates x learning rate; gn2v_block_step picks the hand-over period from
// it).  Hand-overs that overlapped wrote older snapshots over newer ones and kept the view behind
// for good, whatever the period.  The atomics are the device-scope ones: on gfx950 the
// workgroup-scope form compiles to the same global_atomic_add_f32.
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

struct HotLds {
    float *base;
    float *delta;
    uint32_t *grow;  // row inside the part of slot s
    uint32_t *lock;  // 1 while a group hands slot s over
    uint32_t n, ld, mask;
};

__device__ __forceinline__ void lds_add_f32(float *p, float x) {
    (void)__hip_atomic_fetch_add((lds_f32 *)p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ float lds_take_f32(float *p) {
    return __builtin_bit_cast(float, __hip_atomic_exchange((lds_u32 *)p, 0u, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_WORKGROUP));
}

// One 16-lane group hands the pending sum of a slot to its row (lane-contiguous: 64 B per
// instruction and group, the shape in which f32 atomics run 4 x faster than float4-strided).
// REFRESH: under the slot's lock, and the row is read back into base afterwards.
template <int CH, bool REFRESH>
__device__ __forceinline__ void hot_hand_over(float *hb, float *dl, uint32_t *lock, float *row,
                                              int q, uint32_t ld) {
    if constexpr (REFRESH) {
        uint32_t busy = 1;
        if (q == 0)
            busy = __hip_atomic_exchange((lds_u32 *)lock, 1u, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_WORKGROUP);
        // lane 0 of the group decides for its 16 lanes
        const uint64_t free_groups = __ballot(q == 0 && busy == 0);
        if (!((free_groups >> (__lane_id() & 48)) & 1)) return;
    }
    float x[CH * 4];
#pragma unroll
    for (int i = 0; i < CH * 4; ++i) {
        const uint32_t f = i * 16 + q;
        x[i] = 0.f;
        if (f < ld) {
            x[i] = lds_take_f32(dl + f);
            // the view (base + delta) stays what it was until the row is back
            if constexpr (REFRESH) lds_add_f32(hb + f, x[i]);
        }
    }
    if constexpr (!REFRESH) {
#pragma unroll
        for (int i = 0; i < CH * 4; ++i) {
            const uint32_t f = i * 16 + q;
            if (f < ld && x[i] != 0.f) unsafeAtomicAdd(row + f, x[i]);
        }
    } else {
        // returning atomics: the row as it was when this sum arrived -- the other workgroups'
        // sums and this one's earlier ones included -- in the same round trip
        float old[CH * 4];
#pragma unroll
        for (int i = 0; i < CH * 4; ++i) {
            const uint32_t f = i * 16 + q;
            old[i] = f < ld ? unsafeAtomicAdd(row + f, x[i]) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < CH * 4; ++i) {
            const uint32_t f = i * 16 + q;
            if (f < ld) hb[f] = old[i] + x[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (q == 0)
            __hip_atomic_store((lds_u32 *)lock, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// One sample row against the register row u of its centre (a 16-lane group): v = row, var =
// (label - sigmoid(clip(u.v))) * lr, g += var * v, row += var * u -- a read-modify-write of the
// row in memory, or of the workgroup's LDS copy when the row is hot.
template <int CH, int WMX, bool RES = false>
__device__ __forceinline__ void score_sample(const BlockArgs &a, const HotLds &h,
                                             const Row<CH> &u, Row<CH> &g, uint32_t row, float lab,
                                             float lrc, int q, uint32_t nchunks) {
    const bool valid = row != kSentinel;
    if constexpr (RES) {
        // Resident cell (sgns_resident_kernel): `row` is the row inside the cell, every row of
        // which lives in h.base and is touched by this workgroup alone.  Read, score, and add
        // var * u to the row AS IT IS NOW: the row is read a second time right before the
        // 16-byte stores, so that an update of another group is lost only when it lands in the
        // ~100 cycles between that read and the stores (not in the ~300 of the scoring).
        // Measured alternatives, both exact: ds_add_f32 on every element runs at 1.3 cycles per
        // lane and element (15 x slower than the whole rest of the kernel), a per-row LDS lock
        // around read-score-write serialises the cell's hub rows (10 x slower).
        float *rw = h.base + (valid ? row : 0u) * h.ld;
        Row<CH> v;
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) {
            const uint32_t ci = cc * 16 + q;
            v.c[cc] = (valid && ci < nchunks) ? *reinterpret_cast<const float4 *>(rw + ci * 4)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float dot = dot_rows<CH>(u, v);
        const float var = valid ? (lab - sigmoid_clipped_fast(dot, a.clip)) * lrc : 0.f;
        axpy<CH>(g, var, v);
        asm volatile("" ::: "memory");  // the second read is a read, not the first one's registers
        if (valid) {
#pragma unroll
            for (int cc = 0; cc < CH; ++cc) {
                const uint32_t ci = cc * 16 + q;
                if (ci < nchunks) {
                    float4 o = *reinterpret_cast<const float4 *>(rw + ci * 4);
                    o.x += var * u.c[cc].x;
                    o.y += var * u.c[cc].y;
                    o.z += var * u.c[cc].z;
                    o.w += var * u.c[cc].w;
                    *reinterpret_cast<float4 *>(rw + ci * 4) = o;
                }
            }
        }
        return;
    }
    const bool hot = valid && (row & kHubBit) != 0;
    const bool cold = valid && !hot;
    float *base = sample_base(a, a.context, cold ? row : 0);
    Row<CH> v;
    load_row<CH>(v, base, q, nchunks, cold);
    float *hb = nullptr, *dl = nullptr;
    if (hot) {
        const uint32_t off = (row & 0xFFu) * h.ld;
        hb = h.base + off;
        dl = h.delta + off;
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) {
            const uint32_t ci = cc * 16 + q;
            if (ci < nchunks) {
                const float4 b = *reinterpret_cast<const float4 *>(hb + ci * 4);
                const float4 p = *reinterpret_cast<const float4 *>(dl + ci * 4);
                v.c[cc] = make_float4(b.x + p.x, b.y + p.y, b.z + p.z, b.w + p.w);
            }
        }
    }
    const float dot = dot_rows<CH>(u, v);
    const float var = valid ? (lab - sigmoid_clipped(dot, a.clip)) * lrc : 0.f;
    axpy<CH>(g, var, v);
    if (hot) {
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) {
            const uint32_t ci = cc * 16 + q;
            if (ci < nchunks) {
                lds_add_f32(dl + ci * 4 + 0, var * u.c[cc].x);
                lds_add_f32(dl + ci * 4 + 1, var * u.c[cc].y);
                lds_add_f32(dl + ci * 4 + 2, var * u.c[cc].z);
                lds_add_f32(dl + ci * 4 + 3, var * u.c[cc].w);
            }
        }
        if (((__float_as_uint(var) >> 3) & h.mask) == 0)
            hot_hand_over<CH, true>(hb, dl, h.lock + (row & 0xFFu),
                                    sample_base(a, a.context, h.grow[row & 0xFFu]), q, h.ld);
    } else if (cold) {
        if (a.reread) {
            // Small graphs (the tables live in the L2s): add var * u to the row AS IT IS NOW --
            // read again past the CU's L1 right before the stores -- instead of to the copy that
            // was scored: an update of another wave is lost only when it lands between that
            // second read (an L2 hit) and the stores, not anywhere in the memory round trip and
            // the scoring.  One more L2 read per sample row: nothing where HBM is not the bound.
            Row<CH> w;
#pragma unroll
            for (int cc = 0; cc < CH; ++cc) {
                const uint32_t ci = cc * 16 + q;
                if (ci < nchunks) {
                    f32x4 r;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(base + ci * 4)
                                 : "memory");
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r)::"memory");
                    w.c[cc] = make_float4(r.x, r.y, r.z, r.w);
                } else {
                    w.c[cc] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            scatter_add<CH, WMX>(base, q, nchunks, var, u, w);
        } else {
            scatter_add<CH, WMX>(base, q, nchunks, var, u, v);
        }
    }
}

// Resident cells, two samples of a pair side by side: the chain of a sample -- LDS read, dot,
// reduction over the 16 lanes, sigmoid, second read, stores -- is serial, and four waves a SIMD do
// not hide it; two independent chains do.  Exactly the sequential result while the two rows
// differ (the caller takes them one after the other when they are the same row).
// dot product over the 16 lanes of a group with two-wide accumulators (v_pk_fma_f32: half the
// instructions of the element-by-element sum; the kernel is bound by instruction issue)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int CH>
__device__ __forceinline__ float dot_rows_pk(const Row<CH> &a, const Row<CH> &b) {
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        acc += f32x2{a.c[cc].x, a.c[cc].y} * f32x2{b.c[cc].x, b.c[cc].y};
        acc += f32x2{a.c[cc].z, a.c[cc].w} * f32x2{b.c[cc].z, b.c[cc].w};
    }
    return group16_sum(acc.x + acc.y);
}

template <int CH>
__device__ __forceinline__ void score_sample_pair(const BlockArgs &a, const HotLds &h,
                                                  const Row<CH> &u, Row<CH> &g, uint32_t row_a,
                                                  float lab_a, uint32_t row_b, float lab_b,
                                                  float lrc, int q, uint32_t nchunks) {
    // an invalid sample (a negative that fell on the context or the centre; the tail of a record)
    // reads row 0 and contributes var = 0; only its stores are skipped -- a store of row + 0
    // could undo another group's update of that row
    const bool va = row_a != kSentinel, vb = row_b != kSentinel;
    float *ra = h.base + (va ? row_a : 0u) * h.ld, *rb = h.base + (vb ? row_b : 0u) * h.ld;
    Row<CH> xa, xb;
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const uint32_t ci = cc * 16 + q;
        xa.c[cc] = ci < nchunks ? *reinterpret_cast<const float4 *>(ra + ci * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        xb.c[cc] = ci < nchunks ? *reinterpret_cast<const float4 *>(rb + ci * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float dot_a = dot_rows_pk<CH>(u, xa), dot_b = dot_rows_pk<CH>(u, xb);
    const float var_a = va ? (lab_a - sigmoid_clipped_fast(dot_a, a.clip)) * lrc : 0.f;
    const float var_b = vb ? (lab_b - sigmoid_clipped_fast(dot_b, a.clip)) * lrc : 0.f;
    axpy<CH>(g, var_a, xa);
    axpy<CH>(g, var_b, xb);
    asm volatile("" ::: "memory");  // the second reads are reads
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const uint32_t ci = cc * 16 + q;
        if (ci < nchunks) {
            float4 oa = *reinterpret_cast<const float4 *>(ra + ci * 4);
            float4 ob = *reinterpret_cast<const float4 *>(rb + ci * 4);
            oa.x += var_a * u.c[cc].x;
            oa.y += var_a * u.c[cc].y;
            oa.z += var_a * u.c[cc].z;
            oa.w += var_a * u.c[cc].w;
            ob.x += var_b * u.c[cc].x;
            ob.y += var_b * u.c[cc].y;
            ob.z += var_b * u.c[cc].z;
            ob.w += var_b * u.c[cc].w;
            if (va) *reinterpret_cast<float4 *>(ra + ci * 4) = oa;
            if (vb) *reinterpret_cast<float4 *>(rb + ci * 4) = ob;
        }
    }
}

// the sample list of a run against the register row u (replicated in the four groups): a round
// is the four sample rows t0 .. t0 + 3, one per group, with no check for a row named twice in it
template <int CH, int WMX, bool RES = false>
__device__ __forceinline__ void score_run(const BlockArgs &a, const HotLds &h, const Row<CH> &u,
                                          Row<CH> &g, const uint32_t *s_rows, const float *s_lab,
                                          uint32_t n_samples, float lrc, int grp, int q) {
    const uint32_t nchunks = a.ld >> 2;
    for (uint32_t t0 = 0; t0 < n_samples; t0 += 4) {
        const uint32_t t = t0 + grp;
        const uint32_t row = t < n_samples ? s_rows[t] : kSentinel;
        const float lab = t < n_samples ? s_lab[t] : 0.f;
        score_sample<CH, WMX, RES>(a, h, u, g, row, lab, lrc, q, nchunks);
    }
}

template <int CH, int WMX, int WMC, bool DET, bool RES = false>
__device__ __forceinline__ void train_record(const BlockArgs &a, const HotLds &h, uint32_t cell,
                                             uint64_t lo, uint64_t hi, uint64_t p0, uint32_t n,
                                             uint64_t ckey, uint64_t cell_lo, uint64_t cell_n,
                                             uint32_t slice, uint32_t *s_key, uint32_t *s_val,
                                             uint32_t *s_hs, uint32_t *s_rows, float *s_lab,
                                             uint32_t *s_nb, float *s_tr, int lane, int grp, int q,
                                             unsigned long long &pairs,
                                             unsigned long long &runs) {
    const uint32_t k = a.k, nchunks = a.ld >> 2;
    const uint32_t rowmask = a.p.row_bits >= 32 ? 0xFFFFFFFFu : ((1u << a.p.row_bits) - 1u);
    // hot rows are served from the LDS accumulators in the parallel store schedules only
    const uint32_t hot_n = (DET || RES || is_atomic(WMX)) ? 0u : h.n;
    wave_sync();
    if ((uint32_t)lane < n) {
        const unsigned long long word = a.pairs[p0 + lane];
        const uint32_t low = (uint32_t)word & ((1u << a.p.ctx_bits) - 1u);
        const uint32_t hub = low >> (a.p.ctx_bits - 1);
        const uint32_t local = low & ((1u << (a.p.ctx_bits - 1)) - 1u);
        // row inside the part (resident cells: inside the cell)
        uint32_t val = RES ? local : slice + a.p.slices * local;
        uint32_t hs = kNoSlot;
        if (hub && hot_n) {
            hs = a.hot_slot[cell_lo + local];
            if (hs < hot_n) val |= kHubBit;
        }
        s_key[lane] = (uint32_t)(word >> a.p.ctx_bits) & rowmask;
        s_val[lane] = val;
        s_hs[lane] = hs;
    }
    // the centres just before and just after the record in the cell (kSentinel at its borders):
    // a run that has no equal neighbour is the ONLY run of its centre in this cell
    if (lane == 62)
        s_nb[0] = p0 > lo ? (uint32_t)(a.pairs[p0 - 1] >> a.p.ctx_bits) & rowmask : kSentinel;
    if (lane == 63)
        s_nb[1] = p0 + n < hi ? (uint32_t)(a.pairs[p0 + n] >> a.p.ctx_bits) & rowmask : kSentinel;
    wave_sync();
    const uint32_t n_samples = n * (k + 1);
    for (uint32_t t = lane; t < n_samples; t += 64) {
        const uint32_t pr = t / (k + 1), s = t - pr * (k + 1);
        const uint32_t xrow = s_val[pr];  // kHubBit: served from LDS slot s_hs[pr]
        uint32_t row = xrow;
        float lab = 1.f;
        if (s == 0) {
            if (xrow & kHubBit) row = kHubBit | s_hs[pr];
        } else {
            // a negative that falls on the pair's context or centre is drawn again: attempt
            // j + 1 = mix64(attempt j + golden), kNegAttempts draws at most (the oracle's
            // O_NEG_ATTEMPTS), then the sample is given up (cells of one or two rows)
            uint64_t r = draw(ckey, (p0 - lo + pr) * k + (s - 1));
            const uint64_t cgid = (uint64_t)s_key[pr] * a.p.world + a.p.rank;
            row = kSentinel;
            lab = 0.f;
            for (uint32_t att = 0; att < kNegAttempts; ++att, r = mix64(r + kGolden)) {
                uint32_t local = (uint32_t)mulhi64(r, cell_n), hub = 0;
                if (a.alias) {
                    const unsigned long long e = a.alias[cell_lo + local];
                    hub = (uint32_t)e & 1u;
                    if ((uint32_t)r >= ((uint32_t)e & ~1u)) {
                        local = (uint32_t)(e >> 32) & ~kHubBit;
                        hub = (uint32_t)(e >> 63);
                    }
                }
                const uint32_t cand = RES ? local : slice + a.p.slices * local;
                // the node behind the negative: staged with the cell's rows (resident cells; under
                // a placement it is not a function of the row any more)
                const uint64_t ngid =
                    RES ? (uint64_t)h.grow[local]
                        : (uint64_t)(slice + a.p.slices * local) * a.p.parts + a.part;
                if (cand == (xrow & ~kHubBit) || ngid == cgid) continue;
                row = cand;
                if (hub && hot_n) {
                    const uint32_t slot = a.hot_slot[cell_lo + local];
                    if (slot < hot_n) row = kHubBit | slot;
                }
                break;
            }
        }
        s_rows[t] = row;
        s_lab[t] = lab;
    }
    wave_sync();
    // Records of (almost) only single-pair runs -- the rule at 100 M nodes, where a round holds
    // 1.1 pairs per (cell, centre) -- are trained PAIR PER GROUP: each 16-lane group owns one pair
    // at a time (its central row, its gradient, its k + 1 sample rows one after the other), four
    // pairs side by side in one instruction stream.  Same four rows in flight per wave, but the
    // per-run work (row fetch, gradient hand-over, bookkeeping) is issued once for four pairs and
    // the reduction over the groups disappears: the kernel is issue bound as much as bandwidth
    // bound (SIMDs busy 77 % of the time).  A pair whose neighbours share its centre adds its
    // gradient with atomics (several groups, or waves, hold that row), a lone pair stores row +
    // gradient.  Runs proper (the bench graph: 3.4 pairs) keep the run-major loop below: there a
    // group-private centre would re-read the central row per pair.
    // DET (resident cells only): the same code with the four groups taking TURNS -- one pair at a
    // time, in record order -- taken when every run of the record is a single pair, where "a pair
    // per group" and the oracle's "run of equal centre" are the same thing.  That is how the
    // packed dot products, the v_rcp sigmoid, the side-by-side samples, the prefetched central
    // rows and the transposed atomic hand-over are compared with the oracle element by element.
    if constexpr ((!DET || RES) && !is_atomic(WMX) && WMC == kAtomic) {
        uint32_t n_runs = 0;
        {
            const bool starts = (uint32_t)lane < n && (lane == 0 || s_key[lane] != s_key[lane - 1]);
            n_runs = (uint32_t)__popcll(__ballot(starts));
        }
        if (DET ? n_runs == n : n_runs * 100 >= n * kPpgMinPct) {
            const uint32_t kk = k + 1;
            // the central rows of the NEXT four pairs are fetched while these four are scored (a
            // memory round trip per pair otherwise); a pair whose centre one of these four also
            // has then starts from the row before their gradients -- like a pair of another wave
            // (resident cells only: the XCD cells' kernels sit on their register budget)
            Row<CH> u_next;
            if constexpr (RES)
                load_row<CH>(u_next, a.central + (uint64_t)s_key[(uint32_t)grp < n ? grp : 0] * a.cld,
                             q, nchunks, (uint32_t)grp < n);
            for (uint32_t p4 = 0; p4 < n; p4 += 4) {
                const uint32_t pr = p4 + grp;
                const bool in_rec = pr < n;
                const uint32_t crow_id = s_key[in_rec ? pr : 0];
                float *crow = a.central + (uint64_t)crow_id * a.cld;
                Row<CH> u, g;
                if constexpr (RES) {
                    u = u_next;
                    const uint32_t nx = pr + 4;
                    load_row<CH>(u_next, a.central + (uint64_t)s_key[nx < n ? nx : 0] * a.cld, q,
                                 nchunks, nx < n);
                } else {
                    load_row<CH>(u, crow, q, nchunks, in_rec);
                }
                zero_row<CH>(g);
              for (int det_turn = 0; det_turn < (DET ? 4 : 1); ++det_turn) {
                const bool have = in_rec && (!DET || grp == det_turn);
                float lrc = a.lr;
                if (a.flags & kFlagNormLr) {
                    const uint64_t c = (uint64_t)crow_id * a.p.world + a.p.rank;
                    const uint64_t deg = a.g.row_ptr[c + 1] - a.g.row_ptr[c];
                    if (deg) lrc = a.lr / (float)deg;
                }
                uint32_t sidx = 0;
                if constexpr (RES) {
                    for (; sidx + 1 < kk; sidx += 2) {
                        const uint32_t t = (have ? pr : 0) * kk + sidx;
                        const uint32_t row_a = have ? s_rows[t] : kSentinel;
                        const uint32_t row_b = have ? s_rows[t + 1] : kSentinel;
                        if (__ballot(row_a == row_b && row_a != kSentinel)) {  // one row twice
                            score_sample<CH, WMX, RES>(a, h, u, g, row_a, s_lab[t], lrc, q, nchunks);
                            score_sample<CH, WMX, RES>(a, h, u, g, row_b, s_lab[t + 1], lrc, q,
                                                       nchunks);
                        } else {
                            score_sample_pair<CH>(a, h, u, g, row_a, s_lab[t], row_b, s_lab[t + 1],
                                                  lrc, q, nchunks);
                        }
                    }
                }
                for (; sidx < kk; ++sidx) {
                    const uint32_t t = (have ? pr : 0) * kk + sidx;
                    const uint32_t row = have ? s_rows[t] : kSentinel;
                    score_sample<CH, WMX, RES>(a, h, u, g, row, s_lab[t], lrc, q, nchunks);
                }
                // the pair's gradient goes to its central row: one store of row + gradient when
                // the plan allows it and the pair's neighbours have other centres (a.central_store),
                // else f32 atomics in the lane-contiguous shape -- the four groups take turns at
                // the wave's transposition row
                bool store = false;
                if (have && a.central_store) {
                    const bool same_prev = pr > 0 ? s_key[pr - 1] == crow_id : s_nb[0] == crow_id;
                    const bool same_next = pr + 1 < n ? s_key[pr + 1] == crow_id : s_nb[1] == crow_id;
                    store = !same_prev && !same_next;
                }
                if (store) scatter_add<CH, kWriteThrough>(crow, q, nchunks, 1.0f, g, u);
                if (__ballot(have && !store)) {
#pragma unroll
                    for (int turn = 0; turn < 4; ++turn) {
                        wave_sync();
                        if (grp == turn) {
#pragma unroll
                            for (int cc = 0; cc < CH; ++cc) {
                                const uint32_t ci = cc * 16 + q;
                                if (ci < nchunks)
                                    *reinterpret_cast<float4 *>(s_tr + ci * 4) = g.c[cc];
                            }
                        }
                        wave_sync();
                        if (grp == turn && have && !store) {
#pragma unroll
                            for (int cc = 0; cc < CH; ++cc) {
                                const uint32_t f = cc * 64 + q;
                                if (f < a.ld) unsafeAtomicAdd(crow + f, s_tr[f]);
                                if (f + 16 < a.ld) unsafeAtomicAdd(crow + f + 16, s_tr[f + 16]);
                                if (f + 32 < a.ld) unsafeAtomicAdd(crow + f + 32, s_tr[f + 32]);
                                if (f + 48 < a.ld) unsafeAtomicAdd(crow + f + 48, s_tr[f + 48]);
                            }
                        }
                    }
                }
                // the next pair of the record may name a row this one has just changed
                if constexpr (DET) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
              }
            }
            pairs += n;
            runs += n;  // one central row read and one hand-over per pair
            return;
        }
    }
    // Runs of equal centre inside the record (short: a centre's pairs are spread over all cells).
    // The central row of the NEXT run is loaded while the current run is scored (other runs have
    // other centres, so it cannot be changed by this record in between).
    uint32_t r0 = 0;
    Row<CH> u_next;
    if constexpr (!DET)
        load_row<CH>(u_next, a.central + (uint64_t)s_key[0] * a.cld, q, nchunks, n != 0);
    while (r0 < n) {
        const uint32_t crow_id = s_key[r0];
        uint32_t r1 = r0 + 1;
        while (r1 < n && r1 - r0 < kMaxRun && s_key[r1] == crow_id) ++r1;
        float lrc = a.lr;
        if (a.flags & kFlagNormLr) {
            const uint64_t c = (uint64_t)crow_id * a.p.world + a.p.rank;
            const uint64_t deg = a.g.row_ptr[c + 1] - a.g.row_ptr[c];
            if (deg) lrc = a.lr / (float)deg;
        }
        float *crow = a.central + (uint64_t)crow_id * a.cld;
        Row<CH> u, g;
        if constexpr (DET) {
            load_row<CH>(u, crow, q, nchunks, true);
        } else {
            u = u_next;
            load_row<CH>(u_next, a.central + (uint64_t)s_key[r1 < n ? r1 : r0] * a.cld, q, nchunks,
                         r1 < n);
        }
        zero_row<CH>(g);
        if constexpr (DET && RES) {
            // the cell's rows live in LDS: one sample after the other, group 0 alone
            for (uint32_t t = r0 * (k + 1); t < r1 * (k + 1); ++t)
                score_sample<CH, WMX, true>(a, h, u, g, grp == 0 ? s_rows[t] : kSentinel, s_lab[t],
                                            lrc, q, nchunks);
        } else if constexpr (DET) {
            score_samples<CH, WMX, DET>(a, a.context, u, u, g, s_rows + r0 * (k + 1),
                                        s_lab + r0 * (k + 1), (r1 - r0) * (k + 1), lrc, grp, q);
        } else if constexpr (!is_atomic(WMX)) {
            score_run<CH, WMX, RES>(a, h, u, g, s_rows + r0 * (k + 1), s_lab + r0 * (k + 1),
                               (r1 - r0) * (k + 1), lrc, grp, q);
        } else {  // atomics on every row: u in the lane-contiguous layout for the row updates
            Row<CH> u_upd;
            to_contig_layout<CH>(u_upd, u, s_tr, grp, q, a.ld);
            score_samples_racy<CH, WMX>(a, a.context, u, u_upd, g, s_rows + r0 * (k + 1),
                                        s_lab + r0 * (k + 1), (r1 - r0) * (k + 1), lrc, grp, q);
        }
        if constexpr (!DET) reduce_groups<CH>(g);
        // The gradient of the run goes to the central row.  A centre whose pairs in this cell are
        // all in this run (the rule at 100 M nodes, where a round holds 1.1 pairs per cell and
        // centre) is updated with one write-through store of u + g: u is still in registers, no
        // other wave of this XCD names the row in this launch, and the seven other XCDs reach it
        // -- through their own cells -- at some other moment of the launch with probability
        // ~1 - 1e-4.  A centre that spans records (hubs: thousands of consecutive records that many
        // waves train at once) gets its gradient added with hardware f32 atomics: no add is lost.
        // An atomic row add costs ~5 stored rows (128 dword atomics through the L2 atomic units).
        if constexpr (!DET) {
            // the next run continues this centre (a stretch cut at kMaxRun): its row was fetched
            // before this run's gradient existed
            if (r1 < n && s_key[r1] == crow_id) {
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) {
                    u_next.c[cc].x = u.c[cc].x + g.c[cc].x;
                    u_next.c[cc].y = u.c[cc].y + g.c[cc].y;
                    u_next.c[cc].z = u.c[cc].z + g.c[cc].z;
                    u_next.c[cc].w = u.c[cc].w + g.c[cc].w;
                }
            }
        }
        bool alone = false;
        if constexpr (!DET && WMC == kAtomic)
            alone = a.central_store && (r0 > 0 ? s_key[r0 - 1] != crow_id : s_nb[0] != crow_id) &&
                    (r1 < n ? s_key[r1] != crow_id : s_nb[1] != crow_id);
        if constexpr (!DET && WMC == kAtomic) {
            if (alone) {
                if (grp == 0) scatter_add<CH, kWriteThrough>(crow, q, nchunks, 1.0f, g, u);
            } else {
                // every group holds the run's gradient (reduce_groups): through the wave's
                // transposition row into the lane-contiguous shape, a quarter of the row per group
                wave_sync();
                if (grp == 0) {
#pragma unroll
                    for (int cc = 0; cc < CH; ++cc) {
                        const uint32_t ci = cc * 16 + q;
                        if (ci < nchunks) *reinterpret_cast<float4 *>(s_tr + ci * 4) = g.c[cc];
                    }
                }
                wave_sync();
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const uint32_t f = (grp * CH + j) * 16 + q;
                    if (f < a.ld) unsafeAtomicAdd(crow + f, s_tr[f]);
                }
            }
        } else {
            if (grp == 0) scatter_add<CH, DET ? kWriteBack : WMC>(crow, q, nchunks, 1.0f, g, u);
        }
        if constexpr (DET) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        r0 = r1;
        ++runs;  // one central row read + one gradient add per run of equal centre
    }
    pairs += n;
}

// WMX: store flavour of the contextual rows (kWriteBack only when every cell is exclusive to one
// XCD: slices a multiple of the XCDs in use), WMC: of the central rows (shared between XCDs).
// FULL: the row stride is exactly CH * 64 floats (d = 128 -> CH = 2; every d that pads to 64, 128,
// 256, 512 or 1024): the compiler then knows ld, every "is this chunk inside the row" predicate
// folds away (they were sixteen 64-bit masks held in SGPRs across the whole kernel) and the
// per-chunk bounds tests leave the instruction stream.
// LDS: per wave  transpose row[ld] | centre rows[C] | context rows[C] | hot slots[C] |
// sample rows[C (k + 1)] | labels[C (k + 1)] | neighbour centres[2];  then, shared by the
// workgroup, the hot rows  base[hot_n][ld] | delta[hot_n][ld] | rows[hot_n] | locks[hot_n].
// WG: threads of a workgroup.  256 (four workgroups per CU), or 1 024 = one workgroup per CU: its
// sixteen waves share ONE set of hot rows -- four times the rows in the same LDS, a quarter of
// the copies per XCD (what is in limbo between the copies shrinks with them).
template <int CH, int WMX, int WMC, bool DET, bool FULL = false, int WG = kTrainBlock>
__global__ __launch_bounds__(WG) void sgns_block_kernel(BlockArgs a) {
    if constexpr (FULL) a.ld = CH * 64;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, q = lane & 15;
    const uint32_t C = a.p.record, k = a.k;
    const uint32_t per_wave = (a.ld + 3 * C + 2 * C * (k + 1) + 2 + 3) & ~3u;
    float *s_tr = reinterpret_cast<float *>(smem + wave * per_wave);
    uint32_t *s_key = smem + wave * per_wave + a.ld;
    uint32_t *s_val = s_key + C;
    uint32_t *s_hs = s_val + C;
    uint32_t *s_rows = s_hs + C;
    float *s_lab = reinterpret_cast<float *>(s_rows + C * (k + 1));
    uint32_t *s_nb = s_rows + 2 * C * (k + 1);  // centres next to the record
    unsigned long long pairs = 0, runs = 0;
    const uint64_t part_rows = stripe_count(a.n_nodes, a.part, a.p.parts);
    const uint32_t n_waves = blockDim.x >> 6;
    HotLds h{};
    if constexpr (!DET && !is_atomic(WMX)) {
        if (a.hot_n && !a.sweep) {
            h.n = a.hot_n;
            h.ld = a.ld;
            h.mask = a.hot_mask;
            h.base = reinterpret_cast<float *>(smem + n_waves * per_wave);
            h.delta = h.base + h.n * a.ld;
            h.grow = smem + n_waves * per_wave + 2 * h.n * a.ld;
            h.lock = h.grow + h.n;
        }
    }

    // Which slices this workgroup serves.  When the slices are a multiple of the XCDs in use
    // (a.xcds; 8 slices on an MI355X), XCD x serves the slices x, x + xcds, ...: every slice --
    // every contextual row -- is then touched by exactly one XCD during the launch (the host picks
    // write-back stores only in that case).  Otherwise XCD x serves slice x % slices: several XCDs
    // share a slice and the stores are write-through.
    uint32_t first_slice = 0, slice_step = 1, n_my_slices = a.p.slices;
    if (!DET && !a.sweep && a.p.slices > 1) {
        const uint32_t x = xcc_id();
        if (a.xcds != 0 && a.p.slices % a.xcds == 0) {
            first_slice = x;
            slice_step = a.xcds;
            n_my_slices = x < a.xcds ? a.p.slices / a.xcds : 0;
        } else {
            first_slice = x % a.p.slices;
            n_my_slices = 1;
        }
    }
    for (uint32_t si = 0; si < n_my_slices; ++si) {
        const uint32_t slice = first_slice + si * slice_step;
        const uint32_t cell = a.part * a.p.slices + slice;
        const uint64_t lo = a.cell_offsets[cell], hi = a.cell_offsets[cell + 1];
        const uint64_t cell_n = stripe_count(part_rows, slice, a.p.slices);
        const bool work = hi != lo && cell_n != 0;  // pairs imply rows: their contexts
        if (h.n) {  // the same for every wave of the workgroup: the barriers are uniform
            for (uint32_t i = threadIdx.x; i < h.n * a.ld; i += blockDim.x) {
                const uint32_t s = i / a.ld, f = i - s * a.ld;
                const uint32_t r = a.hot_list[(size_t)cell * kHotMax + s];
                h.delta[i] = 0.f;
                h.base[i] = r == kSentinel ? 0.f
                                           : sample_base(a, a.context, slice + a.p.slices * r)[f];
                if (f == 0) {
                    h.grow[s] = r == kSentinel ? kSentinel : slice + a.p.slices * r;
                    h.lock[s] = 0;
                }
            }
            __syncthreads();
        }
        if (work) {
            const uint64_t R = (hi - lo + C - 1) / C;
            const uint64_t A = record_stride(R);
            const uint64_t ckey = cell_stream_key(a.ekey, a.block_id, cell);
            const uint64_t cell_lo = a.cell_rows ? a.cell_rows[cell] : 0;
            if constexpr (DET) {
                for (uint64_t t = 0; t < R; ++t) {
                    const uint64_t rec = (t * A) % R;
                    const uint64_t p0 = lo + rec * C;
                    const uint32_t n = (uint32_t)min((uint64_t)C, hi - p0);
                    train_record<CH, WMX, WMC, DET>(a, h, cell, lo, hi, p0, n, ckey, cell_lo,
                                                    cell_n, slice, s_key, s_val, s_hs, s_rows,
                                                    s_lab, s_nb, s_tr, lane, grp, q, pairs, runs);
                }
            } else {
                // a ticket = kTicket consecutive visiting-order indices (the stride order spreads
                // them over the cell); one returning atomic per ticket on the cell's own cursor line
                // Every cell starts its stride order somewhere else (an offset drawn from the cell's
                // key).  The cells of a part hold the same centres in the same (sorted) order and
                // their XCDs advance at the same pace: without the offset all eight would work on
                // the same centre rows at the same moment, all launch long -- the one situation in
                // which the single-run store of train_record loses updates (and in which the atomic
                // adds of a hub centre queue up behind one another).
                const uint64_t start = mulhi64(ckey, R);
                unsigned long long *cursor = a.cursors + (size_t)slice * kCursorStep;
                for (;;) {
                    unsigned long long t = 0;
                    if (lane == 0) t = atomicAdd(cursor, (unsigned long long)kTicket);
                    t = __shfl(t, 0);
                    if (t >= R) break;
                    const uint64_t t_end = min(t + kTicket, (unsigned long long)R);
                    for (; t < t_end; ++t) {
                        const uint64_t rec = (t * A + start) % R;
                        const uint64_t p0 = lo + rec * C;
                        const uint32_t n = (uint32_t)min((uint64_t)C, hi - p0);
                        train_record<CH, WMX, WMC, DET>(a, h, cell, lo, hi, p0, n, ckey, cell_lo,
                                                        cell_n, slice, s_key, s_val, s_hs, s_rows,
                                                        s_lab, s_nb, s_tr, lane, grp, q, pairs,
                                                        runs);
                    }
                }
            }
        }
        if (h.n) {  // what is still pending goes to the rows before the workgroup leaves the cell
            __syncthreads();
            for (uint32_t s = wave * 4 + grp; s < h.n; s += n_waves * 4) {
                const uint32_t r = h.grow[s];
                if (r != kSentinel)
                    hot_hand_over<CH, false>(h.base + s * a.ld, h.delta + s * a.ld, nullptr,
                                             sample_base(a, a.context, r), q, a.ld);
            }
            __syncthreads();
        }
    }
    if (a.counters && lane == 0 && pairs) {
        atomicAdd(&a.counters[0], pairs);
        atomicAdd(&a.counters[2], runs);  // "centres": runs of equal centre
    }
}

// --------------------------------------------------------------------------------------------
// Resident cells: the plan of a small or mid-sized graph (gn2v_block_auto_plan: up to 115 M nodes
// at d = 128) makes cells whose rows fit ONE workgroup's LDS.  A launch covers a part (or a group
// of parts), one workgroup of sixteen waves per cell: it loads the cell's contextual rows into
// LDS, trains ALL the cell's records (its waves take them from an LDS cursor), reads and updates
// the rows in LDS and writes them back when the cell is done.  No other workgroup touches those
// rows during the launch, so no flavour of global store or atomic is involved and nothing is in
// limbo between copies.  What still races is the workgroup with itself: a sample is a plain LDS
// read -> score -> second read -> 16-byte stores by one of 64 concurrent 16-lane groups
// (score_sample<RES>, score_sample_pair), so two groups that name the same row within the ~100
// cycles between the second read and the stores keep one of the two updates -- always so for two
// groups of one wave that name a row in the same instruction.  The hub rows of a cell (the
// targets of the degree-proportional negatives) are hit that way regularly: the contextual table
// moves 0.85-0.87 x as far as the sequential restatement of the same schedule, where an exact
// accumulation in the same parallel order gives 0.905 (counted per row by
// tests/test_gpu_resident.py).  (The XCD cells of sgns_block_kernel additionally lose what waves
// of different CUs write to a row at the same moment: at 2 708 nodes the contextual table moved
// 0.69 x as far as the sequential schedule)  The central rows are shared between
// the cells and take the run's gradient by f32 atomics, as in sgns_block_kernel: exact.
// DET: the deterministic instantiation (GN2V_TRAIN_DETERMINISTIC on a resident plan) -- ONE
// workgroup walks the launch's cells one after the other, wave 0 trains a cell's records in the
// stride order t * A mod R, samples in order, rows in LDS: the oracle's o_block_step, to 1e-5.
// LDS: per wave the staging of train_record, then rows[cell rows][ld], the rows' node ids and
// the record cursor.
// --------------------------------------------------------------------------------------------
template <int CH, bool DET>
__device__ __forceinline__ void resident_cell(BlockArgs &a, uint32_t *smem, uint32_t slice,
                                              unsigned long long &pairs,
                                              unsigned long long &runs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, q = lane & 15;
    const uint32_t C = a.p.record, k = a.k;
    const uint32_t per_wave = (a.ld + 3 * C + 2 * C * (k + 1) + 2 + 3) & ~3u;
    float *s_tr = reinterpret_cast<float *>(smem + wave * per_wave);
    uint32_t *s_key = smem + wave * per_wave + a.ld;
    uint32_t *s_val = s_key + C;
    uint32_t *s_hs = s_val + C;
    uint32_t *s_rows = s_hs + C;
    float *s_lab = reinterpret_cast<float *>(s_rows + C * (k + 1));
    uint32_t *s_nb = s_rows + 2 * C * (k + 1);
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t cell = a.part * a.p.slices + slice;
    const uint64_t lo = a.cell_offsets[cell], hi = a.cell_offsets[cell + 1];
    if (hi == lo) return;  // the same for every wave of the workgroup
    const uint64_t part_rows = stripe_count(a.n_nodes, a.part, a.p.parts);
    const uint32_t cell_n = (uint32_t)stripe_count(part_rows, slice, a.p.slices);
    HotLds h{};
    h.base = reinterpret_cast<float *>(smem + n_waves * per_wave);
    h.ld = a.ld;
    h.n = cell_n;
    // the node behind every row of the cell (for "is this negative the centre itself", and for
    // where the row lies under a placement)
    h.grow = smem + n_waves * per_wave + cell_n * a.ld;
    uint32_t *s_cursor = h.grow + cell_n;
    for (uint32_t r = threadIdx.x; r < cell_n; r += blockDim.x) {
        const uint64_t xp = (uint64_t)(slice + a.p.slices * r) * a.p.parts + a.part;
        h.grow[r] = a.inv ? a.inv[xp] : (uint32_t)xp;
    }
    if (threadIdx.x == 0) *s_cursor = 0;
    __syncthreads();
    auto row_ptr = [&](uint32_t r) -> float * {
        if (!a.inv) return sample_base(a, a.context, slice + a.p.slices * r);
        const uint32_t x = h.grow[r];
        return a.ctx_table ? a.ctx_table + (uint64_t)x * a.ld
                           : a.context + (uint64_t)(x / a.p.parts) * a.xld;
    };
    for (uint32_t i = threadIdx.x; i < cell_n * (a.ld >> 2); i += blockDim.x) {
        const uint32_t r = i / (a.ld >> 2), c4 = i - r * (a.ld >> 2);
        reinterpret_cast<float4 *>(h.base + r * a.ld)[c4] =
            reinterpret_cast<const float4 *>(row_ptr(r))[c4];
    }
    __syncthreads();
    const uint64_t R = (hi - lo + C - 1) / C;
    const uint64_t A = record_stride(R);
    const uint64_t ckey = cell_stream_key(a.ekey, a.block_id, cell);
    const uint64_t cell_lo = a.cell_rows ? a.cell_rows[cell] : 0;
    if constexpr (DET) {
        if (wave == 0)
            for (uint64_t t = 0; t < R; ++t) {
                const uint64_t rec = (t * A) % R;
                const uint64_t p0 = lo + rec * C;
                const uint32_t n = (uint32_t)min((uint64_t)C, hi - p0);
                train_record<CH, kWriteBack, kAtomic, true, true>(
                    a, h, cell, lo, hi, p0, n, ckey, cell_lo, cell_n, slice, s_key, s_val, s_hs,
                    s_rows, s_lab, s_nb, s_tr, lane, grp, q, pairs, runs);
            }
    } else {
        const uint64_t start = mulhi64(ckey, R);
        for (;;) {
            uint32_t t0 = 0;
            if (lane == 0)
                t0 = __hip_atomic_fetch_add((lds_u32 *)s_cursor, 1u, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint64_t t = __shfl(t0, 0);
            if (t >= R) break;
            const uint64_t rec = (t * A + start) % R;
            const uint64_t p0 = lo + rec * C;
            const uint32_t n = (uint32_t)min((uint64_t)C, hi - p0);
            train_record<CH, kWriteBack, kAtomic, false, true>(
                a, h, cell, lo, hi, p0, n, ckey, cell_lo, cell_n, slice, s_key, s_val, s_hs,
                s_rows, s_lab, s_nb, s_tr, lane, grp, q, pairs, runs);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cell_n * (a.ld >> 2); i += blockDim.x) {
        const uint32_t r = i / (a.ld >> 2), c4 = i - r * (a.ld >> 2);
        reinterpret_cast<float4 *>(row_ptr(r))[c4] =
            reinterpret_cast<const float4 *>(h.base + r * a.ld)[c4];
    }
}

template <int CH, bool FULL = false, bool DET = false>
__global__ __launch_bounds__(1024) void sgns_resident_kernel(BlockArgs a) {
    if constexpr (FULL) a.ld = CH * 64;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    unsigned long long pairs = 0, runs = 0;
    if constexpr (DET) {
        // one workgroup, the cells of the launch in order (parts, then slices): a.sweep carries
        // the number of parts
        const uint32_t part0 = a.part, part_n = a.sweep ? a.sweep : 1;
        for (uint32_t p = 0; p < part_n; ++p) {
            a.part = part0 + p;
            if (a.part_ptrs) a.context = a.part_ptrs[a.part];
            for (uint32_t slice = 0; slice < a.p.slices; ++slice) {
                resident_cell<CH, true>(a, smem, slice, pairs, runs);
                __syncthreads();  // the LDS is the next cell's
            }
        }
    } else {
        // A launch covers a GROUP of parts (blockIdx.y) when the caller holds them all: a launch
        // of one part lasts as long as its heaviest cell -- and the striping puts one of the
        // graph's oldest hubs into every part -- while the workgroups of a group are handed to
        // the CUs as they fall free (BA 10 M: a part's heaviest cell carries 1.5-9 x the average).
        if (a.part_ptrs) {
            a.part += blockIdx.y;
            a.context = a.part_ptrs[a.part];
        }
        resident_cell<CH, false>(a, smem, blockIdx.x, pairs, runs);
    }
    if (a.counters && (threadIdx.x & 63) == 0 && pairs) {
        atomicAdd(&a.counters[0], pairs);
        atomicAdd(&a.counters[2], runs);
    }
}

// natural[x] <- part-major storage: the rows of part p = x % parts lie one after the other from
// row part_first[p] on (row x / parts of the part)
static __global__ void parts_to_natural_kernel(float *__restrict__ natural,
                                               const float *__restrict__ part_major,
                                               const unsigned long long *__restrict__ part_first,
                                               uint64_t n_nodes, uint32_t ld, uint32_t parts) {
    const uint64_t n = n_nodes * (ld >> 2);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t x = i / (ld >> 2);
        const uint32_t c = (uint32_t)(i - x * (ld >> 2));
        const uint64_t src = part_first[x % parts] + x / parts;
        reinterpret_cast<float4 *>(natural + x * ld)[c] =
            reinterpret_cast<const float4 *>(part_major + src * ld)[c];
    }
}

// table row r <- initial values of global row first_row + r * row_stride (shards of a table that
// never exists as a whole on this device)
__global__ void init_rows_kernel(float *__restrict__ t, uint64_t n_rows, uint32_t d, uint32_t ld,
                                 uint64_t key, float scale, uint64_t first_row,
                                 uint64_t row_stride) {
    const uint64_t n = n_rows * ld;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / ld;
        const uint32_t c = (uint32_t)(i - r * ld);
        float v = 0.f;
        if (c < d) {
            const uint64_t h = draw(key, (first_row + r * row_stride) * d + c);
            const float u = __fmul_rn((float)(h >> 40), 1.0f / 16777216.0f);
            v = __fmul_rn(__fsub_rn(__fmul_rn(2.0f, u), 1.0f), scale);
        }
        t[i] = v;
    }
}

}  // namespace gn2v

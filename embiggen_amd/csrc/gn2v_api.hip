// libgn2v.so -- C ABI (include/gn2v.h) over the gfx950 kernels in this directory.
// Host side of the drop-in for `models.SkipGram/CBOW(...).fit_transform(graph)`
// (reference: embiggen/embedders/ensmallen_embedders/node2vec.py:65-69,:99).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gn2v_internal.h"
#include "../../include/gn2v_experimental.h"
#include "rng.h"
#include "edge_kernels.h"
#include "glove_kernels.h"
#include "train_kernels.h"
#include "cbow_lazy_kernel.h"
#include "util_kernels.h"
#include "walk_kernels.h"

#include "handle.h"

namespace gn2v_host {

thread_local std::string g_err;

int fail(const std::string &msg) {
    g_err = msg;
    return 1;
}

int get_events(gn2v_graph *g, EventPair *ev) {
    if (!g->free_events.empty()) {
        *ev = g->free_events.back();
        g->free_events.pop_back();
        return 0;
    }
    HIP_TRY(hipEventCreate(&ev->a));
    HIP_TRY(hipEventCreate(&ev->b));
    return 0;
}

}  // namespace gn2v_host

using namespace gn2v_host;

namespace {

// fold finished event pairs into the ms accumulators (caller has synchronised the stream)
int fold_events(gn2v_graph *g) {
    for (auto &ev : g->train_events) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        g->train_ms += ms;
        g->free_events.push_back(ev);
    }
    g->train_events.clear();
    for (auto &ev : g->walk_events) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        g->walk_ms += ms;
        g->free_events.push_back(ev);
    }
    g->walk_events.clear();
    return 0;
}

int check_walk_params(const gn2v_walk_params *wp) {
    if (!wp) return fail("walk params are NULL");
    if (wp->walk_length < 2) return fail("walk_length must be >= 2");
    if (wp->iterations < 1) return fail("iterations must be >= 1");
    if (!(wp->return_weight > 0.f) || !(wp->explore_weight > 0.f) ||
        !std::isfinite(wp->return_weight) || !std::isfinite(wp->explore_weight))
        return fail("return_weight and explore_weight must be finite and strictly positive");
    if (!(wp->change_node_type_weight >= 0.f) || !(wp->change_edge_type_weight >= 0.f) ||
        !std::isfinite(wp->change_node_type_weight) || !std::isfinite(wp->change_edge_type_weight))
        return fail("change_node_type_weight and change_edge_type_weight must be finite and "
                    "strictly positive (0 = unset)");
    return 0;
}

void type_factors(float weight, uint64_t *same, uint64_t *diff) {
    const double w = weight == 0.0f ? 1.0 : (double)weight;
    const double mx = w > 1.0 ? w : 1.0;
    *same = (uint64_t)std::floor(1.0 / mx * 4294967296.0);
    *diff = (uint64_t)std::floor(w / mx * 4294967296.0);
}

gn2v::WalkConsts walk_consts(const gn2v_graph *g, const gn2v_walk_params *wp) {
    gn2v::WalkConsts c{};
    type_factors(wp->change_node_type_weight, &c.fn_same, &c.fn_diff);
    type_factors(wp->change_edge_type_weight, &c.fe_same, &c.fe_diff);
    c.node_bias = g->view.node_types != nullptr && c.fn_same != c.fn_diff;
    c.edge_bias = g->view.edge_types != nullptr && c.fe_same != c.fe_diff;
    c.walk_length = wp->walk_length;
    c.max_neighbours = wp->max_neighbours;  // 0: exact walks
    c.second_order = !(wp->return_weight == 1.0f && wp->explore_weight == 1.0f);
    const double rw = wp->return_weight, ew = wp->explore_weight;
    double mx = rw > ew ? rw : ew;
    if (mx < 1.0) mx = 1.0;
    const double s = 4294967296.0;
    c.t_ret = (uint64_t)std::floor(rw / mx * s);
    c.t_common = (uint64_t)std::floor(1.0 / mx * s);
    c.t_explore = (uint64_t)std::floor(ew / mx * s);
    c.t_min = std::min(c.t_common, c.t_explore);
    c.t_max = std::max(c.t_common, c.t_explore);
    // trial budget: 37 / (smallest acceptance probability) rejections leave a chance below e^-37
    // of reaching the exact scan; clamped to [32, 1024] (same formula as the oracle)
    double a = 1.0;
    if (c.second_order) a *= (double)std::min(c.t_ret, c.t_min) / s;
    if (c.node_bias) a *= (double)std::min(c.fn_same, c.fn_diff) / s;
    if (c.edge_bias) a *= (double)std::min(c.fe_same, c.fe_diff) / s;
    if (!(a > 37.0 / 1024.0)) {
        c.max_trials = 1024;
    } else {
        const double n = std::ceil(37.0 / a);
        c.max_trials = n < 32.0 ? 32u : (uint32_t)n;
    }
    // the previous node proposed on its own (walk_kernels.h WalkConsts.apart; same rule and
    // arithmetic as the oracle's make_walk_consts)
    const double m = ew > 1.0 ? ew : 1.0;
    c.apart = c.second_order && !c.node_bias && !c.edge_bias && g->view.cumw == nullptr &&
              rw > m && rw <= 1024.0 && m <= 1024.0;
    if (c.apart) {
        c.rq = (uint64_t)std::floor(rw * 1048576.0);
        c.mq = (uint64_t)std::floor(m * 1048576.0);
        c.s_common = (uint64_t)std::floor(1.0 / m * s);
        c.s_explore = (uint64_t)std::floor(ew / m * s);
        c.s_min = std::min(c.s_common, c.s_explore);
        c.s_max = std::max(c.s_common, c.s_explore);
        const double a2 = (double)c.s_min / s;
        if (!(a2 > 37.0 / 1024.0)) {
            c.max_trials = 1024;
        } else {
            const double n = std::ceil(37.0 / a2);
            c.max_trials = n < 32.0 ? 32u : (uint32_t)n;
        }
    }
    return c;
}

// The edge set of the second-order sampler (walk_kernels.h): 8 B per slot, 2 .. 4 slots per
// directed edge (a power of two at load <= 1/2): 3.2 GB for the 10 M / 100 M bench graph, 32 GB at
// 100 M / 1 B.  Built once per handle, on the first walk that needs it, when it fits a quarter of
// the free memory; GN2V_WALK_EDGE_SET=0 keeps the binary searches (same walks either way).
int ensure_edge_set(gn2v_graph *g, hipStream_t s) {
    if (g->edge_set_tried) return 0;
    const char *env = getenv("GN2V_WALK_EDGE_SET");
    const uint64_t E = g->view.n_edges;
    if ((env && *env == '0') || E == 0 || g->view.n_nodes >= 0xFFFFFFFFULL) {
        g->edge_set_tried = true;  // never for this handle
        return 0;
    }
    uint64_t slots = 16;
    while (slots < 2 * E) slots <<= 1;
    // no room now (a quarter of the free memory): the walks take the binary searches -- same
    // walks -- and a later call, with more memory free, tries again
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || slots * 8 > free_b / 4) {
        (void)hipGetLastError();
        return 0;
    }
    unsigned long long *table = nullptr, *filter = nullptr;
    if (hipMalloc((void **)&table, slots * 8) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    // the filter in front of it: one word per 8 edges, rounded up to a power of two
    // (GN2V_WALK_EDGE_FILTER=0: the set alone)
    uint64_t words = 16;
    while (words * 8 < E) words <<= 1;
    const char *fenv = getenv("GN2V_WALK_EDGE_FILTER");
    if ((fenv && *fenv == '0') || hipMalloc((void **)&filter, words * 8) != hipSuccess) {
        (void)hipGetLastError();
        filter = nullptr;
    }
    // Built on `s` and COMPLETE before it is published: a walk launched later on any other
    // stream must never probe a table that is not memset yet (an unset, zero-filled table has no
    // empty slot: the probe loop would not end) or half built (other walks, silently).
    hipError_t e = hipMemsetAsync(table, 0xFF, slots * 8, s);
    if (e == hipSuccess && filter) e = hipMemsetAsync(filter, 0, words * 8, s);
    if (e == hipSuccess) {
        const unsigned blocks =
            (unsigned)std::min<uint64_t>((g->view.n_nodes + 255) / 256, 1u << 20);
        hipLaunchKernelGGL(gn2v::edge_set_kernel, dim3(blocks), dim3(256), 0, s, g->view.row_ptr,
                           g->view.col_idx, g->view.n_nodes, table, slots - 1, filter, words - 1);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        (void)hipFree(table);
        if (filter) (void)hipFree(filter);
        return fail(std::string("building the edge set of the walk sampler: ") +
                    hipGetErrorString(e));
    }
    g->edge_set_tried = true;
    g->edge_set = table;
    g->edge_filter = filter;
    g->view.edge_set = table;
    g->view.edge_mask = slots - 1;
    g->view.edge_filter = filter;
    g->view.filter_mask = words - 1;
    return 0;
}

// The edge records of the walk sampler (walk_kernels.h): 16 B per directed edge (32 B in the typed
// form, for walks with type factors) + 4 B per node -- 3.2 GB (6.4 GB) for the 10 M / 100 M bench
// graph.  Each form is built once per handle, on the first walk that reads it, when it
// fits an eighth of the free memory (a fit of a 100 M-node graph keeps that memory for its tables
// and pair buffers: the walks are 1 % of its time); GN2V_WALK_EDGE_RECORDS=0 keeps the CSR reads
// (same walks either way).
int ensure_edge_records(gn2v_graph *g, hipStream_t s, bool typed) {
    bool &tried = typed ? g->edge_rec_typed_tried : g->edge_rec_tried;
    if (tried) return 0;
    const char *env = getenv("GN2V_WALK_EDGE_RECORDS");
    const uint64_t E = g->view.n_edges, N = g->view.n_nodes;
    if ((env && *env == '0') || E == 0 || E >= gn2v::kRecMaxEdges || N >= 0xFFFFFFFFULL) {
        tried = true;  // never for this handle
        return 0;
    }
    const size_t rec_bytes = (size_t)E * (typed ? 32 : 16);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess ||
        rec_bytes + (g->node_sig ? 0 : (N + 1) * 4) > free_b / 8) {
        (void)hipGetLastError();
        return 0;  // a later call, with more memory free, tries again
    }
    uint4 *rec = nullptr;
    uint32_t *sig = g->node_sig;  // u32[N] signatures, then one flag word
    const bool own_sig = sig == nullptr;
    if ((own_sig && hipMalloc((void **)&sig, (N + 1) * 4) != hipSuccess) ||
        hipMalloc((void **)&rec, rec_bytes) != hipSuccess) {
        (void)hipGetLastError();
        if (own_sig && sig) (void)hipFree(sig);
        return 0;
    }
    // complete before it is published, as the edge set is
    uint32_t flag = 0;
    hipError_t e = hipSuccess;
    if (own_sig) {
        e = hipMemsetAsync(sig + N, 0, 4, s);
        if (e == hipSuccess) {
            const unsigned nb = (unsigned)std::min<uint64_t>((N + 255) / 256, 1u << 20);
            hipLaunchKernelGGL(gn2v::node_sig_kernel, dim3(nb), dim3(256), 0, s, g->view.row_ptr,
                               g->view.col_idx, N, sig, sig + N);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&flag, sig + N, 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (e == hipSuccess && !flag) {
        const unsigned eb = (unsigned)std::min<uint64_t>((E + 255) / 256, 1u << 20);
        if (typed)
            hipLaunchKernelGGL(gn2v::edge_rec_typed_kernel, dim3(eb), dim3(256), 0, s,
                               g->view.row_ptr, g->view.col_idx, sig, g->view.node_types,
                               g->view.edge_types, E, rec);
        else
            hipLaunchKernelGGL(gn2v::edge_rec_kernel, dim3(eb), dim3(256), 0, s, g->view.row_ptr,
                               g->view.col_idx, sig, E, rec);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (e != hipSuccess || flag) {
        (void)hipFree(rec);
        if (own_sig) (void)hipFree(sig);
        if (e != hipSuccess)
            return fail(std::string("building the edge records of the walk sampler: ") +
                        hipGetErrorString(e));
        // a row longer than a record can say: neither form, ever
        g->edge_rec_tried = g->edge_rec_typed_tried = true;
        return 0;
    }
    tried = true;
    g->node_sig = sig;
    g->view.node_sig = sig;
    if (typed) {
        g->edge_rec_typed = rec;
        g->view.edge_rec_typed = rec;
    } else {
        g->edge_rec = rec;
        g->view.edge_rec = rec;
    }
    return 0;
}

__global__ void max_degree_kernel(const uint64_t *__restrict__ row_ptr, uint64_t n,
                                  unsigned long long *__restrict__ out) {
    unsigned long long m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x)
        m = max(m, (unsigned long long)(row_ptr[i + 1] - row_ptr[i]));
    for (int off = 32; off > 0; off >>= 1)
        m = max(m, (unsigned long long)__shfl_xor((long long)m, off));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// the largest out-degree of the graph (one pass over row_ptr, one host read; cached by the caller)
int max_out_degree(gn2v_graph *g, hipStream_t s, uint64_t *out) {
    unsigned long long *d = nullptr, h = 0;
    HIP_TRY(hipMalloc((void **)&d, sizeof(h)));
    hipError_t e = hipMemsetAsync(d, 0, sizeof(h), s);
    if (e == hipSuccess) {
        const uint64_t n = g->view.n_nodes;
        const unsigned blocks = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(max_degree_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s,
                           g->view.row_ptr, n, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(std::string("largest degree: ") + hipGetErrorString(e));
    *out = h;
    return 0;
}

int launch_walks(gn2v_graph *g, const gn2v_walk_params *wp, uint64_t seed, uint64_t epoch,
                 uint64_t first_walk, uint64_t n_walks, uint32_t *d_out, hipStream_t s,
                 uint32_t id_group = 0, uint64_t id_stride = 0) {
    if (n_walks == 0) return 0;
    const gn2v::WalkConsts c = walk_consts(g, wp);
    const bool typed = c.node_bias || c.edge_bias;
    {
        std::lock_guard<std::mutex> lock(g->mu);
        if (c.second_order && ensure_edge_set(g, s)) return 1;
        if (ensure_edge_records(g, s, typed)) return 1;
    }
    const uint64_t blocks = (n_walks + gn2v::kWalkBlock - 1) / gn2v::kWalkBlock;
    if (blocks > 0x7FFFFFFFULL) return fail("too many walks in one launch");
    std::lock_guard<std::mutex> lock(g->mu);
    // max_neighbours acts on rows longer than it: the largest out-degree, once per handle
    bool sub = false;
    if (c.max_neighbours) {
        if (!g->max_out_degree_known) {
            if (max_out_degree(g, s, &g->max_out_degree)) return 1;
            g->max_out_degree_known = true;
        }
        sub = g->max_out_degree > c.max_neighbours;
    }
    EventPair ev;
    if (get_events(g, &ev)) return 1;
    HIP_TRY(hipEventRecord(ev.a, s));
    // (SUB: rows longer than max_neighbours exist -- see walk_kernels.h)
    const bool rec = typed ? g->view.edge_rec_typed != nullptr : g->view.edge_rec != nullptr;
#define GN2V_LAUNCH_WALK(KERNEL, TYPED, SUB)                                                       \
    hipLaunchKernelGGL((gn2v::KERNEL<TYPED, SUB>), dim3((unsigned)blocks), dim3(gn2v::kWalkBlock), \
                       0, s, g->view, c, gn2v::epoch_key(seed, epoch), first_walk, n_walks, d_out, \
                       g->counters, id_group, id_stride)
    switch ((typed ? 4 : 0) | (rec ? 2 : 0) | (sub ? 1 : 0)) {
        case 0: GN2V_LAUNCH_WALK(walk_kernel, false, false); break;
        case 1: GN2V_LAUNCH_WALK(walk_kernel, false, true); break;
        case 2: GN2V_LAUNCH_WALK(walk_rec_kernel, false, false); break;
        case 3: GN2V_LAUNCH_WALK(walk_rec_kernel, false, true); break;
        case 4: GN2V_LAUNCH_WALK(walk_kernel, true, false); break;
        case 5: GN2V_LAUNCH_WALK(walk_kernel, true, true); break;
        case 6: GN2V_LAUNCH_WALK(walk_rec_kernel, true, false); break;
        default: GN2V_LAUNCH_WALK(walk_rec_kernel, true, true); break;
    }
#undef GN2V_LAUNCH_WALK
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev.b, s));
    g->walk_events.push_back(ev);
    g->walk_launches++;
    return 0;
}

int check_train_params(const gn2v_train_params *tp, uint32_t L) {
    if (!tp) return fail("train params are NULL");
    if (tp->d == 0) return fail("embedding size must be strictly positive");
    if (tp->ld < tp->d || (tp->ld & 3)) return fail("ld must be a multiple of 4 and >= d");
    if (tp->ld > 1024) return fail("embedding sizes above 1024 are not supported");
    if (tp->window < 1) return fail("window_size must be >= 1");
    if (tp->min_dist > tp->window) return fail("min_dist must not exceed window_size");
    if (L < 2) return fail("walk_length must be >= 2");
    if (!std::isfinite(tp->lr) || !std::isfinite(tp->clip) || tp->clip <= 0.f)
        return fail("learning rate / clipping value must be finite, clipping value positive");
    return 0;
}

template <int CH>
int launch_train_ch(bool cbow, int wm, bool det, dim3 grid, dim3 block, size_t lds,
                    hipStream_t s, const gn2v::TrainArgs &a) {
#define GN2V_LAUNCH(KERNEL, WM, DT) \
    hipLaunchKernelGGL((gn2v::KERNEL<CH, WM, DT>), grid, block, lds, s, a)
    if (!cbow) {
        if (det)
            GN2V_LAUNCH(sgns_kernel, gn2v::kWriteBack, true);
        else if (wm == gn2v::kAtomic)
            GN2V_LAUNCH(sgns_kernel, gn2v::kAtomic, false);
        else if (wm == gn2v::kWriteBack)
            GN2V_LAUNCH(sgns_kernel, gn2v::kWriteBack, false);
        else
            GN2V_LAUNCH(sgns_kernel, gn2v::kWriteThrough, false);
    } else {
        if (det)
            GN2V_LAUNCH(cbow_kernel, gn2v::kWriteBack, true);
        else if (wm == gn2v::kAtomic)
            GN2V_LAUNCH(cbow_kernel, gn2v::kAtomic, false);
        else if (wm == gn2v::kWriteBack)
            GN2V_LAUNCH(cbow_kernel, gn2v::kWriteBack, false);
        else
            GN2V_LAUNCH(cbow_kernel, gn2v::kWriteThrough, false);
    }
#undef GN2V_LAUNCH
    return 0;
}

int launch_train(gn2v_graph *g, bool cbow, const gn2v_train_params *tp, const gn2v_step_io *io,
                 uint64_t n_walks, uint32_t L, uint64_t seed, uint64_t epoch, uint64_t first_walk,
                 float lr, hipStream_t s) {
    if (check_train_params(tp, L)) return 1;
    if (!io) return fail("step io is NULL");
    if (n_walks == 0) return 0;  // empty batches are legal (and carry NULL walk pointers)
    if (!io->d_walks || !io->d_central || !io->d_contextual)
        return fail("NULL walks / table pointer");
    gn2v::TrainArgs a{};
    a.g = g->view;
    a.walks = io->d_walks;
    a.walk_rows = io->d_walk_rows;
    a.neg_override = io->d_neg_override;
    a.ctx_delta = cbow ? io->d_context_delta : nullptr;
    if (io->d_context_delta && (!cbow || (tp->flags & GN2V_TRAIN_DETERMINISTIC)))
        return fail("d_context_delta is for CBOW in the parallel update modes");
    a.central = io->d_central;
    a.contextual = io->d_contextual;
    float *positive_table = cbow ? io->d_central : io->d_contextual;
    a.negative = io->d_negative ? io->d_negative : positive_table;
    a.split = a.negative != positive_table;
    if (a.split && g->view.n_nodes >= (1ULL << 31))
        return fail("separate negative tables need node ids below 2^31");
    a.neg_pool = io->d_neg_pool;
    a.neg_pool_size = io->neg_pool_size;
    if (a.neg_pool && a.neg_pool_size == 0) return fail("empty negative pool");
    a.neg_id_mul = (io->neg_id_mul == 0 && io->neg_id_add == 0) ? 1u : io->neg_id_mul;
    a.neg_id_add = io->neg_id_add;
    a.counters = g->counters;
    a.n_walks = n_walks;
    a.first_walk = first_walk;
    a.ekey = gn2v::epoch_key(seed, epoch);
    a.L = L;
    a.window = tp->window;
    a.k = tp->k;
    a.ld = tp->ld;
    a.flags = tp->flags & 7u;
    a.min_dist = tp->min_dist ? tp->min_dist : 1;
    a.max_samples = cbow ? (tp->k + 1) : 2 * tp->window * (tp->k + 1);
    a.lr = lr;
    a.clip = tp->clip;

    const bool det = tp->flags & GN2V_TRAIN_DETERMINISTIC;
    // automatic choice: atomics while the tables are small enough for the waves in flight to meet
    // on the same rows all the time -- below 2^16 nodes for SkipGram (which leaves this kernel for
    // the block path at GN2V_BLOCK_PATH_MIN_NODES anyway); for CBOW while a table holds fewer than
    // GN2V_CBOW_STORES_MIN_ELEMENTS floats: narrow rows are trained faster, so more of them are
    // in flight at once and stores need a larger graph to lose nothing (gn2v.h)
    const bool small = cbow ? g->view.n_nodes * (uint64_t)tp->ld < GN2V_CBOW_STORES_MIN_ELEMENTS
                            : g->view.n_nodes < (1ULL << 16);
    const int wm = (tp->flags & GN2V_TRAIN_ATOMIC)          ? gn2v::kAtomic
                   : (tp->flags & GN2V_TRAIN_WRITE_BACK)    ? gn2v::kWriteBack
                   : (tp->flags & GN2V_TRAIN_WRITE_THROUGH) ? gn2v::kWriteThrough
                   : small                                  ? gn2v::kAtomic
                                                            : gn2v::kWriteThrough;
    const int waves_per_block = det ? 1 : gn2v::kTrainBlock / 64;
    const size_t per_wave_words =
        ((size_t)tp->ld + 2 * (size_t)L + 2 * (size_t)a.max_samples + (cbow ? 2 * tp->window : 0) + 3) &
        ~(size_t)3;
    const size_t lds = (size_t)waves_per_block * per_wave_words * 4;
    if (lds > 64 * 1024) return fail("walk_length / window / negatives too large for the LDS plan");
    uint64_t blocks = det ? 1 : (n_walks + waves_per_block - 1) / waves_per_block;
    uint64_t cap = (uint64_t)g->n_cus * 8;
    if (blocks > cap) blocks = cap;
    dim3 grid((unsigned)blocks), block(det ? 64 : gn2v::kTrainBlock);

    // context cache (store modes, walk-ordered SkipGram on whole tables only)
    const uint32_t slots = 2 * tp->window + 1;
    const size_t cache_words = ((size_t)slots * tp->ld + L + 2 * (size_t)a.max_samples +
                                (cbow ? 2 * tp->window + slots : 0) + 2 * slots + 3) &
                               ~(size_t)3;
    const size_t cache_lds = (size_t)waves_per_block * cache_words * 4;
    // LDS budget of a workgroup of the caching kernels: 40 KB leaves room for four of them per CU
    // (160 KB).  The lazy CBOW window may take up to 64 KB -- rows of 132-256 floats, three or two
    // workgroups per CU, which is what their registers allow anyway: + 19 % at d = 200, + 9 % at
    // d = 256 over the uncached kernel on one box; the eager window cache loses there (-5 %).
    constexpr size_t cache_budget = 40 * 1024, lazy_budget = 64 * 1024;
    const bool cacheable = !det && wm != gn2v::kAtomic && !a.split && !a.ctx_delta &&
                           !a.walk_rows && !a.neg_pool && !(tp->flags & GN2V_TRAIN_NO_CTX_CACHE) &&
                           L > 2 * tp->window &&
                           (!cbow || slots <= gn2v::kWinCacheMaxSlots) &&
                           g->view.n_nodes < (1ULL << 30);  // row ids share a word with kCacheBit
    // CBOW, ordinary windows: the lazy form of the window cache (cbow_lazy_kernel.h)
    const size_t lazy_words = ((size_t)(slots + 2) * tp->ld + L + 2 * (size_t)a.max_samples +
                               2 * tp->window + 3 * slots + 3) &
                              ~(size_t)3;
    // Its window takes (2 w + 3) rows of LDS per wave: wide rows and long windows leave a CU's
    // 160 KB fewer waves than its registers would, and workgroups of four waves round that down
    // once more (d = 128, w = 12: 59 KB per workgroup = 8 waves a CU; as single waves 10).  The
    // workgroup is the 4 / 2 / 1 waves that put most waves on a CU; below kLazyMinWaves waves per
    // CU the uncached kernel keeps the path (measured).
    // GN2V_CBOW_LAZY_WAVES forces the workgroup, GN2V_CBOW_LAZY_MIN_WAVES the threshold (A/B).
    static const long lazy_waves_env = [] {
        const char *e = getenv("GN2V_CBOW_LAZY_WAVES");
        return e ? atol(e) : 0L;
    }();
    static const long lazy_min_waves = [] {
        const char *e = getenv("GN2V_CBOW_LAZY_MIN_WAVES");
        return e ? atol(e) : (long)gn2v::kLazyMinWaves;
    }();
    uint32_t lazy_wpb = (uint32_t)waves_per_block, lazy_waves_cu = 0;
    if (!det) {
        // what the kernel's register cap allows (cbow_lazy_kernel.h, __launch_bounds__)
        const uint32_t reg_waves = 4 * gn2v::lazy_min_blocks(tp->ld);
        for (uint32_t wpb : {4u, 2u, 1u}) {
            if (lazy_waves_env && (long)wpb != lazy_waves_env) continue;
            const size_t wg = ((size_t)wpb * lazy_words * 4 + 1023) & ~(size_t)1023;
            if (wg > 64 * 1024) continue;
            const uint32_t waves =
                std::min<uint32_t>((uint32_t)(160 * 1024 / wg) * wpb, reg_waves);
            if (waves > lazy_waves_cu) {
                lazy_waves_cu = waves;
                lazy_wpb = wpb;
            }
        }
    }
    const size_t lazy_lds = (size_t)lazy_wpb * lazy_words * 4;
    // A/B switch, read once: GN2V_CBOW_LAZY=0 keeps cbow_cached_kernel
    static const bool lazy_off = [] {
        const char *e = getenv("GN2V_CBOW_LAZY");
        return e && e[0] == '0';
    }();
    constexpr bool no_full = false;
    const bool use_lazy = cacheable && cbow && a.min_dist == 1 && lazy_lds <= lazy_budget &&
                          (long)lazy_waves_cu >= lazy_min_waves && !lazy_off;
    uint64_t lazy_blocks = (n_walks + lazy_wpb - 1) / lazy_wpb;
    lazy_blocks = std::min<uint64_t>(lazy_blocks, (uint64_t)g->n_cus * 32 / lazy_wpb);
    const dim3 lgrid((unsigned)std::max<uint64_t>(lazy_blocks, 1)), lblock(64 * lazy_wpb);
    const bool use_cache = use_lazy || (cacheable && cache_lds <= cache_budget);
    if (use_cache) {
        if (tp->flags & GN2V_TRAIN_CTX_CACHE_ALL) {
            a.cache_max_degree = 0xFFFFFFFFu;
        } else if (tp->flags & GN2V_TRAIN_CTX_CACHE_NONE) {
            a.cache_max_degree = 0;
        } else {
            // a row is cached by ~(resident waves x window) positions at a time; keep the expected
            // number of waves holding the same row at once below 0.1
            const double holders = (double)g->n_cus * 24.0 * slots;
            const double limit = 0.1 * (double)g->view.n_edges / holders;
            a.cache_max_degree = limit < 1.0 ? 1u : (limit > 4e9 ? 0xFFFFFFFEu : (uint32_t)limit);
        }
    }

    std::lock_guard<std::mutex> lock(g->mu);
    EventPair ev;
    if (get_events(g, &ev)) return 1;
    HIP_TRY(hipEventRecord(ev.a, s));
    const uint32_t nchunks = tp->ld / 4;
    if (use_cache) {
#define GN2V_CACHED(CH)                                                                        \
    do {                                                                                       \
        if (use_lazy && wm == gn2v::kWriteBack)                                                \
            hipLaunchKernelGGL((gn2v::cbow_lazy_kernel<CH, gn2v::kWriteBack>), lgrid, lblock,  \
                               lazy_lds, s, a);                                                \
        else if (use_lazy && tp->ld == CH * 64 && !no_full)                                    \
            hipLaunchKernelGGL((gn2v::cbow_lazy_kernel<CH, gn2v::kWriteThrough, true>), lgrid, \
                               lblock, lazy_lds, s, a);                                        \
        else if (use_lazy)                                                                     \
            hipLaunchKernelGGL((gn2v::cbow_lazy_kernel<CH, gn2v::kWriteThrough>), lgrid,       \
                               lblock, lazy_lds, s, a);                                        \
        else if (cbow && wm == gn2v::kWriteBack)                                               \
            hipLaunchKernelGGL((gn2v::cbow_cached_kernel<CH, gn2v::kWriteBack>), grid, block,  \
                               cache_lds, s, a);                                               \
        else if (cbow)                                                                         \
            hipLaunchKernelGGL((gn2v::cbow_cached_kernel<CH, gn2v::kWriteThrough>), grid,      \
                               block, cache_lds, s, a);                                        \
        else if (wm == gn2v::kWriteBack)                                                       \
            hipLaunchKernelGGL((gn2v::sgns_cached_kernel<CH, gn2v::kWriteBack>), grid, block,  \
                               cache_lds, s, a);                                               \
        else                                                                                   \
            hipLaunchKernelGGL((gn2v::sgns_cached_kernel<CH, gn2v::kWriteThrough>), grid,      \
                               block, cache_lds, s, a);                                        \
    } while (0)
        if (nchunks <= 16)
            GN2V_CACHED(1);
        else if (nchunks <= 32)
            GN2V_CACHED(2);
        else if (nchunks <= 64)
            GN2V_CACHED(4);
        else if (nchunks <= 128)
            GN2V_CACHED(8);
        else
            GN2V_CACHED(16);
#undef GN2V_CACHED
    } else if (nchunks <= 16)
        launch_train_ch<1>(cbow, wm, det, grid, block, lds, s, a);
    else if (nchunks <= 32)
        launch_train_ch<2>(cbow, wm, det, grid, block, lds, s, a);
    else if (nchunks <= 64)
        launch_train_ch<4>(cbow, wm, det, grid, block, lds, s, a);
    else if (nchunks <= 128)
        launch_train_ch<8>(cbow, wm, det, grid, block, lds, s, a);
    else  // 512 < ld <= 1024: 64 registers per row, the compiler parks rows in the AGPRs
        launch_train_ch<16>(cbow, wm, det, grid, block, lds, s, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev.b, s));
    g->train_events.push_back(ev);
    g->train_launches++;
    return 0;
}

}  // namespace

namespace gn2v_host {
void release_kept_buffers(gn2v_graph *g) {
    std::lock_guard<std::mutex> lock(g->kept_mu);
    for (auto &b : g->kept_buffers) (void)hipFree(b.first);
    g->kept_buffers.clear();
    g->kept_bytes = 0;
}

// What the walks of `wp` need beyond the CSR (the second-order sampler's edge set), allocated and
// built NOW: gn2v_train_blocks calls this before it sizes its rounds from the free memory, so
// that the set's bytes (3.4 GB on the bench graph, 18 GB at 100 M nodes) are neither planned
// twice nor found missing at the first walk.
int prepare_walk_sampler(gn2v_graph *g, const gn2v_walk_params *wp, hipStream_t s) {
    if (check_walk_params(wp)) return 1;
    const gn2v::WalkConsts c = walk_consts(g, wp);
    std::lock_guard<std::mutex> lock(g->mu);
    if (c.second_order && ensure_edge_set(g, s)) return 1;
    if (ensure_edge_records(g, s, c.node_bias || c.edge_bias)) return 1;
    return 0;
}
}  // namespace gn2v_host

extern "C" {

int gn2v_version(void) { return GN2V_VERSION; }

const char *gn2v_last_error(void) { return g_err.c_str(); }

int gn2v_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int gn2v_graph_create(const uint64_t *row_ptr, const uint32_t *col_idx, const float *cumw,
                      const uint32_t *sources, uint64_t n_nodes, uint64_t n_edges,
                      uint64_t n_sources, uint32_t flags, int device, gn2v_graph **out) {
    if (!out) return fail("out is NULL");
    *out = nullptr;
    if (!row_ptr || !col_idx) return fail("row_ptr / col_idx are NULL");
    if (n_nodes == 0) return fail("the graph has no nodes");
    if (n_nodes >= 0xFFFFFFFFULL) return fail("node ids must fit in 32 bits (minus the sentinel)");
    if (n_edges == 0) return fail("the graph has no edges");
    if (sources == nullptr) n_sources = n_nodes;
    if (n_sources == 0) return fail("the graph has no source nodes");
    if (gn2v_device_count() <= device || device < 0)
        return fail("no HIP device " + std::to_string(device) +
                    " is visible: the gn2v engine requires an AMD GPU (there is no CPU fallback)");
    DeviceGuard guard(device);
    if (!guard.ok()) return fail("cannot select HIP device " + std::to_string(device));
    gn2v_graph *g = new gn2v_graph();
    g->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        g->n_cus = prop.multiProcessorCount;
    auto cleanup = [&]() { gn2v_graph_destroy(g); };
    if (flags & GN2V_GRAPH_DEVICE_PTRS) {
        g->view.row_ptr = row_ptr;
        g->view.col_idx = col_idx;
        g->view.cumw = cumw;
        g->view.sources = sources;
    } else {
        g->owns = true;
#define GN2V_UPLOAD(dst, src, bytes)                                                       \
    do {                                                                                   \
        if (hipMalloc(&(dst), (bytes)) != hipSuccess ||                                    \
            hipMemcpy((dst), (src), (bytes), hipMemcpyHostToDevice) != hipSuccess) {       \
            cleanup();                                                                     \
            return fail("uploading the CSR graph to the device failed (out of memory?)"); \
        }                                                                                  \
    } while (0)
        GN2V_UPLOAD(g->own_row_ptr, row_ptr, (n_nodes + 1) * sizeof(uint64_t));
        GN2V_UPLOAD(g->own_col_idx, col_idx, n_edges * sizeof(uint32_t));
        if (cumw) GN2V_UPLOAD(g->own_cumw, cumw, n_edges * sizeof(float));
        if (sources) GN2V_UPLOAD(g->own_sources, sources, n_sources * sizeof(uint32_t));
#undef GN2V_UPLOAD
        g->view.row_ptr = (const uint64_t *)g->own_row_ptr;
        g->view.col_idx = (const uint32_t *)g->own_col_idx;
        g->view.cumw = (const float *)g->own_cumw;
        g->view.sources = (const uint32_t *)g->own_sources;
    }
    g->view.n_nodes = n_nodes;
    g->view.n_edges = n_edges;
    g->view.n_sources = n_sources;
    g->view.symmetric = (flags & GN2V_GRAPH_SYMMETRIC) ? 1u : 0u;
    if (hipMalloc(&g->counters, 4 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(g->counters, 0, 4 * sizeof(unsigned long long)) != hipSuccess) {
        cleanup();
        return fail("allocating device counters failed");
    }
    if (hipMalloc(&g->cursors, kCursorRing * kCursorWords * sizeof(unsigned long long)) !=
        hipSuccess) {
        cleanup();
        return fail("allocating device counters failed");
    }
    {   // which XCDs do workgroups land on?  (8 on an MI355X in SPX mode; the block trainer uses
        // write-back stores only for rows that exactly one of them touches)
        unsigned int *mask = reinterpret_cast<unsigned int *>(g->cursors);
        unsigned int seen = 0;
        if (hipMemset(mask, 0, sizeof(unsigned int)) == hipSuccess) {
            hipLaunchKernelGGL(gn2v::xcc_probe_kernel, dim3(g->n_cus * 16), dim3(64), 0,
                               (hipStream_t)0, mask);
            if (hipGetLastError() == hipSuccess &&
                hipMemcpy(&seen, mask, sizeof(seen), hipMemcpyDeviceToHost) == hipSuccess) {
                int n = 0;
                while (seen & (1u << n)) ++n;
                g->n_xcds = (seen == (1u << n) - 1u) ? n : 0;  // contiguous ids only
            }
        }
        (void)hipGetLastError();
    }
    *out = g;
    if (const char *reserve = getenv("GN2V_RESERVE_CUS")) {
        const int k = atoi(reserve);
        if (k > 0 && gn2v_graph_reserve_cus(g, (uint32_t)k, nullptr)) {
            *out = nullptr;
            gn2v_graph_destroy(g);
            return 1;
        }
    }
    return 0;
}

int gn2v_graph_xcds(gn2v_graph *g) { return g ? g->n_xcds : 0; }

int gn2v_graph_reserve_cus(gn2v_graph *g, uint32_t cus_per_xcd, uint32_t *active_per_xcd) {
    if (!g) return fail("graph handle is NULL");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    std::lock_guard<std::mutex> lock(g->mu);
    HIP_TRY(hipDeviceSynchronize());
    if (g->train_stream) {
        (void)hipStreamDestroy(g->train_stream);
        g->train_stream = nullptr;
    }
    g->reserved_cus = 0;
    if (active_per_xcd) std::memset(active_per_xcd, 0, 16 * sizeof(uint32_t));
    if (cus_per_xcd == 0) return 0;
    const uint32_t xcds = g->n_xcds > 0 ? (uint32_t)g->n_xcds : 1u;
    const uint32_t n_cus = (uint32_t)g->n_cus;
    if (cus_per_xcd * xcds >= n_cus) return fail("cannot reserve every compute unit");
    // bit i of the mask = logical CU i; logical CUs are dealt round robin over the XCDs, so the
    // first cus_per_xcd * xcds bits take cus_per_xcd CUs from each -- checked below, not assumed
    std::vector<uint32_t> mask((n_cus + 31) / 32, 0u);
    for (uint32_t i = cus_per_xcd * xcds; i < n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t st = nullptr;
    HIP_TRY(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    unsigned int *seen = reinterpret_cast<unsigned int *>(g->cursors);
    uint32_t host[16 * 8] = {0};
    hipError_t e = hipMemsetAsync(seen, 0, sizeof(host), st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(gn2v::cu_probe_kernel, dim3(n_cus * 32), dim3(64), 0, st, seen);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, seen, sizeof(host), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(st);
        return fail(std::string("probing the CU mask failed: ") + hipGetErrorString(e));
    }
    const uint32_t want = n_cus / xcds - cus_per_xcd;
    bool ok = true;
    for (uint32_t x = 0; x < 16; ++x) {
        uint32_t active = 0;
        for (uint32_t w = 0; w < 8; ++w) active += (uint32_t)__builtin_popcount(host[x * 8 + w]);
        if (active_per_xcd) active_per_xcd[x] = active;
        if (x < xcds ? active > want || active == 0 : active != 0) ok = false;
    }
    if (!ok) {
        (void)hipStreamDestroy(st);
        return fail("the CU mask did not leave " + std::to_string(cus_per_xcd) +
                    " compute unit(s) of every XCD free (logical CU numbering differs from the "
                    "assumed round robin over the XCDs)");
    }
    if (!g->ts_in) {
        HIP_TRY(hipEventCreateWithFlags(&g->ts_in, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g->ts_out, hipEventDisableTiming));
    }
    g->train_stream = st;
    g->reserved_cus = cus_per_xcd;
    return 0;
}

int gn2v_graph_set_types(gn2v_graph *g, const uint32_t *node_types,
                         const uint32_t *edge_types) {
    if (!g) return fail("graph handle is NULL");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    std::lock_guard<std::mutex> lock(g->mu);
    // walks already queued may still read the old arrays
    HIP_TRY(hipDeviceSynchronize());
    // the typed edge records carry the old types: the next typed walk builds them again
    if (g->edge_rec_typed) (void)hipFree(g->edge_rec_typed);
    g->edge_rec_typed = nullptr;
    g->view.edge_rec_typed = nullptr;
    g->edge_rec_typed_tried = false;
    if (!g->owns) {
        g->view.node_types = node_types;
        g->view.edge_types = edge_types;
        return 0;
    }
    struct Kind {
        const uint32_t *src;
        void **own;
        const uint32_t **view;
        uint64_t count;
    } kinds[2] = {{node_types, &g->own_node_types, &g->view.node_types, g->view.n_nodes},
                  {edge_types, &g->own_edge_types, &g->view.edge_types, g->view.n_edges}};
    for (auto &k : kinds) {
        if (*k.own) (void)hipFree(*k.own);
        *k.own = nullptr;
        *k.view = nullptr;
        if (!k.src) continue;
        const size_t bytes = k.count * sizeof(uint32_t);
        if (hipMalloc(k.own, bytes) != hipSuccess ||
            hipMemcpy(*k.own, k.src, bytes, hipMemcpyHostToDevice) != hipSuccess) {
            if (*k.own) (void)hipFree(*k.own);
            *k.own = nullptr;
            return fail("uploading the type ids to the device failed (out of memory?)");
        }
        *k.view = (const uint32_t *)*k.own;
    }
    return 0;
}

int gn2v_graph_destroy(gn2v_graph *g) {
    if (!g) return 0;
    DeviceGuard guard(g->device);
    (void)hipDeviceSynchronize();
    if (g->own_row_ptr) (void)hipFree(g->own_row_ptr);
    if (g->own_col_idx) (void)hipFree(g->own_col_idx);
    if (g->own_cumw) (void)hipFree(g->own_cumw);
    if (g->own_sources) (void)hipFree(g->own_sources);
    if (g->own_node_types) (void)hipFree(g->own_node_types);
    if (g->own_edge_types) (void)hipFree(g->own_edge_types);
    if (g->counters) (void)hipFree(g->counters);
    if (g->part_ptrs_dev) (void)hipFree(g->part_ptrs_dev);
    if (g->indeg) (void)hipFree(g->indeg);
    release_kept_buffers(g);
    if (g->prep_stream) {
        (void)hipStreamDestroy(g->prep_stream);
        for (int i = 0; i < 2; ++i) {
            (void)hipEventDestroy(g->prep_done[i]);
            (void)hipEventDestroy(g->train_done[i]);
        }
    }
    for (int i = 0; i < 2; ++i) {
        if (g->lane_stream[i]) (void)hipStreamDestroy(g->lane_stream[i]);
        if (g->lane_done[i]) (void)hipEventDestroy(g->lane_done[i]);
    }
    if (g->lane_start) (void)hipEventDestroy(g->lane_start);
    if (g->cursors) (void)hipFree(g->cursors);
    if (g->lpt) (void)hipFree(g->lpt);
    if (g->lpt_temp) (void)hipFree(g->lpt_temp);
    if (g->edge_set) (void)hipFree(g->edge_set);
    if (g->edge_filter) (void)hipFree(g->edge_filter);
    if (g->edge_rec) (void)hipFree(g->edge_rec);
    if (g->edge_rec_typed) (void)hipFree(g->edge_rec_typed);
    if (g->node_sig) (void)hipFree(g->node_sig);
    if (g->train_stream) (void)hipStreamDestroy(g->train_stream);
    if (g->ts_in) (void)hipEventDestroy(g->ts_in);
    if (g->ts_out) (void)hipEventDestroy(g->ts_out);
    for (auto *v : {&g->train_events, &g->walk_events, &g->free_events})
        for (auto &ev : *v) {
            (void)hipEventDestroy(ev.a);
            (void)hipEventDestroy(ev.b);
        }
    delete g;
    return 0;
}

int gn2v_ba_edges(uint64_t n_nodes, uint32_t m, uint64_t seed, uint32_t *d_src, uint32_t *d_dst,
                  void *stream) {
    DeviceGuard guard(DeviceGuard::of_pointer(d_src));
    if (n_nodes < 2 || m < 1) return fail("need n_nodes >= 2 and m >= 1");
    if (n_nodes >= 0xFFFFFFFFULL) return fail("node ids must fit in 32 bits");
    if (!d_src || !d_dst) return fail("NULL output pointer");
    const uint64_t n_e = (n_nodes - 1) * (uint64_t)m;
    const unsigned blocks = (unsigned)std::min<uint64_t>((n_e + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(gn2v::ba_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       gn2v::mix64(seed ^ gn2v::kTagBA), n_e, m, d_src, d_dst);
    HIP_TRY(hipGetLastError());
    return 0;
}

int gn2v_walks(gn2v_graph *g, const gn2v_walk_params *wp, uint64_t seed, uint64_t epoch,
               uint64_t first_walk, uint64_t n_walks, uint32_t *d_out, void *stream) {
    if (!g) return fail("graph handle is NULL");
    if (check_walk_params(wp)) return 1;
    if (n_walks == 0) return 0;
    if (!d_out) return fail("NULL output pointer");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    return launch_walks(g, wp, seed, epoch, first_walk, n_walks, d_out, (hipStream_t)stream);
}

int gn2v_walks_strided(gn2v_graph *g, const gn2v_walk_params *wp, uint64_t seed, uint64_t epoch,
                       uint64_t first_walk, uint64_t n_walks, uint32_t group, uint64_t stride,
                       uint32_t *d_out, void *stream) {
    if (!g) return fail("graph handle is NULL");
    if (check_walk_params(wp)) return 1;
    if (group == 0) return fail("group must be at least 1");
    if (n_walks == 0) return 0;
    if (!d_out) return fail("NULL output pointer");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    return launch_walks(g, wp, seed, epoch, first_walk, n_walks, d_out, (hipStream_t)stream, group,
                        stride);
}

int gn2v_window_batch(const uint32_t *d_walks, uint64_t n_walks, uint32_t walk_length,
                      uint32_t window, int32_t *d_contexts, int32_t *d_words, void *stream) {
    DeviceGuard guard(DeviceGuard::of_pointer(d_walks));
    if (window < 1 || walk_length <= 2 * window)
        return fail("walk_length must exceed 2 * window_size");
    const uint64_t n = n_walks * (walk_length - 2 * window);
    if (n == 0) return 0;
    if (!d_walks || !d_contexts || !d_words) return fail("NULL pointer");
    const unsigned blocks = (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(gn2v::window_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       d_walks, n_walks, walk_length, window, d_contexts, d_words);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int launch_pairs(const uint32_t *d_walks, uint64_t n_walks, uint32_t walk_length,
                        uint32_t window, uint32_t min_dist, uint32_t *d_pairs, void *stream) {
    DeviceGuard guard(DeviceGuard::of_pointer(d_walks));
    if (window < 1 || walk_length < 2) return fail("need window_size >= 1 and walk_length >= 2");
    const uint64_t n = n_walks * walk_length * 2 * window;
    if (n == 0) return 0;
    if (!d_walks || !d_pairs) return fail("NULL pointer");
    const unsigned blocks = (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(gn2v::pairs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_walks,
                       n_walks, walk_length, window, min_dist ? min_dist : 1u, d_pairs);
    HIP_TRY(hipGetLastError());
    return 0;
}

int gn2v_walk_pairs(const uint32_t *d_walks, uint64_t n_walks, uint32_t walk_length,
                    uint32_t window, uint32_t min_dist, uint32_t *d_pairs, void *stream) {
    return launch_pairs(d_walks, n_walks, walk_length, window, min_dist, d_pairs, stream);
}

int gn2v_cooc_slots(const uint32_t *d_walks, uint64_t n_walks, uint32_t walk_length,
                    uint32_t window, uint32_t min_dist, uint64_t *d_keys, uint64_t *d_weights,
                    void *stream) {
    DeviceGuard guard(DeviceGuard::of_pointer(d_walks));
    if (window < 1 || walk_length < 2) return fail("need window_size >= 1 and walk_length >= 2");
    const uint64_t n = n_walks * walk_length * 2 * window;
    if (n == 0) return 0;
    if (!d_walks || !d_keys || !d_weights) return fail("NULL pointer");
    const unsigned blocks = (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(gn2v::cooc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_walks,
                       n_walks, walk_length, window, min_dist ? min_dist : 1u,
                       (unsigned long long *)d_keys, (unsigned long long *)d_weights);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C++" {
template <int CH>
static void launch_glove_ch(int wm, bool det, dim3 grid, dim3 block, size_t lds, hipStream_t s,
                            const gn2v::GloveArgs &a) {
    if (det)
        hipLaunchKernelGGL((gn2v::glove_kernel<CH, gn2v::kWriteBack, true>), grid, block, 0, s, a);
    else if (wm == gn2v::kAtomic)
        hipLaunchKernelGGL((gn2v::glove_kernel<CH, gn2v::kAtomic, false>), grid, block, lds, s, a);
    else if (wm == gn2v::kWriteBack)
        hipLaunchKernelGGL((gn2v::glove_kernel<CH, gn2v::kWriteBack, false>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((gn2v::glove_kernel<CH, gn2v::kWriteThrough, false>), grid, block, 0, s, a);
}
}  // extern "C++"

int gn2v_glove_step(gn2v_graph *g, const gn2v_glove_io *io, uint64_t n_entries, uint32_t d,
                    uint32_t ld, float lr, uint32_t flags, void *stream) {
    if (!g) return fail("graph handle is NULL");
    if (!io) return fail("glove io is NULL");
    if (d == 0) return fail("embedding size must be strictly positive");
    if (ld < d || (ld & 3)) return fail("ld must be a multiple of 4 and >= d");
    if (ld > 1024) return fail("embedding sizes above 1024 are not supported");
    if (!std::isfinite(lr)) return fail("learning rate must be finite");
    if (n_entries == 0) return 0;
    if (!io->d_rows || !io->d_cols || !io->d_logx || !io->d_fx || !io->d_central ||
        !io->d_contextual || !io->d_bias_central || !io->d_bias_contextual)
        return fail("NULL entry / table / bias pointer");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    gn2v::GloveArgs a{};
    a.rows = io->d_rows;
    a.cols = io->d_cols;
    a.logx = io->d_logx;
    a.fx = io->d_fx;
    a.central = io->d_central;
    a.contextual = io->d_contextual;
    a.bias_c = io->d_bias_central;
    a.bias_x = io->d_bias_contextual;
    a.n_entries = n_entries;
    a.ld = ld;
    a.lr = lr;
    const bool det = flags & GN2V_TRAIN_DETERMINISTIC;
    const int wm = (flags & GN2V_TRAIN_ATOMIC)          ? gn2v::kAtomic
                   : (flags & GN2V_TRAIN_WRITE_BACK)    ? gn2v::kWriteBack
                   : (flags & GN2V_TRAIN_WRITE_THROUGH) ? gn2v::kWriteThrough
                   : g->view.n_nodes < (1ULL << 16)     ? gn2v::kAtomic
                                                        : gn2v::kWriteThrough;
    const int waves_per_block = det ? 1 : gn2v::kGloveBlock / 64;
    const size_t lds = (!det && wm == gn2v::kAtomic) ? (size_t)waves_per_block * 4 * ld * 4 : 0;
    uint64_t blocks = det ? 1 : (n_entries + 16 * waves_per_block - 1) / (16 * waves_per_block);
    uint64_t cap = (uint64_t)g->n_cus * 8;
    // records of one row are trained from the same stale copy of it by concurrent waves: at most
    // one wave per table row on average (binds on tiny graphs only, as for the SkipGram records)
    const uint64_t max_blocks = std::max<uint64_t>(1, g->view.n_nodes / waves_per_block);
    if (!det && cap > max_blocks) cap = max_blocks;
    if (blocks > cap) blocks = cap;
    dim3 grid((unsigned)blocks), block(det ? 64 : gn2v::kGloveBlock);
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(g->mu);
    EventPair ev;
    if (get_events(g, &ev)) return 1;
    HIP_TRY(hipEventRecord(ev.a, s));
    const uint32_t nchunks = ld / 4;
    if (nchunks <= 16)
        launch_glove_ch<1>(wm, det, grid, block, lds, s, a);
    else if (nchunks <= 32)
        launch_glove_ch<2>(wm, det, grid, block, lds, s, a);
    else if (nchunks <= 64)
        launch_glove_ch<4>(wm, det, grid, block, lds, s, a);
    else if (nchunks <= 128)
        launch_glove_ch<8>(wm, det, grid, block, lds, s, a);
    else
        launch_glove_ch<16>(wm, det, grid, block, lds, s, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev.b, s));
    g->train_events.push_back(ev);
    g->train_launches++;
    return 0;
}

int gn2v_init_table(float *d_table, uint64_t n_rows, uint32_t d, uint32_t ld, uint64_t seed,
                    uint32_t table_id, float scale, void *stream) {
    DeviceGuard guard(DeviceGuard::of_pointer(d_table));
    if (d == 0 || ld < d) return fail("need 0 < d <= ld");
    const uint64_t n = n_rows * ld;
    if (n == 0) return 0;
    if (!d_table) return fail("NULL table pointer");
    const unsigned blocks = (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(gn2v::init_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_table,
                       n_rows, d, ld, gn2v::mix64(seed ^ (gn2v::kTagInit + table_id)), scale);
    HIP_TRY(hipGetLastError());
    return 0;
}

int gn2v_step(gn2v_graph *g, const gn2v_train_params *tp, const gn2v_step_io *io,
              uint64_t n_walks, uint32_t walk_length, uint64_t seed, uint64_t epoch,
              uint64_t first_walk, float lr, void *stream) {
    if (!g) return fail("graph handle is NULL");
    if (!tp) return fail("train params are NULL");
    if (tp->model > GN2V_MODEL_CBOW) return fail("unknown model id");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    return launch_train(g, tp->model == GN2V_MODEL_CBOW, tp, io, n_walks, walk_length, seed, epoch,
                        first_walk, lr, (hipStream_t)stream);
}

static int simple_step(gn2v_graph *g, bool cbow, const gn2v_train_params *tp,
                       const uint32_t *d_walks, uint64_t n_walks, uint32_t walk_length,
                       uint64_t seed, uint64_t epoch, uint64_t first_walk, float lr,
                       float *d_central, float *d_contextual, const uint32_t *d_neg_override,
                       void *stream) {
    if (!g) return fail("graph handle is NULL");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    gn2v_step_io io{};
    io.d_walks = d_walks;
    io.d_central = d_central;
    io.d_contextual = d_contextual;
    io.d_neg_override = d_neg_override;
    return launch_train(g, cbow, tp, &io, n_walks, walk_length, seed, epoch, first_walk, lr,
                        (hipStream_t)stream);
}

int gn2v_sgns_step(gn2v_graph *g, const gn2v_train_params *tp, const uint32_t *d_walks,
                   uint64_t n_walks, uint32_t walk_length, uint64_t seed, uint64_t epoch,
                   uint64_t first_walk, float lr, float *d_central, float *d_contextual,
                   const uint32_t *d_neg_override, void *stream) {
    return simple_step(g, false, tp, d_walks, n_walks, walk_length, seed, epoch, first_walk, lr,
                       d_central, d_contextual, d_neg_override, stream);
}

int gn2v_cbow_step(gn2v_graph *g, const gn2v_train_params *tp, const uint32_t *d_walks,
                   uint64_t n_walks, uint32_t walk_length, uint64_t seed, uint64_t epoch,
                   uint64_t first_walk, float lr, float *d_central, float *d_contextual,
                   const uint32_t *d_neg_override, void *stream) {
    return simple_step(g, true, tp, d_walks, n_walks, walk_length, seed, epoch, first_walk, lr,
                       d_central, d_contextual, d_neg_override, stream);
}

int gn2v_edge_embedding(const float *d_src_table, const float *d_dst_table, uint32_t d, uint32_t ld,
                        const uint32_t *d_src_ids, const uint32_t *d_dst_ids, uint64_t n_edges,
                        uint32_t method, float *d_out, uint32_t out_ld, void *stream) {
    DeviceGuard guard(DeviceGuard::of_pointer(d_out));
    if (d == 0 || ld < d || (ld & 3) || ld > 1024)
        return fail("need 0 < d <= ld <= 1024, ld % 4 == 0");
    if (method >= gn2v::kEdgeMethodCount) return fail("unknown edge embedding method");
    if (n_edges == 0) return 0;  // empty edge lists are legal (and carry NULL pointers)
    if (!d_src_table || !d_dst_table || !d_src_ids || !d_dst_ids || !d_out)
        return fail("NULL pointer");
    const uint32_t need = method == gn2v::kConcatenate ? 2 * d
                          : (method == gn2v::kL2Distance || method == gn2v::kCosineSimilarity) ? 1
                                                                                               : d;
    if (out_ld < need) return fail("out_ld is too small for this method");
    if (n_edges == 0) return 0;
    const unsigned blocks = (unsigned)std::min<uint64_t>((n_edges + 15) / 16, 256 * 16);
    hipStream_t s = (hipStream_t)stream;
    const uint32_t nchunks = ld / 4;
#define GN2V_EDGE(CH)                                                                         \
    hipLaunchKernelGGL((gn2v::edge_embedding_kernel<CH>), dim3(blocks), dim3(256), 0, s,       \
                       d_src_table, d_dst_table, d, ld, d_src_ids, d_dst_ids, n_edges, method, \
                       d_out, out_ld)
    if (nchunks <= 16)
        GN2V_EDGE(1);
    else if (nchunks <= 32)
        GN2V_EDGE(2);
    else if (nchunks <= 64)
        GN2V_EDGE(4);
    else if (nchunks <= 128)
        GN2V_EDGE(8);
    else
        GN2V_EDGE(16);
#undef GN2V_EDGE
    HIP_TRY(hipGetLastError());
    return 0;
}

static bool nchunks_is_32(uint32_t ld) { return ld / 4 > 16 && ld / 4 <= 32; }

int gn2v_touch_rows(float *d_table, uint32_t ld, const uint32_t *d_ids, uint64_t n, uint32_t flags,
                    void *stream) {
    DeviceGuard guard(DeviceGuard::of_pointer(d_table));
    if (!d_table || !d_ids) return fail("NULL pointer");
    if (ld == 0 || (ld & 3) || ld > 1024) return fail("ld must be a multiple of 4 in [4, 1024]");
    if (n == 0) return 0;
    const int wm = (flags & GN2V_TRAIN_ATOMIC)       ? gn2v::kAtomic
                   : (flags & GN2V_TRAIN_WRITE_BACK) ? gn2v::kWriteBack
                                                     : gn2v::kWriteThrough;
    const unsigned blocks = (unsigned)std::min<uint64_t>((n + 15) / 16, 256 * 8);
    hipStream_t s = (hipStream_t)stream;
    if ((flags & 512u) && nchunks_is_32(ld)) {  // experiment: lane-contiguous atomics
        hipLaunchKernelGGL((gn2v::touch_rows_atomic_contig_kernel<2>), dim3(blocks),
                           dim3(gn2v::kTrainBlock), 0, s, d_table, ld, d_ids, n);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if ((flags & 256u) && nchunks_is_32(ld)) {  // experiment: 2 rounds in flight (d = 128 only)
        if (wm == gn2v::kWriteBack)
            hipLaunchKernelGGL((gn2v::touch_rows2_kernel<2, gn2v::kWriteBack>), dim3(blocks),
                               dim3(gn2v::kTrainBlock), 0, s, d_table, ld, d_ids, n);
        else
            hipLaunchKernelGGL((gn2v::touch_rows2_kernel<2, gn2v::kWriteThrough>), dim3(blocks),
                               dim3(gn2v::kTrainBlock), 0, s, d_table, ld, d_ids, n);
        HIP_TRY(hipGetLastError());
        return 0;
    }
#define GN2V_TOUCH(CH)                                                                          \
    do {                                                                                        \
        if (wm == gn2v::kAtomic)                                                                \
            hipLaunchKernelGGL((gn2v::touch_rows_kernel<CH, gn2v::kAtomic>), dim3(blocks),      \
                               dim3(gn2v::kTrainBlock), 0, s, d_table, ld, d_ids, n);           \
        else if (wm == gn2v::kWriteBack)                                                        \
            hipLaunchKernelGGL((gn2v::touch_rows_kernel<CH, gn2v::kWriteBack>), dim3(blocks),   \
                               dim3(gn2v::kTrainBlock), 0, s, d_table, ld, d_ids, n);           \
        else                                                                                    \
            hipLaunchKernelGGL((gn2v::touch_rows_kernel<CH, gn2v::kWriteThrough>), dim3(blocks), \
                               dim3(gn2v::kTrainBlock), 0, s, d_table, ld, d_ids, n);           \
    } while (0)
    const uint32_t nchunks = ld / 4;
    if (nchunks <= 16)
        GN2V_TOUCH(1);
    else if (nchunks <= 32)
        GN2V_TOUCH(2);
    else if (nchunks <= 64)
        GN2V_TOUCH(4);
    else if (nchunks <= 128)
        GN2V_TOUCH(8);
    else
        GN2V_TOUCH(16);
#undef GN2V_TOUCH
    HIP_TRY(hipGetLastError());
    return 0;
}

int gn2v_graph_release_buffers(gn2v_graph *g) {
    if (!g) return fail("graph handle is NULL");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    HIP_TRY(hipDeviceSynchronize());
    release_kept_buffers(g);
    return 0;
}

int gn2v_graph_walk_accel(gn2v_graph *g) {
    if (!g) return -fail("graph handle is NULL");
    std::lock_guard<std::mutex> lock(g->mu);
    return (g->edge_set ? GN2V_WALK_ACCEL_EDGE_SET : 0) |
           (g->edge_filter ? GN2V_WALK_ACCEL_FILTER : 0) |
           (g->edge_rec ? GN2V_WALK_ACCEL_RECORDS : 0) |
           (g->edge_rec_typed ? GN2V_WALK_ACCEL_TYPED_RECORDS : 0);
}

int gn2v_stats_reset(gn2v_graph *g, void *stream) {
    if (!g) return fail("graph handle is NULL");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    std::lock_guard<std::mutex> lock(g->mu);
    if (fold_events(g)) return 1;
    g->train_ms = g->walk_ms = 0.0;
    g->train_launches = g->walk_launches = 0;
    g->resident_launches = g->resident_record = 0;
    HIP_TRY(hipMemsetAsync(g->counters, 0, 4 * sizeof(unsigned long long), (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int gn2v_stats_read(gn2v_graph *g, gn2v_stats *stats, void *stream) {
    if (!g || !stats) return fail("NULL handle / stats");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    std::lock_guard<std::mutex> lock(g->mu);
    if (fold_events(g)) return 1;
    unsigned long long h[4];
    HIP_TRY(hipMemcpy(h, g->counters, sizeof(h), hipMemcpyDeviceToHost));
    stats->pairs = h[0];
    stats->walk_steps = h[1];
    stats->centres = h[2];
    stats->train_ms = g->train_ms;
    stats->walk_ms = g->walk_ms;
    stats->train_launches = g->train_launches;
    stats->walk_launches = g->walk_launches;
    stats->resident_launches = g->resident_launches;
    stats->resident_record = g->resident_record;
    return 0;
}

int gn2v_train(gn2v_graph *g, const gn2v_walk_params *wp, const gn2v_train_params *tp,
               uint64_t seed, uint64_t max_walks_per_epoch, float *d_central,
               float *d_contextual, gn2v_stats *stats, void *stream) {
    if (!g) return fail("graph handle is NULL");
    if (check_walk_params(wp)) return 1;
    if (check_train_params(tp, wp->walk_length)) return 1;
    if (tp->model > GN2V_MODEL_CBOW) return fail("unknown model id");
    if (!d_central || !d_contextual) return fail("NULL table pointer");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    hipStream_t s = (hipStream_t)stream;
    const bool cbow = tp->model == GN2V_MODEL_CBOW;
    const uint32_t L = wp->walk_length;
    if (stats) {
        stats->block_parts = stats->block_slices = stats->block_stripes = 0;
        stats->block_group_parts = 0;
        stats->block_round_walks = 0;
    }

    // SkipGram in the default update mode, unless the graph is tiny: the block path (contextual
    // rows in XCD-exclusive cells)
    const uint32_t explicit_mode = GN2V_TRAIN_DETERMINISTIC | GN2V_TRAIN_ATOMIC |
                                   GN2V_TRAIN_WRITE_BACK | GN2V_TRAIN_WRITE_THROUGH |
                                   GN2V_TRAIN_WALK_ORDERED;
    if (!cbow && ((tp->flags & GN2V_TRAIN_BLOCK_PATH) ||
                  (!(tp->flags & explicit_mode) &&
                   g->view.n_nodes >= GN2V_BLOCK_PATH_MIN_NODES))) {
        const int rc = gn2v_train_blocks(g, wp, tp, seed, max_walks_per_epoch, 0, 0, d_central,
                                         d_contextual, stats, stream);
        // 2 = device memory ran out before anything was trained (alias tables, pair buffers): the
        // automatic choice falls back to the walk-ordered schedule, which needs the walks only
        if (rc != 2 || (tp->flags & GN2V_TRAIN_BLOCK_PATH)) return rc ? 1 : 0;
        if (stats) stats->block_parts = stats->block_slices = stats->block_stripes = 0;
    }

    if (gn2v_init_table(d_central, g->view.n_nodes, tp->d, tp->ld, seed, 0, tp->init_scale, s) ||
        gn2v_init_table(d_contextual, g->view.n_nodes, tp->d, tp->ld, seed, 1, tp->init_scale, s))
        return 1;

    uint64_t walks_per_epoch = g->view.n_sources * (uint64_t)wp->iterations;
    if (max_walks_per_epoch && max_walks_per_epoch < walks_per_epoch)
        walks_per_epoch = max_walks_per_epoch;
    // The walk sampler is latency bound (one walker per lane, dependent loads), so it wants many
    // more walkers in flight than one training launch consumes: walks are generated 2^19 at a
    // time (256 MiB at walk_length 128) and trained in launches of 2^16.
    const uint64_t walk_batch = std::min<uint64_t>(walks_per_epoch, (uint64_t)1 << 19);
    const uint64_t train_batch = (uint64_t)1 << 16;
    uint32_t *d_walks = nullptr;
    HIP_TRY(hipMalloc((void **)&d_walks, walk_batch * L * sizeof(uint32_t)));
    float lr = tp->lr;
    int rc = 0;
    for (uint32_t e = 0; e < tp->epochs && !rc; ++e) {
        for (uint64_t first = 0; first < walks_per_epoch && !rc; first += walk_batch) {
            const uint64_t nw = std::min(walk_batch, walks_per_epoch - first);
            rc = launch_walks(g, wp, seed, e, first, nw, d_walks, s);
            for (uint64_t off = 0; off < nw && !rc; off += train_batch) {
                const uint64_t n = std::min(train_batch, nw - off);
                gn2v_step_io io{};
                io.d_walks = d_walks + off * L;
                io.d_central = d_central;
                io.d_contextual = d_contextual;
                rc = launch_train(g, cbow, tp, &io, n, L, seed, e, first + off, lr, s);
            }
            if (!rc && g->train_events.size() > 2048) {  // bound the event pool on long fits
                if (hipStreamSynchronize(s) != hipSuccess) rc = 1;
                std::lock_guard<std::mutex> lock(g->mu);
                if (!rc && fold_events(g)) rc = 1;
            }
        }
        lr *= tp->lr_decay;
    }
    hipError_t se = hipStreamSynchronize(s);
    (void)hipFree(d_walks);
    if (rc) return rc;
    if (se != hipSuccess) return fail(std::string("training failed: ") + hipGetErrorString(se));
    if (stats) return gn2v_stats_read(g, stats, stream);
    return 0;
}

}  // extern "C"

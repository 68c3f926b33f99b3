// libgn2v.so -- entry points of the block-partitioned (multi-GPU) SkipGram trainer
// (include/gn2v.h "Block-partitioned training").  The reference has no counterpart: its one call,
// self._model.fit_transform(graph) (embedders/ensmallen_embedders/node2vec.py:99), runs in one
// process; this is how the same call is spread over the GPUs of a node.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "block_kernels.h"
#include "resident_kernels.h"
#include "handle.h"

using namespace gn2v_host;

namespace {

constexpr uint32_t kMaxSlices = GN2V_BLOCK_MAX_SLICES;

uint32_t bits_for(uint64_t n) {  // smallest b with 2^b >= n
    uint32_t b = 0;
    while ((1ULL << b) < n) ++b;
    return b;
}

int check_plan(const gn2v_graph *g, const gn2v_block_plan *p) {
    if (!g) return fail("graph handle is NULL");
    if (!p) return fail("block plan is NULL");
    if (p->world < 1 || p->rank >= p->world) return fail("need rank < world");
    // (ranks that exchange parts need parts % world == 0: the host-side trainer checks that; the
    // centre stripes of one GPU -- gn2v_block_io.central_ld -- do not)
    if (p->parts < 1) return fail("parts must be positive");
    if (p->slices < 1 || p->slices > kMaxSlices)
        return fail("slices must be in [1, " + std::to_string(kMaxSlices) + "]");
    if ((uint64_t)p->parts * p->slices > gn2v::kMaxCells) return fail("too many cells (parts x slices)");
    if (p->walk_length < 2 || p->window < 1) return fail("need walk_length >= 2, window_size >= 1");
    if (p->min_dist > p->window) return fail("min_dist must not exceed window_size");
    if (p->record > gn2v::kMaxRecord) return fail("record must be at most 32 pairs");
    if (p->hot_rows > GN2V_BLOCK_HOT_MAX)
        return fail("hot_rows must be at most " + std::to_string(GN2V_BLOCK_HOT_MAX));
    if (p->hot_flush > 1024 || (p->hot_flush & (p->hot_flush - 1)))
        return fail("hot_flush must be a power of two, at most 1024 (0 = 16)");
    return 0;
}

gn2v::BlockPlan device_plan(const gn2v_graph *g, const gn2v_block_plan *p) {
    gn2v::BlockPlan d{};
    d.world = p->world;
    d.rank = p->rank;
    d.parts = p->parts;
    d.slices = p->slices;
    d.L = p->walk_length;
    d.window = p->window;
    d.min_dist = p->min_dist ? p->min_dist : 1;
    d.record = p->record ? p->record : 32;
    const uint64_t rows = (g->view.n_nodes + p->world - 1) / p->world;
    d.row_bits = bits_for(rows);
    // rows of the largest cell: part 0, slice 0
    const uint64_t part_rows = gn2v::stripe_count(g->view.n_nodes, 0, p->parts);
    d.ctx_bits = bits_for(gn2v::stripe_count(part_rows, 0, p->slices)) + 1;  // + the hot flag
    d.flags = p->flags & gn2v::kFlagDownsample;
    d.hot_rows = p->hot_rows;
    d.dparts = gn2v::FastDiv::of(d.parts);
    d.dslices = gn2v::FastDiv::of(d.slices);
    d.dworld = gn2v::FastDiv::of(d.world);
    return d;
}

// LDS words a wavefront of sgns_block_kernel stages a record in (block_kernels.h)
size_t block_lds_words_per_wave(uint32_t ld, uint32_t record, uint32_t k) {
    return ((size_t)ld + 3 * record + 2 * (size_t)record * (k + 1) + 2 + 3) & ~(size_t)3;
}

// the extraction stages, per wave, the walk and three words per position, plus one counter per cell
// walks of at most 128 nodes and windows of at most 31 take block_extract_fast_kernel
bool extract_fast(uint32_t walk_length, uint32_t window) {
    return walk_length <= 128 && window <= 31;
}

size_t extract_lds_bytes(uint32_t walk_length, uint32_t cells) {
    const size_t arrays = walk_length <= 128 ? 5 : 4;  // the fast kernel stages one array more
    return ((size_t)(gn2v::kPrepBlock / 64) * arrays * walk_length + cells) * 4;
}

size_t env_size(const char *name, size_t fallback) {
    const char *v = getenv(name);
    return v && *v ? (size_t)strtoull(v, nullptr, 10) : fallback;
}

// The resident kernel's second form (resident_kernels.h; GN2V_RESIDENT_V2=0: round 4's, for speed
// A/Bs of the parallel form only)
bool resident_v2() {
    static const size_t v = env_size("GN2V_RESIDENT_V2", 1);
    return v != 0;
}

// Waves of a resident workgroup: sixteen for rows up to 128 floats, eight for wider rows (the
// second form only: resident_kernels.h sgns_resident_v2_kernel WAVES)
uint32_t resident_waves(uint32_t ld) { return resident_v2() && ld > 128 ? 8 : 16; }

// Rows of a cell that one workgroup holds in LDS next to its waves' staging; 0: rows too wide for
// the kernel (strides up to 512 floats; round 4's form: 256) or nothing left beside the staging
uint32_t resident_rows(uint32_t ld, uint32_t record, uint32_t k) {
    if (ld == 0 || ld > (resident_v2() ? 512u : 256u)) return 0;
    const size_t lds = 160 * 1024;
    if (resident_v2()) {
        // per wave: four transposition rows + the record's centres and 16-bit samples; shared:
        // the dummy row, then per cell row the row, its alias entry and its node id
        const size_t staging = (size_t)gn2v::res_words_per_wave(ld, record, k) * 4 *
                                   resident_waves(ld) + 64 + (size_t)gn2v::res_lds_stride(ld) * 4;
        // (a staged sample names its row in 12 bits, the dummy row included)
        static const size_t cap = env_size("GN2V_RESIDENT_FIT_ROWS", 4095);  // A/B: pin the cells
        return staging >= lds
                   ? 0
                   : (uint32_t)std::min<size_t>((lds - staging) / gn2v::res_bytes_per_row(ld), cap);
    }
    const size_t staging = block_lds_words_per_wave(ld, record, k) * 4 * 16 + 64;
    // a row and its node id
    return staging >= lds
               ? 0
               : (uint32_t)std::min<size_t>((lds - staging) / ((size_t)ld * 4 + 4), 4096);
}

// The record length the resident kernel runs a cell of `rows` rows with: the plan's, or the
// half or the quarter of it when the sixteen waves' staging of longer records (1 + k samples a
// pair) would not leave the cell its room (k = 50 at d = 128: records of 8).  0: none fits.
// The pairs, their order and their negatives do not depend on it (a record is only how many
// pairs a wave takes from the cursor at a time).
uint32_t resident_record(uint32_t ld, uint32_t record, uint32_t k, uint64_t rows) {
    for (uint32_t r = record; r >= 8 && r * 4 >= record; r >>= 1)
        if (rows <= resident_rows(ld, r, k)) return r;
    return 0;
}

// rows a cell of the automatic plan gets: what records of 32 pairs leave, or -- when that is
// fewer than 64 rows -- what records of 16, then 8 leave (gn2v_block_step picks the record to
// match, resident_record)
uint32_t resident_fit(uint32_t ld, uint32_t k) {
    uint32_t fit = 0;
    for (uint32_t r = 32; r >= 8; r >>= 1) {
        fit = resident_rows(ld, r, k);
        if (fit >= 64) break;
    }
    return fit;
}

uint32_t cell_bits(const gn2v::BlockPlan &d) { return bits_for((uint64_t)d.parts * d.slices); }

int check_key_width(const gn2v::BlockPlan &d) {
    if (d.row_bits > 32) return fail("centre rows need more than 32 bits");
    if (d.ctx_bits > 31) return fail("a cell must have fewer than 2^30 rows");
    if (cell_bits(d) + d.row_bits + d.ctx_bits > 64)
        return fail("cell, centre row and context row do not fit one 64-bit pair word: use more "
                    "ranks or fewer parts");
    return 0;
}

int check_group(const gn2v_block_plan *p, uint32_t part_lo, uint32_t *part_n) {
    if (*part_n == 0 && part_lo == 0) *part_n = p->parts;  // 0, 0 = every part
    if (part_lo >= p->parts || *part_n < 1 || *part_n > p->parts)
        return fail("group of parts out of range");
    if ((uint64_t)*part_n * p->slices > GN2V_BLOCK_MAX_WIDE_GROUP_CELLS)
        return fail("a group of parts may hold at most " +
                    std::to_string(GN2V_BLOCK_MAX_WIDE_GROUP_CELLS) +
                    " cells (parts of the group x slices): extract fewer parts at a time");
    return 0;
}

// a group with more cells than the counting pass has LDS counters for (include/gn2v.h)
bool wide_group(const gn2v::BlockPlan &d, uint32_t part_n) {
    const uint64_t cells = (uint64_t)part_n * d.slices;
    return cells > gn2v::kMaxGroupCells || extract_lds_bytes(d.L, (uint32_t)cells) > 64 * 1024;
}

// Stable radix sort of the pair words on the bits [begin_bit, end_bit) between two buffers of the
// same size (rocPRIM's double-buffer form: no third copy inside the temporary storage); the
// result is left in `out`.
int sort_words(void *temp, size_t temp_bytes, unsigned long long *in, unsigned long long *out,
               uint64_t n, uint32_t begin_bit, uint32_t end_bit, hipStream_t s) {
    rocprim::double_buffer<unsigned long long> kb(in, out);
    size_t need = 0;
    HIP_TRY(rocprim::radix_sort_keys(nullptr, need, kb, n, begin_bit, end_bit, s));
    if (need > temp_bytes) return fail("temporary storage too small for the radix sort");
    HIP_TRY(rocprim::radix_sort_keys(temp, need, kb, n, begin_bit, end_bit, s));
    if (kb.current() != out)
        HIP_TRY(hipMemcpyAsync(out, kb.current(), n * sizeof(unsigned long long),
                               hipMemcpyDeviceToDevice, s));
    return 0;
}

size_t sort_temp_bytes(uint64_t n) {
    size_t need = 0;
    rocprim::double_buffer<unsigned long long> kb(nullptr, nullptr);
    if (rocprim::radix_sort_keys(nullptr, need, kb, n ? n : 1, 0, 64, (hipStream_t)0) !=
        hipSuccess)
        return 0;
    return (need + 255) & ~(size_t)255;
}

constexpr size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// every slice of a part is served by exactly one XCD (block_kernels.h, sgns_block_kernel)
bool slices_are_xcd_exclusive(const gn2v_graph *g, uint32_t slices) {
    return slices > 1 && g->n_xcds > 0 && slices % (uint32_t)g->n_xcds == 0;
}

}  // namespace

namespace {
// Device buffers of one gn2v_train_blocks call, given back on every exit path -- to the graph
// handle, which keeps them for its next fit (up to a third of the device's memory;
// gn2v_graph_release_buffers or gn2v_graph_destroy frees them).  A second fit on the same handle found its 65 GB of round buffers only after the driver
// had cleared them again: 1.6-2.0 s of a 15 s call (profiles/r05_logs/r5_var.log).
struct Buffers {
    gn2v_graph *g;
    std::vector<std::pair<void *, size_t>> ptrs;
    // set once the call has succeeded and its streams are drained: until then an exit is an error
    // exit -- the side stream of the round driver (or a launch on the caller's) may still write
    // the buffers, so the device is drained and they go back to the driver, not to the handle,
    // where the next fit would reuse them at once
    bool done = false;
    explicit Buffers(gn2v_graph *graph) : g(graph) {}
    ~Buffers() {
        if (!done && !ptrs.empty()) (void)hipDeviceSynchronize();
        while (!ptrs.empty()) {
            if (done)
                release_last();
            else
                free_last();
        }
    }
    template <class T>
    int alloc(T **out, size_t bytes) {
        *out = nullptr;
        bytes = bytes ? bytes : 4;
        void *p = nullptr;
        size_t size = bytes;
        {
            // the smallest kept block that holds it without wasting more than a quarter
            std::lock_guard<std::mutex> lock(g->kept_mu);
            size_t best = g->kept_buffers.size();
            for (size_t i = 0; i < g->kept_buffers.size(); ++i) {
                const size_t have = g->kept_buffers[i].second;
                if (have >= bytes && have - bytes <= bytes / 4 &&
                    (best == g->kept_buffers.size() || have < g->kept_buffers[best].second))
                    best = i;
            }
            if (best != g->kept_buffers.size()) {
                p = g->kept_buffers[best].first;
                size = g->kept_buffers[best].second;
                g->kept_bytes -= size;
                g->kept_buffers.erase(g->kept_buffers.begin() + (long)best);
            }
        }
        if (!p && hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            // what the handle still keeps may be exactly what is missing
            gn2v_host::release_kept_buffers(g);
            if (hipMalloc(&p, bytes) != hipSuccess) p = nullptr;
            if (!p) {
                (void)hipGetLastError();
                return fail("out of device memory in gn2v_train_blocks (" +
                            std::to_string(bytes >> 20) + " MiB)");
            }
        }
        ptrs.push_back({p, size});
        *out = (T *)p;
        return 0;
    }
    // the most recent allocation goes back (to the handle, or to the driver)
    void release_last() {
        const auto b = ptrs.back();
        ptrs.pop_back();
        size_t total = 0, free_b = 0;
        if (b.second >= ((size_t)1 << 20) && hipMemGetInfo(&free_b, &total) == hipSuccess) {
            std::lock_guard<std::mutex> lock(g->kept_mu);
            if (g->kept_bytes + b.second <= total / 3) {
                g->kept_buffers.push_back(b);
                g->kept_bytes += b.second;
                return;
            }
        }
        (void)hipFree(b.first);
    }
    void free_last() {  // to the driver, whatever its size (room for another allocation)
        (void)hipFree(ptrs.back().first);
        ptrs.pop_back();
    }
};
}  // namespace

namespace {
__global__ void max_u32_kernel(const uint32_t *__restrict__ v, uint64_t n,
                               unsigned int *__restrict__ out) {
    unsigned int m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x)
        m = max(m, v[i]);
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned int)__shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// The in-degrees of the graph (how often a node is the endpoint of a uniform random directed
// edge: the weights of the degree-proportional negatives) and the largest of them, computed once
// per handle and kept (4 B per node): a placement rebuilds the alias tables every round, and
// 2 x 10^9 atomic increments per rebuild were 2.3 % of the GPU time at 100 M nodes.  Returns 1
// (no error recorded) when the handle cannot keep them: the caller counts into its own storage.
int in_degrees(gn2v_graph *g, hipStream_t s, uint32_t **out) {
    if (!g->indeg) {
        if (g->indeg_failed) return 1;
        const uint64_t n = g->view.n_nodes, E = g->view.n_edges;
        uint32_t *indeg = nullptr;
        if (hipMalloc((void **)&indeg, (n + 1) * sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();
            g->indeg_failed = true;
            return 1;
        }
        hipError_t e = hipMemsetAsync(indeg, 0, (n + 1) * sizeof(uint32_t), s);
        if (e == hipSuccess && E) {
            const unsigned blocks = (unsigned)std::min<uint64_t>((E + 255) / 256, 256 * 16);
            hipLaunchKernelGGL(gn2v::indegree_kernel, dim3(blocks), dim3(256), 0, s,
                               g->view.col_idx, E, indeg);
            const unsigned nb = (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 16);
            hipLaunchKernelGGL(max_u32_kernel, dim3(nb), dim3(256), 0, s, indeg, n, indeg + n);
            e = hipGetLastError();
        }
        unsigned int m = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&m, indeg + n, 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);  // complete before any other stream uses it
        if (e != hipSuccess) {
            (void)hipFree(indeg);
            g->indeg_failed = true;
            return 1;
        }
        g->indeg = indeg;
        g->max_in_degree = m;
        g->max_in_degree_known = true;
    }
    *out = g->indeg;
    return 0;
}
}  // namespace

extern "C" {

int gn2v_block_plan_check(gn2v_graph *g, gn2v_block_plan *plan) {
    if (check_plan(g, plan)) return 1;
    const gn2v::BlockPlan d = device_plan(g, plan);
    if (check_key_width(d)) return 1;
    plan->row_bits = d.row_bits;
    plan->ctx_bits = d.ctx_bits;
    plan->key_bits = cell_bits(d) + d.row_bits + d.ctx_bits;
    plan->record = d.record;
    plan->min_dist = d.min_dist;
    return 0;
}

int gn2v_init_table_rows(float *d_table, uint64_t n_rows, uint32_t d, uint32_t ld, uint64_t seed,
                         uint32_t table_id, float scale, uint64_t first_row, uint64_t row_stride,
                         void *stream) {
    if (d == 0 || ld < d) return fail("need 0 < d <= ld");
    const uint64_t n = n_rows * ld;
    if (n == 0) return 0;
    if (!d_table) return fail("NULL table pointer");
    DeviceGuard guard(DeviceGuard::of_pointer(d_table));
    const unsigned blocks = (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(gn2v::init_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       d_table, n_rows, d, ld, gn2v::mix64(seed ^ (gn2v::kTagInit + table_id)), scale,
                       first_row, row_stride ? row_stride : 1);
    HIP_TRY(hipGetLastError());
    return 0;
}

int gn2v_block_alias_temp_bytes(uint64_t n_nodes, uint64_t *bytes) {
    if (!bytes) return fail("bytes is NULL");
    *bytes = 2 * align256(n_nodes * 4) + align256(n_nodes * 8);
    return 0;
}

int gn2v_block_alias(gn2v_graph *g, const gn2v_block_plan *plan, uint64_t *d_alias,
                     uint64_t *d_cell_rows, uint32_t *d_hub_bits, uint32_t *d_hot_list,
                     uint8_t *d_hot_slot, const uint32_t *d_inv, void *d_temp,
                     uint64_t temp_bytes, void *stream) {
    if (check_plan(g, plan)) return 1;
    if (!d_alias || !d_cell_rows || !d_temp) return fail("NULL pointer");
    if (plan->hot_rows && (!d_hub_bits || !d_hot_list || !d_hot_slot))
        return fail("a plan with hot rows needs d_hub_bits, d_hot_list and d_hot_slot");
    if ((g->view.n_nodes + plan->parts - 1) / plan->parts >= (1ULL << 31))
        return fail("a context part must have fewer than 2^31 rows");
    const uint64_t n = g->view.n_nodes;
    uint64_t need = 0;
    gn2v_block_alias_temp_bytes(n, &need);
    if (temp_bytes < need) return fail("temporary storage too small for the alias tables");
    if (g->view.n_edges >= (1ULL << 47)) return fail("too many edges for the alias tables");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    hipStream_t s = (hipStream_t)stream;
    char *t = (char *)d_temp;
    // the in-degrees: computed once per handle when it can keep them (a placement rebuilds the
    // tables every round), else into the temporary storage
    uint32_t *indeg = nullptr;
    if (in_degrees(g, s, &indeg)) {
        indeg = (uint32_t *)t;
        HIP_TRY(hipMemsetAsync(indeg, 0, n * sizeof(uint32_t), s));
        const uint64_t E = g->view.n_edges;
        const unsigned blocks = (unsigned)std::min<uint64_t>((E + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(gn2v::indegree_kernel, dim3(blocks), dim3(256), 0, s, g->view.col_idx,
                           E, indeg);
        HIP_TRY(hipGetLastError());
    }
    uint32_t *stack = (uint32_t *)(t + align256(n * 4));
    unsigned long long *weight = (unsigned long long *)(t + 2 * align256(n * 4));
    if (d_hub_bits) HIP_TRY(hipMemsetAsync(d_hub_bits, 0, ((n + 31) / 32) * sizeof(uint32_t), s));
    if (d_hot_slot) HIP_TRY(hipMemsetAsync(d_hot_slot, 0xFF, n, s));
    hipLaunchKernelGGL(gn2v::cell_rows_kernel,
                       dim3((unsigned)(((uint64_t)plan->parts * plan->slices + 256) / 256)),
                       dim3(256), 0, s, n, plan->parts, plan->slices,
                       (unsigned long long *)d_cell_rows);
    HIP_TRY(hipGetLastError());
    const uint32_t cells = plan->parts * plan->slices;
    hipLaunchKernelGGL(gn2v::alias_kernel, dim3((cells + 63) / 64), dim3(64), 0, s, indeg, n,
                       plan->parts, plan->slices, (const unsigned long long *)d_cell_rows,
                       (unsigned long long *)d_alias, weight, stack, d_hub_bits, plan->hot_rows,
                       d_hot_list, d_hot_slot, d_inv);
    HIP_TRY(hipGetLastError());
    return 0;
}

int gn2v_block_placement_temp_bytes(uint64_t n_nodes, uint64_t *bytes) {
    if (!bytes) return fail("bytes is NULL");
    size_t sort = 0;
    rocprim::double_buffer<unsigned long long> kb(nullptr, nullptr);
    rocprim::double_buffer<uint32_t> vb(nullptr, nullptr);
    if (rocprim::radix_sort_pairs(nullptr, sort, kb, vb, n_nodes ? n_nodes : 1, 0, 64,
                                  (hipStream_t)0) != hipSuccess)
        return fail("rocprim::radix_sort_pairs (size query)");
    *bytes = 2 * align256(n_nodes * 8) + 2 * align256(n_nodes * 4) + align256(sort);
    return 0;
}

int gn2v_block_placement(gn2v_graph *g, uint32_t classes, uint64_t seed, uint64_t round_id,
                         uint32_t *d_place, uint32_t *d_inv, void *d_temp, uint64_t temp_bytes,
                         void *stream) {
    if (!g) return fail("graph handle is NULL");
    if (!d_place || !d_inv || !d_temp) return fail("NULL pointer");
    const uint64_t n = g->view.n_nodes;
    if (classes < 1 || classes > n || classes >= (1u << 23))
        return fail("classes must be in [1, min(n_nodes, 2^23))");
    uint64_t need = 0;
    if (gn2v_block_placement_temp_bytes(n, &need)) return 1;
    if (temp_bytes < need) return fail("temporary storage too small for the placement");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    hipStream_t s = (hipStream_t)stream;
    char *t = (char *)d_temp;
    unsigned long long *k0 = (unsigned long long *)t, *k1 = (unsigned long long *)(t + align256(n * 8));
    uint32_t *v0 = (uint32_t *)(t + 2 * align256(n * 8));
    uint32_t *v1 = (uint32_t *)(t + 2 * align256(n * 8) + align256(n * 4));
    void *sort_tmp = t + 2 * align256(n * 8) + 2 * align256(n * 4);
    size_t sort_bytes = temp_bytes - (2 * align256(n * 8) + 2 * align256(n * 4));
    const uint64_t pkey = gn2v::draw(gn2v::mix64(seed ^ gn2v::kTagPlace), round_id);
    const unsigned blocks = (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(gn2v::place_keys_kernel, dim3(blocks), dim3(256), 0, s, n, classes, pkey, k0,
                       v0);
    HIP_TRY(hipGetLastError());
    rocprim::double_buffer<unsigned long long> kb(k0, k1);
    rocprim::double_buffer<uint32_t> vb(v0, v1);
    HIP_TRY(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, kb, vb, n, 0, 40 + bits_for(classes), s));
    hipLaunchKernelGGL(gn2v::place_scatter_kernel, dim3(blocks), dim3(256), 0, s, n, classes,
                       vb.current(), d_place, d_inv);
    HIP_TRY(hipGetLastError());
    return 0;
}

int gn2v_block_place_walks(const uint32_t *d_place, const uint32_t *d_walks, uint64_t n_entries,
                           uint32_t *d_out, void *stream) {
    if (n_entries == 0) return 0;
    if (!d_place || !d_walks || !d_out) return fail("NULL pointer");
    DeviceGuard guard(DeviceGuard::of_pointer(d_out));
    const unsigned blocks = (unsigned)std::min<uint64_t>((n_entries + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(gn2v::place_walks_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       d_place, d_walks, n_entries, d_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int launch_extract(gn2v_graph *g, const gn2v::BlockPlan &d, bool write,
                          const uint32_t *d_walks, const uint32_t *d_placed, uint64_t n_walks,
                          uint64_t seed, uint64_t epoch,
                          uint64_t first_walk, uint32_t part_lo, uint32_t part_n, uint64_t *d_work,
                          const uint32_t *d_hub_bits, uint64_t *pairs, hipStream_t s) {
    gn2v::ExtractArgs a{};
    a.g = g->view;
    a.p = d;
    a.walks = d_walks;
    a.placed = d_placed;
    a.n_walks = n_walks;
    a.ekey = gn2v::epoch_key(seed, epoch);
    a.first_walk = first_walk;
    const bool wide = wide_group(d, part_n);
    a.wave_counts = (unsigned long long *)d_work;
    a.cell_counts = wide ? nullptr : (unsigned long long *)d_work + gn2v::kPrepWaves;
    a.hub_bits = d_hub_bits;
    a.pairs = (unsigned long long *)pairs;
    a.part_lo = part_lo;
    a.part_n = part_n;
    // the counting pass keeps one LDS counter per cell of the group; the writing pass needs the
    // walk staging only (and runs twice as many waves per CU without the counters)
    const size_t lds = extract_lds_bytes(d.L, write || wide ? 0 : part_n * d.slices);
    if (lds > 64 * 1024) return fail("walk_length too large for the extraction's LDS plan");
    const dim3 grid(gn2v::kPrepWaves / (gn2v::kPrepBlock / 64)), block(gn2v::kPrepBlock);
    if (extract_fast(d.L, d.window)) {
        if (write)
            hipLaunchKernelGGL((gn2v::block_extract_fast_kernel<true>), grid, block, lds, s, a);
        else
            hipLaunchKernelGGL((gn2v::block_extract_fast_kernel<false>), grid, block, lds, s, a);
    } else if (write)
        hipLaunchKernelGGL((gn2v::block_extract_kernel<true>), grid, block, lds, s, a);
    else
        hipLaunchKernelGGL((gn2v::block_extract_kernel<false>), grid, block, lds, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int gn2v_block_count(gn2v_graph *g, const gn2v_block_plan *plan, const uint32_t *d_walks,
                     const uint32_t *d_placed_walks, uint64_t n_walks, uint64_t seed,
                     uint64_t epoch, uint64_t first_walk,
                     uint32_t part_lo, uint32_t part_n, uint64_t *d_work,
                     uint64_t *d_cell_offsets, void *stream) {
    if (check_plan(g, plan)) return 1;
    const gn2v::BlockPlan d = device_plan(g, plan);
    if (check_key_width(d) || check_group(plan, part_lo, &part_n)) return 1;
    if (!d_work || !d_cell_offsets || (n_walks && !d_walks)) return fail("NULL pointer");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    hipStream_t s = (hipStream_t)stream;
    // the waves' counts and the counters of the plan's cells (the scan reads no more)
    HIP_TRY(hipMemsetAsync(d_work, 0,
                           ((size_t)gn2v::kPrepWaves + (size_t)d.parts * d.slices) * sizeof(uint64_t),
                           s));
    if (n_walks &&
        launch_extract(g, d, false, d_walks, d_placed_walks, n_walks, seed, epoch, first_walk,
                       part_lo, part_n, d_work, nullptr, nullptr, s))
        return 1;
    // (a wide group: no counters -- only d_cell_offsets[cells], the number of pairs, is written)
    hipLaunchKernelGGL(gn2v::block_scan_kernel, dim3(1), dim3(1024), 0, s,
                       (unsigned long long *)d_work,
                       wide_group(d, part_n)
                           ? (const unsigned long long *)nullptr
                           : (const unsigned long long *)d_work + gn2v::kPrepWaves,
                       d.parts * d.slices, (unsigned long long *)d_cell_offsets);
    HIP_TRY(hipGetLastError());
    return 0;
}

int gn2v_block_cell_offsets(gn2v_graph *g, const gn2v_block_plan *plan, uint32_t part_n,
                            const uint64_t *d_pairs, uint64_t n_pairs, uint64_t *d_cell_offsets,
                            void *stream) {
    if (check_plan(g, plan)) return 1;
    const gn2v::BlockPlan d = device_plan(g, plan);
    uint32_t lo = 0;
    if (check_key_width(d) || check_group(plan, lo, &part_n)) return 1;
    if (!wide_group(d, part_n)) return 0;  // counted: gn2v_block_count wrote them
    if (!d_cell_offsets || (n_pairs && !d_pairs)) return fail("NULL pointer");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    const uint32_t cells = d.parts * d.slices;
    hipLaunchKernelGGL(gn2v::cell_offsets_from_sorted_kernel, dim3((cells + 1 + 255) / 256),
                       dim3(256), 0, (hipStream_t)stream, (const unsigned long long *)d_pairs,
                       (unsigned long long)n_pairs, d.row_bits + d.ctx_bits, cells,
                       (unsigned long long *)d_cell_offsets);
    HIP_TRY(hipGetLastError());
    return 0;
}

int gn2v_block_extract_temp_bytes(uint64_t n_pairs, uint64_t *bytes) {
    if (!bytes) return fail("bytes is NULL");
    *bytes = align256(n_pairs * 8) + sort_temp_bytes(n_pairs);
    return 0;
}

int gn2v_block_extract(gn2v_graph *g, const gn2v_block_plan *plan, const uint32_t *d_walks,
                       const uint32_t *d_placed_walks, uint64_t n_walks, uint64_t seed,
                       uint64_t epoch, uint64_t first_walk,
                       uint32_t part_lo, uint32_t part_n, const uint64_t *d_work,
                       const uint32_t *d_hub_bits, uint64_t n_pairs, uint64_t *d_pairs,
                       void *d_temp, uint64_t temp_bytes, void *stream) {
    if (check_plan(g, plan)) return 1;
    const gn2v::BlockPlan d = device_plan(g, plan);
    if (check_key_width(d) || check_group(plan, part_lo, &part_n)) return 1;
    if (n_pairs == 0) return 0;
    if (!d_work || !d_walks || !d_pairs || !d_temp) return fail("NULL pointer");
    uint64_t need = 0;
    gn2v_block_extract_temp_bytes(n_pairs, &need);
    if (temp_bytes < need) return fail("temporary storage too small for the extraction");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    hipStream_t s = (hipStream_t)stream;
    char *t = (char *)d_temp;
    unsigned long long *unsorted = (unsigned long long *)t;
    const size_t head = align256(n_pairs * 8);
    if (launch_extract(g, d, true, d_walks, d_placed_walks, n_walks, seed, epoch, first_walk,
                       part_lo, part_n, const_cast<uint64_t *>(d_work), d_hub_bits,
                       (uint64_t *)unsorted, s))
        return 1;
    const uint32_t end_bit = std::max(d.ctx_bits + 1, d.ctx_bits + d.row_bits + cell_bits(d));
    // resident plans: (cell, the centre's highest bits) -- GN2V_RESIDENT_CENTRE_SORT_BITS
    uint32_t begin_bit = d.ctx_bits;
    // (round 6, same box, bench graph: 4 bits 2.09e9 pairs/s, 8 bits 2.20, 12 / 16 bits 2.16-2.17;
    // no key on the centre at all -- the counted scatter -- 2.15: profiles/r06_logs/
    // r6_sort_centre_bits_ab.log, r6_scatter_vs_sort_ab.log)
    if (d.slices > gn2v_host::kCursorSlices && d.row_bits > GN2V_RESIDENT_CENTRE_SORT_BITS)
        begin_bit += d.row_bits - GN2V_RESIDENT_CENTRE_SORT_BITS;
    return sort_words(t + head, temp_bytes - head, unsorted, (unsigned long long *)d_pairs,
                      n_pairs, begin_bit, end_bit, s);
}

extern "C++" {
// dynamic LDS beyond 64 KB has to be allowed per kernel, once
template <class K>
static void allow_lds(K kernel, size_t lds) {
    // per kernel instantiation (K) and device: the largest size allowed so far
    static std::mutex mu;
    static std::vector<std::pair<std::pair<const void *, int>, size_t>> allowed;
    if (lds <= 64 * 1024) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const std::pair<const void *, int> key(reinterpret_cast<const void *>(kernel), dev);
    std::lock_guard<std::mutex> lock(mu);
    for (auto &e : allowed)
        if (e.first == key) {
            if (e.second >= lds) return;
            e.second = lds;
            (void)hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            return;
        }
    allowed.push_back({key, lds});
    (void)hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <int CH>
static void launch_block_ch(int wmx, int wmc, bool det, dim3 grid, dim3 block, size_t lds,
                            hipStream_t s, const gn2v::BlockArgs &a) {
    // the row stride is a compile-time constant where it fills its instantiation
    const bool full = a.ld == (uint32_t)CH * 64;
#define GN2V_BLOCK(WMX, WMC, DT)                                                                  \
    do {                                                                                          \
        if (full && !DT)                                                                          \
            hipLaunchKernelGGL((gn2v::sgns_block_kernel<CH, WMX, WMC, DT, true>), grid, block,    \
                               lds, s, a);                                                        \
        else                                                                                      \
            hipLaunchKernelGGL((gn2v::sgns_block_kernel<CH, WMX, WMC, DT, false>), grid, block,   \
                               lds, s, a);                                                        \
    } while (0)
// one workgroup of sixteen waves per CU (rows up to 128 floats, the store flavours)
#define GN2V_BLOCK_WIDE(WMX)                                                                      \
    do {                                                                                          \
        if (full) {                                                                               \
            auto kernel = gn2v::sgns_block_kernel<CH, WMX, gn2v::kAtomic, false, true, 1024>;     \
            allow_lds(kernel, lds);                                                               \
            hipLaunchKernelGGL(kernel, grid, block, lds, s, a);                                   \
        } else {                                                                                  \
            auto kernel = gn2v::sgns_block_kernel<CH, WMX, gn2v::kAtomic, false, false, 1024>;    \
            allow_lds(kernel, lds);                                                               \
            hipLaunchKernelGGL(kernel, grid, block, lds, s, a);                                   \
        }                                                                                         \
    } while (0)
    if constexpr (CH <= 2) {
        if (block.x == 1024) {
            if (wmx == gn2v::kWriteBack)
                GN2V_BLOCK_WIDE(gn2v::kWriteBack);
            else
                GN2V_BLOCK_WIDE(gn2v::kWriteThrough);
            return;
        }
    }
    if (det)
        GN2V_BLOCK(gn2v::kWriteBack, gn2v::kWriteBack, true);
    else if (wmx == gn2v::kAtomic)
        GN2V_BLOCK(gn2v::kAtomic, gn2v::kAtomic, false);
    else if (wmx == gn2v::kWriteBack && wmc == gn2v::kWriteBack)
        GN2V_BLOCK(gn2v::kWriteBack, gn2v::kWriteBack, false);
    else if (wmx == gn2v::kWriteBack)
        GN2V_BLOCK(gn2v::kWriteBack, gn2v::kAtomic, false);
    else
        GN2V_BLOCK(gn2v::kWriteThrough, gn2v::kAtomic, false);
#undef GN2V_BLOCK
#undef GN2V_BLOCK_WIDE
}

// Heaviest cell first (BlockArgs.order): the cells of a launch sorted by their pairs, on the
// launch's stream, in a ring of slots owned by the handle -- keys, values and the sort's storage
// of a slot are rewritten kLptRing launches later, long after the launch that read them (the
// launches of a handle are issued stream-ordered by its callers; 64 of them are minutes of
// training).  Launches of fewer than two cells per CU, or of more cells than an extraction group
// may hold, keep the index order.  GN2V_RESIDENT_LPT=0: never (A/B).
constexpr uint32_t kLptRing = 32, kLptCells = GN2V_BLOCK_MAX_WIDE_GROUP_CELLS;
static int lpt_order(gn2v_graph *g, uint32_t n, uint32_t first_cell,
                     const unsigned long long *d_cell_offsets, const uint32_t **order,
                     hipStream_t s) {
    const char *env = getenv("GN2V_RESIDENT_LPT");  // read per launch: tests switch it
    const bool on = !(env && *env == '0');
    *order = nullptr;
    if (!on || n > kLptCells || n < 2u * (uint32_t)g->n_cus) return 0;
    if (!g->lpt) {
        uint32_t *kb = nullptr, *vb = nullptr;
        size_t need = 0;
        if (rocprim::radix_sort_pairs(nullptr, need, kb, kb, vb, vb, kLptCells, 0, 32,
                                      (hipStream_t)0) != hipSuccess)
            return fail("rocprim::radix_sort_pairs (size query)");
        void *temp = nullptr;
        uint32_t *buf = nullptr;
        need = (need + 255) & ~(size_t)255;
        if (hipMalloc((void **)&buf, (size_t)kLptRing * 4 * kLptCells * 4) != hipSuccess ||
            hipMalloc(&temp, (size_t)kLptRing * (need ? need : 256)) != hipSuccess) {
            (void)hipGetLastError();
            if (buf) (void)hipFree(buf);
            return 0;  // no room: index order
        }
        g->lpt = buf;
        g->lpt_temp = temp;
        g->lpt_temp_bytes = need ? need : 256;
    }
    const uint32_t turn = g->lpt_slot++ % kLptRing;
    uint32_t *slot = g->lpt + (size_t)turn * 4 * kLptCells;
    void *temp = (char *)g->lpt_temp + (size_t)turn * g->lpt_temp_bytes;
    uint32_t *keys_in = slot, *keys_out = slot + kLptCells, *vals_in = slot + 2 * kLptCells,
             *vals_out = slot + 3 * kLptCells;
    hipLaunchKernelGGL(gn2v::cell_order_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, s,
                       d_cell_offsets, first_cell, n, keys_in, vals_in);
    HIP_TRY(hipGetLastError());
    size_t bytes = g->lpt_temp_bytes;
    HIP_TRY(rocprim::radix_sort_pairs(temp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 32,
                                      s));
    *order = vals_out;
    return 0;
}

template <int CH, int LDQ, bool DET>
static void launch_resident_v2_one(dim3 grid, size_t lds, hipStream_t s, const gn2v::BlockArgs &a) {
    constexpr int W = CH > 2 ? 8 : 16;  // resident_waves()
    auto kernel = gn2v::sgns_resident_v2_kernel<CH, LDQ, DET, W>;
    allow_lds(kernel, lds);
    hipLaunchKernelGGL(kernel, DET ? dim3(1) : grid, dim3(W * 64), lds, s, a);
}

// The strides the Python classes use (models.SkipGram.padded_size: multiples of 32 floats up to
// 128, of 64 beyond) get an instantiation with the stride as a constant; any other stride -- and
// the deterministic form, where one workgroup walks the cells in order -- reads it from the
// arguments.
template <int CH>
static void launch_resident_v2_ch(bool det, dim3 grid, size_t lds, hipStream_t s,
                                  const gn2v::BlockArgs &a) {
    if (det) return launch_resident_v2_one<CH, 0, true>(grid, lds, s, a);
    const uint32_t ldq = a.ld % 32 == 0 ? a.ld / 32 : 0;
#define GN2V_LDQ(Q) \
    if (ldq == Q) return launch_resident_v2_one<CH, Q, false>(grid, lds, s, a)
    if constexpr (CH == 1) {
        GN2V_LDQ(1);
        GN2V_LDQ(2);
    } else if constexpr (CH == 2) {
        GN2V_LDQ(3);
        GN2V_LDQ(4);
    } else if constexpr (CH == 4) {
        GN2V_LDQ(6);
        GN2V_LDQ(8);
    } else {
        GN2V_LDQ(10);
        GN2V_LDQ(12);
        GN2V_LDQ(14);
        GN2V_LDQ(16);
    }
#undef GN2V_LDQ
    launch_resident_v2_one<CH, 0, false>(grid, lds, s, a);
}

template <int CH>
static void launch_resident_ch(bool det, dim3 grid, size_t lds, hipStream_t s,
                               const gn2v::BlockArgs &a) {
    if (det) {  // one workgroup walks the cells in order (block_kernels.h)
        auto kernel = gn2v::sgns_resident_kernel<CH, false, true>;
        allow_lds(kernel, lds);
        hipLaunchKernelGGL(kernel, dim3(1), dim3(1024), lds, s, a);
    } else if (a.ld == (uint32_t)CH * 64) {
        auto kernel = gn2v::sgns_resident_kernel<CH, true>;
        allow_lds(kernel, lds);
        hipLaunchKernelGGL(kernel, grid, dim3(1024), lds, s, a);
    } else {
        auto kernel = gn2v::sgns_resident_kernel<CH, false>;
        allow_lds(kernel, lds);
        hipLaunchKernelGGL(kernel, grid, dim3(1024), lds, s, a);
    }
}
}  // extern "C++"

// part_n > 1: the parts io->part .. io->part + part_n - 1 in ONE launch (resident cells only;
// d_part_ptrs[p] = the rows of part p, on the device); *took_group says whether that happened
// (false: the plan is not resident here and nothing was launched)
static int block_step(gn2v_graph *g, const gn2v_train_params *tp, const gn2v_block_plan *plan,
                      const gn2v_block_io *io, uint64_t seed, uint64_t epoch, float lr,
                      void *stream, uint32_t part_n, float *const *d_part_ptrs, bool *took_group) {
    if (check_plan(g, plan)) return 1;
    const gn2v::BlockPlan d = device_plan(g, plan);
    if (check_key_width(d)) return 1;
    if (!tp || !io) return fail("NULL params / io");
    if (tp->d == 0) return fail("embedding size must be strictly positive");
    if (tp->ld < tp->d || (tp->ld & 3)) return fail("ld must be a multiple of 4 and >= d");
    if (tp->ld > 1024) return fail("embedding sizes above 1024 are not supported");
    if (!std::isfinite(lr) || !std::isfinite(tp->clip) || tp->clip <= 0.f)
        return fail("learning rate / clipping value must be finite, clipping value positive");
    if (io->part >= plan->parts || part_n < 1 || io->part + part_n > plan->parts)
        return fail("part out of range");
    if (!io->d_pairs || !io->d_cell_offsets || !io->d_central ||
        (!io->d_context && !io->d_context_table && part_n == 1))
        return fail("NULL pointer");
    if ((tp->flags & GN2V_TRAIN_SCALE_FREE) && (!io->d_alias || !io->d_cell_rows))
        return fail("degree-proportional negatives need the tables of gn2v_block_alias");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    hipStream_t s = (hipStream_t)stream;

    gn2v::BlockArgs a{};
    a.g = g->view;
    a.p = d;
    a.pairs = (const unsigned long long *)io->d_pairs;
    a.cell_offsets = (const unsigned long long *)io->d_cell_offsets;
    const bool scale_free = tp->flags & GN2V_TRAIN_SCALE_FREE;
    a.alias = scale_free ? (const unsigned long long *)io->d_alias : nullptr;
    a.cell_rows = (const unsigned long long *)io->d_cell_rows;
    if ((io->d_hot_list != nullptr) != (io->d_hot_slot != nullptr) ||
        (io->d_hot_list && !io->d_cell_rows))
        return fail("the hot rows need d_hot_list, d_hot_slot and d_cell_rows together");
    a.hot_list = io->d_hot_list;
    a.hot_slot = io->d_hot_slot;
    a.central = io->d_central;
    a.cld = io->central_ld ? io->central_ld : tp->ld;
    if (a.cld < tp->ld || (a.cld & 3)) return fail("central_ld must be a multiple of 4 and >= ld");
    a.context = io->d_context;
    a.xld = io->context_ld ? io->context_ld : tp->ld;
    if (a.xld < tp->ld || (a.xld & 3)) return fail("context_ld must be a multiple of 4 and >= ld");
    a.inv = io->d_inv;
    a.ctx_table = io->d_context_table;
    if (a.ctx_table && !a.inv) return fail("d_context_table is for plans under a placement (d_inv)");
    a.counters = g->counters;
    a.n_nodes = g->view.n_nodes;
    a.ekey = gn2v::epoch_key(seed, epoch);
    a.block_id = io->block_id;
    a.part = io->part;
    a.k = tp->k;
    a.ld = tp->ld;
    a.flags = tp->flags & 7u;
    a.lr = lr;
    a.clip = tp->clip;

    const bool det = tp->flags & GN2V_TRAIN_DETERMINISTIC;
    // contextual rows: exclusive to one XCD when the part has one slice per XCD -> plain
    // write-back stores, else write-through.  Central rows: the gradient of a whole record is added with hardware
    // f32 atomics (one row per ~record * (k + 1) sample rows: free, and the records of a hub centre
    // that many waves train at once lose no update).  Small graphs: atomics everywhere.
    // One XCD per slice -- the kernel maps XCD x to the slices x, x + n_xcds, ... -- holds only
    // when the slices are a multiple of the XCDs the workgroups are spread over (8 on an MI355X).
    // With 2 or 4 slices several XCDs (non-coherent L2s) read-modify-write the same rows: they
    // keep the write-through stores, as do unsliced parts (atomics on small graphs, where all
    // wavefronts meet on the same few rows: exclusive slices divide that crowd by the XCDs and
    // keep it inside one L2, which is why they need no atomics from GN2V_BLOCK_PATH_MIN_NODES up).
    const bool exclusive = slices_are_xcd_exclusive(g, d.slices);
    // Resident cells (block_kernels.h sgns_resident_kernel): a plan of MORE THAN 16 SLICES whose
    // every cell fits one workgroup's LDS -- what gn2v_block_auto_plan arranges for graphs up to
    // 115 M nodes at d = 128 (its XCD-cell plans have 1 or 8 slices; an explicit plan of a few
    // slices on a tiny graph keeps sgns_block_kernel although its cells would fit).  One
    // workgroup per cell, contextual rows read and updated in LDS: no other CU touches them, no
    // flavour of global store or atomic is involved.  Inside the workgroup the rows are plain
    // read-modify-writes of its 64 concurrent 16-lane groups, so updates that meet on a row
    // within ~100 cycles lose one (block_kernels.h; counted in tests/test_gpu_resident.py).
    // GN2V_TRAIN_DETERMINISTIC runs the same kernel's deterministic instantiation.
    constexpr size_t resident_env = 1;
    const uint64_t max_cell_rows =
        gn2v::stripe_count(gn2v::stripe_count(g->view.n_nodes, 0, d.parts), 0, d.slices);
    const uint32_t res_record = resident_record(tp->ld, d.record, tp->k, max_cell_rows);
    const bool resident =
        resident_env && d.slices > gn2v_host::kCursorSlices &&
        !(tp->flags & (GN2V_TRAIN_ATOMIC | GN2V_TRAIN_WRITE_THROUGH | GN2V_TRAIN_WRITE_BACK)) &&
        res_record != 0;
    int wmc = (tp->flags & GN2V_TRAIN_ATOMIC)          ? gn2v::kAtomic
              : (tp->flags & GN2V_TRAIN_WRITE_BACK)    ? gn2v::kWriteBack
              : (tp->flags & GN2V_TRAIN_WRITE_THROUGH) ? gn2v::kWriteThrough
              : (g->view.n_nodes < (1ULL << 16) &&
                 !(exclusive && g->view.n_nodes >= GN2V_BLOCK_PATH_MIN_NODES))
                  ? gn2v::kAtomic
                  : gn2v::kWriteThrough;
    int wmx = wmc;
    if (wmc == gn2v::kWriteThrough && exclusive) wmx = gn2v::kWriteBack;
    a.xcds = (uint32_t)g->n_xcds;
    a.central_store = (tp->flags & GN2V_TRAIN_CENTRAL_STORE) ? 1u : 0u;
    // graphs whose contextual table lives in the L2s (and in the Infinity Cache): the second read
    // of a row right before its stores costs no HBM traffic and shrinks the window in which a
    // racing store is lost (block_kernels.h score_sample); GN2V_BLOCK_REREAD=0 / 1 overrides
    static const size_t reread_env = env_size("GN2V_BLOCK_REREAD", 2);
    a.reread = reread_env == 2 ? ((uint64_t)g->view.n_nodes * tp->ld * 4 <= (64ull << 20) ? 1u : 0u)
                               : (uint32_t)reread_env;

    // Hot rows (block_kernels.h "hot rows"): only with ONE workgroup of sixteen waves per CU --
    // rows up to 128 floats, store flavours, central rows by atomics -- whose waves share one set
    // of LDS copies: 32 workgroups per XCD instead of 128 keep what is in limbo between the copies
    // small, and the LDS holds four times the rows (100 at d = 128).  The hand-over period T
    // follows from the stability bound "unseen updates x learning rate": a workgroup does not see
    // the other workgroups' pending sums (T / 2 each on average) nor what they handed over since
    // its own last hand-over (T each): 1.5 T (W - 1) updates for W workgroups side by side on the
    // cell.  Measured on BA 1 M (d = 128, lr = 0.01): 380 such updates train well, 770 diverge;
    // the largest T in {8, 4, 2, 1} that keeps 1.5 T W lr <= 2 is taken (lr = 0.01: T = 4), and
    // when even T = 1 does not, the rows stay ordinary rows (plain stores lose updates on hub rows
    // -- which is also what keeps them stable at any learning rate).
    const bool stores = !det && wmx != gn2v::kAtomic;
    if (d.slices > gn2v_host::kCursorSlices && !resident)
        return fail("more than 16 slices need cells that fit a workgroup's LDS (default update "
                    "mode, rows up to 512 floats)");
    if (a.inv && !resident)
        return fail("a placement (d_inv) needs resident cells: only their kernel reaches the rows "
                    "through it");
    if (part_n > 1) {
        *took_group = resident;
        if (!resident) return 0;
        a.part_ptrs = d_part_ptrs;
    }
    if (resident) {
        a.p.record = res_record;
        const bool v2 = resident_v2();
        if (!v2 && det)
            return fail("GN2V_RESIDENT_V2=0 (round 4's resident kernel) is a speed A/B only: its "
                        "deterministic form assumes that pairs of equal centre are neighbours, "
                        "which the sort key of resident plans no longer guarantees");
        const size_t lds =
            v2 ? (size_t)gn2v::res_words_per_wave(tp->ld, res_record, tp->k) * 4 *
                         resident_waves(tp->ld) +
                     (size_t)(max_cell_rows + 1) * gn2v::res_lds_stride(tp->ld) * 4 +
                     (size_t)max_cell_rows * 12 + 16
               : block_lds_words_per_wave(tp->ld, res_record, tp->k) * 4 * 16 +
                     (size_t)max_cell_rows * (tp->ld * 4 + 4) + 16;
        a.hot_n = (uint32_t)max_cell_rows;  // the rows the LDS plan is made for
        std::lock_guard<std::mutex> lock(g->mu);
        hipStream_t caller = s;
        if (g->train_stream) {
            HIP_TRY(hipEventRecord(g->ts_in, caller));
            HIP_TRY(hipStreamWaitEvent(g->train_stream, g->ts_in, 0));
            s = g->train_stream;
        }
        EventPair ev;
        if (get_events(g, &ev)) return 1;
        HIP_TRY(hipEventRecord(ev.a, s));
        if (det) a.sweep = part_n;  // the deterministic form walks the parts itself
        const dim3 grid(d.slices, part_n);
        if (v2 && !det && lpt_order(g, d.slices * part_n, io->part * d.slices,
                                    (const unsigned long long *)io->d_cell_offsets, &a.order, s))
            return 1;
        if (v2) {
            if (tp->ld <= 64)
                launch_resident_v2_ch<1>(det, grid, lds, s, a);
            else if (tp->ld <= 128)
                launch_resident_v2_ch<2>(det, grid, lds, s, a);
            else if (tp->ld <= 256)
                launch_resident_v2_ch<4>(det, grid, lds, s, a);
            else
                launch_resident_v2_ch<8>(det, grid, lds, s, a);
        } else if (tp->ld <= 64)
            launch_resident_ch<1>(det, grid, lds, s, a);
        else if (tp->ld <= 128)
            launch_resident_ch<2>(det, grid, lds, s, a);
        else
            launch_resident_ch<4>(det, grid, lds, s, a);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev.b, s));
        g->train_events.push_back(ev);
        g->train_launches++;
        g->resident_launches++;
        g->resident_record = res_record;
        if (g->train_stream) {
            HIP_TRY(hipEventRecord(g->ts_out, s));
            HIP_TRY(hipStreamWaitEvent(caller, g->ts_out, 0));
        }
        return 0;
    }
    static const size_t wide_env = env_size("GN2V_BLOCK_WIDE", 1);  // 0: never (A/B)
    const size_t per_wave_words = block_lds_words_per_wave(tp->ld, d.record, tp->k);
    // Not on graphs whose tables live in the L2s (a.reread): there a sample costs ~30 ns of a
    // wave's time, and a hand-over -- a returning atomic's round trip -- costs a hundred of them
    // (BA 2 708 nodes, 100 of a cell's 338 rows hot: 1.06e8 instead of 8.4e8 pairs/s); such graphs
    // get the second read before the store instead.
    bool wide = stores && a.hot_list && d.hot_rows && wide_env && tp->ld <= 128 && !a.reread &&
                wmc != gn2v::kWriteBack && per_wave_words * 4 * 16 <= 96 * 1024;
    const uint64_t cus = (uint64_t)g->n_cus - (uint64_t)g->reserved_cus * std::max(1, g->n_xcds);
    uint32_t hot_period = 0;
    if (wide) {
        const uint64_t rows = g->view.n_nodes / d.parts;
        const uint64_t wgs = std::min<uint64_t>(cus, std::max<uint64_t>(d.slices, rows / 16));
        const double side_by_side = (double)std::max<uint64_t>(1, exclusive ? wgs / g->n_xcds : wgs);
        hot_period = plan->hot_flush;
        if (!hot_period)
            for (uint32_t t = 8; t >= 1 && !hot_period; t >>= 1)
                if (1.5 * t * side_by_side * std::fabs(lr) <= 2.0) hot_period = t;
        static const size_t flush_override = env_size("GN2V_HOT_FLUSH_OVERRIDE", 0);  // A/B
        if (flush_override) hot_period = (uint32_t)flush_override;
        if (!hot_period) wide = false;
    }
    // A graph whose hub alone is more than a CU's share of all pairs (max in-degree x CUs > edges:
    // a star of 120 k leaves on BA 1 M) DIVERGES with hot rows at lr = 0.01 -- |x| 2e8 after
    // three epochs, link AUROC 0.45; the hand-over period above is derived for hubs that are a
    // small part of a cell's samples -- and trains to 0.993 without them, faster
    // (profiles/r05_logs/r5_skew_ab*.log): its rows stay ordinary rows.  (The in-degrees are
    // known here: the hot list was made from them.)
    if (wide && g->max_in_degree_known &&
        (double)g->max_in_degree * g->n_cus > (double)g->view.n_edges)
        wide = false;
    const int waves_per_block = det ? 1 : wide ? 16 : gn2v::kTrainBlock / 64;
    size_t lds = (size_t)waves_per_block * per_wave_words * 4;
    if (lds > 64 * 1024 && !wide) return fail("record / negatives too large for the LDS plan");
    if (wide) {
        static const size_t budget = env_size("GN2V_HOT_LDS_BYTES", 160 * 1024);
        const size_t per_row = (size_t)tp->ld * 8 + 8;
        if (budget > lds) a.hot_n = (uint32_t)std::min<size_t>(d.hot_rows, (budget - lds) / per_row);
        static const size_t cap = env_size("GN2V_HOT_ROWS", GN2V_BLOCK_HOT_MAX);
        a.hot_n = (uint32_t)std::min<size_t>(a.hot_n, cap);
        a.hot_mask = hot_period - 1u;
        lds += (size_t)a.hot_n * per_row;
    }
    uint64_t blocks = det ? 1 : wide ? cus : cus * 8;
    if (!det) {
        // at most one concurrent wave per table row on average (staleness of the records of one
        // centre trained from the same copy of its row; binds on tiny graphs only)
        // Graphs whose tables live in the L2s (a.reread) are not bandwidth bound: every wave more
        // is one more racer on the same few rows.  One wave per FOUR rows there -- BA 2 708 nodes,
        // 10 epochs: cosine-of-central AUROC 0.9937 at 3.0e8 pairs/s, against 0.9845 at 9.0e8 with
        // a wave per row (atomics on every row: 0.9961 at 5.8e7); the contextual table moves 0.82
        // instead of 0.61 x as far as the sequential schedule, the central one 0.98 instead of
        // 1.10 x (overshoot: several runs of a centre in flight add gradients of one stale row).
        static const size_t rows_env = env_size("GN2V_BLOCK_ROWS_PER_WAVE", 0);  // A/B
        const size_t rows_per_wave = rows_env ? rows_env : a.reread ? 4 : 1;
        const uint64_t rows = g->view.n_nodes / d.parts / rows_per_wave;
        const uint64_t max_blocks = std::max<uint64_t>(d.slices, rows / waves_per_block);
        if (blocks > max_blocks) blocks = max_blocks;
    }
    dim3 grid((unsigned)blocks), block(det ? 64 : waves_per_block * 64);

    std::lock_guard<std::mutex> lock(g->mu);
    // gn2v_graph_reserve_cus: the launch runs on the CU-masked stream, between the caller's
    // stream's past and future
    hipStream_t caller = s;
    if (g->train_stream) {
        HIP_TRY(hipEventRecord(g->ts_in, caller));
        HIP_TRY(hipStreamWaitEvent(g->train_stream, g->ts_in, 0));
        s = g->train_stream;
    }
    a.cursors = g->cursors + (size_t)(g->cursor_slot++ % kCursorRing) * kCursorWords;
    HIP_TRY(hipMemsetAsync(a.cursors, 0, kCursorWords * sizeof(unsigned long long), s));
    EventPair ev;
    if (get_events(g, &ev)) return 1;
    HIP_TRY(hipEventRecord(ev.a, s));
    const uint32_t nchunks = tp->ld / 4;
    for (int pass = 0; pass < ((!det && d.slices > 1) ? 2 : 1); ++pass) {
        // pass 1 (sliced parts only): whatever the XCD placement left behind is finished by
        // every workgroup with write-through stores (normally nothing: the kernel exits at once)
        a.sweep = pass;
        const int x = pass ? (wmc == gn2v::kAtomic ? gn2v::kAtomic : gn2v::kWriteThrough) : wmx;
        // the sweep normally finds every cell finished: a small grid keeps its ticket reads cheap
        if (pass) grid = dim3(std::min<unsigned>(grid.x, 8 * d.slices));
        if (nchunks <= 16)
            launch_block_ch<1>(x, wmc, det, grid, block, lds, s, a);
        else if (nchunks <= 32)
            launch_block_ch<2>(x, wmc, det, grid, block, lds, s, a);
        else if (nchunks <= 64)
            launch_block_ch<4>(x, wmc, det, grid, block, lds, s, a);
        else if (nchunks <= 128)
            launch_block_ch<8>(x, wmc, det, grid, block, lds, s, a);
        else
            launch_block_ch<16>(x, wmc, det, grid, block, lds, s, a);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(ev.b, s));
    g->train_events.push_back(ev);
    g->train_launches++;
    if (g->train_stream) {
        HIP_TRY(hipEventRecord(g->ts_out, s));
        HIP_TRY(hipStreamWaitEvent(caller, g->ts_out, 0));
    }
    return 0;
}

int gn2v_block_step(gn2v_graph *g, const gn2v_train_params *tp, const gn2v_block_plan *plan,
                    const gn2v_block_io *io, uint64_t seed, uint64_t epoch, float lr,
                    void *stream) {
    return block_step(g, tp, plan, io, seed, epoch, lr, stream, 1, nullptr, nullptr);
}

}  // extern "C"

namespace {
// largest in-degree of the graph (computed once per handle): the most frequent context
int max_in_degree(gn2v_graph *g, hipStream_t s, uint64_t *out) {
    if (!g->max_in_degree_known) {
        uint32_t *indeg = nullptr;
        if (in_degrees(g, s, &indeg)) return fail("out of device memory for the in-degrees");
    }
    *out = g->max_in_degree;
    return 0;
}

int auto_plan(uint64_t n_nodes, uint32_t world, uint32_t ld, uint32_t k, bool allow_resident,
              uint32_t *parts, uint32_t *slices) {
    if (!parts || !slices || world < 1) return fail("bad arguments");
    constexpr uint64_t kMinRows = 32768, kXcds = 8;
    // Rows up to 256 floats, a graph of GN2V_RESIDENT_MIN_NODES nodes or more that is
    // small enough for cells that fit a workgroup's LDS (524 288 cells x ~200 rows at d = 128: 115 M
    // nodes): RESIDENT CELLS -- every contextual row is read and updated in the LDS of the one
    // workgroup that owns its cell (sgns_resident_kernel).  As few cells as hold the rows, up to
    // 256 slices per part (a launch covers a part: one workgroup per cell and CU).  Smaller
    // graphs keep the XCD cells: with cells of ~200 rows the cosine of the central vectors of a
    // CONVERGED fit separates edges from random pairs less well (link AUROC 0.978 vs 0.996 at
    // 2 708 nodes, 0.994 vs 0.998 at 20 k; equal from 200 k nodes, 0.956 vs 0.920 at 1 M after
    // three epochs: DESIGN.md 7.3) -- the negatives of a pair come from its context's cell.
    const uint64_t max_nodes = env_size("GN2V_RESIDENT_MAX_NODES", GN2V_RESIDENT_MAX_NODES);
    const uint64_t min_nodes = env_size("GN2V_RESIDENT_MIN_NODES", GN2V_RESIDENT_MIN_NODES);
    const uint64_t fit = allow_resident && n_nodes >= min_nodes && n_nodes <= max_nodes
                             ? resident_fit(ld, k) : 0;
    if (fit >= 16 && n_nodes <= fit * (gn2v::kMaxCells - 512)) {
        const uint64_t cells = (n_nodes + fit - 1) / fit;
        // one GPU: 256 slices per part (gn2v_block_round launches a group of parts at once)
        constexpr uint64_t kOneGpuSlices = 256;
        uint64_t sl = std::min<uint64_t>(cells, kOneGpuSlices), p = (cells + sl - 1) / sl;
        // Several ranks: the parts travel -- a multiple of the ranks, two per rank -- and a part is
        // launched by itself (it leaves for the neighbour after its episode), so it brings as many
        // cells as it can (up to 8 192 slices): a launch cannot end before its heaviest cell, and
        // the striping puts one of the graph's oldest hubs into every part; with thousands of
        // workgroups per launch the other cells keep the CUs busy meanwhile.  Graphs too small
        // for 64 cells a part keep the XCD cells.
        if (world > 1) {
            p = std::max<uint64_t>(2ull * world,
                                   ((cells + kMaxSlices - 1) / kMaxSlices + world - 1) / world * world);
            sl = std::min<uint64_t>(kMaxSlices, (cells + p - 1) / p);
        }
        const uint64_t sl_max = world > 1 ? kMaxSlices : kOneGpuSlices;
        // striping rounds up twice: make sure the largest cell fits
        while (gn2v::stripe_count(gn2v::stripe_count(n_nodes, 0, p), 0, sl) > fit) {
            if (world > 1 && sl < sl_max)
                ++sl;
            else
                p += world;
        }
        // (gn2v_block_step runs cells in LDS for plans of more than 16 slices only)
        if (p * sl <= gn2v::kMaxCells && sl > gn2v_host::kCursorSlices && (world == 1 || sl >= 64)) {
            *parts = (uint32_t)p;
            *slices = (uint32_t)sl;
            return 0;
        }
    }
    // Slices: one per XCD or none.  Only then is a contextual row exclusive to one XCD's L2 (plain
    // write-back stores, hub rows L2 resident); 2 or 4 slices would have several XCDs share a
    // slice and fall back to write-through stores without the locality.
    const uint64_t min_parts = world > 1 ? 2ull * world : 1;
    // One GPU: from GN2V_BLOCK_PATH_MIN_NODES nodes up (a single part: the slices only divide the
    // rows among the XCDs, nothing concentrates in time).  Travelling parts: cells of 8 k rows.
    const uint64_t sl = (world == 1 ? n_nodes >= GN2V_BLOCK_PATH_MIN_NODES
                                    : n_nodes / (min_parts * kXcds) >= kMinRows / 4)
                            ? kXcds
                            : 1;
    // Parts: as many as keep kMinRows rows in a cell (any count: 10 M nodes -> 38, 100 M -> 381);
    // a multiple of the ranks when they travel, at least two per rank.
    uint64_t p = n_nodes / (sl * kMinRows);
    const uint64_t max_parts = gn2v::kMaxCells / sl / world * world;
    p = p / world * world;
    p = std::max<uint64_t>(min_parts, std::min<uint64_t>(p, max_parts));
    while (p > min_parts && n_nodes / p == 0) p -= world;
    *parts = (uint32_t)p;
    *slices = (uint32_t)sl;
    return 0;
}
}  // namespace

extern "C" {

int gn2v_block_auto_plan(uint64_t n_nodes, uint32_t world, uint32_t ld, uint32_t k,
                         uint32_t *parts, uint32_t *slices) {
    return auto_plan(n_nodes, world, ld, k, true, parts, slices);
}

int gn2v_block_auto_plan_graph(gn2v_graph *g, uint32_t world, uint32_t ld, uint32_t k,
                               uint32_t *parts, uint32_t *slices, void *stream) {
    if (!g) return fail("NULL handle");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    if (auto_plan(g->view.n_nodes, world, ld, k, true, parts, slices)) return 1;
    if (*slices <= 8) return 0;  // XCD cells
    // Resident cells: one workgroup per cell, and a launch cannot end before its heaviest cell.
    // The cell of the most frequent context receives h = in_degree / edges of ALL pairs on top of
    // its 1 / cells, while a balanced round takes 1 / CUs of them per CU: the other cells of the
    // launch keep the CUs busy meanwhile (the heaviest cell is started first, lpt_order), but a
    // round cannot take less than h.  Round 4 sent graphs with h x CUs > 1 to the XCD cells;
    // measured since (BA 1 M + a star of 120 k / 300 k leaves, h x CUs = 1.6 / 3.8,
    // profiles/r05_logs/r5_skew_ab.log): resident cells 1.48e9 / 7.5e8 pairs/s against 7.0 /
    // 5.9e8, and the same link AUROC as the atomic modes -- so the rule now only applies from
    // h x CUs = 8 (where a round is all hub cell), GN2V_RESIDENT_MAX_SKEW_PCT = 100 restores it.
    uint64_t hub = 0;
    if (max_in_degree(g, (hipStream_t)stream, &hub)) return 1;
    const size_t skew_pct = env_size("GN2V_RESIDENT_MAX_SKEW_PCT", 800);
    if ((double)hub * g->n_cus * 100.0 > (double)skew_pct * (double)g->view.n_edges)
        return auto_plan(g->view.n_nodes, world, ld, k, false, parts, slices);
    return 0;
}


int gn2v_block_round_plan(uint64_t free_bytes, uint64_t n_nodes, uint32_t walk_length,
                          uint32_t window, uint32_t world, uint32_t parts, uint32_t slices,
                          uint32_t overlap, uint64_t *round_walks, uint32_t *group_parts) {
    if (!round_walks || !group_parts || walk_length < 2 || window < 1 || world < 1 || parts < 1 ||
        slices < 1 || n_nodes < 1)
        return fail("bad arguments");
    const uint64_t L = walk_length, pairs = 2ull * window * L;  // per walk (window untrimmed)
    // Long enough that the pairs of a centre meet in a cell -- 64 per (cell, centre) -- within
    // [2^20, 2^23] walks (the bench graph wants more than the cap: 3.5 at 2^23).
    const double want = 64.0 * (double)n_nodes * parts * slices / ((double)world * pairs);
    uint64_t r = 1ull << 20;
    while ((double)r < want && r < (1ull << 23)) r *= 2;
    // *round_walks on entry: the caller's cap (0 = none) -- the rounds-per-epoch rule of resident
    // cells, a walk budget: the groups below are sized for the round that will really be trained
    // (a rank of 8 on the bench graph trains rounds of 2^19 walks: ONE group holds them, where
    // the 2^23 this function would take by itself needed three)
    if (*round_walks && *round_walks < r) r = *round_walks;
    // Memory, three quarters of what is free: the round's walks (this rank's and, with several
    // ranks, the gathered ones), and per group of parts the pair words once sorted (twice when a
    // second group is prepared while the first trains) and once unsorted.  Groups: at least four
    // per round when there are that many parts, more when memory is short.
    const uint64_t budget = free_bytes / 4 * 3;
    const uint64_t copies = overlap ? 3 : 2;
    // (resident cells -- more than 16 slices -- are trained under a round's placement: the walks
    // the extraction reads exist a second time, with placed node ids)
    const uint64_t copies_of_walks = (world > 1 ? world + 1ull : 1ull) + (slices > 16 ? world : 0);
    auto walk_bytes = [&](uint64_t rw) { return 4 * L * rw * copies_of_walks; };
    auto group_bytes = [&](uint64_t rw, uint64_t gp) {  // + 1/8: parts are not equally heavy
        const uint64_t per_part = rw * pairs / parts + 1;
        return copies * 8 * (per_part * gp + per_part * gp / 8);
    };
    while (r > (1ull << 14) && walk_bytes(r) + group_bytes(r, 1) > budget) r /= 2;
    // (one GPU in resident cells: ONE too when the group fits what a handle keeps between fits
    // and the cells a group may hold -- with the rounds of the rounds-per-epoch rule it does on
    // the bench graph: every group scans the round's walks twice, a launch of all 45 568 cells
    // ends on a shorter tail than one of 7 680, and beside-the-training preparation hides a
    // seventh of itself only.  Bench graph, preparation in line, groups a round: 6 2.198e9,
    // 3 2.237, 2 2.250, 1 2.263 pairs/s -- against 2.222 with 6 groups prepared beside the
    // training (profiles/r06_logs/r6_kernel_alone_and_preparation.log).  Round 5 took six because
    // rounds of 2^23 walks made a group's pair buffers larger than a handle keeps)
    // (several ranks in resident cells: ONE when memory allows -- every scan of a group reads the
    // walks of ALL ranks, and a group may be wide there: more cells than the counting pass has
    // LDS counters; a rank of 8 on the bench graph, one group a round against two: 2.07 against
    // 2.01e9 pairs/s, a rank of 4 2.21 / 2.16, profiles/r06_logs/r6_kernel_alone_and_preparation.log)
    const uint64_t min_groups = slices > 16 ? 1 : 4;
    uint64_t gp = std::max<uint64_t>(1, (parts + min_groups - 1) / min_groups);
    // the extraction counts the cells of a group in LDS: kMaxGroupCells at most, and fewer when
    // the walk's staging leaves less of the 64 KB
    const uint64_t lds_cells =
        std::min<uint64_t>(gn2v::kMaxGroupCells,
                           (64 * 1024 - std::min<size_t>(extract_lds_bytes(walk_length, 0), 60 * 1024)) / 4);
    // Resident cells on one GPU: a group is ONE launch, one workgroup per cell, and a launch of
    // few cells ends on its heaviest ones -- BA 1 M (18 parts x 256 cells), kernel pairs/s by
    // cells per launch: 768 1.79e9, 1 280 2.01e9, 2 304 2.15e9, 4 608 2.25e9
    // (profiles/r05_logs/r5_n1m_ab.log).  So a group takes at least 4 096 cells when the plan has
    // them (in equal groups), and when the pair words of such a group do not fit -- or exceed the
    // third of the memory a handle keeps between fits -- the ROUND gets shorter (down to 2^20
    // walks) before the group gets smaller.
    if (world == 1 && slices > 16) {
        const uint64_t min_gp = std::min<uint64_t>(parts, (4096 + slices - 1) / slices);
        uint64_t floor_gp = gp;
        if (gp < min_gp) {
            uint64_t groups = (parts + min_gp - 1) / min_gp;
            // (two groups of a round twice as long hold what one group of the whole round holds:
            // the larger launch wins -- 2.07e9 against 2.00e9 pairs/s on BA 1 M)
            if (groups == 2 && parts * slices <= GN2V_BLOCK_MAX_WIDE_GROUP_CELLS) groups = 1;
            floor_gp = gp = (parts + groups - 1) / groups;
        } else {
            floor_gp = min_gp;
        }
        const uint64_t wide_gp = std::max<uint64_t>(1, GN2V_BLOCK_MAX_WIDE_GROUP_CELLS / slices);
        gp = std::min(gp, wide_gp);
        floor_gp = std::min(floor_gp, gp);
        const uint64_t keep = std::min<uint64_t>(budget, free_bytes / 3);
        auto too_big = [&](uint64_t rw, uint64_t g_) {
            return walk_bytes(rw) + group_bytes(rw, g_) > budget || group_bytes(rw, g_) > keep;
        };
        // the round is shortened for the sake of the floor only; a group larger than the floor
        // (a wide group: BA 100 M) is cut down to what the handle keeps instead
        while (r > (1ull << 20) && too_big(r, floor_gp)) r /= 2;
        while (gp > floor_gp && too_big(r, gp)) --gp;
    }
    // (resident plans: a group may be wide -- gn2v_block_cell_offsets; BA 100 M then takes 7 groups
    // of 254 parts a round instead of 33 of 54)
    const uint64_t cap_cells = slices > 16 ? (uint64_t)GN2V_BLOCK_MAX_WIDE_GROUP_CELLS : lds_cells;
    gp = std::max<uint64_t>(1, std::min(gp, cap_cells / slices));
    while (gp > 1 && walk_bytes(r) + group_bytes(r, gp) > budget) --gp;
    // equal groups: as many as that size needs, none of them a remainder of a part or two
    gp = (parts + (parts + gp - 1) / gp - 1) / ((parts + gp - 1) / gp);
    *round_walks = r;
    *group_parts = (uint32_t)gp;
    return 0;
}

int gn2v_block_round(gn2v_graph *g, const gn2v_train_params *tp, const gn2v_block_plan *plans,
                     uint32_t stripes, gn2v_block_round_io *io, uint64_t n_walks, uint64_t seed,
                     uint64_t epoch, uint64_t first_walk, float lr, uint64_t round_id,
                     void *stream) {
    if (!g || !tp || !plans || !io) return fail("NULL handle / params / plans / io");
    if (stripes < 1 || stripes > 64) return fail("stripes must be in [1, 64]");
    for (uint32_t j = 0; j < stripes; ++j) {
        if (check_plan(g, &plans[j])) return 1;
        if (plans[j].world != stripes || plans[j].rank != j || plans[j].parts != plans[0].parts ||
            plans[j].slices != plans[0].slices)
            return fail("plans[j] must be the plan of stripe j of `stripes` (world = stripes, "
                        "rank = j, one geometry)");
    }
    if (!io->d_central || !io->context_parts || !io->d_work || !io->d_cell_offsets)
        return fail("NULL pointer in the round's io");
    if ((io->d_inv != nullptr) != (io->d_placed_walks != nullptr))
        return fail("a round under a placement needs d_inv and d_placed_walks together");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    hipStream_t s = (hipStream_t)stream;
    const uint32_t parts = plans[0].parts, cells = parts * plans[0].slices, ld = tp->ld;
    const uint32_t gp = io->group_parts ? std::min(io->group_parts, parts) : parts;
    const uint32_t groups = (parts + gp - 1) / gp;
    if (io->next_unit > stripes * groups) return fail("next_unit beyond the round");

    // Two sets of (pair words, cell offsets, work): unit u + 1 is counted, extracted and sorted
    // on a stream of the handle's own WHILE unit u trains -- the training kernel of resident
    // cells is bound by the L2 atomic units, the preparation by HBM, so they overlap (the
    // kernel's workgroups hold their CUs' LDS: gn2v_graph_reserve_cus leaves the preparation
    // CUs of its own).  One set: everything in line on the caller's stream.
    const bool overlapped = io->d_pairs2 && io->d_cell_offsets2 && io->d_work2;
    uint64_t *const pairs_of[2] = {io->d_pairs, overlapped ? io->d_pairs2 : io->d_pairs};
    uint64_t *const offsets_of[2] = {io->d_cell_offsets,
                                     overlapped ? io->d_cell_offsets2 : io->d_cell_offsets};
    uint64_t *const work_of[2] = {io->d_work, overlapped ? io->d_work2 : io->d_work};
    hipStream_t side = s;
    if (overlapped) {
        if (!g->prep_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&g->prep_stream, hipStreamNonBlocking));
            for (int i = 0; i < 2; ++i) {
                HIP_TRY(hipEventCreateWithFlags(&g->prep_done[i], hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&g->train_done[i], hipEventDisableTiming));
            }
        }
        side = g->prep_stream;
        // whatever the caller put on its stream (the walks, the placement, the alias tables)
        // comes before the first preparation
        HIP_TRY(hipEventRecord(g->train_done[0], s));
        HIP_TRY(hipStreamWaitEvent(side, g->train_done[0], 0));
        HIP_TRY(hipEventRecord(g->train_done[1], s));
    }

    // count + one host read + extract + sort of unit u into set `slot`; *n_pairs = 0: nothing
    // to train; returns GN2V_ROUND_GROW when the unit needs larger buffers
    auto prepare = [&](uint32_t u, int slot, uint64_t *n_pairs) -> int {
        const uint32_t j = u / groups, p0 = (u % groups) * gp, pn = std::min(gp, parts - p0);
        const gn2v_block_plan *pj = &plans[j];
        if (gn2v_block_count(g, pj, io->d_walks, io->d_placed_walks, n_walks, seed, epoch,
                             first_walk, p0, pn, work_of[slot], offsets_of[slot], side))
            return 1;
        HIP_TRY(hipMemcpyAsync(n_pairs, offsets_of[slot] + cells, 8, hipMemcpyDeviceToHost, side));
        HIP_TRY(hipStreamSynchronize(side));  // the one host read of the group
        if (*n_pairs == 0) return 0;
        uint64_t need = 0;
        gn2v_block_extract_temp_bytes(*n_pairs, &need);
        if (*n_pairs > io->pairs_capacity || need > io->temp_bytes || !io->d_pairs || !io->d_temp) {
            io->needed_pairs = *n_pairs;
            return GN2V_ROUND_GROW;
        }
        if (gn2v_block_extract(g, pj, io->d_walks, io->d_placed_walks, n_walks, seed, epoch,
                               first_walk, p0, pn, work_of[slot], io->d_hub_bits, *n_pairs,
                               pairs_of[slot], io->d_temp, io->temp_bytes, side))
            return 1;
        if (gn2v_block_cell_offsets(g, pj, pn, pairs_of[slot], *n_pairs, offsets_of[slot], side))
            return 1;
        if (overlapped) HIP_TRY(hipEventRecord(g->prep_done[slot], side));
        return 0;
    };
    auto train = [&](uint32_t u, int slot) -> int {
        const uint32_t j = u / groups, p0 = (u % groups) * gp, pn = std::min(gp, parts - p0);
        const gn2v_block_plan *pj = &plans[j];
        if (overlapped) HIP_TRY(hipStreamWaitEvent(s, g->prep_done[slot], 0));
        if (io->train_after) HIP_TRY(hipStreamWaitEvent(s, (hipEvent_t)io->train_after, 0));
        gn2v_block_io step{};
        step.d_pairs = pairs_of[slot];
        step.d_cell_offsets = offsets_of[slot];
        step.d_alias = io->d_alias;
        step.d_cell_rows = io->d_cell_rows;
        step.d_hot_list = io->d_hot_list;
        step.d_hot_slot = io->d_hot_slot;
        step.d_central = io->d_central + (size_t)j * ld;
        step.central_ld = (uint64_t)stripes * ld;
        step.context_ld = io->context_ld;
        step.d_inv = io->d_inv;
        step.d_context_table = io->d_context_table;
        step.block_id = round_id * stripes + j;
        // resident cells: the whole group in one launch (its workgroups are handed to the CUs
        // as they fall free; a launch per part would wait for the part's heaviest cell)
        bool took_group = false;
        if (pn > 1) {
            std::vector<float *> ptrs(io->context_parts, io->context_parts + parts);
            if (!g->part_ptrs_dev || ptrs != g->part_ptrs_host) {
                HIP_TRY(hipStreamSynchronize(s));  // no launch may still read the old pointers
                if (g->part_ptrs_dev && g->part_ptrs_host.size() < parts) {
                    (void)hipFree(g->part_ptrs_dev);
                    g->part_ptrs_dev = nullptr;
                }
                if (!g->part_ptrs_dev)
                    HIP_TRY(hipMalloc((void **)&g->part_ptrs_dev, parts * sizeof(float *)));
                HIP_TRY(hipMemcpy(g->part_ptrs_dev, ptrs.data(), parts * sizeof(float *),
                                  hipMemcpyHostToDevice));
                g->part_ptrs_host = ptrs;
            }
            step.d_context = nullptr;
            step.part = p0;
            if (block_step(g, tp, pj, &step, seed, epoch, lr, s, pn, g->part_ptrs_dev, &took_group))
                return 1;
        }
        for (uint32_t p = p0; p < p0 + pn && !took_group; ++p) {
            step.d_context = io->context_parts[p];
            step.part = p;
            if (gn2v_block_step(g, tp, pj, &step, seed, epoch, lr, s)) return 1;
        }
        if (overlapped) HIP_TRY(hipEventRecord(g->train_done[slot], s));
        return 0;
    };

    const uint32_t n_units = stripes * groups;
    int slot = 0;
    uint64_t n_pairs = 0;
    bool prepared = false;  // unit io->next_unit sits in `slot`, ready to train
    while (io->next_unit < n_units) {
        if (!prepared) {
            const int rc = prepare(io->next_unit, slot, &n_pairs);
            if (rc) {
                if (rc == GN2V_ROUND_GROW && overlapped) HIP_TRY(hipStreamSynchronize(s));
                return rc;
            }
        }
        prepared = false;
        const uint32_t u = io->next_unit;
        if (n_pairs) {
            if (train(u, slot)) return 1;
            io->pairs_trained += n_pairs;
        }
        io->next_unit = u + 1;
        if (overlapped && u + 1 < n_units) {
            // the next unit's preparation beside this unit's training: its set was last read by
            // the training of unit u - 1
            slot ^= 1;
            HIP_TRY(hipStreamWaitEvent(side, g->train_done[slot], 0));
            const int rc = prepare(u + 1, slot, &n_pairs);
            if (rc) {
                if (rc == GN2V_ROUND_GROW) HIP_TRY(hipStreamSynchronize(s));
                return rc;
            }
            prepared = true;
        }
        if (g->train_events.size() > 2048) {  // bound the event pool on long fits
            HIP_TRY(hipStreamSynchronize(s));
            gn2v_stats scratch;
            if (gn2v_stats_read(g, &scratch, s)) return 1;
        }
    }
    if (overlapped) {  // the caller's stream ends after everything the side stream did
        HIP_TRY(hipEventRecord(g->prep_done[0], side));
        HIP_TRY(hipStreamWaitEvent(s, g->prep_done[0], 0));
    }
    return 0;
}

// gn2v_train_blocks returns this (instead of 1) when it cannot run this fit -- device memory ran
// out, or the walk / the sample list of a record does not fit the kernels' LDS plans -- before
// anything was trained or initialised: gn2v_train then falls back to the walk-ordered schedule.
static constexpr int kOutOfMemory = 2;


int gn2v_train_blocks(gn2v_graph *g, const gn2v_walk_params *wp, const gn2v_train_params *tp,
                      uint64_t seed, uint64_t max_walks_per_epoch, uint64_t round_walks,
                      uint32_t stripes, float *d_central, float *d_contextual, gn2v_stats *stats,
                      void *stream) {
    if (!g || !wp || !tp) return fail("NULL handle / params");
    if (tp->model != GN2V_MODEL_SKIPGRAM) return fail("the block path trains SkipGram only");
    if (!d_central || !d_contextual) return fail("NULL table pointer");
    if (stripes > 64) return fail("at most 64 centre stripes");
    DeviceGuard guard(g->device);
    if (!guard.ok()) return fail("cannot select the graph's HIP device");
    hipStream_t s = (hipStream_t)stream;
    const uint64_t n = g->view.n_nodes;
    const uint32_t L = wp->walk_length, w = tp->window, ld = tp->ld;
    // GN2V_TRAIN_TIMING=1: host-side phase times of this call on stderr (each mark synchronises)
    const bool timing = getenv("GN2V_TRAIN_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[gn2v_train_blocks] %-28s %9.2f ms\n", what,
                std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };

    // Centre stripes ("virtual ranks"): stripe j = the centres c with c % V == j is trained over
    // the pairs of ALL the round's walks before stripe j + 1 -- what V ranks do side by side.
    // Off unless asked for: the stripes of a round are trained one after the other, not side by
    // side as ranks would, and that coarser order costs link quality.
    uint32_t V = stripes ? stripes : 1;
    while (V > 1 && n / V < 2) V /= 2;

    // the walk sampler's own memory first: everything below is sized from what is free
    if (prepare_walk_sampler(g, wp, s)) return 1;

    gn2v_block_plan plan{};
    plan.world = V;
    plan.rank = 0;
    if (gn2v_block_auto_plan_graph(g, 1, ld, tp->k, &plan.parts, &plan.slices, s)) return 1;
    plan.walk_length = L;
    plan.window = w;
    plan.min_dist = tp->min_dist ? tp->min_dist : 1;
    // records of 32 pairs, fewer when 1 + k samples per pair would not fit a workgroup's LDS; a
    // walk the extraction cannot stage, or a sample list too long even for records of 8, is not
    // for this path (status 2, before anything is touched: gn2v_train takes the walk-ordered one)
    plan.record = 0;
    for (uint32_t r = 32; r >= 8 && !plan.record; r >>= 1)
        if (block_lds_words_per_wave(ld, r, tp->k) * 4 * (gn2v::kTrainBlock / 64) <= 64 * 1024)
            plan.record = r;
    if (!plan.record || extract_lds_bytes(L, plan.slices) > 64 * 1024) {  // a group of one part
        fail("walk_length / number_of_negative_samples beyond the block path's LDS plans");
        return kOutOfMemory;
    }
    plan.flags = tp->flags & GN2V_TRAIN_DOWNSAMPLE;
    // Resident cells (more than 16 slices): no hot rows -- every row of a cell lives in LDS -- and
    // a PLACEMENT per round (block_kernels.h "Placement of a round"): the cell-mates of a node,
    // among which the negatives of a pair are drawn, change from round to round.
    // GN2V_BLOCK_PERMUTE=0: the fixed cells of round 4 (A/B).
    const bool resident_plan = plan.slices > gn2v_host::kCursorSlices;
    const bool permute = resident_plan && env_size("GN2V_BLOCK_PERMUTE", 1) != 0;
    plan.hot_rows = resident_plan ? 0 : (uint32_t)env_size("GN2V_HOT_ROWS", GN2V_BLOCK_HOT_DEFAULT);
    plan.hot_flush = (uint32_t)env_size("GN2V_HOT_FLUSH", 0);
    std::vector<gn2v_block_plan> plans(V, plan);
    for (uint32_t j = 0; j < V; ++j) {
        plans[j].rank = j;
        if (gn2v_block_plan_check(g, &plans[j])) return 1;
    }
    plan = plans[0];
    const uint32_t parts = plan.parts, cells = parts * plan.slices;
    const bool scale_free = tp->flags & GN2V_TRAIN_SCALE_FREE;
    mark("plan");

    uint64_t walks_per_epoch = g->view.n_sources * (uint64_t)wp->iterations;
    if (max_walks_per_epoch && max_walks_per_epoch < walks_per_epoch)
        walks_per_epoch = max_walks_per_epoch;
    const uint64_t pairs_per_walk = 2ull * w * L;  // upper bound (window untrimmed)

    Buffers buf(g);
    // alias tables + the hot rows of every cell (flags for the extraction, slot tables)
    uint64_t *alias = nullptr, *cell_rows = nullptr;
    uint32_t *hub_bits = nullptr, *hot_list = nullptr;
    uint8_t *hot_slot = nullptr;
    // under a placement the alias tables are rebuilt every round: their temporary storage stays
    uint32_t *place = nullptr, *inv = nullptr;
    void *alias_tmp = nullptr, *place_tmp = nullptr;
    uint64_t alias_tb = 0, place_tb = 0;
    if (scale_free || plan.hot_rows) {  // uniform negatives: the hot rows (frequent contexts) only
        gn2v_block_alias_temp_bytes(n, &alias_tb);
        if (buf.alloc(&alias, n * 8) || buf.alloc(&cell_rows, (cells + 1) * 8)) return kOutOfMemory;
        if (plan.hot_rows &&
            (buf.alloc(&hub_bits, ((n + 31) / 32) * 4) ||
             buf.alloc(&hot_list, (size_t)cells * GN2V_BLOCK_HOT_MAX * 4) || buf.alloc(&hot_slot, n)))
            return kOutOfMemory;
        if (buf.alloc(&alias_tmp, alias_tb)) return kOutOfMemory;
        if (!permute) {
            if (gn2v_block_alias(g, &plan, alias, cell_rows, hub_bits, hot_list, hot_slot, nullptr,
                                 alias_tmp, alias_tb, s))
                return 1;
            HIP_TRY(hipStreamSynchronize(s));
            buf.release_last();
            alias_tmp = nullptr;
        }
    }
    if (permute) {
        if (gn2v_block_placement_temp_bytes(n, &place_tb)) return 1;
        if (buf.alloc(&place, n * 4) || buf.alloc(&inv, n * 4) || buf.alloc(&place_tmp, place_tb))
            return kOutOfMemory;
    }

    mark("alias tables");
    // round size and groups of parts: `round_walks` = the walks one pass extracts from (a round is
    // V times that); the pairs of a round are extracted, sorted and trained a group at a time.
    // Resident cells: the next group is prepared on a second stream while this one trains (a
    // second set of pair words; gn2v_block_round) -- GN2V_BLOCK_OVERLAP=0: in line (A/B).
    const bool overlap = resident_plan && env_size("GN2V_BLOCK_OVERLAP", 1) != 0;
    const bool automatic = round_walks == 0;
    uint32_t group_parts = 0;
    size_t free_b = 0, total_b = 0;
    {
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        {
            // what the handle keeps from its last fit is this fit's to reuse: a second fit must
            // plan from the same budget as the first (the same rounds, groups and -- the round ids
            // and placements depend on them -- the same embeddings for a seed)
            std::lock_guard<std::mutex> lock(g->kept_mu);
            free_b += g->kept_bytes;
        }
        // Resident cells draw a pair's negatives among the ~220 cell-mates its context has THIS
        // round: an epoch trained as one or two rounds shows every context two or three sets of
        // mates, and the cosine of the central rows pays for it -- config 3's shape (169 k
        // nodes, three epochs of ten walks a node), cos-central AUROC by rounds per epoch:
        // 1 0.9893, 4 0.9946, 8 0.9958, 16 0.9964 (walk-ordered kernel with graph-wide
        // negatives: 0.9972) at the same kernel speed (profiles/r06_logs/r6_rounds_quality.log);
        // rank correlation of the cosines with that kernel's: 16 0.936, 26 0.946, 64 0.955,
        // 128 0.958 where it agrees with ITSELF under other negatives to 0.969
        // (r6_quality_gates_*.log).  What counts is how many sets of mates a context meets
        // while its row forms -- rounds over the whole fit -- so an epoch of the graph
        // (iterations x sources walks, whatever the caller's walk budget) is cut into
        // 192 / epochs rounds, at least 16 and at most 64 (rounds_per_epoch), none shorter than
        // 2^14 walks.  Cost of 64 against 16: 1.6 % on the bench graph, 12 % at 169 k nodes.
        // The plan is made for that round (its cap), or for the caller's.
        uint64_t auto_walks = round_walks;
        if (automatic && permute) {
            const uint64_t rounds = gn2v_host::rounds_per_epoch(tp->epochs);
            const uint64_t epoch = g->view.n_sources * (uint64_t)wp->iterations;
            const uint64_t shortest =
                std::max<uint64_t>(1, env_size("GN2V_ROUND_MIN_WALKS", 1ull << 14));
            auto_walks = std::max<uint64_t>(shortest, (epoch + rounds * V - 1) / (rounds * V));
        }
        if (gn2v_block_round_plan(free_b, n, L, w, V, parts, plan.slices, overlap ? 1 : 0,
                                  &auto_walks, &group_parts))
            return 1;
        if (automatic) round_walks = auto_walks;
    }
    const uint64_t planned_walks = round_walks;
    round_walks = std::max<uint64_t>(1, std::min(round_walks, (walks_per_epoch + V - 1) / V));
    if (automatic) {
        // equal rounds: an epoch of 2.5 planned rounds is trained as three of 5 / 6 of the size,
        // not as two and a half (the half round pays a whole round's placement, alias tables and
        // first preparation for half the pairs)
        const uint64_t per_pass = (walks_per_epoch + V - 1) / V;
        const uint64_t n_rounds = (per_pass + round_walks - 1) / round_walks;
        round_walks = std::max<uint64_t>(1, (per_pass + n_rounds - 1) / n_rounds);
    }
    // GN2V_ROUND_BUFFERS_FOR=<walks per epoch>: a caller that will come back to this handle with
    // that many walks (bench.py's warm-up before its timed call) has the round buffers sized for
    // it now, so that the later call finds them kept -- memory touched for the first time inside
    // a timed region costs what the driver needs to clear it (1.5 s of 14 on a fresh box).
    uint64_t buffer_walks = round_walks;
    if (const uint64_t later = env_size("GN2V_ROUND_BUFFERS_FOR", 0))
        buffer_walks = std::max(round_walks, std::min(planned_walks, (later + V - 1) / V));
    uint32_t *walks = nullptr, *placed = nullptr;
    uint64_t *pairs = nullptr, *pairs2 = nullptr, *work = nullptr, *work2 = nullptr,
             *cell_offsets = nullptr, *cell_offsets2 = nullptr, *part_first = nullptr;
    void *tmp = nullptr;
    uint64_t tb = 0, cap = 0;
    if (buf.alloc(&work, GN2V_BLOCK_WORK_WORDS * 8) || buf.alloc(&cell_offsets, (cells + 1) * 8) ||
        buf.alloc(&part_first, (parts + 1) * 8))
        return kOutOfMemory;
    if (overlap && (buf.alloc(&work2, GN2V_BLOCK_WORK_WORDS * 8) ||
                    buf.alloc(&cell_offsets2, (cells + 1) * 8)))
        return kOutOfMemory;
    const size_t held = buf.ptrs.size();
    auto release_round = [&]() {
        while (buf.ptrs.size() > held) buf.release_last();
    };
    for (;;) {
        // a group's share of the round's pairs by its parts, 1 / 8 of head room on the
        // untrimmed-window bound (parts and stripes are not equally heavy), more on demand (below)
        cap = buffer_walks * pairs_per_walk / parts * group_parts;
        cap += cap / 8 + 1024;
        gn2v_block_extract_temp_bytes(cap, &tb);
        // (the second set of pair words only where a round has a second unit to prepare beside
        // the first: a round that is one group of one stripe trains what it has just prepared)
        const bool second_set = overlap && (group_parts < parts || V > 1);
        pairs2 = nullptr;
        if (!(buf.alloc(&walks, V * buffer_walks * L * 4) ||
              (permute && buf.alloc(&placed, V * buffer_walks * L * 4)) ||
              (second_set && buf.alloc(&pairs2, cap * 8)) || buf.alloc(&pairs, cap * 8) ||
              buf.alloc(&tmp, tb)))
            break;
        // somebody else took the memory between the query and here: smaller groups, then an
        // automatic round halves
        release_round();
        if (group_parts > 1)
            group_parts = (group_parts + 1) / 2;
        else if (buffer_walks > round_walks)
            buffer_walks = round_walks;  // no room for the later call's buffers: this call's
        else if (automatic && round_walks > (1u << 14))
            buffer_walks = round_walks /= 2;
        else
            return kOutOfMemory;
    }
    const uint64_t super_walks = V * round_walks;
    mark("round buffers");

    // The contextual table is trained in the caller's buffer -- no third table exists during the
    // fit -- stored part by part (the rows of part p one after the other from row first_row[p]: a
    // part is one contiguous range for its launches); the natural order is restored at the end
    // through one scratch copy, for which the pair buffers make room.  When even that copy would
    // not fit what is free now (or GN2V_BLOCK_LAYOUT=natural asks for it), the rows stay where
    // they belong and part p is the rows p, p + parts, ... (gn2v_block_io.context_ld).
    const size_t table_bytes = (size_t)n * ld * sizeof(float);
    const char *layout = getenv("GN2V_BLOCK_LAYOUT");
    // (under a placement the kernel reaches every row through the placement's inverse: the
    // table simply stays in node order)
    const bool part_major = !permute && parts > 1 && !(layout && !strcmp(layout, "natural")) &&
                            (free_b > table_bytes + ((size_t)1 << 28) ||
                             (layout && !strcmp(layout, "parts")));
    std::vector<uint64_t> first_row(parts + 1, 0);
    for (uint32_t p = 0; p < parts; ++p)
        first_row[p + 1] = first_row[p] + gn2v::stripe_count(n, p, parts);
    HIP_TRY(hipMemcpyAsync(part_first, first_row.data(), (parts + 1) * 8, hipMemcpyHostToDevice, s));
    if (gn2v_init_table(d_central, n, tp->d, ld, seed, 0, tp->init_scale, s)) return 1;
    if (!part_major) {
        if (gn2v_init_table(d_contextual, n, tp->d, ld, seed, 1, tp->init_scale, s)) return 1;
    } else {
        for (uint32_t p = 0; p < parts; ++p)
            if (gn2v_init_table_rows(d_contextual + first_row[p] * ld,
                                     first_row[p + 1] - first_row[p], tp->d, ld, seed, 1,
                                     tp->init_scale, p, parts, s))
                return 1;
    }

    std::vector<float *> part_rows(parts);
    for (uint32_t p = 0; p < parts; ++p)
        part_rows[p] = part_major ? d_contextual + first_row[p] * ld : d_contextual + (size_t)p * ld;
    gn2v_block_round_io rio{};
    rio.d_walks = walks;
    rio.d_placed_walks = placed;
    rio.d_inv = inv;
    rio.d_context_table = permute ? d_contextual : nullptr;
    rio.d_alias = alias;
    rio.d_cell_rows = cell_rows;
    rio.d_hub_bits = hub_bits;
    rio.d_hot_list = hot_list;
    rio.d_hot_slot = hot_slot;
    rio.d_central = d_central;
    rio.context_parts = part_rows.data();
    rio.context_ld = part_major ? 0 : (uint64_t)parts * ld;
    rio.d_work = work;
    rio.d_cell_offsets = cell_offsets;
    rio.d_pairs = pairs;
    rio.pairs_capacity = cap;
    rio.d_temp = tmp;
    rio.temp_bytes = tb;
    rio.group_parts = group_parts;
    rio.d_pairs2 = pairs2;
    rio.d_cell_offsets2 = cell_offsets2;
    rio.d_work2 = work2;

    mark("tables initialised");
    // Two lanes of round buffers (a round that is ONE unit: one group of one stripe under a
    // placement): round t + 1 -- its walks, placement, alias tables, placed walks, count,
    // extraction and sort -- is enqueued on the other lane's stream while round t trains; only
    // its training launch waits for round t's (two rounds in training at once would stage the
    // same contextual rows in two cells).  What it hides: the host's wait for a round's pair
    // count and a dozen small launches per round -- a tenth of the time when a round trains for
    // 17 ms -- and the share of the preparation that runs beside a training kernel at all (a
    // seventh to a third: DESIGN.md 7).  What it costs: a second set of round buffers.  Same
    // box, pairs/s with two lanes against one, and the second set: config 3's shape (rounds of
    // 26 k walks) 1.94 / 1.75e9, 0.8 GB; config 4's (383 k) 2.28 / 2.23e9, 9.5 GB; the bench
    // graph (1.5 M) 2.30 / 2.27e9, 38 GB -- a seventh of the device for 1 %: two lanes are taken
    // while the second set stays under 16 GiB (GN2V_ROUND_LANES_MAX_BYTES;
    // profiles/r06_logs/r6_round_lanes_ab.log).  The two lanes share the temporary storage of
    // placement and alias tables: the host reads a round's pair count (one stream
    // synchronisation) after enqueueing them, so the other lane's are over before the next ones
    // are enqueued.  GN2V_ROUND_LANES=1: one lane.
    struct Lane {
        uint32_t *walks = nullptr, *placed = nullptr, *place = nullptr, *inv = nullptr;
        uint64_t *alias = nullptr, *cell_rows = nullptr;
        gn2v_block_round_io rio{};
    } lane[2];
    lane[0].walks = walks;
    lane[0].placed = placed;
    lane[0].place = place;
    lane[0].inv = inv;
    lane[0].alias = alias;
    lane[0].cell_rows = cell_rows;
    lane[0].rio = rio;
    const uint64_t rounds_per_epoch_here = (walks_per_epoch + super_walks - 1) / super_walks;
    const uint64_t second_set = 2 * V * buffer_walks * L * 4 + (uint64_t)n * 16 + cap * 8 + tb;
    bool lanes = permute && V == 1 && group_parts >= parts && overlap && !pairs2 &&
                 !(tp->flags & GN2V_TRAIN_DETERMINISTIC) &&
                 rounds_per_epoch_here * tp->epochs >= 2 && env_size("GN2V_ROUND_LANES", 2) >= 2 &&
                 second_set <= env_size("GN2V_ROUND_LANES_MAX_BYTES", 16ull << 30);
    if (lanes) {
        const size_t before = buf.ptrs.size();
        Lane &b = lane[1];
        uint64_t *pairs_b = nullptr;
        void *tmp_b = nullptr;
        if (buf.alloc(&b.walks, V * buffer_walks * L * 4) ||
            buf.alloc(&b.placed, V * buffer_walks * L * 4) || buf.alloc(&b.place, n * 4) ||
            buf.alloc(&b.inv, n * 4) || (alias && buf.alloc(&b.alias, n * 8)) ||
            (cell_rows && buf.alloc(&b.cell_rows, (cells + 1) * 8)) ||
            buf.alloc(&pairs_b, cap * 8) || buf.alloc(&tmp_b, tb)) {
            while (buf.ptrs.size() > before) buf.free_last();  // no room: one lane
            lanes = false;
        } else {
            b.rio = rio;
            b.rio.d_walks = b.walks;
            b.rio.d_placed_walks = b.placed;
            b.rio.d_inv = b.inv;
            b.rio.d_alias = b.alias;
            b.rio.d_cell_rows = b.cell_rows;
            b.rio.d_work = work2;
            b.rio.d_cell_offsets = cell_offsets2;
            b.rio.d_pairs = pairs_b;
            b.rio.d_temp = tmp_b;
            for (int i = 0; i < 2; ++i) {
                lane[i].rio.d_pairs2 = nullptr;
                lane[i].rio.d_cell_offsets2 = nullptr;
                lane[i].rio.d_work2 = nullptr;
                if (!g->lane_stream[i]) {
                    HIP_TRY(hipStreamCreateWithFlags(&g->lane_stream[i], hipStreamNonBlocking));
                    HIP_TRY(hipEventCreateWithFlags(&g->lane_done[i], hipEventDisableTiming));
                }
            }
            if (!g->lane_start)
                HIP_TRY(hipEventCreateWithFlags(&g->lane_start, hipEventDisableTiming));
            // the tables initialised on the caller's stream come before both lanes
            HIP_TRY(hipEventRecord(g->lane_start, s));
            for (int i = 0; i < 2; ++i) HIP_TRY(hipStreamWaitEvent(g->lane_stream[i], g->lane_start, 0));
        }
    }
    float lr = tp->lr;
    uint64_t round_id = 0;
    for (uint32_t e = 0; e < tp->epochs; ++e) {
        for (uint64_t first = 0; first < walks_per_epoch; first += super_walks, ++round_id) {
            const uint64_t nw = std::min(super_walks, walks_per_epoch - first);
            const int li = lanes ? (int)(round_id & 1) : 0;
            Lane &ln = lane[li];
            hipStream_t ls = lanes ? g->lane_stream[li] : s;
            if (gn2v_walks(g, wp, seed, e, first, nw, ln.walks, ls)) return 1;
            if (permute) {  // this round's cells
                if (gn2v_block_placement(g, 1, seed, round_id, ln.place, ln.inv, place_tmp,
                                         place_tb, ls))
                    return 1;
                if (ln.alias && gn2v_block_alias(g, &plan, ln.alias, ln.cell_rows, nullptr, nullptr,
                                                 nullptr, ln.inv, alias_tmp, alias_tb, ls))
                    return 1;
                if (gn2v_block_place_walks(ln.place, ln.walks, nw * L, ln.placed, ls)) return 1;
            }
            ln.rio.next_unit = 0;
            ln.rio.train_after = lanes && round_id > 0 ? (void *)g->lane_done[li ^ 1] : nullptr;
            for (;;) {
                const int rc = gn2v_block_round(g, tp, plans.data(), V, &ln.rio, nw, seed, e, first,
                                                lr, round_id, ls);
                if (rc == 0) break;
                if (rc != GN2V_ROUND_GROW) return 1;
                if (lanes)
                    return fail("a round of two lanes outgrew buffers sized for its upper bound");
                // a group heavier than the head room allows: grow (the driver has waited for
                // whatever still trained from these buffers)
                buf.free_last();  // tmp
                buf.free_last();  // pairs
                if (pairs2) buf.free_last();
                cap = ln.rio.needed_pairs + ln.rio.needed_pairs / 16;
                gn2v_block_extract_temp_bytes(cap, &tb);
                if ((pairs2 && buf.alloc(&pairs2, cap * 8)) || buf.alloc(&pairs, cap * 8) ||
                    buf.alloc(&tmp, tb))
                    return 1;
                ln.rio.d_pairs2 = pairs2;
                ln.rio.d_pairs = pairs;
                ln.rio.pairs_capacity = cap;
                ln.rio.d_temp = tmp;
                ln.rio.temp_bytes = tb;
            }
            if (lanes) HIP_TRY(hipEventRecord(g->lane_done[li], ls));
        }
        lr *= tp->lr_decay;
    }
    if (lanes) {  // the caller's stream continues after both lanes
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipEventRecord(g->lane_done[i], g->lane_stream[i]));
            HIP_TRY(hipStreamWaitEvent(s, g->lane_done[i], 0));
        }
    }
    HIP_TRY(hipStreamSynchronize(s));
    mark("rounds");
    release_round();
    mark("round buffers released");
    if (part_major) {  // part-major -> natural order
        float *scratch = nullptr;
        if (!getenv("GN2V_BLOCK_RESTORE_ON_HOST") && buf.alloc(&scratch, table_bytes) == 0) {
            HIP_TRY(hipMemcpyAsync(scratch, d_contextual, table_bytes, hipMemcpyDeviceToDevice, s));
            const unsigned blocks =
                (unsigned)std::min<uint64_t>((n * (ld >> 2) + 255) / 256, 256 * 32);
            hipLaunchKernelGGL(gn2v::parts_to_natural_kernel, dim3(blocks), dim3(256), 0, s,
                               d_contextual, scratch, (const unsigned long long *)part_first, n,
                               ld, parts);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(s));
        } else {
            // somebody took the memory the copy was planned for while the fit ran: the trained
            // table is not given up -- it goes through host memory, a part at a time coming back
            // as a strided copy into its rows p, p + parts, ...
            (void)hipGetLastError();
            float *host = (float *)malloc(table_bytes);
            if (!host)
                return fail("out of device and host memory while restoring the row order of the "
                            "contextual table (GN2V_BLOCK_LAYOUT=natural trains without that "
                            "copy)");
            hipError_t e = hipMemcpyAsync(host, d_contextual, table_bytes, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            for (uint32_t p = 0; p < parts && e == hipSuccess; ++p)
                e = hipMemcpy2DAsync(d_contextual + (size_t)p * ld, (size_t)parts * ld * 4,
                                     host + first_row[p] * ld, (size_t)ld * 4, (size_t)ld * 4,
                                     first_row[p + 1] - first_row[p], hipMemcpyHostToDevice, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            free(host);
            if (e != hipSuccess)
                return fail(std::string("restoring the row order through host memory: ") +
                            hipGetErrorString(e));
        }
    }
    mark("node order restored");
    if (stats) {
        if (gn2v_stats_read(g, stats, s)) return 1;
        stats->block_parts = parts;
        stats->block_slices = plan.slices;
        stats->block_stripes = V;
        stats->block_group_parts = group_parts;
        stats->block_round_walks = round_walks;
    }
    buf.done = true;  // streams drained above: what is left goes back to the handle
    return 0;
}

}  // extern "C"

#include "world_driver.h"

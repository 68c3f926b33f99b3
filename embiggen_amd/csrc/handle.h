// Host-side pieces shared by the translation units of libgn2v.so: the opaque graph handle of
// include/gn2v.h, error reporting, the device guard and the launch-time bookkeeping.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gn2v_internal.h"
#include "walk_kernels.h"

namespace gn2v_host {

int fail(const std::string &msg);  // records the thread's last error, returns 1

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return gn2v_host::fail(std::string(#expr) + ": " + hipGetErrorString(e_) +    \
                                   " (" __FILE__ ":" + std::to_string(__LINE__) + ")");   \
    } while (0)

struct EventPair {
    hipEvent_t a, b;
};

// record-ticket cursors of the block trainer: up to 16 slices per launch, every cursor on a
// 512 B line of its own (eight XCDs taking tickets from neighbouring words of one line
// serialised on that line); one array per launch, reused round robin (a launch is long over
// before its slot comes round again)
constexpr unsigned kCursorSlices = 16, kCursorStride = 64, kCursorRing = 256;
constexpr unsigned kCursorWords = kCursorSlices * kCursorStride;

// Every entry point runs on the device of its graph handle (or of the buffers it is given) and
// leaves the calling thread's current HIP device as it found it: a one-process multi-device
// program (PyTorch keeps its own notion of the current device) must not see it change.
class DeviceGuard {
  public:
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
        if (device >= 0 && device != prev_) {
            ok_ = hipSetDevice(device) == hipSuccess;
            switched_ = ok_;
        }
    }
    // device that owns `ptr` (device memory); falls back to the current device
    static int of_pointer(const void *ptr) {
        hipPointerAttribute_t attr;
        if (ptr && hipPointerGetAttributes(&attr, ptr) == hipSuccess) return attr.device;
        (void)hipGetLastError();
        return -1;
    }
    ~DeviceGuard() {
        if (switched_ && prev_ >= 0) (void)hipSetDevice(prev_);
    }
    bool ok() const { return ok_; }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;

  private:
    int prev_ = -1;
    bool ok_ = true, switched_ = false;
};

}  // namespace gn2v_host

struct gn2v_graph {
    gn2v::GraphView view{};
    int device = 0;
    int n_cus = 256;
    // XCDs (one L2 each) workgroups are spread over, ids 0 .. n_xcds - 1, found by a probe launch
    // when the handle is made; 0 = unknown (then no row is ever treated as exclusive to one XCD)
    int n_xcds = 0;
    bool owns = false;
    void *own_row_ptr = nullptr, *own_col_idx = nullptr, *own_cumw = nullptr,
         *own_sources = nullptr, *own_node_types = nullptr, *own_edge_types = nullptr;
    // gn2v_graph_reserve_cus: the training kernels run on a stream whose CU mask leaves some CUs
    // of every XCD to other work (RCCL's transfer kernels); nullptr: the caller's stream
    hipStream_t train_stream = nullptr;
    hipEvent_t ts_in = nullptr, ts_out = nullptr;
    uint32_t reserved_cus = 0;
    // the second-order sampler's edge set (GraphView.edge_set), built on the first biased walk
    unsigned long long *edge_set = nullptr, *edge_filter = nullptr;
    bool edge_set_tried = false;
    // the sampler's edge records (GraphView.edge_rec / node_sig), built on the first unweighted walk
    uint4 *edge_rec = nullptr, *edge_rec_typed = nullptr;
    uint32_t *node_sig = nullptr;
    bool edge_rec_tried = false, edge_rec_typed_tried = false;
    // gn2v_block_round with two sets of pair buffers: the preparation's own stream and the
    // events that order it against the training (created on first use)
    hipStream_t prep_stream = nullptr;
    hipEvent_t prep_done[2] = {nullptr, nullptr}, train_done[2] = {nullptr, nullptr};
    // gn2v_train_blocks with two lanes of round buffers: round t + 1 is prepared on the other
    // lane's stream while round t trains (created on first use)
    hipStream_t lane_stream[2] = {nullptr, nullptr};
    hipEvent_t lane_done[2] = {nullptr, nullptr}, lane_start = nullptr;
    // resident cells, a group of parts per launch: the parts' row pointers on the device
    std::vector<float *> part_ptrs_host;
    float **part_ptrs_dev = nullptr;
    // largest in-degree (the most frequent context), computed on the first automatic plan
    uint64_t max_in_degree = 0;
    bool max_in_degree_known = false;
    // largest out-degree (max_neighbours acts on longer rows only), on the first such walk
    uint64_t max_out_degree = 0;
    bool max_out_degree_known = false;
    uint32_t *indeg = nullptr;  // u32[n_nodes + 1] (the last word: their maximum), kept once computed
    bool indeg_failed = false;
    // round buffers of the last block fit, kept for the next one (gn2v_block_api.hip Buffers)
    std::vector<std::pair<void *, size_t>> kept_buffers;
    size_t kept_bytes = 0;
    std::mutex kept_mu;  // guards kept_buffers / kept_bytes (never taken together with `mu`)
    // resident launches, heaviest cell first: ring of (keys, values) x (in, out) + sort storage
    uint32_t *lpt = nullptr;
    void *lpt_temp = nullptr;
    size_t lpt_temp_bytes = 0;
    uint32_t lpt_slot = 0;
    unsigned long long *counters = nullptr;  // device, 4 x u64
    unsigned long long *cursors = nullptr;   // device, ring of record-ticket arrays (block trainer)
    uint32_t cursor_slot = 0;
    std::vector<gn2v_host::EventPair> train_events, walk_events, free_events;
    double train_ms = 0.0, walk_ms = 0.0;
    uint32_t train_launches = 0, walk_launches = 0;
    uint32_t resident_launches = 0, resident_record = 0;  // sgns_resident_kernel (gn2v_stats)
    std::mutex mu;  // guards the event / timing bookkeeping (launches themselves are stream ordered)
};

namespace gn2v_host {
// Rounds an epoch of the graph is cut into under a placement (resident cells): 192 over the
// whole fit, at least 16 and at most 64 an epoch; GN2V_ROUNDS_PER_EPOCH pins it (A/B).
inline uint64_t rounds_per_epoch(uint32_t epochs) {
    if (const char *v = getenv("GN2V_ROUNDS_PER_EPOCH"))
        if (*v) return std::max<uint64_t>(1, strtoull(v, nullptr, 10));
    const uint64_t e = epochs ? epochs : 1;
    return std::min<uint64_t>(64, std::max<uint64_t>(16, (192 + e - 1) / e));
}
int get_events(gn2v_graph *g, EventPair *ev);
int prepare_walk_sampler(gn2v_graph *g, const gn2v_walk_params *wp, hipStream_t s);
void release_kept_buffers(gn2v_graph *g);
}

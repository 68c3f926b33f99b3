// Fused negative-sampling kernels: gather -> dot -> sigmoid -> scatter-add on two f32 tables.
// Replaces the SGD inside ensmallen's `SkipGram/CBOW.fit_transform`
// (reference call site embedders/ensmallen_embedders/node2vec.py:99; parameter semantics
//  node2vec_skipgram.py:37-119; model statement tensorflow_embedders/skipgram.py:28-61,
//  cbow.py:26-60).
//
// Mapping (gfx950, wave64): one wavefront owns one walk at a time.  A wave is four groups of 16
// lanes; a group owns one embedding row per round: lane q of a group holds float4 chunks
// q, q+16, ... of the row (d = 128 -> 2 x 16 B per lane, one 512 B row per group, 2 KiB per wave
// round trip).  Dot products reduce inside a group with 4 DPP steps (quad_perm, quad_perm,
// row_half_mirror, row_mirror) -- no LDS, bit-identical in every lane of the group.  The centre
// row and its gradient stay in registers across the <= 2w*(k+1) samples of a centre.  Sample rows
// are updated by racy read-modify-write 16 B stores (Hogwild, like the CPU reference; default
// write-through so every XCD sees them) or by hardware f32 atomics on HBM (exact accumulation,
// measured ~10x slower: 128 atomics per row).  HBM-bound: ~0.75 flop/B, MFMA is not used.
#pragma once
#include "rng.h"
#include "walk_kernels.h"

namespace gn2v {

constexpr uint32_t kFlagScaleFree = 1u, kFlagDownsample = 2u, kFlagNormLr = 4u;
constexpr uint32_t kNegBit = 0x80000000u;  // staged row id refers to the `negative` table

struct TrainArgs {
    GraphView g;
    const uint32_t *walks;      // global node ids [n_walks][L]
    const uint32_t *walk_rows;  // row of every walk node in central/contextual, or nullptr (= id)
    const uint32_t *neg_override;
    float *ctx_delta;           // CBOW: add the contexts' gradients here instead of to `contextual`
    float *central;
    float *contextual;
    float *negative;            // table the negative rows live in (== contextual / central when
                                // the tables are whole; the local shard in row-sharded training)
    const uint32_t *neg_pool;   // negatives = neg_pool[uniform] (rows of `negative`) or nullptr
    uint64_t neg_pool_size;
    uint32_t neg_id_mul, neg_id_add;  // global id of negative row r = r * mul + add (skip rule)
    uint32_t split;             // 1 when `negative` is a different table than the positive one
    uint32_t cache_max_degree;  // context cache: rows of nodes with degree >= this stay in HBM
                                // (0xFFFFFFFF = cache every row)
    unsigned long long *counters;  // [0] pairs, [1] walk steps, [2] centres
    uint64_t n_walks;
    uint64_t first_walk;
    uint64_t ekey;
    uint32_t L, window, k, ld, flags;
    uint32_t min_dist;  // contexts at walk distance [min_dist, window] (Walklets: == window)
    uint32_t max_samples;  // LDS list capacity per wave
    float lr, clip;
};

// The XCD (accelerator complex) a wavefront runs on.  Each XCD of an MI355X has an L2 of its own;
// L2s are not coherent with one another inside a launch.
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}

template <int CH>
struct Row {
    float4 c[CH];
};


template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}

// sum over the 16 lanes of a DPP row; every lane gets the bit-identical result
__device__ __forceinline__ float group16_sum(float x) {
    x += dpp_mov<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_mov<0x141>(x);  // row_half_mirror
    x += dpp_mov<0x140>(x);  // row_mirror
    return x;
}

// Makes LDS writes of any lane visible to every lane of the same wavefront.  LDS only: a generic
// wavefront-scope fence compiles to `s_waitcnt vmcnt(0) lgkmcnt(0)` on gfx950, i.e. it also waits
// for every outstanding global load and (write-through) store -- which serialised the prefetches
// and cost a memory round trip per call.  The LDS pipeline is in order, so `lgkmcnt(0)` plus a
// compiler barrier is all a single wave needs; global-memory ordering inside a wave relies, as
// before, on program order per address (tested: rows repeated in consecutive rounds).
__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

template <int CH>
__device__ __forceinline__ void load_row(Row<CH> &r, const float *base, int q, uint32_t nchunks,
                                         bool valid) {
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const uint32_t ci = cc * 16 + q;
        r.c[cc] = (valid && ci < nchunks) ? *reinterpret_cast<const float4 *>(base + ci * 4)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int CH>
__device__ __forceinline__ float dot_rows(const Row<CH> &a, const Row<CH> &b) {
    float s = 0.f;
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        s += a.c[cc].x * b.c[cc].x;
        s += a.c[cc].y * b.c[cc].y;
        s += a.c[cc].z * b.c[cc].z;
        s += a.c[cc].w * b.c[cc].w;
    }
    return group16_sum(s);
}

// acc += s * x
template <int CH>
__device__ __forceinline__ void axpy(Row<CH> &acc, float s, const Row<CH> &x) {
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        acc.c[cc].x += s * x.c[cc].x;
        acc.c[cc].y += s * x.c[cc].y;
        acc.c[cc].z += s * x.c[cc].z;
        acc.c[cc].w += s * x.c[cc].w;
    }
}

// How a row update reaches memory.
//   kWriteThrough: read-modify-write, 16 B stores with sc1 (write-through, line dropped from the
//                  XCD's L2) -- Hogwild like the CPU reference, visible to every XCD at once.
//   kWriteBack:    read-modify-write, plain stores (dirty lines stay private to an XCD's L2
//                  until evicted: hot rows diverge per XCD inside a launch).
//   kAtomic:       hardware f32 atomics, one per element, lane-contiguous addresses (no lost
//                  update; ~2-3x the store cost).
enum WriteMode : int { kWriteThrough = 0, kWriteBack = 1, kAtomic = 2 };

__host__ __device__ constexpr bool is_atomic(int wm) { return wm == kAtomic; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_sc1(float *p, float4 v) {
    f32x4 r = {v.x, v.y, v.z, v.w};
    // the trailing s_nop keeps hipcc from reusing the data registers before the store read them
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
}

// table row += s * x   (x: float4 shape for the store modes, lane-contiguous shape for kAtomic)
template <int CH, int WM>
__device__ __forceinline__ void scatter_add(float *base, int q, uint32_t nchunks, float s,
                                            const Row<CH> &x, const Row<CH> &old) {
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        if constexpr (is_atomic(WM)) {
            // slot (cc, e) of lane q is element 64*cc + 16*e + q (to_contig_layout), so each
            // atomic instruction covers 64 contiguous bytes per group -- measured 4x the
            // throughput of float4-shaped (16 B strided) atomics.
            float *pc = base + cc * 64 + q;
            const uint32_t f = cc * 64 + q, ldf = nchunks * 4;
            if (f < ldf) unsafeAtomicAdd(pc + 0, s * x.c[cc].x);
            if (f + 16 < ldf) unsafeAtomicAdd(pc + 16, s * x.c[cc].y);
            if (f + 32 < ldf) unsafeAtomicAdd(pc + 32, s * x.c[cc].z);
            if (f + 48 < ldf) unsafeAtomicAdd(pc + 48, s * x.c[cc].w);
        } else {
            const uint32_t ci = cc * 16 + q;
            if (ci < nchunks) {
                float *p = base + ci * 4;
                float4 o = old.c[cc];
                o.x += s * x.c[cc].x;
                o.y += s * x.c[cc].y;
                o.z += s * x.c[cc].z;
                o.w += s * x.c[cc].w;
                if constexpr (WM == kWriteThrough)
                    store_sc1(p, o);
                else
                    *reinterpret_cast<float4 *>(p) = o;
            }
        }
    }
}

// x summed over the four 16-lane rows of the wave, in every lane, with gfx950's row swaps instead
// of two ds_bpermute round trips through the LDS pipeline: v_permlane16_swap exchanges the odd
// rows of its first operand with the even rows of its second (a = [x0 x0 x2 x2], b = [x1 x1 x3 x3]
// when both start as x), v_permlane32_swap the upper half of the first with the lower half of the
// second.  Inline assembly: the clang builtin of this ROCm release returns the first result twice
// (`v_add_f32 v1, v1, v1` after the swap).  The s_nop covers the VALU-write -> permlane-read hazard.
__device__ __forceinline__ float sum_rows4(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a += b;
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// sum a register row over the four 16-lane groups of the wave
template <int CH>
__device__ __forceinline__ void reduce_groups(Row<CH> &r) {
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        float *f = reinterpret_cast<float *>(&r.c[cc]);
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = sum_rows4(f[e]);
    }
}

// Re-layout a register row from the float4 shape (slot (cc, e) of lane q = element 64cc + 4q + e)
// to the lane-contiguous shape (element 64cc + 16e + q) through a per-wave LDS row.  `in` must be
// identical in all four groups (group 0 writes).  Used once per centre in atomic mode.
template <int CH>
__device__ __forceinline__ void to_contig_layout(Row<CH> &out, const Row<CH> &in, float *s_tr,
                                                 int grp, int q, uint32_t ld) {
    wave_sync();
    if (grp == 0) {
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) {
            const uint32_t ci = cc * 16 + q;
            if (ci * 4 < ld) *reinterpret_cast<float4 *>(s_tr + ci * 4) = in.c[cc];
        }
    }
    wave_sync();
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const uint32_t f = cc * 64 + q;
        out.c[cc].x = f < ld ? s_tr[f] : 0.f;
        out.c[cc].y = f + 16 < ld ? s_tr[f + 16] : 0.f;
        out.c[cc].z = f + 32 < ld ? s_tr[f + 32] : 0.f;
        out.c[cc].w = f + 48 < ld ? s_tr[f + 48] : 0.f;
    }
}

__device__ __forceinline__ float sigmoid_clipped(float dot, float clip) {
    dot = fminf(fmaxf(dot, -clip), clip);
    return 1.0f / (1.0f + __expf(-dot));
}

// the same with v_rcp_f32 (1 ulp) in place of the IEEE division (ten instructions): for the
// racing schedules, where the order of the updates moves the result by far more
__device__ __forceinline__ float sigmoid_clipped_fast(float dot, float clip) {
    dot = fminf(fmaxf(dot, -clip), clip);
    return __builtin_amdgcn_rcpf(1.0f + __expf(-dot));
}

// row of the negative table for draw qi: a pool entry (shard-local sampling), the endpoint of a
// uniform random edge (proportional to degree) or a uniform node
__device__ __forceinline__ uint32_t draw_negative(const TrainArgs &a, uint64_t nkey, uint64_t qi) {
    const uint64_t r = draw(nkey, qi);
    if (a.neg_pool) return a.neg_pool[mulhi64(r, a.neg_pool_size)];
    if (a.flags & kFlagScaleFree) return a.g.col_idx[mulhi64(r, a.g.n_edges)];
    return (uint32_t)mulhi64(r, a.g.n_nodes);
}

__device__ __forceinline__ bool keep_centre(const TrainArgs &a, uint64_t wkey, uint32_t i,
                                            uint32_t c) {
    if (!(a.flags & kFlagDownsample)) return true;
    const uint64_t deg = a.g.row_ptr[c + 1] - a.g.row_ptr[c];
    if (deg == 0) return true;
    const uint64_t r32 = draw(wkey ^ kTagDown, i) >> 32;
    const uint64_t x = r32 * deg;  // < 2^64
    const uint64_t lhs_lo = x * a.g.n_nodes, lhs_hi = mulhi64(x, a.g.n_nodes);
    const uint64_t rhs_lo = a.g.n_edges << 32, rhs_hi = a.g.n_edges >> 32;
    return lhs_hi < rhs_hi || (lhs_hi == rhs_hi && lhs_lo < rhs_lo);
}

__device__ __forceinline__ float centre_lr(const TrainArgs &a, uint32_t c) {
    if (!(a.flags & kFlagNormLr)) return a.lr;
    const uint64_t deg = a.g.row_ptr[c + 1] - a.g.row_ptr[c];
    return deg ? a.lr / (float)deg : a.lr;
}

// Context positions of centre i: [lo, lo + n_left) and [right0, right0 + n_right), i.e. every j
// with min_dist <= |j - i| <= window inside the walk (window trimmed at the borders,
// node2vec_skipgram.py:55-57).
struct Window {
    uint32_t lo, right0, n_left, n_ctx;
    __device__ __forceinline__ Window(uint32_t i, uint32_t Le, uint32_t w, uint32_t md) {
        lo = i > w ? i - w : 0;
        const uint32_t hi = min(i + w, Le - 1);
        n_left = (i >= md && i - md >= lo) ? (i - md - lo + 1) : 0;
        right0 = i + md;
        const uint32_t n_right = right0 <= hi ? hi - right0 + 1 : 0;
        n_ctx = n_left + n_right;
    }
    __device__ __forceinline__ uint32_t position(uint32_t rank) const {
        return rank < n_left ? lo + rank : right0 + (rank - n_left);
    }
};

// Load walk b into LDS and return its effective length (first sentinel).
__device__ __forceinline__ uint32_t stage_walk(const TrainArgs &a, uint64_t b, uint32_t *s_walk,
                                               uint32_t *s_wrow, int lane) {
    uint32_t first_bad = a.L;
    for (uint32_t t = lane; t < a.L; t += 64) {
        const uint32_t v = a.walks[b * a.L + t];
        s_walk[t] = v;
        s_wrow[t] = a.walk_rows ? a.walk_rows[b * a.L + t] : v;
        if (v == kSentinel) first_bad = min(first_bad, t);
    }
    for (int off = 32; off > 0; off >>= 1) first_bad = min(first_bad, (uint32_t)__shfl_xor(first_bad, off));
    wave_sync();
    return first_bad;
}

// The four row ids a wave handles in one round (one per 16-lane group).  Rows repeated inside a
// round are serialised into passes in group order, so a single wave applies its updates in
// exactly the oracle's sequential order even when a walk revisits a node inside the window
// (a later round always sees an earlier round's stores: same wave, program order).
struct RoundIds {
    uint32_t r0, r1, r2, r3;  // named scalars: a runtime-indexed array would live in scratch
    int last_pass;
    __device__ __forceinline__ RoundIds(const uint32_t *ids, uint32_t t0, uint32_t n) {
        r0 = (t0 + 0 < n) ? ids[t0 + 0] : kSentinel;
        r1 = (t0 + 1 < n) ? ids[t0 + 1] : kSentinel;
        r2 = (t0 + 2 < n) ? ids[t0 + 2] : kSentinel;
        r3 = (t0 + 3 < n) ? ids[t0 + 3] : kSentinel;
        last_pass = max(max(pass1(), pass2()), pass3());
    }
    __device__ __forceinline__ uint32_t row_of(int g) const {
        return g == 0 ? r0 : g == 1 ? r1 : g == 2 ? r2 : r3;
    }
    // number of earlier groups of this round holding the same (valid) row
    __device__ __forceinline__ int pass1() const { return (r1 != kSentinel && r1 == r0) ? 1 : 0; }
    __device__ __forceinline__ int pass2() const {
        return r2 == kSentinel ? 0 : (int)(r2 == r0) + (int)(r2 == r1);
    }
    __device__ __forceinline__ int pass3() const {
        return r3 == kSentinel ? 0 : (int)(r3 == r0) + (int)(r3 == r1) + (int)(r3 == r2);
    }
    __device__ __forceinline__ int pass_of(int g) const {
        return g == 0 ? 0 : g == 1 ? pass1() : g == 2 ? pass2() : pass3();
    }
};

// staged row id -> address: ids carrying kNegBit index the `negative` table (split tables only)
__device__ __forceinline__ float *sample_base(const TrainArgs &a, float *table, uint32_t row) {
    if (a.split && (row & kNegBit)) return a.negative + (uint64_t)(row & ~kNegBit) * a.ld;
    return table + (uint64_t)row * a.ld;
}

// Score the staged sample list against the register row `u` (replicated in every group):
// for each sample row v: var = (label - sigmoid(clip(u.v))) * lr ; g += var * v ; v += var * u.
// u_upd is the copy of u the row update consumes (lane-contiguous shape in atomic mode).
// DET: one sample at a time, all groups redundantly, group 0 writes (strict sequential semantics).
template <int CH, int WM, bool DET, class Args>
__device__ __forceinline__ void score_samples(const Args &a, float *table, const Row<CH> &u,
                                              const Row<CH> &u_upd, Row<CH> &g,
                                              const uint32_t *s_rows,
                                              const float *s_lab, uint32_t n_samples, float lrc,
                                              int grp, int q) {
    const uint32_t nchunks = a.ld >> 2;
    if constexpr (DET) {
        for (uint32_t t = 0; t < n_samples; ++t) {
            const uint32_t row = s_rows[t];
            if (row == kSentinel) continue;
            float *base = sample_base(a, table, row);
            Row<CH> v;
            load_row<CH>(v, base, q, nchunks, true);
            const float dot = dot_rows<CH>(u, v);
            const float var = (s_lab[t] - sigmoid_clipped(dot, a.clip)) * lrc;
            axpy<CH>(g, var, v);
            if (grp == 0) scatter_add<CH, kWriteBack>(base, q, nchunks, var, u, v);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        }
    } else {
        for (uint32_t t0 = 0; t0 < n_samples; t0 += 4) {
            const uint32_t t = t0 + grp;
            const RoundIds ids(s_rows, t0, n_samples);
            const uint32_t row = ids.row_of(grp);
            const float lab = t < n_samples ? s_lab[t] : 0.f;
            const bool valid = row != kSentinel;
            const int my_pass = ids.pass_of(grp);
            float *base = sample_base(a, table, valid ? row : 0);
            for (int pass = 0; pass <= ids.last_pass; ++pass) {
                const bool mine = valid && my_pass == pass;
                Row<CH> v;
                load_row<CH>(v, base, q, nchunks, mine);
                const float dot = dot_rows<CH>(u, v);
                const float var = mine ? (lab - sigmoid_clipped(dot, a.clip)) * lrc : 0.f;
                axpy<CH>(g, var, v);
                if (mine) scatter_add<CH, WM>(base, q, nchunks, var, u_upd, v);
            }
        }
    }
}

// The parallel schedules' lean form: a round is the four sample rows t0 .. t0 + 3, one per group,
// with NO check for a row named twice inside the round.  Two groups that meet on a row (a hub
// drawn twice among four samples: ~1e-4 of the rounds in a cell of 32 k rows) both read the old
// value and one store wins -- the loss every pair of concurrent waves risks on every row anyway
// (Hogwild).  Saves the per-round duplicate analysis of score_samples (three LDS reads, the pass
// numbers, the pass loop): the block kernel issues ~800 instructions per pair with its SIMDs
// busy 77 % of the time (rocprofv3 SQ_* counters, round 3), so instructions are not free.
template <int CH, int WM, class Args>
__device__ __forceinline__ void score_samples_racy(const Args &a, float *table, const Row<CH> &u,
                                                   const Row<CH> &u_upd, Row<CH> &g,
                                                   const uint32_t *s_rows, const float *s_lab,
                                                   uint32_t n_samples, float lrc, int grp, int q) {
    const uint32_t nchunks = a.ld >> 2;
    for (uint32_t t0 = 0; t0 < n_samples; t0 += 4) {
        const uint32_t t = t0 + grp;
        const uint32_t row = t < n_samples ? s_rows[t] : kSentinel;
        const float lab = t < n_samples ? s_lab[t] : 0.f;
        const bool valid = row != kSentinel;
        float *base = sample_base(a, table, valid ? row : 0);
        Row<CH> v;
        load_row<CH>(v, base, q, nchunks, valid);
        const float dot = dot_rows<CH>(u, v);
        const float var = valid ? (lab - sigmoid_clipped(dot, a.clip)) * lrc : 0.f;
        axpy<CH>(g, var, v);
        if (valid) scatter_add<CH, WM>(base, q, nchunks, var, u_upd, v);
    }
}

// The same scoring with every sample row of the centre in flight at once (up to three rounds of
// four rows): one memory round trip per centre instead of one per round.  Precondition: no row id
// occurs twice in the list (the caller checks and otherwise takes score_samples, whose per-round
// serialisation keeps repeated rows sequentially exact).  Same accumulation order per group, so
// the result is bit-identical to score_samples on such lists.
constexpr uint32_t kFlightSamples = 12;
template <int CH, int WM, class Args>
__device__ __forceinline__ void score_samples_flight(const Args &a, float *table,
                                                     const Row<CH> &u, const Row<CH> &u_upd,
                                                     Row<CH> &g, const uint32_t *s_rows,
                                                     const float *s_lab, uint32_t n_samples,
                                                     float lrc, int grp, int q) {
    const uint32_t nchunks = a.ld >> 2;
    Row<CH> v0, v1, v2;
    const uint32_t t0 = grp, t1 = 4 + grp, t2 = 8 + grp;
    const uint32_t r0 = t0 < n_samples ? s_rows[t0] : kSentinel;
    const uint32_t r1 = t1 < n_samples ? s_rows[t1] : kSentinel;
    const uint32_t r2 = t2 < n_samples ? s_rows[t2] : kSentinel;
    const float l0 = t0 < n_samples ? s_lab[t0] : 0.f;
    const float l1 = t1 < n_samples ? s_lab[t1] : 0.f;
    const float l2 = t2 < n_samples ? s_lab[t2] : 0.f;
    float *b0 = sample_base(a, table, r0 != kSentinel ? r0 : 0);
    float *b1 = sample_base(a, table, r1 != kSentinel ? r1 : 0);
    float *b2 = sample_base(a, table, r2 != kSentinel ? r2 : 0);
    load_row<CH>(v0, b0, q, nchunks, r0 != kSentinel);
    load_row<CH>(v1, b1, q, nchunks, r1 != kSentinel);
    load_row<CH>(v2, b2, q, nchunks, r2 != kSentinel);
    {
        const float dot = dot_rows<CH>(u, v0);
        const float var = r0 != kSentinel ? (l0 - sigmoid_clipped(dot, a.clip)) * lrc : 0.f;
        axpy<CH>(g, var, v0);
        if (r0 != kSentinel) scatter_add<CH, WM>(b0, q, nchunks, var, u_upd, v0);
    }
    {
        const float dot = dot_rows<CH>(u, v1);
        const float var = r1 != kSentinel ? (l1 - sigmoid_clipped(dot, a.clip)) * lrc : 0.f;
        axpy<CH>(g, var, v1);
        if (r1 != kSentinel) scatter_add<CH, WM>(b1, q, nchunks, var, u_upd, v1);
    }
    {
        const float dot = dot_rows<CH>(u, v2);
        const float var = r2 != kSentinel ? (l2 - sigmoid_clipped(dot, a.clip)) * lrc : 0.f;
        axpy<CH>(g, var, v2);
        if (r2 != kSentinel) scatter_add<CH, WM>(b2, q, nchunks, var, u_upd, v2);
    }
}

template <int CH>
__device__ __forceinline__ void zero_row(Row<CH> &r) {
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) r.c[cc] = make_float4(0.f, 0.f, 0.f, 0.f);
}

constexpr int kTrainBlock = 256;

// SkipGram with negative sampling over a batch of walks.
// LDS per wave: transpose row[ld] | walk ids[L] | walk rows[L] | rows[max_samples] |
// labels[max_samples].
template <int CH, int WM, bool DET>
__global__ __launch_bounds__(kTrainBlock) void sgns_kernel(TrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, q = lane & 15;
    const uint32_t per_wave = (a.ld + 2 * a.L + 2 * a.max_samples + 3) & ~3u;
    float *s_tr = reinterpret_cast<float *>(smem + wave * per_wave);
    uint32_t *s_walk = smem + wave * per_wave + a.ld;
    uint32_t *s_wrow = s_walk + a.L;
    uint32_t *s_rows = s_wrow + a.L;
    const uint32_t negbit = a.split ? kNegBit : 0u;
    float *s_lab = reinterpret_cast<float *>(s_rows + a.max_samples);
    const uint32_t nchunks = a.ld >> 2;
    const uint32_t w = a.window, k = a.k;
    const uint64_t per_walk_neg = (uint64_t)a.L * 2 * w * k;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint64_t wave_stride = (uint64_t)gridDim.x * waves_per_block;

    unsigned long long pairs = 0, centres = 0;  // per wave, flushed with one atomic each at exit

    for (uint64_t b = (uint64_t)blockIdx.x * waves_per_block + wave; b < a.n_walks;
         b += wave_stride) {
        const uint32_t Le = stage_walk(a, b, s_walk, s_wrow, lane);
        const uint64_t wkey = draw(a.ekey, a.first_walk + b);
        const uint64_t nkey = wkey ^ kTagNeg;
        const uint32_t *ov = a.neg_override ? a.neg_override + b * per_walk_neg : nullptr;

        for (uint32_t i = 0; i < Le; ++i) {
            const uint32_t c = s_walk[i];
            if (!keep_centre(a, wkey, i, c)) continue;
            const float lrc = centre_lr(a, c);
            const Window win(i, Le, w, a.min_dist);
            const uint32_t n_ctx = win.n_ctx;
            const uint32_t n_samples = n_ctx * (k + 1);
            if (n_ctx == 0) continue;

            // stage the sample list of this centre: [ctx, neg_0 .. neg_{k-1}] per context slot
            wave_sync();
            for (uint32_t t = lane; t < n_samples; t += 64) {
                const uint32_t rank = t / (k + 1);
                const uint32_t s = t - rank * (k + 1);
                const uint32_t j = win.position(rank);
                const uint32_t slot = j < i ? (j + w - i) : (j + w - i - 1);
                const uint32_t ctx = s_walk[j];
                uint32_t row = s_wrow[j];
                float lab = 1.f;
                if (s != 0) {
                    const uint64_t qi = ((uint64_t)i * 2 * w + slot) * k + (s - 1);
                    row = ov ? ov[qi] : draw_negative(a, nkey, qi);
                    lab = 0.f;
                    const uint32_t gid = row * a.neg_id_mul + a.neg_id_add;
                    row = (gid == c || gid == ctx) ? kSentinel : (row | negbit);
                }
                s_rows[t] = row;
                s_lab[t] = lab;
            }
            wave_sync();

            float *crow = a.central + (uint64_t)s_wrow[i] * a.ld;
            Row<CH> u, g;
            load_row<CH>(u, crow, q, nchunks, true);
            zero_row<CH>(g);
            Row<CH> u_upd = u;
            if constexpr (!DET && WM == kAtomic) to_contig_layout<CH>(u_upd, u, s_tr, grp, q, a.ld);
            score_samples<CH, WM, DET>(a, a.contextual, u, u_upd, g, s_rows, s_lab, n_samples,
                                       lrc, grp, q);
            if constexpr (!DET) reduce_groups<CH>(g);
            if constexpr (!DET && WM == kAtomic) to_contig_layout<CH>(g, g, s_tr, grp, q, a.ld);
            if (grp == 0) scatter_add<CH, DET ? kWriteBack : WM>(crow, q, nchunks, 1.0f, g, u);
            if constexpr (DET) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            pairs += n_ctx;
            ++centres;
        }
        wave_sync();
    }
    // one atomic per wave: a per-walk atomic on a single counter serialises at ~12 ns each
    if (a.counters && lane == 0 && pairs) {
        atomicAdd(&a.counters[0], pairs);
        atomicAdd(&a.counters[2], centres);
    }
}

// ---------------------------------------------------------------------------------------------
// Walk-ordered SkipGram with a per-wave LDS cache of the window's contextual rows.
//
// In the plain kernel the contextual row of walk position j is read and written once for each of
// the <= 2w centres that see it as a context: 1 of the 11.1 rows a pair moves.  Here every wave
// keeps the rows of positions [i-w, i+w] in LDS (2w+1 slots of ld floats, a small directory keyed
// by row id with reference counts for nodes the walk revisits): a row enters when the window
// reaches it (one HBM read), every positive AND every negative sample that names it is served from
// LDS (so the wave stays sequentially consistent with the oracle: one live copy per row), and it
// is written back once when the window leaves it.  Saves ~0.9 row read + write per pair (-7.7 %
// HBM traffic).  Rows of high-degree nodes are not cached (cache_max_degree): many waves would hold
// private copies of a hub row at once and the write-back of one would discard the others' updates.
// Store modes only (atomic mode keeps the plain kernel: a cached row would need a delta).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kCacheBit = 0x40000000u;  // staged row id = kCacheBit | LDS slot

struct CtxCache {
    float *rows;       // [slots][ld]
    uint32_t *node;    // [slots] row id held by the slot
    uint32_t *ref;     // [slots] number of window positions naming it (0 = free)
    uint32_t slots, ld;

    __device__ __forceinline__ int find(uint32_t v) const {
        int hit = -1;
        for (uint32_t s = 0; s < slots; ++s)
            if (ref[s] != 0 && node[s] == v) hit = (int)s;
        return hit;
    }
};

template <int WM>
__device__ __forceinline__ void cache_insert(const TrainArgs &a, CtxCache &c, float *table,
                                             uint32_t v, int lane) {
    const int hit = c.find(v);
    wave_sync();
    if (hit >= 0) {
        if (lane == 0) c.ref[hit] += 1;
        wave_sync();
        return;
    }
    if (a.cache_max_degree != 0xFFFFFFFFu) {
        const uint64_t deg = a.g.row_ptr[v + 1] - a.g.row_ptr[v];
        if (deg >= a.cache_max_degree) return;
    }
    int free_slot = -1;
    for (uint32_t s = 0; s < c.slots; ++s)
        if (c.ref[s] == 0 && free_slot < 0) free_slot = (int)s;
    if (free_slot < 0) return;  // cannot happen: slots >= window positions
    const float *src = table + (uint64_t)v * c.ld;
    float *dst = c.rows + (uint32_t)free_slot * c.ld;
    for (uint32_t ci = lane; ci < (c.ld >> 2); ci += 64)
        *reinterpret_cast<float4 *>(dst + ci * 4) = *reinterpret_cast<const float4 *>(src + ci * 4);
    if (lane == 0) {
        c.node[free_slot] = v;
        c.ref[free_slot] = 1;
    }
    wave_sync();
}

// The row that enters the window at the NEXT centre, fetched one centre ahead into registers
// (its degree too): the insert then costs no memory round trip (win_prefetch_issue / win_insert).
// Only issued for a node that is not in the cache (a cached node is never re-read); it cannot
// become stale in between: this wave changes contextual rows only through the cache or, for
// uncached hub rows, in place -- and a hub row is discarded at commit by the same degree test.
template <int NC>  // float4 chunks per lane: ld / 4 chunks over 64 lanes
struct RowPrefetch {
    uint32_t node;
    bool valid;
    uint64_t deg;
    float4 chunk[NC];
};

template <int WM>
__device__ __forceinline__ void cache_write_back(CtxCache &c, float *table, uint32_t slot,
                                                 int lane) {
    float *dst = table + (uint64_t)c.node[slot] * c.ld;
    const float *src = c.rows + slot * c.ld;
    for (uint32_t ci = lane; ci < (c.ld >> 2); ci += 64) {
        const float4 x = *reinterpret_cast<const float4 *>(src + ci * 4);
        if constexpr (WM == kWriteThrough)
            store_sc1(dst + ci * 4, x);
        else
            *reinterpret_cast<float4 *>(dst + ci * 4) = x;
    }
}

template <int WM>
__device__ __forceinline__ void cache_retire(CtxCache &c, float *table, uint32_t v, int lane) {
    const int hit = c.find(v);
    wave_sync();
    if (hit < 0) return;
    if (c.ref[hit] == 1) cache_write_back<WM>(c, table, (uint32_t)hit, lane);
    wave_sync();
    if (lane == 0) c.ref[hit] -= 1;
    wave_sync();
}

template <int CH, int WM>
__global__ __launch_bounds__(kTrainBlock, (CH <= 2 ? 5 : CH == 4 ? 4 : 1)) void sgns_cached_kernel(TrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, q = lane & 15;
    const uint32_t w = a.window, k = a.k;
    const uint32_t slots = 2 * w + 1;
    const uint32_t per_wave = (slots * a.ld + a.L + 2 * a.max_samples + 2 * slots + 3) & ~3u;
    uint32_t *base_w = smem + wave * per_wave;
    CtxCache cache;
    cache.rows = reinterpret_cast<float *>(base_w);
    uint32_t *s_walk = base_w + slots * a.ld;
    uint32_t *s_rows = s_walk + a.L;
    float *s_lab = reinterpret_cast<float *>(s_rows + a.max_samples);
    cache.node = s_rows + 2 * a.max_samples;
    cache.ref = cache.node + slots;
    cache.slots = slots;
    cache.ld = a.ld;
    const uint32_t nchunks = a.ld >> 2;
    const uint64_t per_walk_neg = (uint64_t)a.L * 2 * w * k;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint64_t wave_stride = (uint64_t)gridDim.x * waves_per_block;
    unsigned long long pairs = 0, centres = 0;

    for (uint64_t b = (uint64_t)blockIdx.x * waves_per_block + wave; b < a.n_walks;
         b += wave_stride) {
        const uint32_t Le = stage_walk(a, b, s_walk, s_walk, lane);
        const uint64_t wkey = draw(a.ekey, a.first_walk + b);
        const uint64_t nkey = wkey ^ kTagNeg;
        const uint32_t *ov = a.neg_override ? a.neg_override + b * per_walk_neg : nullptr;
        if ((uint32_t)lane < slots) cache.ref[lane] = 0;
        wave_sync();
        for (uint32_t p = 0; p < Le && p <= w; ++p) cache_insert<WM>(a, cache, a.contextual, s_walk[p], lane);

        for (uint32_t i = 0; i < Le; ++i) {
            // slide the window: position i-w-1 leaves, position i+w enters
            if (i >= w + 1) cache_retire<WM>(cache, a.contextual, s_walk[i - w - 1], lane);
            if (i >= 1 && i + w < Le) cache_insert<WM>(a, cache, a.contextual, s_walk[i + w], lane);

            const uint32_t c = s_walk[i];
            if (!keep_centre(a, wkey, i, c)) continue;
            const float lrc = centre_lr(a, c);
            const Window win(i, Le, w, a.min_dist);
            const uint32_t n_ctx = win.n_ctx;
            const uint32_t n_samples = n_ctx * (k + 1);
            if (n_ctx == 0) continue;

            wave_sync();
            for (uint32_t t = lane; t < n_samples; t += 64) {
                const uint32_t rank = t / (k + 1);
                const uint32_t s = t - rank * (k + 1);
                const uint32_t j = win.position(rank);
                const uint32_t slot = j < i ? (j + w - i) : (j + w - i - 1);
                const uint32_t ctx = s_walk[j];
                uint32_t row = ctx;
                float lab = 1.f;
                if (s != 0) {
                    const uint64_t qi = ((uint64_t)i * 2 * w + slot) * k + (s - 1);
                    row = ov ? ov[qi] : draw_negative(a, nkey, qi);
                    lab = 0.f;
                    if (row == c || row == ctx) row = kSentinel;
                }
                if (row != kSentinel) {
                    const int hit = cache.find(row);
                    if (hit >= 0) row = kCacheBit | (uint32_t)hit;
                }
                s_rows[t] = row;
                s_lab[t] = lab;
            }
            wave_sync();

            float *crow = a.central + (uint64_t)c * a.ld;
            Row<CH> u, g;
            load_row<CH>(u, crow, q, nchunks, true);
            zero_row<CH>(g);
            for (uint32_t t0 = 0; t0 < n_samples; t0 += 4) {
                const uint32_t t = t0 + grp;
                const RoundIds ids(s_rows, t0, n_samples);
                const uint32_t row = ids.row_of(grp);
                const float lab = t < n_samples ? s_lab[t] : 0.f;
                const bool valid = row != kSentinel;
                const bool cached = valid && (row & kCacheBit);
                const int my_pass = ids.pass_of(grp);
                float *gbase = a.contextual + (uint64_t)((valid && !cached) ? row : 0) * a.ld;
                float *lbase = cache.rows + (cached ? (row & 0xFFFFu) : 0u) * a.ld;
                for (int pass = 0; pass <= ids.last_pass; ++pass) {
                    const bool mine = valid && my_pass == pass;
                    Row<CH> v;
                    load_row<CH>(v, gbase, q, nchunks, mine && !cached);
                    if (mine && cached) {
#pragma unroll
                        for (int cc = 0; cc < CH; ++cc) {
                            const uint32_t ci = cc * 16 + q;
                            if (ci < nchunks) v.c[cc] = *reinterpret_cast<const float4 *>(lbase + ci * 4);
                        }
                    }
                    const float dot = dot_rows<CH>(u, v);
                    const float var = mine ? (lab - sigmoid_clipped(dot, a.clip)) * lrc : 0.f;
                    axpy<CH>(g, var, v);
                    if (mine && !cached) scatter_add<CH, WM>(gbase, q, nchunks, var, u, v);
                    if (mine && cached) {
#pragma unroll
                        for (int cc = 0; cc < CH; ++cc) {
                            const uint32_t ci = cc * 16 + q;
                            if (ci < nchunks) {
                                float4 o = v.c[cc];
                                o.x += var * u.c[cc].x;
                                o.y += var * u.c[cc].y;
                                o.z += var * u.c[cc].z;
                                o.w += var * u.c[cc].w;
                                *reinterpret_cast<float4 *>(lbase + ci * 4) = o;
                            }
                        }
                    }
                }
            }
            reduce_groups<CH>(g);
            // the centre's own contextual row may sit in the cache, but its CENTRAL row never does
            if (grp == 0) scatter_add<CH, WM>(crow, q, nchunks, 1.0f, g, u);
            pairs += n_ctx;
            ++centres;
        }
        // the walk is over: write the remaining window back
        wave_sync();
        for (uint32_t s = 0; s < slots; ++s)
            if (cache.ref[s] != 0) cache_write_back<WM>(cache, a.contextual, s, lane);
        wave_sync();
    }
    if (a.counters && lane == 0 && pairs) {
        atomicAdd(&a.counters[0], pairs);
        atomicAdd(&a.counters[2], centres);
    }
}

// CBOW with negative sampling: h = mean of the window's *contextual* rows (input side), scored
// against the centre and k negatives in the *central* table (output side); the input gradient / C
// goes back to every context row.
template <int CH, int WM, bool DET>
__global__ __launch_bounds__(kTrainBlock) void cbow_kernel(TrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, q = lane & 15;
    const uint32_t per_wave = (a.ld + 2 * a.L + 2 * a.max_samples + 2 * a.window + 3) & ~3u;
    float *s_tr = reinterpret_cast<float *>(smem + wave * per_wave);
    uint32_t *s_walk = smem + wave * per_wave + a.ld;
    uint32_t *s_wrow = s_walk + a.L;
    uint32_t *s_rows = s_wrow + a.L;
    const uint32_t negbit = a.split ? kNegBit : 0u;
    float *s_lab = reinterpret_cast<float *>(s_rows + a.max_samples);
    uint32_t *s_ctx = s_rows + 2 * a.max_samples;  // 2w context ids of the current centre
    const uint32_t nchunks = a.ld >> 2;
    const uint32_t w = a.window, k = a.k;
    const uint64_t per_walk_neg = (uint64_t)a.L * k;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint64_t wave_stride = (uint64_t)gridDim.x * waves_per_block;

    unsigned long long pairs = 0, centres = 0;  // per wave, flushed with one atomic each at exit

    for (uint64_t b = (uint64_t)blockIdx.x * waves_per_block + wave; b < a.n_walks;
         b += wave_stride) {
        const uint32_t Le = stage_walk(a, b, s_walk, s_wrow, lane);
        const uint64_t wkey = draw(a.ekey, a.first_walk + b);
        const uint64_t nkey = wkey ^ kTagNeg;
        const uint32_t *ov = a.neg_override ? a.neg_override + b * per_walk_neg : nullptr;

        for (uint32_t i = 0; i < Le; ++i) {
            const uint32_t c = s_walk[i];
            if (!keep_centre(a, wkey, i, c)) continue;
            const float lrc = centre_lr(a, c);
            const Window win(i, Le, w, a.min_dist);
            const uint32_t n_ctx = win.n_ctx;
            if (n_ctx == 0) continue;
            const float invC = 1.0f / (float)n_ctx;

            wave_sync();
            for (uint32_t t = lane; t <= k; t += 64) {
                uint32_t row = s_wrow[i];
                float lab = 1.f;
                if (t != 0) {
                    const uint64_t qi = (uint64_t)i * k + (t - 1);
                    row = ov ? ov[qi] : draw_negative(a, nkey, qi);
                    lab = 0.f;
                    const uint32_t gid = row * a.neg_id_mul + a.neg_id_add;
                    row = gid == c ? kSentinel : (row | negbit);
                }
                s_rows[t] = row;
                s_lab[t] = lab;
            }
            for (uint32_t t = lane; t < n_ctx; t += 64) s_ctx[t] = s_wrow[win.position(t)];
            wave_sync();

            // h = mean of context rows; in DET mode every group sums all rows in walk order
            Row<CH> h, g;
            zero_row<CH>(h);
            zero_row<CH>(g);
            if constexpr (DET) {
                for (uint32_t t = 0; t < n_ctx; ++t) {
                    Row<CH> v;
                    load_row<CH>(v, a.contextual + (uint64_t)s_ctx[t] * a.ld, q, nchunks, true);
                    axpy<CH>(h, 1.0f, v);
                }
            } else {
                for (uint32_t r0 = 0; r0 < n_ctx; r0 += 4) {
                    const uint32_t rank = r0 + grp;
                    const bool in = rank < n_ctx;
                    Row<CH> v;
                    load_row<CH>(v, a.contextual + (uint64_t)(in ? s_ctx[rank] : 0) * a.ld, q,
                                 nchunks, in);
                    axpy<CH>(h, 1.0f, v);
                }
                reduce_groups<CH>(h);
            }
#pragma unroll
            for (int cc = 0; cc < CH; ++cc) {
                h.c[cc].x *= invC;
                h.c[cc].y *= invC;
                h.c[cc].z *= invC;
                h.c[cc].w *= invC;
            }

            Row<CH> h_upd = h;
            if constexpr (!DET && WM == kAtomic) to_contig_layout<CH>(h_upd, h, s_tr, grp, q, a.ld);
            score_samples<CH, WM, DET>(a, a.central, h, h_upd, g, s_rows, s_lab, k + 1, lrc, grp,
                                       q);
            if constexpr (!DET) reduce_groups<CH>(g);
            if constexpr (!DET && WM == kAtomic) to_contig_layout<CH>(g, g, s_tr, grp, q, a.ld);

            // every context row += g / C
            if constexpr (DET) {
                for (uint32_t t = 0; t < n_ctx; ++t) {
                    float *base = a.contextual + (uint64_t)s_ctx[t] * a.ld;
                    Row<CH> v;
                    load_row<CH>(v, base, q, nchunks, true);
                    if (grp == 0) scatter_add<CH, kWriteBack>(base, q, nchunks, invC, g, v);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                }
            } else if (a.ctx_delta) {
                // batch form: contextual is read-only in this launch, the gradients are summed
                // (exactly: hardware atomics) in the delta table
                Row<CH> gc = g;
                if constexpr (WM != kAtomic) to_contig_layout<CH>(gc, g, s_tr, grp, q, a.ld);
                for (uint32_t r0 = 0; r0 < n_ctx; r0 += 4) {
                    const uint32_t rank = r0 + grp;
                    if (rank < n_ctx)
                        scatter_add<CH, kAtomic>(a.ctx_delta + (uint64_t)s_ctx[rank] * a.ld, q,
                                                 nchunks, invC, gc, gc);
                }
            } else {
                // context node ids of this centre were staged behind the sample list
                for (uint32_t r0 = 0; r0 < n_ctx; r0 += 4) {
                    const RoundIds ids(s_ctx, r0, n_ctx);
                    const uint32_t row = ids.row_of(grp);
                    const bool valid = row != kSentinel;
                    const int my_pass = ids.pass_of(grp);
                    float *base = a.contextual + (uint64_t)(valid ? row : 0) * a.ld;
                    for (int pass = 0; pass <= ids.last_pass; ++pass) {
                        const bool mine = valid && my_pass == pass;
                        Row<CH> v;
                        if constexpr (WM != kAtomic) load_row<CH>(v, base, q, nchunks, mine);
                        if (mine) scatter_add<CH, WM>(base, q, nchunks, invC, g, v);
                    }
                }
            }
            pairs += n_ctx;
            ++centres;
        }
        wave_sync();
    }
    // one atomic per wave: a per-walk atomic on a single counter serialises at ~12 ns each
    if (a.counters && lane == 0 && pairs) {
        atomicAdd(&a.counters[0], pairs);
        atomicAdd(&a.counters[2], centres);
    }
}

// The same window cache with O(1) bookkeeping (CBOW: a centre is only ~40 row transfers, so the
// directory searches of CtxCache were a large part of its time).  Window position p owns the
// directory entry pos_slot[p % slots] = the slot that holds its row (several positions naming the
// same node share a slot through the reference count), or kNoSlot when the row is not cached (hot
// node).  Finding a node is one parallel compare + ballot instead of a loop over the slots.
constexpr uint32_t kNoSlot = 0xFFu;
// lookup / free_slot are one ballot over the 64 lanes: the host only takes this kernel when
// slots = 2 * window + 1 <= kWinCacheMaxSlots (larger windows run the uncached kernel)
constexpr uint32_t kWinCacheMaxSlots = 64;
static_assert(kWinCacheMaxSlots <= 64 && kWinCacheMaxSlots < kNoSlot, "one ballot, one byte");

struct WinCache {
    float *rows;         // [slots][ld]
    uint32_t *node;      // [slots] node held by the slot
    uint32_t *ref;       // [slots] window positions naming it (0 = free)
    uint32_t *pos_slot;  // [slots] indexed by position % slots
    uint32_t slots, ld;

    __device__ __forceinline__ int lookup(uint32_t v, int lane) const {
        const bool m = (uint32_t)lane < slots && ref[lane] != 0 && node[lane] == v;
        const unsigned long long b = __ballot(m);
        return b ? __ffsll((long long)b) - 1 : -1;
    }
    __device__ __forceinline__ int free_slot(uint32_t prefer, int lane) const {
        if (ref[prefer] == 0) return (int)prefer;
        const unsigned long long b = __ballot((uint32_t)lane < slots && ref[lane] == 0);
        return b ? __ffsll((long long)b) - 1 : -1;
    }
};

template <int NC>
__device__ __forceinline__ void win_prefetch_issue(const TrainArgs &a, const WinCache &c,
                                                   const float *table, uint32_t v, int lane,
                                                   RowPrefetch<NC> &pf) {
    pf.node = v;
    pf.valid = c.lookup(v, lane) < 0;
    if (!pf.valid) return;
    pf.deg = a.cache_max_degree != 0xFFFFFFFFu ? a.g.row_ptr[v + 1] - a.g.row_ptr[v] : 0;
    const float *src = table + (uint64_t)v * c.ld;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
        const uint32_t ci = lane + 64 * cc;
        pf.chunk[cc] = ci < (c.ld >> 2) ? *reinterpret_cast<const float4 *>(src + ci * 4)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// position p (node v) enters the window; pf: registers fetched one centre ahead, if for this node
template <int NC>
__device__ __forceinline__ void win_insert(const TrainArgs &a, WinCache &c, const float *table,
                                           uint32_t p, uint32_t v, int lane,
                                           const RowPrefetch<NC> *pf) {
    const uint32_t entry = p % c.slots;
    const int hit = c.lookup(v, lane);
    if (hit >= 0) {
        if (lane == 0) {
            c.ref[hit] += 1;
            c.pos_slot[entry] = (uint32_t)hit;
        }
        wave_sync();
        return;
    }
    const bool have = pf != nullptr && pf->valid && pf->node == v;
    if (a.cache_max_degree != 0xFFFFFFFFu) {
        const uint64_t deg = have ? pf->deg : a.g.row_ptr[v + 1] - a.g.row_ptr[v];
        if (deg >= a.cache_max_degree) {  // hot node: many waves would hold private copies
            if (lane == 0) c.pos_slot[entry] = kNoSlot;
            wave_sync();
            return;
        }
    }
    const int f = c.free_slot(entry, lane);
    if (f < 0) {  // cannot happen: at most `slots` positions are live
        if (lane == 0) c.pos_slot[entry] = kNoSlot;
        wave_sync();
        return;
    }
    float *dst = c.rows + (uint32_t)f * c.ld;
    if (have) {
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            const uint32_t ci = lane + 64 * cc;
            if (ci < (c.ld >> 2)) *reinterpret_cast<float4 *>(dst + ci * 4) = pf->chunk[cc];
        }
    } else {
        const float *src = table + (uint64_t)v * c.ld;
        for (uint32_t ci = lane; ci < (c.ld >> 2); ci += 64)
            *reinterpret_cast<float4 *>(dst + ci * 4) =
                *reinterpret_cast<const float4 *>(src + ci * 4);
    }
    if (lane == 0) {
        c.node[f] = v;
        c.ref[f] = 1;
        c.pos_slot[entry] = (uint32_t)f;
    }
    wave_sync();
}

template <int WM>
__device__ __forceinline__ void win_write_back(const WinCache &c, float *table, uint32_t slot,
                                               int lane) {
    float *dst = table + (uint64_t)c.node[slot] * c.ld;
    const float *src = c.rows + slot * c.ld;
    for (uint32_t ci = lane; ci < (c.ld >> 2); ci += 64) {
        const float4 x = *reinterpret_cast<const float4 *>(src + ci * 4);
        if constexpr (WM == kWriteThrough)
            store_sc1(dst + ci * 4, x);
        else
            *reinterpret_cast<float4 *>(dst + ci * 4) = x;
    }
}

// position p leaves the window
template <int WM>
__device__ __forceinline__ void win_retire(WinCache &c, float *table, uint32_t p, int lane) {
    const uint32_t h = c.pos_slot[p % c.slots];
    if (h == kNoSlot) return;
    if (c.ref[h] == 1) win_write_back<WM>(c, table, h, lane);
    wave_sync();
    if (lane == 0) c.ref[h] -= 1;
    wave_sync();
}

// CBOW with the same LDS context cache: the window's contextual rows are what CBOW reads for the
// mean AND read-modify-writes for the gradient (30 of the 52 row transfers of a centre); with the
// cache they cost one read and one write-back per walk position.  Centre + negatives live in the
// central table, which the cache never holds, so only the context list needs directory lookups.
#ifndef GN2V_CBOW_MIN_BLOCKS
#define GN2V_CBOW_MIN_BLOCKS 1  // occupancy experiments: 4 caps the kernel at 128 VGPRs
#endif
template <int CH, int WM>
__global__ __launch_bounds__(kTrainBlock, GN2V_CBOW_MIN_BLOCKS) void cbow_cached_kernel(TrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, q = lane & 15;
    const uint32_t w = a.window, k = a.k;
    const uint32_t slots = 2 * w + 1;
    const uint32_t per_wave =
        (slots * a.ld + a.L + 2 * a.max_samples + 2 * w + 3 * slots + 3) & ~3u;
    uint32_t *base_w = smem + wave * per_wave;
    WinCache cache;
    cache.rows = reinterpret_cast<float *>(base_w);
    uint32_t *s_walk = base_w + slots * a.ld;
    uint32_t *s_rows = s_walk + a.L;
    float *s_lab = reinterpret_cast<float *>(s_rows + a.max_samples);
    uint32_t *s_ctx = s_rows + 2 * a.max_samples;
    cache.node = s_ctx + 2 * w;
    cache.ref = cache.node + slots;
    cache.pos_slot = cache.ref + slots;
    cache.slots = slots;
    cache.ld = a.ld;
    const uint32_t nchunks = a.ld >> 2;
    const uint64_t per_walk_neg = (uint64_t)a.L * k;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint64_t wave_stride = (uint64_t)gridDim.x * waves_per_block;
    unsigned long long pairs = 0, centres = 0;

    for (uint64_t b = (uint64_t)blockIdx.x * waves_per_block + wave; b < a.n_walks;
         b += wave_stride) {
        const uint32_t Le = stage_walk(a, b, s_walk, s_walk, lane);
        const uint64_t wkey = draw(a.ekey, a.first_walk + b);
        const uint64_t nkey = wkey ^ kTagNeg;
        const uint32_t *ov = a.neg_override ? a.neg_override + b * per_walk_neg : nullptr;
        if ((uint32_t)lane < slots) {
            cache.ref[lane] = 0;
            cache.pos_slot[lane] = kNoSlot;
        }
        wave_sync();
        // float4 chunks per lane of a whole row spread over the 64 lanes: ld / 256, rounded up
        // (CH <= 4: ld <= 256 floats = 64 chunks; CH = 16: ld <= 1024 = 256 chunks)
        constexpr int kPfChunks = CH > 8 ? 4 : CH > 4 ? 2 : 1;
        for (uint32_t p = 0; p < Le && p <= w; ++p)
            win_insert<kPfChunks>(a, cache, a.contextual, p, s_walk[p], lane, nullptr);
        // one centre ahead, in registers: the negatives (their ids cost a random col_idx read) and
        // the contextual row that enters the window -- the centre's latency chain shrinks from
        // ~6 dependent memory round trips to the one of its output rows
        RowPrefetch<kPfChunks> pf;
        pf.valid = false;
        pf.node = kSentinel;
        pf.deg = 0;
#pragma unroll
        for (int cc = 0; cc < kPfChunks; ++cc) pf.chunk[cc] = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t pre_neg = kSentinel, pre_for = kSentinel;
        const bool neg_in_lanes = k < 64;
        auto fetch_negative = [&](uint32_t ii) -> uint32_t {
            if (lane < 1 || (uint32_t)lane > k) return kSentinel;
            const uint64_t qi = (uint64_t)ii * k + (uint32_t)(lane - 1);
            return ov ? ov[qi] : draw_negative(a, nkey, qi);
        };

        for (uint32_t i = 0; i < Le; ++i) {
            if (i >= w + 1) win_retire<WM>(cache, a.contextual, i - w - 1, lane);
            if (i >= 1 && i + w < Le)
                win_insert<kPfChunks>(a, cache, a.contextual, i + w, s_walk[i + w], lane, &pf);
            pf.valid = false;
            if (i + 1 + w < Le)
                win_prefetch_issue(a, cache, a.contextual, s_walk[i + 1 + w], lane, pf);

            const uint32_t c = s_walk[i];
            if (!keep_centre(a, wkey, i, c)) continue;
            const float lrc = centre_lr(a, c);
            const Window win(i, Le, w, a.min_dist);
            const uint32_t n_ctx = win.n_ctx;
            if (n_ctx == 0) continue;
            const float invC = 1.0f / (float)n_ctx;

            wave_sync();
            if (neg_in_lanes) {
                const uint32_t raw = pre_for == i ? pre_neg : fetch_negative(i);
                if (i + 1 < Le) {
                    pre_neg = fetch_negative(i + 1);
                    pre_for = i + 1;
                }
                if ((uint32_t)lane <= k) {
                    s_rows[lane] = lane == 0 ? c : (raw == c ? kSentinel : raw);
                    s_lab[lane] = lane == 0 ? 1.f : 0.f;
                }
            } else {
                for (uint32_t t = lane; t <= k; t += 64) {
                    uint32_t row = c;
                    float lab = 1.f;
                    if (t != 0) {
                        const uint64_t qi = (uint64_t)i * k + (t - 1);
                        row = ov ? ov[qi] : draw_negative(a, nkey, qi);
                        lab = 0.f;
                        if (row == c) row = kSentinel;
                    }
                    s_rows[t] = row;
                    s_lab[t] = lab;
                }
            }
            for (uint32_t t = lane; t < n_ctx; t += 64) {
                const uint32_t j = win.position(t);
                const uint32_t h = cache.pos_slot[j % slots];
                s_ctx[t] = h != kNoSlot ? (kCacheBit | h) : s_walk[j];
            }
            wave_sync();

            Row<CH> h, g;
            zero_row<CH>(h);
            zero_row<CH>(g);
            for (uint32_t r0 = 0; r0 < n_ctx; r0 += 4) {
                const uint32_t rank = r0 + grp;
                const bool in = rank < n_ctx;
                const uint32_t row = in ? s_ctx[rank] : 0u;
                const bool cached = in && (row & kCacheBit);
                Row<CH> v;
                load_row<CH>(v, a.contextual + (uint64_t)(cached ? 0u : row) * a.ld, q, nchunks,
                             in && !cached);
                if (cached) {
                    const float *lbase = cache.rows + (row & 0xFFFFu) * a.ld;
#pragma unroll
                    for (int cc = 0; cc < CH; ++cc) {
                        const uint32_t ci = cc * 16 + q;
                        if (ci < nchunks) v.c[cc] = *reinterpret_cast<const float4 *>(lbase + ci * 4);
                    }
                }
                axpy<CH>(h, 1.0f, v);
            }
            reduce_groups<CH>(h);
#pragma unroll
            for (int cc = 0; cc < CH; ++cc) {
                h.c[cc].x *= invC;
                h.c[cc].y *= invC;
                h.c[cc].z *= invC;
                h.c[cc].w *= invC;
            }

            // all output rows in flight at once unless a row is named twice (then the serialising
            // path keeps the oracle's order)
            bool repeated = false;
            if (k + 1 <= kFlightSamples && (uint32_t)lane <= k) {
                const uint32_t mine = s_rows[lane];
                for (int j = 0; j < lane; ++j) repeated |= mine != kSentinel && s_rows[j] == mine;
            }
            if (k + 1 <= kFlightSamples && __ballot(repeated) == 0)
                score_samples_flight<CH, WM>(a, a.central, h, h, g, s_rows, s_lab, k + 1, lrc, grp,
                                             q);
            else
                score_samples<CH, WM, false>(a, a.central, h, h, g, s_rows, s_lab, k + 1, lrc, grp,
                                             q);
            reduce_groups<CH>(g);

            for (uint32_t r0 = 0; r0 < n_ctx; r0 += 4) {
                const RoundIds ids(s_ctx, r0, n_ctx);
                const uint32_t row = ids.row_of(grp);
                const bool valid = row != kSentinel;
                const bool cached = valid && (row & kCacheBit);
                const int my_pass = ids.pass_of(grp);
                float *gbase = a.contextual + (uint64_t)((valid && !cached) ? row : 0) * a.ld;
                float *lbase = cache.rows + (cached ? (row & 0xFFFFu) : 0u) * a.ld;
                for (int pass = 0; pass <= ids.last_pass; ++pass) {
                    const bool mine = valid && my_pass == pass;
                    if (mine && cached) {
#pragma unroll
                        for (int cc = 0; cc < CH; ++cc) {
                            const uint32_t ci = cc * 16 + q;
                            if (ci < nchunks) {
                                float4 o = *reinterpret_cast<const float4 *>(lbase + ci * 4);
                                o.x += invC * g.c[cc].x;
                                o.y += invC * g.c[cc].y;
                                o.z += invC * g.c[cc].z;
                                o.w += invC * g.c[cc].w;
                                *reinterpret_cast<float4 *>(lbase + ci * 4) = o;
                            }
                        }
                    } else {
                        Row<CH> v;
                        load_row<CH>(v, gbase, q, nchunks, mine);
                        if (mine) scatter_add<CH, WM>(gbase, q, nchunks, invC, g, v);
                    }
                }
            }
            pairs += n_ctx;
            ++centres;
        }
        wave_sync();
        for (uint32_t s = 0; s < slots; ++s)
            if (cache.ref[s] != 0) win_write_back<WM>(cache, a.contextual, s, lane);
        wave_sync();
    }
    if (a.counters && lane == 0 && pairs) {
        atomicAdd(&a.counters[0], pairs);
        atomicAdd(&a.counters[2], centres);
    }
}

// Traffic calibration: every listed row is read and written exactly once with the same access
// shape as the training kernels (16 lanes x float4 per row, same store flavour), so the HBM
// bytes of a launch are known exactly (n * ld * 4 read + written, + 4 B of id per row).  Used to
// calibrate the FETCH_SIZE / WRITE_SIZE counters for this access pattern (profiles/README.md).
// experiment: f32 atomics with lane-contiguous addresses (lane q -> float 64*cc + 16*e + q)
// instead of the float4-shaped stride (float 64*cc + 4*q + e)
template <int CH>
__global__ __launch_bounds__(kTrainBlock) void touch_rows_atomic_contig_kernel(
    float *table, uint32_t ld, const uint32_t *ids, uint64_t n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, q = lane & 15;
    const uint64_t stride = (uint64_t)gridDim.x * (kTrainBlock / 64) * 4;
    for (uint64_t i = ((uint64_t)blockIdx.x * (kTrainBlock / 64) + wave) * 4 + grp; i < n;
         i += stride) {
        float *base = table + (uint64_t)ids[i] * ld;
#pragma unroll
        for (int cc = 0; cc < CH; ++cc)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t f = cc * 64 + e * 16 + q;
                if (f < ld) unsafeAtomicAdd(base + f, 1.0f);
            }
    }
}

// experiment: two rounds of rows in flight per wave (does more memory-level parallelism raise the
// random-row read-modify-write bandwidth?)
template <int CH, int WM>
__global__ __launch_bounds__(kTrainBlock) void touch_rows2_kernel(float *table, uint32_t ld,
                                                                  const uint32_t *ids, uint64_t n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, q = lane & 15;
    const uint32_t nchunks = ld >> 2;
    const uint64_t stride = (uint64_t)gridDim.x * (kTrainBlock / 64) * 8;
    Row<CH> ones;
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) ones.c[cc] = make_float4(1.f, 1.f, 1.f, 1.f);
    for (uint64_t i = ((uint64_t)blockIdx.x * (kTrainBlock / 64) + wave) * 8 + grp; i < n;
         i += stride) {
        const bool two = i + 4 < n;
        float *base0 = table + (uint64_t)ids[i] * ld;
        float *base1 = table + (uint64_t)ids[two ? i + 4 : i] * ld;
        Row<CH> v0, v1;
        load_row<CH>(v0, base0, q, nchunks, true);
        load_row<CH>(v1, base1, q, nchunks, two);
        scatter_add<CH, WM>(base0, q, nchunks, 1.0f, ones, v0);
        if (two) scatter_add<CH, WM>(base1, q, nchunks, 1.0f, ones, v1);
    }
}

template <int CH, int WM>
__global__ __launch_bounds__(kTrainBlock) void touch_rows_kernel(float *table, uint32_t ld,
                                                                 const uint32_t *ids, uint64_t n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, q = lane & 15;
    const uint32_t nchunks = ld >> 2;
    const uint64_t stride = (uint64_t)gridDim.x * (kTrainBlock / 64) * 4;
    Row<CH> ones;
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) ones.c[cc] = make_float4(1.f, 1.f, 1.f, 1.f);
    for (uint64_t i = ((uint64_t)blockIdx.x * (kTrainBlock / 64) + wave) * 4 + grp; i < n;
         i += stride) {
        float *base = table + (uint64_t)ids[i] * ld;
        Row<CH> v;
        load_row<CH>(v, base, q, nchunks, true);
        scatter_add<CH, WM>(base, q, nchunks, 1.0f, ones, v);
    }
}

}  // namespace gn2v

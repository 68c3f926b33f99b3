// CBOW with a LAZY window: the input-side gradients of the cached contextual rows are never
// scattered.  (model statement: embedders/tensorflow_embedders/cbow.py:26-60 -- the mean of the
// window's input rows scored against the centre and k negatives; call site node2vec.py:99.)
//
// cbow_cached_kernel keeps the rows of the window positions [i - w, i + w] in LDS and, per
// centre, reads the <= 2w context rows for the mean and read-modify-writes the same rows with
// g / C: ~20 LDS row operations plus their bookkeeping per centre.  rocprofv3 (round 3,
// SQ_INSTS_* / SQ_ACTIVE_INST_ANY) showed the kernel issue bound, not memory bound: 1 680
// instructions per centre and wave, every wave active a quarter of the time with four waves per
// SIMD.  Here the window is kept in a form in which a centre costs O(1) row operations:
//
//   delta_i = g_i / C_i                       the step every context row of centre i receives
//   D_t     = sum_{i <= t} delta_i            one running row per walk
//   slot of node v:  b_v, m_v   with   x_v(t) = b_v + m_v * D_t
//
// where m_v = number of live window positions naming v.  A position p is live for the centres
// p - w .. p + w and receives delta_i for each of them but i = p:
//   insert p (before centre e):  b_v = x_v(HBM) - D_{e-1}  (new slot), or  b_v -= D_{e-1}, m_v += 1
//   after centre i:              b_{v(i)} -= delta_i       (the centre's own position gets none)
//   retire p (after centre p+w): b_v += D_{p+w}, m_v -= 1; m_v == 0: x_v = b_v goes back to HBM
// and the window sum of centre i is  sum_{j in J(i)} (b_{v(j)} + m_{v(j)} D_{i-1}).  While no
// node occurs twice in the window (all m = 1: the rule), the sum of the b's is slid along:
//   B_{i+1} = B_i - b_{v(i-w)} + b_{v(i+w+1)} + b_{v(i)} - b_{v(i+1)},   S_i = B_i + Mc_i D_{i-1}
// (Mc = cached positions in J(i)); a revisited node switches to the direct sum until the window
// is free of duplicates again.  Rows of high-degree nodes must not live in private copies (many
// waves would hold one at once and the last write-back would win): their slots hold the PENDING
// STEP only,
//   x_v(t) = x_v(HBM, now) + b_v + m_v * D_t,   b_v = -D_{e-1} at entry,
// the row itself is read from HBM for every window sum (so it is always everybody's latest) and
// the pending step reaches it with f32 atomics when the position retires -- one row of atomics
// per position instead of a read-modify-write per centre, and nobody's update is lost.  Exact in
// exact arithmetic; in f32 the sliding sum drifts by ~1e-6 over a walk (tests: 1e-5 against the
// oracle, return-heavy walks included).
// min_dist == 1 only (Walklets scales keep cbow_cached_kernel).
#pragma once
#include "train_kernels.h"

// Four workgroups per CU = four waves per SIMD: the kernel is issue bound and a wave of occupancy
// is worth more than the four registers the cap spills (round 3, same box: 131 VGPRs / 3 waves
// 2.90e8 centres/s, 127 VGPRs + 4 spilled / 4 waves 3.62e8; cbow_cached_kernel 3.21e8).
#ifndef GN2V_CBOW_LAZY_MIN_BLOCKS
#define GN2V_CBOW_LAZY_MIN_BLOCKS 4
#endif
// Wider rows (CH = 4 / 8 / 16: d up to 256 / 512 / 1024) hold 4 * CH registers per row vector:
// under the 128-register cap they spilled 184 / 570 / 1 500 B per lane.  Their caps follow the
// rows and the LDS the window takes (64 KB budget: three workgroups per CU up to ~ 230 floats per
// row, two at 256).  Same box, BA 1 M, centres/s at d = 160 / 200 / 256: cap 4 1.75 / 1.29 /
// 1.14e8, cap 3 1.93 / 1.65 / 1.56e8, cap 2 1.61 / 1.38 / 1.61e8 (uncached kernel: 1.75 / 1.39 /
// 1.31e8) -> 3, and 2 for the rows of exactly 256 floats (FULL).  CH = 8 / 16 never run today
// (their windows exceed the LDS budget; the uncached kernel serves d > 256).
#ifndef GN2V_CBOW_LAZY_MIN_BLOCKS_CH4
#define GN2V_CBOW_LAZY_MIN_BLOCKS_CH4 3
#endif
#ifndef GN2V_CBOW_LAZY_MIN_BLOCKS_CH4_FULL
#define GN2V_CBOW_LAZY_MIN_BLOCKS_CH4_FULL 2
#endif
#ifndef GN2V_CBOW_LAZY_MIN_BLOCKS_CH8
#define GN2V_CBOW_LAZY_MIN_BLOCKS_CH8 2
#endif
#ifndef GN2V_CBOW_LAZY_MIN_BLOCKS_CH16
#define GN2V_CBOW_LAZY_MIN_BLOCKS_CH16 1
#endif

namespace gn2v {

// workgroups of 256 threads per CU the register cap of the instantiation for row stride ld
// leaves room for (the __launch_bounds__ below)
constexpr uint32_t lazy_min_blocks_ch(int ch, bool full) {
    return ch <= 2   ? GN2V_CBOW_LAZY_MIN_BLOCKS
           : ch == 4 ? (full ? GN2V_CBOW_LAZY_MIN_BLOCKS_CH4_FULL : GN2V_CBOW_LAZY_MIN_BLOCKS_CH4)
           : ch == 8 ? GN2V_CBOW_LAZY_MIN_BLOCKS_CH8
                     : GN2V_CBOW_LAZY_MIN_BLOCKS_CH16;
}
inline uint32_t lazy_min_blocks(uint32_t ld) {
    const uint32_t nchunks = ld / 4;
    const int ch = nchunks <= 16 ? 1 : nchunks <= 32 ? 2 : nchunks <= 64 ? 4 : nchunks <= 128 ? 8 : 16;
    return lazy_min_blocks_ch(ch, ld == (uint32_t)ch * 64);
}
// the lazy window runs when a CU's LDS holds at least this many of its waves (measured down to
// three -- rows of 1 024 floats at w = 5: 0.62 of the roofline against the uncached kernel's
// 0.52)
constexpr uint32_t kLazyMinWaves = 3;

template <int CH>
__device__ __forceinline__ void lds_load_row(Row<CH> &r, const float *base, int q,
                                             uint32_t nchunks) {
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const uint32_t ci = cc * 16 + q;
        r.c[cc] = ci < nchunks ? *reinterpret_cast<const float4 *>(base + ci * 4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int CH>
__device__ __forceinline__ void lds_store_row(float *base, const Row<CH> &r, int q,
                                              uint32_t nchunks) {
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const uint32_t ci = cc * 16 + q;
        if (ci < nchunks) *reinterpret_cast<float4 *>(base + ci * 4) = r.c[cc];
    }
}

// acc += s * x for float4 rows (s may be +-1)
template <int CH>
__device__ __forceinline__ void row_axpy(Row<CH> &acc, float s, const Row<CH> &x) {
    axpy<CH>(acc, s, x);
}

// FULL: the row stride is exactly CH * 64 floats (d = 128 -> CH = 2): ld becomes a compile-time
// constant and the per-chunk bounds predicates fold away (fewer instructions, fewer SGPR masks).
template <int CH, int WM, bool FULL = false>
__global__ __launch_bounds__(kTrainBlock, lazy_min_blocks_ch(CH, FULL)) void
cbow_lazy_kernel(TrainArgs a) {
    if constexpr (FULL) a.ld = CH * 64;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, q = lane & 15;
    const uint32_t w = a.window, k = a.k;
    const uint32_t slots = 2 * w + 1;
    const uint32_t per_wave =
        ((slots + 2) * a.ld + a.L + 2 * a.max_samples + 2 * w + 3 * slots + 3) & ~3u;
    uint32_t *base_w = smem + wave * per_wave;
    float *rows = reinterpret_cast<float *>(base_w);             // [slots][ld]: the b's
    float *s_D = rows + slots * a.ld;                            // [ld] running sum of deltas
    float *s_B = s_D + a.ld;                                     // [ld] sliding sum of the b's
    uint32_t *s_walk = base_w + (slots + 2) * a.ld;
    uint32_t *s_rows = s_walk + a.L;
    float *s_lab = reinterpret_cast<float *>(s_rows + a.max_samples);
    uint32_t *s_ctx = s_rows + 2 * a.max_samples;                // uncached context nodes
    uint32_t *s_node = s_ctx + 2 * w;
    uint32_t *s_ref = s_node + slots;
    uint32_t *s_pos = s_ref + slots;                             // position % slots -> slot
    const uint32_t nchunks = a.ld >> 2;
    const uint64_t per_walk_neg = (uint64_t)a.L * k;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint64_t wave_stride = (uint64_t)gridDim.x * waves_per_block;
    unsigned long long pairs = 0, centres = 0;

    // s_node: node id of the slot; bit 31 = the slot holds the pending step of a row that stays
    // in HBM (node ids are below 2^30 here, launch_train)
    constexpr uint32_t kPending = 0x80000000u;
    auto lookup = [&](uint32_t v) -> int {
        const bool m =
            (uint32_t)lane < slots && s_ref[lane] != 0 && (s_node[lane] & ~kPending) == v;
        const unsigned long long b = __ballot(m);
        return b ? __ffsll((long long)b) - 1 : -1;
    };

    for (uint64_t wb = (uint64_t)blockIdx.x * waves_per_block + wave; wb < a.n_walks;
         wb += wave_stride) {
        const uint32_t Le = stage_walk(a, wb, s_walk, s_walk, lane);
        const uint64_t wkey = draw(a.ekey, a.first_walk + wb);
        const uint64_t nkey = wkey ^ kTagNeg;
        const uint32_t *ov = a.neg_override ? a.neg_override + wb * per_walk_neg : nullptr;
        if ((uint32_t)lane < slots) {
            s_ref[lane] = 0;
            s_pos[lane] = kNoSlot;
        }
        {
            Row<CH> z;
            zero_row<CH>(z);
            if (grp == 0) lds_store_row<CH>(s_D, z, q, nchunks);
        }
        wave_sync();
        uint32_t n_dups = 0;     // slots named by more than one live position
        bool b_valid = false;    // s_B holds the sum of the b's of the cached positions of J(i)

        // position p (node v) enters the window; D = the running sum before the first centre
        // that sees it.  Returns the slot (or -1: not cached) and whether it created a duplicate.
        auto insert = [&](uint32_t p, const Row<CH> &D) -> int {
            const uint32_t v = s_walk[p], entry = p % slots;
            const int hit = lookup(v);
            if (hit >= 0) {  // the walk revisits a node inside the window
                float *row = rows + (uint32_t)hit * a.ld;
                Row<CH> b;
                lds_load_row<CH>(b, row, q, nchunks);
                const uint32_t m_old = s_ref[hit];
                row_axpy<CH>(b, -1.0f, D);
                wave_sync();
                if (grp == 0) lds_store_row<CH>(row, b, q, nchunks);
                if (lane == 0) {
                    s_ref[hit] = m_old + 1;
                    s_pos[entry] = (uint32_t)hit;
                }
                if (m_old == 1) ++n_dups;
                b_valid = false;
                wave_sync();
                return hit;
            }
            bool pending = false;  // hot node: the row stays in HBM, the slot holds its step
            if (a.cache_max_degree != 0xFFFFFFFFu) {
                const uint64_t deg = a.g.row_ptr[v + 1] - a.g.row_ptr[v];
                pending = deg >= a.cache_max_degree;
            }
            int f = -1;
            {
                const unsigned long long fb = __ballot((uint32_t)lane < slots && s_ref[lane] == 0);
                if (s_ref[entry] == 0)
                    f = (int)entry;
                else if (fb)
                    f = __ffsll((long long)fb) - 1;
            }
            if (f < 0) {  // cannot happen: at most `slots` positions are live
                if (lane == 0) s_pos[entry] = kNoSlot;
                wave_sync();
                return -1;
            }
            Row<CH> b;
            load_row<CH>(b, a.contextual + (uint64_t)v * a.ld, q, nchunks, !pending);
            row_axpy<CH>(b, -1.0f, D);
            if (grp == 0) lds_store_row<CH>(rows + (uint32_t)f * a.ld, b, q, nchunks);
            if (lane == 0) {
                s_node[f] = pending ? v | kPending : v;
                s_ref[f] = 1;
                s_pos[entry] = (uint32_t)f;
            }
            wave_sync();
            return f;
        };

        {
            Row<CH> D0;
            zero_row<CH>(D0);
            for (uint32_t p = 0; p < Le && p <= w; ++p) insert(p, D0);
        }

        for (uint32_t i = 0; i < Le; ++i) {
            const uint32_t c = s_walk[i];
            const Window win(i, Le, w, 1);
            const uint32_t n_ctx = win.n_ctx;
            const bool train = n_ctx != 0 && keep_centre(a, wkey, i, c);

            if (train) {
                const float lrc = centre_lr(a, c);
                const float invC = 1.0f / (float)n_ctx;
                wave_sync();
                // the k + 1 output rows: the centre and its negatives, in the central table
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
                // which context positions are cached; the others' node ids, compacted
                bool cached = false, hub = false;
                uint32_t node_j = 0;
                if ((uint32_t)lane < n_ctx) {
                    const uint32_t j = win.position(lane);
                    const uint32_t sl = s_pos[j % slots];
                    cached = sl != kNoSlot;
                    hub = !cached || (s_node[sl] & kPending);  // the row is read from HBM
                    node_j = s_walk[j];
                }
                const unsigned long long hub_mask = __ballot(hub);
                const uint32_t n_hub = (uint32_t)__popcll(hub_mask);
                const uint32_t n_cached = (uint32_t)__popcll(__ballot(cached));
                if (hub) s_ctx[__popcll(hub_mask & ((1ULL << lane) - 1))] = node_j;
                wave_sync();

                // S = sum of the current values of the context rows
                Row<CH> S, D;
                lds_load_row<CH>(D, s_D, q, nchunks);  // D_{i-1}
                if (b_valid) {
                    lds_load_row<CH>(S, s_B, q, nchunks);
                    row_axpy<CH>(S, (float)n_cached, D);
                } else {
                    // direct: every cached position contributes b + m * D (a revisited node
                    // once per position)
                    Row<CH> part;
                    zero_row<CH>(part);
                    float m_sum = 0.f;
                    for (uint32_t r0 = 0; r0 < n_ctx; r0 += 4) {
                        const uint32_t rank = r0 + grp;
                        if (rank < n_ctx) {
                            const uint32_t h = s_pos[win.position(rank) % slots];
                            if (h != kNoSlot) {
                                Row<CH> b;
                                lds_load_row<CH>(b, rows + h * a.ld, q, nchunks);
                                row_axpy<CH>(part, 1.0f, b);
                                m_sum += (float)s_ref[h];
                            }
                        }
                    }
                    reduce_groups<CH>(part);
                    m_sum += __shfl_xor(m_sum, 16);
                    m_sum += __shfl_xor(m_sum, 32);
                    if (n_dups == 0) {  // the window is free of duplicates: slide from here on
                        wave_sync();
                        if (grp == 0) lds_store_row<CH>(s_B, part, q, nchunks);
                        b_valid = true;
                    }
                    S = part;
                    row_axpy<CH>(S, m_sum, D);
                }
                if (n_hub) {  // the rows that stay in HBM: eight in flight per wave
                    Row<CH> hs;
                    zero_row<CH>(hs);
                    for (uint32_t r0 = 0; r0 < n_hub; r0 += 8) {
                        const uint32_t ra = r0 + grp, rb = r0 + 4 + grp;
                        Row<CH> va, vb;
                        load_row<CH>(va, a.contextual + (uint64_t)(ra < n_hub ? s_ctx[ra] : 0) * a.ld,
                                     q, nchunks, ra < n_hub);
                        load_row<CH>(vb, a.contextual + (uint64_t)(rb < n_hub ? s_ctx[rb] : 0) * a.ld,
                                     q, nchunks, rb < n_hub);
                        row_axpy<CH>(hs, 1.0f, va);
                        row_axpy<CH>(hs, 1.0f, vb);
                    }
                    reduce_groups<CH>(hs);
                    row_axpy<CH>(S, 1.0f, hs);
                }
                Row<CH> h, g, delta;
                zero_row<CH>(g);
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) {
                    h.c[cc].x = S.c[cc].x * invC;
                    h.c[cc].y = S.c[cc].y * invC;
                    h.c[cc].z = S.c[cc].z * invC;
                    h.c[cc].w = S.c[cc].w * invC;
                }
                // all output rows in flight at once unless there are more than twelve or a row is
                // named twice (then the serialising path keeps the oracle's order)
                bool repeated = false;
                if (k + 1 <= kFlightSamples && (uint32_t)lane <= k) {
                    const uint32_t mine = s_rows[lane];
                    for (int j = 0; j < lane; ++j) repeated |= mine != kSentinel && s_rows[j] == mine;
                }
                if (k + 1 <= kFlightSamples && __ballot(repeated) == 0)
                    score_samples_flight<CH, WM>(a, a.central, h, h, g, s_rows, s_lab, k + 1, lrc,
                                                 grp, q);
                else
                    score_samples<CH, WM, false>(a, a.central, h, h, g, s_rows, s_lab, k + 1, lrc,
                                                 grp, q);
                reduce_groups<CH>(g);
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) {
                    delta.c[cc].x = g.c[cc].x * invC;
                    delta.c[cc].y = g.c[cc].y * invC;
                    delta.c[cc].z = g.c[cc].z * invC;
                    delta.c[cc].w = g.c[cc].w * invC;
                }
                // (a position without a slot cannot happen -- at most `slots` are live -- so
                // every context row receives delta_i through its slot, the rows that stay in HBM
                // too: their slots hold the pending step)
                pairs += n_ctx;
                ++centres;
                // (a) the centre's own position receives no step: take it back
                wave_sync();
                const uint32_t own = s_pos[i % slots];
                if (own != kNoSlot) {
                    float *row = rows + own * a.ld;
                    Row<CH> b;
                    lds_load_row<CH>(b, row, q, nchunks);
                    row_axpy<CH>(b, -1.0f, delta);
                    wave_sync();
                    if (grp == 0) lds_store_row<CH>(row, b, q, nchunks);
                }
                // (b) D_i = D_{i-1} + delta_i
                {
                    Row<CH> Dn;
                    lds_load_row<CH>(Dn, s_D, q, nchunks);
                    row_axpy<CH>(Dn, 1.0f, delta);
                    wave_sync();
                    if (grp == 0) lds_store_row<CH>(s_D, Dn, q, nchunks);
                }
            }

            // ---- advance the window from centre i to centre i + 1
            wave_sync();
            const uint32_t hc = s_pos[i % slots];  // the centre's own slot (or kNoSlot)
            Row<CH> B, D;
            lds_load_row<CH>(D, s_D, q, nchunks);  // D_i
            if (b_valid) lds_load_row<CH>(B, s_B, q, nchunks);
            // (c) position i - w leaves
            if (i >= w) {
                const uint32_t h = s_pos[(i - w) % slots];
                if (h != kNoSlot) {
                    float *row = rows + h * a.ld;
                    Row<CH> b;
                    lds_load_row<CH>(b, row, q, nchunks);
                    if (b_valid) row_axpy<CH>(B, -1.0f, b);
                    row_axpy<CH>(b, 1.0f, D);  // frozen: its share of the steps so far
                    const uint32_t m = s_ref[h];
                    wave_sync();
                    if (m == 1) {
                        const uint32_t node = s_node[h];
                        float *dst = a.contextual + (uint64_t)(node & ~kPending) * a.ld;
                        if (node & kPending) {
                            // b = the pending step: added to everybody's row, 256 B per instruction
                            if (grp == 0) lds_store_row<CH>(row, b, q, nchunks);
                            wave_sync();
                            for (uint32_t f = lane; f < a.ld; f += 64) unsafeAtomicAdd(dst + f, row[f]);
                        } else if (grp == 0) {
                            Row<CH> none;
                            zero_row<CH>(none);
                            // row = b: a plain (or write-through) store of the final value
                            scatter_add<CH, WM == kWriteBack ? kWriteBack : kWriteThrough>(
                                dst, q, nchunks, 1.0f, b, none);
                        }
                    } else {
                        if (grp == 0) lds_store_row<CH>(row, b, q, nchunks);
                        if (m == 2) --n_dups;
                    }
                    if (lane == 0) s_ref[h] = m - 1;
                    wave_sync();
                }
            }
            // (d) position i + 1 + w enters
            if (i + 1 + w < Le) {
                const uint32_t before = n_dups;
                const int f = insert(i + 1 + w, D);
                if (b_valid && f >= 0 && n_dups == before) {
                    Row<CH> b;
                    lds_load_row<CH>(b, rows + (uint32_t)f * a.ld, q, nchunks);
                    row_axpy<CH>(B, 1.0f, b);
                }
            }
            // (e) position i becomes a context, position i + 1 the centre
            if (b_valid) {
                if (hc != kNoSlot) {
                    Row<CH> b;
                    lds_load_row<CH>(b, rows + hc * a.ld, q, nchunks);
                    row_axpy<CH>(B, 1.0f, b);
                }
                if (i + 1 < Le) {
                    const uint32_t hn = s_pos[(i + 1) % slots];
                    if (hn != kNoSlot) {
                        Row<CH> b;
                        lds_load_row<CH>(b, rows + hn * a.ld, q, nchunks);
                        row_axpy<CH>(B, -1.0f, b);
                    }
                }
                wave_sync();
                if (grp == 0) lds_store_row<CH>(s_B, B, q, nchunks);
            }
            wave_sync();
        }
        // the walk is over: every live position is frozen at the final D and written back
        {
            Row<CH> D;
            lds_load_row<CH>(D, s_D, q, nchunks);
            for (uint32_t s = 0; s < slots; ++s) {
                const uint32_t m = s_ref[s];
                if (m == 0) continue;
                Row<CH> b, none;
                float *row = rows + s * a.ld;
                lds_load_row<CH>(b, row, q, nchunks);
                row_axpy<CH>(b, (float)m, D);
                zero_row<CH>(none);
                const uint32_t node = s_node[s];
                float *dst = a.contextual + (uint64_t)(node & ~kPending) * a.ld;
                if (node & kPending) {
                    wave_sync();
                    if (grp == 0) lds_store_row<CH>(row, b, q, nchunks);
                    wave_sync();
                    for (uint32_t f = lane; f < a.ld; f += 64) unsafeAtomicAdd(dst + f, row[f]);
                } else if (grp == 0) {
                    scatter_add<CH, WM == kWriteBack ? kWriteBack : kWriteThrough>(
                        dst, q, nchunks, 1.0f, b, none);
                }
            }
        }
        wave_sync();
    }
    if (a.counters && lane == 0 && pairs) {
        atomicAdd(&a.counters[0], pairs);
        atomicAdd(&a.counters[2], centres);
    }
}

}  // namespace gn2v

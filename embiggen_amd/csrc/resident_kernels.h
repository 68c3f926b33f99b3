// Resident cells, second form (round 5): the record loop of sgns_resident_kernel rewritten for
// what bounds it -- vector-instruction issue (round 4: 196 VALU + 63 SALU + 32 LDS instructions
// per pair and wave against ~60 of arithmetic, vector pipes 64 % busy) -- not bytes:
//   * the sample list of a record is staged as ONE 16-bit word per sample (row inside the cell,
//     bit 15 = positive) instead of a row word + a label word: a group fetches the four samples of
//     a step with one 8-byte LDS read, and the staging is a third of its former size -- room for
//     four transposition rows per wave and ~20 more rows per cell;
//   * a sample that must not be trained (a negative that fell on the context or the centre, the
//     padding of a list) names the cell's DUMMY row and gets the coefficient 0: no store is
//     predicated, the whole step is straight-line code;
//   * the cell's alias table and node ids live in LDS beside its rows (8 + 4 B per row): a
//     negative costs two LDS reads instead of a global 8-byte load;
//   * every pair is trained "pair per group" -- its central row prefetched while the previous four
//     pairs are scored, its gradient handed to the central row with f32 atomics through the
//     group's OWN transposition row: the four groups of a wave hand over at once, not in turns.
// Semantics as before (block_kernels.h train_record<RES>; the oracle's o_block_step): negative n of
// the pair at position p of its cell = draw(cell key, p k + n) through the cell's alias table,
// drawn again (kNegAttempts draws at most) when it is the context or the centre; dot product clamped at +-clip; two samples of a
// pair that name the same row are applied one after the other (deterministic form) or as one
// update with the summed coefficient (parallel form).  The deterministic instantiation (DET)
// runs the same code with the four groups taking turns when every run of the record is a single
// pair, and the oracle's run-major order otherwise.
#pragma once
#include "block_kernels.h"

namespace gn2v {

// a staged sample (16 bits): row inside the cell | phase << 12 | positive << 15
constexpr uint32_t kPkPositive = 0x8000u;  // this is the pair's context (label 1)
constexpr uint32_t kPkRowMask = 0x0FFFu;   // row inside the cell (cells hold at most 4 095 rows)
// The four groups of a wave run in lockstep: two of them that name the same row in the same step
// read it in the same instruction and store it in the same instruction -- one of the two updates
// is ALWAYS lost (counted: that alone was two thirds of what a cell's hub row lost, 5.8 of 9.3 %
// on a row that receives 4 % of the cell's samples).  So the staging gives every sample the number
// of EARLIER groups' samples of its step that name its row -- its phase -- and a step whose
// samples are not all of phase 0 is run once per phase.
constexpr uint32_t kPkPhaseShift = 12, kPkPhaseMask = 0x3000u;

// words of LDS a wave of the resident kernel stages a record in: four transposition rows |
// centre rows [C] | samples, u16 each, (k + 1) rounded up to 4 per pair, C lists + the null list
// (every sample the dummy row: what a group without a pair scores) | neighbour centres [2]
__host__ __device__ __forceinline__ uint32_t res_list_stride(uint32_t k) { return (k + 1 + 3) & ~3u; }
// (a transposition row holds 128 floats: wider rows are handed over in passes)
__host__ __device__ __forceinline__ uint32_t res_tr_floats(uint32_t ld) { return ld < 128 ? ld : 128; }
__host__ __device__ __forceinline__ uint32_t res_words_per_wave(uint32_t ld, uint32_t C, uint32_t k) {
    return (4 * res_tr_floats(ld) + C + (C + 1) * res_list_stride(k) / 2 + 2 + 3) & ~3u;
}
// Stride of a cell's rows IN LDS: rows wider than 64 floats are padded with zeros to the width of
// their instantiation (128, 256 or 512 floats), so that the scoring loops -- all the kernel does
// between loading a cell and writing it back -- run without a per-chunk predicate whatever the
// stride of the tables (a row of 224 floats ran at 0.50 of the atomic ceiling with the predicates,
// one of 256 at 0.83).  The padding columns stay zero: a row's update is coefficient x central row,
// whose padding is zero.
__host__ __device__ __forceinline__ uint32_t res_lds_stride(uint32_t ld) {
    return ld <= 64 ? ld : ld <= 128 ? 128u : ld <= 256 ? 256u : 512u;
}
// bytes of LDS a cell row costs: the row, its alias entry, its node id
__host__ __device__ __forceinline__ uint32_t res_bytes_per_row(uint32_t ld) {
    return res_lds_stride(ld) * 4 + 8 + 4;
}

struct ResCell {
    float *rows;                      // [cell_n + 1][ld]: the cell's rows, then the dummy row
    const unsigned long long *alias;  // [cell_n] (LDS copy of the cell's alias table) or nullptr
    const uint32_t *node;             // [cell_n]: the node behind every row
    uint32_t n, ld;
};

__device__ __forceinline__ float sigmoid_med3(float dot, float clip) {
    dot = __builtin_amdgcn_fmed3f(dot, -clip, clip);
    return __builtin_amdgcn_rcpf(1.0f + __expf(-dot));
}

// One step of a group: two samples (rows ra, rb of the cell; positives flagged) against the
// register row u.  SEQ: one after the other, exactly (deterministic form); else side by side --
// two independent chains -- with a row named twice updated once by the summed coefficient.
// POS: sample a is the FIRST of its pair's list -- the context, label 1; every other sample of a
// list is a negative (label 0): the label follows from the position, not from the staged word.
template <int CH, bool SEQ, bool POS>
__device__ __forceinline__ void res_step2(const ResCell &c, const Row<CH> &u, Row<CH> &g,
                                          uint32_t pa, uint32_t pb, float lrc, float clip, int q,
                                          uint32_t nchunks) {
    const uint32_t ra = pa & kPkRowMask, rb = pb & kPkRowMask;
    constexpr float lab_a = POS ? 1.f : 0.f, lab_b = 0.f;
    float *wa = c.rows + ra * c.ld, *wb = c.rows + rb * c.ld;
    if constexpr (SEQ) {
        float *w[2] = {wa, wb};
        const uint32_t r[2] = {ra, rb};
        const float lab[2] = {lab_a, lab_b};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            Row<CH> x;
#pragma unroll
            for (int cc = 0; cc < CH; ++cc) {
                const uint32_t ci = cc * 16 + q;
                x.c[cc] = ci < nchunks ? *reinterpret_cast<const float4 *>(w[i] + ci * 4)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const float dot = dot_rows_pk<CH>(u, x);
            const float var = r[i] != c.n ? (lab[i] - sigmoid_med3(dot, clip)) * lrc : 0.f;
            axpy<CH>(g, var, x);
#pragma unroll
            for (int cc = 0; cc < CH; ++cc) {
                const uint32_t ci = cc * 16 + q;
                if (ci < nchunks) {
                    float4 o = x.c[cc];
                    o.x += var * u.c[cc].x;
                    o.y += var * u.c[cc].y;
                    o.z += var * u.c[cc].z;
                    o.w += var * u.c[cc].w;
                    *reinterpret_cast<float4 *>(w[i] + ci * 4) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // b may name a's row
        }
        return;
    }
    Row<CH> xa, xb;
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const uint32_t ci = cc * 16 + q;
        xa.c[cc] = ci < nchunks ? *reinterpret_cast<const float4 *>(wa + ci * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        xb.c[cc] = ci < nchunks ? *reinterpret_cast<const float4 *>(wb + ci * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float dot_a = dot_rows_pk<CH>(u, xa), dot_b = dot_rows_pk<CH>(u, xb);
    float var_a = ra != c.n ? (lab_a - sigmoid_med3(dot_a, clip)) * lrc : 0.f;
    float var_b = rb != c.n ? (lab_b - sigmoid_med3(dot_b, clip)) * lrc : 0.f;
    axpy<CH>(g, var_a, xa);
    axpy<CH>(g, var_b, xb);
    // one row twice: a single update with the summed coefficient, b's store goes to the dummy row
    const bool same = ra == rb;
    var_a += same ? var_b : 0.f;
    wb = same ? c.rows + c.n * c.ld : wb;
    // The rows are read a SECOND time right before their 16-byte stores: an update of another
    // group is lost only when it lands between this read and the stores (~100 cycles), not
    // anywhere in the scoring above (~300).  (The barrier names the coefficients: the compiler
    // may not start the second reads before the sigmoids are through.)
    asm volatile("" : "+v"(var_a), "+v"(var_b)::"memory");
    Row<CH> oa, ob;
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const uint32_t ci = cc * 16 + q;
        oa.c[cc] = ci < nchunks ? *reinterpret_cast<const float4 *>(wa + ci * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
        ob.c[cc] = ci < nchunks ? *reinterpret_cast<const float4 *>(wb + ci * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    axpy<CH>(oa, var_a, u);
    axpy<CH>(ob, var_b, u);
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const uint32_t ci = cc * 16 + q;
        if (ci < nchunks) {
            *reinterpret_cast<float4 *>(wa + ci * 4) = oa.c[cc];
            *reinterpret_cast<float4 *>(wb + ci * 4) = ob.c[cc];
        }
    }
}

// a single sample (the odd one of a list)
template <int CH, bool POS>
__device__ __forceinline__ void res_step1(const ResCell &c, const Row<CH> &u, Row<CH> &g,
                                          uint32_t pa, float lrc, float clip, int q,
                                          uint32_t nchunks) {
    const uint32_t ra = pa & kPkRowMask;
    constexpr float lab_a = POS ? 1.f : 0.f;
    float *wa = c.rows + ra * c.ld;
    Row<CH> xa;
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const uint32_t ci = cc * 16 + q;
        xa.c[cc] = ci < nchunks ? *reinterpret_cast<const float4 *>(wa + ci * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float dot_a = dot_rows_pk<CH>(u, xa);
    float var_a = ra != c.n ? (lab_a - sigmoid_med3(dot_a, clip)) * lrc : 0.f;
    axpy<CH>(g, var_a, xa);
    asm volatile("" : "+v"(var_a)::"memory");
    Row<CH> oa;
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const uint32_t ci = cc * 16 + q;
        oa.c[cc] = ci < nchunks ? *reinterpret_cast<const float4 *>(wa + ci * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    axpy<CH>(oa, var_a, u);
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const uint32_t ci = cc * 16 + q;
        if (ci < nchunks) *reinterpret_cast<float4 *>(wa + ci * 4) = oa.c[cc];
    }
}

// A step of the wave = the two samples (packed in `w`: a | b << 16; ONE: only a) of each of its
// four groups.  All of phase 0 -- nearly always --: one pass.  Otherwise one pass per phase, the
// samples of the other phases naming the dummy row meanwhile (coefficient 0).
template <int CH, bool SEQ, bool ONE, bool POS>
__device__ __forceinline__ void res_step(const ResCell &c, const Row<CH> &u, Row<CH> &g,
                                         uint32_t w, float lrc, float clip, int q,
                                         uint32_t nchunks) {
    const uint32_t pa = w & 0xFFFFu, pb = w >> 16;
    if constexpr (!SEQ && !ONE && CH >= 8) {
        // rows of 512 floats: two samples side by side do not fit the 256 registers of a lane
        // next to u, g and the prefetched central row -- one after the other (their phases, counted
        // over both samples of the earlier groups, serialise each half at least as strictly)
        res_step<CH, SEQ, true, POS>(c, u, g, pa, lrc, clip, q, nchunks);
        res_step<CH, SEQ, true, false>(c, u, g, pb, lrc, clip, q, nchunks);
        return;
    }
    if constexpr (SEQ) {  // deterministic form: no phases are staged
        if constexpr (ONE)
            res_step1<CH, POS>(c, u, g, pa, lrc, clip, q, nchunks);
        else
            res_step2<CH, true, POS>(c, u, g, pa, pb, lrc, clip, q, nchunks);
        return;
    }
    const uint32_t phases = ONE ? (w & kPkPhaseMask) : (w & (kPkPhaseMask | (kPkPhaseMask << 16)));
    if (__builtin_expect(__ballot(phases != 0) == 0, 1)) {
        if constexpr (ONE)
            res_step1<CH, POS>(c, u, g, pa, lrc, clip, q, nchunks);
        else
            res_step2<CH, false, POS>(c, u, g, pa, pb, lrc, clip, q, nchunks);
        return;
    }
    const uint32_t pha = (pa & kPkPhaseMask) >> kPkPhaseShift;
    const uint32_t phb = ONE ? 0u : (pb & kPkPhaseMask) >> kPkPhaseShift;
    for (uint32_t pass = 0; pass < 4; ++pass) {
        const uint32_t ea = pha == pass ? (pa & ~kPkPhaseMask) : c.n;
        const uint32_t eb = phb == pass ? (pb & ~kPkPhaseMask) : c.n;
        if constexpr (ONE)
            res_step1<CH, POS>(c, u, g, ea, lrc, clip, q, nchunks);
        else
            res_step2<CH, false, POS>(c, u, g, ea, eb, lrc, clip, q, nchunks);
        if (__ballot(pha > pass || phb > pass) == 0) break;
        // the next phase reads what this one stored
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// the (k + 1) samples of one pair, staged at `list` (u16 each, padded to a multiple of four)
template <int CH, bool SEQ>
__device__ __forceinline__ void res_score_list(const ResCell &c, const Row<CH> &u, Row<CH> &g,
                                               const uint32_t *list, uint32_t kk, float lrc,
                                               float clip, int q, uint32_t nchunks) {
    // the list's first sample is the pair's context (label 1), every other one a negative
    uint32_t s = 0;
    if (kk >= 4) {
        const uint2 w = *reinterpret_cast<const uint2 *>(list);
        res_step<CH, SEQ, false, true>(c, u, g, w.x, lrc, clip, q, nchunks);
        res_step<CH, SEQ, false, false>(c, u, g, w.y, lrc, clip, q, nchunks);
        s = 4;
    }
    for (; s + 4 <= kk; s += 4) {  // four samples = one 8-byte read
        const uint2 w = *reinterpret_cast<const uint2 *>(list + (s >> 1));
        res_step<CH, SEQ, false, false>(c, u, g, w.x, lrc, clip, q, nchunks);
        res_step<CH, SEQ, false, false>(c, u, g, w.y, lrc, clip, q, nchunks);
    }
    const uint32_t rem = kk - s;  // the same for every pair: uniform branches
    if (rem) {
        const uint2 w = *reinterpret_cast<const uint2 *>(list + (s >> 1));
        if (s == 0) {  // lists of fewer than four samples (k < 3): the context is in here
            if (rem >= 2)
                res_step<CH, SEQ, false, true>(c, u, g, w.x, lrc, clip, q, nchunks);
            else
                res_step<CH, SEQ, true, true>(c, u, g, w.x, lrc, clip, q, nchunks);
        } else if (rem >= 2) {
            res_step<CH, SEQ, false, false>(c, u, g, w.x, lrc, clip, q, nchunks);
        } else {
            res_step<CH, SEQ, true, false>(c, u, g, w.x, lrc, clip, q, nchunks);
        }
        if (rem == 3) res_step<CH, SEQ, true, false>(c, u, g, w.y, lrc, clip, q, nchunks);
    }
}

// One record (n <= C consecutive sorted pairs from p0) of the cell.
template <int CH, bool DET>
__device__ __forceinline__ void res_train_record(const BlockArgs &a, const ResCell &c,
                                                 uint64_t lo, uint64_t p0, uint32_t n,
                                                 uint64_t ckey, float *s_tr, uint32_t *s_key,
                                                 uint32_t *s_pk, int lane, int grp, int q,
                                                 unsigned long long &pairs,
                                                 unsigned long long &runs) {
    const uint32_t k = a.k, kk = k + 1, stride = res_list_stride(k), nchunks = a.ld >> 2;
    // chunks of a row in LDS (res_lds_stride): the whole width of the instantiation from CH = 2
    const uint32_t lchunks = CH >= 2 ? CH * 16u : (c.ld >> 2);
    const uint32_t rowmask = a.p.row_bits >= 32 ? 0xFFFFFFFFu : ((1u << a.p.row_bits) - 1u);
    unsigned short *pk16 = reinterpret_cast<unsigned short *>(s_pk);
    wave_sync();
    // ---- staging: lane = (pair, half of its list); one sample per lane and turn
    {
        const uint32_t pr = (uint32_t)lane >> 1, half = (uint32_t)lane & 1u;
        const uint32_t first = half ? (kk + 1) / 2 : 0, last = half ? kk : (kk + 1) / 2;
        const bool have = pr < n;
        uint32_t xrow = 0, crow = 0;
        if (have) {
            const unsigned long long word = a.pairs[p0 + pr];
            const uint32_t low = (uint32_t)word & ((1u << a.p.ctx_bits) - 1u);
            xrow = low & ((1u << (a.p.ctx_bits - 1)) - 1u);
            crow = (uint32_t)(word >> a.p.ctx_bits) & rowmask;
            if (half == 0) s_key[pr] = crow;
        }
        const uint32_t cgid = crow * a.p.world + a.p.rank;
        // draw(ckey, t) = mix64(ckey + (t + 1) golden), t = (p0 - lo + pr) k + (s - 1): the
        // argument advances by the golden ratio per sample
        unsigned long long arg = ckey + ((p0 - lo + pr) * (unsigned long long)k + first) * kGolden;
        unsigned short *out = pk16 + pr * stride;
        for (uint32_t s = first; s < last; ++s, arg += kGolden) {
            uint32_t row = xrow | kPkPositive;
            if (s != 0) {
                // a negative that falls on the context or the centre is drawn again (attempt
                // j + 1 = mix64(attempt j + golden), kNegAttempts draws at most -- the oracle's
                // O_NEG_ATTEMPTS), so that a pair trains k negatives as under the reference's
                // graph-wide draw; only cells of one or two rows still give a sample up
                uint64_t r = mix64(arg);
                row = c.n;
                for (uint32_t att = 0; att < kNegAttempts; ++att, r = mix64(r + kGolden)) {
                    uint32_t local = (uint32_t)mulhi64(r, (uint64_t)c.n);
                    if (c.alias) {
                        const unsigned long long e = c.alias[local];
                        if ((uint32_t)r >= ((uint32_t)e & ~1u))
                            local = (uint32_t)(e >> 32) & ~kHubBit;
                    }
                    if (local != xrow && c.node[local] != cgid) {
                        row = local;
                        break;
                    }
                }
            }
            if (have) out[s] = (unsigned short)row;
        }
        // the padding of the list (up to three samples) names the dummy row
        if (have && half)
            for (uint32_t s = kk; s < stride; ++s) out[s] = (unsigned short)c.n;
        // the null list: what a group without a pair (the tail of a record; a group that waits
        // for its turn in the deterministic form) scores -- the dummy row, coefficient 0
        for (uint32_t s = (uint32_t)lane; s < stride; s += 64)
            pk16[a.p.record * stride + s] = (unsigned short)c.n;
    }
    wave_sync();
    if constexpr (!DET) {
        // ---- phases: lane = (quad of pairs, step of their lists); see kPkPhaseMask
        const uint32_t steps = stride >> 1, quads = (n + 3) >> 2;
        for (uint32_t item = (uint32_t)lane; item < quads * steps; item += 64) {
            const uint32_t quad = item / steps, step = item - quad * steps;
            uint32_t w[4];
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                const uint32_t pr = 4 * quad + gi;
                w[gi] = pr < n ? s_pk[pr * steps + step] : (c.n | (c.n << 16));
            }
#pragma unroll
            for (int gi = 1; gi < 4; ++gi) {
                const uint32_t ra = w[gi] & kPkRowMask, rb = (w[gi] >> 16) & kPkRowMask;
                uint32_t pa = 0, pb = 0;
#pragma unroll
                for (int gj = 0; gj < gi; ++gj) {
                    const uint32_t oa = w[gj] & kPkRowMask, ob = (w[gj] >> 16) & kPkRowMask;
                    pa += (ra == oa) + (ra == ob);
                    pb += (rb == oa) + (rb == ob);
                }
                pa = ra == c.n ? 0u : min(pa, 3u);
                pb = rb == c.n ? 0u : min(pb, 3u);
                if ((pa | pb) && 4 * quad + gi < n)
                    s_pk[(4 * quad + gi) * steps + step] =
                        w[gi] | (pa << kPkPhaseShift) | (pb << (16 + kPkPhaseShift));
            }
        }
        wave_sync();
    }
    bool ppg = true;
    if constexpr (DET) {
        // only records whose centres are ALL different (a resident plan's pairs are sorted by the
        // centre's highest bits: equal centres need not be neighbours, and the pair-per-group
        // loop reads the central rows of four -- with the prefetch, eight -- pairs ahead)
        bool dup = false;
        if ((uint32_t)lane < n)
            for (int j = 0; j < lane; ++j) dup |= s_key[j] == s_key[lane];
        ppg = __ballot(dup) == 0;
    }
    if (ppg) {
        // ---- pair per group: four pairs side by side; the central rows of the NEXT four are
        // fetched while these are scored
        Row<CH> u_next;
        load_row<CH>(u_next, a.central + (uint64_t)s_key[(uint32_t)grp < n ? grp : 0] * a.cld, q,
                     nchunks, (uint32_t)grp < n);
        float *tr = s_tr + grp * res_tr_floats(a.ld);
        for (uint32_t p4 = 0; p4 < n; p4 += 4) {
            const uint32_t pr = p4 + grp;
            const bool in_rec = pr < n;
            const uint32_t crow_id = s_key[in_rec ? pr : 0];
            float *crow = a.central + (uint64_t)crow_id * a.cld;
            Row<CH> u = u_next, g;
            {
                const uint32_t nx = pr + 4;
                load_row<CH>(u_next, a.central + (uint64_t)s_key[nx < n ? nx : 0] * a.cld, q,
                             nchunks, nx < n);
            }
            zero_row<CH>(g);
            float lrc = a.lr;
            if (a.flags & kFlagNormLr) {
                const uint64_t cn = (uint64_t)crow_id * a.p.world + a.p.rank;
                const uint64_t deg = a.g.row_ptr[cn + 1] - a.g.row_ptr[cn];
                if (deg) lrc = a.lr / (float)deg;
            }
            const uint32_t *null_list = s_pk + a.p.record * (stride >> 1);
            for (int turn = 0; turn < (DET ? 4 : 1); ++turn) {
                const bool have = in_rec && (!DET || grp == turn);
                // (a group without a pair scores the null list: it stores nothing but the dummy
                // row -- a stale copy of a real row written back could undo another group's update)
                res_score_list<CH, DET>(c, u, g, have ? s_pk + pr * (stride >> 1) : null_list, kk,
                                        lrc, a.clip, q, lchunks);
                // Hand-over: every group writes its gradient to its own transposition row; then
                // the WHOLE wave adds one pair's gradient after the other to its central row --
                // 64 lanes on 256 contiguous bytes of ONE row per atomic instruction.  (Measured,
                // profiles/r05_logs/r5_resident_v2_*_ab.log: an instruction whose four 16-lane groups add to four
                // different rows runs the kernel at 1.35e9 pairs/s, one row per instruction at
                // 2.26e9, no hand-over at all at 3.5e9 -- the L2 atomic units, 3.1e11 dword adds/s
                // in scripts/atomic_probe.hip, are what bounds this kernel: 128 dwords per pair,
                // 2.45e9 pairs/s at most.  Round 6: the TCC counters confirm it per cycle
                // (profiles/r06_atomic_summary.md) -- and folding the gradients of neighbouring
                // pairs of one centre in these rows before the atomics, with the pairs sorted by
                // the whole centre, took 17 % of the atomic rows away and left the kernel where it
                // was (2.32 against 2.31e9: profiles/r06_logs/r6_fold_ab.log): instruction issue
                // and LDS waits stand right behind the atomic units.  Not kept.)
#pragma unroll
                for (int pass = 0; pass < (CH + 1) / 2; ++pass) {
                    wave_sync();
#pragma unroll
                    for (int cc = 2 * pass; cc < 2 * pass + 2 && cc < CH; ++cc) {
                        const uint32_t ci = cc * 16 + q;
                        if (ci < nchunks)
                            *reinterpret_cast<float4 *>(tr + (ci - 32 * pass) * 4) = g.c[cc];
                    }
                    wave_sync();
                    {
                        const uint32_t tw = res_tr_floats(a.ld);
                        for (uint32_t gi = DET ? (uint32_t)turn : 0u; gi < (DET ? turn + 1u : 4u); ++gi) {
                            if (p4 + gi >= n) break;  // uniform
                            const float *src = s_tr + gi * tw;
                            const uint32_t f = 128 * pass + lane;
                            float *row = a.central + (uint64_t)s_key[p4 + gi] * a.cld + 128 * pass;
                            if (f < a.ld) unsafeAtomicAdd(row + lane, src[lane]);
                            if (f + 64 < a.ld && lane + 64 < tw) unsafeAtomicAdd(row + 64 + lane, src[64 + lane]);
                        }
                    }
                }
                if constexpr (DET) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            }
        }
        pairs += n;
        runs += n;
        return;
    }
    // ---- deterministic form, runs of equal centre (the oracle's order): one copy of the central
    // row per run of at most kMaxRun pairs, samples one after the other, gradient added at the end
    if constexpr (DET) {
        uint32_t r0 = 0;
        while (r0 < n) {
            const uint32_t crow_id = s_key[r0];
            uint32_t r1 = r0 + 1;
            while (r1 < n && r1 - r0 < kMaxRun && s_key[r1] == crow_id) ++r1;
            float lrc = a.lr;
            if (a.flags & kFlagNormLr) {
                const uint64_t cn = (uint64_t)crow_id * a.p.world + a.p.rank;
                const uint64_t deg = a.g.row_ptr[cn + 1] - a.g.row_ptr[cn];
                if (deg) lrc = a.lr / (float)deg;
            }
            float *crow = a.central + (uint64_t)crow_id * a.cld;
            Row<CH> u, g;
            load_row<CH>(u, crow, q, nchunks, true);
            zero_row<CH>(g);
            for (uint32_t pr = r0; pr < r1; ++pr)
                res_score_list<CH, true>(
                    c, u, g, s_pk + (grp == 0 ? pr : a.p.record) * (stride >> 1), kk, lrc, a.clip,
                    q, lchunks);
            if (grp == 0) scatter_add<CH, kWriteBack>(crow, q, nchunks, 1.0f, g, u);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            r0 = r1;
            ++runs;
        }
        pairs += n;
    }
}

// LDS (32-bit words): per wave res_words_per_wave(ld, C, k); then, shared by the workgroup,
// rows [(cell rows + 1) res_lds_stride(ld)] | alias [2 x cell rows] | node ids [cell rows] | cursor.
template <int CH, bool DET>
__device__ __forceinline__ void resident_cell_v2(BlockArgs &a, uint32_t *smem, uint32_t slice,
                                                 uint32_t max_rows, unsigned long long &pairs,
                                                 unsigned long long &runs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, q = lane & 15;
    const uint32_t C = a.p.record, k = a.k;
    const uint32_t per_wave = res_words_per_wave(a.ld, C, k);
    float *s_tr = reinterpret_cast<float *>(smem + wave * per_wave);
    uint32_t *s_key = smem + wave * per_wave + 4 * res_tr_floats(a.ld);
    uint32_t *s_pk = s_key + C;
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t cell = a.part * a.p.slices + slice;
    const uint64_t lo = a.cell_offsets[cell], hi = a.cell_offsets[cell + 1];
    if (hi == lo) return;  // the same for every wave of the workgroup
    const uint64_t part_rows = stripe_count(a.n_nodes, a.part, a.p.parts);
    const uint32_t cell_n = (uint32_t)stripe_count(part_rows, slice, a.p.slices);
    uint32_t *shared = smem + n_waves * per_wave;
    ResCell c{};
    c.rows = reinterpret_cast<float *>(shared);
    c.n = cell_n;
    // res_lds_stride(a.ld); the host picks CH = 1 for strides up to 64 floats, then 2, 4, 8
    c.ld = CH >= 2 ? CH * 64u : a.ld;
    unsigned long long *s_alias =
        reinterpret_cast<unsigned long long *>(shared + (size_t)(max_rows + 1) * c.ld);
    uint32_t *s_node = reinterpret_cast<uint32_t *>(s_alias + max_rows);
    uint32_t *s_cursor = s_node + max_rows;
    c.node = s_node;
    c.alias = a.alias ? s_alias : nullptr;
    const uint64_t cell_lo = a.cell_rows ? a.cell_rows[cell] : 0;
    for (uint32_t r = threadIdx.x; r < cell_n; r += blockDim.x) {
        const uint64_t xp = (uint64_t)(slice + a.p.slices * r) * a.p.parts + a.part;
        s_node[r] = a.inv ? a.inv[xp] : (uint32_t)xp;
        if (a.alias) s_alias[r] = a.alias[cell_lo + r];
    }
    if (threadIdx.x == 0) *s_cursor = 0;
    __syncthreads();
    auto row_ptr = [&](uint32_t r) -> float * {
        if (!a.inv) return sample_base(a, a.context, slice + a.p.slices * r);
        const uint32_t x = s_node[r];
        return a.ctx_table ? a.ctx_table + (uint64_t)x * a.ld
                           : a.context + (uint64_t)(x / a.p.parts) * a.xld;
    };
    for (uint32_t i = threadIdx.x; i < (cell_n + 1) * (c.ld >> 2); i += blockDim.x) {
        const uint32_t r = i / (c.ld >> 2), c4 = i - r * (c.ld >> 2);
        reinterpret_cast<float4 *>(c.rows + r * c.ld)[c4] =
            r < cell_n && c4 < (a.ld >> 2) ? reinterpret_cast<const float4 *>(row_ptr(r))[c4]
                                           : make_float4(0.f, 0.f, 0.f, 0.f);  // padding, dummy row
    }
    __syncthreads();
    const uint64_t R = (hi - lo + C - 1) / C;
    const uint64_t A = record_stride(R);
    const uint64_t ckey = cell_stream_key(a.ekey, a.block_id, cell);
    if constexpr (DET) {
        if (wave == 0)
            for (uint64_t t = 0; t < R; ++t) {
                const uint64_t rec = (t * A) % R;
                const uint64_t p0 = lo + rec * C;
                const uint32_t n = (uint32_t)min((uint64_t)C, hi - p0);
                res_train_record<CH, true>(a, c, lo, p0, n, ckey, s_tr, s_key, s_pk, lane, grp, q,
                                           pairs, runs);
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
            res_train_record<CH, false>(a, c, lo, p0, n, ckey, s_tr, s_key, s_pk, lane, grp, q,
                                        pairs, runs);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cell_n * (a.ld >> 2); i += blockDim.x) {
        const uint32_t r = i / (a.ld >> 2), c4 = i - r * (a.ld >> 2);
        reinterpret_cast<float4 *>(row_ptr(r))[c4] =
            reinterpret_cast<const float4 *>(c.rows + r * c.ld)[c4];
    }
}

// a.hot_n carries the rows the launch's LDS plan was made for (the largest cell of the plan)
// WAVES: sixteen waves per workgroup for rows up to 128 floats (four per SIMD: 128 registers);
// eight for wider rows (two per SIMD, 256 registers: a row of 256 floats is 16 registers a lane,
// of 512 floats 32, and a step holds five of them)
// LDQ: the tables' row stride in units of 32 floats as a compile-time constant (0: read from the
// arguments).  Not a nicety: with a run-time stride the per-chunk predicates of the central rows'
// loads and of the hand-over make the compiler wait for the prefetched central row inside the
// scoring loop and before every atomic (s_waitcnt vmcnt(0)): strides of 192 and 320 floats ran at
// 0.51 of their atomic ceiling where 256 and 512 reached 0.83-0.91.
template <int CH, int LDQ = 0, bool DET = false, int WAVES = 16>
__global__ __launch_bounds__(WAVES * 64) void sgns_resident_v2_kernel(BlockArgs a) {
    if constexpr (LDQ != 0) a.ld = LDQ * 32;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    unsigned long long pairs = 0, runs = 0;
    if constexpr (DET) {
        const uint32_t part0 = a.part, part_n = a.sweep ? a.sweep : 1;
        for (uint32_t p = 0; p < part_n; ++p) {
            a.part = part0 + p;
            if (a.part_ptrs) a.context = a.part_ptrs[a.part];
            for (uint32_t slice = 0; slice < a.p.slices; ++slice) {
                resident_cell_v2<CH, true>(a, smem, slice, a.hot_n, pairs, runs);
                __syncthreads();  // the LDS is the next cell's
            }
        }
    } else {
        uint32_t part_off = blockIdx.y, slice = blockIdx.x;
        if (a.order) {  // heaviest cells first (BlockArgs.order)
            const uint32_t id = a.order[blockIdx.y * gridDim.x + blockIdx.x];
            part_off = a.p.dslices.div(id);
            slice = id - part_off * a.p.slices;
        }
        if (a.part_ptrs) {
            a.part += part_off;
            a.context = a.part_ptrs[a.part];
        }
        resident_cell_v2<CH, false>(a, smem, slice, a.hot_n, pairs, runs);
    }
    if (a.counters && (threadIdx.x & 63) == 0 && pairs) {
        atomicAdd(&a.counters[0], pairs);
        atomicAdd(&a.counters[2], runs);
    }
}

}  // namespace gn2v

// GloVe on the co-occurrences of the walks: the third model of the reference's walk-based table
// (embedders/ensmallen_embedders/node2vec.py:16-26, "Node2Vec GloVe": models.GloVe; wrapper
// kwargs node2vec_glove.py:8-30).  Semantics restated in oracle/gn2v_oracle.c ("GloVe").
//
// cooc_kernel: every (centre, context) slot of the walks -> key = centre << 32 | context and the
// fixed-point weight 2^20 / distance (exact, order-independent sums).
// glove_kernel: one 16-lane group per non-zero entry (i, j, log X, f(X)); per entry two 512 B rows
// are read, g = f (u.v + b_i + b~_j - log X), and both rows are written back: 4 * d * 4 B of HBM
// traffic per entry, random rows -> HBM-bound, same row primitives and update modes as SGNS.
#pragma once
#include "train_kernels.h"

namespace gn2v {

constexpr uint32_t kCoocOne = 1u << 20;
constexpr unsigned long long kCoocUnused = 0x7FFFFFFFFFFFFFFFULL;

__global__ void cooc_kernel(const uint32_t *__restrict__ walks, uint64_t n_walks, uint32_t L,
                            uint32_t w, uint32_t min_dist, unsigned long long *__restrict__ keys,
                            unsigned long long *__restrict__ weights) {
    const uint64_t n = n_walks * L * 2 * w;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t slot = (uint32_t)(t % (2 * w));
        const uint64_t pos = t / (2 * w);
        const uint32_t i = (uint32_t)(pos % L);
        const uint64_t b = pos / L;
        const int64_t j = slot < w ? (int64_t)i - w + slot : (int64_t)i + 1 + (slot - w);
        unsigned long long key = kCoocUnused, weight = 0;
        if (j >= 0 && j < (int64_t)L) {
            const uint32_t dist = (uint32_t)(j > (int64_t)i ? j - i : i - j);
            const uint32_t ci = walks[b * L + i], xj = walks[b * L + j];
            // a sentinel at j means the walk ended before j (sentinels are a suffix)
            if (dist >= min_dist && ci != kSentinel && xj != kSentinel) {
                key = ((unsigned long long)ci << 32) | xj;
                weight = (kCoocOne + dist / 2) / dist;
            }
        }
        keys[t] = key;
        weights[t] = weight;
    }
}

struct GloveArgs {
    const uint32_t *rows, *cols;
    const float *logx, *fx;
    float *central, *contextual, *bias_c, *bias_x;
    uint64_t n_entries;
    uint32_t ld;
    float lr;
};

// float4 shape (slot (cc, e) of lane q = element 64cc + 4q + e) -> lane-contiguous shape (element
// 64cc + 16e + q) through the group's own LDS row
template <int CH>
__device__ __forceinline__ void group_to_contig(Row<CH> &out, const Row<CH> &in, float *s_grp,
                                                int q, uint32_t ld) {
    wave_sync();
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const uint32_t ci = cc * 16 + q;
        if (ci * 4 < ld) *reinterpret_cast<float4 *>(s_grp + ci * 4) = in.c[cc];
    }
    wave_sync();
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const uint32_t f = cc * 64 + q;
        out.c[cc].x = f < ld ? s_grp[f] : 0.f;
        out.c[cc].y = f + 16 < ld ? s_grp[f + 16] : 0.f;
        out.c[cc].z = f + 32 < ld ? s_grp[f + 32] : 0.f;
        out.c[cc].w = f + 48 < ld ? s_grp[f + 48] : 0.f;
    }
}

constexpr int kGloveBlock = 256;

// DET: one wavefront, one entry at a time in entry order (all groups compute it, group 0 writes):
// equals the oracle's sequential loop.  Otherwise Hogwild over entries, one per 16-lane group,
// with the record fast path when the 16 slots of a chunk share their central row.
template <int CH, int WM, bool DET>
__global__ __launch_bounds__(kGloveBlock) void glove_kernel(GloveArgs a) {
    extern __shared__ float s_glove[];  // atomic mode: [waves][4 groups][ld]
    const int lane = threadIdx.x & 63, grp = lane >> 4, q = lane & 15;
    const int wave = threadIdx.x >> 6;
    const uint32_t nchunks = a.ld >> 2;
    const uint64_t wave_id = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    // A wave takes 16 consecutive entries at a time: lane q of every group loads entry q (64 B
    // contiguous per array and instruction), the groups then pick theirs with a 16-wide shuffle.
    for (uint64_t base = wave_id * 16; base < a.n_entries; base += n_waves * 16) {
        const uint64_t el = base + q;
        uint32_t li = 0, lj = 0;
        float llx = 0.f, lf = 0.f;
        if (el < a.n_entries) {
            li = a.rows[el];
            lj = a.cols[el];
            llx = a.logx[el];
            lf = a.fx[el];
        }
        const bool lvalid = el < a.n_entries && lj != kSentinel;  // col = sentinel: padding slot
        const unsigned long long vmask = __ballot(lvalid);
        if (vmask == 0) continue;
        if constexpr (!DET) {
            // Record fast path: the slots of this chunk share their central row (the order
            // cooccurrence.entries / o_glove_entries produce).  The row is read once, kept in
            // registers, refreshed after every round of four entries with the four groups' summed
            // contributions, and written once: 1 KiB + 64 B instead of 2 KiB of HBM traffic per
            // entry.
            const uint32_t i0 = __shfl(li, __ffsll((long long)vmask) - 1, 16);
            if (__ballot(lvalid && li != i0) == 0) {
                float *ub = a.central + (uint64_t)i0 * a.ld;
                Row<CH> u, acc;
                load_row<CH>(u, ub, q, nchunks, true);
                zero_row<CH>(acc);
                float bi = a.bias_c[i0], accb = 0.f;
                float *s_grp = s_glove + ((size_t)wave * 4 + grp) * a.ld;  // atomic mode only
                for (int r = 0; r < 4; ++r) {
                    const int slot = r * 4 + grp;
                    const uint32_t j = __shfl(lj, slot, 16);
                    const float lx = __shfl(llx, slot, 16), f = __shfl(lf, slot, 16);
                    const bool valid = base + slot < a.n_entries && j != kSentinel;
                    float *vb = a.contextual + (uint64_t)(valid ? j : 0) * a.ld;
                    Row<CH> v, c;
                    load_row<CH>(v, vb, q, nchunks, valid);
                    const float dot = dot_rows<CH>(u, v);
                    const float bj = valid ? a.bias_x[j] : 0.f;
                    const float g = f * (((dot + bi) + bj) - lx);
                    const bool apply = valid && isfinite(g);
                    const float s = apply ? -a.lr * g : 0.f;
                    zero_row<CH>(c);
                    axpy<CH>(c, s, v);  // this entry's contribution to the central row
                    if constexpr (WM == kAtomic) {
                        Row<CH> uc;
                        group_to_contig<CH>(uc, u, s_grp, q, a.ld);
                        if (apply) {
                            scatter_add<CH, kAtomic>(vb, q, nchunks, s, uc, v);
                            if (q == 0) unsafeAtomicAdd(a.bias_x + j, s);
                        }
                    } else if (apply) {
                        scatter_add<CH, WM>(vb, q, nchunks, s, u, v);
                        if (q == 0) a.bias_x[j] = bj + s;
                    }
                    reduce_groups<CH>(c);
                    float ds = s;
                    ds += __shfl_xor(ds, 16);
                    ds += __shfl_xor(ds, 32);
                    axpy<CH>(u, 1.0f, c);
                    axpy<CH>(acc, 1.0f, c);
                    bi += ds;
                    accb += ds;
                }
                if constexpr (WM == kAtomic) {
                    Row<CH> accc;
                    group_to_contig<CH>(accc, acc, s_grp, q, a.ld);
                    if (grp == 0) {
                        scatter_add<CH, kAtomic>(ub, q, nchunks, 1.0f, accc, u);
                        if (q == 0) unsafeAtomicAdd(a.bias_c + i0, accb);
                    }
                } else if (grp == 0) {
                    scatter_add<CH, WM>(ub, q, nchunks, 0.0f, u, u);  // stores u
                    if (q == 0) a.bias_c[i0] = bi;
                }
                continue;
            }
        }
        constexpr int kRounds = DET ? 16 : 4;
        for (int r = 0; r < kRounds; ++r) {
            const int slot = DET ? r : r * 4 + grp;
            const uint32_t i = __shfl(li, slot, 16), j = __shfl(lj, slot, 16);
            const bool valid = base + slot < a.n_entries && j != kSentinel;
            const float lx = __shfl(llx, slot, 16), f = __shfl(lf, slot, 16);
            float *ub = a.central + (uint64_t)(valid ? i : 0) * a.ld;
            float *vb = a.contextual + (uint64_t)(valid ? j : 0) * a.ld;
            Row<CH> u, v;
            load_row<CH>(u, ub, q, nchunks, valid);
            load_row<CH>(v, vb, q, nchunks, valid);
            const float dot = dot_rows<CH>(u, v);
            const float bi = valid ? a.bias_c[i] : 0.f, bj = valid ? a.bias_x[j] : 0.f;
            const float g = f * (((dot + bi) + bj) - lx);
            const float s = -a.lr * g;
            const bool apply = valid && isfinite(g) && (!DET || grp == 0);
            if constexpr (WM == kAtomic) {
                float *s_grp = s_glove + ((size_t)wave * 4 + grp) * a.ld;
                Row<CH> uc, vc;
                group_to_contig<CH>(uc, u, s_grp, q, a.ld);
                group_to_contig<CH>(vc, v, s_grp, q, a.ld);
                if (apply) {
                    scatter_add<CH, kAtomic>(ub, q, nchunks, s, vc, u);
                    scatter_add<CH, kAtomic>(vb, q, nchunks, s, uc, v);
                    if (q == 0) {
                        unsafeAtomicAdd(a.bias_c + i, s);
                        unsafeAtomicAdd(a.bias_x + j, s);
                    }
                }
            } else {
                if (apply) {
                    scatter_add<CH, WM>(ub, q, nchunks, s, v, u);
                    scatter_add<CH, WM>(vb, q, nchunks, s, u, v);
                    if (q == 0) {
                        a.bias_c[i] = bi + s;
                        a.bias_x[j] = bj + s;
                    }
                }
            }
            if constexpr (DET) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        }
    }
}

}  // namespace gn2v

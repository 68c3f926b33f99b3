// Edge embeddings from node-embedding tables, fused with the row gather.
// Device form of the reference's node -> edge feature operators
// (embiggen/embedding_transformers/edge_transformer.py:12-343, method table :348-361), which the
// reference applies with numpy after hstack-ing the embedder's tables
// (embedding_transformers/node_transformer.py:110).  One 16-lane group per edge, float4 chunks,
// DPP row reductions for the three norm-style operators.  Pure HBM streaming: 2 rows in, <= 2 rows out.
#pragma once
#include "train_kernels.h"

namespace gn2v {

enum EdgeMethod : uint32_t {
    kHadamard = 0, kSum, kAverage, kL1, kAbsoluteL1, kSquaredL2, kL2, kConcatenate, kMin, kMax,
    kL2Distance, kCosineSimilarity, kEdgeMethodCount
};

template <int CH>
__global__ __launch_bounds__(256) void edge_embedding_kernel(
    const float *__restrict__ src_table, const float *__restrict__ dst_table, uint32_t d,
    uint32_t ld, const uint32_t *__restrict__ src_ids, const uint32_t *__restrict__ dst_ids,
    uint64_t n_edges, uint32_t method, float *__restrict__ out, uint32_t out_ld) {
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 4, q = lane & 15;
    const uint32_t nchunks = ld >> 2;
    // whole 16-byte stores where the output rows allow them (d and out_ld multiples of 4 floats,
    // out aligned: the transformers' case) -- the element-wise operators write as much as they read
    const bool vec = (d & 3u) == 0 && (out_ld & 3u) == 0 && ((uintptr_t)out & 15u) == 0;
    const uint64_t groups = (uint64_t)gridDim.x * (blockDim.x >> 4);
    for (uint64_t e = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; ; e += groups) {
        // all lanes of a wave must reach the DPP reductions together
        const uint64_t wave_first = e - grp;
        if (wave_first >= n_edges) break;
        const bool live = e < n_edges;
        Row<CH> a, b;
        load_row<CH>(a, src_table + (uint64_t)(live ? src_ids[e] : 0) * ld, q, nchunks, live);
        load_row<CH>(b, dst_table + (uint64_t)(live ? dst_ids[e] : 0) * ld, q, nchunks, live);
        float *o = out + e * out_ld;
        if (method == kL2Distance || method == kCosineSimilarity) {
            float dd = 0.f, ab = 0.f, aa = 0.f, bb = 0.f;
#pragma unroll
            for (int cc = 0; cc < CH; ++cc) {
                const float *x = reinterpret_cast<const float *>(&a.c[cc]);
                const float *y = reinterpret_cast<const float *>(&b.c[cc]);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float df = x[t] - y[t];
                    dd += df * df;
                    ab += x[t] * y[t];
                    aa += x[t] * x[t];
                    bb += y[t] * y[t];
                }
            }
            dd = group16_sum(dd);
            ab = group16_sum(ab);
            aa = group16_sum(aa);
            bb = group16_sum(bb);
            if (live && q == 0) {
                if (method == kL2Distance) {
                    o[0] = sqrtf(dd);
                } else {
                    float norm = sqrtf(aa) * sqrtf(bb);
                    if (norm < 1e-6f) norm = 1e-6f;  // edge_transformer.py:266
                    o[0] = ab / norm;
                }
            }
            continue;
        }
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) {
            const uint32_t ci = cc * 16 + q;
            if (!live || ci >= nchunks) continue;
            const float *x = reinterpret_cast<const float *>(&a.c[cc]);
            const float *y = reinterpret_cast<const float *>(&b.c[cc]);
            float4 rv;
            float *rr = reinterpret_cast<float *>(&rv);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t col = ci * 4 + t;
                if (col >= d) continue;
                float r;
                switch (method) {
                    case kHadamard: r = x[t] * y[t]; break;
                    case kSum: r = x[t] + y[t]; break;
                    case kAverage: r = (x[t] + y[t]) / 2.0f; break;
                    case kL1: r = x[t] - y[t]; break;
                    case kAbsoluteL1: r = fabsf(x[t] - y[t]); break;
                    case kSquaredL2: r = (x[t] - y[t]) * (x[t] - y[t]); break;
                    case kL2: r = sqrtf((x[t] - y[t]) * (x[t] - y[t])); break;
                    case kMin: r = fminf(x[t], y[t]); break;
                    case kMax: r = fmaxf(x[t], y[t]); break;
                    default: r = x[t]; break;  // kConcatenate: first half
                }
                rr[t] = r;
                if (!vec) {
                    o[col] = r;
                    if (method == kConcatenate) o[d + col] = y[t];
                }
            }
            if (vec && ci * 4 < d) {
                *reinterpret_cast<float4 *>(o + ci * 4) = rv;
                if (method == kConcatenate) *reinterpret_cast<float4 *>(o + d + ci * 4) = b.c[cc];
            }
        }
    }
}

}  // namespace gn2v

// Microbenchmark (round 5): what a row of f32 atomic adds costs in HBM / L2 on gfx950, by the
// shape of the instruction.  Table of `rows` rows x 128 floats (512 B); every wave adds to
// pseudo-random rows, 128 dwords per row, in one of the shapes
//   0  8 instructions, each: four 16-lane groups -> four DIFFERENT rows, 64 B each   (V2 hand-over)
//   1  32 instructions of 16 active lanes: one row at a time, 64 B per instruction   (round 4)
//   2  8 instructions, each: 64 lanes -> ONE row, 256 contiguous bytes (4 rows in turn)
//   3  as 2 with 128 B per half-wave to two rows (32 lanes a row)
//   4  plain write-through stores of the same bytes (no atomics), for scale
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/atomic_probe.hip -o scripts/r5/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(256) void probe(float *table, unsigned rows, int iters) {
    const int lane = threadIdx.x & 63, grp = lane >> 4, q = lane & 15;
    unsigned s = (blockIdx.x * 256 + (threadIdx.x & ~63)) * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        unsigned r[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            s = s * 1664525u + 1013904223u;
            r[g] = (s >> 4) % rows;
        }
        const float v = 1e-6f * (it + 1);
        if constexpr (MODE == 0) {
            float *row = table + (size_t)r[grp] * 128;
#pragma unroll
            for (int j = 0; j < 8; ++j) unsafeAtomicAdd(row + 16 * j + q, v);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (grp == g) {
                    float *row = table + (size_t)r[g] * 128;
#pragma unroll
                    for (int j = 0; j < 8; ++j) unsafeAtomicAdd(row + 16 * j + q, v);
                }
            }
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float *row = table + (size_t)r[g] * 128;
                unsafeAtomicAdd(row + lane, v);
                unsafeAtomicAdd(row + 64 + lane, v);
            }
        } else if constexpr (MODE == 3) {
            const int half = lane >> 5, l = lane & 31;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float *row = table + (size_t)r[2 * p + half] * 128;
#pragma unroll
                for (int j = 0; j < 4; ++j) unsafeAtomicAdd(row + 32 * j + l, v);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float *row = table + (size_t)r[g] * 128;
                __builtin_nontemporal_store(v, row + lane);
                __builtin_nontemporal_store(v, row + 64 + lane);
            }
        }
    }
}

template <int MODE>
static void run(const char *name, float *table, unsigned rows, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int blocks = 256 * 8;
    probe<MODE><<<blocks, 256>>>(table, rows, 16);
    hipEventRecord(a);
    probe<MODE><<<blocks, 256>>>(table, rows, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double row_adds = (double)blocks * 4 * 4 * iters;  // waves x rows per iteration
    printf("mode %d %-52s %8.2f ms  %.3e row adds/s  %.3e dword atomics/s\n", MODE, name, ms,
           row_adds / (ms * 1e-3), row_adds * 128 / (ms * 1e-3));
}

int main(int argc, char **argv) {
    const unsigned rows = argc > 1 ? atoi(argv[1]) : 10000000;
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    float *table;
    hipMalloc(&table, (size_t)rows * 512);
    hipMemset(table, 0, (size_t)rows * 512);
    run<0>("8 x (4 rows x 64 B)", table, rows, iters);
    run<1>("32 x (1 row x 64 B, 16 lanes)", table, rows, iters);
    run<2>("8 x (1 row x 256 B)", table, rows, iters);
    run<3>("8 x (2 rows x 128 B)", table, rows, iters);
    run<4>("stores, 8 x (1 row x 256 B)", table, rows, iters);
    return 0;
}

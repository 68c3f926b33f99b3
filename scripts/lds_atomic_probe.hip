// Microbenchmark (round 5): what an LDS row update costs on gfx950 in the shapes the resident
// kernel could use -- 16 waves per workgroup, one workgroup per CU, each 16-lane group picks a
// pseudo-random row of 200 rows x 128 floats per step:
//   0  read row (2 x ds_read_b128), re-read, 2 x ds_write_b128      (what ships: racy)
//   1  read row, 8 x ds_add_f32 at the float4 layout's addresses     (lane q: element 4q + e)
//   2  read row, 8 x ds_add_f32 lane-contiguous (element 16 j + q)   (bank-conflict free in a group)
//   3  as 2, the second 16-lane group of a half-wave rotated by one j (both groups on other banks)
//   4  read row only
// Build: hipcc --offload-arch=gfx950 -O3 scripts/lds_atomic_probe.hip -o scripts/r5/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) float lds_f32;
__device__ __forceinline__ void lds_add(float *p, float x) {
    (void)__hip_atomic_fetch_add((lds_f32 *)p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int MODE>
__global__ __launch_bounds__(1024) void probe(float *out, int iters, int rows) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, grp = lane >> 4, q = lane & 15;
    for (int i = threadIdx.x; i < rows * 128; i += blockDim.x) lds[i] = 0.001f * (i & 255);
    __syncthreads();
    unsigned s = (blockIdx.x * 1024 + (threadIdx.x & ~15)) * 2654435761u + 12345u;
    float acc = 0.f;
    float4 u0 = make_float4(1e-6f, 2e-6f, 3e-6f, 4e-6f), u1 = u0;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const unsigned row = (s >> 8) % (unsigned)rows;
        float *rw = lds + row * 128;
        const float4 a = *reinterpret_cast<const float4 *>(rw + 4 * q);
        const float4 b = *reinterpret_cast<const float4 *>(rw + 64 + 4 * q);
        float dot = a.x * u0.x + a.y * u0.y + a.z * u0.z + a.w * u0.w + b.x * u1.x + b.y * u1.y +
                    b.z * u1.z + b.w * u1.w;
        dot += __shfl_xor(dot, 1);
        dot += __shfl_xor(dot, 2);
        const float var = 1e-3f * dot;
        acc += var;
        if constexpr (MODE == 0) {
            asm volatile("" ::: "memory");
            float4 oa = *reinterpret_cast<const float4 *>(rw + 4 * q);
            float4 ob = *reinterpret_cast<const float4 *>(rw + 64 + 4 * q);
            oa.x += var * u0.x; oa.y += var * u0.y; oa.z += var * u0.z; oa.w += var * u0.w;
            ob.x += var * u1.x; ob.y += var * u1.y; ob.z += var * u1.z; ob.w += var * u1.w;
            *reinterpret_cast<float4 *>(rw + 4 * q) = oa;
            *reinterpret_cast<float4 *>(rw + 64 + 4 * q) = ob;
        } else if constexpr (MODE == 1) {
            lds_add(rw + 4 * q + 0, var * u0.x);
            lds_add(rw + 4 * q + 1, var * u0.y);
            lds_add(rw + 4 * q + 2, var * u0.z);
            lds_add(rw + 4 * q + 3, var * u0.w);
            lds_add(rw + 64 + 4 * q + 0, var * u1.x);
            lds_add(rw + 64 + 4 * q + 1, var * u1.y);
            lds_add(rw + 64 + 4 * q + 2, var * u1.z);
            lds_add(rw + 64 + 4 * q + 3, var * u1.w);
        } else if constexpr (MODE == 2) {
            const float v[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) lds_add(rw + 16 * j + q, var * v[j]);
        } else if constexpr (MODE == 3) {
            const float v[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
            const int rot = grp & 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) lds_add(rw + 16 * ((j + rot) & 7) + q, var * v[j]);
        }
    }
    __syncthreads();
    if (acc == 12345.678f) out[threadIdx.x] = acc + lds[threadIdx.x];
}

template <int MODE>
static void run(const char *name, int iters, int rows) {
    float *out;
    hipMalloc(&out, 4096);
    const size_t lds = (size_t)rows * 128 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    probe<MODE><<<256, 1024, lds>>>(out, 100, rows);
    hipEventRecord(a);
    probe<MODE><<<256, 1024, lds>>>(out, iters, rows);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    // per CU: 16 waves x 4 groups = 64 row updates per iteration
    const double per_cu_updates_per_s = 64.0 * iters / (ms * 1e-3);
    printf("mode %d %-44s %8.2f ms  %6.1f ns per wave step  %.3e row updates/s/CU  %.3e /s chip\n",
           MODE, name, ms, ms * 1e6 / iters / 4.0 /* 4 waves per SIMD in turn */,
           per_cu_updates_per_s, per_cu_updates_per_s * 256);
    hipFree(out);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int rows = argc > 2 ? atoi(argv[2]) : 200;
    run<4>("read only", iters, rows);
    run<0>("read, re-read, 2 x ds_write_b128", iters, rows);
    run<1>("read, 8 x ds_add_f32 float4 layout", iters, rows);
    run<2>("read, 8 x ds_add_f32 lane-contiguous", iters, rows);
    run<3>("read, 8 x ds_add_f32 lane-contiguous, rotated", iters, rows);
    return 0;
}

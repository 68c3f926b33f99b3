// Small data-parallel helpers: table initialisation and the synthetic Barabasi-Albert generator
// used by the benchmark configurations.
#pragma once
#include "rng.h"
#include "train_kernels.h"

namespace gn2v {

// mask[0] |= 1 << (XCD of this workgroup): a launch of many workgroups reports the XCDs in use
static __global__ void xcc_probe_kernel(unsigned int *mask) {
    if (threadIdx.x == 0) atomicOr(mask, 1u << xcc_id());
}

// seen[xcc * 8 + word] |= bit of the physical CU (SE, SH, CU id inside the XCD) this workgroup
// runs on; the short sleep keeps the first CUs from draining the whole grid before the others
// get their share.  gn2v_graph_reserve_cus reads back which CUs a masked stream really uses.
static __global__ void cu_probe_kernel(unsigned int *seen) {
    if (threadIdx.x == 0) {
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        const uint32_t id = (se << 5) | (sh << 4) | cu;  // < 256
        atomicOr(&seen[xcc_id() * 8 + (id >> 5)], 1u << (id & 31));
    }
    for (int i = 0; i < 64; ++i) __builtin_amdgcn_s_sleep(127);
}

// table[r][c] = (2*u24 - 1) * scale with u24 from draw(key, r*d + c); padding columns are zero.
__global__ void init_kernel(float *__restrict__ t, uint64_t n_rows, uint32_t d, uint32_t ld,
                            uint64_t key, float scale) {
    const uint64_t n = n_rows * ld;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / ld;
        const uint32_t c = (uint32_t)(i - r * ld);
        float v = 0.f;
        if (c < d) {
            const uint64_t h = draw(key, r * d + c);
            const float u = __fmul_rn((float)(h >> 40), 1.0f / 16777216.0f);
            v = __fmul_rn(__fsub_rn(__fmul_rn(2.0f, u), 1.0f), scale);
        }
        t[i] = v;
    }
}

// Pointer-chasing Barabasi-Albert: edge e belongs to node v = e/m + 1 and copies the endpoint at a
// uniformly random earlier position of the virtual endpoint array; odd positions (targets) are
// resolved by re-deriving that edge's own draw, so every edge is independent of the others.
__device__ __forceinline__ uint32_t ba_target(uint64_t bkey, uint64_t e, uint32_t m) {
    for (;;) {
        const uint64_t v = e / m + 1;
        const uint64_t limit = 2 * (v - 1) * m + 1;
        const uint64_t x = mulhi64(draw(bkey, e), limit);
        if (x == 0) return 0;
        const uint64_t p = x - 1;
        const uint64_t e2 = p >> 1;
        if ((p & 1) == 0) return (uint32_t)(e2 / m + 1);
        e = e2;
    }
}

__global__ void ba_kernel(uint64_t bkey, uint64_t n_e, uint32_t m, uint32_t *__restrict__ src,
                          uint32_t *__restrict__ dst) {
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_e;
         e += (uint64_t)gridDim.x * blockDim.x) {
        src[e] = (uint32_t)(e / m + 1);
        dst[e] = ba_target(bkey, e, m);
    }
}

}  // namespace gn2v

// Counter-based randomness shared by every gn2v kernel (device side).
// Stream layout (DESIGN.md "Randomness"): all draws are splitmix64 finalisers of key + (t+1)*phi;
// keys derive from (seed, epoch, walk id) so any walk / negative can be regenerated anywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gn2v {

constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ULL;
constexpr uint64_t kTagEpoch = 0x6E32764B45590A01ULL;
constexpr uint64_t kTagNeg = 0xA5A5F00DC0FFEE11ULL;
constexpr uint64_t kTagDown = 0x5BD1E995D00D1E55ULL;
constexpr uint64_t kTagBA = 0xBA5EBA11BA5EBA11ULL;
constexpr uint64_t kTagInit = 0x1417AB1E00000000ULL;
constexpr uint32_t kSentinel = 0xFFFFFFFFu;

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

__host__ __device__ __forceinline__ uint64_t draw(uint64_t key, uint64_t t) {
    return mix64(key + (t + 1) * kGolden);
}

__host__ __device__ __forceinline__ uint64_t epoch_key(uint64_t seed, uint64_t epoch) {
    return draw(mix64(seed ^ kTagEpoch), epoch);
}

__device__ __forceinline__ uint64_t mulhi64(uint64_t a, uint64_t b) { return __umul64hi(a, b); }

}  // namespace gn2v

// bcd_common.h -- shared host/device helpers of the gfx950 engine (internal; the public surface is include/bcd_hip.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// pixel states of the marking strategy (bcd_hip_active_set)
enum : uint8_t { BCD_ST_NONE = 0, BCD_ST_IN = 1, BCD_ST_OUT = 2, BCD_ST_UNDECIDED = 3 };

// ---- visiting order ---------------------------------------------------------------------------------
// The reference visits main pixels in scanline order (-r 0, 1 thread; src/core/Denoiser.cpp:136-146) or
// in a wall-clock-seeded shuffle (-r 1; :416-420).  Here the order is the ascending order of a 64-bit
// key: scanline -> key = linear index; random -> (hash32(index, seed) << 32) | index.  Any permutation
// is an admissible outcome of the reference's shuffle; a seeded one is reproducible.
__host__ __device__ inline uint32_t bcd_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
__host__ __device__ inline uint64_t bcd_order_key(uint32_t idx, int random_order, uint32_t seed)
{
    uint32_t hi = random_order ? bcd_mix32(idx ^ bcd_mix32(seed + 0x9E3779B9u)) : 0u;
    return ((uint64_t)hi << 32) | idx;
}
// uniform in [0,1) per pixel, for skip probabilities strictly between 0 and 1
__host__ __device__ inline float bcd_unit_hash(uint32_t idx, uint32_t seed)
{
    return (float)(bcd_mix32(idx * 0x9E3779B1u + bcd_mix32(seed ^ 0x51ed270bu)) >> 8) * (1.0f / 16777216.0f);
}

// ---- displacement tables -----------------------------------------------------------------------------
// half-plane displacement set used by the pair-distance planes: (dl,dc) with dl in [0,b];
// dc in [0,b] for dl == 0 and [-b,b] otherwise.  index(dl,dc):
__host__ __device__ inline int bcd_delta_index(int dl, int dc, int b)
{
    return dl == 0 ? dc : (b + 1) + (dl - 1) * (2 * b + 1) + (dc + b);
}
__host__ __device__ inline int bcd_delta_count(int b) { return (b + 1) + b * (2 * b + 1); }

// ---- fast similarity path (k_similarity_fast.hip): approximate distance planes, exact decision at the threshold ----
// relative half-width of the band around tau inside which a pair is re-evaluated exactly; the worst-case deviation of the
// approximate patch distance from the reference's fp32 value is ~1e-5 (derivation in k_similarity_fast.hip)
#define BCD_APPROX_DELTA 6.103515625e-05f /* 2^-14 */
struct BcdBorderline {
    float tau_hi;      // tau (1 + delta); the kernels get tau (1 - delta) as their threshold
    uint2 *list;       // (pixel index, displacement index) of the pairs with tau_lo < d' <= tau_hi
    int *counter;      // number of appended pairs (may exceed capacity: then the host falls back to the exact kernels)
    int capacity;
};

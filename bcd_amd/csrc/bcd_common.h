// bcd_common.h -- shared host/device helpers of the gfx950 engine (internal; the public surface is include/bcd_hip.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// pixel states of the marking strategy (bcd_hip_active_set)
enum : uint8_t { BCD_ST_NONE = 0, BCD_ST_IN = 1, BCD_ST_OUT = 2, BCD_ST_UNDECIDED = 3 };

// ---- visiting order ---------------------------------------------------------------------------------
// The reference visits main pixels in scanline order (-r 0, 1 thread; src/core/Denoiser.cpp:136-146), in a
// wall-clock-seeded shuffle (-r 1; :416-420), or -- -r 0 with several threads -- strip by strip, the even strips of 2b
// lines first and then the odd ones (reorderPixelSetJumpNextStrip, :381-414; a trailing partial strip stays last).
// Here the order is the ascending order of a 64-bit key:
//   mode 0 scanline -> key = linear index;   mode 1 random -> (hash32(index, seed) << 32) | index  (any permutation is an
//   admissible outcome of the reference's shuffle; a seeded one is reproducible);   mode 2 strips -> (position in the
//   reordered list << 32) | index, with the frame geometry packed into the `seed` argument (bcd_strip_order_seed).
__host__ __device__ inline uint32_t bcd_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
// geometry of the strip order: width and number of main lines (13 bits each), patch radius (2 bits), search radius (4 bits)
__host__ __device__ inline uint32_t bcd_strip_order_seed(int W, int H, int w, int b)
{
    return (uint32_t)W | ((uint32_t)(H - 2 * w) << 13) | ((uint32_t)w << 26) | ((uint32_t)b << 28);
}
__host__ __device__ inline uint64_t bcd_order_key(uint32_t idx, int order_mode, uint32_t seed)
{
    if (order_mode == 2) {
        const int W = (int)(seed & 8191u), Hm = (int)((seed >> 13) & 8191u), w = (int)((seed >> 26) & 3u), b = (int)(seed >> 28);
        const int Wm = W - 2 * w, row = (int)(idx / (uint32_t)W) - w, col = (int)(idx % (uint32_t)W) - w;
        uint32_t pos = idx; // (not a main pixel: never compared)
        if (row >= 0 && row < Hm && col >= 0 && col < Wm && b > 0) {
            const int lines = 2 * b, strip = row / lines, nfull = Hm / lines; // a chunk is Wm * 2b pixels = one strip of 2b lines
            const int first = strip >= nfull ? strip : ((strip & 1) ? (nfull + 1) / 2 + strip / 2 : strip / 2); // even strips, then the odd ones
            pos = (uint32_t)((first * lines + (row - strip * lines)) * Wm + col);
        }
        return ((uint64_t)pos << 32) | idx;
    }
    uint32_t hi = order_mode ? bcd_mix32(idx ^ bcd_mix32(seed + 0x9E3779B9u)) : 0u;
    return ((uint64_t)hi << 32) | idx;
}
// uniform in [0,1) per pixel, for skip probabilities strictly between 0 and 1
__host__ __device__ inline float bcd_unit_hash(uint32_t idx, uint32_t seed)
{
    return (float)(bcd_mix32(idx * 0x9E3779B1u + bcd_mix32(seed ^ 0x51ed270bu)) >> 8) * (1.0f / 16777216.0f);
}

// counters of undecided pixels are spread over BCD_CNT_LINES sub-counters, one per 128-byte line (k_active.hip)
#define BCD_CNT_LINES 64
#define BCD_CNT_STRIDE 32 /* ints */

// work queues of the persistent estimate kernels (k_bayes27.hip): BCD_WORK_QUEUES counters per kernel, one per 128-byte line
#define BCD_WORK_QUEUES 8
#define BCD_WORK_STRIDE 32 /* ints */
#define BCD_WORK_INTS (4 * BCD_WORK_QUEUES * BCD_WORK_STRIDE) /* prepare | solve | finish | finish of the redo list */

// ---- displacement tables -----------------------------------------------------------------------------
// half-plane displacement set used by the pair-distance planes: (dl,dc) with dl in [0,b];
// dc in [0,b] for dl == 0 and [-b,b] otherwise.  index(dl,dc):
__host__ __device__ inline int bcd_delta_index(int dl, int dc, int b)
{
    return dl == 0 ? dc : (b + 1) + (dl - 1) * (2 * b + 1) + (dc + b);
}
__host__ __device__ inline int bcd_delta_count(int b) { return (b + 1) + b * (2 * b + 1); }

// ---- fast similarity path (k_similarity_fast.hip): approximate distance planes, exact decision at the threshold ----
// The approximate T plane is stored in binary16 (3 instead of 5 bytes per plane entry with the count byte): every entry then carries a
// relative error <= 2^-11 on top of the ~1e-5 of the approximate arithmetic (derivation in k_similarity_fast.hip), all entries are
// >= 0, so the patch distance is within 5.0e-4 of the reference's; pairs inside tau (1 +- 2^-10) are re-evaluated exactly.
// Thresholds outside [BCD_APPROX_TAU_MIN, BCD_APPROX_TAU_MAX] take the exact kernels: below, binary16 subnormals (absolute error
// 2^-25 per entry) would matter; above, an entry that overflowed to +inf could belong to a similar pair.
// (The subnormal figure assumes that binary16 subnormals are kept, which is HIP's default -- float_denorm_mode_16_64 = 3 in the
// kernel descriptors; do not build these files with -fgpu-flush-denormals-to-zero.)
#define BCD_APPROX_DELTA 9.765625e-04f /* 2^-10 */
#define BCD_APPROX_TAU_MIN 0.015625f
#define BCD_APPROX_TAU_MAX 64.f
typedef unsigned short bcd_half_bits; // storage type of the approximate T plane (binary16)
struct BcdBorderline {
    float tau_hi;      // tau (1 + delta); the kernels get tau (1 - delta) as their threshold
    uint2 *list;       // (pixel index, displacement index) of the pairs with tau_lo < d' <= tau_hi
    int *counter;      // number of appended pairs (may exceed capacity: then the host falls back to the exact kernels)
    int capacity;
};

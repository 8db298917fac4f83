// k_active.hip -- the "marking strategy" of the reference as a parallel fixed point.
//
// Reference (src/core/DenoisingUnit.cpp:164-173,182-191,690): main pixels are visited in a fixed order;
// a pixel is skipped when it was marked by an earlier processed pixel q with |S(q)| >= 3P+1 and p in S(q);
// pixels processed through the fallback path (|S| < 3P+1) mark nobody.  Because distances are bitwise
// symmetric, p in S(q) <=> q in S(p), so with the visiting order expressed as a key:
//     p is processed  <=>  no q in S(p) with key(q) < key(p), |S(q)| >= 3P+1, q processed.
// This is the lexicographically-first independent-set construction on the similarity graph; it has a
// unique solution, reached by iterating "decide p once all its earlier strong similar neighbours are
// decided".  With a random order the dependency depth is O(log^2 n) (Blelloch, Fineman, Shun 2012).
#include "bcd_common.h"

namespace {

constexpr int LISTS_PER_THREAD = 8;

// Counters of still undecided pixels: one atomic per workgroup, spread over BCD_CNT_LINES sub-counters in cache lines of their own
// (the caller sums them with bcd_launch_sum_counter_lines).  A single counter serves ~50 same-address atomics per microsecond, and
// a 1080p scale has 8 160 tiles: 0.13 ms for k_mark_deps alone was the counter, not the kernel (r3).
__device__ inline int *counter_line(int *base) { return base + ((blockIdx.x + 5 * blockIdx.y) & (BCD_CNT_LINES - 1)) * BCD_CNT_STRIDE; }

// total_out (optional; round 6, the band driver): what the rank contributes to the all-reduced count of undecided pixels, left on the DEVICE for
// ncclAllReduce -- the count after the batch's last launch, plus 2^40 when `flags` (the validity words of the similarity kernels: [0] range / count
// flag, [2] another sample count, [3] borderline pairs against `border_capacity`; same test as similarity_redo_mode in bcd_api.hip) say that
// this rank's masks have to be recomputed
__global__ void k_sum_counter_lines(const int *__restrict__ lines, int rounds, int *__restrict__ out, long long *__restrict__ total_out,
                                    const int *__restrict__ flags, int border_capacity)
{
    const int r = blockIdx.x, t = threadIdx.x; // one wavefront per round
    int v = t < BCD_CNT_LINES ? lines[((size_t)r * BCD_CNT_LINES + t) * BCD_CNT_STRIDE] : 0;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (t == 0 && r < rounds) out[r] = v;
    if (t == 0 && r == rounds - 1 && total_out) {
        long long tot = v;
        if (flags && (flags[0] != 0 || (border_capacity > 0 && (flags[2] != 0 || flags[3] > border_capacity)))) tot += 1ll << 40;
        *total_out = tot;
    }
}


// row_offset: line of the full frame that local line 0 corresponds to (multi-GPU bands): hashes and visiting keys are
// functions of the GLOBAL pixel index, so that a band decomposition follows the same order as the whole frame
__global__ void k_active_init(const int32_t *__restrict__ nsim, int W, int H, int w, int row_begin, int row_end,
                              float skip_prob, uint32_t seed, int row_offset, uint8_t *__restrict__ state)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    size_t p = (size_t)r * W + c;
    uint8_t s = BCD_ST_NONE;
    if (r >= w && r <= H - 1 - w && c >= w && c <= W - 1 - w && r >= row_begin && r < row_end) {
        if (skip_prob <= 0.f) s = BCD_ST_IN;
        else if (skip_prob >= 1.f) s = BCD_ST_UNDECIDED;
        else s = (bcd_unit_hash((uint32_t)(p + (size_t)row_offset * W), seed) < skip_prob) ? BCD_ST_UNDECIDED : BCD_ST_IN; // never skipped when marked
    }
    state[p] = s;
}

__global__ __launch_bounds__(256) void k_active_round(const uint32_t *__restrict__ mask, const int32_t *__restrict__ nsim,
                                                      uint8_t *state, int W, int H, int b, int words, int min_strong,
                                                      int random_order, uint32_t seed, int row_begin, int row_end, int row_offset,
                                                      int *__restrict__ undecided)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    bool still = false;
    const size_t goff = (size_t)row_offset * W;
    if (c < W && r >= row_begin && r < row_end) {
        size_t p = (size_t)r * W + c;
        if (state[p] == BCD_ST_UNDECIDED) {
            const uint64_t keyp = bcd_order_key((uint32_t)(p + goff), random_order, seed);
            const int side = 2 * b + 1;
            bool any_in = false, all_decided = true;
            for (int j = 0; j < words && !any_in; ++j) {
                uint32_t m = mask[p * words + j];
                while (m) {
                    int bit = __ffs(m) - 1;
                    m &= m - 1;
                    int k = j * 32 + bit;
                    int dl = k / side - b, dc = k % side - b;
                    size_t q = (size_t)(r + dl) * W + (c + dc);
                    if (q == p) continue;
                    if (nsim[q] < min_strong) continue;                 // fallback pixels mark nobody
                    if (bcd_order_key((uint32_t)(q + goff), random_order, seed) > keyp) continue; // visited later
                    uint8_t sq = state[q];
                    if (sq == BCD_ST_IN) { any_in = true; break; }
                    if (sq == BCD_ST_UNDECIDED) all_decided = false;
                    // BCD_ST_OUT / BCD_ST_NONE (outside this band): never processed here -> marks nobody
                }
            }
            if (any_in) state[p] = BCD_ST_OUT;
            else if (all_decided) state[p] = BCD_ST_IN;
            else still = true;
        }
    }
    unsigned long long bal = __ballot(still);
    if (bal && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)bal) - 1)) atomicAdd(counter_line(undecided), __popcll(bal));
}

// ---- dependency lists + probe rounds (search radius 6 or 12) ------------------------------------------------------------
// k_mark_deps, once per marking problem: every undecided pixel extracts dep(p) = similar AND strong AND visited earlier
// into global memory.  The "strong" test is word-parallel: the strong flags of the tile and its halo are bit-packed per
// row in LDS, the window's strong bitmap is 2b+1 shifted row fields, and only the (few) strong similar neighbours are
// walked bit by bit for the key comparison.  Pixels with dep = 0 are local minima of the order: processed.
// k_mark_round, one dependency level per launch: a still undecided pixel probes the states of its dep bits (IN -> p is
// marked; decided and not processed -> bit dropped; undecided -> wait).  A state byte only ever changes from UNDECIDED to its
// final value, so a stale read delays a decision and never changes it.
// the (2B+1)^2 window bitmap of a pixel at tile position (lx, ly) from per-row bitmaps of the tile + halo (bit lc of
// rows[lr] = cell (lr, lc)): window line j is the field of 2B+1 bits starting at column lx of row ly + j
// the WORDS mask / dependency words of one pixel as 16- or 8-byte loads (round 6): the per-word form was WORDS separate 4-byte loads per lane at a lane stride
// of WORDS * 4 bytes -- 20 passes of 64 scattered addresses through the L1 at b = 12, and the kernels that read them wait on exactly those loads
// (profiles/r05_pmc_b12_mask_kernels.txt).  (Global loads need dword alignment only; the records are WORDS * 4 bytes apart.)
template <int WORDS>
__device__ inline void load_pixel_words(const uint32_t *__restrict__ src, uint32_t (&out)[WORDS])
{
    if constexpr (WORDS % 4 == 0) {
#pragma unroll
        for (int q = 0; q < WORDS / 4; ++q) {
            const uint4 v = reinterpret_cast<const uint4 *>(src)[q];
            out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
        }
    } else if constexpr (WORDS % 2 == 0) {
#pragma unroll
        for (int q = 0; q < WORDS / 2; ++q) {
            const uint2 v = reinterpret_cast<const uint2 *>(src)[q];
            out[2 * q] = v.x; out[2 * q + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < WORDS; ++j) out[j] = src[j];
    }
}

template <int B, class RowPtr>
__device__ inline void window_bits(uint32_t (&w)[((2 * B + 1) * (2 * B + 1) + 31) / 32], RowPtr rows, int lx, int ly)
{
    constexpr int side = 2 * B + 1, WORDS = (side * side + 31) / 32;
#pragma unroll
    for (int j = 0; j < WORDS; ++j) w[j] = 0;
#pragma unroll
    for (int j = 0; j < side; ++j) {
        const unsigned long long field = (rows[ly + j] >> lx) & ((1ull << side) - 1);
        const int pos = j * side, word = pos >> 5, sh = pos & 31;
        const unsigned long long v = field << sh;
        w[word] |= (uint32_t)v;
        if (word + 1 < WORDS) w[word + 1] |= (uint32_t)(v >> 32);
    }
}

template <int B>
__global__ __launch_bounds__(256) void k_mark_deps(const uint32_t *__restrict__ mask, const int32_t *__restrict__ nsim,
                                                   uint8_t *state, uint32_t *__restrict__ dep, int W, int H, int min_strong,
                                                   int random_order, uint32_t seed, int row_begin, int row_end, int row_offset,
                                                   int *__restrict__ undecided)
{
    constexpr int side = 2 * B + 1, WORDS = (side * side + 31) / 32, tw = 16 + 2 * B;
    static_assert(tw <= 64 && side <= 32, "row bitmaps are 64-bit");
    // (round 5) "visited earlier" is a comparison of order keys, one per similar neighbour -- ~100 of them per pixel in a flat region, and the
    // wavefront walks the longest of its 64 lists: that loop was the kernel (0.13 ms at 1080p, vector pipe saturated).  The keys' top LV_BITS bits
    // sort the cells of the tile into 32 levels; per level l two row bitmaps of the tile + halo -- cells of a level BELOW l (certainly earlier
    // than a pixel of level l) and cells OF level l (to be compared key by key: 1 in 32) -- turn the test into the same word-parallel window
    // extraction as the strong flags, plus a short loop over the few same-level neighbours.
    constexpr int LV_BITS = 5, NLV = 1 << LV_BITS;
    __shared__ uint32_t s_hash[tw * tw];
    __shared__ unsigned long long s_eq[NLV][tw], s_lt[NLV][tw];
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + lx, r = blockIdx.y * 16 + ly;
    const int r0 = blockIdx.y * 16 - B, c0 = blockIdx.x * 16 - B;
    for (int i = threadIdx.x; i < NLV * tw; i += 256) (&s_eq[0][0])[i] = 0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < tw * tw; i += 256) {
        int lr = i / tw, lc = i - lr * tw, gr = r0 + lr, gc = c0 + lc;
        uint32_t h = 0;
        if (gr >= 0 && gr < H && gc >= 0 && gc < W) {
            size_t q = (size_t)gr * W + gc;
            h = (uint32_t)(bcd_order_key((uint32_t)(q + (size_t)row_offset * W), random_order, seed) >> 32);
            if (nsim[q] >= min_strong) {
                atomicOr(&s_eq[h >> (32 - LV_BITS)][lr], 1ull << lc); // (only strong cells can be dependencies: the level bitmaps hold those)
            }
        }
        s_hash[i] = h;
    }
    __syncthreads();
    if (threadIdx.x < tw) { // strictly-lower-level bitmaps: a running OR over the levels, per tile row
        unsigned long long run = 0ull;
#pragma unroll 8
        for (int l = 0; l < NLV; ++l) { s_lt[l][threadIdx.x] = run; run |= s_eq[l][threadIdx.x]; }
    }
    __syncthreads();
    const bool inside = c < W && r < H;
    const size_t p = inside ? (size_t)r * W + c : 0;
    bool pending = inside && r >= row_begin && r < row_end && state[p] == BCD_ST_UNDECIDED;
    if (pending) {
        const int lp = (ly + B) * tw + lx + B;
        const uint32_t hp = s_hash[lp];
        const int lv = (int)(hp >> (32 - LV_BITS));
        uint32_t lower[WORDS], same[WORDS];
        window_bits<B>(lower, s_lt[lv], lx, ly); // strong cells of a lower level: earlier whatever the rest of the key says
        window_bits<B>(same, s_eq[lv], lx, ly);  // strong cells of the pixel's own level: compared key by key below
        uint32_t d[WORDS];
        bool none = true;
        // window cell k = (dl + B) side + (dc + B) lies at lp0 + k + (tw - side) (k / side) of the hash tile; visited earlier: key = (hash, index),
        // and the index order is the order of k (k < KC = the window's centre)
        constexpr int KC = (side * side - 1) / 2;
        const int lp0 = lp - B * tw - B;
        auto earlier_bit = [&](int k, int bit) __attribute__((always_inline)) -> uint32_t {
            const int rw = side == 13 ? (k * 79) >> 10 : k / side; // k / side
            const uint32_t hq = s_hash[lp0 + k + (tw - side) * rw];
            const uint32_t e = (uint32_t)(hq < hp) | ((uint32_t)(hq == hp) & (uint32_t)(k < KC));
            return e << bit;
        };
        uint32_t simw[WORDS];
        load_pixel_words<WORDS>(mask + p * WORDS, simw);
#pragma unroll
        for (int j = 0; j < WORDS; ++j) {
            const uint32_t sim = simw[j];
            uint32_t m = sim & same[j], keep = sim & lower[j];
            while (m) {
                const int b0 = __ffs(m) - 1;
                m &= m - 1;
                const int b1 = m ? __ffs(m) - 1 : b0; // (a second bit, or the first one again: its result is OR-ed in twice)
                m &= m - 1;                            // (0 & anything == 0)
                keep |= earlier_bit(j * 32 + b0, b0) | earlier_bit(j * 32 + b1, b1);
            }
            d[j] = keep;
            none = none && keep == 0;
        }
        if (none) { state[p] = BCD_ST_IN; pending = false; }
        else {
#pragma unroll
            for (int j = 0; j < WORDS; ++j) dep[p * WORDS + j] = d[j];
        }
    }
    int left = __syncthreads_count(pending);
    if (threadIdx.x == 0 && left) atomicAdd(counter_line(undecided), left);
}

template <int B>
__global__ __launch_bounds__(256) void k_mark_round(const uint32_t *__restrict__ dep, uint8_t *state, int W, int H, int row_begin,
                                                    int row_end, int iters, int *__restrict__ undecided)
{
    constexpr int side = 2 * B + 1, WORDS = (side * side + 31) / 32, tw = 16 + 2 * B;
    static_assert(tw <= 64, "row bitmaps are 64-bit");
    // states of the tile + halo as two bitmaps per row: processed (IN) and undecided
    // (plain LDS arrays: a `volatile unsigned long long *` alias of them is a GENERIC pointer -- until round 6 every probe below was a flat load with
    // system scope and a full wait behind it, 50 of them per iteration through 50 precomputed 64-bit addresses: 148 VGPRs at b = 12.  The barriers
    // between the iterations already keep the compiler from carrying bitmap values across them.)
    __shared__ unsigned long long s_in_[tw], s_und_[tw];
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + lx, r = blockIdx.y * 16 + ly;
    const bool inside = c < W && r < H;
    const size_t p = inside ? (size_t)r * W + c : 0;
    bool pending = inside && r >= row_begin && r < row_end && state[p] == BCD_ST_UNDECIDED;
    if (!__syncthreads_or(pending)) return; // nothing left to decide in this tile
    // (round 5) the two row bitmaps come from ballots, not from LDS atomics: a wavefront takes every fourth tile row, lane c loads the state of
    // column c of that row, and the two ballots ARE the row's bitmaps (the 28 lanes of a row used to hit one 64-bit word with an atomic each)
    const int r0 = blockIdx.y * 16 - B, c0 = blockIdx.x * 16 - B;
    // every load of the tile is issued before the first one is used (the halo lines and the dependency words: one round trip to memory, not one per line)
    uint32_t dword[WORDS];
#pragma unroll
    for (int j = 0; j < WORDS; ++j) dword[j] = 0u;
    if (pending) load_pixel_words<WORDS>(dep + p * WORDS, dword);
    {
        constexpr int NL = (tw + 3) / 4;
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        uint8_t hv[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int lr = wave + 4 * u, gr = r0 + lr, gc = c0 + lane;
            hv[u] = BCD_ST_NONE;
            if (lr < tw && lane < tw && gr >= 0 && gr < H && gc >= 0 && gc < W) hv[u] = state[(size_t)gr * W + gc];
        }
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int lr = wave + 4 * u;
            const unsigned long long bin = __ballot(hv[u] == BCD_ST_IN), bund = __ballot(hv[u] == BCD_ST_UNDECIDED);
            if (lane == 0 && lr < tw) { s_in_[lr] = bin; s_und_[lr] = bund; }
        }
    }
    // the pixel's dependency bits, one window line per register (bit dc + B of drow[j] = window cell (j - B, dc)): a probe is then, per window line,
    // one shifted row bitmap AND-ed with that register -- no 169-bit window is assembled per iteration any more (round 5: half the instructions)
    uint32_t drow[side];
    {
        uint32_t d[WORDS + 1];
#pragma unroll
        for (int j = 0; j < WORDS; ++j) d[j] = dword[j];
        d[WORDS] = 0u;
#pragma unroll
        for (int j = 0; j < side; ++j) {
            const int pos = j * side, word = pos >> 5, sh = pos & 31;
            const unsigned long long two = ((unsigned long long)d[word + 1] << 32) | d[word];
            drow[j] = (uint32_t)(two >> sh) & ((1u << side) - 1u);
        }
    }
    __syncthreads();
    // decisions made inside the tile are visible to the tile at once (LDS), the halo is as of the launch: one launch resolves
    // the chains that stay inside a tile, the next one sees the neighbours' results.  p is marked iff a dependency is IN, and waits iff one
    // is still undecided.
    for (int it = 0; it < iters; ++it) {
        uint8_t v = BCD_ST_UNDECIDED;
        if (pending) {
            uint32_t hit_in = 0u, hit_und = 0u;
#pragma unroll
            for (int j = 0; j < side; ++j) {
                hit_in |= (uint32_t)(s_in_[ly + j] >> lx) & drow[j];
                hit_und |= (uint32_t)(s_und_[ly + j] >> lx) & drow[j];
            }
            v = hit_in != 0u ? BCD_ST_OUT : (hit_und != 0u ? BCD_ST_UNDECIDED : BCD_ST_IN);
        }
        __syncthreads(); // every probe of this iteration has read the bitmaps
        const bool changed = pending && v != BCD_ST_UNDECIDED;
        if (changed) { state[p] = v; pending = false; }
        {
            // a wavefront owns four tile rows (16 lanes each): the decisions of a row go into its two bitmap words with one plain update by one lane
            // (nobody else writes these words; the halo bits of a row stay as loaded)
            const unsigned long long bch = __ballot(changed), bin = __ballot(changed && v == BCD_ST_IN);
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            if (lane < 4 && ((bch >> (16 * lane)) & 0xffffull) != 0ull) {
                const int row = 4 * wave + lane + B;
                const unsigned long long ch = ((bch >> (16 * lane)) & 0xffffull) << B, in = ((bin >> (16 * lane)) & 0xffffull) << B;
                s_und_[row] &= ~ch;
                s_in_[row] |= in;
            }
        }
        if (!__syncthreads_or(changed)) break;
    }
    int left = __syncthreads_count(pending);
    if (threadIdx.x == 0 && left) atomicAdd(counter_line(undecided), left);
}

// compact lists of processed pixels: strong (full Bayesian path) and weak (fallback path); one atomic per counter
// per workgroup, wavefront offsets through LDS
__global__ __launch_bounds__(1024) void k_active_lists(const uint8_t *__restrict__ state, const int32_t *__restrict__ nsim,
                                                       int64_t p_begin, int64_t npix /* pixels [p_begin, npix) */, int min_strong,
                                                       int32_t *__restrict__ strong_list, int32_t *__restrict__ weak_list,
                                                       int32_t *__restrict__ counters /* [0]=strong [1]=weak, [2..3] sum |S| (64-bit) */,
                                                       const long long *__restrict__ skip_if /* optional: the launch does nothing when this word is not zero */)
{
    // (round 6) launched behind a marking batch whose outcome the host has not seen yet: undecided pixels left (or masks to recompute) -> no lists,
    // and the estimate kernels that follow find none
    if (skip_if && *skip_if != 0) return;
    // LISTS_PER_THREAD x 1024 pixels per workgroup and ONE set of atomics on the list counters for all of them (same-address atomics
    // are served one at a time: with 1024 pixels per workgroup the 2 025 x 3 atomics of a 1080p scale were most of this kernel's time)
    __shared__ int ws[LISTS_PER_THREAD][16], ww[LISTS_PER_THREAD][16], base[2];
    __shared__ long long wt[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t p0 = p_begin + (int64_t)blockIdx.x * (1024 * LISTS_PER_THREAD) + threadIdx.x;
    unsigned long long bs[LISTS_PER_THREAD], bw[LISTS_PER_THREAD];
    long long tot = 0;
#pragma unroll
    for (int u = 0; u < LISTS_PER_THREAD; ++u) {
        const int64_t p = p0 + u * 1024;
        const bool in = p < npix && state[p] == BCD_ST_IN;
        const int n = in ? nsim[p] : 0;
        bs[u] = __ballot(in && n >= min_strong);
        bw[u] = __ballot(in && n < min_strong);
        tot += n;
    }
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off);
    if (lane == 0) {
#pragma unroll
        for (int u = 0; u < LISTS_PER_THREAD; ++u) { ws[u][wave] = __popcll(bs[u]); ww[u][wave] = __popcll(bw[u]); }
        wt[wave] = tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0, w = 0;
        long long t = 0;
        for (int u = 0; u < LISTS_PER_THREAD; ++u)
            for (int i = 0; i < 16; ++i) { int a = ws[u][i], bq = ww[u][i]; ws[u][i] = s; ww[u][i] = w; s += a; w += bq; }
        for (int i = 0; i < 16; ++i) t += wt[i];
        base[0] = s ? atomicAdd(&counters[0], s) : 0;
        base[1] = w ? atomicAdd(&counters[1], w) : 0;
        if (t) atomicAdd(reinterpret_cast<unsigned long long *>(&counters[2]), (unsigned long long)t);
    }
    __syncthreads();
    const unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int u = 0; u < LISTS_PER_THREAD; ++u) {
        const int64_t p = p0 + u * 1024;
        if ((bs[u] >> lane) & 1ull) strong_list[base[0] + ws[u][wave] + __popcll(bs[u] & lower)] = (int32_t)p;
        if ((bw[u] >> lane) & 1ull) weak_list[base[1] + ww[u][wave] + __popcll(bw[u] & lower)] = (int32_t)p;
    }
}

} // namespace

hipError_t bcd_launch_active_init(const int32_t *nsim, int W, int H, int w, int row_begin, int row_end, float skip_prob,
                                  uint32_t seed, int row_offset, uint8_t *state, hipStream_t st)
{
    hipLaunchKernelGGL(k_active_init, dim3((W + 255) / 256, H), dim3(256), 0, st, nsim, W, H, w, row_begin, row_end, skip_prob, seed, row_offset, state);
    return hipGetLastError();
}

hipError_t bcd_launch_active_round(const uint32_t *mask, const int32_t *nsim, uint8_t *state, int W, int H, int b,
                                   int min_strong, int random_order, uint32_t seed, int row_begin, int row_end, int row_offset,
                                   int *undecided, hipStream_t st)
{
    int side = 2 * b + 1, words = (side * side + 31) / 32;
    hipLaunchKernelGGL(k_active_round, dim3((W + 255) / 256, H), dim3(256), 0, st, mask, nsim, state, W, H, b, words, min_strong,
                       random_order, seed, row_begin, row_end, row_offset, undecided);
    return hipGetLastError();
}

// pixels [p_begin, p_end) of the state image (the owned lines of a row band: halo lines are processed by their owner); skip_if: see the kernel
hipError_t bcd_launch_active_lists(const uint8_t *state, const int32_t *nsim, int64_t p_begin, int64_t p_end, int min_strong,
                                   int32_t *strong_list, int32_t *weak_list, int32_t *counters, hipStream_t st, const long long *skip_if)
{
    if (p_end <= p_begin) return hipSuccess;
    const int64_t n = p_end - p_begin;
    hipLaunchKernelGGL(k_active_lists, dim3((unsigned)((n + 1024 * LISTS_PER_THREAD - 1) / (1024 * LISTS_PER_THREAD))), dim3(1024), 0, st, state, nsim, p_begin, p_end, min_strong,
                       strong_list, weak_list, counters, skip_if);
    return hipGetLastError();
}

hipError_t bcd_launch_sum_counter_lines(const int *lines, int rounds, int *out, hipStream_t st, long long *total_out, const int *flags, int border_capacity)
{
    if (rounds <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_sum_counter_lines, dim3(rounds), dim3(64), 0, st, lines, rounds, out, total_out, flags, border_capacity);
    return hipGetLastError();
}

// marking with stored dependency lists (b = 6 or 12): dependency extraction ...
hipError_t bcd_launch_mark_deps(const uint32_t *mask, const int32_t *nsim, uint8_t *state, uint32_t *dep, int W, int H, int b,
                                int min_strong, int random_order, uint32_t seed, int row_begin, int row_end, int row_offset,
                                int *undecided, hipStream_t st)
{
    dim3 grid((W + 15) / 16, (H + 15) / 16), block(256);
    if (b == 6)
        hipLaunchKernelGGL(k_mark_deps<6>, grid, block, 0, st, mask, nsim, state, dep, W, H, min_strong, random_order, seed, row_begin, row_end, row_offset, undecided);
    else if (b == 12)
        hipLaunchKernelGGL(k_mark_deps<12>, grid, block, 0, st, mask, nsim, state, dep, W, H, min_strong, random_order, seed, row_begin, row_end, row_offset, undecided);
    else
        return hipErrorNotSupported;
    return hipGetLastError();
}

// ... and one round: in-tile fixed point against the halo states of the launch
hipError_t bcd_launch_mark_round(const uint32_t *dep, uint8_t *state, int W, int H, int b, int row_begin, int row_end, int iters, int *undecided,
                                 hipStream_t st)
{
    dim3 grid((W + 15) / 16, (H + 15) / 16), block(256);
    if (b == 6)
        hipLaunchKernelGGL(k_mark_round<6>, grid, block, 0, st, dep, state, W, H, row_begin, row_end, iters, undecided);
    else if (b == 12)
        hipLaunchKernelGGL(k_mark_round<12>, grid, block, 0, st, dep, state, W, H, row_begin, row_end, iters, undecided);
    else
        return hipErrorNotSupported;
    return hipGetLastError();
}

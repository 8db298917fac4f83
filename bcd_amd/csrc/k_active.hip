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
    if (bal && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)bal) - 1)) atomicAdd(undecided, __popcll(bal));
}

// Tile-iterated round: a 16x16 tile of pixels per workgroup.  On entry every undecided pixel extracts its dependency
// bits (similar AND visited earlier AND strong) into registers; the tile then iterates in place -- updates made by the
// same workgroup are visible through the CU's write-through L1 (volatile loads), updates of other workgroups may be
// seen late, which only delays a decision (the iteration is monotone and its fixed point unique).  One launch
// resolves every dependency chain that stays inside a tile, so a frame needs 2-3 launches instead of one per level.
template <int B>
__global__ __launch_bounds__(256) void k_active_tile(const uint32_t *__restrict__ mask, const int32_t *__restrict__ nsim,
                                                     uint8_t *state, int W, int H, int min_strong, int random_order,
                                                     uint32_t seed, int inner_iters, int first_launch, int row_begin, int row_end,
                                                     int row_offset, int *__restrict__ undecided)
{
    constexpr int b = B, WORDS = ((2 * B + 1) * (2 * B + 1) + 31) / 32; // compile-time window: k / side is a multiply-shift
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), r = blockIdx.y * 16 + (threadIdx.x >> 4);
    constexpr int side = 2 * b + 1;
    volatile uint8_t *vstate = state;
    // stage, for the tile and its b-pixel halo, the high word of the visiting key and the "strong" flag (|S| >= 3P+1):
    // the dependency extraction below then probes LDS instead of chasing dependent global loads
    extern __shared__ uint32_t lds_u[];
    constexpr int tw = 16 + 2 * b;
    uint32_t *s_hash = lds_u;
    uint8_t *s_strong = reinterpret_cast<uint8_t *>(lds_u + tw * tw);
    volatile uint8_t *s_state = s_strong + tw * tw; // states of the tile + halo, refreshed from global every iteration
    const int r0 = blockIdx.y * 16 - b, c0 = blockIdx.x * 16 - b;
    for (int i = threadIdx.x; i < tw * tw; i += 256) {
        int lr = i / tw, lc = i - lr * tw, gr = r0 + lr, gc = c0 + lc;
        uint32_t h = 0;
        uint8_t st = 0, sv = BCD_ST_NONE;
        if (gr >= 0 && gr < H && gc >= 0 && gc < W) {
            size_t q = (size_t)gr * W + gc;
            h = (uint32_t)(bcd_order_key((uint32_t)(q + (size_t)row_offset * W), random_order, seed) >> 32);
            st = nsim[q] >= min_strong;
            sv = state[q];
        }
        s_hash[i] = h;
        s_strong[i] = st;
        s_state[i] = sv;
    }
    __syncthreads();
    const bool inside = c < W && r < H;
    const size_t p = inside ? (size_t)r * W + c : 0;
    // only lines [row_begin, row_end) are decided here; undecided pixels outside (the halo of a band) belong to a neighbour
    bool pending = inside && r >= row_begin && r < row_end && state[p] == BCD_ST_UNDECIDED;
    uint32_t dep[WORDS];
    if (pending) {
        const int lp = ((threadIdx.x >> 4) + b) * tw + (threadIdx.x & 15) + b;
        const uint32_t hp = s_hash[lp];
#pragma unroll
        for (int j = 0; j < WORDS; ++j) {
            uint32_t m = mask[p * WORDS + j], keep = 0;
            while (m) {
                int bit = __ffs(m) - 1;
                m &= m - 1;
                int k = j * 32 + bit;
                int dl = k / side - b, dc = k - (k / side) * side - b;
                int lq = lp + dl * tw + dc;
                uint32_t hq = s_hash[lq];
                // strong, and visited earlier: key = (hash, index), index order == (dl, dc) lexicographic order
                bool earlier = hq < hp || (hq == hp && (dl < 0 || (dl == 0 && dc < 0)));
                if (s_strong[lq] && earlier) keep |= 1u << bit;
            }
            dep[j] = keep;
        }
    } else {
#pragma unroll
        for (int j = 0; j < WORDS; ++j) dep[j] = 0;
    }
    const int lp_own = ((threadIdx.x >> 4) + b) * tw + (threadIdx.x & 15) + b;
    for (int it = 0; it < inner_iters; ++it) {
        bool changed = false;
        if (pending) {
            bool any_in = false, wait = false;
            // very first pass of a scale: every neighbour is still undecided, so only pixels without any earlier strong
            // similar neighbour (local minima of the visiting order) can be decided -- no need to probe states
            const bool skip_probe = first_launch && it == 0;
#pragma unroll
            for (int j = 0; j < WORDS; ++j) {
                uint32_t m = dep[j];
                if (skip_probe) { wait = wait || m != 0; continue; }
                while (m && !any_in) {
                    int bit = __ffs(m) - 1;
                    m &= m - 1;
                    int k = j * 32 + bit;
                    int dl = k / side - b, dc = k - (k / side) * side - b;
                    uint8_t sq = s_state[lp_own + dl * tw + dc];
                    if (sq == BCD_ST_IN) any_in = true;
                    else if (sq == BCD_ST_UNDECIDED) wait = true;
                    else dep[j] &= ~(1u << bit); // decided and not processed: can never mark p
                }
            }
            if (any_in) { state[p] = BCD_ST_OUT; s_state[lp_own] = BCD_ST_OUT; pending = false; changed = true; }
            else if (!wait) { state[p] = BCD_ST_IN; s_state[lp_own] = BCD_ST_IN; pending = false; changed = true; }
        }
        // halo cells are owned by other workgroups running concurrently: re-read them (late values only delay decisions)
        for (int i = threadIdx.x; i < tw * tw; i += 256) {
            int lr = i / tw, lc = i - lr * tw;
            if (lr >= b && lr < b + 16 && lc >= b && lc < b + 16) continue;
            int gr = r0 + lr, gc = c0 + lc;
            if (gr >= 0 && gr < H && gc >= 0 && gc < W) {
                uint8_t old = s_state[i];
                if (old == BCD_ST_UNDECIDED) {
                    uint8_t nv = vstate[(size_t)gr * W + gc];
                    if (nv != old) { s_state[i] = nv; changed = true; }
                }
            }
        }
        if (!__syncthreads_or(changed)) break;
    }
    int left = __syncthreads_count(pending);
    if (threadIdx.x == 0 && left) atomicAdd(undecided, left);
}

// compact lists of processed pixels: strong (full Bayesian path) and weak (fallback path); one atomic per counter
// per workgroup, wavefront offsets through LDS
__global__ __launch_bounds__(1024) void k_active_lists(const uint8_t *__restrict__ state, const int32_t *__restrict__ nsim,
                                                       int64_t npix, int min_strong,
                                                       int32_t *__restrict__ strong_list, int32_t *__restrict__ weak_list,
                                                       int32_t *__restrict__ counters /* [0]=strong [1]=weak, [2..3] sum |S| (64-bit) */)
{
    __shared__ int ws[16], ww[16], wt[16], base[2];
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool in = p < npix && state[p] == BCD_ST_IN;
    int n = in ? nsim[p] : 0;
    bool strong = in && n >= min_strong, weak = in && n < min_strong;
    unsigned long long bs = __ballot(strong), bw = __ballot(weak);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tot = n;
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off);
    if (lane == 0) { ws[wave] = __popcll(bs); ww[wave] = __popcll(bw); wt[wave] = tot; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0, w = 0;
        long long t = 0;
        for (int i = 0; i < 16; ++i) { int a = ws[i], bq = ww[i]; ws[i] = s; ww[i] = w; s += a; w += bq; t += wt[i]; }
        base[0] = s ? atomicAdd(&counters[0], s) : 0;
        base[1] = w ? atomicAdd(&counters[1], w) : 0;
        if (t) atomicAdd(reinterpret_cast<unsigned long long *>(&counters[2]), (unsigned long long)t);
    }
    __syncthreads();
    unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (strong) strong_list[base[0] + ws[wave] + __popcll(bs & lower)] = (int32_t)p;
    if (weak) weak_list[base[1] + ww[wave] + __popcll(bw & lower)] = (int32_t)p;
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

hipError_t bcd_launch_active_lists(const uint8_t *state, const int32_t *nsim, int64_t npix, int min_strong,
                                   int32_t *strong_list, int32_t *weak_list, int32_t *counters, hipStream_t st)
{
    hipLaunchKernelGGL(k_active_lists, dim3((unsigned)((npix + 1023) / 1024)), dim3(1024), 0, st, state, nsim, npix, min_strong,
                       strong_list, weak_list, counters);
    return hipGetLastError();
}

hipError_t bcd_launch_active_tile(const uint32_t *mask, const int32_t *nsim, uint8_t *state, int W, int H, int b,
                                  int min_strong, int random_order, uint32_t seed, int inner_iters, int first_launch, int row_begin,
                                  int row_end, int row_offset, int *undecided, hipStream_t st)
{
    int side = 2 * b + 1, words = (side * side + 31) / 32;
    dim3 grid((W + 15) / 16, (H + 15) / 16), block(256);
    const int tw = 16 + 2 * b;
    const size_t lds = (size_t)tw * tw * 4 + 2 * (((size_t)tw * tw + 3) & ~(size_t)3);
    (void)words;
    if (b == 6)
        hipLaunchKernelGGL(k_active_tile<6>, grid, block, lds, st, mask, nsim, state, W, H, min_strong, random_order, seed, inner_iters, first_launch, row_begin, row_end, row_offset, undecided);
    else if (b == 12)
        hipLaunchKernelGGL(k_active_tile<12>, grid, block, lds, st, mask, nsim, state, W, H, min_strong, random_order, seed, inner_iters, first_launch, row_begin, row_end, row_offset, undecided);
    else
        return hipErrorNotSupported;
    return hipGetLastError();
}

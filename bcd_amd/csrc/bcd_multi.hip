// bcd_multi.hip -- one frame over several GPUs of a node, native driver behind the C ABI (bcd_hip_multi_*).
//
// The reference is single-process and single-device; this is the build's own decomposition, the same one
// bcd_amd/tiling.py describes and tests/ checks against the single-GPU frame:
//   * the frame is cut into horizontal bands of main pixels whose boundaries are multiples of 2^(S-1) lines, so every pyramid
//     level of a band is built from exactly the 2x2 blocks the full frame would use (MultiscaleDenoiser.cpp:256-266);
//   * a rank (one host thread + one device) holds, per scale, its band plus (b + w) halo lines of INPUT on each interior side
//     -- no input exchange at run time -- and rebuilds the pyramid for its band;
//   * per scale (one host thread, HIP stream and engine context each, like the scales of bcd_hip_denoise): similar-patch
//     masks, the marking fixed point in the visiting order of the WHOLE frame (keys are functions of the global pixel index;
//     the |S| and the states of the b boundary lines travel between marking batches, one all-reduced integer ends the
//     iteration), the Bayesian estimate, then the (b + w) accumulator halo lines travel and the band is finalised;
//   * two lines of every unmerged output and one line of every merged output travel for the merges at the band edges.
// The result is the single-GPU frame to fp32 round-off for every -m / -r setting.
//
// Transport: RCCL point-to-point (ncclSend / ncclRecv grouped per neighbour exchange, one communicator and stream per scale;
// xGMI links between neighbouring devices) when the ranks sit on distinct devices; ranks that share a device (virtual split,
// tests on a one-GPU box) exchange through device-to-device copies and host barriers.
//
// Issue order of the RCCL operations (CommGate): a communication kernel spins on the device until its peers' kernel of the same
// operation runs, and HIP multiplexes the streams of a process onto a few hardware queues, so two ranks that enqueue the
// operations of two communicators in opposite orders can block each other for good (each queue's head waits for a kernel stuck
// behind the other queue's head).  The scales run on concurrent host threads, so their relative order is a matter of timing;
// the gate makes it a rule instead: a phase-major sequence -- the marking operations of scale S-1, ..., of scale 0, then the
// accumulator exchanges of scale S-1, ..., of scale 0, then the merges' exchanges (struct CommGate).  Every rank then enqueues the
// same global sequence, and operation k only ever waits for operations < k and for compute kernels.  A scale waits for the coarser
// scales' marking, never for their estimates, so the three chains of a band overlap as on a single GPU.
#include "../../include/bcd_hip.h"
#include "bcd_common.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int MAX_RANKS = 64;
constexpr int MAX_S = 8;

// ---- partition bookkeeping (bcd_amd/tiling.py: BandGeometry) -------------------------------------------------------
struct ScaleBand {
    int W, H;       // full-frame size at this scale
    int own0, own1; // owned lines (global, this scale)
    int loc0, loc1; // locally held lines
};

struct Geometry {
    int W, H, S, b, w, world, halo, align;
    std::vector<int> bounds;
    bool init(int W_, int H_, int S_, int b_, int w_, int world_, std::string &err, bool loopback_ = false)
    {
        W = W_; H = H_; S = S_; b = b_; w = w_; world = world_;
        halo = b + w;
        align = 1 << (S - 1);
        const int units = H / align;
        if (units < world) { err = "frame too small to be split into that many bands"; return false; }
        bounds.clear();
        for (int r = 0; r < world; ++r) bounds.push_back((int)((long long)units * r / world) * align);
        bounds.push_back(H);
        for (int r = 0; r < world; ++r) {
            int o0, o1;
            owned(r, S - 1, o0, o1);
            if ((world > 1 || loopback_) && o1 - o0 < std::max(halo, 2)) { err = "a band owns fewer lines than the halo at the coarsest scale: use fewer devices"; return false; }
        }
        return true;
    }
    void owned(int rank, int s, int &o0, int &o1) const
    {
        o0 = bounds[rank] >> s;
        o1 = rank == world - 1 ? (H >> s) : bounds[rank + 1] >> s;
    }
    void bands(int rank, ScaleBand *out) const
    {
        for (int s = S - 1; s >= 0; --s) {
            int o0, o1;
            owned(rank, s, o0, o1);
            const int Hs = H >> s;
            int l0 = rank > 0 ? std::max(0, o0 - halo) : 0;
            int l1 = rank < world - 1 ? std::min(Hs, o1 + halo) : Hs;
            if (s < S - 1) {
                l0 = std::min(l0, 2 * out[s + 1].loc0);
                l1 = std::max(l1, std::min(Hs, 2 * out[s + 1].loc1));
                if (rank == world - 1) l1 = Hs;
            }
            if (l0 % 2) --l0; // merges map local line l / 2 to the coarser level
            out[s] = ScaleBand{ W >> s, Hs, o0, o1, l0, l1 };
        }
    }
};

// ---- reusable host barrier with an abort flag ------------------------------------------------------------------------
struct HostBarrier {
    std::mutex m;
    std::condition_variable cv;
    int count = 0, generation = 0, parties = 1;
    std::atomic<bool> *abort_flag = nullptr;
    bool wait()
    {
        std::unique_lock<std::mutex> lk(m);
        const int gen = generation;
        if (++count == parties) {
            count = 0;
            ++generation;
            cv.notify_all();
            return !abort_flag->load();
        }
        while (gen == generation && !abort_flag->load()) cv.wait_for(lk, std::chrono::milliseconds(50));
        return !abort_flag->load();
    }
    // after a failed frame some ranks have arrived and others never will: forget the arrivals (no thread is waiting when this runs)
    void reset()
    {
        std::lock_guard<std::mutex> lk(m);
        count = 0;
        ++generation;
    }
};

// ---- CommGate: the order in which the scale threads of a rank enqueue their communication operations -------------------------
// Every rank must enqueue the same global sequence (header).  The sequence is PHASE-major (round 4): phase 1 = what the marking needs, phase 2 = the
// accumulator halos, so a scale waits for the COARSER scales' marking before its own, never for their estimates (until round 4 the sequence was
// scale-major: the tails of the three scales of a band ran one after the other, 3.5 ms for a band whose kernels take 2.7 ms,
// profiles/r04_band_timeline_*.txt).  Round 6: phase 1 is cut in two.
//     P1(S-1) ... P1(0)    R(S-1) ... R(0)    P2(S-1) ... P2(0)    merges
// P1(s) = the operations of the scale's FIRST marking batch (|S| of the boundary lines, the boundary states, the all-reduced count): a fixed number of
// operations, so the scale opens the gate for the next finer one as soon as it has ENQUEUED them -- it used to open it when its marking was complete, which
// it only knows after the last all-reduce has come back to the host: the three marking phases of a band ran one after the other in TIME, each behind a
// host round trip and the wake-up of the next thread, and the finest scale sat 0.16 ms behind its masks waiting for its turn
// (profiles/r06_band_timeline_before_gate_split.txt).  R(s) = whatever the scale's marking needs beyond its first batch (further batches, a restart
// after a rank recomputed its masks): a data-dependent number of operations, normally none, but the same number on every rank -- the all-reduce decides
// it -- and enqueued only after every scale's P1 and the coarser scales' R, so the global sequence stays one sequence.  P2 follows every scale's R.
struct CommGate {
    std::mutex m;
    std::condition_variable cv;
    unsigned done1 = 0, doner = 0, done2 = 0; // bit s: scale s has enqueued its last operation of P1 / R / P2 of this frame
    std::atomic<bool> *abort_flag = nullptr;
    void reset() { std::lock_guard<std::mutex> lk(m); done1 = doner = done2 = 0; }
    void set(unsigned &bits, int s)
    {
        { std::lock_guard<std::mutex> lk(m); bits |= 1u << s; }
        cv.notify_all();
    }
    void finish1(int s) { set(done1, s); }
    void finish_r(int s) { set(doner, s); }
    void finish2(int s) { set(done2, s); }
    void finish(int s) // the scale leaves (normally or not): nobody waits for it any longer
    {
        { std::lock_guard<std::mutex> lk(m); done1 |= 1u << s; doner |= 1u << s; done2 |= 1u << s; }
        cv.notify_all();
    }
    static unsigned coarser(int s, int S) { return ((1u << S) - 1u) & ~((2u << s) - 1u); }
    template <class Pred> bool wait(Pred ready)
    {
        std::unique_lock<std::mutex> lk(m);
        while (!ready() && !abort_flag->load()) cv.wait_for(lk, std::chrono::milliseconds(50));
        return !abort_flag->load();
    }
    bool wait1(int s, int S) // before the first P1 operation of scale s
    {
        const unsigned need = coarser(s, S);
        return wait([&]() { return (done1 & need) == need; });
    }
    bool wait_r(int s, int S) // before the first R operation of scale s
    {
        const unsigned all = (1u << S) - 1u, need = coarser(s, S);
        return wait([&]() { return (done1 & all) == all && (doner & need) == need; });
    }
    bool wait_all_p2(int S) // before the merges' operations
    {
        const unsigned all = (1u << S) - 1u;
        return wait([&]() { return (done2 & all) == all; });
    }
    bool wait2(int s, int S) // before the P2 operations of scale s
    {
        const unsigned all = (1u << S) - 1u, need = coarser(s, S);
        return wait([&]() { return (doner & all) == all && (done2 & need) == need; });
    }
};

struct DBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int device = 0;
    bool ensure(size_t n)
    {
        if (bytes >= n && p) return true;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        if (hipMalloc(&p, n + 256) != hipSuccess) return false;
        bytes = n + 256;
        return true;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

} // namespace

namespace { struct Watchdog; }

struct bcd_hip_multi {
    int n = 0;
    int devices[MAX_RANKS];
    bool use_rccl = false;
    std::mutex err_mutex;
    std::string err;
    std::atomic<bool> abort_flag{ false };
    // per rank: one context per scale + one for the pyramid / merges; channel c = scale (c < S) or S (the tail)
    bcd_hip_ctx *ctx[MAX_RANKS][MAX_S + 1];
    hipStream_t stream[MAX_RANKS][MAX_S + 1];
    ncclComm_t comm[MAX_S + 1][MAX_RANKS];
    bool comm_ready[MAX_S + 1];
    HostBarrier barrier[MAX_S + 1];
    CommGate gate[MAX_RANKS];
    bool ordered = false; // CommGate in force: always with RCCL; BCD_HIP_MULTI_ORDERED=1 turns it on for the in-process transport (tests)
    // in-process transport: what every rank offers its neighbours in the current exchange, per channel
    struct Offer { const void *up = nullptr, *down = nullptr; long long value = 0; };
    Offer offer[MAX_S + 1][MAX_RANKS];
    // grow-only device buffers: [rank][scale][kind]
    enum { IN_COL, IN_NS, IN_HIST, IN_COV, OUT, PIXCOV, MASK, NSIM, STATE, SUM, CNT, RX_UP_S, RX_UP_C, RX_DN_S, RX_DN_C, NBUF };
    DBuf buf[MAX_RANKS][MAX_S][NBUF];
    long long *d_red[MAX_RANKS][MAX_S + 1]; // all-reduce scratch (RCCL transport)
    long long *h_red[MAX_RANKS][MAX_S + 1]; // ... and where its result lands (pinned)
    hipEvent_t ev_level[MAX_RANKS][MAX_S]; // pyramid level s of the rank is complete (recorded on the rank's tail stream)
    hipEvent_t ev_tail[MAX_RANKS];         // the finest scale's last kernel of the frame (recorded on its stream; the merges wait for it)
    bcd_hip_multi_stats stats;
    // communication trace of the last frame (bcd_hip_multi_set_comm_trace): per rank, in the order the rank ENQUEUED its operations,
    // (channel, kind, bytes to / from the rank above, bytes to / from the rank below).  Lets a test check on one GPU what decides
    // whether the RCCL transport can deadlock: that all ranks enqueue the same sequence and that neighbours agree on every size.
    bool trace_on = false;
    std::mutex trace_mutex[MAX_RANKS];
    std::vector<int64_t> trace[MAX_RANKS];
    // one-process-per-GPU use (bcd_hip_multi_create_rank): only `local_rank` lives in this process; its communicators are built
    // with ncclCommInitRank from the ids all processes share
    int local_rank = -1;
    std::vector<ncclUniqueId> ids;
    struct Frame { int W = 0, H = 0, D = 0, S = 0; bcd_hip_params prm; bool set = false; } frame;
    // failure handling of the RCCL transport: the first fail() aborts every local communicator (ncclCommAbort: their device kernels
    // stop waiting for peers, so the host threads blocked in stream synchronisations return); prepare() rebuilds them.  A watchdog
    // thread per frame turns "no progress for frame_timeout_s" into such a failure, which is what ends a frame whose PEER (another
    // process or device) has died.
    std::mutex comm_mutex;
    // rank threads hold comm_rw shared while they ENQUEUE on a communicator (group start .. group end, the all-reduce call); an abort takes it
    // exclusively, so ncclCommAbort -- which frees the communicator -- never runs while a thread is inside an RCCL call on it
    std::shared_timed_mutex comm_rw;
    // ... but a rank thread can be stuck INSIDE such a call (blocking communicators: ncclGroupEnd / the first send to a peer that has died
    // never returns).  The abort therefore waits for the section only this long and then proceeds anyway: ncclCommAbort is what releases a
    // blocked call (round-4 ADVICE: with an unbounded wait the watchdog could no longer end a frame whose peer is gone).
    int abort_wait_ms = 5000;                 // BCD_HIP_MULTI_ABORT_WAIT_MS
    std::atomic<bool> abort_forced{ false };  // the last abort did not get the section to itself (diagnostics / self-test; written by the aborting thread, read by others)
    bool comm_aborted = false;
    // loopback (bcd_hip_multi_set_loopback; one rank, tests on a one-GPU box): the rank is its own neighbour on both sides -- every exchange and
    // all-reduce of a frame is enqueued on real RCCL communicators (ncclCommInitRank with n = 1, grouped self send / recv), in the order and with
    // the sizes a band inside a larger world would use; what is received goes to scratch, so the frame is still the single-GPU frame
    bool loopback = false;
    DBuf loop_rx[MAX_S + 1][2];
    Watchdog *watchdog = nullptr;             // created with the first RCCL frame, joined by bcd_hip_multi_destroy
    int frame_timeout_ms = 600 * 1000;        // BCD_HIP_MULTI_TIMEOUT_S / bcd_hip_multi_set_frame_timeout
    // progress reporting (IDenoiser::setProgressCallback): every (rank, scale) adds its owned pixels twice, like bcd_hip_denoise
    bcd_hip_progress_fn progress_fn = nullptr;
    void *progress_user = nullptr;
    std::mutex progress_mutex;
    double progress_done = 0.0, progress_total = 0.0;
};

namespace {

// which RCCL serves this library: version of the loaded library and the file ncclCommInitRank was resolved from.  The process may hold another
// copy (PyTorch bundles one): libbcd_hip.so links librccl.so.1 by SONAME, so the dynamic loader gives it the copy that is already mapped under
// that SONAME, if any -- this string is how a run says which one it got (bench.py prints it next to the copies mapped into the process).
std::string rccl_identity()
{
    int v = 0;
    (void)ncclGetVersion(&v);
    Dl_info info;
    memset(&info, 0, sizeof(info));
    const char *path = (dladdr(reinterpret_cast<const void *>(&ncclCommInitRank), &info) != 0 && info.dli_fname) ? info.dli_fname : "?";
    return "rccl version " + std::to_string(v) + " from " + path;
}

// RCCL transport: stop the communication kernels of every local communicator (idempotent; prepare() creates new ones)
void abort_comms(bcd_hip_multi *m)
{
    if (!m->use_rccl) return;
    std::unique_lock<std::shared_timed_mutex> excl(m->comm_rw, std::defer_lock);
    // normally no rank thread is inside an RCCL call on these communicators when they are freed; a thread that is STUCK in one is released by the abort itself
    // A forced abort (the wait ran out) frees communicators a rank thread may still be inside: right for a thread that is STUCK in an RCCL call (the abort
    // is what releases it), a residual use-after-free risk for one that is merely slow.  ncclCommAbort cannot be split into "stop" and "free", so the
    // bound is a knob (BCD_HIP_MULTI_ABORT_WAIT_MS, default 5 s -- three orders of magnitude above a healthy enqueue section); INTEGRATION.md says so.
    m->abort_forced.store(!excl.try_lock_for(std::chrono::milliseconds(std::max(0, m->abort_wait_ms))));
    std::lock_guard<std::mutex> lk(m->comm_mutex);
    for (int c = 0; c <= MAX_S; ++c) {
        if (!m->comm_ready[c]) continue;
        for (int r = 0; r < m->n; ++r)
            if (m->local_rank < 0 || r == m->local_rank) (void)ncclCommAbort(m->comm[c][r]);
        m->comm_ready[c] = false;
        m->comm_aborted = true;
    }
}

void fail(bcd_hip_multi *m, const std::string &msg)
{
    {
        std::lock_guard<std::mutex> lk(m->err_mutex);
        if (m->err.empty()) m->err = msg;
    }
    m->abort_flag.store(true);
    abort_comms(m); // peers blocked in ncclSend / ncclRecv / all-reduce kernels (and the host threads waiting for them) are released
}

void progress_add(bcd_hip_multi *m, double share)
{
    if (!m->progress_fn || !(m->progress_total > 0.0)) return;
    std::lock_guard<std::mutex> lk(m->progress_mutex);
    m->progress_done = std::min(m->progress_total, m->progress_done + share);
    m->progress_fn((float)(m->progress_done / m->progress_total), m->progress_user);
}

// A frame of several ranks that has not finished after frame_timeout_s is failed (which releases the host barriers and, on the RCCL
// transport, aborts the communicators).  One watchdog thread
// per handle, started with the first such frame and parked between frames; a frame arms it on entry and disarms it on exit (two
// mutex sections: nothing a 3 ms step would notice, unlike a thread per frame).
struct Watchdog {
    std::mutex mu;
    std::condition_variable cv;
    std::thread th;
    bool armed = false, quit = false;
    bool firing = false;          // fail() of a timed-out frame is running: that frame's guard waits for it, so it can never hit the NEXT frame
    unsigned long long epoch = 0; // frames armed so far: a wait that times out only fires if its own frame is still the armed one
    void run(bcd_hip_multi *m)
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [this]() { return armed || quit; });
            if (quit) return;
            const unsigned long long mine = epoch;
            const bool released = cv.wait_for(lk, std::chrono::milliseconds(m->frame_timeout_ms), [this, mine]() { return quit || !armed || epoch != mine; });
            if (quit) return;
            if (!released) {
                // (the predicate was evaluated under the lock: this frame is still the armed one.)  The frame's guard cannot disarm and
                // return while `firing` is set, so the failure lands in the frame that timed out and nowhere else
                armed = false;
                firing = true;
                lk.unlock();
                fail(m, "frame timed out (a peer rank has stopped?): the frame is abandoned, RCCL communicators aborted");
                lk.lock();
                firing = false;
                cv.notify_all();
            }
        }
    }
};

struct FrameWatchdog { // scope guard of one frame
    Watchdog *w = nullptr;
    explicit FrameWatchdog(bcd_hip_multi *m);
    ~FrameWatchdog()
    {
        if (!w) return;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            w->armed = false;
            w->cv.wait(lk, [this]() { return !w->firing; });
        }
        w->cv.notify_all();
    }
};

FrameWatchdog::FrameWatchdog(bcd_hip_multi *m)
{
    if ((m->n < 2 && !m->loopback) || m->frame_timeout_ms <= 0) return; // (both transports: a rank that stops also stalls the in-process barriers)
    if (!m->watchdog) {
        m->watchdog = new Watchdog();
        Watchdog *wd = m->watchdog;
        wd->th = std::thread([wd, m]() { wd->run(m); });
    }
    w = m->watchdog;
    { std::lock_guard<std::mutex> lk(w->mu); w->armed = true; ++w->epoch; }
    w->cv.notify_all();
}

#define MCHK(m, rank, expr)                                                                                            \
    do {                                                                                                               \
        hipError_t e__ = (expr);                                                                                       \
        if (e__ != hipSuccess) { fail((m), std::string("rank ") + std::to_string(rank) + ": " #expr ": " + hipGetErrorString(e__)); return false; } \
    } while (0)
#define ECHK(m, rank, c, expr)                                                                                         \
    do {                                                                                                               \
        int rc__ = (expr);                                                                                             \
        if (rc__ != BCD_HIP_OK) { fail((m), std::string("rank ") + std::to_string(rank) + ": " #expr ": " + bcd_hip_last_error(c)); return false; } \
    } while (0)

// one neighbour exchange on channel `ch`: `bytes_*` to / from the rank above (up) and below (down); null pointers at the borders.
// The data must have been produced on stream[rank][ch]; on return the received data is ordered before later work of that stream.
void trace_op(bcd_hip_multi *m, int rank, int ch, int kind, size_t bytes_up, size_t bytes_down)
{
    if (!m->trace_on) return;
    std::lock_guard<std::mutex> lk(m->trace_mutex[rank]);
    for (int64_t v : { (int64_t)ch, (int64_t)kind, (int64_t)bytes_up, (int64_t)bytes_down }) m->trace[rank].push_back(v);
}

// one piece of a neighbour exchange: `bytes_up` to / from the rank above, `bytes_down` to / from the rank below (null receive pointers at the borders)
struct Seg { const void *send_up; void *recv_up; size_t bytes_up; const void *send_down; void *recv_down; size_t bytes_down; };

// several pieces in ONE operation (one RCCL group = one communication kernel; round 4: the two accumulator planes of a scale and the output
// lines of all scales used to be an operation each)
bool exchange_n(bcd_hip_multi *m, int rank, int ch, Seg *segs, int n)
{
    hipStream_t st = m->stream[rank][ch];
    const bool lb = m->loopback;
    const bool up = rank > 0 || lb, down = rank < m->n - 1 || lb;
    const int peer_up = lb ? rank : rank - 1, peer_down = lb ? rank : rank + 1;
    size_t tot_up = 0, tot_down = 0;
    for (int i = 0; i < n; ++i) { tot_up += segs[i].bytes_up; tot_down += segs[i].bytes_down; }
    trace_op(m, rank, ch, 0, up ? tot_up : 0, down ? tot_down : 0);
    if (lb) { // the rank is its own neighbour: what it "receives" is its own data and goes to scratch unless the caller (the self-test) wants it
        if (!m->loop_rx[ch][0].ensure(tot_up + 256 * (size_t)n) || !m->loop_rx[ch][1].ensure(tot_down + 256 * (size_t)n)) { fail(m, "out of device memory"); return false; }
        size_t ou = 0, od = 0;
        for (int i = 0; i < n; ++i) {
            if (!segs[i].recv_up) segs[i].recv_up = (char *)m->loop_rx[ch][0].p + ou;
            if (!segs[i].recv_down) segs[i].recv_down = (char *)m->loop_rx[ch][1].p + od;
            ou += (segs[i].bytes_up + 255) & ~(size_t)255;
            od += (segs[i].bytes_down + 255) & ~(size_t)255;
        }
    }
    if (m->use_rccl) {
        ncclResult_t r = ncclSuccess, e = ncclSuccess;
        {
            std::shared_lock<std::shared_timed_mutex> enq(m->comm_rw); // (an abort waits -- a bounded time -- for this section to end before it frees the communicator)
            if (!m->comm_ready[ch] || m->abort_flag.load()) return false; // aborted meanwhile: fail() has recorded why
            r = ncclGroupStart();
            // operations to the same peer are matched in issue order (both sides issue the pieces in the same order; in loopback the k-th send pairs with the k-th receive)
            for (int i = 0; i < n && r == ncclSuccess; ++i) {
                const Seg &g = segs[i];
                if (up) { r = ncclSend(g.send_up, g.bytes_up, ncclChar, peer_up, m->comm[ch][rank], st); if (r == ncclSuccess) r = ncclRecv(g.recv_up, g.bytes_up, ncclChar, peer_up, m->comm[ch][rank], st); }
                if (r == ncclSuccess && down) { r = ncclSend(g.send_down, g.bytes_down, ncclChar, peer_down, m->comm[ch][rank], st); if (r == ncclSuccess) r = ncclRecv(g.recv_down, g.bytes_down, ncclChar, peer_down, m->comm[ch][rank], st); }
            }
            e = ncclGroupEnd();
        }
        if (r != ncclSuccess || e != ncclSuccess) { fail(m, std::string("RCCL exchange failed: ") + ncclGetErrorString(r != ncclSuccess ? r : e)); return false; }
        return true;
    }
    // in-process transport: publish, rendezvous, copy from the neighbours' buffers, rendezvous (buffers may be reused afterwards)
    MCHK(m, rank, hipStreamSynchronize(st));
    for (int i = 0; i < n; ++i) {
        m->offer[ch][rank].up = segs[i].send_up;
        m->offer[ch][rank].down = segs[i].send_down;
        if (!m->barrier[ch].wait()) return false;
        if (up) MCHK(m, rank, hipMemcpyAsync(segs[i].recv_up, m->offer[ch][rank - 1].down, segs[i].bytes_up, hipMemcpyDefault, st));
        if (down) MCHK(m, rank, hipMemcpyAsync(segs[i].recv_down, m->offer[ch][rank + 1].up, segs[i].bytes_down, hipMemcpyDefault, st));
        MCHK(m, rank, hipStreamSynchronize(st));
        if (!m->barrier[ch].wait()) return false;
    }
    return true;
}

bool exchange(bcd_hip_multi *m, int rank, int ch, const void *send_up, void *recv_up, size_t bytes_up, const void *send_down, void *recv_down,
              size_t bytes_down)
{
    Seg g{ send_up, recv_up, bytes_up, send_down, recv_down, bytes_down };
    return exchange_n(m, rank, ch, &g, 1);
}

// sum of one integer over all ranks (channel ch); every rank gets the total
bool allreduce(bcd_hip_multi *m, int rank, int ch, long long *value)
{
    trace_op(m, rank, ch, 1, 0, 0);
    if (m->use_rccl) {
        hipStream_t st = m->stream[rank][ch];
        MCHK(m, rank, hipMemcpyAsync(m->d_red[rank][ch], value, sizeof(long long), hipMemcpyHostToDevice, st));
        ncclResult_t r;
        {
            std::shared_lock<std::shared_timed_mutex> enq(m->comm_rw);
            if (!m->comm_ready[ch] || m->abort_flag.load()) return false;
            r = ncclAllReduce(m->d_red[rank][ch], m->d_red[rank][ch], 1, ncclInt64, ncclSum, m->comm[ch][rank], st);
        }
        if (r != ncclSuccess) { fail(m, std::string("RCCL all-reduce failed: ") + ncclGetErrorString(r)); return false; }
        MCHK(m, rank, hipMemcpyAsync(value, m->d_red[rank][ch], sizeof(long long), hipMemcpyDeviceToHost, st));
        MCHK(m, rank, hipStreamSynchronize(st));
        return true;
    }
    m->offer[ch][rank].value = *value;
    if (!m->barrier[ch].wait()) return false;
    long long total = 0;
    for (int r = 0; r < m->n; ++r) total += m->offer[ch][r].value;
    if (!m->barrier[ch].wait()) return false;
    *value = total;
    return true;
}

// RCCL transport: the rank's contribution is already in d_red[rank][ch], produced on the channel's stream (bcd_hip_active_step_enqueue); the sum over
// all ranks is copied to h_red[rank][ch] in stream order; the caller synchronises the stream (or an event behind the copy) before it reads it
bool allreduce_device_enqueue(bcd_hip_multi *m, int rank, int ch)
{
    trace_op(m, rank, ch, 1, 0, 0);
    hipStream_t st = m->stream[rank][ch];
    ncclResult_t r;
    {
        std::shared_lock<std::shared_timed_mutex> enq(m->comm_rw);
        if (!m->comm_ready[ch] || m->abort_flag.load()) return false;
        r = ncclAllReduce(m->d_red[rank][ch], m->d_red[rank][ch], 1, ncclInt64, ncclSum, m->comm[ch][rank], st);
    }
    if (r != ncclSuccess) { fail(m, std::string("RCCL all-reduce failed: ") + ncclGetErrorString(r)); return false; }
    MCHK(m, rank, hipMemcpyAsync(m->h_red[rank][ch], m->d_red[rank][ch], sizeof(long long), hipMemcpyDeviceToHost, st));
    return true;
}

struct Job {
    bcd_hip_multi *m;
    const float *h_col, *h_ns, *h_hist, *h_cov;
    float *h_out;
    int W, H, D, S;
    bcd_hip_params prm;
    Geometry geom;
};

// ---- one scale of one rank: masks, frame-ordered marking, estimate, accumulator halos, finalisation -----------------------
// defer_sync (the finest scale of a multiscale frame, round 6): the scale's last kernel is followed by ev_tail[rank] instead of a host synchronisation --
// the caller enqueues the merges behind that event and waits once, at the end of the frame
bool scale_worker(const Job &job, int rank, int s, const ScaleBand *bands, bool defer_sync = false)
{
    bcd_hip_multi *m = job.m;
    if (hipSetDevice(m->devices[rank]) != hipSuccess) { fail(m, "hipSetDevice failed"); return false; }
    bcd_hip_ctx *c = m->ctx[rank][s];
    hipStream_t st = m->stream[rank][s];
    const Geometry &g = job.geom;
    const ScaleBand &sb = bands[s];
    const bool up = rank > 0, down = rank < g.world - 1;
    const int W = sb.W, D = job.D, b = g.b, w = g.w, halo = g.halo;
    const int o0 = sb.own0 - sb.loc0, o1 = sb.own1 - sb.loc0; // owned lines, local indices
    const int a0 = up ? o0 - halo : 0, a1 = down ? o1 + halo : sb.loc1 - sb.loc0; // the sub-band this scale works on
    const int rows = a1 - a0, r0 = o0 - a0, r1 = o1 - a0;
    const int row_offset = sb.loc0 + a0;
    const uint32_t seed = bcd_hip_scale_seed(job.prm.order_seed, s);
    auto B = [&](int kind) -> DBuf & { return m->buf[rank][s][kind]; };
    const size_t npix = (size_t)rows * W;
    const int side = 2 * b + 1, words = (side * side + 31) / 32;
    for (int k : { (int)bcd_hip_multi::PIXCOV, (int)bcd_hip_multi::MASK, (int)bcd_hip_multi::NSIM, (int)bcd_hip_multi::STATE, (int)bcd_hip_multi::SUM,
                   (int)bcd_hip_multi::CNT, (int)bcd_hip_multi::RX_UP_S, (int)bcd_hip_multi::RX_UP_C, (int)bcd_hip_multi::RX_DN_S, (int)bcd_hip_multi::RX_DN_C }) {
        size_t bytes = 0;
        switch (k) {
        case bcd_hip_multi::PIXCOV: bytes = npix * 6 * 4; break;
        case bcd_hip_multi::MASK: bytes = npix * words * 4; break;
        case bcd_hip_multi::NSIM: bytes = npix * 4; break;
        case bcd_hip_multi::STATE: bytes = npix; break;
        case bcd_hip_multi::SUM: bytes = npix * 12; break;
        case bcd_hip_multi::CNT: bytes = npix * 4; break;
        case bcd_hip_multi::RX_UP_S: case bcd_hip_multi::RX_DN_S: bytes = (size_t)halo * W * 12; break;
        default: bytes = (size_t)halo * W * 4; break;
        }
        if (!B(k).ensure(bytes)) { fail(m, "out of device memory"); return false; }
    }
    const float *col = (const float *)B(bcd_hip_multi::IN_COL).p + (size_t)a0 * W * 3;
    const float *ns = (const float *)B(bcd_hip_multi::IN_NS).p + (size_t)a0 * W;
    const float *hist = (const float *)B(bcd_hip_multi::IN_HIST).p + (size_t)a0 * W * D;
    const float *cov = (const float *)B(bcd_hip_multi::IN_COV).p + (size_t)a0 * W * 6;
    float *pixcov = (float *)B(bcd_hip_multi::PIXCOV).p;
    uint32_t *mask = (uint32_t *)B(bcd_hip_multi::MASK).p;
    int32_t *nsim = (int32_t *)B(bcd_hip_multi::NSIM).p;
    uint8_t *state = (uint8_t *)B(bcd_hip_multi::STATE).p;
    float *sum = (float *)B(bcd_hip_multi::SUM).p;
    int32_t *cnt = (int32_t *)B(bcd_hip_multi::CNT).p;

    // whatever way this scale ends, the finer scales must not wait for it any longer
    struct GateRelease { CommGate &gate; int s; ~GateRelease() { gate.finish(s); } } release{ m->gate[rank], s };
    if (s > 0) MCHK(m, rank, hipStreamWaitEvent(st, m->ev_level[rank][s], 0));
    // the finest scale is the critical path of the band: the coarse scales' persistent estimate kernels keep to a quarter of the CU slots
    ECHK(m, rank, c, bcd_hip_set_cu_share(c, s == 0 || g.S == 1 ? 100 : 25));
    // per-pixel covariances, the accumulators cleared in the same pass, and every counter / flag / work queue of the chain in one launch
    ECHK(m, rank, c, bcd_hip_scale_begin(c, cov, ns, W, rows, pixcov, sum, cnt));
    // Similar-patch masks with the production kernels; whether they are valid (inputs inside the guarded range, borderline list not
    // overflowed) comes back with the first host round trip that follows anyway -- the first marking batch -- instead of one of its own.
    const float tau = job.prm.hist_dist_threshold;
    const bool marking = job.prm.marked_skip_probability > 0.f;
    ECHK(m, rank, c, bcd_hip_similarity_masks_deferred(c, hist, ns, W, rows, D, w, b, tau, mask, nsim));
    bool verdict_known = false;
    if (!marking) { // no marking batch: ask now; the masks are local, so every rank decides for itself
        MCHK(m, rank, hipStreamSynchronize(st));
        int redo = 0;
        ECHK(m, rank, c, bcd_hip_similarity_masks_verdict(c, &redo));
        if (redo) ECHK(m, rank, c, bcd_hip_similarity_masks_exact(c, hist, ns, W, rows, D, w, b, tau, mask, nsim));
        verdict_known = true;
    }
    const bool talk = g.world > 1 || m->loopback; // (loopback: one rank that exchanges with itself, see bcd_hip_multi::loopback)
    const bool gated = m->ordered && talk;
    if (gated && !m->gate[rank].wait1(s, g.S)) return false; // P1 of this scale: after the coarser scales' P1
    int rounds = 0;
    const long long REDO = 1ll << 40; // added to the all-reduced count of undecided pixels by a rank whose masks are not valid
    // Round 6: on the RCCL transport with 3 x 3 patches the estimate is enqueued BEHIND every marking batch and its all-reduce, valid only if the
    // all-reduced count came out zero (the list and fallback kernels look at the word on the device); the host waits once per batch, inside
    // bcd_hip_bayes_accumulate_rows, with the estimate kernels already in the queue -- it used to wait for the batch, for the all-reduce and for the lists,
    // each time with an empty queue behind it.
    const bool speculate = marking && talk && m->use_rccl && w == 1;
    bool estimated = false;
    // CommGate: the operations of the first batch are P1, everything after them R
    bool in_p1 = true, in_r = false;
    auto end_p1 = [&]() { if (in_p1) { in_p1 = false; m->gate[rank].finish1(s); } };
    auto before_op = [&]() -> bool { // called before every communication operation of the marking
        if (in_p1 || in_r || !gated) return true;
        in_r = true;
        return m->gate[rank].wait_r(s, g.S);
    };
    for (;;) { // the marking problem; once more from the start if some rank has to recompute its masks
        if (marking && talk) {
            // |S| of the b boundary lines comes from their owner (locally their windows are cut by the band edge)
            if (!before_op()) return false;
            if (!exchange(m, rank, s, nsim + (size_t)r0 * W, up ? nsim + (size_t)(r0 - b) * W : nullptr, (size_t)b * W * 4,
                          nsim + (size_t)(r1 - b) * W, down ? nsim + (size_t)r1 * W : nullptr, (size_t)b * W * 4)) return false;
        }
        // Initial states: a function of the GLOBAL pixel index alone (skip draws), so the b boundary lines of the neighbours are initialised here like
        // their owners initialise them -- the first batch needs no exchange of states (round 6: one communication operation less in front of every
        // scale's marking); from the second batch on the boundary lines' states come from their owners.
        ECHK(m, rank, c, bcd_hip_active_init(c, nsim, W, rows, w, up ? r0 - b : r0, down ? r1 + b : r1, job.prm.marked_skip_probability, seed, row_offset, state));
        rounds = 0;
        bool restart = false, my_redo = false;
        if (marking) {
            long long before = -1;
            for (bool first_batch = true;; first_batch = false) {
                if (talk && !first_batch) {
                    if (!before_op()) return false;
                    if (!exchange(m, rank, s, state + (size_t)r0 * W, up ? state + (size_t)(r0 - b) * W : nullptr, (size_t)b * W,
                                  state + (size_t)(r1 - b) * W, down ? state + (size_t)r1 * W : nullptr, (size_t)b * W)) return false;
                }
                // Round 6: one synchronisation per batch.  The batch leaves the rank's contribution (undecided pixels, + REDO when its masks are not
                // valid: the same test on the device) in the all-reduce buffer; the all-reduce and the copy of its result follow in stream order.
                long long total = 0;
                if (talk && m->use_rccl) {
                    ECHK(m, rank, c, bcd_hip_active_step_enqueue(c, mask, nsim, W, rows, w, b, r0, r1, job.prm.use_random_pixel_order, seed, row_offset, state,
                                                                 reinterpret_cast<int64_t *>(m->d_red[rank][s]), verdict_known ? 0 : 1));
                    if (!allreduce_device_enqueue(m, rank, s)) return false;
                    end_p1(); // the first batch is in the queue: the next finer scale may enqueue its own (before this one's outcome is known)
                    if (speculate) {
                        int skipped = 0;
                        ECHK(m, rank, c, bcd_hip_bayes_accumulate_rows(c, col, pixcov, mask, nsim, state, W, rows, w, b, job.prm.min_eigen_value, sum, cnt, r0, r1,
                                                                       reinterpret_cast<const int64_t *>(m->d_red[rank][s]), reinterpret_cast<const int64_t *>(m->h_red[rank][s]), &skipped));
                        estimated = !skipped; // (the call waited for an event behind the copy of the all-reduced word)
                    } else
                        MCHK(m, rank, hipStreamSynchronize(st));
                    total = *m->h_red[rank][s];
                    int32_t undecided = 0;
                    ECHK(m, rank, c, bcd_hip_active_step_collect(c, &undecided, nullptr));
                    if (!verdict_known) { int redo = 0; ECHK(m, rank, c, bcd_hip_similarity_masks_verdict(c, &redo)); my_redo = redo != 0; }
                    ++rounds;
                } else {
                    int32_t undecided = 0;
                    ECHK(m, rank, c, bcd_hip_active_step(c, mask, nsim, W, rows, w, b, r0, r1, job.prm.use_random_pixel_order, seed, row_offset,
                                                          rounds == 0 && job.prm.marked_skip_probability >= 1.f, state, &undecided));
                    ++rounds;
                    total = undecided;
                    if (!verdict_known) { // the batch synchronised the stream: the flags of the masks are on the host
                        int redo = 0;
                        ECHK(m, rank, c, bcd_hip_similarity_masks_verdict(c, &redo));
                        my_redo = redo != 0;
                        if (my_redo) total += REDO;
                    }
                    if (talk && !allreduce(m, rank, s, &total)) return false;
                    end_p1();
                }
                if (!verdict_known) {
                    verdict_known = true;
                    if (total >= REDO) { restart = true; break; }
                }
                if (total == 0) break;
                if (before >= 0 && total >= before && rounds > 4 * (W + sb.H) + 64) { fail(m, "marking fixed point made no progress"); return false; }
                before = total;
            }
        }
        if (!restart) break;
        if (my_redo) ECHK(m, rank, c, bcd_hip_similarity_masks_exact(c, hist, ns, W, rows, D, w, b, tau, mask, nsim));
    }
    end_p1();                   // (no marking: nothing was enqueued)
    m->gate[rank].finish_r(s);  // the marking of this scale is complete: nothing of it is enqueued on a communicator until P2
    progress_add(m, 0.5 * (double)(r1 - r0) * W); // similar patches selected, processed set known
    // halo lines are processed by their owner: only the owned lines are listed
    if (!estimated)
        ECHK(m, rank, c, bcd_hip_bayes_accumulate_rows(c, col, pixcov, mask, nsim, state, W, rows, w, b, job.prm.min_eigen_value, sum, cnt, r0, r1, nullptr, nullptr, nullptr));
    // accumulator halos: the (b + w) lines written outside the owned band belong to the neighbours
    float *rx_us = (float *)B(bcd_hip_multi::RX_UP_S).p, *rx_ds = (float *)B(bcd_hip_multi::RX_DN_S).p;
    int32_t *rx_uc = (int32_t *)B(bcd_hip_multi::RX_UP_C).p, *rx_dc = (int32_t *)B(bcd_hip_multi::RX_DN_C).p;
    if (gated && !m->gate[rank].wait2(s, g.S)) return false; // P2: after every scale's marking (P1 and R) and the coarser scales' accumulators
    if (talk) {
        Seg acc[2] = { { sum, rx_us, (size_t)halo * W * 12, sum + (size_t)(rows - halo) * W * 3, rx_ds, (size_t)halo * W * 12 },
                       { cnt, rx_uc, (size_t)halo * W * 4, cnt + (size_t)(rows - halo) * W, rx_dc, (size_t)halo * W * 4 } };
        if (!exchange_n(m, rank, s, acc, 2)) return false; // sums and counts in one operation
    }
    m->gate[rank].finish2(s);
    float *out = (float *)B(bcd_hip_multi::OUT).p;
    ECHK(m, rank, c, bcd_hip_finalize_band(c, sum + (size_t)r0 * W * 3, cnt + (size_t)r0 * W, W, r1 - r0, halo, up ? rx_us : nullptr, up ? rx_uc : nullptr,
                                           down ? rx_ds : nullptr, down ? rx_dc : nullptr, out + (size_t)o0 * W * 3));
    if (rank == 0) m->stats.marking_rounds[s] = rounds;
    if (defer_sync) { MCHK(m, rank, hipEventRecord(m->ev_tail[rank], st)); return true; } // (the caller reports the progress after its own synchronisation)
    MCHK(m, rank, hipStreamSynchronize(st));
    progress_add(m, 0.5 * (double)(r1 - r0) * W);
    return true;
}

// ---- one rank: inputs, pyramid, the scales (concurrently), output halos and merges, result ---------------------------------
// h_* point at the first LOCAL line of the band when `local_band` is set (one process per GPU), else at line 0 of the frame
bool rank_upload(const Job &job, int rank, bool local_band)
{
    bcd_hip_multi *m = job.m;
    if (hipSetDevice(m->devices[rank]) != hipSuccess) { fail(m, "hipSetDevice failed"); return false; }
    const Geometry &g = job.geom;
    const int S = g.S, D = job.D;
    ScaleBand bands[MAX_S];
    g.bands(rank, bands);
    hipStream_t sm = m->stream[rank][S];
    auto B = [&](int s, int kind) -> DBuf & { return m->buf[rank][s][kind]; };
    for (int s = 0; s < S; ++s) {
        const size_t np = (size_t)(bands[s].loc1 - bands[s].loc0) * bands[s].W;
        if (!B(s, bcd_hip_multi::IN_COL).ensure(np * 12) || !B(s, bcd_hip_multi::IN_NS).ensure(np * 4) || !B(s, bcd_hip_multi::IN_HIST).ensure(np * D * 4) ||
            !B(s, bcd_hip_multi::IN_COV).ensure(np * 24) || !B(s, bcd_hip_multi::OUT).ensure(np * 12)) { fail(m, "out of device memory"); return false; }
    }
    const size_t first = local_band ? 0 : (size_t)bands[0].loc0 * job.W, np = (size_t)(bands[0].loc1 - bands[0].loc0) * job.W;
    MCHK(m, rank, hipMemcpyAsync(B(0, bcd_hip_multi::IN_COL).p, job.h_col + first * 3, np * 12, hipMemcpyHostToDevice, sm));
    MCHK(m, rank, hipMemcpyAsync(B(0, bcd_hip_multi::IN_NS).p, job.h_ns + first, np * 4, hipMemcpyHostToDevice, sm));
    MCHK(m, rank, hipMemcpyAsync(B(0, bcd_hip_multi::IN_HIST).p, job.h_hist + first * D, np * D * 4, hipMemcpyHostToDevice, sm));
    MCHK(m, rank, hipMemcpyAsync(B(0, bcd_hip_multi::IN_COV).p, job.h_cov + first * 6, np * 24, hipMemcpyHostToDevice, sm));
    MCHK(m, rank, hipStreamSynchronize(sm));
    return true;
}

// everything between the resident inputs and the resident result of the band (what bench.py times at N > 1)
bool rank_compute(const Job &job, int rank)
{
    bcd_hip_multi *m = job.m;
    if (hipSetDevice(m->devices[rank]) != hipSuccess) { fail(m, "hipSetDevice failed"); return false; }
    const Geometry &g = job.geom;
    const int S = g.S, D = job.D;
    ScaleBand bands[MAX_S];
    g.bands(rank, bands);
    const bool up = rank > 0, down = rank < g.world - 1;
    bcd_hip_ctx *cm = m->ctx[rank][S];
    hipStream_t sm = m->stream[rank][S];
    auto B = [&](int s, int kind) -> DBuf & { return m->buf[rank][s][kind]; };
    // ---- the scales are independent until the merges: one thread, stream and context each.  The finest scale needs no pyramid level and is the critical
    // path of the band: it runs in THIS thread and starts at once (round 6: it used to start behind the enqueue of the eight pyramid kernels and the
    // creation of the other threads -- 0.1 ms of a 2.9 ms band step; the single-GPU call has always started its distance kernel beside the pyramid).  A
    // helper thread enqueues the pyramid and starts a coarse scale's thread as soon as that scale's level is recorded.
    std::vector<char> ok(S, 1);
    bool pyramid_ok = true;
    m->gate[rank].reset();
    auto out_rows = [&](int s, int local_line) { return (float *)B(s, bcd_hip_multi::OUT).p + (size_t)local_line * bands[s].W * 3; };
    const bool talk = g.world > 1 || m->loopback;
    // two lines of every unmerged finer output (hi - up(down(hi)) at the band edge), one line of the coarsest (up(lo)): scales [s_begin, s_end) in one operation
    auto exchange_edge_lines = [&](int s_begin, int s_end) -> bool {
        Seg lines[MAX_S];
        for (int s = s_begin; s < s_end; ++s) {
            const int n = s < S - 1 ? 2 : 1, o0 = bands[s].own0 - bands[s].loc0, o1 = bands[s].own1 - bands[s].loc0;
            const size_t bytes = (size_t)n * bands[s].W * 12;
            lines[s - s_begin] = Seg{ out_rows(s, o0), up ? out_rows(s, o0 - n) : nullptr, bytes, out_rows(s, o1 - n), down ? out_rows(s, o1) : nullptr, bytes };
        }
        return exchange_n(m, rank, S, lines, s_end - s_begin);
    };
    // mergeOutputs of scale s (MultiscaleDenoiser.cpp:453-466) on the band's lines (+ 2 at an interior edge); one line of the merged output then travels
    auto merge_scale = [&](int s) -> bool {
        const ScaleBand &sb = bands[s], &nb = bands[s + 1];
        const int o0 = sb.own0 - sb.loc0, o1 = sb.own1 - sb.loc0;
        const int m0 = up ? o0 - 2 : o0, m1 = down ? o1 + 2 : o1;
        const int lo0 = (sb.loc0 + m0) / 2 - nb.loc0;
        ECHK(m, rank, cm, bcd_hip_merge(cm, out_rows(s, m0), sb.W, m1 - m0, out_rows(s + 1, lo0), 3));
        if (s > 0 && talk) {
            const size_t bytes = (size_t)sb.W * 12;
            if (!exchange(m, rank, S, out_rows(s, o0), up ? out_rows(s, o0 - 1) : nullptr, bytes, out_rows(s, o1 - 1), down ? out_rows(s, o1) : nullptr, bytes)) return false;
        }
        return true;
    };
    auto coarse = [&]() -> bool {
        if (hipSetDevice(m->devices[rank]) != hipSuccess) { fail(m, "hipSetDevice failed"); return false; }
        std::vector<std::thread> th;
        struct Join { std::vector<std::thread> &th; ~Join() { for (auto &t : th) if (t.joinable()) t.join(); } } join{ th };
        // ---- local pyramid (MultiscaleDenoiser.cpp:41-53)
        for (int s = 1; s < S; ++s) {
            const ScaleBand &prev = bands[s - 1], &cur = bands[s];
            const int a = 2 * cur.loc0 - prev.loc0, rows = 2 * (cur.loc1 - cur.loc0);
            const float *pc = (const float *)B(s - 1, bcd_hip_multi::IN_COL).p + (size_t)a * prev.W * 3;
            const float *pn = (const float *)B(s - 1, bcd_hip_multi::IN_NS).p + (size_t)a * prev.W;
            const float *ph = (const float *)B(s - 1, bcd_hip_multi::IN_HIST).p + (size_t)a * prev.W * D;
            const float *pv = (const float *)B(s - 1, bcd_hip_multi::IN_COV).p + (size_t)a * prev.W * 6;
            ECHK(m, rank, cm, bcd_hip_downscale_avg(cm, pc, prev.W, rows, 3, (float *)B(s, bcd_hip_multi::IN_COL).p));
            ECHK(m, rank, cm, bcd_hip_downscale_sum(cm, pn, prev.W, rows, 1, (float *)B(s, bcd_hip_multi::IN_NS).p));
            ECHK(m, rank, cm, bcd_hip_downscale_sum(cm, ph, prev.W, rows, D, (float *)B(s, bcd_hip_multi::IN_HIST).p));
            ECHK(m, rank, cm, bcd_hip_downscale_cov(cm, pv, pn, prev.W, rows, (float *)B(s, bcd_hip_multi::IN_COV).p));
            MCHK(m, rank, hipEventRecord(m->ev_level[rank][s], sm)); // scale s starts when its level is there
            // (the event is recorded before the scale's thread exists: its hipStreamWaitEvent sees this frame's record)
            th.emplace_back([&, s]() { ok[s] = scale_worker(job, rank, s, bands) ? 1 : 0; });
        }
        for (auto &t : th) t.join(); // (each coarse scale has synchronised its stream)
        for (int s = 1; s < S; ++s)
            if (!ok[s]) return true; // (reported by the caller)
        // Round 6: the merges among the coarse scales do not wait for the finest one.  Their exchanges come behind every scale's accumulator exchange
        // in the global sequence (the finest scale's is enqueued long before its estimate has run), the finest scale's own merge follows in the caller.
        if (m->ordered && talk && !m->gate[rank].wait_all_p2(S)) return false;
        if (talk && !exchange_edge_lines(1, S)) return false;
        for (int s = S - 2; s >= 1; --s)
            if (!merge_scale(s)) return false;
        return true;
    };
    {
        std::thread helper;
        if (S > 1) helper = std::thread([&]() { pyramid_ok = coarse(); });
        ok[0] = scale_worker(job, rank, 0, bands, S > 1) ? 1 : 0;
        if (helper.joinable()) helper.join();
    }
    if (hipSetDevice(m->devices[rank]) != hipSuccess) { fail(m, "hipSetDevice failed"); return false; }
    if (!pyramid_ok) return false; // (fail() has raised the abort flag: every scale thread came back)
    for (int s = 0; s < S; ++s)
        if (!ok[s]) { m->abort_flag.store(true); return false; }
    // the merges follow the finest scale's last kernel in stream order (the coarse scales have synchronised their streams)
    if (S > 1) MCHK(m, rank, hipStreamWaitEvent(sm, m->ev_tail[rank], 0));
    // ---- the finest scale's merge: its two edge lines travel, then hi -= up(down(hi)); hi += up(lo) with the (merged) scale 1
    if (S > 1) {
        if (talk && !exchange_edge_lines(0, 1)) return false;
        if (!merge_scale(0)) return false;
    }
    MCHK(m, rank, hipStreamSynchronize(sm));
    if (S > 1) progress_add(m, 0.5 * (double)(bands[0].own1 - bands[0].own0) * bands[0].W); // (the finest scale's estimate: its worker did not wait for it)
    return true;
}

// owned lines of the result; h_out points at line 0 of the frame, or (local_band) at the first OWNED line
bool rank_download(const Job &job, int rank, bool local_band)
{
    bcd_hip_multi *m = job.m;
    if (hipSetDevice(m->devices[rank]) != hipSuccess) { fail(m, "hipSetDevice failed"); return false; }
    ScaleBand bands[MAX_S];
    job.geom.bands(rank, bands);
    hipStream_t sm = m->stream[rank][job.geom.S];
    const int o0 = bands[0].own0 - bands[0].loc0;
    const float *src = (const float *)m->buf[rank][0][bcd_hip_multi::OUT].p + (size_t)o0 * job.W * 3;
    MCHK(m, rank, hipMemcpyAsync(job.h_out + (local_band ? 0 : (size_t)bands[0].own0 * job.W * 3), src, (size_t)(bands[0].own1 - bands[0].own0) * job.W * 12,
                                 hipMemcpyDeviceToHost, sm));
    MCHK(m, rank, hipStreamSynchronize(sm));
    return true;
}

bool rank_worker(const Job &job, int rank)
{
    bcd_hip_multi *m = job.m;
    if (!rank_upload(job, rank, false)) return false;
    // every rank has its inputs resident: what follows is what a resident-data caller would time (stats.compute_ms, rank 0's clock)
    if (!m->barrier[job.S].wait()) return false;
    const auto t0 = std::chrono::steady_clock::now();
    if (!rank_compute(job, rank)) return false;
    if (!m->barrier[job.S].wait()) return false;
    if (rank == 0) m->stats.compute_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rank_download(job, rank, false);
}

// contexts / streams of the local ranks and the communicators of channels 0..S, created on first use
int prepare(bcd_hip_multi *m, int S)
{
    for (int r = 0; r < m->n; ++r) {
        if (m->local_rank >= 0 && r != m->local_rank) continue;
        if (hipSetDevice(m->devices[r]) != hipSuccess) { fail(m, "hipSetDevice failed"); return BCD_HIP_EDEVICE; }
        if (!m->ev_tail[r] && hipEventCreateWithFlags(&m->ev_tail[r], hipEventDisableTiming) != hipSuccess) { fail(m, "hipEventCreate failed"); return BCD_HIP_EDEVICE; }
        for (int c = 0; c <= S; ++c) {
            if (c < S && !m->ev_level[r][c] && hipEventCreateWithFlags(&m->ev_level[r][c], hipEventDisableTiming) != hipSuccess) { fail(m, "hipEventCreate failed"); return BCD_HIP_EDEVICE; }
            if (m->ctx[r][c]) continue;
            if (hipStreamCreateWithFlags(&m->stream[r][c], hipStreamNonBlocking) != hipSuccess ||
                bcd_hip_ctx_create(&m->ctx[r][c], m->devices[r], m->stream[r][c]) != BCD_HIP_OK) { fail(m, "cannot create an engine context"); return BCD_HIP_EDEVICE; }
            if (m->use_rccl && (hipMalloc((void **)&m->d_red[r][c], 64) != hipSuccess || hipHostMalloc((void **)&m->h_red[r][c], 64, hipHostMallocDefault) != hipSuccess)) { fail(m, "hipMalloc failed"); return BCD_HIP_ENOMEM; }
        }
    }
    if (m->use_rccl && (m->n > 1 || m->loopback))
        for (int c = 0; c <= S; ++c) {
            if (m->comm_ready[c]) continue;
            ncclResult_t r;
            if (m->local_rank >= 0 && m->comm_aborted) {
                // the unique ids were consumed by the communicators that have been aborted, and the other processes hold the other halves
                fail(m, "the RCCL communicators of this rank were aborted after a failure: destroy the handle and create it again from fresh unique ids (all processes)");
                return BCD_HIP_EDEVICE;
            }
            if (m->local_rank >= 0) {
                if ((size_t)c >= m->ids.size()) { fail(m, "not enough RCCL unique ids for this number of scales"); return BCD_HIP_EINVAL; }
                r = ncclCommInitRank(&m->comm[c][m->local_rank], m->n, m->ids[c], m->local_rank); // collective: every process, same order
            } else
                r = ncclCommInitAll(m->comm[c], m->n, m->devices);
            if (r != ncclSuccess) { fail(m, std::string("RCCL communicator creation failed: ") + ncclGetErrorString(r)); return BCD_HIP_EDEVICE; }
            if (c == 0) {
                static const bool verbose = [] { const char *e = getenv("BCD_HIP_MULTI_VERBOSE"); return e && e[0] == '1'; }();
                if (verbose) fprintf(stderr, "[bcd_hip_multi] rank %d of %d: communicators from %s\n", m->local_rank < 0 ? 0 : m->local_rank, m->n, rccl_identity().c_str());
            }
            std::lock_guard<std::mutex> lk(m->comm_mutex);
            m->comm_ready[c] = true;
        }
    return BCD_HIP_OK;
}

int make_job(bcd_hip_multi *m, Job &job, int W, int H, int D, int nb_scales, const bcd_hip_params *prm)
{
    if (!prm) { fail(m, "null parameters"); return BCD_HIP_EINVAL; }
    if (W <= 0 || H <= 0 || D <= 0 || nb_scales < 1 || nb_scales > MAX_S) { fail(m, "bad image size or number of scales"); return BCD_HIP_EINVAL; }
    if (prm->use_random_pixel_order == 2) { fail(m, "the strip visiting order (pixel order 2) is not available on row bands"); return BCD_HIP_EINVAL; }
    job.m = m; job.h_col = job.h_ns = job.h_hist = job.h_cov = nullptr; job.h_out = nullptr;
    job.W = W; job.H = H; job.D = D; job.S = nb_scales; job.prm = *prm;
    std::string err;
    if (!job.geom.init(W, H, nb_scales, prm->search_radius, prm->patch_radius, m->n, err, m->loopback)) { fail(m, err); return BCD_HIP_EINVAL; }
    if ((W >> (nb_scales - 1)) < 2 * prm->patch_radius + 1) { fail(m, "too many scales for this image size"); return BCD_HIP_EINVAL; }
    return BCD_HIP_OK;
}

// start of every entry point that runs a frame: no worker thread of an earlier call is alive here.  A frame that failed left
// some ranks' arrivals in the barriers and bits in the gates; a retry must not inherit them (it would pair up the wrong phases).
void reset_error(bcd_hip_multi *m)
{
    {
        std::lock_guard<std::mutex> lk(m->err_mutex);
        m->err.clear();
        m->abort_flag.store(false);
    }
    for (int c = 0; c <= MAX_S; ++c) m->barrier[c].reset();
    for (int r = 0; r < m->n; ++r) m->gate[r].reset();
    std::lock_guard<std::mutex> lk(m->progress_mutex);
    m->progress_done = 0.0;
    m->progress_total = 0.0;
}

} // namespace

extern "C" {

int bcd_hip_multi_create(bcd_hip_multi **out, const int *devices, int n_ranks)
{
    if (!out || !devices || n_ranks < 1 || n_ranks > MAX_RANKS) return BCD_HIP_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return BCD_HIP_EDEVICE;
    bcd_hip_multi *m = new (std::nothrow) bcd_hip_multi();
    if (!m) return BCD_HIP_ENOMEM;
    m->n = n_ranks;
    bool distinct = true;
    for (int r = 0; r < n_ranks; ++r) {
        if (devices[r] < 0 || devices[r] >= ndev) { delete m; return BCD_HIP_EINVAL; }
        m->devices[r] = devices[r];
        for (int q = 0; q < r; ++q) distinct = distinct && devices[q] != devices[r];
    }
    m->use_rccl = distinct && n_ranks > 1;
    memset(m->ctx, 0, sizeof(m->ctx));
    memset(m->stream, 0, sizeof(m->stream));
    memset(m->d_red, 0, sizeof(m->d_red));
    memset(m->h_red, 0, sizeof(m->h_red));
    memset(m->ev_level, 0, sizeof(m->ev_level));
    memset(m->ev_tail, 0, sizeof(m->ev_tail));
    memset(m->comm_ready, 0, sizeof(m->comm_ready));
    memset(&m->stats, 0, sizeof(m->stats));
    for (int c = 0; c <= MAX_S; ++c) { m->barrier[c].parties = n_ranks; m->barrier[c].abort_flag = &m->abort_flag; }
    for (int r = 0; r < n_ranks; ++r) m->gate[r].abort_flag = &m->abort_flag;
    if (const char *t = getenv("BCD_HIP_MULTI_TIMEOUT_S")) m->frame_timeout_ms = atoi(t) * 1000;
    if (const char *t = getenv("BCD_HIP_MULTI_ABORT_WAIT_MS")) m->abort_wait_ms = atoi(t);
    const char *ordered = getenv("BCD_HIP_MULTI_ORDERED");
    m->ordered = m->use_rccl || (ordered && atoi(ordered) != 0);
    m->stats.n_ranks = n_ranks;
    m->stats.transport = m->use_rccl ? 1 : 0;
    *out = m;
    return BCD_HIP_OK;
}

void bcd_hip_multi_destroy(bcd_hip_multi *m)
{
    if (!m) return;
    if (m->watchdog) {
        { std::lock_guard<std::mutex> lk(m->watchdog->mu); m->watchdog->quit = true; }
        m->watchdog->cv.notify_all();
        m->watchdog->th.join();
        delete m->watchdog;
        m->watchdog = nullptr;
    }
    for (int c = 0; c <= MAX_S; ++c)
        if (m->comm_ready[c])
            for (int r = 0; r < m->n; ++r)
                if (m->local_rank < 0 || r == m->local_rank) (void)ncclCommDestroy(m->comm[c][r]);
    for (int r = 0; r < m->n; ++r) {
        (void)hipSetDevice(m->devices[r]);
        for (int s = 0; s < MAX_S; ++s)
            for (int k = 0; k < bcd_hip_multi::NBUF; ++k) m->buf[r][s][k].release();
        if (m->ev_tail[r]) (void)hipEventDestroy(m->ev_tail[r]);
        for (int c = 0; c <= MAX_S; ++c) {
            if (m->d_red[r][c]) (void)hipFree(m->d_red[r][c]);
            if (m->h_red[r][c]) (void)hipHostFree(m->h_red[r][c]);
            if (m->ctx[r][c]) bcd_hip_ctx_destroy(m->ctx[r][c]);
            if (m->stream[r][c]) (void)hipStreamDestroy(m->stream[r][c]);
            if (c < MAX_S && m->ev_level[r][c]) (void)hipEventDestroy(m->ev_level[r][c]);
        }
    }
    delete m;
}

const char *bcd_hip_multi_last_error(const bcd_hip_multi *m) { return m ? m->err.c_str() : "null handle"; }

int bcd_hip_multi_get_stats(const bcd_hip_multi *m, bcd_hip_multi_stats *out)
{
    if (!m || !out) return BCD_HIP_EINVAL;
    *out = m->stats;
    return BCD_HIP_OK;
}

int bcd_hip_multi_set_progress_callback(bcd_hip_multi *m, bcd_hip_progress_fn fn, void *user)
{
    if (!m) return BCD_HIP_EINVAL;
    std::lock_guard<std::mutex> lk(m->progress_mutex);
    m->progress_fn = fn;
    m->progress_user = user;
    return BCD_HIP_OK;
}

int bcd_hip_multi_set_frame_timeout(bcd_hip_multi *m, int milliseconds)
{
    if (!m || milliseconds < 0) return BCD_HIP_EINVAL;
    m->frame_timeout_ms = milliseconds; // (read when a frame arms the watchdog: takes effect with the next frame)
    return BCD_HIP_OK;
}

int bcd_hip_multi_set_comm_trace(bcd_hip_multi *m, int enabled)
{
    if (!m) return BCD_HIP_EINVAL;
    m->trace_on = enabled != 0;
    return BCD_HIP_OK;
}

int bcd_hip_multi_get_comm_trace(bcd_hip_multi *m, int rank, int64_t *out, int capacity)
{
    if (!m || rank < 0 || rank >= m->n || capacity < 0 || (capacity > 0 && !out)) return -1;
    std::lock_guard<std::mutex> lk(m->trace_mutex[rank]);
    const int n = (int)m->trace[rank].size();
    for (int i = 0; i < n && i < capacity; ++i) out[i] = m->trace[rank][i];
    return n;
}

int bcd_hip_multi_denoise_host(bcd_hip_multi *m, const float *h_colors, const float *h_ns, const float *h_hist, const float *h_cov, int W, int H, int D,
                               int nb_scales, const bcd_hip_params *prm, float *h_out)
{
    if (!m) return BCD_HIP_EINVAL;
    reset_error(m);
    if (m->local_rank >= 0) { fail(m, "this handle drives one rank of several processes: use the bcd_hip_multi_rank_* calls"); return BCD_HIP_EINVAL; }
    if (!h_colors || !h_ns || !h_hist || !h_cov || !h_out) { fail(m, "null pointer"); return BCD_HIP_EINVAL; }
    Job job;
    int rc = make_job(m, job, W, H, D, nb_scales, prm);
    if (rc != BCD_HIP_OK) return rc;
    job.h_col = h_colors; job.h_ns = h_ns; job.h_hist = h_hist; job.h_cov = h_cov; job.h_out = h_out;
    rc = prepare(m, nb_scales);
    if (rc != BCD_HIP_OK) return rc;
    for (int r = 0; r < m->n; ++r) m->trace[r].clear();
    {
        std::lock_guard<std::mutex> lk(m->progress_mutex);
        for (int s = 0; s < nb_scales; ++s) m->progress_total += (double)(W >> s) * (double)(H >> s);
    }
    FrameWatchdog watchdog(m);
    std::vector<std::thread> th;
    std::vector<char> ok(m->n, 1);
    for (int r = 1; r < m->n; ++r) th.emplace_back([&, r]() { ok[r] = rank_worker(job, r) ? 1 : 0; });
    ok[0] = rank_worker(job, 0) ? 1 : 0;
    for (auto &t : th) t.join();
    for (int r = 0; r < m->n; ++r)
        if (!ok[r]) { fail(m, "a rank failed"); return BCD_HIP_EDEVICE; }
    m->stats.frames += 1;
    return BCD_HIP_OK;
}

int bcd_hip_multi_rccl_info(char *out, int capacity)
{
    if (!out || capacity <= 0) return BCD_HIP_EINVAL;
    const std::string t = rccl_identity();
    strncpy(out, t.c_str(), (size_t)capacity - 1);
    out[capacity - 1] = 0;
    return BCD_HIP_OK;
}

// ---- one process per GPU -------------------------------------------------------------------------------------------------------
int bcd_hip_multi_unique_id(char *out128)
{
    if (!out128) return BCD_HIP_EINVAL;
    static_assert(sizeof(ncclUniqueId) == BCD_HIP_MULTI_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return BCD_HIP_EDEVICE;
    memcpy(out128, &id, sizeof(id));
    return BCD_HIP_OK;
}

int bcd_hip_multi_create_rank(bcd_hip_multi **out, int rank, int n_ranks, int device, const char *ids, int n_ids)
{
    if (!out || rank < 0 || n_ranks < 1 || rank >= n_ranks || n_ranks > MAX_RANKS || (n_ranks > 1 && (!ids || n_ids < 2))) return BCD_HIP_EINVAL;
    std::vector<int> devices(n_ranks, device);
    int rc = bcd_hip_multi_create(out, devices.data(), n_ranks);
    if (rc != BCD_HIP_OK) return rc;
    bcd_hip_multi *m = *out;
    m->local_rank = rank;
    m->use_rccl = true; // ranks of other processes are only reachable through RCCL
    m->ordered = true;
    m->stats.transport = 1;
    for (int i = 0; i < n_ids && ids; ++i) { // (a world of one only needs them in loopback mode)
        ncclUniqueId id;
        memcpy(&id, ids + (size_t)i * sizeof(id), sizeof(id));
        m->ids.push_back(id);
    }
    return BCD_HIP_OK;
}

int bcd_hip_multi_set_loopback(bcd_hip_multi *m, int enabled)
{
    if (!m) return BCD_HIP_EINVAL;
    if (enabled && (m->n != 1 || m->local_rank != 0)) { fail(m, "loopback needs a one-rank handle made by bcd_hip_multi_create_rank(rank 0 of 1)"); return BCD_HIP_EINVAL; }
    if (enabled && m->ids.empty()) { fail(m, "loopback needs RCCL unique ids (one per scale + 1)"); return BCD_HIP_EINVAL; }
    m->loopback = enabled != 0;
    return BCD_HIP_OK;
}

int bcd_hip_multi_rank_renew_ids(bcd_hip_multi *m, const char *ids, int n_ids)
{
    if (!m || !ids || n_ids < 1) return BCD_HIP_EINVAL;
    if (m->local_rank < 0) { fail(m, "not a one-rank handle"); return BCD_HIP_EINVAL; }
    // communicators that are still up belong to the old ids: a renewal replaces all of them (every process does the same)
    abort_comms(m);
    m->ids.clear();
    for (int i = 0; i < n_ids; ++i) {
        ncclUniqueId id;
        memcpy(&id, ids + (size_t)i * sizeof(id), sizeof(id));
        m->ids.push_back(id);
    }
    std::lock_guard<std::mutex> lk(m->comm_mutex);
    m->comm_aborted = false;
    return BCD_HIP_OK;
}

// One device, real RCCL: everything the band driver does with the library, through the driver's own exchange() / allreduce() / fail() / prepare().
int bcd_hip_multi_selftest_transport(int device, long long halo_bytes, char *report, int report_capacity)
{
    auto say = [&](const std::string &t) { if (report && report_capacity > 0) { strncpy(report, t.c_str(), (size_t)report_capacity - 1); report[report_capacity - 1] = 0; } };
    if (halo_bytes < 16 || halo_bytes > (1ll << 30)) { say("halo_bytes out of range"); return BCD_HIP_EINVAL; }
    const int S = 1, n_ids = S + 1;
    auto fresh_ids = [&](std::vector<char> &v) {
        v.resize((size_t)n_ids * BCD_HIP_MULTI_ID_BYTES);
        for (int i = 0; i < n_ids; ++i)
            if (bcd_hip_multi_unique_id(v.data() + (size_t)i * BCD_HIP_MULTI_ID_BYTES) != BCD_HIP_OK) return false;
        return true;
    };
    std::vector<char> ids;
    if (!fresh_ids(ids)) { say("ncclGetUniqueId failed"); return BCD_HIP_EDEVICE; }
    bcd_hip_multi *m = nullptr;
    int rc = bcd_hip_multi_create_rank(&m, 0, 1, device, ids.data(), n_ids);
    if (rc != BCD_HIP_OK) { say("bcd_hip_multi_create_rank failed"); return rc; }
    struct Guard { bcd_hip_multi *m; void *p[4] = { nullptr, nullptr, nullptr, nullptr }; ~Guard() { for (void *q : p) if (q) (void)hipFree(q); bcd_hip_multi_destroy(m); } } guard{ m };
    auto bail = [&](const char *where, int code) { say(std::string(where) + ": " + bcd_hip_multi_last_error(m)); return code; };
    if ((rc = bcd_hip_multi_set_loopback(m, 1)) != BCD_HIP_OK) return bail("set_loopback", rc);
    reset_error(m);
    if ((rc = prepare(m, S)) != BCD_HIP_OK) return bail("prepare (ncclCommInitRank, n = 1)", rc);
    if (hipSetDevice(device) != hipSuccess) { say("hipSetDevice failed"); return BCD_HIP_EDEVICE; }
    const size_t nb = (size_t)halo_bytes;
    for (int i = 0; i < 4; ++i)
        if (hipMalloc(&guard.p[i], nb) != hipSuccess) { say("out of device memory"); return BCD_HIP_ENOMEM; }
    std::vector<unsigned char> h_up(nb), h_down(nb), back(nb);
    auto round = [&](int ch, unsigned salt, const char *what) -> int {
        hipStream_t st = m->stream[0][ch];
        for (size_t i = 0; i < nb; ++i) { h_up[i] = (unsigned char)(i * 7u + salt); h_down[i] = (unsigned char)(i * 13u + 3u * salt + 1u); }
        if (hipMemcpyAsync(guard.p[0], h_up.data(), nb, hipMemcpyHostToDevice, st) != hipSuccess || hipMemcpyAsync(guard.p[1], h_down.data(), nb, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemsetAsync(guard.p[2], 0, nb, st) != hipSuccess || hipMemsetAsync(guard.p[3], 0, nb, st) != hipSuccess) { say("copy failed"); return BCD_HIP_EDEVICE; }
        if (!exchange(m, 0, ch, guard.p[0], guard.p[2], nb, guard.p[1], guard.p[3], nb)) return bail(what, BCD_HIP_EDEVICE);
        if (hipStreamSynchronize(st) != hipSuccess) { say(std::string(what) + ": stream synchronisation failed"); return BCD_HIP_EDEVICE; }
        // the grouped self send / recv pairs in issue order: what went "up" comes back as the data from "up", likewise down
        if (hipMemcpy(back.data(), guard.p[2], nb, hipMemcpyDeviceToHost) != hipSuccess || back != h_up) { say(std::string(what) + ": data of the first send / recv pair differs"); return BCD_HIP_EDEVICE; }
        if (hipMemcpy(back.data(), guard.p[3], nb, hipMemcpyDeviceToHost) != hipSuccess || back != h_down) { say(std::string(what) + ": data of the second send / recv pair differs"); return BCD_HIP_EDEVICE; }
        long long v = 123456789012ll + salt;
        if (!allreduce(m, 0, ch, &v)) return bail(what, BCD_HIP_EDEVICE);
        if (v != 123456789012ll + salt) { say(std::string(what) + ": all-reduce of one rank changed the value"); return BCD_HIP_EDEVICE; }
        return BCD_HIP_OK;
    };
    for (int ch = 0; ch <= S; ++ch)
        if ((rc = round(ch, 17u + (unsigned)ch, "first exchange")) != BCD_HIP_OK) return rc;
    // failure path: ncclCommAbort on every communicator; the consumed ids cannot make new ones; fresh ids can
    fail(m, "self-test: simulated failure");
    if (m->comm_ready[0] || !m->comm_aborted) { say("abort did not take the communicators down"); return BCD_HIP_EDEVICE; }
    {
        long long v = 1;
        if (exchange(m, 0, 0, guard.p[0], guard.p[2], nb, guard.p[1], guard.p[3], nb) || allreduce(m, 0, 0, &v)) { say("an aborted communicator was used"); return BCD_HIP_EDEVICE; }
    }
    reset_error(m);
    if (prepare(m, S) == BCD_HIP_OK) { say("prepare() rebuilt communicators from consumed ids"); return BCD_HIP_EDEVICE; }
    if (!fresh_ids(ids)) { say("ncclGetUniqueId failed"); return BCD_HIP_EDEVICE; }
    if ((rc = bcd_hip_multi_rank_renew_ids(m, ids.data(), n_ids)) != BCD_HIP_OK) return bail("renew_ids", rc);
    reset_error(m);
    if ((rc = prepare(m, S)) != BCD_HIP_OK) return bail("prepare after the abort", rc);
    for (int ch = 0; ch <= S; ++ch)
        if ((rc = round(ch, 91u + (unsigned)ch, "exchange after abort + rebuild")) != BCD_HIP_OK) return rc;
    // a rank thread stuck inside an RCCL call (a peer that never joins) holds the enqueue section for good: the abort must not wait for it for ever
    {
        std::atomic<bool> held{ false }, release{ false };
        std::thread stuck([&] {
            std::shared_lock<std::shared_timed_mutex> enq(m->comm_rw);
            held.store(true);
            while (!release.load()) std::this_thread::sleep_for(std::chrono::milliseconds(5));
        });
        while (!held.load()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        const int keep = m->abort_wait_ms;
        m->abort_wait_ms = 200;
        const auto t0 = std::chrono::steady_clock::now();
        fail(m, "self-test: failure while a rank thread is stuck in an RCCL call");
        const double waited_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        m->abort_wait_ms = keep;
        release.store(true);
        stuck.join();
        if (!m->abort_forced.load() || m->comm_ready[0] || waited_ms > 5000.0) { say("an abort behind a stuck enqueue section did not go through in bounded time"); return BCD_HIP_EDEVICE; }
    }
    say("ok: ncclCommInitRank(n=1) x2, grouped self send/recv + int64 all-reduce on 2 channels, ncclCommAbort, rebuild from fresh ids, second exchange, "
        "abort behind a stuck enqueue section after a bounded wait; " + rccl_identity());
    return BCD_HIP_OK;
}

int bcd_hip_multi_rank_configure(bcd_hip_multi *m, int W, int H, int D, int nb_scales, const bcd_hip_params *prm, int *first_input_line,
                                 int *nb_input_lines, int *first_owned_line, int *nb_owned_lines)
{
    if (!m) return BCD_HIP_EINVAL;
    reset_error(m);
    if (m->local_rank < 0) { fail(m, "not a one-rank handle"); return BCD_HIP_EINVAL; }
    Job job;
    int rc = make_job(m, job, W, H, D, nb_scales, prm);
    if (rc != BCD_HIP_OK) return rc;
    rc = prepare(m, nb_scales);
    if (rc != BCD_HIP_OK) return rc;
    m->frame.W = W; m->frame.H = H; m->frame.D = D; m->frame.S = nb_scales; m->frame.prm = *prm; m->frame.set = true;
    ScaleBand bands[MAX_S];
    job.geom.bands(m->local_rank, bands);
    if (first_input_line) *first_input_line = bands[0].loc0;
    if (nb_input_lines) *nb_input_lines = bands[0].loc1 - bands[0].loc0;
    if (first_owned_line) *first_owned_line = bands[0].own0;
    if (nb_owned_lines) *nb_owned_lines = bands[0].own1 - bands[0].own0;
    return BCD_HIP_OK;
}

static int rank_job(bcd_hip_multi *m, Job &job)
{
    if (!m) return BCD_HIP_EINVAL;
    reset_error(m);
    if (m->local_rank < 0 || !m->frame.set) { fail(m, "bcd_hip_multi_rank_configure has not been called"); return BCD_HIP_EINVAL; }
    return make_job(m, job, m->frame.W, m->frame.H, m->frame.D, m->frame.S, &m->frame.prm);
}

int bcd_hip_multi_rank_upload(bcd_hip_multi *m, const float *h_colors, const float *h_ns, const float *h_hist, const float *h_cov)
{
    Job job;
    int rc = rank_job(m, job);
    if (rc != BCD_HIP_OK) return rc;
    if (!h_colors || !h_ns || !h_hist || !h_cov) { fail(m, "null pointer"); return BCD_HIP_EINVAL; }
    job.h_col = h_colors; job.h_ns = h_ns; job.h_hist = h_hist; job.h_cov = h_cov;
    return rank_upload(job, m->local_rank, true) ? BCD_HIP_OK : BCD_HIP_EDEVICE;
}

int bcd_hip_multi_rank_step(bcd_hip_multi *m)
{
    Job job;
    int rc = rank_job(m, job);
    if (rc != BCD_HIP_OK) return rc;
    { std::lock_guard<std::mutex> lk(m->trace_mutex[m->local_rank]); m->trace[m->local_rank].clear(); }
    FrameWatchdog watchdog(m);
    if (!rank_compute(job, m->local_rank)) return BCD_HIP_EDEVICE;
    m->stats.frames += 1;
    return BCD_HIP_OK;
}

int bcd_hip_multi_rank_download(bcd_hip_multi *m, float *h_out_owned)
{
    Job job;
    int rc = rank_job(m, job);
    if (rc != BCD_HIP_OK) return rc;
    if (!h_out_owned) { fail(m, "null pointer"); return BCD_HIP_EINVAL; }
    job.h_out = h_out_owned;
    return rank_download(job, m->local_rank, true) ? BCD_HIP_OK : BCD_HIP_EDEVICE;
}

} // extern "C"

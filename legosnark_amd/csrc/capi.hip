// legosnark_amd/csrc/capi.hip -- implementation of the C-ABI in include/legosnark_amd.h.
// No CPU fallback: every compute entry point requires lsa_init() to have found a gfx950
// device and fails with LSA_ERR_NO_DEVICE otherwise.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <sys/random.h>

#include "capi_internal.h"
#include "tower.h"
#include "ntt_core.h"

namespace lsa {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

State g;

// ---- keyed content fingerprints (the CRS cache and the G2 line-table cache): see hash_words below
namespace {
constexpr size_t NH_KEY_WORDS = 4096;           // 32 KiB: longer messages are hashed block by block
struct NhKey {
    uint64_t k[NH_KEY_WORDS + 8];
    NhKey() {
        size_t got = 0;
        while (got < sizeof k) {
            const ssize_t r = getrandom((char *)k + got, sizeof k - got, 0);
            if (r <= 0) break;
            got += (size_t)r;
        }
        if (got < sizeof k) {                   // no CSPRNG (never seen on Linux >= 3.17): time- and address-seeded xorshift
            uint64_t x = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ (uint64_t)(uintptr_t)this;
            for (size_t i = got / 8; i < sizeof k / 8; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; k[i] = x * 0x9E3779B97F4A7C15ull; }
        }
        for (size_t i = NH_KEY_WORDS; i < NH_KEY_WORDS + 8; i++) k[i] |= 1;     // multipliers of the block chain / the fold: odd
    }
};
const NhKey &nh_key() { static const NhKey key; return key; }
}  // namespace
void keyed_hash128(const void *bytes, size_t nbytes, uint64_t tweak, uint64_t out[2]) {
    typedef unsigned __int128 u128;
    const NhKey &K = nh_key();
    const uint64_t *w = (const uint64_t *)bytes;
    const size_t nw = nbytes / 8;
    // the length and the tweak enter as a message pair of their own, so lengths and kinds never collide by construction
    u128 acc = (u128)(K.k[NH_KEY_WORDS + 1] + (uint64_t)nbytes) * (K.k[NH_KEY_WORDS + 2] + tweak);
    const u128 chain = ((u128)K.k[NH_KEY_WORDS + 3] << 64) | K.k[NH_KEY_WORDS];
    for (size_t base = 0; base < nw; base += NH_KEY_WORDS) {
        const size_t m = nw - base < NH_KEY_WORDS ? nw - base : NH_KEY_WORDS;
        const uint64_t *p = w + base;
        u128 a0 = 0, a1 = 0;                    // two independent sums: the multiplier pipe stays full
        size_t i = 0;
        for (; i + 4 <= m; i += 4) {
            a0 += (u128)(p[i] + K.k[i]) * (p[i + 1] + K.k[i + 1]);
            a1 += (u128)(p[i + 2] + K.k[i + 2]) * (p[i + 3] + K.k[i + 3]);
        }
        for (; i + 2 <= m; i += 2) a0 += (u128)(p[i] + K.k[i]) * (p[i + 1] + K.k[i + 1]);
        if (i < m) a0 += (u128)(p[i] + K.k[i]) * K.k[i + 1];                    // odd tail: the missing word is zero
        acc = acc * chain + a0 + a1;
    }
    if (nbytes & 7) {                            // trailing bytes (never the case for the callers here)
        uint64_t t = 0;
        memcpy(&t, (const char *)bytes + nw * 8, nbytes & 7);
        acc += (u128)(t + K.k[NH_KEY_WORDS + 4]) * K.k[NH_KEY_WORDS + 5];
    }
    out[0] = (uint64_t)acc;
    out[1] = (uint64_t)(acc >> 64);
}
uint64_t keyed_hash64(const void *bytes, size_t nbytes, uint64_t tweak) {
    uint64_t h[2];
    keyed_hash128(bytes, nbytes, tweak, h);
    // 128 -> 64 bits by a keyed multilinear map (high half of lo * a + hi * b)
    const NhKey &K = nh_key();
    typedef unsigned __int128 u128;
    const u128 t = (u128)h[0] * K.k[NH_KEY_WORDS + 6] + (u128)h[1] * K.k[NH_KEY_WORDS + 7];
    return (uint64_t)(t >> 64);
}

int require_ready() {
    if (!g.ready) {
        set_error("legosnark_amd: no initialised gfx950 device (call lsa_init first; there is no CPU fallback)");
        return LSA_ERR_NO_DEVICE;
    }
    return LSA_OK;
}

}  // namespace lsa

using namespace lsa;

static void release_stage_buffers();
static void warm_stage_buffers();
static void crs_cache_clear();

extern "C" {

int lsa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *lsa_last_error(void) { return g_err; }
}  // extern "C"
namespace lsa {
bool trace_on() {
    static const bool on = getenv("LSA_TRACE") && (getenv("LSA_TRACE")[0] == '1' || getenv("LSA_TRACE")[0] == '2');
    return on;
}
double CallTrace::now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace lsa
extern "C" {

int lsa_init(int device) {
    if (g.ready && g.device == device) return LSA_OK;
    if (g.ready) lsa_shutdown();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("lsa_init: no HIP device visible (there is no CPU fallback)");
        return LSA_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("lsa_init: device %d out of range (%d visible)", device, n);
        return LSA_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("lsa_init: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return LSA_ERR_NO_DEVICE;
    }
    HIPCHK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    HIPCHK(hipMalloc(&g.d_result, 512));
    HIPCHK(hipHostMalloc(&g.h_result, 512, hipHostMallocDefault));
    g.device = device;
    upload_prepare();
    const int wrc = msm_warmup(g.stream);
    warm_stage_buffers();
    g.ready = true;                                  // (lsa_shutdown releases what the steps above created)
    if (wrc) { lsa_shutdown(); return wrc; }         // a library whose warm-up failed is not handed out as ready
    return LSA_OK;
}

void lsa_shutdown(void) {
    if (!g.ready) return;
    (void)hipStreamSynchronize(g.stream);
    comm_release();
    crs_cache_clear();
    msm_release_workspace();
    batch_exp_release();
    release_stage_buffers();
    upload_release();
    (void)hipFree(g.d_result);
    (void)hipHostFree(g.h_result);
    (void)hipStreamDestroy(g.stream);
    g = State();
}

void *lsa_stream(void) { return g.ready ? (void *)g.stream : nullptr; }

int lsa_stream_join(void) {
    int rc = require_ready();
    if (rc) return rc;
    return msm_join(g.stream);
}

int lsa_synchronize(void) {
    int rc = require_ready();
    if (rc) return rc;
    rc = msm_join(g.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    return LSA_OK;
}

unsigned lsa_msm_window_bits(size_t n) { return msm_window_bits(2 * n); }   // G1 (GLV: 2n virtual scalars)

int lsa_profile_enable(int on) { msm_profile_enable(on != 0); return LSA_OK; }
int lsa_profile_last_msm(float ms[LSA_MSM_STAGES]) { return msm_profile_last(ms); }

}  // extern "C"

// ---------------------------------------------------------------- staged host <-> device copies (capi_internal.h)
namespace lsa {
namespace {
class HostCopier {
    size_t SLOT = (size_t)2 << 20;                    // LSA_H2D_SLOT_KB (experiments)
    static constexpr unsigned MAX_WORKERS = 12, SLOTS_PER_WORKER = 2;
    struct Slot { void *p = nullptr; hipEvent_t ev = nullptr; bool pending = false; };
    struct Worker { Slot slot[SLOTS_PER_WORKER]; unsigned turn = 0; };
    Worker w_[MAX_WORKERS];
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, cv_done_;
    uint64_t gen_ = 0;
    unsigned active_ = 0;
    bool stop_ = false, broken_ = false;
    // the job
    char *dev_ = nullptr, *host_ = nullptr;
    bool down_ = false;
    size_t bytes_ = 0, nchunks_ = 0, chunk_ = 0;
    std::atomic<size_t> next_{0};
    std::atomic<int> err_{0};

    void *arena_ = nullptr;                           // every slot, one pinned allocation
    unsigned nworkers_ = 0;
    // Uploads of two slots and more leave lsa_stream(): worker i issues its copies on copy stream i % ncs_, so that
    // several copy engines run at once (one stream = one engine at a time: 23 GB/s measured, against 40-50 for the
    // runtime's own large copy) and a kernel the caller queues on lsa_stream() between two uploads (the normalisation
    // of the previous wave of a CRS vector) does not hold the next copies back.  run() orders the streams: they wait
    // for what lsa_stream() holds at the start (`order_after`), lsa_stream() waits for them at the end.
    static constexpr unsigned MAX_STREAMS = 4;
    hipStream_t cs_[MAX_STREAMS] = {};
    hipEvent_t ev_begin_ = nullptr, ev_done_[MAX_STREAMS] = {};
    unsigned ncs_ = 0;
    bool on_cs_ = false;
    bool slot_ready(Slot &s) {
        if (!s.p || !s.ev) return false;
        if (s.pending) { if (hipEventSynchronize(s.ev) != hipSuccess) return false; s.pending = false; }
        return true;
    }
    // threads, the pinned slots and their events, all at once (lsa_init: not inside somebody's first timed call --
    // sixteen separate pinned allocations took 8 ms of the unchanged hadamard's Lipmaa prover)
    bool prepare_locked() {
        if (arena_) return true;
        const unsigned hw = std::thread::hardware_concurrency();
        const char *e = getenv("LSA_H2D_THREADS");
        // one thread moves ~10 GB/s through a pinned slot; the link takes ~55: ten on a host with cores to spare
        unsigned want = e ? (unsigned)atoi(e) : (hw >= 32 ? 10 : (hw > 8 ? 6 : (hw > 2 ? hw / 2 : 1)));
        if (want < 1) want = 1;
        if (want > MAX_WORKERS) want = MAX_WORKERS;
        if (const char *sk = getenv("LSA_H2D_SLOT_KB")) { const size_t kb = (size_t)atol(sk); if (kb >= 64 && kb <= 16384) SLOT = kb << 10; }
        if (hipHostMalloc(&arena_, (size_t)want * SLOTS_PER_WORKER * SLOT, hipHostMallocDefault) != hipSuccess) { arena_ = nullptr; return false; }
        for (unsigned i = 0; i < want; i++)
            for (unsigned k = 0; k < SLOTS_PER_WORKER; k++) {
                Slot &s = w_[i].slot[k];
                s.p = (char *)arena_ + ((size_t)i * SLOTS_PER_WORKER + k) * SLOT;
                s.pending = false;
                if (hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) != hipSuccess) { s.ev = nullptr; return false; }
            }
        nworkers_ = want;
        const char *se = getenv("LSA_H2D_STREAMS");
        // ONE copy stream by default: enough to run the copies beside the kernels of lsa_stream() (96 MiB of first-sight
        // bases: 3.3 ms, 4.3 on lsa_stream() itself); with two to four, 40 % of the first calls of fresh processes took
        // another 7 ms (more copy engines and queues brought up lazily, in whichever call first overlaps enough copies)
        unsigned ns = se ? (unsigned)atoi(se) : 1u;
        if (ns > MAX_STREAMS) ns = MAX_STREAMS;
        if (ns > want) ns = want;
        ncs_ = 0;
        if (ns && hipEventCreateWithFlags(&ev_begin_, hipEventDisableTiming) == hipSuccess) {
            for (unsigned k = 0; k < ns; k++) {
                if (hipStreamCreateWithFlags(&cs_[k], hipStreamNonBlocking) != hipSuccess) { cs_[k] = nullptr; break; }
                if (hipEventCreateWithFlags(&ev_done_[k], hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(cs_[k]); cs_[k] = nullptr; ev_done_[k] = nullptr; break; }
                ncs_ = k + 1;
            }
        }
        (void)hipGetLastError();
        for (unsigned i = 1; i < want; i++) th_.emplace_back([this, i] { loop(i); });
        return true;
    }
    void work(unsigned id) {
        if (id) (void)hipSetDevice(g.device);
        Worker &w = w_[id];
        for (;;) {
            const size_t c = next_.fetch_add(1);
            if (c >= nchunks_) break;
            const size_t lo = c * chunk_, len = bytes_ - lo < chunk_ ? bytes_ - lo : chunk_;
            Slot &s = w.slot[w.turn++ % SLOTS_PER_WORKER];
            if (!slot_ready(s)) { err_ = 1; continue; }
            if (down_) {
                if (hipMemcpyAsync(s.p, dev_ + lo, len, hipMemcpyDeviceToHost, g.stream) != hipSuccess || hipEventRecord(s.ev, g.stream) != hipSuccess ||
                    hipEventSynchronize(s.ev) != hipSuccess) { err_ = 1; continue; }
                memcpy(host_ + lo, s.p, len);
            } else {
                hipStream_t st = on_cs_ ? cs_[id % ncs_] : g.stream;
                memcpy(s.p, host_ + lo, len);
                if (hipMemcpyAsync(dev_ + lo, s.p, len, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(s.ev, st) != hipSuccess) { err_ = 1; continue; }
                s.pending = true;
            }
        }
    }
    void loop(unsigned id) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            work(id);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--active_ == 0) cv_done_.notify_all();
            }
        }
    }

  public:
    ~HostCopier() { stop_threads(); }
    // 0: done (up: every copy is on the stream; down: the bytes are in host memory), 1: not taken, < 0: LSA error
    int run(void *dev, void *host, size_t bytes, bool down, bool order_after = true) {
        if (broken_) return 1;
        dev_ = (char *)dev; host_ = (char *)host; bytes_ = bytes; down_ = down;
        // whole slots for large copies; a copy of 0.5 .. 4 MiB (the points of a 2^12-pair product: 0.4 + 0.8 MiB) is cut into
        // up to four pieces of >= 256 KiB so that its memcpy is shared too (one thread moves ~10 GB/s: 80 us per 0.8 MiB)
        chunk_ = SLOT;
        if (bytes >= ((size_t)512 << 10) && bytes < 2 * SLOT) {
            const size_t pieces = bytes / ((size_t)256 << 10) < 4 ? bytes / ((size_t)256 << 10) : 4;
            chunk_ = ((bytes + pieces - 1) / pieces + 4095) & ~(size_t)4095;
            if (chunk_ > SLOT) chunk_ = SLOT;
        }
        nchunks_ = (bytes + chunk_ - 1) / chunk_;
        next_.store(0);
        err_.store(0);
        bool alone = nchunks_ < 2;                        // one piece: not worth waking anybody
        {
            std::lock_guard<std::mutex> lk(m_);
            if (!prepare_locked()) { broken_ = true; return 1; }
            if (th_.empty()) alone = true;
            on_cs_ = !down && !alone && ncs_ > 0;
            if (on_cs_ && order_after) {
                bool ok = hipEventRecord(ev_begin_, g.stream) == hipSuccess;
                for (unsigned k = 0; k < ncs_ && ok; k++) ok = hipStreamWaitEvent(cs_[k], ev_begin_, 0) == hipSuccess;
                if (!ok) { (void)hipGetLastError(); on_cs_ = false; }
            }
            if (!alone) {
                active_ = (unsigned)th_.size();
                gen_++;
            }
        }
        if (!alone) cv_.notify_all();
        work(0);
        if (!alone) {
            std::unique_lock<std::mutex> lk(m_);
            cv_done_.wait(lk, [&] { return active_ == 0; });
        }
        if (on_cs_) {
            for (unsigned k = 0; k < ncs_; k++)
                if (hipEventRecord(ev_done_[k], cs_[k]) != hipSuccess || hipStreamWaitEvent(g.stream, ev_done_[k], 0) != hipSuccess) err_ = 1;
            on_cs_ = false;
        }
        if (err_.load()) {
            broken_ = true;
            set_error("a staged host <-> device copy failed (%s)", hipGetErrorString(hipGetLastError()));
            return LSA_ERR_HIP;
        }
        return 0;
    }
    void stop_threads() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) if (t.joinable()) t.join();
        th_.clear();
        std::lock_guard<std::mutex> lk(m_);
        stop_ = false;
        gen_ = 0;
        active_ = 0;
    }
    void prepare() {
        std::lock_guard<std::mutex> lk(m_);
        if (!prepare_locked()) broken_ = true;
    }
    // the first pinned transfer on a stream sets up that stream's copy-engine queue (~7 ms each, measured inside the
    // first CRS upload of a process): paid in lsa_init, per copy stream and per size class a slot can carry
    // (the runtime adds copy engines when copies overlap, and each new engine is another such set-up: the warm-up keeps
    // every copy stream busy at the same time, ordered against lsa_stream() exactly as run() orders them)
    void warm_streams() {
        std::lock_guard<std::mutex> lk(m_);
        if (!arena_ || !ncs_) return;
        void *d = nullptr;
        if (hipMalloc(&d, (size_t)ncs_ * SLOT) != hipSuccess) { (void)hipGetLastError(); return; }
        (void)hipEventRecord(ev_begin_, g.stream);
        for (unsigned k = 0; k < ncs_; k++) (void)hipStreamWaitEvent(cs_[k], ev_begin_, 0);
        for (int rep = 0; rep < 12; rep++)
            for (unsigned k = 0; k < ncs_; k++)
                (void)hipMemcpyAsync((char *)d + (size_t)k * SLOT, (char *)arena_ + (size_t)k * SLOT, rep & 1 ? SLOT / 4 : SLOT, hipMemcpyHostToDevice, cs_[k]);
        for (unsigned k = 0; k < ncs_; k++) { (void)hipEventRecord(ev_done_[k], cs_[k]); (void)hipStreamWaitEvent(g.stream, ev_done_[k], 0); }
        (void)hipStreamSynchronize(g.stream);
        for (unsigned k = 0; k < ncs_; k++) (void)hipStreamSynchronize(cs_[k]);
        (void)hipFree(d);
        (void)hipGetLastError();
    }
    void release() {
        stop_threads();
        for (auto &w : w_)
            for (auto &s : w.slot) {
                if (s.ev) { if (s.pending) (void)hipEventSynchronize(s.ev); (void)hipEventDestroy(s.ev); }
                s = Slot();
            }
        for (unsigned k = 0; k < MAX_STREAMS; k++) {
            if (cs_[k]) { (void)hipStreamSynchronize(cs_[k]); (void)hipStreamDestroy(cs_[k]); }
            if (ev_done_[k]) (void)hipEventDestroy(ev_done_[k]);
            cs_[k] = nullptr; ev_done_[k] = nullptr;
        }
        if (ev_begin_) (void)hipEventDestroy(ev_begin_);
        ev_begin_ = nullptr;
        ncs_ = 0;
        if (arena_) (void)hipHostFree(arena_);
        arena_ = nullptr;
        nworkers_ = 0;
        broken_ = false;
    }
};
HostCopier g_copier;
// 0 auto, 1 direct, 2 staged at every size
int copy_mode() {
    static const int mode = [] {
        const char *e = getenv("LSA_H2D");
        return !e ? 0 : (strcmp(e, "direct") == 0 ? 1 : (strcmp(e, "staged") == 0 ? 2 : 0));
    }();
    return mode;
}
constexpr size_t STAGE_FROM = (size_t)32 << 10;       // below: the runtime copies through its own staging buffer
// from here on the runtime's own path (pin the caller's pages, one DMA) wins: a staged copy is sixteen and more
// commands and six threads of memcpy -- 32 MiB: 0.3-0.4 ms later on the device than the direct copy, and the
// fingerprint threads get less of the host.  (Buffers this large are the ones whose release stalls the GPU when the
// runtime has pinned them: see capi_internal.h; the shim keeps them on the heap, other callers keep them alive.)
constexpr size_t STAGE_BELOW = (size_t)16 << 20;
static bool staged(size_t bytes) { return copy_mode() != 1 && ((bytes >= STAGE_FROM && bytes < STAGE_BELOW) || copy_mode() == 2); }
// A host range the library has NOT been handed before (the reference's provers: one multiExpMA per key vector and
// process, src/gadgets/subspace.cc:78-85; fresh std::vectors of scalars per call) is the other case: the runtime's direct
// path first has to pin its pages, 4 KiB at a time unless the kernel backs the range with huge pages -- 5-20 ms per
// 96 MiB depending on the box, where the slots move it at the link rate whatever the pages are.  The last few large
// ranges are remembered; a range goes through the slots until it has been seen LSA_H2D_DIRECT_AFTER (default 2) times,
// from then on the runtime's registration is worth its price (bench loops, provers that keep their vectors).
struct SeenRange { const void *p = nullptr; size_t bytes = 0; unsigned count = 0; uint64_t tick = 0; };
static SeenRange g_seen[16];
static uint64_t g_seen_tick = 0;
static unsigned direct_after() {
    static const unsigned v = [] { const char *e = getenv("LSA_H2D_DIRECT_AFTER"); return e ? (unsigned)atoi(e) : 2u; }();
    return v;
}
static bool large_copy_staged(const void *h, size_t bytes) {
    if (copy_mode() == 1) return false;
    if (copy_mode() == 2) return true;
    SeenRange *slot = &g_seen[0];
    for (auto &r : g_seen) {
        if (r.p == h && r.bytes == bytes) { r.tick = ++g_seen_tick; return r.count++ < direct_after(); }
        if (r.tick < slot->tick) slot = &r;
    }
    *slot = SeenRange{h, bytes, 1, ++g_seen_tick};
    return direct_after() > 0;
}
}  // namespace

bool upload_takes_slots(const void *h_src, size_t bytes) {
    return staged(bytes) || (bytes >= STAGE_BELOW && large_copy_staged(h_src, bytes));
}
int upload_host_as(void *d_dst, const void *h_src, size_t bytes, bool slots, bool order_after) {
    if (bytes == 0) return LSA_OK;
    if (slots) {
        const int rc = g_copier.run(d_dst, const_cast<void *>(h_src), bytes, false, order_after);
        if (rc <= 0) return rc;
    }
    HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, g.stream));
    return LSA_OK;
}
int upload_host(void *d_dst, const void *h_src, size_t bytes) {
    return upload_host_as(d_dst, h_src, bytes, bytes != 0 && upload_takes_slots(h_src, bytes), true);
}
int download_host(void *h_dst, const void *d_src, size_t bytes) {
    if (bytes == 0) return LSA_OK;
    // (large destinations the library has not seen before go through the slots as well: the runtime would pin the caller's
    // pages and keep them registered, and a caller that frees such a buffer afterwards -- a Python array, a temporary
    // std::vector outside the shim's allocator settings -- stalls its next GPU submission by 12-25 ms when the pages are
    // unmapped: measured as a 15.5 ms bubble in front of the first MSM after a 96-MiB download, 26 ms after 192 MiB)
    if (staged(bytes) || (bytes >= STAGE_BELOW && large_copy_staged(h_dst, bytes))) {
        const int rc = g_copier.run(const_cast<void *>(d_src), h_dst, bytes, true);
        if (rc <= 0) return rc;
    }
    HIPCHK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return LSA_OK;
}
void upload_release() { g_copier.release(); }
void upload_prepare() {
    if (copy_mode() == 1) return;
    g_copier.prepare();
    // the runtime sets up a copy engine's queue at the first pinned transfer of a direction and size class (measured:
    // 8.4 ms inside the first 128-KiB download of a process that had only done larger ones; none with
    // HSA_ENABLE_SDMA=0): pay that here for every size a slot can carry, not inside a caller's first NTT
    void *d = nullptr;
    if (hipMalloc(&d, (size_t)4 << 20) != hipSuccess) { (void)hipGetLastError(); return; }
    g_copier.warm_streams();
    std::vector<char> h((size_t)4 << 20, 0);
    for (size_t bytes = STAGE_FROM; bytes <= ((size_t)4 << 20); bytes <<= 1) {
        if (upload_host(d, h.data(), bytes) != LSA_OK) break;
        (void)hipStreamSynchronize(g.stream);                       // (a download on an idle stream takes another path)
        if (download_host(h.data(), d, bytes) != LSA_OK) break;
    }
    (void)hipStreamSynchronize(g.stream);
    (void)hipFree(d);
}
}  // namespace lsa
extern "C" {
// ---------------------------------------------------------------- bases
}  // extern "C"
static int stage_jac_ensure(size_t bytes, void **p);      // g_stage_jac (below)
template <class F>
static int bases_create(const void *bases_jac, size_t n, int src_on_device, int group, lsa_bases **out, bool allow_table = true) {
    int rc = require_ready();
    if (rc) return rc;
    if (!out || (n && !bases_jac)) { set_error("bases_create: null argument"); return LSA_ERR_INVALID; }
    lsa_bases *b = new lsa_bases();
    b->n = n;
    b->group = group;
    if (n) {
        const Jac<F> *d_in = nullptr;
        void *tmp = nullptr;
        // Large resident CRS vectors also keep the pre-shifted windows (msm.hip, "merged
        // windows"): nwin x the memory, no Horner fold per MSM.  LSA_PRECOMPUTE=0 opts out; a
        // table that does not fit falls back to the plain layout.
        const char *pe = getenv("LSA_PRECOMPUTE");
        bool table = allow_table && n >= msm_merge_min() && !(pe && pe[0] == '0');
        // G1 handles of up to 2^16 points: all copies in ONE kernel (msm_compact.hip: one inversion per point instead
        // of 25, ~1.2 ms whatever n) -- cheap enough that every such handle carries them unless a threshold was set
        // explicitly, and their MSMs of up to msm_compact_max() pairs take the four-launch pipeline
        // (G2 handles too since round 5: CommScheme::commit's second half, src/prototools/commit.h:155)
        const bool fast_build = n <= ((size_t)1 << 16) && allow_table && !(pe && pe[0] == '0') &&
                                (table || !msm_merge_min_is_explicit());
        if (fast_build) table = true;
        const size_t tw = msm_table_windows(group, n);
        if (table && (uint64_t)n * tw >= (1u << 30)) table = false;
        if (table && hipMalloc(&b->d_aff, tw * n * msm_base_bytes(group)) != hipSuccess) { (void)hipGetLastError(); b->d_aff = nullptr; table = false; }
        if (!table && hipMalloc(&b->d_aff, n * msm_base_bytes(group)) != hipSuccess) {
            delete b;
            set_error("bases_create: hipMalloc of %zu bytes failed", n * msm_base_bytes(group));
            return LSA_ERR_NOMEM;
        }
        bool prepared = false;
        if (src_on_device) {
            d_in = (const Jac<F> *)bases_jac;
        } else {
            // the Jacobian staging area: the library's grow-only buffer up to 512 MiB (no hipMalloc / hipFree -- each an
            // implicit device synchronisation and 0.5-2 ms -- inside a prover's first MSM), a temporary beyond
            const size_t jbytes = n * sizeof(Jac<F>);
            void *d_stage = nullptr;
            if (jbytes <= ((size_t)512 << 20)) {
                if (stage_jac_ensure(jbytes, &d_stage)) d_stage = nullptr;
            }
            if (!d_stage) {
                if (hipMalloc(&tmp, jbytes) != hipSuccess) {
                    (void)hipFree(b->d_aff); delete b;
                    set_error("bases_create: hipMalloc of %zu bytes failed", jbytes);
                    return LSA_ERR_NOMEM;
                }
                d_stage = tmp;
            }
            d_in = (const Jac<F> *)d_stage;
            // In waves: while the host threads fill the pinned slots of wave k + 1, the copy engine moves wave k and
            // the normalisation kernel of wave k - 1 runs (affine coordinates are unique: the same bytes as one
            // prepare_bases over the whole vector).  One wave for short vectors.
            // A vector the runtime has pinned by now (seen several times) goes as one direct copy, as before.
            const bool slots = upload_takes_slots(bases_jac, jbytes);
            const size_t wave_pts = slots ? ((((size_t)16 << 20) / sizeof(Jac<F>)) & ~(size_t)4095) : n;
            int urc = LSA_OK;
            for (size_t lo = 0; lo < n && !urc; lo += wave_pts) {
                const size_t cnt = n - lo < wave_pts ? n - lo : wave_pts;
                urc = upload_host_as((char *)d_stage + lo * sizeof(Jac<F>), (const char *)bases_jac + lo * sizeof(Jac<F>), cnt * sizeof(Jac<F>), slots, lo == 0);
                if (!urc) urc = prepare_bases<F>(d_in + lo, (char *)b->d_aff + lo * msm_base_bytes(group), cnt, g.stream);
            }
            if (urc) { (void)hipStreamSynchronize(g.stream); if (tmp) (void)hipFree(tmp); (void)hipFree(b->d_aff); delete b; return urc; }
            prepared = true;
        }
        if (!prepared) rc = prepare_bases<F>(d_in, b->d_aff, n, g.stream);
        void *scratch = nullptr;
        if (!rc && table && fast_build) {
            if (hipMalloc(&scratch, table_build_scratch_bytes(n, group)) != hipSuccess) { (void)hipGetLastError(); scratch = nullptr; set_error("bases_create: scratch allocation failed"); rc = LSA_ERR_NOMEM; }
            if (!rc) rc = group == 1 ? table_build_g1_device(b->d_aff, n, n, scratch, g.stream) : table_build_g2_device(b->d_aff, n, n, scratch, g.stream);
            if (!rc) b->table_stride = n;
        } else if (!rc && table) {
            rc = precompute_windows<F>(b->d_aff, n, g.stream);
            if (!rc) b->table_stride = n;
        }
        hipError_t e = hipStreamSynchronize(g.stream);
        if (tmp) (void)hipFree(tmp);
        if (scratch) (void)hipFree(scratch);
        if (rc || e != hipSuccess) {
            if (!rc) { set_error("bases_create: kernel failed: %s", hipGetErrorString(e)); rc = LSA_ERR_HIP; }
            (void)hipFree(b->d_aff); delete b;
            return rc;
        }
    }
    *out = b;
    return LSA_OK;
}

extern "C" {
int lsa_g1_bases_create(const void *bases_jac, size_t n, int src_on_device, lsa_bases **out) {
    return bases_create<Fq>(bases_jac, n, src_on_device, 1, out);
}
int lsa_g2_bases_create(const void *bases_jac, size_t n, int src_on_device, lsa_bases **out) {
    return bases_create<Fq2>(bases_jac, n, src_on_device, 2, out);
}
void lsa_bases_destroy(lsa_bases *b) {
    if (!b) return;
    if (b->d_aff) (void)hipFree(b->d_aff);
    delete b;
}
size_t lsa_bases_size(const lsa_bases *b) { return b ? b->n : 0; }
int lsa_bases_has_table(const lsa_bases *b) { return b && b->table_stride ? 1 : 0; }
void lsa_msm_set_table_threshold(size_t n) { msm_set_merge_min(n); }
unsigned lsa_msm_field_mults_per_pair(const lsa_bases *b, size_t n) { return msm_field_mults_per_pair(n, b ? b->table_stride : 0); }
unsigned lsa_bases_table_windows(const lsa_bases *b) { return b && b->table_stride ? msm_table_windows(b->group, b->n) : 0; }
const void *lsa_bases_device_ptr(const lsa_bases *b) { return b ? b->d_aff : nullptr; }

// ---------------------------------------------------------------- MSM
int lsa_msm_run_async(const lsa_bases *bases, size_t first, const void *d_scalars, size_t n, void *d_out_jac) {
    int rc = require_ready();
    if (rc) return rc;
    if (!bases || !d_out_jac || (n && !d_scalars)) { set_error("msm_run: null argument"); return LSA_ERR_INVALID; }
    if (first > bases->n || n > bases->n - first) { set_error("msm_run: range [%zu,%zu) exceeds %zu bases", first, first + n, bases->n); return LSA_ERR_INVALID; }
    if (bases->group == 1)
        return msm_device<Fq>(bases->d_aff, first, (const Fr *)d_scalars, n, (Jac<Fq> *)d_out_jac, g.stream, bases->table_stride);
    return msm_device<Fq2>(bases->d_aff, first, (const Fr *)d_scalars, n, (Jac<Fq2> *)d_out_jac, g.stream, bases->table_stride);
}

int lsa_msm_run_segments_async(const lsa_bases *bases, size_t first, const void *d_scalars, const uint64_t *seg_offsets, size_t nseg,
                               void *d_out_jac) {
    int rc = require_ready();
    if (rc) return rc;
    if (nseg == 0) return LSA_OK;
    if (!bases || !d_out_jac || !seg_offsets || !d_scalars) { set_error("msm_run_segments: null argument"); return LSA_ERR_INVALID; }
    if (!bases->table_stride) { set_error("msm_run_segments: the bases carry no pre-shifted copies (lsa_bases_has_table)"); return LSA_ERR_INVALID; }
    for (size_t j = 0; j < nseg; j++) {
        if (seg_offsets[j + 1] < seg_offsets[j]) { set_error("msm_run_segments: offsets must not decrease"); return LSA_ERR_INVALID; }
        const size_t len = seg_offsets[j + 1] - seg_offsets[j];
        if (first > bases->n || len > bases->n - first) { set_error("msm_run_segments: segment %zu of %zu pairs exceeds the %zu bases", j, len, bases->n); return LSA_ERR_INVALID; }
    }
    if (bases->group == 1)
        return msm_segments_device<Fq>(bases->d_aff, first, (const Fr *)d_scalars, seg_offsets, nseg, (Jac<Fq> *)d_out_jac, g.stream, bases->table_stride);
    return msm_segments_device<Fq2>(bases->d_aff, first, (const Fr *)d_scalars, seg_offsets, nseg, (Jac<Fq2> *)d_out_jac, g.stream, bases->table_stride);
}

int lsa_commit_run_async(const lsa_bases *g1_bases, const lsa_bases *g2_bases, const void *d_scalars, size_t n, void *d_out_g1, void *d_out_g2) {
    int rc = require_ready();
    if (rc) return rc;
    if (!g1_bases || !g2_bases || !d_out_g1 || !d_out_g2 || (n && !d_scalars)) { set_error("commit_run: null argument"); return LSA_ERR_INVALID; }
    if (g1_bases->group != 1 || g2_bases->group != 2) { set_error("commit_run: needs a G1 and a G2 handle, in that order"); return LSA_ERR_INVALID; }
    if (n > g1_bases->n || n > g2_bases->n) { set_error("commit_run: %zu pairs exceed the bases (%zu, %zu)", n, g1_bases->n, g2_bases->n); return LSA_ERR_INVALID; }
    // one shared sort when both handles carry copies over the same number of points AND an MSM of n pairs runs
    // over them (an explicit table threshold above n sends both to the plain pipeline); else two calls
    if (n && g1_bases->table_stride && g1_bases->table_stride == g2_bases->table_stride && msm_uses_table(n))
        return msm_commit_pair_device(g1_bases->d_aff, g2_bases->d_aff, (const Fr *)d_scalars, n, (Jac<Fq> *)d_out_g1, (Jac<Fq2> *)d_out_g2, g.stream,
                                      g1_bases->table_stride);
    rc = msm_device<Fq>(g1_bases->d_aff, 0, (const Fr *)d_scalars, n, (Jac<Fq> *)d_out_g1, g.stream, g1_bases->table_stride);
    if (rc) return rc;
    return msm_device<Fq2>(g2_bases->d_aff, 0, (const Fr *)d_scalars, n, (Jac<Fq2> *)d_out_g2, g.stream, g2_bases->table_stride);
}

// Blocking: the tail runs on lsa_stream() itself and the last kernel writes the point straight into pinned host memory
// (no cross-stream hand-over, no copy command behind the kernels: ~35 us of a small call's latency).
int lsa_msm_run(const lsa_bases *bases, size_t first, const void *d_scalars, size_t n, void *out_jac) {
    int rc = require_ready();
    if (rc) return rc;
    if (!bases || !out_jac || (n && !d_scalars)) { set_error("msm_run: null argument"); return LSA_ERR_INVALID; }
    if (first > bases->n || n > bases->n - first) { set_error("msm_run: range [%zu,%zu) exceeds %zu bases", first, first + n, bases->n); return LSA_ERR_INVALID; }
    rc = msm_join(g.stream);                       // earlier asynchronous calls publish before this one (call order)
    if (rc) return rc;
    if (bases->group == 1)
        rc = msm_device<Fq>(bases->d_aff, first, (const Fr *)d_scalars, n, (Jac<Fq> *)g.h_result, g.stream, bases->table_stride, true);
    else
        rc = msm_device<Fq2>(bases->d_aff, first, (const Fr *)d_scalars, n, (Jac<Fq2> *)g.h_result, g.stream, bases->table_stride, true);
    if (rc) return rc;
    size_t bytes = bases->group == 1 ? sizeof(Jac<Fq>) : sizeof(Jac<Fq2>);
    HIPCHK(hipStreamSynchronize(g.stream));
    memcpy(out_jac, g.h_result, bytes);
    return LSA_OK;
}

}  // extern "C"
// grow-only device staging buffers of the host-buffer entry points (StageBuf, capi_internal.h)
namespace {
StageBuf g_stage_jac, g_stage_bases, g_stage_scalars, g_stage_prefix_scratch;
StageBuf g_stage_ntt_a, g_stage_ntt_tmp;         // lsa_fr_ntt: the data of host callers, the second buffer of the passes
// the Fr recursions (lsa_fr_cppoly_witness, lsa_fr_eval_mle): scratch, and for host callers v / r / w -- grow-only like the
// others (round 5; before, every call paid a hipMalloc + hipFree pair per buffer, and the recursions' cached hipGraphs
// (fr_vec.hip) are keyed by these pointers)
StageBuf g_stage_fr_tmp, g_stage_fr_v, g_stage_fr_r, g_stage_fr_w;
StageBuf g_stage_sc_partial, g_stage_sc_out;     // lsa_fr_sumcheck_round: block partials and the coefficients (one call per round of a prover)
}  // namespace
namespace lsa { StageBuf g_stage_gather; }
static int stage_jac_ensure(size_t bytes, void **p) {
    if (g_stage_jac.ensure(bytes)) return -1;
    *p = g_stage_jac.p;
    return 0;
}
// lsa_init: the staging buffers of the host-vector entry points at the BASELINE scale (2^20 scalars, 2^20 G2 points in
// Jacobian form), beside msm_warmup's workspaces: no allocation inside a prover's first multiExpMA (LSA_WARM=0: lazy)
static void warm_stage_buffers() {
    const char *w = getenv("LSA_WARM");
    if (w && w[0] == '0') return;
    // (LSA_WARM_MB scales these with the MSM workspaces: 400 is the default there)
    const char *mb = getenv("LSA_WARM_MB");
    const size_t scale = mb ? (size_t)atoll(mb) : 400;
    if (scale == 0) return;
    const int e1 = g_stage_scalars.ensure(((size_t)34 << 20) * scale / 400), e2 = g_stage_jac.ensure(((size_t)202 << 20) * scale / 400);
    if ((e1 || e2) && trace_on()) fprintf(stderr, "[lsa]   warm-up: staging buffers not allocated\n");
    (void)hipGetLastError();
}
static void release_stage_buffers() {
    g_stage_jac.release(); g_stage_bases.release(); g_stage_scalars.release(); g_stage_gather.release(); g_stage_prefix_scratch.release();
    g_stage_ntt_a.release(); g_stage_ntt_tmp.release();
    g_stage_fr_tmp.release(); g_stage_fr_v.release(); g_stage_fr_r.release(); g_stage_fr_w.release();
    g_stage_sc_partial.release(); g_stage_sc_out.release();
    ntt_release();                                   // the per-domain twiddle tables (ntt.hip)
    fr_vec_release();                                // the cached hipGraphs of the Fr recursions (fr_vec.hip)
    pairing_release();
}

// ---------------------------------------------------------------- CRS cache behind lsa_g1_msm / lsa_g2_msm
// multiExpMA hands libff the same std::vector<G> on every call of a prover: crs->P in
// SubspaceSnark::prove (src/gadgets/subspace.cc:82), g1s/g2s in CommScheme::commit
// (src/prototools/commit.h:154-155), prefixes of g1s in CPPoly::prove (src/gadgets/poly.h:77-86).
// Uploading 96 B/point over PCIe and normalising it each time costs more than the MSM, so the
// host-buffer entry points keep the prepared (affine, packed; after the first re-use also the
// pre-shifted window copies) bases of recently seen vectors resident, keyed by host pointer and
// VERIFIED by content before every use:
//   mode 2 "full" (default): a 64-bit fingerprint of every 64-point unit, computed by a small
//          host thread pool while the caller thread uploads the scalars; any change of any byte
//          of the vector is a miss (a single changed word always changes its unit's fingerprint).
//          Never reads past the n points the caller passed.
//   mode 1 "sampled": only ~3*log2(n) sampled points are compared -- for callers that promise
//          not to modify a CRS vector in place.
//   mode 0: off.  Env LSA_CRS_CACHE = full | sampled | 0, LSA_CRS_CACHE_MB = budget (LRU by bytes).
// A request (ptr, n) also hits an entry (ptr, n' > n) when n is a multiple of 64: the prefix is
// verified unit-wise (CPPoly::prove's ladder of prefixes 2^k of g1s).
namespace {

constexpr size_t CRS_UNIT_POINTS = 64;          // fingerprint granularity (a prefix of k units can be verified against a longer entry)
constexpr size_t CRS_TASK_UNITS = 64;           // units per hashing task (4096 points)
constexpr unsigned CRS_TABLE_AFTER_DEFAULT = 23;
constexpr size_t CRS_MIN_POINTS = 1024;         // smaller vectors are cheaper to re-upload than to look up

// Content fingerprints are KEYED: NH (the inner hash of UMAC, here on 64-bit words with 128-bit sums:
//   sum_i ((m_2i + k_2i) mod 2^64) * ((m_2i+1 + k_2i+1) mod 2^64) mod 2^128)
// under a per-process key drawn from the kernel's CSPRNG at first use.  For two distinct messages of the same length the
// collision probability over the key is <= 2^-64 whatever the messages are, so a caller who can choose the bytes of a
// vector (or of a G2 point handed to a verifier) but cannot read this process's memory cannot aim at the resident
// entry of another vector.  One multiplication per 16 bytes: faster than the unkeyed multiply-xor lanes it replaces.
inline uint64_t hash_words(const uint64_t *w, size_t nwords) { return lsa::keyed_hash64(w, nwords * 8, 0); }

// persistent workers hashing the chunks of one vector; the caller thread joins in at finish()
class HashPool {
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, cv_done_;
    const uint8_t *base_ = nullptr;
    size_t unit_bytes_ = 0, units_per_task_ = 1, total_bytes_ = 0, nunits_ = 0, ntasks_ = 0;
    uint64_t *out_ = nullptr;
    std::atomic<size_t> next_{0};
    size_t active_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;

    void work() {
        for (;;) {
            size_t t = next_.fetch_add(1);
            if (t >= ntasks_) break;
            size_t u1 = (t + 1) * units_per_task_ < nunits_ ? (t + 1) * units_per_task_ : nunits_;
            for (size_t u = t * units_per_task_; u < u1; u++) {
                size_t lo = u * unit_bytes_, hi = lo + unit_bytes_ < total_bytes_ ? lo + unit_bytes_ : total_bytes_;
                out_[u] = hash_words(reinterpret_cast<const uint64_t *>(base_ + lo), (hi - lo) / 8);   // never past total_bytes_
            }
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            work();
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--active_ == 0) cv_done_.notify_all();
            }
        }
    }

  public:
    ~HashPool() { stop(); }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) if (t.joinable()) t.join();
        th_.clear();
        // back to the initial state: workers started later begin with seen = 0 and must find gen_ = 0, or each
        // would take a phantom first generation and decrement active_ once too often (lsa_shutdown -> lsa_init)
        std::lock_guard<std::mutex> lk(m_);
        stop_ = false;
        gen_ = 0;
        active_ = 0;
    }
    // starts hashing `bytes` bytes (a multiple of 8) in units of unit_bytes (the last one may be
    // short); out: one u64 per unit; a task = units_per_task consecutive units
    void begin(const void *p, size_t bytes, size_t unit_bytes, size_t units_per_task, uint64_t *out) {
        base_ = (const uint8_t *)p; total_bytes_ = bytes; unit_bytes_ = unit_bytes; units_per_task_ = units_per_task; out_ = out;
        nunits_ = (bytes + unit_bytes - 1) / unit_bytes;
        ntasks_ = (nunits_ + units_per_task - 1) / units_per_task;
        next_.store(0);
        if (ntasks_ < 8) return;                     // small: the caller does it alone in finish()
        {
            // workers are spawned inside the block that publishes the generation: none can observe a
            // half-initialised pool
            std::lock_guard<std::mutex> lk(m_);
            if (th_.empty()) {
                unsigned hw = std::thread::hardware_concurrency();
                const char *e = getenv("LSA_HASH_THREADS");
                unsigned want = e ? (unsigned)atoi(e) : (hw > 16 ? 15 : (hw > 1 ? hw - 1 : 0));
                for (unsigned i = 0; i < want; i++) th_.emplace_back([this] { loop(); });
            }
            if (th_.empty()) return;
            active_ = th_.size();
            gen_++;
        }
        cv_.notify_all();
    }
    void finish() {
        work();
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return active_ == 0; });
    }
};

// One background build: allocates the table, fills copy 0 from the entry's prepared bases and derives the other copies
// one at a time on a low-priority stream, each time waiting (at most 2 ms) for lsa_stream() to be idle first -- an MSM
// issued meanwhile overlaps with at most one copy step (~1 ms of the GPU at 2^20 points).
struct TableBuild {
    std::atomic<int> state{0};                      // 0: queued / running, 1: table complete, 2: failed or cancelled (buffers freed)
    std::atomic<bool> cancel{false};
    const void *bases = nullptr;                    // the entry's prepared bases (read-only)
    void *big = nullptr, *tmp = nullptr;
    size_t n = 0;
    int group = 1;
};
class TableBuilder {
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<TableBuild>> q_;
    bool stop_ = false;
    std::atomic<bool> abandon_{false};              // process exit: no more HIP calls from this thread (the runtime may be gone)
    hipStream_t stream_ = nullptr;

    void run(TableBuild &j) {
        const size_t tw = msm_table_windows(j.group, j.n), per = msm_base_bytes(j.group);
        const size_t jac = j.group == 1 ? sizeof(Jac<Fq>) : sizeof(Jac<Fq2>);
        // the call that asked for the build is still running its own MSM: let it finish first (it is the "second call")
        {
            const auto t0 = std::chrono::steady_clock::now();
            std::this_thread::sleep_for(std::chrono::microseconds(200));
            while (hipStreamQuery(g.stream) == hipErrorNotReady && !j.cancel.load() &&
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < 6.0)
                std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        bool ok = hipMalloc(&j.big, tw * j.n * per) == hipSuccess && hipMalloc(&j.tmp, j.n * jac) == hipSuccess;
        ok = ok && hipMemcpyAsync(j.big, j.bases, j.n * per, hipMemcpyDeviceToDevice, stream_) == hipSuccess;
        for (unsigned k = 1; ok && k < tw && !j.cancel.load() && !abandon_.load(); k++) {
            const auto t0 = std::chrono::steady_clock::now();
            while (hipStreamQuery(g.stream) == hipErrorNotReady && !j.cancel.load() &&
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < 2.0)
                std::this_thread::sleep_for(std::chrono::microseconds(30));
            const int rc = j.group == 1 ? precompute_window_step<Fq>(j.big, j.n, j.tmp, k, stream_) : precompute_window_step<Fq2>(j.big, j.n, j.tmp, k, stream_);
            ok = rc == LSA_OK && hipStreamSynchronize(stream_) == hipSuccess;
        }
        if (abandon_.load()) { j.state.store(2); return; }
        if (ok && !j.cancel.load() && hipStreamSynchronize(stream_) == hipSuccess) { j.state.store(1); return; }
        (void)hipGetLastError();
        (void)hipStreamSynchronize(stream_);
        if (j.big) (void)hipFree(j.big);
        if (j.tmp) (void)hipFree(j.tmp);
        j.big = j.tmp = nullptr;
        j.state.store(2);
    }
    void loop(int device) {
        (void)hipSetDevice(device);
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, least) != hipSuccess) stream_ = nullptr;
        for (;;) {
            std::shared_ptr<TableBuild> j;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (stop_ && q_.empty()) break;
                j = q_.front();
                q_.pop_front();
            }
            if (!stream_ || j->cancel.load()) { j->state.store(2); continue; }
            run(*j);
        }
        if (stream_ && !abandon_.load()) (void)hipStreamDestroy(stream_);
        stream_ = nullptr;
    }

  public:
    ~TableBuilder() { abandon_.store(true); shutdown(); }
    void submit(std::shared_ptr<TableBuild> j) {
        std::lock_guard<std::mutex> lk(m_);
        if (!th_.joinable()) { stop_ = false; th_ = std::thread([this, d = g.device] { loop(d); }); }
        q_.push_back(std::move(j));
        cv_.notify_all();
    }
    // after this every submitted build has state != 0
    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(m_);
            for (auto &j : q_) j->cancel.store(true);
            stop_ = true;
        }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
};

struct CrsEntry {
    const void *ptr = nullptr;
    size_t n = 0;
    int group = 1;
    std::vector<uint64_t> unit_fp;                  // mode 2: one fingerprint per 64-point unit (the last may be short)
    std::vector<uint64_t> head_fp;                  // mode 2: one fingerprint per point of the first unit (prefixes shorter than a unit)
    std::vector<size_t> sample_pos;                 // mode 1
    std::vector<uint8_t> sample;
    lsa_bases *b = nullptr;
    size_t bytes = 0;
    uint64_t tick = 0;
    unsigned hits = 0;
    // the pre-shifted window copies are built by a background thread (TableBuilder): the entry keeps serving the
    // plain layout until the build has finished, then switches; nobody ever waits for the 26 copies
    std::shared_ptr<TableBuild> build;
    void *small = nullptr;                          // the plain bases after the switch (an MSM queued earlier may still read them)
    // G1: pre-shifted copies of the entry's first prefix_n points only, built in one kernel on lsa_stream() when the first
    // request that fits arrives (crs_prefix_table): the tails of the reference's prefix ladders run over them
    void *prefix = nullptr;
    size_t prefix_n = 0;
    unsigned prefix_asks = 0;                       // G2: requests a prefix table would have served (it is built at the second)
    unsigned build_attempts = 0;                    // table builds handed to the builder thread (a failed one is retried once)
    bool building() const { return build && build->state.load() == 0; }
};

struct CrsCache {
    int mode = -1;                                  // -1: read the environment on first use
    size_t budget = 0, bytes = 0;
    uint64_t tick = 0, hits = 0, misses = 0;
    std::vector<CrsEntry> entries;
    HashPool pool;
    std::vector<uint64_t> scratch_fp;
    TableBuilder builder;
    // hits before an entry's copies are built (0: never; LSA_CRS_TABLE_AFTER).  The copies of 2^20 G1 points cost ~27 ms of
    // GPU time and save ~1.2 ms per MSM: building them at the 23rd hit is the break-even rule (never more than twice the
    // cost of the better choice in hindsight); a key that proves a handful of times never pays for them
    unsigned table_after = CRS_TABLE_AFTER_DEFAULT;
    // (Round 6 tried an adaptive rule -- build at the second hit when the caller leaves the GPU idle most of the time, as the
    // unchanged examples do -- and measured a LOSS on `hadamard 20`: 7 more of its 126 G1 MSMs ran over copies, but the two
    // background builds ran beside its foreground kernels: msm_g1 76.8-79.6 -> 82.3-86.3 ms.  The break-even rule stays.)
    std::vector<void *> garbage;                    // device buffers to free once the device is idle
} g_crs;

lsa_host_stats g_host_stats = {};

void crs_configure_from_env() {
    if (g_crs.mode >= 0) return;
    const char *e = getenv("LSA_CRS_CACHE");
    g_crs.mode = 2;
    if (e) {
        if (e[0] == '0' || !strcmp(e, "off")) g_crs.mode = 0;
        else if (!strcmp(e, "sampled") || e[0] == '1') g_crs.mode = 1;
    }
    if (const char *ta = getenv("LSA_CRS_TABLE_AFTER")) g_crs.table_after = (unsigned)atoi(ta);
    const char *mb = getenv("LSA_CRS_CACHE_MB");
    if (mb && atoll(mb) > 0) g_crs.budget = (size_t)atoll(mb) << 20;
    if (g_crs.budget == 0) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) tot = (size_t)64 << 30;
        g_crs.budget = std::min<size_t>(tot / 4, (size_t)64 << 30);
    }
}

size_t bases_device_bytes(const lsa_bases *b) {
    size_t per = msm_base_bytes(b->group);
    return b->n * per * (b->table_stride ? msm_table_windows(b->group, b->n) : 1);
}

// frees what an entry owns beyond its handle; a build in flight is cancelled and waited for (it reads the entry's bases)
void crs_release_entry(CrsEntry &e) {
    if (e.build) {
        e.build->cancel.store(true);
        while (e.build->state.load() == 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (e.build->state.load() == 1) {              // complete but never adopted
            if (e.build->big) (void)hipFree(e.build->big);
            if (e.build->tmp) (void)hipFree(e.build->tmp);
        }
        e.build.reset();
    }
    if (e.small) (void)hipFree(e.small);
    e.small = nullptr;
    if (e.prefix) (void)hipFree(e.prefix);
    e.prefix = nullptr; e.prefix_n = 0;
}
bool crs_entry_building(const lsa_bases *b) {
    for (auto &x : g_crs.entries) if (x.b == b) return x.building();
    return false;
}
void crs_collect_garbage() {
    for (void *p : g_crs.garbage) (void)hipFree(p);
    g_crs.garbage.clear();
}

void crs_evict_to(size_t budget) {
    while (g_crs.bytes > budget && !g_crs.entries.empty()) {
        size_t victim = 0;
        for (size_t i = 1; i < g_crs.entries.size(); i++) if (g_crs.entries[i].tick < g_crs.entries[victim].tick) victim = i;
        (void)msm_join(g.stream);
        (void)hipStreamSynchronize(g.stream);       // no MSM may still read the victim
        g_crs.bytes -= g_crs.entries[victim].bytes;
        crs_release_entry(g_crs.entries[victim]);
        lsa_bases_destroy(g_crs.entries[victim].b);
        g_crs.entries.erase(g_crs.entries.begin() + victim);
    }
}

// evicts least-recently-used entries beyond the budget, never the entry with tick `keep`
void crs_trim(uint64_t keep) {
    while (g_crs.bytes > g_crs.budget && g_crs.entries.size() > 1) {
        size_t victim = g_crs.entries.size();
        for (size_t i = 0; i < g_crs.entries.size(); i++) {
            if (g_crs.entries[i].tick == keep) continue;
            if (victim == g_crs.entries.size() || g_crs.entries[i].tick < g_crs.entries[victim].tick) victim = i;
        }
        if (victim == g_crs.entries.size()) break;
        (void)msm_join(g.stream);
        (void)hipStreamSynchronize(g.stream);       // no MSM may still read the victim
        g_crs.bytes -= g_crs.entries[victim].bytes;
        crs_release_entry(g_crs.entries[victim]);
        lsa_bases_destroy(g_crs.entries[victim].b);
        g_crs.entries.erase(g_crs.entries.begin() + victim);
    }
}

std::vector<size_t> sample_positions(size_t n) {
    std::vector<size_t> pos;
    for (size_t i = 0; i < 8 && i < n; i++) pos.push_back(i);
    for (size_t k = 8; k < n; k <<= 1) {
        pos.push_back(k - 1);
        pos.push_back(k);
        if (k + k / 2 < n) pos.push_back(k + k / 2);
    }
    if (n) pos.push_back(n - 1);
    return pos;
}

}  // namespace

static void crs_cache_clear() {
    crs_evict_to(0);
    crs_collect_garbage();
    g_crs.builder.shutdown();
    g_crs.pool.stop();
    g_crs.bytes = 0;
}

// Background build of an entry's pre-shifted copies.  After `table_after` hits the entry's build is handed to the
// builder thread (allocation, copy 0, the 25 shift + normalise steps -- all off this thread): this MSM and the
// following ones keep using the plain layout.  Once the build has finished, the handle switches to the table (the
// plain buffer is kept until the entry goes: an MSM queued earlier may still read it).
template <class F>
static int crs_table_progress(CrsEntry &e) {
    lsa_bases *b = e.b;
    const char *pe = getenv("LSA_PRECOMPUTE");
    if (b->table_stride || b->n < msm_merge_min() || (pe && pe[0] == '0') || g_crs.table_after == 0) return LSA_OK;
    if (e.build) {
        const int st = e.build->state.load();
        if (st == 0) return LSA_OK;                                    // not yet
        if (st == 1) {
            const size_t before = bases_device_bytes(b);
            e.small = b->d_aff;
            b->d_aff = e.build->big;
            b->table_stride = b->n;
            g_crs.garbage.push_back(e.build->tmp);
            e.bytes += bases_device_bytes(b);                          // the plain copy stays resident too
            g_crs.bytes += bases_device_bytes(b);
            (void)before;
        }
        e.build.reset();                                               // a failed build (no memory) leaves a plain entry
        return LSA_OK;
    }
    const unsigned after = g_crs.table_after;
    if (e.hits < after || e.build_attempts >= 2) return LSA_OK;                          // (no endless retries after a failed build)
    e.build_attempts++;
    const size_t tw = msm_table_windows(b->group, b->n);
    if ((uint64_t)b->n * tw >= (1u << 30)) return LSA_OK;
    auto j = std::make_shared<TableBuild>();
    j->bases = b->d_aff; j->n = b->n; j->group = b->group;
    e.build = j;
    g_crs.builder.submit(j);
    return LSA_OK;
}

template <class F>
static int crs_lookup_or_insert(const void *bases_jac, size_t n, int group, const std::vector<uint64_t> &fp, lsa_bases **out, bool *hit) {
    const size_t psz = sizeof(Jac<F>);
    const size_t nun = (n + CRS_UNIT_POINTS - 1) / CRS_UNIT_POINTS;
    *hit = false;
    for (auto &e : g_crs.entries) {
        if (e.ptr != bases_jac || e.group != group || e.n < n) continue;
        bool same = true;
        if (g_crs.mode == 2) {
            // comparable only when the request's units cover the same byte ranges as the entry's
            if (e.n != n && n % CRS_UNIT_POINTS != 0) continue;
            for (size_t u = 0; u < nun && same; u++) same = fp[u] == e.unit_fp[u];
        } else {
            for (size_t k = 0; k < e.sample_pos.size() && same; k++) {
                size_t i = e.sample_pos[k];
                if (i >= n) continue;
                same = memcmp((const uint8_t *)bases_jac + i * psz, e.sample.data() + k * psz, psz) == 0;
            }
        }
        if (!same) continue;
        e.tick = ++g_crs.tick;
        e.hits++;
        g_crs.hits++;
        if (!e.b->table_stride) {                     // re-used: worth the pre-shifted copies -- built in the background
            int rc = crs_table_progress<F>(e);
            if (rc) return rc;
        }
        lsa_bases *found = e.b;
        const uint64_t keep = e.tick;
        crs_trim(keep);                               // (invalidates `e`)
        *out = found;
        *hit = true;
        return LSA_OK;
    }
    // Not under this pointer -- but maybe under another one: the reference hands multiExpMA COPIES of its key vectors
    // (CommScheme::getBases1() returns g1s by value, /root/reference/src/prototools/commit.h:145-147, and CPPoly::prove
    // calls it once per proof, src/gadgets/poly.h:73), so the same bytes arrive at a new address with every proof.
    // With full fingerprints the cache is content-addressed: an entry of the same group whose unit fingerprints equal
    // the request's is the same vector (or, prefix rule as above, starts with it); it is re-keyed to the new pointer.
    if (g_crs.mode == 2) {
        for (auto &e : g_crs.entries) {
            if (e.group != group || e.n < n || e.ptr == bases_jac) continue;
            if (e.n != n && n % CRS_UNIT_POINTS != 0) continue;
            if (memcmp(fp.data(), e.unit_fp.data(), nun * sizeof(uint64_t)) != 0) continue;
            if (e.n == n) e.ptr = bases_jac;          // a whole-vector match follows the copy; a prefix match leaves the entry where it is
            e.tick = ++g_crs.tick;
            e.hits++;
            g_crs.hits++;
            if (!e.b->table_stride) {
                int rc = crs_table_progress<F>(e);
                if (rc) return rc;
            }
            lsa_bases *found = e.b;
            const uint64_t keep = e.tick;
            crs_trim(keep);
            *out = found;
            *hit = true;
            return LSA_OK;
        }
    }
    // miss: drop stale entries for this pointer, upload + normalise (no table yet), insert
    for (size_t i = 0; i < g_crs.entries.size();) {
        if (g_crs.entries[i].ptr == bases_jac && g_crs.entries[i].group == group && g_crs.entries[i].n <= n) {
            (void)msm_join(g.stream);
            (void)hipStreamSynchronize(g.stream);
            g_crs.bytes -= g_crs.entries[i].bytes;
            crs_release_entry(g_crs.entries[i]);
            lsa_bases_destroy(g_crs.entries[i].b);
            g_crs.entries.erase(g_crs.entries.begin() + i);
        } else i++;
    }
    g_crs.misses++;
    CrsEntry e;
    e.ptr = bases_jac; e.n = n; e.group = group;
    int rc = bases_create<F>(bases_jac, n, 0, group, &e.b, false);
    if (rc) return rc;
    if (g_crs.mode == 2) {
        e.unit_fp.assign(fp.begin(), fp.begin() + nun);
        const size_t head = n < CRS_UNIT_POINTS ? n : CRS_UNIT_POINTS;
        e.head_fp.resize(head);
        for (size_t i = 0; i < head; i++) e.head_fp[i] = hash_words((const uint64_t *)((const uint8_t *)bases_jac + i * psz), psz / 8);
    } else {
        e.sample_pos = sample_positions(n);
        e.sample.resize(e.sample_pos.size() * psz);
        for (size_t k = 0; k < e.sample_pos.size(); k++) memcpy(e.sample.data() + k * psz, (const uint8_t *)bases_jac + e.sample_pos[k] * psz, psz);
    }
    e.bytes = bases_device_bytes(e.b);
    e.tick = ++g_crs.tick;
    g_crs.bytes += e.bytes;
    *out = e.b;
    g_crs.entries.push_back(std::move(e));
    const uint64_t keep = g_crs.entries.back().tick;
    crs_trim(keep);
    for (auto &x : g_crs.entries) if (x.tick == keep) *out = x.b;
    return LSA_OK;
}

// Vectors below CRS_MIN_POINTS are not worth an entry of their own -- but the reference asks for them as PREFIXES of its
// key vector: CPPoly::prove's ladder ends with multiExpMA(g1s, 512 .. 1 scalars) (src/gadgets/poly.h:77-86, globl.h:66).
// Such a request is looked up (never inserted): whole 64-point units against the entries' unit fingerprints, fewer than
// 64 points against the per-point fingerprints of an entry's first unit.  A hit saves the upload and normalisation of
// the points and runs on the entry's pre-shifted copies when it has them (0.3 ms instead of 0.9-1.0 for an MSM this small:
// no fold over 127 doublings).  Every byte the request covers is verified, as for large vectors.
template <class F>
static lsa_bases *crs_lookup_small(const void *bases_jac, size_t n, int group) {
    if (g_crs.mode != 2 || n == 0 || g_crs.entries.empty()) return nullptr;
    const size_t psz = sizeof(Jac<F>);
    if (n >= CRS_UNIT_POINTS && n % CRS_UNIT_POINTS != 0) return nullptr;      // a partial last unit cannot be compared
    const uint8_t *p = (const uint8_t *)bases_jac;
    uint64_t fp[CRS_MIN_POINTS / CRS_UNIT_POINTS > CRS_UNIT_POINTS ? CRS_MIN_POINTS / CRS_UNIT_POINTS : CRS_UNIT_POINTS];
    const bool by_point = n < CRS_UNIT_POINTS;
    const size_t cnt = by_point ? n : n / CRS_UNIT_POINTS;
    for (size_t i = 0; i < cnt; i++)
        fp[i] = by_point ? hash_words((const uint64_t *)(p + i * psz), psz / 8)
                         : hash_words((const uint64_t *)(p + i * CRS_UNIT_POINTS * psz), CRS_UNIT_POINTS * psz / 8);
    for (auto &e : g_crs.entries) {
        if (e.group != group || e.n <= n) continue;
        const std::vector<uint64_t> &have = by_point ? e.head_fp : e.unit_fp;
        if (have.size() < cnt || memcmp(fp, have.data(), cnt * sizeof(uint64_t)) != 0) continue;
        e.tick = ++g_crs.tick;
        g_crs.hits++;
        if (!e.b->table_stride && crs_table_progress<F>(e) != LSA_OK) return nullptr;   // (a finished build switches the entry)
        return e.b;
    }
    return nullptr;
}

// The copies of the first points of a G1 entry that has no full table (yet): CPPoly::prove asks for MSMs over prefixes
// 2^(d-1) .. 1 of its key vector (src/gadgets/poly.h:76-88) and most of those calls are small.  On the plain layout
// such an MSM is 0.9 ms (a fold over 127 dependent doublings, ~25 launches); over copies of the prefix it takes the
// four-launch pipeline of msm_compact.hip.  One kernel, ~1.2 ms, on lsa_stream() ahead of the MSM that asked for it;
// paid back by the second small call.  LSA_CRS_PREFIX_TABLE=0 turns it off.
static size_t crs_prefix_max() {
    static const size_t v = [] {
        const char *e = getenv("LSA_CRS_PREFIX_TABLE");
        const size_t want = e ? (size_t)atoll(e) : (size_t)1 << 16;
        return want > ((size_t)1 << 16) ? (size_t)1 << 16 : want;
    }();
    return v;
}
// G2 entries (CommScheme::commit's second MSM, src/prototools/commit.h:155; InterpCommScheme::commit, src/gadgets/lipmaa.cc:27)
// get theirs at the SECOND request that could use one (LSA_CRS_PREFIX_AFTER_G2 requests are served from the plain layout
// first; the G2 builder's 255 Fq2 doublings per point are ~3x the G1 kernel, and a vector used once never pays for them).
static unsigned crs_prefix_after_g2() {
    static const unsigned v = [] { const char *e = getenv("LSA_CRS_PREFIX_AFTER_G2"); return e ? (unsigned)atoi(e) : 1u; }();
    return v;
}
static int crs_prefix_table(lsa_bases *b, size_t n_req, const void **d_bases, size_t *stride) {
    const char *pe = getenv("LSA_PRECOMPUTE");
    const int grp = b->group;
    if (b->table_stride || n_req == 0 || n_req > crs_prefix_max() || n_req > (grp == 1 ? msm_compact_max() : msm_compact_max_g2()) ||
        (pe && pe[0] == '0') || msm_merge_min_is_explicit())
        return LSA_OK;
    CrsEntry *e = nullptr;
    for (auto &x : g_crs.entries) if (x.b == b) e = &x;
    if (!e) return LSA_OK;
    if (!e->prefix) {
        if (grp == 2 && e->prefix_asks++ < crs_prefix_after_g2()) return LSA_OK;
        const size_t pn = std::min(e->n, crs_prefix_max());
        const size_t bytes = (size_t)msm_table_windows(grp, pn) * pn * msm_base_bytes(grp);
        void *t = nullptr;
        if (hipMalloc(&t, bytes) != hipSuccess) { (void)hipGetLastError(); return LSA_OK; }          // no memory: the plain layout serves
        if (g_stage_prefix_scratch.ensure(table_build_scratch_bytes(pn, grp))) { (void)hipFree(t); return LSA_OK; }
        if (hipMemcpyAsync(t, b->d_aff, pn * msm_base_bytes(grp), hipMemcpyDeviceToDevice, g.stream) != hipSuccess) {
            set_error("crs_prefix_table: copy failed: %s", hipGetErrorString(hipGetLastError()));
            (void)hipFree(t);
            return LSA_ERR_HIP;
        }
        const int rc = grp == 1 ? table_build_g1_device(t, pn, pn, g_stage_prefix_scratch.p, g.stream)
                                : table_build_g2_device(t, pn, pn, g_stage_prefix_scratch.p, g.stream);
        if (rc) { (void)hipStreamSynchronize(g.stream); (void)hipFree(t); return rc; }
        e->prefix = t;
        e->prefix_n = pn;
        e->bytes += bytes;
        g_crs.bytes += bytes;
    }
    if (n_req <= e->prefix_n) { *d_bases = e->prefix; *stride = e->prefix_n; }
    return LSA_OK;
}

static inline double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// this process's MSM into g.d_result (stream-ordered, tails joined); the host-side wall-clock split in `st`
template <class F>
static int msm_host_local(const void *bases_jac, const void *scalars, size_t n, int group, lsa_host_stats &st,
                          std::chrono::steady_clock::time_point &t_msm, bool to_host) {
    int rc = LSA_OK;
    crs_configure_from_env();
    const bool cached = g_crs.mode != 0 && n >= CRS_MIN_POINTS;
    if (g_stage_scalars.ensure(n * sizeof(Fr))) { set_error("msm: staging allocation failed"); return LSA_ERR_NOMEM; }
    const void *d_bases = nullptr;
    size_t table_stride = 0;
    if (cached) {
        // the fingerprint of the bases is computed by the pool while this thread uploads the scalars
        if (g_crs.mode == 2) {
            g_crs.scratch_fp.resize((n + CRS_UNIT_POINTS - 1) / CRS_UNIT_POINTS);
            g_crs.pool.begin(bases_jac, n * sizeof(Jac<F>), CRS_UNIT_POINTS * sizeof(Jac<F>), CRS_TASK_UNITS, g_crs.scratch_fp.data());
        }
        auto t0 = std::chrono::steady_clock::now();
        const int ce = upload_host(g_stage_scalars.p, scalars, n * sizeof(Fr));
        st.h2d_scalars_ms = ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        if (g_crs.mode == 2) g_crs.pool.finish();       // also on the error path: the workers read the caller's buffer
        if (ce) return ce;
        st.fingerprint_wait_ms = ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        lsa_bases *b = nullptr;
        bool hit = false;
        rc = crs_lookup_or_insert<F>(bases_jac, n, group, g_crs.scratch_fp, &b, &hit);
        if (rc) return rc;
        st.bases_prepare_ms = ms_since(t0);
        st.cache_hit = hit ? 1 : 0;
        st.table = b->table_stride ? 1 : 0;
        st.table_building = crs_entry_building(b) ? 1 : 0;
        d_bases = b->d_aff;
        table_stride = b->table_stride;
        rc = crs_prefix_table(b, n, &d_bases, &table_stride);
        if (rc) return rc;
        if (table_stride && !b->table_stride) st.table = 2;           // the entry's prefix table
    } else if (lsa_bases *pb = crs_lookup_small<F>(bases_jac, n, group)) {
        auto t0 = std::chrono::steady_clock::now();
        rc = upload_host(g_stage_scalars.p, scalars, n * sizeof(Fr));
        if (rc) return rc;
        st.h2d_scalars_ms = ms_since(t0);
        st.cache_hit = 1;
        st.table = pb->table_stride ? 1 : 0;
        st.table_building = crs_entry_building(pb) ? 1 : 0;
        d_bases = pb->d_aff;
        table_stride = pb->table_stride;
        rc = crs_prefix_table(pb, n, &d_bases, &table_stride);
        if (rc) return rc;
        if (table_stride && !pb->table_stride) st.table = 2;
    } else {
        if (g_stage_jac.ensure(n * sizeof(Jac<F>)) || g_stage_bases.ensure(n * msm_base_bytes(group))) {
            set_error("msm: staging allocation failed");
            return LSA_ERR_NOMEM;
        }
        if (n) {
            auto t0 = std::chrono::steady_clock::now();
            rc = upload_host(g_stage_jac.p, bases_jac, n * sizeof(Jac<F>));
            if (rc) return rc;
            st.bases_prepare_ms = ms_since(t0);
            t0 = std::chrono::steady_clock::now();
            rc = upload_host(g_stage_scalars.p, scalars, n * sizeof(Fr));
            if (rc) return rc;
            st.h2d_scalars_ms = ms_since(t0);
            rc = prepare_bases<F>((const Jac<F> *)g_stage_jac.p, g_stage_bases.p, n, g.stream);
            if (rc) return rc;
        }
        d_bases = g_stage_bases.p;
    }
    t_msm = std::chrono::steady_clock::now();
    rc = msm_join(g.stream);
    if (rc) return rc;
    // blocking call: tail on lsa_stream(); unless the partials still have to be exchanged, the result lands in pinned host memory
    return msm_device<F>(d_bases, 0, (const Fr *)g_stage_scalars.p, n, (Jac<F> *)(to_host ? g.h_result : g.d_result), g.stream, table_stride, true);
}

template <class F>
static int msm_host(const void *bases_jac, const void *scalars, size_t n, void *out_jac, int group, bool sharded = false) {
    LSA_TRACE_CALL(group == 1 ? "msm_g1" : "msm_g2", n);
    int rc = require_ready();
    if (rc) return rc;
    if (sharded && lsa_comm_world() <= 1) sharded = false;
    if (!out_jac || (n && (!bases_jac || !scalars))) { set_error("msm: null argument"); return LSA_ERR_INVALID; }
    const auto t_all = std::chrono::steady_clock::now();
    lsa_host_stats st = {};
    st.n = n;
    auto t0 = std::chrono::steady_clock::now();
    rc = msm_host_local<F>(bases_jac, scalars, n, group, st, t0, !sharded);
    if (sharded) {
        // this rank's slice is one libff chunk: all-gather the partials, sum them in rank order.  A rank whose
        // local part failed still takes part (with the point at infinity: all-zero bytes) and reports its error
        // afterwards -- the other ranks must not be left blocking in ncclAllGather.
        const int local = rc;
        const size_t world = (size_t)lsa_comm_world();
        if (local) (void)hipMemsetAsync(g.d_result, 0, sizeof(Jac<F>), g.stream);
        if (g_stage_gather.ensure(world * sizeof(Jac<F>))) { set_error("msm: staging allocation failed"); return LSA_ERR_NOMEM; }
        rc = lsa_comm_all_gather(g.d_result, g_stage_gather.p, group);
        if (local) return local;
        if (rc) return rc;
        rc = sum_points_device<F>((const Jac<F> *)g_stage_gather.p, world, (Jac<F> *)g.d_result, g.stream);
    }
    if (rc) return rc;
    if (sharded) HIPCHK(hipMemcpyAsync(g.h_result, g.d_result, sizeof(Jac<F>), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    memcpy(out_jac, g.h_result, sizeof(Jac<F>));
    if (!g_crs.garbage.empty()) crs_collect_garbage();     // lsa_stream() has just drained
    st.msm_ms = ms_since(t0);
    st.total_ms = ms_since(t_all);
    g_host_stats = st;
    if (trace_on())
        fprintf(stderr, "[lsa]   msm_split                  h2d_scalars=%.3f fingerprint_wait=%.3f bases_prepare=%.3f kernels=%.3f hit=%d table=%d\n",
                st.h2d_scalars_ms, st.fingerprint_wait_ms, st.bases_prepare_ms, st.msm_ms, st.cache_hit, st.table);
    return LSA_OK;
}

extern "C" {
int lsa_g1_msm(const void *bases_jac, const void *scalars, size_t n, size_t chunks, void *out_jac) {
    (void)chunks;
    return msm_host<Fq>(bases_jac, scalars, n, out_jac, 1);
}
int lsa_g2_msm(const void *bases_jac, const void *scalars, size_t n, size_t chunks, void *out_jac) {
    (void)chunks;
    return msm_host<Fq2>(bases_jac, scalars, n, out_jac, 2);
}
int lsa_g1_msm_sharded(const void *bases_jac, const void *scalars, size_t n_local, void *out_jac) {
    return msm_host<Fq>(bases_jac, scalars, n_local, out_jac, 1, true);
}
int lsa_g2_msm_sharded(const void *bases_jac, const void *scalars, size_t n_local, void *out_jac) {
    return msm_host<Fq2>(bases_jac, scalars, n_local, out_jac, 2, true);
}
int lsa_crs_cache_configure(int mode, size_t max_bytes) {
    int rc = require_ready();
    if (rc) return rc;
    if (mode < 0 || mode > 2) { set_error("crs_cache_configure: mode must be 0 (off), 1 (sampled) or 2 (full)"); return LSA_ERR_INVALID; }
    crs_configure_from_env();
    if (mode != g_crs.mode) crs_evict_to(0);          // fingerprints of the two modes are not comparable
    g_crs.mode = mode;
    if (max_bytes) g_crs.budget = max_bytes;
    crs_evict_to(g_crs.budget);
    return LSA_OK;
}
void lsa_crs_cache_clear(void) { if (g.ready) crs_evict_to(0); }
int lsa_crs_cache_stats(uint64_t *hits, uint64_t *misses, uint64_t *resident_bytes, uint64_t *entries) {
    if (hits) *hits = g_crs.hits;
    if (misses) *misses = g_crs.misses;
    if (resident_bytes) *resident_bytes = g_crs.bytes;
    if (entries) *entries = g_crs.entries.size();
    return LSA_OK;
}
int lsa_crs_cache_table_after(unsigned hits) {
    crs_configure_from_env();
    g_crs.table_after = hits;
    return LSA_OK;
}
// blocks until every background table build has finished and its entry has switched (tests, benchmarks)
int lsa_crs_cache_wait_tables(void) {
    int rc = require_ready();
    if (rc) return rc;
    for (auto &e : g_crs.entries) {
        if (!e.build) continue;
        while (e.build->state.load() == 0) std::this_thread::sleep_for(std::chrono::microseconds(100));
        rc = e.group == 1 ? crs_table_progress<Fq>(e) : crs_table_progress<Fq2>(e);
        if (rc) return rc;
    }
    return LSA_OK;
}
int lsa_msm_host_stats(lsa_host_stats *out) {
    if (!out) return LSA_ERR_INVALID;
    *out = g_host_stats;
    return LSA_OK;
}

// ---------------------------------------------------------------- normalisation
}  // extern "C"
template <class F>
static int normalize_host(const void *in_jac, size_t n, void *out_jac) {
    LSA_TRACE_CALL("normalize", n);
    int rc = require_ready();
    if (rc) return rc;
    if (n == 0) return LSA_OK;
    if (!in_jac || !out_jac) { set_error("normalize: null argument"); return LSA_ERR_INVALID; }
    void *d_in = nullptr, *d_aff = nullptr;
    if (hipMalloc(&d_in, n * sizeof(Jac<F>)) != hipSuccess || hipMalloc(&d_aff, n * sizeof(Aff<F>)) != hipSuccess) {
        if (d_in) (void)hipFree(d_in);
        set_error("normalize: hipMalloc failed");
        return LSA_ERR_NOMEM;
    }
    rc = upload_host(d_in, in_jac, n * sizeof(Jac<F>));
    if (!rc) rc = normalize_to_affine<F>((const Jac<F> *)d_in, (Aff<F> *)d_aff, n, g.stream);
    std::vector<Aff<F>> host(n);
    if (!rc) rc = download_host(host.data(), d_aff, n * sizeof(Aff<F>));
    const hipError_t e = hipStreamSynchronize(g.stream);
    (void)hipFree(d_in);
    (void)hipFree(d_aff);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("normalize: %s", hipGetErrorString(e)); return LSA_ERR_HIP; }
    Jac<F> *o = (Jac<F> *)out_jac;
    for (size_t i = 0; i < n; i++) {
        if (host[i].is_inf()) o[i] = Jac<F>::inf();
        else o[i] = Jac<F>{host[i].x, host[i].y, F::one()};
    }
    return LSA_OK;
}
extern "C" {
int lsa_g1_normalize(const void *in_jac, size_t n, void *out_jac) { return normalize_host<Fq>(in_jac, n, out_jac); }
int lsa_g2_normalize(const void *in_jac, size_t n, void *out_jac) { return normalize_host<Fq2>(in_jac, n, out_jac); }

}  // extern "C"

// ---------------------------------------------------------------- batch_exp / sum
template <class F>
static int batch_exp_any(const void *base_jac, const void *scalars, size_t n, void *out_jac, int on_device) {
    LSA_TRACE_CALL("batch_exp", n);
    int rc = require_ready();
    if (rc) return rc;
    if (n == 0) return LSA_OK;
    if (!base_jac || !scalars || !out_jac) { set_error("batch_exp: null argument"); return LSA_ERR_INVALID; }
    Jac<F> base;
    memcpy(&base, base_jac, sizeof base);
    if (on_device) return batch_exp_device<F>(base, (const Fr *)scalars, n, (Jac<F> *)out_jac, g.stream);
    // grow-only staging (the shim calls this with anything from one scalar to a whole key vector)
    if (g_stage_scalars.ensure(n * sizeof(Fr)) || g_stage_jac.ensure(n * sizeof(Jac<F>))) { set_error("batch_exp: staging allocation failed"); return LSA_ERR_NOMEM; }
    void *d_sc = g_stage_scalars.p, *d_out = g_stage_jac.p;
    rc = upload_host(d_sc, scalars, n * sizeof(Fr));
    if (!rc) rc = batch_exp_device<F>(base, (const Fr *)d_sc, n, (Jac<F> *)d_out, g.stream);
    if (!rc) rc = download_host(out_jac, d_out, n * sizeof(Jac<F>));
    const hipError_t e = hipStreamSynchronize(g.stream);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("batch_exp: %s", hipGetErrorString(e)); return LSA_ERR_HIP; }
    return LSA_OK;
}

extern "C" {
int lsa_g1_batch_exp(const void *base_jac, const void *scalars, size_t n, void *out_jac, int on_device) {
    return batch_exp_any<Fq>(base_jac, scalars, n, out_jac, on_device);
}
int lsa_g2_batch_exp(const void *base_jac, const void *scalars, size_t n, void *out_jac, int on_device) {
    return batch_exp_any<Fq2>(base_jac, scalars, n, out_jac, on_device);
}
// the same on a caller-supplied stream, without touching lsa_stream(): lets the collective and
// the fold of step i run beside the MSM front of step i+1
}  // extern "C"
template <class F>
static int sum_on_any(const void *d_pts, size_t n, void *d_out, void *stream) {
    int rc = require_ready();
    if (rc) return rc;
    if (!d_out || (n && !d_pts)) { set_error("sum: null argument"); return LSA_ERR_INVALID; }
    return sum_points_device<F>((const Jac<F> *)d_pts, n, (Jac<F> *)d_out, (hipStream_t)stream);
}
extern "C" {
int lsa_g1_sum_on(const void *d_pts, size_t n, void *d_out, void *stream) { return sum_on_any<Fq>(d_pts, n, d_out, stream); }
int lsa_g2_sum_on(const void *d_pts, size_t n, void *d_out, void *stream) { return sum_on_any<Fq2>(d_pts, n, d_out, stream); }
int lsa_stream_join_to(void *stream) {
    int rc = require_ready();
    if (rc) return rc;
    return msm_join_to((hipStream_t)stream);
}
int lsa_g1_sum_async(const void *d_pts, size_t n, void *d_out) {
    int rc = require_ready();
    if (rc) return rc;
    if (!d_out || (n && !d_pts)) { set_error("sum: null argument"); return LSA_ERR_INVALID; }
    rc = msm_join(g.stream);
    if (rc) return rc;
    return sum_points_device<Fq>((const Jac<Fq> *)d_pts, n, (Jac<Fq> *)d_out, g.stream);
}
int lsa_g2_sum_async(const void *d_pts, size_t n, void *d_out) {
    int rc = require_ready();
    if (rc) return rc;
    if (!d_out || (n && !d_pts)) { set_error("sum: null argument"); return LSA_ERR_INVALID; }
    rc = msm_join(g.stream);
    if (rc) return rc;
    return sum_points_device<Fq2>((const Jac<Fq2> *)d_pts, n, (Jac<Fq2> *)d_out, g.stream);
}
}  // extern "C"

namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess ? 0 : -1; }
};
}  // namespace

// ---------------------------------------------------------------- variable-base scalar mul / sparse matrix
extern "C" {
int lsa_g1_scalar_mul_batch(const void *pts_jac, const void *scalars, size_t n, void *out_jac, int on_device) {
    LSA_TRACE_CALL("g1_scalar_mul_batch", n);
    int rc = require_ready();
    if (rc) return rc;
    if (n == 0) return LSA_OK;
    if (!pts_jac || !scalars || !out_jac) { set_error("scalar_mul_batch: null argument"); return LSA_ERR_INVALID; }
    rc = msm_join(g.stream);
    if (rc) return rc;
    if (on_device) return g1_scalar_mul_device((const Jac<Fq> *)pts_jac, (const Fr *)scalars, nullptr, n, (Jac<Fq> *)out_jac, g.stream);
    DevBuf d_p, d_s, d_o;
    if (d_p.alloc(n * sizeof(Jac<Fq>)) || d_s.alloc(n * sizeof(Fr)) || d_o.alloc(n * sizeof(Jac<Fq>))) {
        set_error("scalar_mul_batch: hipMalloc failed");
        return LSA_ERR_NOMEM;
    }
    LSA_UPLOAD(d_p.p, pts_jac, n * sizeof(Jac<Fq>));
    LSA_UPLOAD(d_s.p, scalars, n * sizeof(Fr));
    rc = g1_scalar_mul_device((const Jac<Fq> *)d_p.p, (const Fr *)d_s.p, nullptr, n, (Jac<Fq> *)d_o.p, g.stream);
    if (rc) return rc;
    LSA_DOWNLOAD(out_jac, d_o.p, n * sizeof(Jac<Fq>));
    return LSA_OK;
}

int lsa_g1_sparse_matrix_msm(const void *vals_jac, const uint32_t *rows, const uint64_t *col_ptr, size_t ncols,
                             const void *exps, size_t nrows, void *out_jac) {
    int rc = require_ready();
    if (rc) return rc;
    if (ncols == 0) return LSA_OK;
    if (!col_ptr || !out_jac) { set_error("sparse_matrix_msm: null argument"); return LSA_ERR_INVALID; }
    const uint64_t nnz = col_ptr[ncols];
    if (col_ptr[0] != 0) { set_error("sparse_matrix_msm: col_ptr[0] must be 0"); return LSA_ERR_INVALID; }
    for (size_t j = 0; j < ncols; j++)
        if (col_ptr[j] > col_ptr[j + 1]) { set_error("sparse_matrix_msm: col_ptr not monotone at column %zu", j); return LSA_ERR_INVALID; }
    if (nnz && (!vals_jac || !rows || !exps)) { set_error("sparse_matrix_msm: null argument"); return LSA_ERR_INVALID; }
    for (uint64_t e = 0; e < nnz; e++)
        if (rows[e] >= nrows) { set_error("sparse_matrix_msm: row index %u out of range (%zu rows)", rows[e], nrows); return LSA_ERR_INVALID; }
    rc = msm_join(g.stream);
    if (rc) return rc;
    DevBuf d_v, d_r, d_c, d_e, d_items, d_o;
    if (d_v.alloc(nnz * sizeof(Jac<Fq>)) || d_r.alloc(nnz * sizeof(uint32_t)) || d_c.alloc((ncols + 1) * sizeof(uint64_t)) ||
        d_e.alloc(nrows * sizeof(Fr)) || d_items.alloc(nnz * sizeof(Jac<Fq>)) || d_o.alloc(ncols * sizeof(Jac<Fq>))) {
        set_error("sparse_matrix_msm: hipMalloc failed");
        return LSA_ERR_NOMEM;
    }
    if (nnz) {
        LSA_UPLOAD(d_v.p, vals_jac, nnz * sizeof(Jac<Fq>));
        LSA_UPLOAD(d_r.p, rows, nnz * sizeof(uint32_t));
        LSA_UPLOAD(d_e.p, exps, nrows * sizeof(Fr));
    }
    LSA_UPLOAD(d_c.p, col_ptr, (ncols + 1) * sizeof(uint64_t));
    rc = g1_scalar_mul_device((const Jac<Fq> *)d_v.p, (const Fr *)d_e.p, (const uint32_t *)d_r.p, nnz, (Jac<Fq> *)d_items.p, g.stream);
    if (rc) return rc;
    rc = g1_column_sums_device((const Jac<Fq> *)d_items.p, (const uint64_t *)d_c.p, ncols, (Jac<Fq> *)d_o.p, g.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    LSA_DOWNLOAD(out_jac, d_o.p, ncols * sizeof(Jac<Fq>));
    return LSA_OK;
}
}  // extern "C"

// ---------------------------------------------------------------- Fr vectors
extern "C" {
int lsa_fr_cppoly_witness(const void *v, size_t d, const void *r, void *w, int on_device) {
    LSA_TRACE_CALL("fr_cppoly_witness", (size_t)1 << d);
    int rc = require_ready();
    if (rc) return rc;
    if (d > 40) { set_error("cppoly_witness: d = %zu too large", d); return LSA_ERR_INVALID; }
    if (!v || !w || (d && !r)) { set_error("cppoly_witness: null argument"); return LSA_ERR_INVALID; }
    const size_t N = (size_t)1 << d;
    StageBuf &d_tmp = g_stage_fr_tmp, &d_v = g_stage_fr_v, &d_r = g_stage_fr_r, &d_w = g_stage_fr_w;
    if (d_tmp.ensure((N / 2 + N / 4 + 1) * sizeof(Fr))) { set_error("cppoly_witness: hipMalloc failed"); return LSA_ERR_NOMEM; }
    if (on_device) {
        rc = fr_cppoly_fold_device((const Fr *)v, d, (const Fr *)r, (Fr *)w, (Fr *)d_tmp.p, g.stream);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(g.stream));   // (callers read w from their own streams)
        return LSA_OK;
    }
    if (d_v.ensure(N * sizeof(Fr)) || d_r.ensure((d + 1) * sizeof(Fr)) || d_w.ensure(N * sizeof(Fr))) { set_error("cppoly_witness: hipMalloc failed"); return LSA_ERR_NOMEM; }
    LSA_UPLOAD(d_v.p, v, N * sizeof(Fr));
    if (d) LSA_UPLOAD(d_r.p, r, d * sizeof(Fr));
    rc = fr_cppoly_fold_device((const Fr *)d_v.p, d, (const Fr *)d_r.p, (Fr *)d_w.p, (Fr *)d_tmp.p, g.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    LSA_DOWNLOAD(w, d_w.p, N * sizeof(Fr));
    return LSA_OK;
}

int lsa_fr_eval_mle(const void *v, size_t d, const void *r, void *out, int on_device) {
    LSA_TRACE_CALL("fr_eval_mle", (size_t)1 << d);
    int rc = require_ready();
    if (rc) return rc;
    if (d > 40) { set_error("eval_mle: d = %zu too large", d); return LSA_ERR_INVALID; }
    if (!v || !out || (d && !r)) { set_error("eval_mle: null argument"); return LSA_ERR_INVALID; }
    const size_t N = (size_t)1 << d;
    StageBuf &d_tmp = g_stage_fr_tmp, &d_v = g_stage_fr_v, &d_r = g_stage_fr_r, &d_o = g_stage_fr_w;
    if (d_tmp.ensure((N / 2 + N / 4 + 1) * sizeof(Fr))) { set_error("eval_mle: hipMalloc failed"); return LSA_ERR_NOMEM; }
    if (on_device) {
        rc = fr_eval_mle_device((const Fr *)v, d, (const Fr *)r, (Fr *)d_tmp.p, (Fr *)out, g.stream);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(g.stream));
        return LSA_OK;
    }
    if (d_v.ensure(N * sizeof(Fr)) || d_r.ensure((d + 1) * sizeof(Fr)) || d_o.ensure(sizeof(Fr))) { set_error("eval_mle: hipMalloc failed"); return LSA_ERR_NOMEM; }
    LSA_UPLOAD(d_v.p, v, N * sizeof(Fr));
    if (d) LSA_UPLOAD(d_r.p, r, d * sizeof(Fr));
    rc = fr_eval_mle_device((const Fr *)d_v.p, d, (const Fr *)d_r.p, (Fr *)d_tmp.p, (Fr *)d_o.p, g.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    LSA_DOWNLOAD(out, d_o.p, sizeof(Fr));
    return LSA_OK;
}

int lsa_fr_sumcheck_round(const void *suff, const void *const *tables, size_t m, size_t half, const void *pre, const void *rho_j,
                          void *out_coeffs, int on_device) {
    int rc = require_ready();
    if (rc) return rc;
    if (!tables || !out_coeffs || m == 0 || m > 4 || half == 0) { set_error("sumcheck_round: invalid argument (1 <= m <= 4, half > 0)"); return LSA_ERR_INVALID; }
    if (rho_j && !pre) { set_error("sumcheck_round: rho_j without pre"); return LSA_ERR_INVALID; }
    for (size_t t = 0; t < m; t++) if (!tables[t]) { set_error("sumcheck_round: null table"); return LSA_ERR_INVALID; }
    const size_t ncoef = m + (rho_j ? 2 : 1);
    StageBuf &d_partial = g_stage_sc_partial, &d_out = g_stage_sc_out;
    DevBuf d_suff, d_tab[4];
    if (d_partial.ensure(fr_sumcheck_scratch_elems() * sizeof(Fr)) || d_out.ensure(8 * sizeof(Fr))) { set_error("sumcheck_round: hipMalloc failed"); return LSA_ERR_NOMEM; }
    const Fr *tabs[4] = {nullptr, nullptr, nullptr, nullptr};
    const Fr *sf = (const Fr *)suff;
    if (on_device) {
        for (size_t t = 0; t < m; t++) tabs[t] = (const Fr *)tables[t];
    } else {
        for (size_t t = 0; t < m; t++) {
            if (d_tab[t].alloc(2 * half * sizeof(Fr))) { set_error("sumcheck_round: hipMalloc failed"); return LSA_ERR_NOMEM; }
            LSA_UPLOAD(d_tab[t].p, tables[t], 2 * half * sizeof(Fr));
            tabs[t] = (const Fr *)d_tab[t].p;
        }
        if (suff) {
            if (d_suff.alloc(half * sizeof(Fr))) { set_error("sumcheck_round: hipMalloc failed"); return LSA_ERR_NOMEM; }
            LSA_UPLOAD(d_suff.p, suff, half * sizeof(Fr));
            sf = (const Fr *)d_suff.p;
        }
    }
    rc = fr_sumcheck_round_device(sf, tabs, m, half, (const Fr *)pre, (const Fr *)rho_j, (Fr *)d_partial.p, (Fr *)d_out.p, g.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    LSA_DOWNLOAD(out_coeffs, d_out.p, ncoef * sizeof(Fr));
    return LSA_OK;
}

int lsa_fr_scale_upper(const void *old, size_t half, const void *k, void *cur, int on_device) {
    LSA_TRACE_CALL("fr_scale_upper", half);
    int rc = require_ready();
    if (rc) return rc;
    if (half == 0) return LSA_OK;
    if (!old || !cur || !k) { set_error("fr_scale_upper: null argument"); return LSA_ERR_INVALID; }
    Fr kk;
    memcpy(&kk, k, sizeof kk);
    if (on_device) return fr_scale_upper_device((const Fr *)old, half, kk, (Fr *)cur, g.stream);
    DevBuf d_v;
    if (d_v.alloc(2 * half * sizeof(Fr))) { set_error("fr_scale_upper: hipMalloc failed"); return LSA_ERR_NOMEM; }
    LSA_UPLOAD(d_v.p, old, 2 * half * sizeof(Fr));
    rc = fr_scale_upper_device((const Fr *)d_v.p, half, kk, (Fr *)d_v.p, g.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    LSA_DOWNLOAD(cur, d_v.p, half * sizeof(Fr));
    return LSA_OK;
}

int lsa_fr_eq_table(const void *r, size_t d, int variant, void *out, int on_device) {
    LSA_TRACE_CALL("fr_eq_table", (size_t)1 << (d & 63));
    int rc = require_ready();
    if (rc) return rc;
    if (d == 0 || d > 30) { set_error("fr_eq_table: need 1 <= d <= 30 (got %zu)", d); return LSA_ERR_INVALID; }
    if (!out || (d && !r)) { set_error("fr_eq_table: null argument"); return LSA_ERR_INVALID; }
    if (variant != 0 && variant != 1) { set_error("fr_eq_table: variant %d (0: the reference's loop, 1: the eq monomials)", variant); return LSA_ERR_INVALID; }
    const size_t N = (size_t)1 << d;
    StageBuf &d_tmp = g_stage_fr_tmp, &d_r = g_stage_fr_r, &d_o = g_stage_fr_w;
    if (d_tmp.ensure(fr_eq_table_scratch_elems(d) * sizeof(Fr))) { set_error("fr_eq_table: hipMalloc failed"); return LSA_ERR_NOMEM; }
    if (on_device) return fr_eq_table_device((const Fr *)r, d, variant, (Fr *)d_tmp.p, (Fr *)out, g.stream);
    if (d_r.ensure((d + 1) * sizeof(Fr)) || d_o.ensure(N * sizeof(Fr))) { set_error("fr_eq_table: hipMalloc failed"); return LSA_ERR_NOMEM; }
    if (d) LSA_UPLOAD(d_r.p, r, d * sizeof(Fr));
    rc = fr_eq_table_device((const Fr *)d_r.p, d, variant, (Fr *)d_tmp.p, (Fr *)d_o.p, g.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    LSA_DOWNLOAD(out, d_o.p, N * sizeof(Fr));
    return LSA_OK;
}

int lsa_fr_ntt(void *a, size_t log_n, const void *omega, int inverse, const void *coset_g, int on_device) {
    LSA_TRACE_CALL("fr_ntt", (size_t)1 << log_n);
    int rc = require_ready();
    if (rc) return rc;
    if (log_n > 28) { set_error("fr_ntt: log_n = %zu exceeds the 2-adicity of Fr (28)", log_n); return LSA_ERR_INVALID; }
    if (!a || !omega) { set_error("fr_ntt: null argument"); return LSA_ERR_INVALID; }
    if (log_n == 0) return LSA_OK;
    const size_t n = (size_t)1 << log_n;
    Fr w, gco;
    memcpy(&w, omega, sizeof w);
    if (coset_g) memcpy(&gco, coset_g, sizeof gco);
    // grow-only staging (released by lsa_shutdown, like every other staging buffer) for the second buffer of the passes
    // and, for host callers, the data: two hipMalloc / hipFree pairs per call cost a 2^20-point transform of an unchanged
    // prover more than its kernels (the shim's evaluation_domain calls this 7 times per Lipmaa proof).  The twiddle
    // tables are cached per domain inside ntt.hip.
    StageBuf &s_a = g_stage_ntt_a;
    if (log_n > NTT_TILE_LOG && g_stage_ntt_tmp.ensure(n * sizeof(Fr))) { set_error("fr_ntt: hipMalloc failed"); return LSA_ERR_NOMEM; }
    Fr *da = (Fr *)a;
    if (!on_device) {
        if (s_a.ensure(n * sizeof(Fr))) { set_error("fr_ntt: hipMalloc failed"); return LSA_ERR_NOMEM; }
        LSA_UPLOAD(s_a.p, a, n * sizeof(Fr));
        da = (Fr *)s_a.p;
    }
    rc = fr_ntt_device(da, (unsigned)log_n, w, inverse != 0, coset_g ? &gco : nullptr, (Fr *)g_stage_ntt_tmp.p, g.stream);
    if (rc) return rc;
    // (The download is enqueued BEHIND the kernels, not after a wait for them: the first device -> host copy a process
    // issues on an idle stream costs it 8 ms of copy-engine set-up on this stack.)
    if (!on_device) LSA_DOWNLOAD(a, s_a.p, n * sizeof(Fr));
    else HIPCHK(hipStreamSynchronize(g.stream));
    return LSA_OK;
}

int lsa_fr_ntt_step(void *a, size_t big_log, size_t small_log, const void *omega, int inverse, const void *coset_g, int on_device) {
    LSA_TRACE_CALL("fr_ntt_step", ((size_t)1 << (big_log & 31)) + ((size_t)1 << (small_log & 31)));
    int rc = require_ready();
    if (rc) return rc;
    if (big_log > 27 || small_log >= big_log) {
        set_error("fr_ntt_step: need small_log < big_log <= 27 (got %zu, %zu): omega is a 2^(big_log + 1)-th root of unity of a field of 2-adicity 28", small_log, big_log);
        return LSA_ERR_INVALID;
    }
    if (!a || !omega) { set_error("fr_ntt_step: null argument"); return LSA_ERR_INVALID; }
    const size_t big = (size_t)1 << big_log, m = big + ((size_t)1 << small_log);
    Fr w, gco;
    memcpy(&w, omega, sizeof w);
    if (coset_g) memcpy(&gco, coset_g, sizeof gco);
    if (g_stage_ntt_tmp.ensure(big * sizeof(Fr))) { set_error("fr_ntt_step: hipMalloc failed"); return LSA_ERR_NOMEM; }
    Fr *da = (Fr *)a;
    if (!on_device) {
        if (g_stage_ntt_a.ensure(m * sizeof(Fr))) { set_error("fr_ntt_step: hipMalloc failed"); return LSA_ERR_NOMEM; }
        LSA_UPLOAD(g_stage_ntt_a.p, a, m * sizeof(Fr));
        da = (Fr *)g_stage_ntt_a.p;
    }
    rc = fr_ntt_step_device(da, (unsigned)big_log, (unsigned)small_log, w, inverse != 0, coset_g ? &gco : nullptr, (Fr *)g_stage_ntt_tmp.p, g.stream);
    if (rc) return rc;
    if (!on_device) LSA_DOWNLOAD(a, g_stage_ntt_a.p, m * sizeof(Fr));
    else HIPCHK(hipStreamSynchronize(g.stream));
    return LSA_OK;
}

int lsa_fr_fold(const void *old, size_t half, const void *r, void *cur, int on_device) {
    LSA_TRACE_CALL("fr_fold", half);
    int rc = require_ready();
    if (rc) return rc;
    if (half == 0) return LSA_OK;
    if (!old || !cur || !r) { set_error("fr_fold: null argument"); return LSA_ERR_INVALID; }
    if (on_device) return fr_fold_halves_device((const Fr *)old, half, (const Fr *)r, (Fr *)cur, g.stream);
    DevBuf d_v, d_r;
    if (d_v.alloc(2 * half * sizeof(Fr)) || d_r.alloc(sizeof(Fr))) { set_error("fr_fold: hipMalloc failed"); return LSA_ERR_NOMEM; }
    LSA_UPLOAD(d_v.p, old, 2 * half * sizeof(Fr));
    LSA_UPLOAD(d_r.p, r, sizeof(Fr));
    rc = fr_fold_halves_device((const Fr *)d_v.p, half, (const Fr *)d_r.p, (Fr *)d_v.p, g.stream);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g.stream));
    LSA_DOWNLOAD(cur, d_v.p, half * sizeof(Fr));
    return LSA_OK;
}
}  // extern "C"

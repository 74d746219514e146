// legosnark_amd/csrc/capi_internal.h -- state shared by the translation units that implement the
// C-ABI (capi.hip: single-GPU entry points; comm.hip: the multi-GPU exchange step).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "msm.h"

namespace lsa {

struct State {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    void *d_result = nullptr;      // 512-byte device slot for MSM / pairing results
    void *h_result = nullptr;      // pinned host mirror
};
extern State g;

#define HIPCHK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return LSA_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

int require_ready();

// Keyed content fingerprints (NH under a per-process random key, capi.hip): for two distinct inputs of the same
// length the collision probability over the key is <= 2^-64 (128-bit form), whatever the inputs are.
void keyed_hash128(const void *bytes, size_t nbytes, uint64_t tweak, uint64_t out[2]);
uint64_t keyed_hash64(const void *bytes, size_t nbytes, uint64_t tweak);

// LSA_TRACE=1: one line on stderr per host-facing call -- name, size, wall time (what a caller's profile of the
// library looks like from outside; off: one predictable branch).
bool trace_on();
struct CallTrace {
    const char *name;
    size_t n;
    double t0;
    static double now_ms();
    CallTrace(const char *nm, size_t items) : name(nm), n(items), t0(trace_on() ? now_ms() : 0) {}
    ~CallTrace() { if (trace_on()) fprintf(stderr, "[lsa] %-28s n=%-9zu %9.3f ms\n", name, n, now_ms() - t0); }
};
#define LSA_TRACE_CALL(name, items) CallTrace trace_scope_((name), (size_t)(items))
void comm_release();               // comm.hip: called by lsa_shutdown
void pairing_release();            // capi_pairing.hip: staging buffers and the G2 line-table cache

// grow-only device staging buffer of the host-buffer entry points (the libff shim calls them
// thousands of times with tiny inputs: no hipMalloc / hipFree per call)
struct StageBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        static const bool noisy = getenv("LSA_TRACE") && getenv("LSA_TRACE")[0] == '2';
        if (noisy) fprintf(stderr, "[lsa]   stage_grow                  %zu -> %zu bytes\n", cap, bytes < 4096 ? (size_t)4096 : bytes + bytes / 4);
        if (p) { (void)hipStreamSynchronize(g.stream); (void)hipFree(p); }
        p = nullptr; cap = 0;
        size_t want = bytes < 4096 ? 4096 : bytes + bytes / 4;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return -1; }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
extern StageBuf g_stage_gather;

// Pageable host memory <-> device on lsa_stream().  For a large pageable copy hipMemcpy pins the caller's pages in
// place and keeps the registration; when the caller later frees that vector with munmap (glibc: every allocation above
// its mmap threshold, at most 32 MiB -- a std::vector<Fr> of 2^20 scalars is just above) the driver evicts the process's
// queues and the next submission waits 12-25 ms (measured: tools/native/h2d_vectors.cc, a CPPoly::prove-shaped ladder of
// MSMs 29 -> 15 ms without it), and pinning 4-KiB pages is itself slow (32 MiB: 1.0-1.5 ms against 0.6 at the link rate).
// Copies of 32 KiB .. 16 MiB therefore go through pinned 2-MiB slots owned by the library (a few threads share the
// memcpy work from two slots up); larger ones are left to the runtime, whose single DMA from the caller's pages is
// 0.3-0.4 ms ahead of sixteen staged commands at 32 MiB -- the shim keeps such vectors on the heap (never unmapped), other
// callers keep them alive across calls or set LSA_H2D=staged.  upload_host: every copy is ON THE STREAM when it returns
// (then asynchronous, like hipMemcpyAsync); download_host: blocking, the bytes are in h_dst when it returns.
// LSA_H2D=direct|staged: plain hipMemcpyAsync always / slots at every size.
int upload_host(void *d_dst, const void *h_src, size_t bytes);
// the two halves of upload_host for callers that cut one host range into several copies: the decision (recorded once
// per range: a large range goes through the slots until it has been seen LSA_H2D_DIRECT_AFTER times) and the copy
bool upload_takes_slots(const void *h_src, size_t bytes);
// (order_after = false: the destination is not touched by anything queued on lsa_stream() since the previous piece of
// the same range went out -- the copy streams then do not wait for the kernels queued in between)
int upload_host_as(void *d_dst, const void *h_src, size_t bytes, bool slots, bool order_after);
int download_host(void *h_dst, const void *d_src, size_t bytes);
void upload_release();             // threads, pinned slots (lsa_shutdown)
void upload_prepare();             // the same, created (lsa_init)
#define LSA_UPLOAD(dst, src, bytes) do { int u_ = upload_host((dst), (src), (bytes)); if (u_) return u_; } while (0)
#define LSA_DOWNLOAD(dst, src, bytes) do { int u_ = download_host((dst), (src), (bytes)); if (u_) return u_; } while (0)    // all-gather landing zone shared by the sharded MSM and pairing paths

}  // namespace lsa

struct lsa_bases {
    void *d_aff = nullptr;   // prepared bases (msm_base_bytes(group) each); with a table: window-major copies
    size_t n = 0;
    int group = 1;           // 1 = G1, 2 = G2
    size_t table_stride = 0; // n when the pre-shifted window copies 2^(s_k)*P are resident, else 0
};

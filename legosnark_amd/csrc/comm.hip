// legosnark_amd/csrc/comm.hip -- the multi-GPU exchange step of the C-ABI (RCCL over xGMI).
//
// libff's multi_exp splits [0, n) into `chunks` contiguous ranges, runs multi_exp_inner per
// chunk and sums the partials (/root/reference/src/utils/globl.h:67-77 forwards `chunks`;
// SURVEY.md section 8e).  Here a chunk is a GPU and a rank is a process: every rank runs the
// single-GPU pipeline on its slice and contributes ONE Jacobian partial (96 B G1 / 192 B G2);
// ncclAllGather of 12 / 24 u64 per rank + k_sum_points replaces `final = final + partial[i]`
// (RCCL has no elliptic-curve reduce op, so reduce = gather + local fold; payload world x 96 B:
// latency-bound, link bandwidth irrelevant).  Pairing batches split the same way: per-rank
// Miller product, all-gather of the 384-byte Fq12 partials, product, one final exponentiation.
//
// The exchange of call i (wait for the MSM tail, all-gather, fold) runs on a side stream with
// buffers rotating over DEPTH calls, so lsa_stream() starts the front of call i+1 meanwhile.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <chrono>
#include <thread>

#include "capi_internal.h"

using namespace lsa;

namespace {

constexpr int DEPTH = 4;

struct Comm {
    bool active = false;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t side = nullptr;
    hipEvent_t front = nullptr;
    void *partial[DEPTH] = {};       // 384 B each (G1 96, G2 192, Fq12 384)
    void *gathered[DEPTH] = {};      // world x 384 B
    hipEvent_t done[DEPTH] = {};
    bool used[DEPTH] = {};
    unsigned calls = 0;
#if defined(LSA_COMM_TEST_LOOPBACK)
    // TEST BUILD ONLY (liblegosnark_amd_loopback.so, csrc/Makefile; the product library does not contain it): env
    // LSA_COMM_LOOPBACK=W with a one-rank communicator behaves as rank 0 of W ranks whose peers all contribute THIS
    // rank's partial -- the collective becomes W device copies, so the whole multi-rank step (side stream, rotating
    // buffers, events, fold) runs on one GPU, where RCCL itself refuses more than one rank; the result is W times
    // the local one
    int loopback = 0;
#endif
} c;

// all-gather of `bytes` per rank on `stream` (ncclAllGather of u64 words, or the loopback copies)
int gather_all(const void *send, void *recv, size_t bytes, hipStream_t stream) {
#if defined(LSA_COMM_TEST_LOOPBACK)
    if (c.loopback) {
        for (int r = 0; r < c.world; r++)
            if (hipMemcpyAsync((char *)recv + (size_t)r * bytes, send, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) {
                set_error("comm loopback copy failed");
                return LSA_ERR_HIP;
            }
        return LSA_OK;
    }
#endif
    ncclResult_t r_ = ncclAllGather(send, recv, bytes / 8, ncclUint64, c.comm, stream);
    if (r_ != ncclSuccess) { set_error("ncclAllGather failed: %s", ncclGetErrorString(r_)); return LSA_ERR_HIP; }
    return LSA_OK;
}

#define NCCLCHK(x)                                                                      \
    do {                                                                                \
        ncclResult_t r_ = (x);                                                          \
        if (r_ != ncclSuccess) {                                                        \
            set_error("%s failed: %s (%s:%d)", #x, ncclGetErrorString(r_), __FILE__, __LINE__); \
            return LSA_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

int require_comm() {
    int rc = require_ready();
    if (rc) return rc;
    if (!c.active) { set_error("legosnark_amd: no communicator (call lsa_comm_init first)"); return LSA_ERR_INVALID; }
    return LSA_OK;
}

// side stream: after everything queued on lsa_stream() so far and after the MSM tails issued so far
int side_after_main_and_tails() {
    HIPCHK(hipEventRecord(c.front, g.stream));
    HIPCHK(hipStreamWaitEvent(c.side, c.front, 0));
    return msm_join_to(c.side);
}

int next_slot() {
    int j = (int)(c.calls++ % DEPTH);
    return j;
}

}  // namespace

namespace lsa {
void comm_release() {
    if (!c.active) return;
    (void)hipStreamSynchronize(c.side);
    (void)ncclCommDestroy(c.comm);
    for (int j = 0; j < DEPTH; j++) {
        if (c.partial[j]) (void)hipFree(c.partial[j]);
        if (c.gathered[j]) (void)hipFree(c.gathered[j]);
        if (c.done[j]) (void)hipEventDestroy(c.done[j]);
    }
    if (c.front) (void)hipEventDestroy(c.front);
    if (c.side) (void)hipStreamDestroy(c.side);
    c = Comm();
}
}  // namespace lsa

template <class F>
static int msm_sharded_async(const lsa_bases *b, size_t first, const void *d_scalars, size_t n, void *d_out) {
    if (c.world == 1) return msm_device<F>(b->d_aff, first, (const Fr *)d_scalars, n, (Jac<F> *)d_out, g.stream, b->table_stride);
    const int j = next_slot();
    if (c.used[j]) HIPCHK(hipStreamWaitEvent(g.stream, c.done[j], 0));     // buffer set j was last used DEPTH calls ago
    int rc = msm_device<F>(b->d_aff, first, (const Fr *)d_scalars, n, (Jac<F> *)c.partial[j], g.stream, b->table_stride);
    if (rc) return rc;
    rc = side_after_main_and_tails();
    if (rc) return rc;
    rc = gather_all(c.partial[j], c.gathered[j], sizeof(Jac<F>), c.side);
    if (rc) return rc;
    rc = sum_points_device<F>((const Jac<F> *)c.gathered[j], (size_t)c.world, (Jac<F> *)d_out, c.side);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c.done[j], c.side));
    c.used[j] = true;
    return LSA_OK;
}

extern "C" {

int lsa_comm_unique_id(void *out_id128) {
    if (!out_id128) { set_error("comm_unique_id: null argument"); return LSA_ERR_INVALID; }
    static_assert(sizeof(ncclUniqueId) == LSA_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    memcpy(out_id128, &id, sizeof id);
    return LSA_OK;
}

int lsa_comm_init(int rank, int world, const void *id128) {
    int rc = require_ready();
    if (rc) return rc;
    if (world < 1 || rank < 0 || rank >= world || !id128) { set_error("comm_init: bad rank/world/id"); return LSA_ERR_INVALID; }
    if (c.active) comm_release();
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    HIPCHK(hipSetDevice(g.device));
    NCCLCHK(ncclCommInitRank(&c.comm, world, id, rank));
    c.rank = rank;
    c.world = world;
#if defined(LSA_COMM_TEST_LOOPBACK)
    const char *lb = getenv("LSA_COMM_LOOPBACK");
    if (world == 1 && lb && atoi(lb) > 1 && atoi(lb) <= 64) { c.loopback = atoi(lb); c.world = c.loopback; }
#endif
    HIPCHK(hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c.front, hipEventDisableTiming));
    for (int j = 0; j < DEPTH; j++) {
        HIPCHK(hipMalloc(&c.partial[j], 384));
        HIPCHK(hipMalloc(&c.gathered[j], (size_t)c.world * 384));
        HIPCHK(hipEventCreateWithFlags(&c.done[j], hipEventDisableTiming));
    }
    c.active = true;
    return LSA_OK;
}

// when this process started (seconds since the epoch): /proc/self/stat field 22 is in clock ticks since boot; the
// library's own first use is a good enough stand-in when /proc is not there
static time_t process_start_time() {
    static const time_t first_use = time(nullptr);
    FILE *f = fopen("/proc/self/stat", "r");
    if (!f) return first_use;
    char buf[2048];
    size_t got = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[got] = 0;
    const char *p = strrchr(buf, ')');           // the command name may contain spaces
    if (!p) return first_use;
    unsigned long long ticks = 0;
    int field = 2;
    for (p++; *p && field < 22; p++) if (*p == ' ') field++;
    if (field != 22 || sscanf(p, "%llu", &ticks) != 1) return first_use;
    double up = 0;
    FILE *u = fopen("/proc/uptime", "r");
    if (!u) return first_use;
    const int ok = fscanf(u, "%lf", &up);
    fclose(u);
    if (ok != 1) return first_use;
    const long hz = sysconf(_SC_CLK_TCK);
    return (time_t)((double)time(nullptr) - up + (double)ticks / (double)(hz > 0 ? hz : 100));
}

// single-node bootstrap for C++ callers without their own transport: rank 0 writes the id to
// `path` (atomically, via rename), the other ranks poll for it
int lsa_comm_init_file(int rank, int world, const char *path, int timeout_s) {
    int rc = require_ready();
    if (rc) return rc;
    if (!path || world < 1 || rank < 0 || rank >= world) { set_error("comm_init_file: bad argument"); return LSA_ERR_INVALID; }
    unsigned char id[LSA_COMM_ID_BYTES];
    if (rank == 0) {
        rc = lsa_comm_unique_id(id);
        if (rc) return rc;
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp.%d", path, (int)getpid());
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) { if (f) fclose(f); set_error("comm_init_file: cannot write %s", tmp); return LSA_ERR_INVALID; }
        fclose(f);
        if (rename(tmp, path) != 0) { set_error("comm_init_file: cannot rename %s -> %s", tmp, path); return LSA_ERR_INVALID; }
    } else {
        // a file left behind by an earlier (crashed) run must not be taken for this run's id: only a file written
        // after this process started counts (one node, one clock; the ranks of a job start together)
        const auto t0 = std::chrono::steady_clock::now();
        const time_t not_before = process_start_time() - 2;
        for (;;) {
            struct stat sb;
            FILE *f = (stat(path, &sb) == 0 && sb.st_mtime >= not_before) ? fopen(path, "rb") : nullptr;
            if (f) {
                size_t got = fread(id, 1, sizeof id, f);
                fclose(f);
                if (got == sizeof id) break;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > (timeout_s > 0 ? timeout_s : 60)) {
                set_error("comm_init_file: timed out waiting for %s", path);
                return LSA_ERR_INVALID;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    return lsa_comm_init(rank, world, id);
}

void lsa_comm_destroy(void) { comm_release(); }
int lsa_comm_rank(void) { return c.active ? c.rank : 0; }
int lsa_comm_world(void) { return c.active ? c.world : 1; }

void lsa_shard_range(size_t n, int world, int rank, size_t *lo, size_t *hi) {
    // libff multi_exp: one = n / chunks, the last chunk takes the remainder; n < chunks: no split
    size_t l, h;
    if (world <= 1 || n < (size_t)world) { l = rank == 0 ? 0 : n; h = n; }
    else {
        size_t one = n / (size_t)world;
        l = (size_t)rank * one;
        h = rank == world - 1 ? n : l + one;
    }
    if (lo) *lo = l;
    if (hi) *hi = h;
}

int lsa_msm_run_sharded_async(const lsa_bases *bases, size_t first, const void *d_scalars, size_t n, void *d_out_jac) {
    int rc = require_comm();
    if (rc) return rc;
    if (!bases || !d_out_jac || (n && !d_scalars)) { set_error("msm_run_sharded: null argument"); return LSA_ERR_INVALID; }
    if (first > bases->n || n > bases->n - first) { set_error("msm_run_sharded: range [%zu,%zu) exceeds %zu bases", first, first + n, bases->n); return LSA_ERR_INVALID; }
    if (bases->group == 1) return msm_sharded_async<Fq>(bases, first, d_scalars, n, d_out_jac);
    return msm_sharded_async<Fq2>(bases, first, d_scalars, n, d_out_jac);
}

int lsa_comm_join(void) {
    int rc = require_comm();
    if (rc) return rc;
    rc = msm_join(g.stream);
    if (rc) return rc;
    for (int j = 0; j < DEPTH; j++)
        if (c.used[j]) HIPCHK(hipStreamWaitEvent(g.stream, c.done[j], 0));
    return LSA_OK;
}

int lsa_msm_run_sharded(const lsa_bases *bases, size_t first, const void *d_scalars, size_t n, void *out_jac) {
    if (!out_jac) { set_error("msm_run_sharded: null argument"); return LSA_ERR_INVALID; }
    int rc = lsa_msm_run_sharded_async(bases, first, d_scalars, n, g.d_result);
    if (rc) return rc;
    rc = lsa_comm_join();
    if (rc) return rc;
    const size_t bytes = bases->group == 1 ? sizeof(Jac<Fq>) : sizeof(Jac<Fq2>);
    HIPCHK(hipMemcpyAsync(g.h_result, g.d_result, bytes, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    memcpy(out_jac, g.h_result, bytes);
    return LSA_OK;
}

// d_partial (device; 12 / 24 / 48 u64 for kind 1 / 2 / 12) of every rank -> d_gathered
// (world x the same), ordered after everything queued on lsa_stream(); lsa_stream() waits for it
int lsa_comm_all_gather(const void *d_partial, void *d_gathered, int kind) {
    int rc = require_comm();
    if (rc) return rc;
    const size_t words = kind == 1 ? 12 : kind == 2 ? 24 : kind == 12 ? 48 : 0;
    if (!words || !d_partial || !d_gathered) { set_error("comm_all_gather: bad argument"); return LSA_ERR_INVALID; }
    rc = msm_join(g.stream);
    if (rc) return rc;
    if (c.world == 1) {
        HIPCHK(hipMemcpyAsync(d_gathered, d_partial, words * 8, hipMemcpyDeviceToDevice, g.stream));
        return LSA_OK;
    }
    return gather_all(d_partial, d_gathered, words * 8, g.stream);
}

}  // extern "C"

// legosnark_amd/shim/libff/lsa_libff.hpp -- libff-compatible C++17 surface over the C-ABI
// in include/legosnark_amd.h, so that LegoSNARK's unchanged sources
// (/root/reference/src/{utils,prototools,gadgets,examples}) compile and link against the
// MI355X library instead of libff/libsnark.
//
// What lives where:
//  * HOT PATH (forwarded to the GPU through the C-ABI; no CPU fallback):
//      multi_exp, multi_exp_with_mixed_addition            -> lsa_g1_msm / lsa_g2_msm
//      batch_exp                                           -> lsa_g1_batch_exp / lsa_g2_batch_exp
//      miller_loop, double_miller_loop, final_exponentiation, reduced_pairing
//                                                          -> lsa_miller_loop*, lsa_final_exponentiation, lsa_pairing_product
//  * COLD host glue (single field / group operators the gadgets use a handful of times):
//      inline host code from csrc/fp.h, ec.h, tower.h (the same formulas the kernels use).
// Class layouts are libff's (Fr/Fq 32 B Montgomery limbs, G1 {X,Y,Z} 96 B, G2 192 B,
// GT 384 B), so std::vector<G1<pp>> crosses the C-ABI as a plain pointer.
// Names mirror the libff symbols the reference uses.
#pragma once
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <memory>
#include <mutex>
#include <random>
#include <stdexcept>
#include <cerrno>
#include <climits>
#include <malloc.h>
#include <pthread.h>
#include <sys/random.h>
#include <sys/types.h>
#include <string>
#include <vector>

#include "../../../include/legosnark_amd.h"
#include "../../csrc/ec.h"
#include "../../csrc/tower.h"
#include "../../csrc/fixed_base.h"
#include "../../csrc/smul_host.h"
#include <future>
#include <algorithm>
#include <atomic>
#include <functional>
#include <chrono>
#include <cstdlib>
// The reference's -DMULTICORE=ON build (/root/reference/CMakeLists.txt:35-39,57-59,78-80: -fopenmp -DMULTICORE=1) calls
// omp_get_max_threads() without including <omp.h> itself (src/utils/globl.h:52,68; src/utils/sparsemexp.cc:6,17): libff's
// headers pull it in under that macro [upstream, recalled: libff/algebra/scalar_multiplication/multiexp.tcc], so this one
// does too.  `chunks` is then the OpenMP thread count; the GPU computes the same sum whatever its value.  The four
// `#pragma omp parallel for` loops of src/gadgets/lipmaa.cc:125-172 run this header's Fr operators concurrently: they are
// pure functions of their operands (no shared state; random_element's pool is thread_local, the statistics are atomics).
#ifdef MULTICORE
#include <omp.h>
#endif

namespace libff {

// LSA_SHIM_STATS=1: at exit, one JSON line on stderr saying how many calls of each kind the program made through this
// header and how long it spent inside them -- the rest of a reference binary's run time is the reference's own host
// code (witness recursions, sumcheck tables, vector copies).  One relaxed atomic add per call when off.
namespace lsa_shim {
enum StatKind { ST_MSM_G1, ST_MSM_G2, ST_PAIRING, ST_G2_PRECOMP, ST_BATCH_EXP, ST_NORMALIZE, ST_SCALAR_MUL_HOST, ST_SPARSE_MSM, ST_KINDS };
struct Stats {
    std::atomic<uint64_t> calls[ST_KINDS], ns[ST_KINDS], items[ST_KINDS];
    std::atomic<uint64_t> msm_us[4], msm_hits, msm_tables;     // lsa_msm_host_stats summed: h2d, fingerprint wait, bases, kernels
    // host scalar multiplications by group (G1, G2) and route: fixed-base table / scalar below 2^16 / any other (GLV ladder)
    std::atomic<uint64_t> smul_n[2][3] = {}, smul_ns[2][3] = {};
    bool on = false, each = false;                              // LSA_SHIM_STATS=1: totals at exit; =2: also one line per forwarded call
    std::chrono::steady_clock::time_point born = std::chrono::steady_clock::now();
    static const char *name(int k) {
        static const char *const n[ST_KINDS] = {"msm_g1", "msm_g2", "pairing", "g2_precompute", "batch_exp", "normalize", "scalar_mul_host", "sparse_msm"};
        return n[k];
    }
    static void report() {
        Stats &s = get();
        const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - s.born).count();
        double inside = 0;
        fprintf(stderr, "{\"lsa_shim_stats\": {");
        for (int k = 0; k < ST_KINDS; k++) {
            const double ms = (double)s.ns[k].load() * 1e-6;
            inside += ms;
            fprintf(stderr, "\"%s\": {\"calls\": %llu, \"items\": %llu, \"ms\": %.3f}, ", name(k), (unsigned long long)s.calls[k].load(),
                    (unsigned long long)s.items[k].load(), ms);
        }
        fprintf(stderr, "\"msm_host_path\": {\"h2d_scalars_ms\": %.3f, \"fingerprint_wait_ms\": %.3f, \"bases_prepare_ms\": %.3f, \"kernels_ms\": %.3f, "
                        "\"cache_hits\": %llu, \"on_pre_shifted_copies\": %llu}, ",
                (double)s.msm_us[0].load() * 1e-3, (double)s.msm_us[1].load() * 1e-3, (double)s.msm_us[2].load() * 1e-3, (double)s.msm_us[3].load() * 1e-3,
                (unsigned long long)s.msm_hits.load(), (unsigned long long)s.msm_tables.load());
        fprintf(stderr, "\"scalar_mul_host_split\": {");
        for (int g = 0; g < 2; g++) {
            static const char *const route[3] = {"fixed_base_table", "scalar_below_2^16", "other_base"};
            for (int r = 0; r < 3; r++)
                fprintf(stderr, "%s\"g%d_%s\": {\"calls\": %llu, \"ms\": %.3f}", g + r ? ", " : "", g + 1, route[r],
                        (unsigned long long)s.smul_n[g][r].load(), (double)s.smul_ns[g][r].load() * 1e-6);
        }
        fprintf(stderr, "}, ");
        fprintf(stderr, "\"inside_ms\": %.3f, \"process_ms\": %.3f}}\n", inside, wall);
    }
    static Stats &get() {
        static Stats *s = [] {
            Stats *t = new Stats();
            for (int k = 0; k < ST_KINDS; k++) { t->calls[k] = 0; t->ns[k] = 0; t->items[k] = 0; }
            for (int k = 0; k < 4; k++) t->msm_us[k] = 0;
            t->msm_hits = 0;
            t->msm_tables = 0;
            const char *e = getenv("LSA_SHIM_STATS");
            if (e && (e[0] == '1' || e[0] == '2')) { t->on = true; t->each = e[0] == '2'; atexit(report); }
            return t;
        }();
        return *s;
    }
};
struct StatScope {
    const int kind;
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    StatScope(int k, size_t items) : kind(k) {
        Stats &s = Stats::get();
        s.calls[k].fetch_add(1, std::memory_order_relaxed);
        s.items[k].fetch_add(items, std::memory_order_relaxed);
    }
    ~StatScope() {
        Stats &s = Stats::get();
        const auto t1 = std::chrono::steady_clock::now();
        s.ns[kind].fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count(), std::memory_order_relaxed);
        if (s.each)
            fprintf(stderr, "lsa_shim_call %s at_ms %.3f took_ms %.3f\n", Stats::name(kind), std::chrono::duration<double, std::milli>(t0 - s.born).count(),
                    std::chrono::duration<double, std::milli>(t1 - t0).count());
    }
};
struct SmulScope {                                                  // one host scalar multiplication: its route is set before it returns
    const int group;
    int route = 2;
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit SmulScope(int g) : group(g - 1) {}
    ~SmulScope() {
        Stats &s = Stats::get();
        s.smul_n[group][route].fetch_add(1, std::memory_order_relaxed);
        s.smul_ns[group][route].fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(),
                                          std::memory_order_relaxed);
    }
};
}  // namespace lsa_shim

// ---------------------------------------------------------------- errors
inline void lsa_require(int rc, const char *what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + lsa_last_error());
}

// The C-ABI is written for ONE calling thread at a time (SURVEY.md section 8b "Threading"; the reference never issues two
// library calls concurrently, MULTICORE or not).  Every forwarding site of this header takes this lock, so that a caller
// who does reach the library from several threads (an OpenMP loop around commitments, say) is serialised instead of racing
// on the library's stream and staging buffers.  Recursive: window_table::rows() forwards from inside batch_exp's callers.
namespace lsa_shim {
inline std::recursive_mutex &gpu_mutex() { static std::recursive_mutex m; return m; }
struct GpuLock {
    std::lock_guard<std::recursive_mutex> g;
    GpuLock() : g(gpu_mutex()) {}
};
}  // namespace lsa_shim

// ---------------------------------------------------------------- profiling stubs (libff/common/profiling.hpp)
inline bool &inhibit_profiling_info_ref() { static bool b = false; return b; }
#define inhibit_profiling_info (::libff::inhibit_profiling_info_ref())
inline bool &inhibit_profiling_counters_ref() { static bool b = false; return b; }
#define inhibit_profiling_counters (::libff::inhibit_profiling_counters_ref())
inline void print_indent() {}
inline void print_header(const char *msg) { printf("\n================================================================================\n%s\n================================================================================\n\n", msg); }
inline void enter_block(const std::string &, bool = false) {}
inline void leave_block(const std::string &, bool = false) {}
inline void start_profiling() {}
inline size_t log2(size_t n) {   // ceil(log2 n)
    size_t r = ((n & (n - 1)) == 0 ? 0 : 1);
    while (n > 1) { n >>= 1; r++; }
    return r;
}
#ifndef UNUSED
#define UNUSED(x) (void)(x)
#endif

// ---------------------------------------------------------------- bigint<4>
template <long N>
struct bigint {
    uint64_t data[N] = {0};
    bigint() = default;
    bigint(unsigned long x) { data[0] = x; }
    bool test_bit(size_t i) const { return i < 64 * N && ((data[i >> 6] >> (i & 63)) & 1); }
    size_t max_bits() const { return 64 * N; }
    size_t num_bits() const {
        for (long i = N - 1; i >= 0; --i) if (data[i]) return 64 * i + (64 - __builtin_clzll(data[i]));
        return 0;
    }
    unsigned long as_ulong() const { return data[0]; }
    bool operator==(const bigint &o) const { return memcmp(data, o.data, sizeof data) == 0; }
    bool operator!=(const bigint &o) const { return !(*this == o); }
};

// Randomness of random_element(): trapdoors, blinding factors and published challenges of the
// gadgets come from here (src/gadgets/subspace.cc:47, src/examples/cplink.cc:70,
// src/gadgets/sigma.cc:11, src/gadgets/sumcheck.cc:47), so every draw reads the operating
// system's CSPRNG (getrandom(2); std::random_device as the fallback) -- libff's
// bigint::randomize() reads std::random_device for every limb [upstream, recalled].  A
// reproducible stream exists only under the explicit test-only macro LSA_SHIM_TEST_SEED
// (then env LSA_SEED seeds a Mersenne twister): never define it in a production build.
#ifdef LSA_SHIM_TEST_SEED
// (test builds only: checks/resident_prover_check.cc replays one transcript through two provers)
inline std::mt19937_64 &lsa_test_rng() {
    static std::mt19937_64 g = []() {
        const char *s = getenv("LSA_SEED");
        return std::mt19937_64(s ? strtoull(s, nullptr, 0) : 0x4C45474F534E4152ull);
    }();
    return g;
}
inline void lsa_test_reseed(uint64_t seed) { lsa_test_rng().seed(seed); }
#endif
inline void lsa_random_bytes(void *buf, size_t len) {
#ifdef LSA_SHIM_TEST_SEED
    std::mt19937_64 &g = lsa_test_rng();
    unsigned char *p = (unsigned char *)buf;
    for (size_t i = 0; i < len; i += 8) { uint64_t r = g(); memcpy(p + i, &r, len - i < 8 ? len - i : 8); }
#else
    // One system call per 4 KiB, not per field element (a vector of 2^20 random scalars is a million draws): a
    // per-thread pool of CSPRNG output, every byte handed out once and wiped, emptied in the child of a fork.
    struct Pool {
        unsigned char bytes[4096];
        size_t left = 0;
        static void os_fill(unsigned char *p, size_t len) {
            size_t got = 0;
            while (got < len) {
                ssize_t r = getrandom(p + got, len - got, 0);
                if (r > 0) { got += (size_t)r; continue; }
                if (r < 0 && errno == EINTR) continue;
                break;                             // ENOSYS etc.: fall back below
            }
            if (got < len) {
                std::random_device rd;
                for (; got < len; got++) p[got] = (unsigned char)rd();
            }
        }
        static Pool &mine() {
            static thread_local Pool pool;
            static const int registered = pthread_atfork(nullptr, nullptr, [] { Pool &q = mine(); memset(q.bytes, 0, sizeof q.bytes); q.left = 0; });
            (void)registered;
            return pool;
        }
    };
    unsigned char *p = (unsigned char *)buf;
    if (len > sizeof(Pool::bytes) / 4) { Pool::os_fill(p, len); return; }
    Pool &pool = Pool::mine();
    if (pool.left < len) { Pool::os_fill(pool.bytes, sizeof pool.bytes); pool.left = sizeof pool.bytes; }
    pool.left -= len;
    memcpy(p, pool.bytes + pool.left, len);
    memset(pool.bytes + pool.left, 0, len);
#endif
}

// ---------------------------------------------------------------- prime fields
template <class P>
class Fp_shim {
public:
    lsa::Fp<P> v;
    static const long num_limbs = 4;
    static const Fp_shim multiplicative_generator;   // 5 generates Fr^* (libff alt_bn128_init)
    Fp_shim() : v(lsa::Fp<P>::zero()) {}
    Fp_shim(const lsa::Fp<P> &x) : v(x) {}
    Fp_shim(long x) { set_long(x); }
    Fp_shim(int x) { set_long(x); }
    Fp_shim(unsigned long x) { set_ulong(x); }
    Fp_shim(unsigned long long x) { set_ulong(x); }
    Fp_shim(unsigned int x) { set_ulong(x); }
    Fp_shim(const char *dec) {   // decimal string, as libff's bigint(const char*)
        Fp_shim acc, ten(10L);
        for (const char *c = dec; *c; ++c) { if (*c < '0' || *c > '9') continue; acc = acc * ten + Fp_shim((long)(*c - '0')); }
        v = acc.v;
    }
    Fp_shim(const bigint<4> &b) {
        uint32_t l[8];
        for (int i = 0; i < 4; i++) { l[2 * i] = (uint32_t)b.data[i]; l[2 * i + 1] = (uint32_t)(b.data[i] >> 32); }
        v = lsa::Fp<P>::from_canonical(l);
    }
    void set_ulong(unsigned long long x) {
        uint32_t l[8] = {(uint32_t)x, (uint32_t)(x >> 32), 0, 0, 0, 0, 0, 0};
        v = lsa::Fp<P>::from_canonical(l);
    }
    void set_long(long x) {
        if (x >= 0) set_ulong((unsigned long long)x);
        else { set_ulong((unsigned long long)(-x)); v = v.neg(); }
    }
    static Fp_shim zero() { return Fp_shim(); }
    static Fp_shim one() { return Fp_shim(lsa::Fp<P>::one()); }
    static size_t size_in_bits() { return 254; }
    static size_t capacity() { return 253; }
    static Fp_shim random_element() {
        for (;;) {
            uint32_t l[8];
            lsa_random_bytes(l, sizeof l);
            l[7] &= 0x3fffffffu;
            bool lt = false;   // l < MOD ?
            for (int i = 7; i >= 0; --i) { if (l[i] != P::MOD[i]) { lt = l[i] < P::MOD[i]; break; } }
            if (lt) { Fp_shim r; for (int i = 0; i < 8; i++) r.v.l[i] = l[i]; return r; }   // random Montgomery residue
        }
    }
    bool is_zero() const { return v.is_zero(); }
    bool operator==(const Fp_shim &o) const { return v == o.v; }
    bool operator!=(const Fp_shim &o) const { return !(v == o.v); }
    Fp_shim operator+(const Fp_shim &o) const { return Fp_shim(v + o.v); }
    Fp_shim operator-(const Fp_shim &o) const { return Fp_shim(v - o.v); }
    Fp_shim operator*(const Fp_shim &o) const { return Fp_shim(v * o.v); }
    Fp_shim operator-() const { return Fp_shim(v.neg()); }
    Fp_shim &operator+=(const Fp_shim &o) { v = v + o.v; return *this; }
    Fp_shim &operator-=(const Fp_shim &o) { v = v - o.v; return *this; }
    Fp_shim &operator*=(const Fp_shim &o) { v = v * o.v; return *this; }
    Fp_shim squared() const { return Fp_shim(v.sqr()); }
    Fp_shim inverse() const { return Fp_shim(v.inverse()); }
    Fp_shim operator^(unsigned long e) const {
        Fp_shim acc = one(), b = *this;
        while (e) { if (e & 1) acc *= b; b = b.squared(); e >>= 1; }
        return acc;
    }
    Fp_shim operator^(const bigint<4> &e) const {
        Fp_shim acc = one();
        for (long i = 255; i >= 0; --i) { acc = acc.squared(); if (e.test_bit(i)) acc *= *this; }
        return acc;
    }
    bigint<4> as_bigint() const {
        uint32_t l[8];
        v.to_canonical(l);
        bigint<4> b;
        for (int i = 0; i < 4; i++) b.data[i] = (uint64_t)l[2 * i] | ((uint64_t)l[2 * i + 1] << 32);
        return b;
    }
    unsigned long as_ulong() const { return as_bigint().as_ulong(); }
    void print() const { std::cout << *this << "\n"; }
    // libff Fp_model::sqrt for p = 3 mod 4 (Fq; used by point decompression): a^((p+1)/4)
    Fp_shim sqrt() const {
        bigint<4> e;
        uint64_t m[4];
        for (int i = 0; i < 4; i++) m[i] = (uint64_t)P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32);
        m[0] += 1;                                              // p + 1 (no carry: p ends in ...47)
        for (int i = 0; i < 4; i++) e.data[i] = (m[i] >> 2) | (i < 3 ? (m[i + 1] << 62) : 0);
        return *this ^ e;
    }
    // Serialisation = libff's operator<< / operator>> for Fp_model (fp.tcc) and bigint
    // (bigint.tcc) [upstream, recalled], under the same macros libff's build defines:
    //   default            canonical value, decimal text
    //   MONTGOMERY_OUTPUT  the Montgomery representation instead of the canonical value
    //   BINARY_OUTPUT      the 32 raw little-endian bytes instead of decimal text
    // (/root/reference/src/utils/util.h:56-96 streams vectors of these with `<<` / `>>`.)
    static void write_bigint(std::ostream &os, const bigint<4> &b) {
#ifdef BINARY_OUTPUT
        os.write(reinterpret_cast<const char *>(b.data), sizeof b.data);
#else
        uint32_t w[8];
        for (int i = 0; i < 4; i++) { w[2 * i] = (uint32_t)b.data[i]; w[2 * i + 1] = (uint32_t)(b.data[i] >> 32); }
        std::string s;
        bool nz = true;
        while (nz) {
            uint64_t rem = 0; nz = false;
            for (int i = 7; i >= 0; --i) { uint64_t cur = (rem << 32) | w[i]; w[i] = (uint32_t)(cur / 10); rem = cur % 10; if (w[i]) nz = true; }
            s.push_back((char)('0' + rem));
        }
        for (size_t i = 0; i < s.size() / 2; i++) std::swap(s[i], s[s.size() - 1 - i]);
        os << s;
#endif
    }
    static bool read_bigint(std::istream &is, bigint<4> &b) {
#ifdef BINARY_OUTPUT
        is.read(reinterpret_cast<char *>(b.data), sizeof b.data);
        return (bool)is;
#else
        std::string s;
        is >> s;
        if (s.empty()) return false;
        uint32_t w[8] = {0};
        for (char c : s) {
            if (c < '0' || c > '9') { is.setstate(std::ios::failbit); return false; }
            uint64_t carry = (uint64_t)(c - '0');
            for (int i = 0; i < 8; i++) { uint64_t cur = (uint64_t)w[i] * 10 + carry; w[i] = (uint32_t)cur; carry = cur >> 32; }
        }
        for (int i = 0; i < 4; i++) b.data[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
        return true;
#endif
    }
    friend std::ostream &operator<<(std::ostream &os, const Fp_shim &a) {
#ifdef MONTGOMERY_OUTPUT
        bigint<4> b;
        for (int i = 0; i < 4; i++) b.data[i] = (uint64_t)a.v.l[2 * i] | ((uint64_t)a.v.l[2 * i + 1] << 32);
        write_bigint(os, b);
#else
        write_bigint(os, a.as_bigint());
#endif
        return os;
    }
    friend std::istream &operator>>(std::istream &is, Fp_shim &a) {
        bigint<4> b;
        if (!read_bigint(is, b)) return is;
#ifdef MONTGOMERY_OUTPUT
        for (int i = 0; i < 4; i++) { a.v.l[2 * i] = (uint32_t)b.data[i]; a.v.l[2 * i + 1] = (uint32_t)(b.data[i] >> 32); }
#else
        a = Fp_shim(b);
#endif
        return is;
    }
};

// libff serialization.hpp: separators are empty in binary mode
#ifdef BINARY_OUTPUT
#define LSA_OUTPUT_SEPARATOR ""
#define LSA_OUTPUT_NEWLINE ""
#else
#define LSA_OUTPUT_SEPARATOR " "
#define LSA_OUTPUT_NEWLINE "\n"
#endif
inline void consume_OUTPUT_SEPARATOR(std::istream &in) {
#ifndef BINARY_OUTPUT
    char c;
    in.read(&c, 1);
#else
    (void)in;
#endif
}
inline void consume_OUTPUT_NEWLINE(std::istream &in) { consume_OUTPUT_SEPARATOR(in); }
inline void consume_newline(std::istream &in) { char c; in.read(&c, 1); }

template <class P>
const Fp_shim<P> Fp_shim<P>::multiplicative_generator = Fp_shim<P>(5L);

using alt_bn128_Fr = Fp_shim<lsa::FrParams>;
using alt_bn128_Fq = Fp_shim<lsa::FqParams>;

inline alt_bn128_Fr fr_multiplicative_generator() { return alt_bn128_Fr(5L); }

// 2^s-th primitive root of unity in Fr (r - 1 = 2^28 * odd); libff get_root_of_unity
template <class FieldT>
FieldT get_root_of_unity(size_t n) {
    const size_t logn = log2(n);
    if (n != (size_t(1) << logn) || logn > 28) throw std::invalid_argument("get_root_of_unity: n must be a power of two <= 2^28");
    // omega = 5^((r-1)/2^28)
    bigint<4> e;   // (r-1) >> 28
    {
        uint64_t r[4];
        for (int i = 0; i < 4; i++) r[i] = (uint64_t)lsa::FrParams::MOD[2 * i] | ((uint64_t)lsa::FrParams::MOD[2 * i + 1] << 32);
        r[0] -= 1;
        for (int i = 0; i < 4; i++) e.data[i] = (r[i] >> 28) | (i < 3 ? (r[i + 1] << 36) : 0);
    }
    FieldT omega = FieldT(5L) ^ e;
    for (size_t i = 28; i > logn; --i) omega = omega.squared();
    return omega;
}

// ---------------------------------------------------------------- Fq2 / GT
class alt_bn128_Fq2 {
public:
    lsa::Fq2 v;
    alt_bn128_Fq2() : v(lsa::Fq2::zero()) {}
    alt_bn128_Fq2(const lsa::Fq2 &x) : v(x) {}
    static alt_bn128_Fq2 zero() { return alt_bn128_Fq2(); }
    static alt_bn128_Fq2 one() { return alt_bn128_Fq2(lsa::Fq2::one()); }
    bool operator==(const alt_bn128_Fq2 &o) const { return v == o.v; }
    bool operator!=(const alt_bn128_Fq2 &o) const { return !(v == o.v); }
    alt_bn128_Fq2 operator*(const alt_bn128_Fq2 &o) const { return alt_bn128_Fq2(v * o.v); }
    alt_bn128_Fq2 operator+(const alt_bn128_Fq2 &o) const { return alt_bn128_Fq2(v + o.v); }
    alt_bn128_Fq2 operator-() const { return alt_bn128_Fq2(v.neg()); }
    alt_bn128_Fq2 squared() const { return alt_bn128_Fq2(v.sqr()); }
    // a square root in Fq[u]/(u^2+1) by the norm method; libff uses Tonelli-Shanks in Fq2 --
    // either root serves point decompression, which fixes the sign by the parity of c0.
    // Returns false when the element is not a square.
    bool sqrt(alt_bn128_Fq2 &out) const {
        const alt_bn128_Fq a0(v.c0), a1(v.c1);
        if (a1.is_zero()) {
            alt_bn128_Fq r = a0.sqrt();
            if (r.squared() == a0) { out.v = lsa::Fq2{r.v, lsa::Fq::zero()}; return true; }
            alt_bn128_Fq r2 = (-a0).sqrt();                       // a0 = -(r2^2) = (r2 u)^2
            if (r2.squared() != -a0) return false;
            out.v = lsa::Fq2{lsa::Fq::zero(), r2.v};
            return true;
        }
        const alt_bn128_Fq norm = a0.squared() + a1.squared();
        const alt_bn128_Fq sn = norm.sqrt();
        if (sn.squared() != norm) return false;
        const alt_bn128_Fq half = alt_bn128_Fq(2L).inverse();
        alt_bn128_Fq x0sq = (a0 + sn) * half;
        alt_bn128_Fq x0 = x0sq.sqrt();
        if (x0.squared() != x0sq) { x0sq = (a0 - sn) * half; x0 = x0sq.sqrt(); if (x0.squared() != x0sq) return false; }
        const alt_bn128_Fq x1 = a1 * (x0 + x0).inverse();
        out.v = lsa::Fq2{x0.v, x1.v};
        return true;
    }
    // libff Fp2_model operator<< / >>: c0 SEP c1
    friend std::ostream &operator<<(std::ostream &os, const alt_bn128_Fq2 &a) {
        return os << alt_bn128_Fq(a.v.c0) << LSA_OUTPUT_SEPARATOR << alt_bn128_Fq(a.v.c1);
    }
    friend std::istream &operator>>(std::istream &is, alt_bn128_Fq2 &a) {
        alt_bn128_Fq c0, c1;
        is >> c0 >> c1;
        a.v = lsa::Fq2{c0.v, c1.v};
        return is;
    }
};

// GT / Fqk values.  A value that comes out of miller_loop / double_miller_loop / reduced_pairing / final_exponentiation
// is DEFERRED: it is the description "[final_exponentiation] (prod of Miller loops (P_i, Q_i)^(+-1) * constant)" and
// becomes twelve field elements when somebody looks at it (==, <<, squared, a product with a plain value, val()).
// The reference's verifiers only ever multiply such values, conjugate them, final-exponentiate and compare
// (/root/reference/src/utils/globl.h:94-105, src/gadgets/subspace.cc:142-170, src/gadgets/poly.h:105-123,
// src/gadgets/lipmaa.cc:187-207), so a whole check
//     final_exponentiation(lhs * rhs.unitary_inverse()) == GT::one()
// reaches the GPU as ONE lsa_pairing_terms call -- one upload, shared accumulators, one final exponentiation, one
// download -- instead of four blocking calls.  The value is the same Fq12 element whichever way it is evaluated
// (multiplication in Fq12 is exact).  LSA_SHIM_EAGER=1 evaluates every call at once (the round-2 behaviour).
namespace lsa_shim {
struct PairTerm {
    lsa::Jac<lsa::Fq> P;
    lsa::Jac<lsa::Fq2> Q;
    uint8_t conj;
};
struct PairExpr {
    std::vector<PairTerm> terms;
    lsa::Fq12 cst = lsa::Fq12::one();      // product of the already evaluated factors
    bool has_cst = false;
    bool final_exp = false;
    // every copy of a deferred value shares its node, hence its result.  `done` is published with release order after
    // `result` is written (under gpu_mutex): copies of one value may be looked at from several threads.
    mutable std::atomic<bool> done{false};
    mutable lsa::Fq12 result;
    PairExpr() = default;
    PairExpr(const PairExpr &o) : terms(o.terms), cst(o.cst), has_cst(o.has_cst), final_exp(o.final_exp) {
        if (o.done.load(std::memory_order_acquire)) { result = o.result; done.store(true, std::memory_order_release); }
    }
    PairExpr &operator=(const PairExpr &o) {
        if (this != &o) {
            terms = o.terms; cst = o.cst; has_cst = o.has_cst; final_exp = o.final_exp;
            const bool d = o.done.load(std::memory_order_acquire);
            if (d) result = o.result;
            done.store(d, std::memory_order_release);
        }
        return *this;
    }
};
inline bool eager() {
    static const bool e = getenv("LSA_SHIM_EAGER") && getenv("LSA_SHIM_EAGER")[0] == '1';
    return e;
}
inline const lsa::Fq12 &evaluate(const PairExpr &e) {
    if (e.done.load(std::memory_order_acquire)) return e.result;
    GpuLock lock;
    if (e.done.load(std::memory_order_acquire)) return e.result;
    StatScope scope(ST_PAIRING, e.terms.size());
    lsa::Fq12 out = e.cst;
    const size_t n = e.terms.size();
    if (n) {
        std::vector<lsa::Jac<lsa::Fq>> ps(n);
        std::vector<lsa::Jac<lsa::Fq2>> qs(n);
        std::vector<uint8_t> fl(n);
        for (size_t i = 0; i < n; i++) { ps[i] = e.terms[i].P; qs[i] = e.terms[i].Q; fl[i] = e.terms[i].conj; }
        const uint64_t seg[2] = {0, n};
        lsa_require(lsa_pairing_terms(ps.data(), qs.data(), nullptr, fl.data(), seg, 1, &out, e.final_exp && !e.has_cst), "pairing product");
        if (e.has_cst) out = lsa::fq12_mul(out, e.cst);
    }
    if (e.final_exp && (e.has_cst || !n)) {
        lsa::Fq12 in = out;
        lsa_require(lsa_final_exponentiation(&in, 1, &out, 0), "final_exponentiation");
    }
    e.result = out;
    e.done.store(true, std::memory_order_release);
    return e.result;
}
}  // namespace lsa_shim

class alt_bn128_Fq12 {
    lsa::Fq12 v_;
    std::shared_ptr<const lsa_shim::PairExpr> e_;      // non-null: the value lives in the node (evaluated on demand)
public:
    alt_bn128_Fq12() : v_(lsa::Fq12::one()) { v_.c0.c0 = lsa::Fq2::zero(); }
    alt_bn128_Fq12(const lsa::Fq12 &x) : v_(x) {}
    explicit alt_bn128_Fq12(std::shared_ptr<const lsa_shim::PairExpr> e) : v_(lsa::Fq12::one()), e_(std::move(e)) {
        if (lsa_shim::eager()) val();
    }
    // the twelve field elements (libff layout); evaluates a deferred value
    // (a const object is never written: concurrent readers of one value, or of copies of it, are safe)
    const lsa::Fq12 &val() const { return e_ ? lsa_shim::evaluate(*e_) : v_; }
    bool deferred() const { return e_ != nullptr && !e_->done.load(std::memory_order_acquire); }
    static alt_bn128_Fq12 one() { return alt_bn128_Fq12(lsa::Fq12::one()); }
    static alt_bn128_Fq12 zero() { return alt_bn128_Fq12(); }
    bool operator==(const alt_bn128_Fq12 &o) const { return val() == o.val(); }
    bool operator!=(const alt_bn128_Fq12 &o) const { return !(val() == o.val()); }
    alt_bn128_Fq12 operator*(const alt_bn128_Fq12 &o) const {
        if (!deferred() && !o.deferred()) return alt_bn128_Fq12(lsa::fq12_mul(val(), o.val()));
        // a deferred factor that is not final-exponentiated contributes its terms; anything else its value
        auto r = std::make_shared<lsa_shim::PairExpr>();
        for (const alt_bn128_Fq12 *f : {this, &o}) {
            if (f->deferred() && !f->e_->final_exp) {
                r->terms.insert(r->terms.end(), f->e_->terms.begin(), f->e_->terms.end());
                if (f->e_->has_cst) { r->cst = r->has_cst ? lsa::fq12_mul(r->cst, f->e_->cst) : f->e_->cst; r->has_cst = true; }
            } else {
                const lsa::Fq12 &x = f->val();
                r->cst = r->has_cst ? lsa::fq12_mul(r->cst, x) : x;
                r->has_cst = true;
            }
        }
        if (r->has_cst && r->cst == lsa::Fq12::one()) r->has_cst = false;     // acc = Fqk::one(); acc = acc * miller_loop(...)
        return alt_bn128_Fq12(std::shared_ptr<const lsa_shim::PairExpr>(std::move(r)));
    }
    alt_bn128_Fq12 &operator*=(const alt_bn128_Fq12 &o) { *this = *this * o; return *this; }
    alt_bn128_Fq12 squared() const { return alt_bn128_Fq12(lsa::fq12_sqr(val())); }
    alt_bn128_Fq12 inverse() const { return alt_bn128_Fq12(lsa::fq12_inverse(val())); }
    // conjugation over Fq6 (the inverse of a unitary element).  It is a field automorphism, and the conjugate of a
    // Miller loop on (P, Q) is the Miller loop on (-P, Q): a deferred product conjugates term by term
    alt_bn128_Fq12 unitary_inverse() const {
        if (!deferred() || e_->final_exp) return alt_bn128_Fq12(val().unitary_inverse());
        auto r = std::make_shared<lsa_shim::PairExpr>(*e_);
        for (auto &t : r->terms) t.conj ^= 1;
        if (r->has_cst) r->cst = r->cst.unitary_inverse();
        return alt_bn128_Fq12(std::shared_ptr<const lsa_shim::PairExpr>(std::move(r)));
    }
    // libff final_exponentiation (alt_bn128_pp::final_exponentiation forwards here)
    alt_bn128_Fq12 final_exponentiated() const {
        auto r = std::make_shared<lsa_shim::PairExpr>();
        if (deferred() && !e_->final_exp) *r = *e_;
        else { r->cst = val(); r->has_cst = true; }
        r->final_exp = true;
        r->done.store(false, std::memory_order_release);
        return alt_bn128_Fq12(std::shared_ptr<const lsa_shim::PairExpr>(std::move(r)));
    }
    alt_bn128_Fq12 operator^(const bigint<4> &e) const {
        alt_bn128_Fq12 acc = one();
        for (long i = 255; i >= 0; --i) { acc = acc.squared(); if (e.test_bit(i)) acc *= *this; }
        return acc;
    }
    void print() const { std::cout << *this << "\n"; }
    // libff Fp12_2over3over2 / Fp6_3over2 operator<< / >>: c0 SEP c1 (SEP c2) recursively, i.e.
    // the twelve Fq coefficients in tower order separated by OUTPUT_SEPARATOR
    friend std::ostream &operator<<(std::ostream &os, const alt_bn128_Fq12 &a) {
        const lsa::Fq2 *c = reinterpret_cast<const lsa::Fq2 *>(&a.val());  // c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2
        for (int i = 0; i < 6; i++) os << alt_bn128_Fq2(c[i]) << (i < 5 ? LSA_OUTPUT_SEPARATOR : "");
        return os;
    }
    friend std::istream &operator>>(std::istream &is, alt_bn128_Fq12 &a) {
        lsa::Fq12 x;
        lsa::Fq2 *c = reinterpret_cast<lsa::Fq2 *>(&x);
        for (int i = 0; i < 6; i++) { alt_bn128_Fq2 t; is >> t; c[i] = t.v; }
        a = alt_bn128_Fq12(x);
        return is;
    }
};
using alt_bn128_GT = alt_bn128_Fq12;

// ---------------------------------------------------------------- groups
template <class F, int GROUP>
class G_shim {
public:
    F X, Y, Z;   // libff layout: Jacobian, Montgomery limbs
    using Jac = lsa::Jac<F>;
    using scalar_field = alt_bn128_Fr;     // (libff: typedef alt_bn128_Fr scalar_field)
    G_shim() { *this = zero(); }
    G_shim(const F &x, const F &y, const F &z) : X(x), Y(y), Z(z) {}
    G_shim(const Jac &j) : X(j.X), Y(j.Y), Z(j.Z) {}
    Jac jac() const { return Jac{X, Y, Z}; }
    static G_shim zero() { return G_shim(Jac::inf()); }
    static G_shim one() {
        if constexpr (GROUP == 1) {
            return G_shim(lsa::Fq::from_u32(1), lsa::Fq::from_u32(2), lsa::Fq::one());
        } else {
            return G_shim(lsa::fq2_const(LSA_G2_GEN_X), lsa::fq2_const(LSA_G2_GEN_Y), lsa::Fq2::one());
        }
    }
    static G_shim random_element() { return alt_bn128_Fr::random_element() * one(); }
    static size_t size_in_bits() { return GROUP == 1 ? 255 : 509; }
    // Fixed-base table of the generator (csrc/fixed_base.h: 32 signed 8-bit digits, at most 32 mixed additions, no
    // doubling), built once per process -- by whoever asks first, normally the helper thread init_public_params()
    // starts while the GPU comes up.  Read-only afterwards: safe under the OpenMP loops of a MULTICORE build.
    // LSA_SHIM_FIXED_BASE=0 keeps every product on the generic windowed ladder.
    static const lsa::FixedBaseTable<F> &generator_table() {
        static const lsa::FixedBaseTable<F> *tab = [] {
            auto *t = new lsa::FixedBaseTable<F>();
            t->build(one().jac());
            return t;
        }();
        return *tab;
    }
    static bool fixed_base_on() {
        static const bool on = !(getenv("LSA_SHIM_FIXED_BASE") && getenv("LSA_SHIM_FIXED_BASE")[0] == '0');
        return on;
    }
    bool is_generator() const {
        static const G_shim g = one();
        return memcmp((const void *)this, (const void *)&g, sizeof(G_shim)) == 0;
    }
    bool is_zero() const { return Z.is_zero(); }
    bool is_special() const { return is_zero() || Z == F::one(); }
    bool operator==(const G_shim &o) const { return lsa::jac_eq(jac(), o.jac()); }
    bool operator!=(const G_shim &o) const { return !(*this == o); }
    G_shim operator+(const G_shim &o) const { return G_shim(lsa::jac_add(jac(), o.jac())); }
    G_shim operator-() const { return G_shim(lsa::jac_neg(jac())); }
    G_shim operator-(const G_shim &o) const { return *this + (-o); }
    G_shim &operator+=(const G_shim &o) { *this = *this + o; return *this; }
    G_shim add(const G_shim &o) const { return *this + o; }
    G_shim mixed_add(const G_shim &o) const { return *this + o; }
    G_shim dbl() const { return G_shim(lsa::jac_dbl(jac())); }
    void to_affine_coordinates() { *this = G_shim(lsa::jac_normalize(jac())); }
    void to_special() { to_affine_coordinates(); }
    // libff is_well_formed(): the Jacobian curve equation Y^2 = X^3 + b Z^6 (infinity is fine)
    bool is_well_formed() const {
        if (is_zero()) return true;
        const F z2 = Z.sqr(), z6 = (z2.sqr()) * z2;
        return Y.sqr() == X.sqr() * X + curve_b() * z6;
    }
    void print() const { std::cout << *this << "\n"; }
    // libff "scalar * point" on the host (cold path: a verifier's handful of points; vectors of them go through
    // lsa_mtxmultiexp / lsa_scalar_mul_batch).  The same point as libff's double-and-add, by the cheapest route: plain
    // double-and-add for scalars below 2^16, the fixed-base table for the generator, a GLV (G1) / Galbraith-Scott (G2)
    // ladder for any other base (csrc/fixed_base.h, csrc/smul_host.h).
    // the 4-bit window ladder: correct for ANY point of the curve (the endomorphism ladders below need the order-r subgroup)
    static Jac window_ladder(const Jac &base, const uint64_t e[4], long top) {
        Jac res = Jac::inf();
        Jac tbl[16];
        tbl[0] = Jac::inf();
        tbl[1] = base;
        for (int i = 2; i < 16; i++) tbl[i] = (i & 1) ? lsa::jac_add(tbl[i - 1], base) : lsa::jac_dbl(tbl[i / 2]);
        for (long nib = top / 4; nib >= 0; --nib) {
            if (nib != top / 4) for (int j = 0; j < 4; j++) res = lsa::jac_dbl(res);
            const unsigned d = (unsigned)((e[nib / 16] >> (4 * (nib % 16))) & 15u);
            if (d) res = lsa::jac_add(res, tbl[d]);
        }
        return res;
    }
    // G2 is a proper subgroup of the twist's points (cofactor 2p - r), and psi / the cube root of unity act as scalars on it
    // ONLY: the Galbraith-Scott and GLV ladders return [k]P for P in G2, something else outside.  libff's double-and-add is
    // right for any point, and its operator>> checks neither the subgroup nor -- with compression -- more than solvability of
    // y^2 = x^3 + b'.  Every point the shim hands out itself lies in G2; a value cannot carry a tag (the 192-byte layout is
    // libff's), and testing every point that arrives as bytes would cost a key file of 2^20 points minutes.  So: once a
    // process has READ a G2 point, its host products on G2 bases other than the generator take the window ladder (275 us
    // instead of 76) for the rest of its life -- none of the reference's programs reads one; a verifier that does is correct.
    static std::atomic<bool> &external_point_seen() { static std::atomic<bool> f{false}; return f; }
    static void note_external_point(const G_shim &p) {
        if constexpr (GROUP == 2) { if (!p.is_zero()) external_point_seen().store(true, std::memory_order_relaxed); }
    }
    friend G_shim operator*(const alt_bn128_Fr &k, const G_shim &p) {
        lsa_shim::StatScope scope(lsa_shim::ST_SCALAR_MUL_HOST, 1);
        lsa_shim::SmulScope route(GROUP);
        const bigint<4> e = k.as_bigint();
        const Jac base = p.jac();
        Jac res = Jac::inf();
        long top = 255;
        while (top >= 0 && !e.test_bit(top)) --top;
        if (top < 16) {
            route.route = 1;
            for (long i = top; i >= 0; --i) {
                res = lsa::jac_dbl(res);
                if (e.test_bit(i)) res = lsa::jac_add(res, base);
            }
            return G_shim(res);
        }
        // the generator (commit.h:43-44,162-163, polytools.h:126-133, poly.h:117: most sites): table look-ups
        if (fixed_base_on() && p.is_generator()) {
            route.route = 0;
            return G_shim(generator_table().mul(e.data));
        }
        static const bool histogram_on = getenv("LSA_SHIM_BASE_HISTOGRAM") != nullptr;
        if (histogram_on) {                                           // diagnostic: which bases the generic ladder multiplies, how often
            static std::mutex hm;
            static std::vector<std::pair<std::string, unsigned>> hist;
            std::lock_guard<std::mutex> lk(hm);
            const std::string key((const char *)&p, 16);
            bool found = false;
            for (auto &h : hist) if (h.first == key) { h.second++; found = true; }
            if (!found) hist.emplace_back(key, 1u);
            static const int reg = atexit([] {
                std::vector<unsigned> c;
                for (auto &h : hist) c.push_back(h.second);
                std::sort(c.begin(), c.end(), std::greater<unsigned>());
                fprintf(stderr, "{\"lsa_shim_base_histogram\": {\"group\": %d, \"distinct\": %zu, \"top\": [", GROUP, c.size());
                for (size_t i = 0; i < c.size() && i < 12; i++) fprintf(stderr, "%s%u", i ? ", " : "", c[i]);
                fprintf(stderr, "]}}\n");
            });
            (void)reg;
        }
        // any other base (each is met once or twice: no table pays): GLV halves in width-5 NAF over one chain of 127
        // doublings (csrc/smul_host.h); LSA_SHIM_GLV=0 keeps the 4-bit window ladder below for an A/B
        static const bool glv_on = !(getenv("LSA_SHIM_GLV") && getenv("LSA_SHIM_GLV")[0] == '0');
        if constexpr (GROUP == 2) {
            // G2: the four-dimensional split over psi (66 doublings instead of 127); LSA_SHIM_GLS4=0: the two-dimensional one.
            // (Both use endomorphisms that act as scalars on the prime-order subgroup G2 only -- where every point libff
            // hands out lies; a point of the twist outside G2 must go through LSA_SHIM_GLV=0.)
            static const bool gls_on = !(getenv("LSA_SHIM_GLS4") && getenv("LSA_SHIM_GLS4")[0] == '0');
            if (external_point_seen().load(std::memory_order_relaxed)) return G_shim(window_ladder(base, e.data, top));   // (see note_external_point)
            if (glv_on && gls_on) return G_shim(lsa::gls4_mul_host(base, e.data));
        }
        if (glv_on) return G_shim(lsa::glv_mul_host(base, e.data));
        return G_shim(window_ladder(base, e.data, top));
    }
    // libff alt_bn128_G1 / alt_bn128_G2 operator<< / operator>> (alt_bn128_g1.cpp, alt_bn128_g2.cpp)
    // [upstream, recalled]: affine coordinates after to_affine_coordinates();
    //   "<is_zero> SEP X SEP <lsb of Y>"   default (point compression; G2 takes the lsb of Y.c0)
    //   "<is_zero> SEP X SEP Y"            with NO_PT_COMPRESSION
    // field elements in the Fp format above (the same BINARY_OUTPUT / MONTGOMERY_OUTPUT macros).
    using FieldIO = typename std::conditional<GROUP == 1, alt_bn128_Fq, alt_bn128_Fq2>::type;
    static unsigned y_parity(const F &y) {
        if constexpr (GROUP == 1) return (unsigned)(alt_bn128_Fq(y).as_bigint().data[0] & 1);
        else return (unsigned)(alt_bn128_Fq(y.c0).as_bigint().data[0] & 1);
    }
    friend std::ostream &operator<<(std::ostream &os, const G_shim &p) {
        G_shim a = p;
        a.to_affine_coordinates();
        os << (a.is_zero() ? 1 : 0) << LSA_OUTPUT_SEPARATOR;
#ifdef NO_PT_COMPRESSION
        os << FieldIO(a.X) << LSA_OUTPUT_SEPARATOR << FieldIO(a.Y);
#else
        os << FieldIO(a.X) << LSA_OUTPUT_SEPARATOR << y_parity(a.Y);
#endif
        return os;
    }
    friend std::istream &operator>>(std::istream &is, G_shim &p) {
        char is_zero;
        FieldIO tX, tY;
#ifdef NO_PT_COMPRESSION
        is >> is_zero >> tX >> tY;
        is_zero -= '0';
#else
#ifndef BINARY_OUTPUT
        is >> std::ws;                                 // vectors are streamed one element per line
#endif
        is.read(&is_zero, 1);
        is_zero -= '0';
        consume_OUTPUT_SEPARATOR(is);
        unsigned char y_lsb;
        is >> tX;
        consume_OUTPUT_SEPARATOR(is);
        is.read(reinterpret_cast<char *>(&y_lsb), 1);
        y_lsb -= '0';
        if (!is_zero) {                                // y = +/- sqrt(x^3 + b)
            FieldIO rhs = tX.squared() * tX + FieldIO(curve_b());
            if constexpr (GROUP == 1) { tY = rhs.sqrt(); if (tY.squared() != rhs) is.setstate(std::ios::failbit); }
            else { if (!rhs.sqrt(tY)) is.setstate(std::ios::failbit); }
            if (y_parity(tY.v) != y_lsb) tY = -tY;
        }
#endif
        if (!is_zero) p = G_shim(tX.v, tY.v, F::one());
        else p = zero();
        if (is) note_external_point(p);
        return is;
    }
    // curve coefficient: b = 3 (G1), b' = 3 / (9 + u) (G2, the sextic twist)
    static F curve_b() {
        if constexpr (GROUP == 1) return lsa::Fq::from_u32(3);
        else {
            lsa::Fq2 xi{lsa::Fq::from_u32(9), lsa::Fq::from_u32(1)};
            lsa::Fq2 three{lsa::Fq::from_u32(3), lsa::Fq::zero()};
            return three * xi.inverse();
        }
    }
};
using alt_bn128_G1 = G_shim<lsa::Fq, 1>;
using alt_bn128_G2 = G_shim<lsa::Fq2, 2>;
static_assert(sizeof(alt_bn128_G1) == 96 && sizeof(alt_bn128_G2) == 192, "libff point layout");
static_assert(sizeof(alt_bn128_Fr) == 32, "libff field layout");

// ---------------------------------------------------------------- pairing-friendly curve "pp"
// libff alt_bn128_ate_G1_precomp: the affine coordinates (to_affine_coordinates: O -> (0, 1, 0)).  Normalised here, on
// the host, once -- the Miller kernels then never invert.
struct alt_bn128_G1_precomp {
    alt_bn128_G1 P;                                       // Z == 1 or the point at infinity
    alt_bn128_Fq PX() const { return alt_bn128_Fq(P.is_zero() ? lsa::Fq::zero() : P.X); }
    alt_bn128_Fq PY() const { return alt_bn128_Fq(P.is_zero() ? lsa::Fq::one() : P.Y); }
    bool operator==(const alt_bn128_G1_precomp &o) const { return PX() == o.PX() && PY() == o.PY(); }
};
// libff alt_bn128_ate_ell_coeffs / alt_bn128_ate_G2_precomp: QX, QY and the coefficient triple of each of the 102
// steps of the ate loop.  The device keeps its own copy of the table per distinct Q (include/legosnark_amd.h, "G2
// precomputation"), so the host copy is only fetched when somebody asks for it (coeffs(), ==, <<).
struct alt_bn128_ate_ell_coeffs {
    alt_bn128_Fq2 ell_0, ell_VW, ell_VV;
    bool operator==(const alt_bn128_ate_ell_coeffs &o) const { return ell_0 == o.ell_0 && ell_VW == o.ell_VW && ell_VV == o.ell_VV; }
    friend std::ostream &operator<<(std::ostream &os, const alt_bn128_ate_ell_coeffs &c) {
        return os << c.ell_0 << LSA_OUTPUT_SEPARATOR << c.ell_VW << LSA_OUTPUT_SEPARATOR << c.ell_VV;
    }
    friend std::istream &operator>>(std::istream &is, alt_bn128_ate_ell_coeffs &c) { return is >> c.ell_0 >> c.ell_VW >> c.ell_VV; }
};
static_assert(sizeof(alt_bn128_ate_ell_coeffs) == 192, "libff coefficient triple");
struct alt_bn128_G2_precomp {
    alt_bn128_G2 Q;                                       // Z == 1 or the point at infinity
    mutable std::shared_ptr<std::vector<uint8_t>> blob;   // LSA_G2_PRECOMP_BYTES: QX, QY, coefficients (lazy)
    const std::vector<uint8_t> &bytes() const {
        lsa_shim::GpuLock lock;
        if (!blob) {
            blob = std::make_shared<std::vector<uint8_t>>(LSA_G2_PRECOMP_BYTES);
            lsa_require(lsa_g2_precompute(&Q, 1, blob->data()), "precompute_G2");
        }
        return *blob;
    }
    alt_bn128_Fq2 QX() const { return alt_bn128_Fq2(Q.is_zero() ? lsa::Fq2::zero() : Q.X); }
    alt_bn128_Fq2 QY() const { return alt_bn128_Fq2(Q.is_zero() ? lsa::Fq2::one() : Q.Y); }
    std::vector<alt_bn128_ate_ell_coeffs> coeffs() const {
        std::vector<alt_bn128_ate_ell_coeffs> c(LSA_ATE_NUM_COEFFS);
        memcpy((void *)c.data(), bytes().data() + 128, LSA_ATE_NUM_COEFFS * sizeof(alt_bn128_ate_ell_coeffs));
        return c;
    }
    bool operator==(const alt_bn128_G2_precomp &o) const { return bytes() == o.bytes(); }
    // libff operator<<: QX SEP QY NL, the number of coefficients NL, one triple per line
    friend std::ostream &operator<<(std::ostream &os, const alt_bn128_G2_precomp &p) {
        os << p.QX() << LSA_OUTPUT_SEPARATOR << p.QY() << "\n";
        const auto c = p.coeffs();
        os << c.size() << "\n";
        for (const auto &x : c) os << x << LSA_OUTPUT_NEWLINE;
        return os;
    }
    friend std::istream &operator>>(std::istream &is, alt_bn128_G2_precomp &p) {
        alt_bn128_Fq2 qx, qy;
        is >> qx;
        is >> qy;
        size_t n = 0;
        is >> n;
        if (n != (size_t)LSA_ATE_NUM_COEFFS) { is.setstate(std::ios::failbit); return is; }
        auto b = std::make_shared<std::vector<uint8_t>>(LSA_G2_PRECOMP_BYTES);
        memcpy(b->data(), &qx.v, 64);
        memcpy(b->data() + 64, &qy.v, 64);
        for (size_t i = 0; i < n; i++) {
            alt_bn128_ate_ell_coeffs c;
            is >> c;
            memcpy(b->data() + 128 + i * 192, (const void *)&c, 192);
        }
        const bool inf = qx == alt_bn128_Fq2::zero() && qy == alt_bn128_Fq2::one();      // libff's affine form of O is (0, 1)
        p.Q = inf ? alt_bn128_G2::zero() : alt_bn128_G2(qx.v, qy.v, lsa::Fq2::one());
        p.blob = b;
        return is;
    }
};

class alt_bn128_pp {
public:
    typedef alt_bn128_Fr Fp_type;
    typedef alt_bn128_G1 G1_type;
    typedef alt_bn128_G2 G2_type;
    typedef alt_bn128_G1_precomp G1_precomp_type;
    typedef alt_bn128_G2_precomp G2_precomp_type;
    typedef alt_bn128_Fq Fq_type;
    typedef alt_bn128_Fq2 Fqe_type;
    typedef alt_bn128_Fq12 Fqk_type;
    typedef alt_bn128_GT GT_type;
    static const bool has_affine_pairing = false;

    // one-time setup (src/examples/cplink.cc:81): brings the GPU library up.  LSA_DEVICE
    // selects the device (default 0).  Fails loudly without a gfx950 device.
    // The process's large vectors (2^20 scalars = 32 MiB, 2^20 points = 96 MiB: above glibc's mmap threshold, so each is
    // its own mapping, faulted in page by page and unmapped on destruction) are kept on the heap instead and the heap is
    // never trimmed: (1) the HIP runtime pins the pages of a large pageable copy, and unmapping pinned pages stalls the
    // GPU's queues for 12-25 ms (include/../csrc/capi_internal.h); (2) the reference's own loops allocate such vectors by
    // the dozen (src/gadgets/poly.h:57-86, src/gadgets/sumcheck.cc:41-66).  Unchanged hadamard 20: TOTAL Prove 1.70 ->
    // 1.56 s, Lipmaa prover 134 -> 103 ms.  LSA_SHIM_MALLOC=0 leaves the allocator alone.
    static void init_public_params() {
        const char *m = getenv("LSA_SHIM_MALLOC");
        if (!(m && m[0] == '0')) {
#if defined(__GLIBC__)
            mallopt(M_MMAP_THRESHOLD, 1 << 30);
            mallopt(M_TRIM_THRESHOLD, INT_MAX);
            mallopt(M_TOP_PAD, 64 << 20);
#endif
        }
        // the generators' fixed-base tables (a few ms of host work) while the GPU comes up
        static std::future<void> tables;
        if (alt_bn128_G1::fixed_base_on() && !tables.valid())
            tables = std::async(std::launch::async, [] { (void)alt_bn128_G1::generator_table(); (void)alt_bn128_G2::generator_table(); });
        const char *d = getenv("LSA_DEVICE");
        lsa_shim::GpuLock lock;
        lsa_require(lsa_init(d ? atoi(d) : 0), "init_public_params");
    }
    static alt_bn128_G1_precomp precompute_G1(const alt_bn128_G1 &P) {
        alt_bn128_G1 a = P;
        a.to_affine_coordinates();
        return {a};
    }
    // The device starts on the line table of a point it has not seen (asynchronously) and keeps it; the value only
    // carries Q, and its coefficients on demand.
    static alt_bn128_G2_precomp precompute_G2(const alt_bn128_G2 &Q) {
        lsa_shim::StatScope scope(lsa_shim::ST_G2_PRECOMP, 1);
        alt_bn128_G2 a = Q;
        a.to_affine_coordinates();
        lsa_shim::GpuLock lock;
        lsa_require(lsa_g2_tables_prefetch(&a, 1), "precompute_G2");
        return {a, nullptr};
    }
    static alt_bn128_Fq12 miller_term(const alt_bn128_G1_precomp &p, const alt_bn128_G2_precomp &q) {
        auto e = std::make_shared<lsa_shim::PairExpr>();
        e->terms.push_back(lsa_shim::PairTerm{p.P.jac(), q.Q.jac(), 0});
        return alt_bn128_Fq12(std::shared_ptr<const lsa_shim::PairExpr>(std::move(e)));
    }
    static alt_bn128_Fq12 miller_loop(const alt_bn128_G1_precomp &p, const alt_bn128_G2_precomp &q) { return miller_term(p, q); }
    static alt_bn128_Fq12 double_miller_loop(const alt_bn128_G1_precomp &p1, const alt_bn128_G2_precomp &q1,
                                             const alt_bn128_G1_precomp &p2, const alt_bn128_G2_precomp &q2) {
        return miller_term(p1, q1) * miller_term(p2, q2);
    }
    static alt_bn128_GT final_exponentiation(const alt_bn128_Fq12 &elt) { return elt.final_exponentiated(); }
    static alt_bn128_GT reduced_pairing(const alt_bn128_G1 &P, const alt_bn128_G2 &Q) {
        return final_exponentiation(miller_loop(precompute_G1(P), precompute_G2(Q)));
    }
    static alt_bn128_Fq12 pairing(const alt_bn128_G1 &P, const alt_bn128_G2 &Q) { return miller_loop(precompute_G1(P), precompute_G2(Q)); }
};

typedef alt_bn128_pp default_ec_pp;

template <typename EC_ppT> using Fr = typename EC_ppT::Fp_type;
template <typename EC_ppT> using G1 = typename EC_ppT::G1_type;
template <typename EC_ppT> using G2 = typename EC_ppT::G2_type;
template <typename EC_ppT> using G1_precomp = typename EC_ppT::G1_precomp_type;
template <typename EC_ppT> using G2_precomp = typename EC_ppT::G2_precomp_type;
template <typename EC_ppT> using Fq = typename EC_ppT::Fq_type;
template <typename EC_ppT> using Fqe = typename EC_ppT::Fqe_type;
template <typename EC_ppT> using Fqk = typename EC_ppT::Fqk_type;
template <typename EC_ppT> using GT = typename EC_ppT::GT_type;
template <typename EC_ppT> using Fr_vector = std::vector<Fr<EC_ppT>>;
template <typename EC_ppT> using G1_vector = std::vector<G1<EC_ppT>>;
template <typename EC_ppT> using G2_vector = std::vector<G2<EC_ppT>>;

// ---------------------------------------------------------------- multiexp.hpp
enum multi_exp_method {
    multi_exp_method_naive,
    multi_exp_method_naive_plain,
    multi_exp_method_bos_coster,
    multi_exp_method_BDLO12
};

namespace detail {
template <class T> struct group_id;
template <> struct group_id<alt_bn128_G1> { static const int value = 1; };
template <> struct group_id<alt_bn128_G2> { static const int value = 2; };

}  // namespace detail
// RAII: multi_exp / multiExpMA calls inside the scope are the SPMD form -- this rank's chunk in, the sum over all
// ranks out.  Every rank must issue the same sequence of such calls (each is a collective).
struct lsa_sharded_scope {
    lsa_sharded_scope() { depth()++; }
    ~lsa_sharded_scope() { depth()--; }
    lsa_sharded_scope(const lsa_sharded_scope &) = delete;
    lsa_sharded_scope &operator=(const lsa_sharded_scope &) = delete;
    static int &depth() { static thread_local int d = 0; return d; }
};
namespace detail {
template <class T, class FieldT>
T msm_forward(typename std::vector<T>::const_iterator vec_start, typename std::vector<T>::const_iterator vec_end,
              typename std::vector<FieldT>::const_iterator scalar_start, typename std::vector<FieldT>::const_iterator scalar_end,
              size_t chunks) {
    static_assert(std::is_same<FieldT, alt_bn128_Fr>::value, "scalars must be Fr");
    const size_t n = (size_t)(vec_end - vec_start);
    assert((size_t)(scalar_end - scalar_start) == n);
    (void)scalar_end;
    T out;
    lsa_shim::GpuLock lock;
    lsa_shim::StatScope scope(group_id<T>::value == 1 ? lsa_shim::ST_MSM_G1 : lsa_shim::ST_MSM_G2, n);
    const void *b = n ? (const void *)&*vec_start : nullptr;
    const void *s = n ? (const void *)&*scalar_start : nullptr;
    // SPMD provers (one process per GPU, lsa_comm_init done) inside an lsa_sharded_scope: every rank passes ITS
    // chunk of the vectors -- the contiguous range lsa_shard_range() gives it, i.e. libff's own `chunks` split with
    // chunks = world -- and receives the sum over all ranks (one RCCL all-gather of the Jacobian partials).
    // Outside such a scope a multi_exp is a whole sum on this GPU even when a communicator exists: the call site
    // cannot say whether its vectors are a chunk or everything (keygen, commitments over replicated inputs).
    if (lsa_sharded_scope::depth() > 0 && lsa_comm_world() > 1) {
        if (group_id<T>::value == 1) lsa_require(lsa_g1_msm_sharded(b, s, n, &out), "multi_exp<G1> (sharded)");
        else lsa_require(lsa_g2_msm_sharded(b, s, n, &out), "multi_exp<G2> (sharded)");
        return out;
    }
    if (group_id<T>::value == 1) lsa_require(lsa_g1_msm(b, s, n, chunks, &out), "multi_exp<G1>");
    else lsa_require(lsa_g2_msm(b, s, n, chunks, &out), "multi_exp<G2>");
    if (lsa_shim::Stats::get().on) {
        lsa_host_stats hs;
        if (lsa_msm_host_stats(&hs) == LSA_OK) {
            lsa_shim::Stats &st = lsa_shim::Stats::get();
            st.msm_us[0] += (uint64_t)(hs.h2d_scalars_ms * 1e3);
            st.msm_us[1] += (uint64_t)(hs.fingerprint_wait_ms * 1e3);
            st.msm_us[2] += (uint64_t)(hs.bases_prepare_ms * 1e3);
            st.msm_us[3] += (uint64_t)(hs.msm_ms * 1e3);
            st.msm_hits += (uint64_t)(hs.cache_hit != 0);
            st.msm_tables += (uint64_t)(hs.table != 0);
        }
    }
    return out;
}
}  // namespace detail

// Every method computes the same sum; on the GPU they all run the signed-digit bucket
// pipeline (bos_coster is only ever named by the never-called multiExp, globl.h:47-61).
template <typename T, typename FieldT, multi_exp_method Method>
T multi_exp(typename std::vector<T>::const_iterator vec_start, typename std::vector<T>::const_iterator vec_end,
            typename std::vector<FieldT>::const_iterator scalar_start, typename std::vector<FieldT>::const_iterator scalar_end,
            const size_t chunks) {
    return detail::msm_forward<T, FieldT>(vec_start, vec_end, scalar_start, scalar_end, chunks);
}
template <typename T, typename FieldT, multi_exp_method Method>
T multi_exp_with_mixed_addition(typename std::vector<T>::const_iterator vec_start, typename std::vector<T>::const_iterator vec_end,
                                typename std::vector<FieldT>::const_iterator scalar_start,
                                typename std::vector<FieldT>::const_iterator scalar_end, const size_t chunks) {
    return detail::msm_forward<T, FieldT>(vec_start, vec_end, scalar_start, scalar_end, chunks);
}

// ---- batched extensions (not libff API): operators whose reference implementation loops over
// many tiny libff calls on the CPU.  A maintainer swaps the loop body for one call.
//
// mtxmultiexp(out, exps, m) of /root/reference/src/gadgets/subspace.cc:18-25 for a column-major
// sparse matrix of G1 elements (ColG1 = vector<CoeffPos<G1>>, src/utils/matrix.h:35-45):
//   out[j] = sum over (val, pos) in m[j] of exps[pos] * val.
// Flattens the columns to CSC arrays and forwards to lsa_g1_sparse_matrix_msm (one batched
// scalar-multiplication kernel + one per-column sum instead of |m| sparsemexpG calls).
template <class Col, class FieldT>
void lsa_mtxmultiexp(std::vector<alt_bn128_G1> &out, const std::vector<FieldT> &exps, const std::vector<Col> &m) {
    std::vector<alt_bn128_G1> vals;
    std::vector<uint32_t> rows;
    std::vector<uint64_t> col_ptr(m.size() + 1, 0);
    size_t nnz = 0;
    for (const Col &c : m) nnz += c.size();
    vals.reserve(nnz);
    rows.reserve(nnz);
    for (size_t j = 0; j < m.size(); j++) {
        for (const auto &cp : m[j]) {
            vals.push_back(cp.val);
            rows.push_back((uint32_t)cp.pos);
        }
        col_ptr[j + 1] = vals.size();
    }
    out.assign(m.size(), alt_bn128_G1::zero());
    lsa_shim::GpuLock lock;
    lsa_shim::StatScope scope(lsa_shim::ST_SPARSE_MSM, nnz);
    lsa_require(lsa_g1_sparse_matrix_msm(vals.data(), rows.data(), col_ptr.data(), m.size(), exps.data(), exps.size(), out.data()),
                "mtxmultiexp");
}
// out[i] = scalars[i] * pts[i] (the loop of src/examples/cplink.cc:51-58 when the bases differ)
template <class FieldT>
std::vector<alt_bn128_G1> lsa_scalar_mul_batch(const std::vector<alt_bn128_G1> &pts, const std::vector<FieldT> &scalars) {
    const size_t n = pts.size() < scalars.size() ? pts.size() : scalars.size();
    std::vector<alt_bn128_G1> out(n, alt_bn128_G1::zero());
    lsa_shim::GpuLock lock;
    lsa_require(lsa_g1_scalar_mul_batch(pts.data(), scalars.data(), n, out.data(), 0), "scalar_mul_batch");
    return out;
}

// fixed-base tables.  libff: `template<typename T> using window_table = std::vector<std::vector<T>>` with
// powers_of_g[outer][inner] = inner * 2^(outer * window) * g (get_window_table).  The GPU builds its own table inside
// batch_exp, so get_window_table + batch_exp (which always travel together in the reference: src/utils/util.h:125-133,
// src/prototools/interp.h:45-58) only need the base point -- but a caller that INDEXES the table must find libff's
// entries there: the rows are materialised on first access (one lsa_g{1,2}_batch_exp over the scalars
// inner * 2^(outer * window), rows x columns and the short last row exactly as libff lays them out) and behave like
// the vector of vectors from then on.
template <typename T>
struct window_table {
    using row_type = std::vector<T>;
    using value_type = row_type;
    T base;
    size_t scalar_size = 0, window = 0;

    window_table() = default;
    window_table(const T &g, size_t bits, size_t w) : base(g), scalar_size(bits), window(w) {}
    size_t size() const { return window ? (scalar_size + window - 1) / window : 0; }
    bool empty() const { return size() == 0; }
    const row_type &operator[](size_t outer) const { return rows()[outer]; }
    const row_type &at(size_t outer) const { return rows().at(outer); }
    typename std::vector<row_type>::const_iterator begin() const { return rows().begin(); }
    typename std::vector<row_type>::const_iterator end() const { return rows().end(); }
    const std::vector<row_type> &rows() const {
        lsa_shim::GpuLock lock;
        if (!built_) {
            using FieldT = typename T::scalar_field;
            const size_t outerc = size(), in_window = size_t(1) << window;
            const size_t last_in_window = outerc ? size_t(1) << (scalar_size - (outerc - 1) * window) : 0;
            std::vector<FieldT> sc;
            sc.reserve(outerc * in_window);
            FieldT outer_step = FieldT::one();                       // 2^(outer * window)
            FieldT two_w = FieldT::one();
            for (size_t b = 0; b < window; b++) two_w = two_w + two_w;
            for (size_t outer = 0; outer < outerc; outer++) {
                const size_t cur = outer == outerc - 1 ? last_in_window : in_window;
                FieldT v = FieldT::zero();
                for (size_t inner = 0; inner < in_window; inner++) {
                    sc.push_back(inner < cur ? v : FieldT::zero());  // libff leaves the tail of the last row at T::zero()
                    v = v + outer_step;
                }
                outer_step = outer_step * two_w;
            }
            std::vector<T> flat(sc.size());
            if (!sc.empty()) {
                if (detail::group_id<T>::value == 1) lsa_require(lsa_g1_batch_exp(&base, sc.data(), sc.size(), flat.data(), 0), "get_window_table<G1>");
                else lsa_require(lsa_g2_batch_exp(&base, sc.data(), sc.size(), flat.data(), 0), "get_window_table<G2>");
            }
            rows_.assign(outerc, row_type());
            for (size_t outer = 0; outer < outerc; outer++) rows_[outer].assign(flat.begin() + outer * in_window, flat.begin() + (outer + 1) * in_window);
            built_ = true;
        }
        return rows_;
    }

  private:
    mutable std::vector<row_type> rows_;
    mutable bool built_ = false;
};
template <typename T>
size_t get_exp_window_size(const size_t num_scalars) { return num_scalars >= (size_t(1) << 16) ? 12 : 8; }
template <typename T>
window_table<T> get_window_table(const size_t scalar_size, const size_t window, const T &g) { return window_table<T>(g, scalar_size, window); }
template <typename T, typename FieldT>
std::vector<T> batch_exp(const size_t scalar_size, const size_t window, const window_table<T> &table, const std::vector<FieldT> &v) {
    (void)scalar_size; (void)window;
    std::vector<T> out(v.size());
    if (v.empty()) return out;
    lsa_shim::GpuLock lock;
    lsa_shim::StatScope scope(lsa_shim::ST_BATCH_EXP, v.size());
    if (detail::group_id<T>::value == 1) lsa_require(lsa_g1_batch_exp(&table.base, v.data(), v.size(), out.data(), 0), "batch_exp<G1>");
    else lsa_require(lsa_g2_batch_exp(&table.base, v.data(), v.size(), out.data(), 0), "batch_exp<G2>");
    return out;
}
template <typename T, typename FieldT>
T windowed_exp(const size_t scalar_size, const size_t window, const window_table<T> &table, const FieldT &pow) {
    return batch_exp<T, FieldT>(scalar_size, window, table, std::vector<FieldT>{pow})[0];
}
template <typename T>
void batch_to_special(std::vector<T> &vec) {
    if (vec.empty()) return;
    lsa_shim::GpuLock lock;
    lsa_shim::StatScope scope(lsa_shim::ST_NORMALIZE, vec.size());
    if (detail::group_id<T>::value == 1) lsa_require(lsa_g1_normalize(vec.data(), vec.size(), vec.data()), "batch_to_special");
    else lsa_require(lsa_g2_normalize(vec.data(), vec.size(), vec.data()), "batch_to_special");
}
template <typename T>
void batch_to_special_all_non_zeros(std::vector<T> &vec) { batch_to_special(vec); }

}  // namespace libff

// ate-pairing / xbyak backend of the reference's default CURVE=BN128 build
// (src/utils/matrix.h:31, src/gadgets/subspace.cc:13 say `using namespace bn;`)
namespace bn {}

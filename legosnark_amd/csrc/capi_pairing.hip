// legosnark_amd/csrc/capi_pairing.hip -- the pairing entry points of the C-ABI (include/legosnark_amd.h).
//
// Every entry point is one shape of the same job: n terms (P_i, Q_i), grouped into products, each
// product optionally followed by a final exponentiation.  Q_i arrives either as a point (libff
// precompute_G2 is then done here) or as libff's alt_bn128_ate_G2_precomp bytes (lsa_g2_precompute),
// the form the reference's keys hold and re-use in every verification
// (/root/reference/src/gadgets/subspace.cc:48,66-70,152-166; src/gadgets/lipmaa.h:95-96;
// src/gadgets/poly.h:97-121).  The device works on line tables (tmiller.h): a table is built once per
// distinct Q, kept resident in a cache keyed by a 128-bit fingerprint of the bytes the caller passed (the
// point, or the precomp blob), and every Miller loop over it runs the Fq12 chain only.  Products share
// accumulators (f <- f^2 * prod line_i) when there are more pairs than the chip has lanes for.
// No CPU arithmetic: the host only fingerprints, deduplicates and lays out index arrays.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <unordered_map>
#include <vector>

#include "capi_internal.h"
#include "tower.h"

namespace lsa {
// pairing.hip
size_t g2_table_words();
size_t g2_precomp_public_bytes();
int g2_precomp_device(const void *d_g2, size_t n, uint32_t *const *d_tabs, hipStream_t st);
int g2_table_export_device(const uint32_t *const *d_tabs, size_t n, void *d_public, hipStream_t st);
int g2_table_import_device(const void *d_public, size_t n, uint32_t *const *d_tabs, hipStream_t st);
int g2_table_identity_device(uint32_t *d_tab, hipStream_t st);
unsigned miller_tab_max_pairs();
void miller_split_release();
int miller_tab_device(const void *d_g1, const uint32_t *const *d_tabs, const uint8_t *d_flags, const uint32_t *d_acc_off, size_t nacc, unsigned M,
                      const uint32_t *d_ident, void *d_out, hipStream_t st);
int miller_fused_device(const void *d_g1, const void *d_g2, const uint8_t *d_flags, uint32_t *const *d_tabs, size_t n, void *d_out, hipStream_t st, bool gt_only);
}  // namespace lsa

using namespace lsa;

namespace {

StageBuf g_pair_p, g_pair_q, g_pair_f, g_pair_s, g_pair_o;     // points, Miller values, product scratch, results
StageBuf g_pair_meta, g_pair_scratch_tabs, g_pair_pub;         // index arrays, uncached tables, public-form staging

// pinned host staging for the small index arrays of a call (one H2D copy)
struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { (void)hipStreamSynchronize(g.stream); (void)hipHostFree(p); }
        p = nullptr; cap = 0;
        size_t want = bytes < 65536 ? 65536 : bytes + bytes / 4;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return -1; }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};
PinnedBuf g_pin_meta, g_pin_q;
// the pinned staging of a call must not be rewritten before its copies have run (an on_device call returns
// without synchronising): recorded after the uploads, waited for at the start of the next call
hipEvent_t g_uploaded = nullptr;
bool g_upload_pending = false;
int wait_uploads() {
    if (g_upload_pending) { HIPCHK(hipEventSynchronize(g_uploaded)); g_upload_pending = false; }
    return LSA_OK;
}
int mark_uploads() {
    if (!g_uploaded) HIPCHK(hipEventCreateWithFlags(&g_uploaded, hipEventDisableTiming));
    HIPCHK(hipEventRecord(g_uploaded, g.stream));
    g_upload_pending = true;
    return LSA_OK;
}

// ---------------------------------------------------------------- G2 line-table cache
struct Key128 {
    uint64_t a, b;
    bool operator==(const Key128 &o) const { return a == o.a && b == o.b; }
};
struct Key128Hash { size_t operator()(const Key128 &k) const { return (size_t)(k.a ^ (k.b * 0x9E3779B97F4A7C15ull)); } };

// 128-bit KEYED fingerprint (capi.hip: NH under a per-process random key): a verifier's results depend on which
// table a point resolves to, so the key of a table must not be something a prover can aim at -- for two distinct
// points (or blobs) the collision probability over the key is <= 2^-64 whatever their bytes are.
Key128 fingerprint(const void *bytes, size_t nbytes, uint64_t kind) {
    uint64_t h[2];
    keyed_hash128(bytes, nbytes, kind, h);
    return Key128{h[0], h[1]};
}

struct TableCache {
    static constexpr size_t BLOCK = 128;                  // tables per device allocation
    size_t max_tables = 4096;                             // 90 MB; LSA_G2_TABLES / lsa_g2_table_cache
    std::vector<void *> blocks;
    struct Slot { Key128 key; uint64_t tick; bool used; bool pending; };      // pending: promised by a prefetch, not built yet
    std::vector<Slot> slots;
    std::unordered_map<Key128, uint32_t, Key128Hash> map;
    uint64_t tick = 0, hits = 0, misses = 0, evictions = 0;
    uint32_t *ident = nullptr;                            // the table of a pair that is not there
    bool env_read = false;

    uint32_t *ptr(uint32_t slot) const { return (uint32_t *)((char *)blocks[slot / BLOCK] + (size_t)(slot % BLOCK) * g2_table_words() * 4); }
    void read_env() {
        if (env_read) return;
        env_read = true;
        if (const char *e = getenv("LSA_G2_TABLES")) max_tables = (size_t)strtoull(e, nullptr, 10);
    }
    int ensure_ident() {
        if (ident) return 0;
        if (hipMalloc((void **)&ident, g2_table_words() * 4) != hipSuccess) { ident = nullptr; return -1; }
        return g2_table_identity_device(ident, g.stream);
    }
    // slot of `key`, -1 on a miss
    long lookup(const Key128 &key) {
        auto it = map.find(key);
        if (it == map.end()) { misses++; return -1; }
        hits++;
        slots[it->second].tick = tick;
        return (long)it->second;
    }
    // a free slot for `key` (a released one, a new one, or the least recently used one that neither a term of the
    // current call nor an unbuilt promise of an earlier prefetch refers to), -1 if none
    long insert(const Key128 &key, bool pending = false) {
        long s = -1;
        if (!free_slots.empty()) {
            s = (long)free_slots.back();
            free_slots.pop_back();
            slots[(size_t)s] = Slot{key, tick, true, pending};
        } else if (slots.size() < max_tables) {
            if (slots.size() == blocks.size() * BLOCK) {
                void *b = nullptr;
                if (hipMalloc(&b, BLOCK * g2_table_words() * 4) != hipSuccess) { (void)hipGetLastError(); return -1; }
                blocks.push_back(b);
            }
            slots.push_back(Slot{key, tick, true, pending});
            s = (long)slots.size() - 1;
        } else {
            uint64_t best = tick;
            for (size_t i = 0; i < slots.size(); i++)
                if (slots[i].used && !slots[i].pending && slots[i].tick < best) { best = slots[i].tick; s = (long)i; }
            if (s < 0) return -1;
            map.erase(slots[(size_t)s].key);
            evictions++;
            slots[(size_t)s] = Slot{key, tick, true, pending};
        }
        map[key] = (uint32_t)s;
        return s;
    }
    // forgets `key` (a table that was promised or begun and never finished: a later lookup must miss, not read it)
    void erase(const Key128 &key) {
        auto it = map.find(key);
        if (it == map.end()) return;
        const uint32_t s = it->second;
        map.erase(it);
        slots[s].used = false; slots[s].pending = false; slots[s].tick = 0;
        free_slots.push_back(s);
    }
    void built(const Key128 &key) {
        auto it = map.find(key);
        if (it != map.end()) slots[it->second].pending = false;
    }
    std::vector<uint32_t> free_slots;
    void clear() {
        for (void *b : blocks) (void)hipFree(b);
        blocks.clear(); slots.clear(); map.clear(); free_slots.clear();
        if (ident) (void)hipFree(ident);
        ident = nullptr;
    }
} g_tabs;

// Tables lsa_g2_tables_prefetch has promised (their slots are taken, their keys resolve) but not yet built: built
// when a Miller call arrives or when `batch()` of them are waiting, whichever is first -- always before anything on
// lsa_stream() can read them.  One wavefront of k_g2_precomp holds five points and takes as long (0.65 ms) for five as
// for one; further wavefronts run beside it.  A verifier that derives twenty G2 points one after the other
// (CPPoly::verify, src/gadgets/poly.h:115-118) spends 0.45 ms of host time on them since the fixed-base tables of round 5,
// so ONE launch of four workgroups when its Miller call arrives (0.65 ms) beats four launches of one queued behind each
// other (2.6 ms: the round-4 setting, LSA_G2_PREFETCH_BATCH=5, which paid when a product took 0.27 ms of host time).
struct PendingTables {
    std::vector<Jac<Fq2>> pts;
    std::vector<uint64_t> dst;
    std::vector<Key128> keys;
    StageBuf dev;
    static size_t batch() {
        static const size_t b = [] {
            const char *e = getenv("LSA_G2_PREFETCH_BATCH");
            const long v = e ? atol(e) : 60;
            return (size_t)(v < 1 ? 1 : v > 1024 ? 1024 : v);
        }();
        return b;
    }
    int build(const std::vector<char> &h, size_t m) {
        if (dev.ensure(h.size())) { set_error("g2_tables_prefetch: staging allocation failed"); return LSA_ERR_NOMEM; }
        LSA_UPLOAD(dev.p, h.data(), h.size());
        return g2_precomp_device(dev.p, m, (uint32_t *const *)((const char *)dev.p + m * sizeof(Jac<Fq2>)), g.stream);
    }
    int flush() {
        const size_t m = pts.size();
        if (m == 0) return LSA_OK;
        std::vector<char> h(m * (sizeof(Jac<Fq2>) + 8));
        memcpy(h.data(), pts.data(), m * sizeof(Jac<Fq2>));
        memcpy(h.data() + m * sizeof(Jac<Fq2>), dst.data(), m * 8);
        const std::vector<Key128> ks = keys;
        pts.clear();
        dst.clear();
        keys.clear();
        const int rc = build(h, m);
        // a promise that could not be kept is withdrawn: its key must miss from now on, not resolve to an unwritten table
        for (const Key128 &k : ks) { if (rc) g_tabs.erase(k); else g_tabs.built(k); }
        return rc;
    }
    void drop() { pts.clear(); dst.clear(); keys.clear(); }
} g_pending;

// ---------------------------------------------------------------- one job description
struct Terms {
    const void *g1 = nullptr;              // n Jacobian G1 points
    const void *g2 = nullptr;              // n Jacobian G2 points, or null
    const void *const *qpre = nullptr;     // n pointers to LSA_G2_PRECOMP_BYTES blobs (a null entry: use g2[i]), or null
    const uint8_t *flags = nullptr;        // bit 0: conjugate the term (libff unitary_inverse of its Miller value)
    const uint64_t *seg = nullptr;         // nseg + 1 offsets; null: every term is its own product
    size_t n = 0, nseg = 0;
    bool on_device = false;                // g1 / g2 are device pointers (no cache: nothing to fingerprint on the host)
    bool gt_only = false;                  // every value of this job goes through a final exponentiation before it leaves: Miller
                                           // values may differ from libff's by factors the final exponent kills (signed-digit loop)
};

unsigned g_force_m = 0;                    // lsa_pairing_set_chunk: pairs per accumulator, 0 = by batch size
inline int force_kernel() {
    static const int f = getenv("LSA_MILLER_KERNEL") ? atoi(getenv("LSA_MILLER_KERNEL")) : 0;   // 3: one lane per pairing (miller.h), 5: tables, 6: fused
    return f;
}

// The job through the fused kernel (pairing.hip, k_miller_fused: G2 arithmetic and Fq12 chain side by side in one
// workgroup, one Miller value per term), then the products.  emit: per term, the device table to fill on the way
// (0: none) -- the tables of points seen for the first time go into the cache while their first loop runs.
int run_fused(const Terms &t, const std::vector<uint64_t> &emit, void **d_res) {
    const size_t n = t.n, nprod = t.seg ? t.nseg : n;
    int rc = wait_uploads();
    if (rc) return rc;
    const size_t off_seg = 0, off_emit = off_seg + (nprod + 1) * 8, off_flag = off_emit + n * 8, meta_bytes = off_flag + n + 8;
    if (g_pin_meta.ensure(meta_bytes) || g_pair_meta.ensure(meta_bytes) || g_pair_f.ensure(std::max<size_t>(n, 1) * fq12_bytes()) ||
        g_pair_s.ensure(((n + 7) / 8 + nprod + 1) * fq12_bytes())) {
        set_error("pairing: staging allocation failed");
        return LSA_ERR_NOMEM;
    }
    char *hm = (char *)g_pin_meta.p;
    uint64_t *h_seg = (uint64_t *)(hm + off_seg), *h_emit = (uint64_t *)(hm + off_emit);
    uint8_t *h_flag = (uint8_t *)(hm + off_flag);
    for (size_t j = 0; j <= nprod; j++) h_seg[j] = t.seg ? t.seg[j] : j;
    for (size_t i = 0; i < n; i++) { h_emit[i] = emit.empty() ? 0 : emit[i]; h_flag[i] = t.flags ? (uint8_t)(t.flags[i] & 1) : (uint8_t)0; }
    const void *d_p = t.g1, *d_q = t.g2;
    if (!t.on_device) {
        if (g_pair_p.ensure(std::max<size_t>(n, 1) * sizeof(Jac<Fq>)) || g_pair_q.ensure(std::max<size_t>(n, 1) * sizeof(Jac<Fq2>))) { set_error("pairing: hipMalloc failed"); return LSA_ERR_NOMEM; }
        if (n) {
            LSA_UPLOAD(g_pair_p.p, t.g1, n * sizeof(Jac<Fq>));
            LSA_UPLOAD(g_pair_q.p, t.g2, n * sizeof(Jac<Fq2>));
        }
        d_p = g_pair_p.p; d_q = g_pair_q.p;
    }
    HIPCHK(hipMemcpyAsync(g_pair_meta.p, hm, meta_bytes, hipMemcpyHostToDevice, g.stream));
    rc = mark_uploads();
    if (rc) return rc;
    const char *dm = (const char *)g_pair_meta.p;
    rc = miller_fused_device(d_p, d_q, (const uint8_t *)(dm + off_flag), emit.empty() ? nullptr : (uint32_t *const *)(dm + off_emit), n, g_pair_f.p, g.stream, t.gt_only);
    if (rc) return rc;
    if (!t.seg) { *d_res = g_pair_f.p; return LSA_OK; }
    if (nprod == 1 && n > 0) return fq12_product_device(g_pair_f.p, g_pair_s.p, n, d_res, g.stream);
    rc = fq12_segment_products_device(g_pair_f.p, (const uint64_t *)(dm + off_seg), nprod, g_pair_s.p, g.stream);
    *d_res = g_pair_s.p;
    return rc;
}

// Miller products of the job on the device: *d_res points at nseg Fq12 values when this returns (stream-ordered)
int run_miller(const Terms &t, void **d_res) {
    const size_t n = t.n, nseg = t.nseg;
    g_tabs.read_env();
    // ---- points the cache does not take (more than 1024 terms, device-resident, cache off): the fused kernel -- G2
    // arithmetic and Fq12 chain side by side, no table through memory.  (LSA_MILLER_KERNEL = 6 forces it, 5 the tables.)
    const bool no_cache = t.on_device || t.n > 1024 || g_tabs.max_tables == 0;
    if (!t.qpre && t.n && g_force_m == 0 && ((force_kernel() == 0 && no_cache) || force_kernel() == 6)) return run_fused(t, {}, d_res);
    // ---- the one-lane-per-pairing kernel of miller.h (the family's fallback): only when forced
    const bool plain = !t.qpre && !t.flags;
    if (plain && force_kernel() == 3) {
        const void *d_p = t.g1, *d_q = t.g2;
        if (!t.on_device) {
            if (g_pair_p.ensure(n * sizeof(Jac<Fq>)) || g_pair_q.ensure(n * sizeof(Jac<Fq2>))) { set_error("pairing: hipMalloc failed"); return LSA_ERR_NOMEM; }
            LSA_UPLOAD(g_pair_p.p, t.g1, n * sizeof(Jac<Fq>));
            LSA_UPLOAD(g_pair_q.p, t.g2, n * sizeof(Jac<Fq2>));
            d_p = g_pair_p.p; d_q = g_pair_q.p;
        }
        if (g_pair_f.ensure((n ? n : 1) * fq12_bytes())) { set_error("pairing: hipMalloc failed"); return LSA_ERR_NOMEM; }
        int rc = miller_device(d_p, d_q, n, g_pair_f.p, g.stream);
        if (rc) return rc;
        if (!t.seg) { *d_res = g_pair_f.p; return LSA_OK; }
        if (nseg == 1) {
            if (g_pair_s.ensure(((n + 7) / 8 + 1) * fq12_bytes())) { set_error("pairing: hipMalloc failed"); return LSA_ERR_NOMEM; }
            if (n == 0) { Fq12 one = Fq12::one(); HIPCHK(hipMemcpyAsync(g_pair_s.p, &one, sizeof one, hipMemcpyHostToDevice, g.stream)); HIPCHK(hipStreamSynchronize(g.stream)); *d_res = g_pair_s.p; return LSA_OK; }
            return fq12_product_device(g_pair_f.p, g_pair_s.p, n, d_res, g.stream);
        }
        if (g_pair_s.ensure(nseg * fq12_bytes()) || g_pair_meta.ensure((nseg + 1) * sizeof(uint64_t))) { set_error("pairing: hipMalloc failed"); return LSA_ERR_NOMEM; }
        LSA_UPLOAD(g_pair_meta.p, t.seg, (nseg + 1) * sizeof(uint64_t));
        HIPCHK(hipStreamSynchronize(g.stream));     // t.seg is pageable caller memory
        rc = fq12_segment_products_device(g_pair_f.p, (const uint64_t *)g_pair_meta.p, nseg, g_pair_s.p, g.stream);
        *d_res = g_pair_s.p;
        return rc;
    }

    // ---- tables
    { int rcw = wait_uploads(); if (rcw) return rcw; }
    { int rcp = g_pending.flush(); if (rcp) return rcp; }       // promised tables: on the stream before anything reads them
    if (g_tabs.ensure_ident()) { set_error("pairing: hipMalloc failed"); return LSA_ERR_NOMEM; }
    g_tabs.tick++;
    const size_t TW = g2_table_words(), PUB = g2_precomp_public_bytes();
    // accumulators: a product of len pairs takes ceil(len / M) of them; M grows once the chip is full
    // (four accumulators per wavefront, 1024 SIMDs)
    unsigned M = 1;
    while (M < miller_tab_max_pairs() && n > (size_t)M * 8192) M++;
    if (g_force_m) M = g_force_m;
    std::vector<uint64_t> own_seg;
    const uint64_t *seg = t.seg;
    size_t nprod = nseg;
    if (!seg) {
        own_seg.resize(n + 1);
        for (size_t i = 0; i <= n; i++) own_seg[i] = i;
        seg = own_seg.data();
        nprod = n;
    }
    size_t nacc = 0;
    for (size_t j = 0; j < nprod; j++) { const size_t len = (size_t)(seg[j + 1] - seg[j]); nacc += len ? (len + M - 1) / M : 1; }
    // layout of the index arrays (one pinned staging buffer, one copy): table pointers, accumulator offsets,
    // product offsets (in accumulators), flags
    // (a handful of host G1 points ride in the same pinned buffer: one copy command for the whole call)
    const bool g1_in_meta = !t.on_device && n > 0 && n <= 256;
    const size_t off_tab = 0, off_acc = off_tab + n * 8, off_prod = (off_acc + (nacc + 1) * 4 + 7) & ~(size_t)7, off_flag = off_prod + (nprod + 1) * 8,
                 off_g1 = (off_flag + n + 15) & ~(size_t)15, meta_bytes = off_g1 + (g1_in_meta ? n * sizeof(Jac<Fq>) : 0) + 8;
    if (g_pin_meta.ensure(meta_bytes) || g_pair_meta.ensure(meta_bytes)) { set_error("pairing: staging allocation failed"); return LSA_ERR_NOMEM; }
    char *hm = (char *)g_pin_meta.p;
    uint64_t *h_tab = (uint64_t *)(hm + off_tab);
    uint32_t *h_acc = (uint32_t *)(hm + off_acc);
    uint64_t *h_prod = (uint64_t *)(hm + off_prod);
    uint8_t *h_flag = (uint8_t *)(hm + off_flag);
    {
        size_t a = 0;
        for (size_t j = 0; j < nprod; j++) {
            h_prod[j] = a;
            const size_t lo = (size_t)seg[j], hi = (size_t)seg[j + 1];
            if (lo == hi) { h_acc[a++] = (uint32_t)lo; continue; }       // an empty product: one accumulator without pairs = 1
            for (size_t s = lo; s < hi; s += M) h_acc[a++] = (uint32_t)s;
        }
        h_prod[nprod] = a;
        h_acc[a] = (uint32_t)n;
    }
    // accumulator a covers [h_acc[a], min(h_acc[a+1], end of its product)): make that literally h_acc[a+1] by
    // construction -- the next accumulator starts where this one ends, except after an empty product
    // (same start) -- so nothing else is needed.
    for (size_t i = 0; i < n; i++) h_flag[i] = t.flags ? (uint8_t)(t.flags[i] & 1) : (uint8_t)0;

    // ---- resolve a device table for every term
    const bool use_cache = !t.on_device && g_tabs.max_tables > 0 && n <= 1024;
    std::vector<uint32_t> need_pre;        // terms whose table has to be computed from the point
    std::vector<uint32_t> need_imp;        // terms whose table has to be imported from a precomp blob
    size_t scratch_used = 0;
    std::unordered_map<Key128, uint64_t, Key128Hash> seen;   // within this call
    std::unordered_map<const void *, uint64_t> seen_blob;   // cache bypassed: equal blob POINTERS still share one table
    auto scratch_slot = [&](size_t k) { return (uint64_t)(uintptr_t)((uint32_t *)g_pair_scratch_tabs.p + k * TW); };
    // every term is validated BEFORE the cache is touched: a call that fails half-way must not leave keys behind
    for (size_t i = 0; i < n; i++)
        if (!(t.qpre && t.qpre[i]) && !t.g2) { set_error("pairing: term %zu has neither a point nor a precomputed table", i); return LSA_ERR_INVALID; }
    // uncached terms take scratch tables: count them first (the scratch buffer must not move afterwards).  With the
    // cache: at worst all n (the cache full of this call's own tables, n <= 1024); without: one per point term and one
    // per DISTINCT blob (verifyLin3or4 scaled up: 2^16 terms over a handful of key blobs are a handful of tables)
    size_t scratch_need = n;
    if (!use_cache && t.qpre) {
        std::unordered_map<const void *, uint64_t> distinct;
        scratch_need = 0;
        for (size_t i = 0; i < n; i++) {
            if (!t.qpre[i]) scratch_need++;
            else if (distinct.emplace(t.qpre[i], 0).second) scratch_need++;
        }
    }
    if (g_pair_scratch_tabs.ensure(std::max<size_t>(scratch_need, 1) * TW * 4)) { set_error("pairing: table scratch allocation failed"); return LSA_ERR_NOMEM; }
    // keys this call inserts: withdrawn again on ANY error return below (their tables would be unwritten or partial, and
    // the next call with the same Q would run its Miller loop over them)
    struct InsertGuard {
        std::vector<Key128> keys;
        bool ok = false;
        ~InsertGuard() { if (!ok) for (const Key128 &k : keys) g_tabs.erase(k); }
    } guard;
    for (size_t i = 0; i < n; i++) {
        const void *blob = t.qpre ? t.qpre[i] : nullptr;
        uint64_t dev = 0;
        if (use_cache) {
            const Key128 key = blob ? fingerprint(blob, PUB, 2) : fingerprint((const char *)t.g2 + i * sizeof(Jac<Fq2>), sizeof(Jac<Fq2>), 1);
            auto it = seen.find(key);
            if (it != seen.end()) dev = it->second;
            else {
                long s = g_tabs.lookup(key);
                if (s >= 0) dev = (uint64_t)(uintptr_t)g_tabs.ptr((uint32_t)s);
                else {
                    s = g_tabs.insert(key);
                    if (s >= 0) guard.keys.push_back(key);
                    dev = s >= 0 ? (uint64_t)(uintptr_t)g_tabs.ptr((uint32_t)s) : scratch_slot(scratch_used++);
                    (blob ? need_imp : need_pre).push_back((uint32_t)i);
                }
                seen.emplace(key, dev);
            }
        } else if (blob) {
            auto it = seen_blob.find(blob);
            if (it != seen_blob.end()) dev = it->second;
            else {
                dev = scratch_slot(scratch_used++);
                seen_blob.emplace(blob, dev);
                need_imp.push_back((uint32_t)i);
            }
        } else {
            dev = scratch_slot(scratch_used++);
            need_pre.push_back((uint32_t)i);
        }
        h_tab[i] = dev;
    }

    // ---- points seen for the first time and no precomputed tables among the terms: the fused kernel computes every
    // term (while the chip is not full a resident Q gains nothing from its table: the call waits for the new ones
    // anyway) and fills the new tables on the way
    if (use_cache && !t.qpre && !need_pre.empty() && g_force_m == 0 && force_kernel() != 5) {
        std::vector<uint64_t> emit(n, 0);
        for (uint32_t i : need_pre) emit[i] = h_tab[i];
        const int rcf = run_fused(t, emit, d_res);
        guard.ok = rcf == LSA_OK;
        return rcf;
    }

    // ---- uploads
    const void *d_g1 = t.g1;
    if (g1_in_meta) {
        memcpy(hm + off_g1, t.g1, n * sizeof(Jac<Fq>));
        d_g1 = (const char *)g_pair_meta.p + off_g1;
    } else if (!t.on_device) {
        if (g_pair_p.ensure(std::max<size_t>(n, 1) * sizeof(Jac<Fq>))) { set_error("pairing: hipMalloc failed"); return LSA_ERR_NOMEM; }
        if (n) LSA_UPLOAD(g_pair_p.p, t.g1, n * sizeof(Jac<Fq>));
        d_g1 = g_pair_p.p;
    }
    HIPCHK(hipMemcpyAsync(g_pair_meta.p, hm, meta_bytes, hipMemcpyHostToDevice, g.stream));
    const char *dm = (const char *)g_pair_meta.p;

    // ---- missing tables
    if (!need_pre.empty()) {
        const size_t m = need_pre.size();
        const void *d_q = nullptr;
        const uint32_t *const *d_tp = nullptr;
        // the points and the destination pointers of the misses, gathered into pinned staging
        if (g_pin_q.ensure(m * (sizeof(Jac<Fq2>) + 8)) || g_pair_q.ensure(m * (sizeof(Jac<Fq2>) + 8))) { set_error("pairing: staging allocation failed"); return LSA_ERR_NOMEM; }
        char *hq = (char *)g_pin_q.p;
        uint64_t *hp = (uint64_t *)(hq + m * sizeof(Jac<Fq2>));
        if (t.on_device && m == n) {
            d_q = t.g2;                                             // all of them, in place
            d_tp = (const uint32_t *const *)(dm + off_tab);
        } else {
            if (t.on_device) { set_error("pairing: internal (device points need tables for all terms)"); return LSA_ERR_INVALID; }
            for (size_t k = 0; k < m; k++) {
                memcpy(hq + k * sizeof(Jac<Fq2>), (const char *)t.g2 + (size_t)need_pre[k] * sizeof(Jac<Fq2>), sizeof(Jac<Fq2>));
                hp[k] = h_tab[need_pre[k]];
            }
            HIPCHK(hipMemcpyAsync(g_pair_q.p, hq, m * (sizeof(Jac<Fq2>) + 8), hipMemcpyHostToDevice, g.stream));
            d_q = g_pair_q.p;
            d_tp = (const uint32_t *const *)((const char *)g_pair_q.p + m * sizeof(Jac<Fq2>));
        }
        int rc = g2_precomp_device(d_q, m, (uint32_t *const *)d_tp, g.stream);
        if (rc) return rc;
    }
    if (!need_imp.empty()) {
        const size_t m = need_imp.size();
        if (g_pair_pub.ensure(m * (PUB + 8))) { set_error("pairing: staging allocation failed"); return LSA_ERR_NOMEM; }
        std::vector<uint64_t> hp(m);
        for (size_t k = 0; k < m; k++) {
            LSA_UPLOAD((char *)g_pair_pub.p + k * PUB, t.qpre[need_imp[k]], PUB);
            hp[k] = h_tab[need_imp[k]];
        }
        HIPCHK(hipMemcpyAsync((char *)g_pair_pub.p + m * PUB, hp.data(), m * 8, hipMemcpyHostToDevice, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));     // hp is a local
        int rc = g2_table_import_device(g_pair_pub.p, m, (uint32_t *const *)((char *)g_pair_pub.p + m * PUB), g.stream);
        if (rc) return rc;
    }

    { int rcm = mark_uploads(); if (rcm) return rcm; }
    // ---- Miller loops, one Fq12 per accumulator
    if (g_pair_f.ensure(std::max<size_t>(nacc, 1) * fq12_bytes())) { set_error("pairing: hipMalloc failed"); return LSA_ERR_NOMEM; }
    int rc = miller_tab_device(d_g1, (const uint32_t *const *)(dm + off_tab), (const uint8_t *)(dm + off_flag), (const uint32_t *)(dm + off_acc), nacc, M,
                               g_tabs.ident, g_pair_f.p, g.stream);
    if (rc) return rc;
    guard.ok = true;           // every new table is on the stream, ahead of anything that reads it
    // ---- products over the accumulators of each product
    if (nacc == nprod) { *d_res = g_pair_f.p; return LSA_OK; }
    if (nprod == 1) {
        if (g_pair_s.ensure(((nacc + 7) / 8 + 1) * fq12_bytes())) { set_error("pairing: hipMalloc failed"); return LSA_ERR_NOMEM; }
        return fq12_product_device(g_pair_f.p, g_pair_s.p, nacc, d_res, g.stream);
    }
    if (g_pair_s.ensure(nprod * fq12_bytes())) { set_error("pairing: hipMalloc failed"); return LSA_ERR_NOMEM; }
    rc = fq12_segment_products_device(g_pair_f.p, (const uint64_t *)(dm + off_prod), nprod, g_pair_s.p, g.stream);
    *d_res = g_pair_s.p;
    return rc;
}

// the job with host results: out = nseg (or n) Fq12 values
int run_terms_host(const Terms &t_in, void *out, bool final_exp) {
    LSA_TRACE_CALL("pairing_terms", t_in.n);
    Terms t = t_in;
    t.gt_only = final_exp;
    int rc = require_ready();
    if (rc) return rc;
    const size_t nres = t.seg ? t.nseg : t.n;
    if (nres == 0) return LSA_OK;
    if (!out || (t.n && !t.g1)) { set_error("pairing: null argument"); return LSA_ERR_INVALID; }
    void *res = nullptr;
    rc = run_miller(t, &res);
    if (rc) return rc;
    if (final_exp && nres == 1) {
        // one check = one value: the final exponentiation writes it straight into pinned host memory (no copy command
        // behind the kernel: ~15 us of every blocking check of a verifier written call by call)
        rc = final_exp_device(res, 1, g.h_result, g.stream);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(g.stream));
        memcpy(out, g.h_result, fq12_bytes());
        return LSA_OK;
    }
    if (final_exp) {
        if (g_pair_o.ensure(nres * fq12_bytes())) { set_error("pairing: hipMalloc failed"); return LSA_ERR_NOMEM; }
        rc = final_exp_device(res, nres, g_pair_o.p, g.stream);
        if (rc) return rc;
        res = g_pair_o.p;
    }
    LSA_DOWNLOAD(out, res, nres * fq12_bytes());
    HIPCHK(hipStreamSynchronize(g.stream));
    return LSA_OK;
}

int check_segments(const uint64_t *seg, size_t nseg, const char *who) {
    if (!seg) { set_error("%s: null argument", who); return LSA_ERR_INVALID; }
    if (seg[0] != 0) { set_error("%s: offsets must start at 0", who); return LSA_ERR_INVALID; }
    for (size_t j = 0; j < nseg; j++)
        if (seg[j + 1] < seg[j]) { set_error("%s: offsets must not decrease", who); return LSA_ERR_INVALID; }
    return LSA_OK;
}

}  // namespace

namespace lsa {
void pairing_release() {
    g_pair_p.release(); g_pair_q.release(); g_pair_f.release(); g_pair_s.release(); g_pair_o.release();
    g_pair_meta.release(); g_pair_scratch_tabs.release(); g_pair_pub.release();
    g_pin_meta.release(); g_pin_q.release();
    if (g_uploaded) { (void)hipEventDestroy(g_uploaded); g_uploaded = nullptr; }
    g_upload_pending = false;
    g_pending.drop(); g_pending.dev.release();
    miller_split_release();
    g_tabs.clear();
}
}  // namespace lsa

extern "C" {

size_t lsa_g2_precomp_bytes(void) { return g2_precomp_public_bytes(); }

int lsa_g2_precompute(const void *g2_jac, size_t n, void *out_precomp) {
    LSA_TRACE_CALL("g2_precompute", n);
    int rc = require_ready();
    if (rc) return rc;
    if (n == 0) return LSA_OK;
    if (!g2_jac || !out_precomp) { set_error("g2_precompute: null argument"); return LSA_ERR_INVALID; }
    const size_t TW = g2_table_words(), PUB = g2_precomp_public_bytes();
    if (g_pair_q.ensure(n * sizeof(Jac<Fq2>)) || g_pair_scratch_tabs.ensure(n * TW * 4) || g_pair_pub.ensure(n * PUB) || g_pin_meta.ensure(n * 8) ||
        g_pair_meta.ensure(n * 8)) {
        set_error("g2_precompute: staging allocation failed");
        return LSA_ERR_NOMEM;
    }
    rc = wait_uploads();
    if (rc) return rc;
    uint64_t *hp = (uint64_t *)g_pin_meta.p;
    for (size_t i = 0; i < n; i++) hp[i] = (uint64_t)(uintptr_t)((uint32_t *)g_pair_scratch_tabs.p + i * TW);
    HIPCHK(hipMemcpyAsync(g_pair_meta.p, hp, n * 8, hipMemcpyHostToDevice, g.stream));
    LSA_UPLOAD(g_pair_q.p, g2_jac, n * sizeof(Jac<Fq2>));
    rc = g2_precomp_device(g_pair_q.p, n, (uint32_t *const *)g_pair_meta.p, g.stream);
    if (rc) return rc;
    rc = g2_table_export_device((const uint32_t *const *)g_pair_meta.p, n, g_pair_pub.p, g.stream);
    if (rc) return rc;
    LSA_DOWNLOAD(out_precomp, g_pair_pub.p, n * PUB);
    HIPCHK(hipStreamSynchronize(g.stream));
    return LSA_OK;
}

// Builds the line tables of points the cache does not hold yet, asynchronously (no host synchronisation): a
// verifier that derives its G2 points one after the other on the host (CPPoly::verify: pts[i] * g2,
// /root/reference/src/gadgets/poly.h:116-118) overlaps the G2 arithmetic of point i with its own work on point i + 1.
int lsa_g2_tables_prefetch(const void *g2_jac, size_t n) {
    LSA_TRACE_CALL("g2_tables_prefetch", n);
    int rc = require_ready();
    if (rc) return rc;
    if (n == 0) return LSA_OK;
    if (!g2_jac) { set_error("g2_tables_prefetch: null argument"); return LSA_ERR_INVALID; }
    g_tabs.read_env();
    if (g_tabs.max_tables == 0 || n > 1024) return LSA_OK;         // nothing to keep them in
    g_tabs.tick++;
    for (size_t i = 0; i < n; i++) {
        const char *q = (const char *)g2_jac + i * sizeof(Jac<Fq2>);
        const Key128 key = fingerprint(q, sizeof(Jac<Fq2>), 1);
        if (g_tabs.lookup(key) >= 0) continue;
        const long s = g_tabs.insert(key, true);                    // pending: no later insert may evict it before it is built
        if (s < 0) continue;                                        // full of tables of this very call: leave it to the Miller call
        Jac<Fq2> pt;
        memcpy(&pt, q, sizeof pt);
        g_pending.pts.push_back(pt);
        g_pending.dst.push_back((uint64_t)(uintptr_t)g_tabs.ptr((uint32_t)s));
        g_pending.keys.push_back(key);
    }
    // (a cache smaller than a batch could hand a promised slot to somebody else before it is built)
    if (g_pending.pts.size() >= PendingTables::batch() || g_tabs.max_tables < 4 * PendingTables::batch()) return g_pending.flush();
    return LSA_OK;
}

int lsa_pairing_set_chunk(unsigned pairs_per_accumulator) {
    if (pairs_per_accumulator > miller_tab_max_pairs()) { set_error("pairing_set_chunk: at most %u pairs share an accumulator", miller_tab_max_pairs()); return LSA_ERR_INVALID; }
    g_force_m = pairs_per_accumulator;
    return LSA_OK;
}
int lsa_g2_table_cache(size_t max_tables) {
    int rc = require_ready();
    if (rc) return rc;
    g_pending.drop();                                           // their slots go with the cache
    HIPCHK(hipStreamSynchronize(g.stream));
    g_tabs.env_read = true;
    g_tabs.clear();
    g_tabs.max_tables = max_tables;
    return LSA_OK;
}
int lsa_g2_table_cache_stats(uint64_t out[4]) {
    if (!out) return LSA_ERR_INVALID;
    out[0] = g_tabs.hits; out[1] = g_tabs.misses; out[2] = g_tabs.slots.size(); out[3] = g_tabs.evictions;
    return LSA_OK;
}

int lsa_miller_loop(const void *g1, const void *g2, size_t n, void *out, int on_device) {
    int rc = require_ready();
    if (rc) return rc;
    if (n == 0) return LSA_OK;
    if (!g1 || !g2 || !out) { set_error("miller_loop: null argument"); return LSA_ERR_INVALID; }
    Terms t;
    t.g1 = g1; t.g2 = g2; t.n = n; t.on_device = on_device != 0;
    if (!on_device) return run_terms_host(t, out, false);
    void *res = nullptr;
    rc = run_miller(t, &res);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, res, n * fq12_bytes(), hipMemcpyDeviceToDevice, g.stream));
    return LSA_OK;
}
int lsa_miller_loop_precomp(const void *g1, const void *const *q_precomp, size_t n, void *out) {
    if (n && !q_precomp) { set_error("miller_loop_precomp: null argument"); return LSA_ERR_INVALID; }
    for (size_t i = 0; i < n; i++)
        if (!q_precomp[i]) { set_error("miller_loop_precomp: term %zu has no table", i); return LSA_ERR_INVALID; }
    Terms t;
    t.g1 = g1; t.qpre = q_precomp; t.n = n;
    return run_terms_host(t, out, false);
}
int lsa_pairing_terms(const void *g1, const void *g2, const void *const *q_precomp, const uint8_t *flags, const uint64_t *seg_offsets, size_t nseg,
                      void *out, int final_exp) {
    int rc = require_ready();
    if (rc) return rc;
    if (nseg == 0) return LSA_OK;
    rc = check_segments(seg_offsets, nseg, "pairing_terms");
    if (rc) return rc;
    Terms t;
    t.g1 = g1; t.g2 = g2; t.qpre = q_precomp; t.flags = flags; t.seg = seg_offsets; t.nseg = nseg; t.n = (size_t)seg_offsets[nseg];
    if (t.n && !g2 && !q_precomp) { set_error("pairing_terms: neither points nor tables"); return LSA_ERR_INVALID; }
    return run_terms_host(t, out, final_exp != 0);
}

static int product_host(const void *g1, const void *g2, size_t n, void *out, bool final_exp, bool sharded) {
    LSA_TRACE_CALL("pairing_product", n);
    int rc = require_ready();
    if (rc) return rc;
    if (!out || (n && (!g1 || !g2))) { set_error("pairing: null argument"); return LSA_ERR_INVALID; }
    const uint64_t seg[2] = {0, n};
    Terms t;
    t.g1 = g1; t.g2 = g2; t.seg = seg; t.nseg = 1; t.n = n;
    t.gt_only = final_exp;
    if (!(sharded && lsa_comm_world() > 1)) return run_terms_host(t, out, final_exp);
    // per-rank Miller product (1 for an empty slice), all-gather of the Fq12 partials, product in rank
    // order, one final exponentiation on every rank (SURVEY.md 8e "Pairings").  A rank whose local part
    // fails still takes part in the collective (with 1) and reports its error afterwards: the others
    // must not block in ncclAllGather.
    const size_t world = (size_t)lsa_comm_world();
    void *res = nullptr;
    int local = LSA_OK;
    if (g_pair_o.ensure(fq12_bytes()) || g_stage_gather.ensure(world * fq12_bytes()) || g_pair_s.ensure((world + 8) * fq12_bytes())) local = LSA_ERR_NOMEM;
    if (!local) local = run_miller(t, &res);
    if (local == LSA_ERR_NOMEM && !g_pair_o.p) { set_error("pairing: hipMalloc failed"); return local; }   // nothing to contribute from
    if (local) {
        Fq12 one = Fq12::one();
        (void)hipMemcpy(g_pair_o.p, &one, sizeof one, hipMemcpyHostToDevice);
    } else {
        HIPCHK(hipMemcpyAsync(g_pair_o.p, res, fq12_bytes(), hipMemcpyDeviceToDevice, g.stream));
    }
    rc = lsa_comm_all_gather(g_pair_o.p, g_stage_gather.p, 12);
    if (local) return local;
    if (rc) return rc;
    rc = fq12_product_device(g_stage_gather.p, g_pair_s.p, world, &res, g.stream);
    if (rc) return rc;
    if (final_exp) {
        rc = final_exp_device(res, 1, g_pair_o.p, g.stream);
        if (rc) return rc;
        res = g_pair_o.p;
    }
    HIPCHK(hipMemcpyAsync(g.h_result, res, fq12_bytes(), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    memcpy(out, g.h_result, fq12_bytes());
    return LSA_OK;
}
int lsa_miller_loop_product(const void *g1, const void *g2, size_t n, void *out) { return product_host(g1, g2, n, out, false, false); }
int lsa_pairing_product(const void *g1, const void *g2, size_t n, void *out) { return product_host(g1, g2, n, out, true, false); }
int lsa_pairing_product_sharded(const void *g1, const void *g2, size_t n_local, void *out) { return product_host(g1, g2, n_local, out, true, true); }

// many independent products in one pass: one upload, one Miller launch over all pairs, one product
// workgroup per segment, one batched final exponentiation, one download
int lsa_pairing_product_segments(const void *g1, const void *g2, const uint64_t *seg_offsets, size_t nseg, void *out_gt, int final_exp) {
    int rc = require_ready();
    if (rc) return rc;
    if (nseg == 0) return LSA_OK;
    rc = check_segments(seg_offsets, nseg, "pairing_product_segments");
    if (rc) return rc;
    if (!out_gt) { set_error("pairing_product_segments: null argument"); return LSA_ERR_INVALID; }
    const size_t n = (size_t)seg_offsets[nseg];
    if (n && (!g1 || !g2)) { set_error("pairing_product_segments: null argument"); return LSA_ERR_INVALID; }
    Terms t;
    t.g1 = g1; t.g2 = g2; t.seg = seg_offsets; t.nseg = nseg; t.n = n;
    return run_terms_host(t, out_gt, final_exp != 0);
}

int lsa_fq12_product(const void *in, size_t n, void *out) {
    LSA_TRACE_CALL("fq12_product", n);
    int rc = require_ready();
    if (rc) return rc;
    if (!out || (n && !in)) { set_error("fq12_product: null argument"); return LSA_ERR_INVALID; }
    if (n == 0) {
        Fq12 one = Fq12::one();
        memcpy(out, &one, sizeof one);
        return LSA_OK;
    }
    void *res = nullptr;
    if (g_pair_f.ensure(n * fq12_bytes()) || g_pair_s.ensure(((n + 7) / 8 + 1) * fq12_bytes())) { set_error("fq12_product: hipMalloc failed"); return LSA_ERR_NOMEM; }
    LSA_UPLOAD(g_pair_f.p, in, n * fq12_bytes());
    rc = fq12_product_device(g_pair_f.p, g_pair_s.p, n, &res, g.stream);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(g.h_result, res, fq12_bytes(), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    memcpy(out, g.h_result, fq12_bytes());
    return LSA_OK;
}
int lsa_final_exponentiation(const void *in, size_t n, void *out, int on_device) {
    LSA_TRACE_CALL("final_exponentiation", n);
    int rc = require_ready();
    if (rc) return rc;
    if (n == 0) return LSA_OK;
    if (!in || !out) { set_error("final_exponentiation: null argument"); return LSA_ERR_INVALID; }
    if (on_device) return final_exp_device(in, n, out, g.stream);
    if (g_pair_f.ensure(n * fq12_bytes()) || g_pair_s.ensure(n * fq12_bytes())) { set_error("final_exponentiation: hipMalloc failed"); return LSA_ERR_NOMEM; }
    LSA_UPLOAD(g_pair_f.p, in, n * fq12_bytes());
    rc = final_exp_device(g_pair_f.p, n, g_pair_s.p, g.stream);
    if (rc) return rc;
    LSA_DOWNLOAD(out, g_pair_s.p, n * fq12_bytes());
    HIPCHK(hipStreamSynchronize(g.stream));
    return LSA_OK;
}
}  // extern "C"

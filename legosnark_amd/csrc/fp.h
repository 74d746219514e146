// legosnark_amd/csrc/fp.h -- 254-bit prime-field arithmetic for alt_bn128 (Fq, Fr),
// shared by the gfx950 kernels and the host side of the C-ABI.
//
// Layout = libff's Fp_model<4>: 32 bytes, little-endian limbs, Montgomery form with
// R = 2^256 (SURVEY.md section 8 header).  On the device a field element is held as
// 8 x 32-bit limbs so that every limb product is one v_mad_u64_u32 (32x32+64 -> 64);
// 8 x u32 LE and 4 x u64 LE are the same bytes, so buffers cross the C-ABI unchanged.
//
// No MFMA: this is modular integer arithmetic, not a dense contraction.
#pragma once
#include <stdint.h>
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#include <x86intrin.h>                       // _addcarry_u64 / _subborrow_u64 for the host-side limbs
#endif
#include "bn254_constants.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LSA_HD __host__ __device__ __forceinline__
#define LSA_HD_NOINLINE inline __host__ __device__ __noinline__
#else
#define LSA_HD inline
#define LSA_HD_NOINLINE inline
#endif

namespace lsa {

struct FqParams {
    static constexpr uint32_t INV = LSA_P_INV;
    static constexpr const uint32_t (&MOD)[8] = LSA_P;
    static constexpr const uint32_t (&ONE)[8] = LSA_FQ_ONE;
    static constexpr const uint32_t (&R2)[8] = LSA_FQ_R2;
};
struct FrParams {
    static constexpr uint32_t INV = LSA_R_INV;
    static constexpr const uint32_t (&MOD)[8] = LSA_R;
    static constexpr const uint32_t (&ONE)[8] = LSA_FR_ONE;
    static constexpr const uint32_t (&R2)[8] = LSA_FR_R2;
};

#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__) && !defined(LSA_FP_HOST32)
#define LSA_FP_HOST64 1
// Host side (the libff-compatible operators the reference's own loops run on -- sumcheck tables, witness recursions,
// point arithmetic of the cold paths): the same field on four 64-bit limbs, every loop unrolled, no carry word
// (the moduli leave the top two bits free, so the interleaved Montgomery sum never leaves 2 * MOD < 2^255).
// Same canonical results as the 32-bit limb code below, which the device and -DLSA_FP_HOST32 builds use.
namespace host64 {
typedef unsigned __int128 u128;
// a * b + c + carry -> low word, carry <- high word (cannot overflow: (2^64-1)^2 + 2 (2^64-1) = 2^128 - 1)
static inline uint64_t mac(uint64_t a, uint64_t b, uint64_t c, uint64_t &carry) {
    const u128 r = (u128)a * b + c + carry;
    carry = (uint64_t)(r >> 64);
    return (uint64_t)r;
}
// x (4 limbs) += y, carry out returned; d = x - y, borrow out returned (straight-line: no loop survives at -O2)
#if defined(__x86_64__)
static inline unsigned char add4(uint64_t x[4], const uint64_t y[4]) {
    unsigned long long r0, r1, r2, r3;
    unsigned char c = _addcarry_u64(0, x[0], y[0], &r0);
    c = _addcarry_u64(c, x[1], y[1], &r1);
    c = _addcarry_u64(c, x[2], y[2], &r2);
    c = _addcarry_u64(c, x[3], y[3], &r3);
    x[0] = r0; x[1] = r1; x[2] = r2; x[3] = r3;
    return c;
}
static inline unsigned char sub4(uint64_t d[4], const uint64_t x[4], const uint64_t y[4]) {
    unsigned long long r0, r1, r2, r3;
    unsigned char c = _subborrow_u64(0, x[0], y[0], &r0);
    c = _subborrow_u64(c, x[1], y[1], &r1);
    c = _subborrow_u64(c, x[2], y[2], &r2);
    c = _subborrow_u64(c, x[3], y[3], &r3);
    d[0] = r0; d[1] = r1; d[2] = r2; d[3] = r3;
    return c;
}
#else
static inline unsigned char add4(uint64_t x[4], const uint64_t y[4]) {
    u128 c = 0;
    c += (u128)x[0] + y[0]; x[0] = (uint64_t)c; c >>= 64;
    c += (u128)x[1] + y[1]; x[1] = (uint64_t)c; c >>= 64;
    c += (u128)x[2] + y[2]; x[2] = (uint64_t)c; c >>= 64;
    c += (u128)x[3] + y[3]; x[3] = (uint64_t)c; c >>= 64;
    return (unsigned char)c;
}
static inline unsigned char sub4(uint64_t d[4], const uint64_t x[4], const uint64_t y[4]) {
    u128 t;
    uint64_t bw = 0;
    t = (u128)x[0] - y[0] - bw; d[0] = (uint64_t)t; bw = (uint64_t)(t >> 64) & 1;
    t = (u128)x[1] - y[1] - bw; d[1] = (uint64_t)t; bw = (uint64_t)(t >> 64) & 1;
    t = (u128)x[2] - y[2] - bw; d[2] = (uint64_t)t; bw = (uint64_t)(t >> 64) & 1;
    t = (u128)x[3] - y[3] - bw; d[3] = (uint64_t)t; bw = (uint64_t)(t >> 64) & 1;
    return (unsigned char)bw;
}
#endif
constexpr uint64_t limb(const uint32_t (&m)[8], int i) { return (uint64_t)m[2 * i] | ((uint64_t)m[2 * i + 1] << 32); }
// -m^-1 mod 2^64 from -m^-1 mod 2^32 (two Newton steps on the inverse are more than enough)
constexpr uint64_t neg_inv64(uint64_t m0, uint32_t neg_inv32) {
    uint64_t x = (uint64_t)0 - (uint64_t)neg_inv32;
    x *= 2 - m0 * x;
    x *= 2 - m0 * x;
    return (uint64_t)0 - x;
}
// t < 2 * MOD -> t mod MOD, branch-free
static inline void reduce_once(uint64_t t[4], const uint64_t q[4]) {
    uint64_t d[4];
    const uint64_t keep = (uint64_t)0 - (uint64_t)sub4(d, t, q);       // all ones: t < MOD
    t[0] = (t[0] & keep) | (d[0] & ~keep);
    t[1] = (t[1] & keep) | (d[1] & ~keep);
    t[2] = (t[2] & keep) | (d[2] & ~keep);
    t[3] = (t[3] & keep) | (d[3] & ~keep);
}
}  // namespace host64
#endif

template <class P>
struct Fp {
    uint32_t l[8];

#if defined(LSA_FP_HOST64)
    static constexpr uint64_t Q64[4] = {host64::limb(P::MOD, 0), host64::limb(P::MOD, 1), host64::limb(P::MOD, 2), host64::limb(P::MOD, 3)};
    static constexpr uint64_t NINV64 = host64::neg_inv64(host64::limb(P::MOD, 0), P::INV);
    // (written limb by limb, not as a 32-byte copy: compilers merge the halves into 64-bit moves and keep the value
    // in registers through the by-value copies of the operator chains)
    void load64(uint64_t x[4]) const {
        x[0] = (uint64_t)l[0] | ((uint64_t)l[1] << 32);
        x[1] = (uint64_t)l[2] | ((uint64_t)l[3] << 32);
        x[2] = (uint64_t)l[4] | ((uint64_t)l[5] << 32);
        x[3] = (uint64_t)l[6] | ((uint64_t)l[7] << 32);
    }
    void store64(const uint64_t x[4]) {
        l[0] = (uint32_t)x[0]; l[1] = (uint32_t)(x[0] >> 32);
        l[2] = (uint32_t)x[1]; l[3] = (uint32_t)(x[1] >> 32);
        l[4] = (uint32_t)x[2]; l[5] = (uint32_t)(x[2] >> 32);
        l[6] = (uint32_t)x[3]; l[7] = (uint32_t)(x[3] >> 32);
    }
#endif
    static LSA_HD Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = 0;
        return r;
    }
    static LSA_HD Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = P::ONE[i];
        return r;
    }
    static LSA_HD Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = P::R2[i];
        return r;
    }
    LSA_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= l[i];
        return o == 0;
    }
    LSA_HD bool operator==(const Fp &b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= l[i] ^ b.l[i];
        return o == 0;
    }
    LSA_HD bool operator!=(const Fp &b) const { return !(*this == b); }

    // r = t - MOD if t >= MOD else t   (t < 2*MOD)
    static LSA_HD void reduce_once(uint32_t t[8]) {
        uint32_t d[8];
        uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t x = (uint64_t)t[i] - P::MOD[i] - br;
            d[i] = (uint32_t)x;
            br = (x >> 32) & 1;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = br ? t[i] : d[i];
    }

    friend LSA_HD Fp operator+(const Fp &a, const Fp &b) {
        Fp r;
#if defined(LSA_FP_HOST64)
        {
            uint64_t x[4], y[4];
            const uint64_t q[4] = {Q64[0], Q64[1], Q64[2], Q64[3]};
            a.load64(x);
            b.load64(y);
            (void)host64::add4(x, y);                              // a, b < MOD < 2^254: no carry out
            host64::reduce_once(x, q);
            r.store64(x);
            return r;
        }
#endif
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c += (uint64_t)a.l[i] + b.l[i];
            r.l[i] = (uint32_t)c;
            c >>= 32;
        }
        // a,b < MOD < 2^254 so no carry out of limb 7
        reduce_once(r.l);
        return r;
    }
    friend LSA_HD Fp operator-(const Fp &a, const Fp &b) {
        Fp r;
#if defined(LSA_FP_HOST64)
        {
            uint64_t x[4], y[4], d[4];
            a.load64(x);
            b.load64(y);
            const uint64_t m = (uint64_t)0 - (uint64_t)host64::sub4(d, x, y);
            const uint64_t qm[4] = {Q64[0] & m, Q64[1] & m, Q64[2] & m, Q64[3] & m};
            (void)host64::add4(d, qm);
            x[0] = d[0]; x[1] = d[1]; x[2] = d[2]; x[3] = d[3];
            r.store64(x);
            return r;
        }
#endif
        uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t x = (uint64_t)a.l[i] - b.l[i] - br;
            r.l[i] = (uint32_t)x;
            br = (x >> 32) & 1;
        }
        uint32_t mask = br ? 0xffffffffu : 0u;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c += (uint64_t)r.l[i] + (P::MOD[i] & mask);
            r.l[i] = (uint32_t)c;
            c >>= 32;
        }
        return r;
    }
    LSA_HD Fp neg() const { return is_zero() ? *this : (zero() - *this); }
    LSA_HD Fp dbl() const { return *this + *this; }

    // CIOS Montgomery product, 32-bit limbs: a*b*2^-256 mod MOD, result in [0, MOD).
    // Invariant t < 2*MOD < 2^255 after every outer iteration, so only one transient
    // overflow limb is needed.
    static LSA_HD Fp mul_inline(const Fp &a, const Fp &b) {
#if defined(LSA_FP_HOST64)
        // interleaved (CIOS) product without the carry word: after each of the four rounds t < 2 * MOD < 2^255
        uint64_t x[4], y[4], t[4] = {0, 0, 0, 0};
        const uint64_t q[4] = {Q64[0], Q64[1], Q64[2], Q64[3]};
        a.load64(x);
        b.load64(y);
#define LSA_FP_ROUND(i)                                                  \
        {                                                                \
            uint64_t A = 0, C = 0;                                       \
            t[0] = host64::mac(x[0], y[i], t[0], A);                     \
            const uint64_t m = t[0] * NINV64;                            \
            (void)host64::mac(m, q[0], t[0], C);                         \
            t[1] = host64::mac(x[1], y[i], t[1], A);                     \
            t[0] = host64::mac(m, q[1], t[1], C);                        \
            t[2] = host64::mac(x[2], y[i], t[2], A);                     \
            t[1] = host64::mac(m, q[2], t[2], C);                        \
            t[3] = host64::mac(x[3], y[i], t[3], A);                     \
            t[2] = host64::mac(m, q[3], t[3], C);                        \
            t[3] = A + C;                                                \
        }
        LSA_FP_ROUND(0) LSA_FP_ROUND(1) LSA_FP_ROUND(2) LSA_FP_ROUND(3)
#undef LSA_FP_ROUND
        host64::reduce_once(t, q);
        Fp r64;
        r64.store64(t);
        return r64;
#else
        uint32_t t[8];
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t c = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                c += (uint64_t)a.l[j] * b.l[i] + t[j];
                t[j] = (uint32_t)c;
                c >>= 32;
            }
            uint32_t t8 = (uint32_t)c;
            uint32_t m = t[0] * P::INV;
            c = (uint64_t)m * P::MOD[0] + t[0];
            c >>= 32;
#pragma unroll
            for (int j = 1; j < 8; j++) {
                c += (uint64_t)m * P::MOD[j] + t[j];
                t[j - 1] = (uint32_t)c;
                c >>= 32;
            }
            t[7] = (uint32_t)(c + t8);
        }
        Fp r;
        reduce_once(t);
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = t[i];
        return r;
#endif
    }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LSA_INLINE_FIELD_MUL)
    // On the device the ~600-instruction product is a real function (arguments and
    // result in VGPRs): keeps kernels small enough for the instruction cache and
    // compile times sane.  Define LSA_INLINE_FIELD_MUL in a translation unit to inline.
    static __device__ __noinline__ Fp mul_call(Fp a, Fp b) { return mul_inline(a, b); }
    friend __device__ __forceinline__ Fp operator*(const Fp &a, const Fp &b) { return mul_call(a, b); }
#else
    friend LSA_HD Fp operator*(const Fp &a, const Fp &b) { return mul_inline(a, b); }
#endif
    LSA_HD Fp sqr() const { return *this * *this; }

    // Montgomery -> canonical integer limbs (libff as_bigint()).
    LSA_HD void to_canonical(uint32_t out[8]) const {
        Fp o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.l[i] = (i == 0);
        Fp r = *this * o;
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = r.l[i];
    }
    static LSA_HD Fp from_canonical(const uint32_t in[8]) {
        Fp x;
#pragma unroll
        for (int i = 0; i < 8; i++) x.l[i] = in[i];
        return x * r2();
    }
    static LSA_HD Fp from_u32(uint32_t v) {
        Fp x = zero();
        x.l[0] = v;
        return x * r2();
    }

    // a^(MOD-2) (Fermat); not inlined: cold, and large.
    LSA_HD_NOINLINE Fp inverse() const {
#if defined(LSA_FP_HOST64)
        // Host: binary extended Euclid on the Montgomery residue x = a R (u, v shrink by a bit or more per step;
        // b x = u and c x = v mod MOD throughout), then a^-1 R = x^-1 R^2 by one product with R^3.  Same value as
        // the power below (the inverse is unique); 0 -> 0 like the power.  Not constant-time -- neither is libff's
        // (mpn_gcdext).
        if (is_zero()) return *this;
        {
            uint64_t u[4], v[4] = {Q64[0], Q64[1], Q64[2], Q64[3]}, b[4] = {1, 0, 0, 0}, c[4] = {0, 0, 0, 0}, t[4];
            const uint64_t q[4] = {Q64[0], Q64[1], Q64[2], Q64[3]};
            load64(u);
            auto halve = [&](uint64_t w[4], uint64_t k[4]) {       // w even: w /= 2, k /= 2 mod MOD
                w[0] = (w[0] >> 1) | (w[1] << 63); w[1] = (w[1] >> 1) | (w[2] << 63); w[2] = (w[2] >> 1) | (w[3] << 63); w[3] >>= 1;
                uint64_t top = 0;
                if (k[0] & 1) top = host64::add4(k, q);
                k[0] = (k[0] >> 1) | (k[1] << 63); k[1] = (k[1] >> 1) | (k[2] << 63); k[2] = (k[2] >> 1) | (k[3] << 63);
                k[3] = (k[3] >> 1) | (top << 63);
            };
            auto is_one = [](const uint64_t w[4]) { return w[0] == 1 && (w[1] | w[2] | w[3]) == 0; };
            auto sub_mod = [&](uint64_t k[4], const uint64_t m[4]) {   // k <- k - m mod MOD (both < MOD)
                if (host64::sub4(t, k, m)) (void)host64::add4(t, q);
                k[0] = t[0]; k[1] = t[1]; k[2] = t[2]; k[3] = t[3];
            };
            while (!is_one(u) && !is_one(v)) {
                while (!(u[0] & 1)) halve(u, b);
                while (!(v[0] & 1)) halve(v, c);
                if (host64::sub4(t, u, v)) {                        // u < v
                    (void)host64::sub4(v, v, u);
                    sub_mod(c, b);
                } else {
                    u[0] = t[0]; u[1] = t[1]; u[2] = t[2]; u[3] = t[3];
                    sub_mod(b, c);
                }
            }
            Fp xinv;
            xinv.store64(is_one(u) ? b : c);
            static const Fp R3 = r2() * r2();                       // R^2 R^2 R^-1
            return xinv * R3;
        }
#endif
        uint32_t e[8];
        uint64_t br = 2;
        for (int i = 0; i < 8; i++) {
            uint64_t x = (uint64_t)P::MOD[i] - br;
            e[i] = (uint32_t)x;
            br = (x >> 32) & 1;
        }
        Fp acc = one();
        for (int i = 255; i >= 0; --i) {
            acc = acc.sqr();
            if ((e[i >> 5] >> (i & 31)) & 1) acc = acc * *this;
        }
        return acc;
    }
};

using Fq = Fp<FqParams>;
using Fr = Fp<FrParams>;

// ------------------------------------------------------------------ Fq2 = Fq[u]/(u^2+1)
// Templated on the base-field representation B: Fq (libff layout, saturated limbs) for the
// boundary / host code, Fs (fs29.h, 29-bit limbs) inside the pairing kernels.
template <class B>
struct Fq2T {
    B c0, c1;
    static LSA_HD Fq2T zero() { return {B::zero(), B::zero()}; }
    static LSA_HD Fq2T one() { return {B::one(), B::zero()}; }
    LSA_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    LSA_HD bool operator==(const Fq2T &b) const { return c0 == b.c0 && c1 == b.c1; }
    LSA_HD bool operator!=(const Fq2T &b) const { return !(*this == b); }
    friend LSA_HD Fq2T operator+(const Fq2T &a, const Fq2T &b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
    friend LSA_HD Fq2T operator-(const Fq2T &a, const Fq2T &b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
    friend LSA_HD Fq2T operator*(const Fq2T &a, const Fq2T &b) {
        B aa = a.c0 * b.c0, bb = a.c1 * b.c1;
        B s = (a.c0 + a.c1) * (b.c0 + b.c1);
        return {aa - bb, s - aa - bb};
    }
    LSA_HD Fq2T sqr() const {
        B m = c0 * c1;
        return {(c0 + c1) * (c0 - c1), m + m};
    }
    LSA_HD Fq2T neg() const { return {c0.neg(), c1.neg()}; }
    LSA_HD Fq2T dbl() const { return {c0.dbl(), c1.dbl()}; }
    LSA_HD Fq2T conj() const { return {c0, c1.neg()}; }
    LSA_HD Fq2T mul_fq(const B &k) const { return {c0 * k, c1 * k}; }
    // * xi, xi = 9 + u:  (9 c0 - c1) + (9 c1 + c0) u
    LSA_HD Fq2T mul_xi() const {
        B a8 = c0.dbl().dbl().dbl(), b8 = c1.dbl().dbl().dbl();
        return {a8 + c0 - c1, b8 + c1 + c0};
    }
    LSA_HD_NOINLINE Fq2T inverse() const {
        B n = (c0.sqr() + c1.sqr()).inverse();
        return {c0 * n, (c1 * n).neg()};
    }
};
using Fq2 = Fq2T<Fq>;

template <class F> struct FieldName;

}  // namespace lsa

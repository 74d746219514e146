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

template <class P>
struct Fp {
    uint32_t l[8];

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
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__) && !defined(LSA_FP_HOST32)
        // Host side (the shim's cold operators, the host unit tests): the same CIOS on four
        // 64-bit limbs, ~3x fewer multiplications.  Same canonical result.
        typedef unsigned __int128 u128;
        uint64_t x[4], y[4], q[4];
        for (int i = 0; i < 4; i++) {
            x[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
            y[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
            q[i] = (uint64_t)P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32);
        }
        uint64_t pinv = (uint64_t)0 - (uint64_t)P::INV;       // p^-1 mod 2^32 in the low half ...
        pinv *= 2 - q[0] * pinv;                              // ... lifted to 2^64 by one Newton step
        const uint64_t ninv = (uint64_t)0 - pinv;             // -p^-1 mod 2^64
        uint64_t t[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (u128)x[j] * y[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[4] = (uint64_t)c;
            const uint64_t top = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * ninv;
            c = (u128)m * q[0] + t[0];
            c >>= 64;
            for (int j = 1; j < 4; j++) {
                c += (u128)m * q[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[3] = (uint64_t)c;
            t[4] = top + (uint64_t)(c >> 64);
        }
        uint32_t w[8];
        for (int i = 0; i < 4; i++) { w[2 * i] = (uint32_t)t[i]; w[2 * i + 1] = (uint32_t)(t[i] >> 32); }
        reduce_once(w);                                       // t < 2*MOD < 2^255: t[4] == 0
        Fp r64;
        for (int i = 0; i < 8; i++) r64.l[i] = w[i];
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

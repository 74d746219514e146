// tools/ubench_batch_affine.hip -- what batched-affine bucket additions could buy over the XYZZ mixed addition of
// k_accumulate (msm.hip), measured as an UPPER bound: the arithmetic and the memory traffic of the affine scheme with
// none of its bookkeeping (no pairing-up of bucket lists, no rounds, no P + P / P - P handling), against the
// arithmetic and the gathers of the XYZZ scheme.
//
//   xyzz     every lane walks a list of points gathered by random index and adds them to its XYZZ accumulator:
//            8M + 2S per addition, one 64-byte gather (the dominant kernel of the MSM today).
//   affine   every lane performs K independent affine additions (x1,y1) + (x2,y2) per batch with ONE field inversion
//            per WAVEFRONT and batch (Montgomery's trick: per-lane prefix products parked in a global scratch array
//            that stays in L2, a shuffle product tree over the 64 lanes, the constant-time binary-GCD inversion of
//            inv29.h executed once, the tree walked back down): per addition 2 gathers of 64 B in the forward pass,
//            2 again in the backward pass, 36 B of prefix product written and read, 64 B of result written;
//            5M + 1S + (2 * 6 tree products + 1 inversion) / K.
//
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I legosnark_amd/csrc tools/ubench_batch_affine.hip -o /tmp/ubench_batch_affine
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "fp29.h"
#include "fs29.h"

using namespace lsa;

__global__ __launch_bounds__(256) void k_xyzz(const AffPacked *__restrict__ pts, const uint32_t *__restrict__ idx, uint32_t per_lane, uint32_t npts,
                                               XYZZ29 *__restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ29 acc = XYZZ29::inf();
    const uint32_t *e = idx + (size_t)t * per_lane;
    AffPacked cur = pts[e[0] % npts];
    for (uint32_t j = 0; j < per_lane; j++) {
        AffPacked nxt = cur;
        if (j + 1 < per_lane) nxt = pts[e[j + 1] % npts];
        acc = xyzz29_madd(acc, unpack_affine(cur));
        cur = nxt;
    }
    out[t] = acc;
}

template <class A>
__device__ __forceinline__ A shfl_xor_f29(const A &p, int m) {
    A r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (uint32_t)__shfl_xor((int)p.l[i], m, 64);
    return r;
}

// K additions per lane and batch; `batches` batches
template <int K>
__global__ __launch_bounds__(256) void k_affine(const AffPacked *__restrict__ pts, const uint32_t *__restrict__ idx, uint32_t batches, uint32_t npts,
                                                 F29 *__restrict__ prefix, AffPacked *__restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    F29 *pre = prefix + (size_t)t * K;
    const uint32_t *e = idx + (size_t)t * 2 * K * batches;
    for (uint32_t b = 0; b < batches; b++, e += 2 * K) {
        // forward: d_i = x2 - x1, running product
        F29 run = F29::one();
#pragma unroll 1
        for (int i = 0; i < K; i++) {
            const AffPacked p1 = pts[e[2 * i] % npts], p2 = pts[e[2 * i + 1] % npts];
            const F29 d = sub_k<2>(F29::unpack256(p2.x), F29::unpack256(p1.x));
            pre[i] = run;
            run = mul(run, d);
        }
        // product over the wavefront (butterfly: every lane ends with the product of all 64), its inverse once, and every
        // lane's share: inv(run_lane) = inv(total) * (product of the other 63) -- kept as a prefix/suffix pair per level
        F29 tot = run, others = F29::one();
#pragma unroll 1
        for (int m = 1; m < 64; m <<= 1) {
            const F29 o = shfl_xor_f29(tot, m);
            others = mul(others, o);          // product of everything outside this lane's group so far
            tot = mul(tot, o);
        }
        F29 inv = mul(Fs{tot}.inverse().v, others);      // 1 / run_lane
        // backward: inv_i = inv * pre_i; inv *= d_i; lambda, x3, y3
#pragma unroll 1
        for (int i = K - 1; i >= 0; i--) {
            const AffPacked p1 = pts[e[2 * i] % npts], p2 = pts[e[2 * i + 1] % npts];
            const F29 x1 = F29::unpack256(p1.x), y1 = F29::unpack256(p1.y), x2 = F29::unpack256(p2.x), y2 = F29::unpack256(p2.y);
            const F29 d = sub_k<2>(x2, x1);
            const F29 di = mul(inv, pre[i]);
            inv = mul(inv, d);
            const F29 lam = mul(sub_k<2>(y2, y1), di);
            const F29 x3 = sub_k<4>(sqr(lam), add_lazy(x1, x2));
            const F29 y3 = sub_k<2>(mul(lam, sub_k<8>(x1, x3)), y1);
            AffPacked r;
            x3.canonical().pack256(r.x);
            y3.canonical().pack256(r.y);
            out[(size_t)t * K + i] = r;
        }
    }
}

template <class Fn>
static float time_ms(Fn fn, int reps = 3) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    fn();
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        (void)hipEventRecord(e0);
        fn();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const uint32_t npts = 13u << 20;                  // the size of the MSM's table of pre-shifted copies at n = 2^20 (832 MB)
    std::vector<AffPacked> h(1 << 16);
    // points: any canonical residues do for a throughput measurement (no P + P: distinct x)
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto &p : h) { for (int i = 0; i < 8; i++) { p.x[i] = (uint32_t)rnd(); p.y[i] = (uint32_t)rnd(); } p.x[7] &= 0x0fffffffu; p.y[7] &= 0x0fffffffu; }
    AffPacked *d_pts;
    (void)hipMalloc(&d_pts, (size_t)npts * sizeof(AffPacked));
    for (size_t off = 0; off < npts; off += h.size()) (void)hipMemcpy(d_pts + off, h.data(), h.size() * sizeof(AffPacked), hipMemcpyHostToDevice);
    const size_t total_adds = (size_t)13 << 20;       // one MSM's worth
    std::vector<uint32_t> hi(2 * total_adds);
    for (auto &x : hi) x = (uint32_t)rnd();
    uint32_t *d_idx;
    (void)hipMalloc(&d_idx, hi.size() * 4);
    (void)hipMemcpy(d_idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
    void *d_out;
    (void)hipMalloc(&d_out, total_adds * sizeof(XYZZ29));
    F29 *d_pre;
    (void)hipMalloc(&d_pre, total_adds * sizeof(F29));
    printf("device %s, %d CUs; %zu additions per run, points gathered from a table of %u (%.0f MB)\n", prop.gcnArchName, prop.multiProcessorCount, total_adds, npts,
           npts * 64.0 / 1e6);
    {
        const uint32_t lanes = 1u << 19, per = (uint32_t)(total_adds / lanes);      // 26 per lane: the MSM's bucket lists
        const float ms = time_ms([&] { hipLaunchKernelGGL(k_xyzz, dim3(lanes / 256), dim3(256), 0, 0, d_pts, d_idx, per, npts, (XYZZ29 *)d_out); });
        printf("xyzz    mixed additions (8M + 2S, 1 gather):          %7.3f ms  %6.2f G additions/s\n", ms, total_adds / ms * 1e-6);
    }
    auto run_aff = [&](auto kern, int K, uint32_t lanes) {
        const uint32_t batches = (uint32_t)(total_adds / ((size_t)lanes * K));
        const float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(lanes / 256), dim3(256), 0, 0, d_pts, d_idx, batches, npts, d_pre, (AffPacked *)d_out); });
        printf("affine  K = %3d per lane and inversion, %7u lanes:   %7.3f ms  %6.2f G additions/s  (5M + 1S + %.1f products of tree and inversion share)\n", K, lanes, ms,
               (double)lanes * K * batches / ms * 1e-6, (18.0 + 115.0) / K);
    };
    run_aff(k_affine<16>, 16, 1u << 18);
    run_aff(k_affine<32>, 32, 1u << 18);
    run_aff(k_affine<64>, 64, 1u << 17);
    run_aff(k_affine<128>, 128, 1u << 16);
    run_aff(k_affine<64>, 64, 1u << 16);
    return 0;
}

// tools/ubench_int.hip -- instruction-throughput microbenchmark that decides how the
// 254-bit field multiplication is built on gfx950 (DESIGN.md "field multiplication").
// Measures wave-level issue rates of the candidate building blocks:
//   v_mad_u64_u32 (32x32+64), v_mul_lo_u32 / v_mul_hi_u32, v_mad_u32_u24,
//   v_add_co/v_addc, v_lshl_add_u64, v_fma_f64, and the library's Fq product.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I legosnark_amd/csrc tools/ubench_int.hip -o /tmp/ubench_int
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "fp.h"
#include "fp29.h"

using namespace lsa;

#define CHAINS 8

template <int OP>
__global__ __launch_bounds__(256) void k_op(uint32_t *out, uint32_t seed, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a[CHAINS];
    uint32_t b = seed * 2654435761u + tid * 40503u + 1u;
    double d[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { a[c] = (uint64_t)(tid + c) * 0x9E3779B97F4A7C15ull + seed; d[c] = 1.0 + (double)(tid + c) * 1e-9; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (OP == 0) {            // v_mad_u64_u32
                a[c] = (uint64_t)(uint32_t)a[c] * b + a[c];
            } else if (OP == 1) {     // v_mul_lo_u32
                a[c] = (uint32_t)a[c] * b;
            } else if (OP == 2) {     // v_mul_hi_u32
                a[c] = __umulhi((uint32_t)a[c], b);
            } else if (OP == 3) {     // v_mad_u32_u24
                a[c] = __umul24((uint32_t)a[c], b) + (uint32_t)(a[c] >> 3);
            } else if (OP == 4) {     // 64-bit add (v_lshl_add_u64 or add_co/addc)
                a[c] = a[c] + ((uint64_t)b << 7) + it;
            } else if (OP == 5) {     // v_fma_f64
                d[c] = __builtin_fma(d[c], 1.0000001, 1e-7);
            } else if (OP == 6) {     // 32-bit add3
                a[c] = (uint32_t)a[c] + b + (uint32_t)it;
            }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += a[c] + (uint64_t)d[c];
    out[tid] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

template <bool INLINE>
__global__ __launch_bounds__(256) void k_fqmul(Fq *out, const Fq *in, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fq x = in[tid], y = in[tid + 1];
    for (int it = 0; it < iters; it++) {
        if (INLINE) { x = Fq::mul_inline(x, y); y = Fq::mul_inline(y, x); }
        else { x = x * y; y = y * x; }
    }
    out[tid] = x + y;
}

template <int ILP>
__global__ __launch_bounds__(256) void k_mul29(F29 *out, const F29 *in, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    F29 x[ILP], y = in[tid + 1];
#pragma unroll
    for (int j = 0; j < ILP; j++) { x[j] = in[tid]; x[j].l[0] ^= j; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) x[j] = mul(x[j], y);      // ILP independent dependent-chains
    }
    F29 r = x[0];
#pragma unroll
    for (int j = 1; j < ILP; j++) r = add_lazy(r, x[j]);
    out[tid] = r;
}

template <class Fn>
static float time_ms(Fn fn, int reps = 3) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    fn();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0);
        fn();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("device: %s, CUs=%d, clock=%d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    const int blocks = prop.multiProcessorCount * 8, threads = 256, iters = 4096;
    uint32_t *d_out;
    hipMalloc(&d_out, (size_t)blocks * threads * 4);
    const char *names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "add_u64", "v_fma_f64", "add3_u32"};
    double total = (double)blocks * threads * iters * CHAINS;
#define RUN(OP) { float ms = time_ms([&] { hipLaunchKernelGGL((k_op<OP>), dim3(blocks), dim3(threads), 0, 0, d_out, 7u, iters); }); \
                  printf("%-16s %8.3f ms  %8.2f Gop/s  (%.2f lane-ops/clk/CU at %.2f GHz)\n", names[OP], ms, total / ms * 1e-6, \
                         total / (ms * 1e-3) / prop.multiProcessorCount / (prop.clockRate * 1e3), prop.clockRate * 1e-6); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    // field multiplication
    {
        size_t n = (size_t)blocks * threads;
        std::vector<Fq> h(n + 1);
        for (size_t i = 0; i <= n; i++) for (int j = 0; j < 8; j++) h[i].l[j] = (uint32_t)(i * 2654435761u + j * 40503u + 12345u) & (j == 7 ? 0x0fffffffu : 0xffffffffu);
        Fq *d_in, *d_o;
        hipMalloc(&d_in, (n + 1) * sizeof(Fq)); hipMalloc(&d_o, n * sizeof(Fq));
        hipMemcpy(d_in, h.data(), (n + 1) * sizeof(Fq), hipMemcpyHostToDevice);
        const int fi = 512;
        for (int occ = 1; occ <= 8; occ *= 2) {
            int bl = prop.multiProcessorCount * occ;
            double muls = (double)bl * threads * fi * 2;
            float ms1 = time_ms([&] { hipLaunchKernelGGL((k_fqmul<true>), dim3(bl), dim3(threads), 0, 0, d_o, d_in, fi); });
            float ms2 = time_ms([&] { hipLaunchKernelGGL((k_fqmul<false>), dim3(bl), dim3(threads), 0, 0, d_o, d_in, fi); });
            printf("Fq mul, %d blocks/CU: inline %8.3f ms = %7.2f Gmul/s | call %8.3f ms = %7.2f Gmul/s\n", occ, ms1, muls / ms1 * 1e-6, ms2, muls / ms2 * 1e-6);
        }
    }
    // 29-bit-limb multiplication: throughput vs occupancy and per-lane ILP
    {
        size_t n = (size_t)prop.multiProcessorCount * 8 * 256;
        std::vector<F29> h(n + 1);
        for (size_t i = 0; i <= n; i++) for (int j = 0; j < 9; j++) h[i].l[j] = (uint32_t)(i * 2654435761u + j * 40503u + 12345u) & (j == 8 ? 0x3fffffu : 0x1fffffffu);
        F29 *d_in, *d_o;
        hipMalloc(&d_in, (n + 1) * sizeof(F29)); hipMalloc(&d_o, n * sizeof(F29));
        hipMemcpy(d_in, h.data(), (n + 1) * sizeof(F29), hipMemcpyHostToDevice);
        const int fi = 512;
        for (int occ = 1; occ <= 8; occ *= 2) {
            int bl = prop.multiProcessorCount * occ;
            double muls = (double)bl * threads * fi;
            float m1 = time_ms([&] { hipLaunchKernelGGL((k_mul29<1>), dim3(bl), dim3(threads), 0, 0, d_o, d_in, fi); });
            float m2 = time_ms([&] { hipLaunchKernelGGL((k_mul29<2>), dim3(bl), dim3(threads), 0, 0, d_o, d_in, fi); });
            float m4 = time_ms([&] { hipLaunchKernelGGL((k_mul29<4>), dim3(bl), dim3(threads), 0, 0, d_o, d_in, fi); });
            printf("F29 mul, %d waves/SIMD: ILP1 %7.2f Gmul/s | ILP2 %7.2f Gmul/s | ILP4 %7.2f Gmul/s\n", occ, muls / m1 * 1e-6, 2 * muls / m2 * 1e-6, 4 * muls / m4 * 1e-6);
        }
    }
    return 0;
}

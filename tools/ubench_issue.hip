// tools/ubench_issue.hip -- exact VALU issue rates on gfx950 for the instructions a
// 254-bit modular multiplication can be built from.  Each kernel runs ITER iterations of
// 32 independent inline-asm instructions per lane (no compiler rewriting), 8 waves/SIMD.
// Reports cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP4(x) x x x x
#define REP32(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define KERNEL(NAME, ASM, CONSTR_OUT, ...)                                                   \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters) {                   \
        uint32_t a0 = threadIdx.x + 1, a1 = threadIdx.x * 3 + 7, a2 = blockIdx.x + 5, a3 = 11; \
        uint64_t q0 = a0 * 0x10001ull, q1 = a1 * 0x20003ull, q2 = a2, q3 = 99;               \
        double d0 = a0, d1 = 1.0000001, d2 = 0.5;                                             \
        float f0 = a0, f1 = 1.0001f, f2 = 0.5f;                                               \
        (void)a3; (void)q3; (void)d2; (void)f2; (void)q2; (void)d1; (void)f1;                 \
        for (int it = 0; it < iters; it++) {                                                  \
            asm volatile(REP32(ASM "\n") : CONSTR_OUT : __VA_ARGS__);                         \
        }                                                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + (uint32_t)q0 + (uint32_t)q1 + (uint32_t)d0 + (uint32_t)f0; \
    }

KERNEL(k_add_u32, "v_add_u32 %0, %0, %1", "+v"(a0), "v"(a1))
KERNEL(k_mov, "v_mov_b32 %0, %1", "+v"(a0), "v"(a1))
KERNEL(k_and, "v_and_b32 %0, %0, %1", "+v"(a0), "v"(a1))
KERNEL(k_add_co, "v_add_co_u32 %0, vcc, %0, %1", "+v"(a0), "v"(a1) : "vcc")
KERNEL(k_addc_co, "v_addc_co_u32 %0, vcc, %0, %1, vcc", "+v"(a0), "v"(a1) : "vcc")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1", "+v"(a0), "v"(a1))
KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %1", "+v"(a0), "v"(a1))
KERNEL(k_mad_u64, "v_mad_u64_u32 %0, vcc, %1, %2, %0", "+v"(q0), "v"(a0), "v"(a1) : "vcc")
KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %0", "+v"(a0), "v"(a1))
KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %1", "+v"(q0), "v"(q1))
KERNEL(k_lshrrev_b64, "v_lshrrev_b64 %0, 3, %0", "+v"(q0), "v"(q1))
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 29", "+v"(a0), "v"(a1))
KERNEL(k_fma_f64, "v_fma_f64 %0, %0, %1, %0", "+v"(d0), "v"(d1))
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %0", "+v"(f0), "v"(f1))
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1", "+v"(a0), "v"(a1))
KERNEL(k_mul_hi_u24, "v_mul_hi_u32_u24 %0, %0, %1", "+v"(a0), "v"(a1))
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc", "+v"(a0), "v"(a1) : "vcc")
// v_cndmask variants: VCC written once before the loop / SGPR-pair mask (VOP3)
__global__ __launch_bounds__(256) void k_cndmask_vcc_set(uint32_t *out, int iters) {
    uint32_t a0 = threadIdx.x + 1, a1 = threadIdx.x * 3 + 7;
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a0), "v"(a1) : "vcc");
    for (int it = 0; it < iters; it++) { asm volatile(REP32("v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(a0) : "v"(a1) : "vcc"); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
}
__global__ __launch_bounds__(256) void k_cndmask_e64(uint32_t *out, int iters) {
    uint32_t a0 = threadIdx.x + 1, a1 = threadIdx.x * 3 + 7;
    unsigned long long m = __ballot(a0 & 1);
    for (int it = 0; it < iters; it++) { asm volatile(REP32("v_cndmask_b32_e64 %0, %0, %1, %2\n") : "+v"(a0) : "v"(a1), "s"(m)); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
}
KERNEL(k_xor, "v_xor_b32 %0, %0, %1", "+v"(a0), "v"(a1))
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %1", "+v"(a0), "v"(a1))
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %1", "+v"(a0), "v"(a1))
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf", "+v"(a0), "v"(a1))
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %1", "+v"(a0), "v"(a1))
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1", "+v"(a0), "v"(a1))
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 29", "+v"(a0), "v"(a1))

// v_mad_u64_u32 is the unit a field multiplication is counted in (fp29.h: 171 of them), and the row above
// (carry-out to VCC, every instruction dependent on the previous one) is not the shape the compiler emits: its
// column sums are chains of 9 .. 18 multiply-adds into one 64-bit accumulator with the carry-out discarded into an
// SGPR pair, and neighbouring columns are independent.  The variants below separate the three candidates for the
// 7.0-against-4.1 cycles gap: the VCC write, the dependency on the previous result, and the number of wavefronts.
__global__ __launch_bounds__(256) void k_mad_u64_sgpr(uint32_t *out, int iters) {      // dependent, carry-out to s[20:21]
    uint32_t a0 = threadIdx.x + 1, a1 = threadIdx.x * 3 + 7;
    uint64_t q0 = a0 * 0x10001ull;
    for (int it = 0; it < iters; it++) { asm volatile(REP32("v_mad_u64_u32 %0, s[20:21], %1, %2, %0\n") : "+v"(q0) : "v"(a0), "v"(a1) : "s20", "s21"); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)q0;
}
#define MAD4(sd) "v_mad_u64_u32 %0, " sd ", %4, %5, %0\nv_mad_u64_u32 %1, " sd ", %4, %5, %1\nv_mad_u64_u32 %2, " sd ", %4, %5, %2\nv_mad_u64_u32 %3, " sd ", %4, %5, %3\n"
#define REP8(x) x x x x x x x x
__global__ __launch_bounds__(256) void k_mad_u64_ilp4(uint32_t *out, int iters) {      // four independent accumulators, carry-out to VCC
    uint32_t a0 = threadIdx.x + 1, a1 = threadIdx.x * 3 + 7;
    uint64_t q0 = a0, q1 = a1, q2 = a0 * 3ull, q3 = a1 * 5ull;
    for (int it = 0; it < iters; it++) { asm volatile(REP8(MAD4("vcc")) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(a0), "v"(a1) : "vcc"); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(q0 + q1 + q2 + q3);
}
__global__ __launch_bounds__(256) void k_mad_u64_ilp4_sgpr(uint32_t *out, int iters) { // four independent accumulators, carry-out to s[20:21]
    uint32_t a0 = threadIdx.x + 1, a1 = threadIdx.x * 3 + 7;
    uint64_t q0 = a0, q1 = a1, q2 = a0 * 3ull, q3 = a1 * 5ull;
    for (int it = 0; it < iters; it++) { asm volatile(REP8(MAD4("s[20:21]")) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(a0), "v"(a1) : "s20", "s21"); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(q0 + q1 + q2 + q3);
}
// the compiler's own code for the same thing: 32 multiply-adds into one accumulator per iteration
__global__ __launch_bounds__(256) void k_mad_u64_compiled(uint32_t *out, int iters) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * (2 * i + 3) + i;
    uint64_t acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc += (uint64_t)a[i] * a[(i + j + 1) & 7];
        a[it & 7] ^= (uint32_t)acc;        // keeps the products from being hoisted
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32);
}

template <class K>
static void run_w(const char *name, K kern, uint32_t *d_out, int cus, double ghz, int blocks_per_cu) {
    const int blocks = cus * blocks_per_cu, iters = 2048;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, iters);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double wave_instr_per_simd = (double)blocks_per_cu * iters * 32.0;     // 4 wavefronts per block over 4 SIMDs
    printf("%-22s %d wavefront(s)/SIMD %8.3f ms   %6.2f cycles per wave64-instruction per SIMD (at %.1f GHz nominal)\n", name, blocks_per_cu, best,
           best * 1e-3 * ghz * 1e9 / wave_instr_per_simd, ghz);
}

template <class K>
static void run(const char *name, K kern, uint32_t *d_out, int cus, double ghz) {
    const int blocks = cus * 8, iters = 2048;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, iters);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // wave-instructions per SIMD: blocks*4 waves spread over cus*4 SIMDs
    double wave_instr_per_simd = (double)blocks * 4 / (cus * 4) * iters * 32.0;
    double cyc = best * 1e-3 * ghz * 1e9 / wave_instr_per_simd;
    printf("%-18s %8.3f ms   %6.2f cycles per wave64-instruction per SIMD (at %.1f GHz nominal)\n", name, best, cyc, ghz);
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    double ghz = prop.clockRate * 1e-6;
    uint32_t *d_out;
    (void)hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4);
#define R(k) run(#k, k, d_out, cus, ghz);
    R(k_add_u32) R(k_mov) R(k_and) R(k_add_co) R(k_addc_co) R(k_add3) R(k_lshl_add) R(k_bfe) R(k_cndmask) R(k_alignbit)
    R(k_cndmask_vcc_set) R(k_cndmask_e64) R(k_xor) R(k_and_or) R(k_bfi) R(k_mov_dpp)
    R(k_mul_lo) R(k_mul_hi) R(k_mul_u24) R(k_mul_hi_u24) R(k_mad_u32_u24) R(k_mad_u64)
    R(k_lshl_add_u64) R(k_lshrrev_b64) R(k_fma_f32) R(k_fma_f64)
    for (int w : {1, 2, 4, 8}) {
        run_w("k_mad_u64 (vcc, dep)", k_mad_u64, d_out, cus, ghz, w);
        run_w("k_mad_u64_sgpr (dep)", k_mad_u64_sgpr, d_out, cus, ghz, w);
        run_w("k_mad_u64_ilp4 (vcc)", k_mad_u64_ilp4, d_out, cus, ghz, w);
        run_w("k_mad_u64_ilp4_sgpr", k_mad_u64_ilp4_sgpr, d_out, cus, ghz, w);
        run_w("k_mad_u64_compiled", k_mad_u64_compiled, d_out, cus, ghz, w);
        run_w("k_add_u32", k_add_u32, d_out, cus, ghz, w);
    }
    return 0;
}

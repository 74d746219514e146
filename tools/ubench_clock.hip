// tools/ubench_clock.hip -- the shader clock a LONE wavefront really runs at, and what a dependent instruction costs it.
// s_memtime counts shader-engine cycles, s_memrealtime a constant 100 MHz: their ratio over a long dependent chain is the
// clock; the chain length over the s_memtime difference the cycles per instruction.  Run with 1 workgroup (the final
// exponentiation's situation) and with the whole chip busy.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_clock.hip -o /tmp/ubench_clock
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int KIND>
__global__ __launch_bounds__(64) void k_chain(uint32_t iters, uint64_t *out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed ^ 0x55, d = seed + 7;
    uint64_t acc = a;
    const uint64_t t0 = __builtin_readcyclecounter();          // s_memtime
    const uint64_t r0 = wall_clock64();                        // s_memrealtime, 100 MHz
    for (uint32_t i = 0; i < iters; i++) {
        if (KIND == 0) {            // 16 dependent v_add_u32
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
        } else if (KIND == 1) {     // 16 v_add_u32 on four independent chains
#pragma unroll
            for (int k = 0; k < 4; k++)
                asm volatile("v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %4\nv_add_u32 %2, %2, %4\nv_add_u32 %3, %3, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(b) : "v"(seed));
        } else if (KIND == 2) {     // 16 dependent v_mad_u64_u32
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(b), "v"(c) : "vcc");
        } else if (KIND == 4) {     // 16 dependent v_cndmask_b32 (VOP2: the mask in VCC)
            asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
        } else if (KIND == 5) {     // 16 dependent v_cndmask_b32 (VOP3: the mask in an SGPR pair)
            uint64_t m;
            asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(m));
        } else if (KIND == 6) {     // 16 dependent v_bfi_b32 (select by mask register)
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a) : "v"(c), "v"(b));
        } else {                    // a carry step: add, and, shift (dependent triple) x 5 + 1
#pragma unroll
            for (int k = 0; k < 5; k++)
                asm volatile("v_add_u32 %0, %0, %1\nv_and_b32 %1, 0x1fffffff, %0\nv_lshrrev_b32 %0, 29, %0" : "+v"(a), "+v"(b));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t r1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[0] = t1 - t0; out[1] = r1 - r0; out[2] = a + b + c + d + acc; out[3] = xcc & 0xf;
    }
}

template <int KIND>
static void run(const char *name, uint32_t blocks, uint64_t *d_out) {
    const uint32_t iters = 20000;
    uint64_t h[3];
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k_chain<KIND>), dim3(blocks), dim3(64), 0, 0, iters, d_out, 12345u);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    const double us = h[1] / 100.0, instr = 16.0 * iters;
    printf("%-28s %6u workgroups: %8.1f us, s_memtime ticks %10llu (%.1f MHz), %.2f ns per instruction, %.2f ticks per instruction\n", name, blocks, us,
           (unsigned long long)h[0], h[0] / us, us * 1e3 / instr, h[0] / instr);
}

int main() {
    uint64_t *d_out;
    (void)hipMalloc(&d_out, 64);
    for (uint32_t blocks : {1u, 256u, 1024u, 4096u}) {
        run<0>("dependent v_add_u32", blocks, d_out);
        run<1>("v_add_u32, 4 chains", blocks, d_out);
        run<2>("dependent v_mad_u64_u32", blocks, d_out);
        run<3>("carry step (add, and, shr)", blocks, d_out);
        run<4>("dependent v_cndmask_b32_e32 (vcc)", blocks, d_out);
        run<5>("dependent v_cndmask_b32_e64", blocks, d_out);
        run<6>("dependent v_bfi_b32", blocks, d_out);
    }
    // sixteen lone workgroups one after the other: successive dispatches start on successive XCDs
    printf("lone workgroup, dependent v_add_u32, dispatch after dispatch:\n");
    for (int i = 0; i < 16; i++) {
        uint64_t h[4];
        hipLaunchKernelGGL((k_chain<0>), dim3(1), dim3(64), 0, 0, 20000u, d_out, 12345u);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
        printf("  dispatch %2d on XCC %llu: %8.1f us, %.1f MHz, %.2f ticks per instruction\n", i, (unsigned long long)h[3], h[1] / 100.0, h[0] / (h[1] / 100.0), h[0] / (16.0 * 20000));
    }
    return 0;
}

// tools/ubench_exec_mask.hip -- what a sparse EXEC mask costs a LONE workgroup on gfx950 (MI355X).
// A workgroup of three wavefronts runs the same dependent chain (64 v_mad_u64_u32 + 64 v_add_u32 per trip, one barrier per
// trip) with only the lanes of `mask` active in every wavefront, dispatch after dispatch (the dispatcher moves a lone
// workgroup from CU to CU).  All 64 lanes on: the same time on every CU.  Fewer lanes on: slower, and by a different
// factor on every CU -- which is why the lone-workgroup kernels of this library (final exponentiation, products of Fq12
// values, Miller loops over tables) keep every lane computing and predicate only their stores (w12.h: w12_pin).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_exec_mask.hip -o tools/ubench_exec_mask.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ __launch_bounds__(192) void k_chain(uint32_t iters, uint64_t mask, uint64_t *out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    uint64_t acc = a;
    const bool on = (mask >> (threadIdx.x & 63)) & 1;
    const uint64_t r0 = wall_clock64();
    for (uint32_t i = 0; i < iters; i++) {
        if (on) {
#pragma unroll
            for (int k = 0; k < 64; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(b), "v"(a) : "vcc");
#pragma unroll
            for (int k = 0; k < 64; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
        }
        __syncthreads();
    }
    const uint64_t r1 = wall_clock64();
    if (threadIdx.x == 0) {
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[0] = r1 - r0; out[1] = hw; out[2] = a + acc;
    }
}

int main() {
    uint64_t *d_out;
    (void)hipMalloc(&d_out, 64);
    const struct { const char *name; uint64_t mask; } cases[] = {
        {"all 64 lanes", ~0ull}, {"63 lanes (all but lane 63)", ~0ull >> 1}, {"lanes 0-47", (1ull << 48) - 1}, {"lanes 0-31", (1ull << 32) - 1},
        {"lanes 0-15 (one row of 16)", 0xffffull}, {"lanes 0-11", 0xfffull}, {"one lane per row of 16 (4 lanes)", 0x0001000100010001ull},
        {"every second lane (32 lanes)", 0x5555555555555555ull}, {"lane 0 only", 1ull}};
    for (const auto &c : cases) {
        printf("%-36s:", c.name);
        for (int i = 0; i < 9; i++) {
            uint64_t h[3];
            hipLaunchKernelGGL(k_chain, dim3(1), dim3(192), 0, 0, 500u, c.mask, d_out, 12345u);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
            printf(" cu%u %5.1f", (unsigned)(h[1] >> 8) & 15, h[0] / 100.0);
        }
        printf("  us\n");
    }
    return 0;
}

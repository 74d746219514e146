// tools/ubench_placement.hip -- where the dispatcher puts a lone workgroup of three wavefronts, dispatch after dispatch,
// and whether the placement changes how fast it runs (the final exponentiation's kernel alternates between 0.61 and
// 0.75 ms in a sequence of pairing checks: every phase of it 1.24x slower on one CU than on the other).  A workgroup of
// 192 threads runs a chain of one kind of instruction with a barrier every 64; a short kernel of the same shape runs
// between two of them (as k_fq12_prod8_wave does between a Miller loop and a final exponentiation).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_placement.hip -o tools/ubench_placement.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int KIND>
__global__ __launch_bounds__(192) void k_work(uint32_t iters, uint64_t *out, uint32_t seed) {
    __shared__ uint32_t lds[192 * 4];
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    uint64_t acc = a;
    lds[threadIdx.x] = a;
    const uint64_t r0 = wall_clock64();
    for (uint32_t i = 0; i < iters; i++) {
        if (KIND == 0) {
#pragma unroll
            for (int k = 0; k < 64; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
        } else if (KIND == 1) {
#pragma unroll
            for (int k = 0; k < 64; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(b), "v"(a) : "vcc");
        } else if (KIND == 2) {
#pragma unroll
            for (int k = 0; k < 16; k++) { a = lds[(threadIdx.x + a) % 192]; lds[threadIdx.x + 192] = a; }
        } else if (KIND == 3) {
#pragma unroll
            for (int k = 0; k < 64; k++) asm volatile("s_nop 1\nv_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a));
        } else if (KIND == 4) {
#pragma unroll
            for (int k = 0; k < 8; k++) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); __syncthreads(); }
        } else if (KIND == 6) {     // the same dependent chain with ONE active lane per wavefront (sparse EXEC)
            if ((threadIdx.x & 63) == 0) {
#pragma unroll
                for (int k = 0; k < 64; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(b), "v"(a) : "vcc");
#pragma unroll
                for (int k = 0; k < 64; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
            }
        } else if (KIND == 7) {     // and with all lanes active
#pragma unroll
            for (int k = 0; k < 64; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(b), "v"(a) : "vcc");
#pragma unroll
            for (int k = 0; k < 64; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
        } else {                    // a loop body of 4.8 KB (600 x 8 bytes): does the instruction cache hold it?
#pragma unroll
            for (int k = 0; k < 600; k++) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a) : "v"(b));
        }
        __syncthreads();
    }
    const uint64_t r1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[1 + (threadIdx.x >> 6)] = hw;
        if (threadIdx.x == 0) out[0] = r1 - r0;
        out[4] = a + acc;
    }
}

template <int KIND>
static void run(const char *name, int between, uint64_t *d_out, uint64_t *d_tmp) {
    printf("%s, %d short kernel(s) between two long ones\n", name, between);
    for (int i = 0; i < 8; i++) {
        uint64_t h[5];
        for (int k = 0; k < between; k++) hipLaunchKernelGGL(k_work<0>, dim3(1), dim3(192), 0, 0, 10u, d_tmp, 7u);
        hipLaunchKernelGGL(k_work<KIND>, dim3(1), dim3(192), 0, 0, KIND == 5 ? 100u : 1000u, d_out, 12345u);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
        printf("  dispatch %2d: %7.1f us ", i, h[0] / 100.0);
        for (int w = 0; w < 3; w++) {
            const uint32_t hw = (uint32_t)h[1 + w];
            printf(" | se %u cu %2u simd %u", (hw >> 13) & 7, (hw >> 8) & 0xf, (hw >> 4) & 3);
        }
        printf("\n");
    }
}

int main(int argc, char **argv) {
    const int between = argc > 1 ? atoi(argv[1]) : 1;
    uint64_t *d_out, *d_tmp;
    (void)hipMalloc(&d_out, 64);
    (void)hipMalloc(&d_tmp, 64);
    run<0>("v_add_u32", between, d_out, d_tmp);
    run<1>("v_mad_u64_u32", between, d_out, d_tmp);
    run<2>("LDS round trips", between, d_out, d_tmp);
    run<3>("v_add_u32_dpp", between, d_out, d_tmp);
    run<4>("barriers", between, d_out, d_tmp);
    run<5>("4.8-KB loop body", between, d_out, d_tmp);
    run<6>("mads + adds, one active lane per wavefront", between, d_out, d_tmp);
    run<7>("mads + adds, all lanes", between, d_out, d_tmp);
    return 0;
}

// tools/ubench_branch.hip -- what a TAKEN branch costs a lone workgroup, per CU: the final exponentiation's kernel runs at
// 0.97 M cycles on some CUs and 1.3-1.45 M on others when it is alone on the chip, and at 0.93-0.98 M on every CU when
// all of them run it.  A loop of short blocks (16 v_add_u32 each) chained by s_branch, laid out so that consecutive
// blocks are `stride` blocks apart (stride 1: next block adjacent; larger: sequential prefetch is useless).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_branch.hip -o tools/ubench_branch.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// NB blocks of 16 adds; after block i control jumps to block (i + STRIDE) % NB (NB odd, STRIDE coprime)
#define BLOCK(i, nxt) "blk" #i "_%=:\n" ADDS "s_branch blk" #nxt "_%=\n"
#define ADDS "v_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\nv_add_u32 %0, %0, %1\n"

__global__ __launch_bounds__(192) void k_far(uint32_t iters, uint64_t *out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = 3;
    uint32_t n = iters;
    const uint64_t t0 = __builtin_readcyclecounter();
    // 8 blocks visited in the order 0 5 2 7 4 1 6 3 (stride 5): every transfer is a taken branch to a non-adjacent block
    asm volatile(
        "top_%=:\n"
        "s_branch blk0_%=\n"
        "blk0_%=:\n" ADDS "s_branch blk5_%=\n"
        "blk1_%=:\n" ADDS "s_branch blk6_%=\n"
        "blk2_%=:\n" ADDS "s_branch blk7_%=\n"
        "blk3_%=:\n" ADDS "s_branch end_%=\n"
        "blk4_%=:\n" ADDS "s_branch blk1_%=\n"
        "blk5_%=:\n" ADDS "s_branch blk2_%=\n"
        "blk6_%=:\n" ADDS "s_branch blk3_%=\n"
        "blk7_%=:\n" ADDS "s_branch blk4_%=\n"
        "end_%=:\n"
        "s_sub_u32 %2, %2, 1\n"
        "s_cmp_lg_u32 %2, 0\n"
        "s_cbranch_scc1 top_%=\n"
        : "+v"(a), "+v"(b), "+s"(n) : : "scc");
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[3 * blockIdx.x] = t1 - t0; out[3 * blockIdx.x + 1] = hw | ((uint64_t)(xcc & 15) << 32); out[3 * blockIdx.x + 2] = a;
    }
}
__global__ __launch_bounds__(192) void k_seq(uint32_t iters, uint64_t *out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = 3;
    uint32_t n = iters;
    const uint64_t t0 = __builtin_readcyclecounter();
    asm volatile(
        "top_%=:\n" ADDS ADDS ADDS ADDS ADDS ADDS ADDS ADDS
        "s_sub_u32 %2, %2, 1\n"
        "s_cmp_lg_u32 %2, 0\n"
        "s_cbranch_scc1 top_%=\n"
        : "+v"(a), "+v"(b), "+s"(n) : : "scc");
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[3 * blockIdx.x] = t1 - t0; out[3 * blockIdx.x + 1] = hw | ((uint64_t)(xcc & 15) << 32); out[3 * blockIdx.x + 2] = a;
    }
}

int main() {
    uint64_t *d_out;
    (void)hipMalloc(&d_out, 3 * 8 * 512);
    uint64_t h[3 * 512];
    const uint32_t iters = 20000;
    for (int pass = 0; pass < 2; pass++) {
        for (int grid : {1, 1, 1, 1, 1, 1, 1, 1, 16, 256}) {
            if (pass == 0) hipLaunchKernelGGL(k_seq, dim3(grid), dim3(192), 0, 0, iters, d_out, 12345u);
            else hipLaunchKernelGGL(k_far, dim3(grid), dim3(192), 0, 0, iters, d_out, 12345u);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, d_out, sizeof(uint64_t) * 3 * grid, hipMemcpyDeviceToHost);
            printf("%s grid %3d:", pass ? "8 taken branches per trip" : "straight line           ", grid);
            for (int i = 0; i < grid && i < 16; i++) printf(" [xcc %u cu %u: %.1f]", (unsigned)(h[3 * i + 1] >> 32), (unsigned)(h[3 * i + 1] >> 8) & 15, (double)h[3 * i] / iters);
            if (grid > 16) { double mn = 1e30, mx = 0; for (int i = 0; i < grid; i++) { double v = (double)h[3 * i] / iters; if (v < mn) mn = v; if (v > mx) mx = v; } printf(" ... min %.1f max %.1f", mn, mx); }
            printf("  cycles per trip (64 adds)\n");
        }
    }
    return 0;
}

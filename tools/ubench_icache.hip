// tools/ubench_icache.hip -- how much code a LONE workgroup can loop over before instruction fetch slows it down.
// One workgroup of 192 threads runs a loop whose body is N independent-ish VALU instructions (8 bytes each) with one
// barrier per trip; cycles per instruction against the size of the body.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_icache.hip -o tools/ubench_icache.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int N>
__global__ __launch_bounds__(192) void k_body(uint32_t iters, uint64_t *out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed ^ 5, d = seed + 9;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < N / 4; k++)
            asm volatile("v_add_u32_e64 %0, %0, %4\nv_add_u32_e64 %1, %1, %4\nv_add_u32_e64 %2, %2, %4\nv_add_u32_e64 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
        __syncthreads();
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = a + b + c + d; }
}

template <int N>
static void run(uint64_t *d_out) {
    const uint32_t iters = (1u << 22) / N;        // the same number of instructions for every size
    uint64_t h[2];
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_body<N>, dim3(1), dim3(192), 0, 0, iters, d_out, 12345u);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
        printf("body %7d instructions (%6.1f KB), %6u trips: %.2f cycles per instruction\n", N, N * 8 / 1024.0, iters, (double)h[0] / ((double)iters * N));
    }
}

int main() {
    uint64_t *d_out;
    (void)hipMalloc(&d_out, 64);
    run<128>(d_out); run<256>(d_out); run<512>(d_out); run<1024>(d_out); run<2048>(d_out); run<4096>(d_out); run<8192>(d_out); run<16384>(d_out);
    return 0;
}

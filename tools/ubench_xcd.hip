// tools/ubench_xcd.hip -- what does it cost on gfx950 to read, in kernel B, data that kernel A
// has just written from a DIFFERENT XCD (each of the 8 dies has its own L2; workgroup i runs on
// XCD i % 8)?  Producer: block i writes chunk i.  Consumer: block i reads chunk (i + shift) % nblk
// -- shift 0 = the die that wrote it, shift 1 = always another die -- or the column pattern of
// msm.hip's k_tile_scan_rows (every block reads a 64-byte piece of every chunk).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_xcd.hip -o ubench_xcd && ./ubench_xcd
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int FENCE>
__global__ __launch_bounds__(1024) void k_produce(uint32_t *buf, uint32_t words_per_chunk, uint32_t seed) {
    uint32_t *c = buf + (size_t)blockIdx.x * words_per_chunk;
    for (uint32_t x = threadIdx.x; x < words_per_chunk; x += 1024) c[x] = x * 2654435761u + seed + blockIdx.x;
    if (FENCE == 1) __threadfence();
    if (FENCE == 2) __threadfence_system();
}
__global__ __launch_bounds__(1024) void k_consume_rows(const uint32_t *buf, uint32_t words_per_chunk, uint32_t shift, uint32_t *out) {
    const uint32_t *c = buf + (size_t)((blockIdx.x + shift) % gridDim.x) * words_per_chunk;
    uint32_t s = 0;
    for (uint32_t x = threadIdx.x; x < words_per_chunk; x += 1024) s += c[x];
    if (s == 0x12345678u) out[0] = s;
}
// block b reads words [16 b, 16 b + 16) of every chunk (lanes: 16 words x 64 chunks per pass)
__global__ __launch_bounds__(1024) void k_consume_cols(const uint32_t *buf, uint32_t words_per_chunk, uint32_t nchunks, uint32_t *out) {
    const uint32_t w = blockIdx.x * 16 + (threadIdx.x & 15);
    uint32_t s = 0;
    for (uint32_t r = threadIdx.x >> 4; r < nchunks; r += 64) s += buf[(size_t)r * words_per_chunk + w];
    if (s == 0x12345678u) out[0] = s;
}


// ---- the k_tile_scan_rows pattern of msm.hip (256 rows x 8192 u16 counters -> u32 prefix rows)
__global__ __launch_bounds__(1024) void k_rows_hist(uint16_t *tile_hist, uint32_t Bc, uint32_t pitch, uint32_t seed) {
    uint16_t *th = tile_hist + (size_t)blockIdx.x * pitch;
    for (uint32_t x = threadIdx.x; x < Bc; x += 1024) th[x] = (uint16_t)((x * 7 + seed + blockIdx.x) & 15);
}
template <int MODE>   // 0: full; 1: no tile_base stores; 2: no loads
__global__ __launch_bounds__(1024) void k_rows_scan(const uint16_t *__restrict__ tile_hist, uint32_t Bc, uint32_t R, uint32_t pitch,
                                                    uint32_t *__restrict__ tile_base, uint32_t *__restrict__ hist) {
    __shared__ uint32_t part[32][33];
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned bl = lane & 31, grp = 2 * wv + (lane >> 5);
    const uint32_t b = blockIdx.x * 32 + bl;
    const uint32_t per = (R + 31) / 32, r0 = grp * per, r1 = r0 + per < R ? r0 + per : R;
    uint32_t v[16];
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) {
        const uint32_t r = r0 + j;
        v[j] = (MODE != 2 && b < Bc && r < r1) ? tile_hist[(size_t)r * pitch + b] : 1u;
    }
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) sum += v[j];
    part[grp][bl] = sum;
    __syncthreads();
    uint32_t run = 0, tot = 0;
#pragma unroll
    for (unsigned g2 = 0; g2 < 32; g2++) { uint32_t x = part[g2][bl]; if (g2 < grp) run += x; tot += x; }
    if (b < Bc) {
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) {
            const uint32_t r = r0 + j;
            if (r < r1) { if (MODE != 1) tile_base[(size_t)r * pitch + b] = run; run += v[j]; }
        }
        if (grp == 0) hist[b] = tot + (MODE == 1 ? run : 0);
    }
}
__global__ __launch_bounds__(1024) void k_rows_read(const uint32_t *tile_base, uint32_t Bc, uint32_t pitch, uint32_t *out) {
    const uint32_t *tb = tile_base + (size_t)blockIdx.x * pitch;
    uint32_t s = 0;
    for (uint32_t x = threadIdx.x; x < Bc; x += 1024) s += tb[x];
    if (s == 0x12345678u) out[0] = s;
}

int main() {
    const uint32_t nblk = 256;
    uint32_t *out;
    CHK(hipMalloc(&out, 4));
    hipEvent_t e0, e1, e2;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); CHK(hipEventCreate(&e2));
    for (uint32_t kb : {16u}) {                     // KiB per chunk: 4, 16, 64 MiB in all
        const uint32_t wpc = kb * 256;
        uint32_t *buf;
        CHK(hipMalloc(&buf, (size_t)nblk * wpc * 4));
        for (int fence = 0; fence < 3; fence++) {
            for (int mode = 0; mode < 4; mode++) {             // 0: same die, 1: next die, 2: columns, 3: same die after an unrelated kernel
                float tp = 0, tc = 0;
                const int reps = 10;
                for (int it = 0; it < reps + 2; it++) {
                    CHK(hipEventRecord(e0));
                    if (fence == 0) hipLaunchKernelGGL(k_produce<0>, dim3(nblk), dim3(1024), 0, 0, buf, wpc, (uint32_t)it);
                    else if (fence == 1) hipLaunchKernelGGL(k_produce<1>, dim3(nblk), dim3(1024), 0, 0, buf, wpc, (uint32_t)it);
                    else hipLaunchKernelGGL(k_produce<2>, dim3(nblk), dim3(1024), 0, 0, buf, wpc, (uint32_t)it);
                    CHK(hipEventRecord(e1));
                    if (mode == 2) hipLaunchKernelGGL(k_consume_cols, dim3(wpc / 16), dim3(1024), 0, 0, buf, wpc, nblk, out);
                    else hipLaunchKernelGGL(k_consume_rows, dim3(nblk), dim3(1024), 0, 0, buf, wpc, mode == 1 ? 1u : 0u, out);
                    CHK(hipEventRecord(e2));
                    CHK(hipEventSynchronize(e2));
                    float a, b;
                    CHK(hipEventElapsedTime(&a, e0, e1));
                    CHK(hipEventElapsedTime(&b, e1, e2));
                    if (it >= 2) { tp += a; tc += b; }
                }
                printf("chunk %4u KiB (total %3u MiB) fence %d mode %d: produce %8.1f us  consume %8.1f us\n", kb, nblk * kb / 1024, fence, mode,
                       tp / reps * 1e3, tc / reps * 1e3);
            }
        }
        CHK(hipFree(buf));
    }
    {
        const uint32_t Bc = 8192, R = 256, pitch = Bc + 96;
        uint16_t *th; uint32_t *tb, *hist;
        CHK(hipMalloc(&th, (size_t)R * pitch * 2)); CHK(hipMalloc(&tb, (size_t)R * pitch * 4)); CHK(hipMalloc(&hist, Bc * 4));
        for (int mode = 0; mode < 3; mode++) for (int reader = 0; reader < 2; reader++) {
            float t1 = 0, t2 = 0, t3 = 0;
            hipEvent_t e3; CHK(hipEventCreate(&e3));
            const int reps = 10;
            for (int it = 0; it < reps + 2; it++) {
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_rows_hist, dim3(R), dim3(1024), 0, 0, th, Bc, pitch, (uint32_t)it);
                CHK(hipEventRecord(e1));
                if (mode == 0) hipLaunchKernelGGL(k_rows_scan<0>, dim3(Bc / 32), dim3(1024), 0, 0, th, Bc, R, pitch, tb, hist);
                if (mode == 1) hipLaunchKernelGGL(k_rows_scan<1>, dim3(Bc / 32), dim3(1024), 0, 0, th, Bc, R, pitch, tb, hist);
                if (mode == 2) hipLaunchKernelGGL(k_rows_scan<2>, dim3(Bc / 32), dim3(1024), 0, 0, th, Bc, R, pitch, tb, hist);
                CHK(hipEventRecord(e2));
                if (reader) hipLaunchKernelGGL(k_rows_read, dim3(R), dim3(1024), 0, 0, tb, Bc, pitch, out);
                CHK(hipEventRecord(e3));
                CHK(hipEventSynchronize(e3));
                float a, b, c;
                CHK(hipEventElapsedTime(&a, e0, e1)); CHK(hipEventElapsedTime(&b, e1, e2)); CHK(hipEventElapsedTime(&c, e2, e3));
                if (it >= 2) { t1 += a; t2 += b; t3 += c; }
            }
            printf("rows: scan mode %d reader %d: hist %7.1f us  scan %7.1f us  read %7.1f us\n", mode, reader, t1 / reps * 1e3, t2 / reps * 1e3, t3 / reps * 1e3);
        }
    }
    return 0;
}

// Microbenchmark: throughput of 16-byte global stores whose addresses are NOT 16-byte (or even 4-byte)
// aligned, in the pattern a lanes-per-board emit would use (lane l of a 16-board wave writes 32
// consecutive bytes at board*121 + 32*l), against the aligned wave-contiguous pattern.
//   hipcc --offload-arch=gfx950 -O3 -o store_align store_align.hip && ./store_align
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st16(unsigned char *p, v4u v, bool nt)
{
    if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

// MODE 0: aligned, wave-contiguous: each wave writes SPAN bytes as 16-B vectors (lane i -> 16*i + 1024*it)
// MODE 1: per-lane 32-byte windows at board*C + 32*l (4 lanes per board, 16 boards per wave): two 16-B
//         stores per lane (+ the overlapping tail for the last word), byte-misaligned for most boards
// MODE 2: like 1 but the span is shifted so that every address is 4-byte aligned (C rounded to 124)
template <int MODE, int NT> __global__ __launch_bounds__(64) void k(unsigned char *out, int T, size_t row_bytes)
{
    const int lane = threadIdx.x & 63, wave = blockIdx.x;
    constexpr int C = MODE == 2 ? 124 : 121;
    const size_t span = 16 * C; // bytes per wave and row
    v4u v = {(u32)lane, (u32)wave, 0x01010101u, 0xFF00FF01u};
    for (int t = 0; t < T; ++t) {
        unsigned char *base = out + (size_t)t * row_bytes + (size_t)wave * span;
        if (MODE == 0) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int byte0 = (it * 64 + lane) * 16;
                if (byte0 < (int)span) st16(base + byte0, v, NT);
            }
        } else {
            const int b = lane >> 2, l = lane & 3;
            unsigned char *p = base + b * C + 32 * l;
            st16(p, v, NT);
            // last word of a board holds C-96 cells: second store overlaps backwards so it ends at the board end
            st16(l == 3 ? base + b * C + C - 16 : p + 16, v, NT);
        }
        v.x += 1;
    }
}

template <int MODE, int NT> static void run(unsigned char *buf, int waves, int T, size_t row_bytes, const char *name)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, NT>), dim3(waves), dim3(64), 0, 0, buf, T, row_bytes);
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MODE, NT>), dim3(waves), dim3(64), 0, 0, buf, T, row_bytes);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)waves * 16 * 121 * T * reps;
    printf("%-44s %8.3f ms/launch  %7.1f GB/s (useful bytes)\n", name, ms / reps, bytes / (ms * 1e-3) / 1e9);
}

int main()
{
    const int waves = 4096, T = 100; // 65 536 boards, 100 rows of 121 B per board
    const size_t row_bytes = (size_t)waves * 16 * 124;
    unsigned char *buf;
    if (hipMalloc(&buf, row_bytes * T + 4096) != hipSuccess) return 1;
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>(buf, waves, T, row_bytes, "aligned wave-contiguous 16B");
        run<0, 1>(buf, waves, T, row_bytes, "aligned wave-contiguous 16B nt");
        run<1, 0>(buf, waves, T, row_bytes, "per-lane 32B windows, byte-misaligned");
        run<1, 1>(buf, waves, T, row_bytes, "per-lane 32B windows, byte-misaligned nt");
        run<2, 0>(buf, waves, T, row_bytes, "per-lane 32B windows, 4B-aligned (C=124)");
        run<2, 1>(buf, waves, T, row_bytes, "per-lane 32B windows, 4B-aligned (C=124) nt");
    }
    (void)hipFree(buf);
    return 0;
}

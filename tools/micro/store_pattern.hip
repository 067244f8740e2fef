// Microbenchmark: what the write path sustains for the rollout's output pattern -- 1 024 workgroups, each streaming
// its own 64-board chunk (64 * C bytes, C = N*N) of TWO arrays (obs and mask rows [T][E][C]) per step -- against a
// plain fill of the same bytes, on several buffer sets (the level follows where the buffers lie).
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip && ./store_pattern [N]
// PAT 0: the helper's pattern: per step all obs pieces (1 KB per instruction), then all mask pieces, non-temporal
// PAT 1: same, plain stores            PAT 2: alternating obs / mask piece by piece, non-temporal
// PAT 3: one array of twice the width (the two rows of a board back to back: [T][E][2][C]), non-temporal
// PAT 4: pieces padded so that every workgroup's chunk starts on a 128-byte line (row stride 64*C rounded up)
// PACE > 0: every workgroup spins PACE shader cycles per step between its stores (a player's pace)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32;
typedef u32 v4u __attribute__((ext_vector_type(4)));

template <int PAT> __global__ __launch_bounds__(64) void k(unsigned char *obs, unsigned char *legal, int T, int C,
                                                           size_t row_bytes, int chunk, int pace)
{
    const int lane = threadIdx.x & 63;
    const int nb = gridDim.x, bi = blockIdx.x;
    const int blk = (nb & 7) == 0 ? (bi & 7) * (nb >> 3) + (bi >> 3) : bi;
    const int span = 64 * C;
    const int nit = (span + 1023) / 1024;
    v4u v = {(u32)lane, (u32)blk, 0x01010101u, 0xFF00FF01u};
    if (pace < 0) { // staggered start: workgroup b begins b / nb of a step period late, so the chip's writes sweep the row
        pace = -pace;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), d = (unsigned long long)pace * (unsigned)blk / (unsigned)nb;
        while (__builtin_amdgcn_s_memtime() - t0 < d) __builtin_amdgcn_s_sleep(2);
    }
    for (int t = 0; t < T; ++t) {
        if (PAT == 3) {
            unsigned char *b = obs + (size_t)t * row_bytes * 2 + (size_t)blk * chunk * 2;
            for (int it = 0; it < 2 * nit; ++it) {
                const int byte0 = (it * 64 + lane) * 16;
                if (byte0 < 2 * span) __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(b + byte0));
            }
        } else {
            unsigned char *bo = obs + (size_t)t * row_bytes + (size_t)blk * chunk;
            unsigned char *bl = legal + (size_t)t * row_bytes + (size_t)blk * chunk;
            if (PAT == 2) {
                for (int it = 0; it < nit; ++it) {
                    const int byte0 = (it * 64 + lane) * 16;
                    if (byte0 < span) {
                        __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(bo + byte0));
                        __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(bl + byte0));
                    }
                }
            } else if (PAT == 6) { // the step's pieces spread evenly over `pace` cycles (no burst)
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                int done_pieces = 0;
                for (int pass = 0; pass < 2; ++pass)
                    for (int it = 0; it < nit; ++it) {
                        const int byte0 = (it * 64 + lane) * 16;
                        if (byte0 < span) __builtin_nontemporal_store(v, reinterpret_cast<v4u *>((pass ? bl : bo) + byte0));
                        ++done_pieces;
                        const unsigned long long due = (unsigned long long)pace * done_pieces / (2 * nit);
                        while (__builtin_amdgcn_s_memtime() - t0 < due) __builtin_amdgcn_s_sleep(1);
                    }
            } else if (PAT == 5) { // ONE stream: the obs array only (half the bytes per step)
                for (int it = 0; it < nit; ++it) {
                    const int byte0 = (it * 64 + lane) * 16;
                    if (byte0 < span) __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(bo + byte0));
                }
            } else {
                for (int pass = 0; pass < 2; ++pass)
                    for (int it = 0; it < nit; ++it) {
                        const int byte0 = (it * 64 + lane) * 16;
                        if (byte0 < span) {
                            v4u *p = reinterpret_cast<v4u *>((pass ? bl : bo) + byte0);
                            if (PAT == 1) *p = v;
                            else __builtin_nontemporal_store(v, p);
                        }
                    }
            }
        }
        v.x += 1;
        if (pace > 0 && PAT != 6) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)pace) __builtin_amdgcn_s_sleep(2);
        }
    }
}

__global__ void k_fill(v4u *p, size_t n16)
{
    v4u v = {1u, 2u, 3u, 4u};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

template <int PAT> static double run(unsigned char *obs, unsigned char *legal, int T, int C, size_t row_bytes, int chunk, int pace)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<PAT>), dim3(1024), dim3(64), 0, 0, obs, legal, T, C, row_bytes, chunk, pace);
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<PAT>), dim3(1024), dim3(64), 0, 0, obs, legal, T, C, row_bytes, chunk, pace);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)1024 * 64 * C * 2 * T * reps / (ms * 1e-3) / 1e12; // TB/s of useful bytes
}

// sweep: obs at slab + X, mask at slab + X + D, for a grid of X (and a few D): which placement is the fast one?
static int sweep(int N, int T)
{
    const int C = N * N, chunk = 64 * C;
    const size_t row = (size_t)1024 * chunk, arr = (row * T + (1 << 21) - 1) >> 21 << 21;
    unsigned char *slab;
    const size_t slab_bytes = (size_t)6 << 30;
    if (hipMalloc(&slab, slab_bytes) != hipSuccess) return 1;
    printf("%dx%d, T %d: array %zu MB, slab at %p\n", N, N, T, arr >> 20, (void *)slab);
    const size_t steps[] = {0, 1 << 21, 1 << 22, 1 << 23, 1 << 24, 1 << 25, 1 << 26, 1 << 27, 1 << 28, 1 << 29, (size_t)1 << 30, ((size_t)1 << 30) + (1 << 27), (size_t)3 << 29};
    for (size_t D : {arr, arr + (1 << 21), arr + (1 << 24), arr + (1 << 27), ((arr >> 30) + 1) << 30})
        for (size_t X : steps) {
            if (X + D + arr > slab_bytes) continue;
            printf("X %5zu MB  D %5zu MB: %.2f TB/s\n", X >> 20, D >> 20, run<0>(slab + X, slab + X + D, T, C, row, chunk, 0));
        }
    return 0;
}

// many separately allocated buffer pairs, all kept: which allocations are the fast ones?
static int many(int N, int T, int count)
{
    const int C = N * N, chunk = 64 * C;
    const size_t row = (size_t)1024 * chunk;
    size_t arr = row * T + 4096;
    if (getenv("ROUND_MB")) { // allocation size rounded up to a multiple of ROUND_MB MiB
        const size_t m = (size_t)atoi(getenv("ROUND_MB")) << 20;
        arr = (arr + m - 1) / m * m;
    }
    printf("%dx%d, T %d: %zu MB per array, %d pairs\n", N, N, T, arr >> 20, count);
    for (int i = 0; i < count; ++i) {
        unsigned char *obs, *legal;
        if (hipMalloc(&obs, arr) != hipSuccess || hipMalloc(&legal, arr) != hipSuccess) {
            printf("out of memory at pair %d\n", i);
            break;
        }
        const double a = run<0>(obs, legal, T, C, row, chunk, 0);
        // the arrays crossed: this pair's obs with the previous pair's mask would need both kept; one array alone instead
        const double b = 0.5 * run<5>(obs, obs, T, C, row, chunk, 0), c = 0.5 * run<5>(legal, legal, T, C, row, chunk, 0);
        printf("pair %3d obs %p mask %p (mask - obs = %lld MB): both %.2f TB/s; obs alone %.2f, mask alone %.2f\n", i, (void *)obs, (void *)legal,
               (long long)(legal - obs) / (1 << 20), a, b, c);
    }
    return 0;
}

// does a staggered start (the writes sweep the row instead of landing all over it) change what the pattern sustains?
static int stagger(int N, int T)
{
    const int C = N * N, chunk = 64 * C;
    const size_t row = (size_t)1024 * chunk, arr = row * T + 4096;
    for (int i = 0; i < 3; ++i) {
        unsigned char *obs, *legal;
        if (hipMalloc(&obs, arr) != hipSuccess || hipMalloc(&legal, arr) != hipSuccess) return 1;
        printf("pair %d: unpaced %.2f TB/s; the step's pieces spread over P cycles (one step per P cycles = demand D):\n", i, run<0>(obs, legal, T, C, row, chunk, 0));
        for (int pace : {4000, 4400, 4800, 5200, 5600, 6000, 6500})
            printf("   P %d: %.2f TB/s", pace, run<6>(obs, legal, T, C, row, chunk, pace));
        printf("\n");
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 3 && atoi(argv[3]) < 0) return stagger(atoi(argv[1]), atoi(argv[2]));
    if (argc > 3) return many(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]));
    if (argc > 2) return sweep(atoi(argv[1]), atoi(argv[2]));
    const int N = argc > 1 ? atoi(argv[1]) : 11, C = N * N, T = 100;
    const int chunk = 64 * C, chunk_al = (chunk + 127) / 128 * 128;
    const size_t row = (size_t)1024 * chunk, row_al = (size_t)1024 * chunk_al;
    printf("%dx%d: chunk %d B per workgroup and array, %.1f MB per step\n", N, N, chunk, 2.0 * row / 1e6);
    for (int set = 0; set < 4; ++set) {
        unsigned char *obs, *legal, *pad;
        if (hipMalloc(&obs, 2 * row_al * T + 4096) != hipSuccess || hipMalloc(&legal, row_al * T + 4096) != hipSuccess) return 1;
        (void)hipMalloc(&pad, (size_t)(set + 1) * 37 * (1 << 20) + 4096 * set); // shifts the next set (kept: leaked on purpose)
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        const size_t n16 = row * T / 16;
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (v4u *)obs, 2 * n16);
        (void)hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (v4u *)obs, 2 * n16);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("set %d: plain fill %.2f TB/s | obs-then-mask nt %.2f, plain %.2f, alternating nt %.2f, one double-width array nt %.2f, line-aligned chunks nt %.2f"
               " | paced 5000 cycles/step: nt %.2f, aligned %.2f TB/s\n",
               set, 2.0 * row * T * 5 / (ms * 1e-3) / 1e12, run<0>(obs, legal, T, C, row, chunk, 0), run<1>(obs, legal, T, C, row, chunk, 0),
               run<2>(obs, legal, T, C, row, chunk, 0), run<3>(obs, legal, T, C, row, chunk, 0), run<0>(obs, legal, T, C, row_al, chunk_al, 0),
               run<0>(obs, legal, T, C, row, chunk, 5000), run<0>(obs, legal, T, C, row_al, chunk_al, 5000));
    }
    return 0;
}

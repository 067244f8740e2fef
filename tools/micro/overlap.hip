// overlap.hip -- can two consecutive launches of a chip-filling kernel overlap on MI355X, and how?
//   a) same stream, plain launches; b) same stream, hipExtAnyOrderLaunch; c) two streams.
// Each workgroup (128 threads, 24 000 B of LDS: the shape of k_lpb<11,...rollout1>) idles for 50 us + a per-block extra
// of up to 6.3 us and stamps its start and end with s_memrealtime (100 MHz, one clock for the chip).
// build: hipcc --offload-arch=gfx950 -O2 -o overlap tools/micro/overlap.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
__global__ __launch_bounds__(128) void busy(u64 *st, u64 *en, int ticks)
{
    __shared__ char lds[24000];
    lds[threadIdx.x] = 1;
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    const u64 want = (u64)ticks + (u64)((blockIdx.x * 37u) & 63u) * 10u;
    while (__builtin_amdgcn_s_memrealtime() - t0 < want) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        st[blockIdx.x] = t0;
        en[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    }
    if (lds[threadIdx.x] == 7) st[0] = 0;
}
int main()
{
    const int NB = 1024, R = 6;
    u64 *d;
    hipMalloc(&d, sizeof(u64) * NB * 2 * R);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    std::vector<u64> h(NB * 2 * R);
    for (int mode = 0; mode < 3; ++mode) {
        for (int warm = 0; warm < 2; ++warm) {
            hipMemset(d, 0, sizeof(u64) * NB * 2 * R);
            hipDeviceSynchronize();
            for (int r = 0; r < R; ++r) {
                u64 *st = d + (size_t)r * 2 * NB, *en = st + NB;
                if (mode == 0) hipLaunchKernelGGL(busy, dim3(NB), dim3(128), 0, s1, st, en, 5000);
                else if (mode == 1) hipExtLaunchKernelGGL(busy, dim3(NB), dim3(128), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, st, en, 5000);
                else hipLaunchKernelGGL(busy, dim3(NB), dim3(128), 0, (r & 1) ? s2 : s1, st, en, 5000);
            }
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), d, sizeof(u64) * NB * 2 * R, hipMemcpyDeviceToHost);
        u64 base = ~0ull;
        for (int b = 0; b < NB; ++b) base = std::min(base, h[b]);
        printf("mode %d (%s)\n", mode, mode == 0 ? "same stream" : mode == 1 ? "same stream, any-order flag" : "two streams");
        for (int r = 0; r < R; ++r) {
            std::vector<u64> st(h.begin() + (size_t)r * 2 * NB, h.begin() + (size_t)r * 2 * NB + NB);
            std::vector<u64> en(h.begin() + (size_t)r * 2 * NB + NB, h.begin() + (size_t)(r + 1) * 2 * NB);
            std::sort(st.begin(), st.end());
            std::sort(en.begin(), en.end());
            printf("  launch %d: first start %8.2f us  median start %8.2f  last start %8.2f | first end %8.2f  median end %8.2f  last end %8.2f\n",
                   r, (st[0] - base) * 0.01, (st[NB / 2] - base) * 0.01, (st[NB - 1] - base) * 0.01, (en[0] - base) * 0.01,
                   (en[NB / 2] - base) * 0.01, (en[NB - 1] - base) * 0.01);
        }
    }
    return 0;
}

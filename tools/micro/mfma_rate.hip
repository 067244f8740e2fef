// micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 from ONE wavefront under the instruction mixes of
// hexvec_policy.hip (hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o gpurun_out/mfma_rate)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V> __global__ __launch_bounds__(64) void k(float *out, unsigned long long *ticks, const float *wsrc, int reps)
{
    const int lane = threadIdx.x;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float w[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) w[i] = wsrc[i * 64 + lane];
    float x = out[lane], y = x * 0.5f;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (V == 0) { // two chains, nothing else
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * i], x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * i + 1], x, a1, 0, 0, 0);
            } else if (V == 1) { // four chains
                if (i & 1) {
                    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * i], x, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * i + 1], x, a3, 0, 0, 0);
                } else {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * i], x, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * i + 1], x, a1, 0, 0, 0);
                }
            } else if (V == 2) { // two chains + an independent 8-op VALU chain per pair
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * i], x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * i + 1], x, a1, 0, 0, 0);
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) y = __builtin_fmaf(y, 1.0001f, 0.5f);
            } else if (V == 3) { // B operand produced by a VALU chain (exp, rcp) right before the pair
                const float t = __expf(2.0f * y);
                const float xb = 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
                y = y + 0.001f;
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * i], xb, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * i + 1], xb, a1, 0, 0, 0);
            } else if (V == 4) { // one chain only
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * i], x, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * i + 1], x, a0, 0, 0, 0);
            } else if (V == 5) { // two chains + 24 independent VALU ops per pair (the sampling epilogue's share)
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * i], x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * i + 1], x, a1, 0, 0, 0);
#pragma unroll
                for (int k2 = 0; k2 < 24; ++k2) y = __builtin_fmaf(y, 1.0001f, 0.5f);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = y;
#pragma unroll
    for (int v = 0; v < 16; ++v) s += a0[v] + a1[v] + a2[v] + a3[v];
    out[blockIdx.x * 64 + lane] = s;
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int V> void run(const char *name, int blocks, float *out, unsigned long long *ticks, const float *w)
{
    const int reps = 50;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, out, ticks, w, reps);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, out, ticks, w, reps);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t = 0;
    (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double n = 64.0 * reps;
    printf("%-58s blocks %5d: %7.1f s_memtime ticks / MFMA, %7.1f ns / MFMA by events (kernel %.1f us)\n", name, blocks, t / n,
           ms * 1e6 / n, ms * 1e3);
}

int main()
{
    float *out, *w;
    unsigned long long *ticks;
    (void)hipMalloc(&out, 8192 * 64 * 4);
    (void)hipMalloc(&w, 64 * 64 * 4);
    (void)hipMalloc(&ticks, 8192 * 8);
    (void)hipMemset(out, 0, 8192 * 64 * 4);
    (void)hipMemset(w, 0, 64 * 64 * 4);
    for (int blocks : {1, 1024, 2048, 4096}) {
        run<0>("two chains", blocks, out, ticks, w);
        run<1>("four chains", blocks, out, ticks, w);
        run<4>("one chain", blocks, out, ticks, w);
        run<2>("two chains + 8 independent VALU per pair", blocks, out, ticks, w);
        run<5>("two chains + 24 independent VALU per pair", blocks, out, ticks, w);
        run<3>("two chains, B from an exp/rcp chain before the pair", blocks, out, ticks, w);
    }
    return 0;
}

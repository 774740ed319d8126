// Standalone probe: v_mfma_f64_16x16x4_f64 issue rate and in-kernel clock on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) probe(double* out, long long* clk, int iters)
{
    v4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4){0, 0, 0, 0};
    // pseudo-random operands (full mantissas): the clock the chip holds depends on the data
    unsigned long long z = (threadIdx.x + 1) * 0x9E3779B97F4A7C15ull + blockIdx.x * 0xD1B54A32D192ED03ull;
    z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    double a = (double)(z & 0xFFFFFFFFFFFFFull) / (double)0x10000000000000ull - 0.5;
    z *= 0x94D049BB133111EBull; z ^= z >> 31;
    double b = (double)(z & 0xFFFFFFFFFFFFFull) / (double)0x10000000000000ull - 0.5;
    if (iters < 0) { a = 1.0; b = 1.0; iters = -iters; }
    long long t0 = __builtin_amdgcn_s_memtime();
    long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
void run(int blocks_per_cu, int threads, int iters)
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int blocks = p.multiProcessorCount * blocks_per_cu;
    double* out; long long* clk;
    hipMalloc(&out, 8); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NACC><<<blocks, threads>>>(out, clk, 64);
    hipEventRecord(e0);
    probe<NACC><<<blocks, threads>>>(out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 2);
    hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
    double waves = (double)blocks * threads / 64;
    double flops = waves * (double)abs(iters) * NACC * 2048.0;
    double cyc = (double)h[0], real = (double)h[1];
    printf("nacc=%2d blocks/cu=%d threads=%4d  %.1f TF/s  cycles/mfma/wave=%.1f  clock=%.0f MHz  (%.2f ms)\n", NACC, blocks_per_cu, threads,
           flops / (ms * 1e-3) / 1e12, cyc / ((double)abs(iters) * NACC), cyc / real * 100.0, ms);
    hipFree(out); hipFree(clk);
}

int main()
{
    for (int rep = 0; rep < 2; ++rep) {
        run<16>(1, 256, 200000);
        run<16>(2, 256, 200000);
        run<16>(2, 256, -200000);
    }
    return 0;
}

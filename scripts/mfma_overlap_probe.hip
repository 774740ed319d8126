// Standalone probe: how many independent VALU / LDS instructions of the same wavefront issue in the shadow of a
// v_mfma_f64_16x16x4_f64 (one wavefront per SIMD, gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4 __attribute__((ext_vector_type(4)));
typedef double v2 __attribute__((ext_vector_type(2)));

template <int NV, int KIND>
__global__ void __launch_bounds__(1024) probe(double* out, long long* clk, int iters)
{
    __shared__ double sm[4096];
    v4 acc0 = {0, 0, 0, 0};
    v2 acc1 = {0, 0};
    double a = threadIdx.x * 0.001 + 1.0, b = 0.5 + blockIdx.x * 1e-6;
    unsigned x0 = threadIdx.x, x1 = 3, x2 = 5, x3 = 7;
    double d0 = a, d1 = b;
    sm[threadIdx.x] = a;
    __syncthreads();
    double l0 = 0, l1 = 0;
    const double* lp = sm + (threadIdx.x & 63) * 2;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (KIND == 0) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x0) : "v"(x1), "v"(x2));
            if (KIND == 1) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x0) : "v"(x1), "v"(x2));
            if (KIND == 2) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d0) : "v"(d1), "v"(a));
            if (KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(acc1) : "v"((unsigned)(size_t)lp));
            if (KIND == 4) asm volatile("v_lshl_add_u64 %0, %1, 3, %2" : "=v"(d0) : "v"(d1), "v"(a));
        }
        if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = acc0[0] + acc1[0] + x0 + x3 + d0 + l0 + l1;
    if (s == 12345.678) out[0] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NV, int KIND>
void run(const char* name, int threads = 256)
{
    double* out; long long* clk;
    hipMalloc(&out, 8); hipMalloc(&clk, 256 * 8);
    const int iters = 20000;
    probe<NV, KIND><<<256, threads>>>(out, clk, 64);
    probe<NV, KIND><<<256, threads>>>(out, clk, iters);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    printf("%-14s x %2d per mfma, %d wavefronts per SIMD: %.1f cycles per iteration\n", name, NV, threads / 256, (double)h[0] / iters);
    hipFree(out); hipFree(clk);
}

int main()
{
    run<0, 0>("none");
    run<4, 0>("v_add_u32"); run<8, 0>("v_add_u32"); run<12, 0>("v_add_u32"); run<16, 0>("v_add_u32"); run<24, 0>("v_add_u32");
    run<4, 1>("v_cndmask"); run<8, 1>("v_cndmask"); run<16, 1>("v_cndmask");
    run<2, 2>("v_mul_f64"); run<4, 2>("v_mul_f64"); run<8, 2>("v_mul_f64");
    run<2, 3>("ds_read_b128"); run<4, 3>("ds_read_b128"); run<8, 3>("ds_read_b128");
    run<4, 4>("v_lshl_add_u64"); run<8, 4>("v_lshl_add_u64");
    printf("two wavefronts per SIMD:\n");
    run<0, 0>("none", 512);
    run<4, 0>("v_add_u32", 512); run<8, 0>("v_add_u32", 512); run<16, 0>("v_add_u32", 512); run<24, 0>("v_add_u32", 512);
    run<4, 2>("v_mul_f64", 512); run<8, 2>("v_mul_f64", 512);
    run<4, 3>("ds_read_b128", 512); run<8, 3>("ds_read_b128", 512);
    run<0, 0>("none", 1024);
    run<8, 0>("v_add_u32", 1024); run<16, 0>("v_add_u32", 1024);
    return 0;
}

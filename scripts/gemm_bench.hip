// Standalone timing of the library's fp64 MFMA GEMM kernel (no Python in the loop).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../slam_amd/csrc/gemm.hip.h"
using namespace slamk;

template <bool TB>
double run(int M, int N, int K, int batch, int reps)
{
    size_t sa = (size_t)M * K, sb = (size_t)N * K, sc = (size_t)M * N;
    double *A, *B, *C;
    hipMalloc(&A, sa * batch * 8); hipMalloc(&B, sb * batch * 8); hipMalloc(&C, sc * batch * 8);
    std::vector<double> h(std::max(sa, sb) * batch);
    for (auto& x : h) x = (double)rand() / RAND_MAX - 0.5;
    hipMemcpy(A, h.data(), sa * batch * 8, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), sb * batch * 8, hipMemcpyHostToDevice);
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldb = TB ? K : N; g.ldc = N;
    g.sA1 = sa; g.sB1 = sb; g.sC1 = sc; g.nb2 = 1; g.alpha = 1; g.beta = 0;
    launch_gemm<TB, EPI_NONE, false, TRI_NONE>(g, batch, 0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch_gemm<TB, EPI_NONE, false, TRI_NONE>(g, batch, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double tf = 2.0 * M * N * K * batch * reps / (ms * 1e-3) / 1e12;
    printf("%s M=%5d N=%5d K=%5d batch=%4d : %8.3f ms  %6.2f TF/s (%.1f%% of 78.6)\n", TB ? "NT" : "NN", M, N, K, batch,
           ms / reps, tf, tf / 78.6 * 100);
    hipFree(A); hipFree(B); hipFree(C);
    return tf;
}

int main()
{
    run<true>(4096, 4096, 4096, 1, 3);
    run<false>(4096, 4096, 4096, 1, 3);
    run<true>(8192, 8192, 2048, 1, 2);
    run<true>(6016, 6016, 6016, 1, 2);
    run<false>(6016, 10016, 6016, 1, 1);
    run<true>(112, 112, 112, 4096, 5);
    run<false>(112, 112, 112, 4096, 5);
    run<true>(112, 176, 112, 4096, 5);
    run<false>(176, 176, 112, 4096, 5);
    run<true>(128, 128, 128, 4096, 5);
    run<true>(64, 64, 64, 16384, 5);
    run<true>(64, 64, 512, 4096, 5);
    return 0;
}

// One GEMM shape, few launches: for rocprofv3 --pmc passes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../slam_amd/csrc/gemm.hip.h"
using namespace slamk;
int main(int argc, char** argv)
{
    int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    int batch = argc > 4 ? atoi(argv[4]) : 1;
    size_t sa = (size_t)M * K, sb = (size_t)N * K, sc = (size_t)M * N;
    double *A, *B, *C;
    hipMalloc(&A, sa * batch * 8); hipMalloc(&B, sb * batch * 8); hipMalloc(&C, sc * batch * 8);
    std::vector<double> h(std::max(sa, sb) * batch);
    for (auto& x : h) x = (double)rand() / RAND_MAX - 0.5;
    hipMemcpy(A, h.data(), sa * batch * 8, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), sb * batch * 8, hipMemcpyHostToDevice);
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldb = K; g.ldc = N; g.sA1 = sa; g.sB1 = sb; g.sC1 = sc; g.nb2 = 1; g.alpha = 1; g.beta = 0;
    for (int r = 0; r < 3; ++r) launch_gemm<true, EPI_NONE, false, TRI_NONE>(g, batch, 0);
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}

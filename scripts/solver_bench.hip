// Standalone timing of gain_solver_kernel<11> on 4096 filters (ablation macros select what runs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../slam_amd/csrc/gain_solver.hip.h"
using namespace slamk;
int main()
{
    const int B = 4096, MT = 11, NT = 7, ldm = 208, ldn = 112, m = 16 * MT;
    const long long sM = (long long)ldm * ldm, sNM = (long long)ldn * ldm;
    std::vector<double> hS(sM, 0.0), hW(sNM), hr(ldm, 1.0);
    for (int i = 0; i < m; ++i)
        for (int j = 0; j <= i; ++j) {
            double v = (i == j) ? 50.0 + i : 0.3 * ((i * 7 + j * 13) % 11) / 11.0;
            hS[(long long)i * ldm + j] = v; hS[(long long)j * ldm + i] = v;
        }
    for (auto& x : hW) x = (double)rand() / RAND_MAX - 0.5;
    double *S, *W, *r, *K, *KR;
    hipMalloc(&S, sM * B * 8); hipMalloc(&W, sNM * B * 8); hipMalloc(&K, sNM * B * 8); hipMalloc(&KR, sNM * B * 8);
    hipMalloc(&r, (size_t)ldm * B * 8);
    for (int b = 0; b < B; ++b) {
        hipMemcpy(S + b * sM, hS.data(), sM * 8, hipMemcpyHostToDevice);
        hipMemcpy(W + b * sNM, hW.data(), sNM * 8, hipMemcpyHostToDevice);
        hipMemcpy(r + (size_t)b * ldm, hr.data(), ldm * 8, hipMemcpyHostToDevice);
    }
    const size_t lds = ((size_t)MT * (MT + 1) / 2 * 256 + 256) * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(gain_solver_kernel<MT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        gain_solver_kernel<MT><<<B, 512, lds>>>(S, W, r, K, KR, nullptr, ldm, sM, sNM, NT);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms\n", VARIANT, ms);
    }
    return 0;
}

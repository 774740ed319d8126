// Standalone timing of gain_solver_kernel<10, true> on 4096 filters (ablation macros select what runs).
// -DSB_MT / -DSB_NT / -DSB_BORDER / -DSB_SPILL / -DSB_WIDE / -DSB_B select another instantiation and batch;
// -DSB_HEADER="path" times another copy of the kernel source.
// H is all zero here, so S = R (diagonal): the instruction stream is the same as for real data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifndef SB_STAMP_BLOCK
#define SB_STAMP_BLOCK 2000
#endif
#ifdef STAMPS
__device__ unsigned long long g_stamps[8][32];
#define SOLVER_STAMP(i) do { if (blockIdx.x == SB_STAMP_BLOCK && (threadIdx.x & 63) == 0) g_stamps[threadIdx.x >> 6][i] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#ifndef SB_HEADER
#define SB_HEADER "../slam_amd/csrc/gain_solver.hip.h"
#endif
#include SB_HEADER
#ifndef VARIANT
#define VARIANT "solver"
#endif
#ifndef SB_MT
#define SB_MT 10
#endif
#ifndef SB_NT
#define SB_NT 7
#endif
#ifndef SB_BORDER
#define SB_BORDER true
#endif
#ifndef SB_SPILL
#define SB_SPILL false
#endif
#ifndef SB_B
#define SB_B 4096
#endif
#ifdef SB_WIDE
#define SB_KERNEL gain_solver_kernel<MT, BORDER, true, true>
#define SB_LDS SolverLds<MT, true, true>
#else
#define SB_KERNEL gain_solver_kernel<MT, BORDER, SB_SPILL>
#define SB_LDS SolverLds<MT, SB_SPILL>
#endif
using namespace slamk;
template <typename T> T* dev(const std::vector<T>& h, int B)
{
    T* d; hipMalloc(&d, h.size() * sizeof(T) * B);
    for (int b = 0; b < B; ++b) hipMemcpy(d + b * h.size(), h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}
int main()
{
    constexpr int B = SB_B, MT = SB_MT, NT = SB_NT, ldm = 16 * MT + 48, ldn = 16 * NT;
    constexpr bool BORDER = SB_BORDER;
    const long long sM = (long long)ldm * ldm, sNM = (long long)ldn * ldm;
    std::vector<double> hW((size_t)ldn * ldn), hr(ldm), hz(ldm * 3, 0.0);
    std::vector<int> hcs(ldm, -1), hm(1, 16 * MT + 1);
    for (int i = 0; i < ldm; ++i) hr[i] = 50.0 + i;
    for (auto& x : hW) x = (double)rand() / RAND_MAX - 0.5;
    FilterView fv;
    memset(&fv, 0, sizeof fv);
    fv.B = B; fv.F = 32; fv.ldn = ldn; fv.ldm = ldm;
    fv.ha = dev(hz, B); fv.hb = dev(hz, B); fv.hcs = dev(hcs, B); fv.rdiag = dev(hr, B);
    fv.P = dev(hW, B); fv.m = dev(hm, B);
    { std::vector<int> st(1, 0); fv.status = dev(st, B); }
    double* K; hipMalloc(&K, sNM * B * 8);
    const size_t lds = (size_t)SB_LDS::TILES * 256 * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(SB_KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    double* Lsp = nullptr;
    const long long sSp = ((long long)MT * (MT + 1) / 2 + 1) * 256;
    hipMalloc(&Lsp, sSp * B * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 3;
    for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0);
        SB_KERNEL<<<B, 512, lds>>>(fv, K, nullptr, nullptr, sM, sNM, NT, Lsp, sSp);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms\n", VARIANT, ms);
#ifdef SB_WIDE
        hipEventRecord(e0);
        const size_t lr = (size_t)gain_rows_lds_tiles(MT) * 256 * 8;
        hipFuncSetAttribute(reinterpret_cast<const void*>(gain_rows_kernel<MT, BORDER>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lr);
        gain_rows_kernel<MT, BORDER><<<B, 256, lr>>>(fv, K, nullptr, sNM, NT, 0, 1, Lsp, sSp);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s rows 8..: %.3f ms (%s)\n", VARIANT, ms, hipGetErrorString(hipGetLastError()));
#endif
    }
#ifdef STAMPS
    unsigned long long hs[8][32];
    hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_stamps), sizeof hs);
    for (int w = 0; w < 8; ++w) {
        printf("wave %d:", w);
        for (int i = 1; i < 23; ++i) if (hs[w][i]) printf(" [%d]%lld", i, (long long)(hs[w][i] - hs[w][0])); 
        printf("\n");
    }
#endif
    return 0;
}

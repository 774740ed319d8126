// Standalone timing of joseph_kernel<7> on 4096 filters of 32 features (ablation macros select what runs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef STAMPS
__device__ unsigned long long g_stamps[8][32];
#define JOSEPH_STAMP(i) do { if (blockIdx.x == 2000 && (threadIdx.x & 63) == 0) g_stamps[threadIdx.x >> 6][i] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#include "../slam_amd/csrc/feature_kernels.hip.h"
#include "../slam_amd/csrc/joseph_kernel.hip.h"
using namespace slamk;
template <typename T> T* dev(const std::vector<T>& h, int B)
{
    T* d; hipMalloc(&d, h.size() * sizeof(T) * B);
    for (int b = 0; b < B; ++b) hipMemcpy(d + b * h.size(), h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}
int main()
{
    const int B = 4096, NT = 7, F = 32, nf = 32, ldn = 112, ldm = 208, ml = 160;      // ml: columns of K walked in chunks; altitude column = 160
    std::vector<double> P((size_t)ldn * ldn), K((size_t)ldn * ldm), ha(ldm * 3), hb(ldm * 3), rd(ldm, 1.0);
    std::vector<int> rowoff(F), isref(F), nfv(1, nf), mv(1, 161);
    for (auto& x : P) x = (double)rand() / RAND_MAX;
    for (auto& x : K) x = ((double)rand() / RAND_MAX - 0.5) * 0.01;
    for (auto& x : ha) x = (double)rand() / RAND_MAX;
    for (auto& x : hb) x = (double)rand() / RAND_MAX;
    int row = 0;
    for (int i = 0; i < F; ++i) { isref[i] = i & 1; rowoff[i] = row; row += isref[i] ? 6 : 4; }
    FilterView fv;
    memset(&fv, 0, sizeof fv);
    fv.B = B; fv.F = F; fv.ldn = ldn; fv.ldm = ldm;
    fv.P = dev(P, B); fv.ha = dev(ha, B); fv.hb = dev(hb, B); fv.rdiag = dev(rd, B);
    fv.rowoff = dev(rowoff, B); fv.isref = dev(isref, B); fv.nf = dev(nfv, B); fv.m = dev(mv, B);
    double* Kd = dev(K, B);
    fv.e = dev(rd, B);
    double* dxd; hipMalloc(&dxd, (size_t)B * ldn * 8);
    const size_t lds = (size_t)(NT * NT + (NT <= 7 ? 2 : 1) * NT + 2) * 256 * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(joseph_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        joseph_kernel<NT><<<B, 512, lds>>>(fv, Kd, ldm, (long long)ldn * ldm, ml, 0, dxd, 0, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms\n", VARIANT, ms);
    }
#ifdef STAMPS
    unsigned long long hs[8][32];
    hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_stamps), sizeof hs);
    for (int w = 0; w < 8; ++w) {
        printf("wave %d:", w);
        for (int i = 1; i < 12; ++i) if (hs[w][i]) printf(" [%d]%lld", i, (long long)(hs[w][i] - hs[w][0]));
        printf("\n");
    }
#endif
    return 0;
}

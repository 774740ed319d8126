// Standalone timing of joseph_kernel<7> on 4096 filters of 32 features (ablation macros select what runs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#ifdef STAMPS
__device__ unsigned long long g_stamps[8][32];
#define JOSEPH_STAMP(i) do { if (blockIdx.x == 2000 && (threadIdx.x & 63) == 0) g_stamps[threadIdx.x >> 6][i] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#include "../slam_amd/csrc/feature_kernels.hip.h"
#include "../slam_amd/csrc/joseph_kernel.hip.h"
#include "../slam_amd/csrc/joseph_quad.hip.h"
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
    std::vector<int> rowoff(F), isref(F), nfv(1, nf), mv(1, 161), hcs(ldm, -1);
    for (auto& x : P) x = (double)rand() / RAND_MAX;
    for (int i = 0; i < ldn; ++i) for (int j = 0; j < i; ++j) P[(size_t)j * ldn + i] = P[(size_t)i * ldn + j];   // symmetric
    for (auto& x : K) x = ((double)rand() / RAND_MAX - 0.5) * 0.01;
    for (auto& x : ha) x = (double)rand() / RAND_MAX;
    for (auto& x : hb) x = (double)rand() / RAND_MAX;
    int row = 0;
    for (int i = 0; i < F; ++i) {
        isref[i] = i & 1; rowoff[i] = row;
        for (int t = 0; t < (isref[i] ? 6 : 4); ++t) hcs[row + t] = 6 + 3 * i;
        row += isref[i] ? 6 : 4;
    }
    for (int t = 0; t < 3; ++t) hb[160 * 3 + t] = 0.0;
    FilterView fv;
    memset(&fv, 0, sizeof fv);
    fv.B = B; fv.F = F; fv.ldn = ldn; fv.ldm = ldm;
    fv.P = dev(P, B); fv.ha = dev(ha, B); fv.hb = dev(hb, B); fv.rdiag = dev(rd, B);
    fv.hcs = dev(hcs, B);
    fv.rowoff = dev(rowoff, B); fv.isref = dev(isref, B); fv.nf = dev(nfv, B); fv.m = dev(mv, B);
    double* Kd = dev(K, B);
    std::vector<double> ev(ldm); for (auto& x : ev) x = (double)rand() / RAND_MAX - 0.5;
    fv.e = dev(ev, B);
    double* dxd; hipMalloc(&dxd, (size_t)B * ldn * 8);
#ifdef QUAD
#ifndef QGRID
#define QGRID B
#endif
    const size_t lq = (size_t)(NT * NT + 3 * NT + 10) * 256 * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(joseph_quad_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lq);
    hipFuncSetAttribute(reinterpret_cast<const void*>(joseph_quad_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lq);
    {   // against joseph_kernel on the same input
        const size_t l0 = (size_t)(NT * NT + 2 * NT + 2) * 256 * 8;
        hipFuncSetAttribute(reinterpret_cast<const void*>(joseph_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l0);
        for (int form = 0; form < 2; ++form) {
            std::vector<double> r0(P.size()), r1(P.size()), d0(ldn), d1(ldn);
            hipMemcpy(fv.P, P.data(), P.size() * 8, hipMemcpyHostToDevice);
            joseph_kernel<NT><<<1, 512, l0>>>(fv, Kd, ldm, (long long)ldn * ldm, ml, form, dxd, 0, nullptr);
            hipMemcpy(r0.data(), fv.P, P.size() * 8, hipMemcpyDeviceToHost);
            hipMemcpy(d0.data(), dxd, ldn * 8, hipMemcpyDeviceToHost);
            hipMemcpy(fv.P, P.data(), P.size() * 8, hipMemcpyHostToDevice);
            if (form) joseph_quad_kernel<true><<<1, 256, lq>>>(fv, Kd, ldm, (long long)ldn * ldm, ml, dxd, 0, nullptr);
            else joseph_quad_kernel<false><<<1, 256, lq>>>(fv, Kd, ldm, (long long)ldn * ldm, ml, dxd, 0, nullptr);
            hipMemcpy(r1.data(), fv.P, P.size() * 8, hipMemcpyDeviceToHost);
            hipMemcpy(d1.data(), dxd, ldn * 8, hipMemcpyDeviceToHost);
            double worst = 0, wd = 0, big = 0;
            for (size_t i = 0; i < r0.size(); ++i) { worst = fmax(worst, fabs(r0[i] - r1[i])); big = fmax(big, fabs(r0[i])); }
            for (int i = 0; i < ldn; ++i) wd = fmax(wd, fabs(d0[i] - d1[i]));
            printf("form %d: max |P_old - P_quad| = %.3e (max |P| %.3e), max |dx diff| = %.3e (%s)\n", form, worst, big, wd, hipGetErrorString(hipGetLastError()));
        }
        hipMemcpy(fv.P, P.data(), P.size() * 8, hipMemcpyHostToDevice);
    }
    hipEvent_t q0, q1; hipEventCreate(&q0); hipEventCreate(&q1);
    {
        float best = 1e9f, sum = 0.f;
        const int reps = 12;
        for (int rep = 0; rep < reps; ++rep) {
            hipEventRecord(q0);
            joseph_quad_kernel<false><<<QGRID, 256, lq>>>(fv, Kd, ldm, (long long)ldn * ldm, ml, dxd, 0, nullptr);
            hipEventRecord(q1); hipEventSynchronize(q1);
            float ms; hipEventElapsedTime(&ms, q0, q1);
            if (rep >= 2) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("%s: best %.3f ms, mean %.3f ms over %d (%s)\n", VARIANT, best, sum / (reps - 2), reps - 2, hipGetErrorString(hipGetLastError()));
    }
    {
        unsigned long long hs[8][32];
        hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_stamps), sizeof hs);
        for (int w = 0; w < 4; ++w) {
            printf("wave %d:", w);
            for (int i = 1; i < 32; ++i) if (hs[w][i]) printf(" [%d]%lld", i, (long long)(hs[w][i] - hs[w][0]));
            printf("\n");
        }
    }
    return 0;
#endif
#ifdef MERGED
    auto kern = joseph_merged_kernel<NT>;
    const size_t lds = (size_t)(NT * NT + 2 * NT + 10) * 256 * 8;
#else
    auto kern = joseph_kernel<NT>;
    const size_t lds = (size_t)(NT * NT + (NT <= 7 ? 2 : 1) * NT + 2) * 256 * 8;
#endif
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#ifdef COMPARE
    {   // both kernels on the same input: largest difference of the results
        const size_t l0 = (size_t)(NT * NT + 2 * NT + 2) * 256 * 8, l1 = (size_t)(NT * NT + 2 * NT + 10) * 256 * 8;
        hipFuncSetAttribute(reinterpret_cast<const void*>(joseph_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l0);
        hipFuncSetAttribute(reinterpret_cast<const void*>(joseph_merged_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1);
        for (int form = 0; form < 2; ++form) {
            std::vector<double> r0(P.size()), r1(P.size()), d0(ldn), d1(ldn);
            hipMemcpy(fv.P, P.data(), P.size() * 8, hipMemcpyHostToDevice);
            joseph_kernel<NT><<<1, 512, l0>>>(fv, Kd, ldm, (long long)ldn * ldm, ml, form, dxd, 0, nullptr);
            hipMemcpy(r0.data(), fv.P, P.size() * 8, hipMemcpyDeviceToHost);
            hipMemcpy(d0.data(), dxd, ldn * 8, hipMemcpyDeviceToHost);
            hipMemcpy(fv.P, P.data(), P.size() * 8, hipMemcpyHostToDevice);
            joseph_merged_kernel<NT><<<1, 512, l1>>>(fv, Kd, ldm, (long long)ldn * ldm, ml, form, dxd, 0, nullptr);
            hipMemcpy(r1.data(), fv.P, P.size() * 8, hipMemcpyDeviceToHost);
            hipMemcpy(d1.data(), dxd, ldn * 8, hipMemcpyDeviceToHost);
            double worst = 0, wd = 0, big = 0;
            for (size_t i = 0; i < r0.size(); ++i) { worst = fmax(worst, fabs(r0[i] - r1[i])); big = fmax(big, fabs(r0[i])); }
            for (int i = 0; i < ldn; ++i) wd = fmax(wd, fabs(d0[i] - d1[i]));
            printf("form %d: max |P_old - P_merged| = %.3e (max |P| %.3e), max |dx diff| = %.3e (%s)\n", form, worst, big, wd, hipGetErrorString(hipGetLastError()));
        }
        hipMemcpy(fv.P, P.data(), P.size() * 8, hipMemcpyHostToDevice);
    }
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int form = getenv("FORM") ? atoi(getenv("FORM")) : 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        kern<<<B, 512, lds>>>(fv, Kd, ldm, (long long)ldn * ldm, ml, form, dxd, 0, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms\n", VARIANT, ms);
    }
#ifdef STAMPS
    unsigned long long hs[8][32];
    hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_stamps), sizeof hs);
    for (int w = 0; w < 8; ++w) {
        printf("wave %d:", w);
        for (int i = 1; i < 20; ++i) if (hs[w][i]) printf(" [%d]%lld", i, (long long)(hs[w][i] - hs[w][0]));
        printf("\n");
    }
#endif
    return 0;
}

// Standalone timing of seq_update_kernel<7> on 4096 filters of 32 features (cycle stamps of one workgroup with -DSTAMPS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#ifdef STAMPS
__device__ unsigned long long g_stamps[8][32];
#define SEQ_STAMP(i) do { if (blockIdx.x == 2000 && (threadIdx.x & 63) == 0) g_stamps[threadIdx.x >> 6][i] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#include "../slam_amd/csrc/feature_kernels.hip.h"
#include "../slam_amd/csrc/seq_update.hip.h"
using namespace slamk;
template <typename T> T* dev(const std::vector<T>& h, int B)
{
    T* d; (void)hipMalloc(&d, h.size() * sizeof(T) * B);
    for (int b = 0; b < B; ++b) (void)hipMemcpy(d + b * h.size(), h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}
int main()
{
    const int B = 4096, NT = 7, F = 32, nf = 32, n = 105, ldn = 112, ldm = 208, m = 161;
    std::vector<double> P((size_t)ldn * ldn, 0.0), ha(ldm * 3, 0.0), hb(ldm * 3, 0.0), rd(ldm, 1.0), ev(ldm, 0.0);
    std::vector<int> nfv(1, nf), mv(1, m), hcs(ldm, -1);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            const double v = (i == j) ? 1.0 + 0.1 * rand() / RAND_MAX : 0.002 * ((double)rand() / RAND_MAX - 0.5);
            P[(size_t)i * ldn + j] = P[(size_t)j * ldn + i] = v;
        }
    int row = 0;
    for (int i = 0; i < F; ++i) {
        const int cnt = (i & 1) ? 6 : 4;
        for (int t = 0; t < cnt; ++t) {
            hcs[row + t] = 6 + 3 * i;
            for (int k = 0; k < 3; ++k) { ha[3 * (row + t) + k] = (double)rand() / RAND_MAX - 0.5; hb[3 * (row + t) + k] = (double)rand() / RAND_MAX - 0.5; }
            ev[row + t] = 0.01 * ((double)rand() / RAND_MAX - 0.5);
            rd[row + t] = 0.05;
        }
        row += cnt;
    }
    ha[3 * 160 + 2] = -1.0; ev[160] = 0.01; rd[160] = 0.1;      // altitude row
    FilterView fv;
    memset(&fv, 0, sizeof fv);
    fv.B = B; fv.F = F; fv.ldn = ldn; fv.ldm = ldm; fv.gate = 4;
    fv.P = dev(P, B); fv.ha = dev(ha, B); fv.hb = dev(hb, B); fv.rdiag = dev(rd, B); fv.hcs = dev(hcs, B); fv.e = dev(ev, B);
    fv.nf = dev(nfv, B); fv.m = dev(mv, B);
    double* dxd; (void)hipMalloc(&dxd, (size_t)B * ldn * 8);
    const size_t lds = (size_t)SeqUpdateLds<NT>::TILES * 256 * 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(seq_update_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        for (int b = 0; b < B; b += 512) (void)hipMemcpy(fv.P + (size_t)b * ldn * ldn, fv.P, 0, hipMemcpyDeviceToDevice);
        (void)hipEventRecord(e0);
        seq_update_kernel<NT><<<B, 512, lds>>>(fv, dxd, 0, nullptr);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms (%s)\n", VARIANT, ms, hipGetErrorString(hipGetLastError()));
    }
    std::vector<double> dx(ldn), Pout((size_t)ldn * ldn);
    (void)hipMemcpy(dx.data(), dxd, ldn * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(Pout.data(), fv.P, Pout.size() * 8, hipMemcpyDeviceToHost);
    printf("dx[0..3] = %.6e %.6e %.6e %.6e   P[0][0] = %.6e  P[50][50] = %.6e\n", dx[0], dx[1], dx[2], dx[3], Pout[0], Pout[50 * ldn + 50]);
#ifdef STAMPS
    unsigned long long hs[8][32];
    (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_stamps), sizeof hs);
    for (int w = 0; w < 8; w += 7) {
        printf("wave %d:", w);
        for (int i = 1; i < 24; ++i) if (hs[w][i]) printf(" [%d]%lld", i, (long long)(hs[w][i] - hs[w][0]));
        printf("\n");
    }
#endif
    return 0;
}

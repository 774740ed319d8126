// Device side of an image frame (the `if (u & UPDATE_IMG)` branch, src/slam/main.cpp:167-175): what
// States::update_features (src/ARC/states.cpp:88-213) and Feature::initialize (src/ARC/feature.cpp:48-91) do to the
// feature containers and to the covariance, for a whole range of filters per launch.  The association itself (which
// id goes to which slot) is decided on the host, in the reference's scan order; it arrives here as one small record
// per GROUP of filters that share a plan, uploaded in a single copy:
//     smap[i]  previous slot of new slot i, -1 = a feature that starts in this frame
//     imap[r]  previous state index of new state index r, -1 = fresh feature row, -2 = beyond the new state
// All kernels are bandwidth-bound index shuffles; they run on the caller's stream without host synchronisation.
#pragma once
#include <hip/hip_runtime.h>

#include "feature_kernels.hip.h"

namespace slamk {

struct FrameStage {
    const int* grp;                        // [filters of the range] group of each filter
    const int *nf, *m, *changed;           // [G]   new feature count, measurement rows, 1 = slots or rows move
    const int *isref, *rowoff, *smap;      // [G][F]
    const int* imap;                       // [G][ldn]
    const double* sxy;                     // [G][2][F]  first observation of the features that start here
    const double* quat;                    // [G][4]     attitude at the frame (Feature::initialize, feature.cpp:87)
    const double *ms, *mr;                 // [G][2][F]  the frame's observations in slot order
};

constexpr int FRAME_COMPS = 12;            // pib 3, anchor 3, q0 4, pib0 2

__device__ __forceinline__ double* frame_comp(const FilterView& fv, int b, int c)
{
    const long long F = fv.F;
    if (c < 3) return fv.pib + ((long long)b * 3 + c) * F;
    if (c < 6) return fv.anchor + ((long long)b * 3 + (c - 3)) * F;
    if (c < 10) return fv.q0 + ((long long)b * 4 + (c - 6)) * F;
    return fv.pib0 + ((long long)b * 2 + (c - 10)) * F;
}

// Copies the per-feature containers of every filter whose slots move into tmp[b][12][F] (the gather below is not in place).
__global__ void __launch_bounds__(256) frame_stash_kernel(FilterView fv, FrameStage st, double* __restrict__ tmp)
{
    const int b = blockIdx.y, g = st.grp[b];
    if (!st.changed[g]) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= fv.F) return;
#pragma unroll
    for (int c = 0; c < FRAME_COMPS; ++c) tmp[((long long)b * FRAME_COMPS + c) * fv.F + i] = frame_comp(fv, b, c)[i];
}

// New slot i of filter b: the feature that was in slot smap[i], or a new feature from the frame
// (Feature(X, sense, match), feature.cpp:41-91: pib = (source, 1 / DINIT), anchor = X, q0 = attitude, pib0 = source);
// the measurement layout (nf, m, isref, rowoff) and the frame's observations in slot order.
__global__ void __launch_bounds__(256) frame_gather_kernel(FilterView fv, FrameStage st, const double* __restrict__ tmp,
                                                            double inv_dinit)
{
    const int b = blockIdx.y, g = st.grp[b];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int F = fv.F;
    if (i == 0) { fv.nf[b] = st.nf[g]; fv.m[b] = st.m[g]; }
    if (i >= F) return;
    const long long gi = (long long)g * F + i, bi = (long long)b * F + i;
    fv.isref[bi] = st.isref[gi];
    fv.rowoff[bi] = st.rowoff[gi];
    const long long g2 = (long long)g * 2 * F + i, b2 = (long long)b * 2 * F + i;
    fv.msrc[b2] = st.ms[g2]; fv.msrc[b2 + F] = st.ms[g2 + F];
    fv.mrefl[b2] = st.mr[g2]; fv.mrefl[b2 + F] = st.mr[g2 + F];
    if (i >= st.nf[g]) return;
    const int s = st.smap[gi];
    if (s >= 0) {
        if (!st.changed[g]) return;                  // every feature stays where it is
#pragma unroll
        for (int c = 0; c < FRAME_COMPS; ++c) frame_comp(fv, b, c)[i] = tmp[((long long)b * FRAME_COMPS + c) * F + s];
        return;
    }
    const double sx = st.sxy[g2], sy = st.sxy[g2 + F];
    frame_comp(fv, b, 0)[i] = sx; frame_comp(fv, b, 1)[i] = sy; frame_comp(fv, b, 2)[i] = inv_dinit;
#pragma unroll
    for (int c = 0; c < 3; ++c) frame_comp(fv, b, 3 + c)[i] = fv.X[3 * b + c];
#pragma unroll
    for (int c = 0; c < 4; ++c) frame_comp(fv, b, 6 + c)[i] = st.quat[4 * g + c];
    frame_comp(fv, b, 10)[i] = sx; frame_comp(fv, b, 11)[i] = sy;
}

// Covariance surgery as one gather (States::update_features, src/ARC/states.cpp:129-139, 141-166, 179-196):
// out[i][j] = P[imap[i]][imap[j]] when both sources exist, p0 on the diagonal of a fresh row, 0 elsewhere.
// BACK = false: P -> out for the filters whose rows move; BACK = true: copies the result back (out -> P).
template <bool BACK>
__global__ void __launch_bounds__(256) frame_remap_cov_kernel(FilterView fv, FrameStage st, double* __restrict__ P,
                                                               double* __restrict__ out, int nl, double p0)
{
    const int b = blockIdx.z, g = st.grp[b];
    if (!st.changed[g]) return;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= nl || j >= nl) return;
    const long long off = (long long)b * fv.ldn * fv.ldn, e = off + (long long)i * fv.ldn + j;
    if (BACK) { P[e] = out[e]; return; }
    const int* src = st.imap + (long long)g * fv.ldn;
    const int si = src[i], sj = src[j];
    double v = 0.0;
    if (si >= 0 && sj >= 0) v = P[off + (long long)si * fv.ldn + sj];
    else if (i == j && si == -1) v = p0;
    out[e] = v;
}

// The same surgery in place for covariances that fit LDS (nl <= 128): one workgroup per filter reads its nl x nl block
// with contiguous 16-byte loads and writes the gathered block over it, one pass over P instead of two.
// grid (filters), 512 threads, dynamic LDS nl * nl doubles.
__global__ void __launch_bounds__(512) frame_remap_cov_lds_kernel(FilterView fv, FrameStage st, double* __restrict__ P,
                                                                   int nl, double p0)
{
    extern __shared__ double rimg[];                 // [nl][nl]
    const int b = blockIdx.x, g = st.grp[b];
    if (!st.changed[g]) return;                      // the same for the whole workgroup
    double* Pb = P + (long long)b * fv.ldn * fv.ldn;
    const int half = nl >> 1;
    for (int idx = threadIdx.x; idx < nl * half; idx += 512) {
        const int r = idx / half, c2 = idx - r * half;
        reinterpret_cast<double2*>(rimg + r * nl)[c2] = reinterpret_cast<const double2*>(Pb + (long long)r * fv.ldn)[c2];
    }
    __syncthreads();
    const int* src = st.imap + (long long)g * fv.ldn;
    for (int idx = threadIdx.x; idx < nl * half; idx += 512) {
        const int i = idx / half, c2 = idx - i * half, j = 2 * c2;
        const int si = src[i], s0 = src[j], s1 = src[j + 1];
        double v0 = 0.0, v1 = 0.0;
        if (si >= 0 && s0 >= 0) v0 = rimg[si * nl + s0];
        else if (i == j && si == -1) v0 = p0;
        if (si >= 0 && s1 >= 0) v1 = rimg[si * nl + s1];
        else if (i == j + 1 && si == -1) v1 = p0;
        reinterpret_cast<double2*>(Pb + (long long)i * fv.ldn)[c2] = make_double2(v0, v1);
    }
}

}  // namespace slamk

// Per-feature kernels of the EKF path: motion model, motion Jacobian, measurement
// prediction and measurement Jacobian.  All are HBM-bandwidth bound (a few hundred bytes and
// ~250 flops per feature); lanes run over consecutive feature slots of one filter, and every
// per-feature array is stored component-major per filter ([B][C][F]) so those lanes read and
// write contiguous 8-byte words.
#pragma once
#include <hip/hip_runtime.h>

namespace slamk {

// Device view of one batch of filters.  F = max features, ldn/ldm = padded leading dimensions.
struct FilterView {
    int B, F, ldn, ldm;
    double *X, *V, *bias, *old_pos;             // [B][3]
    int *nf, *m;                                // [B]
    double *pib, *anchor, *q0, *pib0;           // [B][3|3|4|2][F]
    int *isref, *rowoff;                        // [B][F]   measurement row of the feature's first row (feature rows
                                                //          start at row 0, the altitude row is row m - 1)
    double *msrc, *mrefl;                       // [B][2][F] kept frame, slot order
    double *P;                                  // [B][ldn][ldn]
    // measurement rows in compact form: entries in columns 0..2, entries in columns cs..cs+2
    double *ha, *hb;                            // [B][ldm][3]
    int *hcs;                                   // [B][ldm]  (-1: no second group)
    double *e, *rdiag, *qdiag;                  // [B][ldm], [B][ldm], [B][ldn]
    // motion-Jacobian rows in compact form: entries in columns 3..5, entries in columns fgs..fgs+2
    double *fa, *fg;                            // [B][ldn][3]
    int *fgs;                                   // [B][ldn]
    int *status;                                // [B]  bit0: reflection rows inconsistent
    double Q0, R0, qbias, gate, inv_dmax, inv_dmin, alt_override;
    int use_alt_override;
};

// Quaternion::rotation, src/ARC/Quaternion.cpp:32-47 (scalar last, not normalised).
__device__ inline void quat_rot(const double* q, double R[9])
{
    const double q1 = q[0], q2 = q[1], q3 = q[2], q4 = q[3];
    R[0] = q4 * q4 + q1 * q1 - q2 * q2 - q3 * q3;
    R[1] = 2 * (q1 * q2 - q4 * q3);
    R[2] = 2 * (q4 * q2 + q1 * q3);
    R[3] = 2 * (q1 * q2 + q4 * q3);
    R[4] = q4 * q4 - q1 * q1 + q2 * q2 - q3 * q3;
    R[5] = 2 * (q2 * q3 - q4 * q1);
    R[6] = 2 * (q1 * q3 - q4 * q2);
    R[7] = 2 * (q4 * q1 + q2 * q3);
    R[8] = q4 * q4 - q1 * q1 - q2 * q2 + q3 * q3;
}

__device__ inline void m3v(const double* A, const double* v, double* o)
{
    o[0] = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
    o[1] = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
    o[2] = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
}
__device__ inline void m3tv(const double* A, const double* v, double* o)   // A^T v
{
    o[0] = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
    o[1] = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
    o[2] = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
}
__device__ inline void m3m(const double* A, const double* Bm, double* C)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = A[3 * i] * Bm[j] + A[3 * i + 1] * Bm[3 + j] + A[3 * i + 2] * Bm[6 + j];
}
__device__ inline void m3tm(const double* A, const double* Bm, double* C)   // A^T B
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = A[i] * Bm[j] + A[3 + i] * Bm[3 + j] + A[6 + i] * Bm[6 + j];
}

// ---------------------------------------------------------------------------------------------
// Dense motion Jacobian F = I + dt A (jacobianMotionModel, src/slam/main.cpp:241-341) and the
// diagonal of Q (initQ, :430-440), evaluated at the PRE-predict state.  One thread per element
// of the padded nl x nl matrix; rows/cols >= n are written as zero so padding never leaks.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) build_F_kernel(FilterView fv, const double* __restrict__ qbw,
                                                       const double* __restrict__ w,
                                                       const double* __restrict__ dtv,
                                                       double* __restrict__ Fm, int nl)
{
    const int b = blockIdx.z;
    const int k = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (j >= nl || k >= nl) return;
    const int nf = fv.nf[b], n = 9 + 3 * nf;
    const double dt = dtv[b];
    double v = 0.0;
    if (j < n && k < n) {
        double a = 0.0;
        bool neg_dt = false;
        if (j < 3) {
            if (k >= 3 && k < 6) {
                double R[9];
                quat_rot(qbw + 4 * b, R);                 // :253-261
                a = R[3 * j + (k - 3)];
            }
        } else if (j < 6) {
            neg_dt = (k == 6 + 3 * nf + (j - 3));         // :338-340
        } else if (j < 6 + 3 * nf) {
            const int i = (j - 6) / 3, r = (j - 6) - 3 * i;
            const long long fb = (long long)b * 3 * fv.F;
            const double p0 = fv.pib[fb + i], p1 = fv.pib[fb + fv.F + i], p2 = fv.pib[fb + 2 * fv.F + i];
            const double* V = fv.V + 3 * b;
            const double* ww = w + 3 * b;
            if (k >= 3 && k < 6) {                        // Fib_ith :314-317
                const int c = k - 3;
                if (r == 0) a = (c == 0) ? p0 * p2 : (c == 1 ? -p2 : 0.0);
                else if (r == 1) a = (c == 0) ? p1 * p2 : (c == 2 ? -p2 : 0.0);
                else a = (c == 0) ? p2 * p2 : 0.0;
            } else if (k >= 6 + 3 * i && k < 9 + 3 * i) {  // FiTemp :290-299
                const int c = k - (6 + 3 * i);
                if (r == 0)
                    a = (c == 0) ? p2 * V[0] - 2 * p0 * ww[2] + p1 * ww[1] : (c == 1 ? ww[0] + p0 * ww[1] : p0 * V[0] - V[1]);
                else if (r == 1)
                    a = (c == 0) ? -ww[0] - p1 * ww[2] : (c == 1 ? p2 * V[0] - p0 * ww[2] + 2 * p1 * ww[1] : p1 * V[0] - V[2]);
                else
                    a = (c == 0) ? -p2 * ww[2] : (c == 1 ? p2 * ww[1] : 2 * p2 * V[0] - p0 * ww[2] + p1 * ww[1]);
            }
        }
        v = dt * a + (j == k ? 1.0 : 0.0);                // :335
        if (neg_dt) v = -1 * dt;
    }
    Fm[(long long)b * fv.ldn * fv.ldn + (long long)j * fv.ldn + k] = v;
    if (k == 0) {
        double q = 0.0;
        if (j < n) {
            const double q0 = fv.Q0 * (dt * dt);
            q = q0;
            if (j < 3) q = q0 * (0.1 * q0);
            if (j >= 6 + 3 * nf) q = fv.qbias * dt * dt;
        }
        fv.qdiag[(long long)b * fv.ldn + j] = q;
    }
}

// ---------------------------------------------------------------------------------------------
// mu += dynamics(mu) * dt  (States::dynamics + add, src/ARC/states.cpp:222-263, :69-86).
// One workgroup per filter; all increments use the pre-step V.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void dynamics_filter(const FilterView& fv, int b, const double* __restrict__ qbw,
                                                const double* __restrict__ w, const double* __restrict__ acc,
                                                const double* __restrict__ dtv)
{
    const int nf = fv.nf[b];
    const double dt = dtv[b];
    const double V0 = fv.V[3 * b], V1 = fv.V[3 * b + 1], V2 = fv.V[3 * b + 2];
    const double w0 = w[3 * b], w1 = w[3 * b + 1], w2 = w[3 * b + 2];
    const long long fb = (long long)b * 3 * fv.F;
    for (int i = threadIdx.x; i < nf; i += blockDim.x) {
        const double p0 = fv.pib[fb + i], p1 = fv.pib[fb + fv.F + i], p2 = fv.pib[fb + 2 * fv.F + i];
        const double d0 = (-V1 + p0 * V0) * p2 + p1 * w0 - (1 + p0 * p0) * w2 + p0 * p1 * w1;   // :254
        const double d1 = (-V2 + p1 * V0) * p2 - p0 * w0 + (1 + p1 * p1) * w1 - p0 * p1 * w2;   // :255
        const double d2 = (-w2 * p0 + w1 * p1) * p2 + V0 * p2 * p2;                             // :256
        fv.pib[fb + i] = p0 + d0 * dt;
        fv.pib[fb + fv.F + i] = p1 + d1 * dt;
        fv.pib[fb + 2 * fv.F + i] = p2 + d2 * dt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double R[9], Vv[3] = {V0, V1, V2}, dX[3];
        quat_rot(qbw + 4 * b, R);
        m3v(R, Vv, dX);                                                                          // :241
        for (int c = 0; c < 3; ++c) {
            fv.X[3 * b + c] += dX[c] * dt;
            fv.V[3 * b + c] = Vv[c] + (acc[3 * b + c] - fv.bias[3 * b + c]) * dt;                // :243,:260
        }
    }
}

__global__ void __launch_bounds__(256) dynamics_kernel(FilterView fv, const double* __restrict__ qbw,
                                                        const double* __restrict__ w,
                                                        const double* __restrict__ acc,
                                                        const double* __restrict__ dtv)
{
    dynamics_filter(fv, blockIdx.x, qbw, w, acc, dtv);
}

// old_pos = X and inverse-depth clamp (src/slam/main.cpp:168, src/ARC/states.cpp:91-103).
// b_only >= 0 restricts the kernel to one filter.
__device__ __forceinline__ void frame_begin_filter(const FilterView& fv, int b)
{
    const int nf = fv.nf[b];
    double* p2 = fv.pib + (long long)b * 3 * fv.F + 2 * fv.F;
    for (int i = threadIdx.x; i < nf; i += blockDim.x) {
        double d = p2[i];
        if (d < fv.inv_dmax) d = fv.inv_dmax;
        else if (d > fv.inv_dmin) d = fv.inv_dmin;
        p2[i] = d;
    }
    if (threadIdx.x < 3) fv.old_pos[3 * b + threadIdx.x] = fv.X[3 * b + threadIdx.x];
}

__global__ void __launch_bounds__(256) frame_begin_kernel(FilterView fv, int b_only)
{
    frame_begin_filter(fv, b_only >= 0 ? b_only : blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Measurement prediction, innovation, R diagonal and the compact measurement Jacobian.
//   Feature::xibHat/xib0Hat/pib0Hat/xpbHat/ppbHat  src/ARC/feature.cpp:143-198
//   jacobianH                                       src/slam/main.cpp:575-624 (chain-rule form)
//   measurementModel row layout                     src/slam/main.cpp:348-408
//   View subtract / toMat                           src/ARC/view.cpp:3-47
//   initR                                           src/slam/main.cpp:449-476
// pib0Hat is evaluated at old_pos, the Jacobians and ppbHat at the post-predict X (:376-386).
// One thread per feature slot; thread (0,0) of each filter also writes the altitude row, and all threads
// fill the padding rows [m, ml) with e = 0, r = 1, H = 0.
// ---------------------------------------------------------------------------------------------
// Thread i of `stride` threads working on filter b (no barrier inside).
__device__ __forceinline__ void measure_filter(const FilterView& fv, int b, int i, int stride,
                                               const double* __restrict__ qbw, const double* __restrict__ alt,
                                               const double* __restrict__ msrc, const double* __restrict__ mrefl, int ml)
{
    const int nf = fv.nf[b], m = fv.m[b];
    const long long rb = (long long)b * fv.ldm;
    double* ha = fv.ha + rb * 3;
    double* hb = fv.hb + rb * 3;
    int* hcs = fv.hcs + rb;
    double* e = fv.e + rb;
    double* rd = fv.rdiag + rb;
    const double X[3] = {fv.X[3 * b], fv.X[3 * b + 1], fv.X[3 * b + 2]};

    if (i == 0) {                                           // altitude row, :351-354, :452
        // The reference puts it first (row 0 of H, R and e); here it is the LAST row, m - 1, behind the feature
        // rows, so that those start on a 16-row tile boundary.  The order of the measurement rows does not change
        // K e, K H or K R K^T; slamk_debug_get hands rows out in the reference's order.
        const int ia = m - 1;
        const double a = fv.use_alt_override ? fv.alt_override : alt[b];
        e[ia] = a - (-X[2]);
        rd[ia] = 0.15 * fv.R0;
        ha[3 * ia] = 0.0; ha[3 * ia + 1] = 0.0; ha[3 * ia + 2] = -1.0;
        hb[3 * ia] = hb[3 * ia + 1] = hb[3 * ia + 2] = 0.0;
        hcs[ia] = -1;
    }
    for (int q = m + i; q < ml; q += stride) {              // padding rows
        e[q] = 0.0; rd[q] = 1.0; hcs[q] = -1;
        for (int c = 0; c < 3; ++c) { ha[3 * q + c] = 0.0; hb[3 * q + c] = 0.0; }
    }
    if (i >= nf) return;

    const int F = fv.F;
    const long long f3 = (long long)b * 3 * F, f4 = (long long)b * 4 * F, f2 = (long long)b * 2 * F;
    const double p[3] = {fv.pib[f3 + i], fv.pib[f3 + F + i], fv.pib[f3 + 2 * F + i]};
    const double an[3] = {fv.anchor[f3 + i], fv.anchor[f3 + F + i], fv.anchor[f3 + 2 * F + i]};
    const double q0[4] = {fv.q0[f4 + i], fv.q0[f4 + F + i], fv.q0[f4 + 2 * F + i], fv.q0[f4 + 3 * F + i]};
    const double pib0[2] = {fv.pib0[f2 + i], fv.pib0[f2 + F + i]};
    const int ref = fv.isref[(long long)b * F + i];
    const int row = fv.rowoff[(long long)b * F + i];
    const double sx = msrc[f2 + i], sy = msrc[f2 + F + i];
    const double op[3] = {fv.old_pos[3 * b], fv.old_pos[3 * b + 1], fv.old_pos[3 * b + 2]};

    double Rb[9], R0[9], Rb2b0[9];
    quat_rot(qbw + 4 * b, Rb);
    quat_rot(q0, R0);
    m3tm(R0, Rb, Rb2b0);                                    // R(q0)^T R(qbw), feature.cpp:117-126
    const double ip = 1 / p[2];
    const double xib[3] = {ip, p[0] * ip, p[1] * ip};       // :171-177

    double rx[3], dX[3], dO[3], fa[3], x0[3], x0o[3];
    m3v(Rb2b0, xib, rx);
    for (int c = 0; c < 3; ++c) { dX[c] = X[c] - an[c]; dO[c] = op[c] - an[c]; }
    m3tv(R0, dX, fa);                                       // fromAnchor :110-114
    for (int c = 0; c < 3; ++c) x0[c] = rx[c] + fa[c];
    m3tv(R0, dO, fa);
    for (int c = 0; c < 3; ++c) x0o[c] = rx[c] + fa[c];

    // d xib / d pib
    const double D[9] = {0, 0, -ip * ip, ip, 0, -p[0] * ip * ip, 0, ip, -p[1] * ip * ip};
    double M0[9];
    m3m(Rb2b0, D, M0);
    const double i0 = 1 / x0[0];
    const double J0[6] = {-x0[1] * i0 * i0, i0, 0, -x0[2] * i0 * i0, 0, i0};

    // rows 0,1: current view (identity on pib0, pib1)
    e[row + 0] = sx - p[0];
    e[row + 1] = sy - p[1];
    // rows 2,3: initial view
    e[row + 2] = pib0[0] - x0o[1] / x0o[0];                 // pib0Hat(old_pos) :189-198
    e[row + 3] = pib0[1] - x0o[2] / x0o[0];
    const double r_cur = 0.01 / 770 * fv.R0, r_ini = 10. / 770 * fv.R0, r_ref = 1. / 770 * fv.R0;
    rd[row + 0] = r_cur; rd[row + 1] = r_cur; rd[row + 2] = r_ini; rd[row + 3] = r_ini;
    const int cs = 6 + 3 * i;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        hcs[row + r] = cs;
        for (int c = 0; c < 3; ++c) { ha[3 * (row + r) + c] = 0.0; hb[3 * (row + r) + c] = (c == r) ? 1.0 : 0.0; }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        hcs[row + 2 + r] = cs;
        for (int c = 0; c < 3; ++c) {
            // J0 * R0^T  and  J0 * Rb2b0 * D
            ha[3 * (row + 2 + r) + c] = J0[3 * r] * R0[3 * c] + J0[3 * r + 1] * R0[3 * c + 1] + J0[3 * r + 2] * R0[3 * c + 2];
            hb[3 * (row + 2 + r) + c] = J0[3 * r] * M0[c] + J0[3 * r + 1] * M0[3 + c] + J0[3 * r + 2] * M0[6 + c];
        }
    }
    if (ref) {
        // xpb = Rb^T (S Rb xib - 2 n n^T X), S = diag(1,1,-1), n = (0,0,1)   feature.cpp:155-169
        double t[3], xp[3];
        m3v(Rb, xib, t);
        t[2] = -t[2];
        t[2] -= 2 * X[2];
        m3tv(Rb, t, xp);
        const double ix = 1 / xp[0];
        const double Jp[6] = {-xp[1] * ix * ix, ix, 0, -xp[2] * ix * ix, 0, ix};
        double SR[9], RSR[9], Mp[9];
        for (int c = 0; c < 3; ++c) { SR[c] = Rb[c]; SR[3 + c] = Rb[3 + c]; SR[6 + c] = -Rb[6 + c]; }
        m3tm(Rb, SR, RSR);
        m3m(RSR, D, Mp);
        const double mrx = mrefl[f2 + i], mry = mrefl[f2 + F + i];
        const double ex = mrx - xp[1] * ix, ey = mry - xp[2] * ix;   // ppbHat :143-153
        e[row + 4] = ex;
        e[row + 5] = ey;
        rd[row + 4] = r_ref; rd[row + 5] = r_ref;
        // the reference drops these rows from R when the observation is (0,0) and from e when
        // its x or the x innovation is exactly 0, and then fails on the size mismatch
        if ((mrx == 0 && mry == 0) || mrx == 0 || ex == 0) atomicOr(fv.status + b, 1);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            hcs[row + 4 + r] = cs;
            // d xpb / d X = Rb^T (-2 n n^T): only column 2
            ha[3 * (row + 4 + r) + 0] = 0.0;
            ha[3 * (row + 4 + r) + 1] = 0.0;
            ha[3 * (row + 4 + r) + 2] = -2 * (Jp[3 * r] * Rb[6] + Jp[3 * r + 1] * Rb[7] + Jp[3 * r + 2] * Rb[8]);
            for (int c = 0; c < 3; ++c)
                hb[3 * (row + 4 + r) + c] = Jp[3 * r] * Mp[c] + Jp[3 * r + 1] * Mp[3 + c] + Jp[3 * r + 2] * Mp[6 + c];
        }
    }
}

__global__ void __launch_bounds__(64) measure_kernel(FilterView fv, const double* __restrict__ qbw,
                                                      const double* __restrict__ alt,
                                                      const double* __restrict__ msrc,
                                                      const double* __restrict__ mrefl, int ml)
{
    measure_filter(fv, blockIdx.y, blockIdx.x * 64 + threadIdx.x, gridDim.x * 64, qbw, alt, msrc, mrefl, ml);
}

// Dense H (ml x nl) from the compact rows.
__global__ void __launch_bounds__(256) build_H_kernel(FilterView fv, double* __restrict__ Hm, int ml, int nl)
{
    const int b = blockIdx.z;
    const int k = blockIdx.x * 64 + (threadIdx.x & 63);
    const int q = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (q >= ml || k >= nl) return;
    const long long rb = (long long)b * fv.ldm + q;
    double v = 0.0;
    if (k < 3) v = fv.ha[rb * 3 + k];
    else {
        const int cs = fv.hcs[rb];
        if (cs >= 0 && k >= cs && k < cs + 3) v = fv.hb[rb * 3 + (k - cs)];
    }
    Hm[(long long)b * fv.ldm * fv.ldn + (long long)q * fv.ldn + k] = v;
}

// dx = K e (src/slam/main.cpp:228): one wavefront per row of K, 16 rows per workgroup.
// Bias rows are skipped (States(const Mat&), src/ARC/states.cpp:24-37, ignores them).
__global__ void __launch_bounds__(256) gain_kx_kernel(FilterView fv, const double* __restrict__ K, int ldk,
                                                       double* __restrict__ dx, int ml)
{
    const int b = blockIdx.y;
    const int n = 9 + 3 * fv.nf[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* Kb = K + (long long)b * fv.ldn * ldk;
    const double* e = fv.e + (long long)b * fv.ldm;
    double* d = dx + (long long)b * fv.ldn;
    for (int rr = 0; rr < 4; ++rr) {
        const int r = blockIdx.x * 16 + wave * 4 + rr;
        if (r >= n - 3) return;
        double s = 0.0;
        for (int q = lane; q < ml; q += 64) s += Kb[(long long)r * ldk + q] * e[q];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) d[r] = s;
    }
}

// Gate and mean update (src/slam/main.cpp:229-232): `if (sum|V - dV| < gate) mu += dx` on X, V, pib.
__global__ void __launch_bounds__(256) gain_apply_kernel(FilterView fv, const double* __restrict__ dx,
                                                          int* __restrict__ applied)
{
    const int b = blockIdx.x;
    const int nf = fv.nf[b];
    const double* d = dx + (long long)b * fv.ldn;
    const double* V = fv.V + 3 * b;
    const double g = fabs(V[0] - d[3]) + fabs(V[1] - d[4]) + fabs(V[2] - d[5]);
    const bool ok = g < fv.gate;
    __syncthreads();                       // every thread has read V before anyone updates it
    if (threadIdx.x == 0 && applied) applied[b] = ok;
    if (!ok) return;
    if (threadIdx.x < 3) {
        fv.X[3 * b + threadIdx.x] += d[threadIdx.x];
        fv.V[3 * b + threadIdx.x] += d[3 + threadIdx.x];
    }
    const long long fb = (long long)b * 3 * fv.F;
    for (int t = threadIdx.x; t < 3 * nf; t += blockDim.x) {
        const int i = t / 3, c = t - 3 * i;
        fv.pib[fb + (long long)c * fv.F + i] += d[6 + t];
    }
}

}  // namespace slamk

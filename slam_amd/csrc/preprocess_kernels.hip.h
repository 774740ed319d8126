// Batched forms of the reference's stream pre-processing filters (one record per lane; all HBM-bound):
//   euler2qbw  src/euler2qbw/euler2qbw.cpp:37-46      Euler angles -> scalar-last quaternion
//   rotate     src/rotate/main.cpp:146-154            v -> R(q) v with Quaternion::rotation (not normalised)
//   img2body   src/img2body/img2body.cpp:31-67        pixel -> body-frame bearing ratios
//   fir        src/lpf/fir.cpp:39-49,105-115          3-channel FIR low-pass, zero history
#pragma once
#include <hip/hip_runtime.h>

#include "feature_kernels.hip.h"

namespace slamk {

__global__ void __launch_bounds__(256) euler2qbw_kernel(const double* __restrict__ eu, double* __restrict__ q, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double phi = eu[3 * i], theta = eu[3 * i + 1], psi = eu[3 * i + 2];
    const double sp = sin(phi / 2), cp = cos(phi / 2), st = sin(theta / 2), ct = cos(theta / 2);
    const double ss = sin(psi / 2), cs = cos(psi / 2);
    q[4 * i + 0] = sp * ct * cs - cp * st * ss;
    q[4 * i + 1] = cp * st * cs + sp * ct * ss;
    q[4 * i + 2] = cp * ct * ss - sp * st * cs;
    q[4 * i + 3] = cp * ct * cs + sp * st * ss;
}

// One quaternion for the whole stream, as the reference's filter takes it from its command line.
__global__ void __launch_bounds__(256) rotate_kernel(double q0, double q1, double q2, double q3,
                                                      const double* __restrict__ v, double* __restrict__ out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double q[4] = {q0, q1, q2, q3};
    double R[9], o[3];
    quat_rot(q, R);
    const double x[3] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};
    m3v(R, x, o);
    out[3 * i] = o[0]; out[3 * i + 1] = o[1]; out[3 * i + 2] = o[2];
}

__global__ void __launch_bounds__(256) img2body_kernel(const double* __restrict__ px, double* __restrict__ out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double cx = 767.521362, cy = 591.428406, fx = 801.362183, fy = 801.362183;           // :31-35
    const double qbc[4] = {-0.707106781186547, -0.000000000000000, 0.000000000000000, 0.707106781186548};
    double Rb2c[9], S[3], Y[3];
    quat_rot(qbc, Rb2c);
    const double x = (px[2 * i] - cx) / fx, y = (px[2 * i + 1] - cy) / fy;
    const double s = 1 / sqrt(x * x + y * y + 1);
    S[0] = x * s; S[1] = y * s; S[2] = 1 * s;
    m3tv(Rb2c, S, Y);                                     // camera -> body
    out[2 * i] = Y[1] / Y[0];
    out[2 * i + 1] = Y[2] / Y[0];
}

// y[i][c] = sum_{k < order} x[i - k][c] coeffs[k], samples before the first one are zero; k ascending as the
// reference's ring buffer visits them.
__global__ void __launch_bounds__(256) fir_kernel(const double* __restrict__ coeffs, const double* __restrict__ x,
                                                   double* __restrict__ y, int order, int n)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 3 * n) return;
    const int i = t / 3, c = t - 3 * i;
    double acc = 0;
    // separate multiply and add, as the reference's x86-64 build rounds them
    for (int k = 0; k < order && k <= i; ++k) {
#pragma clang fp contract(off)
        acc = acc + x[3 * (i - k) + c] * coeffs[k];
    }
    y[t] = acc;
}

}  // namespace slamk

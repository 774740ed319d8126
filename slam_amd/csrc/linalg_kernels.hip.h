// Small kernels around the MFMA GEMM: symmetrisation, covariance row/column gather, copies.
#pragma once
#include <hip/hip_runtime.h>

namespace slamk {

// MIRROR = false: P = (A + A^T) / 2 over the padded nl x nl block (src/slam/main.cpp:567).
// MIRROR = true : P = the symmetric matrix whose lower triangle is A's (A's upper part is not read
//                 as data): used after products that only computed the lower tiles.
template <bool MIRROR>
__global__ void __launch_bounds__(256) symmetrize_kernel(const double* __restrict__ A, double* __restrict__ P,
                                                          int ld, long long stride, int nl)
{
    __shared__ double t[32][33];
    const long long off = (long long)blockIdx.z * stride;
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int i = bx + r, j = by + tx;                    // transposed tile (bx,by)
        t[r][tx] = (i < nl && j < nl) ? A[off + (long long)i * ld + j] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = by + r, j = bx + tx;
        if (i < nl && j < nl) {
            const double a = A[off + (long long)i * ld + j], at = t[tx][r];
            P[off + (long long)i * ld + j] = MIRROR ? (i >= j ? a : at) : (a + at) / 2;
        }
    }
}

// In place: the upper triangle of the nl x nl matrix P becomes the mirror image of its lower triangle (which is only
// read).  grid (ceil(nl / 32), ceil(nl / 32)); blocks above the diagonal have nothing to do.
__global__ void __launch_bounds__(256) mirror_lower_inplace_kernel(double* __restrict__ P, int ld, int nl)
{
    __shared__ double t[32][33];
    if (blockIdx.y > blockIdx.x) return;
    const int bi = blockIdx.x * 32, bj = blockIdx.y * 32;     // lower tile: rows bi.., columns bj..
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int i = bi + r, j = bj + tx;
        t[r][tx] = (i < nl && j < nl) ? P[(long long)i * ld + j] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = bj + r, j = bi + tx;                    // element (i, j) of the upper tile = (j, i) of the lower
        if (i < nl && j < nl && j > i) P[(long long)i * ld + j] = t[tx][r];
    }
}

// dst[i][j] = src[i][j] for a rows x cols block (same ld and batch stride on both sides).
__global__ void __launch_bounds__(256) copy_block_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                                          int ld, long long stride, int rows, int cols)
{
    const long long off = (long long)blockIdx.z * stride;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i < rows && j < cols) dst[off + (long long)i * ld + j] = src[off + (long long)i * ld + j];
}

// MFMA issue-rate probe: every wave runs `iters` rounds of 8 independent accumulators.
__global__ void __launch_bounds__(256) mfma_probe_kernel(double* out, int iters)
{
    typedef double v4 __attribute__((ext_vector_type(4)));
    v4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, bq = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)      // in-place accumulators: the builtin form is given extra register moves here
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(bq));
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;   // keep the chain alive
}

}  // namespace slamk

// Small dense-linear-algebra kernels around the MFMA GEMM: diagonal-block Cholesky with
// triangular inverse, symmetrisation, covariance row/column gather.
#pragma once
#include <hip/hip_runtime.h>

namespace slamk {

// Cholesky of one nb x nb diagonal block (nb <= NB, nb a multiple of 16) of S at (j0, j0), and the
// inverse of its factor.  One 256-thread workgroup per matrix of the batch; the block lives in LDS.
//   in : S  lower triangle of the block
//   out: L  block (j0,j0) = chol factor (lower; strict upper written as 0)
//        Li block (j0,j0) = inverse of that factor (lower; strict upper written as 0)
// Blocked inside LDS with 16-wide panels:
//   1. the 16 x 16 diagonal block is factored in place with one thread per element, and inverted
//      column by column by 16 lanes,
//   2. the rows below are multiplied by that inverse (L21 = A21 inv(L11)^T),
//   3. the trailing block takes the rank-16 update, threads laid out 16 x 16 over (row, column).
// The inverse of the whole factor is then assembled in place by recursive doubling over the
// 16-blocks, inv([[A,0],[B,C]]) = [[inv A,0],[-inv C B inv A, inv C]], with the product B inv A parked
// in the (unused) strict upper triangle of the tile.
template <int NB>
__global__ void __launch_bounds__(256) potf2_inv_kernel(const double* __restrict__ S, double* __restrict__ L,
                                                         double* __restrict__ Li, int ld, long long stride,
                                                         int j0, int nb)
{
    extern __shared__ double sm[];
    constexpr int LDA = NB + 1;
    double* a = sm;                         // [NB][LDA]
    double* dinv = sm + NB * LDA;           // [NB/16][16][17] inverses of the diagonal 16-blocks
    const long long off = (long long)blockIdx.x * stride + (long long)j0 * ld + j0;
    const double* Sb = S + off;
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;

    for (int i = ty; i < nb; i += 16)
        for (int j = tx; j < nb; j += 16) a[i * LDA + j] = (j <= i) ? Sb[(long long)i * ld + j] : 0.0;

    const int np = nb >> 4;
    for (int p = 0; p < np; ++p) {
        const int k0 = 16 * p;
        __syncthreads();
        // 1. diagonal 16 x 16 block: right-looking factorisation, thread (ty, tx) owns element (ty, tx)
        for (int k = 0; k < 16; ++k) {
            const double akk = a[(k0 + k) * LDA + k0 + k];
            const double aik = a[(k0 + ty) * LDA + k0 + k];
            const double ajk = a[(k0 + tx) * LDA + k0 + k];
            __syncthreads();
            const double d = sqrt(akk), id = 1.0 / d;
            if (tx == k) {
                if (ty == k) a[(k0 + k) * LDA + k0 + k] = d;
                else if (ty > k) a[(k0 + ty) * LDA + k0 + k] = aik * id;
            } else if (tx > k && ty >= tx) {
                a[(k0 + ty) * LDA + k0 + tx] -= (aik * id) * (ajk * id);
            }
            __syncthreads();
        }
        //    its inverse, column c by forward substitution (16 lanes)
        if (tid < 16) {
            const int c = tid;
            double x[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                double sacc = (i == c) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < i; ++k) sacc -= a[(k0 + i) * LDA + k0 + k] * x[k];
                x[i] = (i >= c) ? sacc / a[(k0 + i) * LDA + k0 + i] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) dinv[(p * 16 + i) * 17 + c] = x[i];   // inv[i][c]
        }
        __syncthreads();
        const int r0 = k0 + 16;
        if (r0 >= nb) break;
        // L21 = A21 * inv(L11)^T : thread (ty, tx) -> rows r0 + ty + 16 q, column tx
        {
            double v[NB / 16];
            double di[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) di[k] = dinv[(p * 16 + tx) * 17 + k];   // inv[tx][k], zero for k > tx
#pragma unroll
            for (int q = 0; q < NB / 16; ++q) {
                const int i = r0 + ty + 16 * q;
                double acc = 0.0;
                if (i < nb) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc += a[i * LDA + k0 + k] * di[k];
                }
                v[q] = acc;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NB / 16; ++q) {
                const int i = r0 + ty + 16 * q;
                if (i < nb) a[i * LDA + k0 + tx] = v[q];
            }
        }
        __syncthreads();
        // trailing update A22 -= L21 L21^T on the lower 16-tiles
        for (int qi = 0; r0 + 16 * qi < nb; ++qi) {
            const int i = r0 + ty + 16 * qi;
            double li[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) li[k] = a[i * LDA + k0 + k];
            for (int qj = 0; qj <= qi; ++qj) {
                const int j = r0 + tx + 16 * qj;
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) acc += li[k] * a[j * LDA + k0 + k];
                a[i * LDA + j] -= acc;
            }
        }
    }
    __syncthreads();
    double* Lb = L + off;
    for (int i = ty; i < nb; i += 16)
        for (int j = tx; j < nb; j += 16) Lb[(long long)i * ld + j] = (j <= i) ? a[i * LDA + j] : 0.0;
    __syncthreads();
    // diagonal 16-blocks of the tile become their inverses
    for (int t = tid; t < np * 256; t += 256) {
        const int p = t >> 8, i = (t >> 4) & 15, j = t & 15;
        if (j <= i) a[(16 * p + i) * LDA + 16 * p + j] = dinv[(p * 16 + i) * 17 + j];
    }
    int lg = 4;
    for (int s = 16; s < nb; s *= 2, ++lg) {
        __syncthreads();
        // T = B * inv(A), kept transposed-in-place at [top row][bottom col] of the strict upper triangle
        for (int base = 0; base + s < nb; base += 2 * s) {
            const int s2 = min(s, nb - base - s);
            for (int e = tid; e < s2 * s; e += 256) {
                const int i = e >> lg, j = e & (s - 1);
                const double* bi = &a[(base + s + i) * LDA + base];
                const double* aj = &a[base * LDA + base + j];
                double acc = 0.0;
#pragma unroll 8
                for (int k = 0; k < s; ++k) {
                    const double x = bi[k], y = aj[k * LDA];
                    acc += (k >= j) ? x * y : 0.0;
                }
                a[(base + j) * LDA + base + s + i] = acc;
            }
        }
        __syncthreads();
        // B <- -inv(C) * T
        for (int base = 0; base + s < nb; base += 2 * s) {
            const int s2 = min(s, nb - base - s);
            for (int e = tid; e < s2 * s; e += 256) {
                const int i = e >> lg, j = e & (s - 1);
                const double* ci = &a[(base + s + i) * LDA + base + s];
                const double* tj = &a[(base + j) * LDA + base + s];
                double acc = 0.0;
#pragma unroll 8
                for (int k = 0; k < s2; ++k) {
                    const double x = ci[k], y = tj[k];
                    acc += (k <= i) ? x * y : 0.0;
                }
                a[(base + s + i) * LDA + base + j] = -acc;
            }
        }
    }
    __syncthreads();
    double* Lib = Li + off;
    for (int i = ty; i < nb; i += 16)
        for (int j = tx; j < nb; j += 16) Lib[(long long)i * ld + j] = (j <= i) ? a[i * LDA + j] : 0.0;
}

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

// dst[i][j] = src[i][j] for a rows x cols block (same ld and batch stride on both sides).
__global__ void __launch_bounds__(256) copy_block_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                                          int ld, long long stride, int rows, int cols)
{
    const long long off = (long long)blockIdx.z * stride;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i < rows && j < cols) dst[off + (long long)i * ld + j] = src[off + (long long)i * ld + j];
}

// Covariance surgery as one gather (States::update_features, src/ARC/states.cpp:129-139,
// 141-166, 179-196): out[i][j] = P[src[i]][src[j]] when both sources exist, p0 on the diagonal of
// a fresh row, 0 elsewhere.  src has nl entries; -1 = fresh (new feature), -2 = padding.
__global__ void __launch_bounds__(256) remap_cov_kernel(const double* __restrict__ P, double* __restrict__ out,
                                                         const int* __restrict__ src, int ld, int nl, double p0)
{
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= nl || j >= nl) return;
    const int si = src[i], sj = src[j];
    double v = 0.0;
    if (si >= 0 && sj >= 0) v = P[(long long)si * ld + sj];
    else if (i == j && si == -1) v = p0;
    out[(long long)i * ld + j] = v;
}

// Per-feature container gather for the same surgery: dst slot i takes src slot map[i] (>= 0) or is
// left for the initialiser (-1).  ncomp components, arrays are [ncomp][F].
__global__ void gather_slots_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                    const int* __restrict__ map, int F, int nf_new, int ncomp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nf_new) return;
    const int s = map[i];
    if (s < 0) return;
    for (int c = 0; c < ncomp; ++c) dst[(long long)c * F + i] = src[(long long)c * F + s];
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
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;   // keep the chain alive
}

}  // namespace slamk

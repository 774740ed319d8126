// Fused gain solver for small filters: K = W S^-1 (and K R) from S = L L^T, one workgroup per filter.
//
//   in : S  m x m (lower 16-tiles valid), W  n x m,  r[m] (diagonal of R)
//   out: K  n x m, optionally KR = K diag(r) and L (lower)
//   m = 16 MT with MT <= 11 (the lower tiles of S, 2 KB each, live in LDS: MT (MT+1) / 2 <= 66 tiles),
//   n = 16 NT with NT <= 8 (one wavefront per 16 columns of W^T).
//
// Phase 1  tile Cholesky, right-looking, in LDS.  Per tile column j: wavefront 0 factors the diagonal
//          tile and replaces it by its inverse; the tiles below become S_ij inv(L_jj)^T; the trailing
//          tiles take S_ik -= L_ij L_kj^T.  All tile products are v_mfma_f64_16x16x4_f64.
// Phase 2  two triangular solves on the transposed system, so that finished tiles are consumed
//          straight from the accumulator registers: an MFMA result tile (lane = column, register =
//          row group) IS the B operand of the next MFMA, no data movement.
//            Y^T_j = inv(L_jj)   (W^T_j - sum_{k<j} L_jk   Y^T_k)      j = 0 .. MT-1
//            K^T_j = inv(L_jj)^T (Y^T_j - sum_{k>j} L_kj^T K^T_k)      j = MT-1 .. 0
//          Wavefront w owns columns [16 w, 16 w + 16) of W^T / K^T (rows of W / K).
//
// LDS tile image: element (r, c) of a 16 x 16 tile at r * 16 + (c ^ (r & 14)).  With that XOR every
// access pattern used here -- row-fragment reads (lane = row, 4 s + kk = column), the same on the
// transposed tile, and accumulator-shaped reads and writes -- touches 32 distinct 8-byte banks per
// half-wavefront.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm.hip.h"

namespace slamk {

__device__ __forceinline__ int tsw(int r, int c) { return r * 16 + (c ^ (r & 14)); }
__device__ __forceinline__ int tile_index(int i, int j) { return i * (i + 1) / 2 + j; }

// D = A * B + C on 16 x 16 tiles; A read from an LDS tile, rows = lanes (optionally the transposed
// tile, optionally negated); B is an accumulator-shaped register tile.
template <bool TRANS, bool NEG>
__device__ __forceinline__ v4d tile_mma_regB(const double* __restrict__ ta, v4d b, v4d c, int li, int lk)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        double a = TRANS ? ta[tsw(4 * s + lk, li)] : ta[tsw(li, 4 * s + lk)];
        if (NEG) a = -a;
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s], c, 0, 0, 0);
    }
    return c;
}

// D = (+/-) A * Bt^T + C with both operands LDS tiles read by rows (B[k][j] = tb[j][k]).
template <bool NEG>
__device__ __forceinline__ v4d tile_mma_ldsBt(const double* __restrict__ ta, const double* __restrict__ tb, v4d c,
                                              int li, int lk)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        double a = ta[tsw(li, 4 * s + lk)];
        const double b = tb[tsw(li, 4 * s + lk)];
        if (NEG) a = -a;
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    return c;
}

// 1/sqrt(a) to double precision: hardware estimate (v_rsq_f64) + two Newton steps.
__device__ __forceinline__ double rsqrt_nr(double a)
{
    double y = __builtin_amdgcn_rsq(a);
    y = y * (1.5 - 0.5 * a * y * y);
    y = y * (1.5 - 0.5 * a * y * y);
    return y;
}

// One wavefront: Cholesky of the 16 x 16 tile t (lower triangle), then t := inv(L) (lower, zero above).
// Lg != nullptr receives L (lower, zero above) at Lg[row * ldl + col].
// Lane (r = lane & 15, cg = lane >> 4) keeps A[r][4 cg .. 4 cg + 3] in registers; per pivot only column
// k travels through LDS.  The body is branch-free: lanes that own nothing in a column simply keep their
// values (selects), and every lane re-publishes one of its elements per step, which leaves the
// not-yet-final columns of the LDS copy stale but rewrites column k + 1 just before it is read.
__device__ inline void diag_tile_factor_invert(double* __restrict__ t, double* __restrict__ Lg, int ldl, int lane)
{
    const int r = lane & 15, cg = lane >> 4;
    const int rb = r * 16, rm = r & 14;
    int cb[4], cm[4];
    double cur[4], idk[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        cb[q] = (4 * cg + q) * 16;
        cm[q] = (4 * cg + q) & 14;
        cur[q] = t[rb + ((4 * cg + q) ^ rm)];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double akk = t[tsw(k, k)];
        const double lrk = t[rb + (k ^ rm)];
        double lck[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) lck[q] = t[cb[q] + (k ^ cm[q])];
        const double id = rsqrt_nr(akk);
        idk[k] = id;
        const double lr = lrk * id;
        const double colv = (r == k) ? akk * id : lr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 4 * cg + q;
            const double upd = cur[q] - lr * (lck[q] * id);
            cur[q] = (c > k && r >= c) ? upd : cur[q];
            cur[q] = (c == k) ? colv : cur[q];
        }
        t[rb + ((4 * cg + ((k + 1) & 3)) ^ rm)] = cur[(k + 1) & 3];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) t[rb + ((4 * cg + q) ^ rm)] = cur[q];      // the factor, for the substitution below
    // column c of inv(L) by forward substitution; row i of L is fetched in one batch (uniform addresses)
    const int c = r;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double lrow[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) lrow[k] = (k < i) ? t[tsw(i, k)] : 0.0;
        double sacc = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) sacc -= lrow[k] * x[k];
        x[i] = (i >= c) ? sacc * idk[i] : 0.0;
    }
    if (lane < 16) {
        if (Lg) {
#pragma unroll
            for (int i = 0; i < 16; ++i) Lg[(long long)c * ldl + i] = (i <= c) ? t[tsw(c, i)] : 0.0;   // row c of L
        }
        // every lane has read what it needs of L before the tile is overwritten (same wavefront, in order)
#pragma unroll
        for (int i = 0; i < 16; ++i) t[tsw(i, c)] = x[i];
    }
}

template <int MT>
__global__ void __launch_bounds__(512) gain_solver_kernel(const double* __restrict__ S, const double* __restrict__ W,
                                                           const double* __restrict__ rdiag, double* __restrict__ K,
                                                           double* __restrict__ KR, double* __restrict__ L, int ldm,
                                                           long long sM, long long sNM, int NT)
{
    extern __shared__ double lds[];                     // MT (MT + 1) / 2 tiles of 256 doubles + 16
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const double* Sb = S + (long long)b * sM;
    double* Lb = L ? L + (long long)b * sM : nullptr;
    constexpr int NTILES = MT * (MT + 1) / 2;

    // ---- load the lower tiles of S (accumulator-shaped: lanes along columns) -------------------
    for (int t = wave; t < NTILES; t += 8) {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        const int j = t - i * (i + 1) / 2;
        double* tl = lds + t * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = lk + 4 * r;
            tl[tsw(row, li)] = Sb[(long long)(16 * i + row) * ldm + 16 * j + li];
        }
    }

    // ---- phase 1: tile Cholesky ---------------------------------------------------------------------
    for (int j = 0; j < MT; ++j) {
        __syncthreads();
        double* tjj = lds + tile_index(j, j) * 256;
#ifndef SOLVER_ABLATE_DIAG
        if (wave == 0) diag_tile_factor_invert(tjj, Lb ? Lb + (long long)(16 * j) * ldm + 16 * j : nullptr, ldm, lane);
#endif
        __syncthreads();
        // panel: L_ij = S_ij inv(L_jj)^T
        for (int i = j + 1 + wave; i < MT; i += 8) {
            double* tij = lds + tile_index(i, j) * 256;
            v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
            acc = tile_mma_ldsBt<false>(tij, tjj, acc, li, lk);
#pragma unroll
            for (int r = 0; r < 4; ++r) tij[tsw(lk + 4 * r, li)] = acc[r];
        }
        __syncthreads();
        // trailing update: S_ik -= L_ij L_kj^T for j < k <= i
#ifdef SOLVER_ABLATE_TRAIL
        const int w = 0;
#else
        const int w = MT - j - 1;
#endif
        for (int t = wave; t < w * (w + 1) / 2; t += 8) {
            int ii = 0;
            while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
            const int kk = t - ii * (ii + 1) / 2;
            const int i = j + 1 + ii, k = j + 1 + kk;
            double* tik = lds + tile_index(i, k) * 256;
            v4d acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = tik[tsw(lk + 4 * r, li)];
            acc = tile_mma_ldsBt<true>(lds + tile_index(i, j) * 256, lds + tile_index(k, j) * 256, acc, li, lk);
#pragma unroll
            for (int r = 0; r < 4; ++r) tik[tsw(lk + 4 * r, li)] = acc[r];
        }
    }
    __syncthreads();
    if (Lb) {   // off-diagonal tiles of L (the diagonal ones were written while they were still L)
        for (int t = wave; t < NTILES; t += 8) {
            int i = 0;
            while ((i + 1) * (i + 2) / 2 <= t) ++i;
            const int j = t - i * (i + 1) / 2;
            if (i == j) continue;
            const double* tl = lds + t * 256;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = lk + 4 * r;
                Lb[(long long)(16 * i + row) * ldm + 16 * j + li] = tl[tsw(row, li)];
            }
        }
    }

    // ---- phase 2: K^T = L^-T L^-1 W^T, wavefront w owns rows [16 w, 16 w + 16) of W / K -------------
#ifdef SOLVER_ABLATE_TRSM
    return;
#endif
    if (wave >= NT) return;
    const double* Wb = W + (long long)b * sNM + (long long)(16 * wave + li) * ldm;   // row 16 w + li of W
    v4d y[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        v4d acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = Wb[16 * j + lk + 4 * r];                  // W^T tile j, accumulator-shaped
#pragma unroll
        for (int k = 0; k < j; ++k) acc = tile_mma_regB<false, true>(lds + tile_index(j, k) * 256, y[k], acc, li, lk);
        y[j] = tile_mma_regB<false, false>(lds + tile_index(j, j) * 256, acc, (v4d){0.0, 0.0, 0.0, 0.0}, li, lk);
    }
#pragma unroll
    for (int j = MT - 1; j >= 0; --j) {
        v4d acc = y[j];
#pragma unroll
        for (int k = j + 1; k < MT; ++k) acc = tile_mma_regB<true, true>(lds + tile_index(k, j) * 256, y[k], acc, li, lk);
        y[j] = tile_mma_regB<true, false>(lds + tile_index(j, j) * 256, acc, (v4d){0.0, 0.0, 0.0, 0.0}, li, lk);
    }
    double* Kb = K + (long long)b * sNM + (long long)(16 * wave + li) * ldm;
    double* KRb = KR ? KR + (long long)b * sNM + (long long)(16 * wave + li) * ldm : nullptr;
    const double* rd = rdiag + (long long)b * ldm;
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = 16 * j + lk + 4 * r;
            Kb[q] = y[j][r];
            if (KRb) KRb[q] = y[j][r] * rd[q];
        }
}

}  // namespace slamk

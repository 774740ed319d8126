// Fused gain solver for small filters: K = W S^-1 (and K R) from S = L L^T, one workgroup per filter.
//
//   in : P  n x n, the compact rows of H, r[m] (diagonal of R); W = P H^T and S = H W + R are formed in LDS
//   out: K  n x m, optionally KR = K diag(r) and L (lower)
//   m = 16 MT with MT <= 10 when the lower tiles of S, 2 KB each, all live in LDS (MT (MT+1) / 2 <= 55 tiles), MT <= 15
//   with the finished tiles of the factor partly in an image in global memory (SPILL),
//   n = 16 NT with NT <= 8 (one wavefront per 16 columns of W^T).
//
// Tile Cholesky in LDS.  Per tile column j: wavefront 0 factors the diagonal tile and replaces it by its inverse;
//   the tiles below become S_ij inv(L_jj)^T.  gain_solver_kernel applies the updates S_ik -= L_ij L_kj^T
//   left-looking with one column of look-ahead and forms the columns of S beside the factorisation (its "Pipelined
//   form"); tile_cholesky_forward, which chol_inv_block_kernel uses on a block that is already in memory, applies them
//   right-looking, column j + 1 first, so that wavefront 0 can factor the next diagonal tile while the other
//   wavefronts finish the trailing update and advance the forward solve.  All tile products are
//   v_mfma_f64_16x16x4_f64.
// Two triangular solves on the transposed system, so that finished tiles are consumed straight from the
//   accumulator registers: an MFMA result tile (lane = column, register = row group) IS the B operand of
//   the next MFMA, no data movement.
//            Y^T_j = inv(L_jj)   (W^T_j - sum_{k<j} L_jk   Y^T_k)      j = 0 .. MT-1   (beside the factorisation)
//            K^T_j = inv(L_jj)^T (Y^T_j - sum_{k>j} L_kj^T K^T_k)      j = MT-1 .. 0
//   Each of wavefronts 1..7 (and 0 for an eighth block) owns 16 rows of W / K.
//
// LDS tile image: element (r, c) of a 16 x 16 tile at r * 16 + (c ^ (r & 14)).  With that XOR every
// access pattern used here -- row-fragment reads (lane = row, 4 s + kk = column), the same on the
// transposed tile, and accumulator-shaped reads and writes -- touches 32 distinct 8-byte banks per
// half-wavefront.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "feature_kernels.hip.h"
#include "gemm.hip.h"

// Timing builds (scripts/solver_bench.hip) define SOLVER_STAMP(i) to record the cycle counter of one workgroup.
#ifndef SOLVER_STAMP
#define SOLVER_STAMP(i)
#endif

namespace slamk {

__device__ __forceinline__ int tsw(int r, int c) { return r * 16 + (c ^ (r & 14)); }
__device__ __forceinline__ int tile_index(int i, int j) { return i * (i + 1) / 2 + j; }

// D = A * B + C on 16 x 16 tiles; A read from an LDS tile, rows = lanes (optionally the transposed
// tile, optionally negated); B is an accumulator-shaped register tile.
// A negated operand is the instruction's own negate bit (last argument of the builtin: 1 = A, 2 = B, 4 = C), never an
// instruction of the vector ALU: on gfx950 those do not issue beside an fp64 matrix instruction of the same SIMD.
template <bool TRANS, bool NEG>
__device__ __forceinline__ v4d tile_mma_regB(const double* __restrict__ ta, v4d b, v4d c, int li, int lk)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double a = TRANS ? ta[tsw(4 * s + lk, li)] : ta[tsw(li, 4 * s + lk)];
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[s], c, 0, 0, NEG ? 1 : 0);
    }
    return c;
}

// D = (+/-) A * Bt^T + C with both operands LDS tiles read by rows (B[k][j] = tb[j][k]).  Each MFMA sums over four
// values of k, one per lane group lk; WHICH four is free as long as both operands agree, so lane group lk takes
// k = 4 lk + s in step s: its four operand values are neighbours in the tile image and come with two 16-byte reads.
template <bool NEG>
__device__ __forceinline__ v4d tile_mma_ldsBt(const double* __restrict__ ta, const double* __restrict__ tb, v4d c,
                                              int li, int lk)
{
    const int o0 = li * 16 + ((4 * lk) ^ (li & 14)), o1 = li * 16 + ((4 * lk + 2) ^ (li & 14));
    const double2 a01 = *reinterpret_cast<const double2*>(ta + o0), a23 = *reinterpret_cast<const double2*>(ta + o1);
    const double2 b01 = *reinterpret_cast<const double2*>(tb + o0), b23 = *reinterpret_cast<const double2*>(tb + o1);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.x, b01.x, c, 0, 0, NEG ? 1 : 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.y, b01.y, c, 0, 0, NEG ? 1 : 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.x, b23.x, c, 0, 0, NEG ? 1 : 0);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.y, b23.y, c, 0, 0, NEG ? 1 : 0);
    return c;
}

// 1/sqrt(a) to double precision: hardware estimate (v_rsq_f64) + one Halley-type step.
__device__ __forceinline__ double rsqrt_nr(double a)
{
    // one third-order step: with r = 1 - a y^2 the exact factor is (1 - r)^-1/2 = 1 + r / 2 + 3 r^2 / 8 + O(r^3); the
    // estimate is good to 5e-8, so the remainder is far below rounding (measured 2.3e-16 over 40 binades), and the
    // chain is five dependent operations instead of the eight of two Newton steps.
    double y = __builtin_amdgcn_rsq(a);
    const double r = fma(-(a * y), y, 1.0);
    return fma(y, r * fma(r, 0.375, 0.5), y);
}

// Lower triangle of a 4 x 4 block packed by rows: (i, j) at i (i + 1) / 2 + j.
// l = chol(d), id[k] = 1 / l_kk.  A pivot that is not positive turns into NaN or infinity and spreads from there.
__device__ __forceinline__ void chol4(const double* d, double* l, double* id)
{
    id[0] = rsqrt_nr(d[0]);
    l[0] = d[0] * id[0];
    l[1] = d[1] * id[0]; l[3] = d[3] * id[0]; l[6] = d[6] * id[0];
    const double a11 = fma(-l[1], l[1], d[2]);
    id[1] = rsqrt_nr(a11);
    l[2] = a11 * id[1];
    l[4] = fma(-l[3], l[1], d[4]) * id[1];
    l[7] = fma(-l[6], l[1], d[7]) * id[1];
    const double a22 = fma(-l[4], l[4], fma(-l[3], l[3], d[5]));
    id[2] = rsqrt_nr(a22);
    l[5] = a22 * id[2];
    l[8] = fma(-l[7], l[4], fma(-l[6], l[3], d[8])) * id[2];
    const double a33 = fma(-l[8], l[8], fma(-l[7], l[7], fma(-l[6], l[6], d[9])));
    id[3] = rsqrt_nr(a33);
    l[9] = a33 * id[3];
}

// x = a inv(l)^T for one row a of four entries.
__device__ __forceinline__ void solve4(const double* a, const double* l, const double* id, double* x)
{
    x[0] = a[0] * id[0];
    x[1] = fma(-x[0], l[1], a[1]) * id[1];
    x[2] = fma(-x[1], l[4], fma(-x[0], l[3], a[2])) * id[2];
    x[3] = fma(-x[2], l[8], fma(-x[1], l[7], fma(-x[0], l[6], a[3]))) * id[3];
}

// x = inv(l), packed like l.
__device__ __forceinline__ void inv4(const double* l, const double* id, double* x)
{
    x[0] = id[0]; x[2] = id[1]; x[5] = id[2]; x[9] = id[3];
    x[1] = -id[1] * (l[1] * x[0]);
    x[3] = -id[2] * fma(l[4], x[1], l[3] * x[0]);
    x[6] = -id[3] * fma(l[8], x[3], fma(l[7], x[1], l[6] * x[0]));
    x[4] = -id[2] * (l[4] * x[2]);
    x[7] = -id[3] * fma(l[8], x[4], l[7] * x[2]);
    x[8] = -id[3] * (l[8] * x[5]);
}

// Two adjacent columns (c0 even) of row r of a tile in one 16-byte LDS access: the XOR of the tile image only
// touches bits 1..3 of the column, so columns c0 and c0 + 1 stay neighbours.
__device__ __forceinline__ double2 tile_ld2(const double* t, int r, int c0)
{
    return *reinterpret_cast<const double2*>(t + r * 16 + (c0 ^ (r & 14)));
}
__device__ __forceinline__ void tile_st2(double* t, int r, int c0, double a, double b)
{
    *reinterpret_cast<double2*>(t + r * 16 + (c0 ^ (r & 14))) = make_double2(a, b);
}
// four adjacent columns c0 .. c0 + 3 (c0 a multiple of 4) of row r
__device__ __forceinline__ void tile_ld4(const double* t, int r, int c0, double* v)
{
    const double2 a = tile_ld2(t, r, c0), b = tile_ld2(t, r, c0 + 2);
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
__device__ __forceinline__ void tile_st4(double* t, int r, int c0, const double* v)
{
    tile_st2(t, r, c0, v[0], v[1]);
    tile_st2(t, r, c0 + 2, v[2], v[3]);
}

// g[q] = x of the lane with the same position in 16-lane row q (q = 0..3): an all-gather over the four rows of a
// wavefront with the row-swap instructions of gfx950 (v_permlane16_swap, v_permlane32_swap), no LDS access.
__device__ __forceinline__ void rows_allgather(double x, double (&g)[4])
{
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);      // [0]: rows 0 0 2 2, [1]: rows 1 1 3 3
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const auto a0 = __builtin_amdgcn_permlane32_swap(a[0], a[0], false, false); // [0]: row 0, [1]: row 2
    const auto a1 = __builtin_amdgcn_permlane32_swap(a[1], a[1], false, false); // [0]: row 1, [1]: row 3
    const auto b0 = __builtin_amdgcn_permlane32_swap(b[0], b[0], false, false);
    const auto b1 = __builtin_amdgcn_permlane32_swap(b[1], b[1], false, false);
    g[0] = __hiloint2double((int)b0[0], (int)a0[0]);
    g[1] = __hiloint2double((int)b1[0], (int)a1[0]);
    g[2] = __hiloint2double((int)b0[1], (int)a0[1]);
    g[3] = __hiloint2double((int)b1[1], (int)a1[1]);
}

// x of lane `src` (a constant) in every lane
__device__ __forceinline__ double lane_bcast(double x, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), src), __builtin_amdgcn_readlane(__double2loint(x), src));
}

__device__ __forceinline__ double pick4(const double* v, int k)
{
    return k == 0 ? v[0] : (k == 1 ? v[1] : (k == 2 ? v[2] : v[3]));
}

// One wavefront: Cholesky of the symmetric 16 x 16 tile held in registers in accumulator layout (lane (li, lk),
// register r: element (lk + 4 r, li)), and inv(L) (lower, zero above) written to the LDS tile t.  Lg != nullptr
// receives L (lower, zero above) at Lg[row * ldl + col].  Nothing goes through LDS on the way: it is the latency
// chain of the whole tile Cholesky, and LDS round trips beside seven busy wavefronts were most of its time.
// Right-looking in four steps of four columns.  Step kb:
//   - the 4 x 4 pivot block comes to every lane through scalar registers (v_readlane);
//   - the lane's own row of the pivot columns, A[li][4 kb + q] = A[4 kb + q][li], sits in register kb of the four
//     lanes (li, q): one all-gather over the 16-lane rows; every lane solves its row against the pivot block,
//     X = A[li][4 kb ..] inv(L_kk)^T = L[li][4 kb ..];
//   - the trailing update A -= X X^T is ONE v_mfma_f64_16x16x4_f64 (k = the four columns of the step), operand value
//     X[li][lk] for both operands;
//   - the inverse advances beside it on the residual R = I - L_(columns done) X_(rows done):
//     rows 4 kb .. 4 kb + 3 of inv(L) = inv(L_kk) R[4 kb .., :] (all-gather + 4 x 4 product), then
//     R -= L[:, 4 kb ..] X_kb, again one MFMA.
// The two chains (factor, inverse) only share X and interleave.  A pivot that is not positive turns into NaN.
__device__ __forceinline__ void diag_tile_factor_invert_acc(v4d a, double* __restrict__ t, double* __restrict__ Lg,
                                                            int ldl, int lane, double* __restrict__ t2 = nullptr)
{
    const int li = lane & 15, lk = lane >> 4;
    v4d R, out;
#pragma unroll
    for (int r = 0; r < 4; ++r) R[r] = (lk + 4 * r == li) ? 1.0 : 0.0;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        double d[10], l[10], id[4], g[4], x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q <= i; ++q) d[i * (i + 1) / 2 + q] = lane_bcast(a[kb], 4 * kb + q + 16 * i);
        rows_allgather(a[kb], g);
        chol4(d, l, id);
        solve4(g, l, id, x);
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = (li >= 4 * kb + q) ? x[q] : 0.0;   // L is lower triangular; rows above: done
        if (Lg && lk == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Lg[(long long)li * ldl + 4 * kb + q] = x[q];
        }
        const double xo = pick4(x, lk);
        if (kb < 3) a = __builtin_amdgcn_mfma_f64_16x16x4f64(xo, xo, a, 0, 0, 1);
        double iv[10], rr[4], xi[4];
        inv4(l, id, iv);
        rows_allgather(R[kb], rr);
        xi[0] = iv[0] * rr[0];
        xi[1] = fma(iv[2], rr[1], iv[1] * rr[0]);
        xi[2] = fma(iv[5], rr[2], fma(iv[4], rr[1], iv[3] * rr[0]));
        xi[3] = fma(iv[9], rr[3], fma(iv[8], rr[2], fma(iv[7], rr[1], iv[6] * rr[0])));
        const double xs = pick4(xi, lk);
        out[kb] = xs;
        if (kb < 3) R = __builtin_amdgcn_mfma_f64_16x16x4f64(xo, xs, R, 0, 0, 1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) t[tsw(lk + 4 * r, li)] = out[r];
    if (t2) {                                            // second image of inv(L) (see SPILL in gain_solver_kernel)
#pragma unroll
        for (int r = 0; r < 4; ++r) t2[tsw(lk + 4 * r, li)] = out[r];
    }
}

// Tile Cholesky of the MT x MT lower tiles in LDS (tile (i, j) at tile_index(i, j)) together with the forward
// solve Y^T = inv(L) B^T for up to eight 16-column blocks of B^T, one per wavefront.
//   before: nothing (the prologue builds the S tiles);
//   after : off-diagonal tiles hold L, diagonal tiles hold inv(L_jj), y[j] holds tile j of Y^T of the
//           wavefront's block (accumulator layout), and a barrier has been passed.
// On entry y[j] holds tile j of B^T for the block of wavefronts 1..7; wavefront 0 (an eighth block) fetches
// its tiles with late_tile(j) after the factorisation, so that they do not occupy registers during it.
// Wavefront 0 runs the chain
//   L_j+1,j -> S_j+1,j+1 -> factor -> inverse   one tile column ahead of the others; per tile column two
// barriers:  A  inv(L_jj) is in place and step j - 1 is complete;  B  the panel tiles L_ij are final.
// prologue(tag) runs first in either role and leaves the S tiles in LDS (ending with a barrier); with
// tag = std::true_type it also fills y (wavefronts 1..7), with std::false_type it must not touch y (wavefront 0):
// the tiles of B^T are then not live across the diagonal-tile code.
template <int MT, class Prologue, class LateTile>
__device__ __forceinline__ void tile_cholesky_forward(double* lds, v4d (&y)[MT], bool has_blk, double* Lb, int ldl,
                                                      int wave, int lane, Prologue prologue, LateTile late_tile)
{
    constexpr int NTILES = MT * (MT + 1) / 2;
    const int li = lane & 15, lk = lane >> 4;
    auto panel_tile = [&](int i, int j) {                // L_ij = S_ij inv(L_jj)^T
        double* tij = lds + tile_index(i, j) * 256;
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
        acc = tile_mma_ldsBt<false>(tij, lds + tile_index(j, j) * 256, acc, li, lk);
#pragma unroll
        for (int r = 0; r < 4; ++r) tij[tsw(lk + 4 * r, li)] = acc[r];
    };
    auto trail = [&](int i, int k, int j) {              // S_ik -= L_ij L_kj^T
        double* tik = lds + tile_index(i, k) * 256;
        v4d acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = tik[tsw(lk + 4 * r, li)];
        acc = tile_mma_ldsBt<true>(lds + tile_index(i, j) * 256, lds + tile_index(k, j) * 256, acc, li, lk);
#pragma unroll
        for (int r = 0; r < 4; ++r) tik[tsw(lk + 4 * r, li)] = acc[r];
    };
    // Y^T_j = inv(L_jj) (B^T_j - sum_{k<j} L_jk Y^T_k): a finished tile, lane = column, register = row group,
    // IS the B operand of the next MFMA.
    // The sum is ONE chain of matrix instructions on one accumulator: dependent fp64 matrix instructions run back to
    // back on gfx950 (64 cycles each), a second chain would only cost the additions that join them.
    // The operand fragment of the next tile is requested before the products of the current one are issued, so
    // that its LDS latency passes behind them.
    auto forward_step = [&](int j, v4d acc) {
        double an[4];
        auto fetch = [&](int k) {                        // row fragment of L_jk (k < j) or of inv(L_jj) (k == j)
            const double* ta = lds + tile_index(j, k) * 256;
#pragma unroll
            for (int s = 0; s < 4; ++s) an[s] = ta[tsw(li, 4 * s + lk)];
        };
        fetch(j > 0 ? 0 : j);
#pragma unroll
        for (int k = 0; k < MT; ++k)
            if (k < j) {
                const double a0 = an[0], a1 = an[1], a2 = an[2], a3 = an[3];
                fetch(k + 1);                            // k + 1 == j: the diagonal tile
                v4d& c = acc;
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, y[k][0], c, 0, 0, 1);
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, y[k][1], c, 0, 0, 1);
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, y[k][2], c, 0, 0, 1);
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, y[k][3], c, 0, 0, 1);
            }
        v4d out = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) out = __builtin_amdgcn_mfma_f64_16x16x4f64(an[s], acc[s], out, 0, 0, 0);
        return out;
    };

    if (wave == 0) {
        prologue(std::false_type{});
        SOLVER_STAMP(1);
        {
            v4d a0;
#pragma unroll
            for (int r = 0; r < 4; ++r) a0[r] = lds[tsw(lk + 4 * r, li)];
            diag_tile_factor_invert_acc(a0, lds, Lb, ldl, lane);
        }
        for (int j = 0; j < MT; ++j) {
            __syncthreads();                             // A
            if (j + 1 < MT) panel_tile(j + 1, j);
            __syncthreads();                             // B
            if (j + 1 < MT) {                            // S_j+1,j+1 -= L_j+1,j L_j+1,j^T stays in registers
                double* tjj = lds + tile_index(j + 1, j + 1) * 256;
                const double* tp = lds + tile_index(j + 1, j) * 256;
                v4d acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = tjj[tsw(lk + 4 * r, li)];
                acc = tile_mma_ldsBt<true>(tp, tp, acc, li, lk);
                diag_tile_factor_invert_acc(acc, tjj, Lb ? Lb + (long long)(16 * (j + 1)) * ldl + 16 * (j + 1) : nullptr,
                                            ldl, lane);
            }
            SOLVER_STAMP(2 + j);
        }
        __syncthreads();
        if (has_blk) {                                   // an eighth block: its forward solve runs here
#pragma unroll
            for (int j = 0; j < MT; ++j) y[j] = forward_step(j, late_tile(j));
        }
    } else {
        prologue(std::true_type{});
        SOLVER_STAMP(1);
        // Wavefront 4 shares its SIMD with wavefront 0, which runs the diagonal-tile chain: it takes no panel or
        // trailing tiles.
        const int wk6 = wave < 4 ? wave - 1 : (wave == 4 ? 6 : wave - 2);  // 0..5 for wavefronts 1 2 3 5 6 7, 6 for 4
        const bool worker = wave != 4;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            // On the first columns the other wavefronts' trailing tiles, not the chain of wavefront 0, set the pace:
            // wavefront 4 joins them there.
            const int nw = (2 * j + 4 < MT) ? 7 : 6;
            __syncthreads();                             // A
            if (worker)
                for (int i = j + 2 + wk6; i < MT; i += 6) panel_tile(i, j);
            __syncthreads();                             // B
            if (!worker && nw == 7) {
                const int w = MT - j - 2;
                for (int t = 6; t < w * (w + 1) / 2; t += 7) {
                    int ii = 0;
                    while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
                    const int kk = t - ii * (ii + 1) / 2;
                    trail(j + 2 + ii, j + 2 + kk, j);
                }
            }
            if (worker) {
                for (int i = j + 2 + wk6; i < MT; i += 6) trail(i, j + 1, j);      // column j + 1 first
                const int w = MT - j - 2;
                for (int t = wk6; t < w * (w + 1) / 2; t += nw) {
                    int ii = 0;
                    while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
                    const int kk = t - ii * (ii + 1) / 2;
                    trail(j + 2 + ii, j + 2 + kk, j);
                }
                if (has_blk) y[j] = forward_step(j, y[j]);
            }
            // wavefront 4 only advances its forward solve: measured faster inside the loop, beside the chain of
            // wavefront 0 on the same SIMD, than after it with the other wavefronts idle (1.325 -> 1.277 ms)
            if (!worker && has_blk) y[j] = forward_step(j, y[j]);
            SOLVER_STAMP(2 + j);
        }
        __syncthreads();
    }
}

// Off-diagonal tiles of L from LDS to global memory (the diagonal ones are written by the wavefront that
// factors them, while they are still L).
template <int MT>
__device__ __forceinline__ void store_offdiag_tiles(const double* lds, double* Lb, int ldl, int wave, int lane)
{
    const int li = lane & 15, lk = lane >> 4;
    for (int t = wave; t < MT * (MT + 1) / 2; t += 8) {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        const int j = t - i * (i + 1) / 2;
        if (i == j) continue;
        const double* tl = lds + t * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = lk + 4 * r;
            Lb[(long long)(16 * i + row) * ldl + 16 * j + li] = tl[tsw(row, li)];
        }
    }
}

// LDS beyond the S tiles: scratch tile of the diagonal-tile routine + eight per-wavefront staging tiles during
// and after the factorisation; before it, the compact rows of H and one 16-column panel of W = P H^T.
// SPILL (MT > 10: S does not fit): the pipelined form with only SPILL_CAP tiles of S / L in LDS, see the kernel.
// WIDE (SPILL only): filters of up to 12 row blocks of 16 states; a panel of W is 12 tiles.
template <int MT, bool SPILL = false, bool WIDE = false>
struct SolverLds {
    static_assert(SPILL || !WIDE, "the wide form keeps the factor in a ring");
    static constexpr int M = 16 * MT;
    static constexpr int PANEL_TILES = WIDE ? 12 : 8;
    static constexpr int COEF_TILES = (15 * M / 2 + 255) / 256;          // ha, hb [M][3], r [M], cs [M] (int)
    // two panels of W (one barrier per panel instead of two) where 160 KB allow it
    // (SPILL also keeps columns 0..2 of P behind the panels: 2 tiles, 3 for 192 rows)
    static constexpr bool TWO_PANELS = SPILL || MT * (MT + 1) / 2 + COEF_TILES + 16 <= 80;
    static constexpr int EXTRA_TILES = COEF_TILES + (TWO_PANELS ? 2 : 1) * PANEL_TILES + (SPILL ? (WIDE ? 3 : 2) : 0);
    static constexpr int SPILL_CAP = 79 - EXTRA_TILES;                   // tiles of S / L resident at a time
    static constexpr int TILES = (SPILL ? SPILL_CAP : MT * (MT + 1) / 2) + EXTRA_TILES;
};
// first tile (in column order) of column c of the lower triangle
__host__ __device__ constexpr int solver_col_off(int MT, int c) { return (c < MT ? c : MT) * MT - (c < MT ? c : MT) * ((c < MT ? c : MT) - 1) / 2; }

// BORDER: the launch covers 16 MT rows of S as tiles and one more row, the altitude row, which the library keeps
// LAST (row m - 1).  For a filter with m - 1 == 16 MT that row is factored as a border of the tiled block,
//     S = [S_ff s; s^T sigma],  L = [L_ff 0; l^T lambda],  l = inv(L_ff) s,  lambda^2 = sigma - l^T l,
// and carried through the two solves on the vector pipeline:
//     y_a = (w_a - Y_f l) / lambda,   k_a = y_a / lambda,   K_f = (Y_f - k_a l^T) inv(L_ff),
// instead of costing a whole tile row and column for one measurement.  The altitude row of H is -e_2^T
// (src/slam/main.cpp:354), so s = -(row 2 of W_f)^T, hence l = -(row 2 of Y_f)^T: the forward solve has already
// produced it; w_a = -P[:, 2], sigma = P[2][2] + r_a.  Filters of the batch whose altitude row lies inside the tiles
// (m - 1 < 16 MT) take the plain path.
// Launches of up to three tiles per side (filters of up to about ten features) are chains of latency with little LDS: their
// registers are capped at 128 so that two workgroups share a CU (measured at 4 features, 4096 filters: 0.127 -> 0.089 ms).
// SPILL (launches of more than 10 tiles per side, whose S does not fit LDS): the pipelined form only needs the three
// tile columns in flight on chip.  The tiles of S / L take their places in LDS in the order they are formed (column by
// column) in a ring of SPILL_CAP tiles; every tile is also written to an image of the factor in global memory (Lsp, sSp
// doubles per filter, tile c_off(k) + i - k at 256 doubles each, same layout inside a tile) when it becomes final, and a
// finished tile whose place has since gone to a later column is read from there (it sits in L2).  Which it is follows
// from the tile's position in column order and the column being formed -- no flags.  At 12 tile columns 21 of the 78
// tiles end up outside LDS; during the factorisation only a few operand reads go there, in the backward solve a quarter.
// WIDE (with SPILL): 9 to 12 row blocks of 16 states.  Panels of W are 12 tiles, so S = H W sees every row of P, and the
// kernel only factors S: the two solves for K are left to gain_rows_kernel, launched next, which works from the image
// of the factor (a row of tiles of Y^T per wavefront for 15 tile columns is 120 registers this kernel does not have
// beside its panels, and its finished tiles mostly sit in L2 by then: the solves ran at a sixth of their speed here).
template <int MT, bool BORDER, bool SPILL = false, bool WIDE = false>
__global__ void __launch_bounds__(512, (MT <= 3 ? 4 : 2)) gain_solver_kernel(FilterView fv, double* __restrict__ K, double* __restrict__ KR,
                                                           double* __restrict__ L, long long sM, long long sNM, int NT,
                                                           double* __restrict__ Lsp = nullptr, long long sSp = 0)
{
    extern __shared__ double lds[];                     // SolverLds<MT, SPILL, WIDE>::TILES tiles of 256 doubles
    using Lds = SolverLds<MT, SPILL, WIDE>;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: what depends on it runs on the scalar unit)
    const int li = lane & 15, lk = lane >> 4;
    const int ldm = fv.ldm, ldn = fv.ldn;
    double* Lb = L ? L + (long long)b * sM : nullptr;
    constexpr int NTILES = MT * (MT + 1) / 2;
    constexpr int M = 16 * MT;
    // Wavefront 0 factors the diagonal tiles; wavefronts 1..7 own row blocks 0..6 of W / K (wavefront 0 takes
    // block 7 when there is one), so the solves run beside the factorisation.
    const int blk = (wave + 7) & 7;
    const bool has_blk = blk < NT;
    const int nl = 16 * NT;
    constexpr int PANEL = Lds::PANEL_TILES;
    constexpr int CAP = SPILL ? Lds::SPILL_CAP : NTILES;   // tiles of S / L in LDS
    double* scratch = lds + CAP * 256;
    double* stg = scratch + (1 + wave) * 256;           // per-wavefront staging tile (after the build phase)
    double* Lsb = SPILL ? Lsp + (long long)blockIdx.x * sSp : nullptr;
    // tile (i, k), i >= k, of S / L: its place in LDS (valid from the moment it is formed until the ring comes round)
    auto tslot = [&](int i, int k) -> double* {
        if constexpr (SPILL) return lds + ((k * MT - k * (k - 1) / 2 + (i - k)) % CAP) * 256;
        else return lds + tile_index(i, k) * 256;
    };
    // ... and where to read it from once it is final, while the tiles before `thr` (in column order) have left LDS
    auto tread = [&](int i, int k, int thr) -> const double* {
        if constexpr (SPILL) {
            const int t = k * MT - k * (k - 1) / 2 + (i - k);
            return t >= thr ? lds + (t % CAP) * 256 : Lsb + (long long)t * 256;
        } else
            return lds + tile_index(i, k) * 256;
    };
    // the tile's place in the global image (SPILL only)
    auto timage = [&](int i, int k) -> double* { return Lsb + (long long)(k * MT - k * (k - 1) / 2 + (i - k)) * 256; };
    // tiles that have left LDS while column c is the newest one formed / after the factorisation
    auto thr_at = [&](int c) { const int f = solver_col_off(MT, c + 1) - CAP; return f > 0 ? f : 0; };
    constexpr int THR_END = (SPILL && NTILES > CAP) ? NTILES - CAP : 0;
    const double* rd = fv.rdiag + (long long)b * ldm;
    const double* Pb = fv.P + (long long)b * ldn * ldn;

    // ---- W = P H^T (calcK, src/slam/main.cpp:545) one 16-column panel at a time, and from each panel the tiles
    //      of S = H W + R (:541-543) below the diagonal and the wavefronts' tiles of W^T.  Row q of H has entries on
    //      columns 0..2 and cs_q..cs_q+2 only: six products per element of W and of S.  Neither W nor S exists in HBM.
    double* cha = scratch;                               // [M][3]
    double* chb = cha + 3 * M;                           // [M][3]
    double* crd = chb + 3 * M;                           // [M]
    int* ccs = reinterpret_cast<int*>(crd + M);          // [M]
    double* pan0 = scratch + Lds::COEF_TILES * 256;  // two panels of W: 8 tiles (128 rows x 16 columns) each
    constexpr bool TWO_PANELS = Lds::TWO_PANELS;
    // The first loads of P (columns 0..2 of every row, and the first panel's feature columns, whose start comes
    // straight from global memory) are requested before the compact rows of H are staged: one global round trip
    // instead of three in a row at the start of the workgroup.
    // The panels are built by wavefronts 1..7 beside the factorisation, one column ahead of its need (two panel
    // buffers), and wavefront 0 only runs the chain of diagonal tiles; see "Pipelined form" below.
    static_assert(TWO_PANELS, "S and two panels of W do not fit LDS: launch with SPILL");
    constexpr int RS = 28, UR = WIDE ? 7 : 5;            // rows per pass of the building threads, passes (n <= 128 / 192)
    const int pt = tid - 64;                             // index among the building threads (wavefront 0: none)
    const bool builder = pt >= 0;
    const int ec = pt & 15, ea = pt >> 4;                // panel column, row within a group of RS rows
    // P[a][0..2] once (in registers; in LDS with SPILL, whose registers are needed for the longer rows of tiles of the
    // solves: 1.5 % slower per panel, measured at 10 tile columns), P[a][cs..cs+2] per panel
    constexpr bool PXLDS = SPILL;
    double px[PXLDS ? 1 : UR][3], pf[UR][3];
    double* pxs = pan0 + 2 * PANEL * 256;                // [nl][3] (PXLDS)
    // what the border step needs from memory is requested here, a whole factorisation ahead of its use
    double w_alt = 0.0, sigma_alt = 0.0;
    if constexpr (BORDER && !WIDE) {
        w_alt = has_blk ? -Pb[(long long)min(16 * blk + li, nl - 1) * ldn + 2] : 0.0;
        sigma_alt = Pb[2 * ldn + 2] + rd[M];
    }
    {
        const long long rb = (long long)b * ldm;
        if (builder) {
            const int cs0 = max(fv.hcs[rb + ec], 0);
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const double* row = Pb + (long long)min(ea + RS * u, nl - 1) * ldn;
                if constexpr (!PXLDS) { px[u][0] = row[0]; px[u][1] = row[1]; px[u][2] = row[2]; }
                pf[u][0] = row[cs0]; pf[u][1] = row[cs0 + 1]; pf[u][2] = row[cs0 + 2];
            }
        }
        if constexpr (PXLDS) {
            for (int t = tid; t < 3 * nl; t += 512) pxs[t] = Pb[(long long)(t / 3) * ldn + (t % 3)];
        }
        for (int t = tid; t < 3 * M; t += 512) { cha[t] = fv.ha[rb * 3 + t]; chb[t] = fv.hb[rb * 3 + t]; }
        for (int t = tid; t < M; t += 512) { crd[t] = rd[t]; ccs[t] = fv.hcs[rb + t]; }
    }
    __syncthreads();
    SOLVER_STAMP(0);
    v4d y[MT];
    double* Kpark = K + (long long)b * sNM + (long long)(16 * blk) * ldm;
    // ---- factorisation and Y^T = inv(L) W^T ------------------------------------------------------------------
    auto late_tile = [&](int j) {
        v4d t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = Kpark[(long long)li * ldm + 16 * j + lk + 4 * r];
        return t;
    };
    {
        // Pipelined form.  Column c of S is needed for the first time when column c - 1 of L is final, and all it
        // takes to form it is panel c of W: the build runs beside the factorisation instead of in front of it, and
        // the updates are applied left-looking with a look-ahead of one column.  During tile column j (after its
        // panel tiles L_ij are final) wavefronts 1..7
        //   - apply column j to column j + 1                        S_i,j+1 -= L_ij L_j+1,j^T   (the only update
        //     that has to wait for column j),
        //   - form the tiles of column j + 2 from panel j + 2 (written to LDS during column j - 1) and bring them up
        //     to date with columns 0..j,
        //   - write panel j + 3 of W into the buffer panel j + 1 has left and request the entries of P for panel j + 4,
        //   - advance the forward solve,
        // while wavefront 0 updates, factors and inverts the diagonal tile j + 1: what every tile column costs is the
        // chain of wavefront 0, and the build phase disappears behind it.  Same sums in the same order per tile as the
        // right-looking form (columns ascending).
        auto pan_of = [&](int c) { return pan0 + ((c & 1) ? PANEL * 256 : 0); };
        auto write_panel = [&](int p) {                  // panel p of W from the entries of P held in registers
            double* pan = pan_of(p);
            const int q = 16 * p + ec;
            const double a0 = cha[3 * q], a1 = cha[3 * q + 1], a2 = cha[3 * q + 2];
            const double b0 = chb[3 * q], b1 = chb[3 * q + 1], b2 = chb[3 * q + 2];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int a = ea + RS * u;
                double x0, x1, x2;
                if constexpr (PXLDS) {
                    const double* xr = pxs + 3 * min(a, nl - 1);
                    x0 = xr[0]; x1 = xr[1]; x2 = xr[2];
                } else {
                    x0 = px[u][0]; x1 = px[u][1]; x2 = px[u][2];
                }
                double v = x0 * a0;
                v = fma(x1, a1, v);
                v = fma(x2, a2, v);
                v = fma(pf[u][0], b0, v);
                v = fma(pf[u][1], b1, v);
                v = fma(pf[u][2], b2, v);
                if (a < nl) pan[(a >> 4) * 256 + tsw(a & 15, ec)] = v;
            }
        };
        auto load_panel = [&](int p) {                   // the entries of P in the observed feature's columns
            const int cs = max(ccs[16 * p + ec], 0);
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const double* row = Pb + (long long)min(ea + RS * u, nl - 1) * ldn + cs;
                pf[u][0] = row[0]; pf[u][1] = row[1]; pf[u][2] = row[2];
            }
        };
        auto form_tile = [&](int i, int c) {             // S tile (i, c) = H_i W_c + R; lane = row q of S, 4 columns each
            const double* pan = pan_of(c);
            double* tl = tslot(i, c);
            const int q = 16 * i + li;
            const int cs = max(ccs[q], 0);
            const double a0 = cha[3 * q], a1 = cha[3 * q + 1], a2 = cha[3 * q + 2];
            const double b0 = chb[3 * q], b1 = chb[3 * q + 1], b2 = chb[3 * q + 2];
            const double rq = crd[q];
            double x0[4], x1[4], x2[4], y0[4], y1[4], y2[4], v[4];
            tile_ld4(pan, 0, 4 * lk, x0);
            tile_ld4(pan, 1, 4 * lk, x1);
            tile_ld4(pan, 2, 4 * lk, x2);
            tile_ld4(pan + (cs >> 4) * 256, cs & 15, 4 * lk, y0);
            tile_ld4(pan + ((cs + 1) >> 4) * 256, (cs + 1) & 15, 4 * lk, y1);
            tile_ld4(pan + ((cs + 2) >> 4) * 256, (cs + 2) & 15, 4 * lk, y2);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double w = a0 * x0[r];
                w = fma(a1, x1[r], w);
                w = fma(a2, x2[r], w);
                w = fma(b0, y0[r], w);
                w = fma(b1, y1[r], w);
                w = fma(b2, y2[r], w);
                if (q == 16 * c + 4 * lk + r) w += rq;
                v[r] = w;
            }
            tile_st4(tl, li, 4 * lk, v);
        };
        auto panel_tile = [&](int i, int j) {            // L_ij = S_ij inv(L_jj)^T
            double* tij = tslot(i, j);
            v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
            acc = tile_mma_ldsBt<false>(tij, tslot(j, j), acc, li, lk);
#pragma unroll
            for (int r = 0; r < 4; ++r) tij[tsw(lk + 4 * r, li)] = acc[r];
            if constexpr (SPILL) {                       // final: second image in global memory
                double* tg = timage(i, j);
#pragma unroll
                for (int r = 0; r < 4; ++r) tg[tsw(lk + 4 * r, li)] = acc[r];
            }
        };
        // S_ic -= sum_{k = k0}^{k1} L_ik L_ck^T: one accumulator chain, k ascending; the operands
        // of the next product are requested before the matrix instructions of the current one are issued
        auto update_tile = [&](int i, int c, int k0, int k1, int thr) {
            double* tic = tslot(i, c);
            const int o0 = li * 16 + ((4 * lk) ^ (li & 14)), o1 = li * 16 + ((4 * lk + 2) ^ (li & 14));
            double2 a01, a23, b01, b23;
            auto fetch = [&](int k) {
                const double* ri = tread(i, k, thr);
                const double* rc = tread(c, k, thr);
                a01 = *reinterpret_cast<const double2*>(ri + o0);
                a23 = *reinterpret_cast<const double2*>(ri + o1);
                b01 = *reinterpret_cast<const double2*>(rc + o0);
                b23 = *reinterpret_cast<const double2*>(rc + o1);
            };
            fetch(k0);
            v4d acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = tic[tsw(lk + 4 * r, li)];
#pragma unroll
            for (int k = 0; k < MT; ++k)
                if (k >= k0 && k <= k1) {
                    const double x0 = a01.x, x1 = a01.y, x2 = a23.x, x3 = a23.y;
                    const double z0 = b01.x, z1 = b01.y, z2 = b23.x, z3 = b23.y;
                    if (k < k1) fetch(k + 1);
                    v4d& d = acc;
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, z0, d, 0, 0, 1);
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, z1, d, 0, 0, 1);
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, z2, d, 0, 0, 1);
                    d = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, z3, d, 0, 0, 1);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) tic[tsw(lk + 4 * r, li)] = acc[r];
        };
        auto forward_step = [&](int j, v4d acc, int thr) {   // as in tile_cholesky_forward
            double an[4];
            auto fetch = [&](int k) {
                const double* ta = tread(j, k, thr);
#pragma unroll
                for (int s = 0; s < 4; ++s) an[s] = ta[tsw(li, 4 * s + lk)];
            };
            fetch(j > 0 ? 0 : j);
#pragma unroll
            for (int k = 0; k < MT; ++k)
                if (k < j) {
                    const double a0 = an[0], a1 = an[1], a2 = an[2], a3 = an[3];
                    fetch(k + 1);
                    v4d& c = acc;
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, y[k][0], c, 0, 0, 1);
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, y[k][1], c, 0, 0, 1);
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, y[k][2], c, 0, 0, 1);
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, y[k][3], c, 0, 0, 1);
                }
            v4d out = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) out = __builtin_amdgcn_mfma_f64_16x16x4f64(an[s], acc[s], out, 0, 0, 0);
            return out;
        };
        if (wave == 0) {
            // tile (blk, c) of W for the eighth row block, parked in this block's rows of K until the factorisation is
            // done (wavefront 0 writes and reads them itself)
            auto park = [&](int c) {
                const double* pan = pan_of(c);
#pragma unroll
                for (int r = 0; r < 4; ++r) Kpark[(long long)li * ldm + 16 * c + lk + 4 * r] = pan[blk * 256 + tsw(li, lk + 4 * r)];
            };
            __syncthreads();                             // panel 0 is in LDS
            if constexpr (!WIDE) { if (has_blk) park(0); }
            __syncthreads();                             // column 0 of S and panel 1 are in LDS
            SOLVER_STAMP(1);
            if constexpr (!WIDE) { if (has_blk && 1 < MT) park(1); }
            {
                v4d a0;
#pragma unroll
                for (int r = 0; r < 4; ++r) a0[r] = lds[tsw(lk + 4 * r, li)];
                diag_tile_factor_invert_acc(a0, lds, Lb, ldm, lane, SPILL ? timage(0, 0) : nullptr);
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                __syncthreads();                         // A
                if (j + 1 < MT) panel_tile(j + 1, j);
                __syncthreads();                         // B
                if constexpr (!WIDE) { if (has_blk && j + 2 < MT) park(j + 2); }
                if (j + 1 < MT) {                        // the update with column j stays in registers
                    double* tjj = tslot(j + 1, j + 1);
                    const double* tp = tslot(j + 1, j);
                    v4d acc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = tjj[tsw(lk + 4 * r, li)];
                    acc = tile_mma_ldsBt<true>(tp, tp, acc, li, lk);
                    diag_tile_factor_invert_acc(acc, tjj, Lb ? Lb + (long long)(16 * (j + 1)) * ldm + 16 * (j + 1) : nullptr,
                                                ldm, lane, SPILL ? timage(j + 1, j + 1) : nullptr);
                }
                SOLVER_STAMP(2 + j);
            }
            __syncthreads();
            if constexpr (!WIDE) {
                if (has_blk) {
#pragma unroll
                    for (int j = 0; j < MT; ++j) y[j] = forward_step(j, late_tile(j), THR_END);
                }
            }
        } else {
            // wavefront 4 shares its SIMD with wavefront 0: it takes no panel tiles and no updates of column j + 1,
            // and a tile of the look-ahead column only when there are more than six
            const int wk6 = wave < 4 ? wave - 1 : (wave == 4 ? 6 : wave - 2);  // 0..5 for wavefronts 1 2 3 5 6 7, 6 for 4
            const bool worker = wave != 4;
            // the tiles of the look-ahead column go to one wavefront per SIMD first (their sums get long on the late
            // columns, where SIMD 0 has the most room): wavefronts 4 1 2 3 5 6 7
            const int wq = wave == 4 ? 6 : 5 - wk6;
            auto take_y = [&](int c) {                   // tile (blk, c) of W, transposed: accumulator layout of W^T
                const double* pan = pan_of(c);
#pragma unroll
                for (int r = 0; r < 4; ++r) y[c < MT ? c : 0][r] = pan[blk * 256 + tsw(li, lk + 4 * r)];
            };
            // WIDE: every tile of W waits in its place in K (row-major, the row block's rows) for gain_rows_kernel: the
            // seven building wavefronts copy panel c out when it is complete
            auto park_all = [&](int c) {
                const double* pan = pan_of(c);
                for (int t = wave - 1; t < NT; t += 7) {
                    double* kx = K + (long long)b * sNM + (long long)(16 * t) * ldm + 16 * c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) kx[(long long)li * ldm + lk + 4 * r] = pan[t * 256 + tsw(li, lk + 4 * r)];
                }
            };
            write_panel(0);
            if (1 < MT) load_panel(1);
            __syncthreads();                             // panel 0 is in LDS
            for (int i = wk6; i < MT; i += 7) form_tile(i, 0);
            if constexpr (!WIDE) { if (has_blk) take_y(0); } else park_all(0);
            if (1 < MT) {
                write_panel(1);
                if (2 < MT) load_panel(2);
            }
            __syncthreads();                             // column 0 of S and panel 1 are in LDS
            SOLVER_STAMP(1);
            if (1 < MT) {
                for (int i = 1 + wk6; i < MT; i += 7) form_tile(i, 1);
                if constexpr (!WIDE) { if (has_blk) take_y(1); } else park_all(1);
            }
            if (2 < MT) {
                write_panel(2);
                if (3 < MT) load_panel(3);
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                __syncthreads();                         // A
                if (worker)
                    for (int i = j + 2 + wk6; i < MT; i += 6) panel_tile(i, j);
                __syncthreads();                         // B
                if (worker)
                    for (int i = j + 2 + wk6; i < MT; i += 6) update_tile(i, j + 1, j, j, 0);
                if (j + 3 < MT) {
                    write_panel(j + 3);
                    if (j + 4 < MT) load_panel(j + 4);
                }
                if (j + 2 < MT) {
                    for (int i = j + 2 + wq; i < MT; i += 7) {
                        form_tile(i, j + 2);
                        update_tile(i, j + 2, 0, j, thr_at(j + 2));
                    }
                    if constexpr (!WIDE) { if (has_blk) take_y(j + 2); } else park_all(j + 2);
                }
                if constexpr (!WIDE) { if (has_blk) y[j] = forward_step(j, y[j], thr_at(j + 2)); }
                SOLVER_STAMP(2 + j);
            }
            __syncthreads();
        }
    }
    if constexpr (SPILL) {
        if (Lb)                                          // off-diagonal tiles of L for introspection
            for (int t = wave; t < NTILES; t += 8) {
                int i = 0;
                while ((i + 1) * (i + 2) / 2 <= t) ++i;
                const int j = t - i * (i + 1) / 2;
                if (i == j) continue;
                const double* tl = tread(i, j, THR_END);
#pragma unroll
                for (int r = 0; r < 4; ++r) Lb[(long long)(16 * i + lk + 4 * r) * ldm + 16 * j + li] = tl[tsw(lk + 4 * r, li)];
            }
    } else {
        if (Lb) store_offdiag_tiles<MT>(lds, Lb, ldm, wave, lane);
    }
    SOLVER_STAMP(20);
    if constexpr (WIDE) return;                          // K: gain_rows_kernel

    // ---- the altitude row as a border of the factored block -----------------------------------------------
    double k_alt = 0.0;                                  // K[16 blk + li][M], the same in the four lane groups
    bool border = false;
    if constexpr (BORDER) {
        border = (fv.m[b] - 1 == M);                     // uniform over the workgroup
        if (border) {
            double* lvec = scratch;                      // [M]: l; the diagonal-tile scratch tile is free by now
            const double w_a = w_alt, sigma = sigma_alt;
            if (blk == 0 && li == 2) {                   // row 2 of Y_f: lane = column 2 of the wavefront's Y^T tiles
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) lvec[16 * j + lk + 4 * r] = -y[j][r];
            }
            __syncthreads();
            double d = 0.0, ll = 0.0;
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double lq = lvec[16 * j + lk + 4 * r];
                    d = fma(y[j][r], lq, d);             // this lane group's share of (Y_f l)_a
                    ll = fma(lq, lq, ll);
                }
            d += __shfl_xor(d, 16); d += __shfl_xor(d, 32);
            ll += __shfl_xor(ll, 16); ll += __shfl_xor(ll, 32);
            const double lam2 = sigma - ll;
            if (!(lam2 > 0.0) && tid == 0) atomicOr(fv.status + b, 2);     // S was not positive definite
            const double ilam = rsqrt_nr(lam2);
            k_alt = has_blk ? (w_a - d) * ilam * ilam : 0.0;
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[j][r] = fma(-k_alt, lvec[16 * j + lk + 4 * r], y[j][r]);
            if (Lb && wave == 1) {                       // the factor's last row, for introspection
                for (int q = lane; q < M; q += 64) Lb[(long long)M * ldm + q] = lvec[q];
                if (lane == 0) Lb[(long long)M * ldm + M] = lam2 * ilam;
            }
        }
    }

    SOLVER_STAMP(21);
    // ---- K^T_j = inv(L_jj)^T (Y^T_j - sum_{k>j} L_kj^T K^T_k),  j = MT-1 .. 0 -----------------------------
    if (!has_blk) return;
    if (border && lk == 0) {                             // the altitude column of K
        K[(long long)b * sNM + (long long)(16 * blk + li) * ldm + M] = k_alt;
        if (KR) KR[(long long)b * sNM + (long long)(16 * blk + li) * ldm + M] = k_alt * rd[M];
    }
    // Every finished tile of K leaves at once, through the wavefront's staging tile (128-byte runs per row): its
    // stores drain while the products of the next tile run.
    double* Kb = K + (long long)b * sNM + (long long)(16 * blk + lk) * ldm + li;
    double* KRb = KR ? KR + (long long)b * sNM + (long long)(16 * blk + lk) * ldm + li : nullptr;
    double mag = fabs(k_alt);                            // a pivot that was not positive has made K NaN or infinite
#pragma unroll
    for (int j = MT - 1; j >= 0; --j) {
        v4d acc = y[j];
#pragma unroll
        for (int k = j + 1; k < MT; ++k) acc = tile_mma_regB<true, true>(tread(k, j, THR_END), y[k], acc, li, lk);
        y[j] = tile_mma_regB<true, false>(tread(j, j, THR_END), acc, (v4d){0.0, 0.0, 0.0, 0.0}, li, lk);
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[tsw(li, lk + 4 * r)] = y[j][r];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double v = stg[tsw(lk + 4 * r, li)];
            mag += fabs(v);
            Kb[(long long)(4 * r) * ldm + 16 * j] = v;
            if (KRb) KRb[(long long)(4 * r) * ldm + 16 * j] = v * rd[16 * j + li];
        }
    }
    if (!(mag < 1.0e300)) atomicOr(fv.status + b, 2);    // S was not positive definite
    SOLVER_STAMP(22);
}

// K for filters of 9 to 12 row blocks of 16 states, launched behind gain_solver_kernel<MT, BORDER, true, true>, which
// only factors S for them and leaves every tile of the factor (L_ij below the diagonal, inv(L_jj) on it) in the image
// Lsp.  A workgroup of four wavefronts takes up to four row blocks of W / K from blk0 on, one per wavefront:
//   - each wavefront takes its tiles of W = P H^T (src/slam/main.cpp:545) from its row block's rows of K, where the
//     solver kernel has left a copy of every panel it built (and where the gain replaces them at the end);
//   - the forward solve walks the rows of the factor, the backward solve its columns; the four wavefronts need the same
//     tiles at the same time, so the workgroup streams them through LDS: all 256 threads copy row j + 1 (column j - 1)
//     of tiles from the image (L2) into one buffer while the products of row j run from the other: one barrier and one
//     L2 latency per row of tiles instead of one per product;
//   - the border step needs l = -(row 2 of Y_f)^T, which the wavefront with row block 0 has after its forward solve:
//     with BORDER the launch with blk0 = 0 (one workgroup per filter, G = 1) passes it on in LDS and leaves it
//     behind the image's tiles for the launch with blk0 = 4, which follows on the same stream; without, one launch
//     (blk0 = 0) covers all row blocks.
// 256 registers per wavefront hold a whole row of tiles of Y^T; (2 MT + 5) tiles of LDS: two workgroups per CU.
template <int MT, bool BORDER>
__global__ void __launch_bounds__(256, 2) gain_rows_kernel(FilterView fv, double* __restrict__ K, double* __restrict__ KR, long long sNM,
                                                           int NT, int blk0, int G, double* __restrict__ Lsp, long long sSp,
                                                           double* __restrict__ Ldbg = nullptr, long long sM = 0)
{
    constexpr int M = 16 * MT, NTILES = MT * (MT + 1) / 2;
    extern __shared__ double lds[];                     // gain_rows_lds_tiles(MT) tiles
    double* buf[2] = {lds, lds + MT * 256};              // a row / column of tiles of the factor each
    double* stgs = lds + 2 * MT * 256;                   // one staging tile per wavefront
    double* lvec = stgs + 4 * 256;                       // [M]
    // The G workgroups of a filter stream the same image: they sit next to each other on one XCD (consecutive workgroup
    // numbers go round the eight XCDs), so the image comes from HBM once and from that XCD's L2 after that.
    const int wg = (int)blockIdx.x >> 3, by = wg % G;
    const int b = (wg / G) * 8 + ((int)blockIdx.x & 7);
    if (b >= fv.B) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int ldm = fv.ldm, ldn = fv.ldn;
    // (row blocks dealt to the workgroups of a launch round robin: 9 row blocks are 4, then 3 + 2; which wavefronts --
    // which SIMDs -- get them rotates with the workgroup, so that two workgroups sharing a CU load its SIMDs evenly)
    const int slot = (wave + b + 2 * by) & 3;
    const int blk = blk0 + by + G * slot;
    const bool has_blk = blk < NT;
    const bool first = blk0 == 0 && by == 0;             // this workgroup holds row block 0
    const long long rb = (long long)b * ldm;
    const double* rd = fv.rdiag + rb;
    const double* Pb = fv.P + (long long)b * ldn * ldn;
    double* Lsb = Lsp + (long long)b * sSp;
    bool border = false;
    if constexpr (BORDER) border = (fv.m[b] - 1 == M);   // uniform over the workgroup
    for (int t = tid; t < M; t += 256) lvec[t] = (border && !first) ? Lsb[(long long)NTILES * 256 + t] : 0.0;
    SOLVER_STAMP(1);
    // A wavefront without a row block only helps to stream the factor.  The two kinds of wavefront run separate copies of
    // what follows (same barriers): with the products inside `if (has_blk)` blocks, the compiler sinks the last
    // multiply-add of every entry of W into them and keeps its operands alive instead (2 KB of spills per lane).
    auto run = [&](auto solving) {
    constexpr bool SOLVE = decltype(solving)::value;
    const double* prow = Pb + (long long)(16 * min(blk, NT - 1) + li) * ldn;
    // tile (blk, c) of W, transposed (accumulator layout of W^T): lane = state row li, measurement rows lk + 4 r
    v4d y[MT];
    if constexpr (SOLVE) {                               // all requested at once: one trip to memory
        const double* wrow = K + (long long)b * sNM + (long long)(16 * blk + li) * ldm + lk;
#pragma unroll
        for (int c = 0; c < MT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) y[c][r] = wrow[16 * c + 4 * r];
    }
    SOLVER_STAMP(2);
    // place of tile (i, k), i >= k, in the image (column by column); the image is read through a buffer descriptor:
    // the tile's offset is a scalar operand of the load, the thread's element its only address register
    auto timg = [&](int i, int k) { return (k * MT - k * (k - 1) / 2 + (i - k)) * 256; };
    const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(Lsb, 0, (int)(sSp * 8), 0x00020000);
    typedef int v2i_t __attribute__((ext_vector_type(2)));
    auto img_ld = [&](int tile_off) {
        const v2i_t v = __builtin_amdgcn_raw_buffer_load_b64(img, tid * 8, tile_off * 8, 0);
        return __hiloint2double(v.y, v.x);
    };
    // this thread's element of every tile of a row / column, two rows ahead of the products: a row is requested two
    // steps before it is copied to LDS, so its trip from L2 (or HBM: the images of a whole batch do not fit L2) has two
    // steps of products to hide behind; the two register sets alternate with the parity of the row
    double ld[2][MT];
    // ---- Y^T_j = inv(L_jj) (W^T_j - sum_{k<j} L_jk Y^T_k), j ascending: as forward_step of gain_solver_kernel ----
    // (one barrier per row: the buffer written in step j was last read in step j - 2, which every wavefront has left
    // before it passed the barrier of step j - 1)
    ld[0][0] = img_ld(timg(0, 0));
    if (1 < MT) { ld[1][0] = img_ld(timg(1, 0)); ld[1][1] = img_ld(timg(1, 1)); }
    __syncthreads();                                     // (l is in LDS)
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        double* bj = buf[j & 1];
#pragma unroll
        for (int k = 0; k < MT; ++k)
            if (k <= j) bj[k * 256 + tid] = ld[j & 1][k];
        __syncthreads();
        if (j + 2 < MT) {
#pragma unroll
            for (int k = 0; k < MT; ++k)
                if (k <= j + 2) ld[j & 1][k] = img_ld(timg(j + 2, k));
        }
        if constexpr (SOLVE) {
            v4d acc = y[j];
            double an[4];
            auto fetch = [&](int k) {
                const double* ta = bj + k * 256;
#pragma unroll
                for (int s = 0; s < 4; ++s) an[s] = ta[tsw(li, 4 * s + lk)];
            };
            fetch(0);
#pragma unroll
            for (int k = 0; k < MT; ++k)
                if (k < j) {
                    const double a0 = an[0], a1 = an[1], a2 = an[2], a3 = an[3];
                    fetch(k + 1);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, y[k][0], acc, 0, 0, 1);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, y[k][1], acc, 0, 0, 1);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, y[k][2], acc, 0, 0, 1);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, y[k][3], acc, 0, 0, 1);
                }
            v4d out = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) out = __builtin_amdgcn_mfma_f64_16x16x4f64(an[s], acc[s], out, 0, 0, 0);
            y[j] = out;
        }
        __builtin_amdgcn_sched_barrier(0);               // (one row of the pipeline at a time in the scheduler's view)
    }
    SOLVER_STAMP(3);
    // the last two columns of the factor are requested before the border step
    ld[(MT - 1) & 1][MT - 1] = img_ld(timg(MT - 1, MT - 1));
    if (MT > 1) { ld[(MT - 2) & 1][MT - 2] = img_ld(timg(MT - 2, MT - 2)); ld[(MT - 2) & 1][MT - 1] = img_ld(timg(MT - 1, MT - 2)); }

    // ---- the altitude row as a border of the factored block (see gain_solver_kernel) ---------------------------------
    double k_alt = 0.0;
    if constexpr (BORDER) {
        if (border) {
            if (first) {
                if constexpr (SOLVE) {
                    if (blk == 0 && li == 2) {           // row 2 of Y_f: lane = column 2 of row block 0's Y^T tiles
#pragma unroll
                        for (int j = 0; j < MT; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                lvec[16 * j + lk + 4 * r] = -y[j][r];
                                Lsb[(long long)NTILES * 256 + 16 * j + lk + 4 * r] = -y[j][r];
                            }
                    }
                }
                __syncthreads();
            }
            if constexpr (SOLVE) {
                const double w_a = -prow[2], sigma = Pb[2 * ldn + 2] + rd[M];
                double d = 0.0, ll = 0.0;
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double lq = lvec[16 * j + lk + 4 * r];
                        d = fma(y[j][r], lq, d);         // this lane group's share of (Y_f l)_a
                        ll = fma(lq, lq, ll);
                    }
                d += __shfl_xor(d, 16); d += __shfl_xor(d, 32);
                ll += __shfl_xor(ll, 16); ll += __shfl_xor(ll, 32);
                const double lam2 = sigma - ll;
                if (first && !(lam2 > 0.0) && tid == 0) atomicOr(fv.status + b, 2);   // S was not positive definite
                const double ilam = rsqrt_nr(lam2);
                k_alt = (w_a - d) * ilam * ilam;
                if (Ldbg && first && blk == 0) {         // the factor's last row, for introspection
                    double* Lb = Ldbg + (long long)b * sM;
                    for (int q = lane; q < M; q += 64) Lb[(long long)M * ldm + q] = lvec[q];
                    if (lane == 0) Lb[(long long)M * ldm + M] = lam2 * ilam;
                }
                asm volatile("" ::: "memory");           // l is read again below, not kept in 2 M / 16 registers meanwhile
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[j][r] = fma(-k_alt, lvec[16 * j + lk + 4 * r], y[j][r]);
                if (lk == 0) {                           // the altitude column of K
                    K[(long long)b * sNM + (long long)(16 * blk + li) * ldm + M] = k_alt;
                    if (KR) KR[(long long)b * sNM + (long long)(16 * blk + li) * ldm + M] = k_alt * rd[M];
                }
            }
        }
    }
    // ---- K^T_j = inv(L_jj)^T (Y^T_j - sum_{k>j} L_kj^T K^T_k),  j = MT-1 .. 0 ------------------------------------------
    SOLVER_STAMP(4);
    double* stg = stgs + wave * 256;
    double* Kb = K + (long long)b * sNM + (long long)(16 * min(blk, NT - 1) + lk) * ldm + li;
    double* KRb = KR ? KR + (long long)b * sNM + (long long)(16 * min(blk, NT - 1) + lk) * ldm + li : nullptr;
    double mag = fabs(k_alt);                            // a pivot that was not positive has made K NaN or infinite
    __syncthreads();                                     // the last rows of the forward solve are done with
#pragma unroll
    for (int j = MT - 1; j >= 0; --j) {
        double* bj = buf[j & 1];
#pragma unroll
        for (int k = 0; k < MT; ++k)
            if (k >= j) bj[k * 256 + tid] = ld[j & 1][k];       // tile (k, j) in place k
        __syncthreads();
        if (j > 1) {
#pragma unroll
            for (int k = 0; k < MT; ++k)
                if (k >= j - 2) ld[j & 1][k] = img_ld(timg(k, j - 2));
        }
        if constexpr (SOLVE) {
            v4d acc = y[j];
#pragma unroll
            for (int k = j + 1; k < MT; ++k) acc = tile_mma_regB<true, true>(bj + k * 256, y[k], acc, li, lk);
            y[j] = tile_mma_regB<true, false>(bj + j * 256, acc, (v4d){0.0, 0.0, 0.0, 0.0}, li, lk);
#pragma unroll
            for (int r = 0; r < 4; ++r) stg[tsw(li, lk + 4 * r)] = y[j][r];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = stg[tsw(lk + 4 * r, li)];
                mag += fabs(v);
                Kb[(long long)(4 * r) * ldm + 16 * j] = v;
                if (KRb) KRb[(long long)(4 * r) * ldm + 16 * j] = v * rd[16 * j + li];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (SOLVE && !(mag < 1.0e300)) atomicOr(fv.status + b, 2);    // S was not positive definite
    SOLVER_STAMP(5);
    };
    if (has_blk) run(std::true_type{});
    else run(std::false_type{});
}
constexpr int gain_rows_lds_tiles(int MT) { return 2 * MT + 5; }

// Diagonal block of the blocked Cholesky of a large S: factor the 16 MT x 16 MT block at (j0, j0) (lower part
// valid) into L and write inv(L) of the block into Li, one workgroup per matrix of the batch.  The inverse is
// the forward solve above with the identity as right-hand side.
// (registers capped at 128: two workgroups share a CU when a batch of mid-size filters factors its blocks together)
// S points at the block's first element in matrix 0 of the batch and has its own leading dimension and batch stride
// (the block may sit in a panel buffer instead of a square matrix); L and Li are addressed as square matrices.
template <int MT>
__global__ void __launch_bounds__(512, 4) chol_inv_block_kernel(const double* __restrict__ S, int ldS, long long sS,
                                                              double* __restrict__ L, double* __restrict__ Li, int ld,
                                                              long long sM, int j0, int* __restrict__ status)
{
    extern __shared__ double lds[];                     // MT (MT + 1) / 2 tiles + one scratch tile
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const long long o = (long long)b * sM + (long long)j0 * ld + j0;
    const double* Sb = S + (long long)b * sS;
    double* Lb = L + o;
    double* Lib = Li + o;
    const int blk = (wave + 7) & 7;
    const bool has_blk = blk < MT;
    v4d y[MT];
    auto load = [&](auto with_y) {
        for (int t = wave; t < MT * (MT + 1) / 2; t += 8) {
            int i = 0;
            while ((i + 1) * (i + 2) / 2 <= t) ++i;
            const int j = t - i * (i + 1) / 2;
            double* tl = lds + t * 256;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = lk + 4 * r;
                // the diagonal tiles are completed from their lower triangle
                const int rr = (i == j && li > row) ? li : row, cc = (i == j && li > row) ? row : li;
                tl[tsw(row, li)] = Sb[(long long)(16 * i + rr) * ldS + 16 * j + cc];
            }
        }
        if constexpr (decltype(with_y)::value) {
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) y[j][r] = (j == blk && lk + 4 * r == li) ? 1.0 : 0.0;
        }
        __syncthreads();
    };
    tile_cholesky_forward<MT>(lds, y, has_blk, Lb, ld, wave, lane, load, [&](int j) {
        v4d t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = (j == blk && lk + 4 * r == li) ? 1.0 : 0.0;
        return t;
    });
    store_offdiag_tiles<MT>(lds, Lb, ld, wave, lane);
    if (!has_blk) return;
    double mag = 0.0;
#pragma unroll
    for (int j = 0; j < MT; ++j)         // tile (j, blk) of inv(L); zero above the diagonal
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            mag += fabs(y[j][r]);
            Lib[(long long)(16 * j + lk + 4 * r) * ld + 16 * blk + li] = y[j][r];
        }
    // a pivot that was not positive has made the inverse NaN or infinite: the block was not positive definite
    if (status && !(mag < 1.0e300)) atomicOr(status + b, 2);
}

}  // namespace slamk

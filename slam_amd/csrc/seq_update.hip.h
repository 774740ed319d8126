// Block-sequential measurement update for filters whose covariance fits LDS (update_form = 2), one workgroup per
// filter.  R is diagonal (initR, src/slam/main.cpp:449-476) and H is fixed during an update, so processing the
// measurement rows 16 at a time,
//     W_c = P H_c^T,  S_c = H_c W_c + R_c = L_c L_c^T,  K_c = W_c S_c^-1,  P <- P - K_c W_c^T,
//     dx <- dx + K_c (e_c - H_c dx),
// gives the same posterior covariance and the same total correction dx = K e as one gain for the whole measurement
// vector (calcK / updateP / K * eeMat, src/slam/main.cpp:221-228) in exact arithmetic; the gate and the mean update
// (:229-232) use the total dx as there.  It is NOT the reference's sequence of operations (one m x m system), so it
// is an option beside the default, held to the same tolerance.
// With Y_c = W_c L_c^-T:  K_c W_c^T = Y_c Y_c^T  and  K_c e' = Y_c (L_c^-1 e'), so the gain is never formed: per chunk
// one tile product per row block for Y_c and NT (NT + 1) / 2 for the symmetric downdate of P, instead of an m x m
// factorisation, two n x m triangular solves and three n x n products.  Neither S nor K exists in memory; P is read
// and written once.
//
// LDS: the lower tiles of P (NT (NT + 1) / 2, tile (I, J), I >= J, at tile_index(I, J); symmetric reads go through the
// transposed tile), the panel W_c / Y_c (NT tiles), inv(L_c), the chunk's compact rows of H (two buffers): 39 tiles at
// n <= 112, 69 at n = 160.  Per chunk four barriers: W_c and e' -> S_c, factor, inverse (one wavefront, in registers)
// -> Y_c and z -> P and dx.
#pragma once
#include <hip/hip_runtime.h>

#include "gain_solver.hip.h"
#include "structured_kernels.hip.h"

#ifndef SEQ_WAVES_PER_EU
#define SEQ_WAVES_PER_EU 2
#endif
#ifndef SEQ_STAMP
#define SEQ_STAMP(i)       // timing builds (scripts/seq_bench.hip) record the cycle counter of one workgroup here
#endif

namespace slamk {

template <int NT>
struct SeqUpdateLds {
    static constexpr int TILES = NT * (NT + 1) / 2 + NT + 4;
};

// The update of filter b by the whole workgroup.  PRELOADED: the lower tiles of the covariance are already in `lds`
// (and a barrier has been passed since they were written).
template <int NT, bool PRELOADED>
__device__ __forceinline__ void seq_update_filter(const FilterView& fv, int b, double* __restrict__ lds,
                                                   double* __restrict__ dx, int apply, int* __restrict__ applied)
{
    constexpr int NL = 16 * NT, NPT = NT * (NT + 1) / 2;
    double* wp = lds + NPT * 256;                        // W_c, then Y_c: NT tiles, rows = states, 16 columns
    double* linv = wp + NT * 256;                        // inv(L_c)
    double* tab = linv + 256;                            // [2][16][8]: ha[3], hb[3], e, r of the chunk's rows
    double* dxs = tab + 256;                             // [NL]
    double* ev = dxs + NL;                               // [16] e' = e_c - H_c dx
    double* zv = ev + 16;                                // [16] inv(L_c) e'
    int* tcs = reinterpret_cast<int*>(zv + 16);          // [2][16] first column of the feature entries (-1: none)
    static_assert(256 + NL + 16 + 16 + 16 <= 3 * 256, "LDS layout");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int ldn = fv.ldn;
    double* Pb = fv.P + (long long)b * ldn * ldn;
    const long long rb = (long long)b * fv.ldm;
    const int nf = fv.nf[b];
    const int m = fv.m[b];
    const int nchunk = (m + 15) / 16;                    // rows m .. 16 nchunk - 1: H = 0, e = 0, r = 1 (measure_kernel)

    auto psym = [&](int a, int c) -> double {            // P[a][c] from the lower tiles
        const int I = a >> 4, J = c >> 4;
        return I >= J ? lds[tile_index(I, J) * 256 + tsw(a & 15, c & 15)] : lds[tile_index(J, I) * 256 + tsw(c & 15, a & 15)];
    };
    // table rows of a chunk: one thread per row
    double trow[8];
    int tc = -1;
    auto request_rows = [&](int c) {
        if (tid < 16) {
            const long long q = rb + 16 * c + tid;
#pragma unroll
            for (int k = 0; k < 3; ++k) { trow[k] = fv.ha[q * 3 + k]; trow[3 + k] = fv.hb[q * 3 + k]; }
            trow[6] = fv.e[q]; trow[7] = fv.rdiag[q];
            tc = fv.hcs[q];
        }
    };
    auto store_rows = [&](int buf) {
        if (tid < 16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) tab[(buf * 16 + tid) * 8 + k] = trow[k];
            tcs[buf * 16 + tid] = tc;
        }
    };
    SEQ_STAMP(0);
    if (nchunk > 0) request_rows(0);
    if constexpr (!PRELOADED) {
        for (int t = wave; t < NPT; t += 8) {            // lower tiles of P
            int I = 0;
            while ((I + 1) * (I + 2) / 2 <= t) ++I;
            const int J = t - I * (I + 1) / 2;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                lds[t * 256 + tsw(lk + 4 * r, li)] = Pb[(long long)(16 * I + lk + 4 * r) * ldn + 16 * J + li];
        }
    }
    if (tid < NL) dxs[tid] = 0.0;
    store_rows(0);
    __syncthreads();
    SEQ_STAMP(1);

    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        const double* tb = tab + buf * 128;
        const int* cs = tcs + buf * 16;
        if (c + 1 < nchunk) request_rows(c + 1);
        if (c == 5) SEQ_STAMP(2);
        // ---- W_c = P H_c^T and e' = e_c - H_c dx ----------------------------------------------------------------------
        for (int idx = tid; idx < NL * 16; idx += 512) {
            const int a = idx >> 4, k = idx & 15;
            const double* h = tb + 8 * k;
            const int c0 = max(cs[k], 0);                // (no feature entries: hb = 0)
            double w = psym(a, 0) * h[0];
            w = fma(psym(a, 1), h[1], w);
            w = fma(psym(a, 2), h[2], w);
            w = fma(psym(a, c0), h[3], w);
            w = fma(psym(a, c0 + 1), h[4], w);
            w = fma(psym(a, c0 + 2), h[5], w);
            wp[(a >> 4) * 256 + tsw(a & 15, k)] = w;
        }
        if (tid >= 448 && tid < 464) {                   // (a wavefront with the fewest elements above)
            const int k = tid - 448;
            const double* h = tb + 8 * k;
            const int c0 = max(cs[k], 0);
            double v = h[6];
            v = fma(-h[0], dxs[0], v); v = fma(-h[1], dxs[1], v); v = fma(-h[2], dxs[2], v);
            v = fma(-h[3], dxs[c0], v); v = fma(-h[4], dxs[c0 + 1], v); v = fma(-h[5], dxs[c0 + 2], v);
            ev[k] = v;
        }
        __syncthreads();
        if (c == 5) SEQ_STAMP(4);
        // ---- S_c = H_c W_c + R_c in accumulator layout, factor, inverse: one wavefront, registers --------------------
        if (wave == 0) {
            v4d s;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = lk + 4 * r;                // row of S_c, column li
                const double* h = tb + 8 * q;
                const int c0 = max(cs[q], 0);
                double v = h[0] * wp[tsw(0, li)];
                v = fma(h[1], wp[tsw(1, li)], v);
                v = fma(h[2], wp[tsw(2, li)], v);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int a = c0 + k;
                    v = fma(h[3 + k], wp[(a >> 4) * 256 + tsw(a & 15, li)], v);
                }
                s[r] = v + (q == li ? h[7] : 0.0);
            }
            // S_c is symmetric up to rounding; the routine reads it as such (row li of a pivot column from column li)
            if (c == 5) SEQ_STAMP(5);
            diag_tile_factor_invert_acc(s, linv, nullptr, 0, lane);
        }
        __syncthreads();
        if (c == 5) SEQ_STAMP(7);
        // ---- Y_c = W_c inv(L_c)^T by row blocks, in place of W_c (Y^T = inv(L) W^T), and z = inv(L_c) e' ----------------
        // K_c W_c^T = Y_c Y_c^T and K_c e' = Y_c z: the gain itself is never formed.
        for (int blk = wave; blk < NT; blk += 8) {
            v4d wt;
#pragma unroll
            for (int r = 0; r < 4; ++r) wt[r] = wp[blk * 256 + tsw(li, lk + 4 * r)];
            const v4d yt = tile_mma_regB<false, false>(linv, wt, (v4d){0.0, 0.0, 0.0, 0.0}, li, lk);
#pragma unroll
            for (int r = 0; r < 4; ++r) wp[blk * 256 + tsw(li, lk + 4 * r)] = yt[r];
        }
        if (tid >= 448 && tid < 464) {                   // (wavefront 7 owns at most one row block)
            const int i = tid - 448;
            double v = 0.0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                double l4[4];
                tile_ld4(linv, i, 4 * g, l4);
#pragma unroll
                for (int k = 0; k < 4; ++k) v = fma(l4[k], ev[4 * g + k], v);
            }
            zv[i] = v;
            // the diagonal of inv(L_c) is 1 / l_kk: a pivot of S_c that was not positive has made it NaN (or, at zero,
            // infinite) -- the innovation covariance was not positive definite and no Cholesky gain exists
            const double dinv = linv[tsw(i, i)];
            if (!(dinv > 0.0 && dinv < 1.0e300)) atomicOr(fv.status + b, 2);
        }
        store_rows(buf ^ 1);
        __syncthreads();
        if (c == 5) SEQ_STAMP(9);
        // ---- P <- P - Y_c Y_c^T on the lower tiles; dx <- dx + Y_c z --------------------------------------------------
        // (up to four tiles of a wavefront at a time: independent chains of products instead of one)
        constexpr int PER_WAVE = (NPT + 7) / 8, PASSES = (PER_WAVE + 3) / 4, SLOTS = (PER_WAVE + PASSES - 1) / PASSES;
        for (int t0 = __builtin_amdgcn_readfirstlane(wave); t0 < NPT; t0 += 8 * SLOTS) {   // (scalar: the tile indices below too)
            v4d acc[SLOTS];
            double2 ya[SLOTS][2], yb[SLOTS][2];
            const int o0 = li * 16 + ((4 * lk) ^ (li & 14)), o1 = li * 16 + ((4 * lk + 2) ^ (li & 14));
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) {
                const int t = min(t0 + 8 * u, NPT - 1);  // (a slot past the end repeats the last tile and is not stored)
                int I = 0;
                while ((I + 1) * (I + 2) / 2 <= t) ++I;
                const int J = t - I * (I + 1) / 2;
                const double* pt = lds + t * 256;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[u][r] = pt[tsw(lk + 4 * r, li)];
                ya[u][0] = *reinterpret_cast<const double2*>(wp + I * 256 + o0);
                ya[u][1] = *reinterpret_cast<const double2*>(wp + I * 256 + o1);
                yb[u][0] = *reinterpret_cast<const double2*>(wp + J * 256 + o0);
                yb[u][1] = *reinterpret_cast<const double2*>(wp + J * 256 + o1);
            }
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[u][0].x, yb[u][0].x, acc[u], 0, 0, 1);
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[u][0].y, yb[u][0].y, acc[u], 0, 0, 1);
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[u][1].x, yb[u][1].x, acc[u], 0, 0, 1);
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[u][1].y, yb[u][1].y, acc[u], 0, 0, 1);
#pragma unroll
            for (int u = 0; u < SLOTS; ++u) {
                const int t = t0 + 8 * u;
                if (t < NPT) {
                    double* pt = lds + t * 256;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pt[tsw(lk + 4 * r, li)] = acc[u][r];
                }
            }
        }
        if (c == 5) SEQ_STAMP(10);
        if (tid < NL) {
            const double* yr = wp + (tid >> 4) * 256;
            double v = dxs[tid];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                double y4[4];
                tile_ld4(yr, tid & 15, 4 * g, y4);
#pragma unroll
                for (int k = 0; k < 4; ++k) v = fma(y4[k], zv[4 * g + k], v);
            }
            dxs[tid] = v;
        }
        __syncthreads();
        if (c == 5) SEQ_STAMP(11);
    }
    SEQ_STAMP(12);

    // ---- store: every lower tile with its mirror image; diagonal tiles from their lower half -------------------------
    if (dx && tid < NL) dx[(long long)b * ldn + tid] = dxs[tid];
    for (int t = wave; t < NPT; t += 8) {
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        const int J = t - I * (I + 1) / 2;
        const double* pt = lds + t * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = lk + 4 * r;
            const double v = pt[tsw(row, li)], vt = pt[tsw(li, row)];
            if (I != J) {
                Pb[(long long)(16 * I + row) * ldn + 16 * J + li] = v;             // (row, li) of the tile
                Pb[(long long)(16 * J + row) * ldn + 16 * I + li] = vt;            // (li, row) of the tile, transposed
            } else
                Pb[(long long)(16 * I + row) * ldn + 16 * I + li] = row >= li ? v : vt;
        }
    }
    // ---- gate on the velocity correction and mean update (src/slam/main.cpp:229-232) -----------------------------------
    if (apply && wave == 7) {
        const double* V = fv.V + 3 * b;
        const bool gate_ok = fabs(V[0] - dxs[3]) + fabs(V[1] - dxs[4]) + fabs(V[2] - dxs[5]) < fv.gate;
        if (lane == 0 && applied) applied[b] = gate_ok;
        if (gate_ok) {
            const double dv = lane < 3 ? dxs[3 + lane] : 0.0;      // read by every lane before any lane writes V
            if (lane < 3) {
                fv.X[3 * b + lane] += dxs[lane];
                fv.V[3 * b + lane] += dv;
            }
            const long long fb = (long long)b * 3 * fv.F;
            for (int t = lane; t < 3 * nf; t += 64) {
                const int i = t / 3, cc = t - 3 * i;
                fv.pib[fb + (long long)cc * fv.F + i] += dxs[6 + t];
            }
        }
    }
}

template <int NT>
__global__ void __launch_bounds__(512, SEQ_WAVES_PER_EU) seq_update_kernel(FilterView fv, double* __restrict__ dx, int apply,
                                                          int* __restrict__ applied)
{
    extern __shared__ double lds[];
    seq_update_filter<NT, false>(fv, blockIdx.x, lds, dx, apply, applied);
}

// P <- F P F^T + Q (calcP, src/slam/main.cpp:526-530 with G = I) on the lower tiles of P in LDS.  Row j of F in compact
// form: fa[j][0..2] on columns 3..5 (the velocity), fg[j][0..2] on columns gs[j] .. + 2; the three rows of a triple share
// their group, which is the triple itself except for the velocity rows, whose group is the bias triple.  A thread owns
// 3 x 3 blocks (row triple tj, column triple tl >= tj) of the result, each of which depends on the 6 x 6 entries
// P[{g_j.., 3..5}][{g_l.., 3..5}]: its own block, and entries of the velocity and bias rows (or, by symmetry, columns).
// Those six rows are copied aside first, so every block is updated in place.
// `aux`: free LDS of at least 6 NL + 18 NTR + NL + NL / 2 doubles (NL = 16 NT, NTR = ceil(NL / 3)).
// Two barriers inside, the second at the end; the image must be complete on entry.
template <int NT>
__device__ __forceinline__ void predict_lower_image(const FilterView& fv, int b, double* __restrict__ lds,
                                                     double* __restrict__ aux)
{
    constexpr int NL = 16 * NT, NTR = (NL + 2) / 3;
    constexpr int TASKS = ((NTR + 1) / 2) * (NTR + 1);               // rows tj and NTR - 1 - tj together: NTR + 1 blocks
    const int tid = threadIdx.x;
    auto plow = [&](int r, int c) -> double& {                       // storage of P[r][c] = P[c][r]
        const int I = r >> 4, J = c >> 4;
        return I >= J ? lds[tile_index(I, J) * 256 + tsw(r & 15, c & 15)] : lds[tile_index(J, I) * 256 + tsw(c & 15, r & 15)];
    };
    double* sv = aux;                                                // old rows 3..5
    double* sb = sv + 3 * NL;                                        // old rows nb .. nb + 2
    double* pfa = sb + 3 * NL;                                       // [3 NTR][3], zero past NL
    double* pfg = pfa + 9 * NTR;
    double* pq = pfg + 9 * NTR;                                      // [NL]
    int* pgs = reinterpret_cast<int*>(pq + NL);                      // [NL]
    const long long cbn = (long long)b * fv.ldn;
    for (int t = tid; t < 9 * NTR; t += 512) {
        const bool ok = t < 3 * NL;
        pfa[t] = ok ? fv.fa[cbn * 3 + t] : 0.0;
        pfg[t] = ok ? fv.fg[cbn * 3 + t] : 0.0;
    }
    for (int t = tid; t < NL; t += 512) { pgs[t] = fv.fgs[cbn + t]; pq[t] = fv.qdiag[cbn + t]; }
    const int nb = fv.fgs[cbn + 3];                                  // first bias row
    for (int t = tid; t < 3 * NL; t += 512) {
        const int r = t / NL, c = t - r * NL;
        sv[t] = plow(3 + r, c);
        sb[t] = plow(min(nb + r, NL - 1), c);
    }
    __syncthreads();
#pragma unroll 1
    for (int k = tid; k < TASKS; k += 512) {
        const int p = k / (NTR + 1), idx = k - p * (NTR + 1);
        int tj = p, tl = p + idx;
        if (idx >= NTR - p) {
            if (2 * p == NTR - 1) continue;                          // (odd NTR: the middle row only once)
            tj = NTR - 1 - p; tl = tj + (idx - (NTR - p));
        }
        const int j0 = 3 * tj, l0 = 3 * tl;
        const int gj = pgs[min(j0, NL - 1)], gl = pgs[min(l0, NL - 1)];
        // column coefficients: fgl[3 c + k] on column gl + k, fal[3 c + k] on column 3 + k
        const double* fgl = pfg + 3 * l0;
        const double* fal = pfa + 3 * l0;
        double acc[3][3];
#pragma unroll
        for (int h = 0; h < 2; ++h) {                                // h = 0: the rows of the group, h = 1: rows 3..5
#pragma unroll
            for (int aa = 0; aa < 3; ++aa) {
                double pg[3], pv[3];                                 // old P[a][gl + k], old P[a][3 + k]
                if (h == 1) {
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) { pg[kk] = sv[aa * NL + min(gl + kk, NL - 1)]; pv[kk] = sv[aa * NL + 3 + kk]; }
                } else {
                    const int a = min(gj + aa, NL - 1);
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) {
                        const int bq = min(gl + kk, NL - 1);
                        pg[kk] = tj == 1 ? sb[aa * NL + bq] : (tl == 1 ? sb[kk * NL + a] : plow(a, bq));
                        pv[kk] = sv[kk * NL + a];                    // P[a][3 + k] = P[3 + k][a]
                    }
                }
                double t[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double sacc = pg[0] * fgl[3 * c];
                    sacc = fma(pg[1], fgl[3 * c + 1], sacc); sacc = fma(pg[2], fgl[3 * c + 2], sacc);
                    sacc = fma(pv[0], fal[3 * c], sacc); sacc = fma(pv[1], fal[3 * c + 1], sacc); sacc = fma(pv[2], fal[3 * c + 2], sacc);
                    t[c] = sacc;
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double f = h == 0 ? pfg[3 * (j0 + r) + aa] : pfa[3 * (j0 + r) + aa];
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[r][c] = (h == 0 && aa == 0) ? f * t[c] : fma(f, t[c], acc[r][c]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int j = j0 + r, l = l0 + c;
                if (j < NL && l < NL && (tj != tl || l >= j)) {
                    const double v = acc[r][c] + (j == l ? pq[j] : 0.0);
                    plow(l, j) = v;                                  // (l >= j within a block of the upper half)
                    if ((j >> 4) == (l >> 4)) lds[tile_index(j >> 4, j >> 4) * 256 + tsw(j & 15, l & 15)] = v;   // diagonal tiles are kept whole
                }
            }
    }
    __syncthreads();
}

// One whole iteration with an image (src/slam/main.cpp:167-233 without the association) of a small filter in ONE
// launch, for callers that pay for launches, not work (a single filter, a handful of them): frame begin, compact rows
// of F at the state before the motion, mu += dynamics(mu) dt, the covariance propagated on its lower tiles in LDS,
// measurement rows at the new state, the block-sequential update above, gate and mean update.  The covariance is
// read once and written once; F, H, e and R pass through memory for the other kernels' sake (introspection).
// Same arithmetic as the separate kernels up to the summation order of the propagation.
template <int NT>
__global__ void __launch_bounds__(512) seq_step_kernel(FilterView fv, const double* __restrict__ qbw,
                                                        const double* __restrict__ w, const double* __restrict__ acc,
                                                        const double* __restrict__ dtv, const double* __restrict__ alt,
                                                        const double* __restrict__ msrc, const double* __restrict__ mrefl,
                                                        int ml, int frame, double* __restrict__ dx, int apply,
                                                        int* __restrict__ applied)
{
    extern __shared__ double lds[];
    constexpr int NL = 16 * NT, NPT = NT * (NT + 1) / 2;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int ldn = fv.ldn;
    const double* Pb = fv.P + (long long)b * ldn * ldn;
    for (int t = wave; t < NPT; t += 8) {                // the lower tiles of P: in flight during the small steps below
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        const int J = t - I * (I + 1) / 2;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            lds[t * 256 + tsw(lk + 4 * r, li)] = Pb[(long long)(16 * I + lk + 4 * r) * ldn + 16 * J + li];
    }
    if (frame) {
        frame_begin_filter(fv, b);
        __syncthreads();                    // the clamped depths are read by the rows of F and by the dynamics
    }
    const double dt = dtv[b];
    for (int j = tid; j < NL; j += 512) motion_row(fv, b, j, qbw, w, dt);
    __syncthreads();                        // F is taken at the state before the dynamics move it
    dynamics_filter(fv, b, qbw, w, acc, dtv);
    __syncthreads();
    measure_filter(fv, b, tid, 512, qbw, alt, msrc, mrefl, ml);
    predict_lower_image<NT>(fv, b, lds, lds + NPT * 256);             // (the panels of the update are not in use yet)
    seq_update_filter<NT, true>(fv, b, lds, dx, apply, applied);
}

}  // namespace slamk

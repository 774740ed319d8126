// Fused covariance update for small filters, one workgroup per filter:
//     A = I - K H,  AP = A P,  P <- AP A^T + K R K^T      (Joseph form)
//  or P <- (AP + AP^T) / 2                                 (updateP, src/slam/main.cpp:561-569)
// P (n x n, n = 16 NT <= 128) lives in LDS as 16 x 16 tiles, A is built in the same LDS space first and then
// held in registers, K R is never formed in memory: HBM traffic is one read and one write of P plus two
// reads of K.
//
//   phase A  A = I - K H into LDS, and dx = K e (src/slam/main.cpp:228) from the same reads of K.  H is sparse: column 6 + 3 i + c of K H only sums the 4 or 6 measurement
//            rows of feature i (one lane per feature, rows of K read as contiguous runs), columns 0..2 sum
//            all rows (lanes along the row of K), the rest of A is I.  Wavefront `blk` then takes its 16
//            rows of A in MFMA operand layout (lane (li, lk) holds A[16 blk + li][16 kt + 4 lk + s]: four neighbours
//            of the tile image per k tile, so that the operand reads of phases 2 are 16 bytes wide).
//   phase P  P tiles replace A in LDS.
//   phase 1  AP(blk, :) = A(blk, :) P with A from registers and P tiles from LDS; the row block is then
//            written over P(blk, :) (all wavefronts have finished reading P by then).
//   phase 2  wavefront `blk` owns column J = blk of the lower triangle: out(I, J), I >= J,
//              = sum_k AP(I, k) A(J, k)   A operand from LDS, B operand = the A fragments
//              + sum_q K(I, q) r_q K(J, q) K staged through LDS in chunks of 16 columns
//            and stores every tile together with its mirror image.
// The altitude row is the LAST measurement row of a filter (row m - 1).  `mlc` = the columns of K that the chunk
// loop covers (a multiple of 16); a filter whose altitude column is column mlc itself (m - 1 == mlc) adds that
// column's term k_a r_a k_a^T on the vector pipeline instead of paying a seventeenth-of-a-chunk MFMA pass for it.
// Row blocks are dealt to wavefronts so that the two wavefronts sharing a SIMD (w and w + 4) own
// columns of 8 - t and t + 1 tiles.
#pragma once
#include <hip/hip_runtime.h>

#include "gain_solver.hip.h"

#ifndef JOSEPH_STAMP
#define JOSEPH_STAMP(i)    // timing builds (scripts/joseph_bench.hip) record the cycle counter of one workgroup here
#endif

namespace slamk {

// v + (v of the lane selected by the DPP control), lanes without a source add zero.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(hi, lo);
}

// Sum of v over the 32 lanes of half-wavefront `half` (0 or 1), returned in every lane of that half: scan within
// the rows of 16 lanes (row_shr 1, 2, 4, 8), lane 15 of the even row added to the odd row (row_bcast:15), then the
// totals of lanes 31 and 63 read back.
__device__ __forceinline__ double half_sum32(double v, int half)
{
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double t0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 31), __builtin_amdgcn_readlane(lo, 31));
    const double t1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 63), __builtin_amdgcn_readlane(lo, 63));
    return half ? t1 : t0;
}

template <int NT>
__global__ void __launch_bounds__(512) joseph_kernel(FilterView fv, const double* __restrict__ K, int ldk,
                                                      long long sK, int mlc, int form, double* __restrict__ dx,
                                                      int apply, int* __restrict__ applied)
{
    extern __shared__ double lds[];                     // NT * NT tiles of 256 doubles (tile (I, J) at I * NT + J)
    constexpr int KBUF = (NT <= 7) ? 2 : 1;             // chunk buffers of K (LDS holds two up to NT = 7)
    double* kst = lds + NT * NT * 256;                  // + KBUF * NT tiles: 16-column chunks of K
    double* rds = kst + KBUF * NT * 256;                // + 256: diagonal of R
    double* dxs = rds + 256;                            // + 128: dx = K e
    double* kalt = dxs + 128;                           // + 128: the altitude column of K
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int blk = wave < 4 ? wave : 11 - wave;        // 0 1 2 3 7 6 5 4
    const bool active = blk < NT;
    const int ldn = fv.ldn;
    double* Pb = fv.P + (long long)b * ldn * ldn;
    const double* Kb = K + (long long)b * sK;
    const int nf = fv.nf[b];
    const int ia = fv.m[b] - 1;                         // altitude row of H / column of K
    const bool border = (ia == mlc);                    // not covered by the chunk loop

    JOSEPH_STAMP(0);
    // The P tiles of this wavefront go to registers during phase A, one tile per pass over the rows of K and behind
    // that pass's request for the next rows of K: vector memory loads complete in order, so whatever a pass waits for
    // must not sit behind more tile loads than one pass can cover (all 28 at the start held up the first barrier, all
    // of them behind the first rows of K held up the second pass).
    constexpr int PT = (NT * NT + 7) / 8;
    static_assert(PT <= NT, "one P tile per pass");
    double pre_p[PT][4];
    auto request_p = [&](int u) {
        const int t = wave + 8 * u;
        const int I = t / NT, J = t - I * NT;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            pre_p[u][r] = (t < NT * NT) ? Pb[(long long)(16 * I + lk + 4 * r) * ldn + 16 * J + li] : 0.0;
    };

    // ---- phase A: A = I - K H in LDS -------------------------------------------------------------------
    // The altitude row (H row ia = (ha, 0), main.cpp:354) is taken by one thread per row j of A, ahead of the
    // feature rows: A[j][0..2] = I - K[j][ia] ha, and its share of dx_j = sum_q K[j][q] e[q] (main.cpp:228).
    double ka_own = 0.0, e_alt = 0.0, hal[3] = {0.0, 0.0, 0.0};
    if (tid < 16 * NT) {
        ka_own = Kb[(long long)tid * ldk + ia];
        e_alt = fv.e[(long long)b * fv.ldm + ia];
#pragma unroll
        for (int c = 0; c < 3; ++c) hal[c] = fv.ha[((long long)b * fv.ldm + ia) * 3 + c];
    }
    for (int t = tid; t < mlc; t += 512) rds[t] = fv.rdiag[(long long)b * fv.ldm + t];
    for (int t = tid; t < NT * NT * 256; t += 512) lds[t] = 0.0;
    if (tid < 16 * NT) {
        kalt[tid] = ka_own;
        dxs[tid] = ka_own * e_alt;
    }
    __syncthreads();
    if (tid < 16 * NT) {
        lds[((tid >> 4) * (NT + 1)) * 256 + tsw(tid & 15, tid & 15)] = 1.0;
        double* row = lds + ((tid >> 4) * NT) * 256;
#pragma unroll
        for (int c = 0; c < 3; ++c) row[tsw(tid & 15, c)] = (tid == c ? 1.0 : 0.0) - ka_own * hal[c];
    }
    __syncthreads();
    JOSEPH_STAMP(1);
    {
        const double* hb = fv.hb + (long long)b * fv.ldm * 3;
        const double* ha = fv.ha + (long long)b * fv.ldm * 3;
        const int half = lane >> 5, l32 = lane & 31;
        // lane = feature i: columns 6 + 3 i .. + 2 from its 4 or 6 measurement rows, and its share of the
        // position columns 0..2 (summed over the lanes)
        auto feature_block = [&](int i0, auto first) {      // 32 features; the first block also requests the P tiles
            const int i = i0 + l32;
            const bool live = i < nf;
            const int row0 = live ? fv.rowoff[(long long)b * fv.F + i] : 0;
            const bool six = live && fv.isref[(long long)b * fv.F + i] != 0;
            double h[18], g[18];
#pragma unroll
            for (int t = 0; t < 18; ++t) {
                const bool on = live && (t < 12 || six);
                h[t] = on ? hb[(long long)row0 * 3 + t] : 0.0;
                g[t] = on ? ha[(long long)row0 * 3 + t] : 0.0;
            }
            const double* ev = fv.e + (long long)b * fv.ldm;
            double ef[6];
#pragma unroll
            for (int t = 0; t < 6; ++t) ef[t] = (live && (t < 4 || six)) ? ev[row0 + t] : 0.0;
            const int l0 = 6 + 3 * i;
            // The gain entries are requested KDEPTH passes ahead (a pass waits for a round trip to memory otherwise);
            // a feature's 4 or 6 entries of a row of K are a 16-byte aligned run (row offsets and ldk are even).
            constexpr int KDEPTH = (NT <= 7) ? 2 : 1;       // (no registers to spare for two at NT = 8)
            double2 kn[KDEPTH][3];
            auto fetch = [&](int pass) {
                const double2* kr = reinterpret_cast<const double2*>(Kb + (long long)(pass * 16 + wave * 2 + half) * ldk + row0);
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    kn[pass % KDEPTH][t] = (live && (t < 2 || six)) ? kr[t] : make_double2(0.0, 0.0);
            };
#pragma unroll
            for (int pass = 0; pass < KDEPTH && pass < NT; ++pass) fetch(pass);
#pragma unroll
            for (int pass = 0; pass < NT; ++pass) {
                const int j = pass * 16 + wave * 2 + half;
                double k[6];
#pragma unroll
                for (int t = 0; t < 3; ++t) { k[2 * t] = kn[pass % KDEPTH][t].x; k[2 * t + 1] = kn[pass % KDEPTH][t].y; }
                if (pass + KDEPTH < NT) fetch(pass + KDEPTH);
                if constexpr (decltype(first)::value) {
                    if (pass < PT) request_p(pass);
                }
                double x[3] = {0.0, 0.0, 0.0};
                double de = 0.0;                         // this lane's share of dx_j = sum_q K[j][q] e[q]   (main.cpp:228)
#pragma unroll
                for (int t = 0; t < 6; ++t) de = fma(k[t], ef[t], de);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double v = k[0] * h[c];
#pragma unroll
                    for (int t = 1; t < 6; ++t) v = fma(k[t], h[3 * t + c], v);
#pragma unroll
                    for (int t = 0; t < 6; ++t) x[c] = fma(k[t], g[3 * t + c], x[c]);
                    const int l = l0 + c;
                    if (live) lds[((j >> 4) * NT + (l >> 4)) * 256 + tsw(j & 15, l & 15)] = (j == l ? 1.0 : 0.0) - v;
                }
                // sums over the 32 lanes of the half-wavefront with data-parallel-primitive moves (no LDS traffic)
                const double sx0 = half_sum32(x[0], half), sx1 = half_sum32(x[1], half), sx2 = half_sum32(x[2], half);
                const double sde = half_sum32(de, half);
                if (l32 == 3) dxs[j] += sde;
                if (l32 < 3) lds[((j >> 4) * NT) * 256 + tsw(j & 15, l32)] -= (l32 == 0 ? sx0 : (l32 == 1 ? sx1 : sx2));
            }
        };
        feature_block(0, std::true_type{});                 // also without features: the P tiles are requested here
        for (int i0 = 32; i0 < nf; i0 += 32) feature_block(i0, std::false_type{});
    }
    __syncthreads();
    JOSEPH_STAMP(2);
    if (dx && tid < 16 * NT) dx[(long long)b * ldn + tid] = dxs[tid];
    double af[4 * NT];
    if (active) {
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) tile_ld4(lds + (blk * NT + kt) * 256, li, 4 * lk, af + 4 * kt);
    }
    __syncthreads();

    // ---- phase P: P tiles into LDS ------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int t = wave + 8 * u;
        if (t < NT * NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[t * 256 + tsw(lk + 4 * r, li)] = pre_p[u][r];
        }
    }
    __syncthreads();

    // Gate on the velocity correction and mean update (src/slam/main.cpp:229-232), as gain_apply_kernel: the work of
    // wavefront 4, which owns no row block up to NT = 7 and would otherwise wait at the next barrier through phase 1
    // (its loads of V must not sit in front of a barrier the other wavefronts wait at).  Nothing else in this kernel
    // reads X, V or the features.
    if (apply && wave == 4) {
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
                const int i = t / 3, c = t - 3 * i;
                fv.pib[fb + (long long)c * fv.F + i] += dxs[6 + t];
            }
        }
    }
    JOSEPH_STAMP(3);
    // ---- phase 1: AP(blk, :) = A(blk, :) P -------------------------------------------------------------
    v4d acc[NT];
    if (active) {
#pragma unroll
        for (int J = 0; J < NT; ++J) acc[J] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4 * NT; ++s) {
            const double* col = lds + (s >> 2) * NT * 256 + tsw(4 * lk + (s & 3), li);
#pragma unroll
            for (int J = 0; J < NT; ++J) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s], col[J * 256], acc[J], 0, 0, 0);
        }
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[(blk * NT + J) * 256 + tsw(lk + 4 * r, li)] = acc[J][r];
    }
    __syncthreads();

    if (form == 1) {       // P = (AP + AP^T) / 2
        for (int t = wave; t < NT * NT; t += 8) {
            const int I = t / NT, J = t - I * NT;
            const double* tij = lds + t * 256;
            const double* tji = lds + (J * NT + I) * 256;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = lk + 4 * r;
                Pb[(long long)(16 * I + row) * ldn + 16 * J + li] = (tji[tsw(li, row)] + tij[tsw(row, li)]) / 2;
            }
        }
        return;
    }

    JOSEPH_STAMP(4);
    // ---- phase 2: column J = blk of the lower triangle ---------------------------------------------------
    const int J = blk;
    v4d o[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) o[t] = (v4d){0.0, 0.0, 0.0, 0.0};
    if (active) {
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const double* rowp = lds + (J * NT + kt) * 256 + li * 16;
            const int c0 = (4 * lk) ^ (li & 14), c1 = (4 * lk + 2) ^ (li & 14);
#pragma unroll
            for (int h = 0; h < 2; ++h) {                 // two k steps per 16-byte read of every tile of the column
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (J + t < NT) {
                        const double2 a2 = *reinterpret_cast<const double2*>(rowp + t * NT * 256 + (h ? c1 : c0));
                        o[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, af[4 * kt + 2 * h], o[t], 0, 0, 0);
                        o[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, af[4 * kt + 2 * h + 1], o[t], 0, 0, 0);
                    }
            }
        }
    }
    JOSEPH_STAMP(5);
    {
        // K R K^T: wavefront w < NT stages tile (w, chunk) of K; the next chunk is in flight during the products
        const double* ksrc = Kb + (long long)(16 * wave + lk) * ldk + li;     // element (lk + 4 r, li) of tile `wave`
        const long long r4 = (long long)4 * ldk;
        const bool loader = wave < NT;
        // the chunks are requested two ahead: one chunk of products does not cover a round trip to memory
        const int nchunk = mlc / 16;
        double pre[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        if (loader) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pre[0][r] = ksrc[r * r4];
            if (nchunk > 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pre[1][r] = ksrc[r * r4 + 16];
            }
        }
        for (int c = 0; c < nchunk; ++c) {
            double* kc = kst + (KBUF == 2 ? (c & 1) * NT * 256 : 0);
            if (loader) {
#pragma unroll
                for (int r = 0; r < 4; ++r) kc[wave * 256 + tsw(lk + 4 * r, li)] = pre[0][r];
#pragma unroll
                for (int r = 0; r < 4; ++r) pre[0][r] = pre[1][r];
            }
            __syncthreads();
            if (loader && c + 2 < nchunk) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pre[1][r] = ksrc[r * r4 + 16 * (c + 2)];
            }
            if (active) {
                // all operand fragments of the chunk are requested before the first product: one LDS latency per
                // chunk instead of one per k step
                double ka[4][NT], rq[4];
                {
                    const double2 r01 = *reinterpret_cast<const double2*>(rds + 16 * c + 4 * lk);
                    const double2 r23 = *reinterpret_cast<const double2*>(rds + 16 * c + 4 * lk + 2);
                    rq[0] = r01.x; rq[1] = r01.y; rq[2] = r23.x; rq[3] = r23.y;
                }
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (J + t < NT) {
                        double w[4];
                        tile_ld4(kc + (J + t) * 256, li, 4 * lk, w);
                        ka[0][t] = w[0]; ka[1][t] = w[1]; ka[2][t] = w[2]; ka[3][t] = w[3];
                    }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const double bq = ka[s][0] * rq[s];
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (J + t < NT) o[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ka[s][t], bq, o[t], 0, 0, 0);
                }
            }
            if (KBUF == 1) __syncthreads();      // with two buffers the next store goes to the other one
        }
        if (border && active) {                  // + k_a r_a k_a^T for the altitude column, which no chunk holds
            const double cj = kalt[16 * J + li] * fv.rdiag[(long long)b * fv.ldm + ia];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (J + t < NT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[t][r] = fma(kalt[16 * (J + t) + lk + 4 * r], cj, o[t][r]);
                }
        }
    }
    JOSEPH_STAMP(6);
    if (!active) return;
    // Every tile goes out twice: as it is (rows of 128 bytes), and mirrored.  The mirror image is read back
    // transposed from a tile of LDS (the AP tiles are dead by now; wavefront w uses tile w) so that its stores are
    // rows of 128 bytes as well instead of 64 single doubles a leading dimension apart.
    double* sc = lds + wave * 256;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (J + t >= NT) break;
        const int I = J + t;
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[tsw(lk + 4 * r, li)] = o[t][r];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * I + lk + 4 * r, col = 16 * J + li;
            if (row >= col) Pb[(long long)row * ldn + col] = o[t][r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = li, bq = lk + 4 * r;             // element (a, bq) of the tile -> out[16 J + bq][16 I + a]
            const double v = sc[tsw(a, bq)];
            if (16 * I + a > 16 * J + bq) Pb[(long long)(16 * J + bq) * ldn + 16 * I + a] = v;
        }
    }
    JOSEPH_STAMP(7);
}

// ------------------------------------------------------------------------------------------------------------------
// The same update for NT <= 7 with ONE pass over K.  The chunks of 16 columns of K that K R K^T walks through LDS also
// feed A = I - K H and dx = K e, as matrix products: 16 measurement rows are the rows of at most five features, so the
// entries of those rows of H outside columns 0..2 lie within 16 consecutive columns cb .. cb + 15 (cb = first column
// of the chunk's first feature).  Per chunk and row block of K, next to the K R K^T tiles:
//     G1 (16 x 16) = K(blk, chunk) * H(chunk, cb .. cb + 15)         subtracted from A(blk, cb ..) in LDS
//     G2 (16 x 16) += K(blk, chunk) * [ha(chunk, 0..2) | e(chunk) | 0]  kept in registers over all chunks: the position
//                                                                    columns of K H and dx = K e (src/slam/main.cpp:228)
// with the B operands picked out of the compact rows of H (a lane holds column li of row q = 16 c + 4 lk + s).  K is
// read once, and what used to be phase A (a third of the kernel without a single matrix instruction) becomes 8 matrix
// instructions per chunk and wavefront.
//   order of phases: chunks (K R K^T into o; A, dx) -> A fragments to registers, P tiles to LDS -> A P ->
//   (A P) A^T added into o -> store.
// LDS: NT^2 tiles (A, then P / AP) + 2 NT tiles (chunks of K) + 10 tiles (diag R, dx, altitude column, compact rows of H).
// (NT <= 2: registers capped at 128 so that two workgroups share a CU; at NT = 3 the spills cost more than that gains)
template <int NT>
__global__ void __launch_bounds__(512, (NT <= 2 ? 4 : 2)) joseph_merged_kernel(FilterView fv, const double* __restrict__ K, int ldk,
                                                             long long sK, int mlc, int form, double* __restrict__ dx,
                                                             int apply, int* __restrict__ applied)
{
    static_assert(NT <= 7, "LDS: the tables of this variant do not fit beside 64 tiles");
    extern __shared__ double lds[];
    double* kst = lds + NT * NT * 256;                  // 2 NT tiles: 16-column chunks of K, double buffered
    double* rds = kst + 2 * NT * 256;                   // [256] diagonal of R
    double* dxs = rds + 256;                            // [128] dx = K e
    double* kalt = dxs + 128;                           // [128] the altitude column of K
    double* cfa = kalt + 128;                           // [256][4] row q of H: entries in columns 0..2, innovation e_q
    double* chb = cfa + 1024;                           // [256][3] row q of H: entries in columns cs .. cs + 2
    int* ccs = reinterpret_cast<int*>(chb + 768);       // [256] cs of row q (-1: none)
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int blk = wave < 4 ? wave : 11 - wave;        // 0 1 2 3 7 6 5 4
    const bool active = blk < NT;
    const int ldn = fv.ldn;
    double* Pb = fv.P + (long long)b * ldn * ldn;
    const double* Kb = K + (long long)b * sK;
    const long long rb = (long long)b * fv.ldm;
    const int nf = fv.nf[b];
    const int ia = fv.m[b] - 1;                         // altitude row of H / column of K
    const bool border = (ia == mlc);                    // not covered by the chunk loop
    const int nchunk = mlc / 16;
    JOSEPH_STAMP(0);

    // ---- small loads first (vector memory loads complete in order): altitude entries, compact rows, first chunks of K
    double ka_own = 0.0, e_alt = 0.0, hal[3] = {0.0, 0.0, 0.0};
    if (tid < 16 * NT) {
        ka_own = Kb[(long long)tid * ldk + ia];
        e_alt = fv.e[rb + ia];
#pragma unroll
        for (int c = 0; c < 3; ++c) hal[c] = fv.ha[(rb + ia) * 3 + c];
    }
    // No branch around a load of K or P from here on: the wait for a chunk counts the loads requested after it, and
    // the compiler can only count them along straight code (a wavefront without a tile reads the last tile and does
    // not store it; past the last chunk the last one is read again).
    const bool loader = wave < NT;
    const double* ksrc = Kb + (long long)(16 * (loader ? wave : NT - 1) + lk) * ldk + li;     // element (lk + 4 r, li) of the tile
    const long long r4 = (long long)4 * ldk;
    double pre[2][4];                                   // chunk k waits in set k & 1
    auto request_chunk = [&](int k, auto set_) {
        constexpr int S = decltype(set_)::value;
        const int kk = max(min(k, nchunk - 1), 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) pre[S][r] = ksrc[r * r4 + 16 * kk];
    };
    request_chunk(0, std::integral_constant<int, 0>{});
    request_chunk(1, std::integral_constant<int, 1>{});
    // compact rows of H for the chunk loop (mlc <= 208 < 512: one row per thread), requested here, stored below.
    // The altitude row is taken ahead of the loop (one thread per row of A, below), rows past it are padding.
    const bool trow = tid < mlc, tlive = tid < ia;
    double t_rd = 0.0, t_e = 0.0, t_ha[3] = {0.0, 0.0, 0.0}, t_hb[3] = {0.0, 0.0, 0.0};
    int t_cs = -1;
    if (trow) t_rd = fv.rdiag[rb + tid];
    if (tlive) {
        t_cs = fv.hcs[rb + tid];
        t_e = fv.e[rb + tid];
#pragma unroll
        for (int c = 0; c < 3; ++c) { t_ha[c] = fv.ha[(rb + tid) * 3 + c]; t_hb[c] = fv.hb[(rb + tid) * 3 + c]; }
    }
    // P tiles of this wavefront: requested one per chunk during the first chunks, each behind that chunk's request for
    // K (loads complete in order, and a burst of all tiles at once holds the wavefront up at the memory pipeline).
    constexpr int PT = (NT * NT + 7) / 8;
    double pre_p[PT][4];
    auto request_p = [&](int u) {
        const int t = min(wave + 8 * u, NT * NT - 1);
        const int I = t / NT, Jt = t - I * NT;
#pragma unroll
        for (int r = 0; r < 4; ++r) pre_p[u][r] = Pb[(long long)(16 * I + lk + 4 * r) * ldn + 16 * Jt + li];
    };
    for (int t = tid; t < NT * NT * 256; t += 512) lds[t] = 0.0;
    if (tid < 16 * NT) {
        kalt[tid] = ka_own;
        dxs[tid] = ka_own * e_alt;
    }
    if (trow) {
        rds[tid] = t_rd;
        ccs[tid] = t_cs;
        *reinterpret_cast<double2*>(cfa + 4 * tid) = make_double2(t_ha[0], t_ha[1]);
        *reinterpret_cast<double2*>(cfa + 4 * tid + 2) = make_double2(t_ha[2], t_e);
#pragma unroll
        for (int c = 0; c < 3; ++c) chb[3 * tid + c] = t_hb[c];
    }
    __syncthreads();
    JOSEPH_STAMP(1);
    // A = I, and the altitude row (H row ia = (ha, 0), main.cpp:354): A[j][0..2] = I - K[j][ia] ha, one thread per row j
    if (tid < 16 * NT) {
        lds[((tid >> 4) * (NT + 1)) * 256 + tsw(tid & 15, tid & 15)] = 1.0;
        double* row = lds + ((tid >> 4) * NT) * 256;
#pragma unroll
        for (int c = 0; c < 3; ++c) row[tsw(tid & 15, c)] = (tid == c ? 1.0 : 0.0) - ka_own * hal[c];
    }
    // (the first chunk's barrier orders these writes before the chunk loop's updates of A)

    // ---- chunks of K: K R K^T, K H and K e -----------------------------------------------------------------------------
    const int J = blk;
    v4d o[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) o[t] = (v4d){0.0, 0.0, 0.0, 0.0};
    v4d g2 = (v4d){0.0, 0.0, 0.0, 0.0};                  // rows lk + 4 r of the block: columns 0..2 of K H, column 3: K e
    auto chunk = [&](int c, auto par_) {
        constexpr int X = decltype(par_)::value;         // parity of c
        double* kc = kst + X * NT * 256;
        if (loader) {
#pragma unroll
            for (int r = 0; r < 4; ++r) kc[wave * 256 + tsw(lk + 4 * r, li)] = pre[X][r];
        }
        __syncthreads();
        request_chunk(c + 2, par_);
    };
    const int ntl = __builtin_amdgcn_readfirstlane(blk < NT ? NT - blk : 0);   // tiles of column J (scalar: uniform branches)
    double* ablk = lds + (NT - ntl) * NT * 256;                                // row block J of A
    auto chunk_work = [&](int c) {
        if (ntl == 0) return;
        const double* kc = kst + (c & 1) * NT * 256;
        // B operands of the K H products: column li of the rows q = 16 c + 4 lk + s of H (the k index of lane group
        // lk in step s)
        const int cb = max(ccs[16 * c], 0);
        double b1[4], b2[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int q = 16 * c + 4 * lk + s;
            const int d = cb + li - ccs[q];                                // (ccs[q] = -1: d >= 3)
            const bool hit = (unsigned)d < 3u;
            const double v1 = chb[3 * q + (hit ? d : 0)];
            const double v2 = cfa[4 * q + (li & 3)];
            b1[s] = hit ? v1 : 0.0;
            b2[s] = li < 4 ? v2 : 0.0;
        }
        double rq[4];
        {
            const double2 r01 = *reinterpret_cast<const double2*>(rds + 16 * c + 4 * lk);
            const double2 r23 = *reinterpret_cast<const double2*>(rds + 16 * c + 4 * lk + 2);
            rq[0] = r01.x; rq[1] = r01.y; rq[2] = r23.x; rq[3] = r23.y;
        }
        v4d g1 = (v4d){0.0, 0.0, 0.0, 0.0};
        const double* kt0 = kc + (NT - ntl) * 256 + li * 16;               // tile J of the chunk
#pragma unroll
        for (int h = 0; h < 2; ++h) {                        // two of the four k steps at a time
            double2 ka[NT];
            const int col = (4 * lk + 2 * h) ^ (li & 14);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (t < ntl) ka[t] = *reinterpret_cast<const double2*>(kt0 + t * 256 + col);
            if (form != 1) {                                 // (first: these do not wait for the B operands of K H)
                const double b0 = ka[0].x * rq[2 * h], b1q = ka[0].y * rq[2 * h + 1];
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (t < ntl) {
                        o[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ka[t].x, b0, o[t], 0, 0, 0);
                        o[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ka[t].y, b1q, o[t], 0, 0, 0);
                    }
            }
            g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ka[0].x, b1[2 * h], g1, 0, 0, 0);
            g2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ka[0].x, b2[2 * h], g2, 0, 0, 0);
            g1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ka[0].y, b1[2 * h + 1], g1, 0, 0, 0);
            g2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ka[0].y, b2[2 * h + 1], g2, 0, 0, 0);
        }
        // A(blk, cb + li) -= G1(:, li): a feature can continue in the next chunk, whose cb then points at it again
        const int colA = cb + li;
        if (li < 15 && colA < 16 * NT) {
            double* ap = ablk + (colA >> 4) * 256;
#pragma unroll
            for (int r = 0; r < 4; ++r) ap[tsw(lk + 4 * r, colA & 15)] -= g1[r];
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if (nchunk >= PT) {                                      // the first chunks also carry the requests for the P tiles
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            if (u & 1) chunk(u, I1{}); else chunk(u, I0{});
            request_p(u);
            chunk_work(u);
        }
        int c = PT;
        if (PT & 1) {
            for (; c + 1 < nchunk; c += 2) { chunk(c, I1{}); chunk_work(c); chunk(c + 1, I0{}); chunk_work(c + 1); }
            if (c < nchunk) { chunk(c, I1{}); chunk_work(c); }
        } else {
            for (; c + 1 < nchunk; c += 2) { chunk(c, I0{}); chunk_work(c); chunk(c + 1, I1{}); chunk_work(c + 1); }
            if (c < nchunk) { chunk(c, I0{}); chunk_work(c); }
        }
    } else {                                                 // few measurement rows: all tiles requested at once
#pragma unroll
        for (int u = 0; u < PT; ++u) request_p(u);
        int c = 0;
        for (; c + 1 < nchunk; c += 2) { chunk(c, I0{}); chunk_work(c); chunk(c + 1, I1{}); chunk_work(c + 1); }
        if (c < nchunk) { chunk(c, I0{}); chunk_work(c); }
    }
    if (ntl > 0) {                                           // position columns and dx of this row block
        if (li < 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ablk[tsw(lk + 4 * r, li)] -= g2[r];
        } else if (li == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dxs[16 * (NT - ntl) + lk + 4 * r] += g2[r];
        }
    }
    if (border && active && form != 1) {                 // + k_a r_a k_a^T for the altitude column, which no chunk holds
        const double cj = kalt[16 * J + li] * fv.rdiag[rb + ia];
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (J + t < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[t][r] = fma(kalt[16 * (J + t) + lk + 4 * r], cj, o[t][r]);
            }
    }
    __syncthreads();
    JOSEPH_STAMP(2);
    if (dx && tid < 16 * NT) dx[(long long)b * ldn + tid] = dxs[tid];
    double af[4 * NT];
    if (active) {
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) tile_ld4(lds + (blk * NT + kt) * 256, li, 4 * lk, af + 4 * kt);
    }
    __syncthreads();

    // ---- P tiles into LDS ------------------------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int t = wave + 8 * u;
        if (t < NT * NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[t * 256 + tsw(lk + 4 * r, li)] = pre_p[u][r];
        }
    }
    __syncthreads();
    // gate on the velocity correction and mean update (src/slam/main.cpp:229-232): wavefront 4 owns no row block
    if (apply && wave == 4) {
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
                const int i = t / 3, c = t - 3 * i;
                fv.pib[fb + (long long)c * fv.F + i] += dxs[6 + t];
            }
        }
    }
    JOSEPH_STAMP(3);
    // ---- AP(blk, :) = A(blk, :) P ------------------------------------------------------------------------------------
    v4d acc[NT];
    if (active) {
#pragma unroll
        for (int Jc = 0; Jc < NT; ++Jc) acc[Jc] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4 * NT; ++s) {
            const double* col = lds + (s >> 2) * NT * 256 + tsw(4 * lk + (s & 3), li);
#pragma unroll
            for (int Jc = 0; Jc < NT; ++Jc)
                acc[Jc] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s], col[Jc * 256], acc[Jc], 0, 0, 0);
        }
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int Jc = 0; Jc < NT; ++Jc)
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[(blk * NT + Jc) * 256 + tsw(lk + 4 * r, li)] = acc[Jc][r];
    }
    __syncthreads();
    if (form == 1) {       // P = (AP + AP^T) / 2
        for (int t = wave; t < NT * NT; t += 8) {
            const int I = t / NT, Jc = t - I * NT;
            const double* tij = lds + t * 256;
            const double* tji = lds + (Jc * NT + I) * 256;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = lk + 4 * r;
                Pb[(long long)(16 * I + row) * ldn + 16 * Jc + li] = (tji[tsw(li, row)] + tij[tsw(row, li)]) / 2;
            }
        }
        return;
    }
    JOSEPH_STAMP(4);
    // ---- column J = blk of the lower triangle: o(I, J) += sum_k AP(I, k) A(J, k)^T ------------------------------------
    if (active) {
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const double* rowp = lds + (J * NT + kt) * 256 + li * 16;
            const int c0 = (4 * lk) ^ (li & 14), c1 = (4 * lk + 2) ^ (li & 14);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (J + t < NT) {
                        const double2 a2 = *reinterpret_cast<const double2*>(rowp + t * NT * 256 + (h ? c1 : c0));
                        o[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, af[4 * kt + 2 * h], o[t], 0, 0, 0);
                        o[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, af[4 * kt + 2 * h + 1], o[t], 0, 0, 0);
                    }
            }
        }
    }
    JOSEPH_STAMP(6);
    if (!active) return;
    double* sc = kst + wave * 256;                       // the chunk buffers are free: staging tile of this wavefront
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (J + t >= NT) break;
        const int I = J + t;
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[tsw(lk + 4 * r, li)] = o[t][r];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * I + lk + 4 * r, col = 16 * J + li;
            if (row >= col) Pb[(long long)row * ldn + col] = o[t][r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = li, bq = lk + 4 * r;             // element (a, bq) of the tile -> out[16 J + bq][16 I + a]
            const double v = sc[tsw(a, bq)];
            if (16 * I + a > 16 * J + bq) Pb[(long long)(16 * J + bq) * ldn + 16 * I + a] = v;
        }
    }
    JOSEPH_STAMP(7);
}

}  // namespace slamk

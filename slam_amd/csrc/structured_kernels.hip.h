// Products with the motion Jacobian F and the measurement Jacobian H in their native sparse form.
//
// Every row of F and of H has at most six non-zeros, in two groups of three adjacent columns:
//   F row j : columns 3..5 (the velocity block) and columns g..g+2 (its own 3-state block; the
//             bias block for the velocity rows)             jacobianMotionModel, src/slam/main.cpp:241-341
//   H row q : columns 0..2 (position) and columns cs..cs+2 (the observed feature)
//                                                           jacobianH, src/slam/main.cpp:575-624
// so P F^T, F T, P H^T and H W are 12 flop per output element instead of 2 n, and I - K H touches
// only the 4 or 6 measurement rows of one feature per column.  All kernels here are HBM-bound: lanes
// run along the contiguous (column) index of the output, the operand rows they read are contiguous
// in the same index, and the per-row coefficients are wave-uniform.
//
// The products accumulate in ascending column index with fused multiply-adds, the order a dense
// row-by-column product visits the same non-zeros in.
#pragma once
#include <hip/hip_runtime.h>

#include "feature_kernels.hip.h"

namespace slamk {

// Compact rows of F = I + dt A and the diagonal of Q (initQ, src/slam/main.cpp:430-440), at the
// PRE-predict state.  The three rows of one 3-state block share the group start fgs.
__device__ __forceinline__ void motion_row(const FilterView& fv, int b, int j, const double* __restrict__ qbw,
                                           const double* __restrict__ w, double dt)
{
    const int nf = fv.nf[b], n = 9 + 3 * nf;
    double a[3] = {0, 0, 0}, g[3] = {0, 0, 0}, q = 0.0;
    int gs = 0;
    if (j < n) {
        const int r = j % 3;
        if (j < 3) {                                   // X rows: I, dt R(qbw) on V            :253-261
            double R[9];
            quat_rot(qbw + 4 * b, R);
            for (int c = 0; c < 3; ++c) { a[c] = dt * R[3 * j + c] + 0.0; g[c] = (c == r) ? dt * 0.0 + 1.0 : 0.0; }
            gs = 0;
        } else if (j < 6) {                            // V rows: I, -dt on the bias block      :338-340
            for (int c = 0; c < 3; ++c) { a[c] = (c == r) ? 1.0 : 0.0; g[c] = (c == r) ? -1 * dt : 0.0; }
            gs = 6 + 3 * nf;
        } else if (j < 6 + 3 * nf) {                   // feature rows: Fib_ith on V, I + dt FiTemp on itself
            const int i = (j - 6) / 3;
            const long long fb = (long long)b * 3 * fv.F;
            const double p0 = fv.pib[fb + i], p1 = fv.pib[fb + fv.F + i], p2 = fv.pib[fb + 2 * fv.F + i];
            const double* V = fv.V + 3 * b;
            const double* ww = w + 3 * b;
            double av[3], ag[3];
            if (r == 0) {
                av[0] = p0 * p2; av[1] = -p2; av[2] = 0.0;                                    // :314-317
                ag[0] = p2 * V[0] - 2 * p0 * ww[2] + p1 * ww[1]; ag[1] = ww[0] + p0 * ww[1]; ag[2] = p0 * V[0] - V[1];   // :290-299
            } else if (r == 1) {
                av[0] = p1 * p2; av[1] = 0.0; av[2] = -p2;
                ag[0] = -ww[0] - p1 * ww[2]; ag[1] = p2 * V[0] - p0 * ww[2] + 2 * p1 * ww[1]; ag[2] = p1 * V[0] - V[2];
            } else {
                av[0] = p2 * p2; av[1] = 0.0; av[2] = 0.0;
                ag[0] = -p2 * ww[2]; ag[1] = p2 * ww[1]; ag[2] = 2 * p2 * V[0] - p0 * ww[2] + p1 * ww[1];
            }
            for (int c = 0; c < 3; ++c) { a[c] = dt * av[c] + 0.0; g[c] = dt * ag[c] + (c == r ? 1.0 : 0.0); }   // :335
            gs = 6 + 3 * i;
        } else {                                       // bias rows: I
            for (int c = 0; c < 3; ++c) g[c] = (c == r) ? 1.0 : 0.0;
            gs = 6 + 3 * nf;
        }
        const double q0 = fv.Q0 * (dt * dt);
        q = q0;
        if (j < 3) q = q0 * (0.1 * q0);
        if (j >= 6 + 3 * nf) q = fv.qbias * dt * dt;
    }
    const long long rb = (long long)b * fv.ldn + j;
    for (int c = 0; c < 3; ++c) { fv.fa[rb * 3 + c] = a[c]; fv.fg[rb * 3 + c] = g[c]; }
    fv.fgs[rb] = gs;
    fv.qdiag[rb] = q;
}

__global__ void __launch_bounds__(256) motion_rows_kernel(FilterView fv, const double* __restrict__ qbw,
                                                           const double* __restrict__ w,
                                                           const double* __restrict__ dtv, int nl)
{
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < nl) motion_row(fv, b, j, qbw, w, dtv[b]);
}

// Everything of an iteration that precedes the covariance propagation, one workgroup per filter: the start of an image
// frame (old_pos = X, depth clamp; src/slam/main.cpp:168, states.cpp:91-103) when `frame` is set, the compact rows of
// F at the pre-predict state, then mu += dynamics(mu) dt.  One launch instead of three: what a single small filter
// pays for is launches, not work.
__global__ void __launch_bounds__(256) pre_predict_kernel(FilterView fv, const double* __restrict__ qbw,
                                                           const double* __restrict__ w,
                                                           const double* __restrict__ acc,
                                                           const double* __restrict__ dtv, int nl, int frame)
{
    const int b = blockIdx.x;
    if (frame) {
        frame_begin_filter(fv, b);
        __syncthreads();                    // the clamped depths are read by the rows of F and by the dynamics
    }
    const double dt = dtv[b];
    for (int j = threadIdx.x; j < nl; j += 256) motion_row(fv, b, j, qbw, w, dt);
    __syncthreads();                        // F is taken at the state before the dynamics move it
    dynamics_filter(fv, b, qbw, w, acc, dtv);
}

struct RowSparse {            // row q: ca[q][0..2] on columns c0..c0+2, cb[q][0..2] on columns cs[q]..cs[q]+2
    const double* ca;
    const double* cb;
    const int* cs;            // < 0: no second group
    long long stride;         // rows per filter in ca/cb/cs
    int c0;
};

// Out[j][q] = sum_k In[j][k] M[q][k]   (In * M^T), M row-sparse with `cols` rows; lanes along q.
// grid (ceil(cols / 64), ceil(rows / (4 RPW)), B), 256 threads; each wavefront walks RPW rows.
template <int RPW>
__global__ void __launch_bounds__(256) rowsparse_right_kernel(RowSparse M, const double* __restrict__ In, int ldi,
                                                               long long sIn, double* __restrict__ Out, int ldo,
                                                               long long sOut, int rows, int cols)
{
    const int b = blockIdx.z;
    const int q = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * RPW;
    if (q >= cols) return;
    const long long cr = (long long)b * M.stride + q;
    const double a0 = M.ca[cr * 3], a1 = M.ca[cr * 3 + 1], a2 = M.ca[cr * 3 + 2];
    const double b0 = M.cb[cr * 3], b1 = M.cb[cr * 3 + 1], b2 = M.cb[cr * 3 + 2];
    int cs = M.cs[cr];
    const bool none = cs < 0;
    if (none) cs = 0;                                   // coefficients of a missing group are zero
    const bool first = cs < M.c0;                       // which group has the smaller column indices
    const int k1 = first ? cs : M.c0, k2 = first ? M.c0 : cs;
    const double u0 = first ? b0 : a0, u1 = first ? b1 : a1, u2 = first ? b2 : a2;
    const double v0 = first ? a0 : b0, v1 = first ? a1 : b1, v2 = first ? a2 : b2;
    const double* Ib = In + (long long)b * sIn;
    double* Ob = Out + (long long)b * sOut;
    double x[RPW][6];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {                     // all loads of the wavefront's rows in flight together
        const int j = min(j0 + r, rows - 1);
        const double* row = Ib + (long long)j * ldi;
        x[r][0] = row[k1]; x[r][1] = row[k1 + 1]; x[r][2] = row[k1 + 2];
        x[r][3] = row[k2]; x[r][4] = row[k2 + 1]; x[r][5] = row[k2 + 2];
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int j = j0 + r;
        if (j >= rows) break;
        double acc = x[r][0] * u0;
        acc = fma(x[r][1], u1, acc);
        acc = fma(x[r][2], u2, acc);
        acc = fma(x[r][3], v0, acc);
        acc = fma(x[r][4], v1, acc);
        acc = fma(x[r][5], v2, acc);
        Ob[(long long)j * ldo + q] = acc;
    }
}

// Out[q][l] = sum_k M[q][k] In[k][l] (+ diag[q] on q + dcol == l), M row-sparse with `rows` rows; lanes along l.
// LOWER: row q is skipped by wavefronts whose 64 columns all lie right of column (q | 127).
// grid (ceil(cols / 64), ceil(rows / (4 rpw)), B), 256 threads.
template <bool LOWER>
__global__ void __launch_bounds__(256) rowsparse_left_kernel(RowSparse M, const double* __restrict__ In, int ldi,
                                                              long long sIn, double* __restrict__ Out, int ldo,
                                                              long long sOut, const double* __restrict__ diag,
                                                              long long sDiag, int rows, int cols, int rpw, int dcol)
{
    // dcol: column of row 0's diagonal element (non-zero when M, diag and Out start at a later row of the operator)
    const int b = blockIdx.z;
    const int l0 = blockIdx.x * 64;
    const int l = l0 + (threadIdx.x & 63);
    const int q0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * rpw;
    if (l >= cols) return;
    const double* Ib = In + (long long)b * sIn;
    double* Ob = Out + (long long)b * sOut;
    const double* ca = M.ca + (long long)b * M.stride * 3;
    const double* cb = M.cb + (long long)b * M.stride * 3;
    const int* csp = M.cs + (long long)b * M.stride;
    const double* dg = diag + (long long)b * sDiag;
    const double x0 = Ib[(long long)M.c0 * ldi + l], x1 = Ib[(long long)(M.c0 + 1) * ldi + l],
                 x2 = Ib[(long long)(M.c0 + 2) * ldi + l];
    double y0 = 0, y1 = 0, y2 = 0;
    int cur = -1;
    for (int r = 0; r < rpw; ++r) {
        const int q = q0 + r;
        if (q >= rows) break;
        if (LOWER && l0 > (q | 127)) continue;
        int cs = __builtin_amdgcn_readfirstlane(csp[q]);
        if (cs < 0) cs = 0;
        if (cs != cur) {                               // consecutive rows of one block share the operand rows
            y0 = Ib[(long long)cs * ldi + l];
            y1 = Ib[(long long)(cs + 1) * ldi + l];
            y2 = Ib[(long long)(cs + 2) * ldi + l];
            cur = cs;
        }
        const double a0 = ca[3 * q], a1 = ca[3 * q + 1], a2 = ca[3 * q + 2];
        const double b0 = cb[3 * q], b1 = cb[3 * q + 1], b2 = cb[3 * q + 2];
        double acc;
        if (cs < M.c0) {
            acc = b0 * y0; acc = fma(b1, y1, acc); acc = fma(b2, y2, acc);
            acc = fma(a0, x0, acc); acc = fma(a1, x1, acc); acc = fma(a2, x2, acc);
        } else {
            acc = a0 * x0; acc = fma(a1, x1, acc); acc = fma(a2, x2, acc);
            acc = fma(b0, y0, acc); acc = fma(b1, y1, acc); acc = fma(b2, y2, acc);
        }
        if (q + dcol == l) acc += dg[q];
        Ob[(long long)q * ldo + l] = acc;
    }
}

// Pout = F Pin F^T + Q (calcP, src/slam/main.cpp:526-530 with G = I) in one pass.  A thread owns the 3 x 3
// block (row triple tj, column triple tl) of the output; the three rows (columns) of a triple share
// their group start, so the block depends on the 6 x 6 entries Pin[{g_j.., 3..5}][{g_l.., 3..5}] only.
// The inner sums run in ascending index like the two dense products T = P F^T, P = F T would.
// Pin and Pout must be different buffers.  grid (ceil(nt * nt / 256), B), nt = ceil(nl / 3); 256 threads.
__global__ void __launch_bounds__(256) predict_cov_kernel(FilterView fv, const double* __restrict__ Pin,
                                                           double* __restrict__ Pout, int nl)
{
    const int b = blockIdx.y;
    const int nt = (nl + 2) / 3;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int tj = t / nt, tl = t - tj * nt;
    const int j0 = 3 * tj, l0 = 3 * tl;
    if (tj >= nt) return;
    const long long cb = (long long)b * fv.ldn;
    const int ldn = fv.ldn;
    const double* Pb = Pin + cb * ldn;
    double* Ob = Pout + cb * ldn;
    // coefficients; rows past nl (only in the last triple) read as zero
    double ra[3][3], rg[3][3], ca[3][3], cg[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bool okr = j0 + r < nl, okc = l0 + r < nl;
            ra[r][c] = okr ? fv.fa[(cb + j0 + r) * 3 + c] : 0.0;
            rg[r][c] = okr ? fv.fg[(cb + j0 + r) * 3 + c] : 0.0;
            ca[r][c] = okc ? fv.fa[(cb + l0 + r) * 3 + c] : 0.0;
            cg[r][c] = okc ? fv.fg[(cb + l0 + r) * 3 + c] : 0.0;
        }
    const int gj = fv.fgs[cb + j0], gl = fv.fgs[cb + l0];
    // index sets in ascending order: the own group first when it lies left of the velocity block
    const bool jf = gj < 3, lf = gl < 3;
    const int ja = jf ? gj : 3, jb = jf ? 3 : gj;
    const int la = lf ? gl : 3, lb = lf ? 3 : gl;
    double acc[3][3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int aa = 0; aa < 3; ++aa) {
            const int a = (h ? jb : ja) + aa;
            const double* row = Pb + (long long)a * ldn;
            const double p0 = row[la], p1 = row[la + 1], p2 = row[la + 2];
            const double p3 = row[lb], p4 = row[lb + 1], p5 = row[lb + 2];
            double t[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {       // T[a][l0 + c]
                const double u0 = lf ? cg[c][0] : ca[c][0], u1 = lf ? cg[c][1] : ca[c][1], u2 = lf ? cg[c][2] : ca[c][2];
                const double v0 = lf ? ca[c][0] : cg[c][0], v1 = lf ? ca[c][1] : cg[c][1], v2 = lf ? ca[c][2] : cg[c][2];
                double s = p0 * u0;
                s = fma(p1, u1, s); s = fma(p2, u2, s);
                s = fma(p3, v0, s); s = fma(p4, v1, s); s = fma(p5, v2, s);
                t[c] = s;
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                // coefficient of row j0 + r on column a: the own group is first iff jf
                const double f = (h == 0) ? (jf ? rg[r][aa] : ra[r][aa]) : (jf ? ra[r][aa] : rg[r][aa]);
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
            if (j < nl && l < nl) Ob[(long long)j * ldn + l] = acc[r][c] + (j == l ? fv.qdiag[cb + j] : 0.0);
        }
}

// The same product for filters of moderate size, one workgroup per band of `hband` row triples (hband * nt <= blockDim.x):
// the operand rows of the band (the velocity rows 3..5 and each triple's own group) and the compact rows of F go
// through LDS with coalesced 16-byte loads, the finished band leaves through LDS the same way, so every global
// access is a contiguous run; the arithmetic (and its order) is that of predict_cov_kernel.
// grid (ceil(nt / hband) * 8 * ceil(B / 8)), up to 512 threads; dynamic LDS predict_band_lds_bytes(nl, hband).
__host__ __device__ inline size_t predict_band_lds_bytes(int nl, int hband)
{
    const int nt = (nl + 2) / 3;
    return ((size_t)(hband + 1) * 3 * nl + 2 * 9 * (size_t)nt) * sizeof(double) + (size_t)nt * sizeof(int);
}

__global__ void __launch_bounds__(512) predict_cov_band_kernel(FilterView fv, const double* __restrict__ Pin,
                                                                double* __restrict__ Pout, int nl, int hband)
{
    extern __shared__ double plds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int nt = (nl + 2) / 3;
    // Workgroups are dealt round-robin to the 8 XCDs, each with an L2 of its own: the bands of one filter (which all
    // read its velocity rows and its compact rows of F) are given workgroup ids 8 apart, so that they run on one XCD
    // right after one another and those re-reads stay in its L2.  (A matter of speed only: nothing depends on where a
    // workgroup runs.)
    const int nb = (nt + hband - 1) / hband;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int b = (seq / nb) * 8 + xcd, band = seq - (seq / nb) * nb;
    if (b >= fv.B) return;
    const int t0 = band * hband;
    const int hb = min(hband, nt - t0);
    const long long cb = (long long)b * fv.ldn;
    const int ldn = fv.ldn;
    const double* Pb = Pin + cb * ldn;
    double* Ob = Pout + cb * ldn;
    double* img = plds;                                   // [(hband + 1) * 3][nl]
    double* cfa = img + (size_t)(hband + 1) * 3 * nl;     // [3 nt][3], zero past nl
    double* cfg = cfa + 9 * nt;
    int* cgs = reinterpret_cast<int*>(cfg + 9 * nt);      // [nt]

    // stage: operand rows (16-byte pieces), compact rows of F, group starts
    // (the group of a triple is the triple itself, except for the velocity rows, whose group is the bias triple --
    // motion_row: worked out here from the feature count, so that the loads of P do not wait for a load of fgs)
    const int half = nl >> 1, pieces = (hb + 1) * 3 * half;
    // (only the band that holds the velocity triple needs the feature count; rows past the last state are zero like
    // their coefficients, so a padding triple may read its own rows)
    const int bias0 = band == 0 ? 6 + 3 * fv.nf[b] : 0;
    for (int idx = tid; idx < pieces; idx += nthr) {
        const int row = idx / half, c2 = idx - row * half;
        const int tr = t0 + (row - 3) / 3;
        const int g0 = tr == 1 ? bias0 : 3 * tr;
        const int src = row < 3 ? 3 + row : min(g0 + (row - 3) % 3, nl - 1);
        reinterpret_cast<double2*>(img + (size_t)row * nl)[c2] =
            reinterpret_cast<const double2*>(Pb + (long long)src * ldn)[c2];
    }
    for (int idx = tid; idx < 9 * nt; idx += nthr) {
        const bool ok = idx < 3 * nl;
        cfa[idx] = ok ? fv.fa[cb * 3 + idx] : 0.0;
        cfg[idx] = ok ? fv.fg[cb * 3 + idx] : 0.0;
    }
    for (int idx = tid; idx < nt; idx += nthr) cgs[idx] = fv.fgs[cb + min(3 * idx, nl - 1)];
    __syncthreads();

    const int i = tid / nt, tl = tid - i * nt;
    const bool live = i < hb;
    const int tj = t0 + i;
    const int j0 = 3 * tj, l0 = 3 * tl;
    double acc[3][3];
    if (live) {
        const int gj = cgs[tj], gl = cgs[tl];
        const bool jf = gj < 3, lf = gl < 3;
        const int la = lf ? gl : 3, lb = lf ? 3 : gl;
        const double* ra = cfa + 9 * tj;
        const double* rg = cfg + 9 * tj;
        const double* ca = cfa + 9 * tl;
        const double* cg = cfg + 9 * tl;
        double u[3][3], v[3][3];                           // column coefficients in ascending column order
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                u[c][k] = lf ? cg[3 * c + k] : ca[3 * c + k];
                v[c][k] = lf ? ca[3 * c + k] : cg[3 * c + k];
            }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool own = (h == 0) == jf;               // this half of the sum runs over the triple's own group
            const double* rows = own ? img + (size_t)(3 + 3 * i) * nl : img;
            const double* fr = own ? rg : ra;
#pragma unroll
            for (int aa = 0; aa < 3; ++aa) {
                const double* row = rows + (size_t)aa * nl;
                const double p0 = row[la], p1 = row[la + 1], p2 = row[la + 2];
                const double p3 = row[lb], p4 = row[lb + 1], p5 = row[lb + 2];
                double t[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double sacc = p0 * u[c][0];
                    sacc = fma(p1, u[c][1], sacc); sacc = fma(p2, u[c][2], sacc);
                    sacc = fma(p3, v[c][0], sacc); sacc = fma(p4, v[c][1], sacc); sacc = fma(p5, v[c][2], sacc);
                    t[c] = sacc;
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double f = fr[3 * r + aa];
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[r][c] = (h == 0 && aa == 0) ? f * t[c] : fma(f, t[c], acc[r][c]);
                }
            }
        }
        if (tj == tl)
#pragma unroll
            for (int r = 0; r < 3; ++r)
                if (j0 + r < nl) acc[r][r] += fv.qdiag[cb + j0 + r];
    }
    __syncthreads();
    // the band leaves through the image: rows 3 i + r, then contiguous 16-byte stores
    if (live)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (l0 + c < nl) img[(size_t)(3 * i + r) * nl + l0 + c] = acc[r][c];
    __syncthreads();
    const int orows = min(3 * hb, nl - 3 * t0);
    for (int idx = tid; idx < orows * half; idx += nthr) {
        const int row = idx / half, c2 = idx - row * half;
        reinterpret_cast<double2*>(Ob + (long long)(3 * t0 + row) * ldn)[c2] =
            reinterpret_cast<const double2*>(img + (size_t)row * nl)[c2];
    }
}

// A = I - K H (updateP, src/slam/main.cpp:563-565), columns 3 .. nl-1.  Column 6 + 3 i + c only meets the
// measurement rows of feature i; the velocity, bias and padding columns meet none.  Lanes along the column.
// grid (ceil(nl / 64), ceil(nl / (4 rpw)), B), 256 threads.
// K holds `rows` rows of the gain, the first of them row `row0` of the filter (for the unit diagonal); A has
// leading dimension lda and batch stride sA.
__global__ void __launch_bounds__(256) eye_minus_kh_kernel(FilterView fv, const double* __restrict__ K, int ldk,
                                                            long long sK, double* __restrict__ A, int lda, long long sA,
                                                            int nl, int rows, int row0, int rpw)
{
    const int b = blockIdx.z;
    const int l = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * rpw;
    if (l >= nl) return;
    const int nf = fv.nf[b];
    int frow = 0, cnt = 0, c = 0;
    if (l >= 6 && l < 6 + 3 * nf) {
        const int i = (l - 6) / 3;
        c = (l - 6) - 3 * i;
        frow = fv.rowoff[(long long)b * fv.F + i];
        cnt = fv.isref[(long long)b * fv.F + i] ? 6 : 4;
    }
    const double* hb = fv.hb + ((long long)b * fv.ldm + frow) * 3 + c;
    double h[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) h[t] = (t < cnt) ? hb[3 * t] : 0.0;
    const double* Kb = K + (long long)b * sK + frow;
    double* Ab = A + (long long)b * sA;
    for (int r = 0; r < rpw; ++r) {
        const int j = j0 + r;
        if (j >= rows) break;
        if (l < 3) continue;                           // written by eye_minus_kh_xcols_kernel
        const double* kr = Kb + (long long)j * ldk;
        double acc = 0.0;
        if (cnt) {
            acc = kr[0] * h[0];
#pragma unroll
            for (int t = 1; t < 6; ++t)
                if (t < cnt) acc = fma(kr[t], h[t], acc);
        }
        Ab[(long long)j * lda + l] = (j + row0 == l ? 1.0 : 0.0) - acc;
    }
}

// Columns 0..2 of A = I - K H: every measurement row has position entries.  One wavefront per row of K.
// grid (ceil(nl / 4), B), 256 threads.
__global__ void __launch_bounds__(256) eye_minus_kh_xcols_kernel(FilterView fv, const double* __restrict__ K, int ldk,
                                                                  long long sK, double* __restrict__ A, int lda,
                                                                  long long sA, int rows, int row0, int ml)
{
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= rows) return;
    const double* kr = K + (long long)b * sK + (long long)j * ldk;
    const double* ha = fv.ha + (long long)b * fv.ldm * 3;
    double s0 = 0, s1 = 0, s2 = 0;
    for (int q = lane; q < ml; q += 64) {
        const double k = kr[q];
        s0 = fma(k, ha[3 * q], s0);
        s1 = fma(k, ha[3 * q + 1], s1);
        s2 = fma(k, ha[3 * q + 2], s2);
    }
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o);
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (lane < 3) {
        const double s = lane == 0 ? s0 : (lane == 1 ? s1 : s2);
        A[(long long)b * sA + (long long)j * lda + lane] = (j + row0 == lane ? 1.0 : 0.0) - s;
    }
}

// A = I - K H (updateP, src/slam/main.cpp:563-565) and dx = K e (:228) in ONE pass over K, for filters whose rows of K
// fit LDS (the three kernels above and gain_kx_kernel read K once each).  A wavefront takes KH_ROWS rows of the gain
// into LDS while it sums their position columns and their products with the innovation, then forms those rows of A with
// its lanes along the columns.  Same sums in the same order as eye_minus_kh_kernel, eye_minus_kh_xcols_kernel and
// gain_kx_kernel.  grid (ceil(nl / (4 KH_ROWS)), B), 256 threads, 4 KH_ROWS ml doubles of dynamic LDS; no barrier: a
// wavefront only reads what it wrote itself.
constexpr int KH_ROWS = 4;
__global__ void __launch_bounds__(256) eye_minus_kh_rows_kernel(FilterView fv, const double* __restrict__ K, int ldk,
                                                                 long long sK, double* __restrict__ A, int lda,
                                                                 long long sA, int nl, int ml, double* __restrict__ dx)
{
    extern __shared__ double kh_rows[];                  // [4][KH_ROWS][ml]
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j0 = (blockIdx.x * 4 + wave) * KH_ROWS;
    if (j0 >= nl) return;
    const int nf = fv.nf[b];
    double* kw = kh_rows + (size_t)wave * KH_ROWS * ml;
    const double* Kb = K + (long long)b * sK;
    const double* ha = fv.ha + (long long)b * fv.ldm * 3;
    const double* e = fv.e + (long long)b * fv.ldm;
    double s0[KH_ROWS], s1[KH_ROWS], s2[KH_ROWS], sd[KH_ROWS];
#pragma unroll
    for (int r = 0; r < KH_ROWS; ++r) s0[r] = s1[r] = s2[r] = sd[r] = 0.0;
    for (int q = lane; q < ml; q += 64) {
        const double a0 = ha[3 * q], a1 = ha[3 * q + 1], a2 = ha[3 * q + 2], eq = e[q];
#pragma unroll
        for (int r = 0; r < KH_ROWS; ++r) {
            const double k = j0 + r < nl ? Kb[(long long)(j0 + r) * ldk + q] : 0.0;
            kw[r * ml + q] = k;
            s0[r] = fma(k, a0, s0[r]);
            s1[r] = fma(k, a1, s1[r]);
            s2[r] = fma(k, a2, s2[r]);
            sd[r] += k * eq;
        }
    }
#pragma unroll
    for (int r = 0; r < KH_ROWS; ++r)
        for (int o = 32; o > 0; o >>= 1) {
            s0[r] += __shfl_xor(s0[r], o);
            s1[r] += __shfl_xor(s1[r], o);
            s2[r] += __shfl_xor(s2[r], o);
            sd[r] += __shfl_xor(sd[r], o);
        }
    __builtin_amdgcn_wave_barrier();                     // (the rows in LDS are read by other lanes of this wavefront)
    if (dx && lane == 0) {                               // bias rows are skipped, as in gain_kx_kernel
#pragma unroll
        for (int r = 0; r < KH_ROWS; ++r)
            if (j0 + r < 6 + 3 * nf) dx[(long long)b * fv.ldn + j0 + r] = sd[r];
    }
    double* Ab = A + (long long)b * sA;
    for (int l = lane; l < nl; l += 64) {
        int frow = 0, cnt = 0, c = 0;
        if (l >= 6 && l < 6 + 3 * nf) {
            const int i = (l - 6) / 3;
            c = (l - 6) - 3 * i;
            frow = fv.rowoff[(long long)b * fv.F + i];
            cnt = fv.isref[(long long)b * fv.F + i] ? 6 : 4;
        }
        const double* hb = fv.hb + ((long long)b * fv.ldm + frow) * 3 + c;
        double h[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) h[t] = (t < cnt) ? hb[3 * t] : 0.0;
#pragma unroll
        for (int r = 0; r < KH_ROWS; ++r) {
            const int j = j0 + r;
            if (j >= nl) break;
            double acc = 0.0;
            if (l < 3) acc = l == 0 ? s0[r] : (l == 1 ? s1[r] : s2[r]);
            else if (cnt) {
                const double* kr = kw + r * ml + frow;
                acc = kr[0] * h[0];
#pragma unroll
                for (int t = 1; t < 6; ++t)
                    if (t < cnt) acc = fma(kr[t], h[t], acc);
            }
            Ab[(long long)j * lda + l] = (j == l ? 1.0 : 0.0) - acc;
        }
    }
}

}  // namespace slamk

// fp64 MFMA GEMM for gfx950 (v_mfma_f64_16x16x4_f64), row-major, strided-batched.
//
//   C = alpha * A * op(B) + beta * C        A: M x K (lda),  C: M x N (ldc)
//   op(B) = B^T with B: N x K (ldb)  when TB,   op(B) = B with B: K x N (ldb) otherwise.
//
// One 256-thread workgroup (4 waves as 2 x 2) owns a BM x BN tile of C; each wave owns a
// (BM/2) x (BN/2) sub-tile as TM x TN accumulators of 16 x 16.  K is walked in slabs of 16:
// the next slab is fetched from global memory into registers while the MFMAs of the current
// slab run out of LDS (register-staged software pipeline over two LDS buffers, one barrier per slab).
//
// LDS images are laid out so that the operand reads are bank-conflict free:
//   k-contiguous operands (A, and B when TB) : [row][20]  -> lane (i = l&15, kk = l>>4) reads the
//       16 bytes at dword 40 i + 4 kk (+16 t) with ds_read_b128; each of that instruction's four
//       16-lane groups covers all 64 banks exactly once, and the staging ds_write_b128 is clean too.
//   n-contiguous operand (B when !TB)        : [k][BN+16] -> row stride 2(BN+16) dwords == 32 mod 64,
//       so the two k rows a half-wave touches land in opposite halves of the 64 banks.
//
// All of M, N, K must be multiples of 16 (the filter keeps its matrices zero-padded to 16);
// tile edges are handled by zero-filling loads and guarding stores at that granularity.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>

namespace slamk {

typedef double v4d __attribute__((ext_vector_type(4)));

enum Epilogue {
    EPI_NONE = 0,
    EPI_DUAL_COLSCALE = 3  // C = acc and C2[i][j] = acc * D[j]   (K and K R with diagonal R in one pass)
};

// Which k range can be skipped because one operand is block lower-triangular.
enum TriMode {
    TRI_NONE = 0,
    TRI_B_KLE_N = 1,     // op(B)[k][n] == 0 for k >= n0 + BN   (B^T with B lower)
    TRI_B_KGE_N = 2,     // op(B)[k][n] == 0 for k <  n0        (B lower, not transposed)
    TRI_A_KLE_M = 3      // A[m][k]     == 0 for k >= m0 + BM   (A lower)
};

struct GemmArgs {
    const double* A;
    const double* B;
    double* C;
    double* C2;           // second output of EPI_DUAL_COLSCALE (same ld and strides as C)
    const double* D;      // column-scale vector of EPI_DUAL_COLSCALE (null otherwise)
    int M, N, K;
    int lda, ldb, ldc;
    long long sA1, sB1, sC1, sD1;   // batch level 1 (independent filters)
    long long sA2, sB2, sC2;        // batch level 2 (sub-blocks inside one matrix)
    int nb2;                        // level-2 count (>= 1); blockIdx.z = b1 * nb2 + b2
    double alpha, beta;
};

template <int BM, int BN, bool TB, int EPI, bool LOWER, int TRI>
__global__ void __launch_bounds__(256, 2) gemm_f64_kernel(GemmArgs g)
{
    constexpr int LDS_K = 20;                       // row stride of k-contiguous images (doubles)
    constexpr int LDS_N = BN + 16;                  // row stride of the n-contiguous B image
    constexpr int TM = BM / 32, TN = BN / 32;
    constexpr int NBUF = (BM >= 128) ? 2 : 1;       // big tiles: two slabs in flight, one barrier per slab
    constexpr int A_SZ = BM * LDS_K;
    constexpr int B_SZ = TB ? BN * LDS_K : 16 * LDS_N;
    __shared__ __attribute__((aligned(16))) double As[NBUF * A_SZ];
    __shared__ __attribute__((aligned(16))) double Bs[NBUF * B_SZ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // Tile of this workgroup.  With a trimmed k range the tiles differ in length by up to the tile count: the
    // workgroups are dealt in dispatch order (x fastest) to the tiles longest first, so that the short ones fill
    // the tail of the launch.  (Giving each XCD a contiguous eighth of the tile sequence, for L2 locality, measured
    // no difference at 4096..10016: the kernel is not bound by L2 misses.)
    int tm = blockIdx.y, tn = blockIdx.x;
    if (TRI != TRI_NONE) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, nx = gridDim.x, ny = gridDim.y;
        if (TRI == TRI_A_KLE_M) { tm = ny - 1 - lin / nx; tn = lin % nx; }
        else { tn = lin / ny; tm = lin % ny; if (TRI == TRI_B_KLE_N) tn = nx - 1 - tn; }
    }
    const int m0 = tm * BM, n0 = tn * BN;
    if (LOWER && n0 > m0 + BM - 1) return;          // tile lies strictly above the diagonal

    const int b1 = blockIdx.z / g.nb2, b2 = blockIdx.z - b1 * g.nb2;
    const double* __restrict__ A = g.A + b1 * g.sA1 + b2 * g.sA2;
    const double* __restrict__ B = g.B + b1 * g.sB1 + b2 * g.sB2;
    double* __restrict__ C = g.C + b1 * g.sC1 + b2 * g.sC2;
    double* __restrict__ C2 = (EPI == EPI_DUAL_COLSCALE) ? g.C2 + b1 * g.sC1 + b2 * g.sC2 : nullptr;
    const double* __restrict__ D = g.D ? g.D + b1 * g.sD1 : nullptr;

    int kb = 0, ke = g.K;
    if (TRI == TRI_B_KLE_N) ke = min(g.K, n0 + BN);
    if (TRI == TRI_B_KGE_N) kb = min(g.K, n0);
    if (TRI == TRI_A_KLE_M) ke = min(g.K, m0 + BM);

    v4d acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    double2 ra[TM], rb[TN];
    const int arow = tid >> 3, akp = (tid & 7) * 2;                 // k-contiguous staging map
    constexpr int BLPR = BN / 2;                                    // lanes per row of the n-contiguous image
    const int brow = tid / BLPR, bcol = (tid % BLPR) * 2;

    auto fetch = [&](int k0) {
#pragma unroll
        for (int r = 0; r < TM; ++r) {
            const int gr = m0 + arow + 32 * r;
            ra[r] = (gr < g.M) ? *reinterpret_cast<const double2*>(A + (long long)gr * g.lda + k0 + akp)
                               : make_double2(0.0, 0.0);
        }
        if (TB) {
#pragma unroll
            for (int r = 0; r < TN; ++r) {
                const int gr = n0 + arow + 32 * r;
                rb[r] = (gr < g.N) ? *reinterpret_cast<const double2*>(B + (long long)gr * g.ldb + k0 + akp)
                                   : make_double2(0.0, 0.0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < TN; ++r) {
                const int k = brow + (256 / BLPR) * r;
                const int gc = n0 + bcol;
                rb[r] = (gc < g.N) ? *reinterpret_cast<const double2*>(B + (long long)(k0 + k) * g.ldb + gc)
                                   : make_double2(0.0, 0.0);
            }
        }
    };
    auto stash = [&](int buf) {
        double* Ad = As + buf * A_SZ;
        double* Bd = Bs + buf * B_SZ;
#pragma unroll
        for (int r = 0; r < TM; ++r)
            *reinterpret_cast<double2*>(&Ad[(arow + 32 * r) * LDS_K + akp]) = ra[r];
        if (TB) {
#pragma unroll
            for (int r = 0; r < TN; ++r)
                *reinterpret_cast<double2*>(&Bd[(arow + 32 * r) * LDS_K + akp]) = rb[r];
        } else {
#pragma unroll
            for (int r = 0; r < TN; ++r)
                *reinterpret_cast<double2*>(&Bd[(brow + (256 / BLPR) * r) * LDS_N + bcol]) = rb[r];
        }
    };

    // Lane l = (li = l & 15, lk = l >> 4) feeds row/column li of a 16 x 16 tile.  Each MFMA sums over 4
    // values of k, one per lk; WHICH 4 is free as long as A and B agree, so lane lk takes the adjacent
    // pair k = 8 t + 2 lk + {0, 1}: one 16-byte LDS read serves two MFMAs.
    const int li = lane & 15, lk = lane >> 4;
    const double* aBase = &As[(wr * (BM / 2) + li) * LDS_K + 2 * lk];
    const double* bBase = TB ? &Bs[(wc * (BN / 2) + li) * LDS_K + 2 * lk] : &Bs[2 * lk * LDS_N + wc * (BN / 2) + li];

    // Small tile: a wavefront skips the 16 x 16 blocks of its sub-tile that lie outside C or (LOWER) strictly above the
    // diagonal -- batched matrices of 130..300 rows are a few tiles with wide rims, and the rim blocks would otherwise
    // cost as much as the live ones.  The conditions are uniform over the wavefront.
    constexpr bool SKIP = BM < 128;
    bool live[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int r0 = m0 + wr * (BM / 2) + i * 16, c0 = n0 + wc * (BN / 2) + j * 16;
            live[i][j] = !SKIP || (r0 < g.M && c0 < g.N && !(LOWER && c0 > r0 + 15));
        }
    auto compute = [&](int buf) {
        const double* aB = aBase + buf * A_SZ;
        const double* bB = bBase + buf * B_SZ;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            double2 a2[TM], b2[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a2[i] = *reinterpret_cast<const double2*>(&aB[i * 16 * LDS_K + 8 * t]);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (TB) b2[j] = *reinterpret_cast<const double2*>(&bB[j * 16 * LDS_K + 8 * t]);
                else b2[j] = make_double2(bB[(8 * t) * LDS_N + j * 16], bB[(8 * t + 1) * LDS_N + j * 16]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (live[i][j]) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[i].x, b2[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (live[i][j]) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[i].y, b2[j].y, acc[i][j], 0, 0, 0);
        }
    };

    if (NBUF == 2) {
        // one barrier per slab: while the MFMAs of slab k run out of buffer k&1, slab k+1 (already in
        // registers) is written to the other buffer and slab k+2 is requested from global memory
        if (kb < ke) {
            fetch(kb);
            stash(0);
            if (kb + 16 < ke) fetch(kb + 16);
        }
        __syncthreads();
        int buf = 0;
        for (int k0 = kb; k0 < ke; k0 += 16, buf ^= 1) {
            if (k0 + 16 < ke) {
                stash(buf ^ 1);
                if (k0 + 32 < ke) fetch(k0 + 32);
            }
            compute(buf);
            __syncthreads();
        }
    } else {
        if (kb < ke) fetch(kb);
        for (int k0 = kb; k0 < ke; k0 += 16) {
            __syncthreads();             // every wave is done reading the previous slab
            stash(0);
            __syncthreads();
            if (k0 + 16 < ke) fetch(k0 + 16);
            compute(0);
        }
    }

    // accumulator layout of v_mfma_f64_16x16x4_f64: lane l, register r -> row (l>>4) + 4 r, col l & 15
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (!live[i][j]) continue;                 // outside C, or above the diagonal like a skipped tile: not stored
            const int col = n0 + wc * (BN / 2) + j * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr * (BM / 2) + i * 16 + lk + 4 * r;
                if (row < g.M && col < g.N) {
                    double v = g.alpha * acc[i][j][r];
                    double* c = C + (long long)row * g.ldc + col;
                    if (g.beta != 0.0) v += g.beta * (*c);
                    *c = v;
                    if (EPI == EPI_DUAL_COLSCALE) C2[(long long)row * g.ldc + col] = v * D[col];
                }
            }
        }
    }
}

// Host-side launcher.  64 x 64 tiles unless 128 x 128 tiles are expected to be faster:
//  - products that skip tiles (LOWER) or trim their k range (TRI) skip and trim at tile granularity, and measured faster
//    with the small tile at every size up to 10016;
//  - a plain product with the large tile runs about 5 % faster per FLOP, but only when its tile count fills the
//    512 workgroup slots of the GPU (2 per CU) evenly: 752 x 6016 is 282 large tiles, about half a round, against
//    1128 small ones.  The choice compares the two fill fractions.
// SLAMK_GEMM_BIG_MIN (0: never the large tile) overrides the smallest dimension that may take the large tile.
inline bool gemm_big_tile(int M, int N, int batch, bool lower, int tri)
{
    static const int big_min = [] { const char* e = getenv("SLAMK_GEMM_BIG_MIN"); return e ? atoi(e) : 512; }();
    if (lower || tri != TRI_NONE || big_min <= 0 || M < big_min || N < big_min) return false;
    auto fill = [&](int t) {
        const long long tiles = (long long)((M + t - 1) / t) * ((N + t - 1) / t) * batch;
        const long long slots = 512;
        return (double)tiles / (double)((tiles + slots - 1) / slots * slots);
    };
    return fill(128) > 0.95 * fill(64);
}

template <bool TB, int EPI, bool LOWER, int TRI>
inline hipError_t launch_gemm(const GemmArgs& g, int batch, hipStream_t stream)
{
    if (g.M <= 0 || g.N <= 0 || batch <= 0) return hipSuccess;
    const bool big = gemm_big_tile(g.M, g.N, batch, LOWER, TRI);
    if (big) {
        dim3 grid((g.N + 127) / 128, (g.M + 127) / 128, batch);
        hipLaunchKernelGGL((gemm_f64_kernel<128, 128, TB, EPI, LOWER, TRI>), grid, dim3(256), 0, stream, g);
    } else {
        dim3 grid((g.N + 63) / 64, (g.M + 63) / 64, batch);
        hipLaunchKernelGGL((gemm_f64_kernel<64, 64, TB, EPI, LOWER, TRI>), grid, dim3(256), 0, stream, g);
    }
    return hipGetLastError();
}

// C = beta C + alpha sum_i P_i over `parts` partial products of M x N doubles (ld N), summed in ascending order.
// grid (ceil(N / 256), M), 256 threads.
__global__ void __launch_bounds__(256) splitk_reduce_kernel(double* __restrict__ C, int ldc, const double* __restrict__ P,
                                                             int M, int N, int parts, double alpha, double beta)
{
    const int col = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    if (col >= N) return;
    const long long o = (long long)row * N + col, sp = (long long)M * N;
    double acc = P[o];
    for (int i = 1; i < parts; ++i) acc += P[o + i * sp];
    double* c = C + (long long)row * ldc + col;
    double v = alpha * acc;
    if (beta != 0.0) v += beta * (*c);
    *c = v;
}

// How many slices of the k range fill the GPU for an M x N x K product whose tile grid alone is much smaller than the
// 256 CUs (a row block of a sharded matrix times a long k range): 1 when the grid is large enough or k is short.
// SLAMK_SPLITK_MIN_SLICE: shortest slice worth a launch of its own (default 512; tests lower it to reach this path
// with small matrices).
inline int splitk_parts(int M, int N, int K, size_t ws_elems)
{
    const char* e = getenv("SLAMK_SPLITK_MIN_SLICE");
    const int min_slice = e ? std::max(16, atoi(e)) : 512;
    const long long tiles = (long long)((M + 63) / 64) * ((N + 63) / 64);       // a product this small takes 64 x 64 tiles
    if (tiles >= 400 || K < 4 * min_slice) return 1;
    int parts = 8;
    while (parts > 1 && (tiles * parts > 2048 || K / parts < min_slice || K % (16 * parts) != 0 ||
                         (size_t)parts * M * N > ws_elems))
        parts /= 2;
    return parts;
}

// launch_gemm (one matrix, no epilogue, no triangular trimming) with the k range cut into slices that run as one
// batched launch into partial products in `ws`, then summed into C in a fixed order (deterministic).
template <bool TB>
inline hipError_t launch_gemm_splitk(const GemmArgs& g, double* ws, size_t ws_elems, hipStream_t stream)
{
    const int parts = ws ? splitk_parts(g.M, g.N, g.K, ws_elems) : 1;
    if (parts < 2) return launch_gemm<TB, EPI_NONE, false, TRI_NONE>(g, 1, stream);
    GemmArgs p = g;
    p.K = g.K / parts;
    p.C = ws; p.ldc = g.N;
    p.sA2 = p.K;
    p.sB2 = TB ? (long long)p.K : (long long)p.K * g.ldb;
    p.sC2 = (long long)g.M * g.N;
    p.nb2 = parts;
    p.alpha = 1.0; p.beta = 0.0;
    hipError_t e = launch_gemm<TB, EPI_NONE, false, TRI_NONE>(p, parts, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((g.N + 255) / 256, g.M), dim3(256), 0, stream, g.C, g.ldc, ws, g.M, g.N,
                       parts, g.alpha, g.beta);
    return hipGetLastError();
}

}  // namespace slamk

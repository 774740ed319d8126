// Fused covariance update (Joseph form, src/slam/main.cpp:561-569) for filters of exactly seven row blocks
// (97 <= n <= 112) and 112 <= mlc <= 240 measurement columns in chunks, one workgroup of FOUR wavefronts per filter:
// one wavefront per SIMD with up to 512 registers.  Same arithmetic as joseph_merged_kernel (joseph_kernel.hip.h),
// whose header describes the phases.  What shapes this variant is a property of the fp64 matrix instruction on gfx950
// (scripts/mfma_overlap_probe.hip): while one runs, NO vector ALU instruction of any wavefront of that SIMD issues, and
// the first one behind it waits another 40 cycles; LDS and memory instructions of other wavefronts do issue.  Every
// VALU instruction in a loop of matrix instructions therefore costs its 8 cycles in full.  So:
//   * four wavefronts instead of eight: what every wavefront repeats for itself (rows of H per chunk, addresses) is
//     done four times per chunk instead of seven;
//   * the 28 lower tiles of the result are dealt as 7 per wavefront: wavefront w owns column w (7 - w tiles) and, for
//     w >= 1, column 7 - w (w tiles): every SIMD runs the same number of matrix instructions in every phase but A P
//     (seven row blocks on four SIMDs: wavefront 0 has one, and repeats it as a dummy second one);
//   * the chunk loop is unrolled over its up to 15 chunks, so that ring slots, register sets and table rows are
//     constants of the instruction (address + immediate), and the loads are unconditional (the wait for a chunk counts
//     the loads requested after it, which the compiler can only do along straight code);
//   * K H is accumulated ONTO the entries of A it changes (they are the C operand of its matrix instructions, -H is
//     the B operand), rows of H that a lane has no entry of come from a zero in LDS instead of a select;
//   * the chunks of K go through a ring of THREE buffers with one barrier per chunk at the top of the iteration:
//     what a half chunk of products reads from LDS is requested before the half chunk ahead of it is issued;
//   * P is symmetric: only its 28 lower tiles are read and kept in LDS, A P reads the upper ones transposed.
// LDS: 49 tiles (A, then P / AP) + 21 (chunks of K) + 10 (tables) = 160 KB.
#pragma once
#include <hip/hip_runtime.h>

#include "joseph_kernel.hip.h"

namespace slamk {

constexpr int QUAD_MIN_COLS = 112, QUAD_MAX_COLS = 240;     // mlc this kernel is launched for

template <bool SIMPLE>
__global__ void __launch_bounds__(256) joseph_quad_kernel(FilterView fv, const double* __restrict__ K, int ldk,
                                                          long long sK, int mlc, double* __restrict__ dx, int apply,
                                                          int* __restrict__ applied)
{
    constexpr int NT = 7, ROWS = QUAD_MAX_COLS;
    extern __shared__ double lds[];
    double* kst = lds + NT * NT * 256;                  // 3 NT tiles: ring of 16-column chunks of K
    double* rds = kst + 3 * NT * 256;                   // [240] diagonal of R
    double* dxs = rds + ROWS;                           // [128] dx = K e
    double* kalt = dxs + 128;                           // [128] the altitude column of K
    double* cfa = kalt + 128;                           // [240][4] row q of H: entries in columns 0..2, innovation e_q
    double* chb = cfa + 4 * ROWS;                       // [240][4] row q of -H: entries in columns cs .. cs + 2, then 0
    int* ccs = reinterpret_cast<int*>(chb + 4 * ROWS);  // [240] cs of row q (-1: none)
    double* zero = reinterpret_cast<double*>(ccs + ROWS);    // [24] zeros
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int Ja = wave, nta = NT - wave;               // column Ja: tiles (Ja + t, Ja), t < nta
    const bool hasb = wave > 0;
    const int Jb = hasb ? NT - wave : 0;                // column Jb: tiles (t, Jb), nta <= t < NT  (wavefront 0: none)
    auto sec = [&](int t) { return t >= nta; };         // (scalar) tile t belongs to column Jb
    auto rowblk = [&](int t) { return t >= nta ? t : wave + t; };
    const int ldn = fv.ldn;
    double* Pb = fv.P + (long long)b * ldn * ldn;
    const double* Kb = K + (long long)b * sK;
    const long long rb = (long long)b * fv.ldm;
    const int nf = fv.nf[b];
    const int ia = fv.m[b] - 1;                         // altitude row of H / column of K
    const bool border = (ia == mlc);                    // not covered by the chunk loop
    const int nchunk = mlc / 16;
    if (nchunk < QUAD_MIN_COLS / 16 || nchunk > QUAD_MAX_COLS / 16) return;       // (not launched so)
    JOSEPH_STAMP(0);

    // Chunk loader: wavefront w brings tiles w and w + 4 of every chunk (wavefront 3: tile 3 twice), element
    // (lk + 4 r, li) of the tile per lane; a uniform base advances, the lane's offsets stay.
    const int ktile1 = wave + 4 < NT ? wave + 4 : wave;
    // (buffer loads: descriptor and chunk / tile offset in scalar registers, the lane's offset in one vector register,
    // no address arithmetic on the vector ALU)
    typedef int v2i_t __attribute__((ext_vector_type(2)));
    auto buffer_ld = [](__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
        const v2i_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
        return __hiloint2double(v.y, v.x);
    };
    const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Kb), 0, 16 * NT * ldk * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(Pb, 0, 16 * NT * ldn * 8, 0x00020000);
    int kvo[2][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        kvo[0][r] = ((16 * wave + lk + 4 * r) * ldk + li) * 8;
        kvo[1][r] = ((16 * ktile1 + lk + 4 * r) * ldk + li) * 8;
    }
    double pre[2][2][4];                                // chunk k waits in set k & 1, requested two chunks ahead
    auto request_chunk = [&](int k, auto set_) {
        constexpr int S = decltype(set_)::value;
        // (past the last chunk the last one is read again: no branch around a load)
        const int soff = 128 * min(k, nchunk - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pre[S][0][r] = buffer_ld(krsrc, kvo[0][r], soff);
            pre[S][1][r] = buffer_ld(krsrc, kvo[1][r], soff);
        }
    };
    double* st0[4];
    double* st1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        st0[r] = kst + wave * 256 + tsw(lk + 4 * r, li);
        st1[r] = kst + ktile1 * 256 + tsw(lk + 4 * r, li);
    }
    auto stage_chunk = [&](auto slot_, auto set_) {     // (never skipped: past the last chunk it refills a buffer that has been read)
        constexpr int S = decltype(set_)::value, SLOT = decltype(slot_)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) st0[r][SLOT * NT * 256] = pre[S][0][r];
#pragma unroll
        for (int r = 0; r < 4; ++r) st1[r][SLOT * NT * 256] = pre[S][1][r];
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // The first chunks of K are requested before anything that waits for the filter's sizes, then the small loads
    // (vector memory loads complete in order): altitude entries, compact rows.
    request_chunk(0, I0{});
    request_chunk(1, I1{});
    for (int t = tid; t < NT * NT * 128; t += 256) reinterpret_cast<double2*>(lds)[t] = make_double2(0.0, 0.0);
    double ka_own = 0.0, e_alt = 0.0, hal[3] = {0.0, 0.0, 0.0};
    if (tid < 16 * NT) {
        ka_own = Kb[(long long)tid * ldk + ia];
        e_alt = fv.e[rb + ia];
#pragma unroll
        for (int c = 0; c < 3; ++c) hal[c] = fv.ha[(rb + ia) * 3 + c];
    }
    // compact rows of H for the chunk loop (one row per thread), requested here, stored below.  The altitude row is
    // taken ahead of the loop (one thread per row of A, below), rows past it are padding.
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
    // P is symmetric: only its 28 lower tiles are read, seven per wavefront (tile number wavefront + 4 u, counted along
    // the rows of the triangle), one requested per chunk during the first chunks behind that chunk's request for K.
    constexpr int PT = NT * (NT + 1) / 8;
    static_assert(PT * 8 == NT * (NT + 1) && PT == 7, "seven tiles for every wavefront, requested by the first seven chunks");
    double pre_p[PT][4];
    auto ptile_row = [&](int u) {
        const int L = wave + 4 * u;
        return (L >= 1) + (L >= 3) + (L >= 6) + (L >= 10) + (L >= 15) + (L >= 21);
    };
    int pvo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pvo[r] = ((lk + 4 * r) * ldn + li) * 8;
    auto request_p = [&](int u) {
        const int I = ptile_row(u), Jt = wave + 4 * u - I * (I + 1) / 2;
        const int soff = (16 * I * ldn + 16 * Jt) * 8;
#pragma unroll
        for (int r = 0; r < 4; ++r) pre_p[u][r] = buffer_ld(prsrc, pvo[r], soff);
    };

    if (tid < ROWS) {
        rds[tid] = t_rd;
        ccs[tid] = t_cs;
        *reinterpret_cast<double2*>(cfa + 4 * tid) = make_double2(t_ha[0], t_ha[1]);
        *reinterpret_cast<double2*>(cfa + 4 * tid + 2) = make_double2(t_ha[2], t_e);
        *reinterpret_cast<double2*>(chb + 4 * tid) = make_double2(-t_hb[0], -t_hb[1]);
        *reinterpret_cast<double2*>(chb + 4 * tid + 2) = make_double2(-t_hb[2], 0.0);
    }
    if (tid < 24) zero[tid] = 0.0;
    if (tid < 16 * NT) {
        kalt[tid] = ka_own;
        dxs[tid] = ka_own * e_alt;
    }
    stage_chunk(I0{}, I0{});
    request_chunk(2, I0{});
    stage_chunk(I1{}, I1{});
    request_chunk(3, I1{});
    __syncthreads();
    JOSEPH_STAMP(1);
    // A = I, and the altitude row (H row ia = (ha, 0), main.cpp:354): A[j][0..2] = I - K[j][ia] ha, one thread per row j
    // (the barrier of the first chunk orders these writes before the loop's updates of A)
    if (tid < 16 * NT) {
        lds[((tid >> 4) * (NT + 1)) * 256 + tsw(tid & 15, tid & 15)] = 1.0;
        double* row = lds + ((tid >> 4) * NT) * 256;
#pragma unroll
        for (int c = 0; c < 3; ++c) row[tsw(tid & 15, c)] = (tid == c ? 1.0 : 0.0) - ka_own * hal[c];
    }

    // ---- chunks of K: K R K^T, K H and K e -----------------------------------------------------------------------------
    v4d o[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) o[t] = (v4d){0.0, 0.0, 0.0, 0.0};
    v4d g2a = (v4d){0.0, 0.0, 0.0, 0.0}, g2b = g2a;     // rows lk + 4 r of the block: columns 0..2 of K H, column 3: K e
    double b1[2][4], b2[2][4], rq[2][4];                // rows of -H, of [ha | e] and of R for a chunk, by parity of the chunk
    int cb[2] = {0, 0};                                 // first column of the chunk's entries of H outside columns 0..2
    const int col0 = li * 16 + ((4 * lk) ^ (li & 14)), col1 = li * 16 + ((4 * lk + 2) ^ (li & 14));
    // fragments of a chunk: k steps 0, 1 (f0) and 2, 3 (f1) of the seven tiles, f?j: tile Jb; lane addresses in the ring
    double2 f0[NT], f1[NT], f0j, f1j;
    const double* fa0[NT];
    const double* fa1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { fa0[t] = kst + rowblk(t) * 256 + col0; fa1[t] = kst + rowblk(t) * 256 + col1; }
    const double* fj0 = kst + Jb * 256 + col0;
    const double* fj1 = kst + Jb * 256 + col1;
    double* ablk_a = lds + Ja * NT * 256;               // row blocks of A that this wavefront updates
    double* ablk_b = lds + Jb * NT * 256;
    // lane addresses in the tables (row 4 lk of a chunk); a lane with li >= 4 has no entry of [ha | e] and reads zeros
    const int* cqp = ccs + 4 * lk;
    const double* rdp = rds + 4 * lk;
    const double* b2p = li < 4 ? cfa + 16 * lk + li : zero;
    const int b2inc = li < 4 ? 64 : 0;
    const double* chbl = chb + 16 * lk;
    int rowo[4], rowm[4];                               // tile image: element (row, col) at row * 16 + (col ^ (row & 14))
#pragma unroll
    for (int r = 0; r < 4; ++r) { rowo[r] = (lk + 4 * r) * 16; rowm[r] = (lk + 4 * r) & 14; }

    // rows of H and R for chunk C into set S: [ha | e] and R first, -H through the column offsets of its rows
    int4 cq;
    auto tables_a = [&](auto c_, auto set_) {
        constexpr int C = decltype(c_)::value, S = decltype(set_)::value;
        cb[S] = ccs[16 * C];
        cq = *reinterpret_cast<const int4*>(cqp + 16 * C);
#pragma unroll
        for (int s = 0; s < 4; ++s) b2[S][s] = b2p[4 * s];
        b2p += b2inc;                                                      // (row 4 lk of the next chunk, or still the zeros)
        const double2 r01 = *reinterpret_cast<const double2*>(rdp + 16 * C);
        const double2 r23 = *reinterpret_cast<const double2*>(rdp + 16 * C + 2);
        rq[S][0] = r01.x; rq[S][1] = r01.y; rq[S][2] = r23.x; rq[S][3] = r23.y;
    };
    auto tables_b = [&](auto c_, auto set_) {
        constexpr int C = decltype(c_)::value, S = decltype(set_)::value;
        cb[S] = max(cb[S], 0);
        const int t = cb[S] + li;
        const int cqs[4] = {cq.x, cq.y, cq.z, cq.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // entry t - cs of the row if that is 0, 1 or 2, else the zero behind them (cs = -1: t - cs >= 3)
            const unsigned d = min((unsigned)(t - cqs[s]), 3u);
            b1[S][s] = chbl[64 * C + 4 * s + d];
        }
    };
    // One chunk.  C and with it the parity of the register sets, the ring slots and the table rows are compile-time.
    //   top:        barrier (publishes chunk C + 1, frees the buffer of chunk C - 1 for chunk C + 2); requests: fragments
    //               f1 of this chunk, the entries of A under this chunk's K H, [ha | e] and R of chunk C + 1
    //   first half: products of fragments f0
    //   between:    -H of chunk C + 1, fragments f0 of chunk C + 1
    //   second half: products of fragments f1
    //   bottom:     A written back, chunk C + 2 from registers into its buffer, chunk C + 4 and a P tile requested
    auto iteration = [&](auto c_) {
        constexpr int C = decltype(c_)::value, X = C & 1, Y = X ^ 1;
        constexpr int SLOT = C % 3, SLOTN = (C + 1) % 3, SLOTS = (C + 2) % 3;
        constexpr bool NEXT = C + 1 < QUAD_MAX_COLS / 16;                  // (a chunk C + 1 can exist)
        using Cur = std::integral_constant<int, X>;
        using Oth = std::integral_constant<int, Y>;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t) f1[t] = *reinterpret_cast<const double2*>(fa1[t] + SLOT * NT * 256);
        f1j = *reinterpret_cast<const double2*>(fj1 + SLOT * NT * 256);
        // entries of A under K H of this chunk: column cb + li, rows lk + 4 r of the blocks.  Column 16 NT - 1 is
        // padding for every n this kernel takes (n = 9 + 3 nf is never 112) and no row of H reaches it: lanes past
        // the last column meet there and write back what they read.
        const int colA = min(cb[X] + li, 16 * NT - 1);
        int ea[4];
        v4d g1a, g1b;
        {
            const int tb = (colA >> 4) * 256, cl = colA & 15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ea[r] = tb + rowo[r] + (cl ^ rowm[r]);
                g1a[r] = ablk_a[ea[r]];
                g1b[r] = ablk_b[ea[r]];
            }
        }
        if constexpr (NEXT) tables_a(std::integral_constant<int, C + 1>{}, Oth{});
        __builtin_amdgcn_sched_barrier(0);
        // (second half: K H first, so that its result has left the matrix pipeline when A is written back)
        auto products = [&](const double2 (&f)[NT], const double2& fj, auto h_) {
            constexpr int H = decltype(h_)::value;
            auto kh = [&]() {
                g1a = __builtin_amdgcn_mfma_f64_16x16x4f64(f[0].x, b1[X][2 * H], g1a, 0, 0, 0);
                g1b = __builtin_amdgcn_mfma_f64_16x16x4f64(fj.x, b1[X][2 * H], g1b, 0, 0, 0);
                g1a = __builtin_amdgcn_mfma_f64_16x16x4f64(f[0].y, b1[X][2 * H + 1], g1a, 0, 0, 0);
                g1b = __builtin_amdgcn_mfma_f64_16x16x4f64(fj.y, b1[X][2 * H + 1], g1b, 0, 0, 0);
                g2a = __builtin_amdgcn_mfma_f64_16x16x4f64(f[0].x, b2[X][2 * H], g2a, 0, 0, 0);
                g2b = __builtin_amdgcn_mfma_f64_16x16x4f64(fj.x, b2[X][2 * H], g2b, 0, 0, 0);
                g2a = __builtin_amdgcn_mfma_f64_16x16x4f64(f[0].y, b2[X][2 * H + 1], g2a, 0, 0, 0);
                g2b = __builtin_amdgcn_mfma_f64_16x16x4f64(fj.y, b2[X][2 * H + 1], g2b, 0, 0, 0);
            };
            if constexpr (H == 1) kh();
            if constexpr (!SIMPLE) {
                double ba0 = f[0].x * rq[X][2 * H], ba1 = f[0].y * rq[X][2 * H + 1];
                double bb0 = fj.x * rq[X][2 * H], bb1 = fj.y * rq[X][2 * H + 1];
                asm volatile("" : "+v"(ba0), "+v"(ba1), "+v"(bb0), "+v"(bb1));      // (four products, then selects: not a product per select)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bool s2 = (t >= NT - 3) && sec(t);               // (tiles 0..3 are never in column Jb)
                    o[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[t].x, s2 ? bb0 : ba0, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[t].y, s2 ? bb1 : ba1, o[t], 0, 0, 0);
                }
            }
            if constexpr (H == 0) kh();
        };
        products(f0, f0j, I0{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NEXT) tables_b(std::integral_constant<int, C + 1>{}, Oth{});
#pragma unroll
        for (int t = 0; t < NT; ++t) f0[t] = *reinterpret_cast<const double2*>(fa0[t] + SLOTN * NT * 256);
        f0j = *reinterpret_cast<const double2*>(fj0 + SLOTN * NT * 256);       // (past the last chunk: read, never used)
        __builtin_amdgcn_sched_barrier(0);
        products(f1, f1j, I1{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) ablk_a[ea[r]] = g1a[r];
        if (hasb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ablk_b[ea[r]] = g1b[r];
        }
        stage_chunk(std::integral_constant<int, SLOTS>{}, Cur{});
        request_chunk(C + 4, Cur{});
        if constexpr (C < PT) request_p(C);
    };
    // first half of chunk 0, tables of chunk 0
#pragma unroll
    for (int t = 0; t < NT; ++t) f0[t] = *reinterpret_cast<const double2*>(fa0[t]);
    f0j = *reinterpret_cast<const double2*>(fj0);
    tables_a(I0{}, I0{});
    tables_b(I0{}, I0{});
    iteration(std::integral_constant<int, 0>{});
    iteration(std::integral_constant<int, 1>{});
    iteration(std::integral_constant<int, 2>{});
    iteration(std::integral_constant<int, 3>{});
    iteration(std::integral_constant<int, 4>{});
    iteration(std::integral_constant<int, 5>{});
    iteration(std::integral_constant<int, 6>{});
    if (nchunk > 7) { iteration(std::integral_constant<int, 7>{});
    if (nchunk > 8) { iteration(std::integral_constant<int, 8>{});
    if (nchunk > 9) { iteration(std::integral_constant<int, 9>{});
    if (nchunk > 10) { iteration(std::integral_constant<int, 10>{});
    if (nchunk > 11) { iteration(std::integral_constant<int, 11>{});
    if (nchunk > 12) { iteration(std::integral_constant<int, 12>{});
    if (nchunk > 13) { iteration(std::integral_constant<int, 13>{});
    if (nchunk > 14) { iteration(std::integral_constant<int, 14>{}); } } } } } } } }
    {                                                        // position columns and dx of the row blocks
        if (li < 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ablk_a[tsw(lk + 4 * r, li)] -= g2a[r];
            if (hasb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ablk_b[tsw(lk + 4 * r, li)] -= g2b[r];
            }
        } else if (li == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dxs[16 * Ja + lk + 4 * r] += g2a[r];
            if (hasb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dxs[16 * Jb + lk + 4 * r] += g2b[r];
            }
        }
    }
    if (border && !SIMPLE) {                             // + k_a r_a k_a^T for the altitude column, which no chunk holds
        const double ra = fv.rdiag[rb + ia];
        const double cja = kalt[16 * Ja + li] * ra, cjb = kalt[16 * Jb + li] * ra;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const double cj = sec(t) ? cjb : cja;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[t][r] = fma(kalt[16 * rowblk(t) + lk + 4 * r], cj, o[t][r]);
        }
    }
    __syncthreads();
    JOSEPH_STAMP(2);
    if (dx && tid < 16 * NT) dx[(long long)b * ldn + tid] = dxs[tid];
    // rows of A in operand layout: lane (li, lk) holds A[16 J + li][16 kt + 4 lk + s]
    double afa[4 * NT], afb[4 * NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        tile_ld4(ablk_a + kt * 256, li, 4 * lk, afa + 4 * kt);
        tile_ld4(ablk_b + kt * 256, li, 4 * lk, afb + 4 * kt);
    }
    __syncthreads();

    // ---- lower tiles of P into LDS --------------------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int I = ptile_row(u), Jt = wave + 4 * u - I * (I + 1) / 2;
        double* tij = lds + (I * NT + Jt) * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) tij[tsw(lk + 4 * r, li)] = pre_p[u][r];
    }
    __syncthreads();
    // gate on the velocity correction and mean update (src/slam/main.cpp:229-232)
    if (apply && wave == 0) {
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
    // ---- AP(J, :) = A(J, :) P for the row blocks Ja and Jb, sharing the reads of P --------------------------------------
    // P(kt, Jc) is tile (kt, Jc) for kt >= Jc and the transpose of tile (Jc, kt) above the diagonal (the tile image
    // reads without bank conflicts along rows and along columns).
    {
        v4d acca[NT], accb[NT];
#pragma unroll
        for (int Jc = 0; Jc < NT; ++Jc) { acca[Jc] = (v4d){0.0, 0.0, 0.0, 0.0}; accb[Jc] = acca[Jc]; }
        const double* pd[4];
        const double* pt[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { pd[s] = lds + tsw(4 * lk + s, li); pt[s] = lds + tsw(li, 4 * lk + s); }
        // (the seven entries of P of step s + 1 are requested ahead of the fourteen products of step s: alone on its SIMD, a
        // wavefront has nothing else to cover the latency of LDS)
        auto pload = [&](int s, double (&p)[NT]) {
            const int kt = s >> 2;
#pragma unroll
            for (int Jc = 0; Jc < NT; ++Jc)
                p[Jc] = kt >= Jc ? pd[s & 3][(kt * NT + Jc) * 256] : pt[s & 3][(Jc * NT + kt) * 256];
        };
        double pc[2][NT];
        pload(0, pc[0]);
#pragma unroll
        for (int s = 0; s < 4 * NT; ++s) {
            if (s + 1 < 4 * NT) pload(s + 1, pc[(s + 1) & 1]);
#pragma unroll
            for (int Jc = 0; Jc < NT; ++Jc) {
                acca[Jc] = __builtin_amdgcn_mfma_f64_16x16x4f64(afa[s], pc[s & 1][Jc], acca[Jc], 0, 0, 0);
                accb[Jc] = __builtin_amdgcn_mfma_f64_16x16x4f64(afb[s], pc[s & 1][Jc], accb[Jc], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
#pragma unroll
        for (int Jc = 0; Jc < NT; ++Jc)
#pragma unroll
            for (int r = 0; r < 4; ++r) ablk_a[Jc * 256 + tsw(lk + 4 * r, li)] = acca[Jc][r];
        if (hasb) {
#pragma unroll
            for (int Jc = 0; Jc < NT; ++Jc)
#pragma unroll
                for (int r = 0; r < 4; ++r) ablk_b[Jc * 256 + tsw(lk + 4 * r, li)] = accb[Jc][r];
        }
    }
    __syncthreads();
    if constexpr (SIMPLE) {       // P = (AP + AP^T) / 2
        for (int t = wave; t < NT * NT; t += 4) {
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
    // ---- o(I, J) += sum_k AP(I, k) A(J, k)^T for the seven tiles of this wavefront ---------------------------------------
    // (one copy of the loop per number of tiles in column Jb: which of the two row blocks of A a tile multiplies with
    // is then fixed in the instruction instead of selected per operand)
    {
        const double* ap0[NT];
        const double* ap1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { ap0[t] = lds + rowblk(t) * NT * 256 + col0; ap1[t] = lds + rowblk(t) * NT * 256 + col1; }
        auto apat = [&](auto nsec_) {
            constexpr int NSEC = decltype(nsec_)::value;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const double2 a2 = *reinterpret_cast<const double2*>((h ? ap1[t] : ap0[t]) + kt * 256);
                        const bool s2 = t >= NT - NSEC;
                        o[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, s2 ? afb[4 * kt + 2 * h] : afa[4 * kt + 2 * h], o[t], 0, 0, 0);
                        o[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, s2 ? afb[4 * kt + 2 * h + 1] : afa[4 * kt + 2 * h + 1], o[t], 0, 0, 0);
                    }
                }
            }
        };
        if (wave == 0) apat(I0{});
        else if (wave == 1) apat(I1{});
        else if (wave == 2) apat(I2{});
        else apat(std::integral_constant<int, 3>{});
    }
    JOSEPH_STAMP(6);
    // Every tile goes out twice: as it is (rows of 128 bytes), and mirrored; the mirror image is read back transposed
    // from a staging tile (the chunk buffers are free) so that its stores are rows of 128 bytes as well.  Buffer stores:
    // the tile's offset in a scalar register, the lane's offset the same for both images.  On the two diagonal tiles
    // (the first of each column) the lower triangle goes out as it is and mirrored strictly above the diagonal.
    typedef int v2s_t __attribute__((ext_vector_type(2)));
    auto buffer_st = [](double v, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
        v2s_t d;
        d.x = __double2loint(v); d.y = __double2hiint(v);
        __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, voff, soff, 0);
    };
    double* sc = kst + wave * 256;
    double* scw[4];
    const double* scr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { scw[r] = sc + tsw(lk + 4 * r, li); scr[r] = sc + tsw(li, lk + 4 * r); }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int I = rowblk(t), J = sec(t) ? Jb : Ja;
        const int so = (16 * I * ldn + 16 * J) * 8, som = (16 * J * ldn + 16 * I) * 8;
#pragma unroll
        for (int r = 0; r < 4; ++r) *scw[r] = o[t][r];
        if (I != J) {
#pragma unroll
            for (int r = 0; r < 4; ++r) buffer_st(o[t][r], prsrc, pvo[r], so);
#pragma unroll
            for (int r = 0; r < 4; ++r) buffer_st(*scr[r], prsrc, pvo[r], som);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (lk + 4 * r >= li) buffer_st(o[t][r], prsrc, pvo[r], so);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (li > lk + 4 * r) buffer_st(*scr[r], prsrc, pvo[r], som);
        }
    }
    JOSEPH_STAMP(7);
}

}  // namespace slamk

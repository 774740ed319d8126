// libslamk.so -- C ABI (include/slamk.h) over the HIP kernels of the EKF predict/update path.
// gfx950 only.
#include "../../include/slamk.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "feature_kernels.hip.h"
#include "gemm.hip.h"
#include "gain_solver.hip.h"
#include "linalg_kernels.hip.h"
#include "structured_kernels.hip.h"
#include "joseph_kernel.hip.h"
#include "joseph_quad.hip.h"
#include "seq_update.hip.h"
#include "preprocess_kernels.hip.h"
#include "frame_kernels.hip.h"

using namespace slamk;

// Workspace of the entry points that may not allocate (slamk_stage_gain_static_dev): every GPU the library runs
// on holds its own copy of this array, at its own address, until the process ends.  In host code the name below is
// only a stand-in for the symbol: the device address comes from hipGetSymbolAddress (static_scratch()).
constexpr size_t STATIC_SCRATCH_DOUBLES = (size_t)4 << 20;      // 32 MiB
static __device__ __align__(256) double slamk_static_scratch[STATIC_SCRATCH_DOUBLES];

namespace {

inline int pad16(int x) { return (x + 15) / 16 * 16; }

struct HostFilter {            // host mirror of what association needs (ids are never on the hot path)
    int nf = 0;
    std::vector<int32_t> id;
    std::vector<uint8_t> isref;
    int m() const {
        int r = 1;
        for (int i = 0; i < nf; ++i) r += isref[i] ? 6 : 4;
        return r;
    }
};

}  // namespace

struct slamk_handle {
    slamk_params prm;
    int B = 0, F = 0, device = 0;
    int ldn = 0, ldm = 0;
    std::vector<HostFilter> hf;
    FilterView fv{};
    // dense workspaces
    double *Fm = nullptr, *T = nullptr, *Am = nullptr, *AP = nullptr;          // [B][ldn][ldn]
    double *Hm = nullptr;                                                        // [B][ldm][ldn]
    double *W = nullptr, *Y = nullptr, *K = nullptr, *KR = nullptr;              // [B][ldn][ldm]
    double *S = nullptr, *L = nullptr, *Li = nullptr;                            // [B][ldm][ldm]
    double *Lsp = nullptr;                  // [B][SOLVER_SPILL_IMAGE]: tile image of the factor for the gain solver's spill form
    double *dx = nullptr;                                                        // [B][ldn]
    // staging
    double *qbw_d = nullptr, *w_d = nullptr, *acc_d = nullptr, *dt_d = nullptr, *alt_d = nullptr;
    int *applied_d = nullptr;
    double *feat_tmp = nullptr;           // [B][12][F] copy of the per-feature containers while slots move
    // image frames: one pinned staging block and its device mirror per handle (apply_frames)
    void *stage_host = nullptr, *stage_dev = nullptr;
    size_t stage_bytes = 0;
    hipEvent_t stage_copied = nullptr;
    bool stage_in_flight = false;
    std::vector<void*> allocs;
    int have_ws = 0;                      // WS_* groups allocated so far (ensure_workspaces)
    std::string err;
    hipStream_t stream = nullptr;
    int last_nl = 0, last_ml = 0;
    // The stream of the last *_dev call whose work may still be running: the host-pointer entry points (null
    // stream, synchronous) wait for it first, because non-blocking streams are not ordered against the null stream.
    hipStream_t pending_stream = nullptr;
    bool pending = false;
    // slamk_frame has already run the frame begin (old_pos = X, depth clamp) of the coming iteration
    // (per filter; the fused step runs its own frame begin unless every filter has had one)
    std::vector<uint8_t> begun;
    int nbegun = 0;
    void mark_begun(int b0, int cnt) { for (int b = b0; b < b0 + cnt; ++b) if (!begun[b]) { begun[b] = 1; ++nbegun; } }
    void clear_begun() { if (nbegun) { std::fill(begun.begin(), begun.end(), 0); nbegun = 0; } }
    // slamk_step_dev: ranges of the batch on streams of their own
    int groups = 1;
    std::vector<hipStream_t> gstreams;
    std::vector<hipEvent_t> gdone;
    hipEvent_t fork = nullptr;
    // blocked Cholesky of one large S: the trailing update right of the next outer panel runs on a stream of its own
    // beside that panel's factorisation (handles made by slamk_create only)
    bool chol_lookahead = false;
    hipStream_t chol_side = nullptr;
    hipEvent_t chol_panel_done = nullptr, chol_rest_done = nullptr;
    // slamk_step_dev in graph mode: captured launch sequences, keyed by everything baked into the launches
    bool debug = false;                    // materialise dense F, H and L for slamk_debug_get*
    bool graph_mode = false;
    struct GraphKey {      // everything that is baked into a captured launch sequence
        const void* p[8]; const double* P; int nl, ml, nfmax, form, debug, outer_block, inverse_block, frame_begin;
        bool operator==(const GraphKey& o) const { return memcmp(this, &o, sizeof *this) == 0; }
    };
    struct GraphEntry { GraphKey key; hipGraph_t graph; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
    // optional per-stage timing with HIP events on the launch stream (slamk_profile_*)
    unsigned prof_mask = 0;                 // bit c: stage category c is timed
    struct ProfRec { int cat; hipEvent_t a, b; };
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
};

namespace {

int fail(slamk_handle* h, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    return code;
}

#define HIPCHK(h, call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return fail(h, SLAMK_EHIP, "%s: %s", #call, hipGetErrorString(e_));  \
    } while (0)

// Makes the handle's device current for one entry point and restores the caller's device afterwards.
struct DeviceGuard {
    int prev = -1;
    hipError_t err;
    explicit DeviceGuard(int dev)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) err = hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define ENTER(h)                                                                                                   \
    DeviceGuard dg_((h)->device);                                                                                  \
    if (dg_.err != hipSuccess) return fail(h, SLAMK_EHIP, "hipSetDevice(%d): %s", (h)->device, hipGetErrorString(dg_.err))

// Entry of a host-pointer (synchronous, null-stream) entry point: work queued by an earlier *_dev call is complete
// before it touches the handle's buffers.
int wait_pending(slamk_handle* h)
{
    if (h->pending) {
        HIPCHK(h, hipStreamSynchronize(h->pending_stream));
        h->pending = false;
    }
    return 0;
}
#define ENTER_HOST(h)                                                                                              \
    ENTER(h);                                                                                                      \
    { int rc_ = wait_pending(h); if (rc_) return rc_; }
inline void note_stream(slamk_handle* h, void* stream) { h->pending_stream = (hipStream_t)stream; h->pending = true; }

// Device address of slamk_static_scratch on the CURRENT device, looked up once per device.
hipError_t static_scratch(double** out)
{
    constexpr int MAXDEV = 64;
    static double* ptrs[MAXDEV] = {};
    static std::mutex mtx;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAXDEV) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mtx);
    if (!ptrs[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(slamk_static_scratch));
        if (e != hipSuccess) return e;
        ptrs[dev] = static_cast<double*>(p);
    }
    *out = ptrs[dev];
    return hipSuccess;
}

// Stage categories reported by slamk_profile_read.
enum ProfCat {
    PC_FEATURE = 0,        // per-feature and elementwise kernels
    PC_STRUCT_PREDICT = 1, // P F^T, F T + Q with F in compact row form
    PC_STRUCT_GAIN = 2,    // P H^T, H W + R, I - K H with H in compact row form
    PC_GEMM_GAIN = 3,      // Y = W Li^T, K = Y Li (large maps)
    PC_GEMM_COV = 4,       // A P, (A P) A^T, (K R) K^T
    PC_GEMM_CHOL = 5,      // panel / trailing / inverse products of the factorisation
    PC_POTF2 = 6,          // diagonal-block factor + inverse
    PC_SOLVER = 7,         // fused per-filter gain solver (tile Cholesky + triangular solves)
    PC_COUNT = 8
};

struct ProfScope {
    slamk_handle* h; hipStream_t st; hipEvent_t b = nullptr;
    ProfScope(slamk_handle* h_, int cat, hipStream_t st_) : h(h_), st(st_)
    {
        if (!((h->prof_mask >> cat) & 1u)) return;
        hipEvent_t ev[2];
        for (auto& e : ev) {
            if (!h->prof_pool.empty()) { e = h->prof_pool.back(); h->prof_pool.pop_back(); }
            else (void)hipEventCreate(&e);
        }
        (void)hipEventRecord(ev[0], st);
        b = ev[1];
        h->prof_recs.push_back({cat, ev[0], ev[1]});
    }
    ~ProfScope() { if (b) (void)hipEventRecord(b, st); }
};

// Dynamic-LDS limit of a kernel, set once per (kernel, device).
int ensure_dynamic_lds(slamk_handle* h, const void* func, size_t bytes)
{
    static std::vector<std::pair<const void*, int>> done;
    static std::mutex mtx;                       // handles of several host threads share the table
    std::lock_guard<std::mutex> lock(mtx);
    int dev = 0;
    HIPCHK(h, hipGetDevice(&dev));
    for (const auto& d : done)
        if (d.first == func && d.second == dev) return 0;
    HIPCHK(h, hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done.emplace_back(func, dev);
    return 0;
}

template <typename T>
int dalloc(slamk_handle* h, T** p, size_t count)
{
    void* q = nullptr;
    HIPCHK(h, hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
    HIPCHK(h, hipMemset(q, 0, std::max<size_t>(count, 1) * sizeof(T)));
    h->allocs.push_back(q);
    *p = static_cast<T*>(q);
    return 0;
}

int max_nl(const slamk_handle* h)
{
    int nf = 0;
    for (const auto& f : h->hf) nf = std::max(nf, f.nf);
    return pad16(9 + 3 * nf);
}
int max_ml(const slamk_handle* h)
{
    int m = 1;
    for (const auto& f : h->hf) m = std::max(m, f.m());
    return pad16(m);
}

int max_m(const slamk_handle* h)
{
    int m = 1;
    for (const auto& f : h->hf) m = std::max(m, f.m());
    return m;
}

constexpr int SOLVER_MAX_MT = 15, SOLVER_MAX_NT = 8;
// 9 to 12 row blocks of 16 states: the solver's wide form (panels of 12 tiles, factorisation only; K by
// gain_rows_kernel from the image of the factor); it keeps the factor in the ring form, whose LDS holds up to 15 tile
// columns with the wider panels
// (more than 128 states are at least 40 features, i.e. at least 161 measurement rows: 10 tile columns)
constexpr int SOLVER_WIDE_NT = 12, SOLVER_WIDE_MIN_MT = 10;
// more than SOLVER_LDS_MT tile columns: S does not fit LDS and the solver keeps an image of the factor in global memory
constexpr int SOLVER_LDS_MT = 10;
// doubles per filter: the tiles, and one more for the factor's border row (wide form)
constexpr size_t SOLVER_SPILL_IMAGE = ((size_t)SOLVER_MAX_MT * (SOLVER_MAX_MT + 1) / 2 + 1) * 256;

// The altitude row is the last measurement row.  When the feature rows of the largest filter fill their 16-row
// tiles exactly (m - 1 a multiple of 16), the per-filter kernels cover only those tiles and take the altitude row
// as a border (one row / column on the vector pipeline) instead of a tile row of its own.
bool alt_is_border(int m) { return m > 1 && (m - 1) % 16 == 0; }
// measurement rows covered by 16-row tiles in the per-filter kernels
int tiled_rows(int m) { return alt_is_border(m) ? m - 1 : pad16(m); }

// measurement rows (padded) up to which eye_minus_kh_rows_kernel holds its 16 rows of K in LDS (64 KB)
constexpr int KH_ONE_PASS_MAX_ML = 512;
bool solver_is_wide(int nl) { return nl > 16 * SOLVER_MAX_NT; }
bool use_fused_solver(int nl, int m)
{
    if (tiled_rows(m) > 16 * SOLVER_MAX_MT) return false;
    return nl <= 16 * SOLVER_MAX_NT || (nl <= 16 * SOLVER_WIDE_NT && tiled_rows(m) >= 16 * SOLVER_WIDE_MIN_MT);
}
// the fused covariance update only needs P in LDS (and at most 256 tiled measurement rows for its copy of diag R)
bool use_fused_joseph(int nl, int m) { return nl <= 16 * SOLVER_MAX_NT && tiled_rows(m) <= 256; }

// update_form = 2: block-sequential update for filters whose covariance (lower tiles) fits LDS
constexpr int SEQ_MAX_NT = 10;
bool use_seq_update(const slamk_handle* h, int nl) { return h->prm.update_form == 2 && nl <= 16 * SEQ_MAX_NT; }

// Workspaces beyond P, T and K.  The per-filter fused kernels keep every intermediate in LDS, so a batch of small
// filters never touches these: they are allocated when a layout or the debug switch first needs them.
enum { WS_SPILL = 8,    // Lsp
       WS_GAIN = 1,     // W, Y, KR, S, L, Li: gain by blocked Cholesky + GEMMs
       WS_COV = 2,      // Am, AP: covariance update by GEMMs
       WS_DENSE = 4 };  // Fm, Hm (and L): dense Jacobians and the factor for slamk_debug_get*

int ensure_workspaces(slamk_handle* h)
{
    const int nl = max_nl(h), mm = max_m(h);
    int need = 0;
    const bool seq = use_seq_update(h, nl);    // the block-sequential update keeps everything in LDS
    if (!seq && !use_fused_solver(nl, mm)) need |= WS_GAIN;
    if (!seq && use_fused_solver(nl, mm) && (tiled_rows(mm) > 16 * SOLVER_LDS_MT || solver_is_wide(nl))) need |= WS_SPILL;
    if (!seq && !use_fused_joseph(nl, mm)) need |= WS_COV;
    if (h->debug) need |= WS_DENSE;
    need &= ~h->have_ws;
    if (!need) return 0;
    const size_t B = h->B, ldn = h->ldn, ldm = h->ldm;
    int rc = 0;
    if ((need & (WS_GAIN | WS_DENSE)) && !h->L) rc |= dalloc(h, &h->L, B * ldm * ldm);
    if (need & WS_GAIN) {
        rc |= dalloc(h, &h->W, B * ldn * ldm); rc |= dalloc(h, &h->Y, B * ldn * ldm);
        if (!h->KR) rc |= dalloc(h, &h->KR, B * ldn * ldm);
        rc |= dalloc(h, &h->S, B * ldm * ldm); rc |= dalloc(h, &h->Li, B * ldm * ldm);
    }
    if (need & WS_COV) { rc |= dalloc(h, &h->Am, B * ldn * ldn); rc |= dalloc(h, &h->AP, B * ldn * ldn); }
    if ((need & WS_COV) && !h->KR) rc |= dalloc(h, &h->KR, B * ldn * ldm);     // K R for the covariance update by GEMMs
    if (need & WS_SPILL) rc |= dalloc(h, &h->Lsp, B * SOLVER_SPILL_IMAGE);
    if (need & WS_DENSE) { rc |= dalloc(h, &h->Fm, B * ldn * ldn); rc |= dalloc(h, &h->Hm, B * ldm * ldn); }
    if (rc) return SLAMK_EHIP;
    HIPCHK(h, hipDeviceSynchronize());          // the zero fill is complete before any stream uses the buffers
    h->have_ws |= need;
    return 0;
}

GemmArgs gargs(const double* A, int lda, long long sA, const double* Bp, int ldb, long long sB, double* C, int ldc,
               long long sC, int M, int N, int K, double alpha = 1.0, double beta = 0.0, const double* D = nullptr,
               long long sD = 0)
{
    GemmArgs g{};
    g.A = A; g.B = Bp; g.C = C; g.C2 = nullptr; g.D = D;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.sA1 = sA; g.sB1 = sB; g.sC1 = sC; g.sD1 = sD;
    g.sA2 = g.sB2 = g.sC2 = 0;
    g.nb2 = 1;
    g.alpha = alpha; g.beta = beta;
    return g;
}

// Uploads nf/m/isref/rowoff of filters [b0, b0 + cnt), which all share the host mirror of filter b0.
int push_layout(slamk_handle* h, int b0, int cnt)
{
    const HostFilter& f = h->hf[b0];
    const int F = h->F;
    std::vector<int> isr((size_t)cnt * F, 0), ro((size_t)cnt * F, 0);
    int row = 0;                               // feature rows first, the altitude row last (row m - 1)
    for (int i = 0; i < f.nf; ++i) {
        isr[i] = f.isref[i];
        ro[i] = row;
        row += f.isref[i] ? 6 : 4;
    }
    row += 1;
    for (int c = 1; c < cnt; ++c) {
        std::copy(isr.begin(), isr.begin() + F, isr.begin() + (size_t)c * F);
        std::copy(ro.begin(), ro.begin() + F, ro.begin() + (size_t)c * F);
    }
    std::vector<int> nfv(cnt, f.nf), mv(cnt, row);
    HIPCHK(h, hipMemcpy(h->fv.nf + b0, nfv.data(), cnt * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->fv.m + b0, mv.data(), cnt * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->fv.isref + (size_t)b0 * F, isr.data(), isr.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->fv.rowoff + (size_t)b0 * F, ro.data(), ro.size() * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

__global__ void init_cov_kernel(double* P, int ld, long long stride, double pinit)
{
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < 9) P[(long long)b * stride + (long long)t * ld + t] = (t < 3 || t >= 6) ? pinit * 1.0 : 1.0;
}

// ------------------------------------------------------------------------------------------------
// covariance pipelines (all launches on one stream, no host synchronisation)
// ------------------------------------------------------------------------------------------------
// dense F and H are only materialised for introspection (slamk_debug_get*), never for the products
bool keep_dense(const slamk_handle* h, int nl) { (void)nl; return h->debug; }

constexpr int STRUCT_RPW = 8;        // rows per wavefront in the structured product kernels

void launch_right(const RowSparse& M, const double* In, int ldi, long long sIn, double* Out, int ldo, long long sOut,
                  int rows, int cols, int B, hipStream_t st)
{
    hipLaunchKernelGGL(rowsparse_right_kernel<STRUCT_RPW>,
                       dim3((cols + 63) / 64, (rows + 4 * STRUCT_RPW - 1) / (4 * STRUCT_RPW), B), dim3(256), 0, st, M, In,
                       ldi, sIn, Out, ldo, sOut, rows, cols);
}

template <bool LOWER>
void launch_left(const RowSparse& M, const double* In, int ldi, long long sIn, double* Out, int ldo, long long sOut,
                 const double* diag, long long sDiag, int rows, int cols, int B, hipStream_t st)
{
    hipLaunchKernelGGL(rowsparse_left_kernel<LOWER>,
                       dim3((cols + 63) / 64, (rows + 4 * STRUCT_RPW - 1) / (4 * STRUCT_RPW), B), dim3(256), 0, st, M, In,
                       ldi, sIn, Out, ldo, sOut, diag, sDiag, rows, cols, STRUCT_RPW, 0);
}

// A contiguous range of the filters of a handle: the device view and the workspaces, offset to its first filter.
struct Range {
    FilterView fv;
    int B;
    double *T, *W, *K, *KR, *Y, *S, *L, *Li, *Lsp, *Am, *AP, *Hm, *Fm, *dx;
};

Range make_range(const slamk_handle* h, int b0, int cnt)
{
    Range r;
    const size_t F = h->F, ldn = h->ldn, ldm = h->ldm, o = b0;
    r.B = cnt;
    r.fv = h->fv;
    FilterView& v = r.fv;
    v.B = cnt;
    v.X += 3 * o; v.V += 3 * o; v.bias += 3 * o; v.old_pos += 3 * o;
    v.nf += o; v.m += o; v.status += o;
    v.pib += 3 * F * o; v.anchor += 3 * F * o; v.q0 += 4 * F * o; v.pib0 += 2 * F * o;
    v.isref += F * o; v.rowoff += F * o; v.msrc += 2 * F * o; v.mrefl += 2 * F * o;
    v.P += ldn * ldn * o;
    v.ha += 3 * ldm * o; v.hb += 3 * ldm * o; v.hcs += ldm * o;
    v.e += ldm * o; v.rdiag += ldm * o; v.qdiag += ldn * o;
    v.fa += 3 * ldn * o; v.fg += 3 * ldn * o; v.fgs += ldn * o;
    auto at = [](double* p, size_t off) { return p ? p + off : nullptr; };      // workspaces may not exist yet
    r.T = h->T + ldn * ldn * o; r.Am = at(h->Am, ldn * ldn * o); r.AP = at(h->AP, ldn * ldn * o);
    r.Fm = at(h->Fm, ldn * ldn * o); r.Hm = at(h->Hm, ldm * ldn * o);
    r.W = at(h->W, ldn * ldm * o); r.K = h->K + ldn * ldm * o; r.KR = at(h->KR, ldn * ldm * o); r.Y = at(h->Y, ldn * ldm * o);
    r.S = at(h->S, ldm * ldm * o); r.L = at(h->L, ldm * ldm * o); r.Li = at(h->Li, ldm * ldm * o);
    r.Lsp = at(h->Lsp, SOLVER_SPILL_IMAGE * o);
    r.dx = h->dx + ldn * o;
    return r;
}

int run_predict(slamk_handle* h, const Range& rg, const double* qbw, const double* w, const double* acc,
                const double* dt, hipStream_t st, bool frame = false)
{
    // frame: the iteration starts an image frame (old_pos = X, depth clamp) before anything else
    // Reads rg.fv.P, leaves the propagated covariance in rg.T: the caller swaps the two buffers of the handle.
    const int B = rg.B, nl = max_nl(h);
    h->last_nl = nl;
    {
        ProfScope ps(h, PC_FEATURE, st);
        if (keep_dense(h, nl)) {         // the dense F is built between the rows of F and the dynamics
            if (frame) hipLaunchKernelGGL(frame_begin_kernel, dim3(B), dim3(256), 0, st, rg.fv, -1);
            hipLaunchKernelGGL(motion_rows_kernel, dim3((nl + 255) / 256, B), dim3(256), 0, st, rg.fv, qbw, w, dt, nl);
            hipLaunchKernelGGL(build_F_kernel, dim3((nl + 63) / 64, (nl + 3) / 4, B), dim3(256), 0, st, rg.fv, qbw, w,
                               dt, rg.Fm, nl);
            hipLaunchKernelGGL(dynamics_kernel, dim3(B), dim3(256), 0, st, rg.fv, qbw, w, acc, dt);
        } else
            hipLaunchKernelGGL(pre_predict_kernel, dim3(B), dim3(256), 0, st, rg.fv, qbw, w, acc, dt, nl, frame ? 1 : 0);
    }
    ProfScope ps(h, PC_STRUCT_PREDICT, st);
    // P <- F P F^T + Q      (calcP, src/slam/main.cpp:526-530, G = I)
    const int nt = (nl + 2) / 3;
    if (nt <= 128) {           // moderate sizes: bands of row triples staged through LDS, all global accesses contiguous
        const int nthr = 256, hband = nthr / nt;     // 128 and 512 threads measured slower (0.32, 0.30 vs 0.28 ms)
        hipLaunchKernelGGL(predict_cov_band_kernel, dim3(((nt + hband - 1) / hband) * 8 * ((B + 7) / 8)), dim3(nthr),
                           predict_band_lds_bytes(nl, hband), st, rg.fv, rg.fv.P, rg.T, nl, hband);
    } else
        hipLaunchKernelGGL(predict_cov_kernel, dim3((nt * nt + 255) / 256, B), dim3(256), 0, st, rg.fv, rg.fv.P, rg.T, nl);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// The range with the roles of P and T exchanged: what the update sees after run_predict on `rg`.
Range after_predict(const Range& rg)
{
    Range r = rg;
    std::swap(r.fv.P, r.T);
    return r;
}

constexpr int CHOL_NB = 128;       // diagonal block of the blocked Cholesky (one workgroup factors and inverts it)
// Block sizes of the large-matrix path, process-wide (slamk_set_block_sizes; SLAMK_CHOL_OB / SLAMK_INV_BS give the
// initial values):
//   outer panel width of the two-level factorisation, a multiple of CHOL_NB;
//   rows of the diagonal blocks of L whose inverses are formed, a power of two times CHOL_NB.
int round_outer_block(int v) { return std::max(v, CHOL_NB) / CHOL_NB * CHOL_NB; }
int round_inverse_block(int want)
{
    int v = CHOL_NB;
    while (v * 2 <= want) v *= 2;
    return v;
}
int& chol_outer_block_ref()
{
    static int ob = [] { const char* e = getenv("SLAMK_CHOL_OB"); return round_outer_block(e ? atoi(e) : 512); }();
    return ob;
}
int& inv_block_rows_ref()
{
    static int bs = [] { const char* e = getenv("SLAMK_INV_BS"); return round_inverse_block(e ? atoi(e) : 4096); }();
    return bs;
}
int chol_outer_block() { return chol_outer_block_ref(); }
#define INV_BS inv_block_rows_ref()
// Where the symmetric matrix being factored lives: element (r, c) at p + (r - r0) * ld + (c - c0) (+ stride per matrix of
// the batch).  A square matrix has r0 = c0 = 0; a panel buffer that holds the rows from r0 down of columns c0 .. has
// ld = the panel's width.
struct SymView {
    double* p; int ld; long long stride; int r0, c0;
    double* at(int r, int c) const { return p + (long long)(r - r0) * ld + (c - c0); }
};

template <int MT>
int launch_chol_inv_block_t(slamk_handle* h, const SymView& S, double* Lp, double* Lip, int ldm, long long sM, int B, int j0,
                            int* status, hipStream_t st)
{
    const size_t lds = ((size_t)MT * (MT + 1) / 2 + 1) * 256 * sizeof(double);
    { int rc_ = ensure_dynamic_lds(h, reinterpret_cast<const void*>(chol_inv_block_kernel<MT>), lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(chol_inv_block_kernel<MT>, dim3(B), dim3(512), lds, st, S.at(j0, j0), S.ld, S.stride, Lp, Lip, ldm, sM,
                       j0, status);
    return 0;
}

// diagonal block (nb x nb at (j0, j0), nb a multiple of 16 up to 128) of S -> L and inv(L)
int launch_chol_inv_block(slamk_handle* h, const SymView& S, double* Lp, double* Lip, int ldm, long long sM, int B, int j0,
                          int nb, int* status, hipStream_t st)
{
    switch (nb / 16) {
        case 1: return launch_chol_inv_block_t<1>(h, S, Lp, Lip, ldm, sM, B, j0, status, st);
        case 2: return launch_chol_inv_block_t<2>(h, S, Lp, Lip, ldm, sM, B, j0, status, st);
        case 3: return launch_chol_inv_block_t<3>(h, S, Lp, Lip, ldm, sM, B, j0, status, st);
        case 4: return launch_chol_inv_block_t<4>(h, S, Lp, Lip, ldm, sM, B, j0, status, st);
        case 5: return launch_chol_inv_block_t<5>(h, S, Lp, Lip, ldm, sM, B, j0, status, st);
        case 6: return launch_chol_inv_block_t<6>(h, S, Lp, Lip, ldm, sM, B, j0, status, st);
        case 7: return launch_chol_inv_block_t<7>(h, S, Lp, Lip, ldm, sM, B, j0, status, st);
        case 8: return launch_chol_inv_block_t<8>(h, S, Lp, Lip, ldm, sM, B, j0, status, st);
        default: return fail(h, SLAMK_EINVAL, "Cholesky diagonal block of %d rows", nb);
    }
}

// Columns [J0, J1) of the blocked right-looking Cholesky of S (lower) into L, diagonal-block inverses into Li: the
// trailing update after every diagonal block of NB rows only reaches the panel's own columns (rank NB).  On entry
// these columns of S carry the updates of all columns left of J0.
int chol_outer_panel(slamk_handle* h, const SymView& S, double* Lp, double* Lip, int ldm, long long sM, int B, int ml, int J0,
                     int J1, int* status, hipStream_t st)
{
    const int NB = CHOL_NB;
    for (int j0 = J0; j0 < J1; j0 += NB) {
        const int nb = std::min(NB, ml - j0);
        {
            ProfScope ps(h, PC_POTF2, st);
            int rc = launch_chol_inv_block(h, S, Lp, Lip, ldm, sM, B, j0, nb, status, st);
            if (rc) return rc;
        }
        const int rem = ml - j0 - nb;
        if (rem <= 0) break;
        ProfScope ps(h, PC_GEMM_CHOL, st);
        const long long o21 = (long long)(j0 + nb) * ldm + j0, o11 = (long long)j0 * ldm + j0;
        // L21 = S21 inv(L11)^T
        HIPCHK(h, (launch_gemm<true, EPI_NONE, false, TRI_B_KLE_N>(
                      gargs(S.at(j0 + nb, j0), S.ld, S.stride, Lip + o11, ldm, sM, Lp + o21, ldm, sM, rem, nb, nb), B, st)));
        // S22 -= L21 L21^T, lower tiles, columns of this outer panel only
        const int nin = J1 - (j0 + nb);
        if (nin > 0)
            HIPCHK(h, (launch_gemm<true, EPI_NONE, true, TRI_NONE>(
                          gargs(Lp + o21, ldm, sM, Lp + o21, ldm, sM, S.at(j0 + nb, j0 + nb), S.ld, S.stride, rem, nin, nb, -1.0,
                                1.0), B, st)));
    }
    return 0;
}

// Li = the inverses of L's diagonal blocks of INV_BS rows (all of inv(L) when ml <= INV_BS) from the inverses of its
// diagonal NB-blocks, by recursive doubling on the block diagonal:
//   inv([[A,0],[Bk,C]]) = [[inv A, 0], [-inv C * Bk * inv A, inv C]]
// The intermediate Bk * inv A of the pair whose lower block starts at row q is held in X at row q: in the same columns as
// Bk when X is a square matrix like L (`compact` false: S, whose lower part is dead after the factorisation, serves),
// from column 0 when X is a buffer of ml rows and INV_BS / 2 columns (`compact`).
// The doubling stops at blocks of INV_BS rows: gain_from_factor runs the two triangular solves block-wise with
// these diagonal-block inverses, so the levels above (most of the cost of a full inverse) are never built.
int chol_block_inverses(slamk_handle* h, double* Xp, int ldx, long long sX, bool compact, double* Lp, double* Lip, int ldm,
                        long long sM, int B, int ml, hipStream_t st)
{
    ProfScope ps_inv(h, PC_GEMM_CHOL, st);
    for (int s = CHOL_NB; s < std::min(ml, INV_BS); s *= 2) {
        const int nfull = ml / (2 * s), r = ml - nfull * 2 * s;
        for (int pass = 0; pass < 2; ++pass) {
            int cnt, M, base;
            if (pass == 0) { cnt = nfull; M = s; base = 0; }
            else { cnt = (r > s) ? 1 : 0; M = r - s; base = nfull * 2 * s; }
            if (cnt == 0) continue;
            const long long oA = (long long)base * ldm + base;               // top-left block (s x s)
            const long long oB = (long long)(base + s) * ldm + base;         // bottom-left (M x s)
            const long long oC = (long long)(base + s) * ldm + (base + s);   // bottom-right (M x M)
            const long long oX = (long long)(base + s) * ldx + (compact ? 0 : base);
            const long long s2 = (long long)2 * s * (ldm + 1), s2x = (long long)2 * s * ldx + (compact ? 0 : 2 * s);
            GemmArgs g1 = gargs(Lp + oB, ldm, sM, Lip + oA, ldm, sM, Xp + oX, ldx, sX, M, s, s);
            g1.sA2 = g1.sB2 = s2; g1.sC2 = s2x; g1.nb2 = cnt;
            HIPCHK(h, (launch_gemm<false, EPI_NONE, false, TRI_B_KGE_N>(g1, B * cnt, st)));
            GemmArgs g2 = gargs(Lip + oC, ldm, sM, Xp + oX, ldx, sX, Lip + oB, ldm, sM, M, s, M, -1.0, 0.0);
            g2.sA2 = g2.sC2 = s2; g2.sB2 = s2x; g2.nb2 = cnt;
            HIPCHK(h, (launch_gemm<false, EPI_NONE, false, TRI_A_KLE_M>(g2, B * cnt, st)));
        }
    }
    return 0;
}

// columns of the compact scratch of chol_block_inverses for a matrix of ml rows: the widest lower-left block it forms
int inverse_scratch_cols(int ml)
{
    int w = CHOL_NB;
    for (int s = CHOL_NB; s < std::min(ml, INV_BS); s *= 2) w = s;
    return w;
}

// S (lower) -> L (Cholesky factor) and Li = the inverses of L's diagonal blocks of INV_BS rows (all of inv(L) when
// ml <= INV_BS); S is used as scratch afterwards.
int cholesky_inverse(slamk_handle* h, double* Sp, double* Lp, double* Lip, int ldm, long long sM, int B, int ml,
                     int* status, hipStream_t st)
{
    const int OB = chol_outer_block();
    // Blocked right-looking Cholesky on two levels: inside an outer panel of OB columns the updates are of rank
    // CHOL_NB and stay inside the panel; the columns right of the panel take one rank-OB update when the panel is
    // complete, which reads and writes the large trailing matrix OB / CHOL_NB times less often.
    // One large matrix (SLAMK_CHOL_LOOKAHEAD=0 turns it off): that update is cut in two.  The columns of the NEXT panel
    // are updated on the launch stream and the panel's factorisation -- a chain of one-workgroup kernels and narrow
    // products -- follows at once; the columns right of it are updated on a second stream beside that chain.  The two
    // parts write disjoint columns; the part on the launch stream for panel p + 1 waits for the second stream's part of
    // panel p, which wrote those columns before.  Same products and sums per element as the single update.
    static const bool lookahead_env = [] { const char* e = getenv("SLAMK_CHOL_LOOKAHEAD"); return !e || atoi(e) != 0; }();
    bool ahead = lookahead_env && h->chol_lookahead && B == 1 && ml > 3 * OB;
    if (ahead) {                                         // not inside a stream capture (graph mode)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) ahead = false;
    }
    if (ahead && !h->chol_side) {
        // the second stream leaves a few CUs alone, so that a one-workgroup kernel of the chain on the launch stream
        // finds a whole CU's LDS free while the wide product runs (otherwise it waits for that product to drain)
        hipDeviceProp_t prop;
        HIPCHK(h, hipGetDeviceProperties(&prop, h->device));
        const int ncu = prop.multiProcessorCount;
        std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
        static const int spare = [] { const char* e = getenv("SLAMK_CHOL_SPARE_CUS"); return e ? atoi(e) : 32; }();
        for (int c = std::max(0, std::min(spare, ncu - 1)); c < ncu; ++c) mask[c / 32] |= 1u << (c % 32);
        // (a device or partition mode that refuses the mask: the single update on the launch stream, for good)
        if (spare < 0 || hipExtStreamCreateWithCUMask(&h->chol_side, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            h->chol_side = nullptr;
            h->chol_lookahead = false;
            ahead = false;
        } else {
            HIPCHK(h, hipEventCreateWithFlags(&h->chol_panel_done, hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->chol_rest_done, hipEventDisableTiming));
        }
    }
    bool rest_pending = false;                           // the second stream holds an update the launch stream has not joined
    for (int J0 = 0; J0 < ml; J0 += OB) {
        const int J1 = std::min(ml, J0 + OB);
        int rc = chol_outer_panel(h, SymView{Sp, ldm, sM, 0, 0}, Lp, Lip, ldm, sM, B, ml, J0, J1, status, st);
        if (rc) return rc;
        const int remo = ml - J1;
        if (remo <= 0) break;
        ProfScope ps(h, PC_GEMM_CHOL, st);
        const long long oL = (long long)J1 * ldm + J0, oS = (long long)J1 * ldm + J1;
        const int J2 = std::min(ml, J1 + OB);
        if (ahead && ml - J2 > 0) {
            // columns [J2, ml): rows from J2 down, on the second stream
            HIPCHK(h, hipEventRecord(h->chol_panel_done, st));
            HIPCHK(h, hipStreamWaitEvent(h->chol_side, h->chol_panel_done, 0));
            const long long oL2 = (long long)J2 * ldm + J0, oS2 = (long long)J2 * ldm + J2;
            HIPCHK(h, (launch_gemm<true, EPI_NONE, true, TRI_NONE>(
                          gargs(Lp + oL2, ldm, sM, Lp + oL2, ldm, sM, Sp + oS2, ldm, sM, ml - J2, ml - J2, J1 - J0, -1.0, 1.0),
                          B, h->chol_side)));
            // columns [J1, J2) of the next panel: after the previous panel's update of the same columns
            if (rest_pending) HIPCHK(h, hipStreamWaitEvent(st, h->chol_rest_done, 0));
            HIPCHK(h, hipEventRecord(h->chol_rest_done, h->chol_side));
            rest_pending = true;
            HIPCHK(h, (launch_gemm<true, EPI_NONE, true, TRI_NONE>(
                          gargs(Lp + oL, ldm, sM, Lp + oL, ldm, sM, Sp + oS, ldm, sM, remo, J2 - J1, J1 - J0, -1.0, 1.0),
                          B, st)));
        } else {
            if (rest_pending) { HIPCHK(h, hipStreamWaitEvent(st, h->chol_rest_done, 0)); rest_pending = false; }
            // the rest of the trailing matrix: one update of rank J1 - J0
            HIPCHK(h, (launch_gemm<true, EPI_NONE, true, TRI_NONE>(
                          gargs(Lp + oL, ldm, sM, Lp + oL, ldm, sM, Sp + oS, ldm, sM, remo, remo, J1 - J0, -1.0, 1.0),
                          B, st)));
        }
    }
    if (rest_pending) HIPCHK(h, hipStreamWaitEvent(st, h->chol_rest_done, 0));
    return chol_block_inverses(h, Sp, ldm, sM, false, Lp, Lip, ldm, sM, B, ml, st);
}

// K = W S^-1 = (W L^-T) L^-1 and K R from the factor L and the inverses Li of its diagonal INV_BS-blocks, as two
// block triangular solves (calcK, src/slam/main.cpp:547-550 solved by Cholesky instead of LU):
//     Y_j = (W_j - sum_{k<j} Y_k L_jk^T) inv(L_jj)^T,  j ascending;   K_j = (Y_j - sum_{k>j} K_k L_kj) inv(L_jj),  j descending.
// W (rows x ml) is overwritten; every product is a GEMM with at least min(ml, INV_BS) columns.
// ws: workspace for cutting the k range of the off-diagonal products into slices (one matrix with few rows, i.e. a row
// block of a sharded filter); null: never cut.
int gain_from_factor(slamk_handle* h, double* W, double* Y, double* K, double* KR, int ldw, long long sW,
                     const double* L, const double* Li, int ldl, long long sL, const double* rdiag, long long sR,
                     int rows, int ml, int B, hipStream_t st, double* ws = nullptr, size_t ws_elems = 0)
{
    if (B != 1) ws = nullptr;
    const int nbk = (ml + INV_BS - 1) / INV_BS;
    for (int j = 0; j < nbk; ++j) {
        const int c0 = j * INV_BS, wj = std::min(INV_BS, ml - c0);
        const long long odiag = (long long)c0 * ldl + c0;
        if (j > 0 && ws)
            HIPCHK(h, launch_gemm_splitk<true>(gargs(Y, ldw, sW, L + (long long)c0 * ldl, ldl, sL, W + c0, ldw, sW, rows, wj, c0,
                                                     -1.0, 1.0), ws, ws_elems, st));
        else if (j > 0)
            HIPCHK(h, (launch_gemm<true, EPI_NONE, false, TRI_NONE>(
                          gargs(Y, ldw, sW, L + (long long)c0 * ldl, ldl, sL, W + c0, ldw, sW, rows, wj, c0, -1.0, 1.0), B, st)));
        HIPCHK(h, (launch_gemm<true, EPI_NONE, false, TRI_B_KLE_N>(
                      gargs(W + c0, ldw, sW, Li + odiag, ldl, sL, Y + c0, ldw, sW, rows, wj, wj), B, st)));
    }
    for (int j = nbk - 1; j >= 0; --j) {
        const int c0 = j * INV_BS, wj = std::min(INV_BS, ml - c0), c1 = c0 + wj;
        const long long odiag = (long long)c0 * ldl + c0;
        if (c1 < ml && ws)
            HIPCHK(h, launch_gemm_splitk<false>(gargs(K + c1, ldw, sW, L + (long long)c1 * ldl + c0, ldl, sL, Y + c0, ldw, sW, rows,
                                                      wj, ml - c1, -1.0, 1.0), ws, ws_elems, st));
        else if (c1 < ml)
            HIPCHK(h, (launch_gemm<false, EPI_NONE, false, TRI_NONE>(
                          gargs(K + c1, ldw, sW, L + (long long)c1 * ldl + c0, ldl, sL, Y + c0, ldw, sW, rows, wj, ml - c1, -1.0,
                                1.0), B, st)));
        // K_j and K_j R (R diagonal) from the same accumulators
        GemmArgs g = gargs(Y + c0, ldw, sW, Li + odiag, ldl, sL, K + c0, ldw, sW, rows, wj, wj, 1.0, 0.0, rdiag + c0, sR);
        g.C2 = KR + c0;
        HIPCHK(h, (launch_gemm<false, EPI_DUAL_COLSCALE, false, TRI_B_KGE_N>(g, B, st)));
    }
    return 0;
}

int run_cholesky_inverse(slamk_handle* h, const Range& rg, int ml, hipStream_t st)
{
    return cholesky_inverse(h, rg.S, rg.L, rg.Li, h->ldm, (long long)h->ldm * h->ldm, rg.B, ml, rg.fv.status, st);
}

// KRout: K R beside K for the covariance update by GEMMs (null when the per-filter covariance kernel follows)
template <int MT, bool BORDER, bool WIDE = false>
int launch_gain_solver(slamk_handle* h, const Range& rg, int nl, double* Lout, double* KRout, hipStream_t st)
{
    constexpr bool SPILL = MT > SOLVER_LDS_MT || WIDE;
    if constexpr (WIDE && MT < SOLVER_WIDE_MIN_MT) {
        return fail(h, SLAMK_EINVAL, "gain solver: %d measurement tiles with more than %d states", MT, 16 * SOLVER_MAX_NT);
    } else {
        if (!WIDE && nl > 16 * SOLVER_MAX_NT) return fail(h, SLAMK_EINVAL, "gain solver: %d states need the wide form", nl);
        const size_t lds = (size_t)SolverLds<MT, SPILL, WIDE>::TILES * 256 * sizeof(double);
        { int rc_ = ensure_dynamic_lds(h, reinterpret_cast<const void*>(gain_solver_kernel<MT, BORDER, SPILL, WIDE>), lds); if (rc_) return rc_; }
        if (SPILL && !rg.Lsp) return fail(h, SLAMK_EINVAL, "gain solver: no image buffer for %d measurement tiles", MT);
        const long long sM = (long long)h->ldm * h->ldm, sNM = (long long)h->ldn * h->ldm;
        hipLaunchKernelGGL((gain_solver_kernel<MT, BORDER, SPILL, WIDE>), dim3(rg.B), dim3(512), lds, st, rg.fv, rg.K, KRout,
                           Lout, sM, sNM, nl / 16, rg.Lsp, (long long)SOLVER_SPILL_IMAGE);
        if constexpr (WIDE) {                            // the two solves: row blocks 0..3, then the rest
            const size_t lr = (size_t)gain_rows_lds_tiles(MT) * 256 * sizeof(double);
            int rc_ = ensure_dynamic_lds(h, reinterpret_cast<const void*>(gain_rows_kernel<MT, BORDER>), lr);
            if (rc_) return rc_;
            const int NT = nl / 16;
            auto rows = [&](int blk0, int G) {   // G workgroups per filter, the eight XCDs taking filters in turn
                hipLaunchKernelGGL((gain_rows_kernel<MT, BORDER>), dim3(8 * ((rg.B + 7) / 8) * G), dim3(256), lr, st, rg.fv, rg.K,
                                   KRout, sNM, NT, blk0, G, rg.Lsp, (long long)SOLVER_SPILL_IMAGE, Lout, sM);
            };
            if constexpr (BORDER) {      // the border's l comes from row block 0: its workgroup first, the others behind it
                rows(0, 1);
                rows(4, (NT - 4 + 3) / 4);
            } else                       // one launch: 9 row blocks are three workgroups of three
                rows(0, (NT + 3) / 4);
        }
        HIPCHK(h, hipGetLastError());
        return 0;
    }
}

template <bool BORDER, bool WIDE>
int run_gain_solver_b(slamk_handle* h, const Range& rg, int nl, int mt, double* Lout, double* KRout, hipStream_t st)
{
    switch (mt) {
        case 1: return launch_gain_solver<1, BORDER, WIDE>(h, rg, nl, Lout, KRout, st);
        case 2: return launch_gain_solver<2, BORDER, WIDE>(h, rg, nl, Lout, KRout, st);
        case 3: return launch_gain_solver<3, BORDER, WIDE>(h, rg, nl, Lout, KRout, st);
        case 4: return launch_gain_solver<4, BORDER, WIDE>(h, rg, nl, Lout, KRout, st);
        case 5: return launch_gain_solver<5, BORDER, WIDE>(h, rg, nl, Lout, KRout, st);
        case 6: return launch_gain_solver<6, BORDER, WIDE>(h, rg, nl, Lout, KRout, st);
        case 7: return launch_gain_solver<7, BORDER, WIDE>(h, rg, nl, Lout, KRout, st);
        case 8: return launch_gain_solver<8, BORDER, WIDE>(h, rg, nl, Lout, KRout, st);
        case 9: return launch_gain_solver<9, BORDER, WIDE>(h, rg, nl, Lout, KRout, st);
        case 10: return launch_gain_solver<10, BORDER, WIDE>(h, rg, nl, Lout, KRout, st);
        case 11: return launch_gain_solver<11, BORDER, WIDE>(h, rg, nl, Lout, KRout, st);
        case 12: return launch_gain_solver<12, BORDER, WIDE>(h, rg, nl, Lout, KRout, st);
        case 13: return launch_gain_solver<13, BORDER, WIDE>(h, rg, nl, Lout, KRout, st);
        case 14: return launch_gain_solver<14, BORDER, WIDE>(h, rg, nl, Lout, KRout, st);
        case 15: return launch_gain_solver<15, BORDER, WIDE>(h, rg, nl, Lout, KRout, st);
        default: return fail(h, SLAMK_EINVAL, "gain solver: %d measurement tiles out of range", mt);
    }
}

// mm = measurement rows of the largest filter of the batch
int run_gain_solver(slamk_handle* h, const Range& rg, int nl, int mm, hipStream_t st)
{
    ProfScope ps(h, PC_SOLVER, st);
    double* Lout = h->debug ? rg.L : nullptr;         // the factor is only kept for slamk_debug_get
    double* KRout = use_fused_joseph(nl, mm) ? nullptr : rg.KR;
    const int mt = tiled_rows(mm) / 16;
    if (solver_is_wide(nl))
        return alt_is_border(mm) ? run_gain_solver_b<true, true>(h, rg, nl, mt, Lout, KRout, st)
                                 : run_gain_solver_b<false, true>(h, rg, nl, mt, Lout, KRout, st);
    return alt_is_border(mm) ? run_gain_solver_b<true, false>(h, rg, nl, mt, Lout, KRout, st)
                             : run_gain_solver_b<false, false>(h, rg, nl, mt, Lout, KRout, st);
}

// The gate and the mean update ride in the covariance update kernel when launches are what the step costs (few
// filters); for a large batch the separate kernel is cheaper than the extra work inside the long kernel.
bool fuse_apply(int B) { return B <= 256; }

// mlc = the columns of K walked in 16-column chunks (tiled_rows of the largest filter)
template <int NT>
int launch_joseph(slamk_handle* h, const Range& rg, int ml, int* applied, hipStream_t st)
{
    // up to n = 112 the variant that reads K once (its tables hold 256 measurement rows: n <= 112 has m <= 205)
    constexpr bool MERGED = NT <= 7;
    const size_t lds = (size_t)(NT * NT + (MERGED ? 2 * NT + 10 : NT + 2)) * 256 * sizeof(double);
    auto kern = [] { if constexpr (MERGED) return joseph_merged_kernel<NT>; else return joseph_kernel<NT>; }();
    if (MERGED && ml > 256) return fail(h, SLAMK_EINVAL, "fused covariance update: %d measurement rows for n <= %d", ml, 16 * NT);
    if constexpr (NT == 7) {
        // seven row blocks and 112..240 measurement columns in chunks: the variant with four wavefronts per filter
        if (ml >= QUAD_MIN_COLS && ml <= QUAD_MAX_COLS && ml % 16 == 0) {
            const size_t lq = (size_t)(NT * NT + 3 * NT + 10) * 256 * sizeof(double);
            const bool simple = h->prm.update_form == 1;
            const void* kq = simple ? reinterpret_cast<const void*>(joseph_quad_kernel<true>)
                                    : reinterpret_cast<const void*>(joseph_quad_kernel<false>);
            { int rc_ = ensure_dynamic_lds(h, kq, lq); if (rc_) return rc_; }
            double* dxo = fuse_apply(rg.B) ? (double*)nullptr : rg.dx;
            const int ap = fuse_apply(rg.B) ? 1 : 0;
            if (simple)
                hipLaunchKernelGGL(joseph_quad_kernel<true>, dim3(rg.B), dim3(256), lq, st, rg.fv, rg.K, h->ldm,
                                   (long long)h->ldn * h->ldm, ml, dxo, ap, applied);
            else
                hipLaunchKernelGGL(joseph_quad_kernel<false>, dim3(rg.B), dim3(256), lq, st, rg.fv, rg.K, h->ldm,
                                   (long long)h->ldn * h->ldm, ml, dxo, ap, applied);
            HIPCHK(h, hipGetLastError());
            return 0;
        }
    }
    { int rc_ = ensure_dynamic_lds(h, reinterpret_cast<const void*>(kern), lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(kern, dim3(rg.B), dim3(512), lds, st, rg.fv, rg.K, h->ldm,
                       (long long)h->ldn * h->ldm, ml, h->prm.update_form, fuse_apply(rg.B) ? (double*)nullptr : rg.dx,
                       fuse_apply(rg.B) ? 1 : 0, applied);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int run_joseph(slamk_handle* h, const Range& rg, int nl, int ml, int* applied, hipStream_t st)
{
    ProfScope ps(h, PC_GEMM_COV, st);
    switch (nl / 16) {
        case 1: return launch_joseph<1>(h, rg, ml, applied, st);
        case 2: return launch_joseph<2>(h, rg, ml, applied, st);
        case 3: return launch_joseph<3>(h, rg, ml, applied, st);
        case 4: return launch_joseph<4>(h, rg, ml, applied, st);
        case 5: return launch_joseph<5>(h, rg, ml, applied, st);
        case 6: return launch_joseph<6>(h, rg, ml, applied, st);
        case 7: return launch_joseph<7>(h, rg, ml, applied, st);
        case 8: return launch_joseph<8>(h, rg, ml, applied, st);
        default: return fail(h, SLAMK_EINVAL, "fused covariance update: n = %d out of range", nl);
    }
}

template <int NT>
int launch_seq_update(slamk_handle* h, const Range& rg, int* applied, hipStream_t st)
{
    const size_t lds = (size_t)SeqUpdateLds<NT>::TILES * 256 * sizeof(double);
    { int rc_ = ensure_dynamic_lds(h, reinterpret_cast<const void*>(seq_update_kernel<NT>), lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(seq_update_kernel<NT>, dim3(rg.B), dim3(512), lds, st, rg.fv, fuse_apply(rg.B) ? (double*)nullptr : rg.dx,
                       fuse_apply(rg.B) ? 1 : 0, applied);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// A whole iteration with an image in one launch (seq_step_kernel), update_form 2: for small batches, where a step costs
// launches rather than work, and for filters of 81 states and more at any batch size, where saving the propagation's
// pass over the covariance outweighs a workgroup of 512 threads per filter for the small steps (measured at 4096
// filters: 32 features 1.57 -> 1.45 ms per step, 20 features equal, 8 features 0.31 -> 0.40 ms).
// SLAMK_SEQ_STEP_MAXB overrides the largest batch of smaller filters that takes it (default 256).
bool use_seq_step(const slamk_handle* h)
{
    static const int maxb = [] { const char* e = getenv("SLAMK_SEQ_STEP_MAXB"); return e ? atoi(e) : 256; }();
    const int nl = max_nl(h);
    return use_seq_update(h, nl) && !keep_dense(h, nl) && (h->B <= maxb || nl >= 96);
}

template <int NT>
int launch_seq_step(slamk_handle* h, const Range& rg, const double* qbw, const double* w, const double* acc, const double* dt,
                    const double* alt, const double* src, const double* refl, int ml, bool frame, int* applied, hipStream_t st)
{
    const size_t lds = (size_t)SeqUpdateLds<NT>::TILES * 256 * sizeof(double);
    { int rc_ = ensure_dynamic_lds(h, reinterpret_cast<const void*>(seq_step_kernel<NT>), lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(seq_step_kernel<NT>, dim3(rg.B), dim3(512), lds, st, rg.fv, qbw, w, acc, dt, alt, src, refl, ml,
                       frame ? 1 : 0, fuse_apply(rg.B) ? (double*)nullptr : rg.dx, fuse_apply(rg.B) ? 1 : 0, applied);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int run_seq_step(slamk_handle* h, const Range& rg, const double* qbw, const double* w, const double* acc, const double* dt,
                 const double* alt, const double* src, const double* refl, bool frame, int* applied, hipStream_t st)
{
    const int nl = max_nl(h), ml = max_ml(h);
    h->last_nl = nl; h->last_ml = ml;
    if (!src) src = rg.fv.msrc;
    if (!refl) refl = rg.fv.mrefl;
    int rc = 0;
    {
        ProfScope ps(h, PC_GEMM_COV, st);
        switch (nl / 16) {
            case 1: rc = launch_seq_step<1>(h, rg, qbw, w, acc, dt, alt, src, refl, ml, frame, applied, st); break;
            case 2: rc = launch_seq_step<2>(h, rg, qbw, w, acc, dt, alt, src, refl, ml, frame, applied, st); break;
            case 3: rc = launch_seq_step<3>(h, rg, qbw, w, acc, dt, alt, src, refl, ml, frame, applied, st); break;
            case 4: rc = launch_seq_step<4>(h, rg, qbw, w, acc, dt, alt, src, refl, ml, frame, applied, st); break;
            case 5: rc = launch_seq_step<5>(h, rg, qbw, w, acc, dt, alt, src, refl, ml, frame, applied, st); break;
            case 6: rc = launch_seq_step<6>(h, rg, qbw, w, acc, dt, alt, src, refl, ml, frame, applied, st); break;
            case 7: rc = launch_seq_step<7>(h, rg, qbw, w, acc, dt, alt, src, refl, ml, frame, applied, st); break;
            case 8: rc = launch_seq_step<8>(h, rg, qbw, w, acc, dt, alt, src, refl, ml, frame, applied, st); break;
            case 9: rc = launch_seq_step<9>(h, rg, qbw, w, acc, dt, alt, src, refl, ml, frame, applied, st); break;
            case 10: rc = launch_seq_step<10>(h, rg, qbw, w, acc, dt, alt, src, refl, ml, frame, applied, st); break;
            default: rc = fail(h, SLAMK_EINVAL, "one-launch step: n = %d out of range", nl);
        }
    }
    if (rc) return rc;
    if (!fuse_apply(rg.B)) {
        ProfScope ps_tail(h, PC_FEATURE, st);
        hipLaunchKernelGGL(gain_apply_kernel, dim3(rg.B), dim3(256), 0, st, rg.fv, rg.dx, applied);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

int run_seq_update(slamk_handle* h, const Range& rg, int nl, int* applied, hipStream_t st)
{
    ProfScope ps(h, PC_GEMM_COV, st);
    switch (nl / 16) {
        case 1: return launch_seq_update<1>(h, rg, applied, st);
        case 2: return launch_seq_update<2>(h, rg, applied, st);
        case 3: return launch_seq_update<3>(h, rg, applied, st);
        case 4: return launch_seq_update<4>(h, rg, applied, st);
        case 5: return launch_seq_update<5>(h, rg, applied, st);
        case 6: return launch_seq_update<6>(h, rg, applied, st);
        case 7: return launch_seq_update<7>(h, rg, applied, st);
        case 8: return launch_seq_update<8>(h, rg, applied, st);
        case 9: return launch_seq_update<9>(h, rg, applied, st);
        case 10: return launch_seq_update<10>(h, rg, applied, st);
        default: return fail(h, SLAMK_EINVAL, "block-sequential update: n = %d out of range", nl);
    }
}

int run_update(slamk_handle* h, const Range& rg, const double* qbw, const double* alt, const double* src, const double* refl,
               int* applied, hipStream_t st)
{
    const int B = rg.B, ldn = h->ldn, ldm = h->ldm;
    const int nl = max_nl(h), ml = max_ml(h), mm = max_m(h);
    const long long sN = (long long)ldn * ldn, sNM = (long long)ldn * ldm, sM = (long long)ldm * ldm;
    h->last_nl = nl; h->last_ml = ml;
    int nfmax = 0;
    for (const auto& f : h->hf) nfmax = std::max(nfmax, f.nf);
    if (!src) src = rg.fv.msrc;
    if (!refl) refl = rg.fv.mrefl;

    {
        ProfScope ps(h, PC_FEATURE, st);
        hipLaunchKernelGGL(measure_kernel, dim3(std::max(1, (nfmax + 63) / 64), B), dim3(64), 0, st, rg.fv, qbw, alt,
                           src, refl, ml);
        if (keep_dense(h, nl))
            hipLaunchKernelGGL(build_H_kernel, dim3((nl + 63) / 64, (ml + 3) / 4, B), dim3(256), 0, st, rg.fv, rg.Hm,
                               ml, nl);
    }
    if (h->prm.update_form == 2) {
        // measurement rows 16 at a time against the covariance in LDS: gain, covariance, dx, gate in one kernel
        if (!use_seq_update(h, nl))
            return fail(h, SLAMK_EINVAL, "update_form 2 (block-sequential) needs n <= %d, got %d", 16 * SEQ_MAX_NT, nl);
        int rc = run_seq_update(h, rg, nl, applied, st);
        if (rc) return rc;
        if (!fuse_apply(B)) {
            ProfScope ps_tail(h, PC_FEATURE, st);
            hipLaunchKernelGGL(gain_apply_kernel, dim3(B), dim3(256), 0, st, rg.fv, rg.dx, applied);
        }
        HIPCHK(h, hipGetLastError());
        return 0;
    }
    const RowSparse MH{rg.fv.ha, rg.fv.hb, rg.fv.hcs, ldm, 0};
    // covariance update by GEMMs with rows of K that fit LDS four per wavefront: A = I - K H and dx = K e in one pass
    const bool kh_one_pass = !use_fused_joseph(nl, mm) && ml <= KH_ONE_PASS_MAX_ML;
    {
    ProfScope ps(h, PC_STRUCT_GAIN, st);
    if (!use_fused_solver(nl, mm)) {     // the fused solver forms W and S in LDS itself
        // W = P H^T                                  (calcK, src/slam/main.cpp:545)
        launch_right(MH, rg.fv.P, ldn, sN, rg.W, ldm, sNM, nl, ml, B, st);
        // S = H W + R, lower part                    (:541-543)
        launch_left<true>(MH, rg.W, ldm, sNM, rg.S, ldm, sM, rg.fv.rdiag, ldm, ml, ml, B, st);
    }
    }
    if (use_fused_solver(nl, mm)) {
        // small filters: S = L L^T and K = W L^-T L^-1 (plus K R) in one workgroup per filter
        int rc = run_gain_solver(h, rg, nl, mm, st);
        if (rc) return rc;
    } else {
        // S = L L^T, Li = inv(L)
        int rc = run_cholesky_inverse(h, rg, ml, st);
        if (rc) return rc;
        ProfScope ps(h, PC_GEMM_GAIN, st);
        rc = gain_from_factor(h, rg.W, rg.Y, rg.K, rg.KR, ldm, sNM, rg.L, rg.Li, ldm, sM, rg.fv.rdiag, ldm, nl, ml, B, st);
        if (rc) return rc;
    }
    if (use_fused_joseph(nl, mm)) {
        // small filters: A = I - K H, A P and the Joseph (or simple) form in one workgroup per filter
        int rc = run_joseph(h, rg, nl, tiled_rows(mm), applied, st);      // also dx = K e, the gate and the mean update
        if (rc) return rc;
    } else {
        const double* presym = rg.AP;
        bool mirror = false;
        {
        ProfScope ps(h, PC_STRUCT_GAIN, st);
        // A = I - K H                                (updateP, :563-565)
        if (kh_one_pass) {                               // ... and dx = K e (:228) on the same pass over K
            const size_t lkh = (size_t)4 * KH_ROWS * ml * sizeof(double);
            int rc_ = ensure_dynamic_lds(h, reinterpret_cast<const void*>(eye_minus_kh_rows_kernel),
                                         (size_t)4 * KH_ROWS * KH_ONE_PASS_MAX_ML * sizeof(double));
            if (rc_) return rc_;
            hipLaunchKernelGGL(eye_minus_kh_rows_kernel, dim3((nl + 4 * KH_ROWS - 1) / (4 * KH_ROWS), B), dim3(256), lkh, st,
                               rg.fv, rg.K, ldm, sNM, rg.Am, ldn, sN, nl, ml, rg.dx);
        } else {
            hipLaunchKernelGGL(eye_minus_kh_kernel, dim3((nl + 63) / 64, (nl + 4 * STRUCT_RPW - 1) / (4 * STRUCT_RPW), B),
                               dim3(256), 0, st, rg.fv, rg.K, ldm, sNM, rg.Am, ldn, sN, nl, nl, 0, STRUCT_RPW);
            hipLaunchKernelGGL(eye_minus_kh_xcols_kernel, dim3((nl + 3) / 4, B), dim3(256), 0, st, rg.fv, rg.K, ldm, sNM,
                               rg.Am, ldn, sN, nl, 0, ml);
        }
        }
        {
        ProfScope ps(h, PC_GEMM_COV, st);
        // AP = A P
        HIPCHK(h, (launch_gemm<false, EPI_NONE, false, TRI_NONE>(
                      gargs(rg.Am, ldn, sN, rg.fv.P, ldn, sN, rg.AP, ldn, sN, nl, nl, nl), B, st)));
        if (h->prm.update_form == 0) {
            // Joseph: P = AP A^T + (K R) K^T, symmetric by construction: only the lower tiles are computed
            HIPCHK(h, (launch_gemm<true, EPI_NONE, true, TRI_NONE>(
                          gargs(rg.AP, ldn, sN, rg.Am, ldn, sN, rg.T, ldn, sN, nl, nl, nl), B, st)));
            HIPCHK(h, (launch_gemm<true, EPI_NONE, true, TRI_NONE>(
                          gargs(rg.KR, ldm, sNM, rg.K, ldm, sNM, rg.T, ldn, sN, nl, nl, ml, 1.0, 1.0), B, st)));
            presym = rg.T;
            mirror = true;
        }
        }
        ProfScope ps(h, PC_FEATURE, st);
        if (mirror)   // P = the symmetric completion of the lower triangle
            hipLaunchKernelGGL(symmetrize_kernel<true>, dim3((nl + 31) / 32, (nl + 31) / 32, B), dim3(256), 0, st, presym,
                               rg.fv.P, ldn, sN, nl);
        else          // P = (P + P^T)/2                      (:567)
            hipLaunchKernelGGL(symmetrize_kernel<false>, dim3((nl + 31) / 32, (nl + 31) / 32, B), dim3(256), 0, st, presym,
                               rg.fv.P, ldn, sN, nl);
    }
    ProfScope ps_tail(h, PC_FEATURE, st);
    // dx = K e, gate, mean update                (main.cpp:228-232)
    if (!use_fused_joseph(nl, mm) && !kh_one_pass)      // the fused covariance update has already formed dx
        hipLaunchKernelGGL(gain_kx_kernel, dim3((nl + 15) / 16, B), dim3(256), 0, st, rg.fv, rg.K, ldm, rg.dx, ml);
    if (!use_fused_joseph(nl, mm) || !fuse_apply(B))
        hipLaunchKernelGGL(gain_apply_kernel, dim3(B), dim3(256), 0, st, rg.fv, rg.dx, applied);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// Launch sequence of one whole iteration (frame begin, predict, update) for all filters, in `groups` ranges.
// Leaves the updated covariance in the T buffer (the caller swaps P and T) unless use_seq_step(h): then in P itself.
int step_launch(slamk_handle* h, const double* qbw_dev, const double* w_dev, const double* acc_dev, const double* dt_dev,
                const double* alt_dev, const double* src_dev, const double* refl_dev, int32_t* applied_dev,
                hipStream_t st, int groups, bool frame_begin)
{
    const int G = std::max(1, std::min(groups, h->B));
    if ((int)h->gstreams.size() < G && G > 1) {
        const size_t old = h->gstreams.size();
        h->gstreams.resize(G); h->gdone.resize(G);
        for (size_t g = old; g < (size_t)G; ++g) {
            HIPCHK(h, hipStreamCreateWithFlags(&h->gstreams[g], hipStreamNonBlocking));
            HIPCHK(h, hipEventCreateWithFlags(&h->gdone[g], hipEventDisableTiming));
        }
    }
    if (G > 1 && !h->fork) HIPCHK(h, hipEventCreateWithFlags(&h->fork, hipEventDisableTiming));
    const size_t F = h->F;
    const bool one_launch = use_seq_step(h);
    if (G > 1) HIPCHK(h, hipEventRecord(h->fork, st));
    int rc = 0;
    for (int g = 0; g < G; ++g) {
        const int b0 = (int)((long long)h->B * g / G), b1 = (int)((long long)h->B * (g + 1) / G);
        hipStream_t gs = G > 1 ? h->gstreams[g] : st;
        hipError_t he = hipSuccess;
        if (G > 1) he = hipStreamWaitEvent(gs, h->fork, 0);
        if (!rc && he == hipSuccess) {
            const Range rp = make_range(h, b0, b1 - b0);
            if (one_launch)         // the covariance stays in its buffer: the caller does not exchange P and T
                rc = run_seq_step(h, rp, qbw_dev + 4 * (size_t)b0, w_dev + 3 * (size_t)b0, acc_dev + 3 * (size_t)b0, dt_dev + b0,
                                  (alt_dev ? alt_dev : h->alt_d) + b0, src_dev ? src_dev + 2 * F * b0 : nullptr,
                                  refl_dev ? refl_dev + 2 * F * b0 : nullptr, frame_begin,
                                  applied_dev ? applied_dev + b0 : nullptr, gs);
            else {
            rc = run_predict(h, rp, qbw_dev + 4 * (size_t)b0, w_dev + 3 * (size_t)b0, acc_dev + 3 * (size_t)b0, dt_dev + b0,
                             gs, frame_begin);
            if (!rc)
                rc = run_update(h, after_predict(rp), qbw_dev + 4 * (size_t)b0, (alt_dev ? alt_dev : h->alt_d) + b0,
                                src_dev ? src_dev + 2 * F * b0 : nullptr, refl_dev ? refl_dev + 2 * F * b0 : nullptr,
                                applied_dev ? applied_dev + b0 : nullptr, gs);
            }
        }
        // every forked stream is joined back into `st`, also after a failure: the caller's stream never ends up
        // ahead of work that this call queued
        if (G > 1 && he == hipSuccess) {
            he = hipEventRecord(h->gdone[g], gs);
            if (he == hipSuccess) he = hipStreamWaitEvent(st, h->gdone[g], 0);
        }
        if (he != hipSuccess && !rc) rc = fail(h, SLAMK_EHIP, "step: stream fork/join: %s", hipGetErrorString(he));
    }
    return rc;
}

// useful flops of one GEMM launch as the kernel executes it (tile skipping and k-range trimming
// included, zero padding inside edge tiles excluded)
double gemm_flops(int M, int N, int K, bool lower, int tri)
{
    const bool big = gemm_big_tile(M, N, 1, lower, tri);
    const int BM = big ? 128 : 64, BN = BM;
    double f = 0;
    for (int m0 = 0; m0 < M; m0 += BM)
        for (int n0 = 0; n0 < N; n0 += BN) {
            if (lower && n0 > m0 + BM - 1) continue;
            int kb = 0, ke = K;
            if (tri == TRI_B_KLE_N) ke = std::min(K, n0 + BN);
            if (tri == TRI_B_KGE_N) kb = std::min(K, n0);
            if (tri == TRI_A_KLE_M) ke = std::min(K, m0 + BM);
            if (lower && !big) {        // the small tile also skips its 16 x 16 blocks strictly above the diagonal
                for (int r0 = m0; r0 < std::min(M, m0 + BM); r0 += 16)
                    for (int c0 = n0; c0 < std::min(N, n0 + BN); c0 += 16)
                        if (c0 <= r0 + 15) f += 2.0 * std::min(16, M - r0) * std::min(16, N - c0) * std::max(0, ke - kb);
                continue;
            }
            f += 2.0 * std::min(BM, M - m0) * std::min(BN, N - n0) * std::max(0, ke - kb);
        }
    return f;
}

// update_form = 2 only exists for filters whose covariance fits LDS: checked at the entry of every call that runs an
// update, before anything is queued (features arrive with the frames, so a running filter can grow past the limit)
int check_update_form(slamk_handle* h)
{
    if (h->prm.update_form == 2 && max_nl(h) > 16 * SEQ_MAX_NT)
        return fail(h, SLAMK_EINVAL, "update_form 2 (block-sequential) needs n <= %d, got %d", 16 * SEQ_MAX_NT, max_nl(h));
    return 0;
}

// Frame begin (old_pos = X, depth clamp) for the filters of the batch that have not had theirs yet, one launch per run of
// such filters: only taken when SOME filters were begun by slamk_frame(b, ...) -- their old_pos must stay what it was
// before any state override (src/slam/main.cpp:168 comes before :180-185).
int begin_remaining(slamk_handle* h, hipStream_t st)
{
    for (int b = 0; b < h->B;) {
        if (h->begun[b]) { ++b; continue; }
        int e = b;
        while (e < h->B && !h->begun[e]) ++e;
        const Range rg = make_range(h, b, e - b);
        hipLaunchKernelGGL(frame_begin_kernel, dim3(e - b), dim3(256), 0, st, rg.fv, -1);
        b = e;
    }
    HIPCHK(h, hipGetLastError());
    h->mark_begun(0, h->B);
    return 0;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

void slamk_default_params(slamk_params* p)
{
    p->Q0 = 256; p->R0 = 32; p->qbias = 1e-5; p->p0 = 2; p->pinit = 1e-4;
    p->alt_override = -1.34; p->use_alt_override = 1;
    p->gate = 4; p->d_min = 0.1; p->d_max = 10e3; p->dinit = 100;
    p->update_form = 0;
}

const char* slamk_last_error(const slamk_handle* h) { return h ? h->err.c_str() : "null handle"; }

int slamk_create(const slamk_params* p, int max_features, int batch, int device, slamk_handle** out)
{
    if (!out || max_features < 0 || batch < 1) return SLAMK_EINVAL;
    slamk_handle* h = new slamk_handle();
    if (p) h->prm = *p; else slamk_default_params(&h->prm);
    h->B = batch; h->F = std::max(1, max_features); h->device = device;
    h->chol_lookahead = true;
    h->ldn = pad16(9 + 3 * h->F);
    h->ldm = pad16(1 + 6 * h->F);
    h->hf.resize(batch);
    h->begun.assign(batch, 0);
    *out = h;
    ENTER(h);
    { double* sp_ = nullptr; HIPCHK(h, static_scratch(&sp_)); }   // set-up of the no-allocation entry points on this device
    const size_t B = batch, F = h->F, ldn = h->ldn, ldm = h->ldm;
    FilterView& v = h->fv;
    v.B = batch; v.F = h->F; v.ldn = h->ldn; v.ldm = h->ldm;
    int rc = 0;
    rc |= dalloc(h, &v.X, B * 3); rc |= dalloc(h, &v.V, B * 3); rc |= dalloc(h, &v.bias, B * 3);
    rc |= dalloc(h, &v.old_pos, B * 3);
    rc |= dalloc(h, &v.nf, B); rc |= dalloc(h, &v.m, B); rc |= dalloc(h, &v.status, B);
    rc |= dalloc(h, &v.pib, B * 3 * F); rc |= dalloc(h, &v.anchor, B * 3 * F);
    rc |= dalloc(h, &v.q0, B * 4 * F); rc |= dalloc(h, &v.pib0, B * 2 * F);
    rc |= dalloc(h, &v.isref, B * F); rc |= dalloc(h, &v.rowoff, B * F);
    rc |= dalloc(h, &v.msrc, B * 2 * F); rc |= dalloc(h, &v.mrefl, B * 2 * F);
    rc |= dalloc(h, &v.P, B * ldn * ldn);
    rc |= dalloc(h, &v.ha, B * ldm * 3); rc |= dalloc(h, &v.hb, B * ldm * 3); rc |= dalloc(h, &v.hcs, B * ldm);
    rc |= dalloc(h, &v.e, B * ldm); rc |= dalloc(h, &v.rdiag, B * ldm); rc |= dalloc(h, &v.qdiag, B * ldn);
    rc |= dalloc(h, &v.fa, B * ldn * 3); rc |= dalloc(h, &v.fg, B * ldn * 3); rc |= dalloc(h, &v.fgs, B * ldn);
    rc |= dalloc(h, &h->T, B * ldn * ldn);
    rc |= dalloc(h, &h->K, B * ldn * ldm);
    rc |= dalloc(h, &h->dx, B * ldn);
    rc |= dalloc(h, &h->qbw_d, B * 4); rc |= dalloc(h, &h->w_d, B * 3); rc |= dalloc(h, &h->acc_d, B * 3);
    rc |= dalloc(h, &h->dt_d, B); rc |= dalloc(h, &h->alt_d, B); rc |= dalloc(h, &h->applied_d, B);
    rc |= dalloc(h, &h->feat_tmp, B * FRAME_COMPS * F);
    if (rc) return SLAMK_EHIP;
    v.Q0 = h->prm.Q0; v.R0 = h->prm.R0; v.qbias = h->prm.qbias; v.gate = h->prm.gate;
    v.inv_dmax = 1 / h->prm.d_max; v.inv_dmin = 1 / h->prm.d_min;
    v.alt_override = h->prm.alt_override; v.use_alt_override = h->prm.use_alt_override;
    hipLaunchKernelGGL(init_cov_kernel, dim3(batch), dim3(64), 0, 0, v.P, h->ldn, (long long)ldn * ldn, h->prm.pinit);
    const std::vector<int> ones(batch, 1);      // m = 1: the altitude row
    HIPCHK(h, hipMemcpy(v.m, ones.data(), ones.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipDeviceSynchronize());
    return SLAMK_OK;
}

int slamk_destroy(slamk_handle* h)
{
    if (!h) return SLAMK_EINVAL;
    DeviceGuard dg_(h->device);
    (void)hipDeviceSynchronize();
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->stage_host) (void)hipHostFree(h->stage_host);
    if (h->stage_dev) (void)hipFree(h->stage_dev);
    if (h->stage_copied) (void)hipEventDestroy(h->stage_copied);
    for (auto& g : h->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
    for (auto g : h->gstreams) (void)hipStreamDestroy(g);
    for (auto e : h->gdone) (void)hipEventDestroy(e);
    if (h->fork) (void)hipEventDestroy(h->fork);
    if (h->chol_side) (void)hipStreamDestroy(h->chol_side);
    if (h->chol_panel_done) (void)hipEventDestroy(h->chol_panel_done);
    if (h->chol_rest_done) (void)hipEventDestroy(h->chol_rest_done);
    for (auto& r : h->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : h->prof_pool) (void)hipEventDestroy(e);
    delete h;
    return SLAMK_OK;
}

#define CHECK_B(h, b)                                                                  \
    do {                                                                               \
        if (!(h)) return SLAMK_EINVAL;                                                 \
        if ((b) < 0 || (b) >= (h)->B) return fail(h, SLAMK_EINVAL, "filter index %d out of range", (b)); \
    } while (0)

int slamk_set_state(slamk_handle* h, int b, const double* X, const double* V, const double* bias)
{
    CHECK_B(h, b);
    ENTER_HOST(h);
    if (X) HIPCHK(h, hipMemcpy(h->fv.X + 3 * b, X, 24, hipMemcpyHostToDevice));
    if (V) HIPCHK(h, hipMemcpy(h->fv.V + 3 * b, V, 24, hipMemcpyHostToDevice));
    if (bias) HIPCHK(h, hipMemcpy(h->fv.bias + 3 * b, bias, 24, hipMemcpyHostToDevice));
    return SLAMK_OK;
}

int slamk_get_state(slamk_handle* h, int b, double* X, double* V, double* bias)
{
    CHECK_B(h, b);
    ENTER_HOST(h);
    if (X) HIPCHK(h, hipMemcpy(X, h->fv.X + 3 * b, 24, hipMemcpyDeviceToHost));
    if (V) HIPCHK(h, hipMemcpy(V, h->fv.V + 3 * b, 24, hipMemcpyDeviceToHost));
    if (bias) HIPCHK(h, hipMemcpy(bias, h->fv.bias + 3 * b, 24, hipMemcpyDeviceToHost));
    return SLAMK_OK;
}

int slamk_set_cov(slamk_handle* h, int b, const double* P, int n, int ld)
{
    CHECK_B(h, b);
    if (!P || n != 9 + 3 * h->hf[b].nf || ld < n) return fail(h, SLAMK_EINVAL, "set_cov: n=%d does not match nf=%d", n, h->hf[b].nf);
    ENTER_HOST(h);
    double* dst = h->fv.P + (size_t)b * h->ldn * h->ldn;
    HIPCHK(h, hipMemset(dst, 0, (size_t)h->ldn * h->ldn * sizeof(double)));
    HIPCHK(h, hipMemcpy2D(dst, h->ldn * sizeof(double), P, ld * sizeof(double), n * sizeof(double), n, hipMemcpyHostToDevice));
    return SLAMK_OK;
}

int slamk_get_cov(slamk_handle* h, int b, double* P, int n, int ld)
{
    CHECK_B(h, b);
    if (!P || n != 9 + 3 * h->hf[b].nf || ld < n) return fail(h, SLAMK_EINVAL, "get_cov: n=%d does not match nf=%d", n, h->hf[b].nf);
    ENTER_HOST(h);
    const double* src = h->fv.P + (size_t)b * h->ldn * h->ldn;
    HIPCHK(h, hipMemcpy2D(P, ld * sizeof(double), src, h->ldn * sizeof(double), n * sizeof(double), n, hipMemcpyDeviceToHost));
    return SLAMK_OK;
}

int slamk_num_features(slamk_handle* h, int b) { CHECK_B(h, b); return h->hf[b].nf; }
int slamk_num_rows(slamk_handle* h, int b) { CHECK_B(h, b); return h->hf[b].m(); }

int slamk_set_features(slamk_handle* h, int b, int nf, const int32_t* id, const double* pib, const double* anchor,
                       const double* q0, const double* pib0, const uint8_t* is_ref)
{
    CHECK_B(h, b);
    if (nf < 0 || nf > h->F) return fail(h, SLAMK_ECAPACITY, "set_features: nf=%d exceeds max_features=%d", nf, h->F);
    if (nf && (!id || !pib || !anchor || !q0 || !pib0 || !is_ref)) return fail(h, SLAMK_EINVAL, "set_features: null array");
    ENTER_HOST(h);
    const int F = h->F;
    HostFilter& f = h->hf[b];
    f.nf = nf;
    f.id.assign(id, id + nf);
    f.isref.assign(is_ref, is_ref + nf);
    std::vector<double> t3(3 * F, 0.0), t4(4 * F, 0.0), t2(2 * F, 0.0);
    auto up = [&](double* dev, const double* src, int nc, std::vector<double>& tmp) -> int {
        std::fill(tmp.begin(), tmp.end(), 0.0);
        for (int i = 0; i < nf; ++i)
            for (int c = 0; c < nc; ++c) tmp[(size_t)c * F + i] = src[(size_t)nc * i + c];
        HIPCHK(h, hipMemcpy(dev + (size_t)b * nc * F, tmp.data(), (size_t)nc * F * sizeof(double), hipMemcpyHostToDevice));
        return 0;
    };
    if (up(h->fv.pib, pib, 3, t3) || up(h->fv.anchor, anchor, 3, t3) || up(h->fv.q0, q0, 4, t4) ||
        up(h->fv.pib0, pib0, 2, t2))
        return SLAMK_EHIP;
    return push_layout(h, b, 1);
}

int slamk_get_features(slamk_handle* h, int b, int* nf_out, int32_t* id, double* pib, double* anchor, double* q0,
                       double* pib0, uint8_t* is_ref)
{
    CHECK_B(h, b);
    ENTER_HOST(h);
    const HostFilter& f = h->hf[b];
    const int F = h->F, nf = f.nf;
    if (nf_out) *nf_out = nf;
    if (id) std::copy(f.id.begin(), f.id.end(), id);
    if (is_ref) std::copy(f.isref.begin(), f.isref.end(), is_ref);
    auto down = [&](const double* dev, double* dst, int nc) -> int {
        if (!dst) return 0;
        std::vector<double> tmp((size_t)nc * F);
        HIPCHK(h, hipMemcpy(tmp.data(), dev + (size_t)b * nc * F, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < nf; ++i)
            for (int c = 0; c < nc; ++c) dst[(size_t)nc * i + c] = tmp[(size_t)c * F + i];
        return 0;
    };
    if (down(h->fv.pib, pib, 3) || down(h->fv.anchor, anchor, 3) || down(h->fv.q0, q0, 4) || down(h->fv.pib0, pib0, 2))
        return SLAMK_EHIP;
    return SLAMK_OK;
}

// ---- image frame ------------------------------------------------------------------------------
// What a frame does to the feature list of one host mirror (States::update_features, states.cpp:104-200).
struct FramePlan {
    int old_nf = 0, nf = 0;
    std::vector<int32_t> nid;       // ids after the frame, slot order
    std::vector<uint8_t> nisr;
    std::vector<int> nsrc;          // previous slot of every new slot, -1 = fresh
    std::vector<int> nfresh;        // index into the matches for fresh slots, else -1
    bool identity = true;
};

static int plan_frame(slamk_handle* h, const HostFilter& f, const slamk_match* matches, int nmatch, FramePlan& p)
{
    const int old_nf = f.nf;
    std::vector<int32_t> ids = f.id;
    std::vector<uint8_t> isr = f.isref;
    std::vector<int> nomatch(old_nf, 1), srcslot(old_nf), fresh_of;   // fresh_of[slot] = match index or -1
    for (int i = 0; i < old_nf; ++i) {
        srcslot[i] = i;
        for (int k = 0; k < nmatch; ++k)
            if (ids[i] == matches[k].id) { nomatch[i] = 0; break; }
    }
    fresh_of.assign(old_nf, -1);
    for (int k = 0; k < nmatch; ++k) {
        bool found = false;
        for (size_t i = 0; i < ids.size(); ++i)
            if (ids[i] == matches[k].id) { found = true; break; }
        if (found) continue;
        const uint8_t ref = !(matches[k].rx == 0 && matches[k].ry == 0);   // projection::isRef, featureIO.h:36
        int slot = -1;
        for (size_t i = 0; i < ids.size(); ++i)
            if (nomatch[i] != 0) { slot = (int)i; break; }
        if (slot >= 0) {
            ids[slot] = matches[k].id; isr[slot] = ref; nomatch[slot] = 0; srcslot[slot] = -1; fresh_of[slot] = k;
        } else {
            ids.push_back(matches[k].id); isr.push_back(ref); nomatch.push_back(0); srcslot.push_back(-1);
            fresh_of.push_back(k);
        }
    }
    // erase what is still unmatched, keeping order
    p = FramePlan();
    p.old_nf = old_nf;
    for (size_t i = 0; i < ids.size(); ++i) {
        if (nomatch[i] != 0) continue;
        p.nid.push_back(ids[i]); p.nisr.push_back(isr[i]); p.nsrc.push_back(srcslot[i]); p.nfresh.push_back(fresh_of[i]);
    }
    p.nf = (int)p.nid.size();
    if (p.nf > h->F) return fail(h, SLAMK_ECAPACITY, "frame: %d features exceed max_features=%d", p.nf, h->F);
    p.identity = (p.nf == old_nf);
    for (int i = 0; i < p.nf && p.identity; ++i) p.identity = (p.nsrc[i] == i);
    return SLAMK_OK;
}

// One group of filters that take the same plan: the frame's matches and attitude, and the plan.
struct FrameGroup {
    FramePlan plan;
    const slamk_match* matches;
    int nmatch;
    const double* quat;
};

// Applies the plans to the filters [b0, b0 + cnt): grp[c] = group of filter b0 + c.  One upload (from a pinned
// staging block) and a fixed number of launches on `st`, whatever the number of filters; no host synchronisation
// unless `sync`.
static int apply_frames(slamk_handle* h, int b0, int cnt, const std::vector<FrameGroup>& groups, const std::vector<int>& grp,
                        hipStream_t st, bool sync)
{
    const size_t F = h->F, ldn = h->ldn, G = groups.size();
    // staging layout: doubles first (sxy, quat, ms, mr), then ints (grp, nf, m, changed, isref, rowoff, smap, imap)
    const size_t nd = G * (2 * F + 4 + 2 * F + 2 * F), ni = cnt + 3 * G + 3 * G * F + G * ldn;
    const size_t bytes = nd * sizeof(double) + ni * sizeof(int);
    if (h->stage_in_flight) {                     // the previous upload has left the pinned block
        HIPCHK(h, hipEventSynchronize(h->stage_copied));
        h->stage_in_flight = false;
    }
    if (bytes > h->stage_bytes) {
        if (h->stage_host) { HIPCHK(h, hipStreamSynchronize(st)); (void)hipHostFree(h->stage_host); (void)hipFree(h->stage_dev); }
        const size_t cap = std::max(bytes * 2, (size_t)1 << 16);
        HIPCHK(h, hipHostMalloc(&h->stage_host, cap, hipHostMallocDefault));
        HIPCHK(h, hipMalloc(&h->stage_dev, cap));
        h->stage_bytes = cap;
        if (!h->stage_copied) HIPCHK(h, hipEventCreateWithFlags(&h->stage_copied, hipEventDisableTiming));
    }
    double* hd = static_cast<double*>(h->stage_host);
    int* hi = reinterpret_cast<int*>(hd + nd);
    memset(h->stage_host, 0, bytes);
    double *sxy = hd, *quat = sxy + G * 2 * F, *ms = quat + G * 4, *mr = ms + G * 2 * F;
    int *gr = hi, *nfv = gr + cnt, *mv = nfv + G, *chg = mv + G, *isr = chg + G, *ro = isr + G * F, *smap = ro + G * F,
        *imap = smap + G * F;
    bool any_changed = false;
    int nl = 16, nf_hi = 0;
    for (size_t g = 0; g < G; ++g) {
        const FramePlan& p = groups[g].plan;
        nfv[g] = p.nf;
        int row = 0;
        for (int i = 0; i < p.nf; ++i) {
            isr[g * F + i] = p.nisr[i];
            ro[g * F + i] = row;
            row += p.nisr[i] ? 6 : 4;
            smap[g * F + i] = p.nsrc[i];
            // the frame's observation of the feature in slot i (the id search of measurementModel, main.cpp:362-375)
            const slamk_match* mt = nullptr;
            if (p.nfresh[i] >= 0) mt = &groups[g].matches[p.nfresh[i]];
            else
                for (int k = 0; k < groups[g].nmatch; ++k)
                    if (groups[g].matches[k].id == p.nid[i]) { mt = &groups[g].matches[k]; break; }
            if (!mt) return fail(h, SLAMK_ENOMATCH, "Cannot find match.");
            ms[g * 2 * F + i] = mt->sx; ms[g * 2 * F + F + i] = mt->sy;
            mr[g * 2 * F + i] = mt->rx; mr[g * 2 * F + F + i] = mt->ry;
            if (p.nfresh[i] >= 0) { sxy[g * 2 * F + i] = mt->sx; sxy[g * 2 * F + F + i] = mt->sy; }
        }
        mv[g] = row + 1;                           // feature rows first, the altitude row last
        for (int c = 0; c < 4; ++c) quat[4 * g + c] = groups[g].quat[c];
        chg[g] = p.identity ? 0 : 1;
        int* im = imap + g * ldn;
        for (size_t r = 0; r < ldn; ++r) im[r] = -2;
        for (int c = 0; c < 6; ++c) im[c] = c;
        for (int i = 0; i < p.nf; ++i)
            for (int c = 0; c < 3; ++c) im[6 + 3 * i + c] = p.nsrc[i] >= 0 ? 6 + 3 * p.nsrc[i] + c : -1;
        for (int c = 0; c < 3; ++c) im[6 + 3 * p.nf + c] = 6 + 3 * p.old_nf + c;
        if (!p.identity) {
            any_changed = true;
            nl = std::max(nl, pad16(9 + 3 * std::max(p.nf, p.old_nf)));
        }
        nf_hi = std::max(nf_hi, p.nf);
    }
    for (int c = 0; c < cnt; ++c) gr[c] = grp[c];

    HIPCHK(h, hipMemcpyAsync(h->stage_dev, h->stage_host, bytes, hipMemcpyHostToDevice, st));
    HIPCHK(h, hipEventRecord(h->stage_copied, st));
    h->stage_in_flight = true;
    const double* dd = static_cast<const double*>(h->stage_dev);
    const int* di = reinterpret_cast<const int*>(dd + nd);
    FrameStage fs;
    fs.sxy = dd; fs.quat = dd + G * 2 * F; fs.ms = fs.quat + G * 4; fs.mr = fs.ms + G * 2 * F;
    fs.grp = di; fs.nf = di + cnt; fs.m = fs.nf + G; fs.changed = fs.m + G; fs.isref = fs.changed + G;
    fs.rowoff = fs.isref + G * F; fs.smap = fs.rowoff + G * F; fs.imap = fs.smap + G * F;

    const Range rg = make_range(h, b0, cnt);
    double* tmp = h->feat_tmp + (size_t)b0 * FRAME_COMPS * F;
    const dim3 fgrid((unsigned)((F + 255) / 256), (unsigned)cnt);
    // old_pos = X, depth clamp (main.cpp:168, states.cpp:91-103): before any slot moves
    hipLaunchKernelGGL(frame_begin_kernel, dim3(cnt), dim3(256), 0, st, rg.fv, -1);
    if (any_changed) hipLaunchKernelGGL(frame_stash_kernel, fgrid, dim3(256), 0, st, rg.fv, fs, tmp);
    hipLaunchKernelGGL(frame_gather_kernel, fgrid, dim3(256), 0, st, rg.fv, fs, tmp, 1 / h->prm.dinit);
    if (any_changed) {
        const size_t img = (size_t)nl * nl * sizeof(double);
        if (nl <= 128) {         // the covariance of a filter fits LDS: in place, one pass
            // (the attribute is set once per device: ask for the largest image this path takes)
            int rc_ = ensure_dynamic_lds(h, reinterpret_cast<const void*>(frame_remap_cov_lds_kernel), 128 * 128 * sizeof(double));
            if (rc_) return rc_;
            hipLaunchKernelGGL(frame_remap_cov_lds_kernel, dim3(cnt), dim3(512), img, st, rg.fv, fs, rg.fv.P, nl, h->prm.p0);
        } else {
            const dim3 cgrid((nl + 63) / 64, (nl + 3) / 4, cnt);
            hipLaunchKernelGGL(frame_remap_cov_kernel<false>, cgrid, dim3(256), 0, st, rg.fv, fs, rg.fv.P, rg.T, nl, h->prm.p0);
            hipLaunchKernelGGL(frame_remap_cov_kernel<true>, cgrid, dim3(256), 0, st, rg.fv, fs, rg.fv.P, rg.T, nl, h->prm.p0);
        }
    }
    HIPCHK(h, hipGetLastError());
    for (int c = 0; c < cnt; ++c) {
        const FramePlan& p = groups[grp[c]].plan;
        HostFilter& f = h->hf[b0 + c];
        f.nf = p.nf; f.id = p.nid; f.isref = p.nisr;
    }
    h->mark_begun(b0, cnt);
    if (sync) HIPCHK(h, hipStreamSynchronize(st));
    return SLAMK_OK;
}

static bool same_mirror(const HostFilter& a, const HostFilter& b)
{
    return a.nf == b.nf && a.id == b.id && a.isref == b.isref;
}

int slamk_frame(slamk_handle* h, int b, const slamk_match* matches, int nmatch, const double quat[4])
{
    if (!h || nmatch < 0 || (nmatch && !matches) || !quat) return SLAMK_EINVAL;
    ENTER_HOST(h);
    std::vector<FrameGroup> groups;
    std::vector<int> grp;
    if (b >= 0) {
        CHECK_B(h, b);
        groups.resize(1);
        groups[0].matches = matches; groups[0].nmatch = nmatch; groups[0].quat = quat;
        int rc = plan_frame(h, h->hf[b], matches, nmatch, groups[0].plan);
        if (rc) return rc;
        grp.assign(1, 0);
        return apply_frames(h, b, 1, groups, grp, 0, true);
    }
    // the same frame for every filter: runs of filters with identical feature lists share one plan
    grp.resize(h->B);
    for (int i = 0; i < h->B; ++i) {
        if (i > 0 && same_mirror(h->hf[i], h->hf[i - 1])) { grp[i] = grp[i - 1]; continue; }
        groups.emplace_back();
        FrameGroup& g = groups.back();
        g.matches = matches; g.nmatch = nmatch; g.quat = quat;
        int rc = plan_frame(h, h->hf[i], matches, nmatch, g.plan);
        if (rc) return rc;
        grp[i] = (int)groups.size() - 1;
    }
    return apply_frames(h, 0, h->B, groups, grp, 0, true);
}

int slamk_frame_batch_async(slamk_handle* h, const slamk_match* matches, int nmatch, const int32_t* match_off,
                            const double* quat, void* stream)
{
    if (!h || !quat || nmatch < 0) return SLAMK_EINVAL;
    if (match_off) {
        for (int b = 0; b < h->B; ++b)
            if (match_off[b] < 0 || match_off[b + 1] < match_off[b]) return fail(h, SLAMK_EINVAL, "frame_batch: match_off must ascend");
        if (match_off[h->B] > 0 && !matches) return SLAMK_EINVAL;
    } else if (nmatch && !matches)
        return SLAMK_EINVAL;
    ENTER(h);
    note_stream(h, stream);
    std::vector<FrameGroup> groups;
    std::vector<int> grp(h->B);
    for (int i = 0; i < h->B; ++i) {
        const slamk_match* mi = match_off ? matches + match_off[i] : matches;
        const int ni = match_off ? match_off[i + 1] - match_off[i] : nmatch;
        const double* qi = match_off ? quat + 4 * (size_t)i : quat;
        if (i > 0 && same_mirror(h->hf[i], h->hf[i - 1])) {
            // the same feature list and the same frame as the previous filter: the same plan
            const FrameGroup& pg = groups[grp[i - 1]];
            if (pg.nmatch == ni && (pg.matches == mi || memcmp(pg.matches, mi, (size_t)ni * sizeof(slamk_match)) == 0) &&
                (pg.quat == qi || memcmp(pg.quat, qi, 4 * sizeof(double)) == 0)) {
                grp[i] = grp[i - 1];
                continue;
            }
        }
        groups.emplace_back();
        FrameGroup& g = groups.back();
        g.matches = mi; g.nmatch = ni; g.quat = qi;
        int rc = plan_frame(h, h->hf[i], mi, ni, g.plan);      // nothing has been queued yet when a plan fails
        if (rc) return rc;
        grp[i] = (int)groups.size() - 1;
    }
    return apply_frames(h, 0, h->B, groups, grp, (hipStream_t)stream, false);
}

int slamk_set_frame_values(slamk_handle* h, int b, const double* src2, const double* refl2)
{
    CHECK_B(h, b);
    if (!src2 || !refl2) return SLAMK_EINVAL;
    ENTER_HOST(h);
    const int F = h->F, nf = h->hf[b].nf;
    std::vector<double> ms(2 * F, 0.0), mr(2 * F, 0.0);
    for (int i = 0; i < nf; ++i) {
        ms[i] = src2[2 * i]; ms[F + i] = src2[2 * i + 1];
        mr[i] = refl2[2 * i]; mr[F + i] = refl2[2 * i + 1];
    }
    HIPCHK(h, hipMemcpy(h->fv.msrc + (size_t)b * 2 * F, ms.data(), 2 * F * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->fv.mrefl + (size_t)b * 2 * F, mr.data(), 2 * F * sizeof(double), hipMemcpyHostToDevice));
    return SLAMK_OK;
}

// ---- predict / update ----------------------------------------------------------------------------
int slamk_frame_begin_dev(slamk_handle* h, void* stream)
{
    if (!h) return SLAMK_EINVAL;
    ENTER(h);
    note_stream(h, stream);
    hipLaunchKernelGGL(frame_begin_kernel, dim3(h->B), dim3(256), 0, (hipStream_t)stream, h->fv, -1);
    HIPCHK(h, hipGetLastError());
    h->mark_begun(0, h->B);
    return SLAMK_OK;
}

int slamk_predict_dev(slamk_handle* h, const double* qbw_dev, const double* w_dev, const double* acc_dev,
                      const double* dt_dev, void* stream)
{
    if (!h || !qbw_dev || !w_dev || !acc_dev || !dt_dev) return SLAMK_EINVAL;
    ENTER(h);
    { int rc_ = ensure_workspaces(h); if (rc_) return rc_; }
    note_stream(h, stream);
    int rc = run_predict(h, make_range(h, 0, h->B), qbw_dev, w_dev, acc_dev, dt_dev, (hipStream_t)stream);
    if (rc) return rc;                 // P still names the buffer that holds the covariance
    std::swap(h->fv.P, h->T);          // T is whichever of the two covariance-sized buffers is not P
    h->clear_begun();
    return SLAMK_OK;
}

int slamk_update_dev(slamk_handle* h, const double* qbw_dev, const double* alt_dev, const double* src_dev,
                     const double* refl_dev, int32_t* applied_dev, void* stream)
{
    if (!h || !qbw_dev) return SLAMK_EINVAL;
    if (!alt_dev && !h->prm.use_alt_override) return fail(h, SLAMK_EINVAL, "update: alt required without alt_override");
    { int rc_ = check_update_form(h); if (rc_) return rc_; }
    ENTER(h);
    { int rc_ = ensure_workspaces(h); if (rc_) return rc_; }
    note_stream(h, stream);
    return run_update(h, make_range(h, 0, h->B), qbw_dev, alt_dev ? alt_dev : h->alt_d, src_dev, refl_dev, applied_dev,
                      (hipStream_t)stream);
}

// One loop iteration with an image for every filter (src/slam/main.cpp:167-233): frame begin, predict, update.
// The batch is cut into `groups` ranges that run the whole chain on streams of their own, so that the
// HBM-bound kernels of one range overlap the LDS-resident per-filter kernels of another; everything is
// ordered after the work already in `stream` and complete, in stream order, when the call returns.
int slamk_step_dev(slamk_handle* h, const double* qbw_dev, const double* w_dev, const double* acc_dev,
                   const double* dt_dev, const double* alt_dev, const double* src_dev, const double* refl_dev,
                   int32_t* applied_dev, void* stream)
{
    if (!h || !qbw_dev || !w_dev || !acc_dev || !dt_dev) return SLAMK_EINVAL;
    if (!alt_dev && !h->prm.use_alt_override) return fail(h, SLAMK_EINVAL, "step: alt required without alt_override");
    { int rc_ = check_update_form(h); if (rc_) return rc_; }
    ENTER(h);
    { int rc_ = ensure_workspaces(h); if (rc_) return rc_; }
    hipStream_t st = (hipStream_t)stream;
    note_stream(h, stream);
    // slamk_frame / slamk_frame_begin_dev of this iteration have already set old_pos and clamped the depths; a
    // state override between them and this call (UPDATE_POS, src/slam/main.cpp:180-185) must not move old_pos,
    // which the reference takes before the override (:168).  Per filter: when only some filters have been begun, the
    // others get their frame begin here, in launches of their own, and the step itself runs none.
    if (h->nbegun > 0 && h->nbegun < h->B) { int rc_ = begin_remaining(h, st); if (rc_) return rc_; }
    const bool frame_begin = h->nbegun < h->B;
    if (h->graph_mode && !h->prof_mask && st != nullptr) {
        // replay (or first capture) the launch sequence as one hipGraph: one submission per iteration
        slamk_handle::GraphKey key;
        memset(&key, 0, sizeof key);
        const void* ptrs[8] = {qbw_dev, w_dev, acc_dev, dt_dev, alt_dev, src_dev, refl_dev, applied_dev};
        memcpy(key.p, ptrs, sizeof ptrs);
        key.P = h->fv.P; key.nl = max_nl(h); key.ml = max_m(h); key.form = h->prm.update_form;
        key.debug = h->debug; key.outer_block = chol_outer_block(); key.inverse_block = INV_BS;
        key.frame_begin = frame_begin;
        for (const auto& f : h->hf) key.nfmax = std::max(key.nfmax, f.nf);
        slamk_handle::GraphEntry* hit = nullptr;
        for (auto& g : h->graphs)
            if (g.key == key) hit = &g;
        if (!hit) {
            if (h->graphs.size() >= 16) {       // layouts come and go with the features: keep the cache small
                for (auto& g : h->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
                h->graphs.clear();
            }
            slamk_handle::GraphEntry e;
            e.key = key;
            HIPCHK(h, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            e.graph = nullptr;
            int rc = step_launch(h, qbw_dev, w_dev, acc_dev, dt_dev, alt_dev, src_dev, refl_dev, applied_dev, st, 1,
                                 frame_begin);
            hipError_t ce = hipStreamEndCapture(st, &e.graph);
            if (rc || ce != hipSuccess) {
                if (e.graph) (void)hipGraphDestroy(e.graph);
                return rc ? rc : fail(h, SLAMK_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(ce));
            }
            ce = hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0);
            if (ce != hipSuccess) {
                (void)hipGraphDestroy(e.graph);
                return fail(h, SLAMK_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(ce));
            }
            h->graphs.push_back(e);
            hit = &h->graphs.back();
        }
        HIPCHK(h, hipGraphLaunch(hit->exec, st));
        if (!use_seq_step(h)) std::swap(h->fv.P, h->T);
        h->clear_begun();
        return SLAMK_OK;
    }
    int rc = step_launch(h, qbw_dev, w_dev, acc_dev, dt_dev, alt_dev, src_dev, refl_dev, applied_dev, st, h->groups,
                         frame_begin);
    if (rc) return rc;                 // P still names the buffer that holds the last complete covariance
    if (!use_seq_step(h)) std::swap(h->fv.P, h->T);
    h->clear_begun();
    return SLAMK_OK;
}

int slamk_set_graph_mode(slamk_handle* h, int on)
{
    if (!h) return SLAMK_EINVAL;
    h->graph_mode = on != 0;
    return SLAMK_OK;
}

int slamk_set_pipeline_groups(slamk_handle* h, int groups)
{
    if (!h || groups < 1 || groups > 64) return SLAMK_EINVAL;
    h->groups = groups;
    return SLAMK_OK;
}

int slamk_sync(slamk_handle* h, void* stream)
{
    if (!h) return SLAMK_EINVAL;
    ENTER(h);
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    if (h->pending && h->pending_stream == (hipStream_t)stream) h->pending = false;
    std::vector<int> st(h->B);
    HIPCHK(h, hipMemcpy(st.data(), h->fv.status, h->B * sizeof(int), hipMemcpyDeviceToHost));
    int first = -1, nbad = 0;
    for (int b = 0; b < h->B; ++b)
        if (st[b]) { if (first < 0) first = b; ++nbad; }
    if (first < 0) return SLAMK_OK;
    // the words are cleared once they have been reported: the text names the first filter and how many there are,
    // slamk_read_status (before this call) gives all of them
    HIPCHK(h, hipMemset(h->fv.status, 0, h->B * sizeof(int)));
    if (st[first] & 1)
        return fail(h, SLAMK_EREFL, "filter %d (%d of %d filters flagged): reflection rows of H, R and e disagree (zero reflection observation or innovation)", first, nbad, h->B);
    return fail(h, SLAMK_ENOTSPD, "filter %d (%d of %d filters flagged): S = H P H^T + R is not positive definite (Cholesky pivot <= 0)", first, nbad, h->B);
}

int slamk_read_status(slamk_handle* h, int32_t* status)
{
    if (!h || !status) return SLAMK_EINVAL;
    ENTER_HOST(h);
    HIPCHK(h, hipMemcpy(status, h->fv.status, h->B * sizeof(int), hipMemcpyDeviceToHost));
    return SLAMK_OK;
}

int slamk_predict(slamk_handle* h, const double* qbw, const double* w, const double* acc, const double* dt)
{
    if (!h || !qbw || !w || !acc || !dt) return SLAMK_EINVAL;
    ENTER_HOST(h);
    { int rc_ = ensure_workspaces(h); if (rc_) return rc_; }
    const size_t B = h->B;
    HIPCHK(h, hipMemcpy(h->qbw_d, qbw, B * 32, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->w_d, w, B * 24, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->acc_d, acc, B * 24, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->dt_d, dt, B * 8, hipMemcpyHostToDevice));
    int rc = run_predict(h, make_range(h, 0, h->B), h->qbw_d, h->w_d, h->acc_d, h->dt_d, 0);
    if (rc) return rc;
    std::swap(h->fv.P, h->T);
    h->clear_begun();
    HIPCHK(h, hipDeviceSynchronize());
    return SLAMK_OK;
}

int slamk_update(slamk_handle* h, const double* qbw, const double* alt, int* applied)
{
    if (!h || !qbw) return SLAMK_EINVAL;
    if (!alt && !h->prm.use_alt_override) return fail(h, SLAMK_EINVAL, "update: alt required without alt_override");
    { int rc_ = check_update_form(h); if (rc_) return rc_; }
    ENTER_HOST(h);
    { int rc_ = ensure_workspaces(h); if (rc_) return rc_; }
    const size_t B = h->B;
    HIPCHK(h, hipMemcpy(h->qbw_d, qbw, B * 32, hipMemcpyHostToDevice));
    if (alt) HIPCHK(h, hipMemcpy(h->alt_d, alt, B * 8, hipMemcpyHostToDevice));
    int rc = run_update(h, make_range(h, 0, h->B), h->qbw_d, h->alt_d, nullptr, nullptr, h->applied_d, 0);
    if (rc) return rc;
    rc = slamk_sync(h, 0);
    if (applied) HIPCHK(h, hipMemcpy(applied, h->applied_d, B * sizeof(int), hipMemcpyDeviceToHost));
    return rc;
}

// ---- stage-level entry points for the row-block sharded large-map filter ---------------------------
int slamk_stage_motion_dev(slamk_handle* h, const double* qbw_dev, const double* w_dev, const double* acc_dev,
                           const double* dt_dev, void* stream)
{
    if (!h || !qbw_dev || !w_dev || !acc_dev || !dt_dev) return SLAMK_EINVAL;
    ENTER(h);
    note_stream(h, stream);
    { int rc_ = ensure_workspaces(h); if (rc_) return rc_; }
    hipStream_t st = (hipStream_t)stream;
    const int nl = max_nl(h);
    hipLaunchKernelGGL(motion_rows_kernel, dim3((nl + 255) / 256, h->B), dim3(256), 0, st, h->fv, qbw_dev, w_dev, dt_dev, nl);
    if (h->debug)
        hipLaunchKernelGGL(build_F_kernel, dim3((nl + 63) / 64, (nl + 3) / 4, h->B), dim3(256), 0, st, h->fv, qbw_dev,
                           w_dev, dt_dev, h->Fm, nl);
    hipLaunchKernelGGL(dynamics_kernel, dim3(h->B), dim3(256), 0, st, h->fv, qbw_dev, w_dev, acc_dev, dt_dev);
    HIPCHK(h, hipGetLastError());
    return SLAMK_OK;
}

int slamk_stage_measure_dev(slamk_handle* h, const double* qbw_dev, const double* alt_dev, const double* src_dev,
                            const double* refl_dev, void* stream)
{
    if (!h || !qbw_dev) return SLAMK_EINVAL;
    ENTER(h);
    note_stream(h, stream);
    { int rc_ = ensure_workspaces(h); if (rc_) return rc_; }
    hipStream_t st = (hipStream_t)stream;
    const int nl = max_nl(h), ml = max_ml(h);
    int nfmax = 0;
    for (const auto& f : h->hf) nfmax = std::max(nfmax, f.nf);
    hipLaunchKernelGGL(measure_kernel, dim3(std::max(1, (nfmax + 63) / 64), h->B), dim3(64), 0, st, h->fv, qbw_dev,
                       alt_dev ? alt_dev : h->alt_d, src_dev ? src_dev : h->fv.msrc, refl_dev ? refl_dev : h->fv.mrefl,
                       ml);
    if (h->debug)
        hipLaunchKernelGGL(build_H_kernel, dim3((nl + 63) / 64, (ml + 3) / 4, h->B), dim3(256), 0, st, h->fv, h->Hm, ml, nl);
    HIPCHK(h, hipGetLastError());
    return SLAMK_OK;
}

// which: 0 = the motion Jacobian F (nl rows), 1 = the measurement Jacobian H (ml rows), filter 0 of the handle
static bool stage_operator(slamk_handle* h, int which, RowSparse* M, int* mrows, const double** diag)
{
    if (which == 0) { *M = RowSparse{h->fv.fa, h->fv.fg, h->fv.fgs, h->ldn, 3}; *mrows = max_nl(h); *diag = h->fv.qdiag; }
    else if (which == 1) { *M = RowSparse{h->fv.ha, h->fv.hb, h->fv.hcs, h->ldm, 0}; *mrows = max_ml(h); *diag = h->fv.rdiag; }
    else return false;
    return true;
}

int slamk_stage_right_dev(slamk_handle* h, int which, const double* In_dev, int ldi, int rows, double* Out_dev, int ldo,
                          void* stream)
{
    RowSparse M; int mrows; const double* diag;
    if (!h || !In_dev || !Out_dev || rows < 0 || !stage_operator(h, which, &M, &mrows, &diag)) return SLAMK_EINVAL;
    if (ldi < max_nl(h) || ldo < mrows) return fail(h, SLAMK_EINVAL, "stage_right: leading dimensions too small");
    ENTER(h);
    note_stream(h, stream);
    if (rows) launch_right(M, In_dev, ldi, 0, Out_dev, ldo, 0, rows, mrows, 1, (hipStream_t)stream);
    HIPCHK(h, hipGetLastError());
    return SLAMK_OK;
}

int slamk_stage_left_dev(slamk_handle* h, int which, int row0, int rows, const double* In_dev, int ldi, int cols,
                         double* Out_dev, int ldo, int add_diag, void* stream)
{
    RowSparse M; int mrows; const double* diag;
    if (!h || !In_dev || !Out_dev || row0 < 0 || rows < 0 || cols < 0 || !stage_operator(h, which, &M, &mrows, &diag))
        return SLAMK_EINVAL;
    if (ldi < cols || ldo < cols) return fail(h, SLAMK_EINVAL, "stage_left: leading dimensions too small");
    ENTER(h);
    note_stream(h, stream);
    const int r = std::max(0, std::min(rows, mrows - row0));     // rows past the operator are left to the caller (zero)
    if (r && cols) {
        M.ca += 3 * (size_t)row0; M.cb += 3 * (size_t)row0; M.cs += row0;
        hipStream_t st = (hipStream_t)stream;
        if (add_diag)
            hipLaunchKernelGGL(rowsparse_left_kernel<false>,
                               dim3((cols + 63) / 64, (r + 4 * STRUCT_RPW - 1) / (4 * STRUCT_RPW), 1), dim3(256), 0, st, M,
                               In_dev, ldi, 0LL, Out_dev, ldo, 0LL, diag + row0, 0LL, r, cols, STRUCT_RPW, row0);
        else      // a diagonal column no lane reaches
            hipLaunchKernelGGL(rowsparse_left_kernel<false>,
                               dim3((cols + 63) / 64, (r + 4 * STRUCT_RPW - 1) / (4 * STRUCT_RPW), 1), dim3(256), 0, st, M,
                               In_dev, ldi, 0LL, Out_dev, ldo, 0LL, diag + row0, 0LL, r, cols, STRUCT_RPW, -(1 << 30));
    }
    HIPCHK(h, hipGetLastError());
    return SLAMK_OK;
}

int slamk_stage_eye_minus_kh_dev(slamk_handle* h, const double* K_dev, int ldk, int row0, int rows, double* A_dev, int lda,
                                 void* stream)
{
    if (!h || !K_dev || !A_dev || row0 < 0 || rows < 0) return SLAMK_EINVAL;
    const int nl = max_nl(h), ml = max_ml(h);
    if (ldk < ml || lda < nl) return fail(h, SLAMK_EINVAL, "stage_eye_minus_kh: leading dimensions too small");
    ENTER(h);
    note_stream(h, stream);
    hipStream_t st = (hipStream_t)stream;
    if (rows) {
        hipLaunchKernelGGL(eye_minus_kh_kernel, dim3((nl + 63) / 64, (rows + 4 * STRUCT_RPW - 1) / (4 * STRUCT_RPW), 1),
                           dim3(256), 0, st, h->fv, K_dev, ldk, 0LL, A_dev, lda, 0LL, nl, rows, row0, STRUCT_RPW);
        hipLaunchKernelGGL(eye_minus_kh_xcols_kernel, dim3((rows + 3) / 4, 1), dim3(256), 0, st, h->fv, K_dev, ldk, 0LL,
                           A_dev, lda, 0LL, rows, row0, ml);
    }
    HIPCHK(h, hipGetLastError());
    return SLAMK_OK;
}

// K = W S^-1 and K R = K diag(r) for `rows` rows of W from the outputs L, Li of slamk_cholesky_inverse_dev
// (m x m, m a multiple of 16).  W is overwritten, Y is workspace.
int slamk_stage_gain_dev(slamk_handle* h, double* W_dev, int rows, int m, int ld, const double* L_dev, const double* Li_dev,
                         int ldli, double* Y_dev, double* K_dev, double* KR_dev, const double* r_dev, void* stream)
{
    if (!h || !W_dev || !L_dev || !Li_dev || !Y_dev || !K_dev || !KR_dev || !r_dev || rows < 16 || (rows & 15) || m < 16 ||
        (m & 15))
        return SLAMK_EINVAL;
    if (ld < m || ldli < m) return fail(h, SLAMK_EINVAL, "stage_gain: leading dimensions too small");
    ENTER(h);
    note_stream(h, stream);
    int rc = gain_from_factor(h, W_dev, Y_dev, K_dev, KR_dev, ld, 0, L_dev, Li_dev, ldli, 0, r_dev, 0, rows, m, 1,
                              (hipStream_t)stream);
    return rc ? rc : SLAMK_OK;
}

// slamk_stage_gain_dev with the workspace Y in the library's static device array: no allocation of any kind.
int slamk_stage_gain_static_dev(slamk_handle* h, double* W_dev, int rows, int m, int ld, const double* L_dev,
                                const double* Li_dev, int ldli, double* K_dev, double* KR_dev, const double* r_dev,
                                void* stream)
{
    if (!h || !W_dev || !L_dev || !Li_dev || !K_dev || !KR_dev || !r_dev || rows < 16 || (rows & 15) || m < 16 || (m & 15))
        return SLAMK_EINVAL;
    if (ld < m || ldli < m) return fail(h, SLAMK_EINVAL, "stage_gain_static: leading dimensions too small");
    if ((size_t)rows * (size_t)ld > STATIC_SCRATCH_DOUBLES)
        return fail(h, SLAMK_ECAPACITY, "stage_gain_static: rows * ld = %zu doubles exceed the static workspace of %zu",
                    (size_t)rows * (size_t)ld, STATIC_SCRATCH_DOUBLES);
    ENTER(h);
    note_stream(h, stream);
    double* Y = nullptr;
    HIPCHK(h, static_scratch(&Y));       // cached after the first call on this device; the array's name is never passed on
    // Y needs no clearing: every column block of Y is written in full before the solves read it
    int rc = gain_from_factor(h, W_dev, Y, K_dev, KR_dev, ld, 0, L_dev, Li_dev, ldli, 0, r_dev, 0, rows, m, 1,
                              (hipStream_t)stream);
    return rc ? rc : SLAMK_OK;
}

int slamk_stage_apply_dev(slamk_handle* h, const double* K_dev, int ldk, int32_t* applied_dev, void* stream)
{
    if (!h || !K_dev || ldk < 1) return SLAMK_EINVAL;
    ENTER(h);
    note_stream(h, stream);
    hipStream_t st = (hipStream_t)stream;
    const int nl = max_nl(h), ml = max_ml(h);
    hipLaunchKernelGGL(gain_kx_kernel, dim3((nl + 15) / 16, h->B), dim3(256), 0, st, h->fv, K_dev, ldk, h->dx, ml);
    hipLaunchKernelGGL(gain_apply_kernel, dim3(h->B), dim3(256), 0, st, h->fv, h->dx, applied_dev);
    HIPCHK(h, hipGetLastError());
    return SLAMK_OK;
}

int slamk_buffer_dev(slamk_handle* h, int which, double** ptr, int* rows, int* cols, int* ld)
{
    if (!h || !ptr || !rows || !cols || !ld) return SLAMK_EINVAL;
    const int nl = max_nl(h), ml = max_ml(h);
    switch (which) {
        case 0: *ptr = h->Fm; *rows = nl; *cols = nl; *ld = h->ldn; break;          /* F */
        case 1: *ptr = h->Hm; *rows = ml; *cols = nl; *ld = h->ldn; break;          /* H */
        case 2: *ptr = h->fv.e; *rows = 1; *cols = ml; *ld = h->ldm; break;         /* innovation */
        case 3: *ptr = h->fv.rdiag; *rows = 1; *cols = ml; *ld = h->ldm; break;     /* diag R */
        case 4: *ptr = h->fv.qdiag; *rows = 1; *cols = nl; *ld = h->ldn; break;     /* diag Q */
        case 5: *ptr = h->fv.P; *rows = nl; *cols = nl; *ld = h->ldn; break;        /* P */
        default: return fail(h, SLAMK_EINVAL, "buffer_dev: unknown buffer %d", which);
    }
    return SLAMK_OK;
}

int slamk_cholesky_inverse_dev(double* S_dev, double* L_dev, double* Li_dev, int m, int ld, int32_t* status_dev,
                               void* stream)
{
    if (!S_dev || !L_dev || !Li_dev || m < 16 || (m & 15) || ld < m) return SLAMK_EINVAL;
    static thread_local slamk_handle scratch_handle;   // error text / profiling sink for the handle-free call
    return cholesky_inverse(&scratch_handle, S_dev, L_dev, Li_dev, ld, (long long)ld * m, 1, m, status_dev,
                            (hipStream_t)stream);
}

// ---- one large-map filter sharded by row blocks over the GPUs of a node ------------------------------------
// Rank r owns rows [r bn, (r + 1) bn) of P and of every n-row product (T = P F^T, W = P H^T, K).  The O(nf) filter
// state is replicated: every rank runs the per-feature kernels itself and holds the compact rows of F and H, so the
// products with the Jacobians are local sweeps and the traffic is one all-gather per dense contraction (RCCL over
// xGMI, in place, on the caller's stream):
//     predict:  T_r = P_r F^T -> all-gather of the few rows of T other ranks read -> P_r = F_r T + Q_r
//                                                                                       (calcP, main.cpp:526-530)
//     update :  W_r = P_r H^T -> all-gather W                                           (calcK, :539-552)
//               S = H W + R = L L^T, left-looking by panels of columns: every rank forms and updates an equal share
//               of the rows of the panel, all-gather of the panel, panel factorisation on every rank
//               K_r = W_r L^-T L^-1 -> all-gather K;  A = I - K H (every rank, from the gathered K); all-gather P
//               P_r <- (A_r P) A^T + (K_r R) K^T                                        (Joseph form of updateP, :561-569)
//               dx = K e, gate, mean update (replicated)                                (:228-232)
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// The collective library is opened on first use: a process that never shards a filter does not load it.
static const RcclApi* rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return;
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(lib, "ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GetErrorString) api.lib = lib;
    });
    return api.lib ? &api : nullptr;
}

struct slamk_sharded {
    slamk_handle* h = nullptr;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    slamk_allgather_fn ag = nullptr;
    void* ag_ctx = nullptr;
    int n = 0, m = 0, bn = 0, bm = 0, N = 0, M = 0, r0 = 0, q0 = 0;
    double *P = nullptr, *G1 = nullptr, *G2 = nullptr, *AP = nullptr, *W = nullptr, *K = nullptr, *Y = nullptr,
           *KR = nullptr, *X = nullptr, *Lm = nullptr, *Li = nullptr, *rfull = nullptr, *U = nullptr, *halo = nullptr, *ws = nullptr;
    size_t U_elems = 0, ws_elems = 0;
    int ldx = 0;        // X: M rows x min(M, INV_BS) / 2 columns, intermediates of the diagonal-block inverses
    std::vector<void*> allocs;
};

// rows [lo, hi) of S inside a panel buffer whose first column is column `col0` of S and whose row 0 is row `lo`:
// zero with a unit diagonal (rows the measurement operator does not have)
__global__ void pad_identity_rows_kernel(double* __restrict__ U, int ld, int cols, int col0, int lo, int hi)
{
    const int r = lo + blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (r < hi && c < cols) U[(long long)(r - lo) * ld + c] = (c + col0 == r) ? 1.0 : 0.0;
}

// The rows of T = P F^T that a rank other than their owner reads in P_r = F_r T + Q_r (motion_row,
// structured_kernels.hip.h): rows 0..5 (position and velocity: every row of F has entries on columns 3..5, the position
// rows and rows without a second group on columns 0..2), the bias triple (velocity rows) and, because a feature's
// triple may lie across a block boundary, the first and the last two rows of every block.
constexpr int HALO_ROWS = 13;
__device__ __forceinline__ int halo_row(int t, int p, int bn, int bias0)
{
    // global row of slot t in the pack of rank p, or -1 when that rank does not own the slot's row
    int row;
    if (t < 2) row = p * bn + t;
    else if (t < 4) row = (p + 1) * bn - 4 + t;
    else if (t < 10) row = t - 4;
    else row = bias0 + t - 10;
    if (row < p * bn || row >= (p + 1) * bn) return -1;
    return row;
}
// grid (ceil(N / 256), HALO_ROWS): this rank's slots of T (rows at their global position in G) -> its pack
__global__ void halo_pack_kernel(const double* __restrict__ G, double* __restrict__ pack, int N, int rank, int bn, int bias0)
{
    const int c = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
    if (c >= N) return;
    const int row = halo_row(t, rank, bn, bias0);
    pack[((long long)rank * HALO_ROWS + t) * N + c] = row < 0 ? 0.0 : G[(long long)row * N + c];
}
// grid (ceil(N / 256), HALO_ROWS, world): the gathered packs -> rows of G at their global position
__global__ void halo_unpack_kernel(double* __restrict__ G, const double* __restrict__ pack, int N, int rank, int bn, int bias0)
{
    const int c = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y, p = blockIdx.z;
    if (c >= N || p == rank) return;
    const int row = halo_row(t, p, bn, bias0);
    if (row >= 0) G[(long long)row * N + c] = pack[((long long)p * HALO_ROWS + t) * N + c];
}
__global__ void fill_kernel(double* __restrict__ x, int n, double v)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = v;
}

static int sharded_free(slamk_sharded* s)
{
    for (void* p : s->allocs) (void)hipFree(p);
    s->allocs.clear();
    s->P = nullptr;
    return 0;
}

// In-place all-gather of `full`: this rank's block_elems doubles sit at full + rank * block_elems.
static int sharded_allgather(slamk_sharded* s, double* full, size_t block_elems, hipStream_t st)
{
    if (s->world == 1 && !s->comm) return 0;
    if (s->ag) {
        int rc = s->ag(s->ag_ctx, full, block_elems * sizeof(double), s->rank, s->world, (void*)st);
        return rc ? fail(s->h, SLAMK_EHIP, "sharded: the caller's all-gather returned %d", rc) : 0;
    }
    const RcclApi* api = rccl_api();
    ncclResult_t r = api->AllGather(full + (size_t)s->rank * block_elems, full, block_elems, ncclDouble, s->comm, st);
    if (r != ncclSuccess) return fail(s->h, SLAMK_EHIP, "ncclAllGather: %s", api->GetErrorString(r));
    return 0;
}

static int sharded_check(slamk_sharded* s)
{
    if (!s || !s->h) return SLAMK_EINVAL;
    if (!s->P) return fail(s->h, SLAMK_EINVAL, "sharded: no covariance loaded (slamk_sharded_load_cov_dev)");
    if (s->n != 9 + 3 * s->h->hf[0].nf || s->m != s->h->hf[0].m())
        return fail(s->h, SLAMK_EINVAL, "sharded: the feature set changed since the covariance was loaded");
    return 0;
}

int slamk_nccl_unique_id(void* id)
{
    const RcclApi* api = rccl_api();
    if (!id || !api) return SLAMK_EHIP;
    static_assert(sizeof(ncclUniqueId) == SLAMK_NCCL_ID_BYTES, "id size");
    return api->GetUniqueId(static_cast<ncclUniqueId*>(id)) == ncclSuccess ? SLAMK_OK : SLAMK_EHIP;
}

int slamk_sharded_create(slamk_handle* h, int rank, int world, const void* nccl_id, slamk_allgather_fn allgather, void* ctx,
                         slamk_sharded** out)
{
    if (!h || !out || world < 1 || rank < 0 || rank >= world) return SLAMK_EINVAL;
    if (h->B != 1) return fail(h, SLAMK_EINVAL, "sharded: the handle must hold one filter");
    ENTER(h);
    slamk_sharded* s = new slamk_sharded();
    s->h = h; s->rank = rank; s->world = world; s->ag = allgather; s->ag_ctx = ctx;
    if (!allgather && (world > 1 || nccl_id)) {      // (an id with world == 1 makes a one-rank communicator: same code path)
        const RcclApi* api = rccl_api();
        if (!api) { delete s; return fail(h, SLAMK_EHIP, "sharded: librccl.so could not be loaded"); }
        if (!nccl_id) { delete s; return fail(h, SLAMK_EINVAL, "sharded: world > 1 needs the shared id of slamk_nccl_unique_id"); }
        ncclUniqueId id;
        memcpy(&id, nccl_id, sizeof id);
        ncclResult_t r = api->CommInitRank(&s->comm, world, id, rank);
        if (r != ncclSuccess) { delete s; return fail(h, SLAMK_EHIP, "ncclCommInitRank: %s", api->GetErrorString(r)); }
    }
    *out = s;
    return SLAMK_OK;
}

int slamk_sharded_destroy(slamk_sharded* s)
{
    if (!s) return SLAMK_EINVAL;
    DeviceGuard dg_(s->h->device);
    (void)hipDeviceSynchronize();
    sharded_free(s);
    if (s->comm) (void)rccl_api()->CommDestroy(s->comm);
    delete s;
    return SLAMK_OK;
}

int slamk_sharded_dims(slamk_sharded* s, int* n, int* m, int* rows_per_rank, int* srows_per_rank, int* n_padded, int* m_padded)
{
    if (!s || !s->P) return SLAMK_EINVAL;
    if (n) *n = s->n; if (m) *m = s->m; if (rows_per_rank) *rows_per_rank = s->bn; if (srows_per_rank) *srows_per_rank = s->bm;
    if (n_padded) *n_padded = s->N; if (m_padded) *m_padded = s->M;
    return SLAMK_OK;
}

int slamk_sharded_load_cov_dev(slamk_sharded* s, const double* P_full_dev, int ld, void* stream)
{
    if (!s || !P_full_dev) return SLAMK_EINVAL;
    slamk_handle* h = s->h;
    ENTER(h);
    note_stream(h, stream);
    hipStream_t st = (hipStream_t)stream;
    const int n = 9 + 3 * h->hf[0].nf, m = h->hf[0].m(), g = s->world;
    if (ld < n) return fail(h, SLAMK_EINVAL, "sharded_load_cov: ld < n");
    HIPCHK(h, hipStreamSynchronize(st));
    sharded_free(s);
    s->n = n; s->m = m;
    s->bn = pad16((pad16(n) + g - 1) / g); s->bm = pad16((pad16(m) + g - 1) / g);
    s->N = s->bn * g; s->M = s->bm * g;
    s->r0 = s->rank * s->bn; s->q0 = s->rank * s->bm;
    const size_t N = s->N, M = s->M, bn = s->bn;
    auto alloc = [&](double** p, size_t count) -> int {
        void* q = nullptr;
        HIPCHK(h, hipMalloc(&q, count * sizeof(double)));
        HIPCHK(h, hipMemsetAsync(q, 0, count * sizeof(double), st));
        s->allocs.push_back(q);
        *p = static_cast<double*>(q);
        return 0;
    };
    int rc = 0;
    // U: one panel of S, the rows from its diagonal block down dealt to the ranks in equal padded shares
    s->U_elems = (M + 16 * (size_t)g) * std::min<size_t>(M, chol_outer_block());
    rc |= alloc(&s->P, bn * N); rc |= alloc(&s->G1, N * N); rc |= alloc(&s->G2, N * N);
    rc |= alloc(&s->AP, bn * N); rc |= alloc(&s->W, N * M); rc |= alloc(&s->K, N * M); rc |= alloc(&s->Y, bn * M);
    // (S = H W + R only ever exists one gathered panel at a time, in U)
    s->ldx = inverse_scratch_cols((int)M);
    rc |= alloc(&s->KR, bn * M); rc |= alloc(&s->X, M * (size_t)s->ldx); rc |= alloc(&s->Lm, M * M); rc |= alloc(&s->Li, M * M);
    rc |= alloc(&s->rfull, M); rc |= alloc(&s->U, s->U_elems); rc |= alloc(&s->halo, (size_t)g * HALO_ROWS * N);
    // partial products of the GEMMs whose k range is cut into slices (a rank's rows are few, the k ranges long)
    s->ws_elems = std::max(8 * (s->U_elems / g), 4 * bn * std::min<size_t>(M, INV_BS));
    rc |= alloc(&s->ws, s->ws_elems);
    if (rc) { sharded_free(s); return SLAMK_EHIP; }
    const int hi = std::min(n, s->r0 + s->bn);
    if (hi > s->r0)
        HIPCHK(h, hipMemcpy2DAsync(s->P, N * sizeof(double), P_full_dev + (size_t)s->r0 * ld, (size_t)ld * sizeof(double),
                                   (size_t)n * sizeof(double), hi - s->r0, hipMemcpyDeviceToDevice, st));
    return SLAMK_OK;
}

int slamk_sharded_predict_dev(slamk_sharded* s, const double* qbw_dev, const double* w_dev, const double* acc_dev,
                              const double* dt_dev, void* stream)
{
    { int rc = sharded_check(s); if (rc) return rc; }
    slamk_handle* h = s->h;
    hipStream_t st = (hipStream_t)stream;
    int rc = slamk_stage_motion_dev(h, qbw_dev, w_dev, acc_dev, dt_dev, stream);
    if (!rc) rc = slamk_stage_right_dev(h, 0, s->P, s->N, s->bn, s->G1 + (size_t)s->r0 * s->N, s->N, stream);   // T_r = P_r F^T
    if (rc) return rc;
    ENTER(h);
    if (s->world > 1 || s->comm) {       // the rows of T that ranks other than their owner read
        const int bias0 = 6 + 3 * h->hf[0].nf;
        hipLaunchKernelGGL(halo_pack_kernel, dim3((s->N + 255) / 256, HALO_ROWS), dim3(256), 0, st, s->G1, s->halo, s->N,
                           s->rank, s->bn, bias0);
        HIPCHK(h, hipGetLastError());
        rc = sharded_allgather(s, s->halo, (size_t)HALO_ROWS * s->N, st);
        if (rc) return rc;
        hipLaunchKernelGGL(halo_unpack_kernel, dim3((s->N + 255) / 256, HALO_ROWS, s->world), dim3(256), 0, st, s->G1,
                           s->halo, s->N, s->rank, s->bn, bias0);
        HIPCHK(h, hipGetLastError());
    }
    return slamk_stage_left_dev(h, 0, s->r0, s->bn, s->G1, s->N, s->N, s->P, s->N, 1, stream);                  // P_r = F_r T + Q_r
}

// S = H W + R = L L^T and the inverses of L's diagonal blocks, every rank ending with the whole factor.  Left-looking
// by panels of chol_outer_block() columns: the rows of panel [J0, J1) from its diagonal block down are dealt to the ranks
// in equal shares; a rank forms its rows of S from the gathered W and the compact rows of H (S as a whole never
// exists), subtracts L[rows, 0:J0] L[J0:J1, 0:J0]^T, the panel is all-gathered and factored on every rank (diagonal blocks,
// L21, rank-CHOL_NB updates inside the panel).  Per rank m^3 / (3 world) of the m^3 / 3 FLOP of the trailing updates.
static int sharded_cholesky(slamk_sharded* s, hipStream_t st)
{
    slamk_handle* h = s->h;
    const int M = s->M, g = s->world, OB = chol_outer_block();
    const int mrows = std::min(max_ml(h), M);          // rows the measurement operator has
    for (int J0 = 0; J0 < M; J0 += OB) {
        const int J1 = std::min(M, J0 + OB), ob = J1 - J0, R = M - J0;
        const int chunk = pad16((R + g - 1) / g);
        if ((size_t)g * chunk * ob > s->U_elems)
            return fail(h, SLAMK_EINVAL, "sharded: the block sizes changed since the covariance was loaded");
        const int row0 = J0 + s->rank * chunk, rows = std::max(0, std::min(M, row0 + chunk) - row0);
        double* Ur = s->U + (size_t)s->rank * chunk * ob;
        if (rows > 0) {
            const int rop = std::max(0, std::min(rows, mrows - row0));
            if (rop) {
                RowSparse Hs{h->fv.ha + 3 * (size_t)row0, h->fv.hb + 3 * (size_t)row0, h->fv.hcs + row0, h->ldm, 0};
                hipLaunchKernelGGL(rowsparse_left_kernel<false>,
                                   dim3((ob + 63) / 64, (rop + 4 * STRUCT_RPW - 1) / (4 * STRUCT_RPW), 1), dim3(256), 0, st, Hs,
                                   s->W + J0, M, 0LL, Ur, ob, 0LL, h->fv.rdiag + row0, 0LL, rop, ob, STRUCT_RPW, row0 - J0);
            }
            if (rop < rows)
                hipLaunchKernelGGL(pad_identity_rows_kernel, dim3((ob + 255) / 256, rows - rop), dim3(256), 0, st,
                                   Ur + (size_t)rop * ob, ob, ob, J0, row0 + rop, row0 + rows);
            HIPCHK(h, hipGetLastError());
            if (J0 > 0) {
                ProfScope ps(h, PC_GEMM_CHOL, st);
                HIPCHK(h, launch_gemm_splitk<true>(gargs(s->Lm + (size_t)row0 * M, M, 0, s->Lm + (size_t)J0 * M, M, 0, Ur, ob, 0,
                                                         rows, ob, J0, -1.0, 1.0), s->ws, s->ws_elems, st));
            }
        }
        int rc = sharded_allgather(s, s->U, (size_t)chunk * ob, st);
        if (rc) return rc;
        // the gathered panel is factored where it lies: rows J0 .. M - 1 of columns [J0, J1), leading dimension ob
        rc = chol_outer_panel(h, SymView{s->U, ob, 0, J0, J0}, s->Lm, s->Li, M, 0, 1, M, J0, J1, h->fv.status, st);
        if (rc) return rc;
    }
    if (inverse_scratch_cols(M) > s->ldx)
        return fail(h, SLAMK_EINVAL, "sharded: the block sizes changed since the covariance was loaded");
    return chol_block_inverses(h, s->X, s->ldx, 0, true, s->Lm, s->Li, M, 0, 1, M, st);
}

int slamk_sharded_update_dev(slamk_sharded* s, const double* qbw_dev, const double* alt_dev, const double* src_dev,
                             const double* refl_dev, int32_t* applied_dev, void* stream)
{
    { int rc = sharded_check(s); if (rc) return rc; }
    slamk_handle* h = s->h;
    hipStream_t st = (hipStream_t)stream;
    const size_t N = s->N, M = s->M, bn = s->bn, r0 = s->r0;
    int rc = slamk_stage_measure_dev(h, qbw_dev, alt_dev, src_dev, refl_dev, stream);
    if (rc) return rc;
    ENTER(h);
    rc = slamk_stage_right_dev(h, 1, s->P, s->N, s->bn, s->W + r0 * M, s->M, stream);                          // W_r = P_r H^T
    if (!rc) rc = sharded_allgather(s, s->W, bn * M, st);
    if (rc) return rc;
    // diagonal of R over the padded measurement rows: the operator's own rows, 1 beyond them
    const int ml = std::min(max_ml(h), s->M);
    hipLaunchKernelGGL(fill_kernel, dim3((s->M + 255) / 256), dim3(256), 0, st, s->rfull, s->M, 1.0);
    HIPCHK(h, hipMemcpyAsync(s->rfull, h->fv.rdiag, (size_t)ml * sizeof(double), hipMemcpyDeviceToDevice, st));
    // (Lm and Li are zero above the diagonal since they were allocated: the factorisation writes the lower part and
    // zeros, never anything else, above it)
    rc = sharded_cholesky(s, st);
    if (!rc) rc = gain_from_factor(h, s->W + r0 * M, s->Y, s->K + r0 * M, s->KR, s->M, 0, s->Lm, s->Li, s->M, 0, s->rfull, 0,
                                   s->bn, s->M, 1, st, s->ws, s->ws_elems);                                    // K_r, K_r R
    if (!rc) rc = sharded_allgather(s, s->K, bn * M, st);
    if (!rc) rc = slamk_stage_eye_minus_kh_dev(h, s->K, s->M, 0, s->N, s->G2, s->N, stream);                    // A = I - K H
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(s->G1 + r0 * N, s->P, bn * N * sizeof(double), hipMemcpyDeviceToDevice, st));
    rc = sharded_allgather(s, s->G1, bn * N, st);
    if (rc) return rc;
    // entries (i, j) and (j, i) of the propagated covariance come from different ranks' products: the lower triangle
    // is mirrored so that the update works on an exactly symmetric P on every rank (the reference symmetrises after
    // every update, src/slam/main.cpp:567)
    hipLaunchKernelGGL(mirror_lower_inplace_kernel, dim3((s->N + 31) / 32, (s->N + 31) / 32), dim3(256), 0, st, s->G1, s->N, s->N);
    const double* Ar = s->G2 + r0 * N;
    // P_r <- (A_r P) A^T + (K_r R) K^T
    HIPCHK(h, (launch_gemm<false, EPI_NONE, false, TRI_NONE>(gargs(Ar, s->N, 0, s->G1, s->N, 0, s->AP, s->N, 0, s->bn, s->N, s->N), 1, st)));
    HIPCHK(h, (launch_gemm<true, EPI_NONE, false, TRI_NONE>(gargs(s->AP, s->N, 0, s->G2, s->N, 0, s->P, s->N, 0, s->bn, s->N, s->N), 1, st)));
    HIPCHK(h, (launch_gemm<true, EPI_NONE, false, TRI_NONE>(gargs(s->KR, s->M, 0, s->K, s->M, 0, s->P, s->N, 0, s->bn, s->N, s->M, 1.0, 1.0), 1, st)));
    return slamk_stage_apply_dev(h, s->K, s->M, applied_dev, stream);                                           // dx = K e, gate
}

int slamk_sharded_gather_cov_dev(slamk_sharded* s, double* P_full_dev, int ld, void* stream)
{
    { int rc = sharded_check(s); if (rc) return rc; }
    if (!P_full_dev || ld < s->n) return SLAMK_EINVAL;
    slamk_handle* h = s->h;
    ENTER(h);
    note_stream(h, stream);
    hipStream_t st = (hipStream_t)stream;
    const size_t N = s->N, bn = s->bn;
    HIPCHK(h, hipMemcpyAsync(s->G1 + (size_t)s->r0 * N, s->P, bn * N * sizeof(double), hipMemcpyDeviceToDevice, st));
    int rc = sharded_allgather(s, s->G1, bn * N, st);
    if (rc) return rc;
    // the row blocks of the Joseph update come from different ranks: hand out the symmetric completion of the lower part
    hipLaunchKernelGGL(mirror_lower_inplace_kernel, dim3((s->N + 31) / 32, (s->N + 31) / 32), dim3(256), 0, st, s->G1, s->N, s->N);
    HIPCHK(h, hipMemcpy2DAsync(P_full_dev, (size_t)ld * sizeof(double), s->G1, N * sizeof(double), (size_t)s->n * sizeof(double),
                               s->n, hipMemcpyDeviceToDevice, st));
    return SLAMK_OK;
}

// ---- stream pre-processing filters, batched --------------------------------------------------------------
int slamk_pre_euler2qbw_dev(const double* euler_dev, double* q_dev, int n, void* stream)
{
    if (n < 0 || (n && (!euler_dev || !q_dev))) return SLAMK_EINVAL;
    if (n) hipLaunchKernelGGL(euler2qbw_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, euler_dev, q_dev, n);
    return hipGetLastError() == hipSuccess ? SLAMK_OK : SLAMK_EHIP;
}

int slamk_pre_rotate_dev(const double q[4], const double* v_dev, double* out_dev, int n, void* stream)
{
    if (!q || n < 0 || (n && (!v_dev || !out_dev))) return SLAMK_EINVAL;
    if (n) hipLaunchKernelGGL(rotate_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, q[0], q[1], q[2],
                              q[3], v_dev, out_dev, n);
    return hipGetLastError() == hipSuccess ? SLAMK_OK : SLAMK_EHIP;
}

int slamk_pre_img2body_dev(const double* px_dev, double* out_dev, int n, void* stream)
{
    if (n < 0 || (n && (!px_dev || !out_dev))) return SLAMK_EINVAL;
    if (n) hipLaunchKernelGGL(img2body_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, px_dev, out_dev, n);
    return hipGetLastError() == hipSuccess ? SLAMK_OK : SLAMK_EHIP;
}

int slamk_pre_fir_dev(const double* coeffs_dev, int order, const double* x_dev, double* y_dev, int n, void* stream)
{
    if (order < 1 || n < 0 || !coeffs_dev || (n && (!x_dev || !y_dev)) || x_dev == y_dev) return SLAMK_EINVAL;
    if (n) hipLaunchKernelGGL(fir_kernel, dim3((3 * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, coeffs_dev, x_dev,
                              y_dev, order, n);
    return hipGetLastError() == hipSuccess ? SLAMK_OK : SLAMK_EHIP;
}

// ---- introspection -----------------------------------------------------------------------------------
int slamk_set_debug(slamk_handle* h, int on)
{
    if (!h) return SLAMK_EINVAL;
    ENTER_HOST(h);
    h->debug = on != 0;
    for (auto& g : h->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
    h->graphs.clear();                 // captured sequences were recorded without (or with) the dense F, H, L kernels
    return ensure_workspaces(h);
}

int slamk_debug_get(slamk_handle* h, int b, double* H, double* L, double* K, double* e)
{
    CHECK_B(h, b);
    if ((H || L) && !h->debug) return fail(h, SLAMK_EINVAL, "debug_get: H and L are only kept after slamk_set_debug(h, 1)");
    if ((K || L) && h->prm.update_form == 2)
        return fail(h, SLAMK_EINVAL, "debug_get: the block-sequential update (update_form 2) forms neither K nor L");
    ENTER_HOST(h);
    const int n = 9 + 3 * h->hf[b].nf, m = h->hf[b].m();
    const size_t ldn = h->ldn, ldm = h->ldm;
    // The library keeps the altitude row last; H, K and e go out in the reference's order (altitude first).
    std::vector<double> tmp;
    if (H) {
        tmp.resize((size_t)m * n);
        HIPCHK(h, hipMemcpy2D(tmp.data(), n * 8, h->Hm + b * ldm * ldn, ldn * 8, n * 8, m, hipMemcpyDeviceToHost));
        for (int q = 0; q < m; ++q)
            memcpy(H + (size_t)q * n, tmp.data() + (size_t)(q == 0 ? m - 1 : q - 1) * n, (size_t)n * 8);
    }
    if (L) HIPCHK(h, hipMemcpy2D(L, m * 8, h->L + b * ldm * ldm, ldm * 8, m * 8, m, hipMemcpyDeviceToHost));
    if (K) {
        tmp.resize((size_t)n * m);
        HIPCHK(h, hipMemcpy2D(tmp.data(), m * 8, h->K + b * ldn * ldm, ldm * 8, m * 8, n, hipMemcpyDeviceToHost));
        for (int a = 0; a < n; ++a) {
            K[(size_t)a * m] = tmp[(size_t)a * m + m - 1];
            memcpy(K + (size_t)a * m + 1, tmp.data() + (size_t)a * m, (size_t)(m - 1) * 8);
        }
    }
    if (e) {
        tmp.resize(m);
        HIPCHK(h, hipMemcpy(tmp.data(), h->fv.e + b * ldm, m * 8, hipMemcpyDeviceToHost));
        e[0] = tmp[m - 1];
        memcpy(e + 1, tmp.data(), (size_t)(m - 1) * 8);
    }
    return SLAMK_OK;
}

int slamk_debug_get_F(slamk_handle* h, int b, double* F)
{
    CHECK_B(h, b);
    if (!h->debug) return fail(h, SLAMK_EINVAL, "debug_get_F: F is only kept after slamk_set_debug(h, 1)");
    ENTER_HOST(h);
    const int n = 9 + 3 * h->hf[b].nf;
    const size_t ldn = h->ldn;
    HIPCHK(h, hipMemcpy2D(F, n * 8, h->Fm + b * ldn * ldn, ldn * 8, n * 8, n, hipMemcpyDeviceToHost));
    return SLAMK_OK;
}

int slamk_set_block_sizes(int chol_outer_block_cols, int inverse_block_rows)
{
    if (chol_outer_block_cols < 0 || inverse_block_rows < 0) return SLAMK_EINVAL;
    if (chol_outer_block_cols) chol_outer_block_ref() = round_outer_block(chol_outer_block_cols);
    if (inverse_block_rows) inv_block_rows_ref() = round_inverse_block(inverse_block_rows);
    return SLAMK_OK;
}

int slamk_dgemm_dev(int trans_b, int M, int N, int K, double alpha, const double* A_dev, int lda, long long strideA,
                    const double* B_dev, int ldb, long long strideB, double beta, double* C_dev, int ldc,
                    long long strideC, int batch, void* stream)
{
    if ((M | N | K) & 15) return SLAMK_EINVAL;
    GemmArgs g = gargs(A_dev, lda, strideA, B_dev, ldb, strideB, C_dev, ldc, strideC, M, N, K, alpha, beta);
    hipError_t e = trans_b ? launch_gemm<true, EPI_NONE, false, TRI_NONE>(g, batch, (hipStream_t)stream)
                           : launch_gemm<false, EPI_NONE, false, TRI_NONE>(g, batch, (hipStream_t)stream);
    return e == hipSuccess ? SLAMK_OK : SLAMK_EHIP;
}

int slamk_mfma_f64_probe(int iters, double* tflops)
{
    if (iters < 1 || !tflops) return SLAMK_EINVAL;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return SLAMK_EHIP;
    double* out = nullptr;
    if (hipMalloc(&out, 8) != hipSuccess) return SLAMK_EHIP;
    const int blocks = prop.multiProcessorCount * 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, 0, out, 16);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 8 * 2.0 * 16 * 16 * 4;
    *tflops = flops / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(out);
    return SLAMK_OK;
}

int slamk_profile_enable(slamk_handle* h, int on)
{
    if (!h) return SLAMK_EINVAL;
    h->prof_mask = on == 0 ? 0u : (on == 1 ? ~0u : (unsigned)on >> 1);
    return SLAMK_OK;
}

int slamk_profile_read(slamk_handle* h, double* ms, int* count, int ncat)
{
    if (!h || !ms || !count || ncat < PC_COUNT) return SLAMK_EINVAL;
    ENTER(h);
    HIPCHK(h, hipDeviceSynchronize());
    for (int i = 0; i < ncat; ++i) { ms[i] = 0; count[i] = 0; }
    for (auto& r : h->prof_recs) {
        float t = 0;
        HIPCHK(h, hipEventElapsedTime(&t, r.a, r.b));
        ms[r.cat] += t;
        count[r.cat] += 1;
        h->prof_pool.push_back(r.a);
        h->prof_pool.push_back(r.b);
    }
    h->prof_recs.clear();
    return SLAMK_OK;
}

double slamk_flops_predict(int n)
{
    (void)n;
    return 0.0;        // the predict products run on the compact rows of F: no MFMA work
}

double slamk_flops_update_stage(int n, int m, int update_form, int stage)
{
    const int nl = pad16(n), ml = pad16(m);
    double f = 0;
    if (stage == 0 ? use_fused_solver(nl, m) : use_fused_joseph(nl, m)) {
        // 16^3-tile products (2 * 16^3 flop each); the altitude row costs no tile when it is taken as a border
        const int MT = tiled_rows(m) / 16, NT = nl / 16;
        double tp = 0;
        if (stage == 0) {          // fused gain solver: tile Cholesky and the two triangular solves
            for (int j = 0; j < MT; ++j) {
                const int w = MT - j - 1;
                tp += w + w * (w + 1) / 2;
            }
            tp += 2.0 * NT * (MT * (MT - 1) / 2 + MT);
        } else {                   // fused covariance update: A P, then the lower tiles of (A P) A^T + K R K^T
            tp += (double)NT * NT * NT;
            if (update_form == 0) tp += NT * (NT + 1) / 2 * (double)(NT + MT);
        }
        return tp * 8192.0;
    }
    if (stage == 0) {
        const int NB = CHOL_NB, OB = chol_outer_block();
        for (int J0 = 0; J0 < ml; J0 += OB) {
            const int J1 = std::min(ml, J0 + OB);
            for (int j0 = J0; j0 < J1; j0 += NB) {
                const int nb = std::min(NB, ml - j0), rem = ml - j0 - nb;
                if (rem <= 0) break;
                f += gemm_flops(rem, nb, nb, false, TRI_B_KLE_N);
                if (J1 - (j0 + nb) > 0) f += gemm_flops(rem, J1 - (j0 + nb), nb, true, TRI_NONE);
            }
            if (ml - J1 > 0) f += gemm_flops(ml - J1, ml - J1, J1 - J0, true, TRI_NONE);
        }
        for (int s = NB; s < std::min(ml, INV_BS); s *= 2) {
            const int nfull = ml / (2 * s), r = ml - nfull * 2 * s;
            f += nfull * (gemm_flops(s, s, s, false, TRI_B_KGE_N) + gemm_flops(s, s, s, false, TRI_A_KLE_M));
            if (r > s) f += gemm_flops(r - s, s, s, false, TRI_B_KGE_N) + gemm_flops(r - s, s, r - s, false, TRI_A_KLE_M);
        }
        for (int c0 = 0; c0 < ml; c0 += INV_BS) {         // the two block triangular solves
            const int wj = std::min(INV_BS, ml - c0), c1 = c0 + wj;
            if (c0 > 0) f += gemm_flops(nl, wj, c0, false, TRI_NONE);
            f += gemm_flops(nl, wj, wj, false, TRI_B_KLE_N);
            if (c1 < ml) f += gemm_flops(nl, wj, ml - c1, false, TRI_NONE);
            f += gemm_flops(nl, wj, wj, false, TRI_B_KGE_N);
        }
        return f;
    }
    f += gemm_flops(nl, nl, nl, false, TRI_NONE);              // A P
    if (update_form == 0) {
        f += gemm_flops(nl, nl, nl, true, TRI_NONE);           // (A P) A^T, lower tiles
        f += gemm_flops(nl, nl, ml, true, TRI_NONE);           // (K R) K^T, lower tiles
    }
    return f;
}

double slamk_flops_update(int n, int m, int update_form)
{
    return slamk_flops_update_stage(n, m, update_form, 0) + slamk_flops_update_stage(n, m, update_form, 1);
}

}  // extern "C"

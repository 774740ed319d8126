/*
 * slamk.h -- C ABI of the MI355X EKF predict/update library (libslamk.so).
 *
 * The reference estimator (presocratics/slam, src/slam) has no FFI layer: its hot path
 * is a set of C++ free functions and class methods called once per sensor time step from
 * main()'s while loop (src/slam/main.cpp:150-236).  This header is the boundary a
 * maintainer binds instead; every entry point names the reference code it replaces.
 *
 * Conventions
 *   - fp64 everywhere; quaternions are scalar-last (src/ARC/Quaternion.cpp:32-47).
 *   - A handle owns `batch` independent filters of at most `max_features` features each.
 *     State vector order per filter: X(3) V(3) feature_0(3) .. feature_{nf-1}(3) bias(3),
 *     n = 9 + 3 nf (src/ARC/states.cpp:24-37,267-270).
 *   - Measurement vector order per filter: altitude, then per feature slot
 *     current(2) initial(2) [reflection(2) if the feature was initialised with one],
 *     m = 1 + 6 rf + 4 nrf (src/ARC/view.cpp:3-26, src/slam/main.cpp:353).
 *   - Pointers are caller-owned HOST memory unless the name ends in _dev.
 *   - Every function returns 0 or a negative SLAMK_E* code and never exits the process;
 *     slamk_last_error() gives the text.  A handle may be used from one thread at a time.
 *   - Streams.  The *_dev entry points only queue work on the caller's stream.  The host-pointer entry points
 *     (set/get, slamk_frame, slamk_predict, slamk_update, slamk_debug_get*) are synchronous; on entry they wait
 *     for the stream of the handle's most recent *_dev call, so they may follow such a call directly.  A caller
 *     that spreads *_dev calls of one handle over SEVERAL streams orders them itself (events) and calls slamk_sync
 *     on each before a host-pointer entry point.
 *   - Devices.  Handle entry points make the handle's device current for the call and restore the caller's
 *     current device before they return.  The handle-free entry points (slamk_dgemm_dev, slamk_pre_*,
 *     slamk_cholesky_inverse_dev, slamk_mfma_f64_probe) run on the device that is current in the calling thread;
 *     their pointers must belong to it.  slamk_stage_gain_static_dev takes a handle and runs on the HANDLE's device
 *     with that device's static workspace: W, L, Li, K, KR and r must live there, so the L / Li it is given must
 *     come from a slamk_cholesky_inverse_dev call made while the handle's device was current.
 */
#ifndef SLAMK_H
#define SLAMK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct slamk_handle slamk_handle;

enum {
    SLAMK_OK = 0,
    SLAMK_EINVAL = -1,      /* bad argument / size */
    SLAMK_EHIP = -2,        /* HIP runtime error */
    SLAMK_ENOMATCH = -3,    /* "Cannot find match."            src/slam/main.cpp:372-375 */
    SLAMK_EREFL = -4,       /* reflection rows of H, R, e disagree: the reference aborts in
                               cv::Mat arithmetic here (src/slam/main.cpp:211-218,353; src/ARC/view.cpp:17,39) */
    SLAMK_ECAPACITY = -5,   /* more features than max_features */
    SLAMK_ENOTSPD = -6      /* S = H P H^T + R not positive definite: the Cholesky gain does not exist (the
                               reference's LU solve, main.cpp:549, goes on silently); reported by slamk_sync */
};

/* Compile-time constants of the reference, as run-time parameters with the same defaults:
 * Q0,R0 src/slam/main.cpp:115-116; qbias src/slam/main.h:25; p0 src/ARC/states.cpp:136,164;
 * pinit src/slam/main.cpp:44; alt_override :177; gate :232; d_min,d_max src/ARC/states.cpp:91-92;
 * dinit src/ARC/feature.cpp:20. */
typedef struct {
    double Q0, R0, qbias, p0, pinit;
    double alt_override;
    int32_t use_alt_override;
    double gate, d_min, d_max, dinit;
    /* 0: gain by Cholesky of S, covariance by the Joseph form (I-KH)P(I-KH)^T + K R K^T.
     * 1: gain by Cholesky of S, covariance as the reference writes it, (I-KH)P then (P+P^T)/2
     *    (src/slam/main.cpp:561-569).  Both equal the reference's LU form up to rounding.
     * 2: block-sequential update, n <= 160 only (SLAMK_EINVAL beyond): the measurement rows are processed 16 at
     *    a time against the covariance held on chip (W_c = P H_c^T, S_c = H_c W_c + R_c, K_c = W_c S_c^-1,
     *    P -= K_c W_c^T, dx += K_c (e_c - H_c dx)).  R is diagonal and H fixed during an update, so posterior
     *    and total correction dx = K e are those of forms 0 and 1 up to rounding; the gate uses the total dx.
     *    It is not the reference's sequence of operations (one m x m system) and is therefore not the default;
     *    slamk_debug_get has no K or L to return in this form. */
    int32_t update_form;
} slamk_params;

/* One image-feature observation, include/featureIO.h:31-37 (`projection`). */
typedef struct {
    int32_t id;
    double sx, sy;     /* source */
    double rx, ry;     /* reflection; (0,0) = none */
} slamk_match;

void slamk_default_params(slamk_params* p);

/* Allocates device state for `batch` filters on HIP device `device`: per filter the covariance twice
 * (ldn^2), the gain (ldn x ldm) and O(max_features) vectors, ldn = 9 + 3 max_features and ldm = 1 + 6 max_features
 * rounded up to 16.  Filters small enough for the per-filter kernels (n <= 128, m <= 176) never need more; the
 * first predict/update/step/set_debug call that meets a larger layout (or debug mode) allocates the GEMM-path
 * workspaces (about 3 ldm^2 + 3 ldn ldm + 2 ldn^2 doubles per filter) and synchronises the device once.
 * Every filter starts like main() does: X=V=bias=0, no features,
 * P = diag(pinit x3, 1 x3, pinit x3)  (src/slam/main.cpp:121-123,130). */
int slamk_create(const slamk_params* p, int max_features, int batch, int device, slamk_handle** out);
int slamk_destroy(slamk_handle* h);
const char* slamk_last_error(const slamk_handle* h);

/* ---- state access (replaces direct member access to States / Feature / cv::Mat P) ---- */
/* Any of X, V, bias may be NULL = leave unchanged.  b = filter index. */
int slamk_set_state(slamk_handle* h, int b, const double* X, const double* V, const double* bias);
int slamk_get_state(slamk_handle* h, int b, double* X, double* V, double* bias);
/* P is n x n row-major with leading dimension ld, n must equal 9 + 3 nf of filter b. */
int slamk_set_cov(slamk_handle* h, int b, const double* P, int n, int ld);
int slamk_get_cov(slamk_handle* h, int b, double* P, int n, int ld);
/* Feature containers as AoS arrays of nf entries: id[nf], pib[3nf], anchor[3nf], q0[4nf],
 * pib0[2nf], is_ref[nf]  (include/feature.h:15-25,217-220).  Resizes the filter to nf features;
 * the covariance is NOT touched (follow with slamk_set_cov). */
int slamk_set_features(slamk_handle* h, int b, int nf, const int32_t* id, const double* pib,
                       const double* anchor, const double* q0, const double* pib0, const uint8_t* is_ref);
/* Any output pointer may be NULL.  *nf receives the feature count. */
int slamk_get_features(slamk_handle* h, int b, int* nf, int32_t* id, double* pib, double* anchor,
                       double* q0, double* pib0, uint8_t* is_ref);
int slamk_num_features(slamk_handle* h, int b);
/* Measurement dimension m of filter b for its current feature set. */
int slamk_num_rows(slamk_handle* h, int b);

/* ---- image frame: the `if (u & UPDATE_IMG)` branch, src/slam/main.cpp:167-175 -------- */
/* old_pos = X; inverse depths clamped to [1/d_max, 1/d_min]; association of `matches` with the
 * tracked ids: new ids take the first unmatched slot, else are appended; unmatched features are
 * erased; P rows/cols are inserted / reset / deleted to follow (States::update_features,
 * src/ARC/states.cpp:88-213; Feature::initialize, src/ARC/feature.cpp:48-91).  Slot order and
 * ids are bit-exact with the reference.  The frame's observations are kept, in slot order, as the
 * measurement of the next slamk_update for this filter.  b = -1 applies the same frame to every
 * filter of the batch (they must hold the same ids). */
int slamk_frame(slamk_handle* h, int b, const slamk_match* matches, int nmatch, const double quat[4]);
/* The frames of ALL filters of the batch in one call (same reference code as slamk_frame), ordered on `stream` and
 * without host synchronisation: the association is planned on the host, uploaded in one copy and applied by a fixed
 * number of launches, whatever the batch.  match_off == NULL: the nmatch entries of `matches` are the frame of every
 * filter and quat holds 4 doubles.  Otherwise filter b's frame is matches[match_off[b] .. match_off[b + 1]) (nmatch
 * is ignored; an empty range is a frame without features) and quat holds [batch][4].  matches, match_off and quat
 * are HOST memory and may be reused as soon as the call returns.  When a frame does not fit (SLAMK_ECAPACITY)
 * nothing has been queued for any filter. */
int slamk_frame_batch_async(slamk_handle* h, const slamk_match* matches, int nmatch, const int32_t* match_off,
                            const double* quat, void* stream);
/* Replaces only the observation values (not ids / reflection presence) of the frame kept for
 * filter b: src2/refl2 are [nf][2] in slot order.  For Monte-Carlo batches that share one
 * association but carry independent pixel noise. */
int slamk_set_frame_values(slamk_handle* h, int b, const double* src2, const double* refl2);

/* ---- predict: src/slam/main.cpp:192-194,204-206 --------------------------------------- */
/* F = jacobianMotionModel(mu) at the pre-predict state (:241-341); mu += dynamics(mu) dt
 * (States::dynamics, src/ARC/states.cpp:222-263); Q = initQ (:430-440); P = F P F^T + Q
 * (calcP, :526-530, with G = I).  Per filter b: qbw[4b..], w[3b..], acc[3b..], dt[b]. */
int slamk_predict(slamk_handle* h, const double* qbw, const double* w, const double* acc, const double* dt);

/* ---- update: src/slam/main.cpp:198-202,209-233 ------------------------------------------ */
/* measurementModel + jacobianH (:348-408,:575-624; Feature::pib0Hat/ppbHat,
 * src/ARC/feature.cpp:143-198) at the post-predict state, with pib0Hat at old_pos;
 * R = initR (:449-476); S = H P H^T + R; K = P H^T S^-1 (calcK, :539-552);
 * covariance update (updateP, :561-569; see slamk_params.update_form);
 * dx = K (meas - hmu) (View subtract/toMat, src/ARC/view.cpp:3-47); gated mean update
 * `if (sum|V - dV| < gate) mu += dx` (:228-232, States(const Mat&), src/ARC/states.cpp:24-37).
 * qbw[4b..]; alt[b] is ignored when use_alt_override (:177).  The observations are the ones
 * kept by slamk_frame.  applied[b] (may be NULL) = 1 if the mean update passed the gate, else 0. */
int slamk_update(slamk_handle* h, const double* qbw, const double* alt, int* applied);

/* ---- device-pointer variants: inputs already in HBM, launched on `stream` (a hipStream_t),
 * no host synchronisation.  Layouts: qbw_dev [B][4], w_dev/acc_dev [B][3], dt_dev/alt_dev [B],
 * src_dev/refl_dev [B][2][max_features] (component-major per filter, slot order),
 * applied_dev int32 [B] (may be NULL).  src_dev/refl_dev NULL = use the kept frame. */
int slamk_frame_begin_dev(slamk_handle* h, void* stream);   /* old_pos = X and depth clamp, all filters */
int slamk_predict_dev(slamk_handle* h, const double* qbw_dev, const double* w_dev, const double* acc_dev,
                      const double* dt_dev, void* stream);
int slamk_update_dev(slamk_handle* h, const double* qbw_dev, const double* alt_dev, const double* src_dev,
                     const double* refl_dev, int32_t* applied_dev, void* stream);
/* One loop iteration with an image for every filter of the batch (src/slam/main.cpp:167-233): frame
 * begin, predict and update in one call, arguments as for the three calls above.  With
 * slamk_set_pipeline_groups(h, g), g > 1, the batch is cut into g ranges of filters that run the chain
 * on internal streams (forked from `stream`, joined into it before the call returns), so kernels of
 * different ranges overlap; results are identical to the three separate calls.  When slamk_frame (or
 * slamk_frame_begin_dev) has already begun the frame of every filter since the last predict, the call does not
 * run the frame begin again: a state override in between (UPDATE_POS, main.cpp:180-185) leaves old_pos at the
 * value taken before it, as the reference does (:168). */
int slamk_step_dev(slamk_handle* h, const double* qbw_dev, const double* w_dev, const double* acc_dev,
                   const double* dt_dev, const double* alt_dev, const double* src_dev,
                   const double* refl_dev, int32_t* applied_dev, void* stream);
int slamk_set_pipeline_groups(slamk_handle* h, int groups);
/* Graph mode: slamk_step_dev captures its launch sequence into a hipGraph the first time it sees a set of
 * argument pointers and a feature layout, and replays it afterwards (one submission per iteration instead
 * of about ten launches: what a single small filter's latency is made of).  `stream` must not be the null
 * stream; stage timing (slamk_profile_enable) turns graph mode off for the calls it covers. */
int slamk_set_graph_mode(slamk_handle* h, int on);
/* Waits for `stream` and returns SLAMK_EREFL / SLAMK_ENOTSPD / SLAMK_OK from the per-filter status words: the
 * error text names the first flagged filter and the number of flagged filters, and the words are cleared. */
int slamk_sync(slamk_handle* h, void* stream);
/* Copies the per-filter status words (bit 0: reflection rows disagree, bit 1: S not positive definite) to
 * status[batch] without clearing them; call it before slamk_sync to see every flagged filter. */
int slamk_read_status(slamk_handle* h, int32_t* status);

/* ---- stage-level entry points: the pieces a row-block sharded large-map filter needs around its
 * own distributed products (slam_amd/sharded.py).  Device pointers, asynchronous on `stream`. ----- */
/* jacobianMotionModel (main.cpp:241-341) into the handle's dense F, initQ diagonal (:430-440) and
 * mu += dynamics(mu) dt (states.cpp:222-263); no covariance product. */
int slamk_stage_motion_dev(slamk_handle* h, const double* qbw_dev, const double* w_dev, const double* acc_dev,
                           const double* dt_dev, void* stream);
/* measurementModel / jacobianH / initR (main.cpp:348-408,449-476,575-624): innovation, diag R, dense H. */
int slamk_stage_measure_dev(slamk_handle* h, const double* qbw_dev, const double* alt_dev, const double* src_dev,
                            const double* refl_dev, void* stream);
/* Row-block products with the Jacobians in compact form (filter 0 of the handle; slamk_stage_motion_dev /
 * slamk_stage_measure_dev must have run).  which: 0 = F, 1 = H.
 *   right: Out[rows x mrows] = In[rows x n] * M^T                      (T_r = P_r F^T, W_r = P_r H^T)
 *   left : Out[rows x cols]  = M[row0 .. row0+rows, :] * In[n x cols]  (+ diag Q / R on the operator's diagonal)
 *   eye_minus_kh: A[rows x n] = I[row0 .., :] - K[rows x m] H          (src/slam/main.cpp:563-565)
 *   gain : K = W S^-1 by two block triangular solves, KR = K diag(r), for a row block of W (overwritten)   (src/slam/main.cpp:545-550)
 * Rows past the operator (padding of a row-block layout) are not written. */
int slamk_stage_right_dev(slamk_handle* h, int which, const double* In_dev, int ldi, int rows, double* Out_dev, int ldo,
                          void* stream);
int slamk_stage_left_dev(slamk_handle* h, int which, int row0, int rows, const double* In_dev, int ldi, int cols,
                         double* Out_dev, int ldo, int add_diag, void* stream);
int slamk_stage_eye_minus_kh_dev(slamk_handle* h, const double* K_dev, int ldk, int row0, int rows, double* A_dev,
                                 int lda, void* stream);
int slamk_stage_gain_dev(slamk_handle* h, double* W_dev, int rows, int m, int ld, const double* L_dev,
                         const double* Li_dev, int ldli, double* Y_dev, double* K_dev, double* KR_dev,
                         const double* r_dev, void* stream);
/* No counterpart in the reference.  slamk_stage_gain_dev for callers that cannot allocate device memory (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the workspace Y is taken from an
 * array that lives inside the library's code object (32 MiB per GPU, held until the process ends) instead of from
 * the caller, and the call allocates and frees nothing.  Same arguments and results otherwise.  Limit:
 * rows * ld <= 4 Mi doubles (SLAMK_ECAPACITY beyond).  All calls on one GPU share that array: calls that overlap
 * on one GPU (different streams or threads) race. */
int slamk_stage_gain_static_dev(slamk_handle* h, double* W_dev, int rows, int m, int ld, const double* L_dev,
                                const double* Li_dev, int ldli, double* K_dev, double* KR_dev, const double* r_dev,
                                void* stream);
/* dx = K e with the caller's gain K (n x m, leading dimension ldk), gate and mean update (main.cpp:228-232). */
int slamk_stage_apply_dev(slamk_handle* h, const double* K_dev, int ldk, int32_t* applied_dev, void* stream);
/* Device address and shape of one of the handle's matrices (filter 0): which = 0 F, 1 H, 2 innovation e,
 * 3 diag R, 4 diag Q, 5 P.  rows/cols are the padded live dimensions, ld the leading dimension. */
int slamk_buffer_dev(slamk_handle* h, int which, double** ptr, int* rows, int* cols, int* ld);
/* Cholesky of the lower triangle of S (m x m, m a multiple of 16) into L, and into Li the inverses of L's
 * diagonal blocks of 4096 rows (all of inv(L) when m <= 4096); both lower, their strict upper triangles must be
 * zero on entry.  S is overwritten.  slamk_stage_gain_dev takes the pair (L, Li).  status_dev (may be NULL): one
 * int32 in device memory that gets bit 1 set when S is not positive definite (the caller zeroes and reads it). */
int slamk_cholesky_inverse_dev(double* S_dev, double* L_dev, double* Li_dev, int m, int ld, int32_t* status_dev,
                               void* stream);

/* ---- one large-map filter, covariance sharded by row blocks over the GPUs of a node (one process per GPU) ----
 * No counterpart in the reference (a single-threaded program); the arithmetic is calcP / calcK / updateP (Joseph
 * form) / the gated mean update of src/slam/main.cpp:526-530,539-552,561-569,228-232 on a row-distributed P.
 * `h` is a handle with batch 1 that every rank creates and feeds with the SAME frames and sensors (the O(nf) state
 * is replicated); rank r owns rows [r bn, (r + 1) bn) of P.  Collectives: RCCL all-gathers on the caller's stream,
 * through a communicator the library creates from the id of slamk_nccl_unique_id (obtained on one rank and handed to
 * the others by the caller), or through the caller's own in-place all-gather `allgather` (non-NULL: `nccl_id` is
 * ignored), which must gather `bytes_per_rank` bytes per rank inside `buf_dev` (rank r's part at offset
 * r * bytes_per_rank), ordered on `stream`, and return 0. */
#define SLAMK_NCCL_ID_BYTES 128
typedef struct slamk_sharded slamk_sharded;
typedef int (*slamk_allgather_fn)(void* ctx, void* buf_dev, size_t bytes_per_rank, int rank, int world, void* stream);
int slamk_nccl_unique_id(void* id /* SLAMK_NCCL_ID_BYTES bytes */);
int slamk_sharded_create(slamk_handle* h, int rank, int world, const void* nccl_id, slamk_allgather_fn allgather, void* ctx,
                         slamk_sharded** out);
int slamk_sharded_destroy(slamk_sharded* s);
/* Distributes a full n x n covariance (device memory, the same on every rank): keeps this rank's rows and allocates
 * the rank's workspaces for the current feature set and block sizes (call again after the feature set or
 * slamk_set_block_sizes changes). */
int slamk_sharded_load_cov_dev(slamk_sharded* s, const double* P_full_dev, int ld, void* stream);
int slamk_sharded_predict_dev(slamk_sharded* s, const double* qbw_dev, const double* w_dev, const double* acc_dev,
                              const double* dt_dev, void* stream);
/* All-gathers per call: W (n x m), every panel of S = H W + R as it is factored (m^2 / 2 doubles in all: the rows of a
 * panel are formed and updated by the ranks in equal shares, then factored on every rank), K (n x m), P (n x n).
 * slamk_sharded_predict_dev gathers 13 rows of P F^T per rank. */
int slamk_sharded_update_dev(slamk_sharded* s, const double* qbw_dev, const double* alt_dev, const double* src_dev,
                             const double* refl_dev, int32_t* applied_dev, void* stream);
/* The full n x n covariance on every rank (one all-gather). */
int slamk_sharded_gather_cov_dev(slamk_sharded* s, double* P_full_dev, int ld, void* stream);
/* Dimensions of the sharded layout; any pointer may be NULL. */
int slamk_sharded_dims(slamk_sharded* s, int* n, int* m, int* rows_per_rank, int* srows_per_rank, int* n_padded,
                       int* m_padded);

/* Tuning of the large-matrix path, process-wide (0 = leave as is): the outer panel width of the two-level
 * blocked Cholesky (multiple of 128, default 512) and the rows of the diagonal blocks of L that are inverted for
 * the block triangular solves (128 times a power of two, default 4096).  No counterpart in the reference. */
int slamk_set_block_sizes(int chol_outer_block_cols, int inverse_block_rows);

/* ---- introspection for tests and benchmarks (no counterpart in the reference) ----------- */
/* ----- batched forms of the stream pre-processing filters that feed the estimator (device pointers) ----- */
/* src/euler2qbw/euler2qbw.cpp:37-46: n rows (phi, theta, psi) -> n scalar-last quaternions. */
int slamk_pre_euler2qbw_dev(const double* euler_dev, double* q_dev, int n, void* stream);
/* src/rotate/main.cpp:146-154: out = R(q) v for n vectors; q as given, never normalised (Quaternion.cpp:32-47). */
int slamk_pre_rotate_dev(const double q[4], const double* v_dev, double* out_dev, int n, void* stream);
/* src/img2body/img2body.cpp:31-67: n pixels (x, y) -> body-frame ratios (Y1/Y0, Y2/Y0), compiled-in camera. */
int slamk_pre_img2body_dev(const double* px_dev, double* out_dev, int n, void* stream);
/* src/lpf/fir.cpp:39-49,105-115: 3-channel FIR, y[i] = sum_k x[i-k] coeffs[k], zero history; x and y distinct. */
int slamk_pre_fir_dev(const double* coeffs_dev, int order, const double* x_dev, double* y_dev, int n, void* stream);

/* Introspection is opt-in: after slamk_set_debug(h, 1) every predict/update also writes the dense F, H and
 * the Cholesky factor L, which the products themselves never need. */
int slamk_set_debug(slamk_handle* h, int on);
/* Copies of the last update's intermediates of filter b (NULL = skip): H m x n, K n x m and the innovation e[m]
 * in the measurement order of this header (altitude first); L m x m (lower) is the Cholesky factor of S in the
 * order the library computes in, feature rows first and the altitude row LAST (rows and columns of S rotated by
 * one): the factor of a permuted matrix is not a permutation of the factor. */
int slamk_debug_get(slamk_handle* h, int b, double* H, double* L, double* K, double* e);
/* Dense motion Jacobian of the last predict, n x n. */
int slamk_debug_get_F(slamk_handle* h, int b, double* F);
/* Plain strided-batched DGEMM on device pointers through the library's MFMA kernel:
 * C = alpha A op(B) + beta C; all of M, N, K multiples of 16; trans_b != 0 means B is N x K. */
int slamk_dgemm_dev(int trans_b, int M, int N, int K, double alpha, const double* A_dev, int lda,
                    long long strideA, const double* B_dev, int ldb, long long strideB, double beta,
                    double* C_dev, int ldc, long long strideC, int batch, void* stream);
/* Issue-rate probe: runs `iters` dependent-free v_mfma_f64_16x16x4_f64 per wave on every CU and
 * returns the achieved TFLOP/s in *tflops. */
int slamk_mfma_f64_probe(int iters, double* tflops);
/* Stage timing with HIP events recorded on the launch stream around each group of kernels.
 * Categories (index into ms/count): 0 per-feature and elementwise kernels, 1 covariance predict
 * (products with the compact rows of F), 2 products with the compact rows of H (P H^T, H W + R,
 * I - K H), 3 GEMMs of the gain on the large-map path (Y, K), 4 GEMMs of the covariance update,
 * 5 GEMMs inside the Cholesky factorisation / triangular inverse, 6 diagonal-block factorisation
 * kernel, 7 fused per-filter gain solver of the small-filter path (ncat >= 8).
 * slamk_profile_enable: on = 0 no stage is timed, 1 all of them, any other value a set of categories with bit
 * (c + 1) for category c (every timed scope costs two event records on the stream, a few microseconds: a caller who
 * wants one kernel's time beside an undisturbed throughput figure selects that stage only).
 * slamk_profile_read synchronises, returns the accumulated milliseconds
 * and scope counts since the previous read, and resets them. */
int slamk_profile_enable(slamk_handle* h, int on);
int slamk_profile_read(slamk_handle* h, double* ms, int* count, int ncat);
/* FLOPs the MFMA kernels execute for one predict / one update of a filter of state dimension n
 * and measurement dimension m (tile skipping and triangular k-range trimming included; the
 * products with the sparse Jacobians, the diagonal-block factorisations and the K e product, which
 * do not run on MFMA, excluded: slamk_flops_predict is therefore 0).  This is the numerator of the
 * fp64-matrix roofline fraction. */
double slamk_flops_predict(int n);
double slamk_flops_update(int n, int m, int update_form);
/* The same split by stage: 0 = gain (factorisation + solves), 1 = covariance update. */
double slamk_flops_update_stage(int n, int m, int update_form, int stage);

#ifdef __cplusplus
}
#endif
#endif /* SLAMK_H */

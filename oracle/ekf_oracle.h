/*
 * ekf_oracle.h -- CPU restatement of the EKF predict/update path of
 * presocratics/slam `src/slam` (plain C, fp64, single thread).
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the checker the HIP kernels are
 * compared against and the CPU baseline that bench.py times.  Nothing under
 * slam_amd/ may link, load or call it.
 *
 * PARITY STATUS: the reference holds no golden vector for this path and cannot
 * be built in this image (it needs OpenCV 2.4, which is absent).  The
 * per-feature Jacobian (orc_jacobian_h) is pinned against values obtained by
 * evaluating the reference's own expression text (tests/golden/jacobian_h.json);
 * the motion Jacobian (orc_motion_jacobian) and the feature rates of
 * orc_dynamics_step likewise (tests/golden/motion_model.json); orc_rotation and
 * orc_euler2quaternion against the reference's rotate / euler2qbw fixtures.
 * Everything else is "parity unpinned": it follows the reference source line
 * by line (citations below) but no reference-authored output exists to check
 * it against.  Dense products and the LU solve restate OpenCV 2.4 core
 * (cv::gemm accumulating in double with k ascending; cv::solve DECOMP_LU =
 * Gaussian elimination with partial pivoting, lapack.cpp LUImpl).
 *
 * All citations are relative to /root/reference.
 */
#ifndef EKF_ORACLE_H
#define EKF_ORACLE_H

#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Constants of the reference, gathered in one POD so tests can vary them.
 * Defaults (orc_default_params): src/slam/main.cpp:44,115-116,177,232;
 * src/slam/main.h:25-26; src/ARC/feature.cpp:20; src/ARC/states.cpp:91-92. */
typedef struct {
    double Q0;            /* 256   */
    double R0;            /* 32    */
    double qbias;         /* 1e-5  */
    double p0;            /* 2     new-feature covariance */
    double pinit;         /* 1e-4  */
    double alt_override;  /* -1.34 */
    int    use_alt_override; /* 1 */
    double gate;          /* 4     */
    double d_min;         /* 0.1   */
    double d_max;         /* 10e3  */
    double dinit;         /* 100   */
} orc_params;

/* include/feature.h:15-25,217-220 (fields used by the path only) */
typedef struct {
    int    id;
    double pib[3];     /* position.body: (y/x, z/x, 1/x) */
    double anchor[3];  /* initial.anchor */
    double q0[4];      /* initial.quaternion, scalar last */
    double pib0[2];    /* initial.pib */
    int    isRef;      /* initial.isRef */
    int    noMatch;
} orc_feature;

/* include/featureIO.h:31-37 */
typedef struct {
    int    id;
    double sx, sy;     /* source */
    double rx, ry;     /* reflection, (0,0) = none */
} orc_match;

/* include/states.h:21-59 plus the covariance main() carries next to it */
typedef struct {
    double X[3], V[3], b[3];
    int    nf, cap;
    orc_feature *feat;
    int    rf, nrf;
    double *P;         /* n x n row-major, n = 9 + 3 nf, contiguous */
    int    n;
    orc_params prm;
} orc_filter;

/* Update bitmask, include/Sensors.hpp:9-16 */
enum {
    ORC_ACC = 1, ORC_ANG = 1 << 1, ORC_QUAT = 1 << 2, ORC_IMG = 1 << 3,
    ORC_ALT = 1 << 4, ORC_INIT = 1 << 5, ORC_POS = 1 << 6, ORC_VEL = 1 << 7
};

/* Latest sensor values as main() sees them after Sensors::update. */
typedef struct {
    double quat[4], acc[3], ang[3], alt, init[3], vel[3], pos[3];
} orc_sensors;

void orc_default_params(orc_params *p);

/* ---- per-function restatements ---------------------------------------- */
/* src/ARC/Quaternion.cpp:32-47 ; R row-major 3x3 */
void orc_rotation(const double q[4], double R[9]);
/* src/euler2qbw/euler2qbw.cpp:37-46 */
void orc_euler2quaternion(double phi, double theta, double psi, double q[4]);
/* src/img2body/img2body.cpp:46-67 (uses orc_rotation) */
void orc_image2body(double px, double py, double *bx, double *by);
/* src/lpf/fir.cpp:39-49,105-115, one channel */
void orc_fir(const double *coeffs, int order, const double *x, double *y, int n);

/* src/ARC/feature.cpp:143-198 */
void orc_xib_hat(const double pib[3], double out[3]);
void orc_xib0_hat(const orc_feature *f, const double pos[3], const double qbw[4], double out[3]);
void orc_pib0_hat(const orc_feature *f, const double pos[3], const double qbw[4], double out[2]);
void orc_xpb_hat(const orc_feature *f, const double X[3], const double qbw[4], double out[3]);
void orc_ppb_hat(const orc_feature *f, const double X[3], const double qbw[4], double out[2]);

/* src/slam/main.cpp:575-624 in chain-rule form; Hb 4x6, Hi 4x3 row-major */
void orc_jacobian_h(const double X[3], const double qbw[4], const orc_feature *f,
                    double Hb[24], double Hi[12]);

/* src/ARC/states.cpp:222-263 + :69-86 : mu += dynamics(mu)*dt (in place) */
void orc_dynamics_step(orc_filter *f, const orc_sensors *s, double dt);
/* src/slam/main.cpp:241-341 ; F n x n row-major */
void orc_motion_jacobian(const orc_filter *f, const orc_sensors *s, double dt, double *F);
/* src/slam/main.cpp:430-440 ; diagonal only, length n */
void orc_init_q_diag(int nf, double Q0, double qbias, double dt, double *qd);
/* src/slam/main.cpp:449-476 ; diagonal only; returns m */
int  orc_init_r_diag(double R0, const int *refFlag, int nfeat, double *rd);
/* src/slam/main.cpp:526-530 with G = I */
void orc_calc_p(double *P, const double *F, const double *qd, int n);
/* src/slam/main.cpp:539-552 ; H m x n, P n x n, rd diag of R ; K n x m */
int  orc_calc_k(double *K, const double *H, const double *P, const double *rd, int m, int n);
/* src/slam/main.cpp:561-569 */
void orc_update_p(double *P, const double *K, const double *H, int m, int n);

/* src/slam/main.cpp:348-408 + src/ARC/view.cpp:3-26.
 * meas/hmu flattened in View::toMat order, H dense m x n (caller gives
 * m_cap >= 1+6nf rows of storage), refFlag[nf] as main.cpp:211-218.
 * returns m, or <0: -1 "Cannot find match." */
int orc_measurement_model(const orc_filter *f, const double old_pos[3], double alt,
                          const orc_match *matches, int nmatch, const double qbw[4],
                          double *meas, double *hmu, double *H, int *refFlag);

/* ---- filter object ------------------------------------------------------ */
orc_filter *orc_filter_new(const orc_params *p);   /* P = diag(pinit,1,pinit) 9x9 */
void orc_filter_free(orc_filter *f);
/* src/ARC/states.cpp:88-213 ; association + P surgery. */
void orc_update_features(orc_filter *f, const orc_match *matches, int nmatch,
                         const double quat[4]);

/* One iteration of the while loop, src/slam/main.cpp:157-235, after
 * Sensors::update returned `u` and (if IMG) the frame `matches` was read.
 * out_xy receives the two numbers of the printf at :207 (X[1], X[0]).
 * *applied = 1 if the gated mean update ran (:232), 0 if rejected, -1 if no
 * image this step.  pre_P/pre_state (nullable) receive the after-predict,
 * before-update snapshot (the point where the reference prints).
 * returns 0, or <0 on the conditions where the reference exits/throws:
 * -1 cannot find match, -2 measurement row count mismatch. */
int orc_step(orc_filter *f, int u, orc_sensors *s, double dt, double old_pos[3],
             const orc_match *matches, int nmatch, double out_xy[2], int *applied);

/* Whole program, src/slam/main.cpp:106-238 : sensors stream, feature file,
 * stdout.  dump (nullable) gets one record per iteration at the printf point:
 * "STEP iter nf\nX..\nV..\n<id pib>*\nP rows" in %a hex floats.
 * returns process exit status (0 / 1). */
int orc_slam_run(FILE *sensors, FILE *features, FILE *out, FILE *err, FILE *dump,
                 const orc_params *p);

/* Batch helper for the CPU baseline: runs orc_step over `count` independent
 * filters that were prepared by the caller (same code path, no threads). */

#ifdef __cplusplus
}
#endif
#endif

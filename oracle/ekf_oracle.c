/*
 * ekf_oracle.c -- see ekf_oracle.h.  TEST INFRASTRUCTURE ONLY.
 *
 * Build with -ffp-contract=off: every product and sum is rounded once, in the
 * order written here (k ascending in all dense products), which is what a
 * plain x86-64 build of the reference against OpenCV 2.4 core does.
 */
#include "ekf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ------------------------------------------------------------------------ */
/* small dense helpers                                                       */
/* ------------------------------------------------------------------------ */

/* C (M x N) = A (M x K) * B (K x N); every C[i][j] accumulates k = 0..K-1 in
 * ascending order starting from 0 (i-k-j loop order only changes which element
 * is touched next, not the order of additions into one element). */
static void mm(double *C, const double *A, const double *B, int M, int K, int N)
{
    for (int i = 0; i < M; ++i) {
        double *c = C + (size_t)i * N;
        for (int j = 0; j < N; ++j) c[j] = 0.0;
        for (int k = 0; k < K; ++k) {
            const double a = A[(size_t)i * K + k];
            const double *b = B + (size_t)k * N;
            for (int j = 0; j < N; ++j) c[j] += a * b[j];
        }
    }
}

/* C (M x N) = A (M x K) * B^T, B given as N x K row-major. */
static void mm_nt(double *C, const double *A, const double *B, int M, int K, int N)
{
    for (int i = 0; i < M; ++i) {
        const double *a = A + (size_t)i * K;
        for (int j = 0; j < N; ++j) {
            const double *b = B + (size_t)j * K;
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += a[k] * b[k];
            C[(size_t)i * N + j] = s;
        }
    }
}

static void mat3_mul(const double A[9], const double B[9], double C[9])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}
static void mat3_t(const double A[9], double T[9])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * j + i];
}
static void mat3_vec(const double A[9], const double v[3], double o[3])
{
    for (int i = 0; i < 3; ++i) {
        double s = 0.0;
        for (int k = 0; k < 3; ++k) s += A[3 * i + k] * v[k];
        o[i] = s;
    }
}

void orc_default_params(orc_params *p)
{
    p->Q0 = 256;              /* src/slam/main.cpp:115 */
    p->R0 = 32;               /* :116 */
    p->qbias = 1e-5;          /* src/slam/main.h:25 */
    p->p0 = 2;                /* src/ARC/states.cpp:136,164 (literal 2) */
    p->pinit = 1e-4;          /* src/slam/main.cpp:44 */
    p->alt_override = -1.34;  /* :177 */
    p->use_alt_override = 1;
    p->gate = 4;              /* :232 */
    p->d_min = 0.1;           /* src/ARC/states.cpp:91 */
    p->d_max = 10e3;          /* :92 */
    p->dinit = 100;           /* src/ARC/feature.cpp:20 */
}

/* ------------------------------------------------------------------------ */
/* a1: Quaternion::rotation  src/ARC/Quaternion.cpp:32-47                    */
/* ------------------------------------------------------------------------ */
void orc_rotation(const double c[4], double m[9])
{
    m[0] = pow(c[3], 2) + pow(c[0], 2) - pow(c[1], 2) - pow(c[2], 2);
    m[3] = 2 * (c[0] * c[1] + c[3] * c[2]);
    m[6] = 2 * (c[0] * c[2] - c[3] * c[1]);
    m[1] = 2 * (c[0] * c[1] - c[3] * c[2]);
    m[4] = pow(c[3], 2) - pow(c[0], 2) + pow(c[1], 2) - pow(c[2], 2);
    m[7] = 2 * (c[3] * c[0] + c[1] * c[2]);
    m[2] = 2 * (c[3] * c[1] + c[0] * c[2]);
    m[5] = 2 * (c[1] * c[2] - c[3] * c[0]);
    m[8] = pow(c[3], 2) - pow(c[0], 2) - pow(c[1], 2) + pow(c[2], 2);
}

/* src/euler2qbw/euler2qbw.cpp:37-46 */
void orc_euler2quaternion(double phi, double theta, double psi, double q[4])
{
    q[0] = sin(phi / 2) * cos(theta / 2) * cos(psi / 2) - cos(phi / 2) * sin(theta / 2) * sin(psi / 2);
    q[1] = cos(phi / 2) * sin(theta / 2) * cos(psi / 2) + sin(phi / 2) * cos(theta / 2) * sin(psi / 2);
    q[2] = cos(phi / 2) * cos(theta / 2) * sin(psi / 2) - sin(phi / 2) * sin(theta / 2) * cos(psi / 2);
    q[3] = cos(phi / 2) * cos(theta / 2) * cos(psi / 2) + sin(phi / 2) * sin(theta / 2) * sin(psi / 2);
}

/* src/lpf/fir.cpp:39-49 (filter) driven as main() does, :105-115: one channel, ring buffer of `order` zeros */
void orc_fir(const double *coeffs, int order, const double *x, double *y, int n)
{
    double *ring = (double *)calloc((size_t)order, sizeof(double));
    for (int i = 0; i < n; ++i) {
        const int index = i % order;
        ring[index] = x[i];
        double acc = 0;
        for (int k = 0; k < order; ++k) acc += ring[(order + index - k) % order] * coeffs[k];
        y[i] = acc;
    }
    free(ring);
}

/* src/img2body/img2body.cpp:31-37,46-67 */
void orc_image2body(double px, double py, double *bx, double *by)
{
    const double cx = 767.521362, cy = 591.428406, fx = 801.362183, fy = 801.362183;
    const double qbc[4] = { -0.707106781186547, -0.000000000000000, 0.000000000000000, 0.707106781186548 };
    double Rb2c[9], Rc2b[9], S[3], Y[3];
    orc_rotation(qbc, Rb2c);
    mat3_t(Rb2c, Rc2b);
    double x = (px - cx) / fx, y = (py - cy) / fy;
    double s = 1 / sqrt(x * x + y * y + 1);
    S[0] = x * s; S[1] = y * s; S[2] = 1 * s;
    mat3_vec(Rc2b, S, Y);
    *bx = Y[1] / Y[0];
    *by = Y[2] / Y[0];
}

/* ------------------------------------------------------------------------ */
/* a6: Feature::*Hat  src/ARC/feature.cpp:110-198                            */
/* ------------------------------------------------------------------------ */
void orc_xib_hat(const double p[3], double o[3])
{   /* :171-177 */
    o[0] = 1 / p[2];
    o[1] = p[0] / p[2];
    o[2] = p[1] / p[2];
}

void orc_xib0_hat(const orc_feature *f, const double pos[3], const double qbw[4], double out[3])
{   /* :178-188 with fromAnchor :110-114 and rb2b :117-126 */
    double R0[9], R0t[9], Rb[9], Rb2b0[9], xib[3], a[3], d[3], fa[3];
    orc_rotation(f->q0, R0);
    mat3_t(R0, R0t);
    orc_rotation(qbw, Rb);
    mat3_mul(R0t, Rb, Rb2b0);
    orc_xib_hat(f->pib, xib);
    mat3_vec(Rb2b0, xib, a);
    for (int i = 0; i < 3; ++i) d[i] = pos[i] - f->anchor[i];
    mat3_vec(R0t, d, fa);
    for (int i = 0; i < 3; ++i) out[i] = a[i] + fa[i];
}

void orc_pib0_hat(const orc_feature *f, const double pos[3], const double qbw[4], double out[2])
{   /* :189-198 */
    double x[3];
    orc_xib0_hat(f, pos, qbw, x);
    out[0] = x[1] / x[0];
    out[1] = x[2] / x[0];
}

void orc_xpb_hat(const orc_feature *f, const double X[3], const double qbw[4], double out[3])
{   /* :155-169 : temp = S*Rb2w*xib ; temp -= 2 n n^T X ; temp = Rw2b*temp */
    const double S[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 - 2 * 1 * 1 };
    const double NN2[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 2 };
    double Rb[9], Rbt[9], SR[9], xib[3], t[3], nx[3];
    orc_rotation(qbw, Rb);
    mat3_t(Rb, Rbt);
    orc_xib_hat(f->pib, xib);
    mat3_mul(S, Rb, SR);
    mat3_vec(SR, xib, t);
    mat3_vec(NN2, X, nx);
    for (int i = 0; i < 3; ++i) t[i] -= nx[i];
    mat3_vec(Rbt, t, out);
}

void orc_ppb_hat(const orc_feature *f, const double X[3], const double qbw[4], double out[2])
{   /* :143-153 */
    double x[3];
    orc_xpb_hat(f, X, qbw, x);
    out[0] = x[1] / x[0];
    out[1] = x[2] / x[0];
}

/* ------------------------------------------------------------------------ */
/* a7: jacobianH  src/slam/main.cpp:575-624                                  */
/* The reference spells the 4x6 / 4x3 matrices out as symbolic expressions   */
/* (:604-623).  This is the same derivative by the chain rule:               */
/*   pib0 = proj(R0^T Rb xib(p) + R0^T (X - anchor)),                        */
/*   ppb  = proj(Rb^T (S Rb xib(p) - 2 n n^T X)),  proj(x) = (x1,x2)/x0.     */
/* Checked against values of the reference's expression text in              */
/* tests/golden/jacobian_h.json.                                             */
/* ------------------------------------------------------------------------ */
static void proj_jac(const double x[3], double J[6])
{
    const double ix = 1 / x[0];
    J[0] = -x[1] * ix * ix; J[1] = ix; J[2] = 0;
    J[3] = -x[2] * ix * ix; J[4] = 0;  J[5] = ix;
}

void orc_jacobian_h(const double X[3], const double qbw[4], const orc_feature *f,
                    double Hb[24], double Hi[12])
{
    double R0[9], R0t[9], Rb[9], Rbt[9], Rb2b0[9];
    double xib[3], x0[3], xp[3], J0[6], Jp[6];
    const double *p = f->pib;
    orc_rotation(f->q0, R0);
    mat3_t(R0, R0t);
    orc_rotation(qbw, Rb);
    mat3_t(Rb, Rbt);
    mat3_mul(R0t, Rb, Rb2b0);
    orc_xib_hat(p, xib);
    orc_xib0_hat(f, X, qbw, x0);
    orc_xpb_hat(f, X, qbw, xp);
    proj_jac(x0, J0);
    proj_jac(xp, Jp);

    /* d xib / d p */
    const double ip = 1 / p[2];
    const double D[9] = { 0, 0, -ip * ip,
                          ip, 0, -p[0] * ip * ip,
                          0, ip, -p[1] * ip * ip };
    /* M0 = Rb2b0 * D ; Mp = Rb^T * S * Rb * D */
    const double S[9] = { 1, 0, 0, 0, 1, 0, 0, 0, -1 };
    double M0[9], SR[9], RSR[9], Mp[9];
    mat3_mul(Rb2b0, D, M0);
    mat3_mul(S, Rb, SR);
    mat3_mul(Rbt, SR, RSR);
    mat3_mul(RSR, D, Mp);

    memset(Hb, 0, 24 * sizeof(double));
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0, si = 0, sp = 0;
            for (int k = 0; k < 3; ++k) {
                s  += J0[3 * r + k] * R0t[3 * k + c];
                si += J0[3 * r + k] * M0[3 * k + c];
                sp += Jp[3 * r + k] * Mp[3 * k + c];
            }
            Hb[6 * r + c] = s;
            Hi[3 * r + c] = si;
            Hi[3 * (r + 2) + c] = sp;
        }
    /* d xpb / d X = Rb^T * (-2 n n^T): only column 2 is non-zero */
    for (int r = 0; r < 2; ++r) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += Jp[3 * r + k] * (-2 * Rbt[3 * k + 2]);
        Hb[6 * (r + 2) + 2] = s;
    }
}

/* ------------------------------------------------------------------------ */
/* a2: States::dynamics + add  src/ARC/states.cpp:222-263, :69-86, :6-14     */
/* ------------------------------------------------------------------------ */
void orc_dynamics_step(orc_filter *f, const orc_sensors *s, double dt)
{
    double Rb[9], dX[3], dV[3];
    const double *w = s->ang;
    const double *V = f->V;
    orc_rotation(s->quat, Rb);
    mat3_vec(Rb, V, dX);                                   /* :241 */
    for (int i = 0; i < 3; ++i) dV[i] = s->acc[i];          /* :243 */
    /* feature increments use the pre-step V (:254-256) */
    for (int i = 0; i < f->nf; ++i) {
        double *bp = f->feat[i].pib;
        double d0 = (-V[1] + bp[0] * V[0]) * bp[2] + bp[1] * w[0] - (1 + bp[0] * bp[0]) * w[2] + bp[0] * bp[1] * w[1];
        double d1 = (-V[2] + bp[1] * V[0]) * bp[2] - bp[0] * w[0] + (1 + bp[1] * bp[1]) * w[1] - bp[0] * bp[1] * w[2];
        double d2 = (-w[2] * bp[0] + w[1] * bp[1]) * bp[2] + V[0] * bp[2] * bp[2];
        /* operator*= scales, add() adds (:12, :80) */
        bp[0] += d0 * dt;
        bp[1] += d1 * dt;
        bp[2] += d2 * dt;
    }
    for (int i = 0; i < 3; ++i) dV[i] -= f->b[i];           /* :260 */
    for (int i = 0; i < 3; ++i) {
        f->X[i] += dX[i] * dt;
        f->V[i] += dV[i] * dt;
        /* predicted_state.b = 0 (:261), operator*= leaves b alone, add(): b += 0 */
        f->b[i] += 0.0;
    }
}

/* ------------------------------------------------------------------------ */
/* a3: jacobianMotionModel  src/slam/main.cpp:241-341                        */
/* ------------------------------------------------------------------------ */
void orc_motion_jacobian(const orc_filter *f, const orc_sensors *s, double dt, double *F)
{
    const int nf = f->nf, n = 9 + 3 * nf;
    const double *q = s->quat, *w = s->ang, *V = f->V;
    memset(F, 0, (size_t)n * n * sizeof(double));
#define FA(r, c) F[(size_t)(r) * n + (c)]
    /* Fb1 (:253-264): rows 0..2, cols 3..5 */
    FA(0, 3) = pow(q[0], 2) - pow(q[1], 2) - pow(q[2], 2) + pow(q[3], 2);
    FA(0, 4) = 2 * q[0] * q[1] - 2 * q[2] * q[3];
    FA(0, 5) = 2 * q[0] * q[2] + 2 * q[1] * q[3];
    FA(1, 3) = 2 * q[0] * q[1] + 2 * q[2] * q[3];
    FA(1, 4) = -pow(q[0], 2) + pow(q[1], 2) - pow(q[2], 2) + pow(q[3], 2);
    FA(1, 5) = 2 * q[1] * q[2] - 2 * q[0] * q[3];
    FA(2, 3) = 2 * q[0] * q[2] - 2 * q[1] * q[3];
    FA(2, 4) = 2 * q[0] * q[3] + 2 * q[1] * q[2];
    FA(2, 5) = -pow(q[0], 2) - pow(q[1], 2) + pow(q[2], 2) + pow(q[3], 2);
    for (int i = 0; i < nf; ++i) {
        const double p1 = f->feat[i].pib[0], p2 = f->feat[i].pib[1], p3 = f->feat[i].pib[2];
        const int r = 6 + 3 * i;
        /* FiTemp (:290-299) at (r, r) */
        FA(r, r)         = p3 * V[0] - 2 * p1 * w[2] + p2 * w[1];
        FA(r, r + 1)     = w[0] + p1 * w[1];
        FA(r, r + 2)     = p1 * V[0] - V[1];
        FA(r + 1, r)     = -w[0] - p2 * w[2];
        FA(r + 1, r + 1) = p3 * V[0] - p1 * w[2] + 2 * p2 * w[1];
        FA(r + 1, r + 2) = p2 * V[0] - V[2];
        FA(r + 2, r)     = -p3 * w[2];
        FA(r + 2, r + 1) = p3 * w[1];
        FA(r + 2, r + 2) = 2 * p3 * V[0] - p1 * w[2] + p2 * w[1];
        /* Fib_ith (:314-317) at (r, 0), only cols 3..5 non-zero */
        FA(r, 3)     = p1 * p3;  FA(r, 4)     = -p3; FA(r, 5)     = 0;
        FA(r + 1, 3) = p2 * p3;  FA(r + 1, 4) = 0;   FA(r + 1, 5) = -p3;
        FA(r + 2, 3) = pow(p3, 2); FA(r + 2, 4) = 0; FA(r + 2, 5) = 0;
    }
    /* F = dt*F + I (:335) */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            FA(i, j) = dt * FA(i, j) + (i == j ? 1.0 : 0.0);
    FA(3, 6 + 3 * nf) = -1 * dt;   /* :338-340 */
    FA(4, 7 + 3 * nf) = -1 * dt;
    FA(5, 8 + 3 * nf) = -1 * dt;
#undef FA
}

/* a4: initQ  src/slam/main.cpp:430-440 */
void orc_init_q_diag(int nf, double Q0, double qbias, double dt, double *qd)
{
    const int n = 9 + 3 * nf;
    Q0 *= dt * dt;
    for (int i = 0; i < n; ++i) qd[i] = Q0 * 1.0;
    qd[0] *= 0.1 * Q0;
    qd[1] *= 0.1 * Q0;
    qd[2] *= 0.1 * Q0;
    for (int i = 0; i < 3; ++i) qd[6 + 3 * nf + i] = qbias * dt * dt * 1.0;
}

/* a10: initR  src/slam/main.cpp:449-476 */
int orc_init_r_diag(double R0, const int *refFlag, int nfeat, double *rd)
{
    int m = 0;
    rd[m++] = 0.15 * R0;
    for (int i = 0; i < nfeat; ++i) {
        rd[m++] = 0.01 / 770 * R0;
        rd[m++] = 0.01 / 770 * R0;
        rd[m++] = 10. / 770 * R0;
        rd[m++] = 10. / 770 * R0;
        if (refFlag[i]) {
            rd[m++] = 1. / 770 * R0;
            rd[m++] = 1. / 770 * R0;
        }
    }
    return m;
}

/* a5: calcP  src/slam/main.cpp:526-530.  G = I so G*Q*G^T == Q exactly. */
void orc_calc_p(double *P, const double *F, const double *qd, int n)
{
    double *T = (double *)malloc((size_t)n * n * sizeof(double));
    double *U = (double *)malloc((size_t)n * n * sizeof(double));
    mm(T, F, P, n, n, n);        /* F*P          */
    mm_nt(U, T, F, n, n, n);     /* (F*P)*F^T    */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            P[(size_t)i * n + j] = U[(size_t)i * n + j] + (i == j ? qd[i] : 0.0);
    free(T); free(U);
}

/* cv::solve(A, B, X, DECOMP_LU) for square A (m x m) with nrhs columns in B:
 * Gaussian elimination with partial pivoting, OpenCV 2.4 core lapack.cpp
 * (LUImpl): pivot = largest |a| in the column, singular if < eps, rows are
 * eliminated with alpha = -a_ji/a_ii, the reciprocal pivot is kept on the
 * diagonal and used in back substitution.  A and B are overwritten; B holds X.
 * returns 0 if a pivot is below eps (OpenCV then reports failure; the
 * reference ignores the flag, main.cpp:549). */
static int lu_solve(double *A, int m, double *B, int nrhs)
{
    const double eps = DBL_EPSILON * 100;
    for (int i = 0; i < m; ++i) {
        int k = i;
        for (int j = i + 1; j < m; ++j)
            if (fabs(A[(size_t)j * m + i]) > fabs(A[(size_t)k * m + i])) k = j;
        if (fabs(A[(size_t)k * m + i]) < eps) return 0;
        if (k != i) {
            for (int j = i; j < m; ++j) {
                double t = A[(size_t)i * m + j]; A[(size_t)i * m + j] = A[(size_t)k * m + j]; A[(size_t)k * m + j] = t;
            }
            for (int j = 0; j < nrhs; ++j) {
                double t = B[(size_t)i * nrhs + j]; B[(size_t)i * nrhs + j] = B[(size_t)k * nrhs + j]; B[(size_t)k * nrhs + j] = t;
            }
        }
        const double d = -1 / A[(size_t)i * m + i];
        for (int j = i + 1; j < m; ++j) {
            const double alpha = A[(size_t)j * m + i] * d;
            double *aj = A + (size_t)j * m; const double *ai = A + (size_t)i * m;
            for (int c = i + 1; c < m; ++c) aj[c] += alpha * ai[c];
            double *bj = B + (size_t)j * nrhs; const double *bi = B + (size_t)i * nrhs;
            for (int c = 0; c < nrhs; ++c) bj[c] += alpha * bi[c];
        }
        A[(size_t)i * m + i] = -d;
    }
    for (int i = m - 1; i >= 0; --i) {
        double *bi = B + (size_t)i * nrhs;
        for (int k = i + 1; k < m; ++k) {
            const double a = A[(size_t)i * m + k];
            const double *bk = B + (size_t)k * nrhs;
            for (int j = 0; j < nrhs; ++j) bi[j] -= a * bk[j];
        }
        const double r = A[(size_t)i * m + i];
        for (int j = 0; j < nrhs; ++j) bi[j] *= r;
    }
    return 1;
}

/* a11: calcK  src/slam/main.cpp:539-552 */
int orc_calc_k(double *K, const double *H, const double *P, const double *rd, int m, int n)
{
    double *HP = (double *)malloc((size_t)m * n * sizeof(double));
    double *S  = (double *)malloc((size_t)m * m * sizeof(double));
    double *St = (double *)malloc((size_t)m * m * sizeof(double));
    double *Kt = (double *)malloc((size_t)m * n * sizeof(double));
    mm(HP, H, P, m, n, n);            /* tmp = H*P       (:541) */
    mm_nt(S, HP, H, m, n, m);         /* tmp *= H.t()    (:542) */
    for (int i = 0; i < m; ++i) S[(size_t)i * m + i] += rd[i];   /* += R (:543) */
    mm_nt(K, P, H, n, n, m);          /* K = P*H.t()     (:545) */
    for (int i = 0; i < n; ++i)       /* K = K.t()       (:547) */
        for (int j = 0; j < m; ++j) Kt[(size_t)j * n + i] = K[(size_t)i * m + j];
    for (int i = 0; i < m; ++i)       /* tmp = tmp.t()   (:548) */
        for (int j = 0; j < m; ++j) St[(size_t)j * m + i] = S[(size_t)i * m + j];
    int ok = lu_solve(St, m, Kt, n);  /* solve(tmp,K,K)  (:549) */
    if (!ok) memset(Kt, 0, (size_t)m * n * sizeof(double));   /* cv::solve zero-fills on failure */
    for (int i = 0; i < n; ++i)       /* K = K.t()       (:550) */
        for (int j = 0; j < m; ++j) K[(size_t)i * m + j] = Kt[(size_t)j * n + i];
    free(HP); free(S); free(St); free(Kt);
    return ok;
}

/* a12: updateP  src/slam/main.cpp:561-569 */
void orc_update_p(double *P, const double *K, const double *H, int m, int n)
{
    double *KH = (double *)malloc((size_t)n * n * sizeof(double));
    double *A  = (double *)malloc((size_t)n * n * sizeof(double));
    mm(KH, K, H, n, m, n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            KH[(size_t)i * n + j] = (i == j ? 1.0 : 0.0) - KH[(size_t)i * n + j];
    mm(A, KH, P, n, n, n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            P[(size_t)i * n + j] = (A[(size_t)j * n + i] + A[(size_t)i * n + j]) / 2;
    free(KH); free(A);
}

/* ------------------------------------------------------------------------ */
/* a8/a9: measurementModel  src/slam/main.cpp:348-408, View::toMat order     */
/* ------------------------------------------------------------------------ */
int orc_measurement_model(const orc_filter *f, const double old_pos[3], double alt,
                          const orc_match *matches, int nmatch, const double qbw[4],
                          double *meas, double *hmu, double *H, int *refFlag)
{
    const int n = 9 + 3 * f->nf;
    const int m = 1 + 6 * f->rf + 4 * f->nrf;              /* :353 */
    memset(H, 0, (size_t)m * n * sizeof(double));
    meas[0] = alt;                                          /* :351 */
    hmu[0] = -f->X[2];                                      /* :352 */
    H[2] = -1;                                              /* :354 */
    int index = 1;
    for (int i = 0; i < f->nf; ++i) {
        const orc_feature *ft = &f->feat[i];
        const orc_match *mt = NULL;
        for (int k = 0; k < nmatch; ++k)                    /* :362-371 */
            if (matches[k].id == ft->id) { mt = &matches[k]; break; }
        if (!mt) return -1;                                 /* :372-375 */
        double Hb[24], Hi[12], p0h[2], pph[2];
        orc_jacobian_h(f->X, qbw, ft, Hb, Hi);              /* :376 */
        orc_pib0_hat(ft, old_pos, qbw, p0h);
        const int rows = ft->isRef ? 6 : 4;
        /* meas/hmu in View::toMat order (view.cpp:12-22) */
        meas[index + 0] = mt->sx;       meas[index + 1] = mt->sy;
        meas[index + 2] = ft->pib0[0];  meas[index + 3] = ft->pib0[1];
        hmu[index + 0] = ft->pib[0];    hmu[index + 1] = ft->pib[1];
        hmu[index + 2] = p0h[0];        hmu[index + 3] = p0h[1];
        if (ft->isRef) {                                    /* :377-382 */
            orc_ppb_hat(ft, f->X, qbw, pph);
            meas[index + 4] = mt->rx;   meas[index + 5] = mt->ry;
            hmu[index + 4] = pph[0];    hmu[index + 5] = pph[1];
            refFlag[i] = !(mt->rx == 0 && mt->ry == 0);     /* main.cpp:213 */
        } else {
            refFlag[i] = 0;                                 /* Vfeat(...) two-arg ctor: NONREF */
        }
        /* Hfeat (:391-394): rows 0-1 identity at cols 6+3i; rows 2.. = [Hb | Hi at 6+3i] */
        H[(size_t)(index + 0) * n + 6 + 3 * i] = 1;
        H[(size_t)(index + 1) * n + 7 + 3 * i] = 1;
        for (int r = 0; r < rows - 2; ++r) {
            double *row = H + (size_t)(index + 2 + r) * n;
            for (int c = 0; c < 6; ++c) row[c] = Hb[6 * r + c];
            for (int c = 0; c < 3; ++c) row[6 + 3 * i + c] = Hi[3 * r + c];
        }
        index += rows;                                      /* :396-406 */
    }
    return m;
}

/* ------------------------------------------------------------------------ */
/* filter object                                                             */
/* ------------------------------------------------------------------------ */
orc_filter *orc_filter_new(const orc_params *p)
{
    orc_filter *f = (orc_filter *)calloc(1, sizeof(orc_filter));
    if (p) f->prm = *p; else orc_default_params(&f->prm);
    f->n = 9;
    f->P = (double *)calloc(81, sizeof(double));
    for (int i = 0; i < 9; ++i) f->P[i * 9 + i] = 1.0;                 /* main.cpp:121 */
    for (int i = 0; i < 3; ++i) {
        f->P[i * 9 + i] = f->prm.pinit * 1.0;                          /* :122 */
        f->P[(6 + i) * 9 + 6 + i] = f->prm.pinit * 1.0;                /* :123 */
    }
    return f;
}

void orc_filter_free(orc_filter *f)
{
    if (!f) return;
    free(f->feat);
    free(f->P);
    free(f);
}

/* Feature(anchor, sense, match)  src/ARC/feature.cpp:41-91 */
static void feature_init(orc_feature *ft, const double X[3], const double quat[4],
                         const orc_match *mt, double dinit)
{
    ft->pib[0] = mt->sx; ft->pib[1] = mt->sy; ft->pib[2] = 1 / dinit;   /* :61-62 */
    ft->isRef = !(mt->rx == 0 && mt->ry == 0);                          /* :69 */
    ft->id = mt->id;
    memcpy(ft->anchor, X, 3 * sizeof(double));                          /* :71 */
    memcpy(ft->q0, quat, 4 * sizeof(double));                           /* :87 */
    ft->pib0[0] = mt->sx; ft->pib0[1] = mt->sy;                         /* :88 */
    ft->noMatch = 0;                                                    /* :89 */
}

/* a14: States::update_features  src/ARC/states.cpp:88-213 */
void orc_update_features(orc_filter *f, const orc_match *matches, int nmatch,
                         const double quat[4])
{
    const double d_min = f->prm.d_min, d_max = f->prm.d_max;
    /* pass 1 (:94-115): clamp inverse depth, mark unmatched */
    for (int i = 0; i < f->nf; ++i) {
        orc_feature *ft = &f->feat[i];
        if (ft->pib[2] < 1 / d_max) ft->pib[2] = 1 / d_max;
        else if (ft->pib[2] > 1 / d_min) ft->pib[2] = 1 / d_min;
        int found = 0;
        for (int k = 0; k < nmatch; ++k)
            if (ft->id == matches[k].id) { found = 1; break; }
        ft->noMatch = found ? 0 : 1;
    }
    /* pass 2 (:118-167): new IDs take the first unmatched slot, else append */
    for (int k = 0; k < nmatch; ++k) {
        int found = 0;
        for (int i = 0; i < f->nf; ++i)
            if (f->feat[i].id == matches[k].id) { found = 1; break; }
        if (found) continue;
        int slot = -1;
        for (int i = 0; i < f->nf; ++i)
            if (f->feat[i].noMatch != 0) { slot = i; break; }
        if (slot >= 0) {
            const int n = f->n, r = 6 + 3 * slot;
            feature_init(&f->feat[slot], f->X, quat, &matches[k], f->prm.dinit);
            for (int a = r; a < r + 3; ++a)                              /* :134-135 */
                for (int c = 0; c < n; ++c) {
                    f->P[(size_t)a * n + c] = 0;
                    f->P[(size_t)c * n + a] = 0;
                }
            for (int a = 0; a < 3; ++a) f->P[(size_t)(r + a) * n + r + a] = f->prm.p0;   /* :136 */
        } else {
            if (f->nf == f->cap) {
                f->cap = f->cap ? 2 * f->cap : 16;
                f->feat = (orc_feature *)realloc(f->feat, (size_t)f->cap * sizeof(orc_feature));
            }
            feature_init(&f->feat[f->nf], f->X, quat, &matches[k], f->prm.dinit);
            f->nf += 1;
            const int nf = f->nf, no = f->n, nn = 9 + 3 * nf;
            const int ob = 6 + 3 * (nf - 1), nb = 6 + 3 * nf;
            double *B = (double *)calloc((size_t)nn * nn, sizeof(double));    /* :145 */
            for (int a = 0; a < no; ++a)                                      /* :146 */
                memcpy(B + (size_t)a * nn, f->P + (size_t)a * no, (size_t)no * sizeof(double));
            /* bias rows/cols/diagonal move from ob to nb (:150-160) */
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < ob; ++c) {
                    B[(size_t)(nb + a) * nn + c] = B[(size_t)(ob + a) * nn + c];
                    B[(size_t)c * nn + nb + a] = B[(size_t)c * nn + ob + a];
                }
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < 3; ++c)
                    B[(size_t)(nb + a) * nn + nb + c] = B[(size_t)(ob + a) * nn + ob + c];
            /* vacated strips -> 0, new feature block -> p0*I (:162-164) */
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < ob; ++c) {
                    B[(size_t)(ob + a) * nn + c] = 0;
                    B[(size_t)c * nn + ob + a] = 0;
                }
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < 3; ++c)
                    B[(size_t)(ob + a) * nn + ob + c] = (a == c) ? f->prm.p0 : 0.0;
            free(f->P);
            f->P = B;
            f->n = nn;
        }
    }
    /* pass 3 (:170-200): erase features still unmatched; delete their rows/cols */
    int i = 0;
    while (i < f->nf) {
        if (f->feat[i].noMatch == 0) { ++i; continue; }
        const int no = f->n, nn = no - 3, r = 6 + 3 * i;
        double *B = (double *)malloc((size_t)nn * nn * sizeof(double));
        for (int a = 0, ao = 0; ao < no; ++ao) {
            if (ao >= r && ao < r + 3) continue;
            for (int c = 0, co = 0; co < no; ++co) {
                if (co >= r && co < r + 3) continue;
                B[(size_t)a * nn + c] = f->P[(size_t)ao * no + co];
                ++c;
            }
            ++a;
        }
        free(f->P);
        f->P = B;
        f->n = nn;
        memmove(&f->feat[i], &f->feat[i + 1], (size_t)(f->nf - i - 1) * sizeof(orc_feature));
        f->nf -= 1;
    }
    /* :203-211 */
    f->rf = f->nrf = 0;
    for (int k = 0; k < f->nf; ++k) {
        if (f->feat[k].isRef) f->rf++; else f->nrf++;
    }
}

/* ------------------------------------------------------------------------ */
/* one loop iteration  src/slam/main.cpp:157-235                             */
/* ------------------------------------------------------------------------ */
int orc_step(orc_filter *f, int u, orc_sensors *s, double dt, double old_pos[3],
             const orc_match *matches, int nmatch, double out_xy[2], int *applied)
{
    int rc = 0;
    if (applied) *applied = -1;
    if (u & ORC_IMG) {                                       /* :167-175 */
        memcpy(old_pos, f->X, 3 * sizeof(double));
        orc_update_features(f, matches, nmatch, s->quat);
    }
    if (f->prm.use_alt_override) s->alt = f->prm.alt_override;   /* :177 */
    if (u & ORC_POS) {                                       /* :180-185 */
        f->X[0] = s->pos[1]; f->X[1] = s->pos[0]; f->X[2] = s->pos[2];
    }
    if (u & ORC_VEL) {                                       /* :186-191 */
        double Rb[9], Rt[9], v[3];
        orc_rotation(s->quat, Rb);
        mat3_t(Rb, Rt);
        mat3_vec(Rt, s->vel, v);
        memcpy(f->V, v, sizeof v);
    }
    const int n = f->n, nf = f->nf;
    double *F = (double *)malloc((size_t)n * n * sizeof(double));
    double *qd = (double *)malloc((size_t)n * sizeof(double));
    orc_motion_jacobian(f, s, dt, F);                        /* :192 */
    orc_dynamics_step(f, s, dt);                             /* :193-194 */

    int m = 0;
    double *meas = NULL, *hmu = NULL, *H = NULL;
    int *refFlag = NULL;
    if (u & ORC_IMG) {                                       /* :198-202 */
        const int mcap = 1 + 6 * nf;
        meas = (double *)calloc(mcap, sizeof(double));
        hmu = (double *)calloc(mcap, sizeof(double));
        H = (double *)malloc((size_t)mcap * n * sizeof(double));
        refFlag = (int *)calloc(nf + 1, sizeof(int));
        m = orc_measurement_model(f, old_pos, s->alt, matches, nmatch, s->quat, meas, hmu, H, refFlag);
        if (m < 0) { rc = -1; goto done; }
    }
    orc_init_q_diag(nf, f->prm.Q0, f->prm.qbias, dt, qd);    /* :204-205 */
    orc_calc_p(f->P, F, qd, n);                              /* :206 */
    if (out_xy) { out_xy[0] = f->X[1]; out_xy[1] = f->X[0]; }   /* :207 */

    if (u & ORC_IMG) {                                       /* :209-233 */
        double *rd = (double *)malloc((size_t)(1 + 6 * nf) * sizeof(double));
        const int mr = orc_init_r_diag(f->prm.R0, refFlag, nf, rd);
        if (mr != m) { free(rd); rc = -2; goto done; }       /* cv::Exception in `tmp += R` */
        double *K = (double *)malloc((size_t)n * m * sizeof(double));
        orc_calc_k(K, H, f->P, rd, m, n);
        orc_update_p(f->P, K, H, m, n);
        /* subtract + toMat (view.cpp:28-47, :3-26) */
        double *e = (double *)malloc((size_t)m * sizeof(double));
        int me = 0, idx = 1;
        e[me++] = meas[0] - hmu[0];
        for (int i = 0; i < nf; ++i) {
            for (int c = 0; c < 4; ++c) e[me++] = meas[idx + c] - hmu[idx + c];
            if (f->feat[i].isRef) {
                double ex = 0, ey = 0;
                if (meas[idx + 4] != 0) { ex = meas[idx + 4] - hmu[idx + 4]; ey = meas[idx + 5] - hmu[idx + 5]; }
                if (ex != 0) { e[me++] = ex; e[me++] = ey; }
                idx += 6;
            } else {
                idx += 4;
            }
        }
        if (me != m) { free(rd); free(K); free(e); rc = -2; goto done; }   /* K*eeMat size mismatch */
        /* kx = K*e ; kmh = States(kx) (states.cpp:24-37) ; gate (:230-232) */
        double *kx = (double *)malloc((size_t)n * sizeof(double));
        mm(kx, K, e, n, m, 1);
        double d = fabs(f->V[0] - kx[3]) + fabs(f->V[1] - kx[4]) + fabs(f->V[2] - kx[5]);
        int ok = d < f->prm.gate;
        if (ok) {
            for (int i = 0; i < 3; ++i) { f->X[i] += kx[i]; f->V[i] += kx[3 + i]; }
            for (int i = 0; i < nf; ++i)
                for (int c = 0; c < 3; ++c) f->feat[i].pib[c] += kx[6 + 3 * i + c];
        }
        if (applied) *applied = ok;
        free(rd); free(K); free(e); free(kx);
    }
done:
    free(F); free(qd); free(meas); free(hmu); free(H); free(refFlag);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* text front end                                                            */
/* ------------------------------------------------------------------------ */

/* Sensor<T>::set_value(ts, v)  include/Sensor.hpp:41-46 */
typedef struct { double timestamp, dt; } chan;
static void chan_set(chan *c, double ts)
{
    c->dt = (c->timestamp == 0) ? 0 : ts - c->timestamp;
    c->timestamp = ts;
}

typedef struct {
    FILE *fp;
    double t, time;
    char type[32], vals[256];
    int rv;
    chan quat, acc, ang, alt, img, init, vel, pos;
    orc_sensors v;
} sens_reader;

/* Sensors::update  src/ARC/Sensors.cpp:4-70 */
static int sens_update(sens_reader *r)
{
    int updated = 0;
    if (r->time == -1)
        r->rv = fscanf(r->fp, "%lf,%31[^,\n],%255s", &r->time, r->type, r->vals);
    r->t = r->time;
    while (r->time - r->t < 0.005) {                        /* EPS, include/Sensors.hpp:8 */
        if (r->rv == EOF) return -1;
        double a[4] = { 0, 0, 0, 0 };
        if (!strcmp(r->type, "ACC")) {
            updated |= ORC_ACC;
            sscanf(r->vals, "%lf,%lf,%lf", &a[0], &a[1], &a[2]);
            chan_set(&r->acc, r->time); memcpy(r->v.acc, a, 3 * sizeof(double));
        } else if (!strcmp(r->type, "ANG")) {
            updated |= ORC_ANG;
            sscanf(r->vals, "%lf,%lf,%lf", &a[0], &a[1], &a[2]);
            chan_set(&r->ang, r->time); memcpy(r->v.ang, a, 3 * sizeof(double));
        } else if (!strcmp(r->type, "IMG")) {
            updated |= ORC_IMG;
            chan_set(&r->img, r->time);
        } else if (!strcmp(r->type, "QUAT")) {
            updated |= ORC_QUAT;
            sscanf(r->vals, "%lf,%lf,%lf,%lf", &a[0], &a[1], &a[2], &a[3]);
            chan_set(&r->quat, r->time); memcpy(r->v.quat, a, 4 * sizeof(double));
        } else if (!strcmp(r->type, "ALT")) {
            updated |= ORC_ALT;
            sscanf(r->vals, "%lf", &a[0]);
            chan_set(&r->alt, r->time); r->v.alt = a[0];
        } else if (!strcmp(r->type, "POS")) {
            updated |= ORC_POS;
            sscanf(r->vals, "%lf,%lf,%lf", &a[0], &a[1], &a[2]);
            chan_set(&r->pos, r->time); memcpy(r->v.pos, a, 3 * sizeof(double));
        } else if (!strcmp(r->type, "VEL")) {
            updated |= ORC_VEL;
            sscanf(r->vals, "%lf,%lf,%lf", &a[0], &a[1], &a[2]);
            chan_set(&r->vel, r->time); memcpy(r->v.vel, a, 3 * sizeof(double));
        } else if (!strcmp(r->type, "INIT")) {
            updated |= ORC_INIT;
            sscanf(r->vals, "%lf,%lf,%lf", &a[0], &a[1], &a[2]);
            chan_set(&r->init, r->time); memcpy(r->v.init, a, 3 * sizeof(double));
        }
        r->rv = fscanf(r->fp, "%lf,%31[^,\n],%255s", &r->time, r->type, r->vals);
    }
    return updated;
}

/* ImageSensor::get_projections  src/ARC/imagesensor.cpp:29-49 with
 * FeatureIO::get_val  src/ARC/featureIO.cpp:104-128 */
static int read_frame(FILE *fp, orc_match **out, int *cap)
{
    int n = 0;
    char line[1024];
    for (;;) {
        orc_match pj;
        memset(&pj, 0, sizeof pj);                          /* reflection = NONREF */
        if (fgets(line, sizeof line, fp) == NULL) break;    /* -2: EOF */
        int rv = sscanf(line, "%d,%lf,%lf,%lf,%lf", &pj.id, &pj.sx, &pj.sy, &pj.rx, &pj.ry);
        if (rv < 0) break;                                  /* blank line: end of frame */
        if (n == *cap) {
            *cap = *cap ? 2 * *cap : 64;
            *out = (orc_match *)realloc(*out, (size_t)*cap * sizeof(orc_match));
        }
        (*out)[n++] = pj;
    }
    return n;
}

static void dump_state(FILE *d, int iter, const orc_filter *f)
{
    fprintf(d, "STEP %d %d\n", iter, f->nf);
    fprintf(d, "%a %a %a\n", f->X[0], f->X[1], f->X[2]);
    fprintf(d, "%a %a %a\n", f->V[0], f->V[1], f->V[2]);
    for (int i = 0; i < f->nf; ++i)
        fprintf(d, "%d %a %a %a\n", f->feat[i].id, f->feat[i].pib[0], f->feat[i].pib[1], f->feat[i].pib[2]);
    for (int i = 0; i < f->n; ++i) {
        for (int j = 0; j < f->n; ++j) fprintf(d, j ? " %a" : "%a", f->P[(size_t)i * f->n + j]);
        fputc('\n', d);
    }
}

/* main()  src/slam/main.cpp:106-238 (argument handling is the caller's) */
int orc_slam_run(FILE *sensors, FILE *features, FILE *out, FILE *err, FILE *dump,
                 const orc_params *p)
{
    orc_filter *f = orc_filter_new(p);
    sens_reader r;
    memset(&r, 0, sizeof r);
    r.fp = sensors;
    r.time = -1;                                            /* Sensors() ctor */
    orc_match *matches = NULL;
    int mcap = 0, nmatch = 0;
    double old_pos[3];
    int status = 0;

    int u = sens_update(&r);                                /* :131 */
    if (u & ORC_INIT) {                                     /* :132-137 */
        f->X[0] = r.v.init[1]; f->X[1] = r.v.init[0]; f->X[2] = r.v.init[2];
    }
    if (!(u & ORC_IMG)) {                                   /* :138-141 */
        fprintf(err, "No image measurement at first time step.\n");
        status = 1;
        goto end;
    }
    nmatch = read_frame(features, &matches, &mcap);         /* :142 */
    orc_update_features(f, matches, nmatch, r.v.quat);      /* :143 */
    memcpy(old_pos, f->X, sizeof old_pos);                  /* :144 */

    int iter = 0;
    double dt = 0;   /* the reference leaves dt uninitialised when none of ACC/ANG/QUAT/ALT arrives */
    while (u != -1) {                                       /* :150 */
        u = sens_update(&r);                                /* :157 */
        if (u & ORC_ACC) dt = r.acc.dt;                     /* :158-166 */
        else if (u & ORC_ANG) dt = r.ang.dt;
        else if (u & ORC_QUAT) dt = r.quat.dt;
        else if (u & ORC_ALT) dt = r.alt.dt;
        else if (u == 0) {
            fprintf(err, "No update, so what are we doing here?\n");
            status = 1;
            goto end;
        }
        if (u & ORC_IMG) nmatch = read_frame(features, &matches, &mcap);   /* :169 */
        double xy[2];
        int applied;
        /* the dump hook sits at the printf, i.e. between predict and update:
         * run the step in two halves by snapshotting through a callback-free
         * trick: orc_step fills xy at the printf point; for the dump we need
         * P there as well, so replay the predict half on a copy. */
        if (dump) {
            orc_filter *c = orc_filter_new(&f->prm);
            free(c->P);
            *c = *f;
            c->feat = (orc_feature *)malloc((size_t)(f->cap ? f->cap : 1) * sizeof(orc_feature));
            memcpy(c->feat, f->feat, (size_t)f->nf * sizeof(orc_feature));
            c->P = (double *)malloc((size_t)f->n * f->n * sizeof(double));
            memcpy(c->P, f->P, (size_t)f->n * f->n * sizeof(double));
            orc_sensors sv = r.v;
            double op[3];
            memcpy(op, old_pos, sizeof op);
            /* predict-only replay: same calls as orc_step without the IMG update half */
            if (u & ORC_IMG) { memcpy(op, c->X, sizeof op); orc_update_features(c, matches, nmatch, sv.quat); }
            if (u & ORC_POS) { c->X[0] = sv.pos[1]; c->X[1] = sv.pos[0]; c->X[2] = sv.pos[2]; }
            if (u & ORC_VEL) {
                double Rb[9], Rt[9], v[3];
                orc_rotation(sv.quat, Rb); mat3_t(Rb, Rt); mat3_vec(Rt, sv.vel, v);
                memcpy(c->V, v, sizeof v);
            }
            double *F = (double *)malloc((size_t)c->n * c->n * sizeof(double));
            double *qd = (double *)malloc((size_t)c->n * sizeof(double));
            orc_motion_jacobian(c, &sv, dt, F);
            orc_dynamics_step(c, &sv, dt);
            orc_init_q_diag(c->nf, c->prm.Q0, c->prm.qbias, dt, qd);
            orc_calc_p(c->P, F, qd, c->n);
            dump_state(dump, iter, c);
            free(F); free(qd);
            orc_filter_free(c);
        }
        int rc = orc_step(f, u, &r.v, dt, old_pos, matches, nmatch, xy, &applied);
        if (rc == -1) {
            fprintf(err, "Cannot find match.\n");           /* :372-375 */
            status = 1;
            goto end;
        }
        fprintf(out, "%0.5f,%0.5f\n", xy[0], xy[1]);        /* :207 */
        if (rc == -2) {
            fprintf(err, "measurement size mismatch (the reference aborts with a cv::Exception here)\n");
            status = 134;
            goto end;
        }
        ++iter;
    }
end:
    free(matches);
    orc_filter_free(f);
    return status;
}

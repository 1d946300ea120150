/* ORACLE (test infrastructure, NOT product code) -- plain-C fp64 restatement of the corenav-GP
 * slip-GP fit/predict at fixed hyper-parameters.  It is the checker for the HIP path and the
 * "port" CPU baseline that bench.py times beside it; nothing under corenav_gp_amd/ links it.
 *
 * PARITY UNPINNED at the GPy boundary (the reference's arithmetic is inside un-vendored GPy,
 * /root/reference/core_navigation/script/gp_slip_node.py:3,31-36,45-49; no reference test pins a
 * number).  This file is pinned against tests/golden/ (scikit-learn GPR + closed forms) and against
 * oracle/gp_oracle.py by tests/test_oracle.py.
 *
 * Restated steps (reference file:line, GPy internals as documented in SURVEY.md 3B):
 *   gram      K = k(X,X) + (sigma_n^2 + 1e-8) I      gp_slip_node.py:31,35  (kern.K, inference)
 *   potrf     Ky = L L^T, jitter retry <= 5           gp_slip_node.py:35-36  (jitchol/dpotrf)
 *   potrs     alpha = Ky^-1 y                         gp_slip_node.py:36     (dpotrs)
 *   logml     0.5(-N log 2pi - 2 sum log L_ii - y'a)  gp_slip_node.py:36     (objective)
 *   predict   mu = k*' alpha ; var = k** - |L^-1 k*|^2, clip 1e-15, + sigma_n^2   gp_slip_node.py:45-49
 * Storage is row-major, lower triangle; single thread (catkin builds carry no OpenMP flag).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define K_SE_ISO 0
#define K_SE_ARD 1
#define K_RBF_BROWNIAN 2
#define GPY_DIAG_EPS 1e-8
#define GPY_VAR_FLOOR 1e-15
#define NB 96

static int n_theta(int kid, int d) { return kid == K_SE_ISO ? 3 : (kid == K_SE_ARD ? d + 2 : 4); }

static double sgn(double x) { return (x > 0.0) - (x < 0.0); }

/* one covariance entry; `same` marks the auto-covariance diagonal (GPy forces r^2 = 0 there) */
static double kfun(int kid, const double *th, int d, const double *a, const double *b, int same) {
    if (kid == K_SE_ISO || kid == K_SE_ARD) {
        double d2 = 0.0;
        for (int k = 0; k < d; ++k) {
            double ell = (kid == K_SE_ISO) ? th[1] : th[1 + k];
            double t = a[k] / ell - b[k] / ell;
            d2 += t * t;
        }
        return th[0] * exp(-0.5 * d2);
    }
    /* RBF x Brownian, d == 1: GPy r^2 = x^2 + x'^2 - 2 x x' clipped at 0 */
    double x = a[0], xp = b[0];
    double r2 = same ? 0.0 : (-2.0 * x * xp + (x * x + xp * xp));
    if (r2 < 0.0) r2 = 0.0;
    double r = sqrt(r2) / th[1];
    double krbf = th[0] * exp(-0.5 * r * r);
    double kbr = (sgn(x) == sgn(xp)) ? th[2] * fmin(fabs(x), fabs(xp)) : 0.0;
    return krbf * kbr;
}

static double kdiag(int kid, const double *th, const double *a) {
    return (kid == K_RBF_BROWNIAN) ? th[0] * th[2] * fabs(a[0]) : th[0];
}

/* dot product of two contiguous rows */
static inline double dotp(const double *a, const double *b, int n) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int k = 0;
    for (; k + 4 <= n; k += 4) {
        s0 += a[k] * b[k]; s1 += a[k + 1] * b[k + 1]; s2 += a[k + 2] * b[k + 2]; s3 += a[k + 3] * b[k + 3];
    }
    for (; k < n; ++k) s0 += a[k] * b[k];
    return (s0 + s1) + (s2 + s3);
}

/* blocked right-looking Cholesky, lower, in place, row-major.  returns 0 or 1-based failing pivot */
int oracle_potrf(double *A, int n, int lda) {
    for (int k0 = 0; k0 < n; k0 += NB) {
        int kb = (n - k0 < NB) ? n - k0 : NB;
        /* diagonal block (unblocked, row by row) */
        for (int j = k0; j < k0 + kb; ++j) {
            double *rj = A + (size_t)j * lda;
            for (int c = k0; c < j; ++c) {
                const double *rc = A + (size_t)c * lda;
                rj[c] = (rj[c] - dotp(rj + k0, rc + k0, c - k0)) / rc[c];
            }
            double djj = rj[j] - dotp(rj + k0, rj + k0, j - k0);
            if (!(djj > 0.0)) return j + 1;
            rj[j] = sqrt(djj);
        }
        /* panel: rows below solve against the diagonal block */
        for (int i = k0 + kb; i < n; ++i) {
            double *ri = A + (size_t)i * lda;
            for (int c = k0; c < k0 + kb; ++c) {
                const double *rc = A + (size_t)c * lda;
                ri[c] = (ri[c] - dotp(ri + k0, rc + k0, c - k0)) / rc[c];
            }
        }
        /* trailing symmetric rank-kb update, 2x4 register blocking over contiguous row segments */
        int t0 = k0 + kb;
        for (int i = t0; i < n; i += 2) {
            int i2 = (i + 1 < n);
            const double *a0 = A + (size_t)i * lda + k0;
            const double *a1 = A + (size_t)(i + i2) * lda + k0;
            int jmax = i + i2;
            for (int j = t0; j <= jmax; j += 4) {
                int nj = (jmax - j + 1 < 4) ? jmax - j + 1 : 4;
                const double *b0 = A + (size_t)j * lda + k0;
                const double *b1 = A + (size_t)(j + (nj > 1)) * lda + k0;
                const double *b2 = A + (size_t)(j + (nj > 2 ? 2 : 0)) * lda + k0;
                const double *b3 = A + (size_t)(j + (nj > 3 ? 3 : 0)) * lda + k0;
                double c00 = 0, c01 = 0, c02 = 0, c03 = 0, c10 = 0, c11 = 0, c12 = 0, c13 = 0;
                for (int k = 0; k < kb; ++k) {
                    double x0 = a0[k], x1 = a1[k];
                    c00 += x0 * b0[k]; c01 += x0 * b1[k]; c02 += x0 * b2[k]; c03 += x0 * b3[k];
                    c10 += x1 * b0[k]; c11 += x1 * b1[k]; c12 += x1 * b2[k]; c13 += x1 * b3[k];
                }
                double *o0 = A + (size_t)i * lda, *o1 = A + (size_t)(i + 1) * lda;
                if (j <= i) o0[j] -= c00;
                if (nj > 1 && j + 1 <= i) o0[j + 1] -= c01;
                if (nj > 2 && j + 2 <= i) o0[j + 2] -= c02;
                if (nj > 3 && j + 3 <= i) o0[j + 3] -= c03;
                if (i2) {
                    o1[j] -= c10;
                    if (nj > 1) o1[j + 1] -= c11;
                    if (nj > 2) o1[j + 2] -= c12;
                    if (nj > 3) o1[j + 3] -= c13;
                }
            }
        }
    }
    return 0;
}

/* Build Ky (lower triangle incl. diagonal) row-major n x n */
void oracle_gram(int kid, const double *th, int n, int d, const double *X, double diag_add, double *A) {
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < i; ++j) A[(size_t)i * n + j] = kfun(kid, th, d, X + (size_t)i * d, X + (size_t)j * d, 0);
        A[(size_t)i * n + i] = kfun(kid, th, d, X + (size_t)i * d, X + (size_t)i * d, 1) + diag_add;
    }
}

/* Full fixed-theta fit + predict.  X (N,d) row-major, Xs (M,d) row-major.
 * Outputs (any may be NULL): mean[M], var[M], logml[1], alpha[N], Lout[N*N] row-major lower,
 * jitter[1] (the GPy jitter that made the factorisation succeed, 0 if none).
 * Returns 0, or the 1-based failing pivot after the jitter policy is exhausted, or -1 on bad args. */
int oracle_fit_predict(int kid, const double *theta, int N, int d, const double *X, const double *y,
                       int M, const double *Xs, int include_noise, double *mean, double *var,
                       double *logml, double *alpha, double *Lout, double *jitter_out) {
    if (N <= 0 || d <= 0 || kid < 0 || kid > 2 || (kid == K_RBF_BROWNIAN && d != 1)) return -1;
    const double noise = theta[n_theta(kid, d) - 1];
    double *A = (double *)malloc((size_t)N * N * sizeof(double));
    double *z = (double *)malloc((size_t)N * sizeof(double));
    double *al = (double *)malloc((size_t)N * sizeof(double));
    if (!A || !z || !al) { free(A); free(z); free(al); return -1; }
    /* jitchol: plain attempt, then mean(diag)*1e-6*10^k for k = 0..4 */
    double jitter = 0.0, meandiag = 0.0;
    int info = 0;
    for (int attempt = 0; attempt <= 5; ++attempt) {
        oracle_gram(kid, theta, N, d, X, noise + GPY_DIAG_EPS + jitter, A);
        if (attempt == 0) {
            for (int i = 0; i < N; ++i) meandiag += A[(size_t)i * N + i];
            meandiag /= N;
        }
        info = oracle_potrf(A, N, N);
        if (info == 0) break;
        jitter = (attempt == 0) ? meandiag * 1e-6 : jitter * 10.0;
    }
    if (jitter_out) *jitter_out = (info == 0) ? jitter : -1.0;
    if (info != 0) { free(A); free(z); free(al); return info; }
    /* potrs: L z = y ; L^T alpha = z */
    double logdet = 0.0;
    for (int i = 0; i < N; ++i) {
        const double *ri = A + (size_t)i * N;
        z[i] = (y[i] - dotp(ri, z, i)) / ri[i];
        logdet += log(ri[i]);
    }
    memcpy(al, z, (size_t)N * sizeof(double));
    for (int i = N - 1; i >= 0; --i) {
        const double *ri = A + (size_t)i * N;
        al[i] /= ri[i];
        const double ai = al[i];
        for (int j = 0; j < i; ++j) al[j] -= ri[j] * ai;
    }
    double yta = dotp(y, al, N);
    if (logml) *logml = 0.5 * (-(double)N * log(2.0 * M_PI) - 2.0 * logdet - yta);
    if (alpha) memcpy(alpha, al, (size_t)N * sizeof(double));
    if (Lout) {
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) Lout[(size_t)i * N + j] = (j <= i) ? A[(size_t)i * N + j] : 0.0;
    }
    /* predict: four test points at a time share the rows of L */
    if (M > 0 && (mean || var)) {
        double *v = (double *)malloc((size_t)4 * N * sizeof(double));
        for (int m0 = 0; m0 < M; m0 += 4) {
            int nm = (M - m0 < 4) ? M - m0 : 4;
            for (int q = 0; q < nm; ++q) {
                const double *xs = Xs + (size_t)(m0 + q) * d;
                double *vq = v + (size_t)q * N;
                double mu = 0.0;
                for (int i = 0; i < N; ++i) {
                    vq[i] = kfun(kid, theta, d, X + (size_t)i * d, xs, 0);
                    mu += vq[i] * al[i];
                }
                if (mean) mean[m0 + q] = mu;
            }
            if (var) {
                for (int i = 0; i < N; ++i) {
                    const double *ri = A + (size_t)i * N;
                    for (int q = 0; q < nm; ++q) {
                        double *vq = v + (size_t)q * N;
                        vq[i] = (vq[i] - dotp(ri, vq, i)) / ri[i];
                    }
                }
                for (int q = 0; q < nm; ++q) {
                    const double *vq = v + (size_t)q * N;
                    double s = kdiag(kid, theta, Xs + (size_t)(m0 + q) * d) - dotp(vq, vq, N);
                    if (s < GPY_VAR_FLOOR) s = GPY_VAR_FLOOR;
                    var[m0 + q] = include_noise ? s + noise : s;
                }
            }
        }
        free(v);
    }
    free(A); free(z); free(al);
    return 0;
}

/* ORACLE (test infrastructure, NOT product code) -- the "Eigen::LLT-class" single-thread CPU row of bench.py's
 * cpu_baseline: the same fixed-theta fit + predict as oracle/gp_oracle.c (whose header lists the reference lines each
 * step restates: gp_slip_node.py:31-36,45-49 through GPy's kern.K / jitchol / dpotrs / predict), with the dense linear
 * algebra done by LAPACK / BLAS -- dpotrf, dpotrs, dtrsm, dgemv of the OpenBLAS that scipy bundles, dlopen'ed at run time
 * and held to ONE thread -- and a Gram loop a CPU person would accept: structure-of-arrays inputs, the inner loop over
 * the row's columns, an exp that vectorises (no libm call: range reduction + degree-12 polynomial + exponent add, the
 * GPU path's exp_nonpos in C).  gp_oracle.py's numpy path spends three quarters of a fit building an N x N x d
 * difference tensor; this row is what the GPU is priced against.  Checked against gp_oracle.c / the golden fixtures in
 * tests/test_oracle.py.  Nothing under corenav_gp_amd/ links it.
 *
 * Storage: A row-major lower = column-major UPPER for LAPACK: dpotrf('U') gives U with U^T U = Ky, i.e. L = U^T in place. */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define K_SE_ISO 0
#define K_SE_ARD 1
#define K_RBF_BROWNIAN 2
#define GPY_DIAG_EPS 1e-8
#define GPY_VAR_FLOOR 1e-15

typedef void (*dpotrf_t)(const char *, const int *, double *, const int *, int *);
typedef void (*dpotrs_t)(const char *, const int *, const int *, const double *, const int *, double *, const int *, int *);
typedef void (*dtrsm_t)(const char *, const char *, const char *, const char *, const int *, const int *, const double *,
                        const double *, const int *, double *, const int *);
typedef void (*dgemv_t)(const char *, const int *, const int *, const double *, const double *, const int *, const double *,
                        const int *, const double *, double *, const int *);
typedef void (*setthr_t)(int);

static dpotrf_t p_dpotrf;
static dpotrs_t p_dpotrs;
static dtrsm_t p_dtrsm;
static dgemv_t p_dgemv;

static void *sym2(void *h, const char *a, const char *b) {
    void *p = dlsym(h, a);
    return p ? p : dlsym(h, b);
}

/* dlopen the BLAS/LAPACK library at `path` (scipy's bundled OpenBLAS: symbols carry a scipy_ prefix) and hold it to one
 * thread.  0 on success. */
int oracle_lapack_init(const char *path) {
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1;
    p_dpotrf = (dpotrf_t)sym2(h, "scipy_dpotrf_", "dpotrf_");
    p_dpotrs = (dpotrs_t)sym2(h, "scipy_dpotrs_", "dpotrs_");
    p_dtrsm = (dtrsm_t)sym2(h, "scipy_dtrsm_", "dtrsm_");
    p_dgemv = (dgemv_t)sym2(h, "scipy_dgemv_", "dgemv_");
    setthr_t st = (setthr_t)sym2(h, "scipy_openblas_set_num_threads", "openblas_set_num_threads");
    if (!p_dpotrf || !p_dpotrs || !p_dtrsm || !p_dgemv) return -2;
    if (st) st(1);
    return 0;
}

static int n_theta(int kid, int d) { return kid == K_SE_ISO ? 3 : (kid == K_SE_ARD ? d + 2 : 4); }

/* exp(x) for x <= 0: n = rint(x log2 e), r = x - n ln2 (two pieces), degree-12 Horner on |r| <= ln2/2, 2^n by an add
 * into the exponent field (x clamped at -700: the result underflows to 0 there anyway).  Branch-free: gcc vectorises the
 * loops that call it. */
static inline double exp_np(double x) {
    x = x < -700.0 ? -700.0 : x;
    const double n = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(n, -6.93147180369123816490e-01, x);
    r = __builtin_fma(n, -1.90821492927058770002e-10, r);
    double q = 2.08767569878681e-09;
    q = __builtin_fma(q, r, 2.505210838544172e-08);
    q = __builtin_fma(q, r, 2.755731922398589e-07);
    q = __builtin_fma(q, r, 2.7557319223985893e-06);
    q = __builtin_fma(q, r, 2.48015873015873e-05);
    q = __builtin_fma(q, r, 1.984126984126984e-04);
    q = __builtin_fma(q, r, 1.3888888888888889e-03);
    q = __builtin_fma(q, r, 8.333333333333333e-03);
    q = __builtin_fma(q, r, 4.1666666666666664e-02);
    q = __builtin_fma(q, r, 1.6666666666666666e-01);
    q = __builtin_fma(q, r, 0.5);
    q = __builtin_fma(q, r, 1.0);
    q = __builtin_fma(q, r, 1.0);
    union { double d; int64_t i; } u;
    u.d = q;
    u.i += (int64_t)n << 52;
    return u.d;
}

/* one row of covariances k(a, X[j]) for j in [0, cnt): Z = inputs scaled by 1 / ell, structure of arrays [d][ldz] */
static void krow(int kid, const double *th, int d, const double *za, double xa_raw, const double *Z, const double *Xraw, int ldz,
                 int cnt, double *out) {
    if (kid != K_RBF_BROWNIAN) {
        for (int j = 0; j < cnt; ++j) out[j] = 0.0;
        for (int k = 0; k < d; ++k) {
            const double a = za[k];
            const double *zk = Z + (size_t)k * ldz;
            for (int j = 0; j < cnt; ++j) {
                const double t = a - zk[j];
                out[j] += t * t;
            }
        }
        const double amp = th[0];
        for (int j = 0; j < cnt; ++j) out[j] = amp * exp_np(-0.5 * out[j]);
        return;
    }
    /* RBF x Brownian, d = 1: GPy r^2 = x^2 + x'^2 - 2 x x' clipped at 0; Brownian sigma_b^2 min(|x|, |x'|) where the signs agree */
    const double x = xa_raw, iell = 1.0 / th[1], amp = th[0], ab = th[2];
    for (int j = 0; j < cnt; ++j) {
        const double xp = Xraw[j];
        double r2 = -2.0 * x * xp + (x * x + xp * xp);
        r2 = r2 < 0.0 ? 0.0 : r2;
        const double r = sqrt(r2) * iell;
        const double sx = (double)((x > 0.0) - (x < 0.0)), sp = (double)((xp > 0.0) - (xp < 0.0));
        const double kb = (sx == sp) ? ab * fmin(fabs(x), fabs(xp)) : 0.0;
        out[j] = amp * exp_np(-0.5 * r * r) * kb;
    }
}

/* Same contract as oracle_fit_predict (gp_oracle.c).  dpotrf_s (optional): seconds spent inside dpotrf of the successful
 * attempt, for the "dpotrf-only GFLOP/s" of the baseline line. */
int oracle_fit_predict_lapack(int kid, const double *theta, int N, int d, const double *X, const double *y, int M,
                              const double *Xs, int include_noise, double *mean, double *var, double *logml, double *alpha,
                              double *jitter_out, double *dpotrf_s) {
    if (!p_dpotrf) return -2;
    if (N <= 0 || d <= 0 || kid < 0 || kid > 2 || (kid == K_RBF_BROWNIAN && d != 1)) return -1;
    const double noise = theta[n_theta(kid, d) - 1];
    const int brown = kid == K_RBF_BROWNIAN;
    double *A = (double *)malloc((size_t)N * N * sizeof(double));
    double *Z = (double *)malloc((size_t)d * N * sizeof(double));   /* [d][N] scaled inputs */
    double *al = (double *)malloc((size_t)N * sizeof(double));
    double *za = (double *)malloc((size_t)d * sizeof(double));
    if (!A || !Z || !al || !za) { free(A); free(Z); free(al); free(za); return -1; }
    for (int k = 0; k < d; ++k) {
        const double iell = brown ? 1.0 : 1.0 / ((kid == K_SE_ISO) ? theta[1] : theta[1 + k]);
        for (int i = 0; i < N; ++i) Z[(size_t)k * N + i] = X[(size_t)i * d + k] * iell;
    }
    double jitter = 0.0, meandiag = 0.0;
    int info = 0;
    const char U = 'U', L = 'L', T = 'T', Nn = 'N';
    const int one = 1;
    const double done = 1.0, dzero = 0.0;
    for (int attempt = 0; attempt <= 5; ++attempt) {   /* jitchol: plain attempt, then mean(diag) 1e-6 10^k, k = 0..4 */
        for (int i = 0; i < N; ++i) {
            for (int k = 0; k < d; ++k) za[k] = Z[(size_t)k * N + i];
            krow(kid, theta, d, za, X[(size_t)i * d], Z, Z, N, i + 1, A + (size_t)i * N);   /* Brownian: Z row 0 holds the raw ticks */
            if (brown) A[(size_t)i * N + i] = theta[0] * theta[2] * fabs(X[i]);             /* GPy forces r^2 = 0 on the diagonal */
            A[(size_t)i * N + i] += noise + GPY_DIAG_EPS + jitter;
        }
        if (attempt == 0) {
            for (int i = 0; i < N; ++i) meandiag += A[(size_t)i * N + i];
            meandiag /= N;
        }
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        p_dpotrf(&U, &N, A, &N, &info);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (dpotrf_s) *dpotrf_s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (info == 0) break;
        jitter = (attempt == 0) ? meandiag * 1e-6 : jitter * 10.0;
    }
    if (jitter_out) *jitter_out = (info == 0) ? jitter : -1.0;
    if (info != 0) { free(A); free(Z); free(al); free(za); return info; }
    double logdet = 0.0;
    for (int i = 0; i < N; ++i) logdet += log(A[(size_t)i * N + i]);
    memcpy(al, y, (size_t)N * sizeof(double));
    int info2 = 0;
    p_dpotrs(&U, &N, &one, A, &N, al, &N, &info2);
    double yta = 0.0;
    for (int i = 0; i < N; ++i) yta += y[i] * al[i];
    if (logml) *logml = 0.5 * (-(double)N * log(2.0 * M_PI) - 2.0 * logdet - yta);
    if (alpha) memcpy(alpha, al, (size_t)N * sizeof(double));
    if (M > 0 && (mean || var)) {
        double *Ks = (double *)malloc((size_t)M * N * sizeof(double));   /* [M][N] = column-major N x M */
        if (!Ks) { free(A); free(Z); free(al); free(za); return -1; }
        for (int m = 0; m < M; ++m) {
            for (int k = 0; k < d; ++k) {
                const double iell = brown ? 1.0 : 1.0 / ((kid == K_SE_ISO) ? theta[1] : theta[1 + k]);
                za[k] = Xs[(size_t)m * d + k] * iell;
            }
            krow(kid, theta, d, za, Xs[(size_t)m * d], Z, Z, N, N, Ks + (size_t)m * N);
        }
        if (mean) p_dgemv(&T, &N, &M, &done, Ks, &N, al, &one, &dzero, mean, &one);   /* mean = Ks^T alpha */
        if (var) {
            p_dtrsm(&L, &U, &T, &Nn, &N, &M, &done, A, &N, Ks, &N);                  /* V = U^-T Ks = L^-1 Ks, all M at once */
            for (int m = 0; m < M; ++m) {
                const double *v = Ks + (size_t)m * N;
                double q = 0.0;
                for (int i = 0; i < N; ++i) q += v[i] * v[i];
                const double kss = brown ? theta[0] * theta[2] * fabs(Xs[m]) : theta[0];
                double s = kss - q;
                if (s < GPY_VAR_FLOOR) s = GPY_VAR_FLOOR;
                var[m] = include_noise ? s + noise : s;
            }
        }
        free(Ks);
    }
    free(A); free(Z); free(al); free(za);
    return 0;
}

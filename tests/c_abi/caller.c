/* A plain C caller of the C ABI (include/corenav_gp.h), built with `gcc -std=c11 -pedantic -Werror` and linked against
 * libcorenav_gp.so -- the shape of the call the reference's C++ host makes (gp_predictor/src/gp_predictor.cpp:180-190: a
 * compiled node that owns its buffers and calls straight into the library), with no Python and no ctypes in between.
 *   caller <fixture.bin> [fp32]
 * fixture.bin (written by tests/test_c_abi.py from a golden .npz, little-endian doubles):
 *   header  N d M kid ntheta            (five doubles holding integers)
 *   theta[ntheta] X[N d] y[N] Xs[M d] mean[M] var_latent[M] logml alpha[N]
 * Runs cgp_create_ex -> cgp_fit -> cgp_predict -> cgp_get_alpha on it, then the same window four times through
 * cgp_fit_predict_batch and through cgp_sweep_fit_predict over {device 0, device 0}, and checks every output against the
 * fixture's expected values at 1e-6 (1e-3 with `fp32`).  Exit code 0 = all within tolerance; prints what it compared. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "corenav_gp.h"

static double relmax(const double *a, const double *b, int n) {
  double e = 0.0, s = 0.0;
  int i;
  for (i = 0; i < n; ++i) {
    if (fabs(a[i] - b[i]) > e) e = fabs(a[i] - b[i]);
    if (fabs(b[i]) > s) s = fabs(b[i]);
  }
  return e / (s > 0.0 ? s : 1.0);
}

static double releach(const double *a, const double *b, int n) {
  double e = 0.0;
  int i;
  for (i = 0; i < n; ++i)
    if (fabs(a[i] - b[i]) / fabs(b[i]) > e) e = fabs(a[i] - b[i]) / fabs(b[i]);
  return e;
}

#define CHECK(cond, what)                                     \
  do {                                                        \
    if (!(cond)) {                                            \
      fprintf(stderr, "caller.c: FAILED %s (line %d)\n", what, __LINE__); \
      return 1;                                               \
    }                                                         \
  } while (0)

int main(int argc, char **argv) {
  FILE *f;
  double hdr[5], *buf, *theta, *X, *y, *Xs, *emean, *evar, *ealpha, elogml;
  double *mean, *var, *alpha, *bX, *by, *bXs, *bth, *bmean, *bvar, blogml[4], summary[12], logml = 0.0, tol;
  int N, d, M, kid, nth, status = 0, rc, i, b, info[4], devices[2] = {0, 0}, dtype, s0, s1;
  long total;
  cgp_ctx *ctx;
  cgp_sweep *sw;
  if (argc < 2) {
    fprintf(stderr, "usage: caller <fixture.bin> [fp32]\n");
    return 2;
  }
  dtype = (argc > 2 && strcmp(argv[2], "fp32") == 0) ? CGP_F32 : CGP_F64;
  tol = dtype == CGP_F32 ? 1e-3 : 1e-6;
  f = fopen(argv[1], "rb");
  CHECK(f != NULL, "open fixture");
  CHECK(fread(hdr, sizeof(double), 5, f) == 5, "read header");
  N = (int)hdr[0], d = (int)hdr[1], M = (int)hdr[2], kid = (int)hdr[3], nth = (int)hdr[4];
  total = (long)nth + (long)N * d + N + (long)M * d + M + M + 1 + N;
  buf = (double *)malloc(sizeof(double) * (size_t)total);
  CHECK(buf != NULL && fread(buf, sizeof(double), (size_t)total, f) == (size_t)total, "read fixture");
  fclose(f);
  theta = buf, X = theta + nth, y = X + (long)N * d, Xs = y + N, emean = Xs + (long)M * d, evar = emean + M;
  elogml = evar[M], ealpha = evar + M + 1;

  CHECK(cgp_abi_version() == CGP_ABI_VERSION, "ABI revision of the loaded library = the header's");
  ctx = cgp_create_ex(0, N, M, d, 4, dtype, &status);
  if (!ctx) {
    fprintf(stderr, "caller.c: cgp_create_ex failed: %s\n", cgp_strerror(status));
    return 3;
  }
  mean = (double *)malloc(sizeof(double) * (size_t)M);
  var = (double *)malloc(sizeof(double) * (size_t)M);
  alpha = (double *)malloc(sizeof(double) * (size_t)N);

  /* one window: fit -> predict (latent variance: include_noise = 0) -> alpha */
  rc = cgp_fit(ctx, X, y, N, d, kid, theta, &logml);
  CHECK(rc == 0, "cgp_fit");
  CHECK(cgp_predict(ctx, Xs, M, 0, mean, var) == 0, "cgp_predict");
  CHECK(cgp_get_alpha(ctx, alpha) == 0, "cgp_get_alpha");
  printf("single window  N=%d d=%d M=%d kernel=%d %s: logml %.10g (expected %.10g)  mean err %.2e  var err %.2e  alpha err %.2e\n", N, d, M,
         kid, dtype == CGP_F32 ? "fp32" : "fp64", logml, elogml, relmax(mean, emean, M), releach(var, evar, M), relmax(alpha, ealpha, N));
  CHECK(fabs(logml - elogml) <= tol * fabs(elogml), "logml");
  CHECK(relmax(mean, emean, M) < tol, "mean");
  CHECK(releach(var, evar, M) < tol, "variance");
  CHECK(relmax(alpha, ealpha, N) < (dtype == CGP_F32 ? 1e-2 : 1e-6), "alpha");

  /* the same window four times as a batch, through the context and through the two-shard sweep */
  bX = (double *)malloc(sizeof(double) * 4 * (size_t)N * d);
  by = (double *)malloc(sizeof(double) * 4 * (size_t)N);
  bXs = (double *)malloc(sizeof(double) * 4 * (size_t)M * d);
  bth = (double *)malloc(sizeof(double) * 4 * (size_t)nth);
  bmean = (double *)malloc(sizeof(double) * 4 * (size_t)M);
  bvar = (double *)malloc(sizeof(double) * 4 * (size_t)M);
  for (b = 0; b < 4; ++b) {
    memcpy(bX + (size_t)b * N * d, X, sizeof(double) * (size_t)N * d);
    memcpy(by + (size_t)b * N, y, sizeof(double) * (size_t)N);
    memcpy(bXs + (size_t)b * M * d, Xs, sizeof(double) * (size_t)M * d);
    memcpy(bth + (size_t)b * nth, theta, sizeof(double) * (size_t)nth);
  }
  rc = cgp_fit_predict_batch(ctx, 4, N, d, M, kid, bX, by, bXs, bth, nth, 0, bmean, bvar, blogml, info);
  CHECK(rc == 0, "cgp_fit_predict_batch");
  for (b = 0; b < 4; ++b) {
    CHECK(info[b] == 0 && fabs(blogml[b] - elogml) <= tol * fabs(elogml), "batch logml");
    CHECK(relmax(bmean + (size_t)b * M, emean, M) < tol && releach(bvar + (size_t)b * M, evar, M) < tol, "batch mean / variance");
  }
  sw = cgp_sweep_create(devices, 2, N, M, d, 4, dtype);
  CHECK(sw != NULL && cgp_sweep_ndev(sw) == 2, "cgp_sweep_create");
  CHECK(cgp_sweep_shard(sw, 4, 1, &s0, &s1) == 0 && s0 == 2 && s1 == 4, "cgp_sweep_shard");
  memset(bmean, 0, sizeof(double) * 4 * (size_t)M);
  rc = cgp_sweep_fit_predict(sw, 4, N, d, M, kid, bX, by, bXs, bth, nth, 0, bmean, bvar, blogml, info, summary);
  CHECK(rc == 0, "cgp_sweep_fit_predict");
  for (b = 0; b < 4; ++b) {
    double vmax = 0.0;
    for (i = 0; i < M; ++i)
      if (bvar[(size_t)b * M + i] > vmax) vmax = bvar[(size_t)b * M + i];
    CHECK(relmax(bmean + (size_t)b * M, emean, M) < tol && releach(bvar + (size_t)b * M, evar, M) < tol, "sweep mean / variance");
    CHECK(summary[3 * b] == blogml[b] && summary[3 * b + 1] == 2.0 * sqrt(vmax) && summary[3 * b + 2] == 0.0, "sweep summary row");
  }
  printf("batch of 4 and sweep over 2 shards: within %.0e of the fixture\n", tol);
  cgp_sweep_destroy(sw);
  cgp_destroy(ctx);
  free(buf), free(mean), free(var), free(alpha), free(bX), free(by), free(bXs), free(bth), free(bmean), free(bvar);
  printf("caller.c ok\n");
  return 0;
}

/* One host tick of one N = 512 window from a C caller (no Python, no ctypes in the timed loop): cgp_window_push(T = 1) in a loop.
 *   gcc -O2 -std=c11 -Iinclude tools/window_tick_c.c -o tools/window_tick_c -Lcorenav_gp_amd -lcorenav_gp -lm -Wl,-rpath,'$ORIGIN/../corenav_gp_amd'
 *   tools/window_tick_c */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "corenav_gp.h"

static double now_us(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

int main(void) {
  enum { N = 512, D = 3, T = N + 400 };
  static double X[T][D], y[T];
  double theta[D + 2] = {0.02, 0.8, 1.2, 1.6, 1e-3}, pm, pv, lm, best = 1e30;
  int i, q, rep, st = 0;
  unsigned s = 12345u;
  cgp_ctx *c = cgp_create_ex(0, 8, 8, D, 1, CGP_F64, &st);
  if (!c) { fprintf(stderr, "cgp_create_ex: %d\n", st); return 2; }
  for (i = 0; i < T; ++i) {
    for (q = 0; q < D; ++q) { s = s * 1664525u + 1013904223u; X[i][q] = q == 0 ? (i - T / 2) / (double)T * 3.4 : ((s >> 8) / 16777216.0 - 0.5) * 3.0; }
    s = s * 1664525u + 1013904223u;
    y[i] = 0.1 * sin(2 * 3.141592653589793 * i / 40.0) + ((s >> 8) / 16777216.0 - 0.5) * 0.1;
  }
  if (cgp_window_init(c, 1, N, D, CGP_KERNEL_SE_ARD, theta, D + 2) != 0) return 3;
  {
    static double pmv[N], pvv[N], lmv[N];
    if (cgp_window_push(c, N, &X[0][0], y, 1, pmv, pvv, lmv) != 0) return 4;
  }
  for (i = N; i < N + 40; ++i) if (cgp_window_push(c, 1, X[i], &y[i], 1, &pm, &pv, &lm) != 0) return 5;
  for (rep = 0; rep < 3; ++rep) {
    const double t0 = now_us();
    for (i = N + 40 + 100 * rep; i < N + 140 + 100 * rep; ++i) if (cgp_window_push(c, 1, X[i], &y[i], 1, &pm, &pv, &lm) != 0) return 6;
    const double dt = (now_us() - t0) / 100.0;
    if (dt < best) best = dt;
  }
  printf("C caller: %.1f us per cgp_window_push(T = 1) of one N = %d window (last one-step-ahead mean %.6f, logML %.3f)\n", best, N, pm, lm);
  cgp_destroy(c);
  return 0;
}

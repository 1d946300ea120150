// How v_mfma_f32_16x16x32_bf16 (and v_mfma_f32_16x16x4_f32 beside it) rounds a sum of products of unlike magnitude: the 6-term bf16
// form of an fp32 product puts x0 y0 (order 1) and x1 y1 (order 2^-16) into the same instruction.  Each case gives the 32 products and
// the accumulator exactly (host, double) and prints the instruction's result against the exact sum rounded to nearest and truncated.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_round_probe.hip -o /tmp/mfma_round && /tmp/mfma_round
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_bf16(const float *a, const float *b, float c0, float *out) {   // a[k], b[k], k = 0 .. 31: every row / column the same
  const int g = threadIdx.x >> 4;
  bf8 x, y;
  for (int j = 0; j < 8; ++j) { x[j] = (__bf16)a[8 * g + j]; y[j] = (__bf16)b[8 * g + j]; }
  f4 c = {c0, c0, c0, c0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
__global__ void k_f32(const float *a, const float *b, float c0, float *out) {    // the same 32 products as eight v_mfma_f32_16x16x4_f32 in k order
  const int g = threadIdx.x >> 4;
  f4 c = {c0, c0, c0, c0};
  for (int s = 0; s < 8; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * s + g], b[4 * s + g], c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
static float trunc_f32(double v) {   // toward zero at fp32 precision
  float f = (float)v;
  if (std::fabs((double)f) > std::fabs(v)) f = std::nextafterf(f, 0.0f);
  return f;
}
int main() {
  float *da, *db, *dout;
  hipMalloc(&da, 128); hipMalloc(&db, 128); hipMalloc(&dout, 4);
  struct Case { const char *name; std::vector<float> a, b; float c0; };
  std::vector<Case> cases;
  auto mk = [&](const char *name, float c0) { cases.push_back({name, std::vector<float>(32, 0.f), std::vector<float>(32, 0.f), c0}); return &cases.back(); };
  const float u = std::ldexp(1.f, -23);    // ulp of 1.0
  { auto c = mk("1*1 + one product of 0.75 ulp (nearest: +1 ulp, truncation: +0)", 0.f); c->a[0] = 1; c->b[0] = 1; c->a[1] = 0.75f; c->b[1] = u; }
  { auto c = mk("1*1 - one product of 0.75 ulp(below 1: ulp/2) ", 0.f); c->a[0] = 1; c->b[0] = 1; c->a[1] = -0.75f; c->b[1] = u / 2; }
  { auto c = mk("1*1 + 8 products of 0.375 ulp in the same k group (exact sum 1 + 3 ulp)", 0.f); c->a[0] = 1; c->b[0] = 1; for (int j = 1; j <= 8; ++j) { c->a[j] = 0.375f; c->b[j] = u; } }
  { auto c = mk("1*1 + 8 products of 0.375 ulp in the other k half (k = 16 ..)", 0.f); c->a[0] = 1; c->b[0] = 1; for (int j = 16; j < 24; ++j) { c->a[j] = 0.375f; c->b[j] = u; } }
  { auto c = mk("1*1 + 24 products of 3/32 ulp (exact sum 1 + 2.25 ulp)", 0.f); c->a[0] = 1; c->b[0] = 1; for (int j = 1; j <= 24; ++j) { c->a[j] = 0.09375f; c->b[j] = u; } }
  { auto c = mk("1*1 + 16 products of 3/256 ulp + 0.75 ulp (exact 1 + 0.9375 ulp)", 0.f); c->a[0] = 1; c->b[0] = 1; c->a[1] = 0.75f; c->b[1] = u; for (int j = 2; j < 18; ++j) { c->a[j] = 0.01171875f; c->b[j] = u; } }
  { auto c = mk("accumulator 1 + 8 products of 0.375 ulp (exact 1 + 3 ulp)", 1.f); for (int j = 0; j < 8; ++j) { c->a[j] = 0.375f; c->b[j] = u; } }
  { auto c = mk("accumulator 1 + one product of 0.75 ulp", 1.f); c->a[0] = 0.75f; c->b[0] = u; }
  { auto c = mk("accumulator 1 - one product of 0.375 ulp (below 1: 0.75 of ulp/2)", 1.f); c->a[0] = -0.375f; c->b[0] = u; }
  { auto c = mk("16 products of 1 + 16 of 0.46875 ulp (exact 16 + 7.5 ulp(1) = 16 + 0.47 ulp(16))", 0.f); for (int j = 0; j < 16; ++j) { c->a[j] = 1; c->b[j] = 1; c->a[16 + j] = 0.46875f; c->b[16 + j] = u; } }
  { auto c = mk("16 products of 1 + 16 of 0.75 ulp (exact 16 + 12 ulp(1) = 16 + 0.75 ulp(16))", 0.f); for (int j = 0; j < 16; ++j) { c->a[j] = 1; c->b[j] = 1; c->a[16 + j] = 0.75f; c->b[16 + j] = u; } }
  for (auto &c : cases) {
    double exact = c.c0;
    for (int j = 0; j < 32; ++j) exact += (double)c.a[j] * (double)c.b[j];
    float r[2];
    hipMemcpy(da, c.a.data(), 128, hipMemcpyHostToDevice); hipMemcpy(db, c.b.data(), 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_bf16, dim3(1), dim3(64), 0, 0, da, db, c.c0, dout); hipMemcpy(&r[0], dout, 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, da, db, c.c0, dout); hipMemcpy(&r[1], dout, 4, hipMemcpyDeviceToHost);
    const float rn = (float)exact, rz = trunc_f32(exact);
    const double ul = std::ldexp(1.0, std::ilogb(exact) - 23);
    printf("%-90s exact - base %+8.4f ulp | bf16 x32 %+8.4f ulp (%s) | f32 x4 chain %+8.4f ulp | nearest %+.0f, truncated %+.0f\n", c.name,
           (exact - std::floor(exact)) / ul, ((double)r[0] - std::floor(exact)) / ul,
           r[0] == rn && r[0] != rz ? "= nearest" : r[0] == rz && r[0] != rn ? "= truncated" : r[0] == rn ? "= both" : "neither",
           ((double)r[1] - std::floor(exact)) / ul, ((double)rn - std::floor(exact)) / ul, ((double)rz - std::floor(exact)) / ul);
  }
  // scans: one product of 1 at k = 0 and one small product at k = j; results in ulp(1) above 1
  auto run = [&](const std::vector<float> &a, const std::vector<float> &b, float c0) {
    float r;
    hipMemcpy(da, a.data(), 128, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_bf16, dim3(1), dim3(64), 0, 0, da, db, c0, dout); hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost);
    return r;
  };
  printf("\nscan 1: 1*1 at k = 0, +0.75 ulp at k = j (j = 1 .. 31): result - 1 in ulp\n ");
  for (int j = 1; j < 32; ++j) { std::vector<float> a(32, 0.f), b(32, 0.f); a[0] = b[0] = 1; a[j] = 0.75f; b[j] = u; printf(" %g", (run(a, b, 0.f) - 1.0) / u); }
  printf("\nscan 2: 1*1 at k = 0, +1.75 ulp at k = j: result - 1 in ulp\n ");
  for (int j = 1; j < 32; ++j) { std::vector<float> a(32, 0.f), b(32, 0.f); a[0] = b[0] = 1; a[j] = 1.75f; b[j] = u; printf(" %g", (run(a, b, 0.f) - 1.0) / u); }
  printf("\nscan 3: 1*1 at k = 0, a product of (1 + 2^-g) ulp at k = 16, g = 1 .. 10 (guard bits of the combination): result - 1 in ulp\n ");
  for (int g = 1; g <= 7; ++g) { std::vector<float> a(32, 0.f), b(32, 0.f); a[0] = b[0] = 1; a[16] = 1.f + std::ldexp(1.f, -g); b[16] = u; printf(" g=%d: %g", g, (run(a, b, 0.f) - 1.0) / u); }
  printf("\nscan 4: 1*1 at k = 0, n products of 0.25 ulp at k = 16 .. 16 + n - 1 (n = 1 .. 8): result - 1 in ulp\n ");
  for (int n = 1; n <= 8; ++n) { std::vector<float> a(32, 0.f), b(32, 0.f); a[0] = b[0] = 1; for (int j = 0; j < n; ++j) { a[16 + j] = 0.25f; b[16 + j] = u; } printf(" n=%d: %g", n, (run(a, b, 0.f) - 1.0) / u); }
  printf("\nscan 5: 1*1 at k = 0, n products of 0.25 ulp at k = 8 j (one per lane group), then at k = 1 .. (same group): result - 1 in ulp\n ");
  { std::vector<float> a(32, 0.f), b(32, 0.f); a[0] = b[0] = 1; for (int j = 1; j < 4; ++j) { a[8 * j] = 0.25f; b[8 * j] = u; } printf(" one per other lane group (exact 0.75): %g;", (run(a, b, 0.f) - 1.0) / u); }
  { std::vector<float> a(32, 0.f), b(32, 0.f); a[0] = b[0] = 1; for (int j = 1; j < 4; ++j) { a[j] = 0.25f; b[j] = u; } printf(" three in the group of the 1 (exact 0.75): %g", (run(a, b, 0.f) - 1.0) / u); }
  printf("\nscan 6: -1*1 at k = 0, +0.2 ulp(1) at k = j (exact -1 + 0.2 ulp = -(1 - 0.4 ulp below 1)): nearest -1, toward zero -(1 - 2^-24); result + 1 in units of 2^-24\n ");
  for (int j : {1, 4, 8, 16, 24}) { std::vector<float> a(32, 0.f), b(32, 0.f); a[0] = -1; b[0] = 1; a[j] = 0.2001953125f; b[j] = u; printf(" k=%d: %g", j, (run(a, b, 0.f) + 1.0) / (u / 2)); }
  printf("\nscan 7: accumulator c = 1 with 1*1 at k = 0 and +0.75 ulp(2) = 1.5 ulp(1) at k = j: exact 2 + 0.75 ulp(2); result - 2 in ulp(2)\n ");
  for (int j : {1, 8, 16}) { std::vector<float> a(32, 0.f), b(32, 0.f); a[0] = b[0] = 1; a[j] = 1.5f; b[j] = u; printf(" k=%d: %g", j, (run(a, b, 1.f) - 2.0) / (2 * u)); }
  printf("\nscan 8: accumulator c = 2^e, products: 1*1 at k = 0 and 0.75 ulp(1) at k = 16; result - c - 1 in ulp(1): e = 0 .. 4\n ");
  for (int e = 0; e <= 4; ++e) { std::vector<float> a(32, 0.f), b(32, 0.f); a[0] = b[0] = 1; a[16] = 0.75f; b[16] = u; const float c0 = std::ldexp(1.f, e); printf(" e=%d: %g", e, ((double)run(a, b, c0) - c0 - 1.0) / u); }
  printf("\nscan 9: accumulator c = 1, products 2^-8 (1 + 2^-7) * (1 + 2^-7) 2^-8 at all 32 k (a 16-bit product at scale 2^-16, x 32): exact, result - 1 in ulp(1)\n ");
  { std::vector<float> a(32, (1.f + 0.0078125f) / 256), b(32, (1.f + 0.0078125f) / 256); double ex = 0; for (int j = 0; j < 32; ++j) ex += (double)a[j] * b[j]; printf(" exact %.6f, result %.6f", ex / u, ((double)run(a, b, 1.f) - 1.0) / u); }
  printf("\nscan 10: 1*1 at k = 0, a NEGATIVE product -x ulp(1) at k = j: magnitude truncation of the small term gives 1, floor gives 1 - 2^-24; result - 1 in units of 2^-24\n ");
  for (float x : {0.05f, 0.2001953125f, 0.450195312f}) for (int j : {1, 8, 16, 24}) { std::vector<float> a(32, 0.f), b(32, 0.f); a[0] = b[0] = 1; a[j] = -x; b[j] = u; printf(" x=%.2f k=%d: %g;", x, j, (run(a, b, 0.f) - 1.0) / (u / 2)); }
  // statistics: 32 positive products whose magnitudes fall by a factor r per k inside each lane group (a factor column's decay), random 8-bit
  // mantissas; signed error of the instruction against the exact sum, in ulp of the result, over 400 draws
  printf("\nstatistics (400 draws each): 32 positive bf16 products, magnitudes falling by r per k within every group of 8; signed error in ulp of the result: mean, sd\n");
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
  auto bf = [&](double v) { float f = (float)v; unsigned w; memcpy(&w, &f, 4); w &= 0xffff0000u; memcpy(&f, &w, 4); return f; };
  for (double r : {1.0, 0.5, 0.25, 0.0625}) for (int mixed = 0; mixed < 2; ++mixed) for (float c0 : {0.f, 64.f}) {
    double m = 0, m2 = 0;
    const int T = 400;
    for (int t = 0; t < T; ++t) {
      std::vector<float> a(32), b(32);
      double ex = c0;
      for (int j = 0; j < 32; ++j) {
        const double sc = std::pow(r, j & 7) * ((mixed && j >= 16) ? std::ldexp(1.0, -8) : 1.0);   // mixed: the upper k half is an x1 y1 pair (2^-16 of the lower)
        a[j] = bf((1.0 + rnd()) * sc); b[j] = bf((1.0 + rnd()) * ((mixed && j >= 16) ? std::ldexp(1.0, -8) : 1.0));
        ex += (double)a[j] * b[j];
      }
      const double got = run(a, b, c0), ul = std::ldexp(1.0, std::ilogb(ex) - 23), e = (got - ex) / ul;
      m += e; m2 += e * e;
    }
    m /= T; m2 = std::sqrt(m2 / T - m * m);
    printf("  r = %-6g %s accumulator %-3g: mean %+.3f sd %.3f ulp\n", r, mixed ? "k >= 16 at 2^-16 (x0 y0 | x1 y1)," : "one scale,                       ", c0, m, m2);
  }
  printf("\n");
  return 0;
}

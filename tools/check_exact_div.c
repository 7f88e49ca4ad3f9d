/* check_exact_div.c -- CPU check of the exact-reciprocal quotient used by the box test
 * (mdiv in ray-tracer_amd/csrc/rt_kernels.inl): for d, a in the documented safe ranges,
 *   r = 1/d;  q0 = a*r;  e = fma(-d,q0,a);  q1 = fma(e,r,q0);  e2 = fma(-d,q1,a);  q = fma(e2,r,q1)
 * must equal a/d bit for bit (sign of zero aside).
 *   gcc -O2 -mfma -fopenmp -ffp-contract=off tools/check_exact_div.c -o /tmp/check_exact_div -lm
 *   /tmp/check_exact_div [millions of cases per thread]
 * Inputs: uniformly random bit patterns in range, plus structured hard cases (a = RN(d*m)
 * +- few ulps, i.e. quotients next to representable numbers and to rounding midpoints). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float mdiv(float a, float d, float r) {
  float q0 = a * r;
  float e = __builtin_fmaf(-d, q0, a);
  float q1 = __builtin_fmaf(e, r, q0);
  float e2 = __builtin_fmaf(-d, q1, a);
  return __builtin_fmaf(e2, r, q1);
}
static inline uint64_t rng(uint64_t *s) {
  uint64_t x = *s;
  x ^= x << 13; x ^= x >> 7; x ^= x << 17;
  return *s = x;
}
static inline float bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
/* random float with exponent in [elo, ehi] (unbiased), random sign and mantissa */
static inline float rnd_float(uint64_t *s, int elo, int ehi) {
  uint64_t x = rng(s);
  uint32_t man = (uint32_t)x & 0x7fffff, sign = (uint32_t)(x >> 23) & 1;
  int e = elo + (int)((x >> 24) % (uint64_t)(ehi - elo + 1));
  return bits((sign << 31) | ((uint32_t)(e + 127) << 23) | man);
}

int main(int argc, char **argv) {
  long per = (argc > 1 ? atol(argv[1]) : 50) * 1000000L;
  long bad = 0, total = 0;
#pragma omp parallel reduction(+ : bad, total)
  {
    uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(1 +
#ifdef _OPENMP
                 omp_get_thread_num()
#else
                 0
#endif
                 );
    for (long it = 0; it < per; ++it) {
      float d = rnd_float(&s, -40, 39);
      float r = 1.0f / d;
      float a;
      int mode = (int)(rng(&s) & 3);
      if (mode == 0) {
        a = rnd_float(&s, -59, 59);
      } else {
        /* hard cases: a near d*m for a random m, nudged by up to 3 ulps, and a near
         * d*(m + half ulp) (midpoints of the quotient) */
        float m = rnd_float(&s, -19, 19);
        double target = (double)d * (double)m;
        if (mode == 2) target = (double)d * ((double)m + ldexp(1.0, ilogbf(m) - 24));
        a = (float)target;
        int nud = (int)(rng(&s) % 7) - 3;
        a = bits(fbits(a) + (uint32_t)nud);
        float aa = fabsf(a);
        if (!(aa >= ldexpf(1.0f, -59) && aa <= ldexpf(1.0f, 59))) continue;
      }
      float q = mdiv(a, d, r), t = a / d;
      ++total;
      if (fbits(q) != fbits(t) && !(q == 0.0f && t == 0.0f)) {
        if (bad < 10) printf("MISMATCH a=%a d=%a mdiv=%a div=%a\n", a, d, q, t);
        ++bad;
      }
    }
  }
  printf("checked %ld quotients, %ld mismatches\n", total, bad);
  return bad != 0;
}

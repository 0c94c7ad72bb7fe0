/* Test driver for hipstr_amd/csrc/cr_math.h on the host (tests/test_cr_math.py builds it with gcc -O2 -mfma -ffp-contract=off -shared).
 *   cr_exp_batch / cr_log_batch    the functions over an array (compared with Python's decimal, correctly rounded at 60 digits, by the test)
 *   cr_sweep_exp / cr_sweep_log    n arguments from a seeded generator against the host libm; every disagreement is decided by long double
 *                                  (64-bit significand: expl / logl rounded to double) — counts[0] = arguments, [1] = disagreements with libm,
 *                                  [2] = of those, long double sides with cr_math, [3] = with libm, [4] = undecided (the long double value
 *                                  is within 2^-9 ulp of a double midpoint: those arguments are returned for the decimal check)
 */
#include "../../hipstr_amd/csrc/cr_math.h"
#include <math.h>
#include <stdint.h>

void cr_exp_batch(const double* x, double* y, int64_t n){ for (int64_t i = 0; i < n; i++) y[i] = cr_exp(x[i]); }
void cr_log_batch(const double* x, double* y, int64_t n){ for (int64_t i = 0; i < n; i++) y[i] = cr_log(x[i]); }
void libm_exp_batch(const double* x, double* y, int64_t n){ for (int64_t i = 0; i < n; i++) y[i] = exp(x[i]); }
void libm_log_batch(const double* x, double* y, int64_t n){ for (int64_t i = 0; i < n; i++) y[i] = log(x[i]); }

static uint64_t rng_next(uint64_t* s){ uint64_t x = *s; x ^= x << 13; x ^= x >> 7; x ^= x << 17; *s = x; return x; }
static double rng_unit(uint64_t* s){ return (double)(rng_next(s) >> 11) * 0x1p-53; }

/* which double does the long double value round to, and is that decision safe?  returns 1 if |frac - 1/2| of the discarded 11 bits is
 * at least 2^-9 ulp away from a tie (the long double functions are good to ~1 ulp of THEIR precision, 2^-11 ulp of a double) */
static int decide(long double v, double* out){
  const double d = (double)v;                        /* round to nearest even */
  *out = d;
  const long double err = v - (long double)d;        /* exact */
  const double ulp = fabs(nextafter(d, INFINITY) - d);
  const long double frac = fabsl(err) / (long double)ulp;      /* in [0, 1/2] */
  return (0.5L - frac) > 0x1p-9L;
}

static void sweep(int which, uint64_t seed, int64_t n, double lo, double hi, int mode, int64_t counts[5], double* undecided, int cap){
  uint64_t s = seed ? seed : 88172645463325252ull;
  for (int k = 0; k < 5; k++) counts[k] = 0;
  for (int64_t i = 0; i < n; i++){
    double x;
    if (mode == 0) x = lo + (hi - lo) * rng_unit(&s);                      /* uniform in [lo, hi) */
    else x = exp(log(lo) + (log(hi) - log(lo)) * rng_unit(&s));            /* log-uniform in [lo, hi), lo > 0 */
    const double a = which ? cr_log(x) : cr_exp(x), b = which ? log(x) : exp(x);
    counts[0]++;
    if (a == b || (a != a && b != b)) continue;
    counts[1]++;
    double d; const int safe = decide(which ? logl((long double)x) : expl((long double)x), &d);
    if (!safe){ if (counts[4] < cap) undecided[counts[4]] = x; counts[4]++; }
    else if (d == a) counts[2]++;
    else if (d == b) counts[3]++;
    else { if (counts[4] < cap) undecided[counts[4]] = x; counts[4]++; }
  }
}
void cr_sweep_exp(uint64_t seed, int64_t n, double lo, double hi, int64_t counts[5], double* undecided, int cap){ sweep(0, seed, n, lo, hi, 0, counts, undecided, cap); }
void cr_sweep_log(uint64_t seed, int64_t n, double lo, double hi, int log_uniform, int64_t counts[5], double* undecided, int cap){ sweep(1, seed, n, lo, hi, log_uniform, counts, undecided, cap); }

/* the quick phase of cr_exp against the accurate phase: counts[0] = arguments, [1] = accepted by the quick phase, [2] = accepted AND different
 * from the accurate phase's result (must be 0) */
void cr_quick_exp_check(uint64_t seed, int64_t n, double lo, double hi, int64_t counts[3]){
  uint64_t s = seed ? seed : 88172645463325252ull;
  counts[0] = counts[1] = counts[2] = 0;
  for (int64_t i = 0; i < n; i++){
    const double x = lo + (hi - lo) * rng_unit(&s);
    if (fabs(x) < 0x1p-28) continue;
    int m, ok; const double q = cr_exp_quick(x, &m, &ok);
    counts[0]++;
    if (!ok || m <= -1022) continue;
    counts[1]++;
    int m2; const cr_dd y = cr_exp_core(x, &m2);
    if (m2 != m || y.h != q) counts[2]++;
  }
}

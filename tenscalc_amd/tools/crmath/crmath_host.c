/* Host build of tenscalc_amd/csrc/tc_crmath.hip.inc (the file the device evaluators include) next to the host
 * libm's pow: tests/test_crmath.py and scripts/powcheck compare the two and check ours against exact arithmetic. */
#include <math.h>
#include "tc_crmath.hip.inc"
void crm_eval(const double *x, double *rs, double *rs3, long n) {
  for (long i = 0; i < n; i++) { rs[i] = tc_cr_rsqrt(x[i]); rs3[i] = tc_cr_rsqrt3(x[i]); }
}
void libm_eval(const double *x, double *rs, double *rs3, long n) {
  for (long i = 0; i < n; i++) { rs[i] = pow(x[i], -.5); rs3[i] = pow(x[i], -1.5); }
}
#include "tc_glibcpow.hip.inc"
/* fraction of arguments that take the shortcut (the hardware's sqrt) in tc_glibc_pow_half */
long half_shortcuts(const double *x, long n) {
  long k = 0;
  for (long i = 0; i < n; i++) {
    const double v = x[i];
    if (v >= 0x1p-1000 && v < 0x1p1000) {
      const double s = sqrt(v), e = fma(-s, s, v);
      const double u = tcg_f64((tcg_u64(s) & 0x7ff0000000000000ULL) - (52ULL << 52));
      k += fabs(e) <= 0.875 * s * u;
    }
  }
  return k;
}
void half_eval(const double *x, double *emul, double *libm, long n) {
  for (long i = 0; i < n; i++) { emul[i] = tc_glibc_pow_half(x[i]); libm[i] = pow(x[i], .5); }
}
void glibc_emul_eval(const double *x, double *rs, double *rs3, long n) {
  for (long i = 0; i < n; i++) { rs[i] = tc_glibc_pow_negfrac(x[i], -.5); rs3[i] = tc_glibc_pow_negfrac(x[i], -1.5); }
}

/* exp / log (and, when restated, further unary functions) of csrc/tc_glibcpow.hip.inc next to this host's libm:
 * which = 0 exp, 1 log, 2 sin, 3 cos, 4 atan */
void unary_emul_eval(int which, const double *x, double *y, long n) {
  for (long i = 0; i < n; i++)
    y[i] = which == 0 ? tc_glibc_exp(x[i]) : which == 1 ? tc_glibc_log(x[i]) : which == 2 ? tc_glibc_sin(x[i]) : which == 3 ? tc_glibc_cos(x[i]) : tc_glibc_atan(x[i]);
}
void unary_libm_eval(int which, const double *x, double *y, long n) {
  for (long i = 0; i < n; i++) y[i] = which == 0 ? exp(x[i]) : which == 1 ? log(x[i]) : which == 2 ? sin(x[i]) : which == 3 ? cos(x[i]) : atan(x[i]);
}

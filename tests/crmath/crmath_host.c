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
void half_eval(const double *x, double *emul, double *libm, long n) {
  for (long i = 0; i < n; i++) { emul[i] = tc_glibc_pow_frac(x[i], .5); libm[i] = pow(x[i], .5); }
}
void glibc_emul_eval(const double *x, double *rs, double *rs3, long n) {
  for (long i = 0; i < n; i++) { rs[i] = tc_glibc_pow_negfrac(x[i], -.5); rs3[i] = tc_glibc_pow_negfrac(x[i], -1.5); }
}

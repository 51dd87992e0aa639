#include <math.h>
void hostpow(const double *x, double *a, double *b, double *c, int n) {
  for (int i = 0; i < n; i++) { a[i] = .5 * pow(x[i], -.5); b[i] = -.25 * pow(x[i], -1.5); c[i] = sqrt(x[i]); }
}

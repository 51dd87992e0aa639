// Device build of csrc/tc_pow.hip.inc (model 1: the host libm's pow restated) and csrc/tc_crmath.hip.inc behind a
// plain C entry point: tests/test_crmath.py (-m gpu) compares them with the host builds of the same files and with the
// GPU box's own pow() on the same arguments.  Test infrastructure.
#include <hip/hip_runtime.h>
#include <math.h>
#include "tc_pow.hip.inc"
#include "tc_crmath.hip.inc"
__global__ void crm_kernel(const double *x, double *a, double *b, double *c, double *d, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    a[i] = TC_POW_NEG(x[i], -.5); b[i] = TC_POW_NEG(x[i], -1.5);
    c[i] = tc_cr_rsqrt(x[i]); d[i] = TC_POW_HALF(x[i]);      // d: the Newton loops' pow(mu,.5)
  }
}
extern "C" int crm_dev_eval(const double *x, double *a, double *b, double *c, double *d, long n) {
  double *dx = nullptr, *o[4] = {nullptr, nullptr, nullptr, nullptr};
  double *h[4] = {a, b, c, d};
  int rc = (int)hipMalloc(&dx, 8 * n);
  for (int k = 0; k < 4 && !rc; k++) rc = (int)hipMalloc(&o[k], 8 * n);
  if (!rc) rc = (int)hipMemcpy(dx, x, 8 * n, hipMemcpyHostToDevice);
  if (!rc) {
    hipLaunchKernelGGL(crm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, o[0], o[1], o[2], o[3], n);
    rc = (int)hipDeviceSynchronize();
  }
  for (int k = 0; k < 4 && !rc; k++) rc = (int)hipMemcpy(h[k], o[k], 8 * n, hipMemcpyDeviceToHost);
  (void)hipFree(dx);
  for (int k = 0; k < 4; k++) (void)hipFree(o[k]);
  return rc;
}

// the libm functions as the device evaluators compute them (TC_EXP .. TC_ATAN of csrc/tc_pow.hip.inc):
// which = 0 exp, 1 log, 2 sin, 3 cos, 4 atan
__global__ void crm_unary_kernel(int which, const double *x, double *y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = which == 0 ? TC_EXP(x[i]) : which == 1 ? TC_LOG(x[i]) : which == 2 ? TC_SIN(x[i]) : which == 3 ? TC_COS(x[i]) : TC_ATAN(x[i]);
}
extern "C" int crm_dev_unary(int which, const double *x, double *y, long n) {
  double *dx = nullptr, *dy = nullptr;
  int rc = (int)hipMalloc(&dx, 8 * n);
  if (!rc) rc = (int)hipMalloc(&dy, 8 * n);
  if (!rc) rc = (int)hipMemcpy(dx, x, 8 * n, hipMemcpyHostToDevice);
  if (!rc) {
    hipLaunchKernelGGL(crm_unary_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, which, dx, dy, n);
    rc = (int)hipDeviceSynchronize();
  }
  if (!rc) rc = (int)hipMemcpy(y, dy, 8 * n, hipMemcpyDeviceToHost);
  (void)hipFree(dx);
  (void)hipFree(dy);
  return rc;
}

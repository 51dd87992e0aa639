// development aid: how far is the device libm pow() from the host's for the exponents the tape uses?
#include <hip/hip_runtime.h>
#include <math.h>
extern "C" __global__ void k(const double *x, double *a, double *b, double *c, double *d, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { a[i] = .5 * pow(x[i], -.5); b[i] = -.25 * pow(x[i], -1.5); c[i] = sqrt(x[i]); d[i] = 0.5 / sqrt(x[i]); }
}
extern "C" int run(const double *x, double *a, double *b, double *c, double *d, int n) {
  double *dx, *da, *db, *dc, *dd;
  hipMalloc(&dx, 8 * n); hipMalloc(&da, 8 * n); hipMalloc(&db, 8 * n); hipMalloc(&dc, 8 * n); hipMalloc(&dd, 8 * n);
  hipMemcpy(dx, x, 8 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, da, db, dc, dd, n);
  hipMemcpy(a, da, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(b, db, 8 * n, hipMemcpyDeviceToHost);
  hipMemcpy(c, dc, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(d, dd, 8 * n, hipMemcpyDeviceToHost);
  return (int)hipDeviceSynchronize();
}

/* Test driver for integration/tc_mex.c: implements the mex API stand-in of mex.h and calls mexFunction the way
 * MATLAB would.
 *   mex_driver cpu <some .so without tc_* symbols>           argument validation and handle life cycle, no GPU
 *   mex_driver gpu <class lib.so> <in.bin> <out.bin>         create / set / solve / get / destroy on the sls class:
 *        in.bin = int64 B, A[B][8][100], b[B][100], x0[B][8] (doubles; MATLAB [100,8,B], [100,B], [8,B]);
 *        out.bin = status[B], iter[B] (int32), J[B], x[B][8] (doubles)                                              */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mex.h"

static jmp_buf err_jmp;
static char err_msg[1024];
static void (*at_exit_fn)(void) = NULL;

static size_t elsize(mxClassID c) {
  return c == mxDOUBLE_CLASS ? 8 : c == mxINT32_CLASS ? 4 : c == mxUINT64_CLASS ? 8 : 2;
}
size_t mxGetNumberOfElements(const mxArray *a) {
  size_t n = 1;
  for (int d = 0; d < a->ndim; d++) n *= a->dims[d];
  return n;
}
int mxGetString(const mxArray *a, char *buf, mwSize buflen) {
  if (a->cls != mxCHAR_CLASS) return 1;
  size_t n = mxGetNumberOfElements(a);
  if (n + 1 > buflen) return 1;
  memcpy(buf, a->data, n);
  buf[n] = 0;
  return 0;
}
int mxIsChar(const mxArray *a) { return a->cls == mxCHAR_CLASS; }
int mxIsNumeric(const mxArray *a) { return a->cls != mxCHAR_CLASS; }
int mxIsDouble(const mxArray *a) { return a->cls == mxDOUBLE_CLASS; }
int mxIsInt32(const mxArray *a) { return a->cls == mxINT32_CLASS; }
int mxIsUint64(const mxArray *a) { return a->cls == mxUINT64_CLASS; }
int mxIsComplex(const mxArray *a) { (void)a; return 0; }
int mxIsSparse(const mxArray *a) { (void)a; return 0; }
double mxGetScalar(const mxArray *a) {
  if (a->cls == mxDOUBLE_CLASS) return *(double *)a->data;
  if (a->cls == mxINT32_CLASS) return *(int32_t *)a->data;
  return (double)*(uint64_t *)a->data;
}
void *mxGetData(const mxArray *a) { return a->data; }
double *mxGetPr(const mxArray *a) { return (double *)a->data; }
void *mxCalloc(size_t n, size_t size) { return calloc(n ? n : 1, size ? size : 1); }
void mxFree(void *p) { free(p); }
void mexMakeMemoryPersistent(void *p) { (void)p; }
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity c) {
  (void)c;
  mxArray *a = calloc(1, sizeof *a);
  a->cls = cls;
  a->ndim = (int)ndim;
  for (mwSize d = 0; d < ndim; d++) a->dims[d] = dims[d];
  size_t n = mxGetNumberOfElements(a);
  a->data = calloc(n ? n : 1, elsize(cls));
  return a;
}
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity c) {
  mwSize d[2] = {m, n};
  return mxCreateNumericArray(2, d, cls, c);
}
mxArray *mxCreateDoubleScalar(double v) {
  mxArray *a = mxCreateNumericMatrix(1, 1, mxDOUBLE_CLASS, mxREAL);
  *(double *)a->data = v;
  return a;
}
mxArray *mxCreateString(const char *s) {
  mwSize d[2] = {1, strlen(s)};
  mxArray *a = calloc(1, sizeof *a);
  a->cls = mxCHAR_CLASS;
  a->ndim = 2;
  a->dims[0] = d[0];
  a->dims[1] = d[1];
  a->data = calloc(d[1] + 1, 1);       /* one byte per character is enough for the driver's ASCII strings */
  memcpy(a->data, s, d[1]);
  return a;
}
void mxDestroyArray(mxArray *a) {
  if (a) { free(a->data); free(a); }
}
int mexAtExit(void (*fn)(void)) { at_exit_fn = fn; return 0; }
void mexErrMsgTxt(const char *msg) {
  snprintf(err_msg, sizeof err_msg, "%s", msg);
  longjmp(err_jmp, 1);
}
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...) {
  (void)id;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_msg, sizeof err_msg, fmt, ap);
  va_end(ap);
  longjmp(err_jmp, 1);
}

/* one call of the gateway; returns 0, or 1 with err_msg set when it raised a MATLAB error */
static int call(int nlhs, mxArray **plhs, int nrhs, mxArray **prhs) {
  err_msg[0] = 0;
  if (setjmp(err_jmp)) return 1;
  mexFunction(nlhs, plhs, nrhs, (const mxArray **)prhs);
  return 0;
}
static mxArray *u64(uint64_t v) {
  mxArray *a = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
  *(uint64_t *)a->data = v;
  return a;
}
static mxArray *i32(int32_t v) {
  mxArray *a = mxCreateNumericMatrix(1, 1, mxINT32_CLASS, mxREAL);
  *(int32_t *)a->data = v;
  return a;
}
#define EXPECT_ERR(nl, pl, nr, pr, needle)                                                         \
  do {                                                                                             \
    if (!call(nl, pl, nr, pr) || !strstr(err_msg, needle)) {                                       \
      fprintf(stderr, "line %d: expected an error containing '%s', got '%s'\n", __LINE__, needle, err_msg); \
      return 10;                                                                                   \
    }                                                                                              \
  } while (0)
#define EXPECT_OK(nl, pl, nr, pr)                                                       \
  do {                                                                                  \
    if (call(nl, pl, nr, pr)) {                                                         \
      fprintf(stderr, "line %d: unexpected error '%s'\n", __LINE__, err_msg);           \
      return 11;                                                                        \
    }                                                                                   \
  } while (0)

static int run_cpu(const char *notalib) {
  mxArray *out[4] = {0}, *in[8];
  EXPECT_ERR(0, out, 0, in, "command string expected");
  in[0] = mxCreateDoubleScalar(1.0);
  EXPECT_ERR(0, out, 1, in, "command string expected");
  in[0] = mxCreateString("frobnicate");
  EXPECT_ERR(0, out, 1, in, "handle (uint64 scalar) expected");
  in[1] = u64(0);
  EXPECT_ERR(0, out, 2, in, "not a handle");
  in[1] = u64(7);                       /* a slot nobody created: dead */
  EXPECT_ERR(0, out, 2, in, "already destroyed");
  in[1] = u64(((uint64_t)3 << 32) | 9999);
  EXPECT_ERR(0, out, 2, in, "not a handle");
  in[1] = mxCreateDoubleScalar(3.0);
  EXPECT_ERR(0, out, 2, in, "handle (uint64 scalar) expected");
  in[0] = mxCreateString("create");
  EXPECT_ERR(1, out, 1, in, "(libpath, B) expected");
  in[1] = mxCreateString("/nonexistent/libnope.so");
  in[2] = mxCreateDoubleScalar(0.0);
  EXPECT_ERR(1, out, 3, in, "B must be >= 1");
  in[2] = mxCreateDoubleScalar(4.0);
  EXPECT_ERR(1, out, 3, in, "cannot load");
  in[1] = mxCreateString(notalib);
  for (int k = 0; k < 200; k++)         /* the slot is released on the error path: 200 > the table's 64 entries */
    EXPECT_ERR(1, out, 3, in, "not a TensCalc-HIP class library");
  if (at_exit_fn) at_exit_fn();
  printf("MEX_CPU_OK\n");
  return 0;
}

static int run_gpu(const char *lib, const char *inpath, const char *outpath) {
  FILE *f = fopen(inpath, "rb");
  int64_t B;
  if (!f || fread(&B, sizeof B, 1, f) != 1) return 5;
  mwSize dA[3] = {100, 8, (mwSize)B}, db[2] = {100, (mwSize)B}, dx[2] = {8, (mwSize)B};
  mxArray *A = mxCreateNumericArray(3, dA, mxDOUBLE_CLASS, mxREAL), *b = mxCreateNumericArray(2, db, mxDOUBLE_CLASS, mxREAL),
          *x0 = mxCreateNumericArray(2, dx, mxDOUBLE_CLASS, mxREAL);
  if (fread(A->data, 8, B * 800, f) != (size_t)(B * 800) || fread(b->data, 8, B * 100, f) != (size_t)(B * 100) ||
      fread(x0->data, 8, B * 8, f) != (size_t)(B * 8)) return 5;
  fclose(f);
  mxArray *out[4] = {0}, *in[8];
  in[0] = mxCreateString("create"); in[1] = mxCreateString(lib); in[2] = mxCreateDoubleScalar((double)B);
  EXPECT_OK(1, out, 3, in);
  mxArray *h = out[0];
  /* a second object on the same library, destroyed first: handles are independent */
  EXPECT_OK(1, out, 3, in);
  mxArray *h2 = out[0];
  in[0] = mxCreateString("destroy"); in[1] = h2;
  EXPECT_OK(0, out, 2, in);
  EXPECT_ERR(0, out, 2, in, "already destroyed");
  in[0] = mxCreateString("set"); in[1] = h; in[2] = mxCreateString("A"); in[3] = b;
  EXPECT_ERR(0, out, 4, in, "prod(size)*B elements");
  in[2] = mxCreateString("nosuch"); in[3] = A;
  EXPECT_ERR(0, out, 4, in, "no such parameter");
  in[2] = mxCreateString("A");
  EXPECT_OK(0, out, 4, in);
  in[2] = mxCreateString("b"); in[3] = b;
  EXPECT_OK(0, out, 4, in);
  in[2] = mxCreateString("x"); in[3] = x0;
  EXPECT_OK(0, out, 4, in);
  in[0] = mxCreateString("solve"); in[2] = mxCreateDoubleScalar(1.0); in[3] = mxCreateDoubleScalar(30.0); in[4] = i32(-1);
  EXPECT_ERR(3, out, 5, in, "maxIter must be an int32 scalar");
  in[3] = i32(30);
  EXPECT_OK(3, out, 5, in);
  mxArray *st = out[0], *it = out[1], *tm = out[2];
  if (mxGetNumberOfElements(st) != (size_t)B || mxGetNumberOfElements(it) != (size_t)B || !(mxGetScalar(tm) >= 0)) return 6;
  in[0] = mxCreateString("get");
  EXPECT_ERR(3, out, 2, in, "more outputs requested");
  EXPECT_OK(2, out, 2, in);
  mxArray *J = out[0], *x = out[1];
  if (mxGetNumberOfElements(J) != (size_t)B || x->ndim != 2 || x->dims[0] != 8 || x->dims[1] != (mwSize)B) return 7;
  f = fopen(outpath, "wb");
  fwrite(st->data, 4, B, f); fwrite(it->data, 4, B, f); fwrite(J->data, 8, B, f); fwrite(x->data, 8, 8 * B, f);
  fclose(f);
  /* the diagnostics of the reference classes reach the library too: an ordinary class refuses both with ITS message */
  in[0] = mxCreateString("profilingView"); in[2] = mxCreateString("/dev/null");
  EXPECT_ERR(0, out, 3, in, "without profiling");
  in[0] = mxCreateString("trace"); in[2] = mxCreateDoubleScalar(1.0); in[3] = mxCreateString("/dev/null");
  EXPECT_ERR(0, out, 4, in, "solverVerboseLevel < 3");
  EXPECT_ERR(0, out, 3, in, "(handle, instance, filename) expected");
  /* leave h alive: the mexAtExit hook must release it (MATLAB: clear mex) */
  if (!at_exit_fn) { fprintf(stderr, "no mexAtExit hook registered\n"); return 8; }
  at_exit_fn();
  in[0] = mxCreateString("get");
  EXPECT_ERR(2, out, 2, in, "already destroyed");
  printf("MEX_GPU_OK\n");
  return 0;
}

int main(int argc, char **argv) {
  if (argc == 3 && !strcmp(argv[1], "cpu")) return run_cpu(argv[2]);
  if (argc == 5 && !strcmp(argv[1], "gpu")) return run_gpu(argv[2], argv[3], argv[4]);
  fprintf(stderr, "usage: mex_driver cpu <lib without tc symbols> | gpu <class lib> <in.bin> <out.bin>\n");
  return 2;
}

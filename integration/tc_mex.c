/* Generic MATLAB mex gateway onto a TensCalc-HIP class library (include/tenscalc_hip.h).
 *
 * UNVERIFIED: this image has no MATLAB (no mex.h), so this file has never been compiled.  It is
 * the binding a TensCalc maintainer adds next to the gateways that CmexTools' createGateway
 * generates today (lib/cmex2optimizeCS.m:403-417); the dlopen/dlsym pattern is the reference's
 * own (lib/csparse/instructionsTableHeight4MEX.c:40-54).
 *
 *   h   = tc_mex('create', libpath, B)            % obj=<cls>()
 *         tc_mex('set', h, name, value)           % setP_<name>/setV_<name>: value is [size...,B]
 *   [status,iter,time] = tc_mex('solve', h, mu0, int32(maxIter), int32(saveIter), addEye2Hessian)
 *   [y1,y2,...] = tc_mex('get', h)                % getOutputs: each [size...,B]
 *         tc_mex('destroy', h)
 *
 * Build (on a machine with MATLAB + ROCm):  mex tc_mex.c -ldl
 */
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>
#include "mex.h"
#include "tenscalc_hip.h"

typedef struct {
  void *lib;
  tc_solver_t *h;
  int64_t B;
  int (*create)(int64_t, int, tc_solver_t **);
  int (*destroy)(tc_solver_t *);
  int (*set)(tc_solver_t *, const char *, const double *, int64_t);
  int (*solve)(tc_solver_t *, const double *, const int32_t *, const int32_t *, const double *, int32_t *,
               int32_t *, double *);
  int (*get)(tc_solver_t *, double **, int64_t);
  int (*nout)(void);
  int (*ondim)(int);
  int64_t (*odim)(int, int);
  const char *(*err)(void);
} obj_t;

static void check(obj_t *o, int rc, const char *what) {
  if (rc) mexErrMsgIdAndTxt("tenscalc:hip", "%s failed (%d): %s", what, rc, o->err());
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  char cmd[32];
  if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgTxt("tc_mex: command string expected");
  if (!strcmp(cmd, "create")) {
    char path[4096];
    mxGetString(prhs[1], path, sizeof path);
    obj_t *o = (obj_t *)mxCalloc(1, sizeof *o);
    mexMakeMemoryPersistent(o);
    o->lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!o->lib) mexErrMsgIdAndTxt("tenscalc:hip", "cannot load %s: %s", path, dlerror());
    *(void **)&o->create = dlsym(o->lib, "tc_create");
    *(void **)&o->destroy = dlsym(o->lib, "tc_destroy");
    *(void **)&o->set = dlsym(o->lib, "tc_set");
    *(void **)&o->solve = dlsym(o->lib, "tc_solve");
    *(void **)&o->get = dlsym(o->lib, "tc_get_outputs");
    *(void **)&o->nout = dlsym(o->lib, "tc_num_outputs");
    *(void **)&o->ondim = dlsym(o->lib, "tc_output_ndim");
    *(void **)&o->odim = dlsym(o->lib, "tc_output_dim");
    *(void **)&o->err = dlsym(o->lib, "tc_last_error");
    o->B = (int64_t)mxGetScalar(prhs[2]);
    check(o, o->create(o->B, 0, &o->h), "tc_create");
    plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
    *(uint64_t *)mxGetData(plhs[0]) = (uint64_t)(uintptr_t)o;
    return;
  }
  obj_t *o = (obj_t *)(uintptr_t)(*(uint64_t *)mxGetData(prhs[1]));
  if (!strcmp(cmd, "set")) {
    char name[256];
    mxGetString(prhs[2], name, sizeof name);
    /* MATLAB [size...,B] column-major == the C-ABI's instance-major host layout */
    check(o, o->set(o->h, name, mxGetPr(prhs[3]), o->B), "tc_set");
  } else if (!strcmp(cmd, "solve")) {
    double mu0 = mxGetScalar(prhs[2]), tm = 0;
    int32_t maxIter = *(int32_t *)mxGetData(prhs[3]), saveIter = *(int32_t *)mxGetData(prhs[4]);
    double ae_default[2] = {1e-9, 1e-9};   /* template default, cmex2optimizeCS.m:291 */
    const double *ae = nrhs > 5 ? mxGetPr(prhs[5]) : ae_default;
    plhs[0] = mxCreateNumericMatrix(o->B, 1, mxINT32_CLASS, mxREAL);
    mxArray *it = mxCreateNumericMatrix(o->B, 1, mxINT32_CLASS, mxREAL);
    check(o, o->solve(o->h, &mu0, &maxIter, &saveIter, ae, (int32_t *)mxGetData(plhs[0]),
                      (int32_t *)mxGetData(it), &tm), "tc_solve");
    if (nlhs > 1) plhs[1] = it;
    if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(tm);
  } else if (!strcmp(cmd, "get")) {
    int n = o->nout();
    double *ptr[64];
    for (int k = 0; k < n && k < 64; k++) {
      mwSize dims[9];
      int nd = o->ondim(k);
      for (int d = 0; d < nd; d++) dims[d] = (mwSize)o->odim(k, d);
      dims[nd] = (mwSize)o->B;
      mxArray *a = mxCreateNumericArray(nd + 1, dims, mxDOUBLE_CLASS, mxREAL);
      ptr[k] = mxGetPr(a);
      if (k < nlhs || k == 0) plhs[k] = a;
    }
    check(o, o->get(o->h, ptr, o->B), "tc_get_outputs");
  } else if (!strcmp(cmd, "destroy")) {
    o->destroy(o->h);
    dlclose(o->lib);
    mxFree(o);
  } else {
    mexErrMsgIdAndTxt("tenscalc:hip", "unknown command %s", cmd);
  }
}

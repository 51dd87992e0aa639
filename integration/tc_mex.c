/* Generic MATLAB mex gateway onto a TensCalc-HIP class library (include/tenscalc_hip.h).
 *
 * It is the binding a TensCalc maintainer adds next to the gateways that CmexTools' createGateway
 * generates today (lib/cmex2optimizeCS.m:403-417); the dlopen/dlsym pattern is the reference's
 * own (lib/csparse/instructionsTableHeight4MEX.c:40-54).
 *
 * Status: this image has no MATLAB, so the file has never been compiled against MathWorks' mex.h.  It IS
 * compiled and driven end to end against a small stand-in for the mex API (tests/mock_mex/: the dozen mx* / mex*
 * calls used below, errors delivered by longjmp as mexErrMsgIdAndTxt does) -- argument validation and handle life
 * cycle on the CPU, a create / set / solve / get / destroy round trip against the golden vectors on the GPU
 * (tests/test_mex_gateway.py).
 *
 *   h   = tc_mex('create', libpath, B)            % obj=<cls>()
 *         tc_mex('set', h, name, value)           % setP_<name>/setV_<name>: value is [size...,B]
 *   [status,iter,time] = tc_mex('solve', h, mu0, int32(maxIter), int32(saveIter), addEye2Hessian)
 *   [y1,y2,...] = tc_mex('get', h)                % getOutputs: each [size...,B]
 *         tc_mex('destroy', h)
 *
 * Handles are (slot, generation) pairs into a table of live objects kept by this mex file, never raw pointers:
 * a stale or made-up handle is an error message, not a use-after-free; whatever is still alive when MATLAB clears
 * the mex file is destroyed by the mexAtExit hook.
 *
 * Build (on a machine with MATLAB + ROCm):  mex -I<repo>/include tc_mex.c -ldl
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
  uint32_t generation;        /* bumped at every create in this slot: handles of earlier occupants do not match */
  int (*create)(int64_t, int, tc_solver_t **);
  int (*destroy)(tc_solver_t *);
  int (*set)(tc_solver_t *, const char *, const double *, int64_t);
  int (*solve)(tc_solver_t *, const double *, const int32_t *, const int32_t *, const double *, int32_t *,
               int32_t *, double *);
  int (*get)(tc_solver_t *, double **, int64_t);
  int (*nout)(void);
  int (*ondim)(int);
  int64_t (*odim)(int, int);
  int (*nin)(void);
  const char *(*iname)(int);
  int64_t (*inumel)(int);
  const char *(*err)(void);
  int (*profview)(tc_solver_t *, const char *);            /* optional: diagnostics of the reference classes */
  int (*trace)(tc_solver_t *, int64_t, const char *);
} obj_t;

#define TCM_MAX 64
static obj_t tcm_objs[TCM_MAX];       /* static storage: nothing to leak, nothing MATLAB's allocator can take away */
static int tcm_atexit_set = 0;

static void tcm_release(obj_t *o) {
  if (o->h) o->destroy(o->h);
  o->h = NULL;
  if (o->lib) dlclose(o->lib);
  o->lib = NULL;
}

static void tcm_at_exit(void) {
  for (int k = 0; k < TCM_MAX; k++) tcm_release(&tcm_objs[k]);
}

static void need(int ok, const char *msg) {
  if (!ok) mexErrMsgIdAndTxt("tenscalc:hip", "tc_mex: %s", msg);
}

static void check(obj_t *o, int rc, const char *what) {
  if (rc) mexErrMsgIdAndTxt("tenscalc:hip", "%s failed (%d): %s", what, rc, o->err());
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
  char cmd[32];
  if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgTxt("tc_mex: command string expected");
  if (!strcmp(cmd, "create")) {
    char path[4096];
    need(nrhs == 3 && mxIsChar(prhs[1]) && mxIsNumeric(prhs[2]) && mxGetNumberOfElements(prhs[2]) == 1,
         "create: (libpath, B) expected");
    need(!mxGetString(prhs[1], path, sizeof path), "create: library path too long");
    need(mxGetScalar(prhs[2]) >= 1, "create: B must be >= 1");
    int slot = -1;
    for (int k = 0; k < TCM_MAX && slot < 0; k++)
      if (!tcm_objs[k].lib) slot = k;
    need(slot >= 0, "create: too many live solver objects (64)");
    if (!tcm_atexit_set) {
      mexAtExit(tcm_at_exit);
      tcm_atexit_set = 1;
    }
    obj_t *o = &tcm_objs[slot];
    const uint32_t gen = o->generation + 1;
    memset(o, 0, sizeof *o);
    o->generation = gen;
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
    *(void **)&o->nin = dlsym(o->lib, "tc_num_inputs");
    *(void **)&o->iname = dlsym(o->lib, "tc_input_name");
    *(void **)&o->inumel = dlsym(o->lib, "tc_input_numel");
    *(void **)&o->err = dlsym(o->lib, "tc_last_error");
    *(void **)&o->profview = dlsym(o->lib, "tc_profiling_view");
    *(void **)&o->trace = dlsym(o->lib, "tc_trace");
    if (!o->create || !o->destroy || !o->set || !o->solve || !o->get || !o->nout || !o->ondim || !o->odim ||
        !o->nin || !o->iname || !o->inumel || !o->err) {
      dlclose(o->lib);      /* the slot is free again before the error unwinds */
      o->lib = NULL;
      mexErrMsgIdAndTxt("tenscalc:hip", "%s is not a TensCalc-HIP class library (missing tc_* symbols)", path);
    }
    o->B = (int64_t)mxGetScalar(prhs[2]);
    int rc = o->create(o->B, 0, &o->h);
    if (rc) {
      char msg[512];
      strncpy(msg, o->err(), sizeof msg - 1);
      msg[sizeof msg - 1] = 0;
      o->h = NULL;
      dlclose(o->lib);
      o->lib = NULL;
      mexErrMsgIdAndTxt("tenscalc:hip", "tc_create failed (%d): %s", rc, msg);
    }
    plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
    *(uint64_t *)mxGetData(plhs[0]) = ((uint64_t)o->generation << 32) | (uint64_t)(slot + 1);
    return;
  }
  need(nrhs >= 2 && mxIsUint64(prhs[1]) && mxGetNumberOfElements(prhs[1]) == 1, "handle (uint64 scalar) expected");
  const uint64_t hv = *(uint64_t *)mxGetData(prhs[1]);
  const uint64_t slot1 = hv & 0xffffffffu;
  need(slot1 >= 1 && slot1 <= TCM_MAX, "not a handle returned by tc_mex('create', ...)");
  obj_t *o = &tcm_objs[slot1 - 1];
  need(o->lib != NULL && o->h != NULL && o->generation == (uint32_t)(hv >> 32), "handle already destroyed");
  if (!strcmp(cmd, "set")) {
    char name[256];
    need(nrhs == 4 && mxIsChar(prhs[2]), "set: (h, name, value) expected");
    need(!mxGetString(prhs[2], name, sizeof name), "set: name too long");
    need(mxIsDouble(prhs[3]) && !mxIsComplex(prhs[3]) && !mxIsSparse(prhs[3]), "set: value must be a full real double array");
    int64_t numel = -1;
    for (int k = 0; k < o->nin(); k++)
      if (!strcmp(o->iname(k), name)) numel = o->inumel(k);
    need(numel >= 0, "set: the class has no such parameter/variable");
    /* exactly numel*B doubles: tc_set reads that many from the pointer */
    need((int64_t)mxGetNumberOfElements(prhs[3]) == numel * o->B, "set: value must have prod(size)*B elements ([size...,B])");
    /* MATLAB [size...,B] column-major == the C-ABI's instance-major host layout */
    check(o, o->set(o->h, name, mxGetPr(prhs[3]), o->B), "tc_set");
  } else if (!strcmp(cmd, "solve")) {
    need(nrhs == 5 || nrhs == 6, "solve: (h, mu0, int32(maxIter), int32(saveIter)[, addEye2Hessian]) expected");
    need(mxIsDouble(prhs[2]) && mxGetNumberOfElements(prhs[2]) == 1, "solve: mu0 must be a double scalar");
    need(mxIsInt32(prhs[3]) && mxGetNumberOfElements(prhs[3]) == 1, "solve: maxIter must be an int32 scalar");
    need(mxIsInt32(prhs[4]) && mxGetNumberOfElements(prhs[4]) == 1, "solve: saveIter must be an int32 scalar");
    need(nrhs < 6 || (mxIsDouble(prhs[5]) && mxGetNumberOfElements(prhs[5]) == 2), "solve: addEye2Hessian must be 2 doubles");
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
    need(nlhs <= n, "get: more outputs requested than the class has");
    double **ptr = (double **)mxCalloc(n > 0 ? n : 1, sizeof *ptr);   /* tc_get_outputs reads exactly n pointers */
    for (int k = 0; k < n; k++) {
      mwSize dims[9];
      int nd = o->ondim(k);
      need(nd >= 0 && nd <= 8, "get: output with more than 8 dimensions");
      for (int d = 0; d < nd; d++) dims[d] = (mwSize)o->odim(k, d);
      dims[nd] = (mwSize)o->B;
      mxArray *a = mxCreateNumericArray(nd + 1, dims, mxDOUBLE_CLASS, mxREAL);
      ptr[k] = mxGetPr(a);
      if (k < nlhs || k == 0) plhs[k] = a;
    }
    int rc = o->get(o->h, ptr, o->B);
    mxFree(ptr);
    check(o, rc, "tc_get_outputs");
  } else if (!strcmp(cmd, "profilingView")) {
    /* profilingView(obj, filename) of classes generated with profiling=true (writeCprofiling.c:68-75) */
    char path[4096];
    need(nrhs == 3 && mxIsChar(prhs[2]) && !mxGetString(prhs[2], path, sizeof path), "profilingView: (handle, filename) expected");
    need(o->profview != NULL, "profilingView: this class library has no tc_profiling_view");
    check(o, o->profview(o->h, path), "tc_profiling_view");
  } else if (!strcmp(cmd, "trace")) {
    /* the solverVerboseLevel >= 3 iteration table (ipmPD_CSsolver.c:248-345,775-863) of ONE instance, 1-based as in MATLAB */
    char path[4096];
    need(nrhs == 4 && mxIsNumeric(prhs[2]) && mxGetNumberOfElements(prhs[2]) == 1 && mxIsChar(prhs[3]) &&
         !mxGetString(prhs[3], path, sizeof path), "trace: (handle, instance, filename) expected");
    need(o->trace != NULL, "trace: this class library has no tc_trace");
    check(o, o->trace(o->h, (int64_t)mxGetScalar(prhs[2]) - 1, path), "tc_trace");
  } else if (!strcmp(cmd, "destroy")) {
    tcm_release(o);        /* the slot's generation stays: this handle value never matches again */
  } else {
    mexErrMsgIdAndTxt("tenscalc:hip", "unknown command %s", cmd);
  }
}

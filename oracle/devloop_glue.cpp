/* ORACLE - test infrastructure only.
 *
 * Host build of the DEVICE Newton loop: tenscalc_amd/csrc/ipm_newton.hip.inc compiled by g++ with
 * __device__ defined away, its tape entry points forwarded to the oracle's tape interpreter
 * (tc_oracle.c) -- the mirror image of ref_glue.c, which does the same for the reference's
 * lib/ipmPD_CSsolver.c.  tests/test_device_loop_on_host.py runs both loops (and the oracle's own
 * restatement) on the same tape and inputs and requires identical status, iteration count and
 * outputs, so the control flow of the device loop is pinned to the reference file without a GPU.
 */
#include <math.h>
#include <stdint.h>
extern "C" {
#include "tc_oracle.h"
}

#include TC_DEFINES_H /* the solver #define block (cmex2optimizeCS.m:303-331), written by oracle/build.py */

#define __device__
#define __forceinline__ inline
struct TcCtx {
  tco_t *g;
#if defined(verboseLevel) && verboseLevel >= 3
  double *v3, *v3cur;   /* rows of the verbose iteration table (csrc/ipm_newton.hip.inc: TC_V3), as in the generated classes */
  int v3n, v3max;
#endif
};

/* trace of entry-point calls (entry indices) of the last tcdev_solve: tests/test_loop_schedule.py replays
 * it against the static clean/dirty claims of tenscalc_amd/loop_schedule.py */
#define TC_TRACE_MAX (1 << 20)
static int g_trace[TC_TRACE_MAX];
static int g_ntrace;
static inline int traced(tco_t *g, const char *name) {
  const int e = tco_entry(g, name);
  if (g_ntrace < TC_TRACE_MAX) g_trace[g_ntrace++] = e;
  return e;
}

#define COPY(fn) static inline void fn(TcCtx &c) { tco_copy(c.g, traced(c.g, #fn)); }
#define SET(fn) static inline void fn(TcCtx &c, const double *x) { tco_set(c.g, traced(c.g, #fn), x); }
#define GET1(fn) static inline void fn(TcCtx &c, double *x) { tco_get_flat(c.g, traced(c.g, #fn), x); }

COPY(initPrimal__)
COPY(initDualEqX__)
COPY(initDualEq__)
COPY(initDualIneq__)
COPY(updatePrimalDual__)
COPY(scaleIneq__)
COPY(scaleCost__)
SET(setAddEye2HessianU__)
SET(setAddEye2HessianEq__)
SET(setAddEye2Hessian1__)   /* equilibrium loop, ipmPDeq_CSsolver.c:32-33 */
SET(setAddEye2Hessian2__)
SET(setAlphaPrimal__)
SET(setAlphaDualEq__)
SET(setAlphaDualIneq__)
SET(setMu__)
GET1(getJ__)
GET1(getDirectionError__)
GET1(getScale4Cost__)
GET1(getNorminf_G__)
GET1(getNorminf_Grad__)
GET1(getCurvature__)
GET1(getMinF_s__)
GET1(getMinF_a__)
GET1(getRho__)
static inline void getMaxAlphas_a__(TcCtx &c, double *alphaPrimal, double *alphaDual) {
  double t[2];
  tco_get_flat(c.g, traced(c.g, "getMaxAlphas_a__"), t);
  *alphaPrimal = t[0];
  *alphaDual = t[1];
}
static inline void getGapMinFMinLambda__(TcCtx &c, double *gap, double *ineq, double *dual) {
  double t[3];
  tco_get_flat(c.g, traced(c.g, "getGapMinFMinLambda__"), t);
  *gap = t[0];
  *ineq = t[1];
  *dual = t[2];
}
static inline void getMaxAlphas_s__(TcCtx &c, double *alphaPrimal, double *alphaDual) {
  double t[2];
  tco_get_flat(c.g, traced(c.g, "getMaxAlphas_s__"), t);
  *alphaPrimal = t[0];
  *alphaDual = t[1];
}
static inline void getHessInertia__(TcCtx &c, double *mp, double *mn) {
  double t[2];
  tco_get_flat(c.g, traced(c.g, "getHessInertia__"), t);
  *mp = t[0];
  *mn = t[1];
}

#define TC_POW_HALF(x) pow((x), .5)   /* the host libm, as the reference loop calls it */
#ifdef TC_LOOP_EQ   /* the equilibrium solvers' loop: csrc/ipm_newton_eq.hip.inc <-> lib/ipmPDeq_CSsolver.c */
#include "ipm_newton_eq.hip.inc"
#define TC_DEVICE_LOOP ipmPDeq_CSsolver
#else
#include "ipm_newton.hip.inc"
#define TC_DEVICE_LOOP ipmPD_CSsolver
#endif

#if defined(verboseLevel) && verboseLevel >= 3
#define TC_V3_HOSTROWS 4096
static double g_v3[24 * TC_V3_HOSTROWS];
static int g_v3n;
/* rows the last tcdev_solve recorded (24 doubles each): tests/test_verbose_table.py */
extern "C" int tcdev_v3_rows(double *out, int maxRows) {
  const int n = g_v3n < maxRows ? g_v3n : maxRows;
  for (int k = 0; k < 24 * n; k++) out[k] = g_v3[k];
  return g_v3n;
}
#endif

extern "C" void tcdev_solve(tco_t *g, double mu0, int maxIter, double ae0, double ae1, int32_t *status,
                            int32_t *iter) {
#if defined(verboseLevel) && verboseLevel >= 3
  TcCtx c = {g, g_v3, nullptr, 0, TC_V3_HOSTROWS};
#else
  TcCtx c = {g};
#endif
  int st = 0, it = 0;
  g_ntrace = 0;
  TC_DEVICE_LOOP(c, mu0, maxIter, ae0, ae1, &st, &it);
  *status = st;
  *iter = it;
#if defined(verboseLevel) && verboseLevel >= 3
  g_v3n = c.v3n;
#endif
}

extern "C" int tcdev_trace(int32_t *out, int max) {
  const int n = g_ntrace < max ? g_ntrace : max;
  for (int k = 0; k < n; k++) out[k] = g_trace[k];
  return g_ntrace;
}

/* ORACLE - test infrastructure only.
 *
 * Glue that lets the reference's own Newton loops, /root/reference/lib/ipmPD_CSsolver.c and
 * lib/ipmPDeq_CSsolver.c (equilibrium solvers) compiled from where they lie, drive the oracle's tape interpreter: it defines the `extern void xxx__()`
 * tape entry points that file expects (ipmPD_CSsolver.c:28-51) by forwarding to tc_oracle.c.
 * In the reference these functions are MATLAB-generated code (not producible here: no MATLAB),
 * so this build pins ONLY the Newton-loop control flow of tco_solve(), not tape construction.
 */
#include <string.h>
#include "tc_oracle.h"

static tco_t *g;
void tcref_bind(tco_t *t) { g = t; }

#define COPY(fn) void fn(void) { tco_copy(g, tco_entry(g, #fn)); }
#define SET(fn) void fn(const double *x) { tco_set(g, tco_entry(g, #fn), x); }
#define GET1(fn) void fn(double *x) { tco_get_flat(g, tco_entry(g, #fn), x); }

COPY(initPrimal__)
COPY(initDualEq__)
COPY(initDualEqX__)
COPY(initDualIneq__)
COPY(updatePrimalDual__)
COPY(scaleIneq__)
COPY(scaleCost__)
SET(setAddEye2HessianU__)
SET(setAddEye2HessianEq__)
SET(setAddEye2Hessian1__)   /* the equilibrium loop's names, ipmPDeq_CSsolver.c:32-33 */
SET(setAddEye2Hessian2__)
SET(setAlphaPrimal__)
SET(setAlphaDualEq__)
SET(setAlphaDualIneq__)
SET(setMu__)
GET1(getDirectionError__)
GET1(getScale4Cost__)
GET1(getJ__)
GET1(getNorminf_G__)
GET1(getNorminf_Grad__)
GET1(getCurvature__)
GET1(getMinF_s__)
GET1(getMinF_a__)   /* skipAffine=0 builds (lib/ipmPD_CS.m assembly): predictor step, ipmPD_CSsolver.c:576-665 */
GET1(getRho__)
void getMaxAlphas_a__(double *alphaPrimal, double *alphaDual) {
  double t[2];
  tco_get_flat(g, tco_entry(g, "getMaxAlphas_a__"), t);
  *alphaPrimal = t[0];
  *alphaDual = t[1];
}
void getGapMinFMinLambda__(double *gap, double *ineq, double *dual) {
  double t[3];
  tco_get_flat(g, tco_entry(g, "getGapMinFMinLambda__"), t);
  *gap = t[0];
  *ineq = t[1];
  *dual = t[2];
}
void getMaxAlphas_s__(double *alphaPrimal, double *alphaDual) {
  double t[2];
  tco_get_flat(g, tco_entry(g, "getMaxAlphas_s__"), t);
  *alphaPrimal = t[0];
  *alphaDual = t[1];
}
void getHessInertia__(double *mp, double *mn) {
  double t[2];
  tco_get_flat(g, tco_entry(g, "getHessInertia__"), t);
  *mp = t[0];
  *mn = t[1];
}

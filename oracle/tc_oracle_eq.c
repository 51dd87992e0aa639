/* ORACLE - test infrastructure only.
 *
 * CPU restatement of the equilibrium solvers' Newton loop, /root/reference/lib/ipmPDeq_CSsolver.c:122-802, over the
 * tape interpreter of tc_oracle.c (public API only).  Kept in the reference's layout -- straight-line repetitions
 * and all -- so that it can be read side by side with that file; the solver configuration comes from the tape's
 * #define block (cmex2equilibriumLatentCS.m:417-448).  Pinned: tests/test_equilibrium.py compares it bit for bit with
 * the reference's own file compiled in place (oracle/build.py:build_ref(..., solver="ipmPDeq_CSsolver")).
 */
#include <math.h>
#include <string.h>
#include "tc_oracle.h"

#define MIN(A, B) (((A) < (B)) ? (A) : (B))
#define MAX(A, B) (((A) > (B)) ? (A) : (B))

static void get1(tco_t *t, int e, double *x) { tco_get_flat(t, e, x); }

void tco_solve_eq(tco_t *t, double mu0, int32_t maxIter, int32_t saveIter, const double *addEye2Hessian,
                  int32_t *status, int32_t *iter) {
  (void)saveIter;
  const int nF = (int)tco_define(t, "nF", 0), nG = (int)tco_define(t, "nG", 0);
  const double gradTolerance = tco_define(t, "gradTolerance", 1e-4), equalTolerance = tco_define(t, "equalTolerance", 1e-4),
               desiredDualityGap = tco_define(t, "desiredDualityGap", 1e-5),
               addEye2Hessian1tolerance = tco_define(t, "addEye2Hessian1tolerance", 1e-6),
               alphaMin = tco_define(t, "alphaMin", 1e-7), alphaMax = tco_define(t, "alphaMax", 1),
               muFactorAggressive = tco_define(t, "muFactorAggressive", .2),
               muFactorConservative = tco_define(t, "muFactorConservative", .95);
  const int scaleCost = (int)tco_define(t, "scaleCost", 0), scaleInequalities = (int)tco_define(t, "scaleInequalities", 1),
            coupledAlphas = (int)tco_define(t, "coupledAlphas", 1), setAddEye = (int)tco_define(t, "setAddEye2Hessian", 0),
            adjustAddEye = (int)tco_define(t, "adjustAddEye2Hessian", 0), skipAffine = (int)tco_define(t, "skipAffine", 1),
            delta = (int)tco_define(t, "delta", 3);
#define E(name) tco_entry(t, name)
  const int e_initPrimal = E("initPrimal__"), e_scaleCost = E("scaleCost__"), e_scaleIneq = E("scaleIneq__"),
            e_getScale4Cost = E("getScale4Cost__"), e_setMu = E("setMu__"), e_initDualIneq = E("initDualIneq__"),
            e_initDualEq = E("initDualEq__"), e_set1 = E("setAddEye2Hessian1__"), e_set2 = E("setAddEye2Hessian2__"),
            e_grad = E("getNorminf_Grad__"), e_G = E("getNorminf_G__"), e_gap = E("getGapMinFMinLambda__"),
            e_derr = E("getDirectionError__"), e_maxa = E("getMaxAlphas_a__"), e_maxs = E("getMaxAlphas_s__"),
            e_minFa = E("getMinF_a__"), e_minFs = E("getMinF_s__"), e_rho = E("getRho__"),
            e_aP = E("setAlphaPrimal__"), e_aDE = E("setAlphaDualEq__"), e_aDI = E("setAlphaDualIneq__"),
            e_update = E("updatePrimalDual__");
#undef E
  const double addEye2Hessian2MAX = 1e-2, addEye2HessianMIN = 1e-20, maxDirectionError = 1e-9, STEPBACK = .99;
  double tmp[3];
  *iter = 0;
  *status = 0;
  double addEye2Hessian1 = 0, addEye2Hessian2 = 0, derr = 0;
  int updateAddEye2Hessian2 = 0;
  if (setAddEye) { /* :134-139 */
    addEye2Hessian1 = addEye2Hessian[0];
    addEye2Hessian2 = addEye2Hessian[1];
    tco_set(t, e_set1, &addEye2Hessian1);
    tco_set(t, e_set2, &addEye2Hessian2);
  }
  double norminf_grad = 0, alphaMax_ = alphaMax, norminf_eq = 0;
  double mu = mu0, muMin = 0, gap = 0, ineq = 0, ineq1 = 0, dual = 0, scale4cost, desiredDualityGapVar = desiredDualityGap,
         alphaPrimal = 0, alphaDualEq = 0, alphaDualIneq = 0, sigma;

  tco_copy(t, e_initPrimal); /* :163 */
  if (scaleCost) tco_copy(t, e_scaleCost);
  if (nF > 0) {
    if (scaleInequalities) tco_copy(t, e_scaleIneq);
    tco_set(t, e_setMu, &mu);
    if (scaleCost) {
      get1(t, e_getScale4Cost, &scale4cost);
      desiredDualityGapVar *= scale4cost;
    }
    muMin = desiredDualityGapVar / nF / 2;
  }
  if (nF > 0) tco_copy(t, e_initDualIneq); /* :216-221 */
  if (nG > 0) tco_copy(t, e_initDualEq);

  while (1) {
    (*iter)++;
    if ((*iter) > maxIter) {
      (*status) = 8;
      break;
    }
    /* check exit conditions :244-312 */
    get1(t, e_grad, &norminf_grad);
    if (isnan(norminf_grad)) {
      (*status) = 4;
      break;
    }
    if (nG > 0) get1(t, e_G, &norminf_eq);
    if (nF > 0) {
      tco_get_flat(t, e_gap, tmp);
      gap = tmp[0];
      ineq = tmp[1];
      dual = tmp[2];
      if (ineq <= 0) {
        (*status) = 1;
        break;
      }
      if (dual <= 0) {
        (*status) = 2;
        break;
      }
    }
    if (norminf_grad <= gradTolerance && (nF == 0 || gap <= desiredDualityGapVar) && (nG == 0 || norminf_eq <= equalTolerance) &&
        (!(setAddEye && adjustAddEye) || addEye2Hessian1 <= addEye2Hessian1tolerance)) {
      (*status) = 0;
      break;
    }
    /* adjust addEye2Hessian :321-385 */
    if (setAddEye && adjustAddEye) {
      if (updateAddEye2Hessian2) {
        tco_set(t, e_set2, &addEye2Hessian2);
        updateAddEye2Hessian2 = 0;
      }
      get1(t, e_derr, &derr);
      if (derr < maxDirectionError) {
        if (addEye2Hessian2 > addEye2HessianMIN) {
          addEye2Hessian2 = MAX(.75 * addEye2Hessian2, addEye2HessianMIN);
          updateAddEye2Hessian2 = 1;
        }
      } else {
        for (int ii = 0; ii < 20; ii++) {
          int change = 0;
          if (derr >= maxDirectionError) {
            if (addEye2Hessian2 < addEye2Hessian2MAX) {
              addEye2Hessian2 = MIN(2 * MAX(addEye2Hessian2, addEye2HessianMIN), addEye2Hessian2MAX);
              tco_set(t, e_set2, &addEye2Hessian2);
              change = 1;
            }
          }
          if (!change) break;
          get1(t, e_derr, &derr);
        }
      }
    } else if (adjustAddEye) {
      get1(t, e_derr, &derr); /* :382 */
    }

    if (nF == 0) { /* :387-405 */
      tco_set(t, e_aP, &alphaMax_);
      if (nG > 0) tco_set(t, e_aDE, &alphaMax_);
      tco_copy(t, e_update);
      continue;
    }
    if (!skipAffine) { /* affine search direction :419-510 */
      tco_get_flat(t, e_maxa, tmp);
      alphaPrimal = tmp[0];
      alphaDualIneq = tmp[1];
      alphaMax_ = (alphaPrimal < alphaMax) ? alphaPrimal : alphaMax;
      if (alphaDualIneq < alphaMax_) alphaMax_ = alphaDualIneq;
      if (alphaMax_ >= alphaMin) {
        alphaPrimal = alphaMax_; /* try max */
        tco_set(t, e_aP, &alphaPrimal);
        get1(t, e_minFa, &ineq);
        if (ineq < 0) {
          alphaPrimal = alphaMin; /* try min */
          tco_set(t, e_aP, &alphaPrimal);
          get1(t, e_minFa, &ineq);
          if (ineq > 0) { /* try between min and max */
            for (alphaPrimal = alphaMax_ * .95; alphaPrimal >= alphaMin; alphaPrimal /= 2) {
              tco_set(t, e_aP, &alphaPrimal);
              get1(t, e_minFa, &ineq);
              if (ineq >= 0) break;
            }
            if (alphaPrimal < alphaMin) {
              alphaPrimal = 0;
              tco_set(t, e_aP, &alphaPrimal);
            }
          } else {
            alphaPrimal = 0;
            tco_set(t, e_aP, &alphaPrimal);
          }
        }
      } else {
        alphaPrimal = 0;
        tco_set(t, e_aP, &alphaPrimal);
      }
      tco_set(t, e_aDI, &alphaPrimal);
      if (alphaPrimal > alphaMax / 2 && (nG == 0 || norminf_eq < 100 * equalTolerance)) { /* :484-505 */
        get1(t, e_rho, &sigma);
        if (sigma > 1) sigma = 1;
        if (sigma < 0) sigma = 0;
        if (delta == 2)
          sigma = sigma * sigma;
        else
          sigma = sigma * sigma * sigma;
        mu = sigma * gap / nF;
        if (mu < muMin) mu = muMin;
        tco_set(t, e_setMu, &mu);
      }
    }
    /* combined search direction :525-631 */
    tco_get_flat(t, e_maxs, tmp);
    alphaPrimal = tmp[0];
    alphaDualIneq = tmp[1];
    if (coupledAlphas)
      if (alphaDualIneq < alphaPrimal) alphaPrimal = alphaDualIneq;
    alphaPrimal *= STEPBACK;
    alphaMax_ = (alphaPrimal < alphaMax) ? alphaPrimal : alphaMax;
    if (alphaMax_ >= alphaMin) {
      alphaPrimal = alphaMax_ / STEPBACK; /* try max */
      tco_set(t, e_aP, &alphaPrimal);
      get1(t, e_minFs, &ineq);
      if (isnan(ineq)) {
        (*status) = 4;
        break;
      }
      if (ineq > 0) { /* recheck just to be safe in case not convex */
        alphaPrimal *= STEPBACK;
        tco_set(t, e_aP, &alphaPrimal);
        get1(t, e_minFs, &ineq1);
      }
      if (ineq <= 0 || ineq1 <= ineq / 10) {
        alphaPrimal = alphaMin / STEPBACK; /* try min */
        tco_set(t, e_aP, &alphaPrimal);
        get1(t, e_minFs, &ineq);
        if (ineq > 0) { /* try between min and max */
          for (alphaPrimal = alphaMax_ * .95; alphaPrimal >= alphaMin; alphaPrimal /= 2) {
            tco_set(t, e_aP, &alphaPrimal);
            get1(t, e_minFs, &ineq);
            if (ineq > 0) {
              alphaPrimal *= STEPBACK; /* backtrace just a little */
              tco_set(t, e_aP, &alphaPrimal);
              get1(t, e_minFs, &ineq1);
              if (ineq1 > ineq / 10) break;
            }
          }
          if (alphaPrimal < alphaMin) {
            alphaPrimal = 0;
            tco_set(t, e_aP, &alphaPrimal);
          }
        } else {
          alphaPrimal = 0;
          tco_set(t, e_aP, &alphaPrimal);
        }
      }
    } else {
      alphaPrimal = 0;
      tco_set(t, e_aP, &alphaPrimal);
    }
    if (coupledAlphas) { /* :616-624 */
      alphaDualEq = alphaPrimal;
      alphaDualIneq = alphaPrimal;
    } else {
      alphaDualIneq *= STEPBACK;
      if (alphaDualIneq > alphaMax_) alphaDualIneq = alphaMax_;
      alphaDualEq = alphaMax;
    }
    if (nG > 0) tco_set(t, e_aDE, &alphaDualEq);
    tco_set(t, e_aDI, &alphaDualIneq);
    tco_copy(t, e_update);

    if (skipAffine) { /* :640-699 */
      int th_grad = norminf_grad < MAX(1e-3, 1e0 * gradTolerance);
      int th_eq = nG > 0 ? norminf_eq < MAX(1e-5, 1e0 * equalTolerance) : 1;
      if (alphaPrimal > alphaMax / 2 && th_grad && th_eq) {
        mu *= MIN(muFactorAggressive, pow(mu, .5));
        if (mu < muMin) mu = muMin;
        tco_set(t, e_setMu, &mu);
      } else {
        if (alphaPrimal < .02) {
          mu *= 1.1;
          if (mu > 100 * mu0) mu = 100 * mu0;
          tco_set(t, e_setMu, &mu);
          tco_copy(t, e_initDualIneq);
        } else {
          mu *= muFactorConservative;
          if (mu < muMin) mu = muMin;
          tco_set(t, e_setMu, &mu);
        }
      }
    }
    /* if no motion, slowly increase mu :702-709 */
    if (alphaPrimal < alphaMin && alphaDualIneq < alphaMin && alphaDualEq < alphaMin) {
      mu /= (muFactorConservative * muFactorConservative);
      if (mu < muMin) mu = muMin;
      tco_set(t, e_setMu, &mu);
    }
  }
  if ((*status) == 8) { /* :716-744 */
    get1(t, e_grad, &norminf_grad);
    if (norminf_grad > gradTolerance) (*status) |= 16;
    if (nG > 0) {
      get1(t, e_G, &norminf_eq);
      if (norminf_eq > equalTolerance) (*status) |= 32;
    }
    if (nF > 0) {
      tco_get_flat(t, e_gap, tmp);
      gap = tmp[0];
      if (gap > desiredDualityGapVar) (*status) |= 64;
      if (mu > muMin) (*status) |= 128;
      if (alphaPrimal <= alphaMin && alphaDualIneq < alphaMin && alphaDualEq < alphaMin)
        (*status) |= 1792;
      else if (alphaPrimal <= .1 && alphaDualIneq < .1 && alphaDualEq < .1)
        (*status) |= 1536;
      else if (alphaPrimal <= .5 && alphaDualIneq < .5 && alphaDualEq < .5)
        (*status) |= 1024;
    }
  }
}

void tco_solve_eq_batch(tco_t *t, int nSets, const int *setEntries, const double *const *in, int eOut, double *out,
                        int32_t *status, int32_t *iter, int64_t B, double mu0, int32_t maxIter,
                        const double *addEye2Hessian) {
  const int64_t nOut = tco_entry_size(t, eOut);
  for (int64_t k = 0; k < B; k++) {
    for (int s = 0; s < nSets; s++) tco_set(t, setEntries[s], in[s] + k * tco_entry_size(t, setEntries[s]));
    tco_solve_eq(t, mu0, maxIter, -1, addEye2Hessian, &status[k], &iter[k]);
    tco_get_flat(t, eOut, out + k * nOut);
  }
}

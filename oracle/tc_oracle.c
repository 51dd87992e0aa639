/* ORACLE - test infrastructure only.  Nothing under tenscalc_amd/ may import, link or execute
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the
 * checker.
 *
 * CPU restatement of the reference's run-time path for one problem instance:
 *   - scalar tape interpreter with the per-opcode arithmetic (operand order, statement
 *     chunking) that writeCinstructionsC prints:
 *       /root/reference/lib/csparse/instructionsTableUTHash.c:950-1599
 *   - lazy dependency-group evaluation and set/get/copy semantics of the generated C:
 *       /root/reference/lib/@csparse/writeCfunctionpergroup.m:513-852
 *   - the primal-dual Newton loop:
 *       /root/reference/lib/ipmPD_CSsolver.c:132-1000
 *
 * Parity pinning: the reference holds no golden vectors for this path and
 * its tape generator needs MATLAB.  The Newton-loop restatement below is pinned against the
 * reference's own ipmPD_CSsolver.c compiled in place (oracle/_ref, see oracle/Makefile and
 * tests/test_oracle_vs_reference_loop.py); tape construction is pinned against an
 * independent dense closed form and scipy (tests/test_oracle_sls.py).
 */
#include "tc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ tape container ---- */
typedef struct {
  char name[32];
  int64_t dtype, count;
  const void *data;
} rec_t;

struct tco {
  char *file;
  size_t fileLen;
  rec_t *recs;
  int nRecs;
  /* tape */
  int64_t n, nSlots, nGroups;
  const int64_t *types, *opStart, *ops, *parStart, *group, *slot, *grpStart, *grpInstr, *grpFlops;
  const double *pars;
  /* entries */
  int nEntries;
  char **entryNames;
  const int64_t *entryKind;
  const int64_t **eA, **eB, **eParents, **eChildren;
  int64_t *nA, *nB, *nParents, *nChildren;
  /* state */
  double *m;
  unsigned char *groupStatus;
  /* solver configuration (#define block of cmex2optimizeCS.m:303-331) */
  tco_defines_t d;
  /* cached entry indices for the Newton loop */
  int e_initPrimal, e_scaleIneq, e_scaleCost, e_getScale4Cost, e_setMu, e_initDualIneq,
      e_initDualEqX, e_setAddEyeU, e_setAddEyeEq, e_getNorminfGrad, e_getNorminfG,
      e_getGapMinFMinLambda, e_getCurvature, e_getHessInertia, e_getDirectionError,
      e_getMaxAlphas_s, e_getMinF_s, e_setAlphaPrimal, e_setAlphaDualEq, e_setAlphaDualIneq,
      e_updatePrimalDual, e_getJ, e_getMaxAlphas_a, e_getMinF_a, e_getRho;
  /* statistics */
  int64_t flops, groupEvals, minFtrials;
};

static const rec_t *find(const tco_t *t, const char *name) {
  for (int i = 0; i < t->nRecs; i++)
    if (!strcmp(t->recs[i].name, name)) return &t->recs[i];
  return NULL;
}
static const int64_t *geti(const tco_t *t, const char *name, int64_t *cnt) {
  const rec_t *r = find(t, name);
  if (!r) {
    if (cnt) *cnt = 0;
    return NULL;
  }
  if (cnt) *cnt = r->count;
  return (const int64_t *)r->data;
}

static double define(const tco_t *t, const char *name, double dflt) {
  const rec_t *rn = find(t, "defines_names"), *rv = find(t, "defines_values");
  if (!rn || !rv) return dflt;
  const char *s = (const char *)rn->data;
  size_t len = (size_t)rn->count, nl = strlen(name);
  int64_t k = 0;
  size_t i = 0;
  while (i < len) {
    size_t j = i;
    while (j < len && s[j] != '\n') j++;
    if (j - i == nl && !memcmp(s + i, name, nl)) return ((const double *)rv->data)[k];
    k++;
    i = j + 1;
  }
  return dflt;
}

int tco_entry(const tco_t *t, const char *name) {
  for (int i = 0; i < t->nEntries; i++)
    if (!strcmp(t->entryNames[i], name)) return i;
  return -1;
}

tco_t *tco_load(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long len = ftell(f);
  fseek(f, 0, SEEK_SET);
  tco_t *t = (tco_t *)calloc(1, sizeof *t);
  t->file = (char *)malloc((size_t)len);
  t->fileLen = (size_t)len;
  if (fread(t->file, 1, (size_t)len, f) != (size_t)len || memcmp(t->file, "TCTAPE1", 8)) {
    fclose(f);
    free(t->file);
    free(t);
    return NULL;
  }
  fclose(f);
  size_t off = 8;
  int cap = 64;
  t->recs = (rec_t *)malloc(sizeof(rec_t) * cap);
  while (off + 48 <= t->fileLen) {
    if (t->nRecs == cap) t->recs = (rec_t *)realloc(t->recs, sizeof(rec_t) * (cap *= 2));
    rec_t *r = &t->recs[t->nRecs++];
    memcpy(r->name, t->file + off, 32);
    memcpy(&r->dtype, t->file + off + 32, 8);
    memcpy(&r->count, t->file + off + 40, 8);
    off += 48;
    r->data = t->file + off;
    size_t bytes = r->dtype == 2 ? (size_t)r->count + (size_t)((-r->count) & 7) : (size_t)r->count * 8;
    off += bytes;
    if (!strcmp(r->name, "END")) break;
  }
  t->types = geti(t, "types", &t->n);
  t->opStart = geti(t, "opStart", NULL);
  t->ops = geti(t, "ops", NULL);
  t->parStart = geti(t, "parStart", NULL);
  t->pars = (const double *)find(t, "pars")->data;
  t->group = geti(t, "group", NULL);
  t->slot = geti(t, "slot", NULL);
  t->nSlots = geti(t, "nSlots", NULL)[0];
  t->nGroups = geti(t, "nGroups", NULL)[0];
  t->grpStart = geti(t, "grpStart", NULL);
  t->grpInstr = geti(t, "grpInstr", NULL);
  t->grpFlops = geti(t, "grpFlops", NULL);
  /* entries */
  const rec_t *en = find(t, "entry_names");
  int64_t nk;
  t->entryKind = geti(t, "entry_kind", &nk);
  t->nEntries = (int)nk;
  t->entryNames = (char **)calloc((size_t)nk, sizeof(char *));
  {
    const char *s = (const char *)en->data;
    size_t i = 0, L = (size_t)en->count;
    for (int k = 0; k < t->nEntries; k++) {
      size_t j = i;
      while (j < L && s[j] != '\n') j++;
      t->entryNames[k] = (char *)calloc(j - i + 1, 1);
      memcpy(t->entryNames[k], s + i, j - i);
      i = j + 1;
    }
  }
  t->eA = (const int64_t **)calloc((size_t)nk, sizeof(void *));
  t->eB = (const int64_t **)calloc((size_t)nk, sizeof(void *));
  t->eParents = (const int64_t **)calloc((size_t)nk, sizeof(void *));
  t->eChildren = (const int64_t **)calloc((size_t)nk, sizeof(void *));
  t->nA = (int64_t *)calloc((size_t)nk, 8);
  t->nB = (int64_t *)calloc((size_t)nk, 8);
  t->nParents = (int64_t *)calloc((size_t)nk, 8);
  t->nChildren = (int64_t *)calloc((size_t)nk, 8);
  for (int k = 0; k < t->nEntries; k++) {
    char nm[40];
    snprintf(nm, sizeof nm, "e%d_a", k);
    t->eA[k] = geti(t, nm, &t->nA[k]);
    snprintf(nm, sizeof nm, "e%d_b", k);
    t->eB[k] = geti(t, nm, &t->nB[k]);
    snprintf(nm, sizeof nm, "e%d_parents", k);
    t->eParents[k] = geti(t, nm, &t->nParents[k]);
    snprintf(nm, sizeof nm, "e%d_children", k);
    t->eChildren[k] = geti(t, nm, &t->nChildren[k]);
  }
  /* state: constants are loaded once (the reference computes I_load in a group that no set
   * ever invalidates) */
  t->m = (double *)calloc((size_t)t->nSlots + 1, sizeof(double));
  t->groupStatus = (unsigned char *)calloc((size_t)t->nGroups + 1, 1);
  for (int64_t i = 1; i <= t->n; i++)
    if (t->types[i - 1] == 2 /* I_load */ && t->slot[i] >= 0) t->m[t->slot[i]] = t->pars[t->parStart[i - 1]];
  /* defines */
  tco_defines_t *d = &t->d;
  d->nU = (int)define(t, "nU", 0);
  d->nG = (int)define(t, "nG", 0);
  d->nF = (int)define(t, "nF", 0);
  d->gradTolerance = define(t, "gradTolerance", 1e-4);
  d->equalTolerance = define(t, "equalTolerance", 1e-4);
  d->desiredDualityGap = define(t, "desiredDualityGap", 1e-5);
  d->addEye2HessianUtolerance = define(t, "addEye2HessianUtolerance", 1e-6);
  d->scaleCost = (int)define(t, "scaleCost", 0);
  d->scaleInequalities = (int)define(t, "scaleInequalities", 1);
  d->alphaMin = define(t, "alphaMin", 1e-7);
  d->alphaMax = define(t, "alphaMax", 1);
  d->coupledAlphas = (int)define(t, "coupledAlphas", 1);
  d->setAddEye2Hessian = (int)define(t, "setAddEye2Hessian", 1);
  d->adjustAddEye2Hessian = (int)define(t, "adjustAddEye2Hessian", 1);
  d->useInertia = (int)define(t, "useInertia", 0);
  d->muFactorAggressive = define(t, "muFactorAggressive", .2);
  d->muFactorConservative = define(t, "muFactorConservative", .95);
  d->skipAffine = (int)define(t, "skipAffine", 1);
  d->smallerNewtonMatrix = (int)define(t, "smallerNewtonMatrix", 0);
  d->delta = (int)define(t, "delta", 3);
#define E(field, name) t->field = tco_entry(t, name)
  E(e_initPrimal, "initPrimal__");
  E(e_scaleIneq, "scaleIneq__");
  E(e_scaleCost, "scaleCost__");
  E(e_getScale4Cost, "getScale4Cost__");
  E(e_setMu, "setMu__");
  E(e_initDualIneq, "initDualIneq__");
  E(e_initDualEqX, "initDualEqX__");
  E(e_setAddEyeU, "setAddEye2HessianU__");
  E(e_setAddEyeEq, "setAddEye2HessianEq__");
  E(e_getNorminfGrad, "getNorminf_Grad__");
  E(e_getNorminfG, "getNorminf_G__");
  E(e_getGapMinFMinLambda, "getGapMinFMinLambda__");
  E(e_getCurvature, "getCurvature__");
  E(e_getHessInertia, "getHessInertia__");
  E(e_getDirectionError, "getDirectionError__");
  E(e_getMaxAlphas_s, "getMaxAlphas_s__");
  E(e_getMinF_s, "getMinF_s__");
  E(e_getMaxAlphas_a, "getMaxAlphas_a__");
  E(e_getMinF_a, "getMinF_a__");
  E(e_getRho, "getRho__");
  E(e_setAlphaPrimal, "setAlphaPrimal__");
  E(e_setAlphaDualEq, "setAlphaDualEq__");
  E(e_setAlphaDualIneq, "setAlphaDualIneq__");
  E(e_updatePrimalDual, "updatePrimalDual__");
  E(e_getJ, "getJ__");
#undef E
  return t;
}

void tco_free(tco_t *t) {
  if (!t) return;
  for (int k = 0; k < t->nEntries; k++) free(t->entryNames[k]);
  free(t->entryNames);
  free((void *)t->eA);
  free((void *)t->eB);
  free((void *)t->eParents);
  free((void *)t->eChildren);
  free(t->nA);
  free(t->nB);
  free(t->nParents);
  free(t->nChildren);
  free(t->m);
  free(t->groupStatus);
  free(t->recs);
  free(t->file);
  free(t);
}

const tco_defines_t *tco_defines(const tco_t *t) { return &t->d; }
double tco_define(const tco_t *t, const char *name, double dflt) { return define(t, name, dflt); }
int64_t tco_flops(const tco_t *t) { return t->flops; }
int64_t tco_group_evals(const tco_t *t) { return t->groupEvals; }
int64_t tco_minF_trials(const tco_t *t) { return t->minFtrials; }
void tco_reset_counters(tco_t *t) { t->flops = t->groupEvals = t->minFtrials = 0; }
double tco_value(const tco_t *t, int64_t instr) { return t->m[t->slot[instr]]; }
int tco_num_entries(const tco_t *t) { return t->nEntries; }
const char *tco_entry_name(const tco_t *t, int e) { return t->entryNames[e]; }
int tco_entry_kind(const tco_t *t, int e) { return (int)t->entryKind[e]; }
int64_t tco_entry_size(const tco_t *t, int e) { return t->nA[e]; }

/* ------------------------------------------------------------------ interpreter ------- */
#define MAX_TERMS_PERLINE 100 /* instructionsTableUTHash.c:173 */
#define M(k) m[slot[ops[k]]]

static void run_instruction(tco_t *t, int64_t i) {
  double *m = t->m;
  const int64_t *slot = t->slot;
  const int64_t *ops = t->ops + t->opStart[i - 1];
  int64_t nOps = t->opStart[i] - t->opStart[i - 1];
  const double *par = t->pars + t->parStart[i - 1];
  double *r = &m[slot[i]];
  double x, acc;
  int64_t k;
  switch (t->types[i - 1]) {
    case 1: /* I_set */
    case 2: /* I_load */
      break;
    case 3: /* I_sum :960-985  m[r]=+m[a]-m[b]... */
      acc = par[0] > 0 ? +M(0) : -M(0);
      for (k = 1; k < nOps; k++) acc = par[k] > 0 ? acc + M(k) : acc - M(k);
      *r = acc;
      break;
    case 4: { /* I_sumprod :987-1061; a new "m[r]+=" statement every MAX_TERMS_PERLINE factors */
      int64_t nf = (int64_t)par[0], ns = (int64_t)par[1], s, p, nSums = 0, o = 0;
      double total = 0, line = 0;
      int first = 1, lineEmpty = 1;
      for (s = 0; s < ns; s++) {
        x = M(o);
        o++;
        for (p = 1; p < nf; p++, o++) x = x * M(o);
        line = lineEmpty ? x : line + x;
        lineEmpty = 0;
        if (s + 1 < ns && (nSums += nf) >= MAX_TERMS_PERLINE) {
          total = first ? line : total + line;
          first = 0;
          nSums = 0;
          lineEmpty = 1;
        }
      }
      *r = first ? line : total + line;
      break;
    }
    case 5: /* I_inv */
      *r = 1 / M(0);
      break;
    case 6: /* I_minus_inv_sqr */
      *r = -1 / (M(0) * M(0));
      break;
    case 7: /* I_2_inv_cube */
      *r = 2 / (M(0) * M(0) * M(0));
      break;
    case 8: /* I_div */
      *r = M(0) / M(1);
      break;
    case 9: /* I_minus_dot :1110-1127 */
      acc = M(0) * M(1);
      for (k = 2; k < nOps; k += 2) acc = acc + M(k) * M(k + 1);
      *r = -(acc);
      break;
    case 10: /* I_minus_dot_div :1090-1108  -(...)/m[d] */
      acc = M(1) * M(2);
      for (k = 3; k < nOps; k += 2) acc = acc + M(k) * M(k + 1);
      *r = -(acc) / M(0);
      break;
    case 11: /* I_plus_minus_dot :1128-1144 */
      acc = M(1) * M(2);
      for (k = 3; k < nOps; k += 2) acc = acc + M(k) * M(k + 1);
      *r = M(0) - (acc);
      break;
    case 12: /* I_plus_sqr */
      acc = M(0) * M(0);
      for (k = 1; k < nOps; k++) acc = acc + M(k) * M(k);
      *r = acc;
      break;
    case 13: /* I_plus_abs */
      acc = fabs(M(0));
      for (k = 1; k < nOps; k++) acc = acc + fabs(M(k));
      *r = acc;
      break;
    case 14: /* I_plus_minus_dot_div :1067-1088 */
      acc = M(2) * M(3);
      for (k = 4; k < nOps; k += 2) acc = acc + M(k) * M(k + 1);
      *r = (M(0) - (acc)) / M(1);
      break;
    case 15: /* I_min */
      acc = M(0);
      for (k = 1; k < nOps; k++)
        if (acc > M(k)) acc = M(k);
      *r = acc;
      break;
    case 16: /* I_min0 */
      acc = 0;
      for (k = 0; k < nOps; k++)
        if (acc > M(k)) acc = M(k);
      *r = acc;
      break;
    case 17: /* I_max */
      acc = M(0);
      for (k = 1; k < nOps; k++)
        if (acc < M(k)) acc = M(k);
      *r = acc;
      break;
    case 18: /* I_max0 */
      acc = 0;
      for (k = 0; k < nOps; k++)
        if (acc < M(k)) acc = M(k);
      *r = acc;
      break;
    case 19: /* I_max_abs */
      acc = fabs(M(0));
      for (k = 1; k < nOps; k++)
        if (acc < fabs(M(k))) acc = fabs(M(k));
      *r = acc;
      break;
    case 20: /* I_clp :1315-1331 */
      acc = DBL_MAX;
      for (k = 0; k < nOps; k += 2)
        if (M(k + 1) < 0) {
          x = -M(k) / M(k + 1);
          if (acc > x) acc = x;
        }
      *r = acc;
      break;
    case 21: *r = exp(M(0)); break;
    case 22: *r = log(M(0)); break;
    case 23: *r = cos(M(0)); break;
    case 24: *r = -cos(M(0)); break;
    case 25: *r = sin(M(0)); break;
    case 26: *r = -sin(M(0)); break;
    case 27: *r = round(M(0)); break;
    case 28: *r = ceil(M(0)); break;
    case 29: *r = floor(M(0)); break;
    case 30: *r = fabs(M(0)); break;
    case 31: *r = (M(0) >= 0) ? 1 : -1; break;
    case 32: *r = sqrt(M(0)); break;
    case 33: *r = .5 * pow(M(0), -.5); break;
    case 34: *r = -.25 * pow(M(0), -1.5); break;
    case 35: *r = M(0) * M(0); break;
    case 36: *r = 2 * M(0); break;
    case 37: *r = 2; break;
    case 38: *r = M(0) * M(0) * M(0); break;
    case 39: *r = 3 * M(0) * M(0); break;
    case 40: *r = 6 * M(0); break;
    case 41: *r = atan(M(0)); break;
    case 42: *r = 1 / (1 + M(0) * M(0)); break;
    case 43: *r = -2 * M(0) / (1 + M(0) * M(0)) / (1 + M(0) * M(0)); break;
    case 44: *r = log(1 + exp(M(0))); break;
    case 45: *r = 1 / (1 + exp(-M(0))); break;
    case 46: *r = 1 / (2 + exp(-M(0)) + exp(M(0))); break;
    case 47: *r = (M(0) > 0) ? M(0) : 0; break;
    case 48: *r = (M(0) > 0) ? 1 : 0; break;
    case 49: *r = 0; break; /* I_zero */
    case 50: { /* I_componentwise :1333-1345: C text in the parameters, 0 = the operand.  An interpreter cannot run
                  arbitrary C: the plain calls NAME(operand) of the functions below are evaluated, anything else aborts */
      char nm[32];
      int64_t np_ = t->parStart[i] - t->parStart[i - 1], q = 0, hole = -1;
      size_t ln = 0;
      for (q = 0; q < np_; q++) {
        int ch = (int)par[q];
        if (ch == 0) { hole = q; continue; }
        if (ch == ' ' || ch == '(' || ch == ')') continue;
        if (hole >= 0 || ln + 1 >= sizeof nm) { ln = 0; break; }
        nm[ln++] = (char)ch;
      }
      nm[ln] = 0;
      x = M(0);
      if (!strcmp(nm, "fabs") || !strcmp(nm, "abs")) *r = fabs(x);
      else if (!strcmp(nm, "sqrt")) *r = sqrt(x);
      else if (!strcmp(nm, "exp")) *r = exp(x);
      else if (!strcmp(nm, "log")) *r = log(x);
      else if (!strcmp(nm, "sin")) *r = sin(x);
      else if (!strcmp(nm, "cos")) *r = cos(x);
      else if (!strcmp(nm, "atan")) *r = atan(x);
      else if (!strcmp(nm, "round")) *r = round(x);
      else if (!strcmp(nm, "ceil")) *r = ceil(x);
      else if (!strcmp(nm, "floor")) *r = floor(x);
      else {
        fprintf(stderr, "tc_oracle: I_componentwise text not understood (only NAME(operand) calls)\n");
        abort();
      }
      break;
    }
    default:
      fprintf(stderr, "tc_oracle: unsupported opcode %lld\n", (long long)t->types[i - 1]);
      abort();
  }
}

static void compute_groups(tco_t *t, int e) {
  /* "if(groupStatus[g]==0){groupStatus[g]=1;<cls>_group<g>();}" writeCfunctionpergroup.m:44,653-657 */
  for (int64_t k = 0; k < t->nParents[e]; k++) {
    int64_t g = t->eParents[e][k];
    if (!t->groupStatus[g]) {
      t->groupStatus[g] = 1;
      for (int64_t j = t->grpStart[g - 1]; j < t->grpStart[g]; j++) run_instruction(t, t->grpInstr[j]);
      t->flops += t->grpFlops[g];
      t->groupEvals++;
    }
  }
}
static void invalidate(tco_t *t, int e) {
  for (int64_t k = 0; k < t->nChildren[e]; k++) t->groupStatus[t->eChildren[e][k]] = 0;
}

int tco_set(tco_t *t, int e, const double *in) { /* writeCfunctionpergroup.m:513-572 */
  if (e < 0 || e >= t->nEntries || t->entryKind[e] != 0) return -1;
  for (int64_t k = 0; k < t->nA[e]; k++) t->m[t->slot[t->eA[e][k]]] = in[k];
  invalidate(t, e);
  return 0;
}

int tco_get(tco_t *t, int e, double **outs) { /* writeCfunctionpergroup.m:583-751 */
  if (e < 0 || e >= t->nEntries || t->entryKind[e] != 1) return -1;
  compute_groups(t, e);
  int64_t nOut = t->nB[e] - 1;
  for (int64_t o = 0; o < nOut; o++)
    for (int64_t k = t->eB[e][o]; k < t->eB[e][o + 1]; k++)
      outs[o][k - t->eB[e][o]] = t->m[t->slot[t->eA[e][k]]];
  return 0;
}

int tco_get_flat(tco_t *t, int e, double *out) {
  if (e < 0 || e >= t->nEntries || t->entryKind[e] != 1) return -1;
  compute_groups(t, e);
  for (int64_t k = 0; k < t->nA[e]; k++) out[k] = t->m[t->slot[t->eA[e][k]]];
  return 0;
}

int tco_copy(tco_t *t, int e) { /* writeCfunctionpergroup.m:757-852 */
  if (e < 0 || e >= t->nEntries || t->entryKind[e] != 2) return -1;
  compute_groups(t, e);
  for (int64_t k = 0; k < t->nA[e]; k++) t->m[t->slot[t->eA[e][k]]] = t->m[t->slot[t->eB[e][k]]];
  invalidate(t, e);
  return 0;
}

/* ------------------------------------------------------------------ Newton loop -------- */
#define MIN(A, B) (((A) < (B)) ? (A) : (B))
#define MAX(A, B) (((A) > (B)) ? (A) : (B))
static void get1(tco_t *t, int e, double *a) { tco_get_flat(t, e, a); }
static void getMinF(tco_t *t, double *a) {
  t->minFtrials++;
  tco_get_flat(t, t->e_getMinF_s, a);
}

/* ipmPD_CSsolver(), lib/ipmPD_CSsolver.c:132-1000: skipAffine=1 (the variant cmex2optimizeCS generates,
 * cmex2optimizeCS.m:83-95) and skipAffine=0 / smallerNewtonMatrix (lib/ipmPD_CS.m assembly: Mehrotra predictor
 * step :576-665).  Compile-time #if's of the template become run-time tests on tco_defines_t. */
void tco_solve(tco_t *t, double mu0, int32_t maxIter, int32_t saveIter, const double *addEye2Hessian,
               int32_t *status, int32_t *iter) {
  const tco_defines_t *d = &t->d;
  const int nF = d->nF, nG = d->nG;
  const double addEye2HessianUMAX = 1e2, addEye2HessianEqMAX = 1e2, addEye2HessianMIN = 1e-20,
               maxDirectionError = 1e-6; /* :145-148 */
  double addEye2HessianU = 0, addEye2HessianEq = 0;
  double derr = 0, curvature = 0, mp = 0, mn = 0, tmp[4];
  int updateAddEye2HessianU = 0, updateAddEye2HessianEq = 0;
  const int adjust = d->setAddEye2Hessian && d->adjustAddEye2Hessian;
  const double mpDesired = d->nU, mnDesired = d->smallerNewtonMatrix ? nG : nF + nG; /* :164-169 */
  double sigma = 0;
  double norminf_grad, scale4cost, alphaMax_ = d->alphaMax;
  double mu = mu0, stepback = .99, muMin = 0, gap = 0, ineq = 0, ineq1 = 0, dual = 0,
         desiredDualityGapVar = d->desiredDualityGap, alphaPrimal = 0, alphaDualEq = 0,
         alphaDualIneq = 0;
  double norminf_eq = 0;
  (void)saveIter;
  *iter = 0;
  *status = 0;
  if (d->setAddEye2Hessian) { /* :150-155 */
    addEye2HessianU = addEye2Hessian[0];
    addEye2HessianEq = addEye2Hessian[1];
    tco_set(t, t->e_setAddEyeU, &addEye2HessianU);
    tco_set(t, t->e_setAddEyeEq, &addEye2HessianEq);
  }
  tco_copy(t, t->e_initPrimal); /* :204 */
  if (d->scaleCost) {           /* :206-214 */
    tco_copy(t, t->e_scaleCost);
    get1(t, t->e_getScale4Cost, &scale4cost);
    if (nF > 0) desiredDualityGapVar *= scale4cost;
  } else
    scale4cost = 0.0;
  if (nF > 0) { /* :216-230 */
    if (d->scaleInequalities) tco_copy(t, t->e_scaleIneq);
    tco_set(t, t->e_setMu, &mu);
    muMin = desiredDualityGapVar / nF / 2;
    tco_copy(t, t->e_initDualIneq);
  }
  if (nG > 0) tco_copy(t, t->e_initDualEqX); /* :232-235 */

  while (1) { /* :282 */
    (*iter)++;
    if ((*iter) > maxIter) { /* :308-313 */
      (*status) = 8;
      break;
    }
    get1(t, t->e_getNorminfGrad, &norminf_grad); /* :323 */
    if (isnan(norminf_grad)) {
      (*status) = 4;
      break;
    }
    if (nG > 0) get1(t, t->e_getNorminfG, &norminf_eq);
    if (nF > 0) { /* :343-357 */
      tco_get_flat(t, t->e_getGapMinFMinLambda, tmp);
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
    if (norminf_grad <= d->gradTolerance && (nF == 0 || gap <= desiredDualityGapVar) &&
        (nG == 0 || norminf_eq <= d->equalTolerance) &&
        (!adjust || addEye2HessianU <= d->addEye2HessianUtolerance)) { /* :362-381 */
      (*status) = 0;
      break;
    }
    if (adjust) { /* :393-543 */
      if (updateAddEye2HessianU) {
        tco_set(t, t->e_setAddEyeU, &addEye2HessianU);
        updateAddEye2HessianU = 0;
      }
      if (updateAddEye2HessianEq) {
        tco_set(t, t->e_setAddEyeEq, &addEye2HessianEq);
        updateAddEye2HessianEq = 0;
      }
      get1(t, t->e_getCurvature, &curvature);
      if (d->useInertia && curvature <= 0) {
        tco_get_flat(t, t->e_getHessInertia, tmp);
        mp = tmp[0];
        mn = tmp[1];
      }
      get1(t, t->e_getDirectionError, &derr);
      if (curvature > 0 || (d->useInertia && mp == mpDesired && mn == mnDesired)) {
        if (addEye2HessianU > addEye2HessianMIN && derr < maxDirectionError) {
          addEye2HessianU = MAX(.75 * addEye2HessianU, addEye2HessianMIN);
          updateAddEye2HessianU = 1;
        }
        if (addEye2HessianEq > addEye2HessianMIN && derr < maxDirectionError) {
          addEye2HessianEq = MAX(.75 * addEye2HessianEq, addEye2HessianMIN);
          updateAddEye2HessianEq = 1;
        }
        if (addEye2HessianU < addEye2HessianUMAX && derr > maxDirectionError) {
          addEye2HessianU = MIN(10 * addEye2HessianU, addEye2HessianUMAX);
          updateAddEye2HessianU = 1;
        }
        if (addEye2HessianEq < addEye2HessianEqMAX && derr > maxDirectionError) {
          addEye2HessianEq = MIN(10 * addEye2HessianEq, addEye2HessianEqMAX);
          updateAddEye2HessianEq = 1;
        }
      } else {
        for (int ii = 0; ii < 1 + 0 * 20; ii++) { /* :459 */
          int change = 0;
          if (!d->useInertia) {
            if (curvature < 0) { /* :462-480 */
              if (addEye2HessianU < addEye2HessianUMAX) {
                addEye2HessianU = MIN(10 * MAX(addEye2HessianU, addEye2HessianMIN), addEye2HessianUMAX);
                tco_set(t, t->e_setAddEyeU, &addEye2HessianU);
                change = 1;
              }
              if (addEye2HessianEq < addEye2HessianEqMAX) {
                addEye2HessianEq = MIN(10 * MAX(addEye2HessianEq, addEye2HessianMIN), addEye2HessianEqMAX);
                tco_set(t, t->e_setAddEyeEq, &addEye2HessianEq);
                change = 1;
              }
            }
          } else if (curvature < 0) { /* :482-519 */
            if (mp < mpDesired) {
              if (addEye2HessianU < addEye2HessianUMAX) {
                addEye2HessianU = MIN(10 * MAX(addEye2HessianU, addEye2HessianMIN), addEye2HessianUMAX);
                tco_set(t, t->e_setAddEyeU, &addEye2HessianU);
                change = 1;
              }
              if (addEye2HessianEq < addEye2HessianEqMAX) {
                addEye2HessianEq = MIN(2 * MAX(addEye2HessianEq, addEye2HessianMIN), addEye2HessianEqMAX);
                tco_set(t, t->e_setAddEyeEq, &addEye2HessianEq);
                change = 1;
              }
            } else if (mn < mnDesired) {
              if (addEye2HessianU < addEye2HessianUMAX) {
                addEye2HessianU = MIN(2 * MAX(addEye2HessianU, addEye2HessianMIN), addEye2HessianUMAX);
                tco_set(t, t->e_setAddEyeU, &addEye2HessianU);
                change = 1;
              }
              if (addEye2HessianEq < addEye2HessianEqMAX) {
                addEye2HessianEq = MIN(10 * MAX(addEye2HessianEq, addEye2HessianMIN), addEye2HessianEqMAX);
                tco_set(t, t->e_setAddEyeEq, &addEye2HessianEq);
                change = 1;
              }
            }
          }
          if (!change) break;
          get1(t, t->e_getCurvature, &curvature);
          if (d->useInertia && curvature <= 0) {
            tco_get_flat(t, t->e_getHessInertia, tmp);
            mp = tmp[0];
            mn = tmp[1];
          }
          get1(t, t->e_getDirectionError, &derr);
        }
      }
    }
    if (nF == 0) { /* :550-569 */
      tco_set(t, t->e_setAlphaPrimal, &alphaMax_);
      if (nG > 0) tco_set(t, t->e_setAlphaDualEq, &alphaMax_);
      tco_copy(t, t->e_updatePrimalDual);
    } else {
      if (!d->skipAffine) { /* affine search direction :576-665 */
        tco_get_flat(t, t->e_getMaxAlphas_a, tmp);
        alphaPrimal = tmp[0];
        alphaDualIneq = tmp[1];
        alphaMax_ = (alphaPrimal < d->alphaMax) ? alphaPrimal : d->alphaMax;
        if (alphaDualIneq < alphaMax_) alphaMax_ = alphaDualIneq;
        if (alphaMax_ >= d->alphaMin) {
          alphaPrimal = alphaMax_; /* try max */
          tco_set(t, t->e_setAlphaPrimal, &alphaPrimal);
          get1(t, t->e_getMinF_a, &ineq);
          if (ineq < 0) {
            alphaPrimal = d->alphaMin; /* try min */
            tco_set(t, t->e_setAlphaPrimal, &alphaPrimal);
            get1(t, t->e_getMinF_a, &ineq);
            if (ineq > 0) { /* try between min and max */
              for (alphaPrimal = alphaMax_ * .95; alphaPrimal >= d->alphaMin; alphaPrimal /= 2) {
                tco_set(t, t->e_setAlphaPrimal, &alphaPrimal);
                get1(t, t->e_getMinF_a, &ineq);
                if (ineq >= 0) break;
              }
              if (alphaPrimal < d->alphaMin) {
                alphaPrimal = 0;
                tco_set(t, t->e_setAlphaPrimal, &alphaPrimal);
              }
            } else {
              alphaPrimal = 0;
              tco_set(t, t->e_setAlphaPrimal, &alphaPrimal);
            }
          }
        } else {
          alphaPrimal = 0;
          tco_set(t, t->e_setAlphaPrimal, &alphaPrimal);
        }
        tco_set(t, t->e_setAlphaDualIneq, &alphaPrimal);
        /* update mu based on sigma :636-664 */
        if (alphaPrimal > d->alphaMax / 2 &&
            (nG == 0 || norminf_eq < 100 * d->equalTolerance || norminf_eq < 1e-3)) {
          get1(t, t->e_getRho, &sigma);
          if (sigma > 1) sigma = 1;
          if (sigma < 0) sigma = 0;
          if (d->delta == 2)
            sigma = sigma * sigma;
          else
            sigma = sigma * sigma * sigma;
          mu = sigma * gap / nF;
          if (mu < muMin) mu = muMin;
          tco_set(t, t->e_setMu, &mu);
        }
      }
      /* combined search direction :679-772 */
      tco_get_flat(t, t->e_getMaxAlphas_s, tmp);
      alphaPrimal = tmp[0];
      alphaDualIneq = tmp[1];
      if (d->coupledAlphas)
        if (alphaDualIneq < alphaPrimal) alphaPrimal = alphaDualIneq;
      alphaPrimal *= stepback;
      alphaMax_ = (alphaPrimal < d->alphaMax) ? alphaPrimal : d->alphaMax;
      if (alphaMax_ >= d->alphaMin) {
        alphaPrimal = alphaMax_ / stepback; /* try max */
        tco_set(t, t->e_setAlphaPrimal, &alphaPrimal);
        getMinF(t, &ineq);
        if (isnan(ineq)) {
          (*status) = 4;
          break;
        }
        if (ineq > 0) { /* recheck just to be safe in case not convex */
          alphaPrimal *= stepback;
          tco_set(t, t->e_setAlphaPrimal, &alphaPrimal);
          getMinF(t, &ineq1);
        }
        if (ineq <= 0 || ineq1 <= ineq / 10) {
          alphaPrimal = d->alphaMin / stepback; /* try min */
          tco_set(t, t->e_setAlphaPrimal, &alphaPrimal);
          getMinF(t, &ineq);
          if (ineq > 0) { /* try between min and max */
            for (alphaPrimal = alphaMax_ * .95; alphaPrimal >= d->alphaMin; alphaPrimal /= 2) {
              tco_set(t, t->e_setAlphaPrimal, &alphaPrimal);
              getMinF(t, &ineq);
              if (ineq > 0) {
                alphaPrimal *= stepback; /* backtrace just a little */
                tco_set(t, t->e_setAlphaPrimal, &alphaPrimal);
                getMinF(t, &ineq1);
                if (ineq1 > ineq / 10) break;
              }
            }
            if (alphaPrimal < d->alphaMin) {
              alphaPrimal = 0;
              tco_set(t, t->e_setAlphaPrimal, &alphaPrimal);
            }
          } else {
            alphaPrimal = 0;
            tco_set(t, t->e_setAlphaPrimal, &alphaPrimal);
          }
        }
      } else {
        alphaPrimal = 0;
        tco_set(t, t->e_setAlphaPrimal, &alphaPrimal);
      }
      if (d->coupledAlphas) { /* :758-767 */
        alphaDualEq = alphaPrimal;
        alphaDualIneq = alphaPrimal;
      } else {
        alphaDualIneq *= stepback;
        if (alphaDualIneq > alphaMax_) alphaDualIneq = alphaMax_;
        alphaDualEq = alphaDualIneq;
      }
      if (nG > 0) tco_set(t, t->e_setAlphaDualEq, &alphaDualEq);
      tco_set(t, t->e_setAlphaDualIneq, &alphaDualIneq);
      tco_copy(t, t->e_updatePrimalDual);

      /* mu update, skipAffine!=0 :782-847 */
      int th_grad = norminf_grad < MAX(1e-6, 1e0 * d->gradTolerance);
      int th_eq = nG > 0 ? norminf_eq < MAX(1e-5, 1e0 * d->equalTolerance) : 1;
      if (!d->skipAffine) {
        /* :848-850: the predictor step has already chosen mu */
      } else if (alphaPrimal > alphaMax_ / 2 && th_grad && th_eq) {
        mu *= MIN(d->muFactorAggressive, pow(mu, .5));
        if (mu < muMin) mu = muMin;
        tco_set(t, t->e_setMu, &mu);
      } else {
        if (alphaPrimal < .1) {
          mu *= 1.1;
          if (mu > mu0) mu = mu0;
          tco_set(t, t->e_setMu, &mu);
          tco_copy(t, t->e_initDualIneq);
        } else if (alphaPrimal > .99 && th_eq) {
          mu *= d->muFactorConservative;
          if (mu < muMin) mu = muMin;
          tco_set(t, t->e_setMu, &mu);
        }
      }
      /* if no motion, slowly increase mu :852-859 */
      if (alphaPrimal < d->alphaMin && alphaDualIneq < d->alphaMin && alphaDualEq < d->alphaMin) {
        mu /= (d->muFactorConservative * d->muFactorConservative);
        if (mu < muMin) mu = muMin;
        tco_set(t, t->e_setMu, &mu);
      }
    }
  }
  if ((*status) == 8) { /* :885-920 */
    get1(t, t->e_getNorminfGrad, &norminf_grad);
    if (norminf_grad > d->gradTolerance) (*status) |= 16;
    if (nG > 0) {
      get1(t, t->e_getNorminfG, &norminf_eq);
      if (norminf_eq > d->equalTolerance) (*status) |= 32;
    }
    if (nF > 0) {
      tco_get_flat(t, t->e_getGapMinFMinLambda, tmp);
      gap = tmp[0];
      if (gap > desiredDualityGapVar) (*status) |= 64;
      if (mu > muMin) (*status) |= 128;
      if (alphaPrimal <= d->alphaMin && alphaDualIneq < d->alphaMin && alphaDualEq < d->alphaMin)
        (*status) |= 1792;
      else if (alphaPrimal <= .1 && alphaDualIneq < .1 && alphaDualEq < .1)
        (*status) |= 1536;
      else if (alphaPrimal <= .5 && alphaDualIneq < .5 && alphaDualEq < .5)
        (*status) |= 1024;
    }
    if (adjust && addEye2HessianU > d->addEye2HessianUtolerance) (*status) |= 2048;
  }
}

/* batch helper for the CPU baseline / golden generation: instance k reads
 * in[e][k*size_e ..] for each listed set entry, solves, and writes the flat outputs of get
 * entry `eOut` to out[k*nOut ..] */
void tco_solve_batch(tco_t *t, int nSets, const int *setEntries, const double *const *in, int eOut,
                     double *out, int32_t *status, int32_t *iter, int64_t B, double mu0,
                     int32_t maxIter, const double *addEye2Hessian) {
  int64_t nOut = t->nA[eOut];
  for (int64_t k = 0; k < B; k++) {
    for (int s = 0; s < nSets; s++) tco_set(t, setEntries[s], in[s] + k * t->nA[setEntries[s]]);
    tco_solve(t, mu0, maxIter, -1, addEye2Hessian, &status[k], &iter[k]);
    tco_get_flat(t, eOut, out + k * nOut);
  }
}

/* ORACLE - test infrastructure only (see tc_oracle.c header). C API of the CPU restatement. */
#ifndef TC_ORACLE_H
#define TC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct tco tco_t;

/* solver #define block (reference: cmex2optimizeCS.m:303-331) as run-time values */
typedef struct {
  int nU, nG, nF;
  double gradTolerance, equalTolerance, desiredDualityGap, addEye2HessianUtolerance;
  int scaleCost, scaleInequalities;
  double alphaMin, alphaMax;
  int coupledAlphas, setAddEye2Hessian, adjustAddEye2Hessian, useInertia;
  double muFactorAggressive, muFactorConservative;
  int skipAffine;
  int smallerNewtonMatrix; /* ipmPD_CS.m only: the inequality duals are eliminated from the Newton system */
  int delta;               /* exponent of sigma in the Mehrotra mu update (ipmPD_CSsolver.c:650-654) */
} tco_defines_t;

tco_t *tco_load(const char *tapePath);
void tco_free(tco_t *t);
const tco_defines_t *tco_defines(const tco_t *t);
double tco_define(const tco_t *t, const char *name, double dflt);   /* any name of the tape's #define block */

int tco_num_entries(const tco_t *t);
int tco_entry(const tco_t *t, const char *name);       /* index or -1 */
const char *tco_entry_name(const tco_t *t, int e);
int tco_entry_kind(const tco_t *t, int e);              /* 0 set, 1 get, 2 copy */
int64_t tco_entry_size(const tco_t *t, int e);          /* doubles in (set) / out (get, flat) */

int tco_set(tco_t *t, int e, const double *in);         /* <cls>_set_<name>(const double*) */
int tco_get(tco_t *t, int e, double **outs);            /* <cls>_get...(double*y1,double*y2,...) */
int tco_get_flat(tco_t *t, int e, double *out);         /* all outputs concatenated */
int tco_copy(tco_t *t, int e);                          /* copy entry points, e.g. initPrimal__() */

/* ipmPD_CSsolver(mu0,maxIter,saveIter,addEye2Hessian,status,iter,time) */
void tco_solve(tco_t *t, double mu0, int32_t maxIter, int32_t saveIter,
               const double *addEye2Hessian, int32_t *status, int32_t *iter);
void tco_solve_batch(tco_t *t, int nSets, const int *setEntries, const double *const *in, int eOut,
                     double *out, int32_t *status, int32_t *iter, int64_t B, double mu0,
                     int32_t maxIter, const double *addEye2Hessian);

/* ipmPDeq_CSsolver(mu0,maxIter,saveIter,addEye2Hessian,status,iter,time): the equilibrium solvers' loop
 * (oracle/tc_oracle_eq.c, restating lib/ipmPDeq_CSsolver.c:122-802) */
void tco_solve_eq(tco_t *t, double mu0, int32_t maxIter, int32_t saveIter,
                  const double *addEye2Hessian, int32_t *status, int32_t *iter);
void tco_solve_eq_batch(tco_t *t, int nSets, const int *setEntries, const double *const *in, int eOut,
                        double *out, int32_t *status, int32_t *iter, int64_t B, double mu0,
                        int32_t maxIter, const double *addEye2Hessian);

/* introspection for tests */
double tco_value(const tco_t *t, int64_t instr);
int64_t tco_flops(const tco_t *t);
int64_t tco_group_evals(const tco_t *t);
int64_t tco_minF_trials(const tco_t *t);
void tco_reset_counters(tco_t *t);

#ifdef __cplusplus
}
#endif
#endif

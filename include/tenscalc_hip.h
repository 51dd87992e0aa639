/* C-ABI of a generated TensCalc-HIP solver class library (lib<classname>.so).
 *
 * One shared library per generated class, exactly as the reference emits one <classname>.so
 * per cmex2optimizeCS call (lib/cmex2optimizeCS.m:403-417).  Two faces:
 *
 * (1) the reference's own per-class C symbols, single instance (B=1), same names/signatures:
 *       void <cls>_set_<name>(const double *input)      lib/@csparse/writeCfunctionpergroup.m:524
 *       void ipmPD_CSsolver(double *mu0,int32_t *maxIter,int32_t *saveIter,double *addEye2Hessian,
 *                           int32_t *status,int32_t *iter,double *time)
 *                                                       lib/ipmPD_CSsolver.c:132-141
 *       void <cls>_getOutputs(double *y1,double *y2,...) writeCfunctionpergroup.m:607-635
 *     (declared in the generated <cls>.h; they operate on a process-global B=1 handle).
 *
 * (2) the batched, handle-based interface below: every set/get takes a trailing batch
 *     dimension B.  Host arrays are MATLAB column-major with the batch as the slowest
 *     dimension ([size...,B], instance k contiguous); device storage is structure-of-arrays
 *     [element][B] so the batch is the coalesced HBM axis.
 *
 * All functions return 0 on success or a negative TC_E* code; numerical failures are reported
 * per instance through the reference's `status` bits (lib/private/parameters4all.m:311-332).
 * Plain C types only; `stream` arguments are hipStream_t passed as void*.
 *
 * Devices and streams.  Every entry point makes the handle's device current on the calling thread.  Each
 * handle owns one stream; entry points without a `stream` argument (tc_set, tc_solve, tc_get_outputs,
 * tc_probe, tc_synchronize, ...) run on it.  tc_set_device / tc_set_from_output / tc_solve_async accept a
 * caller's stream (NULL = the handle's): work enqueued there is ordered AFTER everything the handle's stream
 * already holds and BEFORE anything enqueued later on the handle's stream (an event links the two), so
 * tc_solve_async(h,...,s) followed by tc_get_outputs(h,...) reads finished results, and a tc_set cannot
 * overwrite inputs a solve on another stream is still reading.  Two DIFFERENT caller streams used on one handle
 * are chained the same way, in call order: an entry point given stream B first joins what an earlier call left in
 * flight on stream A into the handle's stream, then orders B behind it.
 */
#ifndef TENSCALC_HIP_H
#define TENSCALC_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TC_OK 0
#define TC_EINVAL (-1)  /* bad handle / name / size */
#define TC_EHIP (-2)    /* HIP runtime error (see tc_last_error) */
#define TC_ENOMEM (-3)

#define TC_LAYOUT_INSTANCE_MAJOR 0 /* [B][n]: instance k contiguous (host layout) */
#define TC_LAYOUT_BATCH_MINOR 1    /* [n][B]: structure-of-arrays (device layout) */

typedef struct tc_solver tc_solver_t;

/* ---- class introspection (replaces the template struct array of cmex2optimizeCS.m:157-195) */
const char *tc_classname(void);
int tc_num_inputs(void);                 /* parameters followed by optimization variables */
const char *tc_input_name(int k);
int tc_input_is_variable(int k);         /* 0: setP_<name>, 1: setV_<name> */
int64_t tc_input_numel(int k);
int tc_input_ndim(int k);
int64_t tc_input_dim(int k, int d);
int tc_num_outputs(void);
const char *tc_output_name(int k);
int64_t tc_output_numel(int k);
int tc_output_ndim(int k);
int64_t tc_output_dim(int k, int d);
int64_t tc_define_int(const char *name);  /* nU, nG, nF, ... (cmex2optimizeCS.m:303-331) */
double tc_define_double(const char *name);
int64_t tc_flops_static(const char *what); /* static flop counts: "total_instructions", ... */
const char *tc_last_error(void);

/* ---- life cycle: obj=<cls>() / clear obj */
int tc_create(int64_t B, int device, tc_solver_t **out);
int tc_destroy(tc_solver_t *h);
int64_t tc_batch(const tc_solver_t *h);

/* ---- setP_<name>(obj,value) / setV_<name>(obj,value): <cls>_set_<name>(const double*) */
int tc_set(tc_solver_t *h, const char *name, const double *host, int64_t B);
/* same, from device memory in either layout, enqueued on `stream` (no host sync) */
int tc_set_device(tc_solver_t *h, const char *name, const double *dev, int64_t B, int layout,
                  void *stream);

/* warm start / receding horizon without leaving the device: input <- output of the previous solve
 * (e.g. setV_x(obj,out.xWarm) of examples/mpcmhe/mpc_dcmotor.m:224-225); sizes must match */
int tc_set_from_output(tc_solver_t *h, const char *inputName, const char *outputName, void *stream);

/* ---- [status,iter,time]=solve(obj,mu0,maxIter,saveIter,addEye2Hessian): ipmPD_CSsolver(...)
 * scalars by pointer as in the reference; status/iter are [B] host arrays; *time = seconds
 * for the whole batch (kernel time).  saveIter is accepted and ignored (allowSave=0). */
int tc_solve(tc_solver_t *h, const double *mu0, const int32_t *maxIter, const int32_t *saveIter,
             const double *addEye2Hessian, int32_t *status, int32_t *iter, double *time);
/* enqueue only: results stay on the device (use tc_get_device / tc_get_outputs afterwards) */
int tc_solve_async(tc_solver_t *h, double mu0, int32_t maxIter, const double *addEye2Hessian,
                   void *stream);

/* ---- [y1,y2,...]=getOutputs(obj): <cls>_getOutputs(double*y1,double*y2,...) */
int tc_get_outputs(tc_solver_t *h, double **outs, int64_t B);
/* device views (SoA): outputs[tc_output offsets][B], status[B], iter[B] */
int tc_get_device(tc_solver_t *h, const double **outputs, const int32_t **status,
                  const int32_t **iter);
int64_t tc_output_offset(int k);          /* row of output k inside the SoA outputs block */
int tc_synchronize(tc_solver_t *h);
/* the handle's own stream (a hipStream_t as void*): a caller that enqueues its work THERE -- e.g. records its timing
 * events on it and passes it to tc_solve_async -- needs no cross-stream event per call */
int tc_get_stream(tc_solver_t *h, void **stream);

/* ---- diagnostics: evaluate one tape `get` entry point (getGrad__, getJ__, ...) right after
 * the solver's initialisation sequence (initPrimal__, scaleIneq__, setMu__(mu0),
 * initDualIneq__, initDualEqX__) for every instance; out is [numel,B] instance-major on host */
int tc_probe(tc_solver_t *h, const char *getName, double mu0, const double *addEye2Hessian,
             double *out, int64_t B);
int64_t tc_probe_numel(const char *getName);

/* ---- diagnostics of builds made with stats=True (TC_STATS): rows [0,G) wall cycles spent in
 * dependency group g, [G,2G) evaluation counts, 2G total kernel cycles, 2G+1 getMinF_s__ trials,
 * [2G+2,2G+10) wall cycles per phase of the Newton loop (csrc/ipm_newton.hip.inc: TC_PH);
 * out is [tc_stats_rows()][B].  All zero for ordinary builds. */
int64_t tc_stats_rows(void);
int tc_get_stats(tc_solver_t *h, double *out);
/* the reference's profilingView(filename) (lib/@csparse/writeCprofiling.c:8-83, classes generated with profiling=true):
 * per dependency group executions and time of the last batched solve, averaged over its instances, written to
 * `filename` in the reference's table layout.  TC_EINVAL for classes built without stats=True. */
int tc_profiling_view(tc_solver_t *h, const char *filename);
/* the reference's per-iteration table of solverVerboseLevel >= 3 (lib/ipmPD_CSsolver.c:248-260 header, :305-345 exit
 * tests, :384 mu, :422-546 regularisation, :558-577 / :632-663 step lengths, :775-863 update and mu rule; the reference
 * prints it to the MATLAB console while it solves its ONE instance) for instance `instance` of the batch, written to
 * `filename`: the batch is solved once more with the arguments of the last tc_solve (results unchanged) while the
 * device loop records that instance's rows.  TC_EINVAL for classes generated with solverVerboseLevel < 3 (production
 * classes carry no recording code), before the first solve, or for an instance outside the batch. */
int tc_trace(tc_solver_t *h, int64_t instance, const char *filename);

#ifdef __cplusplus
}
#endif
#endif

/* A foreign host binding the class library the way INTEGRATION.md describes: plain C, dlopen + the
 * declarations of include/tenscalc_hip.h, no Python and no torch in the process.  Reads a binary input
 * file written by tests/test_cabi_from_c.py (B, then A[B][800], b[B][100], x0[B][8] as doubles), solves
 * the batch through tc_create / tc_set / tc_solve / tc_get_outputs, and writes status, iter, J, x to
 * the output file.  The reference-named single-instance symbols are exercised on instance 0.
 *   usage: cabi_harness <lib.so> <classname> <in.bin> <out.bin>                                        */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "tenscalc_hip.h"

#define SYM(var, name)                                     \
  *(void **)&var = dlsym(lib, name);                       \
  if (!var) { fprintf(stderr, "missing symbol %s\n", name); return 3; }

int main(int argc, char **argv) {
  if (argc != 5) return 2;
  void *lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
  int (*create)(int64_t, int, tc_solver_t **), (*destroy)(tc_solver_t *);
  int (*set)(tc_solver_t *, const char *, const double *, int64_t);
  int (*solve)(tc_solver_t *, const double *, const int32_t *, const int32_t *, const double *, int32_t *, int32_t *, double *);
  int (*get)(tc_solver_t *, double **, int64_t);
  const char *(*lasterr)(void), *(*cname)(void);
  int (*nin)(void), (*nout)(void);
  SYM(create, "tc_create") SYM(destroy, "tc_destroy") SYM(set, "tc_set") SYM(solve, "tc_solve")
  SYM(get, "tc_get_outputs") SYM(lasterr, "tc_last_error") SYM(cname, "tc_classname")
  SYM(nin, "tc_num_inputs") SYM(nout, "tc_num_outputs")
  if (strcmp(cname(), argv[2]) || nin() != 3 || nout() != 2) { fprintf(stderr, "introspection mismatch\n"); return 4; }

  FILE *f = fopen(argv[3], "rb");
  int64_t B;
  if (!f || fread(&B, sizeof B, 1, f) != 1) return 5;
  double *A = malloc(sizeof(double) * B * 800), *b = malloc(sizeof(double) * B * 100), *x0 = malloc(sizeof(double) * B * 8);
  if (fread(A, sizeof(double), B * 800, f) != (size_t)(B * 800) || fread(b, sizeof(double), B * 100, f) != (size_t)(B * 100) ||
      fread(x0, sizeof(double), B * 8, f) != (size_t)(B * 8)) return 5;
  fclose(f);

  tc_solver_t *h;
  if (create(B, 0, &h)) { fprintf(stderr, "tc_create: %s\n", lasterr()); return 6; }
  if (set(h, "A", A, B) || set(h, "b", b, B) || set(h, "x", x0, B)) { fprintf(stderr, "tc_set: %s\n", lasterr()); return 6; }
  /* error codes of the boundary (API-level failures as non-zero return codes, never a crash) */
  if (set(h, "nosuchinput", A, B) != TC_EINVAL) { fprintf(stderr, "bad name accepted\n"); return 7; }
  if (set(h, "A", A, B + 1) != TC_EINVAL) { fprintf(stderr, "batch mismatch accepted by tc_set\n"); return 7; }
  if (set(h, "A", NULL, B) != TC_EINVAL || set(NULL, "A", A, B) != TC_EINVAL) { fprintf(stderr, "null accepted\n"); return 7; }
  if (!lasterr()[0]) { fprintf(stderr, "tc_last_error empty after a failure\n"); return 7; }
  {
    tc_solver_t *bad = (tc_solver_t *)1;
    if (create(0, 0, &bad) != TC_EINVAL || create(-5, 0, &bad) != TC_EINVAL) { fprintf(stderr, "B<=0 accepted\n"); return 7; }
    if (create(4, 1000, &bad) != TC_EINVAL || bad != NULL) { fprintf(stderr, "device 1000 accepted\n"); return 7; }
    if (create(4, 0, NULL) != TC_EINVAL) { fprintf(stderr, "null out accepted\n"); return 7; }
    if (destroy(NULL) != TC_EINVAL) { fprintf(stderr, "destroy(NULL) accepted\n"); return 7; }
    double *o2[2] = {0, 0};
    int (*get_)(tc_solver_t *, double **, int64_t);
    *(void **)&get_ = dlsym(lib, "tc_get_outputs");
    if (get_(h, o2, B + 3) != TC_EINVAL || get_(h, NULL, B) != TC_EINVAL) { fprintf(stderr, "tc_get_outputs: bad arguments accepted\n"); return 7; }
    /* several handles at once, destroyed in any order (handle-based, no global state besides the B=1 compat one) */
    tc_solver_t *h2, *h3;
    if (create(3, 0, &h2) || create(5, 0, &h3) || destroy(h2) || destroy(h3)) { fprintf(stderr, "second/third handle: %s\n", lasterr()); return 7; }
  }
  double mu0 = 1.0, ae[2] = {1e-9, 1e-9}, tm = 0;
  int32_t maxIter = 30, saveIter = -1;
  int32_t *status = malloc(sizeof(int32_t) * B), *iter = malloc(sizeof(int32_t) * B);
  if (solve(h, &mu0, &maxIter, &saveIter, ae, status, iter, &tm)) { fprintf(stderr, "tc_solve: %s\n", lasterr()); return 6; }
  double *J = malloc(sizeof(double) * B), *x = malloc(sizeof(double) * B * 8);
  double *outs[2] = {J, x};
  if (get(h, outs, B)) { fprintf(stderr, "tc_get_outputs: %s\n", lasterr()); return 6; }
  {
    /* enqueue-only solve followed by tc_get_outputs without an explicit synchronisation (header: stream rule) */
    int (*solve_async)(tc_solver_t *, double, int32_t, const double *, void *);
    SYM(solve_async, "tc_solve_async")
    double *J2 = malloc(sizeof(double) * B), *x2 = malloc(sizeof(double) * B * 8);
    double *outs2[2] = {J2, x2};
    if (set(h, "x", x0, B) || solve_async(h, mu0, maxIter, ae, NULL) || get(h, outs2, B)) { fprintf(stderr, "async path: %s\n", lasterr()); return 6; }
    if (memcmp(J, J2, sizeof(double) * B) || memcmp(x, x2, sizeof(double) * B * 8)) { fprintf(stderr, "async solve + get differs from tc_solve\n"); return 9; }
    free(J2); free(x2);
  }
  destroy(h);

  /* the reference's own symbols, one instance (writeCfunctionpergroup.m:524,607-635; ipmPD_CSsolver.c:132-141) */
  char nm[256];
  void (*setA)(const double *), (*setb)(const double *), (*setx)(const double *), (*getO)(double *, double *);
  void (*ipm)(double *, int32_t *, int32_t *, double *, int32_t *, int32_t *, double *);
  snprintf(nm, sizeof nm, "%s_set_A", argv[2]); SYM(setA, nm)
  snprintf(nm, sizeof nm, "%s_set_b", argv[2]); SYM(setb, nm)
  snprintf(nm, sizeof nm, "%s_set_x", argv[2]); SYM(setx, nm)
  snprintf(nm, sizeof nm, "%s_getOutputs", argv[2]); SYM(getO, nm)
  SYM(ipm, "ipmPD_CSsolver")
  int32_t st1, it1;
  double J1, x1[8];
  setA(A); setb(b); setx(x0);
  ipm(&mu0, &maxIter, &saveIter, ae, &st1, &it1, &tm);
  getO(&J1, x1);
  if (st1 != status[0] || it1 != iter[0] || J1 != J[0] || memcmp(x1, x, sizeof x1)) { fprintf(stderr, "B=1 symbols disagree with the batch\n"); return 8; }

  f = fopen(argv[4], "wb");
  fwrite(status, sizeof(int32_t), B, f); fwrite(iter, sizeof(int32_t), B, f);
  fwrite(J, sizeof(double), B, f); fwrite(x, sizeof(double), B * 8, f);
  fclose(f);
  printf("cabi_harness ok: B=%lld kernel %.3f ms\n", (long long)B, tm * 1e3);
  return 0;
}

/* C-ABI of the native scalar-instruction table (code-generation side).
 *
 * Replaces the reference's only native library, lib/csparse/instructionsTableUTHash.c
 * (interface: lib/csparse/instructionsTableFunctions.h:29-105).  Same operations,
 * but handle-based (the reference keeps one global table) and backed by
 * std::unordered_map instead of uthash.  Instruction indices are 1-based as in the
 * reference (index == SSA value id); -1 signals an error.
 */
#ifndef TC_INSTRUCTIONS_TABLE_H
#define TC_INSTRUCTIONS_TABLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct tcit tcit_t;

/* initInstructionsTable()  (instructionsTableFunctions.h:29-32) */
tcit_t *tcit_create(void);
void tcit_destroy(tcit_t *t);

/* appendInstruction()  (instructionsTableFunctions.h:34-42) */
int64_t tcit_append(tcit_t *t, int32_t type, int64_t nParameters, const double *parameters,
                    int64_t nOperands, const int64_t *operands);

/* appendUniqueInstruction()  (instructionsTableFunctions.h:44-53): returns the index of an
 * identical instruction (type, parameters, operands all equal) when one exists. */
int64_t tcit_append_unique(tcit_t *t, int32_t type, int64_t nParameters, const double *parameters,
                           int64_t nOperands, const int64_t *operands);

/* n instructions of one type with identical operand/parameter counts in one call;
 * out[k] receives the index of the k-th.  unique!=0 -> dedup as tcit_append_unique. */
int tcit_append_many(tcit_t *t, int32_t type, int64_t n, int64_t nParameters,
                     const double *parameters, int64_t nOperands, const int64_t *operands,
                     int32_t unique, int64_t *out);

/* getInstruction()  (instructionsTableFunctions.h:55-64); pointers stay valid until the next append */
int tcit_get(tcit_t *t, int64_t index, int32_t *type, int64_t *nParameters,
             const double **parameters, int64_t *nOperands, const int64_t **operands);

/* instructionsTableHeight() */
int64_t tcit_height(tcit_t *t);
int64_t tcit_total_operands(tcit_t *t);
int64_t tcit_total_parameters(tcit_t *t);

/* getDependencies()  (instructionsTableFunctions.h:66-70): one (child,parent) pair per operand;
 * arrays must hold tcit_total_operands() entries. */
void tcit_dependencies(tcit_t *t, int64_t *children, int64_t *parents);

/* findInstructionsByType()  (instructionsTableFunctions.h:72-76) */
void tcit_find_by_type(tcit_t *t, int32_t type, int8_t *isOfType);

/* bulk export in CSR form: types[n], opStart[n+1], ops[total], parStart[n+1], pars[total] */
void tcit_export(tcit_t *t, int32_t *types, int64_t *opStart, int64_t *ops,
                 int64_t *parStart, double *pars);

#ifdef __cplusplus
}
#endif
#endif

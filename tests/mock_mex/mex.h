/* TEST DOUBLE of the part of MATLAB's mex API that integration/tc_mex.c uses -- so that OUR gateway can be compiled
 * and driven where MATLAB is not installed (tests/test_mex_gateway.py).  Not MathWorks' header and not part of the
 * product: an mxArray here is a plain struct, errors unwind by longjmp into the test driver (mex_driver.c) the way
 * mexErrMsgIdAndTxt unwinds into MATLAB. */
#ifndef MOCK_MEX_H
#define MOCK_MEX_H
#include <stddef.h>
#include <stdint.h>

typedef size_t mwSize;
typedef enum { mxCHAR_CLASS = 4, mxDOUBLE_CLASS = 6, mxINT32_CLASS = 12, mxUINT64_CLASS = 15 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef struct mxArray_tag {
  mxClassID cls;
  int ndim;
  mwSize dims[9];
  void *data;
} mxArray;

int mxGetString(const mxArray *a, char *buf, mwSize buflen);
int mxIsChar(const mxArray *a);
int mxIsNumeric(const mxArray *a);
int mxIsDouble(const mxArray *a);
int mxIsInt32(const mxArray *a);
int mxIsUint64(const mxArray *a);
int mxIsComplex(const mxArray *a);
int mxIsSparse(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
double mxGetScalar(const mxArray *a);
void *mxGetData(const mxArray *a);
double *mxGetPr(const mxArray *a);
void *mxCalloc(size_t n, size_t size);
void mxFree(void *p);
void mexMakeMemoryPersistent(void *p);
mxArray *mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity c);
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity c);
mxArray *mxCreateDoubleScalar(double v);
mxArray *mxCreateString(const char *s);
void mxDestroyArray(mxArray *a);
int mexAtExit(void (*fn)(void));
void mexErrMsgTxt(const char *msg);
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#endif

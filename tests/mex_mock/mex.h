/* mex.h -- declarations of the documented MATLAB C Matrix / MEX API calls the gateways under
 * integration/ use (R2018a+ signatures; mwSize = size_t, mwIndex = size_t).  No MATLAB exists in the
 * build image: tests/test_integration.py compiles the gateways against THESE declarations, and
 * tests/mex_mock/mex_mock.c implements them over a small array type so that tests/test_gpu_mex.py can
 * run the gateways on the GPU box and compare what they return with the Python surface.  Test
 * infrastructure for this repository's own gateways; nothing of the reference is built with it. */
#ifndef TEST_MEX_H
#define TEST_MEX_H
#include <stddef.h>
#include <stdbool.h>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef bool mxLogical;
typedef enum { mxREAL, mxCOMPLEX } mxComplexity;
typedef enum { mxUNKNOWN_CLASS, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS,
               mxDOUBLE_CLASS, mxSINGLE_CLASS, mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS,
               mxINT32_CLASS, mxUINT32_CLASS, mxINT64_CLASS, mxUINT64_CLASS, mxFUNCTION_CLASS } mxClassID;
bool mxIsUint32(const mxArray *pa);
void *mxGetData(const mxArray *pa);
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID classid, mxComplexity flag);
bool mxIsStruct(const mxArray *pa);
bool mxIsCell(const mxArray *pa);
bool mxIsDouble(const mxArray *pa);
bool mxIsLogical(const mxArray *pa);
bool mxIsComplex(const mxArray *pa);
mxArray *mxGetField(const mxArray *pa, mwIndex index, const char *fieldname);
mxArray *mxGetCell(const mxArray *pa, mwIndex index);
double *mxGetPr(const mxArray *pa);
mxLogical *mxGetLogicals(const mxArray *pa);
double mxGetScalar(const mxArray *pa);
double mxGetNaN(void);
size_t mxGetNumberOfElements(const mxArray *pa);
size_t mxGetM(const mxArray *pa);
size_t mxGetN(const mxArray *pa);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **fieldnames);
void mxSetFieldByNumber(mxArray *pa, mwIndex index, int fieldnumber, mxArray *value);
void mxDestroyArray(mxArray *pa);
void *mxMalloc(size_t n);
void mxFree(void *ptr);
void mexErrMsgIdAndTxt(const char *identifier, const char *err_msg, ...);
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#endif

/* mex_mock.c -- a small implementation of the calls tests/mex_mock/mex.h declares, so that the MEX
 * gateways of integration/ can be RUN without MATLAB (tests/test_gpu_mex.py): column-major arrays,
 * cells, 1 x 1 structs.  mexErrMsgIdAndTxt does not return: it records the message and jumps back
 * into mock_call().  Test infrastructure only. */
#include <math.h>
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mex.h"

enum { K_DOUBLE, K_LOGICAL, K_UINT32, K_CELL, K_STRUCT };
struct mxArray_tag {
  int kind;
  size_t ndim, dims[3];
  void *data;            /* doubles / bools / uint32 / mxArray* (cell) */
  int nfields;           /* struct */
  char **names;
  mxArray **fields;
};

static jmp_buf g_jump;
static char g_message[1024];

static size_t numel(const mxArray *a) { return a->dims[0] * a->dims[1] * (a->ndim > 2 ? a->dims[2] : 1); }

static mxArray *make(int kind, size_t ndim, const size_t *dims, size_t elem) {
  mxArray *a = (mxArray *)calloc(1, sizeof *a);
  size_t i;
  a->kind = kind;
  a->ndim = ndim;
  for (i = 0; i < 3; ++i) a->dims[i] = i < ndim ? dims[i] : 1;
  a->data = calloc(numel(a) ? numel(a) : 1, elem);
  return a;
}

/* ---- the API ---- */
bool mxIsStruct(const mxArray *a) { return a && a->kind == K_STRUCT; }
bool mxIsCell(const mxArray *a) { return a && a->kind == K_CELL; }
bool mxIsDouble(const mxArray *a) { return a && a->kind == K_DOUBLE; }
bool mxIsLogical(const mxArray *a) { return a && a->kind == K_LOGICAL; }
bool mxIsUint32(const mxArray *a) { return a && a->kind == K_UINT32; }
bool mxIsComplex(const mxArray *a) { (void)a; return false; }
mxArray *mxGetField(const mxArray *a, mwIndex index, const char *name) {
  int f;
  if (!a || a->kind != K_STRUCT || index != 0) return NULL;
  for (f = 0; f < a->nfields; ++f)
    if (strcmp(a->names[f], name) == 0) return a->fields[f];
  return NULL;
}
mxArray *mxGetCell(const mxArray *a, mwIndex index) { return ((mxArray **)a->data)[index]; }
double *mxGetPr(const mxArray *a) { return (double *)a->data; }
mxLogical *mxGetLogicals(const mxArray *a) { return (mxLogical *)a->data; }
void *mxGetData(const mxArray *a) { return a->data; }
double mxGetScalar(const mxArray *a) {
  return a->kind == K_DOUBLE ? ((double *)a->data)[0] : a->kind == K_LOGICAL ? (double)((mxLogical *)a->data)[0]
                                                                             : (double)((uint32_t *)a->data)[0];
}
double mxGetNaN(void) { return NAN; }
size_t mxGetNumberOfElements(const mxArray *a) { return numel(a); }
size_t mxGetM(const mxArray *a) { return a->dims[0]; }
size_t mxGetN(const mxArray *a) { return a->dims[1] * (a->ndim > 2 ? a->dims[2] : 1); }
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag) {
  size_t dims[2];
  (void)flag;
  dims[0] = m;
  dims[1] = n;
  return make(K_DOUBLE, 2, dims, sizeof(double));
}
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID classid, mxComplexity flag) {
  (void)flag;
  if (classid == mxDOUBLE_CLASS) return make(K_DOUBLE, ndim, dims, sizeof(double));
  if (classid == mxUINT32_CLASS) return make(K_UINT32, ndim, dims, sizeof(uint32_t));
  mexErrMsgIdAndTxt("mock:class", "mxCreateNumericArray: class %d not in the mock", (int)classid);
  return NULL;
}
mxArray *mxCreateStructMatrix(mwSize m, mwSize n, int nfields, const char **fieldnames) {
  size_t dims[2];
  mxArray *a;
  int f;
  dims[0] = m;
  dims[1] = n;
  a = make(K_STRUCT, 2, dims, 1);
  a->nfields = nfields;
  a->names = (char **)calloc((size_t)nfields, sizeof(char *));
  a->fields = (mxArray **)calloc((size_t)nfields, sizeof(mxArray *));
  for (f = 0; f < nfields; ++f) {
    a->names[f] = (char *)malloc(strlen(fieldnames[f]) + 1);
    strcpy(a->names[f], fieldnames[f]);
  }
  return a;
}
void mxSetFieldByNumber(mxArray *a, mwIndex index, int fieldnumber, mxArray *value) {
  (void)index;
  a->fields[fieldnumber] = value;
}
void mxDestroyArray(mxArray *a) {
  size_t i;
  int f;
  if (!a) return;
  if (a->kind == K_CELL)
    for (i = 0; i < numel(a); ++i) mxDestroyArray(((mxArray **)a->data)[i]);
  if (a->kind == K_STRUCT) {
    for (f = 0; f < a->nfields; ++f) {
      mxDestroyArray(a->fields[f]);
      free(a->names[f]);
    }
    free(a->names);
    free(a->fields);
  }
  free(a->data);
  free(a);
}
void *mxMalloc(size_t n) { return malloc(n ? n : 1); }
void mxFree(void *p) { free(p); }
void mexErrMsgIdAndTxt(const char *identifier, const char *err_msg, ...) {
  va_list ap;
  size_t at;
  snprintf(g_message, sizeof g_message, "%s: ", identifier);
  at = strlen(g_message);
  va_start(ap, err_msg);
  vsnprintf(g_message + at, sizeof g_message - at, err_msg, ap);
  va_end(ap);
  longjmp(g_jump, 1);
}

/* ---- what the Python test builds its arguments with ---- */
mxArray *mock_doubles(const double *src, size_t d0, size_t d1, size_t d2) {
  size_t dims[3];
  mxArray *a;
  dims[0] = d0;
  dims[1] = d1;
  dims[2] = d2;
  a = make(K_DOUBLE, d2 > 1 ? 3 : 2, dims, sizeof(double));
  memcpy(a->data, src, numel(a) * sizeof(double));
  return a;
}
mxArray *mock_logicals(const uint8_t *src, size_t d0, size_t d1) {
  size_t dims[2], i;
  mxArray *a;
  dims[0] = d0;
  dims[1] = d1;
  a = make(K_LOGICAL, 2, dims, sizeof(mxLogical));
  for (i = 0; i < numel(a); ++i) ((mxLogical *)a->data)[i] = src[i] != 0;
  return a;
}
mxArray *mock_uint32s(const uint32_t *src, size_t d0, size_t d1, size_t d2) {
  size_t dims[3];
  mxArray *a;
  dims[0] = d0;
  dims[1] = d1;
  dims[2] = d2;
  a = make(K_UINT32, d2 > 1 ? 3 : 2, dims, sizeof(uint32_t));
  memcpy(a->data, src, numel(a) * sizeof(uint32_t));
  return a;
}
mxArray *mock_cell(size_t n) {
  size_t dims[2];
  dims[0] = n;
  dims[1] = 1;
  return make(K_CELL, 2, dims, sizeof(mxArray *));
}
void mock_set_cell(mxArray *cell, size_t index, mxArray *value) { ((mxArray **)cell->data)[index] = value; }
mxArray *mock_struct(int nfields, const char **names) { return mxCreateStructMatrix(1, 1, nfields, names); }
void mock_set_field(mxArray *s, int field, mxArray *value) { mxSetFieldByNumber(s, 0, field, value); }
int mock_ndim(const mxArray *a) { return (int)a->ndim; }
size_t mock_dim(const mxArray *a, int i) { return a->dims[i]; }
int mock_kind(const mxArray *a) { return a->kind; }
const char *mock_message(void) { return g_message; }

/* runs the gateway; 0 and *out = plhs[0], or 1 with mock_message() set when it called mexErrMsgIdAndTxt */
int mock_call(int nrhs, const mxArray **prhs, mxArray **out) {
  mxArray *plhs[1] = {NULL};
  g_message[0] = 0;
  if (setjmp(g_jump)) return 1;
  mexFunction(1, plhs, nrhs, prhs);
  *out = plhs[0];
  return 0;
}

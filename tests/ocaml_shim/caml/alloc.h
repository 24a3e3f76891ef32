#ifndef SHIM_CAML_ALLOC_H
#define SHIM_CAML_ALLOC_H
#include "mlvalues.h"
value caml_alloc(mlsize_t, int);
value caml_alloc_tuple(mlsize_t);
value caml_alloc_small(mlsize_t, int);
value caml_copy_double(double);
value caml_copy_string(const char *);
value caml_copy_int64(int64_t);
#endif

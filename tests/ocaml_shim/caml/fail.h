#ifndef SHIM_CAML_FAIL_H
#define SHIM_CAML_FAIL_H
#include "mlvalues.h"
void caml_failwith(const char *) __attribute__((noreturn));
void caml_invalid_argument(const char *) __attribute__((noreturn));
void caml_raise_out_of_memory(void) __attribute__((noreturn));
#endif

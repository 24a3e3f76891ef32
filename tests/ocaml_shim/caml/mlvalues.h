/* Syntax-check stand-in for the OCaml runtime's C interface (tests/test_host_logic.py compiles ocaml/soundml_amd_stubs.c
 * with `gcc -fsyntax-only` against these declarations: there is no OCaml toolchain in the image).  Names and shapes follow
 * the OCaml manual, chapter "Interfacing C with OCaml"; nothing here is ever linked or run. */
#ifndef SHIM_CAML_MLVALUES_H
#define SHIM_CAML_MLVALUES_H
#include <stdint.h>
#include <stddef.h>
typedef intptr_t value;
typedef uintptr_t mlsize_t;
typedef intptr_t intnat;
typedef uintptr_t uintnat;
#define CAMLprim
#define CAMLextern extern
#define Val_long(x) ((value)(((uintnat)(x) << 1) + 1))
#define Long_val(v) ((intnat)(v) >> 1)
#define Val_int(x) Val_long(x)
#define Int_val(v) ((int)Long_val(v))
#define Val_unit Val_long(0)
#define Val_bool(x) Val_long((x) != 0)
#define Bool_val(v) Int_val(v)
#define Val_true Val_long(1)
#define Val_false Val_long(0)
#define Is_block(v) (((v) & 1) == 0)
#define Is_long(v) (((v) & 1) != 0)
#define Field(v, i) (((value *)(v))[i])
double caml_Double_val(value);
#define Double_val(v) caml_Double_val(v)
#define Data_custom_val(v) ((void *)&Field((v), 1))
#define Wosize_val(v) ((mlsize_t)0)
#endif

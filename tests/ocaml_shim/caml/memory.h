#ifndef SHIM_CAML_MEMORY_H
#define SHIM_CAML_MEMORY_H
#include "mlvalues.h"
#define CAMLparam0() int caml__frame = 0
#define CAMLparam1(a) CAMLparam0(); (void)(a)
#define CAMLparam2(a, b) CAMLparam0(); (void)(a); (void)(b)
#define CAMLparam3(a, b, c) CAMLparam0(); (void)(a); (void)(b); (void)(c)
#define CAMLparam4(a, b, c, d) CAMLparam0(); (void)(a); (void)(b); (void)(c); (void)(d)
#define CAMLparam5(a, b, c, d, e) CAMLparam0(); (void)(a); (void)(b); (void)(c); (void)(d); (void)(e)
#define CAMLxparam1(a) (void)(a)
#define CAMLxparam2(a, b) (void)(a); (void)(b)
#define CAMLxparam3(a, b, c) (void)(a); (void)(b); (void)(c)
#define CAMLxparam4(a, b, c, d) (void)(a); (void)(b); (void)(c); (void)(d)
#define CAMLxparam5(a, b, c, d, e) (void)(a); (void)(b); (void)(c); (void)(d); (void)(e)
#define CAMLlocal1(a) value a = Val_unit
#define CAMLlocal2(a, b) value a = Val_unit, b = Val_unit
#define CAMLlocal3(a, b, c) value a = Val_unit, b = Val_unit, c = Val_unit
#define CAMLreturn(v) do { (void)caml__frame; return (v); } while (0)
#define CAMLreturn0 do { (void)caml__frame; return; } while (0)
void caml_modify(value *, value);
void caml_initialize(value *, value);
#define Store_field(block, i, v) caml_modify(&Field((block), (i)), (v))
#endif

#ifndef SHIM_CAML_BIGARRAY_H
#define SHIM_CAML_BIGARRAY_H
#include "mlvalues.h"
enum caml_ba_kind {
  CAML_BA_FLOAT32, CAML_BA_FLOAT64, CAML_BA_SINT8, CAML_BA_UINT8, CAML_BA_SINT16, CAML_BA_UINT16, CAML_BA_INT32,
  CAML_BA_INT64, CAML_BA_CAML_INT, CAML_BA_NATIVE_INT, CAML_BA_COMPLEX32, CAML_BA_COMPLEX64, CAML_BA_CHAR,
  CAML_BA_FLOAT16, CAML_BA_KIND_MASK = 0xFF
};
enum caml_ba_layout { CAML_BA_C_LAYOUT = 0, CAML_BA_FORTRAN_LAYOUT = 0x100, CAML_BA_LAYOUT_MASK = 0x100 };
enum caml_ba_managed { CAML_BA_EXTERNAL = 0, CAML_BA_MANAGED = 0x200, CAML_BA_MAPPED_FILE = 0x400, CAML_BA_MANAGED_MASK = 0x600 };
struct caml_ba_array {
  void *data;
  intnat num_dims;
  intnat flags;
  void *proxy;
  intnat dim[1];
};
#define Caml_ba_array_val(v) ((struct caml_ba_array *)Data_custom_val(v))
#define Caml_ba_data_val(v) (Caml_ba_array_val(v)->data)
value caml_ba_alloc_dims(int flags, int num_dims, void *data, ...);
#endif

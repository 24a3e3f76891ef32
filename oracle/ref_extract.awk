# Cuts `soundml_cx`, `soundml_cx_mul` and `soundml_resample_shape_run` out of the reference's resample_stubs.c (the caml-free
# shaping arithmetic, :301-372) for oracle/_ref/.  Test infrastructure: see oracle/Makefile.
/^typedef struct \{/ && !done_t { in_t = 1 }
in_t { print; if ($0 ~ /^\} soundml_cx;/) { in_t = 0; done_t = 1 } ; next }
/^static inline soundml_cx soundml_cx_mul\(/ { in_m = 1 }
in_m { print; if ($0 ~ /^\}/) in_m = 0; next }
/^static void soundml_resample_shape_run\(/ { in_f = 1 }
in_f { print; if ($0 ~ /^\}/) in_f = 0; next }

#ifndef SHIM_CAML_THREADS_H
#define SHIM_CAML_THREADS_H
void caml_release_runtime_system(void);
void caml_acquire_runtime_system(void);
#endif

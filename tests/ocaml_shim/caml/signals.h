#ifndef SHIM_CAML_SIGNALS_H
#define SHIM_CAML_SIGNALS_H
void caml_enter_blocking_section(void);
void caml_leave_blocking_section(void);
#endif

/* Test infrastructure (oracle/_ref): one exported entry point around the reference's own `soundml_resample_shape_run`, cut out of
 * /root/reference/soundml/lib/resample_stubs.c by oracle/ref_extract.awk at build time (the extract lives in oracle/_ref/, which is
 * git-ignored; nothing of the reference is committed).  The geometry checks are those of the reference's CAMLprim wrapper
 * (resample_stubs.c:385-400): exactly one of sl / sm may exceed 1, n even and divisible by sm. */
#include <stdint.h>
#include "resample_shape_extract.h"

int ref_resample_shape(const void *x, const void *h, void *y, int64_t lines, int64_t n, int64_t sl, int64_t sm) {
  if (lines < 0 || n < 2 || (n & 1) || sl < 1 || sm < 1 || (sl > 1 && sm > 1) || n % sm != 0) return 1;
  const int64_t w = sl > 1 ? n * sl : n / sm;
  soundml_resample_shape_run((const soundml_cx *)x, (const soundml_cx *)h, (soundml_cx *)y, lines, n, sl, sm, w);
  return 0;
}

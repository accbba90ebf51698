/* Error handling of libkseps: every function returns a PetscErrorCode; the message
 * of the innermost failure plus the call trace is kept per thread (the role of
 * PETSc's error handler stack, reached through SETERRQ/CHKERRQ in the reference). */
#include <stdarg.h>
#include <stdio.h>

#include "ksimpl.h"

static _Thread_local char ks_err[1024] = "";

PetscErrorCode KSError(const char *func, int line, PetscErrorCode code, const char *fmt, ...) {
  va_list ap;
  char msg[768];
  va_start(ap, fmt);
  vsnprintf(msg, sizeof(msg), fmt, ap);
  va_end(ap);
  snprintf(ks_err, sizeof(ks_err), "[%d] %s() line %d: %s", code, func, line, msg);
  return code ? code : PETSC_ERR_PLIB;
}

void KSErrorTrace(const char *func, int line) {
  size_t len = strlen(ks_err);
  if (len + 64 < sizeof(ks_err)) snprintf(ks_err + len, sizeof(ks_err) - len, " <- %s:%d", func, line);
}

const char *KSGetLastError(void) { return ks_err; }

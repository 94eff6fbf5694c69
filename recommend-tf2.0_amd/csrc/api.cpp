// Version + thread-local error string of the C ABI (include/recamd.h).
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace rec {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace rec

extern "C" int rec_version(void) { return REC_VERSION; }

extern "C" int rec_last_error(char* buf, int n) {
  int len = (int)strlen(rec::g_err);
  if (buf && n > 0) {
    int c = len < n - 1 ? len : n - 1;
    memcpy(buf, rec::g_err, c);
    buf[c] = 0;
  }
  return len;
}

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

namespace rec {
namespace {
struct Forced {
  char key[24], val[24];
};
Forced g_forced[16];
int g_nforced = 0;
}  // namespace

const char* forced(const char* key) {
  for (int i = 0; i < g_nforced; ++i)
    if (!strcmp(g_forced[i].key, key)) return g_forced[i].val[0] ? g_forced[i].val : nullptr;
  return nullptr;
}
}  // namespace rec

extern "C" int rec_debug_force(const char* key, const char* value) {
  using namespace rec;
  REC_CHECK_ARG(key && strlen(key) < sizeof(g_forced[0].key), REC_EINVAL, "rec_debug_force: bad key");
  REC_CHECK_ARG(!value || strlen(value) < sizeof(g_forced[0].val), REC_EINVAL, "rec_debug_force: value too long");
  int i = 0;
  while (i < g_nforced && strcmp(g_forced[i].key, key)) ++i;
  if (i == g_nforced) {
    REC_CHECK_ARG(g_nforced < 16, REC_EINVAL, "rec_debug_force: table full");
    strcpy(g_forced[g_nforced++].key, key);
  }
  strcpy(g_forced[i].val, value ? value : "");
  return REC_OK;
}

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

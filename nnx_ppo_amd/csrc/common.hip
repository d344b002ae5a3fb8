// Error plumbing + library identity for libmippo.
#include "common.h"

#include <stdarg.h>
#include <stdio.h>

namespace mippo {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return -EIO;
  }
  return 0;
}

}  // namespace mippo

extern "C" int mi_abi_version(void) { return 1; }

extern "C" const char* mi_last_error(void) { return mippo::g_err; }

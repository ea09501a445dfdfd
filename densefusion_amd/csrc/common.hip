#include "common.h"

namespace df {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(DF_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return DF_OK;
}

}  // namespace df

extern "C" const char *df_last_error(void) { return df::g_err; }
extern "C" int df_version(void) { return 1; }

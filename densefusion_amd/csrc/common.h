// Shared helpers for libdfusion_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <cstdint>
#include <cstdio>

#include "../../include/dfusion.h"

namespace df {

constexpr int DF_MAX_CROP = 3200;      // largest crop side the network entry points take (LDS tables of psp_prior_sum, layers.hip)

int set_error(int code, const char *fmt, ...);   // records a thread-local message, returns code
int check_launch(const char *what);              // hipGetLastError() -> DF_ERR_LAUNCH

// Development switches (A/B runs of kernel variants, verbose profiles) exist only in the DF_DEV build (libdfusion_hip_dev.so, compiled
// with -DDF_DEV and loaded by tests that compare variants): the shipped library reads no environment.
#ifdef DF_DEV
inline const char *dev_getenv(const char *name) { return getenv(name); }
#else
inline const char *dev_getenv(const char *) { return nullptr; }
#endif

inline hipStream_t to_stream(df_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace df

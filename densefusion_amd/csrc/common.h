// Shared helpers for libdfusion_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/dfusion.h"

namespace df {

constexpr int DF_MAX_CROP = 3200;      // largest crop side the network entry points take (LDS tables of psp_prior_sum, layers.hip)

int set_error(int code, const char *fmt, ...);   // records a thread-local message, returns code
int check_launch(const char *what);              // hipGetLastError() -> DF_ERR_LAUNCH

inline hipStream_t to_stream(df_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace df

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

// Shader clock actually delivered under a vector-ALU load: every wave spins on dependent FMAs between two readings of the shader
// cycle counter (s_memtime) and of the constant 100 MHz counter (s_memrealtime); MHz = 100 * cycles / ticks.
namespace df {
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long *out, int iters, float seed) {
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  float a = seed + threadIdx.x, b = 1.0001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) a = a * b + 0.5f;
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[(size_t)blockIdx.x * 2] = c1 - c0;
    out[(size_t)blockIdx.x * 2 + 1] = w1 - w0;
  }
  if (a == 12345.678f) out[0] = 0;      // keep the chain alive
}
}  // namespace df

extern "C" int df_shader_clock_mhz(double *mhz_out, df_stream_t stream) {
  if (!mhz_out) return df::set_error(DF_ERR_ARG, "shader_clock_mhz: null pointer");
  const int blocks = 1024;                  // one round of the chip
  unsigned long long *d = nullptr;
  if (hipMalloc(&d, (size_t)blocks * 2 * sizeof(unsigned long long)) != hipSuccess) return df::set_error(DF_ERR_LAUNCH, "shader_clock_mhz: hipMalloc failed");
  hipStream_t st = df::to_stream(stream);
  hipLaunchKernelGGL(df::clock_probe_kernel, dim3(blocks), dim3(256), 0, st, d, 20000, 1.0f);
  static unsigned long long h[2048];
  hipError_t e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  hipFree(d);
  if (e != hipSuccess) return df::set_error(DF_ERR_LAUNCH, "shader_clock_mhz: %s", hipGetErrorString(e));
  double cyc = 0, ticks = 0;
  for (int i = 0; i < blocks; ++i) { cyc += (double)h[2 * i]; ticks += (double)h[2 * i + 1]; }
  *mhz_out = ticks > 0 ? 100.0 * cyc / ticks : 0.0;
  return DF_OK;
}

extern "C" const char *df_last_error(void) { return df::g_err; }
extern "C" int df_version(void) { return 1; }

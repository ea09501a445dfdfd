// Dev tool: sustained v_mfma_f32_32x32x2_f32 rate of this card (no memory traffic), for ~milliseconds-long launches --
// the practical fp32-MFMA ceiling the GEMM kernels are compared with.   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float a0, float b0) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float *out;
  const int blocks = 256 * 8;              // 8 workgroups of 4 waves per CU
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int iters : {2000, 20000, 100000}) {
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, 100, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 /*waves*/ * iters * 32.0 /*mfma*/ * (32 * 32 * 2 * 2);
    printf("iters %6d: %8.3f ms  %.1f TFLOP/s\n", iters, ms, flop / ms / 1e9);
  }
  return 0;
}

// Dev tool: how fast does a wave that is NOT issuing MFMAs advance while 0..3 other waves of its SIMD issue v_mfma_f32_32x32x2_f32
// back to back?  One 1024-thread workgroup per CU (16 waves, 4 per SIMD): per SIMD, `nhog` waves run an MFMA loop, one "victim" wave
// runs a fixed piece of work of one kind and times it (the GEMM's prologue / epilogue next to other workgroups' main loops).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_coissue tools/dev/mfma_coissue.hip && /tmp/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int GAP>
__global__ __launch_bounds__(1024) void co(int nhog, int kind, int hog_iters, float *gbuf, unsigned long long *out, float a0, int victim_first, int vprio) {
  extern __shared__ float lds[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned simd = (__builtin_amdgcn_s_getreg((2 << 11) | (4 << 6) | 4)) & 3;      // HW_ID bits [5:4]: SIMD id
  __shared__ int rank[4];
  if (threadIdx.x < 4) rank[threadIdx.x] = 0;
  __syncthreads();
  int j = 0;
  if (lane == 0) j = atomicAdd(&rank[simd], 1);              // j-th wave of its SIMD
  j = __builtin_amdgcn_readfirstlane(j);
  __syncthreads();
  const unsigned long long tstart = __builtin_readcyclecounter();
  const int vrank = victim_first ? 0 : 3;
  const bool hog = victim_first ? (j >= 1 && j <= nhog) : (j < nhog);
  if (hog) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + lane, b = 2.f;
    for (int it = 0; it < hog_iters; ++it)
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
          // what the hog does between two MFMAs: nothing (the next MFMA waits at the head of its instruction stream for the pipe),
          // s_nop, or a short sleep
          if (GAP == 1) asm volatile("s_nop 15");
          if (GAP == 2) asm volatile("s_sleep 1");
          if (GAP == 3) asm volatile("s_nop 15\n s_nop 15\n s_nop 15");
          if (GAP == 4) asm volatile("s_nop 0");
          if (GAP == 5) asm volatile("s_nop 3");
        }
    const unsigned long long th = __builtin_readcyclecounter();
    if (lane == 0) atomicMax(&out[1024 + blockIdx.x * 4 + simd], th - tstart);      // the LAST MFMA wave of the SIMD to finish
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
      for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.f) gbuf[threadIdx.x] = s;
    return;
  }
  if (j != vrank) return;
  // victim: let the hogs get going first
  __builtin_amdgcn_s_sleep(100);
  if (vprio) __builtin_amdgcn_s_setprio(3);
  const unsigned long long t0 = __builtin_readcyclecounter();
  float x = a0 + lane, y = 1.0001f;
  if (kind == 0) {                       // 512 DEPENDENT vector adds
    for (int it = 0; it < 32; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
  } else if (kind == 1) {                // 512 vector adds in 8 independent chains
    float c[8];
    for (int u = 0; u < 8; ++u) c[u] = x + u;
    for (int it = 0; it < 64; ++it)
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("v_add_f32 %0, %0, %1" : "+v"(c[u]) : "v"(y));
    for (int u = 0; u < 8; ++u) x += c[u];
  } else if (kind == 2) {                // 256 LDS writes, then 64 LDS reads each waited for
    for (int it = 0; it < 256; ++it) lds[(w * 64 + lane) + 1024 * (it & 7)] = x + it;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int it = 0; it < 64; ++it) { x += lds[(w * 64 + lane) + 1024 * (it & 7)]; asm volatile("" : "+v"(x)); }
  } else if (kind == 3) {                // 64 16-byte global stores (not waited for)
    float4 v = make_float4(x, y, x, y);
    for (int it = 0; it < 64; ++it) reinterpret_cast<float4 *>(gbuf)[((size_t)blockIdx.x * 64 + it) * 1024 + threadIdx.x] = v;
  } else if (kind == 4) {                // 512 scalar adds
    int sx = hog_iters;
    for (int it = 0; it < 32; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("s_add_i32 %0, %0, 3" : "+s"(sx));
    if (sx == 77) x += 1.f;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 4 + simd] = t1 - t0;
  if (x == 12345.f) gbuf[threadIdx.x] = x;
}

template <int GAP>
void run(float *gbuf, unsigned long long *out, int victim_first, int vprio) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(&co<GAP>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  const char *names[] = {"512 dependent v_add", "512 v_add, 8 chains", "256 ds_write + 64 waited ds_read", "64 global 16-B stores", "512 s_add"};
  const char *gaps[] = {"back to back", "s_nop 15 after each", "s_sleep 1 after each", "3 x s_nop 15 after each", "s_nop 0 after each", "s_nop 3 after each"};
  printf("--- MFMA waves issue their MFMAs: %s; the timed wave is the %s of its SIMD%s\n", gaps[GAP], victim_first ? "OLDEST" : "youngest",
         vprio ? ", at s_setprio 3" : "");
  for (int kind = 0; kind < 5; ++kind) {
    printf("%-34s", names[kind]);
    for (int nhog = 0; nhog <= 3; ++nhog) {
      hipMemset(out, 0, 2048 * 8);
      hipLaunchKernelGGL(co<GAP>, dim3(256), dim3(1024), 100 * 1024, 0, nhog, kind, 400, gbuf, out, 1.f, victim_first, vprio);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(2048);
      hipMemcpy(h.data(), out, 2048 * 8, hipMemcpyDeviceToHost);
      double s = 0, sh = 0; int n = 0, nh = 0;
      for (int i = 0; i < 1024; ++i) if (h[i]) { s += h[i]; ++n; }
      for (int i = 1024; i < 2048; ++i) if (h[i]) { sh += h[i]; ++nh; }
      printf("  %d: timed %7.0f pipe %.2f", nhog, n ? s / n : 0.0, nh ? nhog * 819200.0 / (sh / nh) : 0.0);
    }
    printf("\n");
  }
}
int main() {
  float *gbuf; unsigned long long *out;
  hipMalloc(&gbuf, (size_t)256 * 64 * 1024 * 16);
  hipMalloc(&out, 2048 * 8);
  run<0>(gbuf, out, 0, 0); run<0>(gbuf, out, 1, 1);
  run<4>(gbuf, out, 0, 0); run<5>(gbuf, out, 0, 0); run<1>(gbuf, out, 0, 0); run<2>(gbuf, out, 0, 0);
  printf("(MFMA work: 400 x 32 MFMAs x 64 cycles = 819 200 pipe cycles per MFMA wave; pipe = that work / the time until the last MFMA wave of the SIMD is done)\n");
  return 0;
}

// Dev tool: does gfx950 protect the DATA registers of a 128-bit LDS write / buffer store against a vector instruction that overwrites
// them right afterwards (write-after-read)?  Fixed registers and inline assembly, so that nothing sits between the two.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/b128_war_check tools/dev/b128_war_check.hip && /tmp/b128_war_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NQ, int GAP>
__global__ void lds_case(float *out) {
  __shared__ __attribute__((aligned(16))) float s[8 * 64 * 4];
  const unsigned addr = threadIdx.x * 16;
  const float val = 1.0f + threadIdx.x;
  // NQ 128-bit LDS writes of other registers first (queue pressure), then the write under test, then its data is overwritten
  asm volatile(
      "v_mov_b32 v10, %1\n v_mov_b32 v11, %1\n v_mov_b32 v12, %1\n v_mov_b32 v13, %1\n"
      "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
      "s_nop 7\n"
      ".rept %2\n ds_write_b128 %0, v[20:23] offset:4096\n .endr\n"
      "ds_write_b128 %0, v[10:13]\n"
      ".rept %3\n s_nop 0\n .endr\n"
      "v_mov_b32 v13, 0\n v_mov_b32 v12, 0\n v_mov_b32 v11, 0\n v_mov_b32 v10, 0\n"
      "s_waitcnt lgkmcnt(0)\n"
      :
      : "v"(addr), "v"(val), "n"(NQ), "n"(GAP)
      : "v10", "v11", "v12", "v13", "v20", "v21", "v22", "v23", "memory");
  __syncthreads();
  for (int e = 0; e < 4; ++e) out[(size_t)blockIdx.x * 1024 + threadIdx.x * 4 + e] = s[threadIdx.x * 4 + e];
}

template <int NQ, int GAP>
__global__ void buf_case(float *out, int soff) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)blockIdx.x * 1024, 0, 4096u, 0x00020000);
  const unsigned addr = threadIdx.x * 16;
  const float val = 1.0f + threadIdx.x;
  asm volatile(
      "v_mov_b32 v10, %1\n v_mov_b32 v11, %1\n v_mov_b32 v12, %1\n v_mov_b32 v13, %1\n"
      "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
      "s_nop 7\n"
      ".rept %4\n buffer_store_dwordx4 v[20:23], %0, %2, %3 offen offset:2048\n .endr\n"
      "buffer_store_dwordx4 v[10:13], %0, %2, %3 offen\n"
      ".rept %5\n s_nop 0\n .endr\n"
      "v_mov_b32 v13, 0\n v_mov_b32 v12, 0\n v_mov_b32 v11, 0\n v_mov_b32 v10, 0\n"
      "s_waitcnt vmcnt(0)\n"
      :
      : "v"(addr), "v"(val), "s"(rs), "s"(soff), "n"(NQ), "n"(GAP)
      : "v10", "v11", "v12", "v13", "v20", "v21", "v22", "v23", "memory");
}

// global_store_dwordx4 (saddr form): the store the compiler emits for ordinary pointer writes
template <int NQ, int GAP>
__global__ void global_case(float *out) {
  float *base = out + (size_t)blockIdx.x * 1024;
  const unsigned addr = threadIdx.x * 16;
  const float val = 1.0f + threadIdx.x;
  asm volatile(
      "v_mov_b32 v10, %1\n v_mov_b32 v11, %1\n v_mov_b32 v12, %1\n v_mov_b32 v13, %1\n"
      "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
      "s_nop 7\n"
      ".rept %3\n global_store_dwordx4 %0, v[20:23], %2 offset:2048\n .endr\n"
      "global_store_dwordx4 %0, v[10:13], %2\n"
      ".rept %4\n s_nop 0\n .endr\n"
      "v_mov_b32 v13, 0\n v_mov_b32 v12, 0\n v_mov_b32 v11, 0\n v_mov_b32 v10, 0\n"
      "s_waitcnt vmcnt(0)\n"
      :
      : "v"(addr), "v"(val), "s"(base), "n"(NQ), "n"(GAP)
      : "v10", "v11", "v12", "v13", "v20", "v21", "v22", "v23", "memory");
}

// the overwriter is an LDS READ returning into the store's data registers (asynchronous write-back, not a vector instruction)
template <int NQ, int GAP>
__global__ void buf_then_ldsread(float *out, int soff) {
  __shared__ __attribute__((aligned(16))) float s[64 * 4];
  for (int e = 0; e < 4; ++e) s[threadIdx.x * 4 + e] = -7.f;
  __syncthreads();
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)blockIdx.x * 1024, 0, 4096u, 0x00020000);
  const unsigned addr = threadIdx.x * 16;
  const float val = 1.0f + threadIdx.x;
  asm volatile(
      "v_mov_b32 v10, %1\n v_mov_b32 v11, %1\n v_mov_b32 v12, %1\n v_mov_b32 v13, %1\n"
      "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
      "s_nop 7\n"
      ".rept %4\n buffer_store_dwordx4 v[20:23], %0, %2, %3 offen offset:2048\n .endr\n"
      "buffer_store_dwordx4 v[10:13], %0, %2, %3 offen\n"
      ".rept %5\n s_nop 0\n .endr\n"
      "ds_read_b128 v[10:13], %0\n"
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
      :
      : "v"(addr), "v"(val), "s"(rs), "s"(soff), "n"(NQ), "n"(GAP)
      : "v10", "v11", "v12", "v13", "v20", "v21", "v22", "v23", "memory");
}

template <class K, class... A>
static void run(const char *what, K k, A... a) {
  const int blocks = 16384;              // 64 waves per CU: the memory pipelines are busy while each wave runs its sequence
  float *out; hipMalloc(&out, (size_t)blocks * 4096); hipMemset(out, 0, (size_t)blocks * 4096);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, a...);
  hipDeviceSynchronize();
  std::vector<float> h((size_t)blocks * 1024);
  hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
  long bad = 0, first = -1;
  for (long b = 0; b < blocks; ++b)
    for (int i = 0; i < 256; ++i) if (h[b * 1024 + i] != 1.0f + i / 4) { if (first < 0) first = b * 1024 + i; ++bad; }
  printf("%-72s %s", what, bad ? "CORRUPTED" : "ok");
  if (bad) printf("  (%ld of %ld values, first: block %ld lane %ld element %ld)", bad, (long)blocks * 256, first / 1024, (first % 1024) / 4, first % 4);
  printf("\n");
  hipFree(out);
}

int main() {
  run("ds_write_b128, data overwritten by the next instruction", lds_case<0, 0>);
  run("6 ds_write_b128 queued first, then the same", lds_case<6, 0>);
  run("6 queued, 1 s_nop between write and overwrite", lds_case<6, 1>);
  run("6 queued, 4 s_nop between", lds_case<6, 4>);
  run("6 queued, 16 s_nop between", lds_case<6, 16>);
  run("6 queued, 64 s_nop between", lds_case<6, 64>);
  run("buffer_store_dwordx4 (SGPR soffset), data overwritten by the next instr.", buf_case<0, 0>, 0);
  run("6 buffer stores queued first, then the same", buf_case<6, 0>, 0);
  run("6 queued, 1 s_nop between", buf_case<6, 1>, 0);
  run("6 queued, 4 s_nop between", buf_case<6, 4>, 0);
  run("6 queued, 16 s_nop between", buf_case<6, 16>, 0);
  run("global_store_dwordx4 (scalar base), data overwritten by the next instr.", global_case<0, 0>);
  run("6 global stores queued first, then the same", global_case<6, 0>);
  run("global_store_dwordx4, 1 s_nop between", global_case<0, 1>);
  run("global_store_dwordx4, 2 s_nop between", global_case<0, 2>);
  run("global_store_dwordx4, 3 s_nop between", global_case<0, 3>);
  run("global_store_dwordx4, 4 s_nop between", global_case<0, 4>);
  run("6 queued + global_store_dwordx4, 2 s_nop between", global_case<6, 2>);
  run("6 queued + global_store_dwordx4, 3 s_nop between", global_case<6, 3>);
  run("buffer_store_dwordx4 (SGPR soffset), 1 s_nop between", buf_case<0, 1>, 0);
  run("buffer_store_dwordx4 (SGPR soffset), 2 s_nop between", buf_case<0, 2>, 0);
  run("buffer_store_dwordx4 (SGPR soffset), 3 s_nop between", buf_case<0, 3>, 0);
  run("buffer_store_dwordx4, then ds_read_b128 INTO its data registers", buf_then_ldsread<0, 0>, 0);
  run("6 buffer stores queued first, then the same", buf_then_ldsread<6, 0>, 0);
  run("24 buffer stores queued first, then the same", buf_then_ldsread<24, 0>, 0);
  run("24 queued, 16 s_nop between store and LDS read", buf_then_ldsread<24, 16>, 0);
  return 0;
}

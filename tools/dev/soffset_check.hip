// Dev tool: is the scalar offset of a raw buffer access part of the hardware's range check on this GPU?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/soffset_check tools/dev/soffset_check.hip && /tmp/soffset_check
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *buf, unsigned records, int soff, float *rd) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, records, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b32(42.f + threadIdx.x, rs, threadIdx.x * 4, soff, 0);
  rd[threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(rs, threadIdx.x * 4, soff, 0);
}
int main() {
  float *buf, *rd; hipMalloc(&buf, 4096); hipMalloc(&rd, 256); hipMemset(buf, 0, 4096);
  // 64 floats of records (256 B); store with voffset in range, soffset = 1024 B (beyond the records)
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, buf, 256u, 1024, rd);
  float h[1024], r[64]; hipMemcpy(h, buf, 4096, hipMemcpyDeviceToHost); hipMemcpy(r, rd, 256, hipMemcpyDeviceToHost);
  printf("store with soffset past num_records: buf[256 + 0] = %g (42 = written: soffset NOT range-checked; 0 = dropped)\n", h[256]);
  printf("load  with soffset past num_records: %g (42 = read through; 0 = out of range)\n", r[0]);
  // voffset past the records, soffset 0
  hipMemset(buf, 0, 4096);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, buf + 0, 128u, 0, rd);      // lanes 32..63 out of range by voffset
  hipMemcpy(h, buf, 4096, hipMemcpyDeviceToHost);
  printf("voffset past num_records: buf[31] = %g buf[32] = %g (expect 73 and 0)\n", h[31], h[32]);
  // the SUM is what counts: 256 B of records, voffset = 4 * lane (all < 256), soffset = 128: lanes 32..63 have voffset + soffset >= 256
  hipMemset(buf, 0, 4096);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, buf, 256u, 128, rd);
  hipMemcpy(h, buf, 4096, hipMemcpyDeviceToHost);
  unsigned *u = reinterpret_cast<unsigned *>(h);
  printf("sum check: word 32+31 = %u (expect 73), word 32+32 = %u (0 = the sum voffset + soffset is range-checked; 74 = only voffset is)\n", u[63], u[64]);
  return 0;
}

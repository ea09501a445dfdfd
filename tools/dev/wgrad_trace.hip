// Dev tool: per-workgroup timeline of one weight-gradient launch (wgrad_f32_v2_kernel compiled with its DF_WTRACE hooks on):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -o tools/dev/wgrad_trace tools/dev/wgrad_trace.hip densefusion_amd/csrc/common.hip
//   tools/dev/wgrad_trace B H W Cin Cout k pad dil
#include <hip/hip_runtime.h>
__device__ unsigned long long *df_wtrace_buf = nullptr;
#define DF_WTRACE(i)                                                                                              \
  do {                                                                                                           \
    if (threadIdx.x == 0 && df_wtrace_buf) {                                                                     \
      unsigned long long *r = df_wtrace_buf + (size_t)blockIdx.x * 8;                                            \
      r[i] = __builtin_readcyclecounter();                                                                       \
      if ((i) == 0) { r[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4); r[5] = __builtin_amdgcn_s_getreg((3 << 11) | 20); r[6] = wall_clock64(); } \
      if ((i) == 3) r[7] = wall_clock64();                                                                       \
    }                                                                                                            \
  } while (0)
#include "../../densefusion_amd/csrc/igemm.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
using namespace df;

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8, H = argc > 2 ? atoi(argv[2]) : 20, W = argc > 3 ? atoi(argv[3]) : 20;
  const int Cin = argc > 4 ? atoi(argv[4]) : 512, Cout = argc > 5 ? atoi(argv[5]) : 512, k = argc > 6 ? atoi(argv[6]) : 3;
  const int pad = argc > 7 ? atoi(argv[7]) : 4, dil = argc > 8 ? atoi(argv[8]) : 4;
  const size_t M = (size_t)B * H * W, K = (size_t)k * k * Cin;
  float *x, *dy, *dw, *ws;
  hipMalloc(&x, M * Cin * 4); hipMalloc(&dy, M * Cout * 4); hipMalloc(&dw, (size_t)Cout * K * 4);
  hipMemset(x, 0, M * Cin * 4); hipMemset(dy, 0, M * Cout * 4);
  ConvParams p;
  p.in = x; p.out = dy; p.B = B; p.H = H; p.W = W; p.OH = H; p.OW = W; p.Cin = Cin; p.in_ld = Cin; p.Cout = Cout; p.out_ld = Cout;
  p.KH = p.KW = k; p.pad = pad; p.dil = dil;
  const size_t need = wgrad_workspace_bytes(p);
  hipMalloc(&ws, need);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch_wgrad(p, dw, nullptr, ws, need, 0, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) launch_wgrad(p, dw, nullptr, ws, need, 0, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("M=%zu Cout=%d K=%zu: %.1f us/launch (kernel + reduce), %.1f TFLOP/s\n", M, Cout, K, ms * 100, 2.0 * M * Cout * K / (ms / 10) / 1e9);
  const size_t maxwg = 1 << 16;
  unsigned long long *buf;
  hipMalloc(&buf, maxwg * 8 * 8);
  hipMemset(buf, 0, maxwg * 8 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(df_wtrace_buf), &buf, sizeof(buf));
  launch_wgrad(p, dw, nullptr, ws, need, 0, 0);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(maxwg * 8);
  hipMemcpy(h.data(), buf, maxwg * 8 * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, int> per_cu;
  unsigned long long t0 = ~0ull, tend = 0;
  double pro = 0, loop = 0, epi = 0, cyc = 0, wall = 0; long n = 0;
  std::vector<double> starts, ends;
  for (size_t t = 0; t < maxwg; ++t) if (h[t * 8]) {
    t0 = std::min(t0, h[t * 8]); tend = std::max(tend, h[t * 8 + 3]);
    per_cu[(unsigned)((h[t * 8 + 5] & 7) << 8 | ((h[t * 8 + 4] >> 8) & 0xff))]++;
    pro += h[t*8+1] - h[t*8]; loop += h[t*8+2] - h[t*8+1]; epi += h[t*8+3] - h[t*8+2];
    cyc += h[t*8+3] - h[t*8]; wall += h[t*8+7] - h[t*8+6]; ++n;
  }
  for (size_t t = 0; t < maxwg; ++t) if (h[t * 8]) { starts.push_back((double)(h[t*8] - t0)); ends.push_back((double)(h[t*8+3] - t0)); }
  std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
  int mx = 0, mn = 1 << 30;
  for (auto &kv : per_cu) { mx = std::max(mx, kv.second); mn = std::min(mn, kv.second); }
  const double ghz = cyc / wall * 0.1;
  printf("workgroups %ld on %zu CUs (%d..%d per CU); cycle counter %.3f GHz\n", n, per_cu.size(), mn, mx, ghz);
  printf("mean cycles: prologue %.0f  main loop %.0f  epilogue (partial tile stores) %.0f;  launch span %.0f cycles = %.1f us\n", pro / n, loop / n, epi / n,
         (double)(tend - t0), (double)(tend - t0) / ghz / 1e3);
  printf("start times: median %.0f, 90%% %.0f, last %.0f;  end times: first %.0f, median %.0f, last %.0f\n", starts[n / 2], starts[n * 9 / 10], starts[n - 1], ends[0],
         ends[n / 2], ends[n - 1]);
  return 0;
}

// Dev tool: per-workgroup timeline of one igemm launch (where the prologue / main loop / epilogue of the workgroups that share
// a CU fall relative to each other).  Compiles the product kernel with its DF_TRACE hooks turned on:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -o tools/dev/igemm_trace tools/dev/igemm_trace.hip densefusion_amd/csrc/common.hip
//   tools/dev/igemm_trace M N K [unused] [1 = column-sum epilogue only]
#include <hip/hip_runtime.h>
__device__ unsigned long long *df_trace_buf = nullptr;
#define DF_TRACE(i)                                                                                                \
  do {                                                                                                           \
    if (threadIdx.x == 0 && df_trace_buf) {                                                                      \
      unsigned long long *r = df_trace_buf + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 16;                                    \
      r[i] = __builtin_readcyclecounter();                                                                       \
      if ((i) == 0) { r[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4); r[5] = __builtin_amdgcn_s_getreg((3 << 11) | 20); r[6] = wall_clock64(); } \
      if ((i) == 3) r[7] = wall_clock64();                                                                       \
    }                                                                                                            \
  } while (0)
#define DF_TRACE_WAVE_END(w)                                                                                      \
  do {                                                                                                           \
    if ((threadIdx.x & 63) == 0 && df_trace_buf)                                                                 \
      df_trace_buf[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 16 + 12 + (w)] = __builtin_readcyclecounter(); \
  } while (0)
#include "../../densefusion_amd/csrc/igemm.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
using namespace df;

int main(int argc, char **argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 286720, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 256;
  const int stag = argc > 4 ? atoi(argv[4]) : 1;
  const int nostore = argc > 5 ? atoi(argv[5]) : 0;      // 1: column-sum epilogue only, nothing stored per element
  (void)stag;
  float *a, *w, *o;
  hipMalloc(&a, (size_t)M * K * 4); hipMalloc(&w, (size_t)N * K * 4); hipMalloc(&o, (size_t)M * N * 4);
  hipMemset(a, 0, (size_t)M * K * 4); hipMemset(w, 0, (size_t)N * K * 4);
  ConvParams p;
  p.in = a; p.wgt = w; p.out = o; p.B = M; p.Cin = K; p.in_ld = K; p.Cout = N; p.out_ld = N; p.act = ACT_RELU;
  const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  if (nostore) {
    float *cs; hipMalloc(&cs, (size_t)conv_colsum_rows(p) * N * 4);
    p.out = nullptr; p.colsum = cs;
  }
  unsigned long long *buf;
  hipMalloc(&buf, tiles * 4 * 16 * 8);           // (64x64 tiling has 4x the tiles)
  hipMemset(buf, 0, tiles * 4 * 16 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch_conv(p, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) launch_conv(p, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("M=%d N=%d K=%d stagger=%d: %.1f us/launch, %.1f TFLOP/s\n", M, N, K, stag, ms * 100, 2.0 * M * N * K / (ms / 10) / 1e9);
  hipMemcpyToSymbol(HIP_SYMBOL(df_trace_buf), &buf, sizeof(buf));
  launch_conv(p, 0);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(tiles * 16);
  hipMemcpy(h.data(), buf, tiles * 16 * 8, hipMemcpyDeviceToHost);
  // per CU: the workgroups in start order
  std::map<unsigned, std::vector<long>> by_cu;
  unsigned long long t0 = ~0ull;
  for (long t = 0; t < tiles; ++t) if (h[t * 16]) { t0 = std::min(t0, h[t * 16]); by_cu[(unsigned)((h[t * 16 + 5] & 7) << 8 | ((h[t * 16 + 4] >> 8) & 0xff))].push_back(t); }
  double pro = 0, loop = 0, epi = 0; long n = 0;
  for (long t = 0; t < tiles; ++t) if (h[t * 16]) { pro += h[t*16+1] - h[t*16]; loop += h[t*16+2] - h[t*16+1]; epi += h[t*16+3] - h[t*16+2]; ++n; }
  {
    double cyc = 0, wall = 0;
    for (long t = 0; t < tiles; ++t) if (h[t * 16]) { cyc += h[t*16+3] - h[t*16]; wall += h[t*16+7] - h[t*16+6]; }
    printf("cycle counter runs at %.3f GHz (against the 100 MHz wall clock)\n", cyc / wall * 0.1);
  }
  printf("CUs seen %zu, workgroups %ld; mean cycles: prologue %.0f  main loop %.0f (%.0f / k tile)  epilogue %.0f\n", by_cu.size(), n, pro / n, loop / n,
         loop / n / ((K + 31) / 32), epi / n);
  {
    // MFMA-pipe occupancy per CU: every 128x128 workgroup needs nkt * 4096 pipe cycles on each SIMD (64 MFMAs of 64 cycles per k tile)
    // -- (a) over the CU's whole span, (b) over its middle (first and last workgroup lifetimes cut off: the steady state)
    const double need = (double)((K + 31) / 32) * 4096.0 * (pick_cfg(p).bm * pick_cfg(p).bn) / 16384.0;
    double ua = 0, ub = 0, span_max = 0; int nb = 0;
    for (auto &kv : by_cu) {
      auto &v = kv.second;
      unsigned long long s0 = ~0ull, e1 = 0;
      for (long t : v) { s0 = std::min(s0, h[t * 16]); e1 = std::max(e1, h[t * 16 + 3]); }
      ua += need * v.size() / (double)(e1 - s0);
      span_max = std::max(span_max, (double)(e1 - t0));
      const double life = (pro + loop + epi) / n;
      const double w0 = s0 + 1.5 * life, w1 = e1 - 1.5 * life;
      if (w1 > w0 + life) {
        double work = 0;
        for (long t : v) {      // the share of the workgroup's main loop that falls inside the window
          const double a = std::max((double)h[t * 16 + 1], w0), b = std::min((double)h[t * 16 + 2], w1);
          if (b > a) work += need * (b - a) / (double)(h[t * 16 + 2] - h[t * 16 + 1]);
        }
        ub += work / (w1 - w0); ++nb;
      }
    }
    printf("MFMA pipe busy: %.3f over whole CU spans, %.3f in the steady-state window (%d CUs); launch span %.0f cycles\n", ua / by_cu.size(),
           nb ? ub / nb : 0.0, nb, span_max);
  }
  {
    double d[4] = {0, 0, 0, 0};
    for (long t = 0; t < tiles; ++t) if (h[t * 16] && h[t * 16 + 8]) {
      d[0] += h[t*16+8] - h[t*16+2]; d[1] += h[t*16+9] - h[t*16+8];
      if (h[t*16+10]) { d[2] += h[t*16+10] - h[t*16+9]; d[3] += h[t*16+11] - h[t*16+10]; }
    }
    printf("epilogue split (wave 0, mean cycles): band 0 acc->LDS %.0f, read-out + stores %.0f; band 1 acc->LDS %.0f, read-out + stores %.0f\n",
           d[0] / n, d[1] / n, d[2] / n, d[3] / n);
  }
  {
    // slot turnover: per CU, (span * slots - sum of lifetimes) / workgroups, slots = workgroups that started before the first one ended
    double gap = 0, skew = 0; long ng = 0, ns = 0;
    for (auto &kv : by_cu) {
      auto v = kv.second;
      std::sort(v.begin(), v.end(), [&](long x, long y) { return h[x * 16] < h[y * 16]; });
      unsigned long long first_end = ~0ull, e1 = 0; double life = 0;
      for (long t : v) { first_end = std::min(first_end, h[t * 16 + 3]); e1 = std::max(e1, h[t * 16 + 3]); life += h[t*16+3] - h[t*16]; }
      int slots = 0;
      for (long t : v) slots += h[t * 16] < first_end;
      gap += ((double)(e1 - h[v[0] * 16]) * slots - life); ng += v.size();
      for (long t : v) if (h[t * 16 + 12]) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < 4; ++w) { lo = std::min(lo, h[t * 16 + 12 + w]); hi = std::max(hi, h[t * 16 + 12 + w]); }
        skew += hi - lo; ++ns;
      }
    }
    printf("slot turnover: %.0f idle cycles per workgroup (incl. the drain at the end); wave-end skew inside a workgroup %.0f cycles\n", gap / ng,
           ns ? skew / ns : 0.0);
  }
  int shown = 0;
  for (auto &kv : by_cu) {
    if (shown++ >= 3) break;
    auto v = kv.second;
    std::sort(v.begin(), v.end(), [&](long x, long y) { return h[x * 16] < h[y * 16]; });
    printf("CU %03x: %zu workgroups; first 8 [start, loop start, loop end, end] in cycles since the first start\n", kv.first, v.size());
    for (size_t i = 0; i < v.size() && i < 12; ++i)
      printf("   wg %6ld  %8llu %8llu %8llu %8llu\n", v[i], h[v[i]*16] - t0, h[v[i]*16+1] - t0, h[v[i]*16+2] - t0, h[v[i]*16+3] - t0);
  }
  return 0;
}

// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// One kernel covers every dense contraction of the path: the dilated ResNet-18 3x3 convs
// (lib/extractors.py:14-16), the 7x7/2 stem, the 1x1 convs of PSP (lib/pspnet.py:12-18), the three
// 3x3 up-convs (lib/pspnet.py:30-34) and -- as 1x1 "convs" over point rows -- the Conv1d(k=1) MLPs
// of PoseNetFeat / PoseRefineNetFeat and the r/t/c heads (lib/network.py:53-68,107-121).
//
// Why fp32 MFMA: ADD(-S) must match the reference to 1e-4 m through ~25 un-normalised layers, and the
// f32-input MFMA is bit-for-bit an fp32 fma chain at the full fp32 rate (157 TFLOP/s dense).
//
// Tiling (64-wide wavefronts): 256 threads = 4 waves per workgroup, workgroup tile BM x BN, wave tile
// of TM x TN 32x32 accumulators, BK = 32.  A (pixels x k) and B (channels x k) tiles are staged
// global -> VGPR -> LDS with 16-B vectors, k contiguous (NHWC activations, [Cout][kh][kw][Cin]
// weights), rows padded to 36 floats so the ds_read_b128 fragment reads are bank-conflict free.
// Lane l = (i = l&31, h = l>>5) reads k = 8g+4h .. 8g+4h+3 of its row with one b128 read and feeds
// MFMA step j with element j, for A and B alike, so each 32x32x2 step sums k = 8g+j and 8g+4+j.
// LDS is double buffered; the next tile's global loads are in flight while the MFMAs run.
// Epilogue fused in registers: bias (shared or per row group), residual add, ReLU / PReLU, store at a
// channel offset of a wider row (writes straight into concat buffers), and an optional per-wave
// column sum of the activated tile (the AvgPool1d over points, lib/network.py:65).
#include <cstdlib>

#include "igemm.h"
#include <algorithm>
#include <type_traits>

// DF_TRACE(i): per-workgroup time stamps for tools/dev/igemm_trace.hip (compiles this file with the hook defined); nothing otherwise
#ifndef DF_WTRACE
#define DF_WTRACE(i)      // the same for the weight-gradient kernel (tools/dev/wgrad_trace.hip)
#endif
#ifndef DF_TRACE
#define DF_TRACE(i)
#define DF_TRACE_WAVE_END(w)
#endif

namespace df {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDK = 36;   // padded LDS row (floats): 16 lanes x 16 B land on 64 distinct banks

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// n / d for 0 <= n < 2^31 with a host-made magic pair (make_fdiv): one v_mul_hi + shift instead of ~40 instructions
__device__ __forceinline__ int fdiv(int n, unsigned magic, int sh, int d) {
  return d == 1 ? n : (int)(__umulhi((unsigned)n, magic) >> sh);
}

// 16-byte non-temporal buffer store with a SCALAR offset, followed by two wait states.  gfx950 hazard (tools/dev/b128_war_check.hip):
// a vector instruction that overwrites the data registers of a 128-bit store too soon after it corrupts the last dword of lanes 12-15
// of each row of 16 -- always for a global_store_dwordx4 overwritten in the next slot (hipcc guards those with `s_nop 1`), now and then
// under load for a buffer store with a scalar-offset register, which hipcc does not guard.  Measured safe distance: 2 wait states for
// the global form, 1 for the buffer form; 2 are used here.  Store and s_nop are one asm statement: nothing can be scheduled between.
__device__ __forceinline__ void buffer_store_b128_nt(u32x4 data, __amdgpu_buffer_rsrc_t rs, unsigned voffset, int soffset) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" : : "v"(data), "v"(voffset), "s"(rs), "s"(soffset) : "memory");
}

// magic = ceil(2^(31+L) / d) with 2^(L-1) < d <= 2^L: exact for every n < 2^31 (error term n * (magic*d - 2^(31+L)) < 2^(31+L))
void make_fdiv(long d, unsigned &magic, int &sh) {
  magic = 0; sh = 0;
  if (d <= 1) return;
  int L = 0;
  while ((1L << L) < d) ++L;
  magic = (unsigned)((((unsigned __int128)1 << (31 + L)) + (unsigned long)d - 1) / (unsigned long)d);
  sh = L - 1;
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void igemm_f32_kernel(const ConvParams p) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_ROWS = BM / 32, B_ROWS = BN / 32;   // tile rows staged per thread
  constexpr int TILE = (BM + BN) * LDK;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int M = p.B * p.OH * p.OW;
  const int K = p.KH * p.KW * p.Cin;
  const int tiles_n = (p.Cout + BN - 1) / BN;

  // XCD-aware tile order: consecutive workgroup ids are dealt round-robin over the 8 XCDs, so give each
  // XCD a contiguous run of tiles (neighbours share the A rows / weight panel in that XCD's L2).
  int wgid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int n_tile = wgid % tiles_n, m_tile = wgid / tiles_n;
  const int m0 = m_tile * BM, n0 = n_tile * BN;

  const int z = blockIdx.z;
  const float *__restrict__ in = p.in + p.in_coff + z * p.z_in_coff;
  const float *__restrict__ wgt = p.wgt + (size_t)z * p.z_wgt;
  const int out_coff = p.out_coff + (int)(z * p.z_out_coff);

  // ---- loader state: this thread stages k-vector `vec` of rows lrow + 32*i ----
  const int vec = tid & 7, lrow = tid >> 3;
  int a_iy0[A_ROWS], a_ix0[A_ROWS], a_pix[A_ROWS];   // a_pix: index of pixel (b, 0, 0)
#pragma unroll
  for (int i = 0; i < A_ROWS; ++i) {
    const int m = m0 + lrow + 32 * i;
    if (m < M) {
      const int ohw = p.OH * p.OW;
      const int b = m / ohw, rem = m - b * ohw;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_iy0[i] = oy * p.stride - p.pad;
      a_ix0[i] = ox * p.stride - p.pad;
      a_pix[i] = b * p.H * p.W;
    } else {
      a_iy0[i] = -(1 << 28);     // fails every bounds check -> zero rows
      a_ix0[i] = 0;
      a_pix[i] = 0;
    }
  }
  const bool one_tap = (p.KH * p.KW == 1);
  const int cin_shift = 31 - __builtin_clz(p.Cin);   // multi-tap layers have power-of-two Cin (host-checked)
  const int up = p.up;

  f32x4 ra[A_ROWS], rb[B_ROWS];
  auto load_tile = [&](int kt) {
    const int k = kt * BK + vec * 4;
    const bool kok = k < K;
    int c = k, dy = 0, dx = 0;
    if (!one_tap) {
      const int tap = k >> cin_shift;
      c = k & (p.Cin - 1);
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      dy = ky * p.dil;
      dx = kx * p.dil;
    }
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;          // virtual (input-dilated) coordinates
      bool ok = kok && iy >= 0 && ix >= 0;
      if (up > 1) {
        ok = ok && (iy % up == 0) && (ix % up == 0);
        iy /= up;
        ix /= up;
      }
      ok = ok && iy < p.H && ix < p.W;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4 *>(in + (size_t)(a_pix[i] + iy * p.W + ix) * p.in_ld + c);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
      const int n = n0 + lrow + 32 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (kok && n < p.Cout) v = *reinterpret_cast<const f32x4 *>(wgt + (size_t)n * K + k);
      rb[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
    float *sA = smem + buf * TILE, *sB = sA + BM * LDK;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) *reinterpret_cast<f32x4 *>(sA + (lrow + 32 * i) * LDK + vec * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) *reinterpret_cast<f32x4 *>(sB + (lrow + 32 * i) * LDK + vec * 4) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int li = lane & 31, lh = lane >> 5;
  const int nkt = (K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);       // global loads fly while the MFMAs below run
    const float *sA = smem + buf * TILE + (wm * WM + li) * LDK + lh * 4;
    const float *sB = smem + buf * TILE + BM * LDK + (wn * WN + li) * LDK + lh * 4;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4 *>(sA + i * 32 * LDK + g * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4 *>(sB + j * 32 * LDK + g * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) store_tile(buf ^ 1);     // the other buffer was last read before the previous barrier
    __syncthreads();
  }

  // ---- epilogue ----
  const float slope = (p.act == ACT_PRELU) ? p.prelu[0] : 0.f;
  const int grp = (p.rows_per_group > 0) ? m0 / p.rows_per_group : 0;   // a tile never straddles groups
  const float *bias = p.bias ? p.bias + z * p.z_bias + (p.bias_group_ld > 0 ? (size_t)grp * p.bias_group_ld : 0) : nullptr;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WN + j * 32 + li;
    const bool nok = n < p.Cout;
    const float bv = (bias && nok) ? bias[n] : 0.f;
    float csum = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int m = m0 + row;
        float v = acc[i][j][e] + bv;
        const bool ok = nok && m < M;
        if (p.res && ok) v += p.res[(size_t)m * p.res_ld + p.res_coff + n];
        if (p.act == ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (p.act == ACT_PRELU) v = v > 0.f ? v : v * slope;
        if (ok && p.out) p.out[(size_t)m * p.out_ld + out_coff + n] = v;
        if (p.colsum) {
          const bool real = ok && (p.rows_per_group <= 0 || (m % p.rows_per_group) < p.rows_valid);
          csum += real ? v : 0.f;
        }
      }
    }
    if (p.colsum) {
      csum += __shfl_xor(csum, 32);            // the two lane halves hold different rows of one column
      if (lh == 0 && nok) p.colsum[((size_t)z * gridDim.x / tiles_n * WAVES_M + (size_t)m_tile * WAVES_M + wm) * p.Cout + n] = csum;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// v2 -- the latency kernel, for grids that cannot fill the chip (under two workgroups per CU: single-frame training passes,
// the per-object head GEMMs of a small batch).  The v1 tiling, software-pipelined inside the wave so that a workgroup ALONE on
// its CU keeps the matrix pipe fed (81 % MFMA busy solo, tools/dev/igemm_trace.hip); 72 KB of LDS and ~200 registers, so
// at most two workgroups per CU -- on full grids v4 below (4-6 per CU) is 1-9 % faster and takes over.
//   * every global access is a bounds-checked buffer load (out-of-range -> 0, no branches), so the
//     loop body is one basic block and address arithmetic, loads and LDS traffic sit between MFMAs;
//   * 3-stage pipeline: tile kt is multiplied from LDS while tile kt+1 moves registers -> LDS and
//     tile kt+2 is in flight from HBM/L2; one barrier per tile, placed BEFORE the last quarter of
//     the tile's MFMAs so the next tile's first fragment reads hide under them;
//   * epilogue through LDS: each wave transposes its accumulators so that global stores / residual
//     loads are 16-byte vectors covering whole 128/256-byte row segments.
// Same k order and MFMA sequence per output element as v4: which of the two a launch takes never changes a bit of its result.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, int BKT>
__global__ __launch_bounds__(256) void igemm_f32_v2_kernel(const ConvParams p) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  // BKT = k depth of one LDS tile (32 or 64): the MFMA time between two barriers is BKT/2 * TM*TN * 64 cycles,
  // so the small 64x64 tile takes BKT = 64 to keep the barrier cost per MFMA where the 128x128 tile has it.
  constexpr int VPR = BKT / 4;                  // 16-byte vectors per tile row
  constexpr int RPP = 256 / VPR;                // tile rows staged per pass of the 256 threads
  constexpr int A_ROWS = BM / RPP, B_ROWS = BN / RPP;
  constexpr int LDK = BKT + 4;                  // padded row: ds_read_b128 of 16 rows hits 64 distinct banks
  constexpr int BK = BKT;
  constexpr int NP = BKT / 16;                  // pairs of 8-wide k groups per tile
  constexpr int TILE = (BM + BN) * LDK;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int M = p.B * p.OH * p.OW;
  const int K = p.KH * p.KW * p.Cin;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  int wgid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  // column-tile groups: when the weight operand is too big for an XCD's L2 (4 MB), the tiles are walked group by group
  // (p.ngroup column tiles, all row blocks, next group ...) so that the group's weight slice stays L2-resident while the
  // activation rows stream through once per group instead of every column tile missing on both operands
  int n_tile, m_tile;
  {
    const int GN = p.ngroup > 0 && p.ngroup < tiles_n ? p.ngroup : tiles_n;
    const int tiles_m = (int)(gridDim.x / tiles_n);
    const int full = tiles_m * GN;
    const int g = wgid / full, rem = wgid - g * full;
    const int gw = min(GN, tiles_n - g * GN);
    m_tile = rem / gw;
    n_tile = g * GN + (rem - m_tile * gw);
  }
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const int z = blockIdx.z;
  const int out_coff = p.out_coff + (int)(z * p.z_out_coff);

  // buffer descriptors (wave-uniform): reads past num_records return 0
  // (one descriptor per z slice: a z-batched launch may span more than the 4 GB a descriptor can address)
  const unsigned in_bytes = (unsigned)((size_t)p.B * p.H * p.W * p.in_ld * sizeof(float));
  const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in + (size_t)z * p.z_in_coff), 0, in_bytes, 0x00020000);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wgt + (size_t)z * p.z_wgt), 0,
                                                      (unsigned)((size_t)p.Cout * K * sizeof(float)), 0x00020000);
  const int in_c0 = p.in_coff;

  const int vec = tid % VPR, lrow = tid / VPR;
  int a_iy0[A_ROWS], a_ix0[A_ROWS], a_off[A_ROWS];   // a_off: element offset of (b, iy0, ix0, first channel)
#pragma unroll
  for (int i = 0; i < A_ROWS; ++i) {
    const int m = m0 + lrow + RPP * i;
    if (m < M) {
      const int ohw = p.OH * p.OW;
      const int b = m / ohw, rem = m - b * ohw;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_iy0[i] = oy * p.stride - p.pad;
      a_ix0[i] = ox * p.stride - p.pad;
      a_off[i] = ((b * p.H + a_iy0[i]) * p.W + a_ix0[i]) * p.in_ld + in_c0;
    } else {
      a_iy0[i] = -(1 << 28);
      a_ix0[i] = 0;
      a_off[i] = 0;
    }
  }
  int b_off[B_ROWS];
#pragma unroll
  for (int i = 0; i < B_ROWS; ++i) {
    const int n = n0 + lrow + RPP * i;
    b_off[i] = n < p.Cout ? n * K : -1;
  }
  const bool one_tap = (p.KH * p.KW == 1);
  const int cin_shift = 31 - __builtin_clz(p.Cin);
  const int nkt = (K + BK - 1) / BK;

  // branch-free k -> (tap, channel) decode: single-tap layers use shift 31 / mask ~0 (tap = 0, c = k)
  const int k_shift = one_tap ? 31 : cin_shift;
  const int c_mask = one_tap ? 0x7fffffff : p.Cin - 1;
  const int kw_magic = (65536 + p.KW - 1) / p.KW;     // tap / KW == (tap * kw_magic) >> 16 for tap < 64

  u32x4 ra[A_ROWS], rb[B_ROWS];
  auto issue_loads = [&](int kt) {
    // the tile index goes through an opaque asm so that the address arithmetic below cannot be strength-reduced
    // into loop-header induction updates: it has to stay here, between the MFMAs, where its issue slots are free
    asm volatile("" : "+s"(kt));
    const int k = kt * BK + vec * 4;
    const int kok = k < K;
    const int tap = k >> k_shift;
    const int c = k & c_mask;
    const int ky = (tap * kw_magic) >> 16, kx = tap - ky * p.KW;
    const int dy = ky * p.dil, dx = kx * p.dil;
    const int doff = (dy * p.W + dx) * p.in_ld + c;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
      // bitwise (not short-circuit) so no control flow is generated: the loop body stays one basic block
      const int ok = kok & (int)((unsigned)iy < (unsigned)p.H) & (int)((unsigned)ix < (unsigned)p.W);
      unsigned off = ok ? (unsigned)(a_off[i] + doff) * 4u : 0xffffffffu;
      asm("" : "+v"(off));      // opaque: keeps hipcc from turning the select into two branchy loads
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
      const int ok = kok & (int)(b_off[i] >= 0);
      unsigned off = ok ? (unsigned)(b_off[i] + k) * 4u : 0xffffffffu;
      asm("" : "+v"(off));
      rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0);
    }
  };
  auto write_lds = [&](int buf) {
    float *sA = smem + buf * TILE, *sB = sA + BM * LDK;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) *reinterpret_cast<u32x4 *>(sA + (lrow + RPP * i) * LDK + vec * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) *reinterpret_cast<u32x4 *>(sB + (lrow + RPP * i) * LDK + vec * 4) = rb[i];
  };

  const int li = lane & 31, lh = lane >> 5;
  const int fa = (wm * WM + li) * LDK + lh * 4;                  // this lane's A-fragment base (floats)
  const int fb = BM * LDK + (wn * WN + li) * LDK + lh * 4;
  // fragments of two 8-wide k groups: [half][tile]
  f32x4 a0[2][TM], b0[2][TN], a1[2][TM], b1[2][TN];
  auto read_frags = [&](int buf, int gpair, f32x4 (&a)[2][TM], f32x4 (&b)[2][TN]) {
    const float *base = smem + buf * TILE;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[h][i] = *reinterpret_cast<const f32x4 *>(base + fa + i * 32 * LDK + (gpair * 2 + h) * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[h][j] = *reinterpret_cast<const f32x4 *>(base + fb + j * 32 * LDK + (gpair * 2 + h) * 8);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  auto mfma_group = [&](const f32x4 (&a)[TM], const f32x4 (&b)[TN]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
  };

  // prologue: tile 0 -> LDS, tile 1 -> registers
  DF_TRACE(0);
  issue_loads(0);
  write_lds(0);
  issue_loads(1);
  __syncthreads();
  read_frags(0, 0, a0, b0);
  DF_TRACE(1);
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    // pair pp multiplies out of register set pp&1 while pair pp+1 is read into the other set
#pragma unroll
    for (int pp = 0; pp < NP - 1; ++pp) {
      if (pp & 1) read_frags(buf, pp + 1, a0, b0); else read_frags(buf, pp + 1, a1, b1);
      if (pp == NP / 2 - 1) {
        // register -> LDS hand-over of tile kt+1 and the issue of tile kt+2 sit mid-tile: the loads then have
        // about a full tile of MFMA time to land before their ds_write
        __builtin_amdgcn_sched_barrier(0);
        write_lds(buf ^ 1);             // tile kt+1 (zeros past the end): registers -> the idle buffer
        issue_loads(kt + 2);            // tile kt+2 starts its trip; consumed one full tile later
      }
      if (pp & 1) { mfma_group(a1[0], b1[0]); mfma_group(a1[1], b1[1]); }
      else { mfma_group(a0[0], b0[0]); mfma_group(a0[1], b0[1]); }
      if (pp == NP / 2 - 1) {
        // Spread the loader's address arithmetic, the ds_writes and the buffer loads evenly over the MFMAs of
        // this pair: an f32 MFMA keeps the matrix pipe busy for 64 cycles, during which the same wave can issue
        // ~10 other instructions for free -- but a run of 40 of them between two MFMAs leaves the pipe idle.
#pragma unroll
        for (int q = 0; q < 8 * TM * TN; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, TM * TN == 4 ? 3 : (TM * TN == 2 ? 5 : 8), 0);     // VALU
          __builtin_amdgcn_sched_group_barrier(0x004, TM * TN == 4 ? 1 : (TM * TN == 2 ? 2 : 3), 0);     // SALU
          __builtin_amdgcn_sched_group_barrier(0x090, TM * TN == 4 ? 1 : 2, 0);                          // DS | VMEM
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();                    // everyone has read tile kt out of `buf`; tile kt+1 is complete in buf^1
    read_frags(buf ^ 1, 0, a0, b0);     // lands while the last pair of tile kt multiplies
    __builtin_amdgcn_sched_barrier(0);  // hipcc would otherwise hoist these MFMAs above the barrier
    mfma_group(a1[0], b1[0]);
    mfma_group(a1[1], b1[1]);
  }
  __syncthreads();                    // the speculative fragment reads above are done before smem is reused
  DF_TRACE(2);

  // ---- epilogue: accumulators -> LDS (per-wave region) -> 16-byte row segments ----
  constexpr int EP_LD = WN + 4;
  float *ep = smem + wave * (WM * EP_LD);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) ep[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * EP_LD + j * 32 + li] = acc[i][j][e];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const float slope = (p.act == ACT_PRELU) ? p.prelu[0] : 0.f;
  const int grp = (p.rows_per_group > 0) ? m0 / p.rows_per_group : 0;
  const float *bias = p.bias ? p.bias + z * p.z_bias + (p.bias_group_ld > 0 ? (size_t)grp * p.bias_group_ld : 0) : nullptr;
  constexpr int LPR = WN / 4;            // lanes per row
  constexpr int ERPP = 64 / LPR;         // rows per pass
  const int c4 = (lane % LPR) * 4, r0 = lane / LPR;
  const int n = n0 + wn * WN + c4;
  const bool nok = n < p.Cout;           // Cout % 4 == 0 (host-checked): a vector is all-in or all-out
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias && nok) bv = *reinterpret_cast<const f32x4 *>(bias + n);
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  const int valid_rows = p.rows_per_group > 0 ? p.rows_valid - (m0 - grp * p.rows_per_group) : (1 << 30);
  // the residual rows are requested half a wave tile at a time (two trips to memory instead of one per group of rows)
  constexpr int NPS = WM / ERPP, HPS = NPS / 2;
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
  f32x4 rres[HPS];
  if (p.res) {
#pragma unroll
    for (int q = 0; q < HPS; ++q) {
      const int m = m0 + wm * WM + (hh * HPS + q) * ERPP + r0;
      f32x4 rv = {0.f, 0.f, 0.f, 0.f};
      if (nok && m < M) rv = *reinterpret_cast<const f32x4 *>(p.res + (size_t)m * p.res_ld + p.res_coff + n);
      rres[q] = rv;
    }
  }
#pragma unroll
  for (int q = 0; q < HPS; ++q) {
    const int ps = hh * HPS + q;
    const int row = ps * ERPP + r0;
    const int m = m0 + wm * WM + row;
    const bool ok = nok && m < M;
    f32x4 v = *reinterpret_cast<const f32x4 *>(ep + row * EP_LD + c4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += bv[e];
    if (p.res) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += rres[q][e];
    }
    if (p.act == ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    } else if (p.act == ACT_PRELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
    }
    if (ok && p.out) *reinterpret_cast<f32x4 *>(p.out + (size_t)m * p.out_ld + out_coff + n) = v;
    if (p.colsum) {
      // rows of this tile that count: a tile never straddles groups, so (m % rows_per_group) = (m0 % rows_per_group) + local row
      const bool real = ok && wm * WM + row < valid_rows;
#pragma unroll
      for (int e = 0; e < 4; ++e) csum[e] += real ? v[e] : 0.f;
    }
  }
  }
  if (p.colsum) {
    // lanes with equal (lane % LPR) hold different rows of the same 4 columns: fold them (fixed order)
#pragma unroll
    for (int d = 32; d >= LPR; d >>= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) csum[e] += __shfl_xor(csum[e], d);
    if (lane < LPR && nok)
      *reinterpret_cast<f32x4 *>(p.colsum + ((size_t)z * (gridDim.x / tiles_n) * WAVES_M + (size_t)m_tile * WAVES_M + wm) * p.Cout + n) = csum;
  }
  DF_TRACE(3);
}

// ------------------------------------------------------------------------------------------------
// v4 -- the throughput kernel (every grid of two or more workgroups per CU: > 99 % of the inference step's GEMM time): the v1
// tiling, with latency hidden by OCCUPANCY rather than by a software pipeline inside the wave.
//   * every global access is a bounds-checked buffer load (out-of-range -> 0, no branches): zero padding, ragged edges
//     and the k tail cost nothing and the loop body is one basic block;
//   * ONE k-tile buffer in LDS (36 KB for 128x128x32) and no fragment double-buffering: <= 128 VGPRs + the accumulators,
//     so OCC workgroups (4 / 5 / 6 for the 128x128 / 128x64 / 64x64 tile) share a CU.  Per k tile a workgroup does
//     registers -> LDS, issues tile kt+1's loads, barrier, 2*NP MFMA groups fed by ds_read_b128, barrier; while it
//     waits -- at a barrier, for its loads, in its prologue or its epilogue -- the other workgroups own the matrix
//     pipes.  Measured against v2 on the full grids of the inference step (2 workgroups per CU; tools/dev/gemm_list.sh):
//     +9 % at K = 192, +5 % at K = 256, +2 % at K = 512, +1 % at K = 1024 -- the short-K launches, where prologue and
//     epilogue are a third of a workgroup's life, gain most;
//   * epilogue through LDS, one 32-row band at a time (the k-tile buffer is reused): global stores / residual loads are
//     16-byte vectors covering whole 128/256-byte row segments.
// ------------------------------------------------------------------------------------------------
// LOADER: 0 general (any geometry, K tail), 1 plain GEMM, 2 multi-tap with Cin a power of two >= BKT (a k tile lies inside one tap)
// COLSUM: the launches with a fused column sum (p.colsum; p.out may be null) -- their own instantiation, so that the others carry no
// per-pass tests for it.
// MULTI: several crop-size buckets in ONE launch (the training step's direct k x k convolutions over a window of mixed crop sizes).
// The buckets' pixel rows are concatenated in p.in / p.out / p.res; the linear workgroup index selects a bucket (`tab`, by value in the
// kernel arguments), whose map size, tile counts, division magics and row offsets replace the launch-wide ones; everything after that
// is the single-geometry kernel.  Tiles never straddle buckets (a bucket's rows are tiled on their own).
struct ConvBuckets {
  int n;
  int tile0[CONV_MAX_BUCKETS + 1];      // first linear workgroup of bucket g
  int B[CONV_MAX_BUCKETS], H[CONV_MAX_BUCKETS], W[CONV_MAX_BUCKETS], OH[CONV_MAX_BUCKETS], OW[CONV_MAX_BUCKETS];
  int tiles_m[CONV_MAX_BUCKETS], tile_full[CONV_MAX_BUCKETS], full_sh[CONV_MAX_BUCKETS], ohw_sh[CONV_MAX_BUCKETS], ow_sh[CONV_MAX_BUCKETS];
  unsigned full_magic[CONV_MAX_BUCKETS], ohw_magic[CONV_MAX_BUCKETS], ow_magic[CONV_MAX_BUCKETS];
  long in_row0[CONV_MAX_BUCKETS], out_row0[CONV_MAX_BUCKETS];
};

template <int BM, int BN, int WAVES_M, int WAVES_N, int BKT, int OCC, int LOADER, bool COLSUM, bool MULTI>
__device__ __forceinline__ void igemm_f32_v4_body(const ConvParams &pk, const ConvBuckets *tab) {
  constexpr bool PURE = LOADER == 1, TAPU = LOADER == 2;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int VPR = BKT / 4;                  // 16-byte vectors per tile row
  constexpr int RPP = 256 / VPR;                // tile rows staged per pass of the 256 threads
  constexpr int A_ROWS = BM / RPP, B_ROWS = BN / RPP;
  constexpr int LDK = BKT + 4;                  // padded row: ds_read_b128 of 16 rows hits 64 distinct banks
  constexpr int BK = BKT;
  constexpr int NP = BKT / 16;                  // pairs of 8-wide k groups per tile
  constexpr int TILE = (BM + BN) * LDK;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  int wgid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  ConvParams pl;
  if constexpr (MULTI) {
    pl = pk;
    int g = 0;
    while (g + 1 < tab->n && wgid >= tab->tile0[g + 1]) ++g;      // workgroup-uniform: scalar loads of the table
    wgid -= tab->tile0[g];
    pl.B = tab->B[g]; pl.H = tab->H[g]; pl.W = tab->W[g]; pl.OH = tab->OH[g]; pl.OW = tab->OW[g];
    pl.tiles_m = tab->tiles_m[g]; pl.tile_full = tab->tile_full[g];
    pl.full_magic = tab->full_magic[g]; pl.full_sh = tab->full_sh[g];
    pl.ohw_magic = tab->ohw_magic[g]; pl.ohw_sh = tab->ohw_sh[g]; pl.ow_magic = tab->ow_magic[g]; pl.ow_sh = tab->ow_sh[g];
    pl.in = pk.in + tab->in_row0[g] * pk.in_ld;
    pl.out = pk.out + tab->out_row0[g] * pk.out_ld;
    if (pk.res) pl.res = pk.res + tab->out_row0[g] * pk.res_ld;
  }
  const ConvParams &p = MULTI ? pl : pk;
  const int M = p.B * p.OH * p.OW;
  const int K = p.KH * p.KW * p.Cin;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  // column-tile groups: when the weight operand is too big for an XCD's L2 (4 MB), the tiles are walked group by group
  // (p.ngroup column tiles, all row blocks, next group ...) so that the group's weight slice stays L2-resident while the
  // activation rows stream through once per group instead of every column tile missing on both operands
  // (divisions by host-made magics: the prologue is issue time the SIMD's other waves lose)
  int n_tile, m_tile;
  const int tiles_m = p.tiles_m;
  {
    const int GN = p.tile_gn, full = p.tile_full;
    const int g = fdiv(wgid, p.full_magic, p.full_sh, full), rem = wgid - g * full;
    const int gw = min(GN, tiles_n - g * GN);
    m_tile = gw == GN ? fdiv(rem, p.gn_magic, p.gn_sh, GN) : fdiv(rem, p.gl_magic, p.gl_sh, gw);
    n_tile = g * GN + (rem - m_tile * gw);
  }
  const int m0 = m_tile * BM, n0 = n_tile * BN;
  const int z = blockIdx.z;
  const int out_coff = p.out_coff + (int)(z * p.z_out_coff);

  // buffer descriptors (wave-uniform): reads past num_records return 0
  // (one descriptor per z slice: a z-batched launch may span more than the 4 GB a descriptor can address)
  // (TAPU: the descriptor starts `in_shift` elements BEFORE the tensor, so that the offset of a window origin in the padding above /
  // left of the image is non-negative; nothing is ever read there -- such taps are masked)
  const int in_shift = TAPU ? (p.pad * p.W + p.pad) * p.in_ld : 0;
  const unsigned in_bytes = (unsigned)(((size_t)p.B * p.H * p.W * p.in_ld + in_shift) * sizeof(float));
  const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in + (size_t)z * p.z_in_coff) - in_shift, 0, in_bytes, 0x00020000);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wgt + (size_t)z * p.z_wgt), 0,
                                                      (unsigned)((size_t)p.Cout * K * sizeof(float)), 0x00020000);
  const int in_c0 = p.in_coff;

  const int vec = tid % VPR, lrow = tid / VPR;
  // per staged row: (iy0, ix0) of the output pixel's window origin packed as two signed 16-bit halves (H, W < 2^15 - pad:
  // host-checked) -- the register budget of 4 workgroups per CU is exact -- and a_off, the element offset of (b, iy0, ix0, first channel)
  // PURE (single tap, stride 1, no padding, K % BKT == 0 -- a plain GEMM: row m is pixel m, column k its channel k; 85 % of the
  // step's GEMM time): the lane part of every load address is fixed and the k tile advances through the instruction's scalar
  // offset, so the loop carries NO vector arithmetic besides the MFMAs (the general loader spends ~65 VALU instructions per k
  // tile and wave on tap decode, bounds and selects, as many as it issues MFMAs).  Rows past M / Cout read row 0 instead:
  // their products only reach outputs that are never stored.
  int a_yx[A_ROWS], a_off[A_ROWS];
#pragma unroll
  for (int i = 0; i < A_ROWS; ++i) {
    const int m = m0 + lrow + RPP * i;
    if constexpr (PURE) {
      a_yx[i] = 0;
      a_off[i] = (int)((unsigned)((m < M ? m : 0) * p.in_ld + in_c0 + vec * 4) * 4u);          // bytes
    } else if constexpr (TAPU) {
      // a_yx: bit t set <=> tap t of this row's window lies inside the image (all clear for rows past M);
      // a_off: byte offset of (b, iy0, ix0, first channel + this lane's vector) from the shifted descriptor base
      unsigned mask = 0;
      int off = 0;
      if (m < M) {
        const int ohw = p.OH * p.OW;
        const int b = fdiv(m, p.ohw_magic, p.ohw_sh, ohw), rem = m - b * ohw;
        const int oy = fdiv(rem, p.ow_magic, p.ow_sh, p.OW), ox = rem - oy * p.OW;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        unsigned cols = 0;
        for (int kx = 0; kx < p.KW; ++kx) cols |= (unsigned)((unsigned)(ix0 + kx * p.dil) < (unsigned)p.W) << kx;
        for (int ky = 0; ky < p.KH; ++ky) mask |= ((unsigned)(iy0 + ky * p.dil) < (unsigned)p.H) ? cols << (ky * p.KW) : 0u;
        off = (((b * p.H + iy0) * p.W + ix0) * p.in_ld + in_shift + in_c0 + vec * 4) * 4;
      }
      a_yx[i] = (int)mask;
      a_off[i] = off;
    } else if (m < M) {
      const int ohw = p.OH * p.OW;
      const int b = fdiv(m, p.ohw_magic, p.ohw_sh, ohw), rem = m - b * ohw;
      const int oy = fdiv(rem, p.ow_magic, p.ow_sh, p.OW), ox = rem - oy * p.OW;
      const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
      a_yx[i] = (int)(((unsigned)iy0 << 16) | ((unsigned)ix0 & 0xffffu));
      a_off[i] = ((b * p.H + iy0) * p.W + ix0) * p.in_ld + in_c0;
    } else {
      a_yx[i] = (int)0x80000000u;      // iy0 = -32768: never inside the image
      a_off[i] = 0;
    }
  }
  int b_off[B_ROWS];
#pragma unroll
  for (int i = 0; i < B_ROWS; ++i) {
    const int n = n0 + lrow + RPP * i;
    if constexpr (PURE || TAPU) b_off[i] = (int)((unsigned)((n < p.Cout ? n : 0) * K + vec * 4) * 4u);      // bytes
    else b_off[i] = n < p.Cout ? n * K : -1;
  }
  const bool one_tap = (p.KH * p.KW == 1);
  const int cin_shift = 31 - __builtin_clz(p.Cin);
  const int nkt = (K + BK - 1) / BK;

  // branch-free k -> (tap, channel) decode: single-tap layers use shift 31 / mask ~0 (tap = 0, c = k)
  const int k_shift = one_tap ? 31 : cin_shift;
  const int c_mask = one_tap ? 0x7fffffff : p.Cin - 1;
  const int kw_magic = (65536 + p.KW - 1) / p.KW;     // tap / KW == (tap * kw_magic) >> 16 for tap < 64

  u32x4 ra[A_ROWS], rb[B_ROWS];
  auto issue_loads = [&](int kt) {
    if constexpr (PURE) {
      const int soff = kt * (BK * (int)sizeof(float));      // wave-uniform: the instruction's scalar offset
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (unsigned)a_off[i], soff, 0);
#pragma unroll
      for (int i = 0; i < B_ROWS; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)b_off[i], soff, 0);
      return;
    }
    if constexpr (TAPU) {
      // the whole k tile belongs to one tap: tap, its (dy, dx) displacement and the channel base are scalars and travel in the
      // instruction's scalar offset; per row one bit test selects the fixed lane offset or an out-of-range one (-> zeros)
      const int k0 = kt * BK;
      const int tap = k0 >> cin_shift;
      const int ky = (tap * kw_magic) >> 16, kx = tap - ky * p.KW;
      const int soff = (((ky * p.W + kx) * p.dil) * p.in_ld + (k0 & (p.Cin - 1))) * (int)sizeof(float);
      const unsigned bit = k0 < K ? 1u << tap : 0u;      // (0: the prefetch one tile past the end)
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i) {
        unsigned off = ((unsigned)a_yx[i] & bit) ? (unsigned)a_off[i] : 0x80000000u;      // num_records < 2^31: host-checked
        asm("" : "+v"(off));      // opaque: keeps hipcc from turning the select into two branchy loads
        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, soff, 0);
      }
#pragma unroll
      for (int i = 0; i < B_ROWS; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)b_off[i], k0 * (int)sizeof(float), 0);
      return;
    }
    // the tile index goes through an opaque asm so that the address arithmetic below cannot be strength-reduced
    // into loop-header induction updates: it has to stay here, between the MFMAs, where its issue slots are free
    asm volatile("" : "+s"(kt));
    const int k = kt * BK + vec * 4;
    const int kok = k < K;
    const int tap = k >> k_shift;
    const int c = k & c_mask;
    const int ky = (tap * kw_magic) >> 16, kx = tap - ky * p.KW;
    const int dy = ky * p.dil, dx = kx * p.dil;
    const int doff = (dy * p.W + dx) * p.in_ld + c;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      const int iy = (a_yx[i] >> 16) + dy, ix = (int)(short)a_yx[i] + dx;
      // bitwise (not short-circuit) so no control flow is generated: the loop body stays one basic block
      const int ok = kok & (int)((unsigned)iy < (unsigned)p.H) & (int)((unsigned)ix < (unsigned)p.W);
      unsigned off = ok ? (unsigned)(a_off[i] + doff) * 4u : 0xffffffffu;
      asm("" : "+v"(off));      // opaque: keeps hipcc from turning the select into two branchy loads
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
      const int ok = kok & (int)(b_off[i] >= 0);
      unsigned off = ok ? (unsigned)(b_off[i] + k) * 4u : 0xffffffffu;
      asm("" : "+v"(off));
      rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, 0, 0);
    }
  };
  auto write_lds = [&]() {
    float *sA = smem, *sB = sA + BM * LDK;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) *reinterpret_cast<u32x4 *>(sA + (lrow + RPP * i) * LDK + vec * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) *reinterpret_cast<u32x4 *>(sB + (lrow + RPP * i) * LDK + vec * 4) = rb[i];
  };

  const int li = lane & 31, lh = lane >> 5;
  const int fa = (wm * WM + li) * LDK + lh * 4;                  // this lane's A-fragment base (floats)
  const int fb = BM * LDK + (wn * WN + li) * LDK + lh * 4;
  // fragments of two 8-wide k groups: [half][tile]
  f32x4 a0[2][TM], b0[2][TN];
  auto read_frags = [&](int gpair, f32x4 (&a)[2][TM], f32x4 (&b)[2][TN]) {
    const float *base = smem;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[h][i] = *reinterpret_cast<const f32x4 *>(base + fa + i * 32 * LDK + (gpair * 2 + h) * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[h][j] = *reinterpret_cast<const f32x4 *>(base + fb + j * 32 * LDK + (gpair * 2 + h) * 8);
    }
  };

  f32x16 acc[TM][TN];
  // FIRST: the group's first MFMA of every block takes the instruction's literal-zero C operand instead of the accumulator: the first k
  // tile is peeled below, so the 64 v_mov that would clear the accumulators are never issued (every vector instruction a workgroup
  // spends outside its MFMAs is issue time the SIMD's other waves cannot use for theirs: tools/dev/mfma_coissue.hip)
  auto mfma_group = [&](const f32x4 (&a)[TM], const f32x4 (&b)[TN], auto first) {
    constexpr bool FIRST = decltype(first)::value;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (FIRST) {
            if (e == 0) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j][e], a[i][e], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
              continue;
            }
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j][e], a[i][e], acc[i][j], 0, 0, 0);
        }
  };
  // (operands swapped: the MFMA computes the TRANSPOSED 32x32 block, so lane (li, lh) ends up holding, for row li of the block, the
  // columns 8q + 4lh .. +3 in acc[4q .. 4q+3] -- four consecutive output channels of one pixel: the epilogue stages a band with 8
  // ds_write_b128 per lane instead of 32 ds_write2_b32.  Same products, same order per output element.)

  // split-K launches (p.splitk > 1): blockIdx.z selects this workgroup's range of k tiles
  const int kt_begin = p.splitk > 1 ? (int)((long)z * nkt / p.splitk) : 0;
  const int kt_end = p.splitk > 1 ? (int)((long)(z + 1) * nkt / p.splitk) : nkt;
  DF_TRACE(0);
  issue_loads(kt_begin);
  DF_TRACE(1);
  using std::integral_constant;
  auto k_tile = [&](int kt, auto first) {
    constexpr bool FIRST = decltype(first)::value;
    write_lds();                         // tile kt: registers -> LDS (waits for its loads)
    // tile kt+1 flies while tile kt multiplies.  Past the end the general loader's offsets are out of range (zeros, no traffic);
    // the scalar-offset loaders would fetch the next 128 bytes of every row (+6 % reads at K = 512, +17 % at K = 192): skipped
    if (LOADER == 0 || kt + 1 < kt_end) issue_loads(kt + 1);
    __syncthreads();
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      read_frags(pp, a0, b0);
      if (FIRST && pp == 0) mfma_group(a0[0], b0[0], integral_constant<bool, true>{});
      else mfma_group(a0[0], b0[0], integral_constant<bool, false>{});
      mfma_group(a0[1], b0[1], integral_constant<bool, false>{});
    }
    __syncthreads();                     // every wave has read tile kt before tile kt+1 overwrites it
  };
  k_tile(kt_begin, integral_constant<bool, true>{});          // (there is always a first tile)
  for (int kt = kt_begin + 1; kt < kt_end; ++kt) k_tile(kt, integral_constant<bool, false>{});
  DF_TRACE(2);

  // ---- epilogue: accumulators -> LDS (per-wave region) -> 16-byte row segments, one 32-row band of the wave tile at a time (the
  // single 36 KB buffer holds 4 waves x 32 rows x (WN + 4) floats).
  // Next to three workgroups in their main loops a wave gets roughly one VALU issue per MFMA of theirs (tools/dev/igemm_trace.hip:
  // the epilogue's cycles follow its VALU instruction count, 68 k cycles for ~1000 instructions against 2.6 k alone on the CU), so this
  // code is written for few vector instructions: stores and residual loads are buffer accesses off descriptors based at the tile's
  // first row -- the per-pass row offset is a scalar (soffset), the lane part is computed once, rows past M fall outside num_records
  // and columns past Cout get a poisoned offset: no per-pass address arithmetic, compares or exec masking.
  constexpr int EP_LD = WN + 4;
  static_assert(4 * 32 * EP_LD <= TILE, "epilogue staging must fit the k-tile buffer");
  float *ep = smem + wave * (32 * EP_LD);
  const float slope = (p.act == ACT_PRELU) ? p.prelu[0] : 0.f;
  const int grp = (p.rows_per_group > 0) ? fdiv(m0, p.rpg_magic, p.rpg_sh, p.rows_per_group) : 0;
  const float *bias = p.bias ? p.bias + z * p.z_bias + (p.bias_group_ld > 0 ? (size_t)grp * p.bias_group_ld : 0) : nullptr;
  constexpr int LPR = WN / 4;            // lanes per row
  constexpr int ERPP = 64 / LPR;         // rows per pass
  const int c4 = (lane % LPR) * 4, r0 = lane / LPR;
  const int n = n0 + wn * WN + c4;
  const bool nok = n < p.Cout;           // Cout % 4 == 0 (host-checked): a vector is all-in or all-out
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias && nok) bv = *reinterpret_cast<const f32x4 *>(bias + n);
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  const int valid_rows = p.rows_per_group > 0 ? p.rows_valid - (m0 - grp * p.rows_per_group) : (1 << 30);
  constexpr int BPS = 32 / ERPP;         // passes per 32-row band
  const int rows_left = min(M - m0, BM);                  // wave-uniform
  // output stores carry the non-temporal hint: a tile's 64 KB of results are not read again by this launch, and without the hint
  // they push the A rows that sibling column tiles are about to re-read out of the 4 MB L2 (counter reads -4 % over the step's
  // launches, -21 % on the Winograd-domain GEMMs; 34.9 -> 34.4 ms of GEMM time per step)
  constexpr unsigned POISON = 0x80000000u;                // beyond every num_records below (< 2^31: host-checked)
  const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out ? p.out + (size_t)m0 * p.out_ld + out_coff : nullptr, 0,
                                                        p.out ? (unsigned)(((size_t)(rows_left - 1) * p.out_ld + p.Cout) * sizeof(float)) : 0u,
                                                        0x00020000);
  const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.res ? p.res + (size_t)m0 * p.res_ld + p.res_coff : nullptr), 0,
                                                        p.res ? (unsigned)(((size_t)(rows_left - 1) * p.res_ld + p.Cout) * sizeof(float)) : 0u,
                                                        0x00020000);
  const unsigned vo_out = nok ? (unsigned)(r0 * p.out_ld + n) * 4u : POISON;
  const unsigned vo_res = nok ? (unsigned)(r0 * p.res_ld + n) * 4u : POISON;
  const float *ep_rd = ep + r0 * EP_LD + c4;
  float *ep_wr = ep + li * EP_LD + 4 * lh;
  // one instantiation per (residual, activation) pair, picked by a scalar branch: the passes carry no selects between variants
  const int so_out_step = ERPP * p.out_ld * 4, so_res_step = ERPP * p.res_ld * 4;        // scalar byte offsets between passes
  auto bands = [&](auto has_res, auto act_kind) {
    constexpr bool HAS_RES = decltype(has_res)::value;
    constexpr int ACT = decltype(act_kind)::value;
    int so_out = wm * WM * p.out_ld * 4, so_res = wm * WM * p.res_ld * 4;                 // passes walk the wave tile's rows in order
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4 *>(ep_wr + j * 32 + 8 * q) = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      DF_TRACE(8 + 2 * i);
      const int band = wm * WM + i * 32;                     // first tile row of the band: wave-uniform
      f32x4 rres[BPS];
      if constexpr (HAS_RES) {
#pragma unroll
        for (int q = 0; q < BPS; ++q, so_res += so_res_step)
          rres[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, vo_res, so_res, 0));
      }
#pragma unroll
      for (int q = 0; q < BPS; ++q, so_out += so_out_step) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(ep_rd + q * ERPP * EP_LD);
        v += bv;
        if constexpr (HAS_RES) v += rres[q];
        if constexpr (ACT == ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        } else if constexpr (ACT == ACT_PRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
        }
        if (!COLSUM || p.out) buffer_store_b128_nt(__builtin_bit_cast(u32x4, v), rs_out, vo_out, so_out);
        if constexpr (COLSUM) {
          const bool real = nok && band + q * ERPP + r0 < min(rows_left, valid_rows);
#pragma unroll
          for (int e = 0; e < 4; ++e) csum[e] += real ? v[e] : 0.f;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();                 // the band is read out before the next one lands in the same rows
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      DF_TRACE(9 + 2 * i);
    }
  };
  if (!COLSUM && p.res) {      // (a column-sum launch never carries a residual: host-checked)
    if (p.act == ACT_RELU) bands(integral_constant<bool, true>{}, integral_constant<int, ACT_RELU>{});
    else if (p.act == ACT_PRELU) bands(integral_constant<bool, true>{}, integral_constant<int, ACT_PRELU>{});
    else bands(integral_constant<bool, true>{}, integral_constant<int, ACT_NONE>{});
  } else {
    if (p.act == ACT_RELU) bands(integral_constant<bool, false>{}, integral_constant<int, ACT_RELU>{});
    else if (p.act == ACT_PRELU) bands(integral_constant<bool, false>{}, integral_constant<int, ACT_PRELU>{});
    else bands(integral_constant<bool, false>{}, integral_constant<int, ACT_NONE>{});
  }
  if constexpr (COLSUM) {
    // lanes with equal (lane % LPR) hold different rows of the same 4 columns: fold them (fixed order)
#pragma unroll
    for (int d = 32; d >= LPR; d >>= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) csum[e] += __shfl_xor(csum[e], d);
    if (lane < LPR && nok)
      *reinterpret_cast<f32x4 *>(p.colsum + ((size_t)z * tiles_m * WAVES_M + (size_t)m_tile * WAVES_M + wm) * p.Cout + n) = csum;
  }
  DF_TRACE(3);
  DF_TRACE_WAVE_END(wave);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int BKT, int OCC, int LOADER, bool COLSUM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void igemm_f32_v4_kernel(const ConvParams p) {
  igemm_f32_v4_body<BM, BN, WAVES_M, WAVES_N, BKT, OCC, LOADER, COLSUM, false>(p, nullptr);
}
template <int BM, int BN, int WAVES_M, int WAVES_N, int BKT, int OCC, int LOADER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void igemm_f32_v4_multi_kernel(const ConvParams p, const ConvBuckets tab) {
  igemm_f32_v4_body<BM, BN, WAVES_M, WAVES_N, BKT, OCC, LOADER, false, true>(p, &tab);
}

// split-K reduce: out = act(sum_s partial[s] + bias + res), partials added in the order s = 0, 1, ... (deterministic).  thread = 4 channels
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ ws, int S, long M, int Cout, float *__restrict__ out,
                                                            int out_ld, int out_coff, const float *__restrict__ bias,
                                                            const float *__restrict__ res, int res_ld, int res_coff, int act,
                                                            const float *__restrict__ prelu) {
  const int c4n = Cout >> 2;
  const long total = M * c4n, plane = M * Cout;
  const float slope = act == ACT_PRELU ? prelu[0] : 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / c4n;
    const int c = (int)(i - m * c4n) << 2;
    f32x4 v = *reinterpret_cast<const f32x4 *>(ws + m * Cout + c);
    for (int s2 = 1; s2 < S; ++s2) v += *reinterpret_cast<const f32x4 *>(ws + s2 * plane + m * Cout + c);
    if (bias) v += *reinterpret_cast<const f32x4 *>(bias + c);
    if (res) v += *reinterpret_cast<const f32x4 *>(res + m * res_ld + res_coff + c);
    if (act == ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    } else if (act == ACT_PRELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
    }
    *reinterpret_cast<f32x4 *>(out + m * out_ld + out_coff + c) = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient: dW[n][k] += sum_m dY[m][n] * A[m][k], k = (ky,kx,c), A = im2col view of the forward input.
// A GEMM whose reduction runs over the pixels m: both operands are staged [m][channel] (16-B vectors along
// the contiguous channel axis), the MFMA takes one m pair per step with lanes along n (A operand) and along
// k (B operand) -- plain ds_read_b32, conflict-free -- and the pixel range is split over blockIdx.z; every (tile, z)
// workgroup stores its partial tile to its own slice of `dw` ([split][Cout][K]) and wgrad_reduce_kernel adds the slices in a
// fixed order (bit-reproducible gradients, no atomics).  64x64 output tile, 4 waves x one 32x32 accumulator: the small-shape
// fallback of wgrad_f32_v2_kernel (Cout < 128 or K < 128).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const ConvParams p, float *__restrict__ dw, int m_chunk) {
  constexpr int TN_ = 64, TK_ = 64, RM = 32, LDS_LD = 68;
  __shared__ __attribute__((aligned(16))) float sY[2][RM * LDS_LD];
  __shared__ __attribute__((aligned(16))) float sA[2][RM * LDS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = p.B * p.OH * p.OW, K = p.KH * p.KW * p.Cin;
  const int tiles_k = (K + TK_ - 1) / TK_;
  const int n0 = (blockIdx.x / tiles_k) * TN_, k0 = (blockIdx.x % tiles_k) * TK_;
  const int m_begin = blockIdx.z * m_chunk, m_end = min(M, m_begin + m_chunk);
  const float *__restrict__ x = p.in + p.in_coff;
  const float *__restrict__ dy = p.out + p.out_coff;

  const int vec = tid & 15, lrow = tid >> 4;           // 16 float4 per 64-wide row, 16 rows per pass
  // this thread's k vector is fixed for the whole kernel: decode its tap once
  const int kcol = k0 + vec * 4;
  const bool kok = kcol < K;
  int c = kcol, dyy = 0, dxx = 0;
  if (p.KH * p.KW > 1) {
    const int tap = kcol / p.Cin;
    c = kcol - tap * p.Cin;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    dyy = ky * p.dil; dxx = kx * p.dil;
  }
  const int ncol = n0 + vec * 4;
  const bool nok = ncol < p.Cout;
  const int ohw = p.OH * p.OW;

  f32x4 ry[2], ra[2];
  auto load_chunk = [&](int mb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mb + lrow + 16 * i;
      f32x4 vy = {0.f, 0.f, 0.f, 0.f}, va = {0.f, 0.f, 0.f, 0.f};
      if (m < m_end) {
        if (nok) vy = *reinterpret_cast<const f32x4 *>(dy + (size_t)m * p.out_ld + ncol);
        const int b = m / ohw, rem = m - b * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        const int iy = oy * p.stride - p.pad + dyy, ix = ox * p.stride - p.pad + dxx;
        if (kok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          va = *reinterpret_cast<const f32x4 *>(x + ((size_t)(b * p.H + iy) * p.W + ix) * p.in_ld + c);
      }
      ry[i] = vy; ra[i] = va;
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<f32x4 *>(&sY[buf][(lrow + 16 * i) * LDS_LD + vec * 4]) = ry[i];
      *reinterpret_cast<f32x4 *>(&sA[buf][(lrow + 16 * i) * LDS_LD + vec * 4]) = ra[i];
    }
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int li = lane & 31, lh = lane >> 5;
  const int nch = (m_end - m_begin + RM - 1) / RM;
  if (nch > 0) {
    load_chunk(m_begin);
    store_chunk(0);
  }
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nch) load_chunk(m_begin + (ch + 1) * RM);
    const float *py = &sY[buf][lh * LDS_LD + wm * 32 + li];
    const float *pa = &sA[buf][lh * LDS_LD + wn * 32 + li];
#pragma unroll
    for (int kk = 0; kk < RM / 2; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(py[kk * 2 * LDS_LD], pa[kk * 2 * LDS_LD], acc, 0, 0, 0);
    if (ch + 1 < nch) store_chunk(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int n = n0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh, k = k0 + wn * 32 + li;
    if (n < p.Cout && k < K) dw[((size_t)blockIdx.z * p.Cout + n) * K + k] = acc[e];
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient, large shapes: 128 (n) x 128 (k) output tile, 4 waves x (2 x 2) 32x32 accumulators, pixel chunks of 32
// through one LDS tile with the next chunk's global loads in flight in registers (the forward kernel's pipeline, with the
// contraction over the pixel axis).  Both operands are staged as [pixel][channel] rows (16-byte global loads along the
// contiguous channel axis, ds_write_b128); the MFMA takes two pixels per step with lanes along n (A operand, dY) and along k
// (B operand, the im2col view of x): fragment reads are ds_read(2)_b32 over 32 consecutive floats of a row.  The n side of
// the tile is 128 or 64 (Cout = 64 layers: up_2, up_3, layer1).  Every out-of-range element
// (pixels beyond the chunk, padding taps, ragged Cout / K) is a bounds-checked buffer load that returns 0.
// ------------------------------------------------------------------------------------------------
// Several crop-size buckets in one launch (WgTab, by value): the pixel rows of the buckets are concatenated in p.in / p.out, bucket g is
// tab.in_rows[g] input pixels from row tab.in_row0[g] and tab.M[g] output pixels from row tab.out_row0[g].  The contraction axis is cut
// into STEPS of 32 pixels that never straddle buckets (a bucket's last step is ragged: its missing rows load zeros); the steps of all
// buckets form one sequence (bucket g = steps step0[g] .. step0[g + 1]) that is cut into equal chunks of `steps_per_chunk`, one per
// workgroup -- so a window of mixed crop sizes costs what the same pixels in one size would.  A workgroup walks its steps with the
// current bucket's geometry in scalar registers and reloads it when a step crosses into the next bucket.
struct WgTab {
  int n;
  int step0[WGRAD_MAX_SEGS + 1];
  int H[WGRAD_MAX_SEGS], W[WGRAD_MAX_SEGS], OW[WGRAD_MAX_SEGS], ohw[WGRAD_MAX_SEGS], M[WGRAD_MAX_SEGS];
  unsigned ohw_magic[WGRAD_MAX_SEGS], ow_magic[WGRAD_MAX_SEGS];
  int ohw_sh[WGRAD_MAX_SEGS], ow_sh[WGRAD_MAX_SEGS];
  unsigned in_off[WGRAD_MAX_SEGS], out_off[WGRAD_MAX_SEGS];      // byte offsets of the bucket's first input / output row
};

// Workgroups per CU of the big kernel (144 registers: no spill).  Measured on the training step's 18 shapes (tools/dev/wgrad_shapes.py) with
// the pixel split sized for one round: 2 / 3 / 4 per CU = 76.0 / 78.2 / 76.4 TFLOP/s in total -- occupancy is not what bounds it: per
// workgroup the main loop keeps the matrix pipe 95 % busy (tools/dev/wgrad_trace.hip); what is left is the prologue's first loads, the
// 64 KB partial tile every workgroup stores and the reduce launch behind it.
constexpr int WGRAD_OCC = 3;
// The bias gradient (column sums of dY): stage 1 = bias_grad_kernel, stage 2 as extra workgroups of the slice reduction's launch (or
// bias_reduce_kernel when there is no slice reduction).
struct BiasJob { const float *dy; float *part; int M, C, ld, coff, rows, ncx, nblk; };
// part[by][c] = sum over row block `by` of dY[m][c]: 4 row-interleaved partial sums per channel, added in a fixed order
__device__ __forceinline__ void bias_grad_body(const BiasJob &j, int cx, int by) {
  __shared__ float s_b[4][64];
  const int c = cx * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  const int r0 = by * j.rows, r1 = min(j.M, r0 + j.rows);
  float a = 0.f;
  if (c < j.C) {
#pragma unroll 8
    for (int r = r0 + part; r < r1; r += 4) a += j.dy[(size_t)r * j.ld + j.coff + c];      // (loads ahead, the additions in row order)
  }
  s_b[part][threadIdx.x & 63] = a;
  __syncthreads();
  if (part == 0 && c < j.C) j.part[(size_t)by * j.C + c] = (s_b[0][threadIdx.x] + s_b[1][threadIdx.x]) + (s_b[2][threadIdx.x] + s_b[3][threadIdx.x]);
}
// db[c] = sum_b part[b][c]: 8 lanes per channel add b = l, l + 8, ... in ascending order, then meet in LDS in lane order
__device__ __forceinline__ void bias_reduce_body(const float *__restrict__ part, float *__restrict__ db, int C, int nblk, int accumulate, int bx) {
  __shared__ float s_r[8][32];
  const int col = threadIdx.x & 31, zl = threadIdx.x >> 5;
  const int c = bx * 32 + col;
  float a = 0.f;
  if (c < C)
    for (int b = zl; b < nblk; b += 8) a += part[(size_t)b * C + c];
  s_r[zl][col] = a;
  __syncthreads();
  if (zl == 0 && c < C) {
#pragma unroll
    for (int l = 1; l < 8; ++l) a += s_r[l][col];
    db[c] = accumulate ? db[c] + a : a;
  }
}

template <int TN_, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void wgrad_f32_v2_kernel(const ConvParams p, float *__restrict__ part, int steps_per_chunk, int split,
                                                                                                          unsigned in_bytes, unsigned out_bytes, const WgTab tab) {
  static_assert(TN_ == 64 || TN_ == 128, "n side of the tile");
  constexpr int TK_ = 128, RM = 32;
  constexpr int LDY = TN_ + 4, LDA = TK_ + 4;        // row strides: the second pixel of a step lands 4 banks further (2-way at worst)
  constexpr int OPY = RM * LDY;                      // the dY tile (floats); the A tile follows it
  constexpr int NI = TN_ / 64;                       // 32x32 accumulators per wave along n
  constexpr int VY = TN_ / 4, RPY = 256 / VY, PY = RM / RPY;      // dY loader: vectors per row, rows per pass, passes
  extern __shared__ __attribute__((aligned(16))) float smem[];      // [dY tile | A tile]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  // XCD-aware order of the (tile, pixel chunk) pairs (1-D grid; the hardware deals workgroups to the 8 XCDs round-robin): an XCD gets a
  // contiguous run of the sequence below, i.e. output tiles that share their dY column block and -- taps fastest -- the channel block of
  // x they contract with, for all pixel chunks: its L2 (4 MB) then serves the nine taps' re-reads of the same activation rows and the
  // column tiles' re-reads of dY.  (With tiles dealt round-robin every XCD touched every k block: ~6x the traffic behind the L2.)
  int wgid;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int K = p.KH * p.KW * p.Cin;
  const int tiles_k = (K + TK_ - 1) / TK_;
  const int tile_o = wgid / split, zz = wgid - tile_o * split;
  int n_blk = tile_o / tiles_k, kt = tile_o - n_blk * tiles_k;
  if (p.KH * p.KW > 1 && p.Cin % TK_ == 0) {      // k tile = (tap, channel block): walk the taps of a channel block first
    const int T = p.KH * p.KW, cb = kt / T, tap = kt - cb * T;
    kt = tap * (p.Cin / TK_) + cb;
  }
  const int n0 = n_blk * TN_, k0 = kt * TK_;
  const int step_begin = zz * steps_per_chunk, step_end = min(tab.step0[tab.n], step_begin + steps_per_chunk);
  const int nt = step_end - step_begin;
  // the current bucket's geometry (workgroup-uniform: scalar registers), reloaded when a step enters the next bucket
  int sg = 0;
  while (sg + 1 < tab.n && step_begin >= tab.step0[sg + 1]) ++sg;
  int M, sH, sW, sOW, ohw, ohw_sh, ow_sh, seg_step0, seg_step1;
  unsigned ohw_magic, ow_magic, so_x, so_y;
  auto load_seg = [&]() {
    M = tab.M[sg]; sH = tab.H[sg]; sW = tab.W[sg]; sOW = tab.OW[sg]; ohw = tab.ohw[sg];
    ohw_magic = tab.ohw_magic[sg]; ow_magic = tab.ow_magic[sg]; ohw_sh = tab.ohw_sh[sg]; ow_sh = tab.ow_sh[sg];
    so_x = tab.in_off[sg]; so_y = tab.out_off[sg];
    seg_step0 = tab.step0[sg]; seg_step1 = tab.step0[sg + 1];
  };
  load_seg();

  // one descriptor per operand over all buckets; a bucket's base travels in the loads' scalar offset (total bytes < 4 GB: host-checked)
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in), 0, in_bytes, 0x00020000);
  const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, out_bytes, 0x00020000);

  // loader of the A tile: this thread stages channel vector `vec` (4 floats) of the rows lrow + 8 i
  const int vec = tid & 31, lrow = tid >> 5;
  const int kcol = k0 + vec * 4;                     // fixed for the whole kernel: decode its tap once
  const bool kok = kcol < K;
  int c = kcol, dyy = 0, dxx = 0;
  if (p.KH * p.KW > 1) {
    const int tap = kcol / p.Cin;
    c = kcol - tap * p.Cin;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    dyy = ky * p.dil; dxx = kx * p.dil;
  }
  // loader of the dY tile
  const int yvec = tid % VY, yrow = tid / VY;
  const int ncol = n0 + yvec * 4;
  const bool nok = ncol < p.Cout;
  u32x4 ry[PY], ra[4];
  auto issue_loads = [&](int t) {
    const int st = step_begin + t;                      // (steps are requested in increasing order)
    const bool live = st < step_end;
    if (live && st >= seg_step1) {
      while (sg + 1 < tab.n && st >= tab.step0[sg + 1]) ++sg;
      load_seg();
    }
    const int m0 = (st - seg_step0) * RM;               // first pixel of the step inside its bucket
#pragma unroll
    for (int i = 0; i < PY; ++i) {
      const int m = m0 + yrow + RPY * i;
      unsigned o = (live && m < M && nok) ? (unsigned)(m * p.out_ld + p.out_coff + ncol) * 4u : 0xffffffffu;
      asm("" : "+v"(o));
      ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, o, (int)so_y, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + lrow + 8 * i;
      const int b = fdiv(m, ohw_magic, ohw_sh, ohw), rem = m - b * ohw;
      const int oy = fdiv(rem, ow_magic, ow_sh, sOW), ox = rem - oy * sOW;
      const int iy = oy * p.stride - p.pad + dyy, ix = ox * p.stride - p.pad + dxx;
      const bool aok = live && m < M && kok && (unsigned)iy < (unsigned)sH && (unsigned)ix < (unsigned)sW;
      unsigned o = aok ? (unsigned)(((b * sH + iy) * sW + ix) * p.in_ld + p.in_coff + c) * 4u : 0xffffffffu;
      asm("" : "+v"(o));
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, o, (int)so_x, 0);
    }
  };
  auto write_lds = [&]() {
    float *sY = smem, *sA = sY + OPY;
#pragma unroll
    for (int i = 0; i < PY; ++i) *reinterpret_cast<u32x4 *>(sY + (yrow + RPY * i) * LDY + yvec * 4) = ry[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4 *>(sA + (lrow + 8 * i) * LDA + vec * 4) = ra[i];
  };
  f32x16 acc[NI][2];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int li = lane & 31, lh = lane >> 5;
  const int fy = lh * LDY + wn * (TN_ / 2) + li, fa = OPY + lh * LDA + wk * 64 + li;
  DF_WTRACE(0);
  if (nt > 0) {
    issue_loads(0);
    write_lds();
    issue_loads(1);
  }
  __syncthreads();
  DF_WTRACE(1);
  for (int t = 0; t < nt; ++t) {
    const float *base = smem;
    // fragments of pixel pair s + 1 are fetched (into their own registers) before pair s multiplies: with one register set the
    // compiler serialised read -> wait -> 4 MFMAs -> read, and every pair paid the LDS latency
    float fa_[2][NI], fb_[2][2];
    auto frag = [&](int s, float (&a)[NI], float (&b)[2]) {
#pragma unroll
      for (int i = 0; i < NI; ++i) a[i] = base[fy + s * 2 * LDY + i * 32];
      b[0] = base[fa + s * 2 * LDA]; b[1] = base[fa + s * 2 * LDA + 32];
    };
    frag(0, fa_[0], fb_[0]);
#pragma unroll
    for (int s = 0; s < RM / 2; ++s) {
      if (s + 1 < RM / 2) frag(s + 1, fa_[(s + 1) & 1], fb_[(s + 1) & 1]);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[s & 1][i], fb_[s & 1][0], acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[s & 1][i], fb_[s & 1][1], acc[i][1], 0, 0, 0);
      }
      // keep the order written here: the next pair's LDS reads are issued, THEN this pair's MFMAs (hipcc otherwise sinks the reads
      // behind the MFMAs and reuses one register set)
      if (s + 1 < RM / 2) __builtin_amdgcn_sched_group_barrier(0x100, NI == 2 ? 2 : 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NI, 0);
    }
    // ONE LDS tile per workgroup (4 workgroups per CU instead of the 2 a double buffer allows: the other three cover this hand-over);
    // the next chunk has been in flight in registers for the whole tile
    __syncthreads();
    if (t + 1 < nt) {
      write_lds();
      issue_loads(t + 2);
    }
    __syncthreads();
  }
  DF_WTRACE(2);
  float *dst = part + (size_t)zz * p.Cout * K;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + wn * (TN_ / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh, k = k0 + wk * 64 + j * 32 + li;
        if (n < p.Cout && k < K) dst[(size_t)n * K + k] = acc[i][j][e];
      }
  DF_WTRACE(3);
}

// dw[i] = sum_z part[z][i], bit-reproducible: 8 z-lanes per output vector each add their slices z = l, l + 8, ... in ascending
// order (8 independent load chains instead of one of length `split`), then the 8 partial sums meet in LDS and are added in
// lane order.  32 float4 outputs per workgroup.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dw, long n4, int split, int accumulate, int red_blocks,
                                                           const float *__restrict__ bpart, float *__restrict__ db, int bC, int bnblk) {
  if ((int)blockIdx.x >= red_blocks) {      // the bias gradient's second stage rides along (workgroups past the slice reduction's)
    bias_reduce_body(bpart, db, bC, bnblk, accumulate, (int)blockIdx.x - red_blocks);
    return;
  }
  __shared__ f32x4 s_p[8][32];
  const int col = threadIdx.x & 31, zl = threadIdx.x >> 5;
  for (long i0 = blockIdx.x * 32L; i0 < n4; i0 += (long)red_blocks * 32) {
    const long i = i0 + col;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (i < n4)
      for (int z = zl; z < split; z += 8) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(part)[(long)z * n4 + i];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += v[e];
      }
    s_p[zl][col] = a;
    __syncthreads();
    if (zl == 0 && i < n4) {
#pragma unroll
      for (int l = 1; l < 8; ++l)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += s_p[l][col][e];
      if (accumulate) {                      // dw += this launch's sum (gradient accumulation over passes: still one fixed order)
        const f32x4 old = reinterpret_cast<const f32x4 *>(dw)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = old[e] + a[e];
      }
      reinterpret_cast<f32x4 *>(dw)[i] = a;
    }
    __syncthreads();
  }
}

// the two stages as launches of their own (the small-shape product kernel; a gradient that needs no slice reduction)
__global__ __launch_bounds__(256) void bias_grad_kernel(const BiasJob j) { bias_grad_body(j, blockIdx.x, blockIdx.y); }
__global__ __launch_bounds__(256) void bias_reduce_kernel(const float *__restrict__ part, float *__restrict__ db, int C, int nblk, int accumulate) {
  bias_reduce_body(part, db, C, nblk, accumulate, blockIdx.x);
}


struct TileCfg { int bm, bn, wmv; };     // workgroup tile and waves along M

// Launches the product kernel (v4) takes; the rest -- input dilation (the data gradient of a strided convolution), operands of
// 4 GB or more per z slice (a buffer descriptor's reach), channel offsets / strides that are not multiples of 4 -- go to v1.
bool takes_v4(const ConvParams &p) {
  static const bool force_v1 = df::dev_getenv("DF_IGEMM_V1") != nullptr;   // dev switch: A/B against the un-pipelined kernel; read once
  const size_t in_bytes = (size_t)p.B * p.H * p.W * p.in_ld * sizeof(float);      // per z slice
  const size_t w_bytes = (size_t)p.Cout * p.KH * p.KW * p.Cin * sizeof(float);
  return !force_v1 && p.up == 1 && p.H + p.pad < 32768 && p.W + p.pad < 32768 && p.out_ld < (1 << 21) && p.res_ld < (1 << 21) && in_bytes < (1ull << 32) && w_bytes < (1ull << 32) && p.Cout % 4 == 0 && p.out_ld % 4 == 0 &&
         p.out_coff % 4 == 0 && p.z_out_coff % 4 == 0 && (!p.res || (p.res_ld % 4 == 0 && p.res_coff % 4 == 0)) &&
         (!p.bias || (p.bias_group_ld % 4 == 0 && p.z_bias % 4 == 0));
}

TileCfg pick_cfg(const ConvParams &p) {
  const long M = (long)p.B * p.OH * p.OW;
  // (Row-grouped launches -- fused mean / per-object bias -- need BM | rows_per_group; every tile divides the
  // 128-padded groups the engine uses.)  Pick the tile that minimises  ceil(tiles / 256 CUs) * tile_area / efficiency : the chip
  // finishes when its most loaded CU does, so a 128x128 grid of e.g. 800 tiles (4 rounds for 3.1 rounds
  // of work) loses to the same problem cut into 3200 64x64 tiles (13 rounds for 12.5), and Cout = 576 (4.5 column
  // tiles of 128) is cut into 9 columns of 128x64 tiles.
  // The fused column sum adds rows in per-wave groups: keep that grouping independent of the batch size (so
  // a batched call stays bit-identical to solo calls) by always using the 128x128 tile for it.
  if (p.colsum || !p.out) return {128, 128, 2};     // (!p.out: the same launch while its partial buffer is being sized)
  const bool v4 = takes_v4(p);                      // v1 has the two square tiles only
  static const char *const tile_env = df::dev_getenv("DF_IGEMM_TILE");      // dev switch for A/B runs; read once
  if (tile_env) {
    if (tile_env[0] == 'a') return {128, 128, 2};
    if (tile_env[0] == 'b' && v4) return {128, 64, 2};
    if (tile_env[0] == 'c' || tile_env[0] == 'b') return {64, 64, 2};
  }
  auto cost = [&](int bm, int bn, double eff) {
    const long tiles = ((M + bm - 1) / bm) * ((p.Cout + bn - 1) / bn) * p.zcount;
    const long rounds = (tiles + 255) / 256;
    return (double)rounds * bm * bn / eff;
  };
  // efficiencies from the per-shape table of a bench step run with each tile forced (tools/dev/gemm_list.sh)
  const double ca = p.Cout >= 128 ? cost(128, 128, 1.0) : 1e300, cb = v4 ? cost(128, 64, 0.99) : 1e300, cc = cost(64, 64, 0.96);
  if (ca <= cb && ca <= cc) return {128, 128, 2};
  if (cb <= cc) return {128, 64, 2};
  return {64, 64, 2};
}

}  // namespace

int conv_colsum_rows(const ConvParams &p) {
  const TileCfg c = pick_cfg(p);
  const long M = (long)p.B * p.OH * p.OW;
  return (int)(((M + c.bm - 1) / c.bm) * c.wmv) * p.zcount;
}

double conv_bytes(const ConvParams &p) {
  // algorithmic HBM bytes of one launch: every input element, weight and output element once (+ residual)
  const double M = (double)p.B * p.OH * p.OW;
  double b = (double)p.B * p.H * p.W * p.Cin + (double)p.Cout * p.KH * p.KW * p.Cin + (p.out ? M * p.Cout : 0.0) + (p.res ? M * p.Cout : 0.0);
  return b * p.zcount * sizeof(float);
}

double conv_flops(const ConvParams &p) {
  return 2.0 * p.B * p.OH * p.OW * (double)p.Cout * p.KH * p.KW * p.Cin * p.zcount;
}

int launch_conv(const ConvParams &p, hipStream_t st, int *splitk_used) {
  if (!p.in || !p.wgt || (!p.out && !p.colsum)) return set_error(DF_ERR_ARG, "conv: null pointer");
  if (p.Cin % 4 || p.in_ld % 4 || p.in_coff % 4 || p.z_in_coff % 4 || p.z_wgt % 4)
    return set_error(DF_ERR_ARG, "conv: Cin/in_ld/in_coff must be multiples of 4 (16-B vector loads)");
  if (p.KH * p.KW > 1 && (p.Cin & (p.Cin - 1)))
    return set_error(DF_ERR_ARG, "conv: multi-tap convolutions need a power-of-two Cin (got %d)", p.Cin);
  const long M = (long)p.B * p.OH * p.OW;
  if (M <= 0 || p.Cout <= 0) return DF_OK;
  if (M * (long)p.out_ld >= (1L << 40) || (long)p.B * p.H * p.W >= (1L << 31))
    return set_error(DF_ERR_ARG, "conv: tensor too large for 32-bit pixel indexing");
#ifdef DF_DEV
  if (!p.splitk_ws && try_split_gemm(p, st)) {
    if (splitk_used) *splitk_used = 1;
    return check_launch("split gemm");
  }
#endif
  const TileCfg c = pick_cfg(p);
  ConvParams pl = p;       // launch copy: + the column-tile group width and the division magics
  make_fdiv((long)p.OH * p.OW, pl.ohw_magic, pl.ohw_sh);
  make_fdiv(p.OW, pl.ow_magic, pl.ow_sh);
  auto set_tile_decode = [&]() {      // after ngroup is known
    const int tiles_n = (p.Cout + c.bn - 1) / c.bn;
    pl.tiles_m = (int)((M + c.bm - 1) / c.bm);
    pl.tile_gn = pl.ngroup > 0 && pl.ngroup < tiles_n ? pl.ngroup : tiles_n;
    pl.tile_full = pl.tiles_m * pl.tile_gn;
    const int last = tiles_n % pl.tile_gn;
    make_fdiv(pl.tile_full, pl.full_magic, pl.full_sh);
    make_fdiv(pl.tile_gn, pl.gn_magic, pl.gn_sh);
    make_fdiv(last ? last : pl.tile_gn, pl.gl_magic, pl.gl_sh);
    make_fdiv(p.rows_per_group > 0 ? p.rows_per_group : 1, pl.rpg_magic, pl.rpg_sh);
  };
  {
    const size_t slice = (size_t)c.bn * p.KH * p.KW * p.Cin * sizeof(float);       // weights of one column tile
    const long tn = (p.Cout + c.bn - 1) / c.bn;
    static const long budget = df::dev_getenv("DF_IGEMM_WGROUP_KB") ? atol(df::dev_getenv("DF_IGEMM_WGROUP_KB")) * 1024L : 3L << 20;
    if (budget > 0 && (size_t)tn * slice > (size_t)budget) pl.ngroup = (int)std::max<long>(1, budget / (long)slice);
  }
  set_tile_decode();
  if (p.rows_per_group > 0 && (p.rows_per_group % c.bm))
    return set_error(DF_ERR_ARG, "conv: rows_per_group must be a multiple of %d", c.bm);
  const long tiles = ((M + c.bm - 1) / c.bm) * ((p.Cout + c.bn - 1) / c.bn);
  dim3 grid((unsigned)tiles, 1, p.zcount);
  // v1's 128x128 tile double-buffers 72 KiB of dynamic LDS: above the 64 KiB default cap.  The attribute is per device (a process
  // may drive several): set once for every device a launch is seen on.  (v4's single k-tile buffer is 36 KiB at most.)
  {
    static bool attr_done[64] = {};
    int dev = 0;
    hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_done[dev]) {
      hipFuncSetAttribute(reinterpret_cast<const void *>(&igemm_f32_kernel<128, 128, 2, 2>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * LDK * (int)sizeof(float));
      hipFuncSetAttribute(reinterpret_cast<const void *>(&igemm_f32_v2_kernel<128, 128, 2, 2, 32>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 36 * (int)sizeof(float));
      attr_done[dev] = true;
    }
  }
  constexpr size_t ROW = 36 * sizeof(float);      // one padded k-tile row (BKT = 32)
  // workgroups per CU of the v4 kernel: 4 / 5 / 6 (the register budget amdgpu_waves_per_eu leaves each: 128 / 102 / 85 VGPRs + AGPRs)
  static const bool no_pure = df::dev_getenv("DF_IGEMM_NOPURE") != nullptr;      // dev switch: the general loader for every launch; read once
  const bool v4 = takes_v4(p);
  const int taps = p.KH * p.KW;
  const size_t shifted = ((size_t)p.B * p.H * p.W + (size_t)p.pad * p.W + p.pad) * p.in_ld * sizeof(float);
  // loader 1: plain GEMM; loader 2: several taps, every 32-wide k tile inside one tap, tap mask in 32 bits, offsets below 2^31
  const int loader = no_pure ? 0
                     : taps == 1 && p.stride == 1 && p.pad == 0 && p.Cin % 32 == 0 ? 1
                     : taps > 1 && taps <= 32 && p.Cin >= 32 && shifted < (1ull << 31) ? 2 : 0;     // (Cin is a power of two here)
  // split-K (opt-in scratch; training): a grid under one workgroup per CU with a long reduction
  int S = 1;
  if (v4 && p.splitk_ws && !p.colsum && p.zcount == 1 && loader != 0 && tiles < 256 && p.bias_group_ld == 0 && p.rows_per_group == 0) {
    const int nkt = (p.KH * p.KW * p.Cin) / 32;
    S = (int)std::min<long>(8, (1024 + tiles - 1) / tiles);          // one round of workgroups at 4 per CU (512: -1.2 % on the 8-frame training step)
    S = std::min(S, nkt / 4);
    while (S > 1 && (size_t)S * M * p.Cout * sizeof(float) > p.splitk_ws_bytes) --S;
    if ((long)S * M * p.Cout >= (1L << 31)) S = 1;
    if (S < 1) S = 1;
  }
  if (splitk_used) *splitk_used = S;
  if (S > 1) {
    pl.splitk = S;
    pl.out = p.splitk_ws; pl.out_ld = p.Cout; pl.out_coff = 0; pl.z_out_coff = M * p.Cout;
    pl.bias = nullptr; pl.res = nullptr; pl.act = ACT_NONE;
    grid.z = S;
  }
  // grids under two workgroups per CU: the software-pipelined kernel (training, 1 / 8 frames per pass: 142 -> 154 / 670 -> 696 frames/s)
  static const long lowocc = df::dev_getenv("DF_IGEMM_LOWOCC") ? atol(df::dev_getenv("DF_IGEMM_LOWOCC")) : 512;    // dev switch; read once
  if (v4 && S == 1 && tiles * p.zcount < lowocc && c.bn == c.bm) {
    if (c.bm == 128)
      hipLaunchKernelGGL((igemm_f32_v2_kernel<128, 128, 2, 2, 32>), grid, dim3(256), (size_t)2 * 256 * 36 * sizeof(float), st, pl);
    else
      hipLaunchKernelGGL((igemm_f32_v2_kernel<64, 64, 2, 2, 32>), grid, dim3(256), (size_t)2 * 128 * 36 * sizeof(float), st, pl);
  } else if (v4) {
    auto launch = [&](auto bm, auto bn, auto occ, size_t rows) {
      constexpr int BM_ = decltype(bm)::value, BN_ = decltype(bn)::value, OCC_ = decltype(occ)::value;
      if (loader == 1) hipLaunchKernelGGL((igemm_f32_v4_kernel<BM_, BN_, 2, 2, 32, OCC_, 1, false>), grid, dim3(256), rows * ROW, st, pl);
      else if (loader == 2) hipLaunchKernelGGL((igemm_f32_v4_kernel<BM_, BN_, 2, 2, 32, OCC_, 2, false>), grid, dim3(256), rows * ROW, st, pl);
      else hipLaunchKernelGGL((igemm_f32_v4_kernel<BM_, BN_, 2, 2, 32, OCC_, 0, false>), grid, dim3(256), rows * ROW, st, pl);
    };
    using std::integral_constant;
    if (p.colsum && p.res) return set_error(DF_ERR_ARG, "conv: a column-sum launch cannot take a residual");
    if (p.colsum) {      // always the 128x128 tile (pick_cfg); the per-point launches that use it are plain GEMMs
      if (loader == 1) hipLaunchKernelGGL((igemm_f32_v4_kernel<128, 128, 2, 2, 32, 4, 1, true>), grid, dim3(256), 256 * ROW, st, pl);
      else hipLaunchKernelGGL((igemm_f32_v4_kernel<128, 128, 2, 2, 32, 4, 0, true>), grid, dim3(256), 256 * ROW, st, pl);
    } else if (c.bm == 128 && c.bn == 128) launch(integral_constant<int, 128>{}, integral_constant<int, 128>{}, integral_constant<int, 4>{}, 256);
    else if (c.bm == 128) launch(integral_constant<int, 128>{}, integral_constant<int, 64>{}, integral_constant<int, 5>{}, 192);
    else launch(integral_constant<int, 64>{}, integral_constant<int, 64>{}, integral_constant<int, 6>{}, 128);      // (a BK = 64 form of the small tile was measured: 0.85-1.0x, dropped)
    if (S > 1) {
      const long vecs = M * (p.Cout / 4);
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long>((vecs + 255) / 256, 65535L * 8)), dim3(256), 0, st, p.splitk_ws, S, M, p.Cout,
                         p.out, p.out_ld, p.out_coff, p.bias, p.res, p.res_ld, p.res_coff, p.act, p.prelu);
    }
  } else if (c.bm == 128) {
    hipLaunchKernelGGL((igemm_f32_kernel<128, 128, 2, 2>), grid, dim3(256), (size_t)2 * 256 * LDK * sizeof(float), st, p);
  } else {
    hipLaunchKernelGGL((igemm_f32_kernel<64, 64, 2, 2>), grid, dim3(256), (size_t)2 * 128 * LDK * sizeof(float), st, p);
  }
  return check_launch("igemm");
}

// Direct k x k convolution over several crop-size buckets in one launch (igemm_f32_v4_multi_kernel); see igemm.h.
int launch_conv_multi(const ConvParams &p, int nseg, const WgradSeg *segs, hipStream_t st) {
  if (!p.in || !p.wgt || !p.out || nseg <= 0 || !segs) return set_error(DF_ERR_ARG, "conv_multi: null pointer");
  const int taps = p.KH * p.KW;
  // what the multi-bucket instantiations cover: the product kernel's general and tap-uniform loaders, no grouping, no fused sums
  const bool ok = p.up == 1 && p.zcount == 1 && !p.colsum && p.rows_per_group == 0 && p.bias_group_ld == 0 && p.Cin % 4 == 0 && p.in_ld % 4 == 0 &&
                  p.in_coff % 4 == 0 && p.Cout % 4 == 0 && p.out_ld % 4 == 0 && p.out_coff % 4 == 0 && (!p.res || (p.res_ld % 4 == 0 && p.res_coff % 4 == 0)) &&
                  (taps == 1 || !(p.Cin & (p.Cin - 1))) && p.out_ld < (1 << 21) && p.res_ld < (1 << 21);
  if (!ok || nseg == 1) {          // anything else: one launch per bucket
    for (int g = 0; g < nseg; ++g) {
      ConvParams q = p;
      q.B = segs[g].B; q.H = segs[g].H; q.W = segs[g].W; q.OH = segs[g].OH; q.OW = segs[g].OW;
      q.in = p.in + segs[g].in_row0 * p.in_ld; q.out = p.out + segs[g].out_row0 * p.out_ld;
      if (p.res) q.res = p.res + segs[g].out_row0 * p.res_ld;
      const int rc = launch_conv(q, st);
      if (rc != DF_OK) return rc;
    }
    return DF_OK;
  }
  for (int g0 = 0; g0 < nseg; g0 += CONV_MAX_BUCKETS) {
    const int n = std::min(CONV_MAX_BUCKETS, nseg - g0);
    long Mtot = 0;
    bool small_ok = true;
    for (int g = g0; g < g0 + n; ++g) {
      const WgradSeg &sg = segs[g];
      if (sg.B <= 0 || sg.H <= 0 || sg.W <= 0 || sg.OH <= 0 || sg.OW <= 0) return set_error(DF_ERR_ARG, "conv_multi: empty bucket");
      Mtot += (long)sg.B * sg.OH * sg.OW;
      const size_t in_bytes = (size_t)sg.B * sg.H * sg.W * p.in_ld * sizeof(float);
      const size_t shifted = ((size_t)sg.B * sg.H * sg.W + (size_t)p.pad * sg.W + p.pad) * p.in_ld * sizeof(float);
      small_ok = small_ok && in_bytes < (1ull << 32) && shifted < (1ull << 31) && sg.H + p.pad < 32768 && sg.W + p.pad < 32768 &&
                 (size_t)sg.B * sg.OH * sg.OW * p.out_ld < (1ull << 40);
    }
    if (!small_ok) return set_error(DF_ERR_ARG, "conv_multi: a bucket is too large for the multi-bucket kernel");
    ConvParams pt = p;             // tile choice on the whole launch's rows
    pt.B = (int)std::min<long>(Mtot, 1L << 30); pt.OH = pt.OW = 1;
    const TileCfg c = pick_cfg(pt);
    ConvParams pl = p;
    {
      const size_t slice = (size_t)c.bn * taps * p.Cin * sizeof(float);
      const long tn = (p.Cout + c.bn - 1) / c.bn;
      const long budget = 3L << 20;
      pl.ngroup = (size_t)tn * slice > (size_t)budget ? (int)std::max<long>(1, budget / (long)slice) : 0;
    }
    const int tiles_n = (p.Cout + c.bn - 1) / c.bn;
    pl.tile_gn = pl.ngroup > 0 && pl.ngroup < tiles_n ? pl.ngroup : tiles_n;
    const int last = tiles_n % pl.tile_gn;
    make_fdiv(pl.tile_gn, pl.gn_magic, pl.gn_sh);
    make_fdiv(last ? last : pl.tile_gn, pl.gl_magic, pl.gl_sh);
    make_fdiv(1, pl.rpg_magic, pl.rpg_sh);
    pl.splitk = 1;
    ConvBuckets tab;
    tab.n = n;
    tab.tile0[0] = 0;
    for (int i = 0; i < n; ++i) {
      const WgradSeg &sg = segs[g0 + i];
      const long M = (long)sg.B * sg.OH * sg.OW;
      tab.B[i] = sg.B; tab.H[i] = sg.H; tab.W[i] = sg.W; tab.OH[i] = sg.OH; tab.OW[i] = sg.OW;
      tab.tiles_m[i] = (int)((M + c.bm - 1) / c.bm);
      tab.tile_full[i] = tab.tiles_m[i] * pl.tile_gn;
      make_fdiv(tab.tile_full[i], tab.full_magic[i], tab.full_sh[i]);
      make_fdiv((long)sg.OH * sg.OW, tab.ohw_magic[i], tab.ohw_sh[i]);
      make_fdiv(sg.OW, tab.ow_magic[i], tab.ow_sh[i]);
      tab.in_row0[i] = sg.in_row0; tab.out_row0[i] = sg.out_row0;
      tab.tile0[i + 1] = tab.tile0[i] + tab.tiles_m[i] * tiles_n;
    }
    const dim3 grid((unsigned)tab.tile0[n], 1, 1);
    constexpr size_t ROW = 36 * sizeof(float);
    const int loader = taps > 1 && taps <= 32 && p.Cin >= 32 ? 2 : 0;
    auto launch = [&](auto bm, auto bn, auto occ, size_t rows) {
      constexpr int BM_ = decltype(bm)::value, BN_ = decltype(bn)::value, OCC_ = decltype(occ)::value;
      if (loader == 2) hipLaunchKernelGGL((igemm_f32_v4_multi_kernel<BM_, BN_, 2, 2, 32, OCC_, 2>), grid, dim3(256), rows * ROW, st, pl, tab);
      else hipLaunchKernelGGL((igemm_f32_v4_multi_kernel<BM_, BN_, 2, 2, 32, OCC_, 0>), grid, dim3(256), rows * ROW, st, pl, tab);
    };
    using std::integral_constant;
    if (c.bm == 128 && c.bn == 128) launch(integral_constant<int, 128>{}, integral_constant<int, 128>{}, integral_constant<int, 4>{}, 256);
    else if (c.bm == 128) launch(integral_constant<int, 128>{}, integral_constant<int, 64>{}, integral_constant<int, 5>{}, 192);
    else launch(integral_constant<int, 64>{}, integral_constant<int, 64>{}, integral_constant<int, 6>{}, 128);
  }
  return check_launch("igemm (multi-bucket)");
}

namespace {
struct WgradPlan { bool big; int tiles, split, chunk, steps, nblk, bias_rows; size_t part_floats, bias_floats; };

// big kernel: the pixel axis is a sequence of 32-pixel steps (per bucket: ceil(M_g / 32)), cut into `split` chunks of `steps` steps each;
// small kernel: chunks of `chunk` = 32 * steps pixels per bucket
WgradPlan wgrad_plan(const ConvParams &p, int nseg, const WgradSeg *segs) {
  WgradPlan w{};
  long M = 0, total_steps = 0;
  for (int g = 0; g < nseg; ++g) {
    const long Mg = (long)segs[g].B * segs[g].OH * segs[g].OW;
    M += Mg;
    total_steps += (Mg + 31) / 32;
  }
  const int K = p.KH * p.KW * p.Cin;
  w.big = p.Cout >= 64 && K >= 128;
  const int tn = w.big && p.Cout >= 128 ? 128 : 64, tk = w.big ? 128 : 64;
  w.tiles = ((p.Cout + tn - 1) / tn) * ((K + tk - 1) / tk);
  // split the pixel range so that one round of workgroups (WGRAD_OCC per CU) is in flight, each with at least 256 pixels
  long split = (256 * WGRAD_OCC) / w.tiles;
  const long min_px = w.tiles <= 16 ? 128 : 256;        // few output tiles (the per-point layers: 1 - 10): finer slices keep more CUs busy
  const long max_split = (M + min_px - 1) / min_px;
  if (split > max_split) split = max_split;
  if (split > 256) split = 256;          // (the partial slices are re-read by the reduction)
  if (split < 1) split = 1;
  w.steps = (int)((total_steps + split - 1) / split);
  w.chunk = w.steps * 32;
  if (w.big) w.split = (int)((total_steps + w.steps - 1) / w.steps);
  else {
    w.split = 0;
    for (int g = 0; g < nseg; ++g) w.split += (int)(((long)segs[g].B * segs[g].OH * segs[g].OW + w.chunk - 1) / w.chunk);
  }
  // column sums of dY (the bias gradient): row blocks of 128 rows, or of the multiple of 128 that leaves at most 256 blocks -- the second stage
  // walks the blocks 8 lanes per channel (1 600 blocks at 204 800 pixel rows took it 47 us)
  w.bias_rows = 128 * (int)(((M + 127) / 128 + 255) / 256);
  w.nblk = (int)((M + w.bias_rows - 1) / w.bias_rows);
  w.part_floats = (size_t)w.split * p.Cout * K;       // (a single slice goes straight to dw unless the launch accumulates)
  w.bias_floats = (size_t)w.nblk * p.Cout;
  return w;
}
WgradSeg single_seg(const ConvParams &p) { return WgradSeg{p.B, p.H, p.W, p.OH, p.OW, 0, 0}; }
}  // namespace

size_t wgrad_multi_workspace_bytes(const ConvParams &p, int nseg, const WgradSeg *segs) {
  size_t worst = 0;
  for (int g0 = 0; g0 < nseg; g0 += WGRAD_MAX_SEGS) {
    const WgradPlan w = wgrad_plan(p, std::min(WGRAD_MAX_SEGS, nseg - g0), segs + g0);
    worst = std::max(worst, (w.part_floats + w.bias_floats) * sizeof(float));
  }
  return worst;
}

size_t wgrad_workspace_bytes(const ConvParams &p) {
  const WgradSeg sg = single_seg(p);
  return wgrad_multi_workspace_bytes(p, 1, &sg);
}

static int launch_wgrad_segs(ConvParams p, int nseg, const WgradSeg *segs, float *dw, float *db, void *ws, size_t ws_bytes, hipStream_t st, int accumulate) {
  const int K = p.KH * p.KW * p.Cin;
  const WgradPlan w = wgrad_plan(p, nseg, segs);
  const size_t need = (w.part_floats + w.bias_floats) * sizeof(float);
  if (need > ws_bytes || (need && !ws)) return set_error(DF_ERR_WORKSPACE, "wgrad: workspace too small");
  long M = 0, out_lo = segs[0].out_row0, in_lo = segs[0].in_row0, in_hi = 0;
  for (int g = 0; g < nseg; ++g) {
    const WgradSeg &sg = segs[g];
    if (sg.B <= 0 || sg.H <= 0 || sg.W <= 0 || sg.OH <= 0 || sg.OW <= 0) return set_error(DF_ERR_ARG, "wgrad: empty bucket");
    if (sg.out_row0 != out_lo + M) return set_error(DF_ERR_ARG, "wgrad: the buckets' output rows must be contiguous and in order");
    if (sg.in_row0 < in_lo) return set_error(DF_ERR_ARG, "wgrad: the buckets' input rows must be in order");
    in_hi = std::max(in_hi, sg.in_row0 + (long)sg.B * sg.H * sg.W);
    M += (long)sg.B * sg.OH * sg.OW;
  }
  if (M >= (1L << 31)) return set_error(DF_ERR_ARG, "wgrad: too many pixels");
  const size_t in_bytes = (size_t)(in_hi - in_lo) * p.in_ld * sizeof(float), out_bytes = (size_t)M * p.out_ld * sizeof(float);
  if (in_bytes >= (1ull << 32) || out_bytes >= (1ull << 32)) return set_error(DF_ERR_ARG, "wgrad: tensor too large (4 GB per operand)");
  const bool reduce = w.split > 1 || accumulate;
  float *part = reduce ? static_cast<float *>(ws) : dw;
  float *bpart = static_cast<float *>(ws) + w.part_floats;
  // the bias gradient (dY's rows of all buckets are contiguous: one column sum over them): stage 1 is a launch of its own (as extra workgroups of
  // the product kernel it ran at that kernel's 3 workgroups per CU with 34 KB of LDS each: a latency-bound loop at low occupancy, 32-frame
  // window 1 679 -> 1 402 frames/s), stage 2 rides along in the slice reduction's launch
  BiasJob bj{};
  if (db) {
    bj.dy = p.out + out_lo * p.out_ld; bj.part = bpart; bj.M = (int)M; bj.C = p.Cout; bj.ld = p.out_ld; bj.coff = p.out_coff;
    bj.rows = w.bias_rows; bj.ncx = (p.Cout + 63) / 64; bj.nblk = w.nblk;
  }
  if (w.big) {
    WgTab tab;
    tab.n = nseg;
    tab.step0[0] = 0;
    for (int g = 0; g < nseg; ++g) {
      const WgradSeg &sg = segs[g];
      const long Mg = (long)sg.B * sg.OH * sg.OW;
      tab.H[g] = sg.H; tab.W[g] = sg.W; tab.OW[g] = sg.OW; tab.ohw[g] = sg.OH * sg.OW; tab.M[g] = (int)Mg;
      make_fdiv((long)sg.OH * sg.OW, tab.ohw_magic[g], tab.ohw_sh[g]);
      make_fdiv(sg.OW, tab.ow_magic[g], tab.ow_sh[g]);
      tab.in_off[g] = (unsigned)((size_t)(sg.in_row0 - in_lo) * p.in_ld * sizeof(float));
      tab.out_off[g] = (unsigned)((size_t)(sg.out_row0 - out_lo) * p.out_ld * sizeof(float));
      tab.step0[g + 1] = tab.step0[g] + (int)((Mg + 31) / 32);
    }
    ConvParams pk = p;                  // operand bases at the first bucket's rows (the buckets' offsets are relative to them)
    pk.in = p.in + in_lo * p.in_ld;
    pk.out = p.out + out_lo * p.out_ld;
    constexpr size_t lds128 = (size_t)32 * (132 + 132) * 4, lds64 = (size_t)32 * (68 + 132) * 4;
    auto go = [&](auto tn, size_t lds) {
      constexpr int TN = decltype(tn)::value;
      static bool attr_done[64] = {};
      int dev = 0;
      hipGetDevice(&dev);
      if (dev >= 0 && dev < 64 && !attr_done[dev]) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_f32_v2_kernel<TN, WGRAD_OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done[dev] = true;
      }
      hipLaunchKernelGGL((wgrad_f32_v2_kernel<TN, WGRAD_OCC>), dim3((unsigned)(w.tiles * w.split)), dim3(256), lds, st, pk, part, w.steps, w.split, (unsigned)in_bytes,
                         (unsigned)out_bytes, tab);
    };
    using std::integral_constant;
    if (p.Cout >= 128) go(integral_constant<int, 128>{}, lds128);
    else go(integral_constant<int, 64>{}, lds64);
  } else {
    // the small-shape kernel decodes one geometry: one launch per bucket into consecutive partial slices
    int z = 0;
    for (int g = 0; g < nseg; ++g) {
      const WgradSeg &sg = segs[g];
      ConvParams q = p;
      q.B = sg.B; q.H = sg.H; q.W = sg.W; q.OH = sg.OH; q.OW = sg.OW;
      q.in = p.in + sg.in_row0 * p.in_ld; q.out = p.out + sg.out_row0 * p.out_ld;
      const int zc = (int)(((long)sg.B * sg.OH * sg.OW + w.chunk - 1) / w.chunk);
      hipLaunchKernelGGL(wgrad_f32_kernel, dim3(w.tiles, 1, zc), dim3(256), 0, st, q, part + (size_t)z * p.Cout * K, w.chunk);
      z += zc;
    }
  }
  if (db) hipLaunchKernelGGL(bias_grad_kernel, dim3(bj.ncx, bj.nblk), dim3(256), 0, st, bj);
  if (reduce) {
    const long n4 = (long)p.Cout * K / 4;
    const int red_blocks = (int)std::min<long>((n4 + 31) / 32, 8192), extra = db ? (p.Cout + 31) / 32 : 0;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(red_blocks + extra)), dim3(256), 0, st, part, dw, n4, w.split, accumulate, red_blocks, bpart, db, p.Cout,
                       w.nblk);
  } else if (db) {
    hipLaunchKernelGGL(bias_reduce_kernel, dim3((p.Cout + 31) / 32), dim3(256), 0, st, bpart, db, p.Cout, w.nblk, accumulate);
  }
  return check_launch("wgrad");
}

int launch_wgrad_multi(const ConvParams &p, int nseg, const WgradSeg *segs, float *dw, float *db, void *ws, size_t ws_bytes, hipStream_t st, int accumulate) {
  if (!p.in || !p.out || !dw || nseg <= 0 || !segs) return set_error(DF_ERR_ARG, "wgrad: null pointer");
  if (p.Cin % 4 || p.in_ld % 4 || p.in_coff % 4 || p.out_ld % 4 || p.out_coff % 4 || p.Cout % 4)
    return set_error(DF_ERR_ARG, "wgrad: channel counts / strides / offsets must be multiples of 4");
  if (p.up != 1 || p.zcount != 1) return set_error(DF_ERR_ARG, "wgrad: input dilation / grouped launches not supported");
  for (int g0 = 0; g0 < nseg; g0 += WGRAD_MAX_SEGS) {
    const int rc = launch_wgrad_segs(p, std::min(WGRAD_MAX_SEGS, nseg - g0), segs + g0, dw, db, ws, ws_bytes, st, g0 ? 1 : accumulate);
    if (rc != DF_OK) return rc;
  }
  return DF_OK;
}

int launch_wgrad(const ConvParams &p, float *dw, float *db, void *ws, size_t ws_bytes, hipStream_t st, int accumulate) {
  if (!p.in || !p.out || !dw) return set_error(DF_ERR_ARG, "wgrad: null pointer");
  if ((long)p.B * p.OH * p.OW <= 0) return DF_OK;
  const WgradSeg sg = single_seg(p);
  return launch_wgrad_multi(p, 1, &sg, dw, db, ws, ws_bytes, st, accumulate);
}

}  // namespace df

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
#include "igemm.h"

namespace df {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDK = 36;   // padded LDS row (floats): 16 lanes x 16 B land on 64 distinct banks

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
  int a_iy0[A_ROWS], a_ix0[A_ROWS], a_pix[A_ROWS];
#pragma unroll
  for (int i = 0; i < A_ROWS; ++i) {
    const int m = m0 + lrow + 32 * i;
    if (m < M) {
      const int ohw = p.OH * p.OW;
      const int b = m / ohw, rem = m - b * ohw;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_iy0[i] = oy * p.stride - p.pad;
      a_ix0[i] = ox * p.stride - p.pad;
      a_pix[i] = (b * p.H + a_iy0[i]) * p.W + a_ix0[i];
    } else {
      a_iy0[i] = -(1 << 28);     // fails every bounds check -> zero rows
      a_ix0[i] = 0;
      a_pix[i] = 0;
    }
  }
  const bool one_tap = (p.KH * p.KW == 1);
  const int cin_shift = 31 - __builtin_clz(p.Cin);   // multi-tap layers have power-of-two Cin (host-checked)

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
    const int doff = dy * p.W + dx;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
      const bool ok = kok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4 *>(in + (size_t)(a_pix[i] + doff) * p.in_ld + c);
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

struct TileCfg { int bm, bn, wmv; };

TileCfg pick_cfg(const ConvParams &p) {
  const long M = (long)p.B * p.OH * p.OW;
  // 128x128 when it still fills the chip (>= ~1 tile per CU) or when rows are grouped (colsum / grouped
  // bias need BM | rows_per_group); 64x64 for narrow or small problems.
  const long t128 = ((M + 127) / 128) * ((p.Cout + 127) / 128) * p.zcount;
  if (p.Cout % 128 == 0 && (t128 >= 192 || p.rows_per_group > 0)) return {128, 128, 2};
  if (p.rows_per_group > 0) return {128, 128, 2};
  return {64, 64, 2};
}

}  // namespace

int conv_colsum_rows(const ConvParams &p) {
  const TileCfg c = pick_cfg(p);
  const long M = (long)p.B * p.OH * p.OW;
  return (int)(((M + c.bm - 1) / c.bm) * c.wmv) * p.zcount;
}

double conv_flops(const ConvParams &p) {
  return 2.0 * p.B * p.OH * p.OW * (double)p.Cout * p.KH * p.KW * p.Cin * p.zcount;
}

int launch_conv(const ConvParams &p, hipStream_t st) {
  if (!p.in || !p.wgt || (!p.out && !p.colsum)) return set_error(DF_ERR_ARG, "conv: null pointer");
  if (p.Cin % 4 || p.in_ld % 4 || p.in_coff % 4 || p.z_in_coff % 4 || p.z_wgt % 4)
    return set_error(DF_ERR_ARG, "conv: Cin/in_ld/in_coff must be multiples of 4 (16-B vector loads)");
  if (p.KH * p.KW > 1 && (p.Cin & (p.Cin - 1)))
    return set_error(DF_ERR_ARG, "conv: multi-tap convolutions need a power-of-two Cin (got %d)", p.Cin);
  const long M = (long)p.B * p.OH * p.OW;
  if (M <= 0 || p.Cout <= 0) return DF_OK;
  if (M * (long)p.out_ld >= (1L << 40) || (long)p.B * p.H * p.W >= (1L << 31))
    return set_error(DF_ERR_ARG, "conv: tensor too large for 32-bit pixel indexing");
  const TileCfg c = pick_cfg(p);
  if (p.rows_per_group > 0 && (p.rows_per_group % c.bm))
    return set_error(DF_ERR_ARG, "conv: rows_per_group must be a multiple of %d", c.bm);
  const long tiles = ((M + c.bm - 1) / c.bm) * ((p.Cout + c.bn - 1) / c.bn);
  dim3 grid((unsigned)tiles, 1, p.zcount);
  const size_t lds = (size_t)2 * (c.bm + c.bn) * LDK * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {   // 72 KiB of dynamic LDS for the 128x128 tile: above the 64 KiB default cap
    hipFuncSetAttribute(reinterpret_cast<const void *>(&igemm_f32_kernel<128, 128, 2, 2>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * LDK * (int)sizeof(float));
    attr_done = true;
  }
  if (c.bm == 128)
    hipLaunchKernelGGL((igemm_f32_kernel<128, 128, 2, 2>), grid, dim3(256), lds, st, p);
  else
    hipLaunchKernelGGL((igemm_f32_kernel<64, 64, 2, 2>), grid, dim3(256), lds, st, p);
  return check_launch("igemm");
}

}  // namespace df

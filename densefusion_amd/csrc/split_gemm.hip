// EXPERIMENT (development build only, -DDF_DEV; never the default path, never the bench's headline): fp32 GEMM on the bf16 matrix cores.
//
// Every fp32 operand is cut into three bf16 terms, a = hi + mid + lo (round-to-nearest at each cut: |mid| <= 2^-9 |a|, |lo| <= 2^-18 |a|),
// and a product a*b is accumulated in fp32 from the six term pairs whose magnitude is above 2^-27 |a b|:
//   lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi       (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate: 16 / 6 = 2.7x at equal efficiency).
// The weights are cut once per layer (cached planes [3][Cout][K] bf16); the activations are cut while the workgroup stages its tile
// (global fp32 -> registers -> three bf16 planes in LDS): no extra pass over HBM.
// Covered: 1x1 / per-point / Winograd-domain launches (plain GEMMs: stride 1, no padding) with Cout % 128 == 0 and K % 32 == 0, bias (per
// channel or per row group), residual, ReLU / PReLU, fused column sums, blockIdx.z batches; everything else stays on the fp32 kernels.
// Switch: DF_GEMM_SPLIT_BF16=1 in the environment of a process that loaded libdfusion_hip_dev.so.  Results: profiles/r04_experiments/README.md.
#include "igemm.h"
#ifdef DF_DEV
#include <map>
#include <mutex>
#include <tuple>

namespace df {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SBM = 128, SBN = 128, SBK = 32;
constexpr int PLANE_BYTES = 128 * 64;                 // 128 rows x 32 bf16

// (hi, mid, lo) of two floats, each pair packed into one dword (element 0 in the low half)
__device__ __forceinline__ void cut3(float a0, float a1, unsigned &hi, unsigned &mid, unsigned &lo) {
  const bf16x2 h = __builtin_convertvector(f32x2{a0, a1}, bf16x2);                 // v_cvt_pk_bf16_f32 (round to nearest even)
  hi = __builtin_bit_cast(unsigned, h);
  const float r0 = a0 - __builtin_bit_cast(float, hi << 16), r1 = a1 - __builtin_bit_cast(float, hi & 0xFFFF0000u);     // exact
  const bf16x2 m = __builtin_convertvector(f32x2{r0, r1}, bf16x2);
  mid = __builtin_bit_cast(unsigned, m);
  const float s0 = r0 - __builtin_bit_cast(float, mid << 16), s1 = r1 - __builtin_bit_cast(float, mid & 0xFFFF0000u);   // exact
  const bf16x2 l = __builtin_convertvector(f32x2{s0, s1}, bf16x2);
  lo = __builtin_bit_cast(unsigned, l);
}

__global__ void cut_weights_kernel(const float *__restrict__ w, unsigned *__restrict__ planes, long pairs) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pairs) return;
  const float2 v = reinterpret_cast<const float2 *>(w)[i];
  unsigned h, m, l;
  cut3(v.x, v.y, h, m, l);
  planes[i] = h; planes[pairs + i] = m; planes[2 * pairs + i] = l;
}

struct SplitArgs {
  const float *in; const __bf16 *wpl; const float *bias; const float *res; const float *prelu; float *out; float *colsum;
  long M, wplane;                 // rows; elements of one weight plane
  int N, K, in_ld, in_coff, out_ld, out_coff, res_ld, res_coff, act, rows_per_group, rows_valid, bias_group_ld;
  long z_in_coff, z_wgt, z_bias, z_out_coff;
  int tiles_m, tiles_n;         // of the launched kernel's tile
  long cs_rows;                 // rows of the column-sum partial buffer per z: 2 per 128 rows (igemm.h conv_colsum_rows)
};

// byte offset of the 16-byte piece `slot` (8 consecutive k) of row `row` inside a plane: 64-byte rows, the slot XOR-ed with bits 2..3 of the
// row so that the 16 rows a ds_read_b128 lane group touches fall on 16 different 16-byte bank slots
__device__ __forceinline__ int piece(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

// epilogue of one wave's 64 x 64 block: lane = output channel (col), registers = 16 pixel rows: row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).
// row0: first row of the workgroup's tile (inside ONE row group: host-checked), wrow: the wave's row offset in it, nb: its first column,
// cs_row: its row in the column-sum partial buffer (one per 64 tile rows, the fp32 kernel's layout)
__device__ __forceinline__ void split_epilogue(const SplitArgs &a, const f32x16 (&acc)[2][2], int z, long row0, int wrow, int nb, int fr, int fh, long cs_row) {
  const float slope = a.act == ACT_PRELU ? a.prelu[0] : 0.f;
  float *out = a.out + z * a.z_out_coff + a.out_coff;
  const float *bias = a.bias ? a.bias + z * a.z_bias : nullptr;
  const float *gbias = bias && a.bias_group_ld > 0 ? bias + (row0 / a.rows_per_group) * a.bias_group_ld : nullptr;
  // fused column sums (igemm.h): over the rows that are real points
  const int grp = a.rows_per_group > 0 ? (int)(row0 / a.rows_per_group) : 0;
  const long left = a.M - row0;
  const int valid = a.rows_per_group > 0 ? a.rows_valid - (int)(row0 - (long)grp * a.rows_per_group) : (1 << 30);
  const int limit = (int)(left < valid ? left : valid);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = nb + j * 32 + fr;
    const float b1 = gbias ? gbias[n] : bias ? bias[n] : 0.f;
    float csum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int tb = wrow + i * 32 + 4 * fh;          // first tile row of this lane's 16
      const long mb = row0 + tb;
      float r[16];
      if (a.res) {          // all 16 residual loads in flight before the first use (rows past M re-read row M - 1)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          long m = mb + (e & 3) + 8 * (e >> 2);
          m = m < a.M ? m : a.M - 1;
          r[e] = a.res[m * a.res_ld + a.res_coff + n];
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int tr = tb + (e & 3) + 8 * (e >> 2);
        const long m = row0 + tr;
        float v = acc[i][j][e] + b1;
        if (a.res) v += r[e];
        if (a.act == ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (a.act == ACT_PRELU) v = v > 0.f ? v : v * slope;
        if (a.out && m < a.M) out[m * a.out_ld + n] = v;
        csum += tr < limit ? v : 0.f;
      }
    }
    if (a.colsum && cs_row < a.cs_rows) {      // (every partial row the buffer has is written: the finish kernel sums them all)
      csum += __shfl_xor(csum, 32);
      if (fh == 0) a.colsum[((size_t)z * a.cs_rows + (size_t)cs_row) * a.N + n] = csum;
    }
  }
}

__global__ __launch_bounds__(256, 2) void gemm_split_bf16_kernel(const SplitArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[6 * PLANE_BYTES];       // A planes 0..2, B planes 3..5
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware order: the 8 XCDs take whole row tiles, the column tiles of a row tile run back to back on one XCD (its A rows stay in that L2)
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int tm = (seq / a.tiles_n) * 8 + xcd, tn = seq % a.tiles_n;
  if (tm >= a.tiles_m) return;
  const int z = blockIdx.z;
  const long row0 = (long)tm * SBM;
  const int n0 = tn * SBN;
  const float *in = a.in + z * a.z_in_coff + a.in_coff;
  const __bf16 *wpl = a.wpl + z * a.z_wgt;

  // staging roles: pieces tid and tid + 256 of the 512 (row, 8-k) pieces of a 128 x 32 tile
  const int srow0 = tid >> 2, srow1 = srow0 + 64, sslot = tid & 3;
  const int soff0 = piece(srow0, sslot), soff1 = piece(srow1, sslot);
  long r0 = row0 + srow0, r1 = row0 + srow1;
  r0 = r0 < a.M ? r0 : a.M - 1;                                   // the last row tile re-reads row M - 1 (its results are not stored)
  r1 = r1 < a.M ? r1 : a.M - 1;
  const float *ag0 = in + r0 * a.in_ld + sslot * 8, *ag1 = in + r1 * a.in_ld + sslot * 8;
  const __bf16 *bg0 = wpl + (long)(n0 + srow0) * a.K + sslot * 8, *bg1 = wpl + (long)(n0 + srow1) * a.K + sslot * 8;
  // the activation rows come from HBM: their loads for k-step kt + 2 are in flight while step kt is multiplied and step kt + 1 is cut and
  // written to LDS (two register sets x / y: one step of 48 MFMAs per wave = 0.65 us does not cover a miss to HBM); the weight planes sit
  // in the L2: one step ahead
  float4 xa00, xa01, xa10, xa11, ya00, ya01, ya10, ya11;
  uint4 rb00, rb01, rb02, rb10, rb11, rb12;
#define SPLIT_FETCH_A(S, k0)                                                     \
  do {                                                                           \
    S##a00 = *reinterpret_cast<const float4 *>(ag0 + (k0));                      \
    S##a01 = *reinterpret_cast<const float4 *>(ag0 + (k0) + 4);                  \
    S##a10 = *reinterpret_cast<const float4 *>(ag1 + (k0));                      \
    S##a11 = *reinterpret_cast<const float4 *>(ag1 + (k0) + 4);                  \
  } while (0)
#define SPLIT_FETCH_B(k0)                                                        \
  do {                                                                           \
    rb00 = *reinterpret_cast<const uint4 *>(bg0 + (k0));                         \
    rb01 = *reinterpret_cast<const uint4 *>(bg0 + a.wplane + (k0));              \
    rb02 = *reinterpret_cast<const uint4 *>(bg0 + 2 * a.wplane + (k0));          \
    rb10 = *reinterpret_cast<const uint4 *>(bg1 + (k0));                         \
    rb11 = *reinterpret_cast<const uint4 *>(bg1 + a.wplane + (k0));              \
    rb12 = *reinterpret_cast<const uint4 *>(bg1 + 2 * a.wplane + (k0));          \
  } while (0)
#define SPLIT_STAGE1(x0, x1, off, b0, b1, b2)                                    \
  do {                                                                           \
    uint4 h, m, l;                                                               \
    cut3(x0.x, x0.y, h.x, m.x, l.x);                                             \
    cut3(x0.z, x0.w, h.y, m.y, l.y);                                             \
    cut3(x1.x, x1.y, h.z, m.z, l.z);                                             \
    cut3(x1.z, x1.w, h.w, m.w, l.w);                                             \
    *reinterpret_cast<uint4 *>(lds + (off)) = h;                                 \
    *reinterpret_cast<uint4 *>(lds + PLANE_BYTES + (off)) = m;                   \
    *reinterpret_cast<uint4 *>(lds + 2 * PLANE_BYTES + (off)) = l;               \
    *reinterpret_cast<uint4 *>(lds + 3 * PLANE_BYTES + (off)) = b0;              \
    *reinterpret_cast<uint4 *>(lds + 4 * PLANE_BYTES + (off)) = b1;              \
    *reinterpret_cast<uint4 *>(lds + 5 * PLANE_BYTES + (off)) = b2;              \
  } while (0)
#define SPLIT_STAGE(S)                                                           \
  do {                                                                           \
    SPLIT_STAGE1(S##a00, S##a01, soff0, rb00, rb01, rb02);                       \
    SPLIT_STAGE1(S##a10, S##a11, soff1, rb10, rb11, rb12);                       \
  } while (0)

  const int wr = wave >> 1, wc = wave & 1, fr = lane & 31, fh = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  int offa[2][2], offb[2][2];          // [k16 step][tile]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      offa[ks][t] = piece(wr * 64 + t * 32 + fr, ks * 2 + fh);
      offb[ks][t] = 3 * PLANE_BYTES + piece(wc * 64 + t * 32 + fr, ks * 2 + fh);
    }

#define SPLIT_COMPUTE()                                                                                   \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                      \
    bf16x8 fa[2][3], fb[2][3];                                                                            \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                         \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                       \
      fa[t][p] = *reinterpret_cast<const bf16x8 *>(lds + p * PLANE_BYTES + offa[ks][t]);                  \
      fb[t][p] = *reinterpret_cast<const bf16x8 *>(lds + p * PLANE_BYTES + offb[ks][t]);                  \
    }                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                       \
      f32x16 c = acc[i][j]; /* small terms first */                                                       \
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);                        \
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);                        \
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);                        \
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);                        \
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);                        \
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);                        \
      acc[i][j] = c;                                                                                      \
    }                                                                                                     \
  }

  const int nk = a.K / SBK;
  SPLIT_FETCH_A(x, 0);
  SPLIT_FETCH_B(0);
  if (nk > 1) SPLIT_FETCH_A(y, SBK);
  SPLIT_STAGE(x);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    // LDS: step kt; set y: the rows of step kt + 1
    if (kt + 1 < nk) SPLIT_FETCH_B((kt + 1) * SBK);
    if (kt + 2 < nk) SPLIT_FETCH_A(x, (kt + 2) * SBK);
    SPLIT_COMPUTE();
    if (kt + 1 >= nk) break;
    __syncthreads();
    SPLIT_STAGE(y);
    __syncthreads();
    // LDS: step kt + 1; set x: the rows of step kt + 2
    if (kt + 2 < nk) SPLIT_FETCH_B((kt + 2) * SBK);
    if (kt + 3 < nk) SPLIT_FETCH_A(y, (kt + 3) * SBK);
    SPLIT_COMPUTE();
    if (kt + 2 >= nk) break;
    __syncthreads();
    SPLIT_STAGE(x);
    __syncthreads();
  }

  split_epilogue(a, acc, z, row0, wr * 64, n0 + wc * 64, fr, fh, (long)tm * 2 + wr);
}

// Second form: 256 x 128 x 32 tile, 8 waves (4 x 2) of 64 x 64, TWO LDS stages (144 KB: one workgroup per CU, two waves per SIMD) and ONE barrier
// per k-step: while a wave multiplies step kt out of stage kt % 2 it cuts step kt + 1 (loaded one iteration earlier) into the other stage, and
// the loads of step kt + 2 are in flight.
constexpr int V2_APL = 256 * 64, V2_BPL = 128 * 64, V2_STAGE = 3 * V2_APL + 3 * V2_BPL;

__global__ __launch_bounds__(512, 1) void gemm_split_bf16_v2_kernel(const SplitArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int tm = (seq / a.tiles_n) * 8 + xcd, tn = seq % a.tiles_n;
  if (tm >= a.tiles_m) return;
  const int z = blockIdx.z;
  const long row0 = (long)tm * 256;
  const int n0 = tn * SBN;
  const float *in = a.in + z * a.z_in_coff + a.in_coff;
  const __bf16 *wpl = a.wpl + z * a.z_wgt;

  // staging roles: A pieces tid and tid + 512 of the 1024 (row, 8-k) pieces of the 256 x 32 tile; B piece tid of each plane's 512
  const int srow0 = tid >> 2, srow1 = srow0 + 128, sslot = tid & 3;
  const int soff0 = piece(srow0, sslot), soff1 = piece(srow1, sslot);
  long r0 = row0 + srow0, r1 = row0 + srow1;
  r0 = r0 < a.M ? r0 : a.M - 1;
  r1 = r1 < a.M ? r1 : a.M - 1;
  const float *ag0 = in + r0 * a.in_ld + sslot * 8, *ag1 = in + r1 * a.in_ld + sslot * 8;
  const __bf16 *bg0 = wpl + (long)(n0 + srow0) * a.K + sslot * 8;
  float4 xa00, xa01, xa10, xa11, ya00, ya01, ya10, ya11;
  uint4 xb0, xb1, xb2, yb0, yb1, yb2;
#define V2_FETCH(S, k0)                                                          \
  do {                                                                           \
    S##a00 = *reinterpret_cast<const float4 *>(ag0 + (k0));                      \
    S##a01 = *reinterpret_cast<const float4 *>(ag0 + (k0) + 4);                  \
    S##a10 = *reinterpret_cast<const float4 *>(ag1 + (k0));                      \
    S##a11 = *reinterpret_cast<const float4 *>(ag1 + (k0) + 4);                  \
    S##b0 = *reinterpret_cast<const uint4 *>(bg0 + (k0));                        \
    S##b1 = *reinterpret_cast<const uint4 *>(bg0 + a.wplane + (k0));             \
    S##b2 = *reinterpret_cast<const uint4 *>(bg0 + 2 * a.wplane + (k0));         \
  } while (0)
#define V2_CUT(x0, x1, base, off)                                                \
  do {                                                                           \
    uint4 h, m, l;                                                               \
    cut3(x0.x, x0.y, h.x, m.x, l.x);                                             \
    cut3(x0.z, x0.w, h.y, m.y, l.y);                                             \
    cut3(x1.x, x1.y, h.z, m.z, l.z);                                             \
    cut3(x1.z, x1.w, h.w, m.w, l.w);                                             \
    *reinterpret_cast<uint4 *>((base) + (off)) = h;                              \
    *reinterpret_cast<uint4 *>((base) + V2_APL + (off)) = m;                     \
    *reinterpret_cast<uint4 *>((base) + 2 * V2_APL + (off)) = l;                 \
  } while (0)
#define V2_STAGE_TO(S, base)                                                     \
  do {                                                                           \
    V2_CUT(S##a00, S##a01, base, soff0);                                         \
    V2_CUT(S##a10, S##a11, base, soff1);                                         \
    *reinterpret_cast<uint4 *>((base) + 3 * V2_APL + soff0) = S##b0;             \
    *reinterpret_cast<uint4 *>((base) + 3 * V2_APL + V2_BPL + soff0) = S##b1;    \
    *reinterpret_cast<uint4 *>((base) + 3 * V2_APL + 2 * V2_BPL + soff0) = S##b2; \
  } while (0)

  const int wr = wave >> 1, wc = wave & 1, fr = lane & 31, fh = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  int offa[2][2], offb[2][2];          // [k16 step][tile]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      offa[ks][t] = piece(wr * 64 + t * 32 + fr, ks * 2 + fh);
      offb[ks][t] = 3 * V2_APL + piece(wc * 64 + t * 32 + fr, ks * 2 + fh);
    }
  // pair-major order (every tile's lo*hi, then every tile's hi*lo, ...): per accumulator the order is still small terms first, and the lo / mid
  // fragments die early, which leaves registers for the next k16 step's fragments
#define V2_COMPUTE(base)                                                                                  \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                      \
    bf16x8 fa[2][3], fb[2][3];                                                                            \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                         \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                       \
      fa[t][p] = *reinterpret_cast<const bf16x8 *>((base) + p * V2_APL + offa[ks][t]);                    \
      fb[t][p] = *reinterpret_cast<const bf16x8 *>((base) + p * V2_BPL + offb[ks][t]);                    \
    }                                                                                                     \
    _Pragma("unroll") for (int q = 0; q < 6; ++q) {                                                       \
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};                               \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                       \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[q]], fb[j][PB[q]], acc[i][j], 0, 0, 0); \
    }                                                                                                     \
  }
  // one k-step, branch-free: loads of step kt + 2 (clamped to the last step: harmless re-read), cut of step kt + 1 into the other stage, products
  // of step kt -- ONE basic block, with the issue order pinned so that the cut's vector instructions and the LDS traffic sit in the gaps of the
  // MFMA stream of the SAME wave (a wave issuing MFMAs back to back blocks the other wave of its SIMD: DESIGN 6a)
#define V2_BODY(FS, SS, kf, cur, nxt)                                            \
  do {                                                                           \
    V2_FETCH(FS, (kf));                                                          \
    V2_STAGE_TO(SS, nxt);                                                        \
    V2_COMPUTE(cur);                                                             \
    __builtin_amdgcn_sched_group_barrier(0x020, 7, 0);                           \
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);                          \
    _Pragma("unroll") for (int g = 0; g < 48; ++g) {                             \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                         \
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                         \
      if (g >= 8 && g < 32 && (g & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); \
      if (g % 5 == 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);         \
    }                                                                            \
    __syncthreads();                                                             \
  } while (0)

  unsigned char *const st0 = lds2, *const st1 = lds2 + V2_STAGE;
  const int nk = a.K / SBK;
  const int klast = a.K - SBK;
  V2_FETCH(x, 0);
  V2_FETCH(y, SBK < klast ? SBK : klast);
  V2_STAGE_TO(x, st0);
  __syncthreads();
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    // stage 0: step kt; set y: step kt + 1
    const int k2 = (kt + 2) * SBK, k3 = (kt + 3) * SBK;
    V2_BODY(x, y, k2 < klast ? k2 : klast, st0, st1);
    // stage 1: step kt + 1; set x: step kt + 2
    V2_BODY(y, x, k3 < klast ? k3 : klast, st1, st0);
  }
  if (nk & 1) { V2_COMPUTE(st0); }       // the last step of an odd count (cut into stage 0 by the loop's second half)
  split_epilogue(a, acc, z, row0, wr * 64, n0 + wc * 64, fr, fh, (long)tm * 4 + wr);
}

struct Planes { __bf16 *ptr; long plane; long epoch; };
std::mutex g_mu;
long g_epoch = 1;
std::map<std::tuple<const float *, long, int>, Planes> g_cache;        // (weights, elements, device) -> planes; dev build: never freed
std::map<std::tuple<int, hipStream_t>, Planes> g_scratch;              // (device, stream) -> planes of weights that may change: cut per launch

void cut(const float *w, __bf16 *planes, long elems, hipStream_t st) {
  const long pairs = elems / 2;
  hipLaunchKernelGGL(cut_weights_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, w, reinterpret_cast<unsigned *>(planes), pairs);
}

Planes weight_planes(const float *w, long elems, bool constant, hipStream_t st) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> g(g_mu);
  if (!constant) {
    Planes &sc = g_scratch[std::make_tuple(dev, st)];
    if (sc.plane < elems) {
      if (sc.ptr) (void)hipFree(sc.ptr);          // (synchronises the device: launches reading the old planes have finished)
      sc = Planes{nullptr, 0, 0};
      if (hipMalloc(reinterpret_cast<void **>(&sc.ptr), (size_t)elems * 3 * sizeof(__bf16)) != hipSuccess) return Planes{nullptr, 0, 0};
      sc.plane = elems;
    }
    cut(w, sc.ptr, elems, st);                   // plane stride = elems of THIS launch (the buffer may be larger)
    return Planes{sc.ptr, elems, 0};
  }
  Planes &pl = g_cache[std::make_tuple(w, elems, dev)];
  if (!pl.ptr) {
    if (hipMalloc(reinterpret_cast<void **>(&pl.ptr), (size_t)elems * 3 * sizeof(__bf16)) != hipSuccess) return Planes{nullptr, 0, 0};
    pl.plane = elems;
    pl.epoch = 0;
  }
  if (pl.epoch != g_epoch) {
    cut(w, pl.ptr, elems, st);
    pl.epoch = g_epoch;
  }
  return pl;
}

}  // namespace

// true when the launch was taken (results in p.out); false: not eligible, the caller goes on to the fp32 kernels
void split_gemm_invalidate() {
  std::lock_guard<std::mutex> g(g_mu);
  ++g_epoch;
}

bool try_split_gemm(const ConvParams &p, hipStream_t st) {
  static const bool on = dev_getenv("DF_GEMM_SPLIT_BF16") != nullptr;
  static const bool verbose = dev_getenv("DF_GEMM_SPLIT_VERBOSE") != nullptr;
  if (!on) return false;
  const long M = (long)p.B * p.OH * p.OW;
  const int K = p.Cin;
  const bool ok = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.up == 1 && p.H == p.OH && p.W == p.OW && (p.out || p.colsum) && !(p.colsum && p.res) && p.splitk <= 1 &&
                  p.Cout % SBN == 0 && K % SBK == 0 && K >= SBK && M >= 1 && (p.rows_per_group == 0 ? p.bias_group_ld == 0 : p.rows_per_group % SBM == 0) && (p.z_wgt % 2) == 0;
  if (verbose) fprintf(stderr, "[df-split] M=%ld N=%d K=%d z%d -> %s\n", M, p.Cout, K, p.zcount, ok ? "bf16 x 6" : "fp32");
  if (!ok) return false;
  const long elems = (long)(p.zcount - 1) * p.z_wgt + (long)p.Cout * K;
  const Planes pl = weight_planes(p.wgt, elems, p.wgt_const, st);
  if (!pl.ptr) return false;
  SplitArgs a{};
  a.in = p.in; a.wpl = pl.ptr; a.bias = p.bias; a.res = p.res; a.prelu = p.prelu; a.out = p.out; a.colsum = p.colsum;
  a.M = M; a.wplane = pl.plane; a.N = p.Cout; a.K = K; a.in_ld = p.in_ld; a.in_coff = p.in_coff; a.out_ld = p.out_ld; a.out_coff = p.out_coff;
  a.res_ld = p.res_ld; a.res_coff = p.res_coff; a.act = p.act; a.rows_per_group = p.rows_per_group; a.rows_valid = p.rows_valid; a.bias_group_ld = p.bias_group_ld;
  a.z_in_coff = p.z_in_coff; a.z_wgt = p.z_wgt; a.z_bias = p.z_bias; a.z_out_coff = p.z_out_coff;
  a.cs_rows = ((M + 127) / 128) * 2;
  a.tiles_n = p.Cout / SBN;
  static const int variant = dev_getenv("DF_GEMM_SPLIT_V") ? atoi(dev_getenv("DF_GEMM_SPLIT_V")) : 2;
  // measured per shape: the 256-row form wins from K = 384 up (189 against 172 TFLOP/s on 139 000 x 2 304 x 1 024), the 128-row form below
  const bool v2 = variant == 2 && K >= 384 && (p.rows_per_group == 0 || p.rows_per_group % 256 == 0);
  if (v2) {
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_done[dev]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_split_bf16_v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * V2_STAGE);
      attr_done[dev] = true;
    }
    a.tiles_m = (int)((M + 255) / 256);
    const unsigned grid = (unsigned)(((a.tiles_m + 7) / 8) * 8 * a.tiles_n);
    hipLaunchKernelGGL(gemm_split_bf16_v2_kernel, dim3(grid, 1, p.zcount), dim3(512), 2 * V2_STAGE, st, a);
  } else {
    a.tiles_m = (int)((M + SBM - 1) / SBM);
    const unsigned grid = (unsigned)(((a.tiles_m + 7) / 8) * 8 * a.tiles_n);
    hipLaunchKernelGGL(gemm_split_bf16_kernel, dim3(grid, 1, p.zcount), dim3(256), 0, st, a);
  }
  return true;
}

}  // namespace df
#endif

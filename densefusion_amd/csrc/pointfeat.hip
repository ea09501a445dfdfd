// The first two layers of both branches of PoseNetFeat / PoseRefineNetFeat (lib/network.py:53-58,152-157) chained inside one
// launch:   x1 = relu(W1 xyz + b1)  ->  x2 = relu(W2 x1 + b2)      (Conv1d 3 -> 64 -> 128)
//           e1 = relu(We1 emb + b)  ->  e2 = relu(We2 e1 + b)      (Conv1d 32 -> 64 -> 128)
// These are K = 3 / 32 / 64 products over every point: as separate GEMM launches they ran at 40-70 TFLOP/s (a workgroup's prologue
// and epilogue outweigh a two-k-tile main loop) and the 64-wide intermediates went through HBM.  Here a wave owns 32 points: the
// layer-1 output is written to LDS directly in MFMA A-fragment order, layer 2 multiplies from there, every output row is stored
// once with 16-byte vectors.  Weights are staged in LDS once per workgroup, also in fragment order.
//
// Fragment order: v_mfma_f32_32x32x2_f32 takes A[i][k] from lane (i = lane % 32, k = lane / 32).  igemm.hip feeds it from 16-byte LDS
// reads of four consecutive k per lane, so its k-th MFMA step of an 8-wide k group g pairs k = 8g + e (lanes 0-31) with 8g + 4 + e
// (lanes 32-63), e = 0..3.  The same pairing is kept here -- which two products share an instruction is visible in the last bit --
// by storing a K-deep operand row as two half-rows Af[half][row][4g + e] = A[row][8g + 4 half + e] (row stride 36 floats:
// conflict-free ds_read_b128).  Per output element: groups ascending, e ascending, from a zero accumulator, bias added afterwards,
// then ReLU -- so the fused and the layer-by-layer forms agree bit for bit (tests/test_network_gpu.py).
#include <cstdlib>

#include "layers.h"

namespace df {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PF_THREADS = 256, PF_ROWS = 128, RS = 36;          // 4 waves x 32 points; padded half-row stride
constexpr int W2F = 2 * 128 * RS;       // conv2 / e_conv2 in fragment order: [2][128][32 (+4)]
constexpr int WE1F = 2 * 64 * 20;       // e_conv1: [2][64][16 (+4)]
constexpr int A1F = 2 * 32 * RS;        // per wave: a 32-point operand tile, K <= 64

// W [n][K] row-major (global) -> Wf[half][n][4g + e] = W[n][8g + 4 half + e], half-row stride `rs`
__device__ __forceinline__ void stage_weights(const float *__restrict__ w, float *wf, int n_rows, int K, int rs) {
  const int k4n = K / 4;
  for (int i = threadIdx.x; i < n_rows * k4n; i += PF_THREADS) {
    const int n = i / k4n, j = i - n * k4n;               // float4 j of the row: k = 4j .. 4j + 3 -> half j & 1, group j >> 1
    *reinterpret_cast<f32x4 *>(wf + ((j & 1) * n_rows + n) * rs + 4 * (j >> 1)) = *reinterpret_cast<const f32x4 *>(w + (size_t)n * K + 4 * j);
  }
}

// acc[nb] += A (this wave's 32 x K tile in a1) x W^T (n blocks nb of 32 columns), K = 4 * quads k steps
template <int NB>
__device__ __forceinline__ void mma_tile(const float *a1, const float *wf, int w_rows, int w_rs, int quads, int lane, f32x16 (&acc)[NB]) {
  const int r = lane & 31, half = lane >> 5;
  const float *ap = a1 + (half * 32 + r) * RS;
  const float *wp = wf + (half * w_rows + r) * w_rs;
  for (int q = 0; q < quads; ++q) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(ap + 4 * q);
    f32x4 b[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) b[nb] = *reinterpret_cast<const f32x4 *>(wp + nb * 32 * w_rs + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[nb][e], acc[nb], 0, 0, 0);
  }
}

// the wave's 32 x 64 tile, kept in a1 in fragment order, to global rows (16-byte vectors, 256 B per row)
__device__ __forceinline__ void store_frag_tile(const float *a1, float *__restrict__ dst, int ld, int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = lane + 64 * i, row = idx >> 4, c4 = idx & 15;          // columns 4 c4 .. + 3: half c4 & 1, group c4 >> 1
    *reinterpret_cast<f32x4 *>(dst + (size_t)row * ld + 4 * c4) = *reinterpret_cast<const f32x4 *>(a1 + ((c4 & 1) * 32 + row) * RS + 4 * (c4 >> 1));
  }
}

// layer-2 epilogue: bias + ReLU on 4 accumulator blocks (128 columns), through a1 as a row-major 32 x 64 staging tile, two halves
__device__ __forceinline__ void store_acc128(f32x16 (&acc)[4], const float *__restrict__ bias, float *a1, float *__restrict__ dst, int ld, int lane) {
  const int j = lane & 31, lh = lane >> 5;
  constexpr int SS = 68;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int nbl = 0; nbl < 2; ++nbl) {
      const float bv = bias[(h * 2 + nbl) * 32 + j];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = 8 * (e >> 2) + 4 * lh + (e & 3);
        float v = acc[h * 2 + nbl][e] + bv;
        v = v > 0.f ? v : 0.f;
        a1[row * SS + nbl * 32 + j] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = lane + 64 * i, row = idx >> 4, c4 = idx & 15;
      *reinterpret_cast<f32x4 *>(dst + (size_t)row * ld + h * 64 + 4 * c4) = *reinterpret_cast<const f32x4 *>(a1 + row * SS + 4 * c4);
    }
  }
}

template <bool DO_X, bool DO_E>
__global__ __launch_bounds__(PF_THREADS) void pointfeat_kernel(const PointFeatParams p, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // carve: [conv2 | w1 (64 x 4)] [e_conv2 | e_conv1] [4 waves x a1]
  float *s_w2 = smem;
  float *s_w1 = s_w2 + (DO_X ? W2F : 0);
  float *s_we2 = s_w1 + (DO_X ? 256 : 0);
  float *s_we1 = s_we2 + (DO_E ? W2F : 0);
  float *s_a = s_we1 + (DO_E ? WE1F : 0);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *a1 = s_a + wave * (A1F > 32 * 68 ? A1F : 32 * 68);
  if (DO_X) {
    stage_weights(p.w2, s_w2, 128, 64, RS);
    for (int i = threadIdx.x; i < 64; i += PF_THREADS)
      *reinterpret_cast<f32x4 *>(s_w1 + 4 * i) = f32x4{p.w1[i * 3 + 0], p.w1[i * 3 + 1], p.w1[i * 3 + 2], p.b1[i]};
  }
  if (DO_E) {
    stage_weights(p.we2, s_we2, 128, 64, RS);
    stage_weights(p.we1, s_we1, 64, 32, 20);
  }
  __syncthreads();
  const int r = lane & 31, half = lane >> 5;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long row0 = (long)tile * PF_ROWS + wave * 32;            // first row of this wave (rows = B * Npad, Npad % 128 == 0)
    const int b = (int)(row0 / p.Npad), n = (int)(row0 - (long)b * p.Npad) + r;
    float *out = p.pf + (size_t)row0 * p.ld;
    if (DO_X) {
      // x1 = relu(b1 + w1 . xyz): lane (r, half) makes the 32 channels 8q + 4 half + e of its point, straight into fragment order
      float x = 0.f, y = 0.f, z = 0.f;
      if (n < p.N) {
        const float *c = p.cloud + ((size_t)b * p.N + n) * 3;
        x = c[0]; y = c[1]; z = c[2];
        if (p.rt) {   // new = (p - T) . R   (tools/eval_ycb.py:211; same expression as layers.hip cloud_conv1)
          const float *R = p.rt + b * 12, *T = R + 9;
          const float dx = x - T[0], dy = y - T[1], dz = z - T[2];
          x = dx * R[0] + dy * R[3] + dz * R[6];
          y = dx * R[1] + dy * R[4] + dz * R[7];
          z = dx * R[2] + dy * R[5] + dz * R[8];
        }
      }
      float *dstf = a1 + (half * 32 + r) * RS;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x4 w = *reinterpret_cast<const f32x4 *>(s_w1 + 4 * (8 * q + 4 * half + e));
          const float v = w[3] + w[0] * x + w[1] * y + w[2] * z;
          o[e] = v > 0.f ? v : 0.f;
        }
        *reinterpret_cast<f32x4 *>(dstf + 4 * q) = o;
      }
      store_frag_tile(a1, out + p.cx1, p.ld, lane);
      f32x16 acc[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;
      mma_tile<4>(a1, s_w2, 128, RS, 8, lane, acc);
      store_acc128(acc, p.b2, a1, out + p.cx2, p.ld, lane);
    }
    if (DO_E) {
      // emb rows -> fragment order (K = 32): lane (r, half) carries the vectors 8g + 4 half .. + 3, g = 0..3, of its point
      {
        const float *src = p.emb + ((size_t)row0 + r) * 32 + 4 * half;
        float *d = a1 + (half * 32 + r) * RS;
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4 *>(d + 4 * g) = *reinterpret_cast<const f32x4 *>(src + 8 * g);
      }
      f32x16 a64[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) a64[nb][e] = 0.f;
      mma_tile<2>(a1, s_we1, 64, 20, 4, lane, a64);
      // e1 = relu(. + bias) back into a1 in fragment order (K = 64 operand of e_conv2): column n = nb*32 + j -> half (n / 4) & 1,
      // slot 4 (n / 8) + n % 4
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float bv = p.be1[nb * 32 + r];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = 8 * (e >> 2) + 4 * half + (e & 3);
          float v = a64[nb][e] + bv;
          v = v > 0.f ? v : 0.f;
          a1[(((r >> 2) & 1) * 32 + row) * RS + nb * 16 + 4 * (r >> 3) + (r & 3)] = v;
        }
      }
      store_frag_tile(a1, out + p.ce1, p.ld, lane);
      f32x16 acc[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;
      mma_tile<4>(a1, s_we2, 128, RS, 8, lane, acc);
      store_acc128(acc, p.be2, a1, out + p.ce2, p.ld, lane);
    }
  }
}

template <bool DO_X, bool DO_E>
int launch_t(const PointFeatParams &p, hipStream_t st) {
  const size_t lds = ((DO_X ? W2F + 256 : 0) + (DO_E ? W2F + WE1F : 0) + 4 * (A1F > 32 * 68 ? A1F : 32 * 68)) * sizeof(float);
  static bool attr_done[64] = {};
  int dev = 0;
  hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_done[dev]) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(&pointfeat_kernel<DO_X, DO_E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done[dev] = true;
  }
  const int tiles = (int)((long)p.B * p.Npad / PF_ROWS);
  hipLaunchKernelGGL((pointfeat_kernel<DO_X, DO_E>), dim3(tiles < 512 ? tiles : 512), dim3(PF_THREADS), lds, st, p, tiles);
  return check_launch("pointfeat");
}

}  // namespace

int launch_pointfeat(const PointFeatParams &p, hipStream_t st) {
  const bool do_x = p.cloud != nullptr, do_e = p.emb != nullptr;
  if (!p.pf || (!do_x && !do_e) || p.B <= 0 || p.N <= 0 || p.Npad % PF_ROWS || p.Npad < p.N || p.ld % 4)
    return set_error(DF_ERR_ARG, "pointfeat: bad arguments (Npad must be a multiple of %d)", PF_ROWS);
  if ((do_x && (!p.w1 || !p.b1 || !p.w2 || !p.b2)) || (do_e && (!p.we1 || !p.be1 || !p.we2 || !p.be2)))
    return set_error(DF_ERR_ARG, "pointfeat: null weights");
  if ((p.cx1 | p.cx2 | p.ce1 | p.ce2) % 4) return set_error(DF_ERR_ARG, "pointfeat: column offsets must be multiples of 4");
  if (do_x && do_e) return launch_t<true, true>(p, st);
  if (do_x) return launch_t<true, false>(p, st);
  return launch_t<false, true>(p, st);
}

}  // namespace df

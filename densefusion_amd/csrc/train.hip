// Native training step of PoseNet / PoseRefineNet (tools/train.py:131-176 of the reference): forward, loss and backward of B same-size
// frames sequenced in C++ on one stream -- no host synchronisation, no allocation, every launch capturable in a hipGraph.
//
//   * parameters and gradients live in ONE flat fp32 buffer each, in KERNEL layout (conv O(HW)I, up_1 / up_2 tap-major, head layer 1
//     split into its per-point and global-feature column blocks, the three head towers stacked): the caller owns both buffers
//     (Adam and the gradient all-reduce are layout-agnostic); df_trainer_pack_param / _unpack_param convert to and from the
//     reference's state-dict layout, so checkpoints keep the reference's keys and shapes (tools/train.py:172-176).
//   * gradients are ACCUMULATED into the flat gradient buffer (the reference's `loss.backward()` per frame, `optimizer.step()`
//     every batch_size frames, tools/train.py:161-169); every reduction has a fixed order -- no float atomics anywhere: two
//     identical steps give bit-identical gradient buffers.
//   * the forward half keeps the exact rewrites of the inference engine that have a cheap adjoint: concatenations written in
//     place (channel offsets), the global-feature fold of head layer 1, up_1 / up_2 as low-resolution per-tap products followed by
//     the 9-tap interpolation, up_3 + final 1x1 + LogSoftmax only at the chosen pixels, the last head layer only for the frame's
//     object (lib/network.py:119-131: the other objects' rows receive no gradient in the reference either).
//
// Reference lines mirrored: lib/extractors.py:29-43,114-124; lib/pspnet.py:20-24,27-37,64-77; lib/network.py:53-68,95-132,151-206;
// lib/loss.py:13-70; lib/loss_refiner.py:12-62.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "igemm.h"
#include "loss.h"
#include "layers.h"
#include "wino.h"

namespace df {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TB = 256;
inline unsigned nblk(long n, long cap = 16384) { long b = (n + TB - 1) / TB; return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b)); }
#define GRID_STRIDE(i, n) for (long i = blockIdx.x * (long)TB + threadIdx.x; i < (n); i += (long)gridDim.x * TB)
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline int conv_out(int in, int k, int stride, int pad, int dil) { return (in + 2 * pad - dil * (k - 1) - 1) / stride + 1; }

// ------------------------------------------------------------------------------------------------
// kernels (the glue between the MFMA launches; every sum in a fixed order)
// ------------------------------------------------------------------------------------------------

// g <- g * act'(y) in place on a [rows][C] view; PReLU (act 2) also leaves per-workgroup partial sums of dslope in `part`
__global__ __launch_bounds__(TB) void act_bwd2d_kernel(float *__restrict__ g, int g_ld, const float *__restrict__ y, int y_ld, long rows, int C4,
                                                       int act, const float *__restrict__ slope_p, float *__restrict__ part) {
  __shared__ float s_red[TB];
  const float slope = act == 2 ? slope_p[0] : 0.f;
  float ds = 0.f;
  GRID_STRIDE(i, rows * C4) {
    const long r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    f32x4 gv = *reinterpret_cast<f32x4 *>(g + r * g_ld + c);
    const f32x4 yv = *reinterpret_cast<const f32x4 *>(y + r * y_ld + c);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (!(yv[e] > 0.f)) {
        if (act == 2) ds += gv[e] * (yv[e] / slope);        // x = y / slope on the negative side
        gv[e] *= slope;
      }
    *reinterpret_cast<f32x4 *>(g + r * g_ld + c) = gv;
  }
  if (act == 2) {
    s_red[threadIdx.x] = ds;
    __syncthreads();
    for (int d = TB / 2; d >= 1; d >>= 1) { if ((int)threadIdx.x < d) s_red[threadIdx.x] += s_red[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = s_red[0];
  }
}

// dst[j] (+)= sum_i part[i * n + j]: one thread per output, i ascending (n > 1); for a single output (n == 1: the PReLU slope) one
// workgroup, thread t adds part[t], part[t + 256], ... and the 256 sums meet in a fixed tree
__global__ __launch_bounds__(TB) void sum_partials_kernel(const float *__restrict__ part, int count, long n, float *__restrict__ dst, int accumulate) {
  if (n == 1) {
    __shared__ float s_red[TB];
    float a = 0.f;
    for (int i = threadIdx.x; i < count; i += TB) a += part[i];
    s_red[threadIdx.x] = a;
    __syncthreads();
    for (int d = TB / 2; d >= 1; d >>= 1) { if ((int)threadIdx.x < d) s_red[threadIdx.x] += s_red[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) dst[0] = accumulate ? dst[0] + s_red[0] : s_red[0];
    return;
  }
  GRID_STRIDE(j, n) {
    float a = 0.f;
    for (int i = 0; i < count; ++i) a += part[(long)i * n + j];
    dst[j] = accumulate ? dst[j] + a : a;
  }
}

// dst (+)= src on [rows][C] views
__global__ __launch_bounds__(TB) void add2d_kernel(float *__restrict__ dst, int d_ld, const float *__restrict__ src, int s_ld, long rows, int C4) {
  GRID_STRIDE(i, rows * C4) {
    const long r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    f32x4 a = *reinterpret_cast<f32x4 *>(dst + r * d_ld + c);
    const f32x4 b = *reinterpret_cast<const f32x4 *>(src + r * s_ld + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] += b[e];
    *reinterpret_cast<f32x4 *>(dst + r * d_ld + c) = a;
  }
}

// AdaptiveAvgPool2d(s) backward, lib/pspnet.py:16: bin i covers [floor(i*H/s), ceil((i+1)*H/s)); dx is a [B*H*W][C] view
__global__ __launch_bounds__(TB) void pool_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, int dx_ld, int B, int H, int W, int C4,
                                                      int s, int accumulate) {
  GRID_STRIDE(i, (long)B * H * W * C4) {
    const int c = (int)(i % C4) * 4;
    long r = i / C4;
    const long pix = r;
    const int xx = (int)(r % W); r /= W;
    const int yy = (int)(r % H);
    const int b = (int)(r / H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int bi = 0; bi < s; ++bi) {
      const int y0 = (bi * H) / s, y1 = ((bi + 1) * H + s - 1) / s;
      if (yy < y0 || yy >= y1) continue;
      for (int bj = 0; bj < s; ++bj) {
        const int x0 = (bj * W) / s, x1 = ((bj + 1) * W + s - 1) / s;
        if (xx < x0 || xx >= x1) continue;
        const f32x4 v = reinterpret_cast<const f32x4 *>(dy)[((long)(b * s + bi) * s + bj) * C4 + c / 4];
        const float cnt = (float)((y1 - y0) * (x1 - x0));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[e] / cnt;
      }
    }
    float *d = dx + pix * dx_ld + c;
    if (accumulate) {
      const f32x4 o = *reinterpret_cast<const f32x4 *>(d);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = o[e] + acc[e];
    }
    *reinterpret_cast<f32x4 *>(d) = acc;
  }
}

// bilinear source (ATen UpSample semantics, fp32): align != 0 -> src = dst*(in-1)/(out-1); else half-pixel, clamped at 0
__device__ inline void bil_src(int dst, int in_size, int out_size, int align, int &i0, int &i1, float &l0, float &l1) {
  float s;
  if (align) s = (out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f) * (float)dst;
  else { s = ((float)in_size / (float)out_size) * ((float)dst + 0.5f) - 0.5f; if (s < 0.f) s = 0.f; }
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
  l0 = 1.f - l1;
}
// the destination indices that may interpolate from source index q: a conservative range, every candidate is re-checked with bil_src
__device__ inline void bil_cands(int q, int in_size, int out_size, int align, int &lo, int &hi) {
  const float inv = align ? (in_size > 1 ? (float)(out_size - 1) / (float)(in_size - 1) : (float)out_size)
                          : (float)out_size / (float)in_size;
  lo = (int)floorf(((float)q - 1.f) * inv) - 2;
  hi = (int)ceilf(((float)q + 1.5f) * inv) + 2;
  if (lo < 0 || q == 0) lo = 0;
  if (hi > out_size - 1 || q == in_size - 1) hi = out_size - 1;
}

// y[b][oy][ox][c] = bilinear resample of x [B][H][W][C] to (OH, OW); y is a [B*OH*OW][C] view of width y_ld
__global__ __launch_bounds__(TB) void bilinear_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int y_ld, int B, int H, int W, int C4,
                                                          int OH, int OW, int align) {
  GRID_STRIDE(i, (long)B * OH * OW * C4) {
    const int c = (int)(i % C4) * 4;
    long r = i / C4;
    const long pix = r;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const int b = (int)(r / OH);
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    bil_src(oy, H, OH, align, y0, y1, wy0, wy1);
    bil_src(ox, W, OW, align, x0, x1, wx0, wx1);
    const float *p = x + (long)b * H * W * C4 * 4 + c;
    const f32x4 v00 = *reinterpret_cast<const f32x4 *>(p + ((long)y0 * W + x0) * C4 * 4), v01 = *reinterpret_cast<const f32x4 *>(p + ((long)y0 * W + x1) * C4 * 4);
    const f32x4 v10 = *reinterpret_cast<const f32x4 *>(p + ((long)y1 * W + x0) * C4 * 4), v11 = *reinterpret_cast<const f32x4 *>(p + ((long)y1 * W + x1) * C4 * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = wy0 * (wx0 * v00[e] + wx1 * v01[e]) + wy1 * (wx0 * v10[e] + wx1 * v11[e]);
    *reinterpret_cast<f32x4 *>(y + pix * y_ld + c) = o;
  }
}
// adjoint as a gather: dx[b][q] = sum over the destination pixels that read q of weight * dy.  Workgroup = 8 channel vectors x 32
// pixel lanes: lane l takes the candidate destination pixels l, l + 32, ... of q's (row range) x (column range) window, the 32 partial
// sums meet in LDS and are added in lane order -- a fixed order, and 32 load chains per output instead of one (the 1 x 1 pyramid stage
// gathers the whole map into one pixel: with 8 row lanes it was the longest glue kernel of a mixed-size training window)
__global__ __launch_bounds__(TB) void bilinear_bwd_kernel(const float *__restrict__ dy, int dy_ld, float *__restrict__ dx, int B, int H, int W, int C4,
                                                          int OH, int OW, int align) {
  __shared__ f32x4 s_p[32][8];
  const int col = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int cgroups = (C4 + 7) / 8;
  for (long job = blockIdx.x; job < (long)B * H * W * cgroups; job += gridDim.x) {
    const int cg = (int)(job % cgroups);
    long r = job / cgroups;
    const long pix = r;
    const int qx = (int)(r % W); r /= W;
    const int qy = (int)(r % H);
    const int b = (int)(r / H);
    const int c4 = cg * 8 + col;
    int ylo, yhi, xlo, xhi;
    bil_cands(qy, H, OH, align, ylo, yhi);
    bil_cands(qx, W, OW, align, xlo, xhi);
    const int nx = xhi - xlo + 1, total = (yhi - ylo + 1) * nx;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c4 < C4)
      for (int idx = pl; idx < total; idx += 32) {
        const int oy = ylo + idx / nx, ox = xlo + idx % nx;
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        bil_src(oy, H, OH, align, y0, y1, wy0, wy1);
        if (y0 != qy && y1 != qy) continue;
        bil_src(ox, W, OW, align, x0, x1, wx0, wx1);
        if (x0 != qx && x1 != qx) continue;
        const float wy = (y0 == qy ? wy0 : 0.f) + (y1 == qy ? wy1 : 0.f);
        const float wx = (x0 == qx ? wx0 : 0.f) + (x1 == qx ? wx1 : 0.f);
        const f32x4 g = *reinterpret_cast<const f32x4 *>(dy + ((long)(b * OH + oy) * OW + ox) * dy_ld + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += (wy * wx) * g[e];
      }
    s_p[pl][col] = acc;
    __syncthreads();
    if (pl == 0 && c4 < C4) {
#pragma unroll 4
      for (int l = 1; l < 32; ++l)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += s_p[l][col][e];
      reinterpret_cast<f32x4 *>(dx)[pix * C4 + c4] = acc;
    }
    __syncthreads();
  }
}

// Adjoint of layers.hip upconv_gather (PSPUpsample through the low-resolution per-tap products): g [B][2h][2w][Cout] is the
// gradient of the pre-activation; dY[b][qy][qx][tap][c] = sum over the upsampled positions u = P + tap - 1 (inside the image) that
// interpolate from (qy, qx) of weight(u -> q) * g[P]
__global__ __launch_bounds__(TB) void upconv_gather_bwd_kernel(const float *__restrict__ g, float *__restrict__ dY, int B, int h, int w, int Cout) {
  const int C4 = Cout / 4, OH = 2 * h, OW = 2 * w;
  GRID_STRIDE(i, (long)B * h * w * 9 * C4) {
    const int c = (int)(i % C4) * 4;
    long r = i / C4;
    const int tap = (int)(r % 9); r /= 9;
    const int qx = (int)(r % w); r /= w;
    const int qy = (int)(r % h);
    const int b = (int)(r / h);
    const int dy = tap / 3, dx = tap - dy * 3;
    int ylo, yhi, xlo, xhi;
    bil_cands(qy, h, OH, 1, ylo, yhi);
    bil_cands(qx, w, OW, 1, xlo, xhi);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int uy = ylo; uy <= yhi; ++uy) {
      const int py = uy - dy + 1;
      if ((unsigned)py >= (unsigned)OH) continue;
      int y0, y1;
      float wy0, wy1;
      bil_src(uy, h, OH, 1, y0, y1, wy0, wy1);
      if (y0 != qy && y1 != qy) continue;
      const float wy = (y0 == qy ? wy0 : 0.f) + (y1 == qy ? wy1 : 0.f);
      for (int ux = xlo; ux <= xhi; ++ux) {
        const int px = ux - dx + 1;
        if ((unsigned)px >= (unsigned)OW) continue;
        int x0, x1;
        float wx0, wx1;
        bil_src(ux, w, OW, 1, x0, x1, wx0, wx1);
        if (x0 != qx && x1 != qx) continue;
        const float wx = (x0 == qx ? wx0 : 0.f) + (x1 == qx ? wx1 : 0.f);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(g + ((long)(b * OH + py) * OW + px) * Cout + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += (wy * wx) * v[e];
      }
    }
    reinterpret_cast<f32x4 *>(dY)[i] = acc;
  }
}

// Adjoint of layers.hip up3_patch (the 3x3 patch of the bilinearly upsampled half-resolution map at every chosen pixel):
// dU[b][qy][qx][c] = sum over points n (ascending) and taps (ascending) whose upsampled position interpolates from (qy, qx) of
// weight * dpatch[b][n][tap][c] -- a gather, so no atomics.  Three launches: decode every point's pixel and the range of
// half-resolution rows / columns its patch can touch; per half-resolution row the ordered list of points that touch it; then a
// thread per (pixel, 4 channels) walks its row's list (a few dozen points instead of all N).
// tab[b][n] = {py | px << 16, rlo | rhi << 16, clo | chi << 16, 0}: the chosen pixel of point n and the range of half-resolution rows /
// columns its 3 x 3 patch of upsampled positions interpolates from
__global__ __launch_bounds__(TB) void up3_decode_kernel(const int64_t *__restrict__ choose, int4 *__restrict__ tab, int B, int h, int wd, int N) {
  const int OH = 2 * h, OW = 2 * wd, HW = OH * OW;
  GRID_STRIDE(i, (long)B * N) {
    long pix = choose[i];
    pix = pix < 0 ? 0 : (pix >= HW ? HW - 1 : pix);
    const int py = (int)(pix / OW), px = (int)(pix % OW);
    int i0, i1, rlo, rhi, clo, chi;
    float l0, l1;
    bil_src(max(py - 1, 0), h, OH, 1, rlo, i1, l0, l1);
    bil_src(min(py + 1, OH - 1), h, OH, 1, i0, rhi, l0, l1);
    bil_src(max(px - 1, 0), wd, OW, 1, clo, i1, l0, l1);
    bil_src(min(px + 1, OW - 1), wd, OW, 1, i0, chi, l0, l1);
    tab[i] = make_int4(py | (px << 16), rlo | (rhi << 16), clo | (chi << 16), 0);
  }
}

// rows[b][qy][...] = the points (ascending n) whose patch touches half-resolution row qy, cnt[b][qy] their number: a workgroup per row
// scans the table once, 256 points per round, and compacts the hits in order (wave ballots + a scan over the 4 waves)
__global__ __launch_bounds__(TB) void up3_rowlist_kernel(const int4 *__restrict__ tab, int *__restrict__ rows, int *__restrict__ cnt, int h, int N) {
  __shared__ int s_w[4];
  const int b = blockIdx.y, qy = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int *out = rows + ((size_t)b * h + qy) * N;
  int base = 0;
  for (int n0 = 0; n0 < N; n0 += TB) {
    const int n = n0 + threadIdx.x;
    bool hit = false;
    if (n < N) {
      const int4 e = tab[(size_t)b * N + n];
      hit = qy >= (e.y & 0xffff) && qy <= (e.y >> 16);
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int before = base;
    for (int w2 = 0; w2 < wave; ++w2) before += s_w[w2];
    if (hit) out[before + __popcll(m & ((1ull << lane) - 1ull))] = n;
    base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) cnt[b * h + qy] = base;
}

__global__ __launch_bounds__(TB) void up3_patch_bwd_kernel(const float *__restrict__ dpatch, const int4 *__restrict__ tab, const int *__restrict__ rows,
                                                           const int *__restrict__ cnt, float *__restrict__ dU, int h, int wd, int N, int Npad) {
  const int OH = 2 * h, OW = 2 * wd;
  const int b = blockIdx.z, qy = blockIdx.y;
  const int c4 = threadIdx.x & 15, qx = blockIdx.x * (TB / 16) + (threadIdx.x >> 4);
  if (qx >= wd) return;
  const int *list = rows + ((size_t)b * h + qy) * N;
  const int count = cnt[b * h + qy];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < count; ++i) {
    const int j = list[i];
    const int4 e4 = tab[(size_t)b * N + j];
    if (qx < (e4.z & 0xffff) || qx > (e4.z >> 16)) continue;
    const float *row = dpatch + ((size_t)b * Npad + j) * 576 + c4 * 4;
    const int py = e4.x & 0xffff, px = e4.x >> 16;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int uy = py + dy - 1;
      if ((unsigned)uy >= (unsigned)OH) continue;
      int y0, y1;
      float wy0, wy1;
      bil_src(uy, h, OH, 1, y0, y1, wy0, wy1);
      if (y0 != qy && y1 != qy) continue;
      const float wy = (y0 == qy ? wy0 : 0.f) + (y1 == qy ? wy1 : 0.f);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ux = px + dx - 1;
        if ((unsigned)ux >= (unsigned)OW) continue;
        int x0, x1;
        float wx0, wx1;
        bil_src(ux, wd, OW, 1, x0, x1, wx0, wx1);
        if (x0 != qx && x1 != qx) continue;
        const float wx = (x0 == qx ? wx0 : 0.f) + (x1 == qx ? wx1 : 0.f);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(row + (dy * 3 + dx) * 64);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += (wy * wx) * v[e];
      }
    }
  }
  *reinterpret_cast<f32x4 *>(dU + (((size_t)b * h + qy) * wd + qx) * 64 + c4 * 4) = acc;
}

// ------------------------------------------------------------------------------------------------
// Memory-bound kernels over ALL crop-size buckets of a level in one launch: a bucket table travels in the kernel arguments, an element
// finds its bucket by a scan of <= 16 row bounds (the buckets' pixel rows are concatenated, bucket g = B[g] maps of H[g] x W[g] from
// row row0[g]; its frames are b0[g] .. of the pass).  Same arithmetic per element as the per-bucket kernels above.
// ------------------------------------------------------------------------------------------------
constexpr int TAB_MAX = 16;
struct BTab {
  int n;
  int B[TAB_MAX], H[TAB_MAX], W[TAB_MAX], b0[TAB_MAX];
  long row0[TAB_MAX], row1[TAB_MAX];      // first pixel row, one past the last
  long aux0[TAB_MAX];                     // first row of the bucket in a second level (pooled / convolved maps), where a kernel needs one
};
__device__ inline int tab_of_row(const BTab &t, long row) {
  int g = 0;
  while (g + 1 < t.n && row >= t.row1[g]) ++g;
  return g;
}
__device__ inline int tab_of_frame(const BTab &t, int frame) {
  int g = 0;
  while (g + 1 < t.n && frame >= t.b0[g + 1]) ++g;
  return g;
}

// y[r][c] = x[r][c] * scale[frame(r)][c]  (Dropout2d and its adjoint)
__global__ __launch_bounds__(TB) void channel_scale_multi_kernel(const float *__restrict__ x, int x_ld, const float *__restrict__ scale, float *__restrict__ y,
                                                                 int y_ld, int C4, const BTab tab) {
  const long r_lo = tab.row0[0], nrow = tab.row1[tab.n - 1] - r_lo;
  GRID_STRIDE(i, nrow * C4) {
    const long r = r_lo + i / C4;
    const int c = (int)(i % C4) * 4;
    const int g = tab_of_row(tab, r);
    const int frame = tab.b0[g] + (int)((r - tab.row0[g]) / ((long)tab.H[g] * tab.W[g]));
    const f32x4 v = *reinterpret_cast<const f32x4 *>(x + r * x_ld + c), sc = *reinterpret_cast<const f32x4 *>(scale + (size_t)frame * C4 * 4 + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = v[e] * sc[e];
    *reinterpret_cast<f32x4 *>(y + r * y_ld + c) = o;
  }
}

// AdaptiveAvgPool2d adjoint of the FOUR pyramid stages (sizes 1, 2, 3, 6) at once: dx[pix] (+)= sum_s sum over the stage's bins that contain
// the pixel of dy_s[frame][bin] / |bin| (stages ascending, bins row-major: a fixed order); dy_s = [frames][s*s][C] blocks
struct Ptr4 { const float *p[4]; };
struct MPtr4 { float *p[4]; };
__global__ __launch_bounds__(TB) void pool_bwd_all_kernel(const Ptr4 dy, float *__restrict__ dx, int dx_ld, int C4, int accumulate, const BTab tab) {
  const long r_lo = tab.row0[0], nrow = tab.row1[tab.n - 1] - r_lo;
  GRID_STRIDE(i, nrow * C4) {
    const long r = r_lo + i / C4;
    const int c4 = (int)(i % C4);
    const int g = tab_of_row(tab, r);
    const int H = tab.H[g], W = tab.W[g];
    long l = r - tab.row0[g];
    const int xx = (int)(l % W); l /= W;
    const int yy = (int)(l % H);
    const int frame = tab.b0[g] + (int)(l / H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int si = 0; si < 4; ++si) {
      const int s = si == 0 ? 1 : si == 1 ? 2 : si == 2 ? 3 : 6;
      const float *d = dy.p[si] + (size_t)frame * s * s * C4 * 4;
      for (int bi = max(0, yy * s / H - 1); bi <= min(s - 1, yy * s / H + 1); ++bi) {
        const int y0 = (bi * H) / s, y1 = ((bi + 1) * H + s - 1) / s;
        if (yy < y0 || yy >= y1) continue;
        for (int bj = max(0, xx * s / W - 1); bj <= min(s - 1, xx * s / W + 1); ++bj) {
          const int x0 = (bj * W) / s, x1 = ((bj + 1) * W + s - 1) / s;
          if (xx < x0 || xx >= x1) continue;
          const f32x4 v = reinterpret_cast<const f32x4 *>(d)[(size_t)(bi * s + bj) * C4 + c4];
          const float cnt = (float)((y1 - y0) * (x1 - x0));
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += v[e] / cnt;
        }
      }
    }
    float *o = dx + r * dx_ld + c4 * 4;
    if (accumulate) {
      const f32x4 old = *reinterpret_cast<const f32x4 *>(o);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = old[e] + acc[e];
    }
    *reinterpret_cast<f32x4 *>(o) = acc;
  }
}

// the four pyramid priors (bilinear, align_corners = False, lib/pspnet.py:22) of every bucket: y[r][si * C + c] from z_si [frames][s*s][C]
__global__ __launch_bounds__(TB) void bilinear_fwd_all_kernel(const Ptr4 z, float *__restrict__ y, int y_ld, int C4, const BTab tab) {
  const long r_lo = tab.row0[0], nrow = tab.row1[tab.n - 1] - r_lo;
  GRID_STRIDE(i, nrow * 4 * C4) {
    const int c4 = (int)(i % C4);
    const int si = (int)((i / C4) % 4);
    const long r = r_lo + i / (4 * C4);
    const int g = tab_of_row(tab, r);
    const int OH = tab.H[g], OW = tab.W[g];
    long l = r - tab.row0[g];
    const int ox = (int)(l % OW); l /= OW;
    const int oy = (int)(l % OH);
    const int frame = tab.b0[g] + (int)(l / OH);
    const int s = si == 0 ? 1 : si == 1 ? 2 : si == 2 ? 3 : 6;
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    bil_src(oy, s, OH, 0, y0, y1, wy0, wy1);
    bil_src(ox, s, OW, 0, x0, x1, wx0, wx1);
    const f32x4 *p = reinterpret_cast<const f32x4 *>(z.p[si]) + (size_t)frame * s * s * C4 + c4;
    const f32x4 v00 = p[(size_t)(y0 * s + x0) * C4], v01 = p[(size_t)(y0 * s + x1) * C4], v10 = p[(size_t)(y1 * s + x0) * C4], v11 = p[(size_t)(y1 * s + x1) * C4];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = wy0 * (wx0 * v00[e] + wx1 * v01[e]) + wy1 * (wx0 * v10[e] + wx1 * v11[e]);
    *reinterpret_cast<f32x4 *>(y + r * y_ld + (size_t)si * C4 * 4 + c4 * 4) = o;
  }
}

// their adjoint, all four stages and all buckets: job = (stage, frame, stage pixel q, group of 8 channel vectors); 32 pixel lanes per output
// like bilinear_bwd_kernel (same candidate order, same fixed reduction); dz_si = [frames][s*s][C]
__global__ __launch_bounds__(TB) void bilinear_bwd_all_kernel(const float *__restrict__ dy, int dy_ld, const MPtr4 dz, int frames, int C4, const BTab tab) {
  __shared__ f32x4 s_p[32][8];
  const int col = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int cgroups = (C4 + 7) / 8;
  const long per_frame = 50L * cgroups;                   // 1 + 4 + 9 + 36 stage pixels
  for (long job = blockIdx.x; job < (long)frames * per_frame; job += gridDim.x) {
    const int frame = tab.b0[0] + (int)(job / per_frame);          // (`frames` counts the table's frames; dz / dy are indexed by the absolute frame)
    long rem = job - (job / per_frame) * per_frame;
    const int cg = (int)(rem % cgroups);
    int q = (int)(rem / cgroups);
    int si = 0, s = 1;
    if (q >= 14) { si = 3; s = 6; q -= 14; } else if (q >= 5) { si = 2; s = 3; q -= 5; } else if (q >= 1) { si = 1; s = 2; q -= 1; }
    const int qy = q / s, qx = q - qy * s;
    const int g = tab_of_frame(tab, frame);
    const int OH = tab.H[g], OW = tab.W[g];
    const float *src = dy + (tab.row0[g] + (long)(frame - tab.b0[g]) * OH * OW) * dy_ld + (size_t)si * C4 * 4;
    const int c4 = cg * 8 + col;
    int ylo, yhi, xlo, xhi;
    bil_cands(qy, s, OH, 0, ylo, yhi);
    bil_cands(qx, s, OW, 0, xlo, xhi);
    const int nx = xhi - xlo + 1, total = (yhi - ylo + 1) * nx;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c4 < C4)
      for (int idx = pl; idx < total; idx += 32) {
        const int oy = ylo + idx / nx, ox = xlo + idx % nx;
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        bil_src(oy, s, OH, 0, y0, y1, wy0, wy1);
        if (y0 != qy && y1 != qy) continue;
        bil_src(ox, s, OW, 0, x0, x1, wx0, wx1);
        if (x0 != qx && x1 != qx) continue;
        const float wy = (y0 == qy ? wy0 : 0.f) + (y1 == qy ? wy1 : 0.f);
        const float wx = (x0 == qx ? wx0 : 0.f) + (x1 == qx ? wx1 : 0.f);
        const f32x4 gv = *reinterpret_cast<const f32x4 *>(src + ((long)oy * OW + ox) * dy_ld + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += (wy * wx) * gv[e];
      }
    s_p[pl][col] = acc;
    __syncthreads();
    if (pl == 0 && c4 < C4) {
#pragma unroll 4
      for (int l = 1; l < 32; ++l)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += s_p[l][col][e];
      reinterpret_cast<f32x4 *>(dz.p[si])[((size_t)frame * s * s + q) * C4 + c4] = acc;
    }
    __syncthreads();
  }
}

// upconv_gather_bwd_kernel over all buckets: g rows live at the upsampled level (4 x the low-resolution rows of each bucket)
__global__ __launch_bounds__(TB) void upconv_gather_bwd_multi_kernel(const float *__restrict__ gsrc, float *__restrict__ dY, int Cout, const BTab tab) {
  const int C4 = Cout / 4;
  const long r_lo = tab.row0[0], nrow = tab.row1[tab.n - 1] - r_lo;
  GRID_STRIDE(i, nrow * 9 * C4) {
    const int c = (int)(i % C4) * 4;
    long r = i / C4;
    const int tap = (int)(r % 9); r /= 9;
    const long row = r_lo + r;
    const int gi = tab_of_row(tab, row);
    const int h = tab.H[gi], w = tab.W[gi], OH = 2 * h, OW = 2 * w;
    long l = row - tab.row0[gi];
    const int qx = (int)(l % w); l /= w;
    const int qy = (int)(l % h);
    const int b = (int)(l / h);
    const float *g = gsrc + 4 * tab.row0[gi] * Cout;
    const int dy = tap / 3, dx = tap - dy * 3;
    int ylo, yhi, xlo, xhi;
    bil_cands(qy, h, OH, 1, ylo, yhi);
    bil_cands(qx, w, OW, 1, xlo, xhi);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // the column candidates' weights once per thread (they were recomputed under every row candidate); same products, same order
    constexpr int MAXC = 8;      // (a source column feeds at most 5 upsampled columns at scale 2)
    float wxs[MAXC];
    int pxs[MAXC];
    int ncx = 0;
    for (int ux = xlo; ux <= xhi && ncx < MAXC; ++ux) {
      const int px = ux - dx + 1;
      if ((unsigned)px >= (unsigned)OW) continue;
      int x0, x1;
      float wx0, wx1;
      bil_src(ux, w, OW, 1, x0, x1, wx0, wx1);
      if (x0 != qx && x1 != qx) continue;
      wxs[ncx] = (x0 == qx ? wx0 : 0.f) + (x1 == qx ? wx1 : 0.f);
      pxs[ncx++] = px;
    }
    for (int uy = ylo; uy <= yhi; ++uy) {
      const int py = uy - dy + 1;
      if ((unsigned)py >= (unsigned)OH) continue;
      int y0, y1;
      float wy0, wy1;
      bil_src(uy, h, OH, 1, y0, y1, wy0, wy1);
      if (y0 != qy && y1 != qy) continue;
      const float wy = (y0 == qy ? wy0 : 0.f) + (y1 == qy ? wy1 : 0.f);
      for (int k = 0; k < ncx; ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(g + ((long)(b * OH + py) * OW + pxs[k]) * Cout + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += (wy * wxs[k]) * v[e];
      }
    }
    reinterpret_cast<f32x4 *>(dY)[(row * 9 + tap) * C4 + c / 4] = acc;
  }
}

// MaxPool2d(3, stride 2, pad 1) adjoint over all buckets (trainops.hip maxpool3s2_bwd_kernel: first-maximum rule); tab = the INPUT level,
// aux0 = the buckets' first rows at the pooled level
__global__ __launch_bounds__(TB) void maxpool3s2_bwd_multi_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dx, int C,
                                                                  const BTab tab) {
  // thread = 4 channels of one input pixel (16-byte loads; the per-channel decisions and the order of the additions are those of the
  // one-channel form: 79 -> 25 us on the stem's 8 x 80 x 80 x 64 map)
  const int C4 = C / 4;
  const long r_lo = tab.row0[0], nrow = tab.row1[tab.n - 1] - r_lo;
  GRID_STRIDE(i, nrow * C4) {
    const int c = (int)(i % C4) * 4;
    const long row = r_lo + i / C4;
    const int g = tab_of_row(tab, row);
    const int H = tab.H[g], W = tab.W[g], OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    long l = row - tab.row0[g];
    const int ix = (int)(l % W); l /= W;
    const int iy = (int)(l % H);
    const int b = (int)(l / H);
    const float *xb = x + (tab.row0[g] + (long)b * H * W) * C + c;
    const float *dyb = dy + (tab.aux0[g] + (long)b * OH * OW) * C + c;
    const f32x4 xv = *reinterpret_cast<const f32x4 *>(xb + ((long)iy * W + ix) * C);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int oy = (iy + 1) / 2 - 1 < 0 ? 0 : (iy + 1) / 2 - 1; oy <= (iy + 1) / 2 && oy < OH; ++oy) {
      if (iy < oy * 2 - 1 || iy > oy * 2 + 1) continue;
      for (int ox = (ix + 1) / 2 - 1 < 0 ? 0 : (ix + 1) / 2 - 1; ox <= (ix + 1) / 2 && ox < OW; ++ox) {
        if (ix < ox * 2 - 1 || ix > ox * 2 + 1) continue;
        bool win[4] = {true, true, true, true};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int yy = oy * 2 - 1 + ky;
          if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int xx = ox * 2 - 1 + kx;
            if ((unsigned)xx >= (unsigned)W) continue;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xb + ((long)yy * W + xx) * C);
            const bool earlier = yy < iy || (yy == iy && xx < ix);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (v[e] > xv[e] || (earlier && v[e] == xv[e])) win[e] = false;
          }
        }
        const f32x4 d = *reinterpret_cast<const f32x4 *>(dyb + ((long)oy * OW + ox) * C);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += win[e] ? d[e] : 0.f;
      }
    }
    *reinterpret_cast<f32x4 *>(dx + row * C + c) = acc;
  }
}

// Data gradient of a STRIDED convolution from its per-tap products (col2im as a gather): dcol[m][tap * C + c] = sum_n dY[m][n] w[n][tap][c]
// for every output pixel m (one GEMM over the rows of all buckets); an input pixel collects the <= ceil(k / stride)^2 (tap, output pixel)
// pairs that read it, taps in row-major order (fixed order).  tab = the INPUT level, aux0 = the buckets' first rows at the output level.
__global__ __launch_bounds__(TB) void col2im_multi_kernel(const float *__restrict__ dcol, float *__restrict__ dx, int dx_ld, int C, int k, int stride, int pad,
                                                          int dil, int accumulate, const BTab tab) {
  const int C4 = C / 4;
  const long r_lo = tab.row0[0], nrow = tab.row1[tab.n - 1] - r_lo;
  GRID_STRIDE(i, nrow * C4) {
    const int c4 = (int)(i % C4);
    const long row = r_lo + i / C4;
    const int g = tab_of_row(tab, row);
    const int H = tab.H[g], W = tab.W[g];
    const int OH = (H + 2 * pad - dil * (k - 1) - 1) / stride + 1, OW = (W + 2 * pad - dil * (k - 1) - 1) / stride + 1;
    long l = row - tab.row0[g];
    const int ix = (int)(l % W); l /= W;
    const int iy = (int)(l % H);
    const int b = (int)(l / H);
    const float *src = dcol + (tab.aux0[g] + (long)b * OH * OW) * (size_t)(k * k * C);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ky = 0; ky < k; ++ky) {
      const int ty = iy + pad - ky * dil;
      if (ty < 0 || ty % stride) continue;
      const int oy = ty / stride;
      if (oy >= OH) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int tx = ix + pad - kx * dil;
        if (tx < 0 || tx % stride) continue;
        const int ox = tx / stride;
        if (ox >= OW) continue;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + ((long)oy * OW + ox) * (size_t)(k * k * C) + (size_t)(ky * k + kx) * C + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[e];
      }
    }
    float *o = dx + row * dx_ld + c4 * 4;
    if (accumulate) {
      const f32x4 old = *reinterpret_cast<const f32x4 *>(o);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = old[e] + acc[e];
    }
    *reinterpret_cast<f32x4 *>(o) = acc;
  }
}

// the three up_3 patch-adjoint kernels over all buckets (tab = the half-resolution level; frames of all buckets in one grid, rows / columns
// beyond a bucket's map exit): row lists and counts are laid out with the LARGEST map height hmax per frame
__global__ __launch_bounds__(TB) void up3_decode_multi_kernel(const int64_t *__restrict__ choose, int4 *__restrict__ tabo, int frames, int N, const BTab tab) {
  GRID_STRIDE(i, (long)frames * N) {          // (choose / tabo start at the table's first frame)
    const int g = tab_of_frame(tab, tab.b0[0] + (int)(i / N));
    const int h = tab.H[g], wd = tab.W[g];
    const int OH = 2 * h, OW = 2 * wd, HW = OH * OW;
    long pix = choose[i];
    pix = pix < 0 ? 0 : (pix >= HW ? HW - 1 : pix);
    const int py = (int)(pix / OW), px = (int)(pix % OW);
    int i0, i1, rlo, rhi, clo, chi;
    float l0, l1;
    bil_src(max(py - 1, 0), h, OH, 1, rlo, i1, l0, l1);
    bil_src(min(py + 1, OH - 1), h, OH, 1, i0, rhi, l0, l1);
    bil_src(max(px - 1, 0), wd, OW, 1, clo, i1, l0, l1);
    bil_src(min(px + 1, OW - 1), wd, OW, 1, i0, chi, l0, l1);
    tabo[i] = make_int4(py | (px << 16), rlo | (rhi << 16), clo | (chi << 16), 0);
  }
}
__global__ __launch_bounds__(TB) void up3_rowlist_multi_kernel(const int4 *__restrict__ tabi, int *__restrict__ rows, int *__restrict__ cnt, int hmax, int N,
                                                               const BTab tab) {
  __shared__ int s_w[4];
  const int b = blockIdx.y, qy = blockIdx.x;          // b: frame relative to the table's first (tabi / rows / cnt start there)
  if (qy >= tab.H[tab_of_frame(tab, tab.b0[0] + b)]) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int *out = rows + ((size_t)b * hmax + qy) * N;
  int base = 0;
  for (int n0 = 0; n0 < N; n0 += TB) {
    const int n = n0 + threadIdx.x;
    bool hit = false;
    if (n < N) {
      const int4 e = tabi[(size_t)b * N + n];
      hit = qy >= (e.y & 0xffff) && qy <= (e.y >> 16);
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int before = base;
    for (int w2 = 0; w2 < wave; ++w2) before += s_w[w2];
    if (hit) out[before + __popcll(m & ((1ull << lane) - 1ull))] = n;
    base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) cnt[b * hmax + qy] = base;
}
__global__ __launch_bounds__(TB) void up3_patch_bwd_multi_kernel(const float *__restrict__ dpatch, const int4 *__restrict__ tabi, const int *__restrict__ rows,
                                                                 const int *__restrict__ cnt, float *__restrict__ dU, int hmax, int N, int Npad, const BTab tab) {
  const int b = blockIdx.z, qy = blockIdx.y;          // b: frame relative to the table's first (dpatch / tabi / rows / cnt start there)
  const int g = tab_of_frame(tab, tab.b0[0] + b);
  const int h = tab.H[g], wd = tab.W[g];
  const int OH = 2 * h, OW = 2 * wd;
  const int c4 = threadIdx.x & 15, qx = blockIdx.x * (TB / 16) + (threadIdx.x >> 4);
  if (qy >= h || qx >= wd) return;
  const int *list = rows + ((size_t)b * hmax + qy) * N;
  const int count = cnt[b * hmax + qy];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < count; ++i) {
    const int j = list[i];
    const int4 e4 = tabi[(size_t)b * N + j];
    if (qx < (e4.z & 0xffff) || qx > (e4.z >> 16)) continue;
    const float *row = dpatch + ((size_t)b * Npad + j) * 576 + c4 * 4;
    const int py = e4.x & 0xffff, px = e4.x >> 16;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int uy = py + dy - 1;
      if ((unsigned)uy >= (unsigned)OH) continue;
      int y0, y1;
      float wy0, wy1;
      bil_src(uy, h, OH, 1, y0, y1, wy0, wy1);
      if (y0 != qy && y1 != qy) continue;
      const float wy = (y0 == qy ? wy0 : 0.f) + (y1 == qy ? wy1 : 0.f);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ux = px + dx - 1;
        if ((unsigned)ux >= (unsigned)OW) continue;
        int x0, x1;
        float wx0, wx1;
        bil_src(ux, wd, OW, 1, x0, x1, wx0, wx1);
        if (x0 != qx && x1 != qx) continue;
        const float wx = (x0 == qx ? wx0 : 0.f) + (x1 == qx ? wx1 : 0.f);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(row + (dy * 3 + dx) * 64);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += (wy * wx) * v[e];
      }
    }
  }
  *reinterpret_cast<f32x4 *>(dU + (tab.row0[g] + ((long)(tab.b0[0] + b - tab.b0[g]) * h + qy) * wd + qx) * 64 + c4 * 4) = acc;
}

// Conv1d(3, 64, 1) on the cloud (lib/network.py:54): partial sums of dW [64][3], db [64] over a chunk of 64 points;
// g = the masked gradient of its output, a [B*Npad][64] view.  part[chunk][64][4] = (dW_x, dW_y, dW_z, db)
__global__ __launch_bounds__(64) void cloud_conv1_bwd_kernel(const float *__restrict__ g, int g_ld, const float *__restrict__ cloud, int B, int N,
                                                             int Npad, float *__restrict__ part) {
  const int chunks = (N + 63) / 64;
  const int b = blockIdx.x / chunks, n0 = (blockIdx.x % chunks) * 64, n1 = min(N, n0 + 64);
  const int c = threadIdx.x;
  float ax = 0.f, ay = 0.f, az = 0.f, ab = 0.f;
  for (int n = n0; n < n1; ++n) {
    const float gv = g[((size_t)b * Npad + n) * g_ld + c];
    const float *p = cloud + ((size_t)b * N + n) * 3;
    ax += gv * p[0]; ay += gv * p[1]; az += gv * p[2]; ab += gv;
  }
  float *o = part + ((size_t)blockIdx.x * 64 + c) * 4;
  o[0] = ax; o[1] = ay; o[2] = az; o[3] = ab;
}
__global__ __launch_bounds__(64) void cloud_conv1_bwd_finish_kernel(const float *__restrict__ part, int count, float *__restrict__ dw, float *__restrict__ db) {
  const int c = threadIdx.x;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int i = 0; i < count; ++i) {        // (eight loads in flight; the additions stay in order)
    const f32x4 v = *reinterpret_cast<const f32x4 *>(part + ((size_t)i * 64 + c) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] += v[e];
  }
  dw[c * 3 + 0] += a[0]; dw[c * 3 + 1] += a[1]; dw[c * 3 + 2] += a[2];
  db[c] += a[3];
}

// AvgPool1d(N) adjoint + ReLU mask of conv6's output: g6[r][c] = (n < N && x6[r][c] > 0) ? dap[b][c] / N : 0
__global__ __launch_bounds__(TB) void mask_bcast_kernel(const float *__restrict__ x6, const float *__restrict__ dap, float *__restrict__ g6, int B, int N,
                                                        int Npad, int C4) {
  const float inv = 1.f / (float)N;
  GRID_STRIDE(i, (long)B * Npad * C4) {
    const int c = (int)(i % C4);
    const long r = i / C4;
    const int b = (int)(r / Npad), n = (int)(r - (long)b * Npad);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
      const f32x4 x = reinterpret_cast<const f32x4 *>(x6)[i], d = reinterpret_cast<const f32x4 *>(dap)[(long)b * C4 + c];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = x[e] > 0.f ? d[e] * inv : 0.f;
    }
    reinterpret_cast<f32x4 *>(g6)[i] = o;
  }
}

// s[b][c] = sum over the Npad rows of object b of g[.][c]: 32 columns x 8 row lanes per workgroup, rows in ascending order per lane
__global__ __launch_bounds__(256) void colsum_obj_kernel(const float *__restrict__ g, int g_ld, float *__restrict__ s, int Npad, int C, long rows_total,
                                                         int accumulate = 0) {
  __shared__ float s_p[8][32];
  const int col = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + col, b = blockIdx.y;
  float a = 0.f;
  if (c < C) {
    const long left = rows_total - (long)b * Npad;
    const int rmax = (int)(left < Npad ? left : Npad);
#pragma unroll 8
    for (int r = rl; r < rmax; r += 8) a += g[((size_t)b * Npad + r) * g_ld + c];      // (loads ahead, the additions in row order)
  }
  s_p[rl][col] = a;
  __syncthreads();
  if (rl == 0 && c < C) {
#pragma unroll
    for (int l = 1; l < 8; ++l) a += s_p[l][col];
    s[(size_t)b * C + c] = accumulate ? s[(size_t)b * C + c] + a : a;
  }
}

// global-feature half of head layer 1 (the 1024 broadcast channels folded into a per-object bias, engine.hip posenet_points):
//   gbias[b][o] = Wg[o] . ap[b] + bias[o]   =>   dbias[o] += sum_b s[b][o],  dWg[o][j] += sum_b s[b][o] ap[b][j],  dap[b][j] = sum_o Wg[o][j] s[b][o]
// (the last one is a [B x 1920] x [1920 x 1024] product: the GEMM kernel on the cached transpose of Wg)
// with s[b][o] = the column sums over object b's points of the masked gradient of head layer 1's output
__global__ __launch_bounds__(TB) void head1_global_wgrad_kernel(const float *__restrict__ s, const float *__restrict__ ap, float *__restrict__ dWg,
                                                                float *__restrict__ dbias, int B, int O, int J4) {
  GRID_STRIDE(i, (long)O * J4) {
    const int j = (int)(i % J4);
    const int o = (int)(i / J4);
    f32x4 acc = reinterpret_cast<const f32x4 *>(dWg)[i];
    float sb = 0.f;
    for (int b = 0; b < B; ++b) {
      const float sv = s[(size_t)b * O + o];
      const f32x4 a = reinterpret_cast<const f32x4 *>(ap)[(size_t)b * J4 + j];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += sv * a[e];
      sb += sv;
    }
    reinterpret_cast<f32x4 *>(dWg)[i] = acc;
    if (j == 0) dbias[o] += sb;
  }
}
// last head layer for the frame's object only (layers.hip head_final): outputs j = 0..3 quaternion, 4..6 translation, 7 confidence
// (sigmoid).  dz[b][n][j] = upstream gradient of the pre-sigmoid outputs; dh3 = dz . W rows; partial dW rows over chunks of 128 points.
__global__ __launch_bounds__(TB) void head_final_bwd_kernel(const float *__restrict__ d_r, const float *__restrict__ d_t, const float *__restrict__ d_c,
                                                            const float *__restrict__ out_c, const float *__restrict__ w_r, const float *__restrict__ w_t,
                                                            const float *__restrict__ w_c, const int64_t *__restrict__ obj, int num_obj,
                                                            float *__restrict__ dh3, float *__restrict__ dz, int B, int N, int Npad) {
  GRID_STRIDE(i, (long)B * Npad * 96) {          // thread = (row, one float4 of the 384 feature columns)
    const int k4 = (int)(i % 96);
    const long r = i / 96;
    const int b = (int)(r / Npad), n = (int)(r - (long)b * Npad);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
      long ob = obj[b];
      ob = ob < 0 ? 0 : (ob >= num_obj ? num_obj - 1 : ob);
      const size_t p = (size_t)b * N + n;
      const int tower = k4 / 32, k = (k4 % 32) * 4;
      if (tower == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float gz = d_r[p * 4 + j];
          const f32x4 wv = *reinterpret_cast<const f32x4 *>(w_r + (ob * 4 + j) * 128 + k);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += gz * wv[e];
          if (k4 == 0) dz[p * 8 + j] = gz;
        }
      } else if (tower == 1) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float gz = d_t[p * 3 + j];
          const f32x4 wv = *reinterpret_cast<const f32x4 *>(w_t + (ob * 3 + j) * 128 + k);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += gz * wv[e];
          if (k4 == 32) dz[p * 8 + 4 + j] = gz;
        }
      } else {
        const float cv = out_c[p];
        const float gz = d_c[p] * cv * (1.f - cv);
        const f32x4 wv = *reinterpret_cast<const f32x4 *>(w_c + ob * 128 + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = gz * wv[e];
        if (k4 == 64) dz[p * 8 + 7] = gz;
      }
    }
    *reinterpret_cast<f32x4 *>(dh3 + r * 384 + k4 * 4) = o;
  }
}
// part[b][chunk][j][k] = sum over the chunk's points of dz[n][j] * h3[n][tower(j)*128 + k]; block = (chunk, b), thread = (j pair, k)
__global__ __launch_bounds__(TB) void head_final_wgrad_kernel(const float *__restrict__ dz, const float *__restrict__ h3, float *__restrict__ part,
                                                              float *__restrict__ zpart, int N, int Npad, int chunks) {
  const int b = blockIdx.y, ch = blockIdx.x, n0 = ch * 128, n1 = min(N, n0 + 128);
  const int k = threadIdx.x & 127, jh = threadIdx.x >> 7;       // jh 0: outputs 0..3 (r), 1: outputs 4..7 (t, c)
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  for (int n = n0; n < n1; ++n) {
    const float *z = dz + ((size_t)b * N + n) * 8 + jh * 4;
    const float *hrow = h3 + ((size_t)b * Npad + n) * 384;
    const float hr = hrow[(jh == 0 ? 0 : 128) + k], hc = hrow[256 + k];
    a[0] += z[0] * hr; a[1] += z[1] * hr; a[2] += z[2] * hr;
    a[3] += z[3] * (jh == 0 ? hr : hc);
  }
  float *o = part + (((size_t)b * chunks + ch) * 8 + jh * 4) * 128 + k;
  o[0] = a[0]; o[128] = a[1]; o[256] = a[2]; o[384] = a[3];
  if (k == 0) {                       // the chunk's sums of dz (bias gradient)
    float zs[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n = n0; n < n1; ++n) {
      const float *z = dz + ((size_t)b * N + n) * 8 + jh * 4;
      zs[0] += z[0]; zs[1] += z[1]; zs[2] += z[2]; zs[3] += z[3];
    }
    float *zo = zpart + ((size_t)b * chunks + ch) * 8 + jh * 4;
    zo[0] = zs[0]; zo[1] = zs[1]; zo[2] = zs[2]; zo[3] = zs[3];
  }
}
// thread = (j, k): frames in ascending order add their chunks (ascending) into the rows of their object; db from dz directly
__global__ __launch_bounds__(TB) void head_final_wgrad_finish_kernel(const float *__restrict__ part, const float *__restrict__ zpart,
                                                                     const int64_t *__restrict__ obj, int num_obj, float *__restrict__ dw_r,
                                                                     float *__restrict__ db_r, float *__restrict__ dw_t, float *__restrict__ db_t,
                                                                     float *__restrict__ dw_c, float *__restrict__ db_c, int B, int N, int chunks) {
  const int i = blockIdx.x * TB + threadIdx.x;
  if (i >= 8 * 128) return;
  const int j = i >> 7, k = i & 127;
  for (int b0 = 0; b0 < B; b0 += 4) {          // four frames' partial sums are gathered first (their loads in flight together), then added in frame order
    float a4[4] = {0.f, 0.f, 0.f, 0.f}, s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = b0 + u;
      if (b >= B) break;
#pragma unroll 8
      for (int ch = 0; ch < chunks; ++ch) a4[u] += part[(((size_t)b * chunks + ch) * 8 + j) * 128 + k];
      if (k == 0) {
#pragma unroll 8
        for (int ch = 0; ch < chunks; ++ch) s4[u] += zpart[((size_t)b * chunks + ch) * 8 + j];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int b = b0 + u;
      if (b >= B) break;
      long ob = obj[b];
      ob = ob < 0 ? 0 : (ob >= num_obj ? num_obj - 1 : ob);
      float *dst = j < 4 ? dw_r + (ob * 4 + j) * 128 : j < 7 ? dw_t + (ob * 3 + (j - 4)) * 128 : dw_c + ob * 128;
      dst[k] += a4[u];
      if (k == 0) {
        float *bd = j < 4 ? db_r + ob * 4 + j : j < 7 ? db_t + ob * 3 + (j - 4) : db_c + ob;
        *bd += s4[u];
      }
    }
  }
}

// refiner tail (lib/network.py:199-204): out_r [B][4], out_t [B][3] = conv3_r / conv3_t rows of the frame's object on f2 [B][256]
__global__ __launch_bounds__(64) void refiner_tail_fwd_kernel(const float *__restrict__ f2, const float *__restrict__ w_r, const float *__restrict__ b_r,
                                                              const float *__restrict__ w_t, const float *__restrict__ b_t, const int64_t *__restrict__ obj,
                                                              int num_obj, float *__restrict__ out_r, float *__restrict__ out_t, int B) {
  const int b = blockIdx.x, j = threadIdx.x;
  if (b >= B || j >= 7) return;
  long ob = obj[b];
  ob = ob < 0 ? 0 : (ob >= num_obj ? num_obj - 1 : ob);
  const float *w = j < 4 ? w_r + (ob * 4 + j) * 128 : w_t + (ob * 3 + (j - 4)) * 128;
  const float *x = f2 + (size_t)b * 256 + (j < 4 ? 0 : 128);
  float a = 0.f;
  for (int k = 0; k < 128; ++k) a += x[k] * w[k];
  if (j < 4) out_r[b * 4 + j] = a + b_r[ob * 4 + j];
  else out_t[b * 3 + (j - 4)] = a + b_t[ob * 3 + (j - 4)];
}
// one workgroup: frames in ascending order; df2[b][tower*128 + k] = sum_j dz[j] W[j][k]; dW rows += dz[j] f2[k]; db += dz
__global__ __launch_bounds__(128) void refiner_tail_bwd_kernel(const float *__restrict__ d_r, const float *__restrict__ d_t, const float *__restrict__ f2,
                                                               const float *__restrict__ w_r, const float *__restrict__ w_t, const int64_t *__restrict__ obj,
                                                               int num_obj, float *__restrict__ df2, float *__restrict__ dw_r, float *__restrict__ db_r,
                                                               float *__restrict__ dw_t, float *__restrict__ db_t, int B) {
  const int k = threadIdx.x;
  for (int b = 0; b < B; ++b) {
    long ob = obj[b];
    ob = ob < 0 ? 0 : (ob >= num_obj ? num_obj - 1 : ob);
    float ar = 0.f, at = 0.f;
    const float xr = f2[(size_t)b * 256 + k], xt = f2[(size_t)b * 256 + 128 + k];
    for (int j = 0; j < 4; ++j) {
      const float gz = d_r[b * 4 + j];
      ar += gz * w_r[(ob * 4 + j) * 128 + k];
      dw_r[(ob * 4 + j) * 128 + k] += gz * xr;
      if (k == 0) db_r[ob * 4 + j] += gz;
    }
    for (int j = 0; j < 3; ++j) {
      const float gz = d_t[b * 3 + j];
      at += gz * w_t[(ob * 3 + j) * 128 + k];
      dw_t[(ob * 3 + j) * 128 + k] += gz * xt;
      if (k == 0) db_t[ob * 3 + j] += gz;
    }
    df2[(size_t)b * 256 + k] = ar;
    df2[(size_t)b * 256 + 128 + k] = at;
    __syncthreads();
  }
}

// wt[z][c][tap'][n] = w[z][n][tap][c], tap' = the tap mirrored through the kernel centre (what the data gradient convolves with)
__global__ __launch_bounds__(TB) void flip_kernel(const float *__restrict__ w, float *__restrict__ wt, int O, int T, int I, int KH, int KW, int Z) {
  const long per = (long)O * T * I;
  GRID_STRIDE(i, per * Z) {
    const long z = i / per, l = i - z * per;
    const int n = (int)(l % O);
    const long r = l / O;
    const int t = (int)(r % T);
    const long c = r / T;
    const int ky = t / KW, kx = t - ky * KW;
    const int tf = (KH - 1 - ky) * KW + (KW - 1 - kx);
    wt[i] = w[z * per + ((size_t)n * T + tf) * I + c];
  }
}

// every flip of a trainer in ONE launch: segment table in device memory (begin = first element of the segment in the launch's index space)
struct FlipSeg { long off, begin; int O, T, I, KH, KW, Z; };
__global__ __launch_bounds__(TB) void flip_all_kernel(const float *__restrict__ P, float *__restrict__ wf, const FlipSeg *__restrict__ segs, int nseg,
                                                      long total) {
  GRID_STRIDE(g, total) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {                               // last segment whose begin <= g
      const int mid = (lo + hi + 1) >> 1;
      if (segs[mid].begin <= g) lo = mid; else hi = mid - 1;
    }
    const FlipSeg sg = segs[lo];
    const long i = g - sg.begin, per = (long)sg.O * sg.T * sg.I;
    const long z = i / per, l = i - z * per;
    const int n = (int)(l % sg.O);
    const long r = l / sg.O;
    const int t = (int)(r % sg.T);
    const long c = r / sg.T;
    const int ky = t / sg.KW, kx = t - ky * sg.KW;
    const int tf = (sg.KH - 1 - ky) * sg.KW + (sg.KW - 1 - kx);
    wf[sg.off + i] = P[sg.off + z * per + ((size_t)n * sg.T + tf) * sg.I + c];
  }
}

// The same copies as 32 x 32 (output channel, input channel) tiles through LDS: reads run along c (the source's fastest axis), writes along n
// (the destination's) -- the element-wise form read with a stride of T * I floats (167 us per optimizer step for PoseNet's 86 MB; this one is
// bound by the copy).  Tile list: `tbegin` = first tile of the segment in the launch's tile space; a tile = (z, tap, n block, c block).
struct FlipTile { long off; int tbegin; int O, T, I, KH, KW, Z, nb_n, nb_c; };
__global__ __launch_bounds__(256) void flip_tiles_kernel(const float *__restrict__ P, float *__restrict__ wf, const FlipTile *__restrict__ segs, int nseg) {
  __shared__ float s_t[32][33];
  int lo = 0, hi = nseg - 1;
  while (lo < hi) {                               // last segment whose tbegin <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (segs[mid].tbegin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const FlipTile sg = segs[lo];
  int q = (int)blockIdx.x - sg.tbegin;
  const int cb = q % sg.nb_c; q /= sg.nb_c;
  const int nb = q % sg.nb_n; q /= sg.nb_n;
  const int t = q % sg.T, z = q / sg.T;
  const int ky = t / sg.KW, kx = t - ky * sg.KW;
  const int tf = (sg.KH - 1 - ky) * sg.KW + (sg.KW - 1 - kx);
  const long per = (long)sg.O * sg.T * sg.I;
  const float *src = P + sg.off + z * per;
  float *dst = wf + sg.off + z * per;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = nb * 32 + ty + r * 8, c = cb * 32 + tx;
    s_t[ty + r * 8][tx] = n < sg.O && c < sg.I ? src[((size_t)n * sg.T + tf) * sg.I + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = cb * 32 + ty + r * 8, n = nb * 32 + tx;
    if (n < sg.O && c < sg.I) dst[((size_t)c * sg.T + t) * sg.O + n] = s_t[tx][ty + r * 8];
  }
}

// layout conversion between the reference's state-dict tensors and the flat kernel layout
//   mode 0: OIHW [O][I][T] <-> O(T)Ipad          mode 1: OIHW (T = 9) <-> tap-major [9][O][I]          (dir 0: pack, 1: unpack)
__global__ __launch_bounds__(TB) void relayout_kernel(const float *__restrict__ src, float *__restrict__ dst, int O, int I, int T, int Ipad, int mode,
                                                      int dir) {
  GRID_STRIDE(i, (long)O * T * Ipad) {
    const int c = (int)(i % Ipad);
    const long r = i / Ipad;
    const int t = (int)(r % T);
    const long o = r / T;
    const size_t ref = ((size_t)o * I + c) * T + t;
    const size_t ker = mode == 0 ? (size_t)i : ((size_t)t * O + o) * I + c;
    if (dir == 0) dst[ker] = c < I ? src[ref] : 0.f;
    else if (c < I) dst[ref] = src[ker];
  }
}
__global__ __launch_bounds__(TB) void copy2d_kernel(const float *__restrict__ src, long s_ld, float *__restrict__ dst, long d_ld, long rows, long width) {
  GRID_STRIDE(i, rows * width) {
    const long r = i / width, c = i - r * width;
    dst[r * d_ld + c] = src[r * s_ld + c];
  }
}

// ------------------------------------------------------------------------------------------------
// trainer handle: parameter spec (reference keys / shapes) and where every tensor lives in the flat buffer
// ------------------------------------------------------------------------------------------------
struct PSpec {
  std::string key;
  int64_t shape[4] = {1, 1, 1, 1};
  int ndim = 0;
  // kernel-layout placement: up to two pieces (head layer 1's weight splits into the per-point and the global-feature block)
  int mode = 2;             // 0 OIHW->O(T)Ipad, 1 OIHW->tap-major, 2 plain copy, 3 head-1 weight split, 4 rows of a stacked tensor
  size_t off = 0, off2 = 0; // flat offsets (floats); off2: second piece of mode 3
  size_t kfloats = 0;       // floats this tensor occupies in the flat buffer
  int64_t numel() const { int64_t n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i]; return n; }
};

struct Trainer {
  int kind = 0, N = 0, K = 0, device = 0;
  std::map<std::vector<int>, size_t> ws_cache;      // (B, H, W, M) -> workspace bytes (the sizing pass walks the whole step)
  std::vector<PSpec> spec;
  std::map<std::string, int> index;
  std::map<std::string, size_t> slot;        // internal name -> flat offset
  size_t flat = 0;
  // flipped / transposed weights for the data gradients, rebuilt when the caller's parameter version changes
  float *wflip = nullptr;
  long flip_version = -1;
  const float *flip_src = nullptr;
  struct Flip { size_t off; int O, T, I, KH, KW, Z; };
  std::vector<Flip> flips;
  // Winograd F(4x4,3x3)-domain copies of the stride-1 3x3 trunk weights with >= 128 input channels (forward: G w G^T of the packed
  // weights; data gradient: of the flipped ones), [36][O][I] each, rebuilt with the flips
  struct Wino { size_t w_off, fwd, bwd; int O, I; };
  std::map<std::string, Wino> wino;
  float *wino_buf = nullptr;
  size_t wino_floats = 0;
  FlipSeg *flip_tab = nullptr;     // device copy of `flips` for the one-launch flip
  long flip_total = 0;
#ifdef DF_DEV
  std::vector<std::string> ev_desc;      // shape of every profiled launch (DF_PROFILE_VERBOSE)
  hipStream_t side = nullptr;            // DF_TRAIN_OVERLAP experiment
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
#endif
  FlipTile *flip_tiles = nullptr;  // the tiled form's segment table
  int flip_ntiles = 0;
  bool splitk = true;             // df_trainer_set_splitk
  // df_trainer_profile: HIP event pairs around every MFMA launch of a step, executed FLOPs per kind (0 fwd, 1 dgrad, 2 wgrad)
  bool profiling = false;
  std::vector<hipEvent_t> ev;
  size_t ev_used = 0;
  std::vector<int> ev_kind;
  std::vector<double> ev_flops;
};

size_t take(Trainer &t, const std::string &name, size_t floats) {
  const size_t o = t.flat;
  t.slot[name] = o;
  t.flat += (floats + 63) / 64 * 64;          // 256-byte aligned slots
  return o;
}

void add_spec(Trainer &t, const std::string &key, std::initializer_list<int64_t> shp, int mode, size_t off, size_t kfloats, size_t off2 = 0) {
  PSpec p;
  p.key = key;
  p.ndim = (int)shp.size();
  int i = 0;
  for (auto v : shp) p.shape[i++] = v;
  p.mode = mode; p.off = off; p.off2 = off2; p.kfloats = kfloats;
  t.index[key] = (int)t.spec.size();
  t.spec.push_back(p);
}

// as_gemm: the step uses the packed [O][(ky,kx,c)] rows as a plain GEMM operand (up_3 on chosen-pixel patches): its data gradient
// needs the plain transpose, not the tap-mirrored one
void add_conv(Trainer &t, const std::string &key, int O, int I, int k, bool tapmajor = false, bool as_gemm = false) {
  const int Ipad = (I + 3) / 4 * 4, T = k * k;
  const size_t fl = (size_t)O * T * Ipad;
  const size_t off = take(t, key, fl);
  add_spec(t, key, {O, I, k, k}, tapmajor ? 1 : 0, off, fl);
  if (tapmajor) t.flips.push_back({off, 9 * O, 1, I, 1, 1, 1});        // the low-resolution product is a 1x1 conv with 9*O outputs
  else if (as_gemm) t.flips.push_back({off, O, 1, T * Ipad, 1, 1, 1});
  else t.flips.push_back({off, O, T, Ipad, k, k, 1});
  if (k == 3 && !tapmajor && !as_gemm && I >= 128 && I % 4 == 0 && O % 4 == 0 && key.find("feats.layer") != std::string::npos) {
    const size_t n = (size_t)36 * O * I;
    t.wino[key] = Trainer::Wino{off, t.wino_floats, t.wino_floats + n, O, I};
    t.wino_floats += 2 * n;
  }
}
void add_plain(Trainer &t, const std::string &key, std::initializer_list<int64_t> shp) {
  size_t n = 1;
  for (auto v : shp) n *= (size_t)v;
  add_spec(t, key, shp, 2, take(t, key, n), n);
}
// Conv1d(k=1) / Linear weight [O][I] used as a GEMM operand (needs its transpose for the data gradient)
void add_gemm(Trainer &t, const std::string &key, int O, int I, bool conv1d) {
  const size_t off = take(t, key, (size_t)O * I);
  if (conv1d) add_spec(t, key, {O, I, 1}, 2, off, (size_t)O * I);
  else add_spec(t, key, {O, I}, 2, off, (size_t)O * I);
  t.flips.push_back({off, O, 1, I, 1, 1, 1});
}

const char *CNN = "cnn.model.module.";

void build_posenet(Trainer &t) {
  const std::string c = CNN;
  add_conv(t, c + "feats.conv1.weight", 64, 3, 7);
  int inpl = 64;
  const int planes_of[4] = {64, 128, 256, 512};
  for (int li = 1; li <= 4; ++li) {
    const int planes = planes_of[li - 1];
    for (int blk = 0; blk < 2; ++blk) {
      const int cin = blk == 0 ? inpl : planes;
      const std::string base = c + "feats.layer" + std::to_string(li) + "." + std::to_string(blk) + ".";
      // (a strided convolution's data gradient goes through per-tap products + a gather: it multiplies with the plain transpose)
      add_conv(t, base + "conv1.weight", planes, cin, 3, false, blk == 0 && li == 2);
      add_conv(t, base + "conv2.weight", planes, planes, 3);
      if (blk == 0 && cin != planes) add_conv(t, base + "downsample.0.weight", planes, cin, 1);
    }
    inpl = planes;
  }
  for (int s = 0; s < 4; ++s) add_conv(t, c + "psp.stages." + std::to_string(s) + ".1.weight", 512, 512, 1);
  add_conv(t, c + "psp.bottleneck.weight", 1024, 2560, 1);
  add_plain(t, c + "psp.bottleneck.bias", {1024});
  const char *ups[3] = {"up_1", "up_2", "up_3"};
  const int up_in[3] = {1024, 256, 64}, up_out[3] = {256, 64, 64};
  for (int u = 0; u < 3; ++u) {
    add_conv(t, c + ups[u] + ".conv.1.weight", up_out[u], up_in[u], 3, u < 2, u == 2);
    add_plain(t, c + ups[u] + ".conv.1.bias", {up_out[u]});
    add_plain(t, c + ups[u] + ".conv.2.weight", {1});
  }
  add_conv(t, c + "final.0.weight", 32, 64, 1);
  add_plain(t, c + "final.0.bias", {32});
  add_plain(t, c + "classifier.0.weight", {256, 256});     // dead weights (lib/pspnet.py:58-62): carried, never touched
  add_plain(t, c + "classifier.0.bias", {256});
  add_plain(t, c + "classifier.2.weight", {21, 256});
  add_plain(t, c + "classifier.2.bias", {21});
  add_plain(t, "feat.conv1.weight", {64, 3, 1});
  add_plain(t, "feat.conv1.bias", {64});
  const char *fn[5] = {"conv2", "e_conv1", "e_conv2", "conv5", "conv6"};
  const int fi[5] = {64, 32, 64, 256, 512}, fo[5] = {128, 64, 128, 512, 1024};
  // reference order of the keys: conv1, conv2, e_conv1, e_conv2, conv5, conv6 (weights then biases per layer)
  for (int i = 0; i < 5; ++i) {
    add_gemm(t, std::string("feat.") + fn[i] + ".weight", fo[i], fi[i], true);
    add_plain(t, std::string("feat.") + fn[i] + ".bias", {fo[i]});
  }
  // head layer 1: towers stacked r, t, c; per-point block [1920][384], global-feature block [1920][1024], bias [1920]
  const size_t wpt = take(t, "head1.wpt", (size_t)1920 * 384), wg = take(t, "head1.wg", (size_t)1920 * 1024), b1 = take(t, "head1.bias", 1920);
  t.flips.push_back({wpt, 1920, 1, 384, 1, 1, 1});
  t.flips.push_back({wg, 1920, 1, 1024, 1, 1, 1});
  const size_t w2 = take(t, "head2.w", (size_t)3 * 256 * 640), b2 = take(t, "head2.bias", 768);
  t.flips.push_back({w2, 256, 1, 640, 1, 1, 3});
  const size_t w3 = take(t, "head3.w", (size_t)3 * 128 * 256), b3 = take(t, "head3.bias", 384);
  t.flips.push_back({w3, 128, 1, 256, 1, 1, 3});
  const char *hs[3] = {"r", "t", "c"};
  const int hin[3] = {1408, 640, 256}, hout[3] = {640, 256, 128};
  for (int l = 0; l < 3; ++l)
    for (int h = 0; h < 3; ++h) {
      const std::string nm = "conv" + std::to_string(l + 1) + "_" + hs[h];
      if (l == 0) {
        add_spec(t, nm + ".weight", {640, 1408, 1}, 3, wpt + (size_t)h * 640 * 384, (size_t)640 * 1408, wg + (size_t)h * 640 * 1024);
        add_spec(t, nm + ".bias", {640}, 2, b1 + (size_t)h * 640, 640);
      } else {
        const size_t w = l == 1 ? w2 : w3, b = l == 1 ? b2 : b3;
        add_spec(t, nm + ".weight", {hout[l], hin[l], 1}, 2, w + (size_t)h * hout[l] * hin[l], (size_t)hout[l] * hin[l]);
        add_spec(t, nm + ".bias", {hout[l]}, 2, b + (size_t)h * hout[l], hout[l]);
      }
    }
  const int per[3] = {4, 3, 1};
  for (int h = 0; h < 3; ++h) {
    const std::string nm = std::string("conv4_") + hs[h];
    add_plain(t, nm + ".weight", {(int64_t)t.K * per[h], 128, 1});
    add_plain(t, nm + ".bias", {(int64_t)t.K * per[h]});
  }
}

void build_refiner(Trainer &t) {
  add_plain(t, "feat.conv1.weight", {64, 3, 1});
  add_plain(t, "feat.conv1.bias", {64});
  const char *fn[5] = {"conv2", "e_conv1", "e_conv2", "conv5", "conv6"};
  const int fi[5] = {64, 32, 64, 384, 512}, fo[5] = {128, 64, 128, 512, 1024};
  for (int i = 0; i < 5; ++i) {
    add_gemm(t, std::string("feat.") + fn[i] + ".weight", fo[i], fi[i], true);
    add_plain(t, std::string("feat.") + fn[i] + ".bias", {fo[i]});
  }
  const int li[2] = {1024, 512}, lo[2] = {512, 128};
  const char *hs[2] = {"r", "t"};
  for (int l = 0; l < 2; ++l)
    for (int h = 0; h < 2; ++h) {
      const std::string nm = "conv" + std::to_string(l + 1) + "_" + hs[h];
      add_gemm(t, nm + ".weight", lo[l], li[l], false);
      add_plain(t, nm + ".bias", {lo[l]});
    }
  const int per[2] = {4, 3};
  for (int h = 0; h < 2; ++h) {
    const std::string nm = std::string("conv3_") + hs[h];
    add_plain(t, nm + ".weight", {(int64_t)t.K * per[h], 128});
    add_plain(t, nm + ".bias", {(int64_t)t.K * per[h]});
  }
}

// ------------------------------------------------------------------------------------------------
// step plumbing: workspace arena, activation records, the backward tape
// ------------------------------------------------------------------------------------------------
struct View { float *d = nullptr; int ld = 0; };           // [rows][C] view: element (r, c) at d[r * ld + c] (d already offset to its channel)

// One resolution level of a pass: the crop-size buckets' [B_i][H_i][W_i] blocks concatenated along the pixel-row axis (the inference
// engine's scheme, engine.hip `Level`).  Launches whose arithmetic does not depend on the crop geometry (1x1 convolutions, the
// Winograd-domain products, every weight gradient, the whole per-point part) cover the rows of all buckets at once; direct k x k
// convolutions and the memory-bound glue run per bucket on row offsets into the same buffers.  Point rows are a single bucket.
struct Lv {
  std::vector<int> B, H, W;
  std::vector<long> off;      // first pixel row of bucket i
  std::vector<int> b0;        // first frame of bucket i
  long rows = 0;
  int frames = 0;
  int nb() const { return (int)B.size(); }
  void push(int b, int h, int w) {
    B.push_back(b); H.push_back(h); W.push_back(w); off.push_back(rows); b0.push_back(frames);
    rows += (long)b * h * w; frames += b;
  }
};

struct Act {
  View v, g;                    // values; gradient (allocated / aliased during the backward pass)
  const Lv *lv = nullptr;
  int C = 0;
  bool gset = false;
  long rows() const { return lv->rows; }
};

enum GemmKind { GK_FWD = 0, GK_DGRAD = 1, GK_WGRAD = 2 };

struct Step {
  Trainer *t;
  hipStream_t st;
  bool dry;
  char *base;
  size_t off = 0, cap = 0, peak = 0;
  int err = DF_OK;
  const float *P = nullptr;      // flat parameters
  float *G = nullptr;            // flat gradients (accumulated)
  float *splitk = nullptr;
  size_t splitk_bytes = 0;
  std::deque<Act> acts;
  std::deque<Lv> lvs;
  std::vector<std::function<void()>> tape;
#ifdef DF_DEV
  // development experiment (DF_TRAIN_OVERLAP=1): a layer's weight gradient on a side stream beside its data gradient; joined before the layer's
  // backward closure ends, so no buffer hazard crosses a layer (profiles/r04_experiments/README.md)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool keep_wgrad_ws = false;
#endif

  void *bytes(size_t b) {
    b = (b + 255) & ~size_t(255);
    // the sizing pass hands out (never dereferenced) non-null addresses too, so that every "is there a buffer yet" decision of the
    // backward pass comes out as in the real run: both passes allocate the same sequence by construction
    void *p = (dry ? reinterpret_cast<char *>(4096) : base) + off;
    off += b;
    if (off > peak) peak = off;
    if (!dry && off > cap && err == DF_OK) err = set_error(DF_ERR_WORKSPACE, "train step: workspace too small (need > %zu bytes, have %zu)", off, cap);
    return p;
  }
  float *f(size_t n) { return static_cast<float *>(bytes(n * sizeof(float))); }
  bool live() const { return !dry && err == DF_OK; }
#ifdef DF_DEV
  // dev build, DF_TRAIN_DEBUG=1: synchronise after every phase and name it on stderr (localises a faulting launch)
  void dbg(const char *what, const std::string &extra = std::string()) {
    static const bool on = df::dev_getenv("DF_TRAIN_DEBUG") != nullptr;
    if (!on || dry) return;
    const hipError_t e = hipStreamSynchronize(st);
    fprintf(stderr, "[df-train] %s %s: %s\n", what, extra.c_str(), e == hipSuccess ? "ok" : hipGetErrorString(e));
    fflush(stderr);
  }
#else
  void dbg(const char *, const std::string & = std::string()) {}
#endif
  void fail(int rc) { if (rc != DF_OK && err == DF_OK) err = rc; }
  // every MFMA launch of the step goes through here: with df_trainer_profile on, HIP events on the launch stream bracket it and its
  // EXECUTED FLOPs are tallied per kind (forward / data gradient / weight gradient)
  void prof_begin() {
    if (!t->profiling || !live()) return;
    if (t->ev_used + 2 > t->ev.size()) {
      const size_t old = t->ev.size();
      t->ev.resize(old + 512);
      for (size_t i = old; i < t->ev.size(); ++i) hipEventCreate(&t->ev[i]);
    }
    hipEventRecord(t->ev[t->ev_used], st);
  }
  void prof_end(int kind, double flops, const ConvParams *p = nullptr, long M = 0) {
    if (!t->profiling || !live()) return;
    hipEventRecord(t->ev[t->ev_used + 1], st);
    t->ev_kind.push_back(kind);
    t->ev_flops.push_back(flops);
#ifdef DF_DEV
    char d[160] = "";
    if (p) snprintf(d, sizeof(d), "M=%ld N=%d K=%d k%dx%d s%d d%d z%d", M > 0 ? M : (long)p->B * p->OH * p->OW, p->Cout, p->KH * p->KW * p->Cin, p->KH, p->KW, p->stride, p->dil, p->zcount);
    t->ev_desc.push_back(d);
#endif
    t->ev_used += 2;
  }
  void gemm(int kind, const ConvParams &p) {
    if (!live()) return;
    prof_begin();
    fail(launch_conv(p, st));
    prof_end(kind, conv_flops(p), &p);
  }
  // the same convolution over several buckets: one launch (launch_conv_multi)
  void gemm_multi(int kind, const ConvParams &p, const std::vector<WgradSeg> &segs) {
    if (!live() || segs.empty()) return;
    double fl = 0;
    for (const WgradSeg &g : segs) fl += 2.0 * g.B * g.OH * g.OW * (double)p.Cout * p.KH * p.KW * p.Cin;
    prof_begin();
    fail(launch_conv_multi(p, (int)segs.size(), segs.data(), st));
    long M = 0;
    for (const WgradSeg &g : segs) M += (long)g.B * g.OH * g.OW;
    prof_end(kind, fl, &p, M);
  }
  size_t slot(const std::string &name) {
    auto it = t->slot.find(name);
    if (it == t->slot.end()) {
      if (err == DF_OK) err = set_error(DF_ERR_STATE, "trainer: no parameter slot named '%s'", name.c_str());
      return 0;
    }
    return it->second;
  }
  const float *p(const std::string &name, size_t extra = 0) { const size_t o = slot(name); return dry ? nullptr : P + o + extra; }
  float *gr(const std::string &name, size_t extra = 0) { const size_t o = slot(name); return dry ? nullptr : G + o + extra; }
  const float *pf(const std::string &name, size_t extra = 0) { const size_t o = slot(name); return dry ? nullptr : t->wflip + o + extra; }
  const Lv *level(const Lv &l) { lvs.push_back(l); return &lvs.back(); }
  const Lv *flat_level(long rows) { Lv l; l.push((int)rows, 1, 1); return level(l); }
  Act *act(const Lv *lv, int C, float *d = nullptr, int ld = 0) {
    acts.emplace_back();
    Act *a = &acts.back();
    a->lv = lv; a->C = C;
    a->v.d = d ? d : f((size_t)lv->rows * C);
    a->v.ld = d ? ld : C;
    return a;
  }
  Act *act(long rows, int C, float *d = nullptr, int ld = 0) { return act(flat_level(rows), C, d, ld); }
  // gradient storage of `a` for a producer that is about to write (returns true when it has to ACCUMULATE)
  bool grad_of(Act *a) {
    if (!a->g.d) { a->g.d = f((size_t)a->rows() * a->C); a->g.ld = a->C; }
    const bool acc = a->gset;
    a->gset = true;
    return acc;
  }
};

// a plain GEMM over all rows of x: every pixel / point row is one output row (1x1 convolution, stride 1)
ConvParams flat_params(const Act *x, int cin, const float *w, const float *bias, Act *y, int act) {
  ConvParams p;
  p.in = x->v.d; p.wgt = w; p.bias = bias; p.out = y->v.d;
  p.B = (int)x->rows(); p.H = p.W = p.OH = p.OW = 1; p.Cin = cin; p.in_ld = x->v.ld;
  p.Cout = y->C; p.out_ld = y->v.ld;
  p.act = act;
  return p;
}
// bucket tables of a level, TAB_MAX buckets each (aux: a second level whose first rows go into aux0)
std::vector<BTab> make_tabs(const Lv *lv, const Lv *aux = nullptr) {
  std::vector<BTab> out;
  for (int g0 = 0; g0 < lv->nb(); g0 += TAB_MAX) {
    BTab t{};
    t.n = std::min(TAB_MAX, lv->nb() - g0);
    for (int i = 0; i < TAB_MAX; ++i) {
      const int g = g0 + std::min(i, t.n - 1);           // (entries past n repeat the last bucket: never selected)
      t.B[i] = lv->B[g]; t.H[i] = lv->H[g]; t.W[i] = lv->W[g]; t.b0[i] = lv->b0[g];
      t.row0[i] = lv->off[g]; t.row1[i] = lv->off[g] + (long)lv->B[g] * lv->H[g] * lv->W[g];
      t.aux0[i] = aux ? aux->off[g] : 0;
    }
    out.push_back(t);
  }
  return out;
}
inline long tab_rows(const BTab &t) { return t.row1[t.n - 1] - t.row0[0]; }
inline int tab_frames(const BTab &t) { return t.b0[t.n - 1] + t.B[t.n - 1] - t.b0[0]; }

// bucket i of a k x k convolution between two levels
ConvParams bucket_params(const Act *x, int i, int cin, const float *w, const float *bias, Act *y, int k, int stride, int pad, int dil, int act) {
  ConvParams p;
  const Lv *li = x->lv, *lo = y->lv;
  p.in = x->v.d + li->off[i] * x->v.ld; p.wgt = w; p.bias = bias; p.out = y->v.d + lo->off[i] * y->v.ld;
  p.B = li->B[i]; p.H = li->H[i]; p.W = li->W[i]; p.Cin = cin; p.in_ld = x->v.ld;
  p.OH = lo->H[i]; p.OW = lo->W[i]; p.Cout = y->C; p.out_ld = y->v.ld;
  p.KH = p.KW = k; p.stride = stride; p.pad = pad; p.dil = dil; p.act = act;
  return p;
}

void launch_act_bwd(Step &s, Act *y, int act, const float *slope, float *dslope) {
  const long rows = y->rows();
  const int C4 = y->C / 4;
  const unsigned blocks = act == 2 ? nblk(rows * C4, 512) : nblk(rows * C4);
  float *part = act == 2 ? s.f(blocks) : nullptr;
  if (!s.live()) return;
  hipLaunchKernelGGL(act_bwd2d_kernel, dim3(blocks), dim3(TB), 0, s.st, y->g.d, y->g.ld, y->v.d, y->v.ld, rows, C4, act, slope, part);
  if (act == 2) hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(TB), 0, s.st, part, (int)blocks, 1L, dslope, 1);
}

// weight / bias gradient of a forward GEMM / convolution (`f`: channels, strides, kernel geometry; `segs`: the buckets) with upstream
// gradient view gy, accumulated into dw / db: ONE contraction over the pixels of all buckets
void wgrad(Step &s, ConvParams f, const std::vector<WgradSeg> &segs, View gy, float *dw, float *db) {
  f.out = gy.d; f.out_ld = gy.ld; f.out_coff = 0;
  f.bias = nullptr; f.res = nullptr; f.act = ACT_NONE; f.zcount = 1;
  f.rows_per_group = f.rows_valid = f.bias_group_ld = 0;
  const size_t mark = s.off;
  const size_t need = wgrad_multi_workspace_bytes(f, (int)segs.size(), segs.data());
  void *ws = s.bytes(need);
  hipStream_t st = s.st;
  bool keep = false;
#ifdef DF_DEV
  keep = s.keep_wgrad_ws;
  if (keep && s.live() && s.side) {      // fork: the side stream starts where the main stream is now
    hipEventRecord(s.ev_fork, s.st);
    hipStreamWaitEvent(s.side, s.ev_fork, 0);
    st = s.side;
  }
#endif
  if (s.live()) {
    double M = 0;
    for (const WgradSeg &g : segs) M += (double)g.B * g.OH * g.OW;
    s.prof_begin();
    s.fail(launch_wgrad_multi(f, (int)segs.size(), segs.data(), dw, db, ws, need, st, 1));
    s.prof_end(GK_WGRAD, 2.0 * M * f.Cout * f.KH * f.KW * f.Cin, &f, (long)M);
  }
#ifdef DF_DEV
  if (keep && s.live() && s.side) {
    hipEventRecord(s.ev_join, s.side);
    static const bool sync_dbg = df::dev_getenv("DF_TRAIN_OVERLAP_SYNC") != nullptr;      // A/B: the side stream drained at once (no concurrency left)
    if (sync_dbg) hipStreamSynchronize(s.side);
  }
#endif
  s.dbg("wgrad");
  if (!keep) s.off = mark;          // (overlap experiment: the partial slices stay allocated until the layer's closure joins the side stream)
}
// a launch that is one bucket by itself (plain GEMMs over rows; `f` carries B / H / W / OH / OW)
void wgrad(Step &s, const ConvParams &f, View gy, float *dw, float *db) {
  wgrad(s, f, std::vector<WgradSeg>{WgradSeg{f.B, f.H, f.W, f.OH, f.OW, 0, 0}}, gy, dw, db);
}

// data gradient of forward conv `f` (cached flipped weights): dx (+)= conv^T(gy)
void dgrad(Step &s, const ConvParams &f, View gy, View dx, const float *wflip, bool accumulate) {
  ConvParams q;
  q.in = gy.d; q.B = f.B; q.H = f.OH; q.W = f.OW; q.Cin = f.Cout; q.in_ld = gy.ld;
  q.wgt = wflip;
  q.out = dx.d; q.OH = f.H; q.OW = f.W; q.Cout = f.Cin; q.out_ld = dx.ld;
  q.KH = f.KH; q.KW = f.KW; q.stride = 1; q.up = f.stride; q.dil = f.dil; q.pad = f.dil * (f.KH - 1) - f.pad;
  if (accumulate) { q.res = dx.d; q.res_ld = dx.ld; }
  if (f.zcount > 1) {      // (a split-K launch walks blockIdx.z too: the z strides must stay zero for everything else)
    q.zcount = f.zcount; q.z_in_coff = f.z_out_coff; q.z_out_coff = f.z_in_coff; q.z_wgt = (long)f.Cin * f.Cout * f.KH * f.KW;
  }
  q.splitk_ws = s.splitk; q.splitk_ws_bytes = s.splitk_bytes;
  s.gemm(GK_DGRAD, q);
  s.dbg("dgrad");
}

struct ConvW {
  std::string name;            // slot of the weight (its flipped copy and gradient share the offset)
  size_t woff = 0;             // extra offset inside the slot
  std::string bias;            // slot of the bias ("" = none)
  size_t boff = 0;
  std::string slope;           // PReLU slope slot
};

View rows_view(View v, long row0) { return View{v.d + row0 * v.ld, v.ld}; }

// y = act(conv(x) + bias + res); registers its backward.  The first `cin` channels of x's view are consumed.  A 1x1 stride-1
// convolution is ONE GEMM over the rows of all buckets (forward, data and weight gradient); a k x k or strided one runs the direct
// kernel per bucket -- or, for the stride-1 3x3 trunk layers whose map the engine's rule sends through F(4x4,3x3), one transform-domain
// GEMM over the tiles of all such buckets, forward and data gradient alike -- and its weight gradient is one contraction over all
// buckets' pixels (launch_wgrad_multi).
Act *conv(Step &s, Act *x, int cin, const ConvW &cw, int cout, int k, int stride, int pad, int dil, int act, Act *res = nullptr, Act *into = nullptr,
          bool need_dx = true) {
  const Lv *li = x->lv;
  const int nb = li->nb();
  const bool flat = k == 1 && stride == 1 && pad == 0;
  const Lv *lo = li;
  if (!flat) {
    Lv o;
    for (int i = 0; i < nb; ++i) o.push(li->B[i], conv_out(li->H[i], k, stride, pad, dil), conv_out(li->W[i], k, stride, pad, dil));
    lo = s.level(o);
  }
  Act *y = into ? into : s.act(lo, cout);
  if (!flat && into) y->lv = lo;
  const float *wp = s.p(cw.name, cw.woff), *bp = cw.bias.empty() ? nullptr : s.p(cw.bias, cw.boff);
  const float *slope = act == ACT_PRELU ? s.p(cw.slope) : nullptr;
  // the launches of the forward pass, kept for the backward closure
  auto plan = std::make_shared<std::vector<ConvParams>>();
  auto f4 = std::make_shared<std::vector<int>>();          // buckets on the F(4x4,3x3) route
  const auto wit = s.t->wino.find(cw.name);
  const bool wino_ok = !flat && wit != s.t->wino.end() && k == 3 && stride == 1 && pad == dil && cw.bias.empty() && act != ACT_PRELU;
  if (flat) {
    ConvParams p = flat_params(x, cin, wp, bp, y, act);
    if (res) { p.res = res->v.d; p.res_ld = res->v.ld; }
    p.prelu = slope;
    p.splitk_ws = s.splitk; p.splitk_ws_bytes = s.splitk_bytes;
    plan->push_back(p);
    s.gemm(GK_FWD, p);
  } else {
    std::vector<WgradSeg> direct;
    int first_direct = -1;
    for (int i = 0; i < nb; ++i) {
      ConvParams p = bucket_params(x, i, cin, wp, bp, y, k, stride, pad, dil, act);
      if (res) { p.res = res->v.d + lo->off[i] * res->v.ld; p.res_ld = res->v.ld; }
      p.prelu = slope;
      p.splitk_ws = s.splitk; p.splitk_ws_bytes = s.splitk_bytes;
      plan->push_back(p);
      if (wino_ok && wino_route(li->H[i], li->W[i], dil, cin, cout) == 4) { f4->push_back(i); continue; }
      if (first_direct < 0) first_direct = i;
      direct.push_back(WgradSeg{li->B[i], li->H[i], li->W[i], lo->H[i], lo->W[i], li->off[i], lo->off[i]});
    }
    if (direct.size() == 1) s.gemm(GK_FWD, (*plan)[first_direct]);
    else if (!direct.empty()) {      // the direct kernel over all of them in one launch (a workgroup's tile lies inside one bucket)
      ConvParams p = (*plan)[0];
      p.in = x->v.d; p.out = y->v.d;
      if (res) p.res = res->v.d;
      s.gemm_multi(GK_FWD, p, direct);
    }
  }
  // transform-domain pass over the buckets of `f4` (stride 1: input and output levels have the same rows)
  auto wino_pass = [=](Step &s, View in, int ci, const float *U, View out, int co, const float *rs, int rs_ld, int a, int kind) {
    if (f4->empty()) return;
    std::vector<int> tB, tH, tW;
    std::vector<long> trow, t0;
    long T = 0;
    for (int i : *f4) {
      tB.push_back(li->B[i]); tH.push_back(li->H[i]); tW.push_back(li->W[i]); trow.push_back(li->off[i]);
      t0.push_back(T);
      T += wino_geom(li->B[i], li->H[i], li->W[i], dil, 4).T;
    }
    const size_t mark = s.off;
    float *V = s.f((size_t)36 * T * ci), *M = s.f((size_t)36 * T * co);
    if (s.live()) {
      const int nw = (int)f4->size();
      launch_wino4_input_multi(in.d, in.ld, V, nw, tB.data(), tH.data(), tW.data(), trow.data(), t0.data(), ci, dil, T, s.st);
      ConvParams q;
      q.in = V; q.wgt = U; q.out = M;
      q.B = (int)T; q.Cin = ci; q.in_ld = ci; q.Cout = co; q.out_ld = co;
      q.zcount = 36; q.z_in_coff = T * ci; q.z_wgt = (long)co * ci; q.z_out_coff = T * co;
      s.gemm(kind, q);
      launch_wino4_output_multi(M, out.d, out.ld, rs, rs_ld, a, nw, tB.data(), tH.data(), tW.data(), trow.data(), t0.data(), co, dil, T, s.st);
    }
    s.off = mark;
  };
  wino_pass(s, x->v, cin, s.dry || !wino_ok ? nullptr : s.t->wino_buf + wit->second.fwd, y->v, cout, res ? res->v.d : nullptr, res ? res->v.ld : 0, act, GK_FWD);
  s.dbg("conv fwd", cw.name);
  Step *sp = &s;
  s.tape.push_back([=]() {
    Step &s = *sp;
    s.dbg("conv bwd begin", cw.name);
    if (act != ACT_NONE) launch_act_bwd(s, y, act, slope, act == ACT_PRELU ? s.gr(cw.slope) : nullptr);
#ifdef DF_DEV
    static const bool overlap = df::dev_getenv("DF_TRAIN_OVERLAP") != nullptr;
    s.keep_wgrad_ws = overlap && need_dx;
    bool acc_early = false;
    if (s.keep_wgrad_ws) acc_early = s.grad_of(x);      // x's gradient buffer outlives the layer: it is taken BELOW the slices that are released at the join
    const size_t mark_layer = s.off;
#endif
    {   // weight gradient: one contraction over every bucket's pixels
      std::vector<WgradSeg> segs;
      if (flat) segs.push_back(WgradSeg{(int)x->rows(), 1, 1, 1, 1, 0, 0});
      else for (int i = 0; i < nb; ++i) segs.push_back(WgradSeg{li->B[i], li->H[i], li->W[i], lo->H[i], lo->W[i], li->off[i], lo->off[i]});
      ConvParams f = (*plan)[0];
      f.in = x->v.d;
      wgrad(s, f, segs, y->g, s.gr(cw.name, cw.woff), cw.bias.empty() ? nullptr : s.gr(cw.bias, cw.boff));
    }
    if (need_dx) {
#ifdef DF_DEV
      const bool acc = s.keep_wgrad_ws ? acc_early : s.grad_of(x);
#else
      const bool acc = s.grad_of(x);
#endif
      if (flat) dgrad(s, (*plan)[0], y->g, x->g, s.pf(cw.name, cw.woff), acc);
      else {
        std::vector<int> direct;
        for (int i = 0; i < nb; ++i) {
          bool on_f4 = false;
          for (int j : *f4) on_f4 |= j == i;
          if (!on_f4) direct.push_back(i);
        }
        if (stride != 1) {
          // strided: dcol[m][tap * cin + c] = sum_n dY[m][n] w[n][tap][c] for every OUTPUT pixel m -- one GEMM over the rows of all buckets
          // against the plain transpose of the packed weights -- then every input pixel gathers the (tap, output pixel) pairs that read it
          // (col2im_multi_kernel).  (The dilated-input form of the direct kernel multiplies 3/4 zeros at stride 2 and runs per bucket.)
          const int kk = k * k * cin;
          const size_t mark = s.off;
          float *dcol = s.f((size_t)lo->rows * kk);
          ConvParams q;
          q.in = y->g.d; q.B = (int)lo->rows; q.Cin = cout; q.in_ld = y->g.ld;
          q.wgt = s.pf(cw.name, cw.woff);
          q.out = dcol; q.Cout = kk; q.out_ld = kk;
          q.splitk_ws = s.splitk; q.splitk_ws_bytes = s.splitk_bytes;
          s.gemm(GK_DGRAD, q);
          if (s.live())
            for (const BTab &t : make_tabs(li, lo))
              hipLaunchKernelGGL(col2im_multi_kernel, dim3(nblk(tab_rows(t) * (cin / 4))), dim3(TB), 0, s.st, dcol, x->g.d, x->g.ld, cin, k, stride, pad, dil,
                                 acc ? 1 : 0, t);
          s.off = mark;
        } else if (direct.size() == 1)
          for (int i : direct) dgrad(s, (*plan)[i], rows_view(y->g, lo->off[i]), rows_view(x->g, li->off[i]), s.pf(cw.name, cw.woff), acc);
        else if (!direct.empty()) {
          const ConvParams &f = (*plan)[0];
          ConvParams q;
          q.in = y->g.d; q.Cin = f.Cout; q.in_ld = y->g.ld;
          q.wgt = s.pf(cw.name, cw.woff);
          q.out = x->g.d; q.Cout = f.Cin; q.out_ld = x->g.ld;
          q.KH = f.KH; q.KW = f.KW; q.stride = 1; q.dil = f.dil; q.pad = f.dil * (f.KH - 1) - f.pad;
          if (acc) { q.res = x->g.d; q.res_ld = x->g.ld; }
          std::vector<WgradSeg> segs;
          for (int i : direct) segs.push_back(WgradSeg{lo->B[i], lo->H[i], lo->W[i], li->H[i], li->W[i], lo->off[i], li->off[i]});
          s.gemm_multi(GK_DGRAD, q, segs);
        }
        wino_pass(s, y->g, cout, s.dry || !wino_ok ? nullptr : s.t->wino_buf + wit->second.bwd, x->g, cin, acc ? x->g.d : nullptr, x->g.ld, ACT_NONE, GK_DGRAD);
      }
    }
#ifdef DF_DEV
    if (s.keep_wgrad_ws) {      // join: nothing after this layer may touch dY / the slices before its weight gradient is done
      if (s.live() && s.side) hipStreamWaitEvent(s.st, s.ev_join, 0);
      s.off = mark_layer;
      s.keep_wgrad_ws = false;
    }
#endif
    if (res) {
      if (!res->gset) { res->g = y->g; res->gset = true; }          // the residual's gradient IS this (masked) gradient: alias, no copy
      else if (s.live()) hipLaunchKernelGGL(add2d_kernel, dim3(nblk(y->rows() * (y->C / 4))), dim3(TB), 0, s.st, res->g.d, res->g.ld, y->g.d, y->g.ld,
                                            y->rows(), y->C / 4);
    }
  });
  return y;
}

// channel view [c0, c0 + C) of a wider activation record (shares storage; its gradient view is resolved lazily by the caller)
Act *slice(Step &s, Act *a, int c0, int C) {
  s.acts.emplace_back();
  Act *v = &s.acts.back();
  *v = *a;
  v->v.d = a->v.d + c0;
  v->C = C;
  v->g = View{};
  v->gset = false;
  return v;
}

int check_flips(Trainer &t, const float *P, long version, hipStream_t st) {
  if (!t.wflip || (t.wino_floats && !t.wino_buf))
    return set_error(DF_ERR_STATE, "trainer: created without a device (no arena for the data gradients' weight copies)");
  if (t.flip_version == version && t.flip_src == P && version >= 0) return DF_OK;
  if (t.flip_tiles)
    hipLaunchKernelGGL(flip_tiles_kernel, dim3(t.flip_ntiles), dim3(256), 0, st, P, t.wflip, t.flip_tiles, (int)t.flips.size());
  else if (t.flip_tab)
    hipLaunchKernelGGL(flip_all_kernel, dim3(nblk(t.flip_total, 8192)), dim3(TB), 0, st, P, t.wflip, t.flip_tab, (int)t.flips.size(), t.flip_total);
  else
    for (const Trainer::Flip &f : t.flips)
      hipLaunchKernelGGL(flip_kernel, dim3(nblk((long)f.O * f.T * f.I * f.Z, 1024)), dim3(TB), 0, st, P + f.off, t.wflip + f.off, f.O, f.T, f.I, f.KH, f.KW, f.Z);
  {   // the F(4x4,3x3)-domain copies, forward (of the packed weights) and data gradient (of the flipped ones: [I][9][O]): one launch per 32
    WinoWTab tab;
    tab.n = 0; tab.e0[0] = 0;
    auto flush = [&]() { launch_wino4_weight_multi(P, t.wflip, t.wino_buf, tab, st); tab.n = 0; tab.e0[0] = 0; };
    auto push = [&](int O, int C, int from_b, long src, long dst) {
      if (tab.n == WINO_WMAX) flush();
      const int g = tab.n++;
      tab.O[g] = O; tab.C[g] = C; tab.from_b[g] = from_b; tab.src_off[g] = src; tab.dst_off[g] = dst;
      tab.e0[g + 1] = tab.e0[g] + (long)O * C;
    };
    for (const auto &kv : t.wino) {
      const Trainer::Wino &w = kv.second;
      push(w.O, w.I, 0, (long)w.w_off, (long)w.fwd);
      push(w.I, w.O, 1, (long)w.w_off, (long)w.bwd);
    }
    flush();
  }
  t.flip_version = version;
  t.flip_src = P;
  return check_launch("trainer: weight flips");
}

// ------------------------------------------------------------------------------------------------
// PoseNet step
// ------------------------------------------------------------------------------------------------
struct PoseNetIO {
  int nb;                        // crop-size buckets of the pass; frames are concatenated in bucket order everywhere below
  const int *B, *H, *W;          // host [nb]
  const float *const *img;       // host [nb]: device pointers [B_i][3][H_i][W_i]
  int M;
  const float *cloud, *target, *model_points;
  const int64_t *choose, *obj;
  const int *symmetric;     // host [sum B]
  float w;
  int dropout;
  unsigned seed;
  float *loss, *dis, *new_points, *new_target;        // [B], [B], [B][N][3], [B][M][3]
  float *out_r, *out_t, *out_c, *emb;                 // optional copies of the predictions ([B][N][4] ...); emb [B][32][N]
};

Act *basic_block(Step &s, Act *x, int cin, const std::string &base, int planes, int stride, int dil, bool has_ds, Act *out_into = nullptr) {
  Act *t = conv(s, x, cin, ConvW{base + "conv1.weight"}, planes, 3, stride, dil, dil, ACT_RELU);
  Act *res = x;
  if (has_ds) res = conv(s, x, cin, ConvW{base + "downsample.0.weight"}, planes, 1, stride, 0, 1, ACT_NONE);
  return conv(s, t, planes, ConvW{base + "conv2.weight"}, planes, 3, 1, dil, dil, ACT_RELU, res, out_into);
}

// Dropout2d (lib/pspnet.py:46,52): one keep / drop decision per (frame, channel); out of place -- the PReLU gradient upstream needs
// the un-scaled activation
Act *dropout2d(Step &s, Act *a, float p, unsigned seed) {
  const Lv *lv = a->lv;
  float *scale = s.f((size_t)lv->frames * a->C);
  Act *o = s.act(lv, a->C);
  const std::vector<BTab> tabs = make_tabs(lv);
  if (s.live()) {
    s.fail(df_dropout2d_mask(scale, (int64_t)lv->frames * a->C, seed, p, s.st));
    for (const BTab &t : tabs)
      hipLaunchKernelGGL(channel_scale_multi_kernel, dim3(nblk(tab_rows(t) * (a->C / 4))), dim3(TB), 0, s.st, a->v.d, a->v.ld, scale, o->v.d, o->v.ld, a->C / 4, t);
  }
  Step *sp = &s;
  s.tape.push_back([=]() {
    Step &s = *sp;
    s.grad_of(a);
    if (s.live())
      for (const BTab &t : tabs)
        hipLaunchKernelGGL(channel_scale_multi_kernel, dim3(nblk(tab_rows(t) * (a->C / 4))), dim3(TB), 0, s.st, o->g.d, o->g.ld, scale, a->g.d, a->g.ld, a->C / 4, t);
  });
  return o;
}

// PSPUpsample through the low-resolution per-tap products (layers.hip upconv_gather): x [B][h][w][Cin] -> [B][2h][2w][Cout] per bucket;
// the product, its weight and data gradients, the activation / bias adjoints are single launches over the rows of all buckets
Act *upconv(Step &s, Act *x, const std::string &base, int cin, int cout) {
  const Lv *li = x->lv;
  const int nb = li->nb();
  Act *y = s.act(li, 9 * cout);
  const ConvW cw{base + "conv.1.weight"};
  ConvParams p = flat_params(x, cin, s.p(cw.name), nullptr, y, ACT_NONE);
  p.splitk_ws = s.splitk; p.splitk_ws_bytes = s.splitk_bytes;
  s.gemm(GK_FWD, p);
  Lv up;
  for (int i = 0; i < nb; ++i) up.push(li->B[i], 2 * li->H[i], 2 * li->W[i]);
  const Lv *lo = s.level(up);
  Act *o = s.act(lo, cout);
  if (s.live())
    for (int i = 0; i < nb; ++i)
      s.fail(launch_upconv_gather(y->v.d + li->off[i] * 9 * cout, s.p(base + "conv.1.bias"), s.p(base + "conv.2.weight"), o->v.d + lo->off[i] * cout, li->B[i],
                                  li->H[i], li->W[i], cout, s.st));
  Step *sp = &s;
  s.tape.push_back([=]() {
    Step &s = *sp;
    launch_act_bwd(s, o, ACT_PRELU, s.p(base + "conv.2.weight"), s.gr(base + "conv.2.weight"));
    {   // bias gradient: column sums of the pre-activation gradient over all pixels
      const long rows = o->rows();
      const int nbk = (int)((rows + 255) / 256);
      float *part = s.f((size_t)nbk * cout);
      if (s.live()) {
        hipLaunchKernelGGL(colsum_obj_kernel, dim3((cout + 31) / 32, nbk), dim3(256), 0, s.st, o->g.d, o->g.ld, part, 256, cout, rows);
        hipLaunchKernelGGL(colsum_obj_kernel, dim3((cout + 31) / 32, 1), dim3(256), 0, s.st, part, cout, s.gr(base + "conv.1.bias"), nbk, cout, (long)nbk, 1);
      }
    }
    s.grad_of(y);
    if (s.live())
      for (const BTab &t : make_tabs(li))
        hipLaunchKernelGGL(upconv_gather_bwd_multi_kernel, dim3(nblk(tab_rows(t) * 9 * (cout / 4))), dim3(TB), 0, s.st, o->g.d, y->g.d, cout, t);
    wgrad(s, p, y->g, s.gr(cw.name), nullptr);
    const bool acc = s.grad_of(x);
    dgrad(s, p, y->g, x->g, s.pf(cw.name), acc);
  });
  return o;
}

void posenet_step(Step &s, const PoseNetIO &io) {
  Trainer &t = *s.t;
  const std::string C = CNN;
  int B = 0;
  for (int i = 0; i < io.nb; ++i) B += io.B[i];
  const int nb = io.nb, N = t.N, Npad = round_up(N, 128), rows = B * Npad;
  s.splitk_bytes = (size_t)32 << 20;
  s.splitk = static_cast<float *>(s.bytes(s.splitk_bytes));
  if (!t.splitk) { s.splitk = nullptr; s.splitk_bytes = 0; }      // (allocated either way: the workspace size does not depend on the switch)
  Step *sp = &s;

  // ---- colour branch (lib/extractors.py:114-124, lib/pspnet.py:64-77) ----
  Lv limg;
  for (int i = 0; i < nb; ++i) limg.push(io.B[i], io.H[i], io.W[i]);
  Act *img4 = s.act(s.level(limg), 4);
  if (s.live())
    for (int i = 0; i < nb; ++i) launch_nchw3_to_nhwc4(io.img[i], img4->v.d + img4->lv->off[i] * 4, io.B[i], io.H[i], io.W[i], s.st);
  Act *stem = conv(s, img4, 4, ConvW{C + "feats.conv1.weight"}, 64, 7, 2, 3, 1, ACT_RELU, nullptr, nullptr, false);
  Lv lpool;
  for (int i = 0; i < nb; ++i) lpool.push(io.B[i], conv_out(stem->lv->H[i], 3, 2, 1, 1), conv_out(stem->lv->W[i], 3, 2, 1, 1));
  Act *x = s.act(s.level(lpool), 64);
  {
    const Lv *ls = stem->lv, *lx = x->lv;
    if (s.live())
      for (int i = 0; i < nb; ++i)
        launch_maxpool3s2(stem->v.d + ls->off[i] * 64, x->v.d + lx->off[i] * 64, ls->B[i], ls->H[i], ls->W[i], 64, lx->H[i], lx->W[i], s.st);
    Act *xp = x;
    s.tape.push_back([=]() {
      Step &s = *sp;
      s.grad_of(stem);
      if (s.live())
        for (const BTab &t : make_tabs(ls, lx))
          hipLaunchKernelGGL(maxpool3s2_bwd_multi_kernel, dim3(nblk(tab_rows(t) * 16)), dim3(TB), 0, s.st, stem->v.d, xp->g.d, stem->g.d, 64, t);
    });
  }
  int cin = 64;
  const int planes_of[4] = {64, 128, 256, 512}, stride_of[4] = {1, 2, 1, 1}, dil_of[4] = {1, 1, 2, 4};
  Act *cat = nullptr;      // [rows][2560]: the four pyramid priors then layer4's output (lib/pspnet.py:23)
  for (int li = 1; li <= 4; ++li) {
    const int planes = planes_of[li - 1];
    const std::string base = C + "feats.layer" + std::to_string(li) + ".";
    x = basic_block(s, x, cin, base + "0.", planes, stride_of[li - 1], 1, cin != planes || stride_of[li - 1] != 1);
    Act *into = nullptr;
    if (li == 4) {          // the last block writes straight into its slot of the PSP concatenation
      cat = s.act(x->lv, 2560);
      into = slice(s, cat, 2048, 512);
    }
    x = basic_block(s, x, planes, base + "1.", planes, 1, dil_of[li - 1], false, into);
    cin = planes;
  }
  const Lv *l8 = cat->lv;  // the 1/8-resolution level
  // PSP module (lib/pspnet.py:20-24): pool -> 1x1 conv -> bilinear (align_corners=False) into the concat slots.  The pooled maps of
  // every frame of every bucket sit in ONE set of stage blocks ([4][B*36][512], frames in bucket order): the four stage convolutions are
  // single GEMMs; pooling and resampling (and their adjoints) run per bucket
  Act *feat = x;           // == cat[:, 2048:2560]
  float *pyr = s.f((size_t)4 * B * 36 * 512);
  if (s.live())
    for (int i = 0; i < nb; ++i)
      launch_psp_pool(feat->v.d + l8->off[i] * feat->v.ld, feat->v.ld, 0, pyr, l8->B[i], l8->H[i], l8->W[i], 512, s.st, B, l8->b0[i]);
  struct Stages { Act *pooled[4], *z[4]; };
  auto stg = std::make_shared<Stages>();
  const std::vector<BTab> tabs8 = make_tabs(l8);
  s.tape.push_back([=]() {          // (pushed first: runs after the four stage convolutions' backward) the pooling adjoint of all stages, all buckets
    Step &s = *sp;
    const bool acc = s.grad_of(feat);
    Ptr4 dy;
    for (int si = 0; si < 4; ++si) dy.p[si] = stg->pooled[si]->g.d;
    if (s.live())
      for (const BTab &t : tabs8)
        hipLaunchKernelGGL(pool_bwd_all_kernel, dim3(nblk(tab_rows(t) * 128)), dim3(TB), 0, s.st, dy, feat->g.d, feat->g.ld, 128, acc ? 1 : 0, t);
  });
  for (int si = 0; si < 4; ++si) {
    const int sz = si == 0 ? 1 : si == 1 ? 2 : si == 2 ? 3 : 6;
    stg->pooled[si] = s.act((long)B * sz * sz, 512, pyr + (size_t)si * B * 36 * 512, 512);
    stg->z[si] = conv(s, stg->pooled[si], 512, ConvW{C + "psp.stages." + std::to_string(si) + ".1.weight"}, 512, 1, 1, 0, 1, ACT_NONE);
  }
  {
    Ptr4 z;
    for (int si = 0; si < 4; ++si) z.p[si] = stg->z[si]->v.d;
    if (s.live())
      for (const BTab &t : tabs8)
        hipLaunchKernelGGL(bilinear_fwd_all_kernel, dim3(nblk(tab_rows(t) * 4 * 128)), dim3(TB), 0, s.st, z, cat->v.d, cat->v.ld, 128, t);
  }
  s.tape.push_back([=]() {          // (runs before the stage convolutions' backward) the four resampling adjoints in one launch
    Step &s = *sp;
    MPtr4 dz;
    for (int si = 0; si < 4; ++si) { s.grad_of(stg->z[si]); dz.p[si] = stg->z[si]->g.d; }
    if (s.live())
      for (const BTab &t : tabs8)
        hipLaunchKernelGGL(bilinear_bwd_all_kernel, dim3((unsigned)std::min<long>((long)tab_frames(t) * 50 * 16, 65535L * 16)), dim3(TB), 0, s.st, cat->g.d, cat->g.ld,
                           dz, tab_frames(t), 128, t);
  });
  // the concat's gradient buffer is one allocation; layer4's output gradient is its last 512 channels
  s.tape.push_back([=]() {
    feat->g.d = cat->g.d + 2048;
    feat->g.ld = 2560;
    feat->gset = true;
  });
  Act *psp = conv(s, cat, 2560, ConvW{C + "psp.bottleneck.weight", 0, C + "psp.bottleneck.bias"}, 1024, 1, 1, 0, 1, ACT_RELU);
  if (io.dropout) psp = dropout2d(s, psp, 0.3f, io.seed * 4 + 1);
  Act *u1 = upconv(s, psp, C + "up_1.", 1024, 256);
  if (io.dropout) u1 = dropout2d(s, u1, 0.15f, io.seed * 4 + 2);
  Act *u2 = upconv(s, u1, C + "up_2.", 256, 64);
  if (io.dropout) u2 = dropout2d(s, u2, 0.15f, io.seed * 4 + 3);
  // up_3 + final 1x1 + LogSoftmax at the chosen pixels only (lib/network.py:98-102 reads nothing else)
  Act *patch = s.act((long)rows, 576);
  const Lv *l2 = u2->lv;   // half resolution
  if (s.live())
    for (int i = 0; i < nb; ++i)
      launch_up3_patches(u2->v.d + l2->off[i] * 64, io.choose + (size_t)l2->b0[i] * N, patch->v.d + (size_t)l2->b0[i] * Npad * 576, l2->B[i], l2->H[i], l2->W[i], N,
                         Npad, s.st);
  s.tape.push_back([=]() {
    Step &s = *sp;
    s.grad_of(u2);
    int hmax = 0, wmax = 0;
    for (int i = 0; i < nb; ++i) { hmax = std::max(hmax, l2->H[i]); wmax = std::max(wmax, l2->W[i]); }
    for (const BTab &t : make_tabs(l2)) {
      const int fr = tab_frames(t), f0 = t.b0[0];
      const size_t mark = s.off;
      int4 *tabo = reinterpret_cast<int4 *>(s.bytes((size_t)fr * N * sizeof(int4)));
      int *rowlist = reinterpret_cast<int *>(s.bytes((size_t)fr * hmax * N * sizeof(int)));
      int *rowcnt = reinterpret_cast<int *>(s.bytes((size_t)fr * hmax * sizeof(int)));
      if (s.live()) {
        hipLaunchKernelGGL(up3_decode_multi_kernel, dim3(nblk((long)fr * N)), dim3(TB), 0, s.st, io.choose + (size_t)f0 * N, tabo, fr, N, t);
        hipLaunchKernelGGL(up3_rowlist_multi_kernel, dim3(hmax, fr), dim3(TB), 0, s.st, tabo, rowlist, rowcnt, hmax, N, t);
        hipLaunchKernelGGL(up3_patch_bwd_multi_kernel, dim3((wmax + TB / 16 - 1) / (TB / 16), hmax, fr), dim3(TB), 0, s.st, patch->g.d + (size_t)f0 * Npad * 576, tabo,
                           rowlist, rowcnt, u2->g.d, hmax, N, Npad, t);
      }
      s.off = mark;
    }
  });
  Act *z3 = conv(s, patch, 576, ConvW{C + "up_3.conv.1.weight", 0, C + "up_3.conv.1.bias", 0, C + "up_3.conv.2.weight"}, 64, 1, 1, 0, 1, ACT_PRELU);
  Act *emb_pm = s.act((long)rows, 32);
  float *emb = io.emb ? io.emb : s.f((size_t)B * 32 * N);
  if (s.live()) {
    hipMemsetAsync(emb_pm->v.d, 0, (size_t)rows * 32 * sizeof(float), s.st);      // rows n >= N feed e_conv1: keep them finite
    launch_final_logsoftmax(z3->v.d, s.p(C + "final.0.weight"), s.p(C + "final.0.bias"), emb, emb_pm->v.d, B, N, Npad, s.st);
  }
  s.tape.push_back([=]() {
    Step &s = *sp;
    // LogSoftmax adjoint on the log-probabilities, then the 1x1 conv 64 -> 32 as a GEMM over the chosen pixels' rows
    float *dlog = s.f((size_t)rows * 32);
    if (s.live()) s.fail(df_logsoftmax(emb_pm->g.d, emb_pm->v.d, dlog, rows, 32, 1, s.st));
    Act lg;
    lg.lv = z3->lv; lg.C = 32; lg.v.d = nullptr; lg.v.ld = 32;
    ConvParams f = flat_params(z3, 64, s.p(C + "final.0.weight"), nullptr, &lg, ACT_NONE);
    wgrad(s, f, View{dlog, 32}, s.gr(C + "final.0.weight"), s.gr(C + "final.0.bias"));
    s.grad_of(z3);
    dgrad(s, f, View{dlog, 32}, z3->g, s.pf(C + "final.0.weight"), false);
  });

  // ---- PoseNetFeat (lib/network.py:53-68) on point rows padded to Npad per frame: pf = [x1 64 | e1 64 | x2 128 | e2 128] ----
  Act *pf = s.act((long)rows, 384);
  Act *x1 = slice(s, pf, 0, 64), *e1 = slice(s, pf, 64, 64), *x2 = slice(s, pf, 128, 128), *e2 = slice(s, pf, 256, 128);
  if (s.live()) {
    hipMemsetAsync(pf->v.d, 0, (size_t)rows * 384 * sizeof(float), s.st);
    launch_cloud_conv1(io.cloud, nullptr, s.p("feat.conv1.weight"), s.p("feat.conv1.bias"), pf->v.d, 384, B, N, Npad, s.st);
  }
  // every slice's gradient lives in ONE [rows][384] buffer, written first by head layer 1's data gradient
  auto pf_grad_views = [=]() {
    Act *sl[4] = {x1, e1, x2, e2};
    const int c0[4] = {0, 64, 128, 256};
    for (int i = 0; i < 4; ++i) { sl[i]->g.d = pf->g.d + c0[i]; sl[i]->g.ld = 384; sl[i]->gset = true; }
  };
  s.tape.push_back([=]() {          // conv1's parameters (runs last of the point branch: x1's gradient is complete by then)
    Step &s = *sp;
    Act m = *x1;
    launch_act_bwd(s, &m, ACT_RELU, nullptr, nullptr);
    const int chunks = (N + 63) / 64;
    float *part = s.f((size_t)B * chunks * 64 * 4);
    if (s.live()) {
      hipLaunchKernelGGL(cloud_conv1_bwd_kernel, dim3(B * chunks), dim3(64), 0, s.st, x1->g.d, 384, io.cloud, B, N, Npad, part);
      hipLaunchKernelGGL(cloud_conv1_bwd_finish_kernel, dim3(1), dim3(64), 0, s.st, part, B * chunks, s.gr("feat.conv1.weight"), s.gr("feat.conv1.bias"));
    }
  });
  conv(s, emb_pm, 32, ConvW{"feat.e_conv1.weight", 0, "feat.e_conv1.bias"}, 64, 1, 1, 0, 1, ACT_RELU, nullptr, e1);
  conv(s, x1, 64, ConvW{"feat.conv2.weight", 0, "feat.conv2.bias"}, 128, 1, 1, 0, 1, ACT_RELU, nullptr, x2);
  conv(s, e1, 64, ConvW{"feat.e_conv2.weight", 0, "feat.e_conv2.bias"}, 128, 1, 1, 0, 1, ACT_RELU, nullptr, e2);
  Act *pf2 = slice(s, pf, 128, 256);
  Act *x5 = conv(s, pf2, 256, ConvW{"feat.conv5.weight", 0, "feat.conv5.bias"}, 512, 1, 1, 0, 1, ACT_RELU);
  // conv6 + ReLU; its mean over the points (AvgPool1d) from the GEMM's fused column sums
  Act *x6 = s.act((long)rows, 1024);
  ConvParams p6 = flat_params(x5, 512, s.p("feat.conv6.weight"), s.p("feat.conv6.bias"), x6, ACT_RELU);
  p6.rows_per_group = Npad; p6.rows_valid = N;
  int prow;
  {
    ConvParams q6 = p6;
    q6.out = nullptr;            // (the partial-row count is that of the column-sum launch's tile, chosen when out is null or colsum set)
    prow = conv_colsum_rows(q6);
  }
  float *partial = s.f((size_t)prow * 1024);
  p6.colsum = partial;
  s.gemm(GK_FWD, p6);
  float *apx = s.f((size_t)B * 1024), *dap = s.f((size_t)B * 1024);
  if (s.live()) launch_colsum_finish(partial, prow / B, apx, B, 1024, N, s.st);
  s.tape.push_back([=]() {
    Step &s = *sp;
    s.grad_of(x6);
    if (s.live()) hipLaunchKernelGGL(mask_bcast_kernel, dim3(nblk((long)rows * 256)), dim3(TB), 0, s.st, x6->v.d, dap, x6->g.d, B, N, Npad, 256);
    ConvParams f = p6;
    f.colsum = nullptr;
    wgrad(s, f, x6->g, s.gr("feat.conv6.weight"), s.gr("feat.conv6.bias"));
    const bool acc = s.grad_of(x5);
    dgrad(s, f, x6->g, x5->g, s.pf("feat.conv6.weight"), acc);
  });

  // ---- heads (lib/network.py:107-131): layer 1 with the global feature folded into a per-frame bias, towers stacked r, t, c ----
  float *gbias = s.f((size_t)B * 1920), *s1 = s.f((size_t)B * 1920);
  if (s.live()) launch_fc_rows(apx, 1024, 0, s.p("head1.wg"), s.p("head1.bias"), gbias, 1920, B, 1024, 1920, 1, 0, s.st);
  Act *h1 = s.act((long)rows, 1920);
  ConvParams p1 = flat_params(pf, 384, s.p("head1.wpt"), gbias, h1, ACT_RELU);
  p1.rows_per_group = Npad; p1.rows_valid = N; p1.bias_group_ld = 1920;
  s.gemm(GK_FWD, p1);
  s.tape.push_back([=]() {
    Step &s = *sp;
    launch_act_bwd(s, h1, ACT_RELU, nullptr, nullptr);
    wgrad(s, p1, h1->g, s.gr("head1.wpt"), nullptr);
    s.grad_of(pf);
    dgrad(s, p1, h1->g, pf->g, s.pf("head1.wpt"), false);
    pf_grad_views();
    pf2->g.d = pf->g.d + 128; pf2->g.ld = 384; pf2->gset = true;
    if (s.live()) {
      hipLaunchKernelGGL(colsum_obj_kernel, dim3(1920 / 32, B), dim3(256), 0, s.st, h1->g.d, 1920, s1, Npad, 1920, (long)rows);
      hipLaunchKernelGGL(head1_global_wgrad_kernel, dim3(nblk((long)1920 * 256)), dim3(TB), 0, s.st, s1, apx, s.gr("head1.wg"), s.gr("head1.bias"), B, 1920, 256);
    }
    {
      ConvParams q;
      q.in = s1; q.B = B; q.Cin = 1920; q.in_ld = 1920;
      q.wgt = s.pf("head1.wg");
      q.out = dap; q.Cout = 1024; q.out_ld = 1024;
      q.splitk_ws = s.splitk; q.splitk_ws_bytes = s.splitk_bytes;
      s.gemm(GK_DGRAD, q);
    }
  });
  Act *h2 = s.act((long)rows, 768), *h3 = s.act((long)rows, 384);
  auto towers = [&](Act *in, int cin_t, Act *out, int cout_t, const std::string &wname, const std::string &bname) {
    ConvParams p = flat_params(in, cin_t, s.p(wname), s.p(bname), out, ACT_RELU);
    p.Cout = cout_t;
    p.zcount = 3; p.z_in_coff = cin_t; p.z_wgt = (long)cout_t * cin_t; p.z_bias = cout_t; p.z_out_coff = cout_t;
    s.gemm(GK_FWD, p);
    s.tape.push_back([=]() {
      Step &s = *sp;
      launch_act_bwd(s, out, ACT_RELU, nullptr, nullptr);
      for (int z = 0; z < 3; ++z) {
        ConvParams f = p;
        f.in = p.in + (size_t)z * cin_t;
        f.zcount = 1;
        wgrad(s, f, View{out->g.d + (size_t)z * cout_t, out->g.ld}, s.gr(wname, (size_t)z * cout_t * cin_t), s.gr(bname, (size_t)z * cout_t));
      }
      s.grad_of(in);
      dgrad(s, p, out->g, in->g, s.pf(wname), false);
    });
  };
  towers(h1, 640, h2, 256, "head2.w", "head2.bias");
  towers(h2, 256, h3, 128, "head3.w", "head3.bias");
  float *out_r = io.out_r ? io.out_r : s.f((size_t)B * N * 4), *out_t = io.out_t ? io.out_t : s.f((size_t)B * N * 3);
  float *out_c = io.out_c ? io.out_c : s.f((size_t)B * N);
  if (s.live())
    launch_head_final(h3->v.d, s.p("conv4_r.weight"), s.p("conv4_r.bias"), s.p("conv4_t.weight"), s.p("conv4_t.bias"), s.p("conv4_c.weight"),
                      s.p("conv4_c.bias"), io.obj, t.K, out_r, out_t, out_c, B, N, Npad, s.st);
  // ---- loss (lib/loss.py:13-70), one frame at a time like the reference, and its gradient w.r.t. the predictions ----
  float *d_r = s.f((size_t)B * N * 4), *d_t = s.f((size_t)B * N * 3), *d_c = s.f((size_t)B * N);
  float *dis_n = s.f((size_t)B * N);
  int *sel = reinterpret_cast<int *>(s.bytes((size_t)B * N * io.M * sizeof(int)));
  float *np = io.new_points ? io.new_points : s.f((size_t)B * N * 3), *nt = io.new_target ? io.new_target : s.f((size_t)B * io.M * 3);
  if (s.live()) {      // all frames of the pass in a handful of launches (csrc/loss.h), frame by frame the arithmetic of df_loss_forward / _backward
    s.fail(launch_loss_frames(B, io.symmetric, out_r, out_t, out_c, io.target, io.model_points, io.cloud, N, io.M, io.w, io.loss, io.dis, np, nt, dis_n, sel,
                              s.st));
    s.fail(launch_loss_bwd_frames(B, io.symmetric, out_r, out_t, out_c, io.target, io.model_points, io.cloud, sel, dis_n, N, io.M, io.w, 1.f, d_r, d_t, d_c,
                                  s.st));
  }
  s.dbg("forward + loss");
  // ---- backward: the last head layer by hand, then the tape in reverse ----
  {
    s.grad_of(h3);
    float *dz = s.f((size_t)B * N * 8);
    const int chunks = (N + 127) / 128;
    float *part = s.f((size_t)B * chunks * 8 * 128), *zpart = s.f((size_t)B * chunks * 8);
    if (s.live()) {
      hipLaunchKernelGGL(head_final_bwd_kernel, dim3(nblk((long)rows * 96)), dim3(TB), 0, s.st, d_r, d_t, d_c, out_c, s.p("conv4_r.weight"), s.p("conv4_t.weight"),
                         s.p("conv4_c.weight"), io.obj, t.K, h3->g.d, dz, B, N, Npad);
      hipLaunchKernelGGL(head_final_wgrad_kernel, dim3(chunks, B), dim3(TB), 0, s.st, dz, h3->v.d, part, zpart, N, Npad, chunks);
      hipLaunchKernelGGL(head_final_wgrad_finish_kernel, dim3(4), dim3(TB), 0, s.st, part, zpart, io.obj, t.K, s.gr("conv4_r.weight"), s.gr("conv4_r.bias"),
                         s.gr("conv4_t.weight"), s.gr("conv4_t.bias"), s.gr("conv4_c.weight"), s.gr("conv4_c.bias"), B, N, chunks);
    }
  }
  for (size_t i = s.tape.size(); i-- > 0;) {
    s.tape[i]();
    s.dbg("tape entry", std::to_string(i));
  }
}

// ------------------------------------------------------------------------------------------------
// PoseRefineNet step (lib/network.py:151-206 + lib/loss_refiner.py:12-62): one refine iteration of B frames
// ------------------------------------------------------------------------------------------------
struct RefinerIO {
  int B, M;
  const float *points, *emb, *target, *model_points;      // [B][N][3], [B][32][N], [B][M][3], [B][M][3]
  const int64_t *obj;
  const int *symmetric;
  float *dis, *new_points, *new_target;                   // [B], [B][N][3], [B][M][3]
};

void refiner_step(Step &s, const RefinerIO &io) {
  Trainer &t = *s.t;
  const int B = io.B, N = t.N, Npad = round_up(N, 128), rows = B * Npad;
  Step *sp = &s;
  s.splitk_bytes = (size_t)8 << 20;
  s.splitk = static_cast<float *>(s.bytes(s.splitk_bytes));
  if (!t.splitk) { s.splitk = nullptr; s.splitk_bytes = 0; }
  Act *emb_pm = s.act((long)rows, 32);
  if (s.live()) {
    hipMemsetAsync(emb_pm->v.d, 0, (size_t)rows * 32 * sizeof(float), s.st);
    launch_emb_to_pm(io.emb, emb_pm->v.d, B, N, Npad, s.st);
  }
  // pointfeat_3 = [x1 64 | e1 64 | x2 128 | e2 128] (lib/network.py:160-163)
  Act *pf = s.act((long)rows, 384);
  Act *x1 = slice(s, pf, 0, 64), *e1 = slice(s, pf, 64, 64), *x2 = slice(s, pf, 128, 128), *e2 = slice(s, pf, 256, 128);
  if (s.live()) {
    hipMemsetAsync(pf->v.d, 0, (size_t)rows * 384 * sizeof(float), s.st);
    launch_cloud_conv1(io.points, nullptr, s.p("feat.conv1.weight"), s.p("feat.conv1.bias"), pf->v.d, 384, B, N, Npad, s.st);
  }
  s.tape.push_back([=]() {
    Step &s = *sp;
    Act m = *x1;
    launch_act_bwd(s, &m, ACT_RELU, nullptr, nullptr);
    const int chunks = (N + 63) / 64;
    float *part = s.f((size_t)B * chunks * 64 * 4);
    if (s.live()) {
      hipLaunchKernelGGL(cloud_conv1_bwd_kernel, dim3(B * chunks), dim3(64), 0, s.st, x1->g.d, 384, io.points, B, N, Npad, part);
      hipLaunchKernelGGL(cloud_conv1_bwd_finish_kernel, dim3(1), dim3(64), 0, s.st, part, B * chunks, s.gr("feat.conv1.weight"), s.gr("feat.conv1.bias"));
    }
  });
  conv(s, emb_pm, 32, ConvW{"feat.e_conv1.weight", 0, "feat.e_conv1.bias"}, 64, 1, 1, 0, 1, ACT_RELU, nullptr, e1, false);
  conv(s, x1, 64, ConvW{"feat.conv2.weight", 0, "feat.conv2.bias"}, 128, 1, 1, 0, 1, ACT_RELU, nullptr, x2);
  conv(s, e1, 64, ConvW{"feat.e_conv2.weight", 0, "feat.e_conv2.bias"}, 128, 1, 1, 0, 1, ACT_RELU, nullptr, e2);
  // conv5 reads all 384 channels: its data gradient is the first writer of pf's gradient buffer
  Act *x5 = s.act((long)rows, 512);
  {
    ConvParams p5 = flat_params(pf, 384, s.p("feat.conv5.weight"), s.p("feat.conv5.bias"), x5, ACT_RELU);
    s.gemm(GK_FWD, p5);
    s.tape.push_back([=]() {
      Step &s = *sp;
      launch_act_bwd(s, x5, ACT_RELU, nullptr, nullptr);
      wgrad(s, p5, x5->g, s.gr("feat.conv5.weight"), s.gr("feat.conv5.bias"));
      s.grad_of(pf);
      dgrad(s, p5, x5->g, pf->g, s.pf("feat.conv5.weight"), false);
      Act *sl[4] = {x1, e1, x2, e2};
      const int c0[4] = {0, 64, 128, 256};
      for (int i = 0; i < 4; ++i) { sl[i]->g.d = pf->g.d + c0[i]; sl[i]->g.ld = 384; sl[i]->gset = true; }
    });
  }
  Act *x6 = s.act((long)rows, 1024);
  ConvParams p6 = flat_params(x5, 512, s.p("feat.conv6.weight"), s.p("feat.conv6.bias"), x6, ACT_RELU);
  p6.rows_per_group = Npad; p6.rows_valid = N;
  int prow;
  {
    ConvParams q6 = p6;
    q6.out = nullptr;            // (the partial-row count is that of the column-sum launch's tile, chosen when out is null or colsum set)
    prow = conv_colsum_rows(q6);
  }
  float *partial = s.f((size_t)prow * 1024);
  p6.colsum = partial;
  s.gemm(GK_FWD, p6);
  Act *ap = s.act((long)B, 1024);
  if (s.live()) launch_colsum_finish(partial, prow / B, ap->v.d, B, 1024, N, s.st);
  s.tape.push_back([=]() {
    Step &s = *sp;
    s.grad_of(x6);
    if (s.live()) hipLaunchKernelGGL(mask_bcast_kernel, dim3(nblk((long)rows * 256)), dim3(TB), 0, s.st, x6->v.d, ap->g.d, x6->g.d, B, N, Npad, 256);
    ConvParams f = p6;
    f.colsum = nullptr;
    wgrad(s, f, x6->g, s.gr("feat.conv6.weight"), s.gr("feat.conv6.bias"));
    s.grad_of(x5);
    dgrad(s, f, x6->g, x5->g, s.pf("feat.conv6.weight"), false);
  });
  // FC towers 1024 -> 512 -> 128 (lib/network.py:191-196), one row per frame; f2 = [r 128 | t 128]
  Act *f1 = s.act((long)B, 1024), *f2 = s.act((long)B, 256);
  Act *f1r = slice(s, f1, 0, 512), *f1t = slice(s, f1, 512, 512), *f2r = slice(s, f2, 0, 128), *f2t = slice(s, f2, 128, 128);
  conv(s, ap, 1024, ConvW{"conv1_r.weight", 0, "conv1_r.bias"}, 512, 1, 1, 0, 1, ACT_RELU, nullptr, f1r);
  conv(s, ap, 1024, ConvW{"conv1_t.weight", 0, "conv1_t.bias"}, 512, 1, 1, 0, 1, ACT_RELU, nullptr, f1t);
  conv(s, f1r, 512, ConvW{"conv2_r.weight", 0, "conv2_r.bias"}, 128, 1, 1, 0, 1, ACT_RELU, nullptr, f2r);
  conv(s, f1t, 512, ConvW{"conv2_t.weight", 0, "conv2_t.bias"}, 128, 1, 1, 0, 1, ACT_RELU, nullptr, f2t);
  float *out_r = s.f((size_t)B * 4), *out_t = s.f((size_t)B * 3), *d_r = s.f((size_t)B * 4), *d_t = s.f((size_t)B * 3);
  if (s.live())
    hipLaunchKernelGGL(refiner_tail_fwd_kernel, dim3(B), dim3(64), 0, s.st, f2->v.d, s.p("conv3_r.weight"), s.p("conv3_r.bias"), s.p("conv3_t.weight"),
                       s.p("conv3_t.bias"), io.obj, t.K, out_r, out_t, B);
  int *sel = reinterpret_cast<int *>(s.bytes((size_t)B * io.M * sizeof(int)));
  if (s.live()) {      // all frames in a handful of launches (csrc/loss.h)
    s.fail(launch_loss_refine_frames(B, io.symmetric, out_r, out_t, io.target, io.model_points, io.points, N, io.M, io.dis, io.new_points, io.new_target, sel,
                                     s.st));
    s.fail(launch_loss_refine_bwd_frames(B, io.symmetric, out_r, out_t, io.target, io.model_points, sel, io.M, 1.f, d_r, d_t, s.st));
  }
  {
    s.grad_of(f2);
    f2r->g.d = f2->g.d; f2r->g.ld = 256; f2r->gset = true;
    f2t->g.d = f2->g.d + 128; f2t->g.ld = 256; f2t->gset = true;
    s.grad_of(f1);
    f1r->g.d = f1->g.d; f1r->g.ld = 1024; f1r->gset = false;
    f1t->g.d = f1->g.d + 512; f1t->g.ld = 1024; f1t->gset = false;
    if (s.live())
      hipLaunchKernelGGL(refiner_tail_bwd_kernel, dim3(1), dim3(128), 0, s.st, d_r, d_t, f2->v.d, s.p("conv3_r.weight"), s.p("conv3_t.weight"), io.obj, t.K,
                         f2->g.d, s.gr("conv3_r.weight"), s.gr("conv3_r.bias"), s.gr("conv3_t.weight"), s.gr("conv3_t.bias"), B);
  }
  for (size_t i = s.tape.size(); i-- > 0;) {
    s.tape[i]();
    s.dbg("tape entry", std::to_string(i));
  }
}

Trainer *as_trainer(df_trainer *h) { return reinterpret_cast<Trainer *>(h); }
const Trainer *as_trainer(const df_trainer *h) { return reinterpret_cast<const Trainer *>(h); }

}  // namespace
}  // namespace df

using namespace df;

extern "C" df_trainer *df_trainer_create(int kind, int num_points, int num_obj) {
  if ((kind != 0 && kind != 1) || num_points <= 0 || num_obj <= 0) { set_error(DF_ERR_ARG, "trainer_create: bad arguments"); return nullptr; }
  Trainer *t = new Trainer();
  t->kind = kind; t->N = num_points; t->K = num_obj;
  hipGetDevice(&t->device);
  if (kind == 0) build_posenet(*t);
  else build_refiner(*t);
  // the data gradients' flipped / transposed weight copies; without a device (layout / workspace queries on a CPU-only host) the
  // handle still works for everything that launches nothing, and a step reports the missing arena
  if (hipMalloc(&t->wflip, t->flat * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); t->wflip = nullptr; }
  if (t->wino_floats && hipMalloc(&t->wino_buf, t->wino_floats * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); t->wino_buf = nullptr; }
  {
    std::vector<FlipSeg> tab;
    long begin = 0;
    for (const Trainer::Flip &f : t->flips) {
      tab.push_back(FlipSeg{(long)f.off, begin, f.O, f.T, f.I, f.KH, f.KW, f.Z});
      begin += (long)f.O * f.T * f.I * f.Z;
    }
    t->flip_total = begin;
    if (t->wflip && hipMalloc(&t->flip_tab, tab.size() * sizeof(FlipSeg)) == hipSuccess)
      hipMemcpy(t->flip_tab, tab.data(), tab.size() * sizeof(FlipSeg), hipMemcpyHostToDevice);
    else { (void)hipGetLastError(); t->flip_tab = nullptr; }
    std::vector<FlipTile> tiles;
    long tb = 0;
    for (const Trainer::Flip &f : t->flips) {
      const int nb_n = (f.O + 31) / 32, nb_c = (f.I + 31) / 32;
      tiles.push_back(FlipTile{(long)f.off, (int)tb, f.O, f.T, f.I, f.KH, f.KW, f.Z, nb_n, nb_c});
      tb += (long)f.Z * f.T * nb_n * nb_c;
    }
    t->flip_ntiles = (int)tb;
    if (t->wflip && !tiles.empty() && tb < (1L << 31) && hipMalloc(&t->flip_tiles, tiles.size() * sizeof(FlipTile)) == hipSuccess)
      hipMemcpy(t->flip_tiles, tiles.data(), tiles.size() * sizeof(FlipTile), hipMemcpyHostToDevice);
    else { (void)hipGetLastError(); t->flip_tiles = nullptr; }
  }
  return reinterpret_cast<df_trainer *>(t);
}

extern "C" void df_trainer_destroy(df_trainer *h) {
  if (!h) return;
  Trainer *t = as_trainer(h);
  if (t->wflip) hipFree(t->wflip);
  if (t->wino_buf) hipFree(t->wino_buf);
  if (t->flip_tab) hipFree(t->flip_tab);
  if (t->flip_tiles) hipFree(t->flip_tiles);
  for (auto e : t->ev) hipEventDestroy(e);
  delete t;
}

extern "C" int64_t df_trainer_flat_numel(const df_trainer *h) { return h ? (int64_t)as_trainer(h)->flat : 0; }
extern "C" int df_trainer_num_params(const df_trainer *h) { return h ? (int)as_trainer(h)->spec.size() : 0; }

extern "C" int df_trainer_param_info(const df_trainer *h, int i, char *key_out, int key_cap, int64_t *shape4, int *ndim) {
  if (!h) return set_error(DF_ERR_ARG, "trainer_param_info: null handle");
  const Trainer *t = as_trainer(h);
  if (i < 0 || i >= (int)t->spec.size()) return set_error(DF_ERR_ARG, "trainer_param_info: index out of range");
  const PSpec &p = t->spec[i];
  if (key_out && key_cap > 0) { strncpy(key_out, p.key.c_str(), key_cap - 1); key_out[key_cap - 1] = 0; }
  if (shape4) for (int d = 0; d < 4; ++d) shape4[d] = p.shape[d];
  if (ndim) *ndim = p.ndim;
  return DF_OK;
}

// dir 0: reference layout (`ref`, device) -> its place in the flat kernel-layout buffer; dir 1: back
static int relayout(const Trainer &t, const char *key, float *ref, float *flat, int dir, hipStream_t st) {
  if (!key || !ref || !flat) return set_error(DF_ERR_ARG, "trainer pack/unpack: null pointer");
  auto it = t.index.find(key);
  if (it == t.index.end()) return set_error(DF_ERR_ARG, "trainer pack/unpack: unexpected key '%s'", key);
  const PSpec &p = t.spec[it->second];
  if (p.mode == 0 || p.mode == 1) {
    const int O = (int)p.shape[0], I = (int)p.shape[1], T = (int)(p.shape[2] * p.shape[3]), Ipad = p.mode == 0 ? (I + 3) / 4 * 4 : I;
    hipLaunchKernelGGL(relayout_kernel, dim3(nblk((long)O * T * Ipad, 2048)), dim3(TB), 0, st, dir == 0 ? ref : flat + p.off, dir == 0 ? flat + p.off : ref, O, I, T,
                       Ipad, p.mode, dir);
  } else if (p.mode == 3) {       // [640][1408] <-> [640][384] + [640][1024]
    if (dir == 0) {
      hipLaunchKernelGGL(copy2d_kernel, dim3(nblk(640L * 384)), dim3(TB), 0, st, ref, 1408L, flat + p.off, 384L, 640L, 384L);
      hipLaunchKernelGGL(copy2d_kernel, dim3(nblk(640L * 1024)), dim3(TB), 0, st, ref + 384, 1408L, flat + p.off2, 1024L, 640L, 1024L);
    } else {
      hipLaunchKernelGGL(copy2d_kernel, dim3(nblk(640L * 384)), dim3(TB), 0, st, flat + p.off, 384L, ref, 1408L, 640L, 384L);
      hipLaunchKernelGGL(copy2d_kernel, dim3(nblk(640L * 1024)), dim3(TB), 0, st, flat + p.off2, 1024L, ref + 384, 1408L, 640L, 1024L);
    }
  } else {
    const long n = (long)p.numel();
    hipLaunchKernelGGL(copy2d_kernel, dim3(nblk(n)), dim3(TB), 0, st, dir == 0 ? ref : flat + p.off, n, dir == 0 ? flat + p.off : ref, n, 1L, n);
  }
  return check_launch("trainer pack/unpack");
}

extern "C" int df_trainer_pack_param(const df_trainer *h, const char *key, const float *src, float *flat, df_stream_t stream) {
  if (!h) return set_error(DF_ERR_ARG, "trainer_pack_param: null handle");
  return relayout(*as_trainer(h), key, const_cast<float *>(src), flat, 0, to_stream(stream));
}
extern "C" int df_trainer_unpack_param(const df_trainer *h, const char *key, const float *flat, float *dst, df_stream_t stream) {
  if (!h) return set_error(DF_ERR_ARG, "trainer_unpack_param: null handle");
  return relayout(*as_trainer(h), key, dst, const_cast<float *>(flat), 1, to_stream(stream));
}

static int posenet_buckets_ok(const Trainer *t, int nb, const int *B, const int *H, const int *W, int M, const char *what) {
  if (!t || t->kind != 0) return set_error(DF_ERR_ARG, "%s: not a PoseNet trainer", what);
  if (nb <= 0 || nb > 4096 || !B || !H || !W || M <= 0) return set_error(DF_ERR_ARG, "%s: need 1..4096 buckets with B / H / W arrays and M >= 1", what);
  long tot = 0;
  for (int i = 0; i < nb; ++i) {
    if (B[i] <= 0 || H[i] < 8 || W[i] < 8 || H[i] > DF_MAX_CROP || W[i] > DF_MAX_CROP)
      return set_error(DF_ERR_ARG, "%s: bucket %d: need B >= 1 and 8 <= H, W <= %d (got %d, %d, %d)", what, i, DF_MAX_CROP, B[i], H[i], W[i]);
    tot += B[i];
  }
  if (tot > 65535) return set_error(DF_ERR_ARG, "%s: too many frames in one pass (%ld)", what, tot);
  return DF_OK;
}

extern "C" size_t df_posenet_train_multi_workspace_bytes(const df_trainer *h, int nb, const int *B, const int *H, const int *W, int M) {
  if (!h || posenet_buckets_ok(as_trainer(h), nb, B, H, W, M, "posenet_train_workspace_bytes") != DF_OK) return 0;
  Trainer &t = *const_cast<Trainer *>(as_trainer(h));
  std::vector<int> key{nb, M};
  for (int i = 0; i < nb; ++i) { key.push_back(B[i]); key.push_back(H[i]); key.push_back(W[i]); }
  auto it = t.ws_cache.find(key);
  if (it != t.ws_cache.end()) return it->second;
  Step s{&t, nullptr, true, nullptr};
  PoseNetIO io{};
  io.nb = nb; io.B = B; io.H = H; io.W = W; io.M = M; io.dropout = 1;
  posenet_step(s, io);
  if (t.ws_cache.size() > 4096) t.ws_cache.clear();
  t.ws_cache[key] = s.peak;
  return s.peak;
}

extern "C" size_t df_posenet_train_workspace_bytes(const df_trainer *h, int B, int H, int W, int M) {
  return df_posenet_train_multi_workspace_bytes(h, 1, &B, &H, &W, M);
}

extern "C" int df_posenet_train_step_multi(df_trainer *h, const float *flat_param, float *flat_grad, int64_t param_version, int nb, const int *B,
                                           const int *H, const int *W, const float *const *img, const float *cloud, const int64_t *choose,
                                           const int64_t *obj, const float *target, const float *model_points, int M, const int *symmetric_host,
                                           float w, int dropout, unsigned seed, float *loss_out, float *dis_out, float *new_points,
                                           float *new_target, float *out_r, float *out_t, float *out_c, float *emb, void *ws, size_t ws_bytes,
                                           df_stream_t stream) {
  if (!h) return set_error(DF_ERR_ARG, "posenet_train_step: null handle");
  int rc = posenet_buckets_ok(as_trainer(h), nb, B, H, W, M, "posenet_train_step");
  if (rc != DF_OK) return rc;
  if (!flat_param || !flat_grad || !img || !cloud || !choose || !obj || !target || !model_points || !loss_out || !dis_out || !ws)
    return set_error(DF_ERR_ARG, "posenet_train_step: null pointer");
  for (int i = 0; i < nb; ++i)
    if (!img[i]) return set_error(DF_ERR_ARG, "posenet_train_step: bucket %d: null image pointer", i);
  Trainer &t = *as_trainer(h);
  if (df_posenet_train_multi_workspace_bytes(h, nb, B, H, W, M) > ws_bytes) return set_error(DF_ERR_WORKSPACE, "posenet_train_step: workspace too small");
  rc = check_flips(t, flat_param, (long)param_version, to_stream(stream));
  if (rc != DF_OK) return rc;
  Step s{&t, to_stream(stream), false, static_cast<char *>(ws)};
#ifdef DF_DEV
  if (df::dev_getenv("DF_TRAIN_OVERLAP")) {
    if (!t.side) {
      hipStreamCreateWithFlags(&t.side, hipStreamNonBlocking);
      hipEventCreateWithFlags(&t.ev_fork, hipEventDisableTiming);
      hipEventCreateWithFlags(&t.ev_join, hipEventDisableTiming);
    }
    s.side = t.side; s.ev_fork = t.ev_fork; s.ev_join = t.ev_join;
  }
#endif
  s.cap = ws_bytes; s.P = flat_param; s.G = flat_grad;
  PoseNetIO io{nb, B, H, W, img, M, cloud, target, model_points, choose, obj, symmetric_host, w, dropout, seed, loss_out, dis_out, new_points, new_target,
               out_r, out_t, out_c, emb};
  posenet_step(s, io);
  if (s.err != DF_OK) return s.err;
  return check_launch("posenet_train_step");
}

extern "C" int df_posenet_train_step(df_trainer *h, const float *flat_param, float *flat_grad, int64_t param_version, int B, int H, int W,
                                     const float *img, const float *cloud, const int64_t *choose, const int64_t *obj, const float *target,
                                     const float *model_points, int M, const int *symmetric_host, float w, int dropout, unsigned seed,
                                     float *loss_out, float *dis_out, float *new_points, float *new_target, float *out_r, float *out_t,
                                     float *out_c, float *emb, void *ws, size_t ws_bytes, df_stream_t stream) {
  return df_posenet_train_step_multi(h, flat_param, flat_grad, param_version, 1, &B, &H, &W, &img, cloud, choose, obj, target, model_points, M,
                                     symmetric_host, w, dropout, seed, loss_out, dis_out, new_points, new_target, out_r, out_t, out_c, emb, ws, ws_bytes,
                                     stream);
}

// Split-K of the small-grid forward / data-gradient launches (on by default: +4-8 % on one-frame passes).  Off: every output element is
// summed in ONE order whatever the grid, so the gradient of a frame no longer depends on which other frames share its pass (up to the
// weight gradients' own pixel order): what the equality tests of the multi-bucket pass switch off.
extern "C" int df_trainer_set_splitk(df_trainer *h, int enable) {
  if (!h) return set_error(DF_ERR_ARG, "trainer_set_splitk: null handle");
  as_trainer(h)->splitk = enable != 0;
  return DF_OK;
}

// Profile of the MFMA launches of the steps run on this handle since df_trainer_profile(h, 1): HIP events bracket every forward /
// data-gradient / weight-gradient GEMM on the launch stream; the FLOPs are those the launches EXECUTE (2 M N K of the shapes really run:
// low-resolution up-convolutions, folded head layer 1, chosen-pixel up_3, F(4x4,3x3)-domain products), not the reference graph's.
extern "C" int df_trainer_profile(df_trainer *h, int enable) {
  if (!h) return set_error(DF_ERR_ARG, "trainer_profile: null handle");
  Trainer *t = as_trainer(h);
  t->profiling = enable != 0;
  t->ev_used = 0;
  t->ev_kind.clear();
  t->ev_flops.clear();
  return DF_OK;
}

// after a stream sync: per kind (0 forward, 1 data gradient, 2 weight gradient) the summed launch durations (ms), executed FLOPs, launches
extern "C" int df_trainer_profile_read(df_trainer *h, double *ms3, double *flops3, int *launches3) {
  if (!h || !ms3 || !flops3 || !launches3) return set_error(DF_ERR_ARG, "trainer_profile_read: null pointer");
  Trainer *t = as_trainer(h);
  for (int k = 0; k < 3; ++k) { ms3[k] = 0; flops3[k] = 0; launches3[k] = 0; }
  for (size_t i = 0; i + 1 < t->ev_used; i += 2) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, t->ev[i], t->ev[i + 1]) != hipSuccess) return set_error(DF_ERR_LAUNCH, "trainer_profile_read: events not complete");
    const int k = t->ev_kind[i / 2];
    ms3[k] += ms; flops3[k] += t->ev_flops[i / 2]; launches3[k] += 1;
#ifdef DF_DEV
    static const bool verbose = dev_getenv("DF_PROFILE_VERBOSE") != nullptr;      // dev switch: one line per MFMA launch of the profiled steps
    if (verbose && i / 2 < t->ev_desc.size())
      fprintf(stderr, "[df-train-gemm] %s %-44s %9.1f us %7.1f TFLOP/s\n", k == 0 ? "fwd  " : k == 1 ? "dgrad" : "wgrad", t->ev_desc[i / 2].c_str(), ms * 1e3,
              t->ev_flops[i / 2] / (ms * 1e-3) / 1e12);
#endif
  }
#ifdef DF_DEV
  t->ev_desc.clear();
#endif
  t->ev_used = 0;
  t->ev_kind.clear();
  t->ev_flops.clear();
  return DF_OK;
}

extern "C" size_t df_refiner_train_workspace_bytes(const df_trainer *h, int B, int M) {
  if (!h || as_trainer(h)->kind != 1 || B <= 0 || M <= 0) return 0;
  Trainer &t = *const_cast<Trainer *>(as_trainer(h));
  const std::vector<int> key{B, M};
  auto it = t.ws_cache.find(key);
  if (it != t.ws_cache.end()) return it->second;
  Step s{&t, nullptr, true, nullptr};
  RefinerIO io{};
  io.B = B; io.M = M;
  refiner_step(s, io);
  t.ws_cache[key] = s.peak;
  return s.peak;
}

extern "C" int df_refiner_train_step(df_trainer *h, const float *flat_param, float *flat_grad, int64_t param_version, int B, const float *points,
                                     const float *emb, const int64_t *obj, const float *target, const float *model_points, int M,
                                     const int *symmetric_host, float *dis_out, float *new_points, float *new_target, void *ws, size_t ws_bytes,
                                     df_stream_t stream) {
  if (!h || as_trainer(h)->kind != 1) return set_error(DF_ERR_ARG, "refiner_train_step: not a PoseRefineNet trainer");
  if (B <= 0 || M <= 0) return set_error(DF_ERR_ARG, "refiner_train_step: need B >= 1, M >= 1");
  if (!flat_param || !flat_grad || !points || !emb || !obj || !target || !model_points || !dis_out || !new_points || !new_target || !ws)
    return set_error(DF_ERR_ARG, "refiner_train_step: null pointer");
  Trainer &t = *as_trainer(h);
  if (df_refiner_train_workspace_bytes(h, B, M) > ws_bytes) return set_error(DF_ERR_WORKSPACE, "refiner_train_step: workspace too small");
  int rc = check_flips(t, flat_param, (long)param_version, to_stream(stream));
  if (rc != DF_OK) return rc;
  Step s{&t, to_stream(stream), false, static_cast<char *>(ws)};
  s.cap = ws_bytes; s.P = flat_param; s.G = flat_grad;
  RefinerIO io{B, M, points, emb, target, model_points, obj, symmetric_host, dis_out, new_points, new_target};
  refiner_step(s, io);
  if (s.err != DF_OK) return s.err;
  return check_launch("refiner_train_step");
}

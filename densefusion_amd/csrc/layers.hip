// Memory-bound companion kernels of the DenseFusion forward: layout change, pooling, bilinear
// resampling, the gather + 1x1 + LogSoftmax tail, tiny per-point layers and row reductions.
// All tensors fp32, channels-last; 16-byte vector accesses over the channel axis; 64-wide waves.
#include "layers.h"
#include <cstdlib>

namespace df {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TPB = 256;

inline int blocks_for(long n, int cap = 256 * 32) {
  long b = (n + TPB - 1) / TPB;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

__global__ __launch_bounds__(TPB) void nchw3_to_nhwc4_kernel(const float *__restrict__ img, float *__restrict__ out,
                                                             int B, int HW) {
  const long total = (long)B * HW;
  for (long i = blockIdx.x * (long)TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long b = i / HW, p = i - b * HW;
    const float *s = img + b * 3 * HW + p;
    f32x4 v = {s[0], s[HW], s[2 * (long)HW], 0.f};
    reinterpret_cast<f32x4 *>(out)[i] = v;
  }
}

__global__ __launch_bounds__(TPB) void maxpool3s2_kernel(const float *__restrict__ in, float *__restrict__ out, int B,
                                                         int H, int W, int C4, int OH, int OW) {
  const long total = (long)B * OH * OW * C4;
  for (long i = blockIdx.x * (long)TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const int c = (int)(i % C4);
    long r = i / C4;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const int b = (int)(r / OH);
    const float ninf = -__builtin_inff();
    f32x4 m = {ninf, ninf, ninf, ninf};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const f32x4 v = reinterpret_cast<const f32x4 *>(in)[((long)(b * H + iy) * W + ix) * C4 + c];
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
      }
    }
    reinterpret_cast<f32x4 *>(out)[i] = m;
  }
}

// bilinear source coordinate helpers (ATen UpSample.h semantics, fp32)
__device__ inline void src_ac(int dst, float scale, int in_size, int &i0, int &i1, float &l0, float &l1) {
  const float s = scale * (float)dst;                       // align_corners=True
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
  l0 = 1.f - l1;
}
__device__ inline void src_hp(int dst, float scale, int in_size, int &i0, int &i1, float &l0, float &l1) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;             // align_corners=False (half-pixel), clamped at 0
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
  l0 = 1.f - l1;
}
__device__ inline f32x4 lerp4(const f32x4 &v00, const f32x4 &v01, const f32x4 &v10, const f32x4 &v11, float wy0,
                              float wy1, float wx0, float wx1) {
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = wy0 * (wx0 * v00[e] + wx1 * v01[e]) + wy1 * (wx0 * v10[e] + wx1 * v11[e]);
  return o;
}

__device__ inline void psp_bin(int bin, int &s, int &local) {
  if (bin < 1) { s = 1; local = bin; }
  else if (bin < 5) { s = 2; local = bin - 1; }
  else if (bin < 14) { s = 3; local = bin - 5; }
  else { s = 6; local = bin - 14; }
}

// grid (50, B); thread = one float4 of channels.  Output: 4 stage blocks of B*36 rows each
// ([4][B*36][C], stage s uses its first B*s*s rows) so the 4 stage GEMMs run as one grouped launch.
// 1-D grid of 50 * roundup(B, 8) workgroups, XCD-aware: workgroups are dealt round-robin over the 8 XCDs, so id % 8 picks the XCD
// and the 50 bins of one object are given to ONE of them -- the object's map (<= 2.4 MB at 30 x 40) is fetched into that XCD's L2
// once and the other three pyramid levels re-read it there (with the (bin, object) grid every XCD fetched every map: 3.6 TB/s of
// reads for 0.13 of the algorithmic bandwidth).  Placement is for speed only: any placement gives the same sums.
__global__ __launch_bounds__(512) void psp_pool_kernel(const float *__restrict__ in, int in_ld, int in_coff,
                                                       float *__restrict__ out, int B, int H, int W, int C, int Btot, int b0) {
  __shared__ f32x4 s_part[4][128];
  int s, local;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int b = (q / 50) * 8 + xcd;
  if (b >= B) return;
  psp_bin(q % 50, s, local);
  const int i = local / s, j = local - i * s;
  const int y0 = (i * H) / s, y1 = ((i + 1) * H + s - 1) / s;     // [floor(i*H/s), ceil((i+1)*H/s))
  const int x0 = (j * W) / s, x1 = ((j + 1) * W + s - 1) / s;
  const float cnt = (float)((y1 - y0) * (x1 - x0));
  const int stage = s == 1 ? 0 : s == 2 ? 1 : s == 3 ? 2 : 3;
  float *dst = out + ((size_t)stage * Btot * 36 + (size_t)(b0 + b) * s * s + local) * C;      // stage blocks of Btot objects; this call fills objects b0 ..
  const int tx = threadIdx.x & 127, ty = threadIdx.x >> 7;         // 4 row-interleaved partial sums per channel vector
  for (int c0 = 0; c0 < C; c0 += 128 * 4) {
    const int c = c0 + tx * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
      // the window's pixels, flattened; this thread takes every 4th one, four loads in flight (the 1x1 bin walks the
      // whole map: without the unrolling its chain of dependent loads set the kernel's duration)
      const int bw = x1 - x0, npx = (y1 - y0) * bw;
      const float *base = in + (size_t)b * H * W * in_ld + in_coff + c;
      f32x4 a4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      int p = ty;
      for (; p + 28 < npx; p += 32) {            // eight loads in flight; the additions keep the order of the 4-wide loop below
        f32x4 v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int q = p + 4 * u, yy = y0 + q / bw, xx = x0 + q % bw;
          v8[u] = *reinterpret_cast<const f32x4 *>(base + ((size_t)yy * W + xx) * in_ld);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) a4[u & 3][e] += v8[u][e];
      }
      for (; p + 12 < npx; p += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = p + 4 * u, yy = y0 + q / bw, xx = x0 + q % bw;
          const f32x4 v = *reinterpret_cast<const f32x4 *>(base + ((size_t)yy * W + xx) * in_ld);
#pragma unroll
          for (int e = 0; e < 4; ++e) a4[u][e] += v[e];
        }
      }
      for (; p < npx; p += 4) {
        const int yy = y0 + p / bw, xx = x0 + p % bw;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(base + ((size_t)yy * W + xx) * in_ld);
#pragma unroll
        for (int e = 0; e < 4; ++e) a4[0][e] += v[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = (a4[0][e] + a4[1][e]) + (a4[2][e] + a4[3][e]);
    }
    s_part[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && c < C) {
      f32x4 t = s_part[0][tx];
#pragma unroll
      for (int q = 1; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] += s_part[q][tx][e];
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = t[e] / cnt;
      *reinterpret_cast<f32x4 *>(dst + c) = t;
    }
    __syncthreads();
  }
}

// prior[b][y][x][c] = sum over the 4 pyramid stages of the bilinear (align_corners=False) resampling of
// Z_s[b] (s x s x C) to (H, W): the PSP priors after the bottleneck 1x1 has been folded into the stage
// weights (lib/pspnet.py:20-24; 1x1 conv and bilinear resampling are both linear and commute).
// Workgroup = (object, 64 channels): the object's 50 stage rows of those channels (12.8 KB) and the bilinear source rows / columns /
// weights of the four stages for every map row and column are put in LDS once (a thread used to redo the 8 source computations and
// fetch its 16 corner vectors from L2 for every output vector: 0.20 of the algorithmic bandwidth); thread = (pixel slot, 4 channels).
constexpr int PP_CH = 64;
__global__ __launch_bounds__(TPB) void psp_prior_sum_kernel(const float *__restrict__ z, float *__restrict__ out, int B,
                                                            int H, int W, int C) {
  extern __shared__ __attribute__((aligned(16))) float s_pp[];      // [50][64] stage rows, then the row / column tables
  f32x4 *s_z = reinterpret_cast<f32x4 *>(s_pp);
  int *s_i = reinterpret_cast<int *>(s_pp + 50 * PP_CH);               // [4][H + W][2] first / second source index
  float *s_w = reinterpret_cast<float *>(s_i + 8 * (H + W));           // [4][H + W][2] their weights
  const int chunks = C / PP_CH;
  const int b = blockIdx.x / chunks, c0 = (blockIdx.x % chunks) * PP_CH;
  for (int i = threadIdx.x; i < 50 * (PP_CH / 4); i += TPB) {
    const int row = i / (PP_CH / 4), v = i % (PP_CH / 4);
    int s, local;
    psp_bin(row, s, local);
    const int stage = s == 1 ? 0 : s == 2 ? 1 : s == 3 ? 2 : 3;
    s_z[i] = *reinterpret_cast<const f32x4 *>(z + ((size_t)stage * B * 36 + (size_t)b * s * s + local) * C + c0 + v * 4);
  }
  for (int i = threadIdx.x; i < 4 * (H + W); i += TPB) {
    const int stage = i / (H + W), p = i % (H + W);
    const int s = stage == 0 ? 1 : stage == 1 ? 2 : stage == 2 ? 3 : 6;
    int i0, i1;
    float l0, l1;
    if (p < H) src_hp(p, (float)s / (float)H, s, i0, i1, l0, l1);
    else src_hp(p - H, (float)s / (float)W, s, i0, i1, l0, l1);
    s_i[2 * i] = i0; s_i[2 * i + 1] = i1;
    s_w[2 * i] = l0; s_w[2 * i + 1] = l1;
  }
  __syncthreads();
  const int v = threadIdx.x & 15, slot = threadIdx.x >> 4;
  const int base_of[4] = {0, 1, 5, 14};                       // first stage row of s = 1, 2, 3, 6 among the 50
  for (int p = slot; p < H * W; p += TPB / 16) {
    const int y = p / W, x = p - y * W;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int stage = 0; stage < 4; ++stage) {
      const int s = stage == 0 ? 1 : stage == 1 ? 2 : stage == 2 ? 3 : 6;
      const int ty = (stage * (H + W) + y) * 2, tx = (stage * (H + W) + H + x) * 2;
      const int y0 = s_i[ty], y1 = s_i[ty + 1], x0 = s_i[tx], x1 = s_i[tx + 1];
      const float wy0 = s_w[ty], wy1 = s_w[ty + 1], wx0 = s_w[tx], wx1 = s_w[tx + 1];
      const f32x4 *src = s_z + base_of[stage] * (PP_CH / 4) + v;
      const f32x4 v00 = src[(y0 * s + x0) * (PP_CH / 4)], v01 = src[(y0 * s + x1) * (PP_CH / 4)];
      const f32x4 v10 = src[(y1 * s + x0) * (PP_CH / 4)], v11 = src[(y1 * s + x1) * (PP_CH / 4)];
      const f32x4 t = lerp4(v00, v01, v10, v11, wy0, wy1, wx0, wx1);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += t[e];
    }
    *reinterpret_cast<f32x4 *>(out + ((size_t)b * H * W + p) * C + c0 + v * 4) = acc;
  }
}

// ---- "conv3x3 after bilinear x2" evaluated through the low-resolution per-tap products ----------------
// PSPUpsample = Upsample(x2, align_corners=True) -> Conv3x3(pad 1) -> PReLU (lib/pspnet.py:27-37).  Both the
// resampling and the convolution are linear:  conv(up(x))(P) = b + sum_tap W_tap . up(x)(P + tap)
//                                                            = b + sum_tap up(W_tap . x)(P + tap),
// with taps that fall outside the upsampled image contributing zero (the conv's zero padding).  So the
// nine 1x1 products Y_tap = W_tap . x are taken at LOW resolution (one GEMM with N = 9*Cout, a quarter of
// the conv's FLOPs) and this kernel does the 9-tap x 4-corner interpolation of Y at the output pixels.
// y: [B][h][w][9*Cout] (tap-major channel blocks); out: [B][2h][2w][Cout].
struct Tap3 { int i0[3], i1[3]; float w0[3], w1[3]; bool ok[3]; };
__device__ inline Tap3 taps_for(int P, float scale, int in_size, int out_size) {
  Tap3 t;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int q = P + d - 1;
    t.ok[d] = (unsigned)q < (unsigned)out_size;
    src_ac(t.ok[d] ? q : 0, scale, in_size, t.i0[d], t.i1[d], t.w0[d], t.w1[d]);
  }
  return t;
}

// LDS-tiled: a workgroup owns an 8 x 16 tile of output pixels and 16 output channels.  The
// low-resolution rows / columns its 3x3 taps interpolate from (<= 7 x 11 pixels) are staged in LDS once, all nine tap
// blocks of the 16 channels (44 KB), so the 36 vector reads per output vector come from LDS and global traffic drops
// from 36 to ~5.4 vector reads per output vector.  thread = (pixel of the tile, 4 channels).
// Written for few instructions (this kernel issued half of all the vector instructions of the step's memory-bound kernels, and
// next to another step's GEMM those are issue time its MFMA waves lose): the staged window always has the full 7 x 11 shape
// (rows / columns past the map edge are clamped duplicates nobody reads), so every index decomposition divides by a constant;
// taps outside the upsampled image keep their place in the sum with zero weights (acc + 0 * v = acc: same bits as skipping them)
// instead of branching; the four corner addresses of a tap are row offset + column offset, the tap itself an immediate.
typedef unsigned int u32x4l __attribute__((ext_vector_type(4)));
constexpr int UG_TY = 8, UG_TX = 16, UG_CC = 16, UG_RH = 7, UG_RW = 11;

__global__ __launch_bounds__(512) void upconv_gather_tiled_kernel(const float *__restrict__ y, const float *__restrict__ bias,
                                                                  const float *__restrict__ prelu, float *__restrict__ out, int B,
                                                                  int h, int w, int Cout) {
  __shared__ __attribute__((aligned(16))) float s_y[UG_RH * UG_RW * 9 * UG_CC];
  // per tile row / column and tap: the two LDS offsets and weights of the bilinear source (zero weights for taps outside the image):
  // worked out once per workgroup by 72 threads instead of six source computations in every thread
  __shared__ __attribute__((aligned(16))) int s_trow[3 * UG_TY][4], s_tcol[3 * UG_TX][4];
  const int OH = 2 * h, OW = 2 * w, ldy = 9 * Cout, chunks = Cout / UG_CC;
  const float sh = OH > 1 ? (float)(h - 1) / (float)(OH - 1) : 0.f;
  const float sw = OW > 1 ? (float)(w - 1) / (float)(OW - 1) : 0.f;
  // XCD-aware order (1-D grid; workgroups are dealt to the 8 XCDs round-robin): an XCD walks its own tiles with the channel chunk
  // FASTEST, so the 16-channel (64-byte) pieces a tile's chunks take out of the same 128-byte lines of y meet in that XCD's L2 while
  // they are hot.  (With the chunk as the slowest grid axis the other half of every line was fetched from HBM a second time: the
  // kernel ran at 0.34 of the bandwidth its algorithmic bytes need.)
  const int ntx = (OW + UG_TX - 1) / UG_TX, tiles_img = ntx * ((OH + UG_TY - 1) / UG_TY);
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int chunk = seq % chunks, tile_g = (seq / chunks) * 8 + xcd;          // global tile index over (image, tile)
  if (tile_g >= B * tiles_img) return;
  const int b = tile_g / tiles_img, tile = tile_g - b * tiles_img;
  const int c0 = chunk * UG_CC;
  const int Y0 = (tile / ntx) * UG_TY, X0 = (tile % ntx) * UG_TX;
  // first low-resolution row / column any tap of the tile interpolates from (the window then spans <= 7 x 11 from there)
  int r_lo, c_lo, dummy;
  float fd0, fd1;
  src_ac(max(Y0 - 1, 0), sh, h, r_lo, dummy, fd0, fd1);
  src_ac(max(X0 - 1, 0), sw, w, c_lo, dummy, fd0, fd1);
  // staging: thread = (one of 14 pixel slots, tap, channel vector) -- 36 vectors per low-resolution pixel, 77 pixels in 6 rounds;
  // 32-bit offsets off a buffer descriptor over this image's rows
  {
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(y + (size_t)b * h * w * ldy + c0), 0,
                                                        (unsigned)(((size_t)h * w * ldy - c0) * sizeof(float)), 0x00020000);
    const int slot = threadIdx.x / 36, tc = threadIdx.x - slot * 36;      // tc = tap * 4 + c4
    const unsigned tc_off = (unsigned)((tc >> 2) * Cout + (tc & 3) * 4) * 4u;
    const unsigned pix_bytes = (unsigned)ldy * 4u, row_bytes = (unsigned)w * pix_bytes;
    if (slot < 14) {
#pragma unroll
      for (int it = 0; it < (UG_RH * UG_RW + 13) / 14; ++it) {
        const int pix = slot + 14 * it;
        if (pix < UG_RH * UG_RW) {
          const int ly = pix / UG_RW, lx = pix - ly * UG_RW;
          const int gy = min(r_lo + ly, h - 1), gx = min(c_lo + lx, w - 1);
          const unsigned off = __umul24((unsigned)gy, row_bytes) + __umul24((unsigned)gx, pix_bytes) + tc_off;      // 24-bit factors: host-checked
          reinterpret_cast<u32x4l *>(s_y)[pix * 36 + tc] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, off, 0, 0);
        }
      }
    }
  }
  if (threadIdx.x >= 512 - 3 * (UG_TY + UG_TX)) {          // the last 72 threads fill the tables (24 row entries, 48 column entries)
    const int e = threadIdx.x - (512 - 3 * (UG_TY + UG_TX));
    const bool row = e < 3 * UG_TY;
    const int k = row ? e : e - 3 * UG_TY;
    const int d = row ? k / UG_TY : k / UG_TX, l = row ? k % UG_TY : k % UG_TX;
    const int q = (row ? Y0 : X0) + l + d - 1;
    const bool ok = (unsigned)q < (unsigned)(row ? OH : OW);
    int i0, i1;
    float w0, w1;
    src_ac(ok ? q : 0, row ? sh : sw, row ? h : w, i0, i1, w0, w1);
    int *dst = row ? s_trow[k] : s_tcol[k];
    const int unit = row ? UG_RW * 36 : 36, lo = row ? r_lo : c_lo;
    dst[0] = ok ? (i0 - lo) * unit : 0;
    dst[1] = ok ? (i1 - lo) * unit : 0;
    dst[2] = __float_as_int(ok ? w0 : 0.f);
    dst[3] = __float_as_int(ok ? w1 : 0.f);
  }
  __syncthreads();
  const int c4 = threadIdx.x & 3, pt = threadIdx.x >> 2;
  const int ly = pt / UG_TX, lx = pt % UG_TX;
  const int py = Y0 + ly, px = X0 + lx;
  if (py >= OH || px >= OW) return;
  // LDS vector index of (row a of tap row dy) / (column b of tap column dx); weights of taps outside the image are zero
  int ro[3][2], co[3][2];
  float wy[3][2], wx[3][2];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int4 tr = *reinterpret_cast<const int4 *>(s_trow[d * UG_TY + ly]), tc = *reinterpret_cast<const int4 *>(s_tcol[d * UG_TX + lx]);
    ro[d][0] = tr.x; ro[d][1] = tr.y;
    co[d][0] = tc.x + c4; co[d][1] = tc.y + c4;
    wy[d][0] = __int_as_float(tr.z); wy[d][1] = __int_as_float(tr.w);
    wx[d][0] = __int_as_float(tc.z); wx[d][1] = __int_as_float(tc.w);
  }
  const float slope = prelu[0];
  f32x4 acc = *reinterpret_cast<const f32x4 *>(bias + c0 + c4 * 4);
  const f32x4 *sv = reinterpret_cast<const f32x4 *>(s_y);
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int tap = dy * 3 + dx;
      const f32x4 v00 = sv[ro[dy][0] + co[dx][0] + tap * 4], v01 = sv[ro[dy][0] + co[dx][1] + tap * 4];
      const f32x4 v10 = sv[ro[dy][1] + co[dx][0] + tap * 4], v11 = sv[ro[dy][1] + co[dx][1] + tap * 4];
      // the four corner weights once per tap, then 4 fused multiply-adds per channel (packed pairs on gfx950)
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const f32x2 wxp = {wx[dx][0], wx[dx][1]};
      const f32x2 w0 = f32x2{wy[dy][0], wy[dy][0]} * wxp, w1 = f32x2{wy[dy][1], wy[dy][1]} * wxp;      // (w00, w01), (w10, w11)
      acc = __builtin_elementwise_fma(v00, f32x4{w0.x, w0.x, w0.x, w0.x}, acc);
      acc = __builtin_elementwise_fma(v01, f32x4{w0.y, w0.y, w0.y, w0.y}, acc);
      acc = __builtin_elementwise_fma(v10, f32x4{w1.x, w1.x, w1.x, w1.x}, acc);
      acc = __builtin_elementwise_fma(v11, f32x4{w1.y, w1.y, w1.y, w1.y}, acc);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = acc[e] > 0.f ? acc[e] : acc[e] * slope;
  *reinterpret_cast<f32x4 *>(out + ((size_t)(b * OH + py) * OW + px) * Cout + c0 + c4 * 4) = acc;
}

// Tail of the colour branch, only at the N chosen pixels (lib/network.py:98-102).  Only those pixels of up_3's output
// (lib/pspnet.py:75: bilinear x2, conv3x3 64->64, PReLU) are ever read, so the conv is evaluated there alone:
// this kernel builds, per chosen pixel, the 3x3 patch of the UPSAMPLED input (each of the 9 taps bilinearly
// interpolated from the half-resolution map x [B][h][w][64], zero outside the full-resolution image = the conv's
// padding) as one GEMM row [tap][channel] = 576 floats; the conv itself is then a [B*Npad x 576] x [576 x 64] GEMM
// with bias + PReLU fused (engine), a fixed 74 MFLOP per object instead of 0.15 MFLOP per crop pixel.
// thread = (point, tap, 4 channels); rows n >= N of every object's Npad block are written as zeros.
__global__ __launch_bounds__(TPB) void up3_patch_kernel(const float *__restrict__ x, const int64_t *__restrict__ choose,
                                                        float *__restrict__ patch, int B, int h, int wd, int N, int Npad) {
  const int OH = 2 * h, OW = 2 * wd, HW = OH * OW;
  const float sh = OH > 1 ? (float)(h - 1) / (float)(OH - 1) : 0.f;
  const float sw = OW > 1 ? (float)(wd - 1) / (float)(OW - 1) : 0.f;
  const long total = (long)B * Npad * 9 * 16;
  for (long i = blockIdx.x * (long)TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const int c = (int)(i & 15) * 4;
    long r = i >> 4;
    const int tap = (int)(r % 9);
    r /= 9;                                            // row = b * Npad + n
    const int n = (int)(r % Npad), b = (int)(r / Npad);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
      long pix = choose[(size_t)b * N + n];
      pix = pix < 0 ? 0 : (pix >= HW ? HW - 1 : pix);     // torch.gather would raise; clamp keeps the launch safe
      const int py = (int)(pix / OW) + tap / 3 - 1, px = (int)(pix % OW) + tap % 3 - 1;
      if ((unsigned)py < (unsigned)OH && (unsigned)px < (unsigned)OW) {
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        src_ac(py, sh, h, y0, y1, wy0, wy1);
        src_ac(px, sw, wd, x0, x1, wx0, wx1);
        const float *xb = x + (size_t)b * h * wd * 64 + c;
        const f32x4 v00 = *reinterpret_cast<const f32x4 *>(xb + (size_t)(y0 * wd + x0) * 64);
        const f32x4 v01 = *reinterpret_cast<const f32x4 *>(xb + (size_t)(y0 * wd + x1) * 64);
        const f32x4 v10 = *reinterpret_cast<const f32x4 *>(xb + (size_t)(y1 * wd + x0) * 64);
        const f32x4 v11 = *reinterpret_cast<const f32x4 *>(xb + (size_t)(y1 * wd + x1) * 64);
        v = lerp4(v00, v01, v10, v11, wy0, wy1, wx0, wx1);
      }
    }
    reinterpret_cast<f32x4 *>(patch)[i] = v;
  }
}

// final 1x1 conv 64->32 + LogSoftmax over the 32 channels (lib/pspnet.py:53-56, implicit dim=1) on the up_3 rows
// z [B*Npad][64].  A workgroup takes 64 consecutive points of one object per pass: each 32-lane half-wave owns one point
// (lane = output channel), 8 rounds; the point-major copy is stored straight away (128 B per point), the channel-major one
// through an LDS transpose so that its rows go out as 256-byte runs (it was 32 scattered 4-byte stores per point, and the
// 64 weights per thread were re-loaded for every 8 points: 200 us per step).
constexpr int LSM_PTS = 64;
__global__ __launch_bounds__(TPB) void final_lsm_kernel(const float *__restrict__ z, const float *__restrict__ w,
                                                        const float *__restrict__ bias, float *__restrict__ emb,
                                                        float *__restrict__ emb_pm, int B, int N, int Npad) {
  __shared__ float s_t[32][LSM_PTS + 1];
  __shared__ __attribute__((aligned(16))) float s_z[LSM_PTS][64];       // the pass's input rows, loaded once with full-width vectors
  const int tid = threadIdx.x;
  const int o = tid & 31, slot = tid >> 5;
  float wr[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) wr[c] = w[o * 64 + c];
  const float bo = bias[o];
  const int chunks = (N + LSM_PTS - 1) / LSM_PTS;
  for (long job = blockIdx.x; job < (long)B * chunks; job += gridDim.x) {
    const int b = (int)(job / chunks), n0 = (int)(job % chunks) * LSM_PTS;
    {
      const f32x4 *src = reinterpret_cast<const f32x4 *>(z + ((size_t)b * Npad + n0) * 64);      // rows n0 .. n0+63 < Npad (Npad % 64 == 0)
#pragma unroll
      for (int v = 0; v < LSM_PTS * 16 / TPB; ++v) reinterpret_cast<f32x4 *>(&s_z[0][0])[tid + v * TPB] = src[tid + v * TPB];
    }
    __syncthreads();
#pragma unroll 2
    for (int it = 0; it < LSM_PTS / 8; ++it) {
      const int n = n0 + it * 8 + slot;
      const bool valid = n < N;
      const f32x4 *zr = reinterpret_cast<const f32x4 *>(&s_z[it * 8 + slot][0]);                  // same address across the half-wave
      float acc = bo;
#pragma unroll
      for (int c4 = 0; c4 < 16; ++c4) {
        const f32x4 zv = zr[c4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_fmaf(zv[e], wr[c4 * 4 + e], acc);
      }
      float mx = acc;
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) { const float t = __shfl_xor(mx, d); mx = t > mx ? t : mx; }
      const float shf = acc - mx;
      float se = expf(shf);
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) se += __shfl_xor(se, d);
      const float r = shf - logf(se);
      if (valid) emb_pm[((size_t)b * Npad + n) * 32 + o] = r;
      s_t[o][it * 8 + slot] = r;
    }
    __syncthreads();
    const int pt = tid & (LSM_PTS - 1);
    if (n0 + pt < N) {
      for (int row = tid / LSM_PTS; row < 32; row += TPB / LSM_PTS) emb[((size_t)b * 32 + row) * N + n0 + pt] = s_t[row][pt];
    }
    __syncthreads();
  }
}

// conv weights [O][9][I] (OHWI) -> tap-major [9][O][I]
__global__ __launch_bounds__(TPB) void tapmajor_kernel(const float *__restrict__ src, float *__restrict__ dst, int O, int I) {
  const long total = (long)O * 9 * I;
  for (long i = blockIdx.x * (long)TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const int c = (int)(i % I);
    const long r = i / I;
    const int t = (int)(r % 9);
    const long o = r / 9;
    dst[((size_t)t * O + o) * I + c] = src[i];
  }
}

__global__ __launch_bounds__(TPB) void emb_to_pm_kernel(const float *__restrict__ emb, float *__restrict__ emb_pm, int B,
                                                        int N, int Npad) {
  const long total = (long)B * N * 32;
  for (long i = blockIdx.x * (long)TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const int o = (int)(i & 31);
    const long p = i >> 5;
    const long b = p / N, n = p - b * N;
    emb_pm[((size_t)b * Npad + n) * 32 + o] = emb[((size_t)b * 32 + o) * N + n];
  }
}

// thread = (point, group of 4 output channels)
__global__ __launch_bounds__(TPB) void cloud_conv1_kernel(const float *__restrict__ cloud, const float *__restrict__ rt,
                                                          const float *__restrict__ w, const float *__restrict__ bias,
                                                          float *__restrict__ out, int out_ld, int B, int N, int Npad) {
  const long total = (long)B * N * 16;
  for (long i = blockIdx.x * (long)TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const int g = (int)(i & 15);
    const long p = i >> 4;
    const long b = p / N, n = p - b * N;
    float x = cloud[p * 3 + 0], y = cloud[p * 3 + 1], z = cloud[p * 3 + 2];
    if (rt) {   // new = (p - T) . R   (1x3 times 3x3, tools/eval_ycb.py:211)
      const float *R = rt + b * 12, *T = R + 9;
      const float dx = x - T[0], dy = y - T[1], dz = z - T[2];
      x = dx * R[0] + dy * R[3] + dz * R[6];
      y = dx * R[1] + dy * R[4] + dz * R[7];
      z = dx * R[2] + dy * R[5] + dz * R[8];
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = g * 4 + e;
      float v = bias[c] + w[c * 3 + 0] * x + w[c * 3 + 1] * y + w[c * 3 + 2] * z;
      o[e] = v > 0.f ? v : 0.f;
    }
    *reinterpret_cast<f32x4 *>(out + ((size_t)b * Npad + n) * out_ld + g * 4) = o;
  }
}

__global__ __launch_bounds__(TPB) void colsum_finish_kernel(const float *__restrict__ partial, int rows_per_obj,
                                                            float *__restrict__ mean, int B, int C, int N) {
  const long total = (long)B * C;
  for (long i = blockIdx.x * (long)TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const long b = i / C, c = i - b * C;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < rows_per_obj; ++r) s += partial[((size_t)b * rows_per_obj + r) * C + c];      // (loads ahead, the additions in row order)
    mean[i] = s / (float)N;
  }
}

// y[r][g*nout + n] = act(w[g*nout + n] . x[r][g*x_gstride ...] + bias) for a handful of rows (one per object) against
// wide weight matrices (the global-feature half of head layer 1, the refiner's FC towers): as a GEMM this is M <= 64,
// i.e. a latency-bound chain of k-tiles on a few workgroups.  Here lane = row, a workgroup owns FC_COLS output columns
// and its four waves each take a quarter of K: the weight block is staged in LDS once (coalesced; LDS broadcast reads in
// the loop), the x vectors are fetched eight at a time per lane, the four partial sums meet in LDS and are added in a fixed
// order.
constexpr int FC_COLS = 8;

__global__ __launch_bounds__(256) void fc_rows_kernel(const float *__restrict__ x, int x_ld, int x_gstride, const float *__restrict__ w,
                                                      const float *__restrict__ bias, float *__restrict__ y, int y_ld, int rows, int K,
                                                      int nout, int groups, int relu) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];      // [FC_COLS][K] weight block, then 4 x FC_COLS x 64 partial sums
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col0 = blockIdx.x * FC_COLS;                // FC_COLS | nout: a workgroup never straddles tower groups
  const int g = col0 / nout;
  const int K4 = K >> 2, kq4 = K4 >> 2;                  // float4s per row, per wave quarter
  float *s_part = s_w + FC_COLS * K;
  // the workgroup's FC_COLS weight rows are contiguous in memory: one coalesced sweep into LDS
  for (int i = threadIdx.x; i < FC_COLS * K4; i += 256)
    reinterpret_cast<f32x4 *>(s_w)[i] = reinterpret_cast<const f32x4 *>(w + (size_t)col0 * K)[i];
  // blockIdx.y walks the 64-row chunks (gridDim.y workgroups share them): with one row per object and hundreds of objects per
  // call the chunks fill the chip instead of being looped over by nout / FC_COLS workgroups
  for (int r0 = blockIdx.y * 64; r0 < rows; r0 += gridDim.y * 64) {
    const int r = r0 + lane;
    const bool live = r < rows;
    const f32x4 *xr = reinterpret_cast<const f32x4 *>(x + (size_t)(live ? r : r0) * x_ld + g * x_gstride) + wave * kq4;
    const f32x4 *wq = reinterpret_cast<const f32x4 *>(s_w) + wave * kq4;
    float acc[FC_COLS];
#pragma unroll
    for (int c = 0; c < FC_COLS; ++c) acc[c] = 0.f;
    __syncthreads();                                     // weight block staged / previous chunk's partials consumed
    constexpr int PF = 8;                                // x vectors in flight per lane
    for (int k = 0; k < kq4; k += PF) {
      f32x4 xv[PF];
#pragma unroll
      for (int u = 0; u < PF; ++u) xv[u] = xr[k + u];    // kq4 % PF == 0 (K % 128 == 0)
#pragma unroll
      for (int u = 0; u < PF; ++u) {
#pragma unroll
        for (int c = 0; c < FC_COLS; ++c) {
          const f32x4 wv = wq[c * K4 + k + u];           // same address in every lane: LDS broadcast
          acc[c] += (wv[0] * xv[u][0] + wv[1] * xv[u][1]) + (wv[2] * xv[u][2] + wv[3] * xv[u][3]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < FC_COLS; ++c) s_part[(wave * FC_COLS + c) * 64 + lane] = acc[c];
    __syncthreads();
    for (int e = threadIdx.x; e < FC_COLS * 64; e += 256) {
      const int c = e >> 6, rr = e & 63;
      if (r0 + rr < rows) {
        float v = ((s_part[(0 * FC_COLS + c) * 64 + rr] + s_part[(1 * FC_COLS + c) * 64 + rr]) +
                   (s_part[(2 * FC_COLS + c) * 64 + rr] + s_part[(3 * FC_COLS + c) * 64 + rr])) + (bias ? bias[col0 + c] : 0.f);
        if (relu) v = v > 0.f ? v : 0.f;
        y[(size_t)(r0 + rr) * y_ld + col0 + c] = v;
      }
    }
  }
}

// thread = (point, output j): j 0-3 quaternion, 4-6 translation, 7 confidence (sigmoid)
__global__ __launch_bounds__(TPB) void head_final_kernel(const float *__restrict__ h3, const float *__restrict__ w_r,
                                                         const float *__restrict__ b_r, const float *__restrict__ w_t,
                                                         const float *__restrict__ b_t, const float *__restrict__ w_c,
                                                         const float *__restrict__ b_c, const int64_t *__restrict__ obj,
                                                         int num_obj, float *__restrict__ out_r,
                                                         float *__restrict__ out_t, float *__restrict__ out_c, int B,
                                                         int N, int Npad) {
  const long total = (long)B * N * 8;
  for (long i = blockIdx.x * (long)TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
    const int j = (int)(i & 7);
    const long p = i >> 3;
    const long b = p / N, n = p - b * N;
    long o = obj[b];
    o = o < 0 ? 0 : (o >= num_obj ? num_obj - 1 : o);
    const float *wrow;
    float bv;
    int slice;
    if (j < 4) { wrow = w_r + (o * 4 + j) * 128; bv = b_r[o * 4 + j]; slice = 0; }
    else if (j < 7) { wrow = w_t + (o * 3 + (j - 4)) * 128; bv = b_t[o * 3 + (j - 4)]; slice = 128; }
    else { wrow = w_c + o * 128; bv = b_c[o]; slice = 256; }
    const f32x4 *xv = reinterpret_cast<const f32x4 *>(h3 + ((size_t)b * Npad + n) * 384 + slice);
    const f32x4 *wv = reinterpret_cast<const f32x4 *>(wrow);
    float acc = 0.f;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const f32x4 a = xv[k], c = wv[k];
      acc += (a[0] * c[0] + a[1] * c[1]) + (a[2] * c[2] + a[3] * c[3]);
    }
    acc += bv;
    if (j < 4) out_r[p * 4 + j] = acc;
    else if (j < 7) out_t[p * 3 + (j - 4)] = acc;
    else out_c[p] = 1.f / (1.f + expf(-acc));
  }
}

}  // namespace

void launch_nchw3_to_nhwc4(const float *img, float *out, int B, int H, int W, hipStream_t st) {
  hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(blocks_for((long)B * H * W)), dim3(TPB), 0, st, img, out, B, H * W);
}
void launch_maxpool3s2(const float *in, float *out, int B, int H, int W, int C, int OH, int OW, hipStream_t st) {
  hipLaunchKernelGGL(maxpool3s2_kernel, dim3(blocks_for((long)B * OH * OW * (C / 4))), dim3(TPB), 0, st, in, out, B, H, W,
                     C / 4, OH, OW);
}
void launch_psp_pool(const float *in, int in_ld, int in_coff, float *out, int B, int H, int W, int C, hipStream_t st, int Btot, int b0) {
  hipLaunchKernelGGL(psp_pool_kernel, dim3(50 * ((B + 7) / 8) * 8), dim3(512), 0, st, in, in_ld, in_coff, out, B, H, W, C, Btot > 0 ? Btot : B, b0);
}
void launch_psp_prior_sum(const float *z, float *out, int B, int H, int W, int C, hipStream_t st) {
  const size_t lds = ((size_t)50 * PP_CH + (size_t)16 * (H + W)) * sizeof(float);          // C % 64 == 0 (1024 here); 12.8 KB + 64 (H + W) bytes <= 64 KB: H + W <= 824 at 1/8 resolution, guaranteed by DF_MAX_CROP (posenet_args_ok)
  hipLaunchKernelGGL(psp_prior_sum_kernel, dim3(B * (C / PP_CH)), dim3(TPB), lds, st, z, out, B, H, W, C);
}
void launch_up3_patches(const float *x, const int64_t *choose, float *patch, int B, int h, int wd, int N, int Npad, hipStream_t st) {
  hipLaunchKernelGGL(up3_patch_kernel, dim3(blocks_for((long)B * Npad * 9 * 16)), dim3(TPB), 0, st, x, choose, patch, B, h, wd, N, Npad);
}
void launch_final_logsoftmax(const float *z, const float *w, const float *bias, float *emb, float *emb_pm, int B, int N, int Npad,
                             hipStream_t st) {
  const long jobs = (long)B * ((N + LSM_PTS - 1) / LSM_PTS);
  hipLaunchKernelGGL(final_lsm_kernel, dim3((unsigned)(jobs < 2048 ? (jobs < 1 ? 1 : jobs) : 2048)), dim3(TPB), 0, st, z, w, bias, emb, emb_pm, B, N, Npad);
}
int launch_upconv_gather(const float *y, const float *bias, const float *prelu, float *out, int B, int h, int w, int Cout,
                         hipStream_t st) {
  // host-checked: the kernel's 24-bit offset multiplies, its 32-bit buffer offsets, the grid's z range
  const long tiles = (long)B * ((2 * w + UG_TX - 1) / UG_TX) * ((2 * h + UG_TY - 1) / UG_TY);
  const long nwg = (tiles + 7) / 8 * 8 * (Cout / UG_CC);
  if (Cout % UG_CC || nwg >= (1L << 31) || (long)w * 9 * Cout * 4 >= (1L << 24) || h >= (1 << 24) || (long)h * w * 9 * Cout * 4 >= (1L << 32))
    return set_error(DF_ERR_ARG, "upconv_gather: needs Cout %% 16 == 0, under 2^31 workgroups and a low-resolution image under 4 GB (got B %d, %d x %d, Cout %d)",
                     B, h, w, Cout);
  dim3 grid((unsigned)nwg, 1, 1);
  hipLaunchKernelGGL(upconv_gather_tiled_kernel, grid, dim3(512), 0, st, y, bias, prelu, out, B, h, w, Cout);
  return DF_OK;
}
void launch_tapmajor(const float *src, float *dst, int O, int I, hipStream_t st) {
  hipLaunchKernelGGL(tapmajor_kernel, dim3(blocks_for((long)O * 9 * I)), dim3(TPB), 0, st, src, dst, O, I);
}
void launch_emb_to_pm(const float *emb, float *emb_pm, int B, int N, int Npad, hipStream_t st) {
  hipLaunchKernelGGL(emb_to_pm_kernel, dim3(blocks_for((long)B * N * 32)), dim3(TPB), 0, st, emb, emb_pm, B, N, Npad);
}
void launch_cloud_conv1(const float *cloud, const float *rt, const float *w, const float *bias, float *out, int out_ld,
                        int B, int N, int Npad, hipStream_t st) {
  hipLaunchKernelGGL(cloud_conv1_kernel, dim3(blocks_for((long)B * N * 16)), dim3(TPB), 0, st, cloud, rt, w, bias, out,
                     out_ld, B, N, Npad);
}
void launch_colsum_finish(const float *partial, int rows_per_obj, float *mean, int B, int C, int N, hipStream_t st) {
  hipLaunchKernelGGL(colsum_finish_kernel, dim3(blocks_for((long)B * C)), dim3(TPB), 0, st, partial, rows_per_obj, mean,
                     B, C, N);
}
void launch_fc_rows(const float *x, int x_ld, int x_gstride, const float *w, const float *bias, float *y, int y_ld, int rows, int K,
                    int nout, int groups, int relu, hipStream_t st) {
  // K % 128 == 0 and nout % FC_COLS == 0 hold for every caller (K 512 / 1024, nout 128 .. 1920)
  const size_t lds = ((size_t)FC_COLS * K + 4 * FC_COLS * 64) * sizeof(float);
  const int chunks = (rows + 63) / 64;
  hipLaunchKernelGGL(fc_rows_kernel, dim3(nout * groups / FC_COLS, chunks < 8 ? chunks : 8), dim3(256), lds, st, x, x_ld, x_gstride, w, bias, y, y_ld, rows,
                     K, nout, groups, relu);
}
void launch_head_final(const float *h3, const float *w_r, const float *b_r, const float *w_t, const float *b_t,
                       const float *w_c, const float *b_c, const int64_t *obj, int num_obj, float *out_r, float *out_t,
                       float *out_c, int B, int N, int Npad, hipStream_t st) {
  hipLaunchKernelGGL(head_final_kernel, dim3(blocks_for((long)B * N * 8)), dim3(TPB), 0, st, h3, w_r, b_r, w_t, b_t, w_c,
                     b_c, obj, num_obj, out_r, out_t, out_c, B, N, Npad);
}

}  // namespace df

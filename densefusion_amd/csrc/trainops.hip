// Element-wise / resampling layers of the TRAINING graph, forward and backward (channels-last fp32).
// These are the glue layers between the MFMA GEMMs of lib/train_graph.py: activation gradients
// (lib/extractors.py:34-42, lib/pspnet.py:33), MaxPool2d(3,2,1) (:84), AdaptiveAvgPool2d (lib/pspnet.py:16),
// bilinear resize in both align modes (:22,:31), LogSoftmax over channels (:55), Dropout2d (:46,:52),
// the colour-feature gather (lib/network.py:100-102), AvgPool1d over points (:65) and sigmoid (:121).
// Backward passes are written as gathers where the adjoint is cheap to enumerate and as fp32 atomics otherwise.
#include "common.h"

namespace df {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TB = 256;
inline int nblk(long n) { long b = (n + TB - 1) / TB; return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }
#define GRID_STRIDE(i, n) for (long i = blockIdx.x * (long)TB + threadIdx.x; i < (n); i += (long)gridDim.x * TB)

// act: 1 = ReLU, 2 = PReLU (slope > 0 so sign(y) == sign(x)).  dx = dy * act'(x); PReLU also accumulates dslope.
__global__ __launch_bounds__(TB) void act_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y, float *__restrict__ dx,
                                                     long n, int act, const float *__restrict__ slope_p, float *__restrict__ partials) {
  __shared__ float s_red[TB];
  const float slope = act == 2 ? slope_p[0] : 0.f;
  float ds = 0.f;
  GRID_STRIDE(i, n) {
    const float yy = y[i], g = dy[i];
    if (yy > 0.f) dx[i] = g;
    else {
      dx[i] = g * slope;
      if (act == 2) ds += g * (yy / slope);          // x = y / slope on the negative side
    }
  }
  if (act == 2 && partials) {          // one partial per workgroup; act_bwd_finish_kernel adds them in index order (no float atomics)
    s_red[threadIdx.x] = ds;
    __syncthreads();
    for (int d = TB / 2; d >= 1; d >>= 1) { if (threadIdx.x < d) s_red[threadIdx.x] += s_red[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) partials[blockIdx.x] = s_red[0];
  }
}
__global__ __launch_bounds__(TB) void act_bwd_finish_kernel(const float *__restrict__ partials, int nb, float *__restrict__ dslope) {
  __shared__ float s_red[TB];
  float a = 0.f;
  for (int i = threadIdx.x; i < nb; i += TB) a += partials[i];
  s_red[threadIdx.x] = a;
  __syncthreads();
  for (int d = TB / 2; d >= 1; d >>= 1) { if (threadIdx.x < d) s_red[threadIdx.x] += s_red[threadIdx.x + d]; __syncthreads(); }
  if (threadIdx.x == 0) dslope[0] += s_red[0];
}

// MaxPool2d(3, stride 2, pad 1) backward as a gather: every input pixel asks the <= 4 windows that cover it
// whether it is their FIRST maximum (row-major scan with strict '>', the forward kernel's and ATen's rule)
__global__ __launch_bounds__(TB) void maxpool3s2_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dx,
                                                            int B, int H, int W, int C, int OH, int OW) {
  const long total = (long)B * H * W * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    long r = i / C;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H);
    const int b = (int)(r / H);
    const float xv = x[i];
    float acc = 0.f;
    for (int oy = (iy + 1) / 2 - 1 < 0 ? 0 : (iy + 1) / 2 - 1; oy <= (iy + 1) / 2 && oy < OH; ++oy) {
      if (iy < oy * 2 - 1 || iy > oy * 2 + 1) continue;
      for (int ox = (ix + 1) / 2 - 1 < 0 ? 0 : (ix + 1) / 2 - 1; ox <= (ix + 1) / 2 && ox < OW; ++ox) {
        if (ix < ox * 2 - 1 || ix > ox * 2 + 1) continue;
        // is (iy, ix) the first maximum of window (oy, ox)?
        bool win = true;
        for (int ky = 0; ky < 3 && win; ++ky) {
          const int yy = oy * 2 - 1 + ky;
          if ((unsigned)yy >= (unsigned)H) continue;
          for (int kx = 0; kx < 3; ++kx) {
            const int xx = ox * 2 - 1 + kx;
            if ((unsigned)xx >= (unsigned)W) continue;
            const float v = x[((long)(b * H + yy) * W + xx) * C + c];
            const bool earlier = yy < iy || (yy == iy && xx < ix);
            if (v > xv || (earlier && v == xv)) { win = false; break; }
          }
        }
        if (win) acc += dy[((long)(b * OH + oy) * OW + ox) * C + c];
      }
    }
    dx[i] = acc;
  }
}

// AdaptiveAvgPool2d(s): bin i covers [floor(i*H/s), ceil((i+1)*H/s))
__global__ __launch_bounds__(TB) void adaptive_pool_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int H, int W, int C, int s) {
  const long total = (long)B * s * s * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    long r = i / C;
    const int bj = (int)(r % s); r /= s;
    const int bi = (int)(r % s);
    const int b = (int)(r / s);
    const int y0 = (bi * H) / s, y1 = ((bi + 1) * H + s - 1) / s, x0 = (bj * W) / s, x1 = ((bj + 1) * W + s - 1) / s;
    float acc = 0.f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) acc += x[((long)(b * H + yy) * W + xx) * C + c];
    y[i] = acc / (float)((y1 - y0) * (x1 - x0));
  }
}
__global__ __launch_bounds__(TB) void adaptive_pool_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, int B, int H, int W, int C, int s) {
  const long total = (long)B * H * W * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    long r = i / C;
    const int xx = (int)(r % W); r /= W;
    const int yy = (int)(r % H);
    const int b = (int)(r / H);
    float acc = 0.f;
    for (int bi = 0; bi < s; ++bi) {
      const int y0 = (bi * H) / s, y1 = ((bi + 1) * H + s - 1) / s;
      if (yy < y0 || yy >= y1) continue;
      for (int bj = 0; bj < s; ++bj) {
        const int x0 = (bj * W) / s, x1 = ((bj + 1) * W + s - 1) / s;
        if (xx < x0 || xx >= x1) continue;
        acc += dy[((long)(b * s + bi) * s + bj) * C + c] / (float)((y1 - y0) * (x1 - x0));
      }
    }
    dx[i] = acc;
  }
}

// bilinear resize (ATen semantics, fp32): align != 0 -> src = dst*(in-1)/(out-1); else half-pixel, clamped at 0
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
__global__ __launch_bounds__(TB) void bilinear_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int H, int W, int C,
                                                          int OH, int OW, int align) {
  const long total = (long)B * OH * OW * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    long r = i / C;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const int b = (int)(r / OH);
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    bil_src(oy, H, OH, align, y0, y1, wy0, wy1);
    bil_src(ox, W, OW, align, x0, x1, wx0, wx1);
    const float *p = x + (long)b * H * W * C + c;
    y[i] = wy0 * (wx0 * p[((long)y0 * W + x0) * C] + wx1 * p[((long)y0 * W + x1) * C]) +
           wy1 * (wx0 * p[((long)y1 * W + x0) * C] + wx1 * p[((long)y1 * W + x1) * C]);
  }
}
// adjoint of the interpolation as a GATHER: every input element collects, in ascending (oy, ox) order, the output pixels whose two
// source rows / columns include it (a conservative index range from the scale, each candidate re-tested with bil_src): fixed order,
// no atomics -- the gradient is bit-reproducible
__device__ __forceinline__ void bil_range(int i, int in_size, int out_size, int &lo, int &hi) {
  const float inv = (float)out_size / (float)in_size;          // ~ output pixels per input pixel (either align mode, +-2 of slack below)
  lo = (int)floorf(((float)i - 1.f) * inv) - 2;
  hi = (int)ceilf(((float)i + 2.f) * inv) + 2;
  lo = lo < 0 ? 0 : lo;
  hi = hi > out_size - 1 ? out_size - 1 : hi;
}
__global__ __launch_bounds__(TB) void bilinear_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, int B, int H, int W, int C,
                                                          int OH, int OW, int align) {
  const long total = (long)B * H * W * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    long r = i / C;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    int oy_lo, oy_hi, ox_lo, ox_hi;
    bil_range(y, H, OH, oy_lo, oy_hi);
    bil_range(x, W, OW, ox_lo, ox_hi);
    const float *g = dy + (long)b * OH * OW * C + c;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float wy0, wy1;
      bil_src(oy, H, OH, align, y0, y1, wy0, wy1);
      const float wy = (y0 == y ? wy0 : 0.f) + (y1 == y ? wy1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1; float wx0, wx1;
        bil_src(ox, W, OW, align, x0, x1, wx0, wx1);
        const float wx = (x0 == x ? wx0 : 0.f) + (x1 == x ? wx1 : 0.f);
        if (wx != 0.f) acc += g[((long)oy * OW + ox) * C] * wy * wx;
      }
    }
    dx[i] = acc;
  }
}

// LogSoftmax over the last axis (C <= 64): one 32-lane group per row when C == 32, generic loop otherwise
__global__ __launch_bounds__(TB) void logsoftmax_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, long rows, int C) {
  GRID_STRIDE(r, rows) {
    const float *p = x + r * C;
    float mx = p[0];
    for (int c = 1; c < C; ++c) mx = p[c] > mx ? p[c] : mx;
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(p[c] - mx);
    const float lse = logf(se);
    for (int c = 0; c < C; ++c) y[r * C + c] = (p[c] - mx) - lse;
  }
}
__global__ __launch_bounds__(TB) void logsoftmax_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y, float *__restrict__ dx,
                                                            long rows, int C) {
  GRID_STRIDE(r, rows) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dy[r * C + c];
    for (int c = 0; c < C; ++c) dx[r * C + c] = dy[r * C + c] - expf(y[r * C + c]) * s;
  }
}

__device__ inline unsigned mix32(unsigned seed, unsigned i) {
  unsigned x = seed ^ (i * 0x9E3779B9u);
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// Dropout2d: one Bernoulli(1-p) per (sample, channel), survivors scaled by 1/(1-p)
__global__ __launch_bounds__(TB) void dropout_mask_kernel(float *__restrict__ scale, long n, unsigned seed, float p) {
  GRID_STRIDE(i, n) {
    const float u = (float)(mix32(seed, (unsigned)i) >> 8) * (1.f / 16777216.f);
    scale[i] = u < p ? 0.f : 1.f / (1.f - p);
  }
}
__global__ __launch_bounds__(TB) void channel_scale_kernel(const float *__restrict__ x, const float *__restrict__ scale, float *__restrict__ y,
                                                           long n, long hw, int C) {
  GRID_STRIDE(i, n) {
    const int c = (int)(i % C);
    const long b = i / (hw * C);
    y[i] = x[i] * scale[b * C + c];
  }
}

__global__ __launch_bounds__(TB) void gather_rows_kernel(const float *__restrict__ x, const int64_t *__restrict__ idx, float *__restrict__ y,
                                                         long n, int C, long rows) {
  GRID_STRIDE(i, n * C) {
    const long r = i / C;
    long j = idx[r];
    j = j < 0 ? 0 : (j >= rows ? rows - 1 : j);
    y[i] = x[j * C + (i - r * C)];
  }
}
// adjoint of the row gather without atomics: the FIRST source row that names a destination owns it and adds, in ascending source
// order, every source row with the same index (wrap-padded `choose` repeats pixels) -- O(n) index reads per thread from an 8 KB
// list, fixed order, bit-reproducible.  dx is zeroed by the caller.
__global__ __launch_bounds__(TB) void scatter_add_rows_kernel(const float *__restrict__ dy, const int64_t *__restrict__ idx,
                                                              float *__restrict__ dx, long n, int C, long rows) {
  GRID_STRIDE(i, n * C) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    long j = idx[r];
    j = j < 0 ? 0 : (j >= rows ? rows - 1 : j);
    bool owner = true;
    for (long q = 0; q < r; ++q) {
      long jq = idx[q];
      jq = jq < 0 ? 0 : (jq >= rows ? rows - 1 : jq);
      if (jq == j) { owner = false; break; }
    }
    if (!owner) continue;
    float acc = dy[i];
    for (long q = r + 1; q < n; ++q) {
      long jq = idx[q];
      jq = jq < 0 ? 0 : (jq >= rows ? rows - 1 : jq);
      if (jq == j) acc += dy[q * C + c];
    }
    dx[j * C + c] = acc;
  }
}

// mean over rows (AvgPool1d over the points) and its adjoint
// 32 channels per workgroup x 8 row lanes: lane l adds rows l, l + 8, ... in ascending order (four loads in flight), the 8
// partial sums meet in LDS and are added in lane order (fixed order: deterministic)
__global__ __launch_bounds__(256) void colmean_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, long rows, int C) {
  __shared__ float s_p[8][32];
  const int col = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + col;
  float s = 0.f;
  if (c < C) {
    long r = rl;
    for (; r + 24 < rows; r += 32) {
      const float v0 = x[r * C + c], v1 = x[(r + 8) * C + c], v2 = x[(r + 16) * C + c], v3 = x[(r + 24) * C + c];
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; r < rows; r += 8) s += x[r * C + c];
  }
  s_p[rl][col] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
#pragma unroll
    for (int l = 1; l < 8; ++l) s += s_p[l][col];
    y[c] = s / (float)rows;
  }
}
__global__ __launch_bounds__(TB) void colmean_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, long rows, int C) {
  GRID_STRIDE(i, rows * C) dx[i] = dy[i % C] / (float)rows;
}

__global__ __launch_bounds__(TB) void sigmoid_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, long n) {
  GRID_STRIDE(i, n) y[i] = 1.f / (1.f + expf(-x[i]));
}
__global__ __launch_bounds__(TB) void sigmoid_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y, float *__restrict__ dx, long n) {
  GRID_STRIDE(i, n) dx[i] = dy[i] * y[i] * (1.f - y[i]);
}

}  // namespace
}  // namespace df

using namespace df;
#define ST to_stream(stream)
#define NN(p) if (!(p)) return set_error(DF_ERR_ARG, "%s: null pointer", __func__)

extern "C" int df_act_bwd(const float *dy, const float *y, float *dx, int64_t n, int act, const float *slope, float *dslope, float *partials,
                          df_stream_t stream) {
  NN(dy); NN(y); NN(dx);
  if (act == 2 && !slope) return set_error(DF_ERR_ARG, "act_bwd: PReLU needs its slope");
  if (act == 2 && dslope && !partials) return set_error(DF_ERR_ARG, "act_bwd: the PReLU slope gradient needs DF_ACT_BWD_PARTIALS floats of scratch");
  if (n > 0) {
    const int nb = nblk(n);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(nb), dim3(TB), 0, ST, dy, y, dx, (long)n, act, slope, act == 2 && dslope ? partials : nullptr);
    if (act == 2 && dslope) hipLaunchKernelGGL(act_bwd_finish_kernel, dim3(1), dim3(TB), 0, ST, partials, nb, dslope);
  }
  return check_launch("act_bwd");
}
extern "C" int df_maxpool3s2_bwd(const float *x, const float *dy, float *dx, int B, int H, int W, int C, int OH, int OW, df_stream_t stream) {
  NN(x); NN(dy); NN(dx);
  hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(nblk((long)B * H * W * C)), dim3(TB), 0, ST, x, dy, dx, B, H, W, C, OH, OW);
  return check_launch("maxpool3s2_bwd");
}
extern "C" int df_adaptive_avgpool(const float *in, float *out, int B, int H, int W, int C, int s, int backward, df_stream_t stream) {
  NN(in); NN(out);
  if (!backward) hipLaunchKernelGGL(adaptive_pool_fwd_kernel, dim3(nblk((long)B * s * s * C)), dim3(TB), 0, ST, in, out, B, H, W, C, s);
  else hipLaunchKernelGGL(adaptive_pool_bwd_kernel, dim3(nblk((long)B * H * W * C)), dim3(TB), 0, ST, in, out, B, H, W, C, s);
  return check_launch("adaptive_avgpool");
}
extern "C" int df_bilinear(const float *in, float *out, int B, int H, int W, int C, int OH, int OW, int align_corners, int backward,
                           df_stream_t stream) {
  NN(in); NN(out);
  if (!backward) hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(nblk((long)B * OH * OW * C)), dim3(TB), 0, ST, in, out, B, H, W, C, OH, OW, align_corners);
  else hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(nblk((long)B * H * W * C)), dim3(TB), 0, ST, in, out, B, H, W, C, OH, OW, align_corners);
  return check_launch("bilinear");
}
extern "C" int df_logsoftmax(const float *a, const float *y, float *out, int64_t rows, int C, int backward, df_stream_t stream) {
  NN(a); NN(out);
  if (!backward) hipLaunchKernelGGL(logsoftmax_fwd_kernel, dim3(nblk(rows)), dim3(TB), 0, ST, a, out, (long)rows, C);
  else { NN(y); hipLaunchKernelGGL(logsoftmax_bwd_kernel, dim3(nblk(rows)), dim3(TB), 0, ST, a, y, out, (long)rows, C); }
  return check_launch("logsoftmax");
}
extern "C" int df_dropout2d_mask(float *scale, int64_t n, unsigned seed, float p, df_stream_t stream) {
  NN(scale);
  if (p < 0.f || p >= 1.f) return set_error(DF_ERR_ARG, "dropout2d_mask: p must be in [0, 1)");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(nblk(n)), dim3(TB), 0, ST, scale, (long)n, seed, p);
  return check_launch("dropout2d_mask");
}
extern "C" int df_channel_scale(const float *x, const float *scale, float *y, int B, int64_t hw, int C, df_stream_t stream) {
  NN(x); NN(scale); NN(y);
  hipLaunchKernelGGL(channel_scale_kernel, dim3(nblk((long)B * hw * C)), dim3(TB), 0, ST, x, scale, y, (long)B * hw * C, (long)hw, C);
  return check_launch("channel_scale");
}
extern "C" int df_gather_rows(const float *in, const int64_t *idx, float *out, int64_t n, int C, int64_t rows, int backward, df_stream_t stream) {
  NN(in); NN(idx); NN(out);
  if (!backward) hipLaunchKernelGGL(gather_rows_kernel, dim3(nblk(n * C)), dim3(TB), 0, ST, in, idx, out, (long)n, C, (long)rows);
  else {
    hipMemsetAsync(out, 0, (size_t)rows * C * sizeof(float), ST);
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(nblk(n * C)), dim3(TB), 0, ST, in, idx, out, (long)n, C, (long)rows);
  }
  return check_launch("gather_rows");
}
extern "C" int df_colmean(const float *in, float *out, int64_t rows, int C, int backward, df_stream_t stream) {
  NN(in); NN(out);
  if (!backward) hipLaunchKernelGGL(colmean_fwd_kernel, dim3((C + 31) / 32), dim3(256), 0, ST, in, out, (long)rows, C);
  else hipLaunchKernelGGL(colmean_bwd_kernel, dim3(nblk(rows * C)), dim3(TB), 0, ST, in, out, (long)rows, C);
  return check_launch("colmean");
}
extern "C" int df_sigmoid(const float *a, const float *y, float *out, int64_t n, int backward, df_stream_t stream) {
  NN(a); NN(out);
  if (!backward) hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3(nblk(n)), dim3(TB), 0, ST, a, out, (long)n);
  else { NN(y); hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(nblk(n)), dim3(TB), 0, ST, a, y, out, (long)n); }
  return check_launch("sigmoid");
}

namespace {
// MaxPool2d(2, stride 2, return_indices) / MaxUnpool2d(2, stride 2) of the SegNet encoder / decoder
// (vanilla_segmentation/segnet.py:78-116), channels-last.  The "index" is the position 0..3 inside the 2x2 window (row-major),
// first maximum wins (strict '>' scan, NaN taken like ATen's `val > max || isnan(val)`); one byte per element.
__global__ __launch_bounds__(TB) void maxpool2x2_idx_kernel(const float *__restrict__ x, float *__restrict__ y, unsigned char *__restrict__ idx,
                                                            int B, int H, int W, int C) {
  const int OH = H / 2, OW = W / 2;
  const long total = (long)B * OH * OW * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    long r = i / C;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const int b = (int)(r / OH);
    const float *p = x + (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + c;
    float best = p[0];
    int bi = 0;
    const float v1 = p[C], v2 = p[(size_t)W * C], v3 = p[(size_t)W * C + C];
    if (v1 > best || v1 != v1) { best = v1; bi = 1; }
    if (v2 > best || v2 != v2) { best = v2; bi = 2; }
    if (v3 > best || v3 != v3) { best = v3; bi = 3; }
    y[i] = best;
    idx[i] = (unsigned char)bi;
  }
}

__global__ __launch_bounds__(TB) void maxunpool2x2_kernel(const float *__restrict__ x, const unsigned char *__restrict__ idx,
                                                          float *__restrict__ y, int B, int H, int W, int C) {   // H, W: pooled size
  const long total = (long)B * H * W * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    long r = i / C;
    const int ox = (int)(r % W); r /= W;
    const int oy = (int)(r % H);
    const int b = (int)(r / H);
    const float v = x[i];
    const int k = idx[i];
    float *q = y + (((size_t)b * 2 * H + 2 * oy) * (2 * W) + 2 * ox) * C + c;
    q[0] = k == 0 ? v : 0.f;
    q[C] = k == 1 ? v : 0.f;
    q[(size_t)2 * W * C] = k == 2 ? v : 0.f;
    q[(size_t)2 * W * C + C] = k == 3 ? v : 0.f;
  }
}
}  // namespace

extern "C" int df_maxpool2x2_idx(const float *x, float *y, unsigned char *idx, int B, int H, int W, int C, df_stream_t stream) {
  NN(x); NN(y); NN(idx);
  if (B <= 0 || H < 2 || W < 2 || C <= 0 || (H & 1) || (W & 1)) return set_error(DF_ERR_ARG, "maxpool2x2_idx: need even H, W >= 2");
  hipLaunchKernelGGL(maxpool2x2_idx_kernel, dim3(nblk((long)B * (H / 2) * (W / 2) * C)), dim3(TB), 0, ST, x, y, idx, B, H, W, C);
  return check_launch("maxpool2x2_idx");
}
extern "C" int df_maxunpool2x2(const float *x, const unsigned char *idx, float *y, int B, int H, int W, int C, df_stream_t stream) {
  NN(x); NN(y); NN(idx);
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return set_error(DF_ERR_ARG, "maxunpool2x2: bad sizes");
  hipLaunchKernelGGL(maxunpool2x2_kernel, dim3(nblk((long)B * H * W * C)), dim3(TB), 0, ST, x, idx, y, B, H, W, C);
  return check_launch("maxunpool2x2");
}

#include "layers.h"
extern "C" int df_maxpool3s2_fwd(const float *x, float *y, int B, int H, int W, int C, int OH, int OW, df_stream_t stream) {
  NN(x); NN(y);
  if (C % 4) return set_error(DF_ERR_ARG, "maxpool3s2_fwd: C must be a multiple of 4");
  launch_maxpool3s2(x, y, B, H, W, C, OH, OW, ST);
  return check_launch("maxpool3s2_fwd");
}

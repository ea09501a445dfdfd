// Winograd F(2x2, 3x3) transforms around the batched fp32-MFMA GEMM, for the stride-1 3x3 convolutions of the
// dilated ResNet trunk (lib/extractors.py:29-43,107-110: layer3 / layer4, 256 and 512 channels, dilation 1 / 2 / 4).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A          per 4x4 input patch d -> 2x2 outputs, summed over input channels
//
// 16 multiplies per 4 outputs instead of 36: the element-wise product summed over channels is 16 independent GEMMs
// [tiles x Cin] x [Cin x Cout] (the igemm kernel's z batching).  A dilated convolution is the same thing on each of
// the dil^2 pixel sub-lattices (y = ry + dil*i, x = rx + dil*j): a patch takes its 4 rows / columns `dil` apart.
// Exact in real arithmetic; in fp32 it re-associates sums (measured ~2x the rounding error of the direct sum,
// tests/test_conv_gpu.py), weights are transformed once at load time in fp64.
//
// Tile numbering: t = ((b*dil + ry)*dil + rx)*TH*TW + ty*TW + tx,  TH = ceil(ceil(H/dil)/m) (same for all sub-lattices).
//
// F(4x4, 3x3) (m = 4): 6x6 input patches -> 4x4 outputs, 36 multiplies per 16 outputs (2.25 per output against 4 for F(2x2) and 9
// for the direct sum), V / M are 2.25x the activation instead of 4x.  Interpolation points {0, 1, -1, 1/2, -2, inf}: the mixed
// pair (1/2, -2) keeps the transform entries within [1/8, 8] and measures 6.5-8.7x the direct sum's rounding error per layer
// (the textbook points {0, +-1, +-2}: 10-14x; F(2x2): 1.6-2.2x); end to end the selected-pose ADD against an fp64 evaluation is
// 1-2e-7 m with either route (the error is not made in these layers).  The weights G g G^T are formed in fp64.
#include "wino.h"

namespace df {
namespace {

constexpr int WB = 256;

struct TileId { int b, ry, rx, ty, tx; };

__device__ __forceinline__ TileId tile_of(int t, int d, int TH, int TW) {
  TileId r;
  r.tx = t % TW; t /= TW;
  r.ty = t % TH; t /= TH;
  r.rx = t % d; t /= d;
  r.ry = t % d;
  r.b = t / d;
  return r;
}

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// U[z = i*4+j][o][c] = (G g G^T)[i][j],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]];  w packed [O][3][3][C]
__global__ __launch_bounds__(WB) void wino_weight_kernel(const float *__restrict__ w, float *__restrict__ U, int O, int C) {
  const long n = (long)O * C;
  for (long e = (long)blockIdx.x * WB + threadIdx.x; e < n; e += (long)gridDim.x * WB) {
    const int o = (int)(e / C), c = (int)(e % C);
    double g[3][3], t[4][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) g[i][j] = (double)w[((size_t)o * 9 + i * 3 + j) * C + c];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      t[0][j] = g[0][j];
      t[1][j] = 0.5 * (g[0][j] + g[1][j] + g[2][j]);
      t[2][j] = 0.5 * (g[0][j] - g[1][j] + g[2][j]);
      t[3][j] = g[2][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double u0 = t[i][0], u1 = 0.5 * (t[i][0] + t[i][1] + t[i][2]), u2 = 0.5 * (t[i][0] - t[i][1] + t[i][2]), u3 = t[i][2];
      U[((size_t)(i * 4 + 0) * O + o) * C + c] = (float)u0;
      U[((size_t)(i * 4 + 1) * O + o) * C + c] = (float)u1;
      U[((size_t)(i * 4 + 2) * O + o) * C + c] = (float)u2;
      U[((size_t)(i * 4 + 3) * O + o) * C + c] = (float)u3;
    }
  }
}

// V[z][t][c] = (B^T d B)[i][j],  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]];  one thread per (tile, 4 channels)
__global__ __launch_bounds__(WB) void wino_input_kernel(const float *__restrict__ x, int in_ld, int in_coff, float *__restrict__ V, int H,
                                                        int W, int C, int d, int TH, int TW, long T, long Ttot, long t0) {
  const int c4n = C >> 2;
  const long total = T * c4n;
  for (long e = (long)blockIdx.x * WB + threadIdx.x; e < total; e += (long)gridDim.x * WB) {
    const int c = (int)(e % c4n) << 2;
    const long t = e / c4n;
    const TileId id = tile_of((int)t, d, TH, TW);
    const int y0 = id.ry + d * (2 * id.ty - 1), x0 = id.rx + d * (2 * id.tx - 1);
    float4 p[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = y0 + d * i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int xx = x0 + d * j;
        const bool ok = (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
        p[i][j] = ok ? *reinterpret_cast<const float4 *>(x + ((size_t)(id.b * H + y) * W + xx) * in_ld + in_coff + c)
                     : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float4 r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r[0][j] = f4sub(p[0][j], p[2][j]);
      r[1][j] = f4add(p[1][j], p[2][j]);
      r[2][j] = f4sub(p[2][j], p[1][j]);
      r[3][j] = f4sub(p[1][j], p[3][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 *dst = reinterpret_cast<float4 *>(V + ((size_t)(i * 4) * Ttot + t0 + t) * C + c);
      const size_t zs = (size_t)Ttot * C / 4;       // float4 stride between z planes
      dst[0] = f4sub(r[i][0], r[i][2]);
      dst[zs] = f4add(r[i][1], r[i][2]);
      dst[2 * zs] = f4sub(r[i][2], r[i][1]);
      dst[3 * zs] = f4sub(r[i][1], r[i][3]);
    }
  }
}

// out = act( A^T M A + bias + res ),  A^T = [[1,1,1,0],[0,1,-1,-1]];  one thread per (tile, 4 output channels)
__global__ __launch_bounds__(WB) void wino_output_kernel(const float *__restrict__ Mz, float *__restrict__ out, int out_ld, int out_coff,
                                                         const float *__restrict__ bias, const float *__restrict__ res, int res_ld,
                                                         int res_coff, int act, int H, int W,
                                                         int C, int d, int TH, int TW, long T, long Ttot, long t0) {
  const int c4n = C >> 2;
  const long total = T * c4n;
  for (long e = (long)blockIdx.x * WB + threadIdx.x; e < total; e += (long)gridDim.x * WB) {
    const int c = (int)(e % c4n) << 2;
    const long t = e / c4n;
    const TileId id = tile_of((int)t, d, TH, TW);
    const int oy = id.ry + d * 2 * id.ty, ox = id.rx + d * 2 * id.tx;
    if (oy >= H || ox >= W) continue;                // padding tile of a short sub-lattice
    const size_t zs = (size_t)Ttot * C / 4;
    const float4 *src = reinterpret_cast<const float4 *>(Mz + (size_t)(t0 + t) * C + c);
    float4 s0[4], s1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 m0 = src[(0 * 4 + j) * zs], m1 = src[(1 * 4 + j) * zs], m2 = src[(2 * 4 + j) * zs], m3 = src[(3 * 4 + j) * zs];
      s0[j] = f4add(f4add(m0, m1), m2);
      s1[j] = f4sub(f4sub(m1, m2), m3);
    }
    float4 y[2][2];
    y[0][0] = f4add(f4add(s0[0], s0[1]), s0[2]);
    y[0][1] = f4sub(f4sub(s0[1], s0[2]), s0[3]);
    y[1][0] = f4add(f4add(s1[0], s1[1]), s1[2]);
    y[1][1] = f4sub(f4sub(s1[1], s1[2]), s1[3]);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int yy = oy + d * a;
      if (yy >= H) continue;
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const int xx = ox + d * b2;
        if (xx >= W) continue;
        const size_t pix = (size_t)(id.b * H + yy) * W + xx;
        float4 v = y[a][b2];
        if (bias) v = f4add(v, *reinterpret_cast<const float4 *>(bias + c));
        if (res) v = f4add(v, *reinterpret_cast<const float4 *>(res + pix * res_ld + res_coff + c));
        if (act == ACT_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        *reinterpret_cast<float4 *>(out + pix * out_ld + out_coff + c) = v;
      }
    }
  }
}

// ---- F(4x4, 3x3), points {0, 1, -1, 1/2, -2, inf} ----
//   A^T = [[1,1,1,1,1,0],[0,1,-1,1/2,-2,0],[0,1,1,1/4,4,0],[0,1,-1,1/8,-8,1]]
//   G   = [[1,0,0],[1/3,1/3,1/3],[-1/3,1/3,-1/3],[-16/15,-8/15,-4/15],[1/15,-2/15,4/15],[0,0,1]]
//   B^T = [[1,-3/2,-2,3/2,1,0],[0,-1,1/2,5/2,1,0],[0,1,-5/2,1/2,1,0],[0,-2,-1,2,1,0],[0,1/2,-1,-1/2,1,0],[0,1,-3/2,-2,3/2,1]]
__device__ __forceinline__ float4 f4s(float a, float4 v) { return make_float4(a * v.x, a * v.y, a * v.z, a * v.w); }

__device__ __forceinline__ void bt6(const float4 d0, const float4 d1, const float4 d2, const float4 d3, const float4 d4, const float4 d5,
                                    float4 *r, int stride) {
  const float4 a = f4add(d4, f4s(-2.f, d2)), b = f4s(1.5f, f4sub(d3, d1));          // shared by rows 0 / 5
  r[0 * stride] = f4add(f4add(d0, a), b);
  r[1 * stride] = f4add(f4sub(d4, d1), f4add(f4s(0.5f, d2), f4s(2.5f, d3)));
  r[2 * stride] = f4add(f4add(d4, d1), f4add(f4s(-2.5f, d2), f4s(0.5f, d3)));
  r[3 * stride] = f4add(f4sub(d4, d2), f4s(2.f, f4sub(d3, d1)));
  r[4 * stride] = f4add(f4sub(d4, d2), f4s(0.5f, f4sub(d1, d3)));
  r[5 * stride] = f4add(f4add(d5, f4add(d1, f4s(-2.f, d3))), f4s(1.5f, f4sub(d4, d2)));
  (void)a; (void)b;
}

__global__ __launch_bounds__(WB) void wino4_weight_kernel(const float *__restrict__ w, float *__restrict__ U, int O, int C) {
  const long n = (long)O * C;
  const double G[6][3] = {{1, 0, 0}, {1. / 3, 1. / 3, 1. / 3}, {-1. / 3, 1. / 3, -1. / 3}, {-16. / 15, -8. / 15, -4. / 15},
                          {1. / 15, -2. / 15, 4. / 15}, {0, 0, 1}};
  for (long e = (long)blockIdx.x * WB + threadIdx.x; e < n; e += (long)gridDim.x * WB) {
    const int o = (int)(e / C), c = (int)(e % C);
    double g[3][3], t[6][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) g[i][j] = (double)w[((size_t)o * 9 + i * 3 + j) * C + c];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0][j] + G[i][1] * g[1][j] + G[i][2] * g[2][j];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j)
        U[((size_t)(i * 6 + j) * O + o) * C + c] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
  }
}

// the same for a list of weight tensors in one launch (the trainer re-derives 22 of them after every optimizer step): segment g =
// O[g] x C[g] kernels at src + src_off[g] -> dst + dst_off[g]; a thread finds its segment by a scan of the element prefix
__global__ __launch_bounds__(WB) void wino4_weight_multi_kernel(const float *__restrict__ src_a, const float *__restrict__ src_b, float *__restrict__ dst,
                                                                const WinoWTab tab) {
  const double G[6][3] = {{1, 0, 0}, {1. / 3, 1. / 3, 1. / 3}, {-1. / 3, 1. / 3, -1. / 3}, {-16. / 15, -8. / 15, -4. / 15},
                          {1. / 15, -2. / 15, 4. / 15}, {0, 0, 1}};
  const long n = tab.e0[tab.n];
  for (long e0 = (long)blockIdx.x * WB + threadIdx.x; e0 < n; e0 += (long)gridDim.x * WB) {
    int sgi = 0;
    while (sgi + 1 < tab.n && e0 >= tab.e0[sgi + 1]) ++sgi;
    const long e = e0 - tab.e0[sgi];
    const int O = tab.O[sgi], C = tab.C[sgi];
    const float *w = (tab.from_b[sgi] ? src_b : src_a) + tab.src_off[sgi];
    float *U = dst + tab.dst_off[sgi];
    const int o = (int)(e / C), c = (int)(e % C);
    double g[3][3], t[6][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) g[i][j] = (double)w[((size_t)o * 9 + i * 3 + j) * C + c];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0][j] + G[i][1] * g[1][j] + G[i][2] * g[2][j];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j)
        U[((size_t)(i * 6 + j) * O + o) * C + c] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
  }
}

// V[z = i*6+j][t][c] = (B^T d B)[i][j]; one thread per (tile, 4 channels): 36 vector loads in flight per thread
__device__ __forceinline__ void wino4_input_body(const float *__restrict__ x, int in_ld, int in_coff, float *__restrict__ V, int H, int W, int C, int d,
                                                 int TH, int TW, long T, long Ttot, long t0, long e_begin, long e_step) {
  const int c4n = C >> 2;
  const long total = T * c4n;
  for (long e = e_begin; e < total; e += e_step) {
    const int c = (int)(e % c4n) << 2;
    const long t = e / c4n;
    const TileId id = tile_of((int)t, d, TH, TW);
    const int y0 = id.ry + d * (4 * id.ty - 1), x0 = id.rx + d * (4 * id.tx - 1);
    float4 r[6][6];                  // r[i][j] = (B^T d)[i][j]: column pass as the patch arrives
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int xx = x0 + d * j;
      float4 p[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int y = y0 + d * i;
        const bool ok = (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
        p[i] = ok ? *reinterpret_cast<const float4 *>(x + ((size_t)(id.b * H + y) * W + xx) * in_ld + in_coff + c)
                  : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      bt6(p[0], p[1], p[2], p[3], p[4], p[5], &r[0][j], 6);
    }
    const size_t zs = (size_t)Ttot * C / 4;       // float4 stride between z planes
    float4 *dst = reinterpret_cast<float4 *>(V + ((size_t)t0 + t) * C + c);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float4 v[6];
      bt6(r[i][0], r[i][1], r[i][2], r[i][3], r[i][4], r[i][5], v, 1);
#pragma unroll
      for (int j = 0; j < 6; ++j) dst[(size_t)(i * 6 + j) * zs] = v[j];
    }
  }
}

__device__ __forceinline__ void at6(const float4 m0, const float4 m1, const float4 m2, const float4 m3, const float4 m4, const float4 m5,
                                    float4 *y, int stride) {
  const float4 s12 = f4add(m1, m2), d12 = f4sub(m1, m2);
  y[0 * stride] = f4add(f4add(m0, s12), f4add(m3, m4));
  y[1 * stride] = f4add(d12, f4add(f4s(0.5f, m3), f4s(-2.f, m4)));
  y[2 * stride] = f4add(s12, f4add(f4s(0.25f, m3), f4s(4.f, m4)));
  y[3 * stride] = f4add(f4add(d12, m5), f4add(f4s(0.125f, m3), f4s(-8.f, m4)));
}

__global__ __launch_bounds__(WB) void wino4_input_kernel(const float *__restrict__ x, int in_ld, int in_coff, float *__restrict__ V, int H,
                                                         int W, int C, int d, int TH, int TW, long T, long Ttot, long t0) {
  wino4_input_body(x, in_ld, in_coff, V, H, W, C, d, TH, TW, T, Ttot, t0, (long)blockIdx.x * WB + threadIdx.x, (long)gridDim.x * WB);
}
// the same for up to WINO_MAXB crop-size buckets in ONE launch: a workgroup belongs to one bucket (blocks[k] .. blocks[k + 1]), found by
// a wave-uniform scan of the table in the kernel arguments
__global__ __launch_bounds__(WB) void wino4_input_multi_kernel(const float *__restrict__ x, int in_ld, float *__restrict__ V, int C, int d, long Ttot,
                                                               const WinoTab tab) {
  int k = 0;
  while (k + 1 < tab.n && (int)blockIdx.x >= tab.blocks[k + 1]) ++k;
  wino4_input_body(x + tab.row0[k] * in_ld, in_ld, 0, V, tab.H[k], tab.W[k], C, d, tab.TH[k], tab.TW[k], tab.T[k], Ttot, tab.t0[k],
                   (long)((int)blockIdx.x - tab.blocks[k]) * WB + threadIdx.x, (long)(tab.blocks[k + 1] - tab.blocks[k]) * WB);
}

// out = act( A^T M A + bias + res ); one thread per (tile, 4 output channels)
__device__ __forceinline__ void wino4_output_body(const float *__restrict__ Mz, float *__restrict__ out, int out_ld, int out_coff,
                                                  const float *__restrict__ bias, const float *__restrict__ res, int res_ld, int res_coff, int act, int H,
                                                  int W, int C, int d, int TH, int TW, long T, long Ttot, long t0, long e_begin, long e_step) {
  const int c4n = C >> 2;
  const long total = T * c4n;
  for (long e = e_begin; e < total; e += e_step) {
    const int c = (int)(e % c4n) << 2;
    const long t = e / c4n;
    const TileId id = tile_of((int)t, d, TH, TW);
    const int oy = id.ry + d * 4 * id.ty, ox = id.rx + d * 4 * id.tx;
    if (oy >= H || ox >= W) continue;                // padding tile of a short sub-lattice
    const size_t zs = (size_t)Ttot * C / 4;
    const float4 *src = reinterpret_cast<const float4 *>(Mz + ((size_t)t0 + t) * C + c);
    float4 s[4][6];                   // s = A^T M, one column of M at a time
#pragma unroll
    for (int j = 0; j < 6; ++j)
      at6(src[(size_t)(0 * 6 + j) * zs], src[(size_t)(1 * 6 + j) * zs], src[(size_t)(2 * 6 + j) * zs], src[(size_t)(3 * 6 + j) * zs],
          src[(size_t)(4 * 6 + j) * zs], src[(size_t)(5 * 6 + j) * zs], &s[0][j], 6);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = *reinterpret_cast<const float4 *>(bias + c);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = oy + d * a;
      if (yy >= H) continue;
      float4 y[4];
      at6(s[a][0], s[a][1], s[a][2], s[a][3], s[a][4], s[a][5], y, 1);
#pragma unroll
      for (int b2 = 0; b2 < 4; ++b2) {
        const int xx = ox + d * b2;
        if (xx >= W) continue;
        const size_t pix = (size_t)(id.b * H + yy) * W + xx;
        float4 v = y[b2];
        if (bias) v = f4add(v, bv);
        if (res) v = f4add(v, *reinterpret_cast<const float4 *>(res + pix * res_ld + res_coff + c));
        if (act == ACT_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        *reinterpret_cast<float4 *>(out + pix * out_ld + out_coff + c) = v;
      }
    }
  }
}

__global__ __launch_bounds__(WB) void wino4_output_kernel(const float *__restrict__ Mz, float *__restrict__ out, int out_ld, int out_coff,
                                                          const float *__restrict__ bias, const float *__restrict__ res, int res_ld,
                                                          int res_coff, int act, int H, int W, int C, int d, int TH, int TW, long T,
                                                          long Ttot, long t0) {
  wino4_output_body(Mz, out, out_ld, out_coff, bias, res, res_ld, res_coff, act, H, W, C, d, TH, TW, T, Ttot, t0,
                    (long)blockIdx.x * WB + threadIdx.x, (long)gridDim.x * WB);
}
__global__ __launch_bounds__(WB) void wino4_output_multi_kernel(const float *__restrict__ Mz, float *__restrict__ out, int out_ld, const float *__restrict__ res,
                                                                int res_ld, int act, int C, int d, long Ttot, const WinoTab tab) {
  int k = 0;
  while (k + 1 < tab.n && (int)blockIdx.x >= tab.blocks[k + 1]) ++k;
  wino4_output_body(Mz, out + tab.row0[k] * out_ld, out_ld, 0, nullptr, res ? res + tab.row0[k] * res_ld : nullptr, res_ld, 0, act, tab.H[k], tab.W[k], C,
                    d, tab.TH[k], tab.TW[k], tab.T[k], Ttot, tab.t0[k], (long)((int)blockIdx.x - tab.blocks[k]) * WB + threadIdx.x,
                    (long)(tab.blocks[k + 1] - tab.blocks[k]) * WB);
}

inline unsigned blocks_for(long n) {
  const long b = (n + WB - 1) / WB;
  return (unsigned)(b < 1 ? 1 : (b > 65535L * 16 ? 65535L * 16 : b));
}

}  // namespace

WinoGeom wino_geom(int B, int H, int W, int dil, int m) {
  WinoGeom g;
  g.TH = ((H + dil - 1) / dil + m - 1) / m;
  g.TW = ((W + dil - 1) / dil + m - 1) / m;
  g.T = (long)B * dil * dil * g.TH * g.TW;
  return g;
}

int wino_route(int H, int W, int dil, int Cin, int Cout) {
  // Estimated time per output pixel of the three routes, from the layer geometry only -- never the batch -- so batched and solo
  // calls take the same path: the matrix-core time of the multiplies each route performs (the transform-domain GEMMs have K = Cin,
  // shorter than the direct sum's 9 Cin: slower per FLOP) plus the HBM time of the transforms (read x, write V, read M, read the
  // residual, write y).  0: direct implicit GEMM, 2: F(2x2,3x3), 4: F(4x4,3x3).
  if (Cin < 128 || Cin % 4 || Cout % 4) return 0;
  const double px = (double)H * W, cc = (double)Cin * Cout;
  const double rate = Cin >= 512 ? 144e12 : Cin >= 256 ? 128e12 : 100e12;
  double best = 18.0 * cc / 145e12;
  int route = 0;
  for (int m = 2; m <= 4; m += 2) {
    const double n2 = (double)(m + 2) * (m + 2);
    const double per_px = n2 * (double)wino_geom(1, H, W, dil, m).T / px;          // transform-domain values per output pixel
    const double t = 2.0 * per_px * cc / rate + 4.0 * ((1.0 + per_px) * Cin + (per_px + 2.0) * Cout) / 4.0e12;
    if (t < 0.9 * best) { best = t / 0.9; route = m; }                           // a route has to win by a margin to replace a simpler one
  }
  return route;
}

void launch_wino_weight(const float *w_packed, float *U, int O, int C, hipStream_t st, int m) {
  if (m == 4) hipLaunchKernelGGL(wino4_weight_kernel, dim3(blocks_for((long)O * C)), dim3(WB), 0, st, w_packed, U, O, C);
  else hipLaunchKernelGGL(wino_weight_kernel, dim3(blocks_for((long)O * C)), dim3(WB), 0, st, w_packed, U, O, C);
}

void launch_wino4_weight_multi(const float *src_a, const float *src_b, float *dst, const WinoWTab &tab, hipStream_t st) {
  if (tab.n <= 0) return;
  hipLaunchKernelGGL(wino4_weight_multi_kernel, dim3(blocks_for(tab.e0[tab.n])), dim3(WB), 0, st, src_a, src_b, dst, tab);
}

void launch_wino_input(const float *x, int in_ld, int in_coff, float *V, int B, int H, int W, int C, int dil, hipStream_t st, long Ttot,
                       long t0, int m) {
  const WinoGeom g = wino_geom(B, H, W, dil, m);
  if (Ttot <= 0) { Ttot = g.T; t0 = 0; }
  if (m == 4)
    hipLaunchKernelGGL(wino4_input_kernel, dim3(blocks_for(g.T * (C / 4))), dim3(WB), 0, st, x, in_ld, in_coff, V, H, W, C, dil, g.TH, g.TW, g.T,
                       Ttot, t0);
  else
    hipLaunchKernelGGL(wino_input_kernel, dim3(blocks_for(g.T * (C / 4))), dim3(WB), 0, st, x, in_ld, in_coff, V, H, W, C, dil, g.TH, g.TW, g.T,
                       Ttot, t0);
}

void launch_wino_output(const float *M, float *out, int out_ld, int out_coff, const float *bias, const float *res, int res_ld, int res_coff,
                        int act, int B, int H, int W, int C, int dil, hipStream_t st, long Ttot, long t0, int m) {
  const WinoGeom g = wino_geom(B, H, W, dil, m);
  if (Ttot <= 0) { Ttot = g.T; t0 = 0; }
  if (m == 4)
    hipLaunchKernelGGL(wino4_output_kernel, dim3(blocks_for(g.T * (C / 4))), dim3(WB), 0, st, M, out, out_ld, out_coff, bias, res, res_ld,
                       res_coff, act, H, W, C, dil, g.TH, g.TW, g.T, Ttot, t0);
  else
    hipLaunchKernelGGL(wino_output_kernel, dim3(blocks_for(g.T * (C / 4))), dim3(WB), 0, st, M, out, out_ld, out_coff, bias, res, res_ld,
                       res_coff, act, H, W, C, dil, g.TH, g.TW, g.T, Ttot, t0);
}

// F(4x4,3x3) transforms of several crop-size buckets in one launch each (chunks of WINO_MAXB buckets): bucket k = B[k] maps of
// H[k] x W[k] whose pixel rows start at row0[k] of x / out / res, and whose tiles are rows t0[k] .. of the Ttot-row planes
static WinoTab make_tab(int n, const int *B, const int *H, const int *W, const long *row0, const long *t0, int C, int dil) {
  WinoTab tab;
  tab.n = n;
  tab.blocks[0] = 0;
  for (int k = 0; k < n; ++k) {
    const WinoGeom g = wino_geom(B[k], H[k], W[k], dil, 4);
    tab.H[k] = H[k]; tab.W[k] = W[k]; tab.TH[k] = g.TH; tab.TW[k] = g.TW; tab.T[k] = g.T; tab.row0[k] = row0[k]; tab.t0[k] = t0[k];
    tab.blocks[k + 1] = tab.blocks[k] + (int)blocks_for(g.T * (C / 4));
  }
  return tab;
}

void launch_wino4_input_multi(const float *x, int in_ld, float *V, int nb, const int *B, const int *H, const int *W, const long *row0, const long *t0,
                              int C, int dil, long Ttot, hipStream_t st) {
  for (int k0 = 0; k0 < nb; k0 += WINO_MAXB) {
    const int n = nb - k0 < WINO_MAXB ? nb - k0 : WINO_MAXB;
    const WinoTab tab = make_tab(n, B + k0, H + k0, W + k0, row0 + k0, t0 + k0, C, dil);
    hipLaunchKernelGGL(wino4_input_multi_kernel, dim3(tab.blocks[n]), dim3(WB), 0, st, x, in_ld, V, C, dil, Ttot, tab);
  }
}

void launch_wino4_output_multi(const float *M, float *out, int out_ld, const float *res, int res_ld, int act, int nb, const int *B, const int *H,
                               const int *W, const long *row0, const long *t0, int C, int dil, long Ttot, hipStream_t st) {
  for (int k0 = 0; k0 < nb; k0 += WINO_MAXB) {
    const int n = nb - k0 < WINO_MAXB ? nb - k0 : WINO_MAXB;
    const WinoTab tab = make_tab(n, B + k0, H + k0, W + k0, row0 + k0, t0 + k0, C, dil);
    hipLaunchKernelGGL(wino4_output_multi_kernel, dim3(tab.blocks[n]), dim3(WB), 0, st, M, out, out_ld, res, res_ld, act, C, dil, Ttot, tab);
  }
}

}  // namespace df

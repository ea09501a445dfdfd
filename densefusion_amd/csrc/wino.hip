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
// Tile numbering: t = ((b*dil + ry)*dil + rx)*TH*TW + ty*TW + tx,  TH = ceil(ceil(H/dil)/2) (same for all sub-lattices).
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

inline unsigned blocks_for(long n) {
  const long b = (n + WB - 1) / WB;
  return (unsigned)(b < 1 ? 1 : (b > 65535L * 16 ? 65535L * 16 : b));
}

}  // namespace

WinoGeom wino_geom(int B, int H, int W, int dil) {
  WinoGeom g;
  g.TH = ((H + dil - 1) / dil + 1) / 2;
  g.TW = ((W + dil - 1) / dil + 1) / 2;
  g.T = (long)B * dil * dil * g.TH * g.TW;
  return g;
}

bool wino_pays(int H, int W, int dil, int Cin, int Cout) {
  // per output map: 16 multiplies per tile against 9 per pixel, and the transforms move ~10 activation-sized tensors;
  // take it only when the multiply count drops by more than a quarter, on channel counts whose GEMM (K = Cin) is deep enough.
  // Depends on the layer geometry only -- never on the batch -- so batched and solo calls take the same path.
  if (Cin < 256 || Cin % 4 || Cout % 4) return false;
  const WinoGeom g = wino_geom(1, H, W, dil);
  return 16.0 * (double)g.T <= 0.72 * 9.0 * (double)H * W;
}

void launch_wino_weight(const float *w_packed, float *U, int O, int C, hipStream_t st) {
  hipLaunchKernelGGL(wino_weight_kernel, dim3(blocks_for((long)O * C)), dim3(WB), 0, st, w_packed, U, O, C);
}

void launch_wino_input(const float *x, int in_ld, int in_coff, float *V, int B, int H, int W, int C, int dil, hipStream_t st, long Ttot,
                       long t0) {
  const WinoGeom g = wino_geom(B, H, W, dil);
  if (Ttot <= 0) { Ttot = g.T; t0 = 0; }
  hipLaunchKernelGGL(wino_input_kernel, dim3(blocks_for(g.T * (C / 4))), dim3(WB), 0, st, x, in_ld, in_coff, V, H, W, C, dil, g.TH, g.TW, g.T,
                     Ttot, t0);
}

void launch_wino_output(const float *M, float *out, int out_ld, int out_coff, const float *bias, const float *res, int res_ld, int res_coff,
                        int act, int B, int H, int W, int C, int dil, hipStream_t st, long Ttot, long t0) {
  const WinoGeom g = wino_geom(B, H, W, dil);
  if (Ttot <= 0) { Ttot = g.T; t0 = 0; }
  hipLaunchKernelGGL(wino_output_kernel, dim3(blocks_for(g.T * (C / 4))), dim3(WB), 0, st, M, out, out_ld, out_coff, bias, res, res_ld,
                     res_coff, act, H, W, C, dil, g.TH, g.TW, g.T, Ttot, t0);
}

}  // namespace df

// Implicit-GEMM convolution / batched per-point GEMM on fp32 MFMA (gfx950).
#pragma once
#include "common.h"

namespace df {

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_PRELU = 2 };

// out[m][n] = act( sum_k A[m][k] * Wt[n][k] + bias[n] (+ res[m][n]) )
//   m = (b, oy, ox) over B*OH*OW output pixels (or points), n = output channel,
//   k = (ky, kx, c): A[m][k] = in[b][oy*stride - pad + ky*dil][ox*stride - pad + kx*dil][c]  (0 outside)
// Activations are channels-last (NHWC / point-major) fp32; weights are [Cout][KH*KW*Cin] fp32.
struct ConvParams {
  const float *in = nullptr;   // [B][H][W][in_ld], channels [in_coff, in_coff + Cin) are consumed
  const float *wgt = nullptr;  // [Cout][K]
  const float *bias = nullptr; // [Cout], or [groups][bias_group_ld] when bias_group_ld > 0, or null
  const float *res = nullptr;  // residual [M][res_ld] (channel offset res_coff) or null
  const float *prelu = nullptr;  // one shared slope (ACT_PRELU)
  float *out = nullptr;        // [M][out_ld], written at channel offset out_coff; may be null with colsum
  float *colsum = nullptr;     // optional [M/BM * WAVES_M][Cout] partial column sums of the activated output
  int B = 1, H = 1, W = 1, Cin = 0, in_ld = 0, in_coff = 0;
  int OH = 1, OW = 1, Cout = 0, out_ld = 0, out_coff = 0;
  int res_ld = 0, res_coff = 0;
  int KH = 1, KW = 1, stride = 1, pad = 0, dil = 1;
  // input dilation ("transposed conv" / dgrad of a strided conv): the input is read as if `up`-1 zeros sat between
  // its pixels: virtual coordinate v = o*stride - pad + k*dil is a real pixel v/up only when v % up == 0
  int up = 1;
  int act = ACT_NONE;
  // row groups (per-object point blocks): rows_per_group > 0 => row m belongs to group m / rows_per_group
  // and is a real point iff (m % rows_per_group) < rows_valid; used by bias_group_ld and colsum
  int rows_per_group = 0, rows_valid = 0, bias_group_ld = 0;
  // blockIdx.z "head" groups (the r/t/c towers): per-z element offsets
  int zcount = 1;
  int ngroup = 0;   // column tiles per L2-resident weight group (0 = one group); set by launch_conv
  long z_in_coff = 0, z_wgt = 0, z_bias = 0, z_out_coff = 0;
  // magic pairs for the kernels' divisions by OH*OW and OW (set by the launchers)
  // split-K (training path only: opt-in through a caller-provided scratch, df_conv_desc.splitk_ws): launches that would fill less than
  // half the chip cut their reduction into `splitk` ranges (blockIdx.z), partial sums go to the scratch and a fixed-order reduce
  // kernel adds them and applies bias / residual / activation (deterministic).  splitk is set by launch_conv.
  // development build only (csrc/split_gemm.hip): the weights do not change between split_gemm_invalidate() calls, so their bf16 planes
  // may be cached (set by the inference engine for its own packed parameters); false: the planes are cut again on every launch
  bool wgt_const = false;
  float *splitk_ws = nullptr;
  size_t splitk_ws_bytes = 0;
  int splitk = 1;
  // tile decode of the v4 kernel without integer divisions (set by launch_conv): row-tile count, tiles per full weight group, and
  // division magics (make_fdiv) for the group size, the widths of a full / the last group and rows_per_group
  int tiles_m = 0, tile_gn = 0, tile_full = 0;
  unsigned full_magic = 0, gn_magic = 0, gl_magic = 0, rpg_magic = 0;
  int full_sh = 0, gn_sh = 0, gl_sh = 0, rpg_sh = 0;
  unsigned ohw_magic = 0, ow_magic = 0;
  int ohw_sh = 0, ow_sh = 0;
};

// number of colsum partial rows a launch with these params writes (so callers can size the buffer)
int conv_colsum_rows(const ConvParams &p);
// FLOPs (2*MAC) of the launch, algorithmic (no padding)
double conv_flops(const ConvParams &p);
// algorithmic HBM bytes (inputs, weights, outputs and residual touched once)
double conv_bytes(const ConvParams &p);
int launch_conv(const ConvParams &p, hipStream_t st, int *splitk_used = nullptr);     // splitk_used: the K ranges the launch was cut into

// One crop-size bucket of a multi-bucket launch: B maps of H x W (outputs OH x OW) whose input / output pixel rows start at
// in_row0 / out_row0 of the concatenated buffers
struct WgradSeg { int B, H, W, OH, OW; long in_row0, out_row0; };
// The same convolution over SEVERAL crop-size buckets whose pixel rows are concatenated in p.in / p.out / p.res (p.B / H / W / OH / OW are
// ignored): ONE launch of the product kernel (a workgroup's tile lies inside one bucket), CONV_MAX_BUCKETS buckets per launch.  Same
// sums in the same order per output element as per-bucket launch_conv calls without split-K.  Shapes the multi-bucket instantiations
// do not cover (input dilation, grouped / column-sum launches) fall back to one launch per bucket.
constexpr int CONV_MAX_BUCKETS = 16;
int launch_conv_multi(const ConvParams &p, int nseg, const WgradSeg *segs, hipStream_t st);

// dW[n][(ky,kx,c)] += sum_m dY[m][n] * A[m][(ky,kx,c)]  (A = the im2col view of x of the forward conv `p`; p.out = dY)
// dw / db (optional: column sums of dY) are overwritten.  The pixel range is split over workgroups; the partial tiles go to `ws`
// (wgrad_workspace_bytes) and are added in a fixed order: bit-reproducible, no atomics.
size_t wgrad_workspace_bytes(const ConvParams &p);
// accumulate != 0: dw / db += this launch's gradient (accumulation over the frames of an optimizer step)
int launch_wgrad(const ConvParams &p, float *dw, float *db, void *ws, size_t ws_bytes, hipStream_t st, int accumulate = 0);

// The same gradient over SEVERAL crop-size buckets whose pixel rows are concatenated in p.in (input) and p.out (dY): bucket g =
// B maps of H x W (outputs OH x OW) whose input / output pixel rows start at in_row0 / out_row0.  One contraction over the pixels of
// all buckets (chunks of the pixel axis never straddle buckets), one fixed-order reduction: bit-reproducible.  p.B / H / W / OH / OW
// are ignored; everything else (channels, strides, kernel geometry) comes from p.  Up to WGRAD_MAX_SEGS buckets per launch (more:
// several launches, the later ones accumulating).
constexpr int WGRAD_MAX_SEGS = 32;
size_t wgrad_multi_workspace_bytes(const ConvParams &p, int nseg, const WgradSeg *segs);
int launch_wgrad_multi(const ConvParams &p, int nseg, const WgradSeg *segs, float *dw, float *db, void *ws, size_t ws_bytes, hipStream_t st,
                       int accumulate = 0);

#ifdef DF_DEV
// development build only (csrc/split_gemm.hip): DF_GEMM_SPLIT_BF16=1 routes eligible plain-GEMM launches to the bf16 x 6 experiment
bool try_split_gemm(const ConvParams &p, hipStream_t st);
void split_gemm_invalidate();      // cached weight planes are cut again at their next use (a parameter was loaded / a network destroyed)
#endif

}  // namespace df

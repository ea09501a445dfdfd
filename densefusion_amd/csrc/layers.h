// Memory-bound companion kernels of the DenseFusion forward (channels-last fp32, gfx950).
#pragma once
#include "common.h"

namespace df {

// img [B][3][H][W] -> out [B][H][W][4] (4th channel zero) : feeds the 7x7 stem as a Cin=4 implicit GEMM
void launch_nchw3_to_nhwc4(const float *img, float *out, int B, int H, int W, hipStream_t st);
// MaxPool2d(3, stride 2, pad 1) on NHWC (lib/extractors.py:84)
void launch_maxpool3s2(const float *in, float *out, int B, int H, int W, int C, int OH, int OW, hipStream_t st);
// AdaptiveAvgPool2d(s) for s in {1,2,3,6} (lib/pspnet.py:15-17); in = NHWC rows of width in_ld at channel
// offset in_coff; out = 4 stage blocks of B*36 rows each ([4][B*36][C]; stage s fills its first B*s*s rows).  Btot > 0: the stage blocks
// hold Btot objects and this call fills objects b0 .. b0 + B (several crop-size buckets pooled into one set of stage blocks)
void launch_psp_pool(const float *in, int in_ld, int in_coff, float *out, int B, int H, int W, int C, hipStream_t st, int Btot = 0, int b0 = 0);
// sum over the 4 stages of F.upsample(size=(H,W), bilinear, align_corners=False) (lib/pspnet.py:22) of the
// stage maps z ([4][B*36][C], already multiplied by the folded stage x bottleneck weights) -> out [B][H][W][C]
void launch_psp_prior_sum(const float *z, float *out, int B, int H, int W, int C, hipStream_t st);
// PSPUpsample (x2 bilinear align_corners=True -> conv3x3 pad 1 -> PReLU, lib/pspnet.py:27-37) from the nine
// low-resolution 1x1 products y [B][h][w][9*Cout] (tap-major): out [B][2h][2w][Cout].  See layers.hip.
int launch_upconv_gather(const float *y, const float *bias, const float *prelu, float *out, int B, int h, int w, int Cout,
                         hipStream_t st);
void launch_tapmajor(const float *src, float *dst, int O, int I, hipStream_t st);   // [O][9][I] -> [9][O][I]
// up_3 only at the N chosen pixels (lib/network.py:98-102): per chosen pixel the 3x3 patch of the bilinearly upsampled
// (align_corners) half-resolution map x [B][h][w][64], as GEMM rows patch [B*Npad][9*64] (zero rows for n >= N);
// choose [B][N] int64 indexes the (2h x 2w) map
void launch_up3_patches(const float *x, const int64_t *choose, float *patch, int B, int h, int wd, int N, int Npad, hipStream_t st);
// final 1x1 conv 64->32 + LogSoftmax over channels (lib/pspnet.py:53-56) on rows z [B*Npad][64] ->
// emb [B][32][N] (reference layout) and emb_pm [B][Npad][32] (point-major, for the MLPs)
void launch_final_logsoftmax(const float *z, const float *w, const float *bias, float *emb, float *emb_pm, int B, int N, int Npad,
                             hipStream_t st);
// emb [B][32][N] -> emb_pm [B][Npad][32]  (PoseRefineNet.forward called on its own)
void launch_emb_to_pm(const float *emb, float *emb_pm, int B, int N, int Npad, hipStream_t st);
// Conv1d(3,64,1)+ReLU on the cloud (lib/network.py:54,152); optional rigid pre-transform
// new = (p - T[b]) . R[b]  (tools/eval_ycb.py:211) with rt [B][12] = R row-major (9) then T (3)
void launch_cloud_conv1(const float *cloud, const float *rt, const float *w, const float *bias, float *out, int out_ld,
                        int B, int N, int Npad, hipStream_t st);
// The K = 3 / 32 / 64 head of the per-point MLPs as ONE launch (csrc/pointfeat.hip): x1 = relu(conv1(cloud)), x2 = relu(conv2(x1))
// and / or e1 = relu(e_conv1(emb)), e2 = relu(e_conv2(e1)) (lib/network.py:53-58,152-157), written once into the [rows][ld] point-feature
// rows at the given column offsets; the layer-pair intermediates stay in LDS (fp32 MFMA, weights staged in LDS in fragment order).
// cloud [B][N][3] (+ optional rigid pre-transform rt [B][12], tools/eval_ycb.py:211), emb_pm [B*Npad][32]; either branch may be off
// (null cloud / null emb).  Same sums in the same order as the layer-by-layer launches: bit-identical outputs.
struct PointFeatParams {
  const float *cloud = nullptr, *rt = nullptr, *emb = nullptr;
  float *pf = nullptr;
  int ld = 384, cx1 = 0, cx2 = 0, ce1 = 0, ce2 = 0;
  const float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr, *we1 = nullptr, *be1 = nullptr, *we2 = nullptr, *be2 = nullptr;
  int B = 0, N = 0, Npad = 0;
};
int launch_pointfeat(const PointFeatParams &p, hipStream_t st);
// mean over the points of each object from the GEMM's per-wave partial column sums
void launch_colsum_finish(const float *partial, int rows_per_obj, float *mean, int B, int C, int N, hipStream_t st);
// y[b][g*nout + n] = act(sum_k x[b][g*x_gstride + k] * W[g*nout + n][k] + bias[..]) for a handful of rows b (one per object);
// needs K % 128 == 0, nout % 8 == 0
void launch_fc_rows(const float *x, int x_ld, int x_gstride, const float *w, const float *bias, float *y, int y_ld, int rows, int K,
                    int nout, int groups, int relu, hipStream_t st);
// last head layer for the selected object only (lib/network.py:119-131): h3 [B][Npad][384] = r|t|c towers
void launch_head_final(const float *h3, const float *w_r, const float *b_r, const float *w_t, const float *b_t,
                       const float *w_c, const float *b_c, const int64_t *obj, int num_obj, float *out_r, float *out_t,
                       float *out_c, int B, int N, int Npad, hipStream_t st);

}  // namespace df

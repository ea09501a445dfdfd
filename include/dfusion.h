/* dfusion.h -- C ABI of libdfusion_hip.so, the MI355X (gfx950) implementation of the DenseFusion
 * 6D-pose hot path.  Plain pointers and sizes only; every pointer is a DEVICE pointer unless a
 * parameter says "host".  No function allocates persistent device memory behind the caller's back
 * during a forward call, none synchronises the device, and every launch goes on the stream that is
 * passed in (so a caller may capture a call sequence into a hipGraph).
 *
 * Status convention: 0 = ok, negative = error; df_last_error() returns a thread-local message.
 * (The reference's `int knn(...)` returns 1 on success and raises through THError otherwise,
 * lib/knn/src/knn_pytorch.c:41-47; the Python shim maps a non-zero status to RuntimeError.)
 */
#ifndef DFUSION_H_
#define DFUSION_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *df_stream_t; /* a hipStream_t; NULL = the default stream */

#define DF_OK 0
#define DF_ERR_ARG (-1)       /* bad argument (THArgCheck in the reference, knn_pytorch.c:11-16) */
#define DF_ERR_LAUNCH (-2)    /* hipGetLastError() != hipSuccess after a launch (knn_pytorch.c:41-45) */
#define DF_ERR_WORKSPACE (-3) /* caller-provided workspace too small */
#define DF_ERR_STATE (-4)     /* handle not ready (missing parameters) */

const char *df_last_error(void);
int df_version(void);

/* ------------------------------------------------------------------------------------------------
 * 1-NN / k-NN  (replaces lib/knn: `knn_device`, lib/knn/src/knn_cuda_kernel.h:14-16 + .cu:211-259)
 *
 * ref_dev   [dim][ref_nb]   fp32, point index fastest (the layout knn_pytorch.c:21-29 hands over)
 * query_dev [dim][query_nb] fp32
 * ind_dev   [k][query_nb]   int64, 1-BASED row index of the k nearest reference points, nearest first;
 *                           ties go to the lowest index (strict '<' in cuInsertionSort, .cu:128,152)
 * The reference's `dist_dev` scratch (ref_nb*query_nb floats, knn_pytorch.c:31) does not exist here:
 * distances never leave registers.  dim == 3, k == 1 (the only use on the path) takes the fused
 * kernel; any other dim / k <= DF_KNN_MAX_K takes a generic one.
 */
#define DF_KNN_MAX_K 32
int df_knn_device(const float *ref_dev, int ref_nb, const float *query_dev, int query_nb, int dim, int k,
                  int64_t *ind_dev, df_stream_t stream);

/* Batched form = `int knn(ref, query, idx)` of lib/knn/src/knn_pytorch.h:1-2 with the tensors spelled out:
 * ref [batch][dim][ref_nb], query [batch][dim][query_nb], idx [batch][k][query_nb].  One launch for
 * the whole batch (the reference loops batches on the host, knn_pytorch.c:33-36). */
int df_knn(const float *ref, const float *query, int64_t *idx, int batch, int dim, int ref_nb, int query_nb,
           int k, df_stream_t stream);

/* Exact-signature replacement of the reference's native symbol (lib/knn/src/knn_cuda_kernel.h:14-16): the call at
 * lib/knn/src/knn_pytorch.c:35 binds to it unchanged.  `dist_dev` is ignored (may be NULL); `stream` is a hipStream_t;
 * failures are reported through df_last_error() (the reference's wrapper polls the runtime's last error, knn_pytorch.c:41-45). */
void knn_device(float *ref_dev, int ref_width, float *query_dev, int query_width, int height, int k, float *dist_dev,
                long *ind_dev, df_stream_t stream);

/* Measurement helper: the shader clock (MHz) the device delivers under a full-chip vector-ALU load right now (a ~1 ms spin kernel
 * read against the constant 100 MHz counter).  Synchronises `stream`.  bench.py logs it next to the 1-NN roofline fractions. */
int df_shader_clock_mhz(double *mhz_out, df_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Network handles (replace the nn.Module objects of lib/network.py; the Python classes of
 * densefusion_amd/lib/network.py own one handle each and keep the reference's state_dict keys)
 *
 * A handle owns device copies of the parameters in kernel-friendly layouts (channels-last conv
 * weights, the r/t/c towers stacked, head layer 1 split into its per-point and global-feature parts).
 * Parameters are handed over one state-dict tensor at a time, in the REFERENCE's layout and under the
 * REFERENCE's key (e.g. "cnn.model.module.feats.layer3.0.conv1.weight", "feat.conv5.bias",
 * "conv4_r.weight"), so a checkpoint written by tools/train.py:172-176,211-217 loads unchanged.
 * Creation/loading may allocate and synchronise; forward calls never do.
 */
typedef struct df_net df_net;

df_net *df_posenet_create(int num_points, int num_obj);   /* PoseNet(num_points, num_obj), lib/network.py:70-93 */
df_net *df_refiner_create(int num_points, int num_obj);   /* PoseRefineNet(...),          lib/network.py:170-185 */
void df_net_destroy(df_net *net);
int df_net_num_params(const df_net *net);                  /* 77 tensors for PoseNet, 24 for the refiner */
int df_net_param_info(const df_net *net, int i, char *key_out, int key_cap, int64_t *shape4, int *ndim);
/* ptr: `numel` fp32 values in the reference layout (device or host pointer).  The copy, the re-layout and the derived copies
 * (Winograd-domain, tap-major) are ENQUEUED (null stream) without host synchronisation; the next forward call on this handle waits
 * for the batch once.  A DEVICE source must stay allocated until then (a host source is consumed before the call returns). */
int df_net_load_param(df_net *net, const char *key, const float *ptr, int64_t numel);

/* Batched PoseNet.forward (lib/network.py:95-132), eval mode (Dropout2d = identity).
 *   img    [B][3][H][W] fp32 NCHW (normalised crop, tools/eval_ycb.py:175-181)
 *   cloud  [B][N][3]    fp32        choose [B][N] int64 (row-major pixel index into the H x W crop)
 *   obj    [B]          int64
 *   out_r  [B][N][4] (un-normalised quaternion w,x,y,z)   out_t [B][N][3]   out_c [B][N] (sigmoid)
 *   emb    [B][32][N]  (log-softmax colour features at the chosen pixels; what the refiner consumes)
 * ws: caller-provided scratch of at least df_posenet_workspace_bytes(net, B, H, W) bytes. */
size_t df_posenet_workspace_bytes(const df_net *net, int B, int H, int W);
int df_posenet_forward(df_net *net, int B, int H, int W, const float *img, const float *cloud, const int64_t *choose,
                       const int64_t *obj, float *out_r, float *out_t, float *out_c, float *emb, void *ws,
                       size_t ws_bytes, df_stream_t stream);

/* The same forward for nb buckets of different crop sizes in one pass: bucket i holds B[i] objects of H[i] x W[i] (img[i]:
 * [B[i]][3][H[i]][W[i]]); cloud / choose / obj / outputs are concatenated in bucket order.  Bit-identical to per-bucket
 * df_posenet_forward calls.  (The frozen estimator of the refiner phase, tools/train.py:139-145, over a whole window.) */
size_t df_posenet_multi_workspace_bytes(const df_net *net, int nb, const int *B, const int *H, const int *W);
int df_posenet_forward_multi(df_net *net, int nb, const int *B, const int *H, const int *W, const float *const *img,
                             const float *cloud, const int64_t *choose, const int64_t *obj, float *out_r, float *out_t,
                             float *out_c, float *emb, void *ws, size_t ws_bytes, df_stream_t stream);

/* Batched PoseRefineNet.forward (lib/network.py:187-206): x [B][N][3], emb [B][32][N], obj [B]
 * -> out_r [B][4], out_t [B][3]. */
size_t df_refiner_workspace_bytes(const df_net *net, int B);
int df_refiner_forward(df_net *net, int B, const float *x, const float *emb, const int64_t *obj, float *out_r,
                       float *out_t, void *ws, size_t ws_bytes, df_stream_t stream);

/* The inner loop of tools/eval_ycb.py:192-229 / tools/eval_linemod.py:81-114 for B objects, with no
 * host round trip: PoseNet -> arg-max-confidence pose -> `iters` x (cloud into the pose frame ->
 * refiner -> compose).  pose_wo (optional) and pose: [B][7] fp64 = quaternion (w,x,y,z) then
 * translation, i.e. the rows the reference writes to its result .mat files (eval_ycb.py:203,231). */
size_t df_estimate_workspace_bytes(const df_net *posenet, const df_net *refiner, int B, int H, int W);
int df_estimate_poses(df_net *posenet, df_net *refiner, int B, int H, int W, const float *img, const float *cloud,
                      const int64_t *choose, const int64_t *obj, int iters, double *pose_wo, double *pose, void *ws,
                      size_t ws_bytes, df_stream_t stream);

/* The same for a window of detections of DIFFERENT crop sizes (the objects tools/eval_ycb.py:147-237 meets over a run of frames,
 * bucketed by their snapped box size, eval_ycb.py:54-90): nb buckets, bucket i = B[i] objects of H[i] x W[i] with images
 * img[i] = [B[i]][3][H[i]][W[i]] (host arrays of nb entries; the pointers are device pointers).  cloud / choose / obj / pose_wo /
 * pose hold the objects of all buckets concatenated in bucket order ([sum B][N][3], [sum B][N], [sum B], [sum B][7]).
 * Everything that does not depend on the crop geometry (1x1 convolutions, Winograd-domain products, low-resolution up-conv
 * products, the whole per-point part and every refine iteration) is ONE launch over all buckets; results are bit-identical to
 * per-bucket df_estimate_poses calls. */
size_t df_estimate_multi_workspace_bytes(const df_net *posenet, const df_net *refiner, int nb, const int *B, const int *H,
                                         const int *W);
int df_estimate_poses_multi(df_net *posenet, df_net *refiner, int nb, const int *B, const int *H, const int *W,
                            const float *const *img, const float *cloud, const int64_t *choose, const int64_t *obj, int iters,
                            double *pose_wo, double *pose, void *ws, size_t ws_bytes, df_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * ADD / ADD-S loss and metric, forward (replace lib/loss.py:13-70, lib/loss_refiner.py:12-62 and the
 * metric of tools/eval_linemod.py:118-130).  One object per call, like the reference (bs = 1).
 * `symmetric` != 0 selects the nearest-neighbour (ADD-S) branch -- what `idx[0].item() in sym_list`
 * (and `not refine` for the PoseNet loss) decides in the reference; the transform, the 1-NN and the
 * distance reduction are fused, nothing of size N*M is written to memory.
 *
 * df_loss_forward:  pred_r [N][4], pred_t [N][3], pred_c [N], target [M][3], model_points [M][3],
 *   points [N][3], w  ->  loss_out[1] = mean_n(dis_n c_n - w log c_n), dis_out[1] = dis at arg-max c,
 *   new_points [N][3], new_target [M][3] (re-centred by the arg-max pose); dis_scratch: N floats.
 * df_loss_refine_forward: pred_r [4], pred_t [3], points [N][3] -> dis_out[1], new_points, new_target.
 * df_add_metric: B objects; pose [B][7] fp64 (q wxyz, t), model_points/target [B][M][3] fp32,
 *   symmetric [B] int32 or NULL -> dis_out [B] fp64 (ADD, or ADD-S pred->nearest target). */
int df_loss_forward(const float *pred_r, const float *pred_t, const float *pred_c, const float *target,
                    const float *model_points, const float *points, int N, int M, float w, int symmetric,
                    float *loss_out, float *dis_out, float *new_points, float *new_target, float *dis_scratch,
                    int *sel_out /* optional [N][M]: matched target index per transformed point, for backward */,
                    df_stream_t stream);
/* df_loss_forward for B stacked frames in a handful of launches (each frame's arithmetic is df_loss_forward's): pred_r [B][N][4],
 * pred_t [B][N][3], pred_c [B][N], target / model_points [B][M][3], points [B][N][3]; symmetric: HOST int [B] (or NULL = none)
 * -> loss_out [B], dis_out [B], new_points [B][N][3], new_target [B][M][3]; dis_scratch: B*N floats; sel_out: optional [B][N][M]
 * (needed when any frame is symmetric and the matches are wanted; NULL: the symmetric frames do not keep them).  What the trainer's
 * window passes call (tools/train.py:139-153 runs lib/loss.py once per frame). */
int df_loss_forward_frames(int B, const int *symmetric, const float *pred_r, const float *pred_t, const float *pred_c,
                           const float *target, const float *model_points, const float *points, int N, int M, float w,
                           float *loss_out, float *dis_out, float *new_points, float *new_target, float *dis_scratch,
                           int *sel_out, df_stream_t stream);
int df_loss_refine_forward(const float *pred_r, const float *pred_t, const float *target, const float *model_points,
                           const float *points, int N, int M, int symmetric, float *dis_out, float *new_points,
                           float *new_target, int *sel_out /* optional [M] */, df_stream_t stream);
/* Backward of the two losses (what autograd derives from lib/loss.py:16-50 / lib/loss_refiner.py:17-48): gradients
 * w.r.t. the un-normalised quaternions, the translations and the confidences; the nearest-neighbour match (sel, from
 * the forward call; NULL = identity) is a constant.  dis = the per-point distances the forward left in dis_scratch.
 * g_loss / g_dis = upstream gradient of the scalar output (1 for loss.backward()). */
int df_loss_backward(const float *pred_r, const float *pred_t, const float *pred_c, const float *target,
                     const float *model_points, const float *points, const int *sel, const float *dis, int N, int M,
                     float w, float g_loss, float *d_pred_r, float *d_pred_t, float *d_pred_c, df_stream_t stream);
int df_loss_refine_backward(const float *pred_r, const float *pred_t, const float *target, const float *model_points,
                            const int *sel, int M, float g_dis, float *d_pred_r, float *d_pred_t, df_stream_t stream);
int df_add_metric(const double *pose, const float *model_points, const float *target, const int *symmetric, int B,
                  int M, double *dis_out, df_stream_t stream);
/* YCB-Video toolbox distances (replace_ycb_toolbox/evaluate_poses_keyframe.m:160-193), fp64 like MATLAB:
 * rt_est / rt_gt [B][12] = 3x4 row-major [R|t], pts [B][M][3] fp64 model points ->
 * add_out [B] = mean ||est_m - gt_m||,  adi_out [B] = mean_m min_j ||est_j - gt_m|| (KDTreeSearcher direction). */
int df_ycb_distances(const double *rt_est, const double *rt_gt, const double *pts, int B, int M, double *add_out,
                     double *adi_out, df_stream_t stream);

/* Element-wise / resampling layers of the training graph (channels-last fp32), forward and backward; `backward != 0`
 * selects the adjoint (in = upstream gradient, out = input gradient).  See csrc/trainops.hip for the reference lines. */
#define DF_ACT_BWD_PARTIALS 16384   /* floats of scratch the PReLU slope gradient needs (one partial per workgroup, added in order) */
int df_act_bwd(const float *dy, const float *y, float *dx, int64_t n, int act /*1 ReLU, 2 PReLU*/, const float *slope,
               float *dslope /* PReLU: += */, float *partials /* PReLU with dslope: DF_ACT_BWD_PARTIALS floats; else may be NULL */,
               df_stream_t stream);
int df_maxpool3s2_fwd(const float *x, float *y, int B, int H, int W, int C, int OH, int OW, df_stream_t stream);
/* MaxPool2d(2, 2, return_indices) / MaxUnpool2d(2, 2) of the SegNet encoder / decoder (vanilla_segmentation/segnet.py:78-116),
 * channels-last [B][H][W][C]; idx holds the winner's position 0..3 inside its 2x2 window (first maximum wins); unpool takes
 * the POOLED size H x W and writes [B][2H][2W][C]. */
int df_maxpool2x2_idx(const float *x, float *y, unsigned char *idx, int B, int H, int W, int C, df_stream_t stream);
int df_maxunpool2x2(const float *x, const unsigned char *idx, float *y, int B, int H, int W, int C, df_stream_t stream);
int df_maxpool3s2_bwd(const float *x, const float *dy, float *dx, int B, int H, int W, int C, int OH, int OW, df_stream_t stream);
int df_adaptive_avgpool(const float *in, float *out, int B, int H, int W, int C, int s, int backward, df_stream_t stream);
int df_bilinear(const float *in, float *out, int B, int H, int W, int C, int OH, int OW, int align_corners, int backward,
                df_stream_t stream);
int df_logsoftmax(const float *x_or_dy, const float *y, float *out, int64_t rows, int C, int backward, df_stream_t stream);
int df_dropout2d_mask(float *scale, int64_t n /* B*C */, unsigned seed, float p, df_stream_t stream);
int df_channel_scale(const float *x, const float *scale /*[B][C]*/, float *y, int B, int64_t hw, int C, df_stream_t stream);
int df_gather_rows(const float *in, const int64_t *idx, float *out, int64_t n, int C, int64_t rows, int backward, df_stream_t stream);
int df_colmean(const float *in, float *out, int64_t rows, int C, int backward, df_stream_t stream);
int df_sigmoid(const float *x_or_dy, const float *y, float *out, int64_t n, int backward, df_stream_t stream);

/* Adam update of a flat fp32 parameter buffer (the optimizer of tools/train.py:99 with its default betas / eps):
 *   m = m + (1-b1)(g' - m);  v = b2 v + (1-b2) g'^2;  p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps),  g' = grad_scale * g
 * grad_scale carries the 1/(ranks x accumulated samples) of the data-parallel gradient average. */
int df_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                 float beta2, float eps, int step, float grad_scale, df_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Input preparation on the device (the numpy block of tools/eval_ycb.py:147-181, per detected object):
 * mask = (depth != 0) & (label == itemid) inside the snapped box; choose = num_points mask pixels (random
 * subset / wrap-padding); back-projected cloud; ImageNet-normalised CHW crop.  B objects of one crop size
 * (H x W) per call.
 *   rgb [F][IH][IW][3] u8, depth [F][IH][IW] u16, label [F][IH][IW] i32 (F frames resident on the device)
 *   obj_desc [B][8] int32 = {frame, itemid, rmin, rmax, cmin, cmax, seed, given}; rmax-rmin == H, cmax-cmin == W;
 *   given != 0: the object's row of choose_out is an INPUT -- the caller's chosen pixel indices (how a test hands over the very subset
 *   the reference's np.random.shuffle drew); sampling is skipped, cloud / img are formed from those indices
 *   scratch: B*H*W int32.  Outputs: img_out [B][3][H][W], cloud_out [B][N][3], choose_out [B][N] int64,
 *   count_out [B] = number of mask pixels (0 = detector lost the object, eval_ycb.py:234-237).
 * Subset rule when count > N (replaces np.random.shuffle, whose stream a GPU cannot share): keep the N mask
 * pixels with the smallest mix32(seed, flat index) keys, in index order -- a uniformly random ordered subset.
 * cloud = ((col - cx) * z / fx, (row - cy) * z / fy, z) / cloud_div with z = depth / cam_scale: YCB passes (10000, 1)
 * (eval_ycb.py:165-173), LineMOD (1, 1000) and itemid 255 (datasets/linemod/dataset.py:106-112,152-157). */
int df_preprocess_objects(const unsigned char *rgb, const unsigned short *depth, const int *label, int num_frames, int IH,
                          int IW, const int *obj_desc, int B, int H, int W, int num_points, float cam_cx, float cam_cy,
                          float cam_fx, float cam_fy, float cam_scale, float cloud_div, int *scratch, float *img_out, float *cloud_out,
                          int64_t *choose_out, int *count_out, df_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Building block, exposed for unit parity tests and for callers that want single layers: channels-last
 * convolution / per-point GEMM on the fp32 matrix cores with the fused epilogue.
 *   out[b][oy][ox][out_coff + n] = act( sum_{ky,kx,c} in[b][oy*stride - pad + ky*dil][ox*stride - pad + kx*dil][in_coff + c]
 *                                        * wgt[n][ky][kx][c] + bias[n] + res[b][oy][ox][res_coff + n] )
 * in rows are in_ld floats wide, out rows out_ld, res rows res_ld; Cin, in_ld, in_coff multiples of 4;
 * multi-tap kernels need a power-of-two Cin.  act: 0 none, 1 ReLU, 2 PReLU (one slope at *prelu).
 * bias / res / prelu may be NULL. */
typedef struct df_conv_desc {
  const float *in, *wgt, *bias, *res, *prelu;
  float *out;
  int32_t B, H, W, Cin, in_ld, in_coff;
  int32_t OH, OW, Cout, out_ld, out_coff;
  int32_t res_ld, res_coff;
  int32_t KH, KW, stride, pad, dil, act;
  /* optional split-K scratch (device, owned by the caller, NULL = never split): with it df_conv2d_nhwc and df_conv2d_dgrad_nhwc cut
   * the reduction of launches that would fill less than half the chip (a few hundred pixels x a few thousand taps*channels: the
   * layer3 / layer4 convolutions of a training pass) into up to 8 ranges, keep the partial sums there and add them in a fixed order
   * with bias / residual / activation (deterministic; results differ from the unsplit launch by fp32 re-association).  Per call:
   * concurrent calls on different streams bring their own scratch.  The engine entry points never split. */
  void *splitk_ws;
  size_t splitk_ws_bytes;
} df_conv_desc;
int df_conv2d_nhwc(const df_conv_desc *d, df_stream_t stream);
/* The same convolution over nb crop-size buckets in ONE launch (the training step's direct k x k convolutions on a window of mixed crop
 * sizes): bucket i = B[i] maps of H[i] x W[i] (host arrays); the buckets' pixel rows are concatenated in bucket order in d->in, d->out and
 * d->res; d->B / H / W / OH / OW are ignored.  A workgroup's output tile lies inside one bucket; per output element the same sums in the same
 * order as a df_conv2d_nhwc call on that bucket alone without split-K: bit-identical. */
int df_conv2d_nhwc_multi(const df_conv_desc *d, int nb, const int *B, const int *H, const int *W, df_stream_t stream);
/* number of K ranges the calling thread's last df_conv2d_nhwc / df_conv2d_dgrad_nhwc launch was cut into (1 = not split): tests */
int df_conv_last_splitk(void);

/* The same operator for 3x3 / stride 1 / pad == dil (any dilation) evaluated through the Winograd F(2x2,3x3) domain:
 * 16 multiplies per 2x2 outputs instead of 36 (the path the engine takes for the 256/512-channel convs of the
 * dilated ResNet trunk, lib/extractors.py:29-43,107-110).  Same descriptor; act none / ReLU; prelu unused.
 * scratch (device) of df_conv3x3_winograd_scratch_bytes(d) holds the transformed weights and both transformed
 * activations; results differ from df_conv2d_nhwc by fp32 re-association only. */
size_t df_conv3x3_winograd_scratch_bytes(const df_conv_desc *d);
int df_conv3x3_winograd_nhwc(const df_conv_desc *d, void *scratch, size_t scratch_bytes, df_stream_t stream);
/* The same with the output tile chosen: tile = 2 is the call above; tile = 4 is F(4x4,3x3) on the interpolation points
 * {0, 1, -1, 1/2, -2, inf}: 36 multiplies per 4x4 outputs (2.25 per output), transformed activations 2.25x the map instead of 4x;
 * rounding error ~4x that of tile 2 per layer (tests/test_conv_gpu.py), invisible in the selected pose.  The
 * engine picks direct / 2 / 4 per layer and map size from a cost estimate that depends on the layer geometry only. */
int df_wino_route(int H, int W, int dil, int Cin, int Cout);   /* the engine's choice for an H x W map: 0 direct, 2, 4 (host only) */
size_t df_conv3x3_winograd_tile_scratch_bytes(const df_conv_desc *d, int tile);
int df_conv3x3_winograd_tile_nhwc(const df_conv_desc *d, int tile, void *scratch, size_t scratch_bytes, df_stream_t stream);
/* Gradients of df_conv2d_nhwc (training path; `d` describes the FORWARD convolution, d->wgt = its weights):
 *   dgrad: dx[b][iy][ix][in_coff + c] (+)= sum dy[b][oy][ox][out_coff + n] * wgt[n][ky][kx][c] over the taps/outputs that
 *          read that input pixel; runs on the same MFMA kernel as the forward pass on flipped, transposed weights
 *          (w_scratch: Cout*KH*KW*Cin floats) with input dilation = the forward stride.  accumulate != 0 adds into dx.
 *   wgrad: dw[n][ky][kx][c] = sum_pixels dy[.][n] * x[. shifted by tap][c];  db[n] = sum_pixels dy[.][n] (db may be NULL).
 *          The pixel range is split over workgroups; their partial tiles go through `ws` (df_conv2d_wgrad_workspace_bytes)
 *          and are added in a fixed order, so gradients are bit-reproducible run to run (no atomics).
 * dy has the geometry of the forward output (out_ld / out_coff of `d`); activation masks are the caller's business. */
int df_conv2d_dgrad_nhwc(const df_conv_desc *d, const float *dy, float *dx, float *w_scratch, int accumulate,
                         df_stream_t stream);
size_t df_conv2d_wgrad_workspace_bytes(const df_conv_desc *d);
int df_conv2d_wgrad_nhwc(const df_conv_desc *d, const float *dy, float *dw, float *db, void *ws, size_t ws_bytes,
                         df_stream_t stream);
/* Weight gradient of df_conv2d_nhwc_multi: ONE contraction over the pixels of all buckets (bucket layout as there; dy rows concatenated like
 * d->out), dw / db overwritten; fixed-order reduction: bit-reproducible. */
size_t df_conv2d_wgrad_multi_workspace_bytes(const df_conv_desc *d, int nb, const int *B, const int *H, const int *W);
int df_conv2d_wgrad_nhwc_multi(const df_conv_desc *d, int nb, const int *B, const int *H, const int *W, const float *dy, float *dw, float *db,
                               void *ws, size_t ws_bytes, df_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Native training step (replaces the loop body of tools/train.py:146-163 of the reference: estimator / refiner forward,
 * criterion, loss.backward()) for B same-size frames, sequenced in C++ on one stream: no host synchronisation, no allocation,
 * every launch capturable in a hipGraph.  Parameters and gradients live in ONE flat fp32 buffer each, owned by the caller, in
 * KERNEL layout (conv O(HW)I, up_1 / up_2 tap-major, head layer 1 split, towers stacked); df_trainer_pack_param /
 * df_trainer_unpack_param convert a tensor from / to the reference's state-dict layout (same keys and shapes as
 * df_net_param_info), so checkpoints keep the reference's format.  Gradients are ACCUMULATED into flat_grad (loss.backward()
 * per frame, optimizer.step() every batch_size frames, tools/train.py:161-169) with fixed-order reductions only: two
 * identical steps give bit-identical gradient buffers.  param_version: any number that changes whenever flat_param's contents
 * change (the optimizer step count): the flipped weight copies of the data gradients are rebuilt only then; pass -1 to
 * rebuild them in every step (what a captured graph needs).
 *   kind 0: PoseNet + Loss (lib/network.py:95-132, lib/loss.py:13-70);  kind 1: PoseRefineNet + Loss_refine
 *   (lib/network.py:187-206, lib/loss_refiner.py:12-62).
 * df_posenet_train_step: img [B][3][H][W], cloud [B][N][3], choose [B][N] int64, obj [B] int64 (device), target /
 *   model_points [B][M][3], symmetric_host [B] (HOST ints or NULL: which frames take the nearest-neighbour branch),
 *   dropout != 0: Dropout2d of lib/pspnet.py:46,52 with masks hashed from `seed`;  outputs loss_out [B], dis_out [B],
 *   optional new_points [B][N][3], new_target [B][M][3], out_r [B][N][4], out_t [B][N][3], out_c [B][N], emb [B][32][N].
 * df_refiner_train_step: one refine iteration: points [B][N][3] (already in the current pose's frame), emb [B][32][N] ->
 *   dis_out [B], new_points, new_target for the next iteration. */
typedef struct df_trainer df_trainer;
df_trainer *df_trainer_create(int kind, int num_points, int num_obj);
void df_trainer_destroy(df_trainer *t);
int64_t df_trainer_flat_numel(const df_trainer *t);
int df_trainer_num_params(const df_trainer *t);
int df_trainer_param_info(const df_trainer *t, int i, char *key_out, int key_cap, int64_t *shape4, int *ndim);
int df_trainer_pack_param(const df_trainer *t, const char *key, const float *src, float *flat, df_stream_t stream);
int df_trainer_unpack_param(const df_trainer *t, const char *key, const float *flat, float *dst, df_stream_t stream);
size_t df_posenet_train_workspace_bytes(const df_trainer *t, int B, int H, int W, int M);
int df_posenet_train_step(df_trainer *t, const float *flat_param, float *flat_grad, int64_t param_version, int B, int H, int W,
                          const float *img, const float *cloud, const int64_t *choose, const int64_t *obj, const float *target,
                          const float *model_points, int M, const int *symmetric_host, float w, int dropout, unsigned seed,
                          float *loss_out, float *dis_out, float *new_points, float *new_target, float *out_r, float *out_t,
                          float *out_c, float *emb, void *ws, size_t ws_bytes, df_stream_t stream);
/* A window of frames of DIFFERENT crop sizes as one pass (what real data gives: the reference trains on whatever crop each frame has,
 * tools/train.py:131-176): nb crop-size buckets, bucket i = B[i] frames of H[i] x W[i] with their images at img[i] ([B_i][3][H_i][W_i],
 * device; the B / H / W / img arrays are HOST arrays); every other per-frame tensor (cloud, choose, obj, target, model_points, symmetric_host,
 * the outputs) is concatenated in bucket order, sum B frames.  Per-point layers, 1x1 convolutions, the F(4x4,3x3)-domain products and every
 * weight / bias gradient run ONCE over the rows of all buckets; direct k x k convolutions and the resampling kernels run per bucket.  The
 * gradient added to flat_grad equals the sum of the frames' one-per-pass gradients up to fp32 summation order (<= 1e-6 of each tensor's
 * scale; the order itself is fixed: bit-reproducible).  df_posenet_train_step is the one-bucket case. */
size_t df_posenet_train_multi_workspace_bytes(const df_trainer *t, int nb, const int *B, const int *H, const int *W, int M);
int df_posenet_train_step_multi(df_trainer *t, const float *flat_param, float *flat_grad, int64_t param_version, int nb, const int *B,
                                const int *H, const int *W, const float *const *img, const float *cloud, const int64_t *choose,
                                const int64_t *obj, const float *target, const float *model_points, int M, const int *symmetric_host,
                                float w, int dropout, unsigned seed, float *loss_out, float *dis_out, float *new_points,
                                float *new_target, float *out_r, float *out_t, float *out_c, float *emb, void *ws, size_t ws_bytes,
                                df_stream_t stream);
/* Per-launch timing of a trainer's MFMA launches (HIP events on the step's stream; measurement only).  df_trainer_profile(t, 1) arms
 * it; after the stream has been synchronised df_trainer_profile_read fills, per kind (0 forward, 1 data gradient, 2 weight gradient), the
 * summed launch durations in ms, the FLOPs the launches EXECUTE (2 M N K of the shapes really run, not the reference graph's) and the
 * launch counts since arming, and re-arms. */
/* Split-K of the small-grid forward / data-gradient launches of a step (default on: faster one-frame passes; results then depend on a
 * launch's grid size through fp32 re-association, and ReLU-gated gradients amplify that to ~1e-3 of a tensor's scale).  Off: every output
 * element is summed in one order whatever else shares the pass. */
int df_trainer_set_splitk(df_trainer *t, int enable);
int df_trainer_profile(df_trainer *t, int enable);
int df_trainer_profile_read(df_trainer *t, double *ms3, double *flops3, int *launches3);
size_t df_refiner_train_workspace_bytes(const df_trainer *t, int B, int M);
int df_refiner_train_step(df_trainer *t, const float *flat_param, float *flat_grad, int64_t param_version, int B, const float *points,
                          const float *emb, const int64_t *obj, const float *target, const float *model_points, int M,
                          const int *symmetric_host, float *dis_out, float *new_points, float *new_target, void *ws, size_t ws_bytes,
                          df_stream_t stream);

/* Per-launch timing of the GEMM kernel with HIP events on the call's stream (bench.py roofline).
 * df_net_profile(net, 1) arms it; after the stream has been synchronised df_net_profile_read returns the
 * summed duration (ms), the FLOPs the launches perform (2 M N K per launch, M = the rows launched), the part of them spent on
 * rows that are not padding (`gemm_useful_flops`: points beyond num_points in a 128-padded object block and Winograd tiles
 * beyond the map edge excluded), the algorithmic HBM bytes (each input, weight, output element once) and the number of GEMM
 * launches since arming, and re-arms. */
int df_net_profile(df_net *net, int enable);
int df_net_profile_read(df_net *net, double *gemm_ms, double *gemm_flops, double *gemm_useful_flops, double *gemm_bytes,
                        int *launches);

/* Debug taps: with df_net_debug_taps(net, 1) armed, a single-bucket PoseNet forward keeps device copies of its named
 * intermediates (channels-last): "stem" [B][H/2][W/2][64] (conv1 + ReLU, lib/extractors.py:115-117), "layer1".."layer4",
 * "psp" [B][H/8][W/8][1024] (lib/pspnet.py:20-24), "up_1" [B][H/4][W/4][256], "up_2" [B][H/2][W/2][64], "up_3" [B][Npad][64]
 * (rows of the chosen pixels only, Npad = num_points rounded up to 128), "ap_x" [B][1024] (lib/network.py:65).  It allocates, so
 * it is for debugging / layer-level parity tests only (not under hipGraph capture).  df_net_debug_tap_read copies a tap to dst
 * (host or device, `cap` floats) after the caller has synchronised the stream and returns its shape; dst == NULL only queries
 * the shape.  df_net_debug_taps(net, 0) frees the copies. */
int df_net_debug_taps(df_net *net, int enable);
int df_net_debug_tap_read(df_net *net, const char *name, float *dst, int64_t cap, int64_t *shape4);

#ifdef __cplusplus
}
#endif
#endif /* DFUSION_H_ */

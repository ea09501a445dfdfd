// The PoseNet loss (lib/loss.py:13-70) and its gradient for B stacked frames in a handful of launches (csrc/loss.hip): what df_loss_forward /
// df_loss_backward do per frame, frame by frame the same arithmetic, with blockIdx.y walking the frames.
#pragma once
#include "common.h"

namespace df {

// pred_r [B][N][4], pred_t [B][N][3], pred_c [B][N], target / model_points [B][M][3], points [B][N][3]; symmetric: HOST flags [B] (or null).
// -> loss_out [B], dis_out [B], new_points [B][N][3], new_target [B][M][3]; dis_scratch [B][N]; sel [B][N][M] (written for symmetric frames).
int launch_loss_frames(int B, const int *symmetric, const float *pred_r, const float *pred_t, const float *pred_c, const float *target,
                       const float *model_points, const float *points, int N, int M, float w, float *loss_out, float *dis_out, float *new_points,
                       float *new_target, float *dis_scratch, int *sel, hipStream_t st);
// gradient of sum_b g_loss * loss[b] w.r.t. the predictions: d_pred_r [B][N][4], d_pred_t [B][N][3], d_pred_c [B][N]
int launch_loss_bwd_frames(int B, const int *symmetric, const float *pred_r, const float *pred_t, const float *pred_c, const float *target,
                           const float *model_points, const float *points, const int *sel, const float *dis, int N, int M, float w, float g_loss,
                           float *d_pred_r, float *d_pred_t, float *d_pred_c, hipStream_t st);

// Loss_refine (lib/loss_refiner.py:12-62) and its gradient for B stacked frames, one pose each (pred_r [B][4], pred_t [B][3]; sel [B][M])
int launch_loss_refine_frames(int B, const int *symmetric, const float *pred_r, const float *pred_t, const float *target, const float *model_points,
                              const float *points, int N, int M, float *dis_out, float *new_points, float *new_target, int *sel, hipStream_t st);
int launch_loss_refine_bwd_frames(int B, const int *symmetric, const float *pred_r, const float *pred_t, const float *target, const float *model_points,
                                  const int *sel, int M, float g_dis, float *d_pred_r, float *d_pred_t, hipStream_t st);

}  // namespace df

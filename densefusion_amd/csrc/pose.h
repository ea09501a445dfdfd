// On-device pose selection / composition (fp64), see pose.hip.
#pragma once
#include "common.h"

namespace df {

// eval loop only: confidence (conv4_c + sigmoid on the confidence tower's features h3c [B*Npad][128]) -> arg-max point,
// then the r / t towers (head layers 1..4) at that one point, then the pose record: state[B][7] (fp64 quaternion + translation),
// optional copy pose_wo[B][7], the fp32 R|T record rt[B][12] for the first refine pass, optional which[B]
void launch_head_select(const float *h3c, const float *w_c, const float *b_c, const float *pf, const float *gbias, const float *w1,
                        const float *w2, const float *b2, const float *w3, const float *b3, const float *w_r, const float *b_r,
                        const float *w_t, const float *b_t, const int64_t *obj, int num_obj, const float *cloud, int B, int N, int Npad,
                        float *conf /*[B][N] scratch*/, double *pose_wo, double *state, float *rt, int *which, hipStream_t st);

// per object: refiner conv3_r/conv3_t rows of the selected object on f2 [B][256] (r|t towers), raw outputs to
// out_r[B][4] / out_t[B][3] when non-null; when state != null also compose into state and refresh rt
void launch_refiner_tail(const float *f2, const float *w_r, const float *b_r, const float *w_t, const float *b_t,
                         const int64_t *obj, int num_obj, float *out_r, float *out_t, double *state, float *rt,
                         double *pose_out, int B, hipStream_t st);

}  // namespace df

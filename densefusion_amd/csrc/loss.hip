// ADD / ADD-S loss and metric forward on gfx950.
//
// Reference: lib/loss.py:13-70 (PoseNet loss: N per-point poses x M model points), lib/loss_refiner.py:12-62
// (one pose), tools/eval_linemod.py:118-130 (ADD / ADD-S metric).  For symmetric objects the reference
// (as intended, lib/loss.py:9,41-47 with lib/knn) materialises all N*M transformed points, runs the
// 1-NN against the M target points and gathers the matches.  Here the transform, the 1-NN search and
// the distance reduction are ONE kernel: a workgroup owns one pose, every lane owns model points,
// the M target points sit in LDS (broadcast reads), and neither the [N,M,3] prediction nor the index
// tensor ever reaches HBM.  The nearest-neighbour choice uses the same arithmetic as csrc/knn.hip
// (fma chain in coordinate order, strict '<', lowest index wins).
#include <algorithm>

#include "common.h"
#include "knn_core.h"
#include "loss.h"

namespace df {
namespace {

constexpr int LB = 256;

// Several frames in ONE launch (the training step's loss: B frames of N per-point poses, lib/loss.py is called once per frame): blockIdx.y picks
// the frame idx[blockIdx.y]; every per-frame tensor starts at frame * its per-frame element count (frames are stacked, [B][N][..] / [B][M][3]).
// Single-frame entry points pass {1, N, M, {0}}.  Each frame's arithmetic is exactly the single-frame launch's.
constexpr int LOSS_MAX_FRAMES = 60;
struct FrameTab { int n, N, M; int idx[LOSS_MAX_FRAMES]; };
inline FrameTab one_frame(int N, int M) { FrameTab f{}; f.n = 1; f.N = N; f.M = M; f.idx[0] = 0; return f; }

struct Rot { float m[9]; };

// lib/loss.py:16-26: normalise, then the nine entries in the reference's expression order (fp32)
__device__ inline Rot quat_rot(const float *q4) {
  const float nrm = sqrtf(q4[0] * q4[0] + q4[1] * q4[1] + q4[2] * q4[2] + q4[3] * q4[3]);
  const float a = q4[0] / nrm, b = q4[1] / nrm, c = q4[2] / nrm, d = q4[3] / nrm;
  Rot r;
  r.m[0] = 1.0f - 2.0f * (c * c + d * d);
  r.m[1] = 2.0f * b * c - 2.0f * a * d;
  r.m[2] = 2.0f * a * c + 2.0f * b * d;
  r.m[3] = 2.0f * b * c + 2.0f * d * a;
  r.m[4] = 1.0f - 2.0f * (b * b + d * d);
  r.m[5] = -2.0f * a * b + 2.0f * c * d;
  r.m[6] = -2.0f * a * c + 2.0f * b * d;
  r.m[7] = 2.0f * a * b + 2.0f * c * d;
  r.m[8] = 1.0f - 2.0f * (b * b + c * c);
  return r;
}

__device__ inline float block_sum(float v, float *s_red) {
  const int tid = threadIdx.x;
  s_red[tid] = v;
  __syncthreads();
  for (int d = LB / 2; d >= 1; d >>= 1) {
    if (tid < d) s_red[tid] += s_red[tid + d];
    __syncthreads();
  }
  const float r = s_red[0];
  __syncthreads();
  return r;
}

// grid = P poses.  pose p: quaternion pred_r[p], translation pred_t[p] (+ points[p] when `points`).
// dis[p] = mean_m || R_p model[m] + t_p  -  target[m or nearest] ||       (lib/loss.py:38-49)
__global__ __launch_bounds__(LB) void add_dis_kernel(const float *__restrict__ pred_r, const float *__restrict__ pred_t,
                                                     const float *__restrict__ points, const float *__restrict__ target,
                                                     const float *__restrict__ model, int M, int symmetric,
                                                     float *__restrict__ dis, int *__restrict__ sel_out, const FrameTab ft) {
  extern __shared__ __attribute__((aligned(16))) float s_tgt[];   // [M][4]
  __shared__ float s_red[LB];
  const int p = blockIdx.x, tid = threadIdx.x;
  {
    const size_t f = ft.idx[blockIdx.y], fn = f * ft.N;
    pred_r += fn * 4; pred_t += fn * 3; target += f * ft.M * 3; model += f * ft.M * 3; dis += fn;
    if (points) points += fn * 3;
    if (sel_out) sel_out += fn * ft.M;
  }
  for (int m = tid; m < M; m += LB)
    reinterpret_cast<float4 *>(s_tgt)[m] = make_float4(target[m * 3], target[m * 3 + 1], target[m * 3 + 2], 0.f);
  const Rot R = quat_rot(pred_r + p * 4);
  float t0 = pred_t[p * 3], t1 = pred_t[p * 3 + 1], t2 = pred_t[p * 3 + 2];
  if (points) { t0 = points[p * 3] + t0; t1 = points[p * 3 + 1] + t1; t2 = points[p * 3 + 2] + t2; }
  __syncthreads();
  float acc = 0.f;
  for (int m = tid; m < M; m += LB) {
    const float x = model[m * 3], y = model[m * 3 + 1], z = model[m * 3 + 2];
    // bmm(model_points, base) with base = R^T (lib/loss.py:29,38), then + (points + pred_t)
    const float px = (x * R.m[0] + y * R.m[1] + z * R.m[2]) + t0;
    const float py = (x * R.m[3] + y * R.m[4] + z * R.m[5]) + t1;
    const float pz = (x * R.m[6] + y * R.m[7] + z * R.m[8]) + t2;
    int sel = m;
    if (symmetric) {
      float best = __builtin_inff();
      sel = 0;
#pragma unroll 4
      for (int r = 0; r < M; ++r) {
        const float4 q = reinterpret_cast<const float4 *>(s_tgt)[r];
        const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
        float d = dx * dx;
        d = __builtin_fmaf(dy, dy, d);
        d = __builtin_fmaf(dz, dz, d);
        const bool lt = d < best;
        best = lt ? d : best;
        sel = lt ? r : sel;
      }
    }
    if (sel_out) sel_out[(size_t)p * M + m] = sel;      // kept for the backward pass (the match is a constant there)
    const float4 q = reinterpret_cast<const float4 *>(s_tgt)[sel];
    const float ex = px - q.x, ey = py - q.y, ez = pz - q.z;
    acc += sqrtf(ex * ex + ey * ey + ez * ez);
  }
  const float tot = block_sum(acc, s_red);
  if (tid == 0) dis[p] = tot / (float)M;
}

// Symmetric objects, many poses (the PoseNet loss: N per-point poses, lib/loss.py:41-47): the N*M transformed model points
// are the queries of a 1-NN against the M target points.  Same scan as df_knn (knn_core.h knn1_scan_sc: packed v_pk_*_f32 pairs, two
// queries per lane, chunked arg-min, the targets through the SCALAR cache -- three s_load_dwordx8 per chunk of 8, one chunk ahead, no
// LDS staging, no barrier in the scan), so `sel` is bit for bit what KNearestNeighbor(1)(target, pred) would return; a workgroup owns
// `ppb` whole poses (their ppb*M queries in passes of 256*QPL) so that the per-pose distance sums stay inside the workgroup and
// deterministic.
constexpr int SYM_QPL = 2;      // queries per lane (the scalar-cache scan's best: knn.hip)
__global__ __launch_bounds__(LB) void add_dis_sym_kernel(const float *__restrict__ pred_r, const float *__restrict__ pred_t,
                                                         const float *__restrict__ points, const float *__restrict__ target,
                                                         const float *__restrict__ model, int P, int M, int ppb,
                                                         float *__restrict__ dis, int *__restrict__ sel_out, const FrameTab ft) {
  extern __shared__ __attribute__((aligned(16))) float s_e[];   // LB*SYM_QPL distances, then ppb totals
  __shared__ float s_red[LB];
  {
    const size_t f = ft.idx[blockIdx.y], fn = f * ft.N;
    pred_r += fn * 4; pred_t += fn * 3; target += f * ft.M * 3; model += f * ft.M * 3; dis += fn;
    if (points) points += fn * 3;
    if (sel_out) sel_out += fn * ft.M;
  }
  float *s_tot = s_e + LB * SYM_QPL;
  const int tid = threadIdx.x;
  const int p0 = blockIdx.x * ppb, np = min(ppb, P - p0);
  if (tid < ppb) s_tot[tid] = 0.f;
  __syncthreads();
  const int total = np * M;
  for (int base = 0; base < total; base += LB * SYM_QPL) {
    float qx[SYM_QPL], qy[SYM_QPL], qz[SYM_QPL];
    int bi[SYM_QPL], pi[SYM_QPL], mi[SYM_QPL];
#pragma unroll
    for (int j = 0; j < SYM_QPL; ++j) {
      const int i = base + tid + j * LB;
      const int ic = i < total ? i : total - 1;        // clamp: idle lanes work on a valid point and store nothing
      const int slot = ic / M, m = ic - slot * M, p = p0 + slot;
      pi[j] = p; mi[j] = m;
      const Rot R = quat_rot(pred_r + p * 4);
      float t0 = pred_t[p * 3], t1 = pred_t[p * 3 + 1], t2 = pred_t[p * 3 + 2];
      if (points) { t0 = points[p * 3] + t0; t1 = points[p * 3 + 1] + t1; t2 = points[p * 3 + 2] + t2; }
      const float x = model[m * 3], y = model[m * 3 + 1], z = model[m * 3 + 2];
      // bmm(model_points, base) with base = R^T (lib/loss.py:29,38), then + (points + pred_t): the same expression as add_dis_kernel
      qx[j] = (x * R.m[0] + y * R.m[1] + z * R.m[2]) + t0;
      qy[j] = (x * R.m[3] + y * R.m[4] + z * R.m[5]) + t1;
      qz[j] = (x * R.m[6] + y * R.m[7] + z * R.m[8]) + t2;
    }
    knn1_scan_sc<SYM_QPL>(target, M, qx, qy, qz, bi);      // the targets through the scalar cache ([M][3] as it comes: knn_core.h)
#pragma unroll
    for (int j = 0; j < SYM_QPL; ++j) {
      const int i = base + tid + j * LB;
      const bool ok = i < total;
      const float ex = qx[j] - target[bi[j] * 3], ey = qy[j] - target[bi[j] * 3 + 1], ez = qz[j] - target[bi[j] * 3 + 2];
      s_e[tid + j * LB] = ok ? sqrtf(ex * ex + ey * ey + ez * ez) : 0.f;
      if (ok && sel_out) sel_out[(size_t)pi[j] * M + mi[j]] = bi[j];      // kept for the backward pass
    }
    __syncthreads();
    // per-pose sums of this pass, in a fixed order
    const int s_lo = base / M, s_hi = (min(base + LB * SYM_QPL, total) - 1) / M;
    for (int sl = s_lo; sl <= s_hi; ++sl) {
      const int lo = max(sl * M, base) - base, hi = min(min((sl + 1) * M, base + LB * SYM_QPL), total) - base;
      float v = 0.f;
      for (int k = lo + tid; k < hi; k += LB) v += s_e[k];
      const float tot = block_sum(v, s_red);
      if (tid == 0) s_tot[sl] += tot;
    }
    __syncthreads();
  }
  if (tid < np) dis[p0 + tid] = s_tot[tid] / (float)M;
}

// Backward of dis_p = mean_m || R(q_p/|q_p|) x_m + t_p - tgt_sel(p,m) ||  w.r.t. q_p (un-normalised) and t_p,
// scaled by an upstream weight wgt[p] (PoseNet loss: g * c_p / N, lib/loss.py:49-50; refiner loss: g).
// The nearest-neighbour match enters as a constant (torch.index_select on a non-leaf index, lib/loss.py:46).
// grid = P poses; 12 running sums per thread (sum g, sum g x^T), block-reduced; thread 0 chains through
// the quaternion -> rotation map (lib/loss.py:18-26) and the normalisation (:16).
__global__ __launch_bounds__(LB) void add_dis_bwd_kernel(const float *__restrict__ pred_r, const float *__restrict__ pred_t,
                                                         const float *__restrict__ points, const float *__restrict__ target,
                                                         const float *__restrict__ model, const int *__restrict__ sel, int M,
                                                         const float *__restrict__ wgt, float wscale, float *__restrict__ d_r,
                                                         float *__restrict__ d_t, const FrameTab ft) {
  __shared__ float s_acc[12][LB];
  const int p = blockIdx.x, tid = threadIdx.x;
  {
    const size_t f = ft.idx[blockIdx.y], fn = f * ft.N;
    pred_r += fn * 4; pred_t += fn * 3; target += f * ft.M * 3; model += f * ft.M * 3; d_r += fn * 4; d_t += fn * 3;
    if (points) points += fn * 3;
    if (wgt) wgt += fn;
    if (sel) sel += fn * ft.M;
  }
  const Rot R = quat_rot(pred_r + p * 4);
  float t0 = pred_t[p * 3], t1 = pred_t[p * 3 + 1], t2 = pred_t[p * 3 + 2];
  if (points) { t0 = points[p * 3] + t0; t1 = points[p * 3 + 1] + t1; t2 = points[p * 3 + 2] + t2; }
  float a[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) a[e] = 0.f;
  for (int m = tid; m < M; m += LB) {
    const float x = model[m * 3], y = model[m * 3 + 1], z = model[m * 3 + 2];
    const float px = (x * R.m[0] + y * R.m[1] + z * R.m[2]) + t0;
    const float py = (x * R.m[3] + y * R.m[4] + z * R.m[5]) + t1;
    const float pz = (x * R.m[6] + y * R.m[7] + z * R.m[8]) + t2;
    const int j = sel ? sel[(size_t)p * M + m] : m;
    const float ex = px - target[j * 3], ey = py - target[j * 3 + 1], ez = pz - target[j * 3 + 2];
    const float nrm = sqrtf(ex * ex + ey * ey + ez * ez);
    const float inv = nrm > 0.f ? 1.f / nrm : 0.f;          // torch.norm's subgradient at 0 is 0
    const float gx = ex * inv, gy = ey * inv, gz = ez * inv;
    a[0] += gx; a[1] += gy; a[2] += gz;
    a[3] += gx * x; a[4] += gx * y; a[5] += gx * z;
    a[6] += gy * x; a[7] += gy * y; a[8] += gy * z;
    a[9] += gz * x; a[10] += gz * y; a[11] += gz * z;
  }
#pragma unroll
  for (int e = 0; e < 12; ++e) s_acc[e][tid] = a[e];
  __syncthreads();
  for (int d = LB / 2; d >= 1; d >>= 1) {
    if (tid < d)
#pragma unroll
      for (int e = 0; e < 12; ++e) s_acc[e][tid] += s_acc[e][tid + d];
    __syncthreads();
  }
  if (tid != 0) return;
  const float sc = (wgt ? wgt[p] : 1.f) * wscale / (float)M;
  float G[9];
  for (int e = 0; e < 9; ++e) G[e] = s_acc[3 + e][0] * sc;       // dL/dR (row-major)
  d_t[p * 3 + 0] = s_acc[0][0] * sc; d_t[p * 3 + 1] = s_acc[1][0] * sc; d_t[p * 3 + 2] = s_acc[2][0] * sc;
  const float *q = pred_r + p * 4;
  const float nq = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float qa = q[0] / nq, qb = q[1] / nq, qc = q[2] / nq, qd = q[3] / nq;
  // dL/dq_hat: partial derivatives of the nine entries of lib/loss.py:18-26
  const float ga = -2.f * qd * G[1] + 2.f * qc * G[2] + 2.f * qd * G[3] - 2.f * qb * G[5] - 2.f * qc * G[6] + 2.f * qb * G[7];
  const float gb = 2.f * qc * G[1] + 2.f * qd * G[2] + 2.f * qc * G[3] - 4.f * qb * G[4] - 2.f * qa * G[5] + 2.f * qd * G[6] + 2.f * qa * G[7] - 4.f * qb * G[8];
  const float gc = -4.f * qc * G[0] + 2.f * qb * G[1] + 2.f * qa * G[2] + 2.f * qb * G[3] + 2.f * qd * G[5] - 2.f * qa * G[6] + 2.f * qd * G[7] - 4.f * qc * G[8];
  const float gd = -4.f * qd * G[0] - 2.f * qa * G[1] + 2.f * qb * G[2] + 2.f * qa * G[3] - 4.f * qd * G[4] + 2.f * qc * G[5] + 2.f * qb * G[6] + 2.f * qc * G[7];
  // through q_hat = q / |q|:  dq = (g - q_hat (q_hat . g)) / |q|
  const float dot = qa * ga + qb * gb + qc * gc + qd * gd;
  d_r[p * 4 + 0] = (ga - qa * dot) / nq;
  d_r[p * 4 + 1] = (gb - qb * dot) / nq;
  d_r[p * 4 + 2] = (gc - qc * dot) / nq;
  d_r[p * 4 + 3] = (gd - qd * dot) / nq;
}

// PoseNet-loss weights: wgt[n] = c_n (the 1/N and upstream factor go into wscale); d_c[n] = g (dis_n - w/c_n) / N
__global__ __launch_bounds__(LB) void loss_dc_kernel(const float *__restrict__ pred_c, const float *__restrict__ dis, int N,
                                                     float w, float g, float *__restrict__ d_c, const FrameTab ft) {
  {
    const size_t fn = (size_t)ft.idx[blockIdx.y] * ft.N;
    pred_c += fn; dis += fn; d_c += fn;
  }
  for (int n = blockIdx.x * LB + threadIdx.x; n < N; n += gridDim.x * LB) d_c[n] = g * (dis[n] - w / pred_c[n]) / (float)N;
}

// one workgroup: loss = mean_n(dis*c - w*log c); which = argmax c; dis_sel = dis[which];
// new_points = (points - t*) . R*, new_target = (target - t*) . R*   (lib/loss.py:50-70)
__global__ __launch_bounds__(LB) void loss_finish_kernel(const float *__restrict__ pred_r, const float *__restrict__ pred_t,
                                                         const float *__restrict__ pred_c, const float *__restrict__ points,
                                                         const float *__restrict__ target, const float *__restrict__ dis,
                                                         int N, int M, float w, float *__restrict__ loss_out,
                                                         float *__restrict__ dis_out, float *__restrict__ new_points,
                                                         float *__restrict__ new_target, const FrameTab ft) {
  __shared__ float s_red[LB];
  __shared__ float s_v[LB];
  __shared__ int s_i[LB];
  const int tid = threadIdx.x;
  {
    const size_t f = ft.idx[blockIdx.y], fn = f * ft.N;
    pred_r += fn * 4; pred_t += fn * 3; pred_c += fn; points += fn * 3; target += f * ft.M * 3; dis += fn;
    loss_out += f; dis_out += f; new_points += fn * 3; new_target += f * ft.M * 3;
  }
  float acc = 0.f, best = -__builtin_inff();
  int bi = 0x7fffffff;
  for (int n = tid; n < N; n += LB) {
    const float c = pred_c[n];
    acc += dis[n] * c - w * logf(c);
    if (c > best) { best = c; bi = n; }
  }
  const float tot = block_sum(acc, s_red);
  s_v[tid] = best; s_i[tid] = bi;
  __syncthreads();
  for (int d = LB / 2; d >= 1; d >>= 1) {
    if (tid < d) {
      const float ov = s_v[tid + d];
      const int oi = s_i[tid + d];
      if (ov > s_v[tid] || (ov == s_v[tid] && oi < s_i[tid])) { s_v[tid] = ov; s_i[tid] = oi; }
    }
    __syncthreads();
  }
  int wm = s_i[0];
  if (wm < 0 || wm >= N) wm = 0;
  if (tid == 0) { loss_out[0] = tot / (float)N; dis_out[0] = dis[wm]; }
  const Rot R = quat_rot(pred_r + wm * 4);
  const float t0 = pred_t[wm * 3] + points[wm * 3], t1 = pred_t[wm * 3 + 1] + points[wm * 3 + 1],
              t2 = pred_t[wm * 3 + 2] + points[wm * 3 + 2];
  for (int i = tid; i < N + M; i += LB) {
    const float *src = i < N ? points + i * 3 : target + (i - N) * 3;
    float *dst = i < N ? new_points + i * 3 : new_target + (i - N) * 3;
    const float dx = src[0] - t0, dy = src[1] - t1, dz = src[2] - t2;
    dst[0] = dx * R.m[0] + dy * R.m[3] + dz * R.m[6];      // bmm(p - t, ori_base): row vector times R
    dst[1] = dx * R.m[1] + dy * R.m[4] + dz * R.m[7];
    dst[2] = dx * R.m[2] + dy * R.m[5] + dz * R.m[8];
  }
}

// refiner loss tail: re-centre by the single pose (lib/loss_refiner.py:50-59)
__global__ __launch_bounds__(LB) void recentre_kernel(const float *__restrict__ pred_r, const float *__restrict__ pred_t,
                                                      const float *__restrict__ points, const float *__restrict__ target,
                                                      int N, int M, float *__restrict__ new_points,
                                                      float *__restrict__ new_target, const FrameTab ft) {
  {      // one pose per frame: pred_r [B][4], pred_t [B][3]
    const size_t f = ft.idx[blockIdx.y];
    pred_r += f * 4; pred_t += f * 3; points += f * N * 3; target += f * M * 3; new_points += f * N * 3; new_target += f * M * 3;
  }
  const Rot R = quat_rot(pred_r);
  const float t0 = pred_t[0], t1 = pred_t[1], t2 = pred_t[2];
  for (int i = blockIdx.x * LB + threadIdx.x; i < N + M; i += gridDim.x * LB) {
    const float *src = i < N ? points + i * 3 : target + (i - N) * 3;
    float *dst = i < N ? new_points + i * 3 : new_target + (i - N) * 3;
    const float dx = src[0] - t0, dy = src[1] - t1, dz = src[2] - t2;
    dst[0] = dx * R.m[0] + dy * R.m[3] + dz * R.m[6];
    dst[1] = dx * R.m[1] + dy * R.m[4] + dz * R.m[7];
    dst[2] = dx * R.m[2] + dy * R.m[5] + dz * R.m[8];
  }
}

// tools/eval_linemod.py:118-130, one workgroup per object.  pred = model . R(q)^T + t in fp64 (numpy);
// non-symmetric: fp64 mean of fp64 norms; symmetric: pred/target cast to fp32, 1-NN, fp32 mean of norms.
__device__ void quat_to_mat64(const double *qin, double *M) {
  double q[4] = {qin[0], qin[1], qin[2], qin[3]};
  const double n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (n < 2.220446049250313e-16 * 4.0) { for (int i = 0; i < 9; ++i) M[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
  const double s = sqrt(2.0 / n);
  for (int i = 0; i < 4; ++i) q[i] *= s;
  M[0] = 1.0 - q[2] * q[2] - q[3] * q[3]; M[1] = q[1] * q[2] - q[3] * q[0];       M[2] = q[1] * q[3] + q[2] * q[0];
  M[3] = q[1] * q[2] + q[3] * q[0];       M[4] = 1.0 - q[1] * q[1] - q[3] * q[3]; M[5] = q[2] * q[3] - q[1] * q[0];
  M[6] = q[1] * q[3] - q[2] * q[0];       M[7] = q[2] * q[3] + q[1] * q[0];       M[8] = 1.0 - q[1] * q[1] - q[2] * q[2];
}

__global__ __launch_bounds__(LB) void add_metric_kernel(const double *__restrict__ pose, const float *__restrict__ model,
                                                        const float *__restrict__ target, const int *__restrict__ symmetric,
                                                        int M, double *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float s_tgt[];   // [M][4]
  __shared__ double s_red[LB];
  const int b = blockIdx.x, tid = threadIdx.x;
  model += (size_t)b * M * 3;
  target += (size_t)b * M * 3;
  const bool sym = symmetric && symmetric[b];
  for (int m = tid; m < M; m += LB)
    reinterpret_cast<float4 *>(s_tgt)[m] = make_float4(target[m * 3], target[m * 3 + 1], target[m * 3 + 2], 0.f);
  double R[9];
  quat_to_mat64(pose + b * 7, R);
  const double t0 = pose[b * 7 + 4], t1 = pose[b * 7 + 5], t2 = pose[b * 7 + 6];
  __syncthreads();
  double acc = 0.0;
  float accf = 0.f;
  for (int m = tid; m < M; m += LB) {
    const double x = model[m * 3], y = model[m * 3 + 1], z = model[m * 3 + 2];
    const double px = x * R[0] + y * R[1] + z * R[2] + t0;
    const double py = x * R[3] + y * R[4] + z * R[5] + t1;
    const double pz = x * R[6] + y * R[7] + z * R[8] + t2;
    if (!sym) {
      const double ex = px - (double)s_tgt[m * 4], ey = py - (double)s_tgt[m * 4 + 1], ez = pz - (double)s_tgt[m * 4 + 2];
      acc += sqrt(ex * ex + ey * ey + ez * ez);
    } else {
      const float fx = (float)px, fy = (float)py, fz = (float)pz;
      float best = __builtin_inff();
      int sel = 0;
      for (int r = 0; r < M; ++r) {
        const float4 q = reinterpret_cast<const float4 *>(s_tgt)[r];
        const float dx = q.x - fx, dy = q.y - fy, dz = q.z - fz;
        float d = dx * dx;
        d = __builtin_fmaf(dy, dy, d);
        d = __builtin_fmaf(dz, dz, d);
        const bool lt = d < best;
        best = lt ? d : best;
        sel = lt ? r : sel;
      }
      const float4 q = reinterpret_cast<const float4 *>(s_tgt)[sel];
      const float ex = fx - q.x, ey = fy - q.y, ez = fz - q.z;
      accf += sqrtf(ex * ex + ey * ey + ez * ez);
    }
  }
  s_red[tid] = sym ? (double)accf : acc;
  __syncthreads();
  for (int d = LB / 2; d >= 1; d >>= 1) {
    if (tid < d) s_red[tid] += s_red[tid + d];
    __syncthreads();
  }
  if (tid == 0) out[b] = sym ? (double)((float)s_red[0] / (float)M) : s_red[0] / (double)M;
}

// YCB-Video toolbox distances (replace_ycb_toolbox/evaluate_poses_keyframe.m:160-193), fp64 like MATLAB:
//   add = mean_m || RT_est p_m - RT_gt p_m ||
//   adi = mean_m  min_j || RT_est p_j - RT_gt p_m ||     (nearest ESTIMATED point for every GT point --
//                                                         the opposite direction to tools/eval_linemod.py)
// rt: [B][12] = 3x4 row-major [R|t]; pts: [B][M][3] fp64.  One workgroup per object, est points in LDS.
__global__ __launch_bounds__(LB) void ycb_dist_kernel(const double *__restrict__ rt_est, const double *__restrict__ rt_gt,
                                                      const double *__restrict__ pts, int M, double *__restrict__ add_out,
                                                      double *__restrict__ adi_out) {
  extern __shared__ __attribute__((aligned(16))) double s_est[];   // [M][3]
  __shared__ double s_red[2][LB];
  const int b = blockIdx.x, tid = threadIdx.x;
  const double *E = rt_est + b * 12, *G = rt_gt + b * 12, *P = pts + (size_t)b * M * 3;
  for (int m = tid; m < M; m += LB) {
    const double x = P[m * 3], y = P[m * 3 + 1], z = P[m * 3 + 2];
    s_est[m * 3 + 0] = E[0] * x + E[1] * y + E[2] * z + E[3];
    s_est[m * 3 + 1] = E[4] * x + E[5] * y + E[6] * z + E[7];
    s_est[m * 3 + 2] = E[8] * x + E[9] * y + E[10] * z + E[11];
  }
  __syncthreads();
  double a_add = 0.0, a_adi = 0.0;
  for (int m = tid; m < M; m += LB) {
    const double x = P[m * 3], y = P[m * 3 + 1], z = P[m * 3 + 2];
    const double gx = G[0] * x + G[1] * y + G[2] * z + G[3];
    const double gy = G[4] * x + G[5] * y + G[6] * z + G[7];
    const double gz = G[8] * x + G[9] * y + G[10] * z + G[11];
    const double dx = s_est[m * 3] - gx, dy = s_est[m * 3 + 1] - gy, dz = s_est[m * 3 + 2] - gz;
    a_add += sqrt(dx * dx + dy * dy + dz * dz);
    double best = __builtin_inf();
    for (int j = 0; j < M; ++j) {
      const double ex = s_est[j * 3] - gx, ey = s_est[j * 3 + 1] - gy, ez = s_est[j * 3 + 2] - gz;
      const double d = ex * ex + ey * ey + ez * ez;
      best = d < best ? d : best;
    }
    a_adi += sqrt(best);
  }
  s_red[0][tid] = a_add; s_red[1][tid] = a_adi;
  __syncthreads();
  for (int d = LB / 2; d >= 1; d >>= 1) {
    if (tid < d) { s_red[0][tid] += s_red[0][tid + d]; s_red[1][tid] += s_red[1][tid + d]; }
    __syncthreads();
  }
  if (tid == 0) { add_out[b] = s_red[0][0] / (double)M; adi_out[b] = s_red[1][0] / (double)M; }
}

// 150 KB of dynamic LDS for the kernels that keep a whole point set resident: a per-DEVICE attribute (a process may drive
// several devices), set the first time a launch is seen on a device
void loss_lds_attrs();

int check_m(int M, const char *what) {
  if (M <= 0 || (size_t)M * 16 > 150 * 1024) return set_error(DF_ERR_ARG, "%s: num_points_mesh must be in [1, 9600] (got %d)", what, M);
  return DF_OK;
}

void loss_lds_attrs() {
  static bool done[64] = {};
  int dev = 0;
  hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || done[dev]) return;
  const void *ks[] = {reinterpret_cast<const void *>(&add_dis_kernel), reinterpret_cast<const void *>(&add_dis_sym_kernel),
                      reinterpret_cast<const void *>(&add_metric_kernel), reinterpret_cast<const void *>(&ycb_dist_kernel)};
  for (const void *k : ks) hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  done[dev] = true;
}

}  // namespace
}  // namespace df

using namespace df;

namespace df {

int launch_loss_frames(int B, const int *symmetric, const float *pred_r, const float *pred_t, const float *pred_c, const float *target,
                       const float *model_points, const float *points, int N, int M, float w, float *loss_out, float *dis_out, float *new_points,
                       float *new_target, float *dis_scratch, int *sel, hipStream_t st) {
  if (B <= 0 || N <= 0) return set_error(DF_ERR_ARG, "loss_forward: B and N must be >= 1");
  int rc = check_m(M, "loss_forward");
  if (rc != DF_OK) return rc;
  const size_t lds = (size_t)M * 16;
  loss_lds_attrs();
  const int ppb = M >= LB * SYM_QPL ? 1 : std::min(LB, (LB * SYM_QPL) / M);      // (the per-pose totals are kept by tid < ppb <= LB)
  const size_t lds2 = (size_t)LB * SYM_QPL * 4 + (size_t)ppb * 4;
  const bool fused = N >= 2 && lds2 <= 150 * 1024;
  for (int b0 = 0; b0 < B; b0 += LOSS_MAX_FRAMES) {
    FrameTab all{}, sym{}, non{};
    all.N = sym.N = non.N = N; all.M = sym.M = non.M = M;
    for (int b = b0; b < std::min(B, b0 + LOSS_MAX_FRAMES); ++b) {
      all.idx[all.n++] = b;
      if (symmetric && symmetric[b]) sym.idx[sym.n++] = b; else non.idx[non.n++] = b;
    }
    if (sym.n && fused)
      // the fused transform + shared 1-NN scan (knn_core.h) + distance reduction; ppb whole poses per workgroup fill its 256 lanes x 2 queries
      hipLaunchKernelGGL(add_dis_sym_kernel, dim3((N + ppb - 1) / ppb, sym.n), dim3(LB), lds2, st, pred_r, pred_t, points, target, model_points, N, M, ppb,
                         dis_scratch, sel, sym);
    else if (sym.n)
      hipLaunchKernelGGL(add_dis_kernel, dim3(N, sym.n), dim3(LB), lds, st, pred_r, pred_t, points, target, model_points, M, 1, dis_scratch, sel, sym);
    if (non.n)
      hipLaunchKernelGGL(add_dis_kernel, dim3(N, non.n), dim3(LB), lds, st, pred_r, pred_t, points, target, model_points, M, 0, dis_scratch, (int *)nullptr, non);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1, all.n), dim3(LB), 0, st, pred_r, pred_t, pred_c, points, target, dis_scratch, N, M, w, loss_out, dis_out,
                       new_points, new_target, all);
  }
  return check_launch("loss_forward");
}

int launch_loss_bwd_frames(int B, const int *symmetric, const float *pred_r, const float *pred_t, const float *pred_c, const float *target,
                           const float *model_points, const float *points, const int *sel, const float *dis, int N, int M, float w, float g_loss,
                           float *d_pred_r, float *d_pred_t, float *d_pred_c, hipStream_t st) {
  if (B <= 0 || N <= 0 || M <= 0) return set_error(DF_ERR_ARG, "loss_backward: bad sizes");
  for (int b0 = 0; b0 < B; b0 += LOSS_MAX_FRAMES) {
    FrameTab all{}, sym{}, non{};
    all.N = sym.N = non.N = N; all.M = sym.M = non.M = M;
    for (int b = b0; b < std::min(B, b0 + LOSS_MAX_FRAMES); ++b) {
      all.idx[all.n++] = b;
      if (symmetric && symmetric[b]) sym.idx[sym.n++] = b; else non.idx[non.n++] = b;
    }
    if (sym.n)      // the nearest-neighbour matches of the forward pass enter as constants
      hipLaunchKernelGGL(add_dis_bwd_kernel, dim3(N, sym.n), dim3(LB), 0, st, pred_r, pred_t, points, target, model_points, sel, M, pred_c, g_loss / (float)N,
                         d_pred_r, d_pred_t, sym);
    if (non.n)
      hipLaunchKernelGGL(add_dis_bwd_kernel, dim3(N, non.n), dim3(LB), 0, st, pred_r, pred_t, points, target, model_points, (const int *)nullptr, M, pred_c,
                         g_loss / (float)N, d_pred_r, d_pred_t, non);
    hipLaunchKernelGGL(loss_dc_kernel, dim3(cdiv(N, LB), all.n), dim3(LB), 0, st, pred_c, dis, N, w, g_loss, d_pred_c, all);
  }
  return check_launch("loss_backward");
}

// Loss_refine (lib/loss_refiner.py:12-62) for B stacked frames, one pose each: pred_r [B][4], pred_t [B][3], points [B][N][3], target /
// model_points [B][M][3] -> dis_out [B], new_points, new_target; sel [B][M] (written for symmetric frames)
int launch_loss_refine_frames(int B, const int *symmetric, const float *pred_r, const float *pred_t, const float *target, const float *model_points,
                              const float *points, int N, int M, float *dis_out, float *new_points, float *new_target, int *sel, hipStream_t st) {
  if (B <= 0 || N <= 0) return set_error(DF_ERR_ARG, "loss_refine_forward: B and N must be >= 1");
  int rc = check_m(M, "loss_refine_forward");
  if (rc != DF_OK) return rc;
  loss_lds_attrs();
  for (int b0 = 0; b0 < B; b0 += LOSS_MAX_FRAMES) {
    FrameTab all{}, sym{}, non{};
    all.N = sym.N = non.N = 1; all.M = sym.M = non.M = M;          // one pose per frame: the per-frame strides of add_dis_kernel with N = 1
    for (int b = b0; b < std::min(B, b0 + LOSS_MAX_FRAMES); ++b) {
      all.idx[all.n++] = b;
      if (symmetric && symmetric[b]) sym.idx[sym.n++] = b; else non.idx[non.n++] = b;
    }
    if (sym.n)
      hipLaunchKernelGGL(add_dis_kernel, dim3(1, sym.n), dim3(LB), (size_t)M * 16, st, pred_r, pred_t, (const float *)nullptr, target, model_points, M, 1, dis_out,
                         sel, sym);
    if (non.n)
      hipLaunchKernelGGL(add_dis_kernel, dim3(1, non.n), dim3(LB), (size_t)M * 16, st, pred_r, pred_t, (const float *)nullptr, target, model_points, M, 0, dis_out,
                         (int *)nullptr, non);
    hipLaunchKernelGGL(recentre_kernel, dim3(cdiv(N + M, LB), all.n), dim3(LB), 0, st, pred_r, pred_t, points, target, N, M, new_points, new_target, all);
  }
  return check_launch("loss_refine_forward");
}

int launch_loss_refine_bwd_frames(int B, const int *symmetric, const float *pred_r, const float *pred_t, const float *target, const float *model_points,
                                  const int *sel, int M, float g_dis, float *d_pred_r, float *d_pred_t, hipStream_t st) {
  if (B <= 0 || M <= 0) return set_error(DF_ERR_ARG, "loss_refine_backward: bad sizes");
  for (int b0 = 0; b0 < B; b0 += LOSS_MAX_FRAMES) {
    FrameTab sym{}, non{};
    sym.N = non.N = 1; sym.M = non.M = M;
    for (int b = b0; b < std::min(B, b0 + LOSS_MAX_FRAMES); ++b)
      if (symmetric && symmetric[b]) sym.idx[sym.n++] = b; else non.idx[non.n++] = b;
    if (sym.n)
      hipLaunchKernelGGL(add_dis_bwd_kernel, dim3(1, sym.n), dim3(LB), 0, st, pred_r, pred_t, (const float *)nullptr, target, model_points, sel, M,
                         (const float *)nullptr, g_dis, d_pred_r, d_pred_t, sym);
    if (non.n)
      hipLaunchKernelGGL(add_dis_bwd_kernel, dim3(1, non.n), dim3(LB), 0, st, pred_r, pred_t, (const float *)nullptr, target, model_points, (const int *)nullptr, M,
                         (const float *)nullptr, g_dis, d_pred_r, d_pred_t, non);
  }
  return check_launch("loss_refine_backward");
}

}  // namespace df

extern "C" int df_loss_forward(const float *pred_r, const float *pred_t, const float *pred_c, const float *target,
                               const float *model_points, const float *points, int N, int M, float w, int symmetric,
                               float *loss_out, float *dis_out, float *new_points, float *new_target, float *dis_scratch,
                               int *sel_out, df_stream_t stream) {
  if (!pred_r || !pred_t || !pred_c || !target || !model_points || !points || !loss_out || !dis_out || !new_points ||
      !new_target || !dis_scratch)
    return set_error(DF_ERR_ARG, "loss_forward: null pointer");
  if (N <= 0) return set_error(DF_ERR_ARG, "loss_forward: N must be >= 1");
  const int sym = symmetric != 0;
  if (sym && !sel_out) {      // (the caller does not keep the matches: the one-frame kernels take a null selection buffer)
    int rc = check_m(M, "loss_forward");
    if (rc != DF_OK) return rc;
    hipStream_t st = to_stream(stream);
    loss_lds_attrs();
    const int ppb = M >= LB * SYM_QPL ? 1 : std::min(LB, (LB * SYM_QPL) / M);
    const size_t lds2 = (size_t)LB * SYM_QPL * 4 + (size_t)ppb * 4;
    const FrameTab one = one_frame(N, M);
    if (N >= 2 && lds2 <= 150 * 1024)
      hipLaunchKernelGGL(add_dis_sym_kernel, dim3((N + ppb - 1) / ppb), dim3(LB), lds2, st, pred_r, pred_t, points, target, model_points, N, M, ppb,
                         dis_scratch, (int *)nullptr, one);
    else
      hipLaunchKernelGGL(add_dis_kernel, dim3(N), dim3(LB), (size_t)M * 16, st, pred_r, pred_t, points, target, model_points, M, 1, dis_scratch, (int *)nullptr, one);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(LB), 0, st, pred_r, pred_t, pred_c, points, target, dis_scratch, N, M, w, loss_out, dis_out,
                       new_points, new_target, one);
    return check_launch("loss_forward");
  }
  return launch_loss_frames(1, &sym, pred_r, pred_t, pred_c, target, model_points, points, N, M, w, loss_out, dis_out, new_points, new_target, dis_scratch,
                            sel_out, to_stream(stream));
}

extern "C" int df_loss_forward_frames(int B, const int *symmetric, const float *pred_r, const float *pred_t, const float *pred_c, const float *target,
                                      const float *model_points, const float *points, int N, int M, float w, float *loss_out, float *dis_out,
                                      float *new_points, float *new_target, float *dis_scratch, int *sel_out, df_stream_t stream) {
  if (!pred_r || !pred_t || !pred_c || !target || !model_points || !points || !loss_out || !dis_out || !new_points || !new_target || !dis_scratch)
    return set_error(DF_ERR_ARG, "loss_forward_frames: null pointer");
  if (B <= 0 || N <= 0) return set_error(DF_ERR_ARG, "loss_forward_frames: B and N must be >= 1");
  if (!sel_out && symmetric) {      // no room for the matches: the symmetric frames go one at a time through the form that keeps none
    for (int b = 0; b < B; ++b) {
      const int rc = df_loss_forward(pred_r + (size_t)b * N * 4, pred_t + (size_t)b * N * 3, pred_c + (size_t)b * N, target + (size_t)b * M * 3,
                                     model_points + (size_t)b * M * 3, points + (size_t)b * N * 3, N, M, w, symmetric[b], loss_out + b, dis_out + b,
                                     new_points + (size_t)b * N * 3, new_target + (size_t)b * M * 3, dis_scratch + (size_t)b * N, nullptr, stream);
      if (rc != DF_OK) return rc;
    }
    return DF_OK;
  }
  return launch_loss_frames(B, symmetric, pred_r, pred_t, pred_c, target, model_points, points, N, M, w, loss_out, dis_out, new_points, new_target, dis_scratch,
                            sel_out, to_stream(stream));
}

extern "C" int df_loss_refine_forward(const float *pred_r, const float *pred_t, const float *target, const float *model_points,
                                      const float *points, int N, int M, int symmetric, float *dis_out, float *new_points,
                                      float *new_target, int *sel_out, df_stream_t stream) {
  if (!pred_r || !pred_t || !target || !model_points || !points || !dis_out || !new_points || !new_target)
    return set_error(DF_ERR_ARG, "loss_refine_forward: null pointer");
  if (N <= 0) return set_error(DF_ERR_ARG, "loss_refine_forward: N must be >= 1");
  int rc = check_m(M, "loss_refine_forward");
  if (rc != DF_OK) return rc;
  hipStream_t st = to_stream(stream);
  loss_lds_attrs();
  hipLaunchKernelGGL(add_dis_kernel, dim3(1), dim3(LB), (size_t)M * 16, st, pred_r, pred_t, (const float *)nullptr, target,
                     model_points, M, symmetric, dis_out, sel_out, one_frame(1, M));
  hipLaunchKernelGGL(recentre_kernel, dim3(cdiv(N + M, LB)), dim3(LB), 0, st, pred_r, pred_t, points, target, N, M, new_points, new_target, one_frame(1, M));
  return check_launch("loss_refine_forward");
}

extern "C" int df_add_metric(const double *pose, const float *model_points, const float *target, const int *symmetric, int B,
                             int M, double *dis_out, df_stream_t stream) {
  if (!pose || !model_points || !target || !dis_out) return set_error(DF_ERR_ARG, "add_metric: null pointer");
  if (B <= 0) return set_error(DF_ERR_ARG, "add_metric: B must be >= 1");
  int rc = check_m(M, "add_metric");
  if (rc != DF_OK) return rc;
  loss_lds_attrs();
  hipLaunchKernelGGL(add_metric_kernel, dim3(B), dim3(LB), (size_t)M * 16, to_stream(stream), pose, model_points, target, symmetric, M, dis_out);
  return check_launch("add_metric");
}

extern "C" int df_ycb_distances(const double *rt_est, const double *rt_gt, const double *pts, int B, int M, double *add_out,
                                double *adi_out, df_stream_t stream) {
  if (!rt_est || !rt_gt || !pts || !add_out || !adi_out) return set_error(DF_ERR_ARG, "ycb_distances: null pointer");
  if (B <= 0 || M <= 0 || (size_t)M * 24 > 150 * 1024) return set_error(DF_ERR_ARG, "ycb_distances: need B >= 1 and 1 <= M <= 6400");
  loss_lds_attrs();
  hipLaunchKernelGGL(ycb_dist_kernel, dim3(B), dim3(LB), (size_t)M * 24, to_stream(stream), rt_est, rt_gt, pts, M, add_out, adi_out);
  return check_launch("ycb_distances");
}

extern "C" int df_loss_backward(const float *pred_r, const float *pred_t, const float *pred_c, const float *target,
                                const float *model_points, const float *points, const int *sel, const float *dis, int N, int M,
                                float w, float g_loss, float *d_pred_r, float *d_pred_t, float *d_pred_c, df_stream_t stream) {
  if (!pred_r || !pred_t || !pred_c || !target || !model_points || !points || !dis || !d_pred_r || !d_pred_t || !d_pred_c)
    return set_error(DF_ERR_ARG, "loss_backward: null pointer");
  const int sym = sel != nullptr;
  return launch_loss_bwd_frames(1, &sym, pred_r, pred_t, pred_c, target, model_points, points, sel, dis, N, M, w, g_loss, d_pred_r, d_pred_t, d_pred_c,
                                to_stream(stream));
}

extern "C" int df_loss_refine_backward(const float *pred_r, const float *pred_t, const float *target, const float *model_points,
                                       const int *sel, int M, float g_dis, float *d_pred_r, float *d_pred_t, df_stream_t stream) {
  if (!pred_r || !pred_t || !target || !model_points || !d_pred_r || !d_pred_t) return set_error(DF_ERR_ARG, "loss_refine_backward: null pointer");
  if (M <= 0) return set_error(DF_ERR_ARG, "loss_refine_backward: bad sizes");
  hipLaunchKernelGGL(add_dis_bwd_kernel, dim3(1), dim3(LB), 0, to_stream(stream), pred_r, pred_t, (const float *)nullptr, target,
                     model_points, sel, M, (const float *)nullptr, g_dis, d_pred_r, d_pred_t, one_frame(1, M));
  return check_launch("loss_refine_backward");
}

// ------------------------------------------------------------------------------------------------
// Adam step on a flat fp32 buffer (what optim.Adam does per tensor in tools/train.py:99,166-169: default
// betas/eps, no weight decay, no amsgrad); grad_scale folds the 1/(world*accumulated) averaging of the
// data-parallel all-reduce into the same pass.
// ------------------------------------------------------------------------------------------------
namespace df {
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt, float grad_scale) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * grad_scale;
    const float mi = m[i] + (1.f - b1) * (gi - m[i]);             // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - (lr / bc1) * (mi / denom);
  }
}
}  // namespace df

extern "C" int df_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                            float beta2, float eps, int step, float grad_scale, df_stream_t stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq) return set_error(DF_ERR_ARG, "adam_step: null pointer");
  if (n < 0 || step < 1) return set_error(DF_ERR_ARG, "adam_step: need n >= 0 and step >= 1");
  if (n == 0) return DF_OK;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, to_stream(stream), param, grad, exp_avg, exp_avg_sq, (long)n,
                     lr, beta1, beta2, eps, bc1, sqrtf(bc2), grad_scale);
  return check_launch("adam_step");
}

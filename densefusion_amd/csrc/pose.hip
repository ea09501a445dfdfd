// On-device pose algebra of the eval loops, so the refine loop never returns to the host.
//
// The reference does this part in numpy float64 on the host with one D2H sync per step
// (tools/eval_ycb.py:193-229, tools/eval_linemod.py:82-114).  Here one tiny workgroup per object does
// the same arithmetic in fp64 on the GPU: per-pixel pose selection (arg-max confidence), quaternion ->
// rotation (lib/transformations.py:1266-1278), 4x4 composition and rotation -> quaternion
// (lib/transformations.py:1320-1341,1361-1363), and hands the fp32 R|T the next refiner pass needs
// to the cloud kernel through a 12-float record per object.
#include "pose.h"

namespace df {
namespace {

constexpr double QEPS = 2.220446049250313e-16 * 4.0;   // numpy.finfo(float).eps * 4 (transformations.py:1893)

// lib/transformations.py:1266-1278 ; M is 3x3 row-major
__device__ void quat_to_mat(const double qin[4], double M[9]) {
  double q[4] = {qin[0], qin[1], qin[2], qin[3]};
  const double n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (n < QEPS) {
    M[0] = 1; M[1] = 0; M[2] = 0; M[3] = 0; M[4] = 1; M[5] = 0; M[6] = 0; M[7] = 0; M[8] = 1;
    return;
  }
  const double s = sqrt(2.0 / n);
  for (int i = 0; i < 4; ++i) q[i] *= s;
  double o[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) o[i][j] = q[i] * q[j];
  M[0] = 1.0 - o[2][2] - o[3][3]; M[1] = o[1][2] - o[3][0];       M[2] = o[1][3] + o[2][0];
  M[3] = o[1][2] + o[3][0];       M[4] = 1.0 - o[1][1] - o[3][3]; M[5] = o[2][3] - o[1][0];
  M[6] = o[1][3] - o[2][0];       M[7] = o[2][3] + o[1][0];       M[8] = 1.0 - o[1][1] - o[2][2];
}

// lib/transformations.py:1320-1341 (isprecise=True; homogeneous M[3][3] == 1) + sign rule :1361-1363
__device__ void mat_to_quat_precise(const double M[9], double q[4]) {
  const double m33 = 1.0;
  double t = M[0] + M[4] + M[8] + m33;
  if (t > m33) {
    q[0] = t;
    q[3] = M[3] - M[1];
    q[2] = M[2] - M[6];
    q[1] = M[7] - M[5];
  } else {
    int i = 0, j = 1, k = 2;
    if (M[4] > M[0]) { i = 1; j = 2; k = 0; }
    if (M[8] > M[i * 3 + i]) { i = 2; j = 0; k = 1; }
    t = M[i * 3 + i] - (M[j * 3 + j] + M[k * 3 + k]) + m33;
    double p[4];
    p[i] = t;
    p[j] = M[i * 3 + j] + M[j * 3 + i];
    p[k] = M[k * 3 + i] + M[i * 3 + k];
    p[3] = M[k * 3 + j] - M[j * 3 + k];
    q[0] = p[3]; q[1] = p[0]; q[2] = p[1]; q[3] = p[2];
  }
  const double s = 0.5 / sqrt(t * m33);
  for (int e = 0; e < 4; ++e) q[e] *= s;
  if (q[0] < 0.0)
    for (int e = 0; e < 4; ++e) q[e] = -q[e];
}

// R (fp64) and t -> the fp32 record the refiner's cloud kernel reads (eval_ycb.py:206-209 casts)
__device__ void write_rt(const double M[9], const double t[3], float *rt) {
  for (int e = 0; e < 9; ++e) rt[e] = (float)M[e];
  for (int e = 0; e < 3; ++e) rt[9 + e] = (float)t[e];
}

// one wave per object: last refiner layer for the selected object (lib/network.py:198-204), then
// normalise, compose with the running pose and refresh R|T (eval_ycb.py:213-229)
__global__ __launch_bounds__(64) void refiner_tail_kernel(const float *__restrict__ f2, const float *__restrict__ w_r,
                                                          const float *__restrict__ b_r, const float *__restrict__ w_t,
                                                          const float *__restrict__ b_t, const int64_t *__restrict__ obj,
                                                          int num_obj, float *__restrict__ out_r, float *__restrict__ out_t,
                                                          double *__restrict__ state, float *__restrict__ rt,
                                                          double *__restrict__ pose_out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  long o = obj[b];
  o = o < 0 ? 0 : (o >= num_obj ? num_obj - 1 : o);
  float y[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const float *w = j < 4 ? w_r + (o * 4 + j) * 128 : w_t + (o * 3 + (j - 4)) * 128;
    const float *x = f2 + (size_t)b * 256 + (j < 4 ? 0 : 128);
    float acc = x[lane] * w[lane] + x[lane + 64] * w[lane + 64];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    y[j] = acc + (j < 4 ? b_r[o * 4 + j] : b_t[o * 3 + (j - 4)]);
  }
  if (lane != 0) return;
  if (out_r) for (int e = 0; e < 4; ++e) out_r[b * 4 + e] = y[e];
  if (out_t) for (int e = 0; e < 3; ++e) out_t[b * 3 + e] = y[4 + e];
  if (!state) return;
  const float nrm = sqrtf(y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3]);
  double q2[4], t2[3], M1[9], M2[9], Mf[9], tf[3], qf[4];
  for (int e = 0; e < 4; ++e) q2[e] = (double)(y[e] / nrm);
  for (int e = 0; e < 3; ++e) t2[e] = (double)y[4 + e];
  double *st = state + b * 7;
  quat_to_mat(st, M1);
  quat_to_mat(q2, M2);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Mf[i * 3 + j] = M1[i * 3 + 0] * M2[0 * 3 + j] + M1[i * 3 + 1] * M2[1 * 3 + j] + M1[i * 3 + 2] * M2[2 * 3 + j];
    tf[i] = M1[i * 3 + 0] * t2[0] + M1[i * 3 + 1] * t2[1] + M1[i * 3 + 2] * t2[2] + st[4 + i];
  }
  mat_to_quat_precise(Mf, qf);
  for (int e = 0; e < 4; ++e) st[e] = qf[e];
  for (int e = 0; e < 3; ++e) st[4 + e] = tf[e];
  if (pose_out) for (int e = 0; e < 7; ++e) pose_out[b * 7 + e] = st[e];
  double Mn[9];
  quat_to_mat(st, Mn);
  write_rt(Mn, st + 4, rt + b * 12);
}


// ---- confidence-first head evaluation (eval loop only) -------------------------------------------------------------
// tools/eval_ycb.py:193-203 uses pred_r / pred_t at ONE pixel per object -- the arg-max of the confidence -- so the
// r and t towers (lib/network.py:107-121) need evaluating at that pixel only.  The engine runs the confidence tower
// for all N points as GEMMs; head_conf_kernel adds conv4_c + sigmoid, and head_select_kernel finishes the job: arg-max
// over the N points, then conv1..conv4 of tower blockIdx.y (0 = r, 1 = t) for the winning point as wave-per-output dot
// products, then the pose record.
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct HeadSel {
  const float *conf;                      // [B][N] confidences (head_conf_kernel)
  const float *pf;                        // [rows][384] pointfeat_1 | pointfeat_2
  const float *gbias;                     // [B][1920] global-feature part of head layer 1 (+ bias), towers r|t|c
  const float *w1;                        // [1920][384] per-point part of head layer 1
  const float *w2, *b2;                   // [3][256][640], [768]
  const float *w3, *b3;                   // [3][128][256], [384]
  const float *w_r, *b_r, *w_t, *b_t;     // conv4_r [4K][128], conv4_t [3K][128]
  const int64_t *obj;
  const float *cloud;                     // [B][N][3]
  int num_obj, N, Npad;
  double *pose_wo, *state;
  float *rt;
  int *which;
};

// NB outputs at once: dot products of NB consecutive K4*4-long rows of `w` (row stride K4*4) with the LDS vector `x`.
// Lanes stride the float4s; all NB*ceil(K4/64) weight loads are issued before the first use, then the NB partial sums
// are wave-reduced together (independent shuffle chains).
template <int K4, int NB>
__device__ __forceinline__ void wave_dots(const float *__restrict__ w, const float *x, int lane, float (&out)[NB]) {
  constexpr int IT = (K4 + 63) / 64;
  float4 wv[NB][IT];
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int k = lane + 64 * i;
      wv[n][i] = k < K4 ? reinterpret_cast<const float4 *>(w + (size_t)n * K4 * 4)[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
  for (int n = 0; n < NB; ++n) out[n] = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int k = lane + 64 * i;
    const float4 v = k < K4 ? reinterpret_cast<const float4 *>(x)[k] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int n = 0; n < NB; ++n) out[n] += (wv[n][i].x * v.x + wv[n][i].y * v.y) + (wv[n][i].z * v.z + wv[n][i].w * v.w);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
    for (int n = 0; n < NB; ++n) out[n] += __shfl_xor(out[n], d);
}

// confidence of every point: thread = point, conv4_c row of the object + sigmoid -- the expression and summation order of
// head_final_kernel (layers.hip), so the full forward and the eval loop see the same values and pick the same point
__global__ __launch_bounds__(256) void head_conf_kernel(const float *__restrict__ h3c, const float *__restrict__ w_c,
                                                        const float *__restrict__ b_c, const int64_t *__restrict__ obj, int num_obj,
                                                        float *__restrict__ conf, int B, int N, int Npad) {
  const long total = (long)B * N;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
    const long b = p / N, n = p - b * N;
    long o = obj[b];
    o = o < 0 ? 0 : (o >= num_obj ? num_obj - 1 : o);
    const f32x4 *wv = reinterpret_cast<const f32x4 *>(w_c + o * 128);
    const f32x4 *xv = reinterpret_cast<const f32x4 *>(h3c + ((size_t)b * Npad + n) * 128);
    float acc = 0.f;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const f32x4 x = xv[k], c = wv[k];
      acc += (x[0] * c[0] + x[1] * c[1]) + (x[2] * c[2] + x[3] * c[3]);
    }
    acc += b_c[o];
    conf[p] = 1.f / (1.f + expf(-acc));
  }
}

__global__ __launch_bounds__(256) void head_select_kernel(HeadSel a) {
  __shared__ float s_v[256];
  __shared__ int s_i[256];
  __shared__ __attribute__((aligned(16))) float s_x[384], s_h1[640], s_h2[256], s_h3[128];
  __shared__ float s_y[4];
  const int b = blockIdx.x, tower = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long o = a.obj[b];
  o = o < 0 ? 0 : (o >= a.num_obj ? a.num_obj - 1 : o);
  // 1. first maximum of the confidences wins (torch.max; eval_ycb.py:196)
  float best = -__builtin_inff();
  int bi = 0x7fffffff;
  for (int n = tid; n < a.N; n += 256) {
    const float conf = a.conf[(size_t)b * a.N + n];
    if (conf > best) { best = conf; bi = n; }
  }
  s_v[tid] = best; s_i[tid] = bi;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if (tid < d) {
      const float ov = s_v[tid + d];
      const int oi = s_i[tid + d];
      if (ov > s_v[tid] || (ov == s_v[tid] && oi < s_i[tid])) { s_v[tid] = ov; s_i[tid] = oi; }
    }
    __syncthreads();
  }
  int wm = s_i[0];
  if (wm < 0 || wm >= a.N) wm = 0;             // all-NaN confidences
  const size_t row = (size_t)b * a.Npad + wm;
  // 2. tower `tower` at the winning point
  for (int k = tid; k < 384; k += 256) s_x[k] = a.pf[row * 384 + k];
  __syncthreads();
  constexpr int NB = 8;                      // outputs per wave step: 8 independent weight-row streams in flight
  for (int j0 = wave * NB; j0 < 640; j0 += 4 * NB) {
    float v[NB];
    wave_dots<96, NB>(a.w1 + (size_t)(tower * 640 + j0) * 384, s_x, lane, v);
#pragma unroll
    for (int n = 0; n < NB; ++n)
      if (lane == n) s_h1[j0 + n] = fmaxf(v[n] + a.gbias[(size_t)b * 1920 + tower * 640 + j0 + n], 0.f);
  }
  __syncthreads();
  for (int j0 = wave * NB; j0 < 256; j0 += 4 * NB) {
    float v[NB];
    wave_dots<160, NB>(a.w2 + ((size_t)tower * 256 + j0) * 640, s_h1, lane, v);
#pragma unroll
    for (int n = 0; n < NB; ++n)
      if (lane == n) s_h2[j0 + n] = fmaxf(v[n] + a.b2[tower * 256 + j0 + n], 0.f);
  }
  __syncthreads();
  for (int j0 = wave * NB; j0 < 128; j0 += 4 * NB) {
    float v[NB];
    wave_dots<64, NB>(a.w3 + ((size_t)tower * 128 + j0) * 256, s_h2, lane, v);
#pragma unroll
    for (int n = 0; n < NB; ++n)
      if (lane == n) s_h3[j0 + n] = fmaxf(v[n] + a.b3[tower * 128 + j0 + n], 0.f);
  }
  __syncthreads();
  const int nout = tower == 0 ? 4 : 3;
  if (wave < nout) {
    const float *w = tower == 0 ? a.w_r + (o * 4 + wave) * 128 : a.w_t + (o * 3 + wave) * 128;
    float v[1];
    wave_dots<32, 1>(w, s_h3, lane, v);
    if (lane == 0) s_y[wave] = v[0] + (tower == 0 ? a.b_r[o * 4 + wave] : a.b_t[o * 3 + wave]);
  }
  __syncthreads();
  if (tid != 0) return;
  // 3. pose record (eval_ycb.py:197-203): normalised quaternion, point + offset, and the fp32 R|T the refiner's cloud kernel reads
  if (tower == 0) {
    const float nrm = sqrtf(s_y[0] * s_y[0] + s_y[1] * s_y[1] + s_y[2] * s_y[2] + s_y[3] * s_y[3]);
    double q[4], M[9];
    for (int e = 0; e < 4; ++e) {
      q[e] = (double)(s_y[e] / nrm);
      a.state[b * 7 + e] = q[e];
      if (a.pose_wo) a.pose_wo[b * 7 + e] = q[e];
    }
    quat_to_mat(q, M);
    for (int e = 0; e < 9; ++e) a.rt[b * 12 + e] = (float)M[e];
    if (a.which) a.which[b] = wm;
  } else {
    for (int e = 0; e < 3; ++e) {
      const double t = (double)(a.cloud[((size_t)b * a.N + wm) * 3 + e] + s_y[e]);
      a.state[b * 7 + 4 + e] = t;
      if (a.pose_wo) a.pose_wo[b * 7 + 4 + e] = t;
      a.rt[b * 12 + 9 + e] = (float)t;
    }
  }
}

}  // namespace

void launch_head_select(const float *h3c, const float *w_c, const float *b_c, const float *pf, const float *gbias, const float *w1,
                        const float *w2, const float *b2, const float *w3, const float *b3, const float *w_r, const float *b_r,
                        const float *w_t, const float *b_t, const int64_t *obj, int num_obj, const float *cloud, int B, int N, int Npad,
                        float *conf, double *pose_wo, double *state, float *rt, int *which, hipStream_t st) {
  const long pts = (long)B * N;
  hipLaunchKernelGGL(head_conf_kernel, dim3((unsigned)((pts + 255) / 256)), dim3(256), 0, st, h3c, w_c, b_c, obj, num_obj, conf, B, N, Npad);
  HeadSel a;
  a.conf = conf; a.pf = pf; a.gbias = gbias; a.w1 = w1; a.w2 = w2; a.b2 = b2; a.w3 = w3; a.b3 = b3;
  a.w_r = w_r; a.b_r = b_r; a.w_t = w_t; a.b_t = b_t; a.obj = obj; a.cloud = cloud; a.num_obj = num_obj; a.N = N; a.Npad = Npad;
  a.pose_wo = pose_wo; a.state = state; a.rt = rt; a.which = which;
  hipLaunchKernelGGL(head_select_kernel, dim3(B, 2), dim3(256), 0, st, a);
}

void launch_refiner_tail(const float *f2, const float *w_r, const float *b_r, const float *w_t, const float *b_t,
                         const int64_t *obj, int num_obj, float *out_r, float *out_t, double *state, float *rt,
                         double *pose_out, int B, hipStream_t st) {
  hipLaunchKernelGGL(refiner_tail_kernel, dim3(B), dim3(64), 0, st, f2, w_r, b_r, w_t, b_t, obj, num_obj, out_r, out_t,
                     state, rt, pose_out);
}

}  // namespace df

// Per-object input preparation on the device: what tools/eval_ycb.py:147-181 does in numpy
// between the detector's ROI and the network call -- mask = (depth != 0) & (label == itemid) inside the snapped
// bounding box, `choose` = up to num_points mask pixels (random subset when there are more, wrap-padding when
// fewer, eval_ycb.py:155-163), back-projection of the chosen depth pixels to a cloud (:165-173) and the
// ImageNet-normalised crop of the colour image (:175-181, on 0..255-scale values exactly like the reference).
//
// RNG contract (the reference uses np.random.shuffle, whose stream cannot be shared with a GPU): every mask
// pixel gets the key mix32(seed, flat crop index); the num_points pixels with the smallest keys (ties: lower
// index) are kept, in increasing index order.  Same distribution (a uniformly random subset, order preserved),
// reproducible from `seed`; the CPU checker of the test suite implements the same contract.
#include "common.h"

namespace df {
namespace {

constexpr int PB = 1024;

__device__ __host__ inline unsigned mix32(unsigned seed, unsigned i) {
  unsigned x = seed ^ (i * 0x9E3779B9u);
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__device__ inline int block_excl_scan(int v, int *s_buf, int &total) {
  const int tid = threadIdx.x;
  s_buf[tid] = v;
  __syncthreads();
  for (int d = 1; d < PB; d <<= 1) {
    const int t = tid >= d ? s_buf[tid - d] : 0;
    __syncthreads();
    s_buf[tid] += t;
    __syncthreads();
  }
  total = s_buf[PB - 1];
  const int r = s_buf[tid] - v;
  __syncthreads();
  return r;
}

struct ObjDesc {   // one object: frame index and snapped bounding box (host side: get_bbox, eval_ycb.py:54-90)
  int frame, itemid, rmin, rmax, cmin, cmax;
  unsigned seed;
  int given;         // != 0: this object's row of `choose` was filled in by the caller (chosen indices as an INPUT): step 2 is skipped
};

// grid = B objects, block = 1024.  rgb [F][IH][IW][3] u8, depth [F][IH][IW] u16, label [F][IH][IW] i32.
__global__ __launch_bounds__(PB) void preprocess_kernel(const unsigned char *__restrict__ rgb, const unsigned short *__restrict__ depth,
                                                        const int *__restrict__ label, const ObjDesc *__restrict__ objs, int IH,
                                                        int IW, int H, int W, int N, float cx, float cy, float fx, float fy,
                                                        float cam_scale, float cloud_div, int *__restrict__ nz_scratch, float *__restrict__ img,
                                                        float *__restrict__ cloud, int64_t *__restrict__ choose,
                                                        int *__restrict__ count_out) {
  __shared__ int s_scan[PB];
  __shared__ unsigned s_hist[256];
  __shared__ unsigned s_prefix, s_remaining;
  const int b = blockIdx.x, tid = threadIdx.x;
  const ObjDesc o = objs[b];
  const size_t fbase = (size_t)o.frame * IH * IW;
  const int HW = H * W;
  int *nz = nz_scratch + (size_t)b * HW;
  const int chunk = (HW + PB - 1) / PB;
  const int i0 = tid * chunk, i1 = min(HW, i0 + chunk);
  auto in_mask = [&](int i) {
    const int r = o.rmin + i / W, c = o.cmin + i % W;
    const size_t p = fbase + (size_t)r * IW + c;
    return depth[p] != 0 && label[p] == o.itemid;
  };
  // 1. ordered compaction of the mask pixels (flat crop indices)
  int cnt = 0;
  for (int i = i0; i < i1; ++i) cnt += in_mask(i);
  int total;
  int off = block_excl_scan(cnt, s_scan, total);
  for (int i = i0; i < i1; ++i)
    if (in_mask(i)) nz[off++] = i;
  if (tid == 0) count_out[b] = total;
  __syncthreads();
  // 2. choose
  int64_t *ch = choose + (size_t)b * N;
  if (o.given) {
    // the caller's indices (e.g. the subset the reference's np.random.shuffle drew): clamped into the crop, otherwise taken as they are
    for (int j = tid; j < N; j += PB) { const int64_t v = ch[j]; ch[j] = v < 0 ? 0 : (v >= HW ? HW - 1 : v); }
  } else if (total == 0) {
    for (int j = tid; j < N; j += PB) ch[j] = 0;     // detector lost the object; the caller checks count
  } else if (total <= N) {
    for (int j = tid; j < N; j += PB) ch[j] = nz[j % total];        // np.pad(..., 'wrap')
  } else {
    // radix select of the N-th smallest key (4 rounds of 8 bits)
    if (tid == 0) { s_prefix = 0; s_remaining = (unsigned)N; }
    __syncthreads();
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) s_hist[tid] = 0;
      __syncthreads();
      const unsigned prefix = s_prefix, hmask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
      for (int j = tid; j < total; j += PB) {
        const unsigned k = mix32(o.seed, (unsigned)nz[j]);
        if ((k & hmask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        unsigned rem = s_remaining, bin = 0;
        while (s_hist[bin] < rem) { rem -= s_hist[bin]; ++bin; }
        s_prefix = prefix | (bin << shift);
        s_remaining = rem;          // how many keys equal to the final threshold are still to be taken
      }
      __syncthreads();
    }
    const unsigned T = s_prefix, ties = s_remaining;
    // keep keys < T, plus the `ties` lowest-index entries with key == T; ordered compaction into choose
    const int c2 = (total + PB - 1) / PB;
    const int j0 = tid * c2, j1 = min(total, j0 + c2);
    int less = 0, eq = 0;
    for (int j = j0; j < j1; ++j) {
      const unsigned k = mix32(o.seed, (unsigned)nz[j]);
      less += k < T; eq += k == T;
    }
    int tot_eq, tot_less;
    int eq_off = block_excl_scan(eq, s_scan, tot_eq);
    // number of selected entries before this thread's chunk = less-before + min(eq-before, ties)
    int less_off = block_excl_scan(less, s_scan, tot_less);
    int out = less_off + min(eq_off, (int)ties);
    for (int j = j0; j < j1; ++j) {
      const unsigned k = mix32(o.seed, (unsigned)nz[j]);
      bool take = k < T;
      if (k == T) { take = eq_off < (int)ties; ++eq_off; }
      if (take) ch[out++] = nz[j];
    }
  }
  __syncthreads();
  // 3. cloud from the chosen depth pixels (eval_ycb.py:165-173; xmap = row index, ymap = column index)
  float *cl = cloud + (size_t)b * N * 3;
  for (int j = tid; j < N; j += PB) {
    const int i = (int)ch[j];
    const int r = o.rmin + i / W, c = o.cmin + i % W;
    const float d = (float)depth[fbase + (size_t)r * IW + c];
    const float pt2 = d / cam_scale;
    // cloud_div: the LineMOD loader back-projects in depth units and divides the finished cloud by 1000
    // (datasets/linemod/dataset.py:152-157); the YCB path passes 1 (x / 1 == x)
    cl[j * 3 + 0] = ((float)c - cx) * pt2 / fx / cloud_div;
    cl[j * 3 + 1] = ((float)r - cy) * pt2 / fy / cloud_div;
    cl[j * 3 + 2] = pt2 / cloud_div;
  }
  // 4. normalised colour crop, CHW (eval_ycb.py:175-181)
  float *im = img + (size_t)b * 3 * HW;
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  for (int i = tid; i < HW; i += PB) {
    const int r = o.rmin + i / W, c = o.cmin + i % W;
    const unsigned char *px = rgb + (fbase + (size_t)r * IW + c) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) im[(size_t)k * HW + i] = ((float)px[k] - mean[k]) / stdv[k];
  }
}

}  // namespace
}  // namespace df

using namespace df;

extern "C" int df_preprocess_objects(const unsigned char *rgb, const unsigned short *depth, const int *label, int num_frames,
                                     int IH, int IW, const int *obj_desc, int B, int H, int W, int num_points, float cam_cx,
                                     float cam_cy, float cam_fx, float cam_fy, float cam_scale, float cloud_div, int *scratch, float *img_out,
                                     float *cloud_out, int64_t *choose_out, int *count_out, df_stream_t stream) {
  if (!rgb || !depth || !label || !obj_desc || !scratch || !img_out || !cloud_out || !choose_out || !count_out)
    return set_error(DF_ERR_ARG, "preprocess: null pointer");
  if (B <= 0 || num_frames <= 0 || H <= 0 || W <= 0 || H > IH || W > IW || num_points <= 0 || !(cam_scale > 0.f) || !(cloud_div > 0.f))
    return set_error(DF_ERR_ARG, "preprocess: bad sizes");
  hipLaunchKernelGGL(preprocess_kernel, dim3(B), dim3(PB), 0, to_stream(stream), rgb, depth, label,
                     reinterpret_cast<const ObjDesc *>(obj_desc), IH, IW, H, W, num_points, cam_cx, cam_cy, cam_fx, cam_fy, cam_scale,
                     cloud_div, scratch, img_out, cloud_out, choose_out, count_out);
  return check_launch("preprocess");
}

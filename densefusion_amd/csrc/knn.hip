// Fused brute-force nearest neighbour for gfx950.
//
// Replaces the reference's two-kernel algorithm (lib/knn/src/knn_cuda_kernel.cu:31-95 all-pairs
// distance matrix in HBM, :107-170 per-column insertion sort) with one pass that never materialises
// the R x Q matrix: every lane owns QPL query points in registers, the reference points are
// wave-uniform (read through the scalar cache / broadcast), and the running arg-min lives in VGPRs.
//
// Bit-exactness contract (the oracle is oracle/knn_ref.c): the squared distance is accumulated in
// coordinate order as ssd = fma(t, t, ssd) starting from 0 -- what nvcc's default contraction makes of
// `ssd += tmp*tmp` (.cu:83-84) -- and a candidate replaces the best only on strict '<', so the lowest
// reference index wins ties (.cu:128,152).  Indices are written 1-based as int64 (.cu:122,141,165).
#include <cstdlib>

#include "common.h"
#include "knn_core.h"

namespace {

constexpr int KNN_BLOCK = 256;

#ifdef DF_DEV      // the LDS-staged round-2 kernel: kept for A/B runs of the dev build only (DF_KNN_VARIANT)
// dim == 3, k == 1.  grid = (ceil(Q / (KNN_BLOCK*QPL)), batch)
template <int QPL>
__global__ __launch_bounds__(KNN_BLOCK) void knn1_dim3_kernel(const float *__restrict__ ref, int R,
                                                              const float *__restrict__ query, int Q,
                                                              int64_t *__restrict__ ind) {
  extern __shared__ __attribute__((aligned(16))) float s_ref[];   // [R][4] = x,y,z,pad
  const int b = blockIdx.y;
  ref += (size_t)b * 3 * R;
  query += (size_t)b * 3 * Q;
  ind += (size_t)b * Q;

  for (int r = threadIdx.x; r < R; r += KNN_BLOCK) {
    float4 p = make_float4(ref[r], ref[R + r], ref[2 * R + r], 0.f);
    reinterpret_cast<float4 *>(s_ref)[r] = p;
  }
  __syncthreads();

  const int q0 = blockIdx.x * (KNN_BLOCK * QPL) + threadIdx.x;
  float qx[QPL], qy[QPL], qz[QPL];
  int bi[QPL];
#pragma unroll
  for (int j = 0; j < QPL; ++j) {
    int q = q0 + j * KNN_BLOCK;
    int qc = q < Q ? q : Q - 1;          // clamp: out-of-range lanes compute on a valid point, never store
    qx[j] = query[qc];
    qy[j] = query[Q + qc];
    qz[j] = query[2 * Q + qc];
  }

  df::knn1_scan<QPL>(s_ref, R, qx, qy, qz, bi);
  // the reference seeds the column with row 0 unconditionally (.cu:120-122); with best = +inf the first
  // finite distance wins, and an all-NaN column keeps index 0 -> 1-based 1, the same answer
#pragma unroll
  for (int j = 0; j < QPL; ++j) {
    int q = q0 + j * KNN_BLOCK;
    if (q < Q) ind[q] = (int64_t)bi[j] + 1;
  }
}

#endif

// The same scan with the reference points read through the SCALAR cache instead of LDS (s_load_dwordx8 of eight x, eight y, eight z:
// the planar [3][R] layout of the reference's API is exactly what a scalar load wants): no staging pass, no barrier, no LDS reads
// in the loop -- the packed subtractions take the coordinates straight from SGPR pairs.  Same operations in the same order as
// knn1_scan (bit-identical indices); the winning chunk is re-scanned from global memory (per-lane addresses).
// SPLIT > 1 (one launch that would leave the chip a handful of long-running waves per SIMD: 500 x 500 000 is 3.8 waves per SIMD, the
// 4-wave SIMDs set the time and nothing hides a wave's dependent chains): SPLIT waves share the same 64 x QPL queries, wave w scans
// the chunks w, w + SPLIT, ...; the partial winners meet in LDS and are combined by smaller distance, then lower index --
// the winner of the single ascending scan (as in knn1_dim3_rsplit_kernel).
template <int QPL, int SPLIT>
__global__ __launch_bounds__(KNN_BLOCK) void knn1_dim3_sgpr_kernel(const float *__restrict__ ref, int R, const float *__restrict__ query, int Q,
                                                                   int64_t *__restrict__ ind) {
  static_assert(QPL % 2 == 0, "queries are processed as packed pairs");
  static_assert(SPLIT == 1 || SPLIT == 2 || SPLIT == 4, "waves per query group");
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int CH = 8, GROUPS = KNN_BLOCK / 64 / SPLIT, QPB = GROUPS * 64 * QPL;      // query groups / queries per workgroup
  __shared__ float s_d[SPLIT > 1 ? KNN_BLOCK * QPL : 1];
  __shared__ int s_i[SPLIT > 1 ? KNN_BLOCK * QPL : 1];
  const int b = blockIdx.y;
  ref += (size_t)b * 3 * R;
  query += (size_t)b * 3 * Q;
  ind += (size_t)b * Q;
  const float *__restrict__ rx = ref, *__restrict__ ry = ref + R, *__restrict__ rz = ref + 2 * (size_t)R;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave / SPLIT, part = wave - grp * SPLIT;
  const int q0 = blockIdx.x * QPB + grp * 64 + lane;
  constexpr int QSTRIDE = GROUPS * 64;
  float qx[QPL], qy[QPL], qz[QPL], best[QPL];
  int bc[QPL], bi[QPL];
  f32x2 qx2[QPL / 2], qy2[QPL / 2], qz2[QPL / 2];
#pragma unroll
  for (int j = 0; j < QPL; ++j) {
    const int q = q0 + j * QSTRIDE, qc = q < Q ? q : Q - 1;
    qx[j] = query[qc]; qy[j] = query[Q + qc]; qz[j] = query[2 * (size_t)Q + qc];
    best[j] = __builtin_inff(); bc[j] = part; bi[j] = 0;
  }
#pragma unroll
  for (int j = 0; j < QPL / 2; ++j) {
    qx2[j] = f32x2{qx[2 * j], qx[2 * j + 1]};
    qy2[j] = f32x2{qy[2 * j], qy[2 * j + 1]};
    qz2[j] = f32x2{qz[2 * j], qz[2 * j + 1]};
  }
  const int nchunk = R / CH;
  // chunk c + 1 is requested before chunk c is multiplied (scalar loads return out of order: the wait in front of a chunk's first
  // use covers everything outstanding, so without the look-ahead every chunk would expose a scalar-cache round trip)
  float px[CH], py[CH], pz[CH];
  if (nchunk > part) {
#pragma unroll
    for (int i = 0; i < CH; ++i) { px[i] = rx[part * CH + i]; py[i] = ry[part * CH + i]; pz[i] = rz[part * CH + i]; }
  }
  for (int c = part; c < nchunk; c += SPLIT) {
    float nx[CH], ny[CH], nz[CH];
    const int cn = c + SPLIT < nchunk ? c + SPLIT : c;      // (the last round re-reads its own chunk: no branch in the loop)
#pragma unroll
    for (int i = 0; i < CH; ++i) {                 // uniform addresses, read-only data: scalar loads
      nx[i] = rx[cn * CH + i]; ny[i] = ry[cn * CH + i]; nz[i] = rz[cn * CH + i];
    }
    __builtin_amdgcn_sched_barrier(0);             // the requests go out BEFORE this round's arithmetic, not in the middle of it
#pragma unroll
    for (int j = 0; j < QPL / 2; ++j) {
      float m0 = __builtin_inff(), m1 = __builtin_inff();
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const f32x2 tx = f32x2{px[i], px[i]} - qx2[j];
        const f32x2 ty = f32x2{py[i], py[i]} - qy2[j];
        const f32x2 tz = f32x2{pz[i], pz[i]} - qz2[j];
        f32x2 d = tx * tx;
        d = __builtin_elementwise_fma(ty, ty, d);
        d = __builtin_elementwise_fma(tz, tz, d);
        m0 = __builtin_fminf(m0, d.x);
        m1 = __builtin_fminf(m1, d.y);
      }
      const bool lt0 = m0 < best[2 * j], lt1 = m1 < best[2 * j + 1];
      best[2 * j] = lt0 ? m0 : best[2 * j];
      bc[2 * j] = lt0 ? c : bc[2 * j];
      best[2 * j + 1] = lt1 ? m1 : best[2 * j + 1];
      bc[2 * j + 1] = lt1 ? c : bc[2 * j + 1];
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) { px[i] = nx[i]; py[i] = ny[i]; pz[i] = nz[i]; }
  }
#pragma unroll
  for (int j = 0; j < QPL; ++j) {
    float b2 = __builtin_inff();
    int i2 = 0;
    const int base = bc[j] * CH;
    if (nchunk > part) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const float tx = rx[base + i] - qx[j], ty = ry[base + i] - qy[j], tz = rz[base + i] - qz[j];
        float d = tx * tx;
        d = __builtin_fmaf(ty, ty, d);
        d = __builtin_fmaf(tz, tz, d);
        const bool lt = d < b2;
        b2 = lt ? d : b2;
        i2 = lt ? i : i2;
      }
    }
    bi[j] = b2 < __builtin_inff() ? base + i2 : (SPLIT > 1 ? 0x7fffffff : 0);
    best[j] = b2;
  }
  for (int r = nchunk * CH; r < R && part == SPLIT - 1; ++r) {                         // R % CH tail: plain scan (by the last part)
    const float x = rx[r], y = ry[r], z = rz[r];
#pragma unroll
    for (int j = 0; j < QPL; ++j) {
      const float tx = x - qx[j], ty = y - qy[j], tz = z - qz[j];
      float d = tx * tx;
      d = __builtin_fmaf(ty, ty, d);
      d = __builtin_fmaf(tz, tz, d);
      const bool lt = d < best[j];
      best[j] = lt ? d : best[j];
      bi[j] = lt ? r : bi[j];
    }
  }
  if constexpr (SPLIT > 1) {
#pragma unroll
    for (int j = 0; j < QPL; ++j) { s_d[(wave * QPL + j) * 64 + lane] = best[j]; s_i[(wave * QPL + j) * 64 + lane] = bi[j]; }
    __syncthreads();
    if (part != 0) return;
#pragma unroll
    for (int j = 0; j < QPL; ++j) {
#pragma unroll
      for (int w = 1; w < SPLIT; ++w) {
        const float d = s_d[((wave + w) * QPL + j) * 64 + lane];
        const int i = s_i[((wave + w) * QPL + j) * 64 + lane];
        const bool take = d < best[j] || (d == best[j] && i < bi[j]);
        best[j] = take ? d : best[j];
        bi[j] = take ? i : bi[j];
      }
      if (bi[j] == 0x7fffffff) bi[j] = 0;        // all-NaN column: row 0, like the reference's seed (.cu:120-122)
    }
  }
#pragma unroll
  for (int j = 0; j < QPL; ++j) {
    const int q = q0 + j * QSTRIDE;
    if (q < Q) ind[q] = (int64_t)bi[j] + 1;
  }
}

// dim == 3, k == 1, FEW queries: the query set alone cannot fill the chip, so the reference points are split
// over the lanes as well.  A workgroup owns 64 queries x all R references: wave w scans the slice
// r = w, w + WAVES, ... (ascending), every lane keeps its running arg-min, and the WAVES partial results per query
// meet in LDS where lane order = slice order; the combine keeps the smaller distance and, on equal distances, the
// lower reference index -- exactly the winner a single ascending scan with strict '<' would have kept.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void knn1_dim3_rsplit_kernel(const float *__restrict__ ref, int R,
                                                                     const float *__restrict__ query, int Q,
                                                                     int64_t *__restrict__ ind) {
  extern __shared__ __attribute__((aligned(16))) float s_ref[];   // [R][4], then WAVES*64 (dist, idx) pairs
  const int b = blockIdx.y;
  ref += (size_t)b * 3 * R;
  query += (size_t)b * 3 * Q;
  ind += (size_t)b * Q;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int r = tid; r < R; r += WAVES * 64)
    reinterpret_cast<float4 *>(s_ref)[r] = make_float4(ref[r], ref[R + r], ref[2 * R + r], 0.f);
  float *s_d = s_ref + (size_t)R * 4;
  int *s_i = reinterpret_cast<int *>(s_d + WAVES * 64);
  __syncthreads();
  const int q = blockIdx.x * 64 + lane;
  const int qc = q < Q ? q : Q - 1;
  const float qx = query[qc], qy = query[Q + qc], qz = query[2 * Q + qc];
  float best = __builtin_inff();
  int bi = 0x7fffffff;
  for (int r = wave; r < R; r += WAVES) {
    const float4 p = reinterpret_cast<const float4 *>(s_ref)[r];
    const float tx = p.x - qx, ty = p.y - qy, tz = p.z - qz;
    float d = tx * tx;
    d = __builtin_fmaf(ty, ty, d);
    d = __builtin_fmaf(tz, tz, d);
    const bool lt = d < best;
    best = lt ? d : best;
    bi = lt ? r : bi;
  }
  s_d[wave * 64 + lane] = best;
  s_i[wave * 64 + lane] = bi;
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 1; w < WAVES; ++w) {
      const float d = s_d[w * 64 + lane];
      const int i = s_i[w * 64 + lane];
      const bool take = d < best || (d == best && i < bi);
      best = take ? d : best;
      bi = take ? i : bi;
    }
    if (bi == 0x7fffffff) bi = 0;          // all-NaN column: row 0, like the reference's seed (.cu:120-122)
    if (q < Q) ind[q] = (int64_t)bi + 1;
  }
}

// Generic dim / k: one lane per query, sorted top-k kept in registers/scratch (stable: ties keep the
// lower index first, which is what the reference's insertion with strict comparisons produces).
template <int KMAX>
__global__ __launch_bounds__(KNN_BLOCK) void knn_generic_kernel(const float *__restrict__ ref, int R,
                                                                const float *__restrict__ query, int Q, int dim,
                                                                int k, int64_t *__restrict__ ind) {
  const int b = blockIdx.y;
  ref += (size_t)b * dim * R;
  query += (size_t)b * dim * Q;
  ind += (size_t)b * k * Q;
  const int q = blockIdx.x * KNN_BLOCK + threadIdx.x;
  if (q >= Q) return;
  float bd[KMAX];
  int bidx[KMAX];
  int filled = 0;
  for (int r = 0; r < R; ++r) {
    float ssd = 0.f;
    for (int d = 0; d < dim; ++d) {
      float t = ref[(size_t)d * R + r] - query[(size_t)d * Q + q];
      ssd = __builtin_fmaf(t, t, ssd);
    }
    if (filled < k) {
      // part 1 of cuInsertionSort: place among the first `filled` entries
      int i = filled;
      if (filled > 0 && ssd < bd[filled - 1]) {
        i = filled - 1;
        for (int a = 0; a < filled - 1; ++a)
          if (bd[a] > ssd) { i = a; break; }
      }
      for (int j = filled; j > i; --j) { bd[j] = bd[j - 1]; bidx[j] = bidx[j - 1]; }
      bd[i] = ssd;
      bidx[i] = r;
      ++filled;
    } else if (ssd < bd[k - 1]) {
      int i = k - 1;
      for (int a = 0; a < k - 1; ++a)
        if (bd[a] > ssd) { i = a; break; }
      for (int j = k - 1; j > i; --j) { bd[j] = bd[j - 1]; bidx[j] = bidx[j - 1]; }
      bd[i] = ssd;
      bidx[i] = r;
    }
  }
  for (int j = 0; j < k; ++j) ind[(size_t)j * Q + q] = (int64_t)bidx[j] + 1;
}

int launch_knn(const float *ref, const float *query, int64_t *idx, int batch, int dim, int R, int Q, int k,
               hipStream_t st) {
  if (batch < 0 || dim <= 0 || R <= 0 || Q < 0) return df::set_error(DF_ERR_ARG, "knn: bad sizes");
  if (k <= 0 || k > R || k > DF_KNN_MAX_K)
    return df::set_error(DF_ERR_ARG, "knn: k=%d must be in [1, min(ref_nb=%d, %d)]", k, R, DF_KNN_MAX_K);
  if (batch == 0 || Q == 0) return DF_OK;      // empty query set: nothing to write
  if (!ref || !query || !idx) return df::set_error(DF_ERR_ARG, "knn: null pointer");
  if (batch > 65535) return df::set_error(DF_ERR_ARG, "knn: batch > 65535");
  if (dim == 3 && k == 1 && (size_t)R * 16 <= 56 * 1024 && (long)Q * batch <= 64L * 1024 && R >= 32) {
    constexpr int WAVES = 16;     // 1024 threads: 64 queries x 16 reference slices per workgroup
    dim3 grid(df::cdiv(Q, 64), batch);
    hipLaunchKernelGGL(knn1_dim3_rsplit_kernel<WAVES>, grid, dim3(WAVES * 64), (size_t)R * 16 + WAVES * 64 * 8, st, ref, R, query, Q, idx);
  } else if (dim == 3 && k == 1) {
    // default: reference points through the scalar cache, 2 queries per lane (measured on 500 x 500 000 / 64 x 500 x 1 000 000:
    // 41.8 us / 3.46 ms against 48.3 us / 3.86 ms for the LDS-staged kernel; 4 queries per lane 45.8 us / 3.53 ms).
    // DF_KNN_VARIANT (dev switch, A/B runs): 2 = scalar cache, 4 per lane; 3 / 4 = LDS-staged, 4 / 2 per lane (R <= 4096)
#ifdef DF_DEV
    static const int variant = df::dev_getenv("DF_KNN_VARIANT") ? atoi(df::dev_getenv("DF_KNN_VARIANT")) : 0;
    const bool lds_ok = (size_t)R * 16 <= 64 * 1024;
    if (variant == 3 && lds_ok) {
      hipLaunchKernelGGL(knn1_dim3_kernel<4>, dim3(df::cdiv(Q, KNN_BLOCK * 4), batch), dim3(KNN_BLOCK), (size_t)R * 16, st, ref, R, query, Q, idx);
    } else if (variant == 4 && lds_ok) {
      hipLaunchKernelGGL(knn1_dim3_kernel<2>, dim3(df::cdiv(Q, KNN_BLOCK * 2), batch), dim3(KNN_BLOCK), (size_t)R * 16, st, ref, R, query, Q, idx);
    } else if (variant == 2) {
      hipLaunchKernelGGL((knn1_dim3_sgpr_kernel<4, 1>), dim3(df::cdiv(Q, KNN_BLOCK * 4), batch), dim3(KNN_BLOCK), 0, st, ref, R, query, Q, idx);
    } else
#else
    constexpr int variant = 0;
#endif
    {
      // waves the launch would give every SIMD with one wave per 128 queries; below ~2 rounds of 8 the references are split too
      const double wps = (double)Q * batch / 128.0 / 1024.0;
      const int split = variant == 5 ? 1 : variant == 6 ? 2 : variant == 7 ? 4 : (wps < 6.0 && R >= 64 ? (wps < 3.0 ? 4 : 2) : 1);
      if (split == 4) hipLaunchKernelGGL((knn1_dim3_sgpr_kernel<2, 4>), dim3(df::cdiv(Q, 128), batch), dim3(KNN_BLOCK), 0, st, ref, R, query, Q, idx);
      else if (split == 2) hipLaunchKernelGGL((knn1_dim3_sgpr_kernel<2, 2>), dim3(df::cdiv(Q, 256), batch), dim3(KNN_BLOCK), 0, st, ref, R, query, Q, idx);
      else hipLaunchKernelGGL((knn1_dim3_sgpr_kernel<2, 1>), dim3(df::cdiv(Q, 512), batch), dim3(KNN_BLOCK), 0, st, ref, R, query, Q, idx);
    }
  } else {
    dim3 grid(df::cdiv(Q, KNN_BLOCK), batch);
    hipLaunchKernelGGL(knn_generic_kernel<DF_KNN_MAX_K>, grid, dim3(KNN_BLOCK), 0, st, ref, R, query, Q, dim, k, idx);
  }
  return df::check_launch("knn");
}

}  // namespace

extern "C" int df_knn_device(const float *ref_dev, int ref_nb, const float *query_dev, int query_nb, int dim, int k,
                             int64_t *ind_dev, df_stream_t stream) {
  return launch_knn(ref_dev, query_dev, ind_dev, 1, dim, ref_nb, query_nb, k, df::to_stream(stream));
}

extern "C" int df_knn(const float *ref, const float *query, int64_t *idx, int batch, int dim, int ref_nb,
                      int query_nb, int k, df_stream_t stream) {
  return launch_knn(ref, query, idx, batch, dim, ref_nb, query_nb, k, df::to_stream(stream));
}

// Exact-signature drop-in for the reference's native entry point (lib/knn/src/knn_cuda_kernel.h:14-16, called at
// lib/knn/src/knn_pytorch.c:35): same name, same argument list, void return.  `dist_dev` (the reference's ref_nb x query_nb
// scratch, knn_pytorch.c:31) is IGNORED -- distances never leave registers here -- so a caller may pass NULL and skip the
// allocation.  Errors cannot be returned through a void function: like the reference (whose wrapper polls cudaGetLastError,
// knn_pytorch.c:41-45) the failure is left for the caller to find, here in df_last_error() / hipGetLastError().
extern "C" void knn_device(float *ref_dev, int ref_width, float *query_dev, int query_width, int height, int k, float *dist_dev, long *ind_dev,
                           df_stream_t stream) {
  (void)dist_dev;
  static_assert(sizeof(long) == sizeof(int64_t), "the reference's `long` indices are 64-bit on this ABI");
  (void)launch_knn(ref_dev, query_dev, reinterpret_cast<int64_t *>(ind_dev), 1, height, ref_width, query_width, k, df::to_stream(stream));
}

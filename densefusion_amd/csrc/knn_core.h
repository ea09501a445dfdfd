// The 1-NN scan shared by csrc/knn.hip (df_knn) and csrc/loss.hip (the symmetric ADD-S loss): one implementation, so
// the loss picks bit for bit the neighbour df_knn would return for the same points.
#pragma once
#include <hip/hip_runtime.h>

namespace df {

// s_ref: R reference points in LDS as [R][4] (x, y, z, pad); every lane owns QPL query points (qx, qy, qz);
// out: bi[j] = index of the nearest reference (squared distance accumulated as fma(t, t, ssd) in coordinate order from
// t = ref - query, strict '<': the lowest index wins ties; 0 when every distance is NaN).
template <int QPL>
__device__ __forceinline__ void knn1_scan(const float *s_ref, int R, const float (&qx)[QPL], const float (&qy)[QPL],
                                          const float (&qz)[QPL], int (&bi)[QPL]) {
  float best[QPL];
#pragma unroll
  for (int j = 0; j < QPL; ++j) { best[j] = __builtin_inff(); bi[j] = 0; }
  // Chunked arg-min.  The plain scan costs 3 packed-math + 3 select instructions per pair (compare, keep distance,
  // keep index); here the index bookkeeping is paid once per CH references: within a chunk only the running minimum
  // is kept (v_min3: 0.5 instruction per pair), the chunk number is recorded when the chunk minimum is STRICTLY below
  // the best so far, and the winning chunk is re-scanned at the end to recover the reference index.  Same winner as
  // the sequential strict-'<' scan: the first chunk that attains the global minimum holds its first occurrence, and
  // the re-scan (same operations, same bits) takes the first element of that chunk that attains it.
  // NaN distances: fminf drops them and they never satisfy '<' -- same as the reference's comparison.
  constexpr int CH = 8;
  int bc[QPL];
#pragma unroll
  for (int j = 0; j < QPL; ++j) bc[j] = 0;
  const int nchunk = R / CH;
  static_assert(QPL % 2 == 0, "queries are processed as packed pairs");
  typedef float f32x2 __attribute__((ext_vector_type(2)));      // two queries per packed-math instruction (v_pk_*_f32)
  f32x2 qx2[QPL / 2], qy2[QPL / 2], qz2[QPL / 2];
#pragma unroll
  for (int j = 0; j < QPL / 2; ++j) {
    qx2[j] = f32x2{qx[2 * j], qx[2 * j + 1]};
    qy2[j] = f32x2{qy[2 * j], qy[2 * j + 1]};
    qz2[j] = f32x2{qz[2 * j], qz[2 * j + 1]};
  }
  for (int c = 0; c < nchunk; ++c) {
    float4 p[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) p[i] = reinterpret_cast<const float4 *>(s_ref)[c * CH + i];   // wave-uniform: LDS broadcast
#pragma unroll
    for (int j = 0; j < QPL / 2; ++j) {
      float m0 = __builtin_inff(), m1 = __builtin_inff();
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const f32x2 tx = f32x2{p[i].x, p[i].x} - qx2[j];
        const f32x2 ty = f32x2{p[i].y, p[i].y} - qy2[j];
        const f32x2 tz = f32x2{p[i].z, p[i].z} - qz2[j];
        f32x2 d = tx * tx;               // == fma(tx, tx, 0)
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
  }
#pragma unroll
  for (int j = 0; j < QPL; ++j) {
    float b2 = __builtin_inff();
    int i2 = 0;
    const int base = bc[j] * CH;
    if (nchunk > 0) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const float4 pp = reinterpret_cast<const float4 *>(s_ref)[base + i];      // per-lane address, CH reads per query
        const float tx = pp.x - qx[j], ty = pp.y - qy[j], tz = pp.z - qz[j];
        float d = tx * tx;
        d = __builtin_fmaf(ty, ty, d);
        d = __builtin_fmaf(tz, tz, d);
        const bool lt = d < b2;
        b2 = lt ? d : b2;
        i2 = lt ? i : i2;
      }
    }
    bi[j] = b2 < __builtin_inff() ? base + i2 : 0;
  }
  for (int r = nchunk * CH; r < R; ++r) {                                             // R % CH tail: plain scan
    const float4 pp = reinterpret_cast<const float4 *>(s_ref)[r];
#pragma unroll
    for (int j = 0; j < QPL; ++j) {
      const float tx = pp.x - qx[j], ty = pp.y - qy[j], tz = pp.z - qz[j];
      float d = tx * tx;
      d = __builtin_fmaf(ty, ty, d);
      d = __builtin_fmaf(tz, tz, d);
      const bool lt = d < best[j];
      best[j] = lt ? d : best[j];
      bi[j] = lt ? r : bi[j];
    }
  }
}

// The same scan with the reference points read through the SCALAR cache instead of LDS (the scheme of knn.hip's
// knn1_dim3_sgpr_kernel), for references stored as [R][3] (x, y, z interleaved: the loss's target points): a chunk of 8 references is 24
// consecutive floats = three s_load_dwordx8 of a wave-uniform address, requested one chunk ahead; no staging pass, no barrier, no LDS
// read in the loop.  Same operations in the same order as knn1_scan: the same index, bit for bit.  `tgt` must be wave-uniform.
template <int QPL>
__device__ __forceinline__ void knn1_scan_sc(const float *__restrict__ tgt, int R, const float (&qx)[QPL], const float (&qy)[QPL],
                                             const float (&qz)[QPL], int (&bi)[QPL]) {
  static_assert(QPL % 2 == 0, "queries are processed as packed pairs");
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int CH = 8;
  float best[QPL];
  int bc[QPL];
  f32x2 qx2[QPL / 2], qy2[QPL / 2], qz2[QPL / 2];
#pragma unroll
  for (int j = 0; j < QPL; ++j) { best[j] = __builtin_inff(); bi[j] = 0; bc[j] = 0; }
#pragma unroll
  for (int j = 0; j < QPL / 2; ++j) {
    qx2[j] = f32x2{qx[2 * j], qx[2 * j + 1]};
    qy2[j] = f32x2{qy[2 * j], qy[2 * j + 1]};
    qz2[j] = f32x2{qz[2 * j], qz[2 * j + 1]};
  }
  const int nchunk = R / CH;
  float p[CH * 3];
  if (nchunk > 0) {
#pragma unroll
    for (int i = 0; i < CH * 3; ++i) p[i] = tgt[i];
  }
  for (int c = 0; c < nchunk; ++c) {
    float nx[CH * 3];
    const int cn = c + 1 < nchunk ? c + 1 : c;      // (the last round re-reads its own chunk: no branch in the loop)
#pragma unroll
    for (int i = 0; i < CH * 3; ++i) nx[i] = tgt[cn * (CH * 3) + i];      // uniform address, read-only data: scalar loads
    __builtin_amdgcn_sched_barrier(0);               // the requests go out BEFORE this round's arithmetic
#pragma unroll
    for (int j = 0; j < QPL / 2; ++j) {
      float m0 = __builtin_inff(), m1 = __builtin_inff();
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const f32x2 tx = f32x2{p[3 * i], p[3 * i]} - qx2[j];
        const f32x2 ty = f32x2{p[3 * i + 1], p[3 * i + 1]} - qy2[j];
        const f32x2 tz = f32x2{p[3 * i + 2], p[3 * i + 2]} - qz2[j];
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
    for (int i = 0; i < CH * 3; ++i) p[i] = nx[i];
  }
#pragma unroll
  for (int j = 0; j < QPL; ++j) {
    float b2 = __builtin_inff();
    int i2 = 0;
    const int base = bc[j] * CH;
    if (nchunk > 0) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {                 // the winning chunk again, per-lane addresses
        const float tx = tgt[(base + i) * 3] - qx[j], ty = tgt[(base + i) * 3 + 1] - qy[j], tz = tgt[(base + i) * 3 + 2] - qz[j];
        float d = tx * tx;
        d = __builtin_fmaf(ty, ty, d);
        d = __builtin_fmaf(tz, tz, d);
        const bool lt = d < b2;
        b2 = lt ? d : b2;
        i2 = lt ? i : i2;
      }
    }
    bi[j] = b2 < __builtin_inff() ? base + i2 : 0;
    best[j] = b2;
  }
  for (int r = nchunk * CH; r < R; ++r) {            // R % CH tail: plain scan
    const float x = tgt[r * 3], y = tgt[r * 3 + 1], z = tgt[r * 3 + 2];
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
}

}  // namespace df

/* Oracle: plain-C restatement of the reference's brute-force k-NN.  TEST INFRASTRUCTURE ONLY
 * (see oracle/__init__.py) -- never linked into the product library.
 *
 * Follows lib/knn/src/knn_cuda_kernel.cu:
 *   :31-95   cuComputeDistanceGlobal  dist[r][q] = sum_d (ref[d][r] - query[d][q])^2, fp32,
 *            accumulated in d order as `ssd += tmp*tmp` (nvcc default -fmad=true contracts this to
 *            one fused multiply-add per coordinate; the zero-padded tile slots add exact zeros)
 *   :107-170 cuInsertionSort          per query column, top-k by insertion over rows in order,
 *            strict '<' so the lowest row wins ties, indices written 1-BASED
 * and the batch loop of lib/knn/src/knn_pytorch.c:21-36 (ref [B,dim,R], query [B,dim,Q], idx [B,k,Q]).
 *
 * Parity status: the CUDA op cannot be built in this image (nvcc/THC absent); this restatement is
 * pinned through lib/nn.py nn_distance on small sizes (tests/test_oracle_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float pair_dist(const float *ref, int R, int r, const float *query, int Q, int q, int dim) {
  float ssd = 0.0f;
  for (int d = 0; d < dim; ++d) {
    float tmp = ref[(size_t)d * R + r] - query[(size_t)d * Q + q];
    ssd = fmaf(tmp, tmp, ssd);
  }
  return ssd;
}

/* one query column: the insertion sort of knn_cuda_kernel.cu:107-170 on a private distance column */
static void column_topk(float *p_dist, int64_t *p_ind, int height, int k) {
  float max_dist = p_dist[0];
  p_ind[0] = 1;
  for (int l = 1; l < k; ++l) {              /* part 1: sort the first k elements */
    float curr = p_dist[l];
    if (curr < max_dist) {
      int i = l - 1;
      for (int a = 0; a < l - 1; ++a)
        if (p_dist[a] > curr) { i = a; break; }
      for (int j = l; j > i; --j) { p_dist[j] = p_dist[j - 1]; p_ind[j] = p_ind[j - 1]; }
      p_dist[i] = curr;
      p_ind[i] = l + 1;
    } else {
      p_ind[l] = l + 1;
    }
    max_dist = p_dist[l];
  }
  for (int l = k; l < height; ++l) {         /* part 2: insert the remaining rows */
    float curr = p_dist[l];
    if (curr < max_dist) {
      int i = k - 1;
      for (int a = 0; a < k - 1; ++a)
        if (p_dist[a] > curr) { i = a; break; }
      for (int j = k - 1; j > i; --j) { p_dist[j] = p_dist[j - 1]; p_ind[j] = p_ind[j - 1]; }
      p_dist[i] = curr;
      p_ind[i] = l + 1;
      max_dist = p_dist[k - 1];
    }
  }
}

/* ref [B][dim][R], query [B][dim][Q] fp32; idx [B][k][Q] int64 (1-based). returns 0, or -1 on bad args */
int oracle_knn(const float *ref, const float *query, int64_t *idx, int B, int dim, int R, int Q, int k,
               int threads) {
  if (B < 0 || dim <= 0 || R <= 0 || Q < 0 || k <= 0 || k > R) return -1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
  for (int b = 0; b < B; ++b) {
    const float *rb = ref + (size_t)b * dim * R;
    const float *qb = query + (size_t)b * dim * Q;
    int64_t *ib = idx + (size_t)b * k * Q;
#pragma omp parallel
    {
      float *col = (float *)malloc(sizeof(float) * (size_t)R);
      int64_t *ind = (int64_t *)malloc(sizeof(int64_t) * (size_t)R);
#pragma omp for schedule(static)
      for (int q = 0; q < Q; ++q) {
        if (k == 1) {                          /* same result as the general path, without the column */
          float best = pair_dist(rb, R, 0, qb, Q, q, dim);
          int64_t bi = 1;
          for (int r = 1; r < R; ++r) {
            float d = pair_dist(rb, R, r, qb, Q, q, dim);
            if (d < best) { best = d; bi = r + 1; }
          }
          ib[q] = bi;
        } else {
          for (int r = 0; r < R; ++r) col[r] = pair_dist(rb, R, r, qb, Q, q, dim);
          column_topk(col, ind, R, k);
          for (int j = 0; j < k; ++j) ib[(size_t)j * Q + q] = ind[j];
        }
      }
      free(col);
      free(ind);
    }
  }
  return 0;
}

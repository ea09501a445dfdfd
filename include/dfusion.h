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

#ifdef __cplusplus
}
#endif
#endif /* DFUSION_H_ */

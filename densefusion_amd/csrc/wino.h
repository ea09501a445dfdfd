// Winograd F(2x2, 3x3) / F(4x4, 3x3) companion kernels of the batched GEMM (wino.hip).
#pragma once
#include "igemm.h"

namespace df {

constexpr int WINO_MAXB = 16;
// per-bucket table of a multi-bucket transform launch (kernel argument, by value)
struct WinoTab { int n; int H[WINO_MAXB], W[WINO_MAXB], TH[WINO_MAXB], TW[WINO_MAXB], blocks[WINO_MAXB + 1]; long T[WINO_MAXB], row0[WINO_MAXB], t0[WINO_MAXB]; };
struct WinoGeom { int TH, TW; long T; };                 // tiles per sub-lattice (rows, columns), tiles in total
// m = 2: F(2x2,3x3) (16 planes), m = 4: F(4x4,3x3) (36 planes)
WinoGeom wino_geom(int B, int H, int W, int dil, int m = 2);
// layer-geometry-only decision (never batch dependent): 0 = direct implicit GEMM, 2 / 4 = the transform-domain product with that tile
int wino_route(int H, int W, int dil, int Cin, int Cout);
void launch_wino_weight(const float *w_packed /*[O][3][3][C]*/, float *U /*[(m+2)^2][O][C]*/, int O, int C, hipStream_t st, int m = 2);
// F(4x4,3x3) weight transforms of up to WINO_WMAX tensors in one launch: segment g = O[g] x C[g] packed 3x3 kernels at
// (from_b[g] ? src_b : src_a) + src_off[g] -> dst + dst_off[g]; e0 = prefix of O * C
constexpr int WINO_WMAX = 32;
struct WinoWTab { int n; int O[WINO_WMAX], C[WINO_WMAX], from_b[WINO_WMAX]; long src_off[WINO_WMAX], dst_off[WINO_WMAX], e0[WINO_WMAX + 1]; };
void launch_wino4_weight_multi(const float *src_a, const float *src_b, float *dst, const WinoWTab &tab, hipStream_t st);
// Ttot / t0: the tiles of this call are rows [t0, t0 + T) of planes that hold Ttot tiles each (several crop-size buckets share
// one Winograd-domain GEMM); Ttot = 0 means the call owns the planes (Ttot = T, t0 = 0)
void launch_wino_input(const float *x, int in_ld, int in_coff, float *V /*[(m+2)^2][Ttot][C]*/, int B, int H, int W, int C, int dil,
                       hipStream_t st, long Ttot = 0, long t0 = 0, int m = 2);
void launch_wino_output(const float *M /*[(m+2)^2][Ttot][C]*/, float *out, int out_ld, int out_coff, const float *bias, const float *res,
                        int res_ld, int res_coff, int act, int B, int H, int W, int C, int dil, hipStream_t st, long Ttot = 0, long t0 = 0,
                        int m = 2);

// F(4x4,3x3) transforms of nb crop-size buckets in one launch per 16 buckets (B / H / W / first pixel row / first tile per bucket)
void launch_wino4_input_multi(const float *x, int in_ld, float *V, int nb, const int *B, const int *H, const int *W, const long *row0, const long *t0,
                              int C, int dil, long Ttot, hipStream_t st);
void launch_wino4_output_multi(const float *M, float *out, int out_ld, const float *res, int res_ld, int act, int nb, const int *B, const int *H,
                               const int *W, const long *row0, const long *t0, int C, int dil, long Ttot, hipStream_t st);

}  // namespace df

// Winograd F(2x2, 3x3) companion kernels of the batched GEMM (wino.hip).
#pragma once
#include "igemm.h"

namespace df {

struct WinoGeom { int TH, TW; long T; };                 // tiles per sub-lattice (rows, columns), tiles in total
WinoGeom wino_geom(int B, int H, int W, int dil);
// layer-geometry-only decision (never batch dependent): does the transform-domain product beat the direct sum?
bool wino_pays(int H, int W, int dil, int Cin, int Cout);
void launch_wino_weight(const float *w_packed /*[O][3][3][C]*/, float *U /*[16][O][C]*/, int O, int C, hipStream_t st);
// Ttot / t0: the tiles of this call are rows [t0, t0 + T) of planes that hold Ttot tiles each (several crop-size buckets share
// one Winograd-domain GEMM); Ttot = 0 means the call owns the planes (Ttot = T, t0 = 0)
void launch_wino_input(const float *x, int in_ld, int in_coff, float *V /*[16][Ttot][C]*/, int B, int H, int W, int C, int dil, hipStream_t st,
                       long Ttot = 0, long t0 = 0);
void launch_wino_output(const float *M /*[16][Ttot][C]*/, float *out, int out_ld, int out_coff, const float *bias, const float *res, int res_ld,
                        int res_coff, int act, int B, int H, int W, int C, int dil, hipStream_t st, long Ttot = 0, long t0 = 0);

}  // namespace df

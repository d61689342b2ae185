// Arguments of the fused BiFPN node kernel (csrc/bifpn_node.hip).
#pragma once
#include "jh_common.h"

namespace jh {

struct NodeArgs {
  const float* in[3];
  const double* st[3];     // statistics of in[i] (sum, sumsq per (n,c)) or nullptr
  float inv_cnt[3];        // 1 / pixels the statistics were taken over
  int mode[3];
  float w[3];
  int n_in, act;
  const float* dw;         // [9][Cp]
  const float* pw;         // packed [Cp/8][cout_p16/16][64][2]
  const float* bias;       // [cout_p16]
  float* y;                // raw output [N][H][W][cout_p]
  double* stats;           // [N][cout_p][2]
  int N, H, W, Cp, cout_p, cout_p16, cf;
  int blds = 0;            // pointwise weights staged in LDS behind the operand tile
  int alias = 0;           // operand tile written over the halo tile (single channel chunk)
  int abl = 0;             // ablation bits for timing experiments (0 in production)   // cf = channels per halo chunk (multiple of 4)
};


}  // namespace jh

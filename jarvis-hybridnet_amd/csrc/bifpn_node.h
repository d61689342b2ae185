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
  float* y_pool = nullptr; // row-streaming form, two inputs: also the 2x2-max-pooled raw output [N][H/2][W/2][cout_p]
  int rows = -1;           // row-streaming form (bifpn_rows.hip): 1 wherever the shape allows, 0 never, -1 by launch size,
                           // 2 = the workgroup form with 8-row segments (wide pyramids, time batches below 8)
  int abl = 0;             // ablation bits for timing experiments (0 in production)   // cf = channels per halo chunk (multiple of 4)
};


// csrc/bifpn_rows.hip: will this node take the row-streaming form?  (host; also used when a plan is built)
bool bifpn_rows_eligible(const NodeArgs& a);
bool bifpn_rows_ragged88(const NodeArgs& a);    // 88 channels, class >= 8, width no multiple of 16: workgroup form

#if defined(__HIPCC__)
constexpr int kNodeTY = 8, kNodeTX = 16, kNodePY = 10, kNodePX = 18, kNodeNRG = 4;

// Fetch of one channel quad of input k at output pixel (oy, ox) of the node: `b` is the
// (uniform) base of image n of that input at ITS resolution, offsets are 32-bit.
__device__ __forceinline__ float4 node_fetch_n(const float* b, int mode, int oy, int ox, int H, int W,
                                               int Cp, int c) {
  if (mode == FUSE_SAME) return *reinterpret_cast<const float4*>(b + ((oy * W + ox) * Cp + c));
  if (mode == FUSE_UP2)
    return *reinterpret_cast<const float4*>(b + (((oy >> 1) * (W >> 1) + (ox >> 1)) * Cp + c));
  if (mode == FUSE_UP4)
    return *reinterpret_cast<const float4*>(b + (((oy >> 2) * (W >> 2) + (ox >> 2)) * Cp + c));
  const int w = W * 2;                // FUSE_POOL2 (max commutes with the monotone IN map)
  const float* s = b + ((oy * 2 * w + ox * 2) * Cp + c);
  const float4 a0 = *reinterpret_cast<const float4*>(s);
  const float4 a1 = *reinterpret_cast<const float4*>(s + Cp);
  const float4 a2 = *reinterpret_cast<const float4*>(s + w * Cp);
  const float4 a3 = *reinterpret_cast<const float4*>(s + w * Cp + Cp);
  return make_float4(fmaxf(fmaxf(a0.x, a1.x), fmaxf(a2.x, a3.x)), fmaxf(fmaxf(a0.y, a1.y), fmaxf(a2.y, a3.y)),
                     fmaxf(fmaxf(a0.z, a1.z), fmaxf(a2.z, a3.z)), fmaxf(fmaxf(a0.w, a1.w), fmaxf(a2.w, a3.w)));
}
__device__ __forceinline__ size_t node_plane(int mode, int H, int W) {
  if (mode == FUSE_SAME) return (size_t)H * W;
  if (mode == FUSE_UP2) return (size_t)(H >> 1) * (W >> 1);
  if (mode == FUSE_UP4) return (size_t)(H >> 2) * (W >> 2);
  return (size_t)H * W * 4;
}

__device__ __forceinline__ float node_act(float v, int act) {
  // SiLU with the hardware exp2 / reciprocal (about 1e-7 relative error; the prologue is
  // instruction-bound on the IEEE expf + division otherwise)
  if (act == ACT_SILU) return silu_fast(v);
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

#endif

}  // namespace jh

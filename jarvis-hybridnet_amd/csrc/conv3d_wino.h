// Shared declarations of the Winograd 3D convolution kernels (conv3d_wino.hip: one role per
// workgroup, one tile each; conv3d_wino_pw.hip: persistent, wave-specialised).
#pragma once
#include "jh_common.h"

namespace jh {

struct WinoArgs {
  const float* x;          // [N][D][H][W][cin_p]
  float* y;                // [N][D][H][W][cout_p] raw output
  const float* u;          // transformed weights [16 f][3 dz][cin_p/8][cout_p16/16][64][2]
  const float* bias;       // [cout_p16] or nullptr
  const double* in_stats;  // InstanceNorm (+ in_act) of the input applied on load, or nullptr
  float in_inv;
  int in_act;
  double* stats;           // [N][cout_p][2] or nullptr
  int N, D, H, W, cin_p, cout_p, cout_p16;
  int abl;                 // experiment knob (JH_WS_ABL), 0 in production
  long long* dbg;          // per-phase cycle sums of workgroup 0 (JH_WINO_DBG), nullptr in production
};

constexpr int kWTY = 8, kWTX = 8;                           // (y, x) outputs per workgroup
constexpr int kWPY = kWTY + 2, kWPX = kWTX + 2;

// persistent form; returns -1 when the launch should fall back to the one-role kernel
int launch_conv3d_wino_pw(const WinoArgs& a, int nr, hipStream_t s);

}  // namespace jh

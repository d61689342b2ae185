// Per-pixel arithmetic of the frame pre-processing (jarvis/prediction/jarvis3D.py:143-145 resize +
// normalise, :168-178 crop + normalise), shared by the stand-alone kernels (geometry.hip) and the stem
// convolution that applies it while staging its input patch (stem.hip): ONE definition, so the fused
// and the unfused path produce the same bits.
#pragma once
#include "jh_common.h"

namespace jh {

// Two frame formats: SRC = 0: [N][3][H][W] fp32 RGB in [0,1] (the API input of
// JarvisPredictor3D.forward); SRC = 1: [N][H][W][3] uint8 BGR as the video decoder
// delivers it, converted like predict3D.py:79-80 (`.float()...[:, [2,1,0]] / 255.`) on
// the fly, so the 4x larger fp32 frame never exists.
template <int SRC>
__device__ __forceinline__ float frame_px(const void* frames, size_t n, int c, int y, int x, int H,
                                          int W) {
  if (SRC == 0)
    return static_cast<const float*>(frames)[((n * 3 + c) * H + y) * W + x];
  const unsigned char* p = static_cast<const unsigned char*>(frames) + ((n * H + y) * W + x) * 3;
  // the reference driver divides on the GPU, where torch evaluates `x / 255.` as
  // x * (1.f / 255.f) (division by a host scalar is a multiplication by its reciprocal)
  return __fmul_rn((float)p[2 - c], __fdiv_rn(1.f, 255.f));
}

// pixel (oy, ox) of the S x S resized + normalised image n: torchvision tensor resize (bilinear,
// align_corners = False, no antialias), then (x - mean) / std; (r, g, b, 0)
template <int SRC>
__device__ __forceinline__ float4 resize_px(const void* frames, int n, int oy, int ox, int H, int W, float sy,
                                            float sx, float3 mean, float3 stdv) {
  float ry = fmaxf(__fsub_rn(__fmul_rn(sy, __fadd_rn((float)oy, 0.5f)), 0.5f), 0.f);
  float rx = fmaxf(__fsub_rn(__fmul_rn(sx, __fadd_rn((float)ox, 0.5f)), 0.5f), 0.f);
  int y0 = min((int)floorf(ry), H - 1), x0 = min((int)floorf(rx), W - 1);
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly1 = fminf(fmaxf(__fsub_rn(ry, (float)y0), 0.f), 1.f), ly0 = __fsub_rn(1.f, ly1);
  const float lx1 = fminf(fmaxf(__fsub_rn(rx, (float)x0), 0.f), 1.f), lx0 = __fsub_rn(1.f, lx1);
  const float mv[3] = {mean.x, mean.y, mean.z}, sv[3] = {stdv.x, stdv.y, stdv.z};
  float r[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float p00 = frame_px<SRC>(frames, n, c, y0, x0, H, W), p01 = frame_px<SRC>(frames, n, c, y0, x1, H, W);
    const float p10 = frame_px<SRC>(frames, n, c, y1, x0, H, W), p11 = frame_px<SRC>(frames, n, c, y1, x1, H, W);
    const float a = __fmaf_rn(p00, lx0, __fmul_rn(p01, lx1));
    const float b = __fmaf_rn(p10, lx0, __fmul_rn(p11, lx1));
    const float v = __fmaf_rn(a, ly0, __fmul_rn(b, ly1));
    r[c] = __fdiv_rn(__fsub_rn(v, mv[c]), sv[c]);
  }
  return make_float4(r[0], r[1], r[2], 0.f);
}

// pixel (oy, ox) of the B x B crop of image n around (cx, cy), normalised; pixels of the window that
// lie outside the frame are 0 BEFORE the normalisation (jarvis3D.py:168-178 never leaves the frame: the
// crop centre is clamped; kept for safety as the stand-alone kernel has it)
template <int SRC>
__device__ __forceinline__ float4 crop_px(const void* frames, int n, int cx, int cy, int oy, int ox, int H,
                                          int W, int B, float3 mean, float3 stdv) {
  const int hw = B / 2;
  const int ix = cx - hw + ox, iy = cy - hw + oy;
  const float mv[3] = {mean.x, mean.y, mean.z}, sv[3] = {stdv.x, stdv.y, stdv.z};
  float r[3];
  const bool ok = ix >= 0 && ix < W && iy >= 0 && iy < H;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = ok ? frame_px<SRC>(frames, n, c, iy, ix, H, W) : 0.f;
    r[c] = __fdiv_rn(__fsub_rn(v, mv[c]), sv[c]);
  }
  return make_float4(r[0], r[1], r[2], 0.f);
}

// What the stem convolution reads when the pre-processing is fused into its patch staging.
struct StemSource {
  int mode = 0;               // 0: the plan's own input tensor; 1: resize of the frames; 2: crop of the frames
  const void* frames = nullptr;
  const void* const* frames_cell = nullptr;    // graph replays: the frame pointer of the current call
  int src_u8 = 0;
  const int* center_hm = nullptr;              // crop: [T][C][2]
  int Cloc = 0, C = 0, cam0 = 0;               // crop: image n = (t, local camera)
  int H = 0, W = 0;                            // frame size
  float mean[3] = {0, 0, 0}, stdv[3] = {1, 1, 1};
};

}  // namespace jh

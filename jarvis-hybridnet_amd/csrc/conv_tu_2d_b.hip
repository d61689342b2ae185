// 2D kernel family k=3 (stem, fused-MBConv stages).
#include "conv_mfma.h"
namespace jh {
int conv_launch_2d_k3(const ConvArgs& a, int stride, int nr, int small, size_t budget, hipStream_t s) {
  if (stride == 1)
    return small ? launch_conv_geom<2, 3, 1, 1, 8, 8>(a, nr, budget, s)
                 : launch_conv_geom<2, 3, 1, 1, 8, 16>(a, nr, budget, s);
  return small ? launch_conv_geom<2, 3, 2, 1, 8, 8>(a, nr, budget, s)
               : launch_conv_geom<2, 3, 2, 1, 8, 16>(a, nr, budget, s);
}
// Whole-image tile for 3 x 3 stride-1 layers on images 17..20 pixels wide and at most 22 high (stride 16 of the reference's
// DEFAULT 320-pixel geometry: 20 x 20 fills 52 % of its six 8 x 16 tiles): 23 x 20 pixel slots of which the first 448
// (7 row blocks per wave) are computed -- 89 % of them inside a 20 x 20 image.  One tile per image.
int conv_launch_2d_k3_w20(const ConvArgs& a, int nr, size_t budget, hipStream_t s) {
  return launch_conv_geom<2, 3, 1, 1, 23, 20>(a, nr, budget, s);
}
}  // namespace jh

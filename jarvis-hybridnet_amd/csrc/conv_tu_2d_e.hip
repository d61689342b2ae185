// 2D kernel family: whole-row / whole-image tiles for the 40- and 20-pixel levels of the reference's DEFAULT 320-pixel
// geometry (round 6; the time-batch class >= 8 only, like conv_launch_2d_k3_w20: csrc/conv_host.hip).
#include "conv_mfma.h"
namespace jh {
// 5 x 5 layers on images 33..40 pixels wide: 8 x 40 pixel slots (five row blocks per wave) -- a 40 x 40 image is five
// tiles, all of them full, where the 8 x 16 tiles compute 40 x 48 (83 %).
int conv_launch_2d_k5_w40(const ConvArgs& a, int stride, int nr, size_t budget, hipStream_t s) {
  if (stride == 1) return launch_conv_geom<2, 5, 1, 1, 8, 40>(a, nr, budget, s);
  return launch_conv_geom<2, 5, 2, 1, 8, 40>(a, nr, budget, s);
}
// 3 x 3 stride-2 layers whose OUTPUT is 17..20 wide and at most 22 high: the 23 x 20 whole-image tile (20 x 20 fills 52 %
// of its six 8 x 16 tiles).
int conv_launch_2d_k3s2_w20(const ConvArgs& a, int nr, size_t budget, hipStream_t s) {
  return launch_conv_geom<2, 3, 2, 1, 23, 20>(a, nr, budget, s);
}
}  // namespace jh

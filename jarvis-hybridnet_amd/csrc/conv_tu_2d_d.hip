// 2D kernel families with 16 x 16 pixel tiles (4 row blocks per wave) for the
// high-resolution, few-channel layers at the front of the EfficientNet trunk: their
// workgroups are otherwise too small to amortise the staging round trip and epilogue.
#include "conv_mfma.h"
namespace jh {
int conv_launch_2d_big(const ConvArgs& a, int k, int stride, int nr, size_t budget, hipStream_t s) {
  if (k == 1) return launch_conv_geom<2, 1, 1, 1, 16, 16>(a, nr, budget, s);
  if (k == 3 && stride == 1) return launch_conv_geom<2, 3, 1, 1, 16, 16>(a, nr, budget, s);
  JH_REQUIRE(false, "no 16x16-tile kernel for this conv");
}
}  // namespace jh

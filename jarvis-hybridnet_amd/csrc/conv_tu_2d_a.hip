// 2D kernel families k=1 (pointwise) and k=2 (transposed-conv phases).
#include "conv_mfma.h"
namespace jh {
int conv_launch_2d_k1(const ConvArgs& a, int nr, int small, size_t budget, hipStream_t s) {
  return small ? launch_conv_geom<2, 1, 1, 1, 8, 8>(a, nr, budget, s)
               : launch_conv_geom<2, 1, 1, 1, 8, 16>(a, nr, budget, s);
}
// pointwise layer on an image whose sides do not tile into 8 x 16: the image as ONE row of H * W pixels (a 1 x 1
// convolution has no neighbourhood), cut into 1 x 128 tiles
int conv_launch_2d_k1_flat(const ConvArgs& a, int nr, size_t budget, hipStream_t s) {
  return launch_conv_geom<2, 1, 1, 1, 1, 128>(a, nr, budget, s);
}
int conv_launch_2d_k2(const ConvArgs& a, int nr, int small, size_t budget, hipStream_t s) {
  return small ? launch_conv_geom<2, 2, 1, 1, 8, 8>(a, nr, budget, s)
               : launch_conv_geom<2, 2, 1, 1, 8, 16>(a, nr, budget, s);
}
}  // namespace jh

// 2D kernel families k=1 (pointwise) and k=2 (transposed-conv phases).
#include "conv_mfma.h"
namespace jh {
int conv_launch_2d_k1(const ConvArgs& a, int nr, int small, size_t budget, hipStream_t s) {
  return small ? launch_conv_geom<2, 1, 1, 1, 8, 8>(a, nr, budget, s)
               : launch_conv_geom<2, 1, 1, 1, 8, 16>(a, nr, budget, s);
}
int conv_launch_2d_k2(const ConvArgs& a, int nr, int small, size_t budget, hipStream_t s) {
  return small ? launch_conv_geom<2, 2, 1, 1, 8, 8>(a, nr, budget, s)
               : launch_conv_geom<2, 2, 1, 1, 8, 16>(a, nr, budget, s);
}
}  // namespace jh

// 2D kernel family k=5 (fused-MBConv stage 2).
#include "conv_mfma.h"
namespace jh {
int conv_launch_2d_k5(const ConvArgs& a, int stride, int nr, int small, size_t budget, hipStream_t s) {
  if (stride == 1)
    return small ? launch_conv_geom<2, 5, 1, 1, 8, 8>(a, nr, budget, s)
                 : launch_conv_geom<2, 5, 1, 1, 8, 16>(a, nr, budget, s);
  return small ? launch_conv_geom<2, 5, 2, 1, 8, 8>(a, nr, budget, s)
               : launch_conv_geom<2, 5, 2, 1, 8, 16>(a, nr, budget, s);
}
}  // namespace jh

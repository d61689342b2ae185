// 3D kernel families k=1 (output layer, transposed-conv phases) and k=2 s2.
#include "conv_mfma.h"
namespace jh {
int conv_launch_3d_k1(const ConvArgs& a, int nr, int small, size_t budget, hipStream_t s) {
  return small ? launch_conv_geom<3, 1, 1, 1, 4, 16>(a, nr, budget, s)
               : launch_conv_geom<3, 1, 1, 2, 4, 16>(a, nr, budget, s);
}
int conv_launch_3d_k2s2(const ConvArgs& a, int nr, int small, size_t budget, hipStream_t s) {
  (void)small;
  return launch_conv_geom<3, 2, 2, 1, 4, 16>(a, nr, budget, s);
}
}  // namespace jh

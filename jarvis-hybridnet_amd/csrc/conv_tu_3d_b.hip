// 3D kernel family k=3 (the V2V residual blocks: 80% of the path's FLOPs).
#include "conv_mfma.h"
namespace jh {
int conv_launch_3d_k3(const ConvArgs& a, int stride, int nr, int small, size_t budget, hipStream_t s) {
  if (stride == 1) {
    if (small == 2) return launch_conv_geom<3, 3, 1, 4, 4, 16>(a, nr, budget, s);
    return small ? launch_conv_geom<3, 3, 1, 1, 4, 16>(a, nr, budget, s)
                 : launch_conv_geom<3, 3, 1, 2, 4, 16>(a, nr, budget, s);
  }
  return launch_conv_geom<3, 3, 2, 1, 4, 16>(a, nr, budget, s);
}
}  // namespace jh

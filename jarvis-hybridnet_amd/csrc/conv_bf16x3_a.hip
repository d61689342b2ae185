// explicit instantiations of conv_bf16x3.h (split for parallel compilation): 3D k3 s2, 2D k3 s2
#include "conv_bf16x3.h"
namespace jh {
JH_XCONV_DEFINE(3, 3, 2, 2, 4, 1)
JH_XCONV_DEFINE(2, 3, 2, 1, 16, 1)
}  // namespace jh

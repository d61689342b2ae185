// explicit instantiations of conv_bf16x3.h (split for parallel compilation): 2D k3 s1
#include "conv_bf16x3.h"
namespace jh {
JH_XCONV_DEFINE(2, 3, 1, 1, 16, 1)
JH_XCONV_DEFINE(2, 3, 1, 1, 16, 2)
}  // namespace jh

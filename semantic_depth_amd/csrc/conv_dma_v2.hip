// conv_dma.hip, block shape 2 (256 x 128: 48 / 32 / 24): its twelve instantiations of conv_dma_kernel in a translation unit of their own
#include "conv_dma_kernel.hpp"

namespace sd {

void launch_dma_v2(const ConvParams& p, long M, hipStream_t s) { launch_dma_variant<4, 2, 2, 2, 3, 3, 3>(p, M, s); }

}  // namespace sd

// conv_dma.hip, block shape 3 (256 x 64: 40 / 24 / 20): its twelve instantiations of conv_dma_kernel in a translation unit of their own
#include "conv_dma_kernel.hpp"

namespace sd {

void launch_dma_v3(const ConvParams& p, long M, hipStream_t s) { launch_dma_variant<4, 2, 2, 1, 3, 3, 4>(p, M, s); }

}  // namespace sd

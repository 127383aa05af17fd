// conv_dma.hip, block shape 4 (256 x 32): its twelve instantiations of conv_dma_kernel in a translation unit of their own
#include "conv_dma_kernel.hpp"

namespace sd {

void launch_dma_v4(const ConvParams& p, long M, hipStream_t s) { launch_dma_variant<8, 1, 1, 1, 3, 3, 3>(p, M, s); }

}  // namespace sd

// conv_dma.hip, block shape 1 (128 x 256 block: 48 / 40 / 24 KB per stage): its twelve instantiations of conv_dma_kernel in a translation unit of their own
#include "conv_dma_kernel.hpp"

namespace sd {

void launch_dma_v1(const ConvParams& p, long M, hipStream_t s) { launch_dma_variant<2, 4, 2, 2, 3, 3, 3>(p, M, s); }

}  // namespace sd

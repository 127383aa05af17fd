// conv_dma.hip, block shape 5 (256 x 256: 64 / 48 / 32): its twelve instantiations of conv_dma_kernel in a translation unit of their own
#include "conv_dma_kernel.hpp"

namespace sd {

void launch_dma_v5(const ConvParams& p, long M, hipStream_t s) { launch_dma_variant<2, 4, 4, 2, 2, 2, 4>(p, M, s); }

}  // namespace sd

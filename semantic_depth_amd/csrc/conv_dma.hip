// LDS-DMA conv pipeline of the split engine: variant choice and dispatch.  The kernel template lives in conv_dma_kernel.hpp; each block shape
// is instantiated in its own translation unit (conv_dma_v*.hip: twelve kernels each -- three gather modes x four precisions), so that a clean
// build compiles them side by side instead of sixty kernels in one three-minute hipcc run.
#include "conv_dma_kernel.hpp"

namespace sd {

void launch_dma_v1(const ConvParams& p, long M, hipStream_t s);      // 128 x 256
void launch_dma_v2(const ConvParams& p, long M, hipStream_t s);      // 256 x 128
void launch_dma_v3(const ConvParams& p, long M, hipStream_t s);      // 256 x 64
void launch_dma_v4(const ConvParams& p, long M, hipStream_t s);      // 256 x 32
void launch_dma_v5(const ConvParams& p, long M, hipStream_t s);      // 256 x 256

int conv_dma_variant(const ConvParams& p) {
    if (!p.vec || !p.zero16 || p.Cout % 32 || p.Kpad < 64) return 0;
    // the block shape is chosen on a FULL pass of the engine (ConvParams::Nmax), not on the frames of this call: a call with fewer frames runs the same
    // kernels on fewer tiles, so a frame's result cannot depend on the call it is computed in even where two block shapes round differently (ADVICE r5 #3;
    // conv_dma3_eligible has counted this way since round 5)
    const long M = (long)(p.Nmax > 0 ? p.Nmax : p.N) * p.Hout * p.Wout;
    const long thr = (p.pool || p.out_planar16 || p.fold) ? 0 : 96;     // (a folded layer runs here whatever the batch: its results must not depend on it) tiles needed: the DMA pipeline at half occupancy still beats the register-staged
                                          // kernel (a fused pool exists only here: such layers always take this kernel)
    const bool big = !(p.sw & SW_NO_DMA_BIG);
    // 256 x 256 (one workgroup per CU): 128 flop per byte of L2 -> LDS DMA, the resource the long-K GEMMs run against.  The one-product
    // form of 128 x 256 / 256 x 128 fits TWO workgroups per CU (72 KB of ring, 124 VGPRs): one's epilogue and pipeline fill run under
    // the other's k-loop, which wins below ~32 k-tiles (measured: K <= 640 +5..14 %, K = 768..1280 equal, K >= 1536 and fc6 -2..4 %)
    const bool shortk = p.f16 == 2 && p.Kpad < 1024;
    if (p.x3 && (p.pool || p.out_planar16)) return 0;         // (bf16 x 3: plain outputs only)
    if (p.fold && p.x3) return 0;                             // (folded GEMMs of bf16 x 3: conv_dma3.hip)
    if (big && !p.x3 && !shortk && !p.pool && p.Cout % 256 == 0 && ((M + 255) / 256) * (p.Cout / 256) * (p.fold ? 4 : 1) >= 512) return 5;
    if (p.Cout % 256 == 0 && ((M + 127) / 128) * (p.Cout / 256) >= thr) return 1;     // 128 x 256
    if (p.Cout % 128 == 0 && ((M + 255) / 256) * (p.Cout / 128) >= thr) return 2;     // 256 x 128
    if (p.Cout % 64 == 0 && p.Cout % 128 != 0 && ((M + 255) / 256) * (p.Cout / 64) >= thr) return 3;      // 256 x 64
    if (p.Cout == 32 && !p.pool && !p.out_planar16 && (M + 255) / 256 >= thr && !(p.sw & SW_NO_DMA32)) return 4;      // 256 x 32 (monodepth-vgg conv1b)
    return 0;
}

hipError_t launch_conv_dma(const ConvParams& p0, hipStream_t s) {
    ConvParams p = p0;
    p.dbg = (p.sw & SW_DMA_DBG16) ? 16 : 0;
    const long M = (long)p.N * p.Hout * p.Wout;
    const int v = conv_dma_variant(p);
    if (v == 1) launch_dma_v1(p, M, s);
    else if (v == 2) launch_dma_v2(p, M, s);
    else if (v == 3) launch_dma_v3(p, M, s);
    else if (v == 4) launch_dma_v4(p, M, s);
    else if (v == 5 && !p.x3) launch_dma_v5(p, M, s);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

const char* conv_dma_kernel_name(const ConvParams& p) {
    // f16w: fp16 activations x two fp16 weight planes (2 products); f16x1: fp16 x fp16 (1 product)
    static const char* const names3[5] = {"conv_dma_x3_kernel<2,4,2,2>", "conv_dma_x3_kernel<4,2,2,2>", "conv_dma_x3_kernel<4,2,2,1>", "conv_dma_x3_kernel<8,1,1,1>",
                                          "conv_dma_x3_kernel<2,4,2,2>"};
    if (p.x3) { const int v3 = conv_dma_variant(p); return names3[v3 >= 1 && v3 <= 5 ? v3 - 1 : 2]; }
    static const char* const names[3][5] = {
        {"conv_dma_kernel<2,4,2,2>", "conv_dma_kernel<4,2,2,2>", "conv_dma_kernel<4,2,2,1>", "conv_dma_kernel<8,1,1,1>", "conv_dma_kernel<2,4,4,2>"},
        {"conv_dma_f16w_kernel<2,4,2,2>", "conv_dma_f16w_kernel<4,2,2,2>", "conv_dma_f16w_kernel<4,2,2,1>", "conv_dma_f16w_kernel<8,1,1,1>",
         "conv_dma_f16w_kernel<2,4,4,2>"},
        {"conv_dma_f16x1_kernel<2,4,2,2>", "conv_dma_f16x1_kernel<4,2,2,2>", "conv_dma_f16x1_kernel<4,2,2,1>", "conv_dma_f16x1_kernel<8,1,1,1>",
         "conv_dma_f16x1_kernel<2,4,4,2>"}};
    const int v = conv_dma_variant(p);
    static const char* const namesh[5] = {"conv_dma_hs_kernel<2,4,2,2>", "conv_dma_hs_kernel<4,2,2,2>", "conv_dma_hs_kernel<4,2,2,1>", "conv_dma_hs_kernel<8,1,1,1>",
                                          "conv_dma_hs_kernel<2,4,4,2>"};
    if (p.f16 == 4) return namesh[v >= 1 && v <= 5 ? v - 1 : 2];
    return names[p.f16 == 2 ? 2 : (p.f16 ? 1 : 0)][v >= 1 && v <= 5 ? v - 1 : 2];
}

}  // namespace sd

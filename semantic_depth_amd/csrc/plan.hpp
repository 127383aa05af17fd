// Layer plans of the two networks: tensors, ops, weight slots, activation-arena layout.
// Host-side only.  The C++ statement of the architectures (the Python statement is semantic_depth_amd/weights.py;
// tests/test_abi.py checks that the two agree).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace sd {

struct TensorDesc {
    std::string name;
    int N = 0, H = 0, W = 0, C = 0;   // C = stored channels
    int Ctf = 0;                      // channels the TensorFlow graph sees (input_pre stores 4, TF sees 3)
    int fmt = 0;                      // 0: f32 NHWC; 1: split-bf16 planes (split_fmt.hpp)
    int f16 = 0;                      // 3: fp16 hi + SCALED lo planes ("HS": every tensor of SD_PREC_F16X2, NetPlan::h2);
                                      // 1: ONE fp16 plane, 2: fp16 hi + lo planes, instead of two bf16 planes (split_fmt.hpp): every conv that reads it runs the
                                      // 2-product scheme x * (w_hi + w_lo) (precision plan, see NetPlan::f16_spec)
    int x3 = 0;                       // 1: bf16 x 3 planes (SD_PREC_BF16X3; 6 bytes per element, planes at 0, plane, 2 * plane)
    int planar16 = 0;                 // split planes stored as C/16 sub-planes of 16 channels ([C/16][N][H][W][16] per plane): the
                                      // producer is the stem kernel, the only consumer a direct conv, whose 16-channel halo DMA
                                      // then reads whole 128-byte lines
    size_t bytes = 0;
    size_t offset = 0;      // byte offset in the activation arena
    int first = -1, last = -1;   // op indices (liveness)
};

enum OpKind { OP_PRE_VGG, OP_PRE_MONO, OP_CONV, OP_CONV_DIRECT, OP_SMALLN, OP_POOL2, OP_POOL3Z, OP_DECONV4_ADD, OP_HEAD16,
              OP_DEC_TAIL1 /* dec_tail.hip: upconv1 -> iconv1 -> disp1 of the bf16 x 3 monodepth in one launch */ };
enum WLayout { WL_RAW = 0, WL_IGEMM = 1, WL_SMALLN = 2, WL_BIAS4 = 3, WL_IGEMM_SPLIT = 4, WL_DIRECT_SPLIT = 5,
               WL_TAIL_UP = 6, WL_TAIL_ICONV = 7 /* MFMA A fragments of dec_tail.hip (DecTailParams::w1 / w2) */ };

struct WeightSlot {
    std::string name;
    int64_t shape[4] = {0, 0, 0, 0};
    int rank = 0;
    int layout = WL_RAW;
    int Kpad = 0, CoutPad = 0, nout = 0;
    int nsplit = 1;        // WL_DIRECT_SPLIT: output channels are stored as nsplit blocks of CoutPad (128-channel layers: 2 x 64)
    int nsrc = 1, srcCtf[3] = {0, 0, 0}, srcCpad[3] = {0, 0, 0};   // WL_IGEMM: channel structure of the K axis
    int vec = 0;           // 1: K axis is ordered (32-channel block, tap, channel) instead of (tap, channel)
    int f16 = 0;           // split layouts: two fp16 planes, hi = fp16(w), lo = fp16(w - hi) (the 2-product scheme of split_fmt.hpp)
    int x3 = 0;            // split layouts: THREE bf16 planes hi, mid, lo (exact: w = hi + mid + lo), SD_PREC_BF16X3
    int hs = 0;            // split layouts (with f16): the two fp16 planes hold w' = w * wscale (SD_PREC_F16X2: hi = fp16(w'), lo = fp16(w' - hi); the conv
                           // epilogues multiply the accumulator by 1 / wscale)
    float wscale = 4096.f; // hs: the layer's power-of-two weight scale 2^k, chosen by sd_load_weight from the tensor it is given so that the largest
                           // stored |w'| lies in [2^12, 2^13) (folded layers: the bound 4 max|w| on a sum of four taps); slots that share an
                           // accumulator (ResNet conv3 + projection: `owner`) share one scale
    float scale = 1.f;     // the tensor is multiplied by this while it is loaded (monodepth stem with integer input: 1/255, see NetPlan::input_scale)
    // a slot may be a VIEW of rows [k_off, k_off+Kpad) of a larger device matrix of Ktotal rows owned by slot `owner`
    // (ResNet block: conv3 and the projection shortcut are one GEMM over the concatenated K axis); a bias view is ADDED
    // to its owner's bias
    int owner = -1, k_off = 0, Ktotal = 0;
    int fold = 0;          // WL_IGEMM_SPLIT: upsample-folded 3x3 weights (OpDesc::fold): four parity panels of 4 C rows each, stacked along K
    int srcVec[3] = {0, 0, 0};   // mixed layers: sources with C % 32 == 0 live in the vec region of K, the others in the quad tail
    size_t offset = 0, bytes = 0;   // in the weight arena (re-laid-out form)
    bool loaded = false;
};

struct OpDesc {
    OpKind kind;
    std::string name;
    int nsrc = 0;
    int src[3] = {-1, -1, -1};
    int up[3] = {0, 0, 0};
    int sstride[3] = {1, 1, 1};   // per-source stride (a fused ResNet shortcut reads its source at stride 2)
    int dst = -1;
    int residual = -1;     // OP_CONV: tensor added before the activation; OP_DECONV4_ADD: the skip tensor
    int k = 1, stride = 1, pad = 0, act = ACT_NONE, nout = 0;
    int w = -1, b = -1;    // weight slots
    int w2 = -1, b2 = -1, w3 = -1, b3 = -1;      // OP_DEC_TAIL1: iconv1 and disp1 (w, b = upconv1)
    // conv engine
    int Ctot = 0;          // padded channels per tap (every source rounded up to a multiple of 4)
    int K = 0, Kpad = 0, vec = 0, m_fastest = 0;
    int f16 = 0;                 // conv ops: 4 = SD_PREC_F16X2: fp16 hi + scaled lo sources x fp16 hi + lo weights (times 2^12), THREE products;
                                 // 1 = sources are single fp16 planes, weights two fp16 planes, TWO MFMA products per product;
                                 // 2 = the same sources, w_hi only, ONE product (plain fp16 x fp16)
    int fold = 0;                // OP_CONV: a 3x3 stride-1 conv on a x2 nearest-neighbour upsampled source, run as FOUR 2x2 convs on the source itself
                                 // (one per output parity (y & 1, x & 1)): out[2i+py, 2j+px] = sum_{a,b in 0..1} Wf[py][px][a][b] . src[i+a-1+py, j+b-1+px]
                                 // with Wf = the 3x3 taps that read the same source pixel added up (plan.cpp relayout_weight).  4/9 of the
                                 // multiplications; Kpad = 4 C is the K axis of ONE parity
    int fuse_pool = 0;           // OP_CONV_DIRECT / OP_CONV (LDS-DMA kernel): the 2x2 max pool that follows is applied in the epilogue
    int nchunks = 0;             // OP_CONV_DIRECT: 32-channel chunks over the concatenated sources
    int nsplit = 1;              // OP_CONV_DIRECT: passes of <= 64 output channels per tile
    int Kvec = 0, CqPad = 0;     // mixed layers: K = [vec region: (32-channel block, tap, channel)] + [quad tail: (tap, channel quads)]
    size_t tab_offset = 0, tab_bytes = 0;    // KEntry table, in the weight arena
    double flops = 0;                        // 2*M*N*K for the whole chunk
};

struct NetPlan {
    std::string net;
    int frames = 0;        // frames per chunk
    int images = 0;        // images per chunk (monodepth: 2 per frame)
    int H = 0, W = 0;
    int prec = 0;          // 0: exact f32 MFMA, 1: split engine (planes of split_fmt.hpp)
    int x3 = 0;            // split engine with bf16 x 3 planes everywhere (SD_PREC_BF16X3): fp32-grade, six MFMA products per product
    int h2 = 0;            // split engine with fp16 hi + scaled lo planes everywhere (SD_PREC_F16X2, split_fmt.hpp "HS"): fp32-grade, THREE fp16 MFMA
                           // products per product, 4 bytes per activation element; the kernels of the bf16 x 2 engine in their H2 form
    // precision plan of the split engine: which conv layers run the 2-product fp16 scheme (fp16x2 activations x fp16 weights)
    // instead of the 3-product bf16 one.  f16_spec = what was asked for (comma-separated op names, a trailing '*' matches a
    // prefix, "*" = every layer, empty = none); f16_ops = the layers that run it after the consistency closure (a tensor has ONE
    // plane format, so every conv reading an fp16 tensor is a 2-product layer and every source of a 2-product layer is fp16)
    std::string f16_spec, f16_ops;
    double flops_f16 = 0;  // per image, of the fp16 layers (2-product and 1-product)
    double flops_f16x1 = 0;   // per image, of the 1-product layers among them
    std::vector<TensorDesc> tensors;
    std::vector<OpDesc> ops;
    std::vector<WeightSlot> weights;
    std::map<std::string, int> tensor_by_name, weight_by_name;
    size_t weight_bytes = 0;   // weights + tables
    size_t act_bytes = 0;
    double flops_per_image = 0;
    int t_input = -1, t_output = -1;
    // monodepth, fp16 stem: the input tensor holds the pixel VALUES 0..255 (exact in fp16) instead of value / 255 and the stem's weights
    // carry the 1/255 -- the two-product stem then has no activation rounding at all.  1/255 otherwise.
    float input_scale = 1.f / 255.f;
};

// prec: 0 exact f32 MFMA, 1 split engine (bf16 x 2 + the fp16 forms of f16_layers), 2 split engine with bf16 x 3 planes (f16_layers ignored),
// 3 split engine with fp16 hi + scaled lo planes (f16_layers ignored)
NetPlan build_fcn8s(int frames, int H, int W, int prec, const char* f16_layers = nullptr);
NetPlan build_monodepth(int encoder /*0 vgg, 1 resnet50*/, int frames, int H, int W, int prec, const char* f16_layers = nullptr);

// host-side re-layout of one TensorFlow-layout weight into its slot's kernel layout
void relayout_weight(const WeightSlot& s, const float* tf_data, std::vector<float>& out);
// gather-descriptor table of one conv op, given the bound activation arena
void build_conv_tables(const NetPlan& p, const OpDesc& op, const char* act_base, const char* tab_dev, std::vector<KEntry>& ktab);
void build_direct_chunks(const NetPlan& p, const OpDesc& op, const char* act_base, std::vector<DirectChunk>& chunks);

int conv_tile_n(int Cout);   // conv_igemm.hip
std::string sd_disable_unknown();   // plan.cpp: first token of SEMDEPTH_DISABLE that is not a switch name ("" = all known)

}  // namespace sd

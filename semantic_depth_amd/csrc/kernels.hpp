// Launch interfaces of the HIP kernels (gfx950).  Plain structs; everything is NHWC float32 unless stated.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sd {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// run-time switches (SEMDEPTH_DISABLE=name,... -- plan.cpp sd_disabled; the closed A/Bs and the decomposition runs of earlier rounds in -DSD_DEV_VARIANTS builds
// only), latched ONCE per handle in sd_create (plan.cpp latch_switches) and handed to the launchers in their parameter structs: nothing on the launch path calls getenv
enum Switch : unsigned {
    SW_NO_N16 = 1u << 0, SW_NO_UPTILE = 1u << 1, SW_NO_N16_MT1 = 1u << 2, SW_NO_DMA_BIG = 1u << 3, SW_NO_DMA32 = 1u << 4,
    SW_NO_STEM = 1u << 5, SW_NO_FUSE4 = 1u << 6, SW_NO_SMALLN_TILE = 1u << 7, SW_NO_DMA = 1u << 8, SW_DMA_DBG16 = 1u << 9,
    SW_PROFILE_VERBOSE = 1u << 10, SW_NO_FUSE1 = 1u << 11,
    // bf16 x 3 engine (round 3): SEMDEPTH_X3_KEEP=0 (no register-cached X fragments), SEMDEPTH_X3_RING3 (three-slot weight ring of the
    // NB = 1 layers), SEMDEPTH_NO_DMA3 (128 x 256 two-stage GEMM block instead of the phased 256 x 256 one), SEMDEPTH_X3_DIAG=1|2|3
    // (decomposition runs of conv_direct3: no output stores / no MFMAs)
    SW_X3_NOKEEP = 1u << 12, SW_X3_RING3 = 1u << 13, SW_NO_DMA3 = 1u << 14, SW_X3_DIAG_NOSTORE = 1u << 15, SW_X3_DIAG_NOMFMA = 1u << 16,
    SW_NO_FOLD = 1u << 17,       // SEMDEPTH_NO_FOLD: the upconv layers as 3x3 convs on the upsampled source (plan-time switch)
    SW_NO_TAIL1 = 1u << 18,      // SEMDEPTH_NO_TAIL1: upconv1 / iconv1 / disp1 of the bf16 x 3 monodepth as three launches (plan-time switch)
    SW_NO_ROWSKIP = 1u << 19,    // SEMDEPTH_NO_ROWSKIP: conv_dma3 without the row-grouped pixel order (ConvParams::rowgrp)
    SW_NO_FLAT = 1u << 20,       // SEMDEPTH_NO_FLAT: conv_dma3's 1x1 layers through the general gather
    SW_X3_DIAG_TIMED = 1u << 21, // SEMDEPTH_X3_DIAG=4: conv_direct3's timed copy (s_memtime stamps per item; decomposition runs)
    SW_HS_TAPS = 1u << 23,       // SEMDEPTH_HS_PHASED_TAPS: ALL tap layers of the three-product engine (folded upconvs, strided 3x3; fc6 is there anyway) on conv_dma3's two-phase ring
    SW_MFMA32 = 1u << 22         // SEMDEPTH_MFMA32: conv_dma3's bf16 x 3 layers on 32x32x16 MFMAs instead of 16x16x32 (round 5; conv_dma3.hip "S16")
};
unsigned latch_switches();      // plan.cpp
bool sd_disabled(const char* what);   // plan.cpp: is `what` in SEMDEPTH_DISABLE?
// decomposition runs (SEMDEPTH_X3_DIAG: 1 no output stores, 2 no MFMAs / no epilogue): compiled into -DSD_DEV_VARIANTS builds only -- in the shipped library
// the expression is the constant 0 and every branch on it folds away
#ifdef SD_DEV_VARIANTS
#define SD_DIAG_BITS(sw) ((((sw) & sd::SW_X3_DIAG_NOSTORE) ? 1 : 0) | (((sw) & sd::SW_X3_DIAG_NOMFMA) ? 2 : 0))
#else
#define SD_DIAG_BITS(sw) 0
#endif

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_ELU = 2, ACT_SIGMOID03 = 3 /* 0.3*sigmoid, monodepth get_disp */ };

// ---------------------------------------------------------------------------------------------
// implicit-GEMM convolution on the f32 MFMA (conv_igemm.hip)
// ---------------------------------------------------------------------------------------------
// one gather descriptor = 32 bytes, fetched with a single scalar load:
//   vec path : one per 32-wide k-tile  (32 consecutive channels of one tap of one source)
//   quad path: one per channel quad    (<= 4 consecutive channels of one tap of one source)
struct KEntry {
    const float* base;  // source tensor + first channel of this entry
    int H, W, C;        // physical dims of the source tensor
    int dy, dx;         // tap offset minus pad (in logical, i.e. post-upsample, input coordinates)
    int flags;          // bit 0: read through a x2 nearest-neighbour upsample; bits 4..5: stride of this source (1 or 2);
                        // bits 8..10: valid floats (quad path);
                        // bit 16: entry is live (0 = zero padding of the K axis); bit 17: k-tile head of a QUAD tile
                        // (base then points at the tile's 8 quad descriptors)
};
static_assert(sizeof(KEntry) == 32, "KEntry must be 32 bytes");

struct ConvParams {
    int nsrc, Ctot;    // channel-concatenated sources (tf.concat order), described by ktab
    int N, Hin, Win;   // logical input dims (after upsample)
    int Hout, Wout, Cout, CoutPad;
    int kh, kw, stride, pad;
    int K, Kpad;       // K = kh*kw*Ctot (Ctot = sum of sources rounded up to quads) ; Kpad = multiple of 32
    const float* wt;   // re-laid-out weights [Kpad/4][CoutPad][4]
    const float* bias; // [Cout]
    const KEntry* ktab;  // DEVICE: [Kpad/32] k-tile descriptors, followed by 8 quad descriptors per quad tile
    int vtiles;        // leading k-tiles that are vec tiles (the rest are quad tiles)
    int vec;           // 1: every source has C % 32 == 0 (all k-tiles are vec tiles); 0: the K axis ends in quad tiles
    const float* residual;  // nullable [M][Cout], added before the activation (resnet shortcut)
    float* out;             // [M][Cout]
    int act;
    size_t out_plane;  // split engine: element offset of the output's lo plane
    const void* zero16;  // split engine: 16 zero bytes in device memory (source of out-of-image taps for the LDS-DMA pipeline)
    int Nmax;          // split engine: images of a full chunk (plane stride of a source = Nmax*H*W*C elements)
    const void* src0;  // first source tensor (hi plane) and the element offset of its lo plane: conv_stem.hip reads it directly
    size_t src0_plane;
    int f16;           // 1: the INPUT is ONE fp16 plane and the weights two fp16 planes (2 MFMA products, split_fmt.hpp); 2: the same
                       // input and w_hi only (1 MFMA product: plain fp16 x fp16; conv_dma / conv_direct, the others run the 2-product form)
                       // 4 (SD_PREC_F16X2): fp16 hi + SCALED lo input planes x fp16 hi + lo weights, THREE products, `alpha` applied (split_fmt.hpp "HS")
    int out_f16;       // OUTPUT split planes are fp16 (the format the consumers of the output tensor compute in); 3 = fp16 hi + scaled lo (HS)
    float alpha;       // f16 == 4: 1 / WeightSlot::wscale = 2^-k of the layer's weight scale: out = act(acc * alpha + bias)
    int out_planar16;  // conv_stem.hip: write the output as Cout/16 sub-planes of 16 channels (TensorDesc::planar16)
    int pool;          // conv_dma.hip: output pixels are walked in 2x2-window-major order and the epilogue max-pools each
                       // window: out is [N,Hout/2,Wout/2,Cout]
    int simple;        // 1: one source, stride 1, no upsample, all k-tiles vec: the DMA kernel computes its gather addresses;
                       // 2: two-source 1x1 GEMM (ResNet conv3 + projection), per-source strides, no upsample
    int dbg;           // SEMDEPTH_DMA_DBG=16: general gather path on SIMPLE layers too (A/B switch; 0 in production)
    int m_fastest;     // block order: 1 = consecutive blocks walk M (share a weight panel), 0 = walk N
    unsigned sw;       // Switch bits of the handle
    int reserve_cus;   // persistent launches use (CUs - reserve_cus) workgroups: the CUs left free take the per-frame tail of the previous step
                       // that runs beside the networks on a side stream (sd_set_reserved_cus; 0 = every CU)
    unsigned long long* sat;   // DEVICE counter of fp16-saturated output values (split_fmt.hpp sat_report; sd_saturation_count)
    int x3;            // 1: bf16 x 3 planes in, out and in the weights (SD_PREC_BF16X3: six MFMA products per product, split_fmt.hpp)
    int flat;          // 1 (conv_dma3.hip): a 1x1 conv without upsample (every source read at tap (0, 0), per-source strides allowed): the gather
                       // offset of a lane is computed once per source geometry (conv_dma3_kernel<FLAT>); SEMDEPTH_NO_FLAT switches it off
    int noup;          // 1 (conv_dma3.hip): the ONE source is not read through an upsample, stride 1 or 2, planes below 4 GB: the gather of a
                       // k-tile is a precomputed per-lane pixel offset + a scalar tap offset + the in-image test (conv_dma3_kernel<2>)
    int rowgrp;        // > 0 (conv_dma3.hip): the GEMM's pixels are ordered (image group of rowgrp images, row, image, column) with rowgrp * Wout = 256,
                       // so that a 256-pixel tile is ONE output row of rowgrp images and skips the k-tiles of the taps whose input row is zero
                       // padding (fc6: 7x7 on 16 rows, 10.7 % of the k-tiles).  Results are bit-identical to the plain order.
    int fold;          // 1 (conv_dma3.hip): upsample-folded 3x3 conv (OpDesc::fold).  Hout x Wout are the SOURCE dims (the GEMM's pixel space), the
                       // grid carries four parities, parity q = 2 py + px reads weight rows [q Kpad, (q + 1) Kpad) and table entries [q Kpad/32, ...)
                       // and writes source pixel (i, j) to output pixel (2 i + py, 2 j + px) of the [N, 2 Hout, 2 Wout, Cout] tensor
};
hipError_t launch_conv_igemm(const ConvParams& p, hipStream_t s);
const char* conv_igemm_kernel_name(const ConvParams& p);
// split-bf16 engine (conv_split.hip): wt = two bf16 planes [Kpad/8][CoutPad][8] (hi, then lo)
hipError_t launch_conv_split(const ConvParams& p, hipStream_t s);
const char* conv_split_kernel_name(const ConvParams& p);
int conv_split_tile_n(int Cout);
// LDS-DMA pipeline for the wide vec layers of the split engine (conv_dma.hip); variant 0 = not applicable
int conv_dma_variant(const ConvParams& p);
hipError_t launch_conv_dma(const ConvParams& p, hipStream_t s);
bool conv_stem_eligible(const ConvParams& p);                        // conv_stem.hip: layers on the 4-channel network input
hipError_t launch_conv_stem(const ConvParams& p, hipStream_t s);
const char* conv_dma_kernel_name(const ConvParams& p);
bool conv_dma3_eligible(const ConvParams& p);                        // conv_dma3.hip: bf16 x 3, 256 x 256 block with phased weight planes
hipError_t launch_conv_dma3(const ConvParams& p, hipStream_t s);
int conv_dma3_mode(const ConvParams& p);                             // gather variant (conv_dma3_kernel<MODE>) the layer runs

#ifdef __HIPCC__
// ELU for the conv epilogues.  expm1f() is a ~60-instruction library routine and the monodepth layers apply it to every
// output element; this is exp2 (one transcendental) away from zero and the Taylor series near it, |abs error| < 2.5e-7
// (the reference graph itself evaluates exp(x) - 1 in f32: monodepth_model.py conv -> tf.nn.elu).
__device__ __forceinline__ float fast_elu(float v) {
    const float e = __builtin_amdgcn_exp2f(v * 1.4426950408889634f) - 1.0f;
    const float q = v * (1.0f + v * (0.5f + v * (0.16666667f + v * (0.041666668f + v * 0.008333334f))));
    const float r = v > -0.125f ? q : e;
    return v > 0.f ? v : r;
}
// branch-free form for the split-bf16 engine (abs error < 2e-7, below the 2^-17 relative step of the split format for
// every |value| > 0.02): elu(v) = max(v, 0) + (exp(min(v, 0)) - 1)
__device__ __forceinline__ float fast_elu_split(float v) {
    return fmaxf(v, 0.f) + (__builtin_amdgcn_exp2f(fminf(v, 0.f) * 1.4426950408889634f) - 1.0f);
}
template <int ACT> struct ActTag { static constexpr int value = ACT; };
template <bool B> struct BoolTag { static constexpr bool value = B; };   // output plane format of an epilogue: true = fp16 planes
template <int I> struct IntTag { static constexpr int value = I; };      // the same, three-way (split_fmt.hpp split2_fmt)
// activation selected at compile time inside the epilogues (a run-time switch per value costs more than the arithmetic)
template <int ACT> __device__ __forceinline__ float act_split(float v) {
    if (ACT == 1) return fmaxf(v, 0.f);
    if (ACT == 2) return fast_elu_split(v);
    if (ACT == 3) return 0.3f * (1.0f / (1.0f + expf(-v)));      // as smalln_act (ops_misc.hip)
    return v;
}
// the same on a vector of four values (round 6): written on the whole vector so that the additions of the ELU become v_pk_add_f32 (two values per issue slot;
// hipcc keeps the per-element loop scalar): the same operations in the same order on every element -- bit-identical to act_split<ACT> element by element --,
// 42 instead of 48 VALU instructions per four values of an ELU + HS-split epilogue
template <int ACT, class V4> __device__ __forceinline__ V4 act_split4(V4 v) {
    if constexpr (ACT == 1) {
        const V4 z = {0.f, 0.f, 0.f, 0.f};
        return __builtin_elementwise_max(v, z);
    } else if constexpr (ACT == 2) {
        const V4 z = {0.f, 0.f, 0.f, 0.f};
        const V4 m = __builtin_elementwise_max(v, z);
        const V4 n = __builtin_elementwise_min(v, z) * 1.4426950408889634f;
        V4 e;
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(n[r]);
        return m + (e - 1.0f);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act_split<ACT>(v[r]);
        return v;
    }
}
// the bf16 x 3 engine's activation (round 5): its ELU is the branch-free form of the split engines, max(v, 0) + (exp2(min(v, 0) log2 e) - 1) -- six VALU slots
// instead of thirteen (exp2 AND a degree-5 polynomial AND two selects), which is ~35 % of the VALU work of an epilogue that the two waves of a SIMD execute one
// after the other (profiles/r05_conv_dma3_timed_bf16x3.txt).  Its absolute error near zero (<= 1.3e-7: v_exp_f32 is good to an ulp of a value near 1) was the
// reason round 4 kept the polynomial; the three-product engine has run this form since it exists and measures CLOSER to the float64 oracle than this engine did
// (profiles/r05_f32_grade_check.txt), and the reference's own tf.nn.elu is exp(x) - 1 in f32.
template <int ACT> __device__ __forceinline__ float act_x3(float v) {
    if (ACT == 1) return fmaxf(v, 0.f);
    if (ACT == 2) return fast_elu_split(v);
    return v;
}
template <int ACT> __device__ __forceinline__ float act_f32(float v) {
    if (ACT == 1) return fmaxf(v, 0.f);
    if (ACT == 2) return fast_elu(v);
    return v;
}
#endif

// direct 3x3 convolution for full-resolution, few-channel layers of the split engine (conv_direct.hip)
struct DirectChunk {       // one 16-channel chunk of one source (32 bytes)
    const void* base;      // hi plane of the source + first channel of the chunk
    int H, W, C;           // physical dims of the source tensor
    int up;                // 1: read through a x2 nearest-neighbour upsample
    int nvalid;            // channel octets of the chunk that exist (1..2)
    int pad;               // 0, or the tensor's channel count when the source is stored as 16-channel sub-planes (then C = 16)
};
static_assert(sizeof(DirectChunk) == 32, "DirectChunk must be 32 bytes");
struct ConvDirectParams {
    const DirectChunk* chunks;   // DEVICE [nchunks]
    int nchunks;
    int N, H, W;                 // output (= logical input) dims
    int Cout;                    // output channels PER SPLIT: <= 32 or 64, multiple of 8
    int nsplit, Cstride;         // layers with 128 / 256 output channels run as 2 / 4 passes of 64 per tile (work item = tile x pass);
                                 // Cstride = channels of the output tensor
    int out_planar16;            // write the output as 16-channel sub-planes (TensorDesc::planar16)
    int nreal;                   // > 0: only the first nreal of the Cout stored channels are real, the rest are written as zeros
                                 // (disparity heads: 2 channels in a zero-padded octet; N16 kernel)
    int all_up;                  // every chunk has up == 1 (upconv layers): the kernel keeps source-resolution halo tiles
    int fold;                    // conv_direct3.hip: upsample-folded weights [split][plane][chunk][parity 4][2x2 tap 4][octet 2][32 or 64][8] (OpDesc::fold)
    const u32x4_t* wt;           // [split][plane][chunk][tap 9][octet 2][32 or 64][8 bf16]
    const float* bias;
    float* out;                  // split planes [N,H,W,Cout]
    size_t out_plane;
    int act, Nmax;
    const void* zero16;
    int rows_per_wave;           // 1: 8 x 32 tiles, 2: 16 x 32 tiles (see conv_direct.hip)
    int f16;                     // 1: ONE fp16 input plane x two fp16 weight planes (2 MFMA products); 2: x w_hi only (1 product);
                                 // 3: fp16 hi + lo input planes x w_hi (2 products: x_hi*w_hi + x_lo*w_hi)
                                 // 4 (SD_PREC_F16X2): fp16 hi + scaled lo input planes x fp16 hi + lo weights, THREE products, `alpha` applied
    int out_f16;                 // OUTPUT planes: 0 bf16 hi + lo, 1 ONE fp16 plane, 2 fp16 hi + lo, 3 fp16 hi + scaled lo (HS)
    float alpha;                 // f16 == 4: 2^-k of the layer's weight scale: out = act(acc * alpha + bias)
    int pool;                    // 1: fused 2x2 stride-2 max pool, out is [N,H/2,W/2,Cout] (needs rows_per_wave == 2)
    unsigned sw;                 // Switch bits of the handle
    int reserve_cus;             // the persistent grid is (CUs - reserve_cus) workgroups (ConvParams::reserve_cus)
    unsigned long long* sat;     // DEVICE counter of fp16-saturated output values (split_fmt.hpp sat_report; sd_saturation_count)
};
hipError_t launch_conv_direct3(const ConvDirectParams& p, hipStream_t s);       // conv_direct3.hip: the bf16 x 3 form (SD_PREC_BF16X3)
const char* conv_direct3_kernel_name(const ConvDirectParams& p);
hipError_t launch_conv_direct(const ConvDirectParams& p, hipStream_t s);
const char* conv_direct_kernel_name(const ConvDirectParams& p);

// the full-resolution decoder tail of the bf16 x 3 monodepth in one launch (dec_tail.hip): upconv1 -> iconv1 -> disp1[..., 0]
struct DecTailParams {
    const void* a;         // dec/iconv2: bf16 x 3 planes [N, H/2, W/2, 32]
    size_t a_plane;        // element offset between its planes
    const void* d2;        // dec/disp2: bf16 x 3 planes [N, H/2, W/2, 8] (2 real channels in a zero-padded octet)
    size_t d_plane;
    int N, H, W;           // full resolution
    const u32x4_t* w1;     // upconv1, upsample-folded: MFMA A fragments [parity 4][2x2 tap 4][plane 3][lane 64] (plan.cpp WL_TAIL_UP)
    const float* b1;       // [16]
    const u32x4_t* w2;     // iconv1: A fragments [row block 3 x half 2][plane 3][lane 64] (WL_TAIL_ICONV)
    const float* b2;       // [16]
    const float* wd;       // disp1, output channel 0: [tap 9][16]
    const float* bd;       // [1]
    float* out;            // [N, H, W] f32
    unsigned sw;
    int hs;                // 1 (SD_PREC_F16X2): a / d2 are fp16 hi + scaled lo planes, w1 / w2 hold fp16 hi + lo of w * 2^k in planes 0 and 1 (split_fmt.hpp "HS")
    int reserve_cus;       // the persistent grid is (CUs - reserve_cus) workgroups (ConvParams::reserve_cus)
    float alpha, alpha2;   // hs: 2^-k of upconv1's / iconv1's weight scale, applied to the accumulators of stage 1 / stage 2
};
bool dec_tail1_eligible(int H, int W);
hipError_t launch_dec_tail1(const DecTailParams& p, hipStream_t s);

// small-N convolution (N <= 4 output channels: score 1x1 convs, monodepth disparity heads)
struct SmallNParams {
    const float* x;     // [N,H,W,C] (f32, or split planes when in_split)
    int in_split, out_split;       // activation formats (split_fmt.hpp); out_split needs nout == 2
    int out_c;                     // stored channels of a split output (2, or 8 = zero-padded octet for the direct conv)
    size_t in_plane, out_plane;    // element offset of the lo plane
    size_t in_sub;                 // > 0: the input is stored as C/16 sub-planes of 16 channels, in_sub elements each
                                   // (TensorDesc::planar16; tiled kernel only)
    int N, H, W, C;
    int k;              // 1 or 3 (stride 1, zero pad (k-1)/2)
    int nout;           // computed output channels (<= 4)
    const float* wt;    // [nout][k*k*C]
    const float* bias;  // [4]
    float* out;         // [N,H,W,nout]
    int act;
    const void* zero16; // 16 zero bytes (padding source of the LDS-DMA halo loads of the tiled kernel)
    int f16;            // INPUT planes: 0 bf16 hi + lo, 1 ONE fp16 plane, 2 fp16 hi + lo (per-thread / per-wave kernels), 3 fp16 hi + scaled lo (HS)
    int x3;             // split planes (in and out) are bf16 x 3 (per-thread / per-wave kernels)
    int out_f16;        // OUTPUT split planes (out_split) are fp16 (3: fp16 hi + scaled lo)
    unsigned sw;        // Switch bits of the handle
};
hipError_t launch_conv_smalln(const SmallNParams& p, hipStream_t s);
bool conv_smalln_tiled(int in_split, int k, int W, int C, int nout, unsigned sw);   // the LDS-tiled kernel takes this layer (it alone reads sub-planar inputs)

// ---------------------------------------------------------------------------------------------
// misc network ops (ops_misc.hip)
// ---------------------------------------------------------------------------------------------
// `split`: 0 f32, 1 split-bf16 planes (split_fmt.hpp), 2 ONE fp16 plane, 4 bf16 x 3 planes, 5 fp16 hi + scaled lo (HS); `plane*` = element offset of the lo plane
hipError_t launch_pre_vgg(const uint8_t* frames, float* out, long npix, int split, size_t plane, hipStream_t s);                 // K1
hipError_t launch_pre_mono(const uint8_t* frames, float* out, int B, int H, int W, int split, size_t plane, int raw, hipStream_t s);      // /255 (raw: not) + fliplr pair
hipError_t launch_maxpool2(const float* x, float* y, int N, int H, int W, int C, int split, size_t plane_in, size_t plane_out, hipStream_t s);
hipError_t launch_maxpool3z(const float* x, float* y, int N, int H, int W, int C, int split, size_t plane_in, size_t plane_out, int sub_nmax, hipStream_t s);
hipError_t launch_unsplit(const float* x, float* y, long npix, int C, int Ctf, size_t plane, size_t sub, int f16, hipStream_t s);  // split planes -> f32 [npix][Ctf]; f16 = TensorDesc::f16, or -1: bf16 x 3
// y[n,2i+ky-1,2j+kx-1,o] += x[n,i,j,c]*w[ky,kx,o,c]; y += bias + skip   (3->3 channels; fcn8s/fcn.py:186-204)
hipError_t launch_deconv4s2_add(const float* x, const float* w, const float* bias, const float* skip, float* y,
                                int N, int H, int W, hipStream_t s);
// 16x16 s8 transposed conv + softmax + 0.5 thresholds + argmax (fcn8s/fcn.py:207-224, semantic_depth.py:550-564)
hipError_t launch_deconv16s8_head(const float* x, const float* w, const float* bias, int N, int H, int W,
                                  float* logits, uint8_t* road, uint8_t* fence, uint8_t* argmax, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// fusion (fuse.hip; compiled with -ffp-contract=off: bit-exact vs numpy / OpenCV arithmetic)
// ---------------------------------------------------------------------------------------------
struct CamDev { double q[16]; float mult; };   // Q rounded to f32 then widened; multiplier as f32
hipError_t launch_post_process(const float* disp_raw, float* disp_pp, int B, int H, int W, hipStream_t s);
// resize.hip: cv2.INTER_CUBIC for u8 frames; xi/xa (yi/ya): [dw][4] ([dh][4]) tap indices and fixed-point weights (device)
hipError_t launch_resize_cubic_u8(const uint8_t* src, uint8_t* dst, int B, int sh, int sw, int dh, int dw, int C, const int* xi, const int* xa,
                                  const int* yi, const int* ya, hipStream_t s);
struct FuseParams {
    const float* disp_pp;      // [B,H,W]
    const uint8_t* road; const uint8_t* fence; const uint8_t* frames;
    const CamDev* cams;        // device [B]
    int B, H, W, cap;
    float* dense;              // nullable [B,H,W,3]
    float* road_xyz; uint8_t* road_rgb; int32_t* n_road;
    float* fence_xyz; uint8_t* fence_rgb; int32_t* n_fence;
    int32_t* blk_counts;       // scratch [B][nblk][2]
    int32_t* blk_offsets;      // scratch [B][nblk][2]
    unsigned sw;               // Switch bits of the handle
    // one-pass form (fuse.hip fuse_onepass_kernel): post-processing folded in, decoupled look-back compaction
    const float* disp_raw;     // nullable [B,2,H,W]: the net's raw pair; disp_pp is then COMPUTED here (post_processing) and written
    float* pp_out;             // [B,H,W], written when disp_raw is given
    unsigned long long* lb_state;   // scratch [B][nblk1024]: look-back words (epoch | flag | road count | fence count)
    int32_t* lb_ticket;        // scratch [B]: arrival tickets (zero between launches)
    unsigned epoch;            // 1..1023, different from the previous launch on this scratch (0 = freshly zeroed scratch)
};
size_t fuse_onepass_scratch_bytes(int B, int H, int W);
bool fuse_onepass_eligible(const FuseParams& p);
size_t fuse_scratch_bytes(int B, int H, int W);
hipError_t launch_fuse(const FuseParams& p, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// road-width tail (pcl.hip; -ffp-contract=off)
// ---------------------------------------------------------------------------------------------
struct RwParamsDev {
    double depth, z_cut, mad_y, mad_x, plane_thr, sor_ratio, ror_r, window, depth_offset;
    int sor_k, ror_n, use_o3d;
};
struct RwResultDev {   // layout == sd_rw_result
    double width; float x_left, x_right; float left_pt[3], right_pt[3];
    int32_t found, n_road, n_zcut, n_mad_y, n_mad_x, n_plane, n_sor, n_ror; double plane[4];
};
struct CloudView { const float* xyz; const uint8_t* rgb; const int32_t* n; };   // per-frame stride = cap points
struct CloudOut { float* xyz; uint8_t* rgb; int32_t* n; };

enum FilterKind { F_LT_NEG = 0 /* coord < -t */, F_ABS_LT = 1 /* |coord| < t */ };
// cscratch (nullable, cmp_scratch_bytes(B)): with it and in != out the ordered compaction runs on 64 workgroups per frame
size_t cmp_scratch_bytes(int B);
hipError_t launch_filter_coord(CloudView in, CloudOut out, int B, int cap, int kind, int axis, double t, void* cscratch, hipStream_t s);
hipError_t launch_mad_filter(CloudView in, CloudOut out, int B, int cap, int axis, double thr, float* stats, void* cscratch, hipStream_t s);
hipError_t launch_plane_filter(CloudView in, CloudOut out, int B, int cap, int axis, double thr, double* coeff, void* cscratch, hipStream_t s);
// fence chain
struct F2fResultDev {   // layout == sd_f2f_result
    double dist; double left_pt[3], right_pt[3]; double plane_left[4], plane_right[4];
    int32_t counts[7];   // n_fence, after MAD(y), after |z| threshold, left, right, left final, right final
    int32_t ok;
};
hipError_t launch_extract_pcls(CloudView in, CloudOut outl, CloudOut outr, int B, int cap, int axis, float* mean_out, hipStream_t s);
hipError_t launch_f2f(const double* road_plane, const double* left_plane, const double* right_plane, int B, double depth,
                      const int32_t* cnt /* [7][B] */, F2fResultDev* out, hipStream_t s);
hipError_t launch_gather_planes(const RwResultDev* res, int B, double* planes, hipStream_t s);
hipError_t launch_end_points(CloudView in, int B, int cap, double depth, double window, RwResultDev* res, hipStream_t s);
size_t o3d_scratch_bytes(int B, int cap);
hipError_t launch_sor(CloudView in, CloudOut out, int B, int cap, int k, double ratio, void* scratch, double* mean_out, void* cscratch,
                      hipStream_t s);
hipError_t launch_ror(CloudView in, CloudOut out, int B, int cap, int nb, double radius, void* scratch, void* cscratch, hipStream_t s);
// writes the count fields of the per-frame records from device-side counters
hipError_t launch_record_counts(RwResultDev* res, int B, const int32_t* n_road, const int32_t* n_zcut, const int32_t* n_mad_y,
                                const int32_t* n_mad_x, const int32_t* n_plane, const int32_t* n_sor, const int32_t* n_ror,
                                const double* plane, hipStream_t s);

}  // namespace sd

// Activation formats of the split precisions (a tensor [N,H,W,C] is stored as 16-bit planes; same bytes reserved as f32):
//   bf16 x 2 planes (SD_PREC_BF16X2; 16 mantissa bits, f32 range): hi = RNE(v), lo = RNE(v - hi), the lo plane following the hi
//     plane at a fixed element offset (plane stride = Nmax*H*W*C, Nmax = images of a full chunk).  Weights are split the same
//     way; a product is THREE MFMA products  x_hi*w_hi + x_hi*w_lo + x_lo*w_hi  (dropped term ~2^-18).
//   fp16 x 1 plane (the 2-product layers of a precision plan; |v| < 65504, the MFMA honours fp16 subnormals): the activation is
//     rounded ONCE to fp16 (11 bits) by the epilogue that produces it and only the hi plane exists; the WEIGHTS are split into two
//     fp16 planes (w_hi = RNE(w), w_lo = RNE(w - w_hi): 22 bits); a product is TWO MFMA products  x*w_hi + x*w_lo.  The only
//     error is the 2^-12 relative rounding of the activation -- the same size as rounding the weights instead (round 1's form
//     of the scheme: fp16 x 2 activation planes times ONE fp16 weight plane), but every activation tensor is half the bytes in
//     HBM, L2 and the LDS-DMA, which is what the HBM-bound layers (ResNet 1x1 layers, the full-resolution decoder) run against.
//   fp16 x 2 planes (hi as above + lo = RNE(v - hi)): for the few layers whose input tensor is precision-critical (it also feeds a
//     score head) the plan keeps 22 bits of the activation and drops the weight's low part instead: TWO products x_hi*w_hi + x_lo*w_hi
//     (conv_direct X2).  Any other fp16 layer reads the hi plane of such a tensor as if it were the one-plane format.
//   bf16 x 3 planes (SD_PREC_BF16X3; the fp32-grade split engine): hi = RNE(v), mid = RNE(v - hi), lo = v - hi - mid -- an f32 value is
//     the EXACT sum of three bf16 values (8 + 8 + 8 significand bits, the same exponent range), so nothing is rounded when a tensor or a
//     weight is stored; planes at element offsets 0, plane, 2 * plane.  A product is SIX MFMA products
//        x_hi*w_hi + x_hi*w_mid + x_mid*w_hi + x_mid*w_mid + x_hi*w_lo + x_lo*w_hi
//     (every bf16 x bf16 product is exact in the f32 accumulator); the three dropped terms x_mid*w_lo, x_lo*w_mid, x_lo*w_lo are below
//     2^-23 |x w| together, the size of ONE f32 rounding of the product, which an f32 FMA chain commits on every accumulation anyway.
//   fp16 hi + SCALED lo ("HS"; SD_PREC_F16X2, round 5: an fp32-grade engine on THREE MFMA products): hi = RNE_fp16(clamp(v)),
//     lo = RNE_fp16((v - hi) * 2^11) -- the residual is at most half an fp16 ulp of hi, so the scaled lo plane has the magnitude of hi / 2
//     and stays in fp16's NORMAL range wherever hi does: v is carried to 22 significand bits for every |v| in [1.2e-4, 65504] (29 binades:
//     no per-tensor calibration), 4 bytes per element.  Weights: w' = w * 2^k (k per layer, chosen at load time so that max |w'| lies in
//     [2^12, 2^13); the epilogue multiplies the accumulator by 2^-k), planes w_hi = RNE(w'), w_lo = RNE(w' - w_hi) (unscaled: w' is large).
//     A product is THREE fp16 MFMA products
//        x_hi * w_hi  +  x_hi * w_lo  +  x_lo * (w_hi * 2^-11)
//     the third weight operand being formed IN REGISTERS from the w_hi fragment (one v_pk_mul_f16 per register, exact: a power of two;
//     VALU beside 8-pass MFMAs is free, profiles/r05_probe_mfma_valu_overlap2.txt).  Dropped: x_lo * w_lo < 2^-24 |x w| and the 2^-23
//     representation error of each operand -- the size of ONE f32 rounding of the product; the accumulation is the same f32 chain as every
//     other engine's.  The MFMA honours fp16 subnormals (scripts/probe_mfma_f16_denorm.hip).
// The MFMA operands of the conv engine are read straight from the planes (16-byte runs of 8 channels), nothing is converted at
// load time.  The F16 template argument of the helpers selects the format; their `l` argument is ignored / zero for fp16.
#pragma once
#include <hip/hip_runtime.h>

namespace sd {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16lo_to_f(unsigned w) { return __uint_as_float(w << 16); }          // element 0 of a pair
__device__ __forceinline__ float bf16hi_to_f(unsigned w) { return __uint_as_float(w & 0xffff0000u); }  // element 1 of a pair

// 4 values from (hi pair-words, lo pair-words): v = hi + lo (exact in f32)
__device__ __forceinline__ f32x4_t recon4(uint2 h, uint2 l) {
    f32x4_t v;
    v[0] = bf16lo_to_f(h.x) + bf16lo_to_f(l.x);
    v[1] = bf16hi_to_f(h.x) + bf16hi_to_f(l.x);
    v[2] = bf16lo_to_f(h.y) + bf16lo_to_f(l.y);
    v[3] = bf16hi_to_f(h.y) + bf16hi_to_f(l.y);
    return v;
}
__device__ __forceinline__ f32x2_t recon2(unsigned h, unsigned l) {
    f32x2_t v;
    v[0] = bf16lo_to_f(h) + bf16lo_to_f(l);
    v[1] = bf16hi_to_f(h) + bf16hi_to_f(l);
    return v;
}
__device__ __forceinline__ void split2(f32x2_t v, unsigned& h, unsigned& l) {
    const bf16x2_t hb = __builtin_convertvector(v, bf16x2_t);                 // v_cvt_pk_bf16_f32, round to nearest even
    const f32x2_t r = v - __builtin_convertvector(hb, f32x2_t);               // exact
    const bf16x2_t lb = __builtin_convertvector(r, bf16x2_t);
    h = __builtin_bit_cast(unsigned, hb);
    l = __builtin_bit_cast(unsigned, lb);
}
__device__ __forceinline__ void split4(f32x4_t v, uint2& h, uint2& l) {
    split2(f32x2_t{v[0], v[1]}, h.x, l.x);
    split2(f32x2_t{v[2], v[3]}, h.y, l.y);
}

// bf16 x 3: exact three-way split and its inverse (both exact for finite v above the bf16 subnormal range)
__device__ __forceinline__ void split2_x3(f32x2_t v, unsigned& h, unsigned& m, unsigned& l) {
    const bf16x2_t hb = __builtin_convertvector(v, bf16x2_t);
    const f32x2_t r1 = v - __builtin_convertvector(hb, f32x2_t);              // exact (<= 16 significant bits)
    const bf16x2_t mb = __builtin_convertvector(r1, bf16x2_t);
    const f32x2_t r2 = r1 - __builtin_convertvector(mb, f32x2_t);             // exact (<= 8 significant bits)
    const bf16x2_t lb = __builtin_convertvector(r2, bf16x2_t);                // exact
    h = __builtin_bit_cast(unsigned, hb);
    m = __builtin_bit_cast(unsigned, mb);
    l = __builtin_bit_cast(unsigned, lb);
}
__device__ __forceinline__ void split4_x3(f32x4_t v, uint2& h, uint2& m, uint2& l) {
    split2_x3(f32x2_t{v[0], v[1]}, h.x, m.x, l.x);
    split2_x3(f32x2_t{v[2], v[3]}, h.y, m.y, l.y);
}
__device__ __forceinline__ f32x2_t recon2_x3(unsigned h, unsigned m, unsigned l) {
    f32x2_t v;
    v[0] = (bf16lo_to_f(h) + bf16lo_to_f(m)) + bf16lo_to_f(l);
    v[1] = (bf16hi_to_f(h) + bf16hi_to_f(m)) + bf16hi_to_f(l);
    return v;
}
__device__ __forceinline__ f32x4_t recon4_x3(uint2 h, uint2 m, uint2 l) {
    const f32x2_t a = recon2_x3(h.x, m.x, l.x), b = recon2_x3(h.y, m.y, l.y);
    return f32x4_t{a[0], a[1], b[0], b[1]};
}

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4s_t __attribute__((ext_vector_type(4)));

template <bool F16> __device__ __forceinline__ f32x2_t recon2_t(unsigned h, unsigned l) {
    if constexpr (F16) {
        return __builtin_convertvector(__builtin_bit_cast(f16x2_t, h), f32x2_t);      // one plane: the lo argument is not read
    } else {
        return recon2(h, l);
    }
}
template <bool F16> __device__ __forceinline__ f32x4_t recon4_t(uint2 h, uint2 l) {
    const f32x2_t a = recon2_t<F16>(h.x, l.x), b = recon2_t<F16>(h.y, l.y);
    return f32x4_t{a[0], a[1], b[0], b[1]};
}
// `sat` (the conv epilogues): the device counter of the values the fp16 formats had to clamp (or that were NaN) -- the layer's output
// left the fp16 range, i.e. the precision plan does not fit these weights; the host reads it through sd_saturation_count (a silent
// clamp must not pass for a result).  A compare and a never-taken branch per value pair: a per-lane count carried through the
// epilogue instead costs the register-capped kernels (two workgroups per CU at 128 VGPRs) hundreds of spilled registers.
typedef unsigned long long* sat_ptr_t;
__device__ __forceinline__ void sat_check(f32x2_t c, f32x2_t v, sat_ptr_t sat) {
    if (__builtin_expect((c[0] != v[0]) | (c[1] != v[1]), 0))
        if (sat) atomicAdd(sat, (unsigned long long)((c[0] != v[0] ? 1 : 0) + (c[1] != v[1] ? 1 : 0)));
}
template <bool F16> __device__ __forceinline__ void split2_t(f32x2_t v, unsigned& h, unsigned& l, sat_ptr_t sat) {
    if constexpr (F16) {
        // saturate at the largest finite fp16 (v_med3_f32): an activation beyond 65504 must not become inf and poison the layers
        // behind it (NaN passes through); then round to nearest even.  There is no lo plane.
        const f32x2_t c = {__builtin_amdgcn_fmed3f(v[0], -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v[1], -65504.f, 65504.f)};
        sat_check(c, v, sat);
        const f16x2_t hb = __builtin_convertvector(c, f16x2_t);
        h = __builtin_bit_cast(unsigned, hb);
        l = 0u;
    } else {
        split2(v, h, l);
    }
}
template <bool F16> __device__ __forceinline__ void split2_t(f32x2_t v, unsigned& h, unsigned& l) {
    split2_t<F16>(v, h, l, (sat_ptr_t) nullptr);
}
template <bool F16> __device__ __forceinline__ void split4_t(f32x4_t v, uint2& h, uint2& l, sat_ptr_t sat) {
    split2_t<F16>(f32x2_t{v[0], v[1]}, h.x, l.x, sat);
    split2_t<F16>(f32x2_t{v[2], v[3]}, h.y, l.y, sat);
}
template <bool F16> __device__ __forceinline__ void split4_t(f32x4_t v, uint2& h, uint2& l) {
    split4_t<F16>(v, h, l, (sat_ptr_t) nullptr);
}
// fp16 hi + SCALED lo (HS): split and its inverse
__device__ __forceinline__ void split2_hs(f32x2_t v, unsigned& h, unsigned& l, sat_ptr_t sat) {
    const f32x2_t c = {__builtin_amdgcn_fmed3f(v[0], -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v[1], -65504.f, 65504.f)};
    sat_check(c, v, sat);
    const f16x2_t hb = __builtin_convertvector(c, f16x2_t);
    const f32x2_t r = (c - __builtin_convertvector(hb, f32x2_t)) * 2048.f;        // exact residual, exact scaling
    const f16x2_t lb = __builtin_convertvector(r, f16x2_t);
    h = __builtin_bit_cast(unsigned, hb);
    l = __builtin_bit_cast(unsigned, lb);
}
__device__ __forceinline__ void split4_hs(f32x4_t v, uint2& h, uint2& l, sat_ptr_t sat) {
    split2_hs(f32x2_t{v[0], v[1]}, h.x, l.x, sat);
    split2_hs(f32x2_t{v[2], v[3]}, h.y, l.y, sat);
}
__device__ __forceinline__ f32x2_t recon2_hs(unsigned h, unsigned l) {
    return __builtin_convertvector(__builtin_bit_cast(f16x2_t, h), f32x2_t) + __builtin_convertvector(__builtin_bit_cast(f16x2_t, l), f32x2_t) * (1.f / 2048.f);
}
__device__ __forceinline__ f32x4_t recon4_hs(uint2 h, uint2 l) {
    const f32x2_t a = recon2_hs(h.x, l.x), b = recon2_hs(h.y, l.y);
    return f32x4_t{a[0], a[1], b[0], b[1]};
}
// the weight operand of the x_lo product: an fp16 w_hi fragment times 2^-11 (exact; w' >= 2^-3 stays normal, smaller weights are below
// 2^-15 of the layer's largest)
__device__ __forceinline__ u32x4s_t hs_wscaled(u32x4s_t w) {
    const f16x8_t s = __builtin_bit_cast(f16x8_t, w) * (_Float16)(1.0 / 2048.0);
    return __builtin_bit_cast(u32x4s_t, s);
}
// output format of an epilogue: 0 = bf16 hi + lo, 1 = ONE fp16 plane, 3 = fp16 hi + SCALED lo (HS), 2 = fp16 hi + lo (the hi plane is bit for bit the one of format
// 1, so every fp16 layer can read such a tensor; the lo plane serves the layers that multiply x_hi and x_lo by ONE weight plane)
template <int FMT> __device__ __forceinline__ void split2_fmt(f32x2_t v, unsigned& h, unsigned& l, sat_ptr_t sat) {
    if constexpr (FMT == 3) {          // fp16 hi + scaled lo (SD_PREC_F16X2)
        split2_hs(v, h, l, sat);
    } else if constexpr (FMT == 2) {
        const f32x2_t c = {__builtin_amdgcn_fmed3f(v[0], -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v[1], -65504.f, 65504.f)};
        sat_check(c, v, sat);
        const f16x2_t hb = __builtin_convertvector(c, f16x2_t);
        const f32x2_t r = c - __builtin_convertvector(hb, f32x2_t);               // exact
        const f16x2_t lb = __builtin_convertvector(r, f16x2_t);
        h = __builtin_bit_cast(unsigned, hb);
        l = __builtin_bit_cast(unsigned, lb);
    } else {
        split2_t<FMT == 1>(v, h, l, sat);
    }
}
template <int FMT> __device__ __forceinline__ void split4_fmt(f32x4_t v, uint2& h, uint2& l, sat_ptr_t sat) {
    split2_fmt<FMT>(f32x2_t{v[0], v[1]}, h.x, l.x, sat);
    split2_fmt<FMT>(f32x2_t{v[2], v[3]}, h.y, l.y, sat);
}
// fp16 hi + lo -> f32
__device__ __forceinline__ f32x2_t recon2_f16x2(unsigned h, unsigned l) {
    return __builtin_convertvector(__builtin_bit_cast(f16x2_t, h), f32x2_t) + __builtin_convertvector(__builtin_bit_cast(f16x2_t, l), f32x2_t);
}
// one 32x32x16 MFMA on raw 16-byte fragments of the selected element type
template <bool F16> __device__ __forceinline__ f32x16_t mfma_frag(u32x4s_t a, u32x4s_t b, f32x16_t c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// one 16x16x32 MFMA (A: 16 rows x 32 k, B: 32 k x 16 columns; lane l supplies row/column l & 15, k = 8 (l >> 4) .. + 7 and
// holds D[4 (l >> 4) + r][l & 15]) on raw 16-byte fragments
template <bool F16> __device__ __forceinline__ f32x4_t mfma_frag16(u32x4s_t a, u32x4s_t b, f32x4_t c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

}  // namespace sd

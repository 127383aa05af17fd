// "Split planes" activation format of the split-bf16 precision (SD_PREC_BF16X2): a tensor [N,H,W,C] is stored as two
// bf16 planes, hi = RNE_bf16(v) and lo = RNE_bf16(v - hi), the lo plane following the hi plane at a fixed element
// offset (plane stride = Nmax*H*W*C, Nmax = images of a full chunk).  Same bytes as f32; the MFMA operands of the conv
// engine are read straight from the planes (16-byte runs of 8 channels), nothing is split at load time.
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

}  // namespace sd
